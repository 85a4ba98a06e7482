"""Operator provider registry — the reference's own pattern (operators/registry.py:120-194), so that a provider class
written for the reference registers here unchanged and the other way round:

    @REGISTRY.register
    class Provider:
        name = "..."; priority = N
        @classmethod
        def supports(cls, spec, caps: DeviceCaps) -> SupportResult: ...

`OpRegistry(family)` keeps provider CLASSES by `name` (a second registration of a name is a ValueError),
`OpResolver(registry).resolve(spec, caps, **provider_kwargs)` asks every class, orders the supporting ones by
(-priority, name), instantiates the first with `provider_kwargs` and returns `ResolvedProvider(provider, rejected)`
where `rejected` lists (name, reason) of the classes that declined; no supporting class is a RuntimeError that names the
family, the spec, the device and every reason.  Bound providers are tracked per family for `operator_runtime_stats()`
(operators/registry.py:17-72).
"""

from __future__ import annotations

import re
import weakref
from dataclasses import dataclass
from typing import Generic, Protocol, TypeVar

from ..platforms.interface import DeviceCaps, PlatformEnum  # noqa: F401  (re-exported: earlier rounds imported them from here)

SpecT = TypeVar("SpecT")
ProviderT = TypeVar("ProviderT", bound="OperatorProvider")

_OPERATOR_BINDINGS: dict[str, "weakref.WeakSet[object]"] = {}


def record_operator_binding(operator_type: str, provider: object) -> None:
    _OPERATOR_BINDINGS.setdefault(operator_type, weakref.WeakSet()).add(provider)


def _implementation_name(provider: object) -> str:
    return getattr(provider, "implementation_name", None) or getattr(provider, "name", None) or provider.provider_name


def operator_runtime_stats() -> dict[str, list[dict[str, object]]]:
    """Per operator family, per implementation: how many live providers are bound and the kernel-path / fallback counters
    of those that expose `runtime_kernel_stats()` (summed)."""
    out: dict[str, list[dict[str, object]]] = {}
    for family in sorted(_OPERATOR_BINDINGS):
        by_impl: dict[str, list[object]] = {}
        for provider in _OPERATOR_BINDINGS[family]:
            by_impl.setdefault(_implementation_name(provider), []).append(provider)
        entries = []
        for impl in sorted(by_impl):
            paths: dict[str, dict[str, int]] = {}
            fallbacks: dict[str, int] = {}
            instrumented = 0
            for provider in by_impl[impl]:
                fn = getattr(provider, "runtime_kernel_stats", None)
                if not callable(fn):
                    continue
                instrumented += 1
                stats = fn()
                for path, counts in stats.get("kernel_paths", {}).items():
                    agg = paths.setdefault(str(path), {})
                    for key, n in counts.items():
                        agg[str(key)] = int(agg.get(str(key), 0)) + int(n)
                for reason, n in stats.get("fallback_reasons", {}).items():
                    fallbacks[str(reason)] = int(fallbacks.get(str(reason), 0)) + int(n)
            entries.append({"implementation": impl, "bound_provider_count": len(by_impl[impl]),
                            "instrumented_provider_count": instrumented,
                            "kernel_paths": {p: dict(sorted(c.items())) for p, c in sorted(paths.items())},
                            "fallback_reasons": dict(sorted(fallbacks.items()))})
        if entries:
            out[family] = entries
    return out


def runtime_version_at_least(version: str | None, minimum: tuple[int, int]) -> bool:
    """operators/registry.py:107-117: "major.minor..." >= minimum; None / unparsable -> False."""
    if version is None:
        return False
    m = re.match(r"^\s*(\d+)\.(\d+)", str(version))
    return m is not None and (int(m.group(1)), int(m.group(2))) >= tuple(minimum)


@dataclass(frozen=True)
class SupportResult:
    supported: bool
    reason: str

    @classmethod
    def yes(cls, reason: str = "supported") -> "SupportResult":
        return cls(True, reason)

    @classmethod
    def no(cls, reason: str) -> "SupportResult":
        return cls(False, reason)


class OperatorProvider(Protocol[SpecT]):
    name: str
    priority: int

    @classmethod
    def supports(cls, spec: SpecT, caps: DeviceCaps) -> SupportResult: ...


class OpRegistry(Generic[SpecT, ProviderT]):
    def __init__(self, family: str) -> None:
        self.family = str(family)
        self._providers: dict[str, type] = {}

    def register(self, provider: type) -> type:
        name = str(provider.name)
        if name in self._providers:
            raise ValueError(f"Provider {name!r} is already registered for {self.family!r}.")
        self._providers[name] = provider
        return provider

    @property
    def providers(self) -> tuple[type, ...]:
        return tuple(self._providers.values())


@dataclass(frozen=True)
class ResolvedProvider(Generic[ProviderT]):
    provider: ProviderT
    rejected: tuple[tuple[str, str], ...]


class OpResolver(Generic[SpecT, ProviderT]):
    def __init__(self, registry: OpRegistry) -> None:
        self.registry = registry

    def resolve(self, spec, caps: DeviceCaps, **provider_kwargs) -> ResolvedProvider:
        accepted: list[type] = []
        rejected: list[tuple[str, str]] = []
        for cls in self.registry.providers:
            verdict = cls.supports(spec, caps)
            if verdict.supported:
                accepted.append(cls)
            else:
                rejected.append((cls.name, verdict.reason))
        if not accepted:
            details = "; ".join(f"{name}: {reason}" for name, reason in rejected)
            raise RuntimeError(f"No {self.registry.family} provider supports spec={spec!r} on "
                               f"device={caps.device_name!r}: {details or 'no providers registered'}.")
        accepted.sort(key=lambda cls: (-int(cls.priority), cls.name))
        chosen = accepted[0](**provider_kwargs)
        record_operator_binding(self.registry.family, chosen)
        return ResolvedProvider(chosen, tuple(rejected))
