"""Operator provider registry behind the reference's registration pattern, written from the behaviour
table in tests/golden/operator_registry.json (what the reference answered to tests/registry_scenarios.py), not from
the reference's file.  The contract those scenarios pin, and nothing more:

* a provider is a CLASS with `name`, `priority` and a classmethod `supports(spec, caps) -> SupportResult`;
* `OpRegistry(family).register(cls)` is usable as a decorator, keeps insertion order in `.providers`, and refuses a
  second class of the same `name` with ValueError("Provider 'x' is already registered for 'family'.");
* `OpResolver(registry).resolve(spec, caps, **ctor_kwargs)` instantiates the accepting class of highest priority (ties:
  smallest name) and reports every declining class as `(name, reason)` in registration order; when nobody accepts it
  raises RuntimeError("No <family> provider supports spec=... on device='...': a: why; b: why.") — or
  "...: no providers registered." for an empty family;
* every resolved instance is remembered weakly per family so that `operator_runtime_stats()` can sum the counters of the
  live ones.
"""

from __future__ import annotations

import weakref
from collections import Counter, defaultdict
from dataclasses import dataclass
from typing import Generic, Protocol, TypeVar

from ..platforms.interface import DeviceCaps, PlatformEnum  # noqa: F401  (re-exported: earlier rounds imported them from here)

SpecT = TypeVar("SpecT")
ProviderT = TypeVar("ProviderT", bound="OperatorProvider")

# family -> weak set of bound provider instances (a provider that died drops out of the statistics by itself)
_OPERATOR_BINDINGS: dict[str, "weakref.WeakSet[object]"] = {}


def record_operator_binding(operator_type: str, provider: object) -> None:
    live = _OPERATOR_BINDINGS.get(operator_type)
    if live is None:
        live = _OPERATOR_BINDINGS[operator_type] = weakref.WeakSet()
    live.add(provider)


class _ImplTally:
    """Counters of all live providers that share one implementation label."""

    __slots__ = ("bound", "instrumented", "paths", "fallbacks")

    def __init__(self) -> None:
        self.bound = 0
        self.instrumented = 0
        self.paths: defaultdict[str, Counter] = defaultdict(Counter)
        self.fallbacks: Counter = Counter()

    def absorb(self, provider: object) -> None:
        self.bound += 1
        probe = getattr(provider, "runtime_kernel_stats", None)
        if not callable(probe):
            return
        self.instrumented += 1
        report = probe() or {}
        for path, counters in (report.get("kernel_paths") or {}).items():
            self.paths[str(path)].update({str(k): int(v) for k, v in counters.items()})
        self.fallbacks.update({str(k): int(v) for k, v in (report.get("fallback_reasons") or {}).items()})

    def entry(self, label: str) -> dict[str, object]:
        # Counter.update() keeps zero counts, which the table shows ("noncontiguous:output": 0)
        return {"implementation": label, "bound_provider_count": self.bound,
                "instrumented_provider_count": self.instrumented,
                "kernel_paths": {p: {k: self.paths[p][k] for k in sorted(self.paths[p])} for p in sorted(self.paths)},
                "fallback_reasons": {k: self.fallbacks[k] for k in sorted(self.fallbacks)}}


def _label_of(provider: object) -> str:
    for attr in ("implementation_name", "name", "provider_name"):
        label = getattr(provider, attr, None)
        if label:
            return str(label)
    return type(provider).__name__


def operator_runtime_stats() -> dict[str, list[dict[str, object]]]:
    """{family: [one entry per implementation label, sorted]} over the providers that are still alive; families whose
    providers are all gone are left out."""
    tallies: dict[tuple[str, str], _ImplTally] = {}
    for family, live in _OPERATOR_BINDINGS.items():
        for provider in tuple(live):
            key = (family, _label_of(provider))
            tally = tallies.get(key)
            if tally is None:
                tally = tallies[key] = _ImplTally()
            tally.absorb(provider)
    report: dict[str, list[dict[str, object]]] = {}
    for family, label in sorted(tallies):
        report.setdefault(family, []).append(tallies[(family, label)].entry(label))
    return report


def runtime_version_at_least(version: str | None, minimum: tuple[int, int]) -> bool:
    """True when `version` starts with "<major>.<minor>" (anything may follow the minor digits) and (major, minor) is
    not below `minimum`.  None, a bare major ("13") and non-numeric text are False."""
    if version is None:
        return False
    major, dot, rest = str(version).strip().partition(".")
    digits = rest[: len(rest) - len(rest.lstrip("0123456789"))]
    if not (dot and major.isdigit() and digits):
        return False
    want_major, want_minor = (int(x) for x in minimum)
    have = (int(major), int(digits))
    return have[0] > want_major or (have[0] == want_major and have[1] >= want_minor)


@dataclass(frozen=True)
class SupportResult:
    supported: bool
    reason: str

    @classmethod
    def yes(cls, reason: str = "supported") -> "SupportResult":
        return cls(supported=True, reason=reason)

    @classmethod
    def no(cls, reason: str) -> "SupportResult":
        return cls(supported=False, reason=reason)


class OperatorProvider(Protocol[SpecT]):
    name: str
    priority: int

    @classmethod
    def supports(cls, spec: SpecT, caps: DeviceCaps) -> SupportResult: ...


class OpRegistry(Generic[SpecT, ProviderT]):
    """The provider classes of one operator family, in registration order."""

    def __init__(self, family: str) -> None:
        self.family = str(family)
        self._providers: dict[str, type] = {}

    def register(self, provider: type) -> type:
        key = str(provider.name)
        if key in self._providers:            # also the same class twice
            raise ValueError(f"Provider {key!r} is already registered for {self.family!r}.")
        self._providers[key] = provider
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
        winner: type | None = None
        declined: list[tuple[str, str]] = []
        for candidate in self.registry.providers:
            answer = candidate.supports(spec, caps)
            if not answer.supported:
                declined.append((candidate.name, answer.reason))
            elif winner is None or self._rank(candidate) < self._rank(winner):
                winner = candidate
        if winner is None:
            why = "; ".join(f"{n}: {r}" for n, r in declined) if declined else "no providers registered"
            raise RuntimeError(f"No {self.registry.family} provider supports spec={spec!r} "
                               f"on device={caps.device_name!r}: {why}.")
        instance = winner(**provider_kwargs)
        record_operator_binding(self.registry.family, instance)
        return ResolvedProvider(provider=instance, rejected=tuple(declined))

    @staticmethod
    def _rank(provider: type) -> tuple[int, str]:
        """Smaller is better: higher priority first, then the smaller name."""
        return (-int(provider.priority), str(provider.name))
