"""Operator provider registry (mirror of operators/registry.py:120-194): providers declare
`name`, `priority`, `supports(spec, caps) -> SupportResult`; the resolver picks the highest
priority supporting provider for the current device capabilities."""

from __future__ import annotations

from dataclasses import dataclass
from enum import Enum


class PlatformEnum(str, Enum):
    CUDA = "cuda"
    ROCM = "rocm"
    CPU = "cpu"


@dataclass(frozen=True)
class DeviceCaps:
    platform: PlatformEnum
    arch: str = ""              # e.g. "gfx950"
    num_cus: int = 0
    lds_bytes: int = 0
    hbm_bytes: int = 0


@dataclass(frozen=True)
class SupportResult:
    ok: bool
    reason: str = ""

    @staticmethod
    def yes() -> "SupportResult":
        return SupportResult(True)

    @staticmethod
    def no(reason: str) -> "SupportResult":
        return SupportResult(False, reason)


class OpRegistry:
    def __init__(self, op_name: str):
        self.op_name = op_name
        self._providers: list[type] = []

    def register(self, cls):
        self._providers.append(cls)
        return cls

    @property
    def providers(self):
        return tuple(self._providers)


class OpResolver:
    def __init__(self, registry: OpRegistry):
        self.registry = registry

    def resolve(self, spec, caps: DeviceCaps):
        rejected = []
        for cls in sorted(self.registry.providers, key=lambda c: -int(getattr(c, "priority", 0))):
            provider = cls()
            res = provider.supports(spec, caps)
            if res.ok:
                return provider
            rejected.append(f"{cls.name}: {res.reason}")
        raise RuntimeError(f"No provider for op {self.registry.op_name!r} on {caps}: " + "; ".join(rejected))
