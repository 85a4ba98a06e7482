"""Platform selection (platforms/__init__.py:67-112 of the reference): `SPARSEVLLM_PLATFORM` picks a builtin platform by
name, otherwise ROCm.  Unlike the reference this build has exactly one inference platform — MI355X under PyTorch-ROCm —
and `current_platform` is that platform also in the GPU-less build container, where it answers capability queries with
the target's figures (hipcc cross-compiles there; nothing runs)."""

from __future__ import annotations

import os

from .cpu import CpuPlatform
from .interface import AllocatorStats, DeviceCaps, Platform, PlatformEnum
from .rocm import RocmPlatform

_current_platform: Platform | None = None


def _resolve_platform() -> Platform:
    selected = os.getenv("SPARSEVLLM_PLATFORM", "").strip().lower()
    if selected == "cpu":
        return CpuPlatform()
    if selected in ("", "rocm"):
        return RocmPlatform()
    if selected == "cuda":
        raise RuntimeError("SPARSEVLLM_PLATFORM='cuda': this build targets AMD MI355X (ROCm) only.")
    raise RuntimeError(f"SPARSEVLLM_PLATFORM={selected!r} is not a platform of this build (rocm, cpu).")


def get_current_platform() -> Platform:
    global _current_platform
    if _current_platform is None:
        _current_platform = _resolve_platform()
    return _current_platform


def _set_current_platform_for_tests(platform: Platform | None) -> None:
    global _current_platform
    _current_platform = platform


def __getattr__(name: str):
    if name == "current_platform":
        return get_current_platform()
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


__all__ = ["AllocatorStats", "CpuPlatform", "DeviceCaps", "Platform", "PlatformEnum", "RocmPlatform", "current_platform",
           "get_current_platform"]
