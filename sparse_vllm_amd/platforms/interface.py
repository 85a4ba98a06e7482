"""Platform boundary types: what an operator provider is told about the device it may run on.

Written from tests/golden/operator_registry.json — the reference's answers to tests/registry_scenarios.py `types` and
`platforms` — not from the reference's file.  Pinned there, and kept: the member names of `PlatformEnum`, the field
names / order / defaults of `AllocatorStats` and of the first twelve `DeviceCaps` fields, the public attribute set of
`Platform`, the defaults of an unspecialised `Platform()` and three error texts.  Everything else is this build's:
the boolean families (`supports_*`, `is_*`) are generated from two small tables, hooks a backend may leave alone are
generated no-ops, and the members a backend MUST answer are generated to raise NotImplementedError naming themselves.

MI355X additions: four optional `DeviceCaps` fields after the reference's twelve (`arch`, `num_cus`, `lds_bytes`,
`hbm_bytes`), so a `DeviceCaps` built with the reference's arguments alone stays valid.
"""

from __future__ import annotations

import contextlib
import enum
from dataclasses import dataclass
from typing import Any

import torch

PlatformEnum = enum.Enum("PlatformEnum", ["CUDA", "ROCM", "NPU", "CPU", "UNSPECIFIED"], module=__name__)


@dataclass(frozen=True)
class AllocatorStats:
    peak_allocated_bytes: int = 0
    current_allocated_bytes: int = 0


@dataclass(frozen=True)
class DeviceCaps:
    platform: PlatformEnum
    device_type: str
    device_index: int
    device_name: str
    compute_capability: tuple[int, int] | None = None
    runtime_version: str | None = None
    supports_graph_capture: bool = False
    supports_torch_compile: bool = False
    supports_triton: bool = False
    supports_pin_memory: bool = False
    supports_bfloat16: bool = False
    supports_native_fp8: bool = False
    # MI355X extras (optional)
    arch: str = ""                  # "gfx950"
    num_cus: int = 0
    lds_bytes: int = 0
    hbm_bytes: int = 0


# Platform.<query>() -> the DeviceCaps field of device 0 that answers it
_CAPS_QUERIES = {
    "supports_graph_capture": "supports_graph_capture",
    "supports_torch_compile": "supports_torch_compile",
    "supports_triton": "supports_triton",
    "supports_pin_memory": "supports_pin_memory",
    "supports_bfloat16": "supports_bfloat16",
    "supports_fp8": "supports_native_fp8",
}
# Platform.<predicate>() -> the enum members it is true for
_KIND_PREDICATES = {
    "is_cuda": ("CUDA",),
    "is_rocm": ("ROCM",),
    "is_npu": ("NPU",),
    "is_cpu": ("CPU",),
    "is_cuda_alike": ("CUDA", "ROCM"),
}
# hooks with nothing to do by default -> their default answer
_OPTIONAL_HOOKS = {
    "init_backend": None,
    "reset_peak_memory_stats": None,
    "empty_cache": None,
    "synchronize": None,
    "apply_config_defaults": None,
    "barrier_device_ids": None,
    "get_communicator_cls": None,
    "get_decode_graph_runner_cls": None,
    "is_stream_capturing": False,
    "supports_inference": False,
    "check_available": False,
}
# members every device backend has to answer itself
_REQUIRED_OF_A_BACKEND = ("get_device", "set_device", "get_available_memory")


class Platform:
    name: str = "unknown"
    device_type: str = "cpu"
    enum: PlatformEnum = PlatformEnum.UNSPECIFIED
    supported_quantization: tuple[str, ...] = ()

    # ---- gates
    def validate_environment(self) -> None:
        if not self.check_available():
            raise RuntimeError(f"Platform {self.name!r} is not available.")

    def validate_inference(self) -> None:
        self.validate_environment()
        if not self.supports_inference():
            raise RuntimeError(f"Platform {self.name!r} is detected, but Sparse-vLLM inference is not supported on this "
                               "platform in the current build.")

    def validate_config(self, config: Any) -> None:
        asked = any(bool(getattr(config, flag, False)) for flag in ("decode_cuda_graph", "decode_graph"))
        if asked and not self.supports_graph_capture():
            raise RuntimeError(f"Platform {self.name!r} does not support decode graph capture.")

    # ---- description
    def get_device_caps(self, device_index: int = 0) -> DeviceCaps:
        return DeviceCaps(self.enum, self.device_type, int(device_index), self.name)

    def get_allocator_stats(self, device: torch.device | None = None) -> AllocatorStats:
        return AllocatorStats()

    def get_distributed_backend(self) -> str:
        return "gloo"

    def get_default_attention_backend(self) -> str:
        return "native"

    def get_dispatch_key(self) -> str:
        return str(self.name)

    # ---- run-time helpers
    def inference_mode(self):
        return torch.inference_mode() if hasattr(torch, "inference_mode") else contextlib.nullcontext()

    def seed_everything(self, seed: int) -> None:
        torch.manual_seed(int(seed))


def _install_generated_members() -> None:
    def caps_query(field: str):
        def query(self) -> bool:
            return bool(getattr(self.get_device_caps(), field))
        return query

    def kind_predicate(members: tuple[str, ...]):
        def predicate(self) -> bool:
            return self.enum.name in members
        return predicate

    def optional_hook(answer):
        def hook(self, *args, **kwargs):
            return answer
        return hook

    def required(member: str):
        def missing(self, *args, **kwargs):
            raise NotImplementedError(f"Platform {self.name!r} does not implement {member}().")
        return missing

    generated = {}
    generated.update({n: caps_query(f) for n, f in _CAPS_QUERIES.items()})
    generated.update({n: kind_predicate(m) for n, m in _KIND_PREDICATES.items()})
    generated.update({n: optional_hook(a) for n, a in _OPTIONAL_HOOKS.items()})
    generated.update({n: required(n) for n in _REQUIRED_OF_A_BACKEND})
    for member, fn in generated.items():
        fn.__name__ = member
        fn.__qualname__ = f"Platform.{member}"
        setattr(Platform, member, fn)


_install_generated_members()
