"""A *working* ROCm platform.  The reference's `RocmPlatform` (platforms/rocm.py:8-27) detects ROCm and then refuses
inference ("ROCm inference is not supported yet ... until a ROCm platform/op backend is implemented"); this is that
backend for the sparse-attention path, with the method set of `CudaPlatform` (platforms/cuda.py:15-90) answered for
PyTorch-ROCm on MI355X."""

from __future__ import annotations

from functools import lru_cache

import torch

from .interface import AllocatorStats, DeviceCaps, Platform, PlatformEnum

MI355X_LDS_BYTES = 160 * 1024
MI355X_HBM_BYTES = 288 * 2 ** 30
MI355X_CUS = 256


class RocmPlatform(Platform):
    name = "rocm"
    device_type = "cuda"              # torch device strings stay "cuda:N" on PyTorch-ROCm
    enum = PlatformEnum.ROCM

    def check_available(self) -> bool:
        return bool(torch.cuda.is_available() and torch.version.hip is not None)

    def validate_environment(self) -> None:
        if not self.check_available():
            raise RuntimeError("ROCm platform was selected, but PyTorch is not running with HIP support.")

    def supports_inference(self) -> bool:
        return True

    def validate_inference(self) -> None:
        super().validate_inference()
        from .. import _lib
        _lib.load()                   # fail loudly when the HIP extension is missing: there is no other path

    def get_device(self, local_rank: int = 0) -> torch.device:
        return torch.device(self.device_type, int(local_rank) % max(1, torch.cuda.device_count()))

    def set_device(self, device) -> None:
        torch.cuda.set_device(device)

    def get_available_memory(self, device_id: int = 0) -> tuple[int, int]:
        return torch.cuda.mem_get_info(int(device_id))

    def get_allocator_stats(self, device: torch.device | None = None) -> AllocatorStats:
        stats = torch.cuda.memory_stats(device)
        return AllocatorStats(peak_allocated_bytes=int(stats.get("allocated_bytes.all.peak", 0)),
                              current_allocated_bytes=int(stats.get("allocated_bytes.all.current", 0)))

    def reset_peak_memory_stats(self, device: torch.device | None = None) -> None:
        torch.cuda.reset_peak_memory_stats(device)

    def empty_cache(self) -> None:
        torch.cuda.empty_cache()

    def synchronize(self) -> None:
        torch.cuda.synchronize()

    def is_stream_capturing(self) -> bool:
        return bool(torch.cuda.is_available() and torch.cuda.is_current_stream_capturing())

    def get_distributed_backend(self) -> str:
        return "nccl"                 # == RCCL on PyTorch-ROCm (xGMI inside a node)

    def barrier_device_ids(self, rank: int) -> list[int] | None:
        return [int(rank)]

    @lru_cache(maxsize=None)
    def get_device_caps(self, device_index: int = 0) -> DeviceCaps:
        device_index = int(device_index)
        common = dict(platform=self.enum, device_type=self.device_type, device_index=device_index,
                      runtime_version=torch.version.hip, supports_graph_capture=True, supports_torch_compile=False,
                      supports_triton=False,          # by design: hand-written HIP only
                      supports_pin_memory=True, supports_bfloat16=True, supports_native_fp8=True)
        if not torch.cuda.is_available():
            # build container (no GPU): the capabilities of the one target, so that provider resolution is testable
            return DeviceCaps(device_name="AMD Instinct MI355X", compute_capability=(9, 5), arch="gfx950",
                              num_cus=MI355X_CUS, lds_bytes=MI355X_LDS_BYTES, hbm_bytes=MI355X_HBM_BYTES, **common)
        p = torch.cuda.get_device_properties(device_index)
        major, minor = torch.cuda.get_device_capability(device_index)
        return DeviceCaps(device_name=str(torch.cuda.get_device_name(device_index)),
                          compute_capability=(int(major), int(minor)),
                          arch=str(getattr(p, "gcnArchName", "")).split(":")[0], num_cus=int(p.multi_processor_count),
                          lds_bytes=MI355X_LDS_BYTES, hbm_bytes=int(p.total_memory), **common)

    def get_default_attention_backend(self) -> str:
        return "hip"

    def get_dispatch_key(self) -> str:
        return "rocm"

    # ---- the names this build used before the interface was mirrored (kept: tools and tests call them)
    def mem_get_info(self, device=None) -> tuple[int, int]:
        return torch.cuda.mem_get_info(device)

    def device_caps(self, device=None) -> DeviceCaps:
        idx = 0 if device is None else (torch.device(device).index or 0)
        return self.get_device_caps(int(idx))
