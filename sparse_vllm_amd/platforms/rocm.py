"""A *working* ROCm platform (the reference's RocmPlatform refuses inference,
platforms/rocm.py:182-201).  Mirrors the Platform interface of platforms/interface.py:20-174
for what the hot path needs: device, memory info, distributed backend name, device caps."""

from __future__ import annotations

import torch

from ..operators.registry import DeviceCaps, PlatformEnum


class RocmPlatform:
    name = "rocm"
    platform_enum = PlatformEnum.ROCM

    def get_device(self, rank: int = 0) -> torch.device:
        return torch.device(f"cuda:{int(rank) % max(1, torch.cuda.device_count())}")

    def get_distributed_backend(self) -> str:
        return "nccl"        # == RCCL on PyTorch-ROCm (xGMI inside a node)

    def mem_get_info(self, device=None) -> tuple[int, int]:
        return torch.cuda.mem_get_info(device)

    def is_stream_capturing(self) -> bool:
        return bool(torch.cuda.is_available() and torch.cuda.is_current_stream_capturing())

    def validate_inference(self) -> None:
        if not torch.cuda.is_available():
            raise RuntimeError("ROCm inference needs a visible AMD GPU (torch.cuda.is_available() is False).")
        from .. import _lib
        _lib.load()     # fail loudly when the HIP extension is missing

    def device_caps(self, device=None) -> DeviceCaps:
        if not torch.cuda.is_available():
            return DeviceCaps(platform=PlatformEnum.ROCM, arch="gfx950", num_cus=256, lds_bytes=160 * 1024,
                              hbm_bytes=288 * 2 ** 30)
        p = torch.cuda.get_device_properties(device or 0)
        arch = getattr(p, "gcnArchName", "").split(":")[0]
        return DeviceCaps(platform=PlatformEnum.ROCM, arch=arch, num_cus=int(p.multi_processor_count),
                          lds_bytes=160 * 1024, hbm_bytes=int(p.total_memory))


current_platform = RocmPlatform()
