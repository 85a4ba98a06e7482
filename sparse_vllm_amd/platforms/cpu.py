"""CPU platform: import / unit-test paths only, as in the reference (platforms/cpu.py:8-44) — no inference."""

from __future__ import annotations

import torch

from .interface import DeviceCaps, Platform, PlatformEnum


class CpuPlatform(Platform):
    name = "cpu"
    device_type = "cpu"
    enum = PlatformEnum.CPU

    def check_available(self) -> bool:
        return True

    def get_device(self, local_rank: int = 0) -> torch.device:
        return torch.device("cpu")

    def set_device(self, device) -> None:
        return None

    def get_device_caps(self, device_index: int = 0) -> DeviceCaps:
        return DeviceCaps(platform=self.enum, device_type=self.device_type, device_index=int(device_index),
                          device_name="cpu", supports_bfloat16=True)
