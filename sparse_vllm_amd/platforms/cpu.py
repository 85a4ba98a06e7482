"""CPU platform: import / unit-test paths only (the reference's CPU platform never runs inference either).
Behaviour pinned by tests/golden/operator_registry.json `platforms.cpu`."""

from __future__ import annotations

import dataclasses

import torch

from .interface import Platform, PlatformEnum

_HOST = torch.device("cpu")


class CpuPlatform(Platform):
    name = "cpu"
    device_type = _HOST.type
    enum = PlatformEnum.CPU

    check_available = staticmethod(lambda: True)            # the host is always there
    set_device = staticmethod(lambda device=None: None)     # one host, nothing to select
    get_device = staticmethod(lambda local_rank=0: _HOST)

    def get_device_caps(self, device_index: int = 0):
        # torch's CPU kernels do bf16 arithmetic; nothing else a GPU provider asks for exists here
        return dataclasses.replace(super().get_device_caps(device_index), supports_bfloat16=True)
