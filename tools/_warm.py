"""Shared by the kbench tools: sustained GPU work before the first timed configuration.  Clocks ramp over tens of
milliseconds; a handful of warm-up launches is not enough and the first configuration timed in a process then reads
10-20 % slow (seen with tools/kbench_kivi.py: 199.6 us first, 181.5 us for the same launch later in the process)."""
import time

import torch


def warm(seconds: float = 0.5):
    a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
    t_end = time.perf_counter() + seconds
    while time.perf_counter() < t_end:
        for _ in range(20):
            a = (a @ a).clamp_(-1, 1)
        torch.cuda.synchronize()
