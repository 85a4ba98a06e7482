#!/usr/bin/env python3
"""Phase stamps of the main loop of svk_deltakv_up_reconstruct (developer tool; needs a library built with
`make -C sparse_vllm_amd/csrc EXTRA=-DSVK_UR_TIMING BUILD=... LIB=...` and SVK_AB_LIB pointing at it).

    SVK_AB_LIB=sparse_vllm_amd/libsvk_ab.so [SVK_UP_RECON_TM=128|256|2564] python tools/ur_timing.py [rows]
Per wave of the first 64 workgroups, steps 8..15: cycles (s_memtime, 100 MHz x ... shader clock) between the stamps
0 step start | 1 reads + MFMAs of k-substeps 0, 1 issued | 2 next tile waited for | 3 behind the barrier | 4 DMA issued |
5 step end; printed as medians over waves and steps."""
import ctypes as C
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (before the library: both must bind the same HIP runtime)

from sparse_vllm_amd import _lib
if os.environ.get("SVK_AB_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["SVK_AB_LIB"])
lib = _lib.load()
if not (lib.svk_build_flags() & 8):
    raise SystemExit("ur_timing.py needs a developer build: EXTRA=-DSVK_UR_TIMING")
import runpy
sys.argv = [sys.argv[0]] + sys.argv[1:]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ur_one.py"), run_name="__main__")
out = np.zeros(64 * 8 * 8 * 6, dtype=np.uint64)
lib.svk_debug_up_recon_stamps.argtypes = [C.c_void_p]
lib.svk_debug_up_recon_stamps(out.ctypes.data_as(C.c_void_p))
st = out.reshape(64, 8, 8, 6).astype(np.int64)
waves = int(os.environ.get("UR_WAVES", "8"))
st = st[:, :waves]
ok = (st > 0).all(axis=-1)
d = np.diff(st, axis=-1)[ok]
step = (st[:, :, 1:, 0] - st[:, :, :-1, 0])[ok[:, :, 1:] & ok[:, :, :-1]]
names = ["reads + 8 MFMAs issued", "wait for the next tile", "barrier", "DMA issue", "reads + 8 MFMAs issued (2nd half)"]
print(f"{ok.sum()} (wave, step) samples; step start -> next step start: median {np.median(step):.0f} cycles (p10 {np.percentile(step, 10):.0f}, p90 {np.percentile(step, 90):.0f})")
for i, n in enumerate(names):
    print(f"  {n:38s} median {np.median(d[:, i]):7.0f}   p90 {np.percentile(d[:, i], 90):7.0f} cycles")
