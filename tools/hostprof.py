#!/usr/bin/env python3
"""Host-side cost of one graph-replayed decode step (development tool, GPU only): cProfile over N steps at batch B.

    python tools/hostprof.py [--batch 1] [--steps 300] [--method h2o]
"""
import argparse
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sparse_vllm_amd.config import Config
from tools.synthetic import SyntheticDecodeDriver as SparseDecodeDriver


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--top", type=int, default=28)
    args = ap.parse_args()
    B = args.batch
    cfg = Config.from_kwargs(sparse_method="h2o", h2o_decode_budget=4096, h2o_decode_eviction_interval=128,
                             h2o_prefill_budget=8192, max_model_len=4224 + 64, max_num_seqs_in_gpu=B,
                             num_kvcache_slots=B * 4224 + 4096)
    drv = SparseDecodeDriver(cfg)
    drv.cache_manager.permute_free_slots(1)
    drv.admit_resident_rows(B, 4096, logical_len=131072, seed=0, device_rng=True)
    q, k, v = drv.random_step_inputs(seed=1)
    drv.enable_decode_graph()
    for _ in range(132):
        drv.step(q, k, v)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        drv.step(q, k, v)
    torch.cuda.synchronize()
    print("ms/step %.4f" % ((time.perf_counter() - t0) * 1e3 / args.steps))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(args.steps):
        drv.step(q, k, v)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(args.top)


if __name__ == "__main__":
    main()
