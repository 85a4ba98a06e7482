#!/usr/bin/env python3
"""prefill_score micro-benchmark (development tool, GPU only): the last `--window` queries of an 8 k chunk against
`--keys` compressed keys of one sequence, Qwen2.5-7B heads, randomly permuted paged K (W=128, Lc=16384 ->
2 x 15.0 GFLOP for the two Q.K^T passes of the probability mode, 15.0 for the logits mode).
FLOPs counted = 2 * W * Hq * Lc * D per Q.K^T pass; the bound is the dense bf16 MFMA peak (2.5 PFLOP/s)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sparse_vllm_amd.kernels.prefill_score import PrefillScoreWorkspace, prefill_score_fwd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--window", type=int, default=128)
    ap.add_argument("--keys", default="16384,8192")
    ap.add_argument("--chunk", type=int, default=8192)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    from _warm import warm
    warm()                                  # clocks settled before the first timed configuration (tools/_warm.py)
    d = torch.device("cuda:0")
    Hq, Hkv, D, W = 28, 4, 128, args.window
    torch.manual_seed(0)
    for Lc in [int(x) for x in args.keys.split(",")]:
        chunk = min(args.chunk, Lc)
        slots = Lc + 1024
        q = (torch.randn(chunk, Hq, D, device=d) * 0.3).bfloat16()
        k = (torch.randn(slots, Hkv, D, device=d) * 0.3).bfloat16()
        table = torch.randperm(slots, device=d)[:Lc].to(torch.int32).view(1, Lc)
        z = torch.zeros(1, dtype=torch.int32, device=d)
        seq = torch.tensor([Lc], dtype=torch.int32, device=d)
        pcl = torch.tensor([Lc - chunk], dtype=torch.int32, device=d)
        qs = torch.tensor([Lc - W], dtype=torch.int32, device=d)
        qe = torch.tensor([Lc], dtype=torch.int32, device=d)
        score = torch.zeros(1, Lc, dtype=torch.float32, device=d)
        ws = PrefillScoreWorkspace()
        for mode, passes in (("probability", 2), ("logits", 1)):
            def run():
                prefill_score_fwd(q, k, score, z, z, seq, pcl, W, table, qs, qe, candidate_start=0, num_recent_tokens=0,
                                  score_mode=mode, workspace=ws)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / args.iters
            flops = passes * 2.0 * W * Hq * Lc * D
            print(f"prefill_score {mode:11s} W={W} keys={Lc:6d}: {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s "
                  f"({flops / us / 1e6 / 2500 * 100:4.1f}% of 2.5 PF dense bf16; {passes} Q.K^T pass{'es' if passes > 1 else ''})")


if __name__ == "__main__":
    main()
