#!/usr/bin/env python3
"""Prefill attention micro-benchmark (development tool, GPU only): one chunk of `--chunk` query tokens of a sequence
with `--prefix` cached tokens, Qwen2.5-7B heads, randomly permuted paged KV.  FLOPs = 4 * D * Hq * sum_i (prefix + i + 1)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

if os.environ.get("SVK_AB_LIB"):          # developer A/B: another build of the library (e.g. sparse_vllm_amd/libsvk_ab.so)
    import sparse_vllm_amd._lib as _svk_lib
    _svk_lib.LIB_PATH = os.path.abspath(os.environ["SVK_AB_LIB"])
from sparse_vllm_amd.kernels.context_flashattention_nopad import context_attention_fwd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunk", type=int, default=8192)
    ap.add_argument("--prefixes", default="0,8192,57344")
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    from _warm import warm
    warm()                                  # clocks settled before the first timed configuration (tools/_warm.py)
    d = torch.device("cuda:0")
    Hq, Hkv, D = 28, 4, 128
    torch.manual_seed(0)
    for pc in [int(x) for x in args.prefixes.split(",")]:
        L = pc + args.chunk
        slots = L + 1024
        q = (torch.randn(args.chunk, Hq, D, device=d) * 0.3).bfloat16()
        k = (torch.randn(slots, Hkv, D, device=d) * 0.3).bfloat16()
        v = (torch.randn(slots, Hkv, D, device=d) * 0.3).bfloat16()
        o = torch.empty_like(q)
        table = torch.randperm(slots, device=d)[:L].to(torch.int32).view(1, L)
        z = torch.zeros(1, dtype=torch.int32, device=d)
        seq = torch.tensor([L], dtype=torch.int32, device=d)
        pcl = torch.tensor([pc], dtype=torch.int32, device=d)
        run = lambda: context_attention_fwd(q, k, v, o, z, z, seq, pcl, args.chunk, table)
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        flops = 4.0 * D * Hq * (args.chunk * pc + args.chunk * (args.chunk + 1) / 2)
        print(f"prefill attention chunk={args.chunk} prefix={pc}: {ms:8.3f} ms  {flops / ms / 1e9:8.1f} TFLOP/s "
              f"({flops / ms / 1e9 / 2500 * 100:4.1f}% of 2.5 PF dense bf16)", flush=True)


if __name__ == "__main__":
    main()
