#!/usr/bin/env python3
"""Kernel micro-benchmark for the scored decode path (development tool, GPU only).

    python tools/kbench.py [--batches 1,8,64] [--block-seqs 64,128,256] [--iters 50]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sparse_vllm_amd.kernels import flash_decode_stage1, flash_decode_stage1_with_score, flash_decode_stage2
from sparse_vllm_amd.kernels.h2o_ops import h2o_decode_score_update


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="1,8,64")
    ap.add_argument("--block-seqs", default="64,128,256")
    ap.add_argument("--len", type=int, default=4224)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--modes", default="2,0")
    ap.add_argument("--layers", type=int, default=6, help="distinct K/V sets cycled through (defeats the 256 MB MALL)")
    ap.add_argument("--graph-pair", action="store_true", help="time stage 1 + stage 2 of one layer, back to back in a hipGraph "
                    "(small batches: eager launches are CPU-bound)")
    ap.add_argument("--heads", default="28,4", help="query heads, KV heads of the rank (7,1 = one TP=4 rank of Qwen2.5-7B)")
    args = ap.parse_args()
    from _warm import warm
    warm()                                  # clocks settled before the first timed configuration (tools/_warm.py)
    d = torch.device("cuda:0")
    Hq, Hkv = (int(x) for x in args.heads.split(","))
    D, L = 128, args.len
    torch.manual_seed(20260625)
    for B in [int(x) for x in args.batches.split(",")]:
        slots = B * L + 4096
        kcs = [(torch.randn(slots, Hkv, D, device=d) * 0.3).bfloat16() for _ in range(args.layers)]
        vcs = [(torch.randn(slots, Hkv, D, device=d) * 0.3).bfloat16() for _ in range(args.layers)]
        it = [0]
        q = (torch.randn(B, Hq, D, device=d) * 0.3).bfloat16()
        perm = torch.randperm(slots, device=d)[: B * L].to(torch.int32).view(B, L)
        req = torch.zeros(B, L + 128, dtype=torch.int32, device=d)
        req[:, :L] = perm
        bidx = torch.arange(B, dtype=torch.int32, device=d)
        blen = torch.full((B,), L, dtype=torch.int32, device=d)
        for bs in [int(x) for x in args.block_seqs.split(",")]:
            nblk = (L + bs - 1) // bs
            mid = torch.empty(B, Hq, nblk, D, device=d)
            lse = torch.empty(B, Hq, nblk, device=d)
            score = torch.full((B, L), -1e20, device=d)
            o = torch.empty_like(q)
            if args.graph_pair:
                def pair():
                    kc, vc = kcs[it[0] % args.layers], vcs[it[0] % args.layers]
                    it[0] += 1
                    if bs >= L:          # one block per row: stage 1 writes the output itself, no merge launch (as the layer does)
                        flash_decode_stage1(q, kc, vc, req, bidx, blen, L, mid, lse, bs, direct_out=o)
                    else:
                        flash_decode_stage1(q, kc, vc, req, bidx, blen, L, mid, lse, bs)
                        flash_decode_stage2(mid, lse, blen, o, bs)
                pair()
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for _ in range(args.iters):
                        pair()
                g.replay()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    g.replay()
                e1.record()
                torch.cuda.synchronize()
                print(f"stage1+stage2 (graph) B={B:3d} L={L} block_seq={bs:4d} nblk={nblk}: "
                      f"{e0.elapsed_time(e1) * 1e3 / (5 * args.iters):8.2f} us", flush=True)
                continue
            for mode in [int(x) for x in args.modes.split(",")]:
                def run():
                    kc, vc = kcs[it[0] % args.layers], vcs[it[0] % args.layers]
                    it[0] += 1
                    if mode == 2:
                        flash_decode_stage1_with_score(q, kc, vc, req, bidx, blen, L, mid, lse, score, bs)
                    else:
                        flash_decode_stage1(q, kc, vc, req, bidx, blen, L, mid, lse, bs)
                for _ in range(5):
                    run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    run()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / args.iters
                byts = B * L * (2 * Hkv * D * 2 + 4 + (4 if mode == 2 else 0))
                print(f"stage1 B={B:3d} L={L} block_seq={bs:4d} mode={mode}: {us:9.2f} us  "
                      f"{byts / us / 1e6:8.3f} TB/s  ({byts / us / 1e6 / 8.0 * 100:5.1f}% of 8 TB/s)", flush=True)
            # stage2 + score update
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3):
                flash_decode_stage2(mid, lse, blen, o, bs)
            e0.record()
            for _ in range(args.iters):
                flash_decode_stage2(mid, lse, blen, o, bs)
            e1.record()
            torch.cuda.synchronize()
            print(f"   stage2 B={B} nblk={nblk}: {e0.elapsed_time(e1) * 1e3 / args.iters:8.2f} us", flush=True)
        cum = torch.zeros(B, L + 128, device=d)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            h2o_decode_score_update(score, D ** -0.5, cum_score=cum, b_req_idx=bidx, b_seqlen=blen)
        e0.record()
        for _ in range(args.iters):
            h2o_decode_score_update(score, D ** -0.5, cum_score=cum, b_req_idx=bidx, b_seqlen=blen)
        e1.record()
        torch.cuda.synchronize()
        print(f"   score_update B={B}: {e0.elapsed_time(e1) * 1e3 / args.iters:8.2f} us", flush=True)
        del kcs, vcs


if __name__ == "__main__":
    main()
