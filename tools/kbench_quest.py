#!/usr/bin/env python3
"""Quest decode selection micro-benchmark (development tool, GPU only): svk_quest_score_pages and svk_quest_build_view
alone, hipGraph-replayed over `--sets` rotating metadata sets (every layer of a step has its own page metadata, so a
single set would be served from L2 / MALL).

    python tools/kbench_quest.py [--ctx 131072] [--batch 4] [--budget 4672] [--sets 8] [--iters 20]
Algorithmic bytes of the scoring: pages x Hkv x D x 2 (max, min) x 2 B per sequence."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sparse_vllm_amd.kernels import quest_ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ctx", type=int, default=131072)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--budget", type=int, default=4672)
    ap.add_argument("--sets", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--heads", default="28,4")
    args = ap.parse_args()
    from _warm import warm
    warm()                                  # clocks settled before the first timed configuration (tools/_warm.py)
    d = torch.device("cuda:0")
    Hq, Hkv = (int(x) for x in args.heads.split(","))
    D, ps, B, ctx = 128, 16, args.batch, args.ctx
    pages = ctx // ps
    n_prev = pages - 1
    prev_budget = args.budget // ps - 1
    torch.manual_seed(0)
    pool = pages * B + 7
    metas = [((torch.randn(pool, Hkv, D, device=d) * 0.5 + 1).bfloat16(), (torch.randn(pool, Hkv, D, device=d) * 0.5 - 1).bfloat16())
             for _ in range(args.sets)]
    q = (torch.randn(B, Hq, D, device=d) * 0.5).bfloat16()
    ptab = torch.stack([torch.randperm(pool, device=d)[:pages] for _ in range(B)]).to(torch.int32)
    ttab = torch.zeros(B, ctx, dtype=torch.int32, device=d)
    req = torch.arange(B, dtype=torch.int32, device=d)
    lens = torch.full((B,), ctx - 3, dtype=torch.int32, device=d)
    scores = torch.zeros(B, n_prev, dtype=torch.float32, device=d)
    keep = (prev_budget + 1) * ps
    packed = torch.zeros(B, keep, dtype=torch.int32, device=d)
    ll = torch.zeros(B, dtype=torch.int32, device=d)
    lr = torch.zeros(B, dtype=torch.int32, device=d)

    def score(i):
        mx, mn = metas[i % args.sets]
        quest_ops.score_pages(q, mx, mn, ptab, req, lens, scores, page_size=ps, n_prev=n_prev)

    def view(i):
        quest_ops.build_view(scores, ptab, ttab, req, lens, packed, ll, lr, page_size=ps, n_prev=n_prev, prev_budget=prev_budget,
                             token_budget=args.budget, page_budget_base=args.budget // ps, max_keep=keep, is_long_text=True)

    def timed(fn, name, nbytes):
        for i in range(3):
            fn(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                for i in range(args.sets):
                    fn(i)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (args.iters * args.sets)
        extra = f"  {nbytes / us / 1e6:6.3f} TB/s" if nbytes else ""
        print(f"quest {name:12s} B={B} ctx={ctx} budget={args.budget}: {us:7.2f} us per launch (graph){extra}", flush=True)

    timed(score, "score_pages", B * n_prev * Hkv * D * 2 * 2)
    timed(view, "build_view", 0)
    timed(lambda i: (score(i), view(i)), "score+view", 0)


if __name__ == "__main__":
    main()
