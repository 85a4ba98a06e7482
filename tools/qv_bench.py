#!/usr/bin/env python3
"""Average duration of svk_quest_build_view back to back inside one hipGraph (developer tool).

    python tools/qv_bench.py [context] [batch] [token_budget] [normal|bf16] [paged]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sparse_vllm_amd.kernels import quest_ops

ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
budget = int(sys.argv[3]) if len(sys.argv) > 3 else 4672
mode = sys.argv[4] if len(sys.argv) > 4 else "bf16"
paged = len(sys.argv) > 5 and sys.argv[5] == "paged"
ps = 16
d = torch.device("cuda:0")
pages = ctx // ps
n_prev = pages - 1
prev_budget = budget // ps - 1
torch.manual_seed(0)
scores = torch.randn(B, n_prev, device=d) * 3
if mode == "bf16":
    scores = scores.bfloat16().float()
elif mode == "pages":
    # real page scores: svk_quest_score_pages over random min / max metadata (all positive, a narrow band of bf16 values)
    Hq, Hkv, D = 28, 4, 128
    pool = pages * B + 7
    mx = (torch.randn(pool, Hkv, D, device=d) * 0.5 + 1).bfloat16()
    mn = (torch.randn(pool, Hkv, D, device=d) * 0.5 - 1).bfloat16()
    qq = (torch.randn(B, Hq, D, device=d) * 0.5).bfloat16()
    pt = torch.stack([torch.randperm(pool, device=d)[:pages] for _ in range(B)]).to(torch.int32)
    quest_ops.score_pages(qq, mx, mn, pt, torch.arange(B, dtype=torch.int32, device=d),
                          torch.full((B,), ctx - 3, dtype=torch.int32, device=d), scores, page_size=ps, n_prev=n_prev)
    torch.cuda.synchronize()
    print("distinct values per row:", [int(torch.unique(scores[b]).numel()) for b in range(B)], "min/max", float(scores.min()), float(scores.max()))
ptab = torch.stack([torch.randperm(pages * B, device=d)[:pages] for _ in range(B)]).to(torch.int32)
ttab = torch.zeros(B, ctx, dtype=torch.int32, device=d)
req = torch.arange(B, dtype=torch.int32, device=d)
lens = torch.full((B,), ctx - 3, dtype=torch.int32, device=d)
keep = (prev_budget + 1) * ps
packed = torch.zeros(B, keep, dtype=torch.int32, device=d)
ll = torch.zeros(B, dtype=torch.int32, device=d)
lr = torch.zeros(B, dtype=torch.int32, device=d)


def launch():
    quest_ops.build_view(scores, ptab, ttab, req, lens, packed, ll, lr, page_size=ps, n_prev=n_prev, prev_budget=prev_budget,
                         token_budget=budget, page_budget_base=budget // ps, max_keep=keep, is_long_text=True,
                         emit_page_slots=paged)


launch()
torch.cuda.synchronize()
N = 50
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(N):
        launch()
g.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    g.replay()
e1.record()
torch.cuda.synchronize()
print(f"ctx {ctx} B {B} budget {budget} {mode} {'paged' if paged else 'tokens'}: {e0.elapsed_time(e1) * 1e3 / (10 * N):.2f} us per launch")
