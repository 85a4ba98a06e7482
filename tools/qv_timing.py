#!/usr/bin/env python3
"""Phase stamps of quest_build_view_kernel (developer tool; needs `make -C sparse_vllm_amd/csrc EXTRA=-DSVK_QV_TIMING`).

    python tools/qv_timing.py [context] [batch] [token_budget]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sparse_vllm_amd import _lib
from sparse_vllm_amd.kernels import quest_ops

if not (_lib.load().svk_build_flags() & 2):
    raise SystemExit("qv_timing.py needs a developer build: make -C sparse_vllm_amd/csrc EXTRA=-DSVK_QV_TIMING")
ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
budget = int(sys.argv[3]) if len(sys.argv) > 3 else 4672
mode = sys.argv[4] if len(sys.argv) > 4 else "normal"
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
lib = _lib.load()
out = (C.c_ulonglong * 16)()
for it in range(4):
    quest_ops.build_view(scores, ptab, ttab, req, lens, packed, ll, lr, page_size=ps, n_prev=n_prev, prev_budget=prev_budget,
                         token_budget=budget, page_budget_base=budget // ps, max_keep=keep, is_long_text=True, emit_page_slots=paged)
    torch.cuda.synchronize()
    lib.svk_debug_quest_view_stamps(out)
    t = list(out)
    names = [(1, "stage keys + shared bits"), (5, "bin scan + candidates"), (8, "pass b31.. / 16-bit histogram"), (9, "pass b23.."),
             (10, "pass b15.."), (11, "pass b7.."), (2, "threshold"),
             (3, "ordered emit (+ view, keys in registers)"), (4, "page gather + view / tail")]
    prev, line = t[0], []
    for i, nm in names:
        if t[i] > prev:
            line.append(f"{nm} {(t[i] - prev) / 100:.2f}")
            prev = t[i]
    print(f"total {(t[4] - t[0]) / 100:.2f} us | " + " | ".join(line))
    for i in range(16):
        out[i] = 0
