#!/usr/bin/env python3
"""Phase stamps of the wide KIVI stage-1 kernel (developer tool; needs `make -C sparse_vllm_amd/csrc EXTRA=-DSVK_KV_TIMING`).

    python tools/kv_timing.py [batch] [block_seq] [ctx]
Per workgroup (wave 0): 0 entry, 1 range known, 2 before the tile loop, 3 first K tile in LDS, 4 first tile done, 5 tile
loop done, 6 partials written; printed relative to the earliest entry of the launch, as medians / maxima over the
regular workgroups and the values of the extra workgroups."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from sparse_vllm_amd import _lib
if os.environ.get("SVK_AB_LIB"):          # developer A/B: another build of the library
    _lib.LIB_PATH = os.path.abspath(os.environ["SVK_AB_LIB"])
from sparse_vllm_amd.kernels.deltakv_kernels import full_layer_kivi_flash_decode_stage1

lib = _lib.load()
if not (lib.svk_build_flags() & 4):
    raise SystemExit("kv_timing.py needs a developer build: make -C sparse_vllm_amd/csrc EXTRA=-DSVK_KV_TIMING")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
L = int(sys.argv[3]) if len(sys.argv) > 3 else 262152
d = torch.device("cuda:0")
Hq, Hkv, D, G, sink, tail = 28, 4, 128, 32, 8, 48
torch.manual_seed(1)
nb_row = (L - sink - tail) // G
nblocks = B * nb_row
raw_slots = B * (L - nb_row * G) + 8
raw_k = (torch.randn(raw_slots, Hkv, D, device=d) * 0.3).bfloat16()
raw_v = (torch.randn(raw_slots, Hkv, D, device=d) * 0.3).bfloat16()
q = (torch.randn(B, Hq, D, device=d) * 0.3).bfloat16()
raw_map = torch.full((B, L + 8), -1, dtype=torch.int32, device=d)
blk_map = torch.full((B, L + 8), -1, dtype=torch.int32, device=d)
blk_start = torch.zeros(nblocks, dtype=torch.int32, device=d)
perm = torch.randperm(nblocks, device=d).to(torch.int32).view(B, nb_row)
rp = torch.randperm(raw_slots, device=d).to(torch.int32)
ru = 0
for b in range(B):
    raw_map[b, :sink] = rp[ru: ru + sink]; ru += sink
    blk_map[b, sink: sink + nb_row * G] = perm[b].repeat_interleave(G)
    blk_start[perm[b].long()] = torch.arange(sink, sink + nb_row * G, G, dtype=torch.int32, device=d)
    n_tail = L - sink - nb_row * G
    raw_map[b, sink + nb_row * G: L] = rp[ru: ru + n_tail]; ru += n_tail
ri = lambda *shape: torch.randint(-2 ** 31, 2 ** 31 - 1, shape, device=d, dtype=torch.int64).to(torch.int32)
kp, vp = ri(nblocks, Hkv, D, G // 8), ri(nblocks, Hkv, G, D // 8)
ks = torch.rand(nblocks, Hkv, D, device=d) * 0.1 + 0.02
km = ks * -7.5
vs = (torch.rand(nblocks, Hkv, G, D // G, device=d) * 0.1 + 0.02).bfloat16()
vm = (vs.float() * -7.5).bfloat16()
req = torch.arange(B, dtype=torch.int32, device=d)
lens = torch.full((B,), L, dtype=torch.int32, device=d)
nblk = (L + bs - 1) // bs
mid = torch.empty(B, Hq, nblk + 3, D, device=d)
lse = torch.empty(B, Hq, nblk + 3, device=d)


def run():
    return full_layer_kivi_flash_decode_stage1(
        q=q, raw_k=raw_k, raw_v=raw_v, raw_slots_map=raw_map, kivi_block_slots_map=blk_map, kivi_block_start_pos=blk_start,
        key_packed=kp, key_scales=ks, key_mins=km, value_packed=vp, value_scales=vs, value_mins=vm, req_indices=req,
        context_lens=lens, max_len_in_batch=L, mid_out=mid, mid_out_logsumexp=lse, group_size=G, block_seq=bs,
        extra_partial_slots=3)


lib.svk_debug_kivi_stamps.argtypes = [C.c_void_p]
out = (C.c_ulonglong * (4096 * 8))()
for it in range(4):
    extra = run()
    torch.cuda.synchronize()
    lib.svk_debug_kivi_stamps(out)
    st = np.frombuffer(out, dtype=np.uint64).reshape(4096, 8).astype(np.int64)
    nwg = (nblk + extra) * B
    st = st[:nwg]
    t0 = st[:, 0].min()
    rel = (st - t0) / 100.0          # us
    # grid.x = nblk + extra: workgroup id = blockIdx.x + gridDim.x * blockIdx.y; the kernel's own mapping puts the extras first
    names = ["entry", "range", "pre-loop", "K tile 1", "tile 1 done", "loop done", "end"]
    dur = rel[:, 6] - rel[:, 0]
    order = np.argsort(-dur)
    print(f"launch {it}: {nwg} workgroups, kernel span {rel[:, 6].max():.1f} us; longest workgroups: "
          + ", ".join(f"wg{int(i)}={dur[i]:.1f}" for i in order[:4]))
    reg = order[B * extra:] if extra else order
    sel = np.argsort(dur)[nwg // 2]
    for label, rows in (("median workgroup", rel[sel:sel + 1]), ("slowest regular", rel[order[0]:order[0] + 1])):
        print(f"  {label:18s} " + " | ".join(f"{n} {rows[0, i]:.2f}" for i, n in enumerate(names)))
    print("  medians over workgroups: " + " | ".join(f"{n} {np.median(rel[:, i]):.2f}" for i, n in enumerate(names)))
    if it == 3:
        ids = np.arange(nwg)
        print("  end by workgroup id % 8 (XCD of a round-robin dispatch): "
              + ", ".join(f"{x}: med {np.median(rel[ids % 8 == x, 6]):.0f} max {rel[ids % 8 == x, 6].max():.0f}" for x in range(8)))
        q = np.quantile(rel[:, 6], [0.1, 0.25, 0.5, 0.75, 0.9, 1.0])
        print("  end quantiles 10/25/50/75/90/100 %: " + " ".join(f"{v:.0f}" for v in q))
        per_tile = (rel[:, 5] - rel[:, 4])
        print("  loop time after tile 1, quantiles: " + " ".join(f"{v:.0f}" for v in np.quantile(per_tile, [0.1, 0.5, 0.9, 1.0])))
        slow = np.argsort(-rel[:, 6])[:40]
        print("  40 slowest ids: " + " ".join(str(int(i)) for i in sorted(slow)))
