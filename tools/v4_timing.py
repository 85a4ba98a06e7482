import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np, ctypes as C
from sparse_vllm_amd import _lib
from sparse_vllm_amd.kernels import gqa_flash_decoding_stage1 as g1
d = torch.device("cuda:0")
B, Hq, Hkv, D, L = 64, 28, 4, 128, 4224
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 1056
torch.manual_seed(1)
slots = B * L + 4096
sets = [((torch.randn(slots, Hkv, D, device=d) * 0.3).bfloat16(), (torch.randn(slots, Hkv, D, device=d) * 0.3).bfloat16()) for _ in range(4)]
q = (torch.randn(B, Hq, D, device=d) * 0.3).bfloat16()
req = torch.zeros(B, L + 128, dtype=torch.int32, device=d)
req[:, :L] = torch.randperm(slots, device=d)[: B * L].to(torch.int32).view(B, L)
bidx = torch.arange(B, dtype=torch.int32, device=d); blen = torch.full((B,), L, dtype=torch.int32, device=d)
nblk = (L + bs - 1) // bs
mid = torch.empty(B, Hq, nblk, D, device=d); lse = torch.empty(B, Hq, nblk, device=d)
score = torch.full((B, L), -1e20, device=d)
dbg = torch.zeros(B * nblk * Hkv * 8, dtype=torch.int64, device=d)
lib = _lib.load()
for it in range(8):
    kc, vc = sets[it % 4]
    a = g1._stage1_args(q, kc, vc, req, bidx, blen, L, mid, lse, score, bs)
    a.slot_mapping = dbg.data_ptr()
    _lib.check(lib.svk_flash_decode_stage1(C.byref(a), _lib.current_stream_handle()), lib)
torch.cuda.synchronize()
t = dbg.cpu().numpy().reshape(B * nblk, Hkv, 8).astype(np.float64) * 0.01   # 100 MHz -> us
t0 = t[:, :, 0].min()
print("block_seq", bs, "WGs", B * nblk)
print("kernel span (first entry -> last end): %.1f us" % (t[:, :, 4].max() - t0))
print("entry skew: max entry - min entry = %.1f us" % (t[:, :, 0].max() - t0))
for name, a_, b_ in [("staging (entry->barrier)", 0, 1), ("barrier->first K landed", 1, 2), ("first K->loop end", 2, 3), ("epilogue", 3, 4), ("total per wave", 0, 4)]:
    x = t[:, :, b_] - t[:, :, a_]
    print("%-28s mean %.1f  min %.1f  max %.1f us" % (name, x.mean(), x.min(), x.max()))
print("sum of K waits per wave: mean %.1f max %.1f us;  V waits: mean %.1f max %.1f us" % (t[:, :, 5].mean(), t[:, :, 5].max(), t[:, :, 6].mean(), t[:, :, 6].max()))
