import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np, ctypes as C
from sparse_vllm_amd import _lib
from sparse_vllm_amd.kernels import context_flashattention_nopad as cf
d = torch.device("cuda:0")
Hq, Hkv, D, chunk = 28, 4, 128, 8192
pc = int(sys.argv[1]) if len(sys.argv) > 1 else 57344
L = pc + chunk; slots = L + 1024
torch.manual_seed(0)
q = (torch.randn(chunk, Hq, D, device=d) * 0.3).bfloat16()
k = (torch.randn(slots, Hkv, D, device=d) * 0.3).bfloat16(); v = (torch.randn(slots, Hkv, D, device=d) * 0.3).bfloat16()
nwg = (chunk // 32) * Hkv
extra = (nwg * 8 * 8 * 8 + Hq * D * 2 - 1) // (Hq * D * 2) + 1
obuf = torch.zeros(chunk + extra, Hq, D, dtype=torch.bfloat16, device=d)
o = obuf[:chunk]
mode = sys.argv[2] if len(sys.argv) > 2 else "rand"
if mode == "rand":
    table = torch.randperm(slots, device=d)[:L].to(torch.int32).view(1, L)
elif mode == "seq":
    table = torch.arange(L, device=d, dtype=torch.int32).view(1, L)
else:
    table = (torch.arange(L, device=d, dtype=torch.int32) % int(mode)).view(1, L).contiguous()
print("table mode", mode)
z = torch.zeros(1, dtype=torch.int32, device=d); seq = torch.tensor([L], dtype=torch.int32, device=d); pcl = torch.tensor([pc], dtype=torch.int32, device=d)
orig = cf._lib.load
lib = _lib.load()
if not (lib.svk_build_flags() & 1):
    raise SystemExit("pa_timing.py needs a developer build: make -C sparse_vllm_amd/csrc EXTRA=-DSVK_PA_TIMING "
                     "(in a product build bit 30 of max_input_len is part of the length: ~33 M empty workgroups)")
real = lib.svk_context_attention_fwd
class Hook:
    def __call__(self, a, s):
        a._obj.max_input_len |= (1 << 30)
        return real(a, s)
lib_svk = real
import types
cf_args = {}
# monkeypatch: wrap the C entry
def patched(a, s):
    a._obj.max_input_len = a._obj.max_input_len | (1 << 30)
    return real(a, s)
lib.svk_context_attention_fwd = patched
for _ in range(2):
    cf.context_attention_fwd(q, k, v, o, z, z, seq, pcl, chunk, table)
torch.cuda.synchronize()
raw = obuf[chunk:].view(torch.int16).cpu().numpy().reshape(-1).view(np.int64)[: nwg * 64].reshape(nwg, 8, 8).astype(np.float64)
t = raw[:, :7, :] * 0.01   # us
nt = raw[:, 0, 6]
tot = t[:, :, 5]
print("prefix", pc, "WGs", nwg, "tiles per WG mean %.0f" % nt.mean())
for i, name in enumerate(["issue DMA", "QK phase", "softmax+PV", "wait vmcnt", "barrier"]):
    print("%-12s per tile: mean %.3f us" % (name, (t[:, :, i].sum(axis=1) / 7 / nt).mean()))
print("total per wave mean %.1f us; per tile %.3f us" % (tot.mean(), (tot.mean(axis=1) / nt).mean()))
print("per-wave means (us per tile):  wave: issue  qk  smpv  wait  barrier")
for w in range(7):
    print("  wave %d: " % w + "  ".join("%.3f" % (t[:, w, i] / nt).mean() for i in range(5)))
