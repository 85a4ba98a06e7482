"""developer: a few eager launches of svk_deltakv_up_reconstruct at one shape (for counter collection)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sparse_vllm_amd.kernels import deltakv_kernels as dk

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
hid, nl, Hkv, D, kf = 2048, 2, 4, 128, 4
d = torch.device("cuda:0")
torch.manual_seed(0)
latents, slots = 300000, 400000
hbuf = (torch.randn(nl, n, hid + 64, device=d) * 0.5).bfloat16()
wbuf = (torch.randn(nl, 2 * Hkv * D, hid + 64, device=d) * hid ** -0.5).bfloat16()
bias = (torch.randn(nl, 2 * Hkv * D, device=d) * 0.1).bfloat16()
table = torch.randint(0, slots // 2, (nl, latents, kf), dtype=torch.int32, device=d)
row_index = torch.randperm(latents, device=d)[:n].to(torch.int32)
slot_to_pos = torch.randint(0, 500, (slots,), dtype=torch.int32, device=d)
out_slots = (slots // 2 + torch.randperm(slots // 2, device=d)[:n]).to(torch.int32)
out_pos = torch.randint(0, 262144, (n,), dtype=torch.int32, device=d)
cos_sin = torch.randn(262144, D, device=d)
kc = (torch.randn(nl, slots, Hkv, D, device=d) * 0.3).bfloat16()
vc = (torch.randn(nl, slots, Hkv, D, device=d) * 0.3).bfloat16()
B, K = max(1, n // 2048), min(n, 2048)
W = K + 136
vk = torch.zeros(nl, B * W, Hkv, D, dtype=torch.bfloat16, device=d)
vv = torch.zeros_like(vk)
for _ in range(6):
    dk.deltakv_up_reconstruct_layers(hbuf[:, :, :hid], wbuf[:, :, :hid], bias, table, row_index, slot_to_pos, out_slots, out_pos,
                                     cos_sin, kc, vc, view_out=(vk, vv, W, 8, K))
torch.cuda.synchronize()
