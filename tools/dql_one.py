import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from sparse_vllm_amd.kernels import deltakv_kernels as dk
d = torch.device("cuda:0"); g = torch.Generator(device=d).manual_seed(0)
rows, L, src = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 2, 300000
code = torch.randint(-2**31, 2**31-1, (L, src, 32), dtype=torch.int32, device=d, generator=g)
scale = (torch.rand((L, src, 8), device=d, generator=g) * 0.1 + 0.01).to(torch.bfloat16); mn = (scale.float() * -7.5).to(torch.bfloat16)
w1 = (torch.randn((L, 2048, 256), device=d, generator=g) / 16).to(torch.bfloat16); b1 = (torch.randn((L, 2048), device=d, generator=g) * 0.1).to(torch.bfloat16)
ridx = torch.randint(0, src, (rows,), dtype=torch.int32, device=d, generator=g)
out = torch.empty((L, rows, 2048), dtype=torch.bfloat16, device=d)
for _ in range(6): dk.dequant_linear_act(code, scale, mn, 32, w1, b1, activation="gelu", row_index=ridx, out=out, layers=True)
torch.cuda.synchronize()
