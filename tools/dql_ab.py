import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
if os.environ.get("SVK_AB_LIB"):          # developer A/B: another build of the library
    from sparse_vllm_amd import _lib as _svk_lib
    _svk_lib.LIB_PATH = os.path.abspath(os.environ["SVK_AB_LIB"])
from sparse_vllm_amd.kernels import deltakv_kernels as dk
d = torch.device("cuda:0"); g = torch.Generator(device=d).manual_seed(0)
def t(fn, iters=40):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / iters
for rows in (2048, 8192, 16384):
    L, src = 2, 300000
    code = torch.randint(-2**31, 2**31-1, (L, src, 32), dtype=torch.int32, device=d, generator=g)
    scale = (torch.rand((L, src, 8), device=d, generator=g) * 0.1 + 0.01).to(torch.bfloat16); mn = (scale.float() * -7.5).to(torch.bfloat16)
    w1 = (torch.randn((L, 2048, 256), device=d, generator=g) / 16).to(torch.bfloat16); b1 = (torch.randn((L, 2048), device=d, generator=g) * 0.1).to(torch.bfloat16)
    ridx = torch.randint(0, src, (rows,), dtype=torch.int32, device=d, generator=g)
    out = torch.empty((L, rows, 2048), dtype=torch.bfloat16, device=d)
    f = lambda: dk.dequant_linear_act(code, scale, mn, 32, w1, b1, activation="gelu", row_index=ridx, out=out, layers=True)
    f(); torch.cuda.synchronize(); ref = out.clone()
    print(rows, "us", round(t(f), 1), "checksum", int(ref.view(torch.int16).long().sum()))
