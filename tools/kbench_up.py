#!/usr/bin/env python3
"""Residual-load micro-benchmark (development tool, GPU only): fused dequant+Linear+GELU vs the three separate ops.

    python tools/kbench_up.py [--rows 2048] [--latent 256] [--hidden 2048] [--out 1024]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from sparse_vllm_amd.kernels.deltakv_kernels import dequant_linear_act, dequantize_grouped


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=2048)
    ap.add_argument("--latent", type=int, default=256)
    ap.add_argument("--hidden", type=int, default=2048)
    ap.add_argument("--out", type=int, default=1024)
    a = ap.parse_args()
    d = torch.device("cuda:0")
    g = torch.Generator(device=d).manual_seed(0)
    src = 8192
    code = torch.randint(-2 ** 31, 2 ** 31 - 1, (src, a.latent // 8), dtype=torch.int32, device=d, generator=g)
    scale = (torch.rand((src, a.latent // 32), device=d, generator=g) * 0.1 + 0.01).to(torch.bfloat16)
    mn = (scale.float() * -7.5).to(torch.bfloat16)
    w1 = (torch.randn((a.hidden, a.latent), device=d, generator=g) / a.latent ** 0.5).to(torch.bfloat16)
    b1 = (torch.randn((a.hidden,), device=d, generator=g) * 0.1).to(torch.bfloat16)
    w2 = (torch.randn((a.out, a.hidden), device=d, generator=g) / a.hidden ** 0.5).to(torch.bfloat16)
    b2 = (torch.randn((a.out,), device=d, generator=g) * 0.1).to(torch.bfloat16)
    ridx = torch.randint(0, src, (a.rows,), dtype=torch.int32, device=d, generator=g)

    def separate():
        x = dequantize_grouped(code, scale, mn, 32, a.latent, 4, row_index=ridx)
        return F.linear(F.gelu(F.linear(x, w1, b1)), w2, b2)

    def fused():
        return F.linear(dequant_linear_act(code, scale, mn, 32, w1, b1, activation="gelu", row_index=ridx), w2, b2)

    def fused_only():
        return dequant_linear_act(code, scale, mn, 32, w1, b1, activation="gelu", row_index=ridx)

    y0, y1 = separate().float(), fused().float()
    print("max |diff| %.4g (|y| max %.3g), identical bf16 %.4f" % ((y0 - y1).abs().max().item(), y0.abs().max().item(),
                                                                 (y0 == y1).float().mean().item()))
    flop = 2.0 * a.rows * a.latent * a.hidden
    t = timeit(fused_only)
    print("fused dequant+linear+gelu : %7.1f us  (%.0f TFLOP/s)" % (t, flop / t * 1e-6))
    timeit(lambda: dequant_linear_act(code, scale, mn, 32, w1, b1, activation="none", row_index=ridx))
    print("residual load, separate   : %7.1f us" % timeit(separate))
    print("residual load, fused      : %7.1f us" % timeit(fused))


if __name__ == "__main__":
    main()
