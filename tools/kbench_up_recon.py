#!/usr/bin/env python3
"""Micro-benchmark of svk_deltakv_up_reconstruct against the library GEMM + svk_deltakv_reconstruct_writeback_batched pair
(development tool, GPU only; each variant replayed from a hipGraph over rotating buffers).

    python tools/kbench_up_recon.py [--rows 2048,8192] [--hidden 2048] [--layers 2]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

if os.environ.get("SVK_AB_LIB"):          # developer A/B: another build of the library
    from sparse_vllm_amd import _lib as _svk_lib
    _svk_lib.LIB_PATH = os.path.abspath(os.environ["SVK_AB_LIB"])
from sparse_vllm_amd.kernels import deltakv_kernels as dk


def graph_time(fn, iters=30):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * iters)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="2048,8192")
    ap.add_argument("--hidden", default="2048")
    ap.add_argument("--layers", type=int, default=2)
    a = ap.parse_args()
    d = torch.device("cuda:0")
    Hkv, D, kf, nl = 4, 128, 4, a.layers
    torch.manual_seed(0)
    for hid in [int(x) for x in a.hidden.split(",")]:
        for n in [int(x) for x in a.rows.split(",")]:
            latents, slots, sets = 300000, 400000, 4
            B, K, W = max(1, n // 2048), min(n, 2048), min(n, 2048) + 136
            bufs = []
            for _ in range(sets):
                hbuf = torch.zeros(nl, n, hid + 64, dtype=torch.bfloat16, device=d)
                hbuf[:, :, :hid] = (torch.randn(nl, n, hid, device=d) * 0.5).bfloat16()
                hbuf[:, :, hid] = 1.0
                wbuf = torch.zeros(nl, 2 * Hkv * D, hid + 64, dtype=torch.bfloat16, device=d)
                wbuf[:, :, :hid] = (torch.randn(nl, 2 * Hkv * D, hid, device=d) * (hid ** -0.5)).bfloat16()
                bias = (torch.randn(nl, 2 * Hkv * D, device=d) * 0.1).bfloat16()
                wbuf[:, :, hid] = bias
                bufs.append((hbuf, wbuf, bias))
            hot = int(os.environ.get("KB_HOT_FATHERS", "0"))      # fathers from the first `hot` slots only (cache-resident gathers)
            table = torch.randint(0, hot or slots // 2, (nl, latents, kf), dtype=torch.int32, device=d)
            row_index = torch.randperm(latents, device=d)[:n].to(torch.int32)
            slot_to_pos = torch.randint(0, 500, (slots,), dtype=torch.int32, device=d)
            out_slots = (slots // 2 + torch.randperm(slots // 2, device=d)[:n]).to(torch.int32)
            out_pos = torch.randint(0, 512 if hot else 262144, (n,), dtype=torch.int32, device=d)
            cos_sin = torch.randn(262144, D, device=d)
            kc = (torch.randn(nl, slots, Hkv, D, device=d) * 0.3).bfloat16()
            vc = (torch.randn(nl, slots, Hkv, D, device=d) * 0.3).bfloat16()
            vk = torch.zeros(nl, B * W, Hkv, D, dtype=torch.bfloat16, device=d)
            vv = torch.zeros_like(vk)
            view = (vk, vv, W, 8, K)
            delta = torch.empty(nl, n, 2 * Hkv * D, dtype=torch.bfloat16, device=d)
            it = [0]

            def lib_pair():
                hbuf, wbuf, _ = bufs[it[0] % sets]
                it[0] += 1
                for c0 in range(0, n, 4096):
                    torch.bmm(hbuf[:, c0:c0 + 4096], wbuf.transpose(1, 2), out=delta[:, c0:c0 + 4096])
                dk.deltakv_reconstruct_writeback_layers(delta, table, row_index, slot_to_pos, out_slots, out_pos, cos_sin, kc, vc,
                                                        raw_k_cache=True, store_raw_k=False, view_out=view)

            def lib_gemm():
                hbuf, wbuf, _ = bufs[it[0] % sets]
                it[0] += 1
                for c0 in range(0, n, 4096):
                    torch.bmm(hbuf[:, c0:c0 + 4096], wbuf.transpose(1, 2), out=delta[:, c0:c0 + 4096])

            def fused():
                hbuf, wbuf, bias = bufs[it[0] % sets]
                it[0] += 1
                dk.deltakv_up_reconstruct_layers(hbuf[:, :, :hid], wbuf[:, :, :hid], bias, table, row_index, slot_to_pos, out_slots,
                                                 out_pos, cos_sin, kc, vc, view_out=view)

            t_pair, t_gemm, t_fused = graph_time(lib_pair), graph_time(lib_gemm), graph_time(fused)
            fl = 2.0 * nl * n * hid * 2 * Hkv * D
            print(f"layers={nl} rows={n:6d} hidden={hid:5d}: library GEMM {t_gemm:7.1f} us ({fl / t_gemm / 1e6:6.0f} TFLOP/s)  "
                  f"GEMM + reconstruct {t_pair:7.1f} us   fused {t_fused:7.1f} us ({fl / t_fused / 1e6:6.0f} TFLOP/s)")


if __name__ == "__main__":
    main()
