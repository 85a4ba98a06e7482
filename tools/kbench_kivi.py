#!/usr/bin/env python3
"""KIVI-int4 full-layer decode stage 1 micro-benchmark (development tool, GPU only).

    python tools/kbench_kivi.py [--batches 1,4] [--ctx 262152] [--block-seqs 256,512] [--iters 20]
Algorithmic bytes per token: KV heads x (64 B K codes + 64 B V codes + 32 B fp32 per-channel K scale/min
+ 16 B bf16 per-token V scale/min) + 8 B of slot maps = 712 B with 4 KV heads (184 B for a one-KV-head TP rank).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

if os.environ.get("SVK_AB_LIB"):          # developer A/B: another build of the library (e.g. sparse_vllm_amd/libsvk_ab.so)
    import sparse_vllm_amd._lib as _svk_lib
    _svk_lib.LIB_PATH = os.path.abspath(os.environ["SVK_AB_LIB"])
from sparse_vllm_amd.kernels.deltakv_kernels import full_layer_kivi_flash_decode_stage1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="1,4")
    ap.add_argument("--ctx", type=int, default=262152)
    ap.add_argument("--block-seqs", default="256,512,1024")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--score", action="store_true")
    ap.add_argument("--warmup-s", type=float, default=0.4, help="seconds of launches before each timed configuration")
    ap.add_argument("--no-extra", action="store_true", help="no spare partial slots: the launch without extra workgroups")
    ap.add_argument("--sink", type=int, default=8)
    ap.add_argument("--tail", type=int, default=48)
    ap.add_argument("--heads", default="28,4", help="query heads, KV heads of the rank (7,1 = one TP=4 rank of Qwen2.5-7B)")
    args = ap.parse_args()
    d = torch.device("cuda:0")
    Hq, Hkv = (int(x) for x in args.heads.split(","))
    D, G, sink, tail = 128, 32, args.sink, args.tail
    torch.manual_seed(1)
    for B in [int(x) for x in args.batches.split(",")]:
        L = args.ctx
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
        score = torch.empty(B, Hq, L, device=d) if args.score else None
        for bs in [int(x) for x in args.block_seqs.split(",")]:
            nblk = (L + bs - 1) // bs
            spare = 0 if args.no_extra else 3
            mid = torch.empty(B, Hq, nblk + spare, D, device=d)
            lse = torch.empty(B, Hq, nblk + spare, device=d)

            def run():
                full_layer_kivi_flash_decode_stage1(
                    q=q, raw_k=raw_k, raw_v=raw_v, raw_slots_map=raw_map, kivi_block_slots_map=blk_map,
                    kivi_block_start_pos=blk_start, key_packed=kp, key_scales=ks, key_mins=km, value_packed=vp,
                    value_scales=vs, value_mins=vm, req_indices=req, context_lens=lens, max_len_in_batch=L, mid_out=mid,
                    mid_out_logsumexp=lse, group_size=G, block_seq=bs, attn_score=score, extra_partial_slots=spare)
            # warm-up by time, not by count: the first configuration timed in a process otherwise reads 10-20 % slow
            # (clocks ramp over tens of milliseconds of sustained work; 30 launches of 200 us are not enough)
            import time
            t_end = time.perf_counter() + args.warmup_s
            while True:
                for _ in range(10):
                    run()
                torch.cuda.synchronize()
                if time.perf_counter() >= t_end:
                    break
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / args.iters
            # per token: Hkv x (64 B K codes + 64 B V codes + 32 B K scale/min + 16 B V scale/min) + 8 B of slot maps
            byts = B * L * (Hkv * 176 + 8 + (4 * Hq if args.score else 0))
            print(f"kivi stage1 B={B} L={L} block_seq={bs:5d} score={int(args.score)} extra={spare}: {us:9.1f} us  {byts / us / 1e6:7.3f} TB/s "
                  f"({byts / us / 1e6 / 8.0 * 100:5.1f}% of 8 TB/s)", flush=True)


if __name__ == "__main__":
    main()
