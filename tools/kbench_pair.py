"""developer: unscored stage 1 + merge of B rows of L tokens inside a replayed graph, over block_seq (what the launch provider
chooses is marked).  python3 tools/kbench_pair.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_vllm_amd.kernels import flash_decode_stage1, flash_decode_stage2  # noqa: E402
from sparse_vllm_amd.operators.decode_attention import Mi355xDecodeLaunchProvider  # noqa: E402


def main():
    d = torch.device("cuda:0")
    Hq, Hkv, D = 28, 4, 128
    prov = Mi355xDecodeLaunchProvider()
    for B, L in ((1, 2184), (4, 2184), (1, 4672), (4, 4672), (8, 4672), (64, 1151), (64, 600), (1, 1151)):
        slots = B * L + 1000
        kc = (torch.randn(6, slots, Hkv, D, device=d) * 0.3).bfloat16()
        vc = (torch.randn(6, slots, Hkv, D, device=d) * 0.3).bfloat16()
        q = (torch.randn(B, Hq, D, device=d) * 0.3).bfloat16()
        tab = torch.stack([torch.randperm(slots, device=d)[:L] for _ in range(B)]).to(torch.int32)
        req = torch.arange(B, dtype=torch.int32, device=d)
        lens = torch.full((B,), L, dtype=torch.int32, device=d)
        o = torch.empty_like(q)
        chosen = prov.launch_config(block_seq=256, max_context_len=L, requires_attention_scores=False, batch_size=B, num_kv_heads=Hkv)[0]
        res = []
        for bs in sorted({32, 48, 64, 96, 128, 192, 256, chosen}):
            nblk = -(-L // bs)
            mid = torch.empty((B, Hq, nblk, D), dtype=torch.float32, device=d)
            lse = torch.empty((B, Hq, nblk), dtype=torch.float32, device=d)

            def run(i):
                flash_decode_stage1(q, kc[i % 6], vc[i % 6], tab, req, lens, L, mid, lse, bs, 16, 4)
                flash_decode_stage2(mid, lse, lens, o, bs)
            run(0)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for i in range(24):
                    run(i)
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            res.append(f"{bs}{'*' if bs == chosen else ''}: {e0.elapsed_time(e1) / 120 * 1e3:5.1f}")
        print(f"B {B:3d} L {L:5d}  us per stage 1 + merge by block_seq (* = provider):  " + "  ".join(res))


if __name__ == "__main__":
    main()
