"""developer: svk_cluster_l2_topk (fused ranking product + mask + top-k) against the library path (gather + torch.matmul +
svk_cluster_topk) at the shapes of a DeltaKV eviction.  python3 tools/kbench_cluster.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_vllm_amd.kernels import deltakv_kernels as dk  # noqa: E402


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    d = torch.device("cuda:0")
    H, D = 4, 128
    for rows, m in ((128, 1000), (128, 8000), (512, 8000), (2048, 8000), (8192, 8000)):
        slots = m + 4096
        ck = (torch.randn(slots, H, D, device=d) * 0.5).bfloat16()
        cv = (torch.randn(slots, H, D, device=d) * 0.5).bfloat16()
        kv = (torch.randn(rows, 2 * H * D, device=d) * 0.5).bfloat16()
        m_new = max(1, rows // 33)
        center_slots = torch.randperm(slots, device=d)[:m].to(torch.int32)
        rel = (torch.arange(m_new, device=d) * 33).to(torch.int32)
        m0 = m - m_new

        def fused():
            return dk.cluster_l2_topk(kv, ck, cv, center_slots, m0=m0, new_center_rel=rel, k=4)

        def library():
            idx = center_slots.long()
            centers = torch.cat((ck[idx].reshape(m, -1), cv[idx].reshape(m, -1)), dim=1)
            dot = torch.matmul(kv, centers.t())
            sc = dot.mul(2.0).sub_((centers * centers).sum(dim=1, dtype=torch.float32).to(dot.dtype).unsqueeze(0))
            return dk.cluster_topk(sc, m0=m0, new_center_rel=rel, k=4)

        same = float((fused() == library()).all(dim=1).float().mean())
        tf, tl = timeit(fused), timeit(library)
        flop = 2.0 * rows * m * 2 * H * D
        print(f"rows {rows:5d} centres {m:5d}: fused {tf:8.1f} us ({flop / tf / 1e6:6.1f} TFLOP/s)   library path {tl:8.1f} us   "
              f"rows with identical fathers {same:.4f}   score matrix avoided {rows * m * 2 / 1e6:.1f} MB")


if __name__ == "__main__":
    main()
