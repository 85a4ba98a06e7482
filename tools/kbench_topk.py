"""developer: svk_topk_sorted_desc at the DeltaKV observation shape (n = 262 152 scores, k = 2048) over score distributions;
SVK_TOPK_FINAL=select python3 tools/kbench_topk.py for the single-workgroup final stage."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_vllm_amd.kernels.deltakv_kernels import topk_sorted_desc  # noqa: E402


def main():
    d = torch.device("cuda:0")
    n, k = 262152, 2048
    g = torch.Generator(device=d).manual_seed(1)
    logits = torch.randn(4, n, generator=g, device=d) * 2.0
    cases = {
        "fp32 normal": logits,
        "softmax prob fp32": torch.softmax(logits, dim=-1),
        "softmax prob bf16-valued": torch.softmax(logits, dim=-1).bfloat16().float(),
        "max-of-28-heads prob bf16": torch.softmax(torch.randn(4, 28, n, generator=g, device=d) * 2.0, dim=-1).max(dim=1).values.bfloat16().float(),
        "coarse (64 values)": (torch.rand(4, n, generator=g, device=d) * 64).floor(),
    }
    for name, sc in cases.items():
        for rows in (1, 4):
            x = sc[:rows].contiguous()
            ref = torch.sort(x, dim=1, descending=True, stable=True).indices[:, :k].to(torch.int32)
            got = topk_sorted_desc(x, k)
            ok = bool(torch.equal(got, ref))
            # candidates of the final stage: keys at or above the k-th value's histogram bin ~ count of scores >= k-th value
            kth = torch.gather(x, 1, ref[:, -1:].long())
            m_est = int((x >= kth).sum(dim=1).max())
            for _ in range(5):
                topk_sorted_desc(x, k)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                topk_sorted_desc(x, k)
            e1.record()
            torch.cuda.synchronize()
            print(f"{name:28s} rows {rows}: {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us per call (4 launches, eager)   exact {ok}   "
                  f"scores >= k-th value: {m_est}")


if __name__ == "__main__":
    main()
