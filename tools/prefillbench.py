#!/usr/bin/env python3
"""Sparse-path cost of a long chunked prefill under H2O (development tool, GPU only): one sequence of `--prompt` tokens
in `--chunk`-token chunks through SparseDecodeDriver.prefill_chunk - per layer: slot allocation, store_kvcache, the
chunk's causal attention over the compressed row, prefill_score on the last-window queries, and the chunk-end
eviction/compaction (Qwen2.5-7B heads, synthetic q/k/v; the dense model layers are not part of this build).

    python tools/prefillbench.py [--prompt 131072] [--chunk 8192] [--layers 28]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sparse_vllm_amd.config import Config
from tools.synthetic import SyntheticDecodeDriver as SparseDecodeDriver
from sparse_vllm_amd.engine.sequence import Sequence


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prompt", type=int, default=131072)
    ap.add_argument("--chunk", type=int, default=8192)
    ap.add_argument("--layers", type=int, default=28)
    ap.add_argument("--no-attention", action="store_true", help="scores / eviction only (skip the chunk's attention)")
    a = ap.parse_args()
    L, Hq, Hkv, D = a.layers, 28, 4, 128
    conf = Config.from_kwargs(sparse_method="h2o", num_hidden_layers=L, num_attention_heads=Hq, num_key_value_heads=Hkv, head_dim=D,
                              max_model_len=a.prompt + 256, max_num_seqs_in_gpu=1, num_kvcache_slots=8192 + a.chunk + 4096,
                              h2o_decode_budget=4096, h2o_decode_eviction_interval=128, h2o_prefill_budget=8192,
                              engine_prefill_chunk_size=a.chunk)
    drv = SparseDecodeDriver(conf)
    drv.cache_manager.permute_free_slots(1)
    seq = Sequence(num_prompt_tokens=a.prompt)
    d = drv.device
    g = torch.Generator(device=d).manual_seed(0)
    q = (torch.randn((L, a.chunk, Hq, D), device=d, generator=g) * 0.3).to(torch.bfloat16)
    k = (torch.randn((L, a.chunk, Hkv, D), device=d, generator=g) * 0.3).to(torch.bfloat16)
    v = (torch.randn((L, a.chunk, Hkv, D), device=d, generator=g) * 0.3).to(torch.bfloat16)
    outs = None if a.no_attention else torch.empty_like(q)
    times = []
    while seq.num_prefilled_tokens < seq.num_prompt_tokens:
        n = min(a.chunk, seq.num_prompt_tokens - seq.num_prefilled_tokens)
        seq.current_chunk_size = n
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        drv.prefill_chunk([seq], q[:, :n], k[:, :n], v[:, :n], outputs=None if outs is None else outs[:, :n])
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    row = int(drv.cache_manager.row_seq_lens[0][drv.cache_manager.seq_id_to_row[0][seq.seq_id]])
    print("H2O prefill %d tokens, chunk %d, %d layers: %.1f ms total (first chunk %.1f, steady %.1f ms/chunk), "
          "%.0f prompt tokens/s through the sparse path; resident row after prefill: %d" % (
              a.prompt, a.chunk, L, sum(times), times[0], sum(times[2:]) / max(1, len(times) - 2),
              a.prompt / sum(times) * 1e3, row))


if __name__ == "__main__":
    main()
