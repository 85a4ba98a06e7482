"""Synthetic decode state for tests, benches and profiling tools (NOT product code).

`SyntheticDecodeDriver` is `sparse_vllm_amd.engine.decode_driver.SparseDecodeDriver` plus the helpers that put a
cache manager into the state a long prompt leaves behind (resident rows with random K/V payload, random cumulative
H2O scores, Quest page metadata, DeltaKV compressed rows) and that draw random per-layer q / k / v for a step: the
inputs SURVEY.md 8(d) prescribes.  The product driver itself only steps sequences it is handed.
"""

from __future__ import annotations

import numpy as np
import torch

from sparse_vllm_amd.engine.decode_driver import SparseDecodeDriver
from sparse_vllm_amd.engine.sequence import Sequence


class SyntheticDecodeDriver(SparseDecodeDriver):
    # ------------------------------------------------------------------ synthetic state
    def admit_resident_rows(self, batch: int, resident_len: int, *, logical_len: int | None = None, seed: int = 0,
                            kv_scale: float = 0.3, fill_kv: bool = True, device_rng: bool = False):
        """Create `batch` sequences whose physical rows already hold `resident_len` tokens
        per layer (the state a long prompt is in after chunked prefill + final compaction),
        with random K/V payload and, for H2O, random positive cumulative scores."""
        cm = self.cache_manager
        g = torch.Generator(device="cpu").manual_seed(seed)
        gd = torch.Generator(device=self.device).manual_seed(seed) if device_rng else None
        self.seqs = [Sequence(num_prompt_tokens=int(logical_len or resident_len)) for _ in range(batch)]
        for s in self.seqs:
            s.num_prefilled_tokens = s.num_prompt_tokens
        paged = hasattr(cm, "page_size")          # Quest: one page table for all layers
        if paged:
            for s in self.seqs:
                cm._allocate(s.seq_id, resident_len)
        else:
            for layer_idx in cm.kv_transformer_layer_indices():
                for s in self.seqs:
                    cm._allocate(layer_idx, s.seq_id, resident_len)
        if fill_kv:
            # fill only the slots in use, layer by layer, to bound host memory
            for layer_idx in cm.kv_transformer_layer_indices():
                n = cm.num_slots
                for kv in range(2):
                    chunk = 1 << 16
                    for s0 in range(0, n, chunk):
                        s1 = min(n, s0 + chunk)
                        shape = (s1 - s0, cm.num_kv_heads, cm.head_dim)
                        if gd is not None:
                            blk = torch.randn(shape, generator=gd, device=self.device) * kv_scale
                        else:
                            blk = torch.randn(shape, generator=g) * kv_scale
                        cm.kv_cache[kv, layer_idx, s0:s1].copy_(blk.to(torch.bfloat16))
        if paged and fill_kv:
            # min/max metadata of every complete page, as prefill would have left it
            from sparse_vllm_amd.kernels import quest_ops
            pages = []
            for s in self.seqs:
                row = cm.seq_id_to_row[s.seq_id]
                pages.extend(int(x) for x in cm.buffer_req_to_page_slots_cpu[row, : resident_len // cm.page_size])
            if pages:
                quest_ops.page_minmax(cm.kv_cache, cm.metadata_cache,
                                      torch.tensor(pages, dtype=torch.long, device=self.device), page_size=cm.page_size)
        if hasattr(cm, "h2o_score_tensor"):
            for layer_idx in cm.kv_transformer_layer_indices():
                for s in self.seqs:
                    sc0 = (torch.rand(resident_len, generator=gd, device=self.device) if gd is not None
                           else torch.rand(resident_len, generator=g).to(self.device))
                    cm.set_h2o_score(layer_idx, s.seq_id, sc0)
        return self.seqs

    def admit_compressed_rows(self, batch: int, total_len: int, *, seed: int = 0, kv_scale: float = 0.3):
        """DeltaKV: create `batch` sequences in the state a `total_len`-token prompt is in after prefill +
        compression (the reference's deltakv_evict / KIVI store) with synthetic payload:
        centres every int(1/cluster_ratio) tokens of each `recent`-sized evicted block
        (deltakv_base.py:255-275), K causal fathers per compressed token, random int4 / bf16 latents, and on
        KIVI full layers int4 blocks over [sink, quant_end) (deltakv_less_memory.py:3495-3520)."""
        cm, cfg, d = self.cache_manager, self.config, self.device
        lens = [int(total_len)] * batch if np.ndim(total_len) == 0 else [int(x) for x in total_len]
        assert len(lens) == batch
        gd = torch.Generator(device=d).manual_seed(seed)
        self.seqs = [Sequence(num_prompt_tokens=n) for n in lens]
        for s, n in zip(self.seqs, lens):
            s.num_prefilled_tokens = s.num_prompt_tokens
            self._admit_one_compressed_row(s, n, gd, kv_scale)
        return self.seqs

    def _admit_one_compressed_row(self, seq, total_len: int, gd, kv_scale: float):
        cm, cfg, d = self.cache_manager, self.config, self.device
        sink, recent = int(cfg.num_sink_tokens), int(cfg.num_recent_tokens)
        step = max(1, int(1.0 / max(1e-6, float(cfg.cluster_ratio))))
        buf = max(0, total_len - sink)
        clen = ((buf - recent) // recent) * recent if buf > recent else 0
        centers = np.concatenate([np.arange(s, min(s + recent, sink + clen), step)
                                  for s in range(sink, sink + clen, recent)] or [np.empty(0, np.int64)]).astype(np.int64)
        n_sink = min(sink, total_len)
        n_raw = n_sink + centers.size + (total_len - n_sink - clen)
        Ls, Lf, H, D = len(cm.deltakv_layer_ids), len(cm.full_layer_ids), cm.num_kv_heads, cm.head_dim
        K = int(cfg.deltakv_k_neighbors)
        rn = lambda *shape: (torch.randn(shape, generator=gd, device=d) * kv_scale).to(torch.bfloat16)
        ri = lambda *shape: torch.randint(-2 ** 31, 2 ** 31 - 1, shape, generator=gd, device=d, dtype=torch.int64).to(torch.int32)
        # causal father candidates of a compressed position: the sink tokens + the centres at positions <= it
        avail = n_sink + np.searchsorted(centers, np.arange(sink, sink + clen), side="right")
        avail_gpu = torch.from_numpy(np.maximum(avail, 1).astype(np.float32)).to(d)
        fidx = (torch.rand((clen, K), generator=gd, device=d) * avail_gpu[:, None]).long().clamp_max_(
            torch.from_numpy(np.maximum(avail, 1) - 1).to(d)[:, None]) if clen > 0 else torch.empty((0, K), dtype=torch.long, device=d)
        if int(cfg.kv_quant_bits or 0) == 4:
            W, g = cm.deltakv_latent_cache.shape[-1], cm.deltakv_latent_scales.shape[-1]
            sc = (torch.rand((Ls, clen, g), generator=gd, device=d) * 0.05 + 0.01).to(torch.bfloat16)
            latent = dict(code=ri(Ls, clen, W), scale=sc, mn=(sc.float() * -7.5).to(torch.bfloat16))
        else:
            latent = dict(dense=rn(Ls, clen, cm.deltakv_latent_cache.shape[-1]))
        qend, blocks = 0, None
        if cm._full_layer_kivi_enabled():
            G = cm._full_layer_kivi_group_size()
            qend = sink + (max(0, buf - int(cfg.full_layer_kivi_residual_length)) // G) * G
            if qend > n_sink:
                nb = (qend - n_sink) // G
                ks = torch.rand((Lf, nb, H, D), generator=gd, device=d) * 0.1 + 0.02
                vs = (torch.rand((Lf, nb, H, G, D // G), generator=gd, device=d) * 0.1 + 0.02).to(torch.bfloat16)
                blocks = dict(key_packed=ri(Lf, nb, H, D, G // 8), key_scales=ks, key_mins=ks * -7.5,
                              value_packed=ri(Lf, nb, H, G, D // 8), value_scales=vs,
                              value_mins=(vs.float() * -7.5).to(torch.bfloat16))
        n_full_raw = total_len - max(0, qend - n_sink)
        cm.admit_compressed_row(seq, total_len=total_len, compressed_len=clen, center_positions=centers,
                                father_center_index=fidx.cpu().numpy(), sparse_k_raw=rn(Ls, n_raw, H, D),
                                sparse_v=rn(Ls, n_raw, H, D), latent=latent, full_k=rn(Lf, n_full_raw, H, D),
                                full_v=rn(Lf, n_full_raw, H, D), kivi_quantized_end=qend, kivi_blocks=blocks)

    def random_step_inputs(self, seed: int = 1, scale: float = 0.3):
        """Per-layer q [L,B,Hq,D] and new-token k,v [L,B,Hkv,D] (bf16)."""
        cm = self.cache_manager
        g = torch.Generator(device="cpu").manual_seed(seed)
        B, L = int(self.graph_batch_size or len(self.seqs)), cm.num_layers
        mk = lambda h: (torch.randn((L, B, h, cm.head_dim), generator=g) * scale).to(torch.bfloat16).to(self.device)
        return mk(cm.num_heads), mk(cm.num_kv_heads), mk(cm.num_kv_heads)
