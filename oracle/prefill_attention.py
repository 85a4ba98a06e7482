"""Oracle: chunked-prefill causal attention over the paged slot table (TEST INFRASTRUCTURE ONLY).

Restates `_fwd_kernel` / `context_attention_fwd(attn_score=None)`, kernels/triton/context_flashattention_nopad.py:10-78,
242-276: per (sequence, q head, BLOCK_M query rows) a base-2 online softmax over key tiles of BLOCK_N, logits scaled by
D^-0.5 * log2(e), masked logits = -1e8, P cast to the value dtype before P.V (`p.to(v.dtype)`, :70), fp32 accumulation.
"""

from __future__ import annotations

import numpy as np

from .bf16 import bf16_round

LOG2E = np.float32(1.4426950408889634)


def context_attention_fwd(q, k, v, b_req_idx, b_start_loc, b_seq_len, b_prompt_cache_len, req_to_tokens, *,
                          block_m: int = 128, block_n: int = 128, p_dtype_bf16: bool = True):
    """q [tokens, Hq, D]; k, v [slots, Hkv, D] (f32 arrays holding bf16 values) -> o [tokens, Hq, D] f32."""
    T, Hq, D = q.shape
    Hkv = k.shape[1]
    G = Hq // Hkv
    sm_scale = np.float32(np.float32(1.0) / np.sqrt(np.float32(D)) * LOG2E)
    o = np.zeros((T, Hq, D), np.float32)
    for b in range(len(b_req_idx)):
        pc = int(b_prompt_cache_len[b])
        n_q = int(b_seq_len[b]) - pc
        start = int(b_start_loc[b])
        row = req_to_tokens[int(b_req_idx[b])]
        for h in range(Hq):
            kvh = h // G
            for m0 in range(0, n_q, block_m):
                rows = np.arange(m0, min(m0 + block_m, n_q))
                qq = q[start + rows, h].astype(np.float32)
                m_i = np.full(len(rows), -np.inf, np.float32)
                l_i = np.zeros(len(rows), np.float32)
                acc = np.zeros((len(rows), D), np.float32)
                end = min(m0 + block_m + pc, n_q + pc)
                for n0 in range(0, end, block_n):
                    cols = np.arange(n0, min(n0 + block_n, end))
                    kk = k[row[cols], kvh].astype(np.float32)
                    vv = v[row[cols], kvh].astype(np.float32)
                    qk = (qq @ kk.T).astype(np.float32)
                    mask = (rows[:, None] + pc) >= cols[None, :]
                    qk = np.where(mask, qk * sm_scale, np.float32(-1.0e8)).astype(np.float32)
                    m_ij = np.maximum(m_i, qk.max(axis=1))
                    p = np.exp2(qk - m_ij[:, None]).astype(np.float32)
                    alpha = np.exp2(m_i - m_ij).astype(np.float32)
                    l_i = l_i * alpha + p.sum(axis=1, dtype=np.float32)
                    acc = acc * alpha[:, None]
                    pp = bf16_round(p) if p_dtype_bf16 else p
                    acc = acc + (pp @ vv).astype(np.float32)
                    m_i = m_ij
                o[start + rows, h] = acc / l_i[:, None]
    return o


def context_attention_dense(q, k, v, b_req_idx, b_start_loc, b_seq_len, b_prompt_cache_len, req_to_tokens):
    """Un-tiled float64 ground truth."""
    T, Hq, D = q.shape
    G = Hq // k.shape[1]
    o = np.zeros((T, Hq, D), np.float64)
    for b in range(len(b_req_idx)):
        pc = int(b_prompt_cache_len[b])
        n_q = int(b_seq_len[b]) - pc
        start = int(b_start_loc[b])
        row = req_to_tokens[int(b_req_idx[b])]
        for h in range(Hq):
            kk = k[row[: n_q + pc], h // G].astype(np.float64)
            vv = v[row[: n_q + pc], h // G].astype(np.float64)
            s = q[start: start + n_q, h].astype(np.float64) @ kk.T / np.sqrt(D)
            s = np.where(np.arange(n_q)[:, None] + pc >= np.arange(n_q + pc)[None, :], s, -np.inf)
            s = s - s.max(axis=1, keepdims=True)
            p = np.exp(s)
            o[start: start + n_q, h] = (p / p.sum(axis=1, keepdims=True)) @ vv
    return o


def context_attention_scores(q, k, b_req_idx, b_start_loc, b_seq_len, b_prompt_cache_len, req_to_tokens, attn_score, *,
                             block_m: int = 128):
    """The score-collecting forms of context_attention_fwd (context_flashattention_nopad.py:82-240), in place on
    `attn_score`:
      3-D [B, Hq, L]: += sum over the chunk's query rows r with cache_len + r >= t of q[r, h] . k[t]   (:127-132, atomic_add)
      2-D [B, L]:     = max(old, max over heads and over the BLOCK_M-row query blocks of the block's masked sum / chunk_len)
                      (:205-212, atomic_max; BLOCK_M = 128, :246)
    for the keys t < b_seq_len[b]; raw logits (no D^-0.5), fp32."""
    Hq, D = q.shape[1], q.shape[2]
    G = Hq // k.shape[1]
    for b in range(len(b_seq_len)):
        pc, L = int(b_prompt_cache_len[b]), int(b_seq_len[b])
        chunk = L - pc
        if chunk <= 0:
            continue
        row = req_to_tokens[int(b_req_idx[b])]
        kk = k[row[:L].astype(np.int64)].astype(np.float32)                       # [L, Hkv, D]
        qq = q[int(b_start_loc[b]): int(b_start_loc[b]) + chunk].astype(np.float32)   # [chunk, Hq, D]
        kh = np.repeat(kk, G, axis=1)                                             # [L, Hq, D]
        qk = np.einsum("rhd,thd->hrt", qq, kh, dtype=np.float32)                  # [Hq, chunk, L]
        mask = (np.arange(chunk)[:, None] + pc) >= np.arange(L)[None, :]
        qk = np.where(mask[None], qk, np.float32(0))
        for r0 in range(0, chunk, block_m):
            r1 = min(r0 + block_m, chunk)
            end = min(r0 + block_m + pc, L)                                       # block_end_loc
            part = qk[:, r0:r1, :end].sum(axis=1, dtype=np.float32)               # [Hq, end]
            if attn_score.ndim == 3:
                attn_score[b, :, :end] += part
            else:
                attn_score[b, :end] = np.maximum(attn_score[b, :end], (part / np.float32(chunk)).max(axis=0))
    return attn_score
