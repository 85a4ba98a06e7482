"""Oracle: prefill token scores (TEST INFRASTRUCTURE ONLY).

Restates kernels/triton/prefill_score.py: `prefill_score_fwd` :432-670 with
  probability mode  _prefill_score_partial_stats_kernel :70-174, _reduce_stats :177-207,
                    _prefill_score_final_kernel :210-317
  logits mode       _prefill_logit_score_kernel :320-429
as one dense computation per score range (same masks, same normalisation set, same reductions).

For score range i (source row s = batch_indices[i] if given else i):
  queries   absolute positions p in [score_q_start[i], score_q_end[i]) whose chunk-relative
            position p - b_prompt_cache_len[s] lies in [0, chunk_len); row of `q` =
            b_start_loc[s] + (p - b_prompt_cache_len[s])
  keys      positions t in [candidate_start, max(candidate_start, context_len - num_recent)),
            physical slot = req_to_token[b_req_idx[s], t], causal t <= p
  probability: score[i,t] = max_h (1/max(end-start,1)) * sum_p softmax_t(q_{p,h}.k_t * D^-0.5)
               (softmax over the valid candidate keys of that query only), others stay 0
  logits:      score[i,t] = max_{p,h} q_{p,h}.k_t (raw), others stay -inf
"""

from __future__ import annotations

import numpy as np


def prefill_score_fwd(q, k, attn_score, b_req_idx, b_start_loc, b_seq_len, b_prompt_cache_len, max_query_len,
                      req_to_token_indexs, score_q_start, score_q_end, *, candidate_start=0, num_recent_tokens=0,
                      score_mode="probability", batch_indices=None):
    """q [tokens, Hq, D], k [slots, Hkv, D] (float32 holding bf16 values); attn_score [n, Lc] written in place."""
    score_mode = str(score_mode).strip().lower()
    if score_mode not in {"probability", "logits"}:
        raise ValueError(f"prefill score_mode must be 'probability' or 'logits', got {score_mode!r}.")
    n, Lc = attn_score.shape
    Hq, D = q.shape[1], q.shape[2]
    Hkv = k.shape[1]
    G = Hq // Hkv
    if score_q_end.shape != score_q_start.shape:
        raise ValueError("score_q_start and score_q_end must have the same shape")
    if int(max_query_len) <= 0 or Lc <= 0:
        return
    if score_mode == "probability" and max(16, 1 << (int(max_query_len) - 1).bit_length()) > 128:
        raise ValueError(f"probability prefill score query range is too large for this kernel: {int(max_query_len)} > 128")
    attn_score[...] = -np.inf if score_mode == "logits" else 0.0
    sm_scale = np.float32(float(D) ** -0.5)
    for i in range(n):
        s = int(batch_indices[i]) if batch_indices is not None else i
        start_loc = int(b_start_loc[s])
        cache_len = int(b_prompt_cache_len[s])
        ctx = int(b_seq_len[s])
        chunk = ctx - cache_len
        row = req_to_token_indexs[int(b_req_idx[s])]
        qs, qe = int(score_q_start[i]), int(score_q_end[i])
        # the kernel tiles at most max_query_len (prob: one block of pow2>=16) query positions from qs
        limit = int(max_query_len)
        pos = np.array([p for p in range(qs, min(qe, qs + max(limit, 0) if score_mode == "logits" else qs + max(16, 1 << (limit - 1).bit_length())))
                        if 0 <= p - cache_len < chunk], dtype=np.int64)
        cand_end = max(int(candidate_start), ctx - int(num_recent_tokens))
        t = np.arange(int(candidate_start), min(cand_end, Lc), dtype=np.int64)
        if pos.size == 0 or t.size == 0:
            continue
        qq = q[start_loc + (pos - cache_len)].astype(np.float32)            # [P, Hq, D]
        kk = k[row[t].astype(np.int64)].astype(np.float32)                  # [T, Hkv, D]
        qk = np.einsum("phgd,thd->hgpt", qq.reshape(len(pos), Hkv, G, D), kk, dtype=np.float32)
        valid = pos[:, None] >= t[None, :]                                   # causal
        if score_mode == "logits":
            sc = np.where(valid[None, None], qk, np.float32(-1.0e20)).max(axis=(0, 1, 2))
            has = valid.any(axis=0)
            attn_score[i, t[has]] = np.maximum(attn_score[i, t[has]], sc[has])
            # tokens with no valid query keep the kernel's atomic_max(-inf, -1e20) = -1e20
            attn_score[i, t[~has]] = np.maximum(attn_score[i, t[~has]], np.float32(-1.0e20))
            continue
        x = np.where(valid[None, None], qk * sm_scale, np.float32(-1.0e20))
        m = x.max(axis=3, keepdims=True)
        p_ = np.where(valid[None, None], np.exp(x - m, dtype=np.float32), np.float32(0))
        l = p_.sum(axis=3, keepdims=True, dtype=np.float32)
        probs = p_ / np.where(l > 0, l, np.float32(1))
        q_len = np.float32(max(qe - qs, 1))
        head_score = probs.sum(axis=2, dtype=np.float32) / q_len              # [Hkv, G, T]
        attn_score[i, t] = np.maximum(attn_score[i, t], head_score.max(axis=(0, 1)))
