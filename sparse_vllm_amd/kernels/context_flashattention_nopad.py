"""Chunked-prefill causal attention over the paged slot table on gfx950.

Mirror of the reference's kernels/triton/context_flashattention_nopad.py `context_attention_fwd` (:242-302): same
argument order and layout asserts, including the score-collecting forms (`attn_score` 3-D: per-head sums of the raw
logits over the chunk's query rows, :82-160; 2-D: head / query-block maximum of their means, :163-240) that the
reference's sparse controller hands to observation layers in prefill (sparse_controller.py:355-364).  H2O / SnapKV
prefill scores come from `prefill_score_fwd`.
"""

from __future__ import annotations

import ctypes as C

import torch

from .. import _lib

_SCORE_WS: dict = {}          # device -> fp32 workspace of the score-collecting forms (query suffix sums)


@torch.no_grad()
def context_attention_fwd(q, k, v, o, b_req_idx, b_start_loc, b_seq_len, b_prompt_cache_len, max_input_len,
                          req_to_token_indexs, attn_score=None, *, score_stats=None):
    """`score_stats` (MI355X extension) = (row_stats f32 [B * Hq * wpad], q_start int32 [B], wpad, score_rows f32 [B, cols] or
    None): the rows of the score window [q_start[b], b_seq_len[b]) leave their final softmax statistics in `row_stats`
    for `prefill_score_fwd(row_stats=...)`, and `score_rows` is zeroed - see include/svk.h."""
    Lq, Lk, Lv = q.shape[-1], k.shape[-1], v.shape[-1]
    assert Lq == Lk and Lk == Lv
    assert Lk in {16, 32, 64, 128, 256}
    assert q.dtype == k.dtype and k.dtype == v.dtype
    assert q.stride(-1) == 1 and k.stride(-1) == 1 and v.stride(-1) == 1 and o.stride(-1) == 1
    assert q.dtype == torch.bfloat16 and o.dtype == torch.bfloat16, f"the gfx950 prefill kernel computes in bf16, got {q.dtype}"
    assert k.stride() == v.stride()
    for t in (b_req_idx, b_start_loc, b_seq_len, b_prompt_cache_len, req_to_token_indexs):
        assert t.dtype == torch.int32
    assert req_to_token_indexs.stride(-1) == 1
    lib = _lib.load()
    a = _lib.SvkContextAttentionArgs(
        q=_lib.ptr(q), k_cache=_lib.ptr(k), v_cache=_lib.ptr(v), o=_lib.ptr(o), b_req_idx=_lib.ptr(b_req_idx),
        b_start_loc=_lib.ptr(b_start_loc), b_seq_len=_lib.ptr(b_seq_len), b_prompt_cache_len=_lib.ptr(b_prompt_cache_len),
        req_to_tokens=_lib.ptr(req_to_token_indexs), q_stride_t=q.stride(0), q_stride_h=q.stride(1),
        kv_slot_stride=k.stride(0), kv_head_stride=k.stride(1), o_stride_t=o.stride(0), o_stride_h=o.stride(1),
        req_stride=req_to_token_indexs.stride(0), batch=int(b_seq_len.shape[0]), num_q_heads=int(q.shape[1]),
        num_kv_heads=int(k.shape[1]), head_dim=int(Lk), max_input_len=int(max_input_len), kv_num_slots=int(k.shape[0]))
    if attn_score is not None:
        if attn_score.dim() not in (2, 3):
            raise ValueError(f"attn_score must be [batch, heads, len] or [batch, len], got {tuple(attn_score.shape)}.")
        assert attn_score.dtype == torch.float32 and attn_score.stride(-1) == 1, "score collection accumulates in fp32"
        assert attn_score.shape[0] >= b_seq_len.shape[0]
        if attn_score.dim() == 3:
            assert attn_score.shape[1] == q.shape[1]
        nbytes = int(lib.svk_context_attention_score_workspace_bytes(int(q.shape[0]), int(q.shape[1]), int(Lq)))
        ws = _SCORE_WS.get(q.device)
        if ws is None or ws.numel() * 4 < nbytes:
            ws = _SCORE_WS[q.device] = torch.empty(((nbytes + 3) // 4,), dtype=torch.float32, device=q.device)
        a.attn_score, a.score_workspace = _lib.ptr(attn_score), _lib.ptr(ws)
        a.attn_score_stride_b = attn_score.stride(0)
        a.attn_score_stride_h = attn_score.stride(1) if attn_score.dim() == 3 else 0
        a.attn_score_dim, a.attn_score_cols = int(attn_score.dim()), int(attn_score.shape[-1])
    if score_stats is not None:
        row_stats, q_start, wpad, score_rows = score_stats
        assert row_stats.dtype == torch.float32 and row_stats.is_contiguous() and q_start.dtype == torch.int32
        assert row_stats.numel() >= int(b_seq_len.shape[0]) * int(q.shape[1]) * int(wpad)
        a.score_row_stats, a.score_q_start, a.score_wpad = _lib.ptr(row_stats), _lib.ptr(q_start), int(wpad)
        if score_rows is not None:
            assert score_rows.dtype == torch.float32 and score_rows.stride(1) == 1 and score_rows.shape[0] >= b_seq_len.shape[0]
            a.score_clear, a.score_clear_stride, a.score_clear_cols = _lib.ptr(score_rows), score_rows.stride(0), int(score_rows.shape[1])
    _lib.check(lib.svk_context_attention_fwd(C.byref(a), _lib.current_stream_handle()), lib)
