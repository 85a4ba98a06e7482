"""Prefill token scores on gfx950.  Mirror of kernels/triton/prefill_score.py: `prefill_score_fwd`
(:432-670, same signature and validation) and `PrefillScoreWorkspace` (:6-67)."""

from __future__ import annotations

import ctypes as C

import torch

from .. import _lib


class PrefillScoreWorkspace:
    """Reusable probability-score workspace owned by one runtime worker."""

    def __init__(self) -> None:
        self._buf: torch.Tensor | None = None

    def reserve(self, nbytes: int, device: torch.device) -> torch.Tensor:
        elems = (int(nbytes) + 3) // 4
        if self._buf is None or self._buf.device != device or self._buf.numel() < elems:
            self._buf = torch.empty((max(elems, 1),), dtype=torch.float32, device=device)
        return self._buf


@torch.no_grad()
def prefill_score_fwd(q, k, attn_score, b_req_idx, b_start_loc, b_seq_len, b_prompt_cache_len, max_query_len,
                      req_to_token_indexs, score_q_start, score_q_end, *, candidate_start: int = 0,
                      num_recent_tokens: int = 0, score_mode: str = "probability",
                      workspace: PrefillScoreWorkspace | None = None, batch_indices: torch.Tensor | None = None,
                      row_stats: torch.Tensor | None = None):
    """`row_stats` (MI355X extension): the window rows' softmax statistics that `context_attention_fwd(score_stats=...)`
    of the same chunk left behind; the probability mode then runs its final pass only (attn_score must be the tensor that
    launch cleared)."""
    head_dim = q.shape[-1]
    assert k.shape[-1] == head_dim
    assert q.dtype == k.dtype
    assert q.stride(-1) == 1 and k.stride(-1) == 1
    assert attn_score.dim() == 2
    assert head_dim in {16, 32, 64, 128, 256}
    assert q.dtype == torch.bfloat16 and attn_score.dtype == torch.float32 and attn_score.stride(1) == 1
    batch, head = score_q_start.shape[0], q.shape[1]
    if score_q_end.shape != score_q_start.shape:
        raise ValueError("score_q_start and score_q_end must have the same shape, got "
                         f"{tuple(score_q_start.shape)} and {tuple(score_q_end.shape)}.")
    if int(attn_score.shape[0]) != int(batch):
        raise ValueError("attn_score must have one row per score range, got "
                         f"score_batch={batch} output_shape={tuple(attn_score.shape)}.")
    if batch_indices is not None:
        if batch_indices.shape != score_q_start.shape:
            raise ValueError("batch_indices must have one entry per score range, got "
                             f"{tuple(batch_indices.shape)} and {tuple(score_q_start.shape)}.")
        if batch_indices.dtype != torch.int32 or batch_indices.device != q.device:
            raise TypeError(f"batch_indices must be int32 on the query device, got {batch_indices.dtype} on {batch_indices.device}.")
    kv_head = k.shape[1]
    if head // kv_head <= 0 or head % kv_head != 0:
        raise ValueError(f"num query heads must be divisible by num kv heads: q={head} k={kv_head}")
    score_mode = str(score_mode).strip().lower()
    if score_mode not in {"probability", "logits"}:
        raise ValueError(f"prefill score_mode must be 'probability' or 'logits', got {score_mode!r}.")
    if int(max_query_len) <= 0 or int(attn_score.shape[1]) <= 0:
        return
    mode = _lib.SVK_PREFILL_SCORE_LOGITS if score_mode == "logits" else _lib.SVK_PREFILL_SCORE_PROBABILITY
    lib = _lib.load()
    ws = None
    if row_stats is not None:
        assert row_stats.dtype == torch.float32 and row_stats.is_contiguous()
    if mode == _lib.SVK_PREFILL_SCORE_PROBABILITY and row_stats is None:
        nbytes = lib.svk_prefill_score_workspace_bytes(batch, head, kv_head, int(max_query_len), int(attn_score.shape[1]))
        workspace = PrefillScoreWorkspace() if workspace is None else workspace
        ws = workspace.reserve(nbytes, q.device)
    a = _lib.SvkPrefillScoreArgs(
        q=_lib.ptr(q), k_cache=_lib.ptr(k), attn_score=_lib.ptr(attn_score), b_req_idx=_lib.ptr(b_req_idx),
        b_start_loc=_lib.ptr(b_start_loc), b_seq_len=_lib.ptr(b_seq_len), b_prompt_cache_len=_lib.ptr(b_prompt_cache_len),
        req_to_tokens=_lib.ptr(req_to_token_indexs), score_q_start=_lib.ptr(score_q_start),
        score_q_end=_lib.ptr(score_q_end), batch_indices=_lib.ptr(batch_indices), workspace=_lib.ptr(ws),
        q_stride_t=q.stride(0), q_stride_h=q.stride(1), kv_slot_stride=k.stride(0), kv_head_stride=k.stride(1),
        req_stride=req_to_token_indexs.stride(0), score_stride=attn_score.stride(0), n_ranges=batch,
        num_q_heads=head, num_kv_heads=kv_head, head_dim=head_dim, max_query_len=int(max_query_len),
        score_cols=int(attn_score.shape[1]), candidate_start=int(candidate_start),
        num_recent_tokens=int(num_recent_tokens), score_mode=mode, row_stats=_lib.ptr(row_stats))
    _lib.check(lib.svk_prefill_score(C.byref(a), _lib.current_stream_handle()), lib)


def prefill_score_window_pad(num_q_heads: int, num_kv_heads: int, max_query_len: int) -> int:
    """Columns per (sequence, head) of a `row_stats` tensor: the score window padded the way the kernel tiles it."""
    return int(_lib.load().svk_prefill_score_window_pad(int(num_q_heads), int(num_kv_heads), int(max_query_len)))
