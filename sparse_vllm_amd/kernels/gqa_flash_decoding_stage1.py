"""GQA split-KV decode stage 1 (optionally with fused token scores) on gfx950.

Mirror of the reference's kernels/triton/gqa_flash_decoding_stage1.py:
`flash_decode_stage1` (:329-394) and `flash_decode_stage1_with_score` (:397-445), same
argument order, same layout asserts (:298-326).  `num_warps`/`num_stages`/`block_n` are
Triton launch knobs with no meaning for the HIP kernel; they are accepted and ignored so
existing call sites (layers/attention_backend.py:284-349) keep working.
"""

from __future__ import annotations

import ctypes as C
import os

import torch

from .. import _lib


def _assert_supported_layout(q, k, v, req_to_tokens, b_req_idx, b_seqlen, mid_out, mid_out_logsumexp):
    assert q.stride(-1) == 1, f"q head_dim must be contiguous, got stride={q.stride()}."
    assert k.stride(-1) == 1, f"k head_dim must be contiguous, got stride={k.stride()}."
    assert v.stride(-1) == 1, f"v head_dim must be contiguous, got stride={v.stride()}."
    assert k.stride() == v.stride(), (
        "k and v must have identical layouts because the GQA decode kernel shares their strides, "
        f"got k_stride={k.stride()} v_stride={v.stride()}.")
    assert req_to_tokens.stride(-1) == 1, (
        f"req_to_tokens sequence dimension must be contiguous, got stride={req_to_tokens.stride()}.")
    assert b_req_idx.stride(0) == 1, f"b_req_idx must be contiguous, got stride={b_req_idx.stride()}."
    assert b_seqlen.stride(0) == 1, f"b_seqlen must be contiguous, got stride={b_seqlen.stride()}."
    assert mid_out.stride(-1) == 1, f"mid_out head_dim must be contiguous, got stride={mid_out.stride()}."
    assert mid_out_logsumexp.stride(-1) == 1, (
        f"mid_out_logsumexp block dimension must be contiguous, got stride={mid_out_logsumexp.stride()}.")


def direct_out_supported(max_len_in_batch, block_seq) -> bool:
    """True when stage 1 may write the attention output itself (`direct_out=`): one block per sequence, default kernel."""
    return int(max_len_in_batch) <= int(block_seq) and os.environ.get("SVK_DECODE_DIRECT_OUT", "1") != "0"


def _stage1_args(q, k, v, Req_to_tokens, B_req_idx, B_Seqlen, max_len_in_batch, mid_out, mid_out_logsumexp,
                 attn_score, block_seq, new_kv=None, direct_out=None, score_overwrite=False, slot_page_size=0,
                 rotated_store=None):
    Lq, Lk = q.shape[-1], k.shape[-1]
    assert Lq == Lk
    assert Lk in {16, 32, 64, 128, 256}
    assert q.dtype == k.dtype and k.dtype == v.dtype
    assert q.dtype == torch.bfloat16, f"the gfx950 decode kernel computes in bf16, got {q.dtype}"
    assert int(block_seq) % 16 == 0
    _assert_supported_layout(q, k, v, Req_to_tokens, B_req_idx, B_Seqlen, mid_out, mid_out_logsumexp)
    assert Req_to_tokens.dtype == torch.int32 and B_req_idx.dtype == torch.int32 and B_Seqlen.dtype == torch.int32
    assert mid_out.dtype == torch.float32 and mid_out_logsumexp.dtype == torch.float32
    batch, kv_head_num = B_req_idx.shape[0], k.shape[1]
    nblk = (int(max_len_in_batch) + int(block_seq) - 1) // int(block_seq)
    assert mid_out.shape[2] >= nblk and mid_out_logsumexp.shape[2] >= nblk
    mode = _lib.SVK_SCORE_NONE
    ss_b = ss_h = 0
    if attn_score is not None:
        assert attn_score.dtype == torch.float32 and attn_score.stride(-1) == 1
        if attn_score.dim() == 3:
            mode = _lib.SVK_SCORE_PERHEAD
            ss_b, ss_h = attn_score.stride(0), attn_score.stride(1)
        else:
            mode = _lib.SVK_SCORE_HEADMAX
            ss_b = attn_score.stride(0)
    store = {}
    if new_kv is not None:
        # MI355X: this step's store_kvcache rides in the attention launch (include/svk.h, SvkFlashDecodeStage1Args)
        new_k, new_v, slot_mapping = new_kv
        assert new_k.dtype == k.dtype and new_v.dtype == v.dtype and new_k.shape == new_v.shape
        assert new_k.dim() == 3 and new_k.shape[0] == batch and new_k.shape[1] == kv_head_num and new_k.shape[2] == Lk
        assert new_k.stride(-1) == 1 and new_v.stride(-1) == 1 and new_k.stride() == new_v.stride()
        assert slot_mapping.dtype == torch.int32 and slot_mapping.is_contiguous() and slot_mapping.numel() >= batch
        store = dict(new_k=_lib.ptr(new_k), new_v=_lib.ptr(new_v), slot_mapping=_lib.ptr(slot_mapping),
                     new_stride_b=new_k.stride(0), new_stride_h=new_k.stride(1))
    if rotated_store is not None:
        # MI355X: the fused store in its rotated form (include/svk.h): k / v are an attention VIEW (rotated copy), the raw
        # rows go to the pre-RoPE cache `raw_k` / `raw_v` at slot_mapping[b], the view's newest row is written rotated
        assert new_kv is not None and attn_score is None and int(slot_page_size) == 0
        rs = rotated_store
        raw_k, raw_v, cos_sin = rs["raw_k"], rs["raw_v"], rs["cos_sin"]
        assert raw_k.dtype == torch.bfloat16 and raw_v.dtype == torch.bfloat16 and raw_k.stride() == raw_v.stride()
        assert raw_k.dim() == 3 and raw_k.shape[1] == kv_head_num and raw_k.shape[2] == Lk and raw_k.stride(2) == 1
        if cos_sin.dim() == 3:
            cos_sin = cos_sin[:, 0, :]
        assert cos_sin.dim() == 2 and cos_sin.shape[1] == Lk and cos_sin.stride(1) == 1
        pos = rs["slot_to_pos"]
        assert pos.dtype == torch.int32 and pos.is_contiguous() and pos.numel() >= raw_k.shape[0]
        kw = rs.get("k_norm_weight")
        if kw is not None:
            assert kw.dtype == torch.float32 and kw.is_contiguous() and kw.numel() == Lk
        row_lens = rs.get("row_lens")
        if row_lens is not None:
            assert row_lens.dtype == torch.int32 and row_lens.is_contiguous() and row_lens.numel() >= batch
            store.update(new_row_lens=_lib.ptr(row_lens))
        cos_dt = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[cos_sin.dtype]
        store.update(new_cos_sin=_lib.ptr(cos_sin), new_slot_to_pos=_lib.ptr(pos), new_k_norm_weight=_lib.ptr(kw),
                     raw_k_cache=_lib.ptr(raw_k), raw_v_cache=_lib.ptr(raw_v), raw_slot_stride=raw_k.stride(0),
                     raw_head_stride=raw_k.stride(1), new_cos_stride=cos_sin.stride(0), raw_num_slots=int(raw_k.shape[0]),
                     new_cos_dtype=cos_dt, new_k_norm_eps=float(rs.get("k_norm_eps", 1e-6)))
    if direct_out is not None:
        # MI355X: a single-block launch writes bf16(acc / l) itself; the stage-2 launch disappears (include/svk.h)
        assert direct_out.dtype == torch.bfloat16 and direct_out.shape == q.shape and direct_out.stride(-1) == 1
        store.update(direct_o=_lib.ptr(direct_out), direct_stride_b=direct_out.stride(0), direct_stride_h=direct_out.stride(1))
    return _lib.SvkFlashDecodeStage1Args(
        q=_lib.ptr(q), k_cache=_lib.ptr(k), v_cache=_lib.ptr(v), req_to_tokens=_lib.ptr(Req_to_tokens),
        b_req_idx=_lib.ptr(B_req_idx), b_seqlen=_lib.ptr(B_Seqlen), mid_o=_lib.ptr(mid_out),
        mid_lse=_lib.ptr(mid_out_logsumexp), attn_score=_lib.ptr(attn_score),
        q_stride_b=q.stride(0), q_stride_h=q.stride(1), kv_slot_stride=k.stride(0), kv_head_stride=k.stride(1),
        kv_num_slots=k.shape[0],
        req_stride=Req_to_tokens.stride(0),
        mid_o_stride_b=mid_out.stride(0), mid_o_stride_h=mid_out.stride(1), mid_o_stride_s=mid_out.stride(2),
        mid_lse_stride_b=mid_out_logsumexp.stride(0), mid_lse_stride_h=mid_out_logsumexp.stride(1),
        score_stride_b=ss_b, score_stride_h=ss_h,
        batch=batch, num_q_heads=q.shape[1], num_kv_heads=kv_head_num, head_dim=Lk,
        max_len_in_batch=int(max_len_in_batch), block_seq=int(block_seq), score_mode=mode,
        score_overwrite=int(bool(score_overwrite) and mode == _lib.SVK_SCORE_HEADMAX), slot_page_size=int(slot_page_size),
        **store)


def h2o_score_args(attn_score, scale, *, cum_score=None, b_req_idx=None, b_seqlen=None, b_new_slot=None, mask_by_len=False):
    """SvkH2oDecodeScoreArgs of one layer's score epilogue (what `h2o_ops.h2o_decode_score_update` would launch).
    `mask_by_len`: the raw scores were stored with `score_overwrite` into a buffer that was NOT pre-filled with -1e20."""
    assert not mask_by_len or b_seqlen is not None
    assert attn_score.dim() == 2 and attn_score.dtype == torch.float32 and attn_score.stride(1) == 1
    if cum_score is not None:
        assert cum_score.dim() == 2 and cum_score.dtype == torch.float32 and cum_score.stride(1) == 1
        assert b_req_idx is not None and b_seqlen is not None
    return _lib.SvkH2oDecodeScoreArgs(
        attn_score=_lib.ptr(attn_score), cum_score=_lib.ptr(cum_score), b_req_idx=_lib.ptr(b_req_idx),
        b_seqlen=_lib.ptr(b_seqlen), b_new_slot=_lib.ptr(b_new_slot), score_stride_b=attn_score.stride(0),
        cum_stride=0 if cum_score is None else cum_score.stride(0), scale=float(scale),
        batch=attn_score.shape[0], width=attn_score.shape[1], mask_by_len=int(bool(mask_by_len)))


def _launch(q, k, v, Req_to_tokens, B_req_idx, B_Seqlen, max_len_in_batch, mid_out, mid_out_logsumexp,
            attn_score, block_seq, new_kv=None, direct_out=None, score_overwrite=False, slot_page_size=0, rotated_store=None):
    a = _stage1_args(q, k, v, Req_to_tokens, B_req_idx, B_Seqlen, max_len_in_batch, mid_out, mid_out_logsumexp,
                     attn_score, block_seq, new_kv, direct_out, score_overwrite, slot_page_size, rotated_store)
    lib = _lib.load()
    _lib.check(lib.svk_flash_decode_stage1(C.byref(a), _lib.current_stream_handle()), lib)


@torch.no_grad()
def flash_decode_stage1(q, k, v, Req_to_tokens, B_req_idx, B_Seqlen, max_len_in_batch, mid_out,
                        mid_out_logsumexp, block_seq, block_n=16, num_warps=2, num_stages=2, *, new_kv=None, direct_out=None,
                        slot_page_size=0, rotated_store=None):
    """`new_kv=(new_k, new_v, slot_mapping)` (MI355X extension): store this step's K/V rows inside the launch.
    `direct_out` [B, Hq, D] bf16 (MI355X extension, see `direct_out_supported`): the launch writes the attention output
    itself and no stage 2 is needed.  `slot_page_size` (MI355X extension): Req_to_tokens holds page slots of that many
    tokens (include/svk.h).  `rotated_store` (MI355X extension, with `new_kv`): dict(raw_k, raw_v, slot_to_pos, cos_sin,
    k_norm_weight=None, k_norm_eps=1e-6, row_lens=None) - k / v are a rotated VIEW of the pre-RoPE cache raw_k / raw_v: the raw rows go to
    raw slot slot_mapping[b], the view's row of the newest position gets the k-normed, rotated key (include/svk.h)."""
    _launch(q, k, v, Req_to_tokens, B_req_idx, B_Seqlen, max_len_in_batch, mid_out, mid_out_logsumexp, None,
            block_seq, new_kv, direct_out=direct_out, slot_page_size=slot_page_size, rotated_store=rotated_store)


@torch.no_grad()
def flash_decode_stage1_with_score(q, k, v, Req_to_tokens, B_req_idx, B_Seqlen, max_len_in_batch, mid_out,
                                   mid_out_logsumexp, attn_score, block_seq, *, new_kv=None, direct_out=None,
                                   score_overwrite=False, slot_page_size=0):
    """2-D `attn_score` [B, W]: head-max raw logits fused; 3-D [B, Hq, W]: per head.  `new_kv` / `direct_out` /
    `slot_page_size`: as in `flash_decode_stage1`.  `score_overwrite` (MI355X extension, 2-D scores): store instead of
    max-combine - the caller skips the -1e20 pre-fill and masks positions beyond the lengths itself (include/svk.h)."""
    _launch(q, k, v, Req_to_tokens, B_req_idx, B_Seqlen, max_len_in_batch, mid_out, mid_out_logsumexp,
            attn_score, block_seq, new_kv, direct_out=direct_out, score_overwrite=score_overwrite,
            slot_page_size=slot_page_size)
