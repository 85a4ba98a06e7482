"""H2O / SnapKV-family bookkeeping ops on device (libsvk.so): score normalisation and
accumulation, exact selection, slot-table compaction, decode slot allocation, slot copies.
Each function cites the reference code it replaces (see include/svk.h for line numbers)."""

from __future__ import annotations

import ctypes as C

import torch

from .. import _lib


def fill_f32(t: torch.Tensor, value: float):
    """`view.fill_(-1e20)` of SparseController._get_h2o_decode_score_buffer (sparse_controller.py:460)."""
    assert t.dtype == torch.float32 and t.is_contiguous()
    lib = _lib.load()
    _lib.check(lib.svk_fill_f32(_lib.ptr(t), t.numel(), float(value), _lib.current_stream_handle()), lib)


def h2o_decode_score_update(attn_score: torch.Tensor, scale: float, *, cum_score: torch.Tensor | None = None,
                            b_req_idx: torch.Tensor | None = None, b_seqlen: torch.Tensor | None = None,
                            b_new_slot: torch.Tensor | None = None):
    """In place `attn_score.mul_(scale); softmax(attn_score, -1, out=attn_score)`
    (sparse_controller.py:762-767) and, when `cum_score` is given, the cumulative update of
    H2OCacheManager.update_decode_attention_scores_all_layers (h2o.py:957-1038) on the
    persistent [rows, cap] score rows.  `b_new_slot` = the step's slot_mapping: lanes holding -1 (padded
    hipGraph lanes of prepare_decode_static, which mirror lane 0's row) are normalised but not accumulated,
    like the reference's `normalized[:, :len(seqs)]` (sparse_controller.py:1226-1282)."""
    assert attn_score.dim() == 2 and attn_score.dtype == torch.float32 and attn_score.stride(1) == 1
    lib = _lib.load()
    a = _lib.SvkH2oDecodeScoreArgs(
        attn_score=_lib.ptr(attn_score), cum_score=_lib.ptr(cum_score), b_req_idx=_lib.ptr(b_req_idx),
        b_seqlen=_lib.ptr(b_seqlen), b_new_slot=_lib.ptr(b_new_slot), score_stride_b=attn_score.stride(0),
        cum_stride=0 if cum_score is None else cum_score.stride(0), scale=float(scale),
        batch=attn_score.shape[0], width=attn_score.shape[1])
    if b_new_slot is not None:
        assert b_new_slot.dtype == torch.int32 and b_new_slot.numel() >= attn_score.shape[0] and b_new_slot.stride(-1) == 1
    if cum_score is not None:
        assert cum_score.dim() == 2 and cum_score.dtype == torch.float32 and cum_score.stride(1) == 1
        assert b_req_idx is not None and b_seqlen is not None
    _lib.check(lib.svk_h2o_decode_score_update(C.byref(a), _lib.current_stream_handle()), lib)


SCORE_LAYERS_LAUNCHES = {"batched": 0, "per_layer": 0}     # which form h2o_decode_score_update_layers took (tests)


def h2o_decode_score_update_layers(pending) -> None:
    """`h2o_decode_score_update` of several layers of one decode step (H2OCacheManager.
    update_decode_attention_scores_all_layers, h2o.py:957-1038).  `pending` = the layers' SvkH2oDecodeScoreArgs in
    layer order.  When the layers' buffers are equally spaced slices of one tensor each (the controller's [L, B, W]
    raw-score scratch, the manager's [L, rows, cap] cumulative tensor, the [L, B] slot mapping / row ids / lengths),
    they run as ONE launch (svk_h2o_decode_score_update_layers); otherwise one launch per layer."""
    if not pending:
        return
    lib = _lib.load()
    stream = _lib.current_stream_handle()
    first = pending[0]

    def delta(name, esize):
        a0 = getattr(first, name)
        if not a0:
            return 0 if all(not getattr(x, name) for x in pending) else None
        if len(pending) == 1:
            return 0
        d = getattr(pending[1], name) - a0
        if d % esize or any(getattr(x, name) != a0 + i * d for i, x in enumerate(pending)):
            return None
        return d // esize

    same = all(x.batch == first.batch and x.width == first.width and x.score_stride_b == first.score_stride_b and
               x.cum_stride == first.cum_stride and x.scale == first.scale and x.mask_by_len == first.mask_by_len
               for x in pending)
    strides = [delta(name, 4) for name in ("attn_score", "cum_score", "b_new_slot", "b_req_idx", "b_seqlen")]
    if same and None not in strides:
        _lib.check(lib.svk_h2o_decode_score_update_layers(C.byref(first), len(pending), *strides, stream), lib)
        SCORE_LAYERS_LAUNCHES["batched"] += 1
        return
    for a in pending:
        _lib.check(lib.svk_h2o_decode_score_update(C.byref(a), stream), lib)
    SCORE_LAYERS_LAUNCHES["per_layer"] += 1


def h2o_recent_count(budget: int, recent_ratio: float, kv_len: int) -> int:
    """h2o.py:495-496 / :540-541."""
    rc = max(1, int(int(budget) * float(recent_ratio)))
    return min(rc, int(budget), int(kv_len))


def select_h2o_indices_batch(scores: torch.Tensor, *, budget: int, recent_ratio: float,
                             out: torch.Tensor | None = None) -> torch.Tensor:
    """Device restatement of H2OCacheManager.select_h2o_indices_batch (h2o.py:518-563):
    same validation, same result (ascending int64 indices), bit-exact incl. ties."""
    if scores.dim() != 2:
        raise ValueError(f"Batched H2O scores must have shape [batch, kv_len], got {tuple(scores.shape)}.")
    kv_len = int(scores.shape[-1])
    budget = int(budget)
    if budget <= 0:
        raise ValueError(f"H2O budget must be positive, got {budget}.")
    if not 0.0 < float(recent_ratio) < 1.0:
        raise ValueError(f"H2O recent_ratio must be in (0, 1), got {recent_ratio}.")
    assert scores.dtype == torch.float32 and scores.stride(1) == 1
    rows = int(scores.shape[0])
    keep_len = min(kv_len, budget)
    if out is None:
        out = torch.empty((rows, keep_len), dtype=torch.long, device=scores.device)
    assert out.dtype == torch.long and out.stride(1) == 1 and tuple(out.shape) == (rows, keep_len)
    lib = _lib.load()
    a = _lib.SvkH2oSelectArgs(
        scores=_lib.ptr(scores), keep=_lib.ptr(out), score_stride=scores.stride(0), keep_stride=out.stride(0),
        rows=rows, kv_len=kv_len, budget=budget, recent_count=h2o_recent_count(budget, recent_ratio, kv_len))
    _lib.check(lib.svk_h2o_select_indices(C.byref(a), _lib.current_stream_handle()), lib)
    return out


def select_prefix_topk_suffix(scores: torch.Tensor, *, kv_len: int, prefix: int, topk: int, suffix: int,
                              out: torch.Tensor | None = None) -> torch.Tensor:
    """[0,prefix) ++ top-`topk` of scores[:, prefix:kv_len-suffix] ++ [kv_len-suffix, kv_len), ascending int64."""
    assert scores.dim() == 2 and scores.dtype == torch.float32 and scores.stride(1) == 1 and scores.shape[1] >= kv_len
    rows = scores.shape[0]
    n = prefix + topk + suffix
    if out is None:
        out = torch.empty((rows, n), dtype=torch.long, device=scores.device)
    assert out.dtype == torch.long and tuple(out.shape) == (rows, n) and out.stride(1) == 1
    lib = _lib.load()
    a = _lib.SvkSelectTopkArgs(scores=_lib.ptr(scores), keep=_lib.ptr(out), score_stride=scores.stride(0),
                               keep_stride=out.stride(0), rows=rows, kv_len=int(kv_len), prefix=int(prefix),
                               topk=int(topk), suffix=int(suffix))
    _lib.check(lib.svk_select_prefix_topk_suffix(C.byref(a), _lib.current_stream_handle()), lib)
    return out


def compact_rows(slot_table: torch.Tensor, free_stack: torch.Tensor, keep: torch.Tensor, layer_ids: torch.Tensor,
                 row_ids: torch.Tensor, free_base: torch.Tensor, *, cur_len: int,
                 row_payload: torch.Tensor | None = None):
    """SnapKVCacheManager.free_part_slots_batch_layers device half (snapkv.py:1681-1803) for
    uniform-length rows; optional f32 payload rows (H2O cumulative scores) gathered alike."""
    assert slot_table.dim() == 3 and slot_table.dtype == torch.int32 and slot_table.stride(2) == 1
    assert free_stack.dim() == 2 and free_stack.dtype == torch.int32 and free_stack.stride(1) == 1
    assert keep.dim() == 3 and keep.dtype == torch.long and keep.is_contiguous()
    n_layers, n_lanes, keep_len = keep.shape
    assert layer_ids.dtype == torch.int32 and layer_ids.numel() == n_layers
    assert row_ids.dtype == torch.int32 and tuple(row_ids.shape) == (n_layers, n_lanes) and row_ids.is_contiguous()
    assert free_base.dtype == torch.long and free_base.numel() == n_layers
    if row_payload is not None:
        assert row_payload.dim() == 3 and row_payload.dtype == torch.float32 and row_payload.stride(2) == 1
    lib = _lib.load()
    a = _lib.SvkCompactRowsArgs(
        slot_table=_lib.ptr(slot_table), free_stack=_lib.ptr(free_stack), row_payload=_lib.ptr(row_payload),
        keep=_lib.ptr(keep), layer_ids=_lib.ptr(layer_ids), row_ids=_lib.ptr(row_ids), free_base=_lib.ptr(free_base),
        table_stride_layer=slot_table.stride(0), table_stride_row=slot_table.stride(1),
        stack_stride=free_stack.stride(0),
        payload_stride_layer=0 if row_payload is None else row_payload.stride(0),
        payload_stride_row=0 if row_payload is None else row_payload.stride(1),
        n_layers=n_layers, n_lanes=n_lanes, cur_len=int(cur_len), keep_len=keep_len)
    _lib.check(lib.svk_compact_rows(C.byref(a), _lib.current_stream_handle()), lib)


def decode_alloc_slots(slot_table, free_stack, layer_ids, row_ids, cur_lens, slot_mapping, context_lens,
                       req_indices, *, free_ptr: int, batch: int, free_ptrs: torch.Tensor | None = None):
    """Device half of H2OCacheManager.prepare_decode_static (h2o.py:386-437).  `row_ids` / `cur_lens` are [B]
    (uniform rows) or [n_layers, B] together with int64 `free_ptrs` [n_layers]: the per-layer branch of
    SnapKVCacheManager._prepare_decode (snapkv.py:2656-2673)."""
    assert slot_mapping.dim() == 2 and slot_mapping.dtype == torch.int32 and slot_mapping.stride(1) == 1
    assert context_lens.stride() == slot_mapping.stride() and req_indices.stride() == slot_mapping.stride()
    assert row_ids.dtype == torch.int32 and cur_lens.dtype == torch.int32 and row_ids.shape == cur_lens.shape
    assert row_ids.is_contiguous() and cur_lens.is_contiguous()
    n_layers = layer_ids.numel()
    meta_stride = 0
    if row_ids.dim() == 2:
        assert row_ids.shape[0] == n_layers and row_ids.shape[1] >= int(batch)
        meta_stride = row_ids.stride(0)
    if free_ptrs is not None:
        assert free_ptrs.dtype == torch.long and free_ptrs.numel() == n_layers and free_ptrs.is_contiguous()
    lib = _lib.load()
    a = _lib.SvkDecodeAllocArgs(
        slot_table=_lib.ptr(slot_table), free_stack=_lib.ptr(free_stack), layer_ids=_lib.ptr(layer_ids),
        row_ids=_lib.ptr(row_ids), cur_lens=_lib.ptr(cur_lens), free_ptrs=_lib.ptr(free_ptrs),
        slot_mapping=_lib.ptr(slot_mapping),
        context_lens=_lib.ptr(context_lens), req_indices=_lib.ptr(req_indices),
        table_stride_layer=slot_table.stride(0), table_stride_row=slot_table.stride(1),
        stack_stride=free_stack.stride(0), out_stride=slot_mapping.stride(0), meta_stride_layer=meta_stride,
        free_ptr=int(free_ptr), n_layers=n_layers, batch=int(batch), graph_batch=slot_mapping.shape[1])
    _lib.check(lib.svk_decode_alloc_slots(C.byref(a), _lib.current_stream_handle()), lib)


def h2o_device_step_args(slot_table, free_stack, scores, row_len, free_ptr, row_ids, slot_mapping, context_lens, req_indices,
                         keep, *, batch: int, budget: int, recent_count: int, trigger_len: int, select_mode: int = 0,
                         prefix_count: int = 0, tickets=None):
    """Arguments of svk_h2o_device_step_begin / svk_h2o_device_burst (device-resident row lengths and free-stack
    pointers, include/svk.h): built once per batch composition, every pointer in them is graph-stable.
    `select_mode` 0 = H2O heavy hitters over `scores` [L, rows, cap]; 1 = sink + recent window (`scores` may be None);
    2 = SnapKV sink ++ top-k ++ recent over this step's lane-indexed `scores` [L, lanes, width] (`prefix_count` = sink).
    `tickets` int32 [L] (zeroed once): the burst runs as one launch instead of three."""
    assert tickets is None or (tickets.dtype == torch.int32 and tickets.numel() >= slot_table.shape[0])
    assert slot_table.dim() == 3 and slot_table.dtype == torch.int32 and slot_table.stride(2) == 1
    assert free_stack.dim() == 2 and free_stack.dtype == torch.int32 and free_stack.stride(1) == 1
    assert int(select_mode) in (0, 1, 2) and (scores is not None or int(select_mode) == 1)
    assert scores is None or (scores.dim() == 3 and scores.dtype == torch.float32 and scores.stride(2) == 1)
    assert row_len.dim() == 2 and row_len.dtype == torch.int32 and row_len.is_contiguous()
    assert free_ptr.dtype == torch.long and free_ptr.numel() == slot_table.shape[0]
    assert row_ids.dtype == torch.int32 and row_ids.numel() >= int(batch)
    assert slot_mapping.dim() == 2 and slot_mapping.dtype == torch.int32 and slot_mapping.stride(1) == 1
    assert context_lens.stride() == slot_mapping.stride() and req_indices.stride() == slot_mapping.stride()
    assert keep.dtype == torch.long and keep.is_contiguous() and keep.numel() >= slot_table.shape[0] * int(batch) * int(budget)
    return _lib.SvkH2oDeviceStepArgs(
        slot_table=_lib.ptr(slot_table), free_stack=_lib.ptr(free_stack), scores=_lib.ptr(scores), row_len=_lib.ptr(row_len),
        free_ptr=_lib.ptr(free_ptr), row_ids=_lib.ptr(row_ids), slot_mapping=_lib.ptr(slot_mapping),
        context_lens=_lib.ptr(context_lens), req_indices=_lib.ptr(req_indices), keep=_lib.ptr(keep),
        table_stride_layer=slot_table.stride(0), table_stride_row=slot_table.stride(1), stack_stride=free_stack.stride(0),
        score_stride_layer=0 if scores is None else scores.stride(0), score_stride_row=0 if scores is None else scores.stride(1),
        out_stride=slot_mapping.stride(0),
        n_layers=int(slot_table.shape[0]), rows_total=int(row_len.shape[1]), batch=int(batch),
        graph_batch=int(slot_mapping.shape[1]), budget=int(budget), recent_count=int(recent_count),
        trigger_len=int(trigger_len), select_mode=int(select_mode), prefix_count=int(prefix_count), tickets=_lib.ptr(tickets))


def h2o_device_step_begin(args):
    lib = _lib.load()
    _lib.check(lib.svk_h2o_device_step_begin(C.byref(args), _lib.current_stream_handle()), lib)


def h2o_device_burst(args):
    lib = _lib.load()
    _lib.check(lib.svk_h2o_device_burst(C.byref(args), _lib.current_stream_handle()), lib)


def copy_slots(k_cache, v_cache, src_slots, dst_slots, workspace):
    """K/V move of H2O final-prefill dense compaction (h2o.py:1296-1329): gather all
    sources into `workspace` [2, n, Hkv, D], then scatter -> overlap safe."""
    assert k_cache.is_contiguous() and v_cache.is_contiguous() and k_cache.dtype == torch.bfloat16
    assert src_slots.dtype == torch.long and dst_slots.dtype == torch.long
    n = src_slots.numel()
    row = k_cache.shape[1] * k_cache.shape[2]
    assert workspace.is_contiguous() and workspace.numel() >= 2 * n * row and workspace.dtype == torch.bfloat16
    lib = _lib.load()
    a = _lib.SvkCopySlotsArgs(k_cache=_lib.ptr(k_cache), v_cache=_lib.ptr(v_cache), src_slots=_lib.ptr(src_slots),
                              dst_slots=_lib.ptr(dst_slots), workspace=_lib.ptr(workspace), n=n, row_elems=row)
    _lib.check(lib.svk_copy_slots(C.byref(a), _lib.current_stream_handle()), lib)
