"""Quest device ops (libsvk.so): page min/max metadata, page scoring, top-k packed view, paging.
include/svk.h cites the reference lines each one replaces (engine/cache_manager/quest.py)."""

from __future__ import annotations

import ctypes as C

import torch

from .. import _lib


def page_minmax(kv_cache: torch.Tensor, metadata_cache: torch.Tensor, page_slots: torch.Tensor, *, page_size: int,
                layers: slice | None = None):
    """metadata[0/1, l, p] = max/min over the page's tokens of K, for every layer in `layers`
    (default all) and every page in `page_slots` (int64)."""
    assert kv_cache.dim() == 5 and metadata_cache.dim() == 5 and kv_cache.dtype == torch.bfloat16
    assert page_slots.dtype == torch.long and page_slots.is_contiguous()
    k = kv_cache[0]
    meta = metadata_cache
    l0, l1 = (0, k.shape[0]) if layers is None else (layers.start, layers.stop)
    row = k.shape[2] * k.shape[3]
    assert k[0].is_contiguous() and meta[0, 0].is_contiguous()
    lib = _lib.load()
    a = _lib.SvkQuestPageMinmaxArgs(
        k_cache=_lib.ptr(k[l0]), metadata=_lib.ptr(meta[0, l0]), page_slots=_lib.ptr(page_slots),
        k_layer_stride=k.stride(0), meta_kind_stride=meta.stride(0), meta_layer_stride=meta.stride(1),
        n_pages=page_slots.numel(), n_layers=l1 - l0, page_size=int(page_size), row_elems=row)
    _lib.check(lib.svk_quest_page_minmax(C.byref(a), _lib.current_stream_handle()), lib)


def score_pages(q, page_max, page_min, page_table, req_indices, context_lens, page_scores, *, page_size: int,
                n_prev: int):
    assert q.dtype == torch.bfloat16 and q.stride(-1) == 1 and page_max.is_contiguous() and page_min.is_contiguous()
    assert page_table.dtype == torch.int32 and page_table.stride(1) == 1
    assert page_scores.dtype == torch.float32 and page_scores.stride(1) == 1 and page_scores.shape[1] >= n_prev
    lib = _lib.load()
    a = _lib.SvkQuestScorePagesArgs(
        q=_lib.ptr(q), page_max=_lib.ptr(page_max), page_min=_lib.ptr(page_min), page_table=_lib.ptr(page_table),
        req_indices=_lib.ptr(req_indices), context_lens=_lib.ptr(context_lens), page_scores=_lib.ptr(page_scores),
        q_stride_b=q.stride(0), q_stride_h=q.stride(1), page_table_stride=page_table.stride(0),
        score_stride=page_scores.stride(0), batch=q.shape[0], num_q_heads=q.shape[1],
        num_kv_heads=page_max.shape[1], head_dim=q.shape[2], page_size=int(page_size), n_prev=int(n_prev))
    _lib.check(lib.svk_quest_score_pages(C.byref(a), _lib.current_stream_handle()), lib)


def build_view(page_scores, page_table, token_table, req_indices, context_lens, packed_slots, local_lens, local_req, *,
               page_size: int, n_prev: int, prev_budget: int, token_budget: int, page_budget_base: int, max_keep: int,
               is_long_text: bool, emit_page_slots: bool = False):
    """`emit_page_slots` (MI355X): the view is written as page slots (`flash_decode_stage1(..., slot_page_size=page_size)`
    reads it)."""
    assert packed_slots.dtype == torch.int32 and packed_slots.stride(1) == 1
    lib = _lib.load()
    a = _lib.SvkQuestBuildViewArgs(
        page_scores=_lib.ptr(page_scores), page_table=_lib.ptr(page_table), token_table=_lib.ptr(token_table),
        req_indices=_lib.ptr(req_indices), context_lens=_lib.ptr(context_lens), packed_slots=_lib.ptr(packed_slots),
        local_lens=_lib.ptr(local_lens), local_req=_lib.ptr(local_req), score_stride=page_scores.stride(0),
        page_table_stride=page_table.stride(0), token_table_stride=token_table.stride(0),
        packed_stride=packed_slots.stride(0), batch=packed_slots.shape[0], page_size=int(page_size), n_prev=int(n_prev),
        prev_budget=int(prev_budget), token_budget=int(token_budget), page_budget_base=int(page_budget_base),
        max_keep=int(max_keep), is_long_text=int(bool(is_long_text)), emit_page_slots=int(bool(emit_page_slots)))
    _lib.check(lib.svk_quest_build_view(C.byref(a), _lib.current_stream_handle()), lib)


def decode_alloc(page_table, token_table, row_ids, cur_lens, new_page_slots, slot_mapping, context_lens, req_indices, *,
                 batch: int, page_size: int):
    lib = _lib.load()
    a = _lib.SvkQuestDecodeAllocArgs(
        page_table=_lib.ptr(page_table), token_table=_lib.ptr(token_table), row_ids=_lib.ptr(row_ids),
        cur_lens=_lib.ptr(cur_lens), new_page_slots=_lib.ptr(new_page_slots), slot_mapping=_lib.ptr(slot_mapping),
        context_lens=_lib.ptr(context_lens), req_indices=_lib.ptr(req_indices),
        page_table_stride=page_table.stride(0), token_table_stride=token_table.stride(0),
        batch=int(batch), graph_batch=slot_mapping.numel(), page_size=int(page_size))
    _lib.check(lib.svk_quest_decode_alloc(C.byref(a), _lib.current_stream_handle()), lib)


def device_step_args(page_table, token_table, row_len, free_pages, free_page_ptr, row_ids, slot_mapping, context_lens,
                     req_indices, kv_cache, metadata_cache, *, batch: int, page_size: int):
    """Arguments of svk_quest_device_step_begin / _end (device-resident row lengths and page stack, include/svk.h): built
    once per batch composition, every pointer in them is graph-stable."""
    assert page_table.dtype == torch.int32 and token_table.dtype == torch.int32 and page_table.stride(1) == 1
    assert row_len.dtype == torch.int32 and row_len.is_contiguous() and free_pages.dtype == torch.int32 and free_pages.is_contiguous()
    assert free_page_ptr.dtype == torch.int32 and free_page_ptr.numel() == 1 and row_ids.dtype == torch.int32
    assert kv_cache.dim() == 5 and metadata_cache.dim() == 5 and kv_cache.dtype == torch.bfloat16
    k, meta = kv_cache[0], metadata_cache
    assert k[0].is_contiguous() and meta[0, 0].is_contiguous()
    return _lib.SvkQuestDeviceStepArgs(
        page_table=_lib.ptr(page_table), token_table=_lib.ptr(token_table), row_len=_lib.ptr(row_len),
        free_pages=_lib.ptr(free_pages), free_page_ptr=_lib.ptr(free_page_ptr), row_ids=_lib.ptr(row_ids),
        slot_mapping=_lib.ptr(slot_mapping), context_lens=_lib.ptr(context_lens), req_indices=_lib.ptr(req_indices),
        k_cache=_lib.ptr(k[0]), metadata=_lib.ptr(meta[0, 0]), page_table_stride=page_table.stride(0),
        token_table_stride=token_table.stride(0), k_layer_stride=k.stride(0), meta_kind_stride=meta.stride(0),
        meta_layer_stride=meta.stride(1), batch=int(batch), graph_batch=int(slot_mapping.numel()), page_size=int(page_size),
        n_layers=int(k.shape[0]), row_elems=int(k.shape[2] * k.shape[3]))


def device_step_begin(args):
    lib = _lib.load()
    _lib.check(lib.svk_quest_device_step_begin(C.byref(args), _lib.current_stream_handle()), lib)


def device_step_end(args):
    lib = _lib.load()
    _lib.check(lib.svk_quest_device_step_end(C.byref(args), _lib.current_stream_handle()), lib)
