"""Scatter new K/V rows into the paged cache.  Mirror of kernels/triton/store_kvcache.py:33-71."""

from __future__ import annotations

import ctypes as C

import torch

from .. import _lib


def store_kvcache(key: torch.Tensor, value: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                  slot_mapping: torch.Tensor):
    n_tokens, num_heads, head_dim = key.shape
    d_model = num_heads * head_dim
    assert key.stride(-1) == 1 and value.stride(-1) == 1
    assert key.stride(1) == head_dim and value.stride(1) == head_dim
    assert k_cache.stride(-1) == 1
    assert slot_mapping.numel() == n_tokens
    assert key.dtype == torch.bfloat16 and k_cache.dtype == torch.bfloat16 and slot_mapping.dtype == torch.int32
    assert k_cache.is_contiguous() and v_cache.is_contiguous()
    lib = _lib.load()
    a = _lib.SvkStoreKvcacheArgs(
        key=_lib.ptr(key), value=_lib.ptr(value), k_cache=_lib.ptr(k_cache), v_cache=_lib.ptr(v_cache),
        slot_mapping=_lib.ptr(slot_mapping), key_stride=key.stride(0), value_stride=value.stride(0),
        n_tokens=n_tokens, row_elems=d_model)
    _lib.check(lib.svk_store_kvcache(C.byref(a), _lib.current_stream_handle()), lib)


_SLOT_STATUS: dict = {}
_SLOT_CHECK_KINDS = {1: "decode req row index out of bounds", 2: "decode visible length exceeds Req_to_tokens width",
                     3: "decode physical slot out of bounds before attention"}


def slot_check_status(device) -> torch.Tensor:
    """The device's [8] int32 status record of `check_slot_table_async` (zero = clean), created on first use."""
    dev = torch.device(device)
    st = _SLOT_STATUS.get(dev.index)
    if st is None:
        st = _SLOT_STATUS[dev.index] = torch.zeros((8,), dtype=torch.int32, device=dev)
    return st


@torch.no_grad()
def check_slot_table_async(slot_table, req_indices, context_lens, *, slot_cap: int, slot_page_size: int = 0, status=None):
    """MI355X: the reference's decode bounds check (layers/attention_backend.py:397-439) as a launch - request rows inside
    the table, visible length inside its width, every visible slot inside the KV pool - that never synchronises and may be
    captured into the step's hipGraph.  The first violation since the status was cleared stays recorded;
    `raise_if_slot_check_failed` reads it (a synchronisation: call it where the caller synchronises anyway)."""
    assert slot_table.dim() == 2 and slot_table.dtype == torch.int32 and slot_table.stride(1) == 1
    assert req_indices.dtype == torch.int32 and context_lens.dtype == torch.int32
    st = slot_check_status(slot_table.device) if status is None else status
    lib = _lib.load()
    a = _lib.SvkCheckSlotTableArgs(slot_table=_lib.ptr(slot_table), req_indices=_lib.ptr(req_indices), context_lens=_lib.ptr(context_lens),
                                   status=_lib.ptr(st), table_stride=slot_table.stride(0), batch=int(req_indices.numel()),
                                   num_rows=int(slot_table.shape[0]), width=int(slot_table.shape[1]), slot_cap=int(slot_cap),
                                   slot_page_size=int(slot_page_size))
    _lib.check(lib.svk_check_slot_table(C.byref(a), _lib.current_stream_handle()), lib)
    return st


def raise_if_slot_check_failed(device=None, *, status=None, clear: bool = True) -> None:
    """Read the status record (synchronises); a recorded violation raises the reference's RuntimeError and clears the record."""
    st = slot_check_status(device) if status is None else status
    rec = [int(x) for x in st.cpu().tolist()]
    if rec[0] == 0:
        return
    if clear:
        st.zero_()
    kind, b, row, pos, slot, length = rec[:6]
    raise RuntimeError(f"{_SLOT_CHECK_KINDS.get(kind, 'decode slot table check failed')}: batch={b} req_row={row} pos={pos} "
                       f"slot={slot} context_len={length}")
