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
