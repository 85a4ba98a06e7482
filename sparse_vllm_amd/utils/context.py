"""Per-forward context (mirror of utils/context.py:1-46)."""

from __future__ import annotations

from dataclasses import dataclass
from typing import Any

import torch


@dataclass
class Context:
    is_prefill: bool = False
    cu_seqlens_q: torch.Tensor | None = None
    cache_manager: Any = None
    sparse_controller: Any = None
    now_layer_idx: int = 0
    decode_mid_o: torch.Tensor | None = None
    decode_mid_o_logexpsum: torch.Tensor | None = None
    is_long_text: bool = False
    max_chunk_len: int | None = None       # host-known longest chunk of a prefill step (avoids a device sync)
    seqs: Any = None                       # the step's sequences (utils/context.py:11 `seqs`)


_CONTEXT = Context()


def get_context() -> Context:
    return _CONTEXT


def set_context(is_prefill: bool, cu_seqlens_q=None, cache_manager=None, sparse_controller=None, now_layer_idx: int = 0,
                is_long_text: bool = False) -> Context:
    global _CONTEXT
    keep = _CONTEXT
    _CONTEXT = Context(is_prefill, cu_seqlens_q, cache_manager, sparse_controller, now_layer_idx,
                       keep.decode_mid_o, keep.decode_mid_o_logexpsum, is_long_text)
    return _CONTEXT


def reset_context():
    global _CONTEXT
    _CONTEXT = Context()
