"""Decode stage 2 (LSE merge).  Mirror of kernels/triton/flash_decoding_stage2.py:49-81."""

from __future__ import annotations

import ctypes as C
import os

import torch

from .. import _lib


# scratch of the two-level merge (include/svk.h `split_ws`), grow-only, one per device (the merges of a decode step are
# ordered on one stream): zero when allocated (the kernel's tickets), never allocated while a stream capture is running
# (that launch then takes the one-level form)
_SPLIT_WS: dict = {}


def _split_workspace(lib, batch: int, heads: int, head_dim: int, max_partials: int, device):
    need = int(lib.svk_flash_decode_stage2_split_workspace_bytes(batch, heads, head_dim, max_partials))
    if need <= 0 or os.environ.get("SVK_STAGE2_SPLIT", "1") == "0":
        return None
    key = device.index          # (not per stream: a step's launches are captured on one stream and replayed on another)
    held = _SPLIT_WS.get(key)
    if held is None or held[0].numel() < need or held[1].numel() < batch * heads:
        if torch.cuda.is_current_stream_capturing():
            return None
        scratch = held[0] if held is not None and held[0].numel() >= need else torch.empty((need,), dtype=torch.uint8, device=device)
        tickets = (held[1] if held is not None and held[1].numel() >= batch * heads
                   else torch.zeros((max(4096, batch * heads),), dtype=torch.int32, device=device))
        held = _SPLIT_WS[key] = (scratch, tickets)
    return held


@torch.no_grad()
def flash_decode_stage2(mid_out, mid_out_logexpsum, B_Seqlen, O, block_seq, extra_partials: int = 0):
    """`extra_partials`: partial slots merged beyond ceil(len / block_seq) per row (the wide KIVI stage 1 puts the raw /
    ragged pieces of a row there, `full_layer_kivi_flash_decode_stage1(extra_partial_slots=...)`).  Launches that may
    merge more than 256 partials per row run the two-level merge (MI355X extension, `SVK_STAGE2_SPLIT=0`: one level)."""
    Lk = mid_out.shape[-1]
    assert Lk in {16, 32, 64, 128, 256}
    assert B_Seqlen.stride(0) == 1, f"B_Seqlen must be contiguous, got stride={B_Seqlen.stride()}."
    assert mid_out.stride(-1) == 1 and mid_out_logexpsum.stride(-1) == 1 and O.stride(-1) == 1
    assert O.dtype == torch.bfloat16 and mid_out.dtype == torch.float32
    batch, head_num = mid_out.shape[0], mid_out.shape[1]
    lib = _lib.load()
    a = _lib.SvkFlashDecodeStage2Args(
        mid_o=_lib.ptr(mid_out), mid_lse=_lib.ptr(mid_out_logexpsum), b_seqlen=_lib.ptr(B_Seqlen), o=_lib.ptr(O),
        mid_o_stride_b=mid_out.stride(0), mid_o_stride_h=mid_out.stride(1), mid_o_stride_s=mid_out.stride(2),
        mid_lse_stride_b=mid_out_logexpsum.stride(0), mid_lse_stride_h=mid_out_logexpsum.stride(1),
        o_stride_b=O.stride(0), o_stride_h=O.stride(1),
        batch=batch, num_q_heads=head_num, head_dim=Lk, block_seq=int(block_seq), extra_partials=int(extra_partials),
        max_partials=int(mid_out.shape[2]))      # the caller's view of the workspace is this launch's partial count
    ws = _split_workspace(lib, batch, head_num, Lk, int(mid_out.shape[2]), mid_out.device) if mid_out.is_cuda else None
    if ws is not None:
        a.split_ws, a.split_ws_bytes, a.split_tickets = _lib.ptr(ws[0]), int(ws[0].numel()), _lib.ptr(ws[1])
    _lib.check(lib.svk_flash_decode_stage2(C.byref(a), _lib.current_stream_handle()), lib)
