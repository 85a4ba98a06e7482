"""Decode stage 2 (LSE merge).  Mirror of kernels/triton/flash_decoding_stage2.py:49-81."""

from __future__ import annotations

import ctypes as C
import os

import torch

from .. import _lib


# Scratch + tickets of the two-level merge (include/svk.h `split_ws`, `split_tickets`: one launch at a time per pair).
#   * one pair per device: the reference's engine is one thread issuing a decode step's launches in order (SURVEY 8(b)
#     "Threading"), the eager first step sizes the pair and the capture of the next step finds it (a per-stream key would
#     send every captured launch to the one-level form: capture runs on torch's capture stream);
#   * a pair is NEVER released: a hipGraph captured earlier has the raw pointers baked into its kernel arguments, so a pair
#     that is outgrown is retired to `_RETIRED` (kept alive for the life of the process) instead of being returned to the
#     caching allocator where another tensor could be handed the same memory while a replay still writes to it;
#   * never allocated while a capture is running (that launch takes the one-level form);
#   * a caller that runs merges CONCURRENTLY on one device (several engines / graphs replayed on different streams at the
#     same time) passes its own `workspace=(scratch, tickets)` (`split_workspace(...)` below builds one) next to its other
#     per-context buffers.
_SPLIT_WS: dict = {}
_RETIRED: list = []


def split_workspace(batch: int, heads: int, head_dim: int, max_partials: int, device):
    """A caller-owned (scratch uint8, tickets int32 zeroed) pair for launches up to this shape, or None when the shape
    never takes the two-level merge."""
    need = int(_lib.load().svk_flash_decode_stage2_split_workspace_bytes(batch, heads, head_dim, max_partials))
    if need <= 0:
        return None
    return (torch.empty((need,), dtype=torch.uint8, device=device),
            torch.zeros((max(4096, batch * heads),), dtype=torch.int32, device=device))


def _split_workspace(lib, batch: int, heads: int, head_dim: int, max_partials: int, device):
    need = int(lib.svk_flash_decode_stage2_split_workspace_bytes(batch, heads, head_dim, max_partials))
    if need <= 0 or os.environ.get("SVK_STAGE2_SPLIT", "1") == "0":
        return None
    key = device.index
    held = _SPLIT_WS.get(key)
    if held is None or held[0].numel() < need or held[1].numel() < batch * heads:
        if torch.cuda.is_current_stream_capturing():
            return None
        if held is not None:
            _RETIRED.append(held)                 # an earlier capture may hold its pointers
        held = _SPLIT_WS[key] = (torch.empty((need,), dtype=torch.uint8, device=device),
                                 torch.zeros((max(4096, batch * heads),), dtype=torch.int32, device=device))
    return held


@torch.no_grad()
def flash_decode_stage2(mid_out, mid_out_logexpsum, B_Seqlen, O, block_seq, extra_partials: int = 0, workspace=None):
    """`extra_partials`: partial slots merged beyond ceil(len / block_seq) per row (the wide KIVI stage 1 puts the raw /
    ragged pieces of a row there, `full_layer_kivi_flash_decode_stage1(extra_partial_slots=...)`).  Launches that may
    merge more than 256 partials per row run the two-level merge (MI355X extension, `SVK_STAGE2_SPLIT=0`: one level);
    `workspace`: a caller-owned `split_workspace(...)` pair instead of the per-(device, stream) one."""
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
    ws = workspace
    if ws is None and mid_out.is_cuda:
        ws = _split_workspace(lib, batch, head_num, Lk, int(mid_out.shape[2]), mid_out.device)
    if ws is not None:
        a.split_ws, a.split_ws_bytes, a.split_tickets = _lib.ptr(ws[0]), int(ws[0].numel()), _lib.ptr(ws[1])
    _lib.check(lib.svk_flash_decode_stage2(C.byref(a), _lib.current_stream_handle()), lib)
