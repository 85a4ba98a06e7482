"""Decode stage 2 (LSE merge).  Mirror of kernels/triton/flash_decoding_stage2.py:49-81."""

from __future__ import annotations

import ctypes as C

import torch

from .. import _lib


@torch.no_grad()
def flash_decode_stage2(mid_out, mid_out_logexpsum, B_Seqlen, O, block_seq, extra_partials: int = 0):
    """`extra_partials`: partial slots merged beyond ceil(len / block_seq) per row (the wide KIVI stage 1 puts the raw /
    ragged pieces of a row there, `full_layer_kivi_flash_decode_stage1(extra_partial_slots=...)`)."""
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
    _lib.check(lib.svk_flash_decode_stage2(C.byref(a), _lib.current_stream_handle()), lib)
