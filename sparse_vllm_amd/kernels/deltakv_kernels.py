"""DeltaKV decode-side ops on gfx950.  Mirrors (names, keyword arguments, validation) of
kernels/triton/deltakv_kernels.py `deltakv_static_decode_plan` (:3854-3942),
`deltakv_reconstruct_writeback_grouped_heads` (:2909-3012),
`deltakv_less_memory_reconstruct_writeback_quantized` / `_int4` (:3344-3485) and of
kernels/triton/quant.py `triton_dequantize_2d_int4_grouped` (:160-216), `unpack_quantized_to_16bit` (:326-349)."""

from __future__ import annotations

import ctypes as C

import torch

from .. import _lib

_DT = {torch.float32: _lib.SVK_DTYPE_F32, torch.bfloat16: _lib.SVK_DTYPE_BF16, torch.float16: _lib.SVK_DTYPE_F16}


def _dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError as e:
        raise TypeError(f"unsupported dtype {t.dtype}") from e


@torch.no_grad()
def deltakv_static_decode_plan(*, raw_slots_map, latent_slots_map, active_compressed_indices, req_indices, context_lens,
                               compressed_lens, temp_slots, active_slots_out, active_pos_out, new_context_lens_out,
                               recon_pos_out, recon_latent_out, recon_out_slot_out, sink: int, max_buffer: int):
    assert raw_slots_map.dim() == 2 and latent_slots_map.dim() == 2
    assert active_compressed_indices.dim() == 2 and temp_slots.dim() == 2
    assert req_indices.dim() == 1 and context_lens.dim() == 1 and compressed_lens.dim() == 1
    assert active_slots_out.dim() == 2 and active_pos_out.dim() == 2
    for t in (raw_slots_map, latent_slots_map, active_compressed_indices, req_indices, context_lens, compressed_lens,
              temp_slots, active_slots_out, active_pos_out, new_context_lens_out, recon_pos_out, recon_latent_out,
              recon_out_slot_out):
        assert t.dtype == torch.int32 and t.stride(-1) == 1
    batch_size = int(req_indices.shape[0])
    k_max = int(active_compressed_indices.shape[1])
    sink, max_buffer = int(sink), int(max_buffer)
    max_s = sink + k_max + max_buffer
    if active_slots_out.shape != (batch_size, max_s):
        raise ValueError("DeltaKV static decode plan active_slots_out shape mismatch: "
                         f"got={tuple(active_slots_out.shape)}, expected=({batch_size}, {max_s}).")
    if active_pos_out.shape != (batch_size, max_s):
        raise ValueError("DeltaKV static decode plan active_pos_out shape mismatch: "
                         f"got={tuple(active_pos_out.shape)}, expected=({batch_size}, {max_s}).")
    if temp_slots.shape != (batch_size, k_max):
        raise ValueError("DeltaKV static decode plan temp_slots shape mismatch: "
                         f"got={tuple(temp_slots.shape)}, expected=({batch_size}, {k_max}).")
    if recon_pos_out.numel() != batch_size * k_max:
        raise ValueError("DeltaKV static decode plan recon outputs must have B*K elements.")
    if batch_size == 0:
        return
    lib = _lib.load()
    a = _lib.SvkDeltakvPlanArgs(
        raw_slots_map=_lib.ptr(raw_slots_map), latent_slots_map=_lib.ptr(latent_slots_map),
        active_compressed=_lib.ptr(active_compressed_indices), req_indices=_lib.ptr(req_indices),
        context_lens=_lib.ptr(context_lens), compressed_lens=_lib.ptr(compressed_lens), temp_slots=_lib.ptr(temp_slots),
        active_slots_out=_lib.ptr(active_slots_out), active_pos_out=_lib.ptr(active_pos_out),
        new_context_lens_out=_lib.ptr(new_context_lens_out), recon_pos_out=_lib.ptr(recon_pos_out),
        recon_latent_out=_lib.ptr(recon_latent_out), recon_out_slot_out=_lib.ptr(recon_out_slot_out),
        raw_stride=raw_slots_map.stride(0), latent_stride=latent_slots_map.stride(0),
        active_stride=active_compressed_indices.stride(0), temp_stride=temp_slots.stride(0),
        out_stride=active_slots_out.stride(0), pos_stride=active_pos_out.stride(0), batch=batch_size, k_max=k_max,
        sink=sink, max_buffer=max_buffer, max_positions=int(raw_slots_map.shape[1]))
    _lib.check(lib.svk_deltakv_static_decode_plan(C.byref(a), _lib.current_stream_handle()), lib)


def _reconstruct(*, delta, scale, mn, latent_slots, father_slots, slot_to_pos, out_slots, out_pos, cos_sin, k_cache,
                 v_cache, bits, group_size, k_norm_weight, k_norm_eps, raw_k_cache, store_raw_k, father_index=None, batch=None,
                 view_out=None, up=None):
    assert k_cache.dtype == torch.bfloat16 and k_cache.stride() == v_cache.stride() and k_cache.stride(-1) == 1
    assert father_slots.dim() == 2 and father_slots.dtype == torch.int32 and father_slots.stride(1) == 1
    if father_index is not None:      # father_slots is the [latents, K] table, indexed in-kernel
        assert father_index.dtype == torch.int32 and father_index.is_contiguous()
    assert cos_sin.stride(1) == 1
    if k_norm_weight is not None:
        assert k_norm_weight.dim() == 1 and k_norm_weight.shape[0] == k_cache.shape[2]
        k_norm_weight = k_norm_weight.float().contiguous()
    lib = _lib.load()
    a = _lib.SvkDeltakvReconstructArgs(
        delta=_lib.ptr(delta), scale=_lib.ptr(scale), mn=_lib.ptr(mn), latent_slots=_lib.ptr(latent_slots),
        father_slots=_lib.ptr(father_slots), slot_to_pos=_lib.ptr(slot_to_pos), out_slots=_lib.ptr(out_slots),
        out_pos=_lib.ptr(out_pos), cos_sin=_lib.ptr(cos_sin), k_cache=_lib.ptr(k_cache), v_cache=_lib.ptr(v_cache),
        k_norm_weight=_lib.ptr(k_norm_weight), delta_stride=delta.stride(0),
        scale_stride=0 if scale is None else scale.stride(0), father_stride=father_slots.stride(0),
        cos_stride=cos_sin.stride(0), kv_slot_stride=k_cache.stride(0), kv_head_stride=k_cache.stride(1),
        k_norm_eps=float(k_norm_eps), n=int(father_slots.shape[0] if father_index is None else father_index.numel()),
        k_fathers=int(father_slots.shape[1]),
        num_kv_heads=int(k_cache.shape[1]), head_dim=int(k_cache.shape[2]), delta_bits=int(bits),
        group_size=int(group_size), delta_dtype=_dt(delta) if bits == 0 else 0,
        scale_dtype=0 if scale is None else _dt(scale), cos_dtype=_dt(cos_sin), raw_k_cache=int(bool(raw_k_cache)),
        store_raw_k=int(bool(store_raw_k)), father_table=_lib.ptr(father_slots) if father_index is not None else None,
        father_index=_lib.ptr(father_index), father_table_stride=father_slots.stride(0))
    if view_out is not None:
        # MI355X: the rows go straight into the attention view (include/svk.h `out_k_cache`): (out_k, out_v, view width,
        # first column of the selected block, entries per batch row)
        vk, vv, width, offset, per_row = view_out
        assert vk.dtype == torch.bfloat16 and vk.dim() == 3 and vk.stride() == vv.stride() and vk.stride(2) == 1
        a.out_k_cache, a.out_v_cache = _lib.ptr(vk), _lib.ptr(vv)
        a.out_slot_stride, a.out_head_stride = vk.stride(0), vk.stride(1)
        a.out_view_width, a.out_view_offset, a.out_entries_per_row = int(width), int(offset), int(per_row)
    if up is not None:
        _lib.check(lib.svk_deltakv_up_reconstruct(C.byref(up), C.byref(a), C.byref(batch), _lib.current_stream_handle()), lib)
        return
    if batch is not None:
        _lib.check(lib.svk_deltakv_reconstruct_writeback_batched(C.byref(a), C.byref(batch), _lib.current_stream_handle()), lib)
        return
    _lib.check(lib.svk_deltakv_reconstruct_writeback(C.byref(a), _lib.current_stream_handle()), lib)


@torch.no_grad()
def deltakv_reconstruct_writeback_layers(kv_delta, father_table, father_index, slot_to_pos, out_slots, out_pos, cos_sin, k_cache,
                                         v_cache, *, k_norm_weight=None, k_norm_eps: float = 1e-6, raw_k_cache: bool = False,
                                         store_raw_k: bool = False, view_out=None):
    """`deltakv_reconstruct_writeback_grouped_heads` for several layers that share one plan, in ONE launch: kv_delta
    [n_layers, N, 2*Hkv*D] bf16, father_table [n_layers, latents, K], k_cache / v_cache [n_layers, slots, Hkv, D],
    k_norm_weight None or [n_layers, D] f32; father_index / out_slots / out_pos / slot_to_pos are the plan's.
    `view_out` = (out_k [n_layers, rows, Hkv, D], out_v, width, offset, entries_per_row): write into the layers' views."""
    nl = int(kv_delta.shape[0])
    assert kv_delta.dim() == 3 and kv_delta.dtype == torch.bfloat16 and kv_delta.stride(2) == 1
    assert father_table.dim() == 3 and father_table.shape[0] == nl and k_cache.dim() == 4 and k_cache.shape[0] == nl
    assert k_cache.stride() == v_cache.stride()
    if k_norm_weight is not None:
        assert k_norm_weight.dim() == 2 and k_norm_weight.shape[0] == nl and k_norm_weight.dtype == torch.float32 and k_norm_weight.is_contiguous()
    batch = _lib.SvkDeltakvReconstructBatch(
        n_batch=nl, delta_stride_batch=kv_delta.stride(0), father_table_stride_batch=father_table.stride(0),
        kv_cache_stride_batch=k_cache.stride(0), k_norm_stride_batch=0 if k_norm_weight is None else k_norm_weight.stride(0))
    view0 = None
    if view_out is not None:
        vk, vv, width, offset, per_row = view_out
        assert vk.dim() == 4 and vk.shape[0] == nl and vk.stride() == vv.stride()
        batch.out_cache_stride_batch = vk.stride(0)
        view0 = (vk[0], vv[0], width, offset, per_row)
    _reconstruct(delta=kv_delta[0], scale=None, mn=None, latent_slots=None, father_slots=father_table[0],
                 slot_to_pos=slot_to_pos, out_slots=out_slots, out_pos=out_pos, cos_sin=cos_sin, k_cache=k_cache[0],
                 v_cache=v_cache[0], bits=0, group_size=0, k_norm_weight=None if k_norm_weight is None else k_norm_weight[0],
                 k_norm_eps=k_norm_eps, raw_k_cache=raw_k_cache, store_raw_k=store_raw_k, father_index=father_index, batch=batch,
                 view_out=view0)


def deltakv_up_reconstruct_supported(*, head_dim: int, num_kv_heads: int, k_fathers: int, hidden_features: int) -> bool:
    """Shapes `svk_deltakv_up_reconstruct` serves (the caller keeps the library GEMM + reconstruct pair otherwise)."""
    return int(head_dim) == 128 and 1 <= int(num_kv_heads) <= 8 and 1 <= int(k_fathers) <= 4 and int(hidden_features) >= 64 \
        and int(hidden_features) % 64 == 0


@torch.no_grad()
def deltakv_up_reconstruct_layers(hidden, weight, bias, father_table, father_index, slot_to_pos, out_slots, out_pos, cos_sin,
                                  k_cache, v_cache, *, k_norm_weight=None, k_norm_eps: float = 1e-6, view_out=None):
    """MI355X: `F.linear(hidden, weight, bias)` (the second Linear of compress_up, utils/compressor.py:69-73) and
    `deltakv_reconstruct_writeback_layers` of its output in ONE launch for the layers of a look-ahead sub-batch: hidden
    [n_layers, N, K] bf16 (rows may be strided), weight [n_layers, 2*Hkv*D, K] bf16 (rows may be strided), bias None or
    [n_layers, 2*Hkv*D] bf16; the rest as `deltakv_reconstruct_writeback_layers` (un-rotated father keys, rotated
    output).  The delta rows stay in LDS (include/svk.h SvkDeltakvUpReconArgs)."""
    nl = int(hidden.shape[0])
    assert hidden.dim() == 3 and hidden.dtype == torch.bfloat16 and hidden.stride(2) == 1
    assert weight.dim() == 3 and weight.dtype == torch.bfloat16 and weight.stride(2) == 1 and weight.shape[0] == nl
    assert weight.shape[2] == hidden.shape[2] and int(weight.shape[1]) == 2 * int(k_cache.shape[2]) * int(k_cache.shape[3])
    assert father_table.dim() == 3 and father_table.shape[0] == nl and k_cache.dim() == 4 and k_cache.shape[0] == nl
    assert k_cache.stride() == v_cache.stride() and int(father_index.numel()) == int(hidden.shape[1])
    if bias is not None:
        assert bias.dim() == 2 and bias.dtype == torch.bfloat16 and bias.shape[0] == nl and bias.stride(1) == 1
    if k_norm_weight is not None:
        assert k_norm_weight.dim() == 2 and k_norm_weight.shape[0] == nl and k_norm_weight.dtype == torch.float32 and k_norm_weight.is_contiguous()
    up = _lib.SvkDeltakvUpReconArgs(
        hidden=_lib.ptr(hidden), weight=_lib.ptr(weight), bias=_lib.ptr(bias), hidden_stride=hidden.stride(1),
        hidden_stride_batch=hidden.stride(0), weight_stride=weight.stride(1), weight_stride_batch=weight.stride(0),
        bias_stride_batch=0 if bias is None else bias.stride(0), k=int(hidden.shape[2]))
    batch = _lib.SvkDeltakvReconstructBatch(
        n_batch=nl, delta_stride_batch=0, father_table_stride_batch=father_table.stride(0),
        kv_cache_stride_batch=k_cache.stride(0), k_norm_stride_batch=0 if k_norm_weight is None else k_norm_weight.stride(0))
    view0 = None
    if view_out is not None:
        vk, vv, width, offset, per_row = view_out
        assert vk.dim() == 4 and vk.shape[0] == nl and vk.stride() == vv.stride()
        batch.out_cache_stride_batch = vk.stride(0)
        view0 = (vk[0], vv[0], width, offset, per_row)
    _reconstruct(delta=hidden[0], scale=None, mn=None, latent_slots=None, father_slots=father_table[0],
                 slot_to_pos=slot_to_pos, out_slots=out_slots, out_pos=out_pos, cos_sin=cos_sin, k_cache=k_cache[0],
                 v_cache=v_cache[0], bits=0, group_size=0, k_norm_weight=None if k_norm_weight is None else k_norm_weight[0],
                 k_norm_eps=k_norm_eps, raw_k_cache=True, store_raw_k=False, father_index=father_index, batch=batch,
                 view_out=view0, up=up)


@torch.no_grad()
def deltakv_reconstruct_writeback_grouped_heads(kv_delta, father_slots, slot_to_pos, out_slots, out_pos, cos_sin, k_cache,
                                                v_cache, *, heads_per_program: int = 4, pre_rope_k_cache=None,
                                                ref_v_cache=None, k_norm_weight=None, k_norm_eps: float = 1e-6,
                                                raw_k_cache: bool = False, store_raw_k: bool = False, father_index=None,
                                                view_out=None):
    """`father_index` (extension): `father_slots` is then the whole `[latents, K]` father table and entry n uses row
    max(father_index[n], 0) with negative fathers clamped to 0 (the static-decode gather fused into the kernel).
    `view_out` (extension) = (out_k [rows, Hkv, D], out_v, width, offset, entries_per_row): the rows are written into
    the attention view instead of the cache (include/svk.h `out_k_cache`)."""
    if int(heads_per_program) <= 0:
        raise ValueError("heads_per_program must be a positive integer.")
    if pre_rope_k_cache is not None or ref_v_cache is not None:
        raise NotImplementedError("separate pre-RoPE K / reference V caches are not part of this build")
    assert kv_delta.dim() == 2 and kv_delta.stride(1) == 1
    _reconstruct(delta=kv_delta, scale=None, mn=None, latent_slots=None, father_slots=father_slots,
                 slot_to_pos=slot_to_pos, out_slots=out_slots, out_pos=out_pos, cos_sin=cos_sin, k_cache=k_cache,
                 v_cache=v_cache, bits=0, group_size=0, k_norm_weight=k_norm_weight, k_norm_eps=k_norm_eps,
                 raw_k_cache=raw_k_cache, store_raw_k=store_raw_k, father_index=father_index, view_out=view_out)


@torch.no_grad()
def deltakv_less_memory_reconstruct_writeback_quantized(packed_delta_cache, scale_cache, min_cache, latent_slots,
                                                        father_slots, slot_to_pos, out_slots, out_pos, cos_sin, k_cache,
                                                        v_cache, *, quant_bits: int, group_size: int | None = None,
                                                        heads_per_program: int = 4, k_norm_weight=None,
                                                        k_norm_eps: float = 1e-6, raw_k_cache: bool = False,
                                                        store_raw_k: bool = False):
    assert packed_delta_cache.dim() == 2 and packed_delta_cache.dtype == torch.int32
    assert scale_cache.dim() == 2 and min_cache.dim() == 2 and scale_cache.dtype == min_cache.dtype
    assert latent_slots.dim() == 1 and father_slots.dim() == 2
    num_kv_heads, head_dim = k_cache.shape[1], k_cache.shape[2]
    assert head_dim % 2 == 0
    D = num_kv_heads * head_dim
    quant_bits = int(quant_bits)
    if quant_bits not in (2, 4):
        raise ValueError(f"DeltaKV fused residual reconstruction supports quant_bits=2 or 4, got {quant_bits}.")
    feat_per_int = 32 // quant_bits
    if (2 * D) % feat_per_int != 0:
        raise ValueError(f"int{quant_bits} residual packing requires 2*D={2 * D} divisible by {feat_per_int}.")
    assert packed_delta_cache.shape[1] == (2 * D) // feat_per_int
    group_size = int(group_size or (2 * D))
    if group_size <= 0 or (2 * D) % group_size != 0:
        raise ValueError("DeltaKV fused residual reconstruction requires 2*D divisible by group_size; "
                         f"2*D={2 * D}, group_size={group_size}.")
    num_groups = (2 * D) // group_size
    if scale_cache.shape[1] != num_groups or min_cache.shape[1] != num_groups:
        raise ValueError("DeltaKV fused residual reconstruction scale/min group count mismatch: "
                         f"expected={num_groups}, scale={tuple(scale_cache.shape)}, min={tuple(min_cache.shape)}.")
    if int(heads_per_program) <= 0:
        raise ValueError("heads_per_program must be a positive integer.")
    assert scale_cache.stride() == min_cache.stride() and scale_cache.stride(1) == 1
    _reconstruct(delta=packed_delta_cache, scale=scale_cache, mn=min_cache, latent_slots=latent_slots,
                 father_slots=father_slots, slot_to_pos=slot_to_pos, out_slots=out_slots, out_pos=out_pos,
                 cos_sin=cos_sin, k_cache=k_cache, v_cache=v_cache, bits=quant_bits, group_size=group_size,
                 k_norm_weight=k_norm_weight, k_norm_eps=k_norm_eps, raw_k_cache=raw_k_cache, store_raw_k=store_raw_k)


def deltakv_less_memory_reconstruct_writeback_int4(*args, **kwargs):
    return deltakv_less_memory_reconstruct_writeback_quantized(*args, quant_bits=4, **kwargs)


def dequantize_grouped(packed, scale, mn, group_size: int, output_dim: int, bits: int, out_dtype=None, *, row_index=None):
    """q * scale + mn for LSB-first `bits`-wide codes (quant.py:120-157, :304-349).  With `row_index` (int32 [n]) the
    output row r is decoded from source row max(row_index[r], 0): the latent gather of `_load_residual` fused in."""
    if packed.dim() != 2 or scale.dim() != 2 or mn.dim() != 2:
        raise ValueError("2D dequantization expects rank-2 packed/scale/min tensors, "
                         f"got packed={tuple(packed.shape)}, scale={tuple(scale.shape)}, mn={tuple(mn.shape)}.")
    src_rows = int(packed.shape[0])
    n = src_rows if row_index is None else int(row_index.numel())
    output_dim, group_size, bits = int(output_dim), int(group_size), int(bits)
    if bits not in (2, 4, 8):
        raise ValueError(f"Packed quantization supports bits=(2, 4, 8), got {bits}.")
    fpi = 32 // bits
    if output_dim <= 0 or output_dim % fpi != 0:
        raise ValueError(f"dequantization requires output_dim divisible by {fpi}, got {output_dim}.")
    if group_size <= 0 or output_dim % group_size != 0:
        raise ValueError("dequantization requires output_dim divisible by group_size, "
                         f"got output_dim={output_dim}, group_size={group_size}.")
    if int(packed.shape[1]) != output_dim // fpi:
        raise ValueError(f"dequantization packed width mismatch: packed={packed.shape[1]}, expected={output_dim // fpi}.")
    if tuple(scale.shape) != (src_rows, output_dim // group_size) or tuple(mn.shape) != tuple(scale.shape):
        raise ValueError("dequantization scale/min shape mismatch: "
                         f"scale={tuple(scale.shape)}, mn={tuple(mn.shape)}, expected={(src_rows, output_dim // group_size)}.")
    if row_index is not None:
        assert row_index.dtype == torch.int32 and row_index.is_contiguous()
        assert packed.stride(1) == 1 and scale.stride(1) == 1 and scale.stride() == mn.stride()
    else:
        packed, scale, mn = packed.contiguous(), scale.contiguous(), mn.contiguous()
    out = torch.empty((n, output_dim), device=packed.device, dtype=out_dtype or scale.dtype)
    lib = _lib.load()
    a = _lib.SvkDequantGroupedArgs(packed=_lib.ptr(packed), scale=_lib.ptr(scale), mn=_lib.ptr(mn), out=_lib.ptr(out),
                                   packed_stride=packed.stride(0), scale_stride=scale.stride(0), out_stride=out.stride(0),
                                   rows=n, features=output_dim, bits=bits, group_size=group_size, scale_dtype=_dt(scale),
                                   out_dtype=_dt(out), row_index=_lib.ptr(row_index))
    _lib.check(lib.svk_dequantize_grouped(C.byref(a), _lib.current_stream_handle()), lib)
    return out


def deltakv_decode_alloc(meta, *, batch: int, full_slots_map, full_slot_to_pos, sparse_raw_slots_map, sparse_slot_to_pos,
                         context_lens, req_indices, slot_mapping, sparse_slot_mapping, compressed_lens):
    """Slot bookkeeping of one DeltaKV decode step (deltakv_base.py:2038-2154 `prepare_decode_static`, device half) in
    one launch: `meta` [5, >= batch] int32 on the device = row, cur_len, full_slot, sparse_slot, compressed_len per real
    lane; the five graph-stable buffers are written for all their lanes (padded lanes mirror lane 0, slots -1)."""
    gb = int(context_lens.numel())
    assert meta.dtype == torch.int32 and meta.dim() == 2 and meta.shape[0] == 5 and meta.stride(1) == 1 and meta.shape[1] >= batch
    for t in (full_slots_map, full_slot_to_pos, sparse_raw_slots_map, sparse_slot_to_pos, context_lens, req_indices,
              slot_mapping, sparse_slot_mapping, compressed_lens):
        assert t.dtype == torch.int32 and t.stride(-1) == 1
    for t in (req_indices, slot_mapping, sparse_slot_mapping, compressed_lens):
        assert int(t.numel()) >= gb
    lib = _lib.load()
    a = _lib.SvkDeltakvDecodeAllocArgs(
        meta=_lib.ptr(meta), meta_stride=meta.stride(0), full_slots_map=_lib.ptr(full_slots_map),
        full_map_stride=full_slots_map.stride(0), full_slot_to_pos=_lib.ptr(full_slot_to_pos),
        sparse_raw_slots_map=_lib.ptr(sparse_raw_slots_map), sparse_map_stride=sparse_raw_slots_map.stride(0),
        sparse_slot_to_pos=_lib.ptr(sparse_slot_to_pos), context_lens=_lib.ptr(context_lens), req_indices=_lib.ptr(req_indices),
        slot_mapping=_lib.ptr(slot_mapping), sparse_slot_mapping=_lib.ptr(sparse_slot_mapping),
        compressed_lens=_lib.ptr(compressed_lens), batch=int(batch), graph_batch=gb)
    _lib.check(lib.svk_deltakv_decode_alloc(C.byref(a), _lib.current_stream_handle()), lib)


def deltakv_device_step_args(*, rows, row_len, compressed_len, full_stack, full_ptr, sparse_stack, sparse_ptr, full_slots_map,
                             full_slot_to_pos, sparse_raw_slots_map, sparse_slot_to_pos, context_lens, req_indices,
                             slot_mapping, sparse_slot_mapping, compressed_lens, batch: int):
    """Arguments of the device-resident DeltaKV decode step (`svk_deltakv_device_step_begin`), built once per batch
    composition: every pointer is a persistent tensor of the manager (the caller keeps them alive)."""
    gb = int(context_lens.numel())
    for t in (rows, row_len, compressed_len, full_stack, full_ptr, sparse_stack, sparse_ptr, full_slots_map, full_slot_to_pos,
              sparse_raw_slots_map, sparse_slot_to_pos, context_lens, req_indices, slot_mapping, sparse_slot_mapping, compressed_lens):
        assert t.dtype == torch.int32 and t.stride(-1) == 1 and t.is_cuda
    assert int(rows.numel()) >= batch and 0 < batch <= gb
    for t in (req_indices, slot_mapping, sparse_slot_mapping, compressed_lens):
        assert int(t.numel()) >= gb
    return _lib.SvkDeltakvDeviceStepArgs(
        rows=_lib.ptr(rows), row_len=_lib.ptr(row_len), compressed_len=_lib.ptr(compressed_len),
        full_stack=_lib.ptr(full_stack), full_ptr=_lib.ptr(full_ptr), sparse_stack=_lib.ptr(sparse_stack),
        sparse_ptr=_lib.ptr(sparse_ptr), full_slots_map=_lib.ptr(full_slots_map), full_map_stride=full_slots_map.stride(0),
        full_slot_to_pos=_lib.ptr(full_slot_to_pos), sparse_raw_slots_map=_lib.ptr(sparse_raw_slots_map),
        sparse_map_stride=sparse_raw_slots_map.stride(0), sparse_slot_to_pos=_lib.ptr(sparse_slot_to_pos),
        context_lens=_lib.ptr(context_lens), req_indices=_lib.ptr(req_indices), slot_mapping=_lib.ptr(slot_mapping),
        sparse_slot_mapping=_lib.ptr(sparse_slot_mapping), compressed_lens=_lib.ptr(compressed_lens), batch=int(batch),
        graph_batch=gb)


def deltakv_device_step_begin(args):
    lib = _lib.load()
    _lib.check(lib.svk_deltakv_device_step_begin(C.byref(args), _lib.current_stream_handle()), lib)


def dequant_linear_act(packed, scale, mn, group_size: int, weight, bias=None, *, activation: str = "gelu", row_index=None,
                       out=None, layers: bool = False):
    """`act(F.linear(dequant_int4(packed[row_index]), weight, bias))` as one MFMA launch (MI355X fusion of the residual
    load's dequantisation with the first Linear (+ erf-GELU) of `compress_up`; `_load_residual`,
    deltakv_less_memory.py:2841-2848, utils/compressor.py:69-73).  bf16 weights and output, int4 codes."""
    batch = None
    if layers:
        # several layers in one launch: packed / scale / mn [n_layers, rows, ...], weight [n_layers, N, K], bias
        # [n_layers, N], out [n_layers, n, N]; row_index is shared
        assert packed.dim() == 3 and scale.dim() == 3 and mn.dim() == 3 and weight.dim() == 3 and out is not None and out.dim() == 3
        nl = int(packed.shape[0])
        assert scale.shape[0] == nl and weight.shape[0] == nl and out.shape[0] == nl and (bias is None or (bias.dim() == 2 and bias.shape[0] == nl))
        assert scale.stride() == mn.stride()
        batch = _lib.SvkDequantLinearBatch(n_batch=nl, packed_stride_batch=packed.stride(0), scale_stride_batch=scale.stride(0),
                                           weight_stride_batch=weight.stride(0), bias_stride_batch=0 if bias is None else bias.stride(0),
                                           out_stride_batch=out.stride(0))
        out_all = out
        packed, scale, mn, weight, out = packed[0], scale[0], mn[0], weight[0], out[0]
        bias = None if bias is None else bias[0]
    if packed.dim() != 2 or scale.dim() != 2 or mn.dim() != 2:
        raise ValueError("2D dequantization expects rank-2 packed/scale/min tensors, "
                         f"got packed={tuple(packed.shape)}, scale={tuple(scale.shape)}, mn={tuple(mn.shape)}.")
    if activation not in ("none", "gelu"):
        raise ValueError(f"activation must be 'none' or 'gelu', got {activation!r}")
    K, N = int(weight.shape[1]), int(weight.shape[0])
    group_size = int(group_size)
    if int(packed.shape[1]) * 8 != K:
        raise ValueError(f"dequantization packed width mismatch: packed={packed.shape[1]}, expected={K // 8}.")
    if group_size <= 0 or K % group_size != 0:
        raise ValueError("dequantization requires output_dim divisible by group_size, "
                         f"got output_dim={K}, group_size={group_size}.")
    src_rows = int(packed.shape[0])
    if tuple(scale.shape) != (src_rows, K // group_size) or tuple(mn.shape) != tuple(scale.shape):
        raise ValueError("dequantization scale/min shape mismatch: "
                         f"scale={tuple(scale.shape)}, mn={tuple(mn.shape)}, expected={(src_rows, K // group_size)}.")
    assert weight.dtype == torch.bfloat16 and weight.stride(1) == 1
    assert bias is None or (bias.dtype == torch.bfloat16 and bias.is_contiguous() and bias.numel() == N)
    assert packed.dtype == torch.int32 and packed.stride(1) == 1 and scale.stride(1) == 1 and scale.stride() == mn.stride()
    assert scale.dtype == mn.dtype
    if row_index is not None:
        assert row_index.dtype == torch.int32 and row_index.is_contiguous()
    n = src_rows if row_index is None else int(row_index.numel())
    if out is None:
        out = torch.empty((n, N), device=packed.device, dtype=torch.bfloat16)
    else:       # caller-owned buffer (work issued ahead on a side stream must not borrow from the caching allocator)
        assert out.dtype == torch.bfloat16 and tuple(out.shape) == (n, N) and out.stride(1) == 1
    lib = _lib.load()
    a = _lib.SvkDequantLinearArgs(packed=_lib.ptr(packed), scale=_lib.ptr(scale), mn=_lib.ptr(mn), row_index=_lib.ptr(row_index),
                                  weight=_lib.ptr(weight), bias=_lib.ptr(bias), out=_lib.ptr(out),
                                  packed_stride=packed.stride(0), scale_stride=scale.stride(0), weight_stride=weight.stride(0),
                                  out_stride=out.stride(0), rows=n, k=K, n=N, group_size=group_size, scale_dtype=_dt(scale),
                                  activation=1 if activation == "gelu" else 0)
    if batch is not None:
        _lib.check(lib.svk_dequant_linear_act_batched(C.byref(a), C.byref(batch), _lib.current_stream_handle()), lib)
        return out_all
    _lib.check(lib.svk_dequant_linear_act(C.byref(a), _lib.current_stream_handle()), lib)
    return out


def triton_dequantize_2d_int4_grouped(packed, scale, mn, group_size: int, output_dim: int):
    """Reference name kept for call-site compatibility (quant.py:160-216)."""
    return dequantize_grouped(packed, scale, mn, group_size, output_dim, 4)


_TOKEN_SCORE_WS: dict = {}


def decode_softmax_token_scores(scores, *, candidate_start: int, candidate_lens, scale: float, round_dtype=None,
                                fill_value: float | None = None):
    """SparseController._decode_softmax_token_scores (sparse_controller.py:255-299)."""
    if scores.dim() != 3:
        raise ValueError(f"Expected decode scores with shape [B, H, L], got {tuple(scores.shape)}.")
    candidate_start = int(candidate_start)
    if candidate_start < 0 or candidate_start > scores.shape[-1]:
        raise ValueError(f"candidate_start must be within score length; got {candidate_start} for L={scores.shape[-1]}.")
    assert scores.dtype == torch.float32 and scores.stride(2) == 1
    B, H, L = scores.shape
    rd = torch.float32 if round_dtype is None else round_dtype
    if fill_value is None:
        fill_value = torch.finfo(rd).min
    out = torch.empty((B, L), dtype=torch.float32, device=scores.device)
    lib = _lib.load()
    # statistics + tickets: zero-filled once, self-cleaning afterwards (include/svk.h), one buffer per launch shape
    key = (scores.device.index, B, H, L)
    ws = _TOKEN_SCORE_WS.get(key)
    if ws is None:
        ws = _TOKEN_SCORE_WS[key] = torch.zeros((B, H, int(lib.svk_deltakv_token_scores_chunks(L)), 2), dtype=torch.float32,
                                                device=scores.device)
    a = _lib.SvkDeltakvTokenScoresArgs(
        raw_scores=_lib.ptr(scores), candidate_lens=_lib.ptr(candidate_lens.to(torch.int32)), token_scores=_lib.ptr(out),
        workspace=_lib.ptr(ws), raw_stride_b=scores.stride(0), raw_stride_h=scores.stride(1), out_stride=out.stride(0),
        scale=float(scale), fill_value=float(fill_value), batch=B, num_heads=H, length=L, candidate_start=candidate_start,
        round_dtype=_DT[rd])
    _lib.check(lib.svk_deltakv_token_scores(C.byref(a), _lib.current_stream_handle()), lib)
    return out


_TOPK_WS: dict = {}


def topk_sorted_desc(scores, k: int, *, valid_len=None, masked_value: float = -1e10):
    """`scores.topk(k, sorted=True).indices` (int32) with ties broken by the lower index."""
    assert scores.dim() == 2 and scores.dtype == torch.float32 and scores.stride(1) == 1
    rows, n = scores.shape
    out = torch.empty((rows, int(k)), dtype=torch.int32, device=scores.device)
    lib = _lib.load()
    ws_bytes = int(lib.svk_topk_sorted_workspace_bytes(rows, n, int(k)))
    ws = None
    if ws_bytes > 0:
        # include/svk.h: the workspace's first rows x 16 KiB (level-1 histograms) are zero on entry and zero again on exit,
        # so ONE zero-filled buffer per (device, launch shape) serves every call (launches on a device are ordered; a
        # captured step keeps pointing at it: the buffer is never released)
        import os
        if os.environ.get("SVK_TOPK_PLAN") or os.environ.get("SVK_TOPK_FINAL"):
            # the A/B plans (one histogram level) do not keep the contract: a buffer of their own per call
            ws = torch.zeros((ws_bytes,), dtype=torch.uint8, device=scores.device)
        else:
            key = (scores.device.index, rows, n)
            ws = _TOPK_WS.get(key)
            if ws is None or ws.numel() < ws_bytes:
                ws = _TOPK_WS[key] = torch.zeros((ws_bytes,), dtype=torch.uint8, device=scores.device)
    a = _lib.SvkTopkSortedArgs(scores=_lib.ptr(scores), valid_len=_lib.ptr(valid_len), indices=_lib.ptr(out),
                               score_stride=scores.stride(0), index_stride=out.stride(0), masked_value=float(masked_value),
                               rows=rows, n=n, k=int(k))
    _lib.check(lib.svk_topk_sorted_desc(C.byref(a), _lib.ptr(ws), _lib.current_stream_handle()), lib)
    return out


@torch.no_grad()
def deltakv_materialize_sparse_view(active_slots, context_lens, slot_to_pos, postrope_mask, k_cache, v_cache, out_k, out_v,
                                    cos_sin, *, k_norm_weight=None, k_norm_eps: float = 1e-6, block_tokens: int = 16,
                                    temp_slots=None, temp_offset: int = 0, new_k=None, new_v=None, new_slots=None,
                                    skip_temp: bool = False, skip_new: bool = False):
    """Reference wrapper deltakv_kernels.py:3489-3585 (same arguments; `block_tokens` is a Triton tile knob).
    Extensions: `temp_slots` [B, K] + `temp_offset` replace the mask in static decode (an entry in columns
    [temp_offset, temp_offset + K) is post-RoPE iff its slot is this step's reconstruct scratch slot);
    `new_k`/`new_v` [B, Hkv, D] + `new_slots` [B] carry this step's raw store in the same launch (row b's new token
    goes to cache slot new_slots[b]; the view reads it from new_k/new_v), equal to store_kvcache followed by the
    plain call; `skip_temp`: the scratch entries are left alone (the reconstruction wrote them into out_k / out_v);
    `skip_new` (with `new_slots`, without `new_k`): the entry of row b's newest token is left alone too - the layer's
    attention launch stores it rotated (`flash_decode_stage1(rotated_store=...)`).  k_cache / v_cache / out_k / out_v may
    carry a leading LAYER dimension ([n, slots, Hkv, D], `k_norm_weight` [n, D]): the same slot table materialised for n
    consecutive layers in one launch (no store)."""
    for t in (active_slots, context_lens, slot_to_pos, k_cache, v_cache, out_k, out_v, cos_sin):
        assert t.is_cuda
    assert active_slots.dim() == 2
    assert context_lens.dim() == 1 and context_lens.shape[0] == active_slots.shape[0]
    layers = 1
    kv_ls = out_ls = kn_ls = 0
    if k_cache.dim() == 4:
        layers = int(k_cache.shape[0])
        assert v_cache.shape == k_cache.shape and out_k.dim() == 4 and out_k.shape[0] == layers and out_v.shape == out_k.shape
        assert new_k is None and k_cache.stride() == v_cache.stride() and out_k.stride() == out_v.stride()
        kv_ls, out_ls = k_cache.stride(0), out_k.stride(0)
        if k_norm_weight is not None:
            assert k_norm_weight.dim() == 2 and k_norm_weight.shape[0] == layers and k_norm_weight.dtype == torch.float32
            assert k_norm_weight.stride(1) == 1
            kn_ls = k_norm_weight.stride(0)
            k_norm_weight = k_norm_weight[0]
        k_cache, v_cache, out_k, out_v = k_cache[0], v_cache[0], out_k[0], out_v[0]
    assert k_cache.dim() == 3 and v_cache.shape == k_cache.shape
    assert out_k.dim() == 3 and out_v.shape == out_k.shape
    batch, width = active_slots.shape
    total = int(batch) * int(width)
    if total == 0:
        return
    if out_k.shape[0] < total or out_v.shape[0] < total:
        raise RuntimeError("DeltaKV materialize sparse view output is too small: "
                           f"out={tuple(out_k.shape)}/{tuple(out_v.shape)} need={total}.")
    num_kv_heads, head_dim = int(k_cache.shape[1]), int(k_cache.shape[2])
    if head_dim % 2 != 0:
        raise RuntimeError(f"DeltaKV materialize sparse view requires an even head_dim, got {head_dim}.")
    if cos_sin.dim() == 3:
        cos_sin = cos_sin[:, 0, :]
    assert cos_sin.dim() == 2 and cos_sin.shape[1] == head_dim and cos_sin.stride(1) == 1
    if k_norm_weight is not None:
        assert k_norm_weight.is_cuda and k_norm_weight.dim() == 1 and k_norm_weight.shape[0] == head_dim
        if layers == 1:
            k_norm_weight = k_norm_weight.to(torch.float32).contiguous()
    if postrope_mask is not None:
        assert postrope_mask.is_cuda and postrope_mask.dim() == 1 and postrope_mask.shape[0] >= k_cache.shape[0]
        assert postrope_mask.dtype in (torch.bool, torch.uint8) and postrope_mask.is_contiguous()
    if temp_slots is not None:
        assert temp_slots.dtype == torch.int32 and temp_slots.dim() == 2 and temp_slots.stride(1) == 1
        assert temp_slots.shape[0] == batch
    assert active_slots.dtype == torch.int32 and active_slots.stride(1) == 1
    assert slot_to_pos.dtype == torch.int32 and slot_to_pos.is_contiguous()
    for t in (k_cache, v_cache, out_k, out_v):
        assert t.dtype == torch.bfloat16 and t.stride(2) == 1
    assert k_cache.stride() == v_cache.stride() and out_k.stride() == out_v.stride()
    lib = _lib.load()
    a = _lib.SvkDeltakvMaterializeArgs(
        active_slots=_lib.ptr(active_slots), slot_to_pos=_lib.ptr(slot_to_pos), postrope_mask=_lib.ptr(postrope_mask),
        k_cache=_lib.ptr(k_cache), v_cache=_lib.ptr(v_cache), out_k=_lib.ptr(out_k), out_v=_lib.ptr(out_v),
        cos_sin=_lib.ptr(cos_sin), k_norm_weight=_lib.ptr(k_norm_weight),
        active_stride=active_slots.stride(0), kv_slot_stride=k_cache.stride(0), kv_head_stride=k_cache.stride(1),
        out_slot_stride=out_k.stride(0), out_head_stride=out_k.stride(1), cos_stride=cos_sin.stride(0),
        k_norm_eps=float(k_norm_eps), batch=int(batch), width=int(width), num_slots=int(k_cache.shape[0]),
        num_kv_heads=num_kv_heads, head_dim=head_dim, cos_dtype=_dt(cos_sin), temp_slots=_lib.ptr(temp_slots),
        temp_stride=0 if temp_slots is None else temp_slots.stride(0), temp_offset=int(temp_offset),
        temp_count=0 if temp_slots is None else int(temp_slots.shape[1]), skip_temp=int(bool(skip_temp) and temp_slots is not None),
        layer_count=layers, kv_layer_stride=kv_ls, out_layer_stride=out_ls, k_norm_layer_stride=kn_ls)
    if skip_new:
        assert new_slots is not None and new_k is None and new_v is None
        assert new_slots.dtype == torch.int32 and new_slots.is_contiguous() and new_slots.numel() == batch
        a.new_slots, a.skip_new = _lib.ptr(new_slots), 1
    elif new_slots is not None:
        assert new_k is not None and new_v is not None and new_k.shape == new_v.shape and new_k.stride() == new_v.stride()
        assert new_k.dtype == torch.bfloat16 and new_v.dtype == torch.bfloat16 and new_k.stride(-1) == 1
        assert tuple(new_k.shape) == (batch, num_kv_heads, head_dim)
        assert new_slots.dtype == torch.int32 and new_slots.is_contiguous() and new_slots.numel() == batch
        a.new_k, a.new_v, a.new_slots = _lib.ptr(new_k), _lib.ptr(new_v), _lib.ptr(new_slots)
        a.new_token_stride, a.new_head_stride = new_k.stride(0), new_k.stride(1)
    _lib.check(lib.svk_deltakv_materialize_sparse_view(C.byref(a), _lib.current_stream_handle()), lib)


def full_layer_kivi_flash_decode_stage1(*, q, raw_k, raw_v, raw_slots_map, kivi_block_slots_map, kivi_block_start_pos,
                                        key_packed, key_scales, key_mins, value_packed, value_scales, value_mins,
                                        req_indices, context_lens, max_len_in_batch: int, mid_out, mid_out_logsumexp,
                                        group_size: int, block_seq: int, block_n: int = 16, num_warps: int = 2,
                                        num_stages: int = 3, attn_score=None, extra_partial_slots: int = 0,
                                        new_kv=None) -> int:
    """Decode stage 1 over KIVI-int4 blocks + raw tail (reference wrapper deltakv_kernels.py:973-1142; same
    argument names and ValueErrors).  block_n / num_warps / num_stages are Triton launch knobs: validated,
    otherwise unused (the HIP kernel tiles 32 tokens per wave).
    MI355X: `extra_partial_slots` = partial slots of mid_out / mid_out_logsumexp beyond ceil(max_len / block_seq) the
    launch may use for the raw / ragged pieces of every row; returns how many it used (0 or 3) - the `extra_partials`
    stage 2 has to merge.  `new_kv` = (k [B, Hkv, D], v, slots [B] int32): this step's raw store of the layer riding in
    the launch (only when `kivi_fused_store_supported(...)`; equal to store_kvcache(k, v, raw_k, raw_v, slots) first)."""
    for t in (q, raw_k, raw_v, raw_slots_map, kivi_block_slots_map, kivi_block_start_pos, key_packed, key_scales,
              key_mins, value_packed, value_scales, value_mins, req_indices, context_lens, mid_out, mid_out_logsumexp):
        assert t.is_cuda
    if attn_score is not None:
        assert attn_score.is_cuda
        if attn_score.dim() != 3:
            raise ValueError("Full-layer KIVI fused decode currently supports rank-3 attention scores only.")
    if q.dim() != 3 or raw_k.dim() != 3 or raw_v.dim() != 3:
        raise ValueError(f"Expected q/raw_k/raw_v rank-3 tensors, got {q.dim()}, {raw_k.dim()}, {raw_v.dim()}.")
    if raw_slots_map.dim() != 2 or kivi_block_slots_map.dim() != 2:
        raise ValueError("Full-layer KIVI decode maps must be rank-2 tensors.")
    if tuple(raw_slots_map.shape) != tuple(kivi_block_slots_map.shape):
        raise ValueError("Full-layer KIVI raw and block slot maps must have identical shapes, "
                         f"got raw={tuple(raw_slots_map.shape)} block={tuple(kivi_block_slots_map.shape)}.")
    batch = int(q.shape[0])
    if int(req_indices.numel()) != batch or int(context_lens.numel()) != batch:
        raise ValueError("Full-layer KIVI decode expects one req index/context length per batch item.")
    head_dim = int(q.shape[-1])
    if head_dim != int(raw_k.shape[-1]) or head_dim != int(raw_v.shape[-1]):
        raise ValueError("Full-layer KIVI decode head_dim mismatch.")
    if head_dim not in {64, 128}:
        raise ValueError(f"Unsupported decode head_dim={head_dim}.")
    group_size = int(group_size)
    if group_size <= 0 or head_dim % group_size != 0:
        raise ValueError(f"Invalid KIVI group_size={group_size} for head_dim={head_dim}.")
    if group_size % 8 != 0 or head_dim % 8 != 0:
        raise ValueError(f"int4 KIVI requires group_size/head_dim divisible by 8, got {group_size}/{head_dim}.")
    block_seq, block_n = int(block_seq), int(block_n)
    if block_seq <= 0 or block_seq % 16 != 0:
        raise ValueError(f"block_seq must be a positive multiple of 16, got {block_seq}.")
    if block_n <= 0 or block_n % 16 != 0 or block_seq % block_n != 0:
        raise ValueError("block_n must be a positive multiple of 16 and divide block_seq, "
                         f"got block_n={block_n}, block_seq={block_seq}.")
    max_len_in_batch = int(max_len_in_batch)
    if max_len_in_batch <= 0:
        return 0
    if max_len_in_batch > int(raw_slots_map.shape[1]):
        raise ValueError("Full-layer KIVI max_len_in_batch exceeds map width: "
                         f"max_len={max_len_in_batch} width={int(raw_slots_map.shape[1])}.")
    num_kv_heads = int(raw_k.shape[1])
    if int(raw_v.shape[1]) != num_kv_heads:
        raise ValueError("Full-layer KIVI decode raw K/V head count mismatch.")
    if int(key_packed.shape[1]) != num_kv_heads or int(value_packed.shape[1]) != num_kv_heads:
        raise ValueError("Full-layer KIVI packed K/V head count mismatch.")
    if int(q.shape[1]) % num_kv_heads != 0:
        raise ValueError(f"Q heads must be divisible by KV heads, got {q.shape[1]}/{num_kv_heads}.")
    assert q.dtype == torch.bfloat16 and raw_k.dtype == torch.bfloat16 and raw_v.dtype == torch.bfloat16
    assert q.stride(2) == 1 and raw_k.stride(2) == 1 and raw_k.stride() == raw_v.stride()
    assert raw_slots_map.dtype == torch.int32 and kivi_block_slots_map.dtype == torch.int32
    assert raw_slots_map.stride(1) == 1 and raw_slots_map.stride() == kivi_block_slots_map.stride()
    assert kivi_block_start_pos.dtype == torch.int32 and kivi_block_start_pos.is_contiguous()
    for t in (key_packed, value_packed):
        assert t.dtype == torch.int32 and t.is_contiguous()
    for t in (value_scales, value_mins):
        assert t.dtype == torch.bfloat16 and t.is_contiguous()
    assert key_scales.dtype in (torch.float32, torch.bfloat16) and key_mins.dtype == key_scales.dtype
    assert key_scales.is_contiguous() and key_mins.is_contiguous()
    assert tuple(key_packed.shape[2:]) == (head_dim, group_size // 8)
    assert tuple(value_packed.shape[2:]) == (group_size, head_dim // 8)
    assert mid_out.dtype == torch.float32 and mid_out.stride(3) == 1 and mid_out_logsumexp.stride(2) == 1
    if attn_score is not None:
        assert attn_score.dtype == torch.float32 and attn_score.stride(2) == 1
    lib = _lib.load()
    a = _lib.SvkKiviDecodeStage1Args(
        q=_lib.ptr(q), raw_k=_lib.ptr(raw_k), raw_v=_lib.ptr(raw_v), raw_slots_map=_lib.ptr(raw_slots_map),
        kivi_block_slots_map=_lib.ptr(kivi_block_slots_map), kivi_block_start_pos=_lib.ptr(kivi_block_start_pos),
        key_packed=_lib.ptr(key_packed), key_scales=_lib.ptr(key_scales), key_mins=_lib.ptr(key_mins),
        value_packed=_lib.ptr(value_packed), value_scales=_lib.ptr(value_scales), value_mins=_lib.ptr(value_mins),
        req_indices=_lib.ptr(req_indices.to(torch.int32).contiguous()),
        context_lens=_lib.ptr(context_lens.to(torch.int32).contiguous()),
        mid_o=_lib.ptr(mid_out), mid_lse=_lib.ptr(mid_out_logsumexp), attn_score=_lib.ptr(attn_score),
        q_stride_b=q.stride(0), q_stride_h=q.stride(1), raw_slot_stride=raw_k.stride(0), raw_head_stride=raw_k.stride(1),
        map_stride=raw_slots_map.stride(0), mid_o_stride_b=mid_out.stride(0), mid_o_stride_h=mid_out.stride(1),
        mid_o_stride_s=mid_out.stride(2), mid_lse_stride_b=mid_out_logsumexp.stride(0),
        mid_lse_stride_h=mid_out_logsumexp.stride(1),
        score_stride_b=attn_score.stride(0) if attn_score is not None else 0,
        score_stride_h=attn_score.stride(1) if attn_score is not None else 0,
        batch=batch, num_q_heads=int(q.shape[1]), num_kv_heads=num_kv_heads, head_dim=head_dim,
        max_len_in_batch=max_len_in_batch, block_seq=block_seq, group_size=group_size,
        key_param_dtype=_dt(key_scales), extra_partials=0)
    extra = int(lib.svk_kivi_decode_stage1_extra_partials(C.byref(a))) if int(extra_partial_slots) > 0 else 0
    nblk = (max_len_in_batch + block_seq - 1) // block_seq
    if extra > int(extra_partial_slots) or int(mid_out.shape[2]) < nblk + extra or int(mid_out_logsumexp.shape[2]) < nblk + extra:
        extra = 0
    a.extra_partials = extra
    if new_kv is not None:
        nk, nv, slots = new_kv
        assert nk.dtype == torch.bfloat16 and nv.dtype == torch.bfloat16 and nk.stride(-1) == 1 and nk.stride() == nv.stride()
        assert slots.dtype == torch.int32 and slots.is_contiguous() and int(slots.numel()) == batch and int(nk.shape[0]) == batch
        a.new_k, a.new_v, a.new_slots = _lib.ptr(nk), _lib.ptr(nv), _lib.ptr(slots)
        a.new_stride_b, a.new_stride_h = nk.stride(0), nk.stride(1)
    _lib.check(lib.svk_kivi_decode_stage1(C.byref(a), _lib.current_stream_handle()), lib)
    return extra


def kivi_fused_store_supported(*, head_dim: int, num_kv_heads: int, group_size: int, block_seq: int, key_param_dtype) -> bool:
    """Does a launch of this shape take the wide kernel, which can carry the step's raw store (`new_kv`)?"""
    lib = _lib.load()
    a = _lib.SvkKiviDecodeStage1Args(head_dim=int(head_dim), num_kv_heads=int(num_kv_heads), group_size=int(group_size),
                                     block_seq=int(block_seq), key_param_dtype=_DT[key_param_dtype])
    return int(lib.svk_kivi_decode_stage1_extra_partials(C.byref(a))) > 0


# ------------------------------------------------------------------------------------------------------------------
# compression side
# ------------------------------------------------------------------------------------------------------------------

@torch.no_grad()
def triton_quantize_and_pack_2d_int4_grouped(data, group_size: int, *, out=None, dst_rows=None, bits: int = 4):
    """kernels/triton/quant.py:79-117 (same ValueErrors).  Extension: `out=(code, scale, mn)` + `dst_rows` write row r of
    the result to row dst_rows[r] of caller-owned caches (the latent-cache stores of _store_residual fused in)."""
    if data.dim() != 2:
        raise ValueError(f"2D int4 quantization expects rank-2 input, got shape={tuple(data.shape)}.")
    if not data.is_cuda:
        raise ValueError("2D int4 quantization expects a CUDA tensor.")
    group_size = int(group_size)
    if group_size <= 0 or group_size % 8 != 0:
        raise ValueError(f"2D int4 quantization requires group_size to be a positive multiple of 8, got {group_size}.")
    n, d = data.shape
    if d % group_size != 0:
        raise ValueError(f"2D int4 quantization requires D divisible by group_size, got D={d}, group={group_size}.")
    if d % 8 != 0:
        raise ValueError(f"2D int4 quantization requires D divisible by 8, got D={d}.")
    data = data.contiguous()
    fpi = 32 // int(bits)
    if out is None:
        code = torch.empty((n, d // fpi), device=data.device, dtype=torch.int32)
        scale = torch.empty((n, d // group_size), device=data.device, dtype=data.dtype)
        mn = torch.empty_like(scale)
    else:
        code, scale, mn = out
        assert code.dtype == torch.int32 and code.stride(1) == 1 and scale.dtype == data.dtype and mn.dtype == data.dtype
        assert scale.stride() == mn.stride() and scale.stride(1) == 1
    if dst_rows is not None:
        assert dst_rows.dtype == torch.int32 and dst_rows.is_contiguous() and dst_rows.numel() == n
    lib = _lib.load()
    a = _lib.SvkQuantPackArgs(data=_lib.ptr(data), dst_rows=_lib.ptr(dst_rows), code=_lib.ptr(code), scale=_lib.ptr(scale),
                              mn=_lib.ptr(mn), data_stride=data.stride(0), code_stride=code.stride(0),
                              scale_stride=scale.stride(0), rows=n, features=d, bits=int(bits), group_size=group_size,
                              data_dtype=_dt(data))
    _lib.check(lib.svk_quantize_pack_grouped(C.byref(a), _lib.current_stream_handle()), lib)
    return code, scale, mn


@torch.no_grad()
def kivi_store_blocks(*, k_cache, v_cache, raw_slots, block_slots, key_packed, key_scales, key_mins, value_packed,
                      value_scales, value_mins, group_size: int):
    """_store_full_layer_kivi_blocks (deltakv_less_memory.py:1741-1780) with the gather of the block's raw rows
    (:3544-3545) and the six indexed stores fused: raw_slots [blocks, G] cache rows -> KIVI block block_slots[j]."""
    assert k_cache.dtype == torch.bfloat16 and k_cache.stride() == v_cache.stride() and k_cache.stride(2) == 1
    assert raw_slots.dtype == torch.int32 and raw_slots.dim() == 2 and raw_slots.is_contiguous()
    assert block_slots.dtype == torch.int32 and block_slots.is_contiguous() and block_slots.numel() == raw_slots.shape[0]
    G = int(group_size)
    if int(raw_slots.shape[1]) != G:
        raise ValueError(f"Full-layer KIVI key blocks shape mismatch: got={tuple(raw_slots.shape)} expected=(*, {G}).")
    H, D = int(k_cache.shape[1]), int(k_cache.shape[2])
    for t_ in (key_packed, value_packed):
        assert t_.dtype == torch.int32 and t_.is_contiguous()
    assert tuple(key_packed.shape[1:]) == (H, D, G // 8) and tuple(value_packed.shape[1:]) == (H, G, D // 8)
    assert key_scales.dtype in (torch.float32, torch.bfloat16) and key_mins.dtype == key_scales.dtype
    assert key_scales.is_contiguous() and key_mins.is_contiguous()
    for t_ in (value_scales, value_mins):
        assert t_.dtype == torch.bfloat16 and t_.is_contiguous()
    lib = _lib.load()
    a = _lib.SvkKiviStoreArgs(k_cache=_lib.ptr(k_cache), v_cache=_lib.ptr(v_cache), raw_slots=_lib.ptr(raw_slots),
                              block_slots=_lib.ptr(block_slots), key_packed=_lib.ptr(key_packed),
                              key_scales=_lib.ptr(key_scales), key_mins=_lib.ptr(key_mins),
                              value_packed=_lib.ptr(value_packed), value_scales=_lib.ptr(value_scales),
                              value_mins=_lib.ptr(value_mins), kv_slot_stride=k_cache.stride(0),
                              kv_head_stride=k_cache.stride(1), blocks=int(raw_slots.shape[0]), num_kv_heads=H, head_dim=D,
                              group_size=G, key_param_dtype=_dt(key_scales))
    _lib.check(lib.svk_kivi_store_blocks(C.byref(a), _lib.current_stream_handle()), lib)


@torch.no_grad()
def cluster_topk(scores, *, m0: int, new_center_rel, k: int, row_offset: int = 0):
    """Causal mask over the block's own centres + `topk(k, sorted=False)` of _cluster_compress
    (deltakv_less_memory.py:2780-2787); returns int32 [rows, k] column indices, best first."""
    assert scores.dim() == 2 and scores.stride(1) == 1
    rows, m = scores.shape
    out = torch.empty((rows, int(k)), dtype=torch.int32, device=scores.device)
    if new_center_rel is not None:
        assert new_center_rel.dtype == torch.int32 and new_center_rel.is_contiguous() and new_center_rel.numel() == m - int(m0)
    else:
        assert int(m0) == m
    lib = _lib.load()
    a = _lib.SvkClusterTopkArgs(scores=_lib.ptr(scores), new_center_rel=_lib.ptr(new_center_rel), topk=_lib.ptr(out),
                                score_stride=scores.stride(0), topk_stride=out.stride(0), rows=rows, m=m, m0=int(m0),
                                k=int(k), row_offset=int(row_offset), score_dtype=_dt(scores))
    _lib.check(lib.svk_cluster_topk(C.byref(a), _lib.current_stream_handle()), lib)
    return out


CLUSTER_L2_FUSED_MAX_ROWS = 1024


def cluster_l2_topk_supported(*, num_kv_heads: int, head_dim: int, dtype, rows: int | None = None) -> bool:
    """Shapes `svk_cluster_l2_topk` serves and wins at (the rest keeps the library product + `cluster_topk`): bf16 rows of at
    most 1024 values, and token blocks of at most 1024 rows - a wave re-streams every centre tile for its 16 tokens, so a
    decode-time eviction (128 tokens per row) runs 3-4x faster than gather + GEMM + top-k (44 against 138 us at 8000 centres)
    while a prefill-sized block (2048+ rows) is L2-bound and loses to the library GEMM (303 against 207 us)."""
    half = int(num_kv_heads) * int(head_dim)
    if rows is not None and int(rows) > CLUSTER_L2_FUSED_MAX_ROWS:
        return False
    return dtype == torch.bfloat16 and half % 32 == 0 and 0 < half <= 512


_CLUSTER_WS: dict = {}


@torch.no_grad()
def cluster_l2_topk(tokens, k_cache, v_cache, center_slots, *, m0: int, new_center_rel, k: int, row_offset: int = 0):
    """MI355X: `_metric_l2` (deltakv_base.py:2168-2190) + causal mask + top-k (deltakv_less_memory.py:2766-2787) of
    `_cluster_compress` in one MFMA launch over the layer caches - the [rows, m] score matrix is never written (the
    reference's fused form: `deltakv_l2_topk_blockwise`, kernels/triton/deltakv_kernels.py:3945-4134).  tokens [rows,
    2*Hkv*D] bf16; k_cache / v_cache [slots, Hkv, D] bf16; center_slots [m] int32 (every centre column's slot, the
    block's own centres last); returns int32 [rows, k] column indices, best first, lower column on ties."""
    assert tokens.dim() == 2 and tokens.dtype == torch.bfloat16 and tokens.stride(1) == 1
    assert k_cache.dtype == torch.bfloat16 and k_cache.dim() == 3 and k_cache.stride() == v_cache.stride()
    H, D = int(k_cache.shape[1]), int(k_cache.shape[2])
    assert k_cache.stride(2) == 1 and k_cache.stride(1) == D and int(tokens.shape[1]) == 2 * H * D
    assert center_slots.dtype == torch.int32 and center_slots.is_contiguous()
    rows, m = int(tokens.shape[0]), int(center_slots.numel())
    if new_center_rel is not None:
        assert new_center_rel.dtype == torch.int32 and new_center_rel.is_contiguous() and new_center_rel.numel() == m - int(m0)
    else:
        assert int(m0) == m
    out = torch.empty((rows, int(k)), dtype=torch.int32, device=tokens.device)
    lib = _lib.load()
    need = int(lib.svk_cluster_l2_topk_workspace_bytes(rows, m, int(k)))
    ws = _CLUSTER_WS.get(tokens.device.index)          # grow-only scratch; eviction runs outside graph capture
    if ws is None or ws.numel() < need:
        ws = _CLUSTER_WS[tokens.device.index] = torch.empty((max(need, 1 << 20),), dtype=torch.uint8, device=tokens.device)
    a = _lib.SvkClusterL2TopkArgs(
        tokens=_lib.ptr(tokens), k_cache=_lib.ptr(k_cache), v_cache=_lib.ptr(v_cache), center_slots=_lib.ptr(center_slots),
        new_center_rel=_lib.ptr(new_center_rel), topk=_lib.ptr(out), workspace=_lib.ptr(ws), workspace_bytes=int(ws.numel()),
        token_stride=tokens.stride(0), kv_slot_stride=k_cache.stride(0), topk_stride=out.stride(0), rows=rows, m=m,
        m0=int(m0), k=int(k), row_offset=int(row_offset), half_dim=H * D)
    _lib.check(lib.svk_cluster_l2_topk(C.byref(a), _lib.current_stream_handle()), lib)
    return out


@torch.no_grad()
def gather_mean_fathers(k_cache, v_cache, center_slots, topk, *, k_out: int | None = None):
    """base = mean over fathers of concat(K[slot], V[slot]) (deltakv_less_memory.py:2788-2789, batch_gather_mean
    deltakv_kernels.py:2268-2301) read from the layer cache; also returns father slots [rows, k_out] padded with
    the first father (deltakv_less_memory.py:3721-3727)."""
    assert k_cache.dtype == torch.bfloat16 and k_cache.stride() == v_cache.stride() and k_cache.stride(2) == 1
    assert center_slots.dtype == torch.int32 and center_slots.is_contiguous()
    assert topk.dtype == torch.int32 and topk.stride(1) == 1
    rows, k = topk.shape
    H, D = int(k_cache.shape[1]), int(k_cache.shape[2])
    k_out = int(k_out or k)
    base = torch.empty((rows, 2 * H * D), dtype=torch.bfloat16, device=k_cache.device)
    fathers = torch.empty((rows, k_out), dtype=torch.int32, device=k_cache.device)
    lib = _lib.load()
    a = _lib.SvkGatherMeanArgs(k_cache=_lib.ptr(k_cache), v_cache=_lib.ptr(v_cache), center_slots=_lib.ptr(center_slots),
                               topk=_lib.ptr(topk), base=_lib.ptr(base), father_slots=_lib.ptr(fathers),
                               kv_slot_stride=k_cache.stride(0), kv_head_stride=k_cache.stride(1), topk_stride=topk.stride(0),
                               base_stride=base.stride(0), father_stride=fathers.stride(0), rows=rows, k=k,
                               k_out=k_out, num_kv_heads=H, head_dim=D)
    _lib.check(lib.svk_gather_mean_fathers(C.byref(a), _lib.current_stream_handle()), lib)
    return base, fathers
