"""Oracle: DeltaKV compression side (TEST INFRASTRUCTURE ONLY) - SURVEY section 8 a26.

Restates
  kernels/triton/quant.py:29-117      _quantize_pack_2d_int4_grouped_kernel / triton_quantize_and_pack_2d_int4_grouped
  kernels/triton/quant.py:243-301     triton_quantize_and_pack_along_last_dim (torch arithmetic between two Triton passes)
  engine/cache_manager/deltakv_less_memory.py:1741-1780  _store_full_layer_kivi_blocks (per-channel K, per-token V)
  engine/cache_manager/deltakv_base.py:2168-2190         _metric_l2
  engine/cache_manager/deltakv_less_memory.py:2719-2802  _cluster_compress (causal top-k fathers + mean base)
  engine/cache_manager/deltakv_less_memory.py:2181-2240  _deltakv_store_layer_latent (down(kv) - down(base))

Both quantisers do their arithmetic in the *storage dtype* (every operation rounds to it): `rnd` is that rounding
(identity for fp32, fp16 in the interpreter fixtures, bf16 in deployment).
"""

from __future__ import annotations

import numpy as np

from .bf16 import bf16_round
from .deltakv import round_half_even


def rounding(name: str):
    if name in ("f32", "float32"):
        return lambda x: np.asarray(x, dtype=np.float32)
    if name in ("f16", "float16"):
        return lambda x: np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float32)
    if name in ("bf16", "bfloat16"):
        return lambda x: bf16_round(np.asarray(x, dtype=np.float32))
    raise ValueError(name)


def quantize_groups(groups: np.ndarray, bits: int, rnd, flavor: str):
    """groups [..., g] (values representable in the storage dtype) -> (q int [..., g], scale [...], mn [...]).
    scale = (max - min) / (2^b - 1); q = round_half_even(clamp((x - min) / (scale + 1e-6), 0, 2^b - 1)).
    Where the storage-dtype roundings fall depends on who does the arithmetic (pinned by the fp16 fixtures):
      "triton2d" (quant.py:53-64): tl.max / tl.min promote to fp32, so scale = rnd((max - min) / 15) once; `x - min` is a
                 storage-dtype subtraction (rounded), the quotient by the fp32 `scale + 1e-6` stays fp32;
      "torch"    (quant.py:283-287): every tensor op rounds: rnd(rnd(max - min) / 15), rnd(rnd(x - min) / rnd(scale + 1e-6))."""
    g = np.asarray(groups, dtype=np.float32)
    mx, mn = g.max(axis=-1), g.min(axis=-1)
    qmax = np.float32(2 ** bits - 1)
    if flavor == "triton2d":
        scale = rnd((mx - mn) / qmax)
        norm = rnd(g - mn[..., None]) / (scale + np.float32(1e-6))[..., None]
    elif flavor == "torch":
        scale = rnd(rnd(mx - mn) / qmax)
        norm = rnd(rnd(g - mn[..., None]) / rnd(scale + np.float32(1e-6))[..., None])
    else:
        raise ValueError(flavor)
    q = round_half_even(np.clip(norm, 0, qmax)).astype(np.int64)
    return q, scale.astype(np.float32), mn.astype(np.float32)


def pack_last_dim(q: np.ndarray, bits: int) -> np.ndarray:
    """[..., n] ints -> [..., n*bits/32] int32, element j of a word at bit j*bits (quant.py:66-76, :219-240)."""
    fpi = 32 // bits
    qq = q.reshape(q.shape[:-1] + (q.shape[-1] // fpi, fpi)).astype(np.int64)
    word = np.zeros(qq.shape[:-1], np.int64)
    for j in range(fpi):
        word |= qq[..., j] << (j * bits)
    return (word & 0xFFFFFFFF).astype(np.uint32).view(np.int32)


def quantize_pack_2d(data: np.ndarray, group_size: int, bits: int, rnd):
    """[n, d] -> (code [n, d*bits/32] int32, scale [n, d/group], mn [n, d/group])."""
    n, d = data.shape
    q, scale, mn = quantize_groups(data.reshape(n, d // group_size, group_size), bits, rnd, "triton2d")
    return pack_last_dim(q.reshape(n, d), bits), scale, mn


def kivi_quantize_blocks(k: np.ndarray, v: np.ndarray, value_group: int, rnd):
    """k, v [blocks, G, H, D] -> the KIVI block tensors of _store_full_layer_kivi_blocks:
    key_packed [blocks, H, D, G/8], key_scale/min [blocks, H, D] (one group = the block's G tokens of a channel),
    value_packed [blocks, H, G, D/8], value_scale/min [blocks, H, G, D/value_group]."""
    blocks, G, H, D = k.shape
    ks = np.transpose(k, (0, 2, 3, 1))                      # [blocks, H, D, G]
    qk, sk, mk = quantize_groups(ks[..., None, :], 4, rnd, "torch")  # one group of G tokens
    key_packed = pack_last_dim(qk.reshape(blocks, H, D, G), 4)
    vs = np.transpose(v, (0, 2, 1, 3))                      # [blocks, H, G, D]
    qv, sv, mv = quantize_groups(vs.reshape(blocks, H, G, D // value_group, value_group), 4, rnd, "torch")
    value_packed = pack_last_dim(qv.reshape(blocks, H, G, D), 4)
    return dict(key_packed=key_packed, key_scales=sk[..., 0], key_mins=mk[..., 0], value_packed=value_packed,
                value_scales=sv, value_mins=mv)


def l2_scores(kv: np.ndarray, centers: np.ndarray, rnd) -> np.ndarray:
    """deltakv_base.py:2168-2190: 2*dot(a, b) - ||b||^2 with the [N, M] matrix kept in the storage dtype."""
    dot = rnd(kv.astype(np.float32) @ centers.astype(np.float32).T)
    # `(b * b).sum(dim=1, dtype=torch.float32)`: the product is a tensor of the storage dtype (each square rounded),
    # the sum runs in fp32, `.to(dot.dtype)` rounds it once more
    b_norm = rnd(rnd(centers.astype(np.float32) ** 2).sum(axis=1, dtype=np.float32))
    return rnd(dot * np.float32(2.0) - b_norm[None, :])


def cluster_compress(kv: np.ndarray, existing: np.ndarray, new_center_rel: np.ndarray, k_neighbors: int, rnd=None):
    """kv [n, kv_dim], existing centres [m0, kv_dim] -> (scores [n, m0+m_new] with the causal mask applied,
    top-k indices [n, k_eff] (one valid choice: best score first, lower index among ties), base [n, kv_dim])."""
    rnd = rnd or rounding("bf16")
    new_centers = kv[np.asarray(new_center_rel, dtype=np.int64)]
    allc = np.concatenate((existing, new_centers), axis=0)
    m0 = existing.shape[0]
    scores = l2_scores(kv, allc, rnd)
    rows = np.arange(kv.shape[0])[:, None]
    scores[:, m0:] = np.where(np.asarray(new_center_rel)[None, :] <= rows, scores[:, m0:], -np.inf)
    k_eff = min(int(k_neighbors), allc.shape[0])
    topk = np.argsort(-scores, axis=1, kind="stable")[:, :k_eff]
    return scores, topk, mean_of_rows(allc, topk, rnd), allc


def mean_of_rows(rows: np.ndarray, index: np.ndarray, rnd) -> np.ndarray:
    """`all_centers.gather(...).mean(dim=2)` in the storage dtype: fp32 accumulation, one rounding."""
    return rnd(rows[index].astype(np.float32).sum(axis=1, dtype=np.float32) / np.float32(index.shape[1]))
