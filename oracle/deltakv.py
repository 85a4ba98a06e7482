"""Oracle: DeltaKV decode-side kernels (TEST INFRASTRUCTURE ONLY).

Restates kernels/triton/deltakv_kernels.py
  deltakv_static_decode_plan                          :3854-3942 (kernel :3695-3851)
  deltakv_reconstruct_writeback_grouped_heads         :2909-3012 (kernel :2732-2907)
  deltakv_less_memory_reconstruct_writeback_quantized :3344-3450 (kernel :3173-3341)
and kernels/triton/quant.py
  triton_dequantize_2d_int4_grouped :160-216, unpack_tensor :304-324, unpack_quantized_to_16bit :326-349,
  the pack format of triton_quantize_and_pack_2d_int4_grouped :28-117 / _pack_along_last_dim :219-239
  (asymmetric min/max groups, round-half-even, LSB-first fields inside an int32).
"""

from __future__ import annotations

import numpy as np

from .bf16 import bf16_round


# ------------------------------------------------------------------------------------------------
# static decode plan
# ------------------------------------------------------------------------------------------------

def static_decode_plan(raw_slots_map, latent_slots_map, active_compressed, req_indices, context_lens, compressed_lens,
                       temp_slots, *, sink: int, max_buffer: int):
    """Returns dict(active_slots [B, S], active_pos [B, S], new_context_lens [B],
    recon_pos / recon_latent / recon_out_slot [B*K]), S = sink + K + max_buffer."""
    B, K = active_compressed.shape
    S = sink + K + max_buffer
    max_pos = raw_slots_map.shape[1] - 1
    out_slot = np.zeros((B, S), np.int32)
    out_pos = np.zeros((B, S), np.int32)
    new_len = np.zeros((B,), np.int32)
    r_pos = np.full((B * K,), -1, np.int32)
    r_lat = np.full((B * K,), -1, np.int32)
    r_out = np.full((B * K,), -1, np.int32)
    cols = np.arange(S)
    for b in range(B):
        row = int(req_indices[b])
        ctx = int(context_lens[b])
        clen = int(compressed_lens[b])
        top_len = min(max(clen, 0), K)
        safe = max(int(raw_slots_map[row, 0]), 0) if sink > 0 else 0
        o_slot = np.full((S,), safe, np.int32)
        o_pos = np.zeros((S,), np.int32)
        if sink > 0:
            sp = np.minimum(cols[:sink], max_pos)
            o_slot[:sink] = raw_slots_map[row, sp]
            o_pos[:sink] = sp
        for j in range(K):
            c = sink + j
            rel = int(active_compressed[b, j])
            top_pos = rel + sink
            in_top = j < top_len
            valid = in_top and rel >= 0 and rel < clen and top_pos < ctx
            safe_pos = min(max(top_pos, 0), max_pos)
            raw = int(raw_slots_map[row, safe_pos]) if in_top else 0
            lat = int(latent_slots_map[row, safe_pos]) if in_top else -1
            tmp = int(temp_slots[b, j]) if in_top else 0
            need = valid and lat >= 0
            if in_top:
                o_slot[c] = tmp if need else (max(raw, 0) if valid else safe)
                o_pos[c] = top_pos if valid else 0
            r_pos[b * K + j] = top_pos if need else -1
            r_lat[b * K + j] = lat if need else -1
            r_out[b * K + j] = tmp if need else -1
        buf_start = sink + clen
        buf_len = min(max(ctx - buf_start, 0), max_buffer)
        start_out = sink + top_len
        for c in range(start_out, S):
            j = c - start_out
            pos = min(max(buf_start + j, 0), max_pos)
            ok = j < buf_len
            o_slot[c] = max(int(raw_slots_map[row, pos]), 0) if ok else safe
            o_pos[c] = pos if ok else 0
        out_slot[b], out_pos[b] = o_slot, o_pos
        new_len[b] = sink + top_len + buf_len
    return dict(active_slots=out_slot, active_pos=out_pos, new_context_lens=new_len, recon_pos=r_pos,
                recon_latent=r_lat, recon_out_slot=r_out)


# ------------------------------------------------------------------------------------------------
# quantisation format
# ------------------------------------------------------------------------------------------------

def round_half_even(x):
    return np.rint(x)          # numpy rint == round half to even


def quantize_pack_grouped(data: np.ndarray, group_size: int, bits: int):
    """data [n, d] f32 -> (code int32 [n, d*bits/32], scale [n, d/group], mn [n, d/group]) (f32 maths,
    quant.py:41-76 / :262-301: scale=(max-min)/(2^b-1), q=rhe(clamp((x-min)/(scale+1e-6))), LSB-first)."""
    n, d = data.shape
    fpi = 32 // bits
    g = data.reshape(n, d // group_size, group_size).astype(np.float32)
    mx, mn = g.max(axis=2), g.min(axis=2)
    scale = ((mx - mn) / np.float32(2 ** bits - 1)).astype(np.float32)
    norm = (g - mn[..., None]) / (scale[..., None] + np.float32(1e-6))
    qv = round_half_even(np.clip(norm, 0, 2 ** bits - 1)).astype(np.int64).reshape(n, d)
    code = np.zeros((n, d // fpi), np.int64)
    for j in range(fpi):
        code |= qv[:, j::fpi] << (j * bits)
    code = (code & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
    return code, scale, mn.astype(np.float32)


def unpack_codes(code: np.ndarray, bits: int) -> np.ndarray:
    """[n, p] int32 -> [n, p*32/bits] ints (quant.py:304-324)."""
    fpi = 32 // bits
    u = code.view(np.uint32).astype(np.int64)
    n, p = u.shape
    out = np.zeros((n, p * fpi), np.int64)
    for j in range(fpi):
        out[:, j::fpi] = (u >> (j * bits)) & ((1 << bits) - 1)
    return out


def dequantize_grouped(code, scale, mn, group_size: int, bits: int) -> np.ndarray:
    """q*scale + mn in fp32 (quant.py:120-157)."""
    q = unpack_codes(code, bits).astype(np.float32)
    n, d = q.shape
    s = np.repeat(scale.astype(np.float32), group_size, axis=1)
    m = np.repeat(mn.astype(np.float32), group_size, axis=1)
    return q * s + m


# ------------------------------------------------------------------------------------------------
# reconstruct + RoPE write-back
# ------------------------------------------------------------------------------------------------

def reconstruct_writeback(k_cache, v_cache, *, father_slots, slot_to_pos, out_slots, out_pos, cos_sin, delta=None,
                          packed=None, scale=None, mn=None, latent_slots=None, bits=0, group_size=0,
                          k_norm_weight=None, k_norm_eps=1e-6, raw_k_cache=False, store_raw_k=False,
                          out_bf16=True):
    """In place on k_cache/v_cache [slots, Hkv, D] (f32 holding bf16 values).  Dense form: `delta` [N, 2*Hkv*D]
    (valid entry: out_slot>=0 and out_pos>=0, kernel :2770-2777).  Quantised form: `packed/scale/mn` indexed by
    `latent_slots` (valid entry: latent_slot>=0, kernel :3219).  K = delta_K + mean_f(de-RoPE(father K)),
    optional RMS k-norm, RoPE at out_pos; V = delta_V + mean_f(father V)."""
    N, Kf = father_slots.shape
    Hkv, D = k_cache.shape[1], k_cache.shape[2]
    HD2 = D // 2
    Dtot = Hkv * D
    cs = cos_sin.astype(np.float32)
    for n in range(N):
        if delta is not None:
            if not (out_slots[n] >= 0 and out_pos[n] >= 0):
                continue
            dl = delta[n].astype(np.float32)
        else:
            ls = int(latent_slots[n])
            if ls < 0:
                continue
            dl = dequantize_grouped(packed[ls:ls + 1], scale[ls:ls + 1], mn[ls:ls + 1], group_size, bits)[0]
        dk = dl[:Dtot].reshape(Hkv, D)
        dv = dl[Dtot:].reshape(Hkv, D)
        acc_k1 = np.zeros((Hkv, HD2), np.float32); acc_k2 = np.zeros((Hkv, HD2), np.float32)
        acc_v = np.zeros((Hkv, D), np.float32)
        for kk in range(Kf):
            fs = int(father_slots[n, kk])
            y1 = k_cache[fs, :, :HD2].astype(np.float32)
            y2 = k_cache[fs, :, HD2:].astype(np.float32)
            if raw_k_cache:
                x1, x2 = y1, y2
            else:
                fp = int(slot_to_pos[fs])
                c, s_ = cs[fp, :HD2], cs[fp, HD2:]
                x1 = y1 * c + y2 * s_
                x2 = y2 * c - y1 * s_
            acc_k1 += x1; acc_k2 += x2
            acc_v += v_cache[fs].astype(np.float32)
        inv = np.float32(1.0 / Kf)
        k1 = dk[:, :HD2] + acc_k1 * inv
        k2 = dk[:, HD2:] + acc_k2 * inv
        vv = dv + acc_v * inv
        if k_norm_weight is not None and not store_raw_k:
            w = k_norm_weight.astype(np.float32)
            var = (k1 * k1 + k2 * k2).sum(axis=1) / np.float32(D)
            rstd = 1.0 / np.sqrt(var + np.float32(k_norm_eps))
            k1 = k1 * rstd[:, None] * w[None, :HD2]
            k2 = k2 * rstd[:, None] * w[None, HD2:]
        if store_raw_k:
            o1, o2 = k1, k2
        else:
            op = int(out_pos[n])
            c, s_ = cs[op, :HD2], cs[op, HD2:]
            o1 = k1 * c - k2 * s_
            o2 = k2 * c + k1 * s_
        ko = np.concatenate((o1, o2), axis=1).astype(np.float32)
        os_ = int(out_slots[n])
        k_cache[os_] = bf16_round(ko) if out_bf16 else ko
        v_cache[os_] = bf16_round(vv) if out_bf16 else vv


# ------------------------------------------------------------------------------------------------
# attention-facing copy of a sparse layer's active slots (deltakv_kernels.py:3489-3693)
# ------------------------------------------------------------------------------------------------

def materialize_sparse_view(active_slots, slot_to_pos, k_cache, v_cache, cos_sin, *, postrope_mask=None,
                            k_norm_weight=None, k_norm_eps=1e-6, out_bf16=True):
    """-> (out_k, out_v) [B*W, Hkv, D]: entry n = b*W + w copies slot active_slots[b, w] (clamped into range);
    K is k-normed (optional) and rotated at slot_to_pos[slot] (clamped at 0) unless the slot is flagged
    post-RoPE, in which case it is copied verbatim (kernel :3636-3681).  Every entry is written (no length mask)."""
    B, W = active_slots.shape
    S, Hkv, D = k_cache.shape
    HD2 = D // 2
    flat = np.clip(active_slots.reshape(-1).astype(np.int64), 0, S - 1)
    valid = (active_slots.reshape(-1) >= 0) & (active_slots.reshape(-1) < S)
    pos = np.maximum(slot_to_pos[flat].astype(np.int64), 0)
    k = k_cache[flat].astype(np.float32)
    k1, k2 = k[..., :HD2], k[..., HD2:]
    if k_norm_weight is not None:
        w = k_norm_weight.astype(np.float32)
        var = (k1 * k1 + k2 * k2).sum(axis=-1) / np.float32(D)
        rstd = (1.0 / np.sqrt(var + np.float32(k_norm_eps))).astype(np.float32)
        n1 = k1 * rstd[..., None] * w[None, None, :HD2]
        n2 = k2 * rstd[..., None] * w[None, None, HD2:]
    else:
        n1, n2 = k1, k2
    cs = cos_sin.astype(np.float32)[pos]
    c, s_ = cs[:, None, :HD2], cs[:, None, HD2:]
    r1 = n1 * c - n2 * s_
    r2 = n2 * c + n1 * s_
    if postrope_mask is not None:
        already = (postrope_mask[flat].astype(bool) & valid)[:, None, None]
        r1 = np.where(already, k1, r1)
        r2 = np.where(already, k2, r2)
    out_k = np.concatenate((r1, r2), axis=-1).astype(np.float32)
    out_v = v_cache[flat].astype(np.float32)
    return (bf16_round(out_k) if out_bf16 else out_k), out_v


# ------------------------------------------------------------------------------------------------
# query-aware top-k over the compressed range (sparse_controller.py:255-299, :1813-1822)
# ------------------------------------------------------------------------------------------------

def decode_softmax_token_scores(raw_scores, *, sink: int, compressed_lens, scale: float):
    """raw [B, H, L] head logits -> [B, Lc] : scale, mask to the compressed range [sink, sink+clen),
    softmax over it per head, max over heads (positions outside get 0)."""
    B, H, L = raw_scores.shape
    out = np.zeros((B, L - sink), np.float32)
    for b in range(B):
        c = int(compressed_lens[b])
        if c <= 0:
            continue
        x = raw_scores[b, :, sink:sink + c].astype(np.float32) * np.float32(scale)
        x = x - x.max(axis=1, keepdims=True)
        e = np.exp(x, dtype=np.float32)
        p = e / e.sum(axis=1, keepdims=True, dtype=np.float32)
        out[b, :c] = p.max(axis=0)
    return out


def token_scores_full(raw_scores, *, candidate_start: int, candidate_lens, scale: float, model_dtype: str = "bfloat16"):
    """sparse_controller.py:255-299 exactly as the caller sees it: [B, H, L] raw logits -> [B, L] token scores.
    Candidates are positions [candidate_start, candidate_start + len): logits * scale, per-head softmax over the
    candidates, max over heads, cast to the model dtype (bf16 / fp16 checkpoints; fp32 stays), every other
    position = finfo(dtype).min.  Returned as float32 holding the dtype's values."""
    raw = np.asarray(raw_scores, np.float32)
    B, H, L = raw.shape
    cs = int(candidate_start)
    fill = {"bfloat16": np.float32(-3.3895313892515355e38), "float32": np.finfo(np.float32).min}[model_dtype]
    out = np.full((B, L), fill, np.float32)
    lens = np.clip(np.asarray(candidate_lens, np.int64), 0, L - cs)
    for b in range(B):
        c = int(lens[b])
        if c <= 0:
            continue
        x = raw[b, :, cs:cs + c] * np.float32(scale)
        x = x - x.max(axis=1, keepdims=True)
        e = np.exp(x, dtype=np.float32)
        p = (e / e.sum(axis=1, keepdims=True, dtype=np.float32)).max(axis=0)
        out[b, cs:cs + c] = bf16_round(p) if model_dtype == "bfloat16" else p
    return out


def dynamic_topk_keys(token_scores, *, sink: int, compressed_lens, tiebreak: bool = False, model_dtype: str = "bfloat16"):
    """The keys the reference's topk ranks (sparse_controller.py:1784-1811): token_scores[:, sink:] with
    positions >= compressed_len set to -1e10 (stored in the score tensor's dtype: bf16 holds -9.999221e9); with the
    deterministic tie-break, `float(s) + max(|s|, 1) * (pos / n * 1e-6)` first (fp32 op by op, so LATER positions
    win among equal scores), masked again with the fp32 -1e10."""
    s = np.asarray(token_scores, np.float32)[:, sink:].copy()
    n = s.shape[1]
    mask = np.arange(n)[None, :] >= np.asarray(compressed_lens, np.int64)[:, None]
    s[mask] = bf16_round(np.float32(-1e10)) if model_dtype == "bfloat16" else np.float32(-1e10)
    if tiebreak and s.size:
        pos = np.arange(n, dtype=np.float32) / np.float32(max(1, n))
        key = (pos * np.float32(1.0e-6))[None, :]
        s = s + np.maximum(np.abs(s), np.float32(1.0)) * key
        s[mask] = np.float32(-1e10)
    return s


def dynamic_topk_indices(token_scores, *, sink: int, compressed_lens, keep: int, tiebreak: bool = False,
                         model_dtype: str = "bfloat16"):
    """topk(min(keep, n), sorted=True) over `dynamic_topk_keys` -> int32 [B, k] relative to `sink`; equal keys in
    ascending position (torch leaves that order unspecified — compare with `check_sorted_topk`)."""
    keys = dynamic_topk_keys(token_scores, sink=sink, compressed_lens=compressed_lens, tiebreak=tiebreak,
                             model_dtype=model_dtype)
    k = min(int(keep), keys.shape[1])
    if k <= 0:
        return np.zeros((keys.shape[0], 0), np.int32)
    return np.stack([np.argsort(-keys[b], kind="stable")[:k] for b in range(keys.shape[0])]).astype(np.int32)


def check_sorted_topk(keys_row: np.ndarray, got: np.ndarray, ref: np.ndarray) -> bool:
    """Two sorted top-k results over the same key row agree up to topk's freedom: the same key at every rank,
    distinct indices, and the same index set for every key value strictly above the last rank's (indices that
    share a key may be permuted among themselves, and at the last key value either may hold any of the tied
    positions).  Returns True when the index arrays are identical."""
    got, ref = np.asarray(got).reshape(-1), np.asarray(ref).reshape(-1)
    assert got.shape == ref.shape
    assert np.unique(got).size == got.size, "duplicate indices"
    np.testing.assert_array_equal(keys_row[got], keys_row[ref])
    if np.array_equal(got, ref):
        return True
    if got.size:
        last = keys_row[ref[-1]]
        above_g = np.sort(got[keys_row[got] > last])
        above_r = np.sort(ref[keys_row[ref] > last])
        np.testing.assert_array_equal(above_g, above_r)
    return False
