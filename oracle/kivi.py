"""Oracle: full-layer KIVI-int4 decode stage 1 (TEST INFRASTRUCTURE ONLY).

Restates full_layer_kivi_flash_decode_stage1, kernels/triton/deltakv_kernels.py:973-1142 (kernel :675-929):
token t of row r is either raw (raw_slots_map[r,t] >= 0 -> bf16 K/V rows) or lives in KIVI block
b = kivi_block_slots_map[r,t] at local index t - kivi_block_start_pos[b] in [0, group_size):
  K[t, h, d] = code_K[b, h, d, local]  * key_scales[b, h, d]          + key_mins[b, h, d]        (per channel)
  V[t, h, d] = code_V[b, h, local, d]  * value_scales[b, h, local, d/G] + value_mins[...]        (per token)
codes are 4-bit fields packed LSB-first, 8 per int32.  Dequantised values are cast to q's dtype
(`.to(q.dtype)`, :852/:888) and the rest is the ordinary split-KV decode with optional 3-D raw scores.
"""

from __future__ import annotations

import numpy as np

from .bf16 import bf16_round
from .decode_attention import flash_decode_stage1


def dequant_tokens(row: int, length: int, raw_k, raw_v, raw_slots_map, kivi_block_slots_map, kivi_block_start_pos,
                   key_packed, key_scales, key_mins, value_packed, value_scales, value_mins, group_size: int,
                   round_bf16: bool):
    """-> K, V [length, Hkv, D] f32 for the tokens of one row, valid [length] bool.  A token is valid when it is raw
    (raw_slots_map >= 0) or maps to a KIVI block that contains its position (deltakv_kernels.py:806-823); every other
    token is excluded from the softmax and gets no score (`slot_valid`, :823, :853-866)."""
    Hkv, D = raw_k.shape[1], raw_k.shape[2]
    K = np.zeros((length, Hkv, D), np.float32)
    V = np.zeros((length, Hkv, D), np.float32)
    valid = np.ones((length,), bool)
    kp = key_packed.view(np.uint32).astype(np.int64)
    vp = value_packed.view(np.uint32).astype(np.int64)
    d = np.arange(D)
    for t in range(length):
        rs = int(raw_slots_map[row, t])
        if rs >= 0:
            K[t], V[t] = raw_k[rs], raw_v[rs]
            continue
        b = int(kivi_block_slots_map[row, t])
        lt = t - int(kivi_block_start_pos[max(b, 0)])
        if b < 0 or not 0 <= lt < group_size:
            valid[t] = False
            continue
        kq = ((kp[b, :, :, lt // 8] >> ((lt % 8) * 4)) & 15).astype(np.float32)            # [Hkv, D]
        kk = kq * key_scales[b].astype(np.float32) + key_mins[b].astype(np.float32)
        vq = ((vp[b, :, lt, :][:, d // 8] >> ((d % 8) * 4)[None, :]) & 15).astype(np.float32)
        vv = vq * value_scales[b, :, lt, :][:, d // group_size].astype(np.float32) + \
            value_mins[b, :, lt, :][:, d // group_size].astype(np.float32)
        K[t] = bf16_round(kk) if round_bf16 else kk
        V[t] = bf16_round(vv) if round_bf16 else vv
    return K, V, valid


def full_layer_kivi_flash_decode_stage1(*, q, raw_k, raw_v, raw_slots_map, kivi_block_slots_map, kivi_block_start_pos,
                                        key_packed, key_scales, key_mins, value_packed, value_scales, value_mins,
                                        req_indices, context_lens, max_len_in_batch, group_size, block_seq,
                                        attn_score=None, round_bf16=True, p_dtype_bf16=True):
    """Returns (mid_o, mid_lse); attn_score [B, Hq, L] (if given) receives the raw logits."""
    B = q.shape[0]
    Hkv, D = raw_k.shape[1], raw_k.shape[2]
    lens = np.asarray(context_lens, dtype=np.int32)
    tot = int(lens.sum())
    Kp = np.zeros((max(tot, 1), Hkv, D), np.float32)
    Vp = np.zeros_like(Kp)
    table = np.zeros((B, int(max_len_in_batch)), np.int32)
    token_valid = np.ones((B, int(max_len_in_batch)), bool)
    off = 0
    for b in range(B):
        n = int(lens[b])
        K, V, token_valid[b, :n] = dequant_tokens(int(req_indices[b]), n, raw_k, raw_v, raw_slots_map, kivi_block_slots_map,
                              kivi_block_start_pos, key_packed, key_scales, key_mins, value_packed, value_scales,
                              value_mins, group_size, round_bf16)
        Kp[off: off + n], Vp[off: off + n] = K, V
        table[b, :n] = np.arange(off, off + n)
        off += n
    return flash_decode_stage1(q, Kp, Vp, table, np.arange(B, dtype=np.int32), lens, int(max_len_in_batch), block_seq,
                               attn_score=attn_score, p_dtype_bf16=p_dtype_bf16,
                               token_valid=None if token_valid.all() else token_valid)
