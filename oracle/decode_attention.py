"""Oracle: split-KV GQA decode attention with fused token scores.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates, tile for tile, the reference Triton kernels
  kernels/triton/gqa_flash_decoding_stage1.py:6-121   (unscored stage 1)
  kernels/triton/gqa_flash_decoding_stage1.py:124-208 (3-D per-head raw scores)
  kernels/triton/gqa_flash_decoding_stage1.py:211-295 (2-D head-max raw scores)
  kernels/triton/flash_decoding_stage2.py:8-46        (LSE merge)
and the H2O per-layer normalisation of engine/sparse_controller.py:748-768.

All tensors are float32 numpy arrays holding bf16-representable values where the
reference tensor is bf16 (q, k, v).  The only bf16 rounding inside the kernels is
`exp_logic.to(v.dtype)` before P·V (gqa_flash_decoding_stage1.py:280).
"""

from __future__ import annotations

import numpy as np

from .bf16 import bf16_round

NEG_INF = np.float32(-np.inf)


def flash_decode_stage1(
    q: np.ndarray,               # [B, Hq, D]
    k: np.ndarray,               # [slots, Hkv, D]
    v: np.ndarray,               # [slots, Hkv, D]
    req_to_tokens: np.ndarray,   # [rows, max_len] int32
    b_req_idx: np.ndarray,       # [B] int32
    b_seqlen: np.ndarray,        # [B] int32
    max_len_in_batch: int,
    block_seq: int,
    *,
    block_n: int = 16,
    attn_score: np.ndarray | None = None,   # [B, W] (2-D, max-reduced in place) or [B, Hq, W]
    p_dtype_bf16: bool = True,
    token_valid: np.ndarray | None = None,  # [B, max_len] bool: tokens a caller's view excludes (KIVI maps, see oracle/kivi.py)
):
    """Returns (mid_o [B,Hq,nblk,D] f32, mid_lse [B,Hq,nblk] f32).

    Follows gqa_flash_decoding_stage1.py:228-295: grid (B, Hkv, nblk); per program
    an online softmax over BLOCK_N-token tiles of its BLOCK_SEQ block.  Empty
    blocks write o=0, lse=-inf (:288-294).  `attn_score` (if 2-D) receives
    max(old, max_over_q_heads(raw q.k)) for every valid token (:262-268); if 3-D it
    receives the raw logits per q head (:167-173).  Raw = before sm_scale.
    """
    assert block_seq % block_n == 0
    B, Hq, D = q.shape
    Hkv = k.shape[1]
    G = Hq // Hkv
    sm_scale = np.float32(1.0 / (D ** 0.5))
    nblk = (int(max_len_in_batch) + block_seq - 1) // block_seq
    tiles = block_seq // block_n

    mid_o = np.zeros((B, Hq, nblk, D), dtype=np.float32)
    mid_lse = np.full((B, Hq, nblk), NEG_INF, dtype=np.float32)

    q = q.astype(np.float32, copy=False)
    for b in range(B):
        L = int(b_seqlen[b])
        row = req_to_tokens[int(b_req_idx[b])]
        # token index grid [nblk, tiles, block_n]
        t_idx = (np.arange(nblk)[:, None, None] * block_seq
                 + np.arange(tiles)[None, :, None] * block_n
                 + np.arange(block_n)[None, None, :])
        valid = t_idx < L                                  # masks both seq end and block end
        if token_valid is not None:
            tv = np.asarray(token_valid[b], dtype=bool)
            valid = valid & tv[np.minimum(t_idx, tv.shape[0] - 1)]
        safe_t = np.where(valid, t_idx, 0)
        safe_t = np.minimum(safe_t, row.shape[0] - 1)
        slots = np.where(valid, row[safe_t], 0).astype(np.int64)   # other=0 (:254-255)
        qb = q[b].reshape(Hkv, G, D)

        m = np.full((nblk, Hkv, G), NEG_INF, dtype=np.float32)
        l = np.zeros((nblk, Hkv, G), dtype=np.float32)
        acc = np.zeros((nblk, Hkv, G, D), dtype=np.float32)
        touched = np.zeros((nblk,), dtype=bool)

        for j in range(tiles):
            vj = valid[:, j, :]                            # [nblk, n]
            has = vj.any(axis=1)                           # tile entered iff start_n < block_n_size
            if not has.any():
                continue
            sj = slots[:, j, :]                            # [nblk, n]
            kj = np.where(vj[:, :, None, None], k[sj], np.float32(0))   # [nblk, n, Hkv, D]
            vv = np.where(vj[:, :, None, None], v[sj], np.float32(0))
            # att[nblk, Hkv, G, n] = q . k   (fp32 accumulate, tl.dot :259)
            att = np.einsum("hgd,bnhd->bhgn", qb, kj, dtype=np.float32, optimize=True)
            att = np.where(vj[:, None, None, :], att, NEG_INF)
            if attn_score is not None:
                tt = t_idx[:, j, :]
                if attn_score.ndim == 2:
                    hm = att.max(axis=(1, 2))              # max over all q heads -> [nblk, n]
                    sel = vj
                    cur = attn_score[b, tt[sel]]
                    attn_score[b, tt[sel]] = np.maximum(cur, hm[sel])
                else:
                    a3 = att.reshape(nblk, Hq, block_n)
                    for blk in np.nonzero(has)[0]:
                        cols = tt[blk][vj[blk]]
                        attn_score[b, :, cols] = a3[blk][:, vj[blk]].T
            att = att * sm_scale
            cur_max = att.max(axis=3)
            new_m = np.maximum(cur_max, m)
            upd = has[:, None, None]
            safe_new_m = np.where(np.isfinite(new_m), new_m, np.float32(0))
            p = np.exp(att - safe_new_m[..., None], dtype=np.float32)
            p = np.where(vj[:, None, None, :], p, np.float32(0))
            scale = np.where(np.isfinite(m), np.exp(m - safe_new_m, dtype=np.float32), np.float32(0))
            pv_in = bf16_round(p) if p_dtype_bf16 else p
            pv = np.einsum("bhgn,bnhd->bhgd", pv_in, vv, dtype=np.float32, optimize=True)
            new_acc = acc * scale[..., None] + pv
            new_l = l * scale + p.sum(axis=3, dtype=np.float32)
            acc = np.where(upd[..., None], new_acc, acc)
            l = np.where(upd, new_l, l)
            m = np.where(upd, new_m, m)
            touched |= has

        safe_l = np.where(touched[:, None, None], l, np.float32(1))
        o = np.where(touched[:, None, None, None], acc / safe_l[..., None], np.float32(0))
        with np.errstate(divide="ignore"):
            lse = np.where(touched[:, None, None], m + np.log(safe_l, dtype=np.float32), NEG_INF)
        mid_o[b] = o.reshape(nblk, Hq, D).transpose(1, 0, 2)
        mid_lse[b] = lse.reshape(nblk, Hq).T
    return mid_o, mid_lse


def flash_decode_stage2(
    mid_o: np.ndarray,       # [B, Hq, nblk, D]
    mid_lse: np.ndarray,     # [B, Hq, nblk]
    b_seqlen: np.ndarray,    # [B]
    block_seq: int,
) -> np.ndarray:
    """LSE-weighted merge, flash_decoding_stage2.py:19-46.  Returns o [B,Hq,D] f32.

    (The reference stores into `torch.empty_like(q)`, i.e. rounds to bf16 on store;
    callers apply `bf16_round` when comparing with a bf16 output.)
    """
    B, Hq, nblk, D = mid_o.shape
    o = np.zeros((B, Hq, D), dtype=np.float32)
    for b in range(B):
        L = int(b_seqlen[b])
        n = 0 if L <= 0 else (L + block_seq - 1) // block_seq
        s = np.zeros((Hq,), dtype=np.float32)
        m = np.full((Hq,), NEG_INF, dtype=np.float32)
        acc = np.zeros((Hq, D), dtype=np.float32)
        for blk in range(n):
            tv = mid_o[b, :, blk]
            tl_ = mid_lse[b, :, blk]
            new_m = np.maximum(tl_, m)
            with np.errstate(invalid="ignore"):
                old_scale = np.exp(m - new_m, dtype=np.float32)
                e = np.exp(tl_ - new_m, dtype=np.float32)
            old_scale = np.where(np.isnan(old_scale), np.float32(0), old_scale)
            e = np.where(np.isnan(e), np.float32(0), e)
            acc = acc * old_scale[:, None] + e[:, None] * tv
            s = s * old_scale + e
            m = new_m
        with np.errstate(invalid="ignore", divide="ignore"):
            o[b] = acc / s[:, None]
    return o


def decode_attention_dense(q, k, v, req_to_tokens, b_req_idx, b_seqlen):
    """Un-tiled float64 ground truth (same role as the reference's
    scripts/validation/test_gqa_flash_decoding_score.py:66-133).  Returns
    (o [B,Hq,D] f64, raw_logits list of [Hq, L] f64)."""
    B, Hq, D = q.shape
    Hkv = k.shape[1]
    G = Hq // Hkv
    outs = np.zeros((B, Hq, D), dtype=np.float64)
    raws = []
    for b in range(B):
        L = int(b_seqlen[b])
        sl = req_to_tokens[int(b_req_idx[b]), :L].astype(np.int64)
        kk = k[sl].astype(np.float64)          # [L, Hkv, D]
        vv = v[sl].astype(np.float64)
        qb = q[b].astype(np.float64).reshape(Hkv, G, D)
        raw = np.einsum("hgd,lhd->hgl", qb, kk)
        raws.append(raw.reshape(Hq, L))
        if L == 0:
            continue
        s = raw / np.sqrt(D)
        s = s - s.max(axis=2, keepdims=True)
        p = np.exp(s)
        p /= p.sum(axis=2, keepdims=True)
        outs[b] = np.einsum("hgl,lhd->hgd", p, vv).reshape(Hq, D)
    return outs, raws


def h2o_normalize_decode_scores(attn_score: np.ndarray, head_dim: int) -> np.ndarray:
    """engine/sparse_controller.py:762-767: `attn_score.mul_(D**-0.5);
    softmax(attn_score, dim=-1, out=attn_score)` on the [B, W] buffer that was
    pre-filled with -1e20 (sparse_controller.py:460).  Returns a new array."""
    x = attn_score.astype(np.float32) * np.float32(float(head_dim) ** -0.5)
    mx = x.max(axis=-1, keepdims=True)
    e = np.exp(x - mx, dtype=np.float32)
    return (e / e.sum(axis=-1, keepdims=True, dtype=np.float32)).astype(np.float32)
