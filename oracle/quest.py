"""Oracle: Quest page metadata, page scoring, top-k view (TEST INFRASTRUCTURE ONLY).

Restates engine/cache_manager/quest.py:
  page allocator        _allocate :1227-1277, _allocate_batch :1279-1360, free_seq :1379-1420
  page min/max metadata on_kv_stored :1607-1685, on_forward_end :1718-1771
  page scoring          _score_pages_batched :1773-1802
  decode view           build_decode_view / _build_decode_view_static :1804-1913

bf16 tensors are float32 arrays holding bf16 values.  torch.bmm on bf16 returns bf16, and the
in-place `+=` rounds again, so page scores are bf16 values: bf16(bf16(q+ . max) + bf16(q- . min)).
"""

from __future__ import annotations

import numpy as np

from .bf16 import bf16_round


def page_minmax(k_cache: np.ndarray, page_slots: np.ndarray, page_size: int):
    """k_cache [slots, Hkv, D] -> (page_max, page_min) [n, Hkv, D] for whole pages
    (quest.py:1754-1771: aminmax over the page's 16 token rows; exact)."""
    idx = (np.asarray(page_slots, dtype=np.int64)[:, None] * page_size + np.arange(page_size)[None, :])
    pk = k_cache[idx]                     # [n, page, Hkv, D]
    return pk.max(axis=1), pk.min(axis=1)


def score_pages_batched(q_heads: np.ndarray, page_max: np.ndarray, page_min: np.ndarray, num_kv_heads: int) -> np.ndarray:
    """quest.py:1773-1802.  q [B, Hq, D]; page_max/min [B, Hkv, P, D] -> [B, P] (bf16 values)."""
    B, Hq, D = q_heads.shape
    G = Hq // num_kv_heads
    P = page_max.shape[2]
    qg = bf16_round(q_heads.astype(np.float32)).reshape(B, num_kv_heads, G, D)
    q_pos = np.maximum(qg, 0)
    q_neg = np.minimum(qg, 0)
    s = bf16_round(np.einsum("bhgd,bhpd->bhgp", q_pos, page_max.astype(np.float32), dtype=np.float32))
    s = bf16_round(s + bf16_round(np.einsum("bhgd,bhpd->bhgp", q_neg, page_min.astype(np.float32), dtype=np.float32)))
    return s.reshape(B, num_kv_heads * G, P).max(axis=1)


def build_decode_view(q, metadata_max, metadata_min, req_to_token_slots, req_to_page_slots, req_indices, context_lens,
                      *, page_size: int, token_budget: int, max_context_len: int, max_pages_per_row: int,
                      num_kv_heads: int, is_long_text: bool = True):
    """quest.py:1833-1913.  Returns (packed_slots [B, keep], req_indices', lens', info) or None when the
    dense path applies to the whole batch.  `info` carries page_scores / valid mask / prev_budget so a
    test can verify that ANY valid top-k set was chosen (topk(sorted=False) ties are implementation-defined)."""
    page_budget_base = max(3, int(token_budget) // page_size)
    max_keep = max(int(token_budget), page_budget_base * page_size, page_size)
    if max_context_len <= max_keep:
        return None
    B = q.shape[0]
    max_pages = min(max_pages_per_row, (max_context_len + page_size - 1) // page_size)
    prev_budget = min(page_budget_base - 1, max_pages - 1)
    if prev_budget <= 0:
        return None
    context_lens = np.asarray(context_lens, dtype=np.int64)
    num_pages = (context_lens + page_size - 1) // page_size
    rows = np.asarray(req_indices, dtype=np.int64)
    row_page_slots = req_to_page_slots[rows][:, :max_pages]
    prev = np.maximum(row_page_slots[:, : max_pages - 1].astype(np.int64), 0)
    pmax = metadata_max[prev].transpose(0, 2, 1, 3)      # [B, Hkv, P, D]
    pmin = metadata_min[prev].transpose(0, 2, 1, 3)
    scores = score_pages_batched(q, pmax, pmin, num_kv_heads)
    safe_np = np.maximum(num_pages, 1)
    valid = np.arange(max_pages - 1)[None, :] < (safe_np - 1)[:, None]
    scores = np.where(valid, scores, -np.inf).astype(np.float32)
    # a deterministic valid choice: highest score first, lower page index among ties
    order = np.argsort(-scores, axis=1, kind="stable")[:, :prev_budget]
    top_prev = np.sort(order, axis=1)
    last_page = (safe_np - 1)[:, None]
    selected = np.concatenate((top_prev, last_page), axis=1)
    sel_slots = np.take_along_axis(row_page_slots.astype(np.int64), selected, axis=1)
    sparse_slots = (sel_slots[:, :, None] * page_size + np.arange(page_size)[None, None, :]).reshape(B, -1)
    last_page_len = context_lens - (num_pages - 1) * page_size
    sparse_lens = (prev_budget * page_size + last_page_len).astype(np.int32)
    if is_long_text:
        packed, lens = sparse_slots.astype(np.int32), sparse_lens
    else:
        dense = req_to_token_slots[rows][:, :max_keep]
        dense_mask = (context_lens <= int(token_budget)) | (num_pages <= page_budget_base)
        keep = sparse_slots.shape[1]
        packed = np.empty((B, max_keep), dtype=np.int32)
        packed[:, :keep] = np.where(dense_mask[:, None], dense[:, :keep], sparse_slots)
        if max_keep > keep:
            packed[:, keep:] = dense[:, keep:]
        lens = np.where(dense_mask, context_lens, sparse_lens).astype(np.int32)
    info = dict(page_scores=scores, valid=valid, prev_budget=prev_budget, max_pages=max_pages, selected_pages=selected,
                num_pages=num_pages)
    return packed, np.arange(B, dtype=np.int32), lens, info


def check_topk_set(scores_row: np.ndarray, chosen: np.ndarray, k: int, *, atol: float = 0.0) -> None:
    """`chosen` (k distinct indices) is a valid top-k of scores_row: every chosen score >= every
    non-chosen score (ties at the threshold may go either way)."""
    chosen = np.asarray(chosen)
    assert chosen.size == k and np.unique(chosen).size == k
    mask = np.zeros(scores_row.shape[0], dtype=bool)
    mask[chosen] = True
    lo = scores_row[mask].min()
    rest = scores_row[~mask]
    if rest.size:
        assert lo + atol >= rest.max(), f"not a top-{k} set: min chosen {lo} < max rejected {rest.max()}"
