"""Oracle: SnapKV / StreamingLLM selection, the prefill score accumulator and the decode re-eviction.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates, in numpy,
  engine/sparse_controller.py  _snapkv_select_indices{,_batch} :1670-1747 (F.max_pool1d :1694/:1735),
                               _streamingllm_select_indices :1661-1668, _get_streamingllm_budget :1655-1659,
                               _get_layer_budget :2036-2048 (snapkv branch), _snapkv_decode_trigger_len :2050-2054,
                               _snapkv_prefill_eviction :1059-1102, _snapkv_decode_eviction :1104-1223
  engine/cache_manager/snapkv.py  _prefill_score_rows :935-1009 (non-chain prompts), the accumulator
                               :1017-1044 / :1046-1048 and its elementwise-max update :1299-1303.

Pinned by tests/golden/snapkv_select.npz and snapkv_e2e.npz (tests/test_oracle_golden.py), which
tests/golden/gen_fixtures.py produced by running those reference functions on CPU.

`torch.topk` leaves the order of its result and the choice among exact ties unspecified; the functions here return
the keep set in ascending order and take the LOWER index among ties at the threshold.  `check_keep_set` is the
comparison that honours that freedom.
"""

from __future__ import annotations

import numpy as np

from . import h2o as oh


def max_pool1d_same(x: np.ndarray, kernel: int) -> np.ndarray:
    """F.max_pool1d(x[None, None], kernel_size=kernel, padding=kernel // 2, stride=1) (-inf padding).
    Output length = n + 2 * (kernel // 2) - kernel + 1 (= n for odd kernels)."""
    kernel = int(kernel)
    pad = kernel // 2
    n = x.shape[-1]
    xp = np.concatenate((np.full(x.shape[:-1] + (pad,), -np.inf, x.dtype), x, np.full(x.shape[:-1] + (pad,), -np.inf, x.dtype)), -1)
    out_n = n + 2 * pad - kernel + 1
    win = np.stack([xp[..., j: j + out_n] for j in range(kernel)], 0)
    return win.max(0)


def snapkv_layer_budget(sink: int, keep: int, recent: int) -> int:
    """sparse_controller.py:2036-2048, `snapkv` branch."""
    return int(sink) + int(keep) + int(recent)


def snapkv_decode_trigger_len(budget: int, sink: int, recent: int) -> int:
    """sparse_controller.py:2050-2054."""
    return int(2.0 * (int(budget) - int(sink) - int(recent)))


def streamingllm_budget(sink: int, recent: int):
    b = int(sink) + int(recent)
    return b if b > 0 else None


def streamingllm_select_indices(kv_len: int, sink: int, recent: int) -> np.ndarray:
    """sparse_controller.py:1661-1668."""
    assert kv_len > 0
    sink_end = min(int(sink), kv_len)
    recent_start = max(sink_end, kv_len - int(recent))
    return np.concatenate((np.arange(sink_end), np.arange(recent_start, kv_len))).astype(np.int64)


def snapkv_middle_scores(scores: np.ndarray, kv_len: int, *, sink: int, recent: int, pool: int = 1) -> np.ndarray:
    """The (optionally pooled) scores the top-k runs over: positions [sink, kv_len - recent)."""
    mid = np.asarray(scores, np.float32)[..., sink: kv_len - recent]
    return max_pool1d_same(mid, pool) if int(pool) > 1 else mid


def snapkv_select_indices(scores: np.ndarray, kv_len: int, budget: int, *, sink: int, recent: int, pool: int = 1) -> np.ndarray:
    """sink ++ topk(middle) ++ recent as an ASCENDING index list (the reference's order is unspecified and
    free_part_slots sorts it, snapkv.py:1546-1548)."""
    assert kv_len > budget
    recent_start = kv_len - recent
    num_topk = budget - sink - recent
    parts = [np.arange(sink)]
    if num_topk > 0 and recent_start > sink:
        mid = snapkv_middle_scores(scores, kv_len, sink=sink, recent=recent, pool=pool)
        k = min(num_topk, mid.shape[-1])
        order = np.argsort(-mid, kind="stable")[:k]
        parts.append(np.sort(order) + sink)
    parts.append(np.arange(recent_start, kv_len))
    return np.concatenate(parts).astype(np.int64)


def check_keep_set(scores: np.ndarray, kv_len: int, budget: int, got: np.ndarray, ref: np.ndarray, *, sink: int,
                   recent: int, pool: int = 1, atol: float = 0.0) -> bool:
    """`got` and `ref` are two results for the same row.  They must agree on sink, recent and on every middle
    position whose (pooled) score is strictly above the k-th value; positions AT the k-th value may differ
    (topk's tie freedom; `atol` widens "at" for scores that carry float noise).  Returns True when the two
    sets are identical."""
    got, ref = np.sort(np.asarray(got).reshape(-1)), np.sort(np.asarray(ref).reshape(-1))
    assert got.size == ref.size, (got.size, ref.size)
    assert np.unique(got).size == got.size, "duplicate keep indices"
    if np.array_equal(got, ref):
        return True
    mid = snapkv_middle_scores(scores, kv_len, sink=sink, recent=recent, pool=pool)
    k = min(budget - sink - recent, mid.shape[-1])
    thr = np.sort(mid)[::-1][k - 1]
    diff = np.setxor1d(got, ref)
    assert ((diff >= sink) & (diff < kv_len - recent)).all(), "sink / recent part differs"
    vals = mid[diff - sink]
    assert (np.abs(vals - thr) <= atol).all(), f"keep sets differ away from the top-k threshold {thr}: {vals}"
    return False


# ------------------------------------------------------------------------------------------------
# prefill score rows + accumulator (snapkv.py:935-1048, :1299-1303)
# ------------------------------------------------------------------------------------------------

def prefill_score_rows(prompt_lens, prefilled, chunk_lens, *, budget: int, window: int):
    """-> [(batch index, score_start, score_end)]: a prompt is scored in the chunk that contains its whole score
    window (the last `window` prompt tokens) and only when it is longer than the layer budget."""
    rows = []
    if window <= 0:
        return rows
    for b, (p, done, n) in enumerate(zip(prompt_lens, prefilled, chunk_lens)):
        p, done, n = int(p), int(done), int(n)
        if p <= budget:
            continue
        w = min(window, p) if p else window
        end, start = p, max(0, p - w)
        if done <= start and done + n >= end:
            rows.append((b, start, end))
    return rows


def prefill_score_initial_value(mode: str) -> float:
    return -np.inf if mode == "logits" else 0.0


def accumulate_prefill_score(acc, step: np.ndarray, *, mode: str) -> np.ndarray:
    """acc = max(acc, step) elementwise; a fresh accumulator starts at 0 (probability) / -inf (logits)."""
    if acc is None:
        acc = np.full(step.shape, prefill_score_initial_value(mode), np.float32)
    return np.maximum(acc, step.astype(np.float32))


# ------------------------------------------------------------------------------------------------
# evictions on a SlotState
# ------------------------------------------------------------------------------------------------

def snapkv_prefill_eviction(state: oh.SlotState, layers, rows, kv_lens, finals, scores_by_layer_row, *, sink: int,
                            recent: int, keep: int, pool: int = 1):
    """sparse_controller.py:1059-1102: per layer, per final-chunk sequence longer than the budget."""
    budget = snapkv_layer_budget(sink, keep, recent)
    for l in layers:
        for r, n, fin in zip(rows, kv_lens, finals):
            if not fin or n <= budget:
                continue
            sc = scores_by_layer_row[(l, r)]
            idx = snapkv_select_indices(sc[:n], n, budget, sink=sink, recent=recent, pool=pool)
            oh.free_part_slots(state, l, r, idx, keep_sorted=True)


def snapkv_decode_eviction(state: oh.SlotState, layers, rows, scores_by_layer, *, sink: int, recent: int, keep: int):
    """sparse_controller.py:1104-1223.  Per layer: rows with len >= trigger and > budget are grouped by length in
    batch order; a lone row is compacted immediately, a group of several rows is deferred until every layer has
    been visited and then compacted for all its layers at once (per layer that is: lone rows first, groups after,
    each in first-seen order — the order that fixes the free-stack contents)."""
    budget = snapkv_layer_budget(sink, keep, recent)
    trigger = snapkv_decode_trigger_len(budget, sink, recent)
    pending: dict = {}
    for l in layers:
        sc = scores_by_layer.get(l)
        if sc is None:
            continue
        lens = [int(state.row_len[l, r]) for r in rows]
        if max(lens) <= budget or max(lens) < trigger:
            continue
        by_len: dict = {}
        for b, (r, n) in enumerate(zip(rows, lens)):
            if n <= budget or n < trigger:
                continue
            by_len.setdefault(n, []).append((b, r))
        for n, group in by_len.items():
            keeps = np.stack([snapkv_select_indices(sc[b, :n], n, budget, sink=sink, recent=recent) for b, _ in group])
            if len(group) == 1:
                oh.free_part_slots(state, l, group[0][1], keeps[0], keep_sorted=True)
                continue
            key = (tuple(r for _, r in group), keeps.shape)
            pending.setdefault(key, []).append((l, [r for _, r in group], keeps))
    for entries in pending.values():
        for l, grows, keeps in entries:
            oh.free_part_slots_batch_layers(state, [l], grows, keeps[None], keep_sorted=True)
