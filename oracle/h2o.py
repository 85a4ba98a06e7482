"""Oracle: H2O selection, score accumulation and slot-table compaction.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates the torch code of
  engine/cache_manager/h2o.py      (selection :478-563, scores :590-655, :897-1038,
                                    final-prefill dense compaction :1181-1349,
                                    trigger policy :1498-1538)
  engine/cache_manager/snapkv.py   (free_part_slots* :1528-1803, _allocate :1319-1340,
                                    free_seq :1489-1514)
with plain numpy on a tiny explicit state object (`SlotState`) that holds the
same arrays the reference manager owns:
  slot_table [L, rows, cap] i32   == buffer_req_to_token_slots_tensor
  free_stack [L, nslots]    i32   == free_slots_stack_tensor
  free_ptr   [L]            i64   == _num_free_slots
  row_len    [L, rows]      i32   == row_seq_lens
"""

from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np


# --------------------------------------------------------------------------------------
# selection  (h2o.py:478-563)
# --------------------------------------------------------------------------------------

def h2o_budget_partition(budget: int, recent_ratio: float) -> tuple[int, int]:
    """h2o.py:67-71 -> (heavy_count, recent_count)."""
    recent = max(1, int(budget * float(recent_ratio)))
    recent = min(recent, budget)
    return budget - recent, recent


def select_h2o_indices_batch(scores: np.ndarray, *, budget: int, recent_ratio: float) -> np.ndarray:
    """h2o.py:518-563.  scores [rows, kv_len] f32 -> keep [rows, min(kv_len,budget)] i64,
    ascending.  heavy = first `heavy_count` of a *stable descending* argsort of
    scores[:, :recent_start] (ties -> lower index first), recent = the newest
    `recent_count` positions."""
    if scores.ndim != 2:
        raise ValueError(f"Batched H2O scores must have shape [batch, kv_len], got {scores.shape}.")
    rows, kv_len = scores.shape
    budget = int(budget)
    if budget <= 0:
        raise ValueError(f"H2O budget must be positive, got {budget}.")
    if not 0.0 < float(recent_ratio) < 1.0:
        raise ValueError(f"H2O recent_ratio must be in (0, 1), got {recent_ratio}.")
    if kv_len <= budget:
        return np.broadcast_to(np.arange(kv_len, dtype=np.int64), (rows, kv_len)).copy()
    recent_count = max(1, int(budget * float(recent_ratio)))
    recent_count = min(recent_count, budget, kv_len)
    heavy_count = budget - recent_count
    recent_start = kv_len - recent_count
    recent = np.broadcast_to(np.arange(recent_start, kv_len, dtype=np.int64), (rows, recent_count))
    if heavy_count == 0:
        return recent.copy()
    # stable descending argsort == stable ascending argsort of the negated keys
    # (x == y  <=>  -x == -y, so tie groups and their index order are preserved).
    order = np.argsort(-scores[:, :recent_start].astype(np.float32), axis=1, kind="stable")
    heavy = order[:, : min(heavy_count, recent_start)].astype(np.int64)
    keep = np.concatenate((heavy, recent), axis=1)
    if keep.shape[1] != budget:
        raise RuntimeError("Batched H2O selection did not fill the requested budget")
    return np.sort(keep, axis=1)


def select_h2o_indices(scores: np.ndarray, *, budget: int, recent_ratio: float) -> np.ndarray:
    """h2o.py:478-516 (1-D form)."""
    if scores.ndim != 1:
        raise ValueError(f"H2O scores must be 1D, got shape={scores.shape}.")
    return select_h2o_indices_batch(scores[None, :], budget=budget, recent_ratio=recent_ratio)[0]


# --------------------------------------------------------------------------------------
# scores  (h2o.py:590-655, :897-1038)
# --------------------------------------------------------------------------------------

def expand_score(score: np.ndarray | None, new_len: int) -> np.ndarray:
    """h2o.py:590-608."""
    old = 0 if score is None else int(score.shape[0])
    if old > new_len:
        raise RuntimeError(f"H2O score vector cannot shrink without keep_indices: old={old} new={new_len}.")
    out = np.zeros((int(new_len),), dtype=np.float32)
    if old:
        out[:old] = score.astype(np.float32)
    return out


def accumulate_score(previous: np.ndarray | None, step_score: np.ndarray, *, new_len: int, weight: float) -> np.ndarray:
    """h2o.py:610-626: cumulative = expand(previous) + weight * step[:new_len]
    (torch `add_(x, alpha=w)` == fp32 multiply then add, one rounding each)."""
    cum = expand_score(previous, new_len)
    cum += (step_score[:new_len].astype(np.float32) * np.float32(weight)).astype(np.float32)
    return cum


def normalize_logit_prefill_score(step_score: np.ndarray, *, new_len: int) -> np.ndarray:
    """h2o.py:628-655: softmax over the max-reduced prefill logits (-inf -> 0)."""
    logits = step_score[: int(new_len)].astype(np.float32)
    if np.isnan(logits).any() or (logits == np.inf).any() or not np.isfinite(logits).any():
        raise RuntimeError("H2O logit prefill score contains invalid non-finite values")
    m = logits.max()
    e = np.exp(logits - m, dtype=np.float32)
    return (e / e.sum(dtype=np.float32)).astype(np.float32)


def update_decode_scores(previous: np.ndarray, normalized: np.ndarray, kv_len: int) -> np.ndarray:
    """h2o.py:957-1038 fast path: previous [.., kv_len-1], normalized [.., W>=kv_len]
    -> cumulative [.., kv_len] = pad(previous, 1) + normalized[..., :kv_len]."""
    cum = normalized[..., :kv_len].astype(np.float32).copy()
    cum[..., : kv_len - 1] += previous.astype(np.float32)
    return cum


# --------------------------------------------------------------------------------------
# slot-table state and compaction  (snapkv.py)
# --------------------------------------------------------------------------------------

@dataclass
class SlotState:
    slot_table: np.ndarray          # [L, rows, cap] int32
    free_stack: np.ndarray          # [L, nslots] int32
    free_ptr: np.ndarray            # [L] int64
    row_len: np.ndarray             # [L, rows] int32
    scores: dict = field(default_factory=dict)   # (layer,row) -> f32 vector (H2O)

    def copy(self) -> "SlotState":
        return SlotState(self.slot_table.copy(), self.free_stack.copy(), self.free_ptr.copy(),
                         self.row_len.copy(), {k: v.copy() for k, v in self.scores.items()})


def make_slot_state(num_layers: int, rows: int, cap: int, nslots: int, *, permute_seed: int | None = None) -> SlotState:
    """Fresh state: every slot free.  snapkv.py:135-200 initialises the stack to
    arange(nslots); `permute_seed` shuffles it so gathers are genuinely paged
    (SURVEY.md 8(d))."""
    stack = np.tile(np.arange(nslots, dtype=np.int32), (num_layers, 1))
    if permute_seed is not None:
        rng = np.random.default_rng(permute_seed)
        for l in range(num_layers):
            stack[l] = rng.permutation(nslots).astype(np.int32)
    return SlotState(
        slot_table=np.zeros((num_layers, rows, cap), dtype=np.int32),
        free_stack=stack,
        free_ptr=np.full((num_layers,), nslots, dtype=np.int64),
        row_len=np.zeros((num_layers, rows), dtype=np.int32),
    )


def allocate(state: SlotState, layer: int, row: int, size: int) -> np.ndarray:
    """snapkv.py:1319-1340: LIFO pop of stack[ptr-size:ptr], appended to the row."""
    ptr = int(state.free_ptr[layer])
    if ptr < size:
        raise AssertionError(f"Out of KV cache slots: need {size}, free {ptr}")
    cur = int(state.row_len[layer, row])
    if cur + size > state.slot_table.shape[2]:
        raise RuntimeError("KV row length exceeds max_model_len in _allocate")
    sel = state.free_stack[layer, ptr - size: ptr].copy()
    state.free_ptr[layer] -= size
    state.slot_table[layer, row, cur: cur + size] = sel
    state.row_len[layer, row] += size
    return sel


def decode_allocate_batch_layers(state: SlotState, layers, rows) -> np.ndarray:
    """h2o.py:386-415 (prepare_decode_static core): the *same* stack window
    [ptr-B, ptr) on every layer, lane b -> rows[b], written at column cur_len.
    Returns new_slots [len(layers), B] i32."""
    layers = list(layers)
    rows = list(rows)
    B = len(rows)
    ptrs = [int(state.free_ptr[l]) for l in layers]
    if min(ptrs) < B:
        raise RuntimeError(f"Out of KV cache slots in H2O static decode: need={B} free={min(ptrs)}.")
    ptr = ptrs[0]
    out = np.zeros((len(layers), B), dtype=np.int32)
    for i, l in enumerate(layers):
        new = state.free_stack[l, ptr - B: ptr].copy()
        out[i] = new
        state.free_ptr[l] -= B
        for b, r in enumerate(rows):
            cur = int(state.row_len[l, r])
            state.slot_table[l, r, cur] = new[b]
            state.row_len[l, r] = cur + 1
    return out


def decode_allocate_per_layer(state: SlotState, layers, rows_by_layer):
    """SnapKVCacheManager._prepare_decode, non-uniform branch (snapkv.py:2656-2673): every layer pops its OWN
    window [ptr_l - B, ptr_l) and appends lane b's slot at that layer's own row length.
    -> (new_slots [len(layers), B], context_lens [len(layers), B], max_context_len [len(layers)])."""
    layers = list(layers)
    B = len(rows_by_layer[0])
    new_slots = np.zeros((len(layers), B), dtype=np.int32)
    ctx = np.zeros((len(layers), B), dtype=np.int32)
    for i, l in enumerate(layers):
        ptr = int(state.free_ptr[l])
        if ptr < B:
            raise AssertionError(f"Out of KV slots: need {B}, free {ptr}")
        new = state.free_stack[l, ptr - B: ptr].copy()
        state.free_ptr[l] -= B
        for b, r in enumerate(rows_by_layer[i]):
            cur = int(state.row_len[l, r])
            state.slot_table[l, r, cur] = new[b]
            state.row_len[l, r] = cur + 1
            ctx[i, b] = cur + 1
        new_slots[i] = new
    return new_slots, ctx, ctx.max(axis=1)


def pad_static_decode_metadata(new_slots: np.ndarray, context_lens: np.ndarray, rows, graph_batch: int):
    """Padded lanes of a graph-sized decode batch (h2o.py:419-437): slot -1, lane 0's length and row.
    new_slots / context_lens [L, B] -> (slot_mapping, context_lens, req_indices) [L, graph_batch] i32."""
    L, B = new_slots.shape
    rows = np.asarray(rows, dtype=np.int32)
    rows2 = np.broadcast_to(rows, (L, B)) if rows.ndim == 1 else rows
    sm = np.full((L, graph_batch), -1, dtype=np.int32)
    cl = np.repeat(context_lens[:, :1].astype(np.int32), graph_batch, axis=1)
    ri = np.repeat(rows2[:, :1].astype(np.int32), graph_batch, axis=1)
    sm[:, :B], cl[:, :B], ri[:, :B] = new_slots, context_lens, rows2
    return sm, cl, ri


def free_seq(state: SlotState, layers, row: int) -> None:
    """snapkv.py:1489-1514: push the whole row back, zero it."""
    for l in layers:
        cur = int(state.row_len[l, row])
        if cur > 0:
            ptr = int(state.free_ptr[l])
            state.free_stack[l, ptr: ptr + cur] = state.slot_table[l, row, :cur]
            state.free_ptr[l] += cur
        state.slot_table[l, row, :] = 0
        state.row_len[l, row] = 0
        state.scores.pop((l, row), None)


def free_part_slots(state: SlotState, layer: int, row: int, keep: np.ndarray, *, keep_sorted: bool = False) -> None:
    """snapkv.py:1528-1591.  NOTE the scalar form zeroes the *whole* row tail
    (`row[:] = 0` then rewrite) whereas the batched forms zero only
    [new_len, cur_len) - identical result because columns >= cur_len are already 0."""
    cur = int(state.row_len[layer, row])
    keep = np.asarray(keep, dtype=np.int64)
    if keep.size <= 0:
        raise RuntimeError("free_part_slots got empty keep_indices")
    if (keep < 0).any() or (keep >= cur).any():
        raise RuntimeError("free_part_slots keep_indices out of bounds")
    if not keep_sorted:
        keep = np.sort(keep)
    old = state.slot_table[layer, row, :cur].copy()
    new = old[keep]
    mask = np.ones((cur,), dtype=bool)
    mask[keep] = False
    dropped = old[mask]
    if dropped.size:
        ptr = int(state.free_ptr[layer])
        state.free_stack[layer, ptr: ptr + dropped.size] = dropped
        state.free_ptr[layer] += dropped.size
    state.slot_table[layer, row, :] = 0
    state.slot_table[layer, row, : new.size] = new
    state.row_len[layer, row] = new.size


def free_part_slots_batch_layers(state: SlotState, layers, rows, keep: np.ndarray, *, keep_sorted: bool = False) -> None:
    """snapkv.py:1681-1803 (and the single-layer form :1593-1679).

    keep [len(layers), len(rows), K] int64.  Uniform current length required for
    the fused path; otherwise falls back per (layer,row) exactly like the reference.
    Free-stack order per layer = row-major over (batch row, position) of the
    dropped entries (`old_slots[mask].view(num_layers, -1)` :1756).
    """
    layers = list(layers)
    rows = list(rows)
    keep = np.asarray(keep, dtype=np.int64)
    assert keep.shape[:2] == (len(layers), len(rows))
    cur_lens = np.array([[int(state.row_len[l, r]) for r in rows] for l in layers])
    cur = int(cur_lens[0, 0])
    if not np.all(cur_lens == cur):
        for i, l in enumerate(layers):
            for j, r in enumerate(rows):
                free_part_slots(state, l, r, keep[i, j], keep_sorted=keep_sorted)
        return
    if (keep < 0).any() or (keep >= cur).any():
        raise RuntimeError("free_part_slots_batch_layers keep_indices out of bounds")
    if not keep_sorted:
        keep = np.sort(keep, axis=2)
    new_len = keep.shape[2]
    for i, l in enumerate(layers):
        dropped_all = []
        for j, r in enumerate(rows):
            old = state.slot_table[l, r, :cur].copy()
            mask = np.ones((cur,), dtype=bool)
            mask[keep[i, j]] = False
            dropped_all.append(old[mask])
            state.slot_table[l, r, :new_len] = old[keep[i, j]]
            state.slot_table[l, r, new_len:cur] = 0
            state.row_len[l, r] = new_len
        dropped = np.concatenate(dropped_all) if dropped_all else np.zeros((0,), np.int32)
        if dropped.size:
            ptr = int(state.free_ptr[l])
            state.free_stack[l, ptr: ptr + dropped.size] = dropped
            state.free_ptr[l] += dropped.size


def free_prefix_recent_slots(state: SlotState, layers, rows, *, kv_len: int, prefix_tokens: int, recent_tokens: int) -> None:
    """snapkv.py:1805-1896 (StreamingLLM sink+recent compaction): keep
    [0,prefix) U [kv_len-recent, kv_len); dropped middle goes to the free stack in
    (row, position) order; the tail is zeroed."""
    layers = list(layers)
    rows = list(rows)
    kv_len = int(kv_len)
    sink_end = min(int(prefix_tokens), kv_len)
    recent_start = max(sink_end, kv_len - int(recent_tokens))
    new_len = sink_end + (kv_len - recent_start)
    if new_len <= 0:
        raise RuntimeError("prefix/recent compaction cannot keep zero tokens.")
    if new_len >= kv_len:
        return
    for l in layers:
        for r in rows:
            if int(state.row_len[l, r]) != kv_len:
                raise RuntimeError("prefix/recent compaction expected uniform row lengths")
    keep = np.concatenate((np.arange(sink_end), np.arange(recent_start, kv_len))).astype(np.int64)
    k3 = np.broadcast_to(keep, (len(layers), len(rows), keep.size))
    free_part_slots_batch_layers(state, layers, rows, k3, keep_sorted=True)


def compact_final_prefill_dense_batch(state: SlotState, layer: int, rows, keep: np.ndarray, budget: int,
                                      k_cache: np.ndarray, v_cache: np.ndarray) -> None:
    """h2o.py:1181-1349: selected K/V rows move into the `budget` *smallest*
    physical slots the row already owns (ascending), the larger slots are released
    (per row ascending, rows concatenated), table row = destination slots."""
    rows = list(rows)
    keep = np.asarray(keep, dtype=np.int64)
    assert keep.shape == (len(rows), budget)
    lens = [int(state.row_len[layer, r]) for r in rows]
    kv_len = lens[0]
    if any(x != kv_len for x in lens):
        raise RuntimeError("H2O final-prefill dense batch requires uniform physical lengths")
    if kv_len <= budget:
        raise RuntimeError("H2O final-prefill dense compaction requires an over-budget row")
    if (keep < 0).any() or (keep >= kv_len).any():
        raise RuntimeError("H2O final-prefill keep indices are out of bounds")
    if budget > 1 and not (keep[:, 1:] > keep[:, :-1]).all():
        raise RuntimeError("H2O final-prefill keep indices must be strictly increasing")
    old = np.stack([state.slot_table[layer, r, :kv_len] for r in rows]).astype(np.int64)
    flat_sorted = np.sort(old.reshape(-1))
    if flat_sorted.size > 1 and (flat_sorted[1:] == flat_sorted[:-1]).any():
        raise RuntimeError("H2O final-prefill active physical slots must be unique across the batch")
    selected = np.take_along_axis(old, keep, axis=1)
    sorted_old = np.sort(old, axis=1)
    dest = sorted_old[:, :budget]
    released = sorted_old[:, budget:].reshape(-1)
    ws_k = k_cache[selected.reshape(-1)].copy()
    ws_v = v_cache[selected.reshape(-1)].copy()
    k_cache[dest.reshape(-1)] = ws_k
    v_cache[dest.reshape(-1)] = ws_v
    ptr = int(state.free_ptr[layer])
    state.free_stack[layer, ptr: ptr + released.size] = released.astype(np.int32)
    state.free_ptr[layer] = ptr + released.size
    for j, r in enumerate(rows):
        state.slot_table[layer, r, :budget] = dest[j].astype(np.int32)
        state.slot_table[layer, r, budget:kv_len] = 0
        state.row_len[layer, r] = budget


# --------------------------------------------------------------------------------------
# trigger policy (h2o.py:1498-1538) and the decode burst (h2o.py:1558-1625)
# --------------------------------------------------------------------------------------

def decode_eviction_groups(row_lens: dict, scheduled_rows, active_rows, *, budget: int, interval: int,
                           num_free_slots: int) -> dict:
    """h2o.py:1498-1538.  row_lens: row -> kv_len (aligned across layers).
    Returns {kv_len: [rows...]} in the reference's insertion order: scheduled rows in
    batch order, then (only under slot pressure) the other active decode rows sorted
    by id.  trigger = budget+interval, or budget+1 when num_free_slots <= 0."""
    under_pressure = num_free_slots <= 0
    trigger = budget + 1 if under_pressure else budget + interval
    cand = list(scheduled_rows)
    if under_pressure:
        cand.extend(sorted(set(active_rows).difference(scheduled_rows)))
    groups: dict[int, list[int]] = {}
    for r in cand:
        kv_len = int(row_lens[r])
        if kv_len >= trigger:
            groups.setdefault(kv_len, []).append(r)
    return groups


def evict_decode_rows(state: SlotState, layers, groups: dict, *, budget: int, recent_ratio: float) -> dict:
    """h2o.py:1558-1625 fast path.  Mutates `state` (slot table, free stack,
    row lens, per-(layer,row) scores).  Returns {kv_len: keep[L, n, budget]}."""
    layers = list(layers)
    out = {}
    for kv_len, rows in groups.items():
        scores = np.stack([state.scores[(l, r)] for l in layers for r in rows]).reshape(len(layers), len(rows), kv_len)
        keep = select_h2o_indices_batch(scores.reshape(-1, kv_len), budget=budget,
                                        recent_ratio=recent_ratio).reshape(len(layers), len(rows), budget)
        kept = np.take_along_axis(scores, keep, axis=2)
        free_part_slots_batch_layers(state, layers, rows, keep, keep_sorted=True)
        for i, l in enumerate(layers):
            for j, r in enumerate(rows):
                state.scores[(l, r)] = kept[i, j].copy()
        out[kv_len] = keep
    return out
