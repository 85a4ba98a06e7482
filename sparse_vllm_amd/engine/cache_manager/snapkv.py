"""SnapKV-family physical cache manager: per-layer slot tables + LIFO free-slot stacks.

Host mirror of `SnapKVCacheManager` (engine/cache_manager/snapkv.py:98-3055) for the
parts on the hot path.  Same state, same names:

  kv_cache                          [2, L, slots, Hkv, D] bf16     (snapkv.py:135-200)
  buffer_req_to_token_slots_tensor  [L, rows, max_model_len] i32   row-major slot table
  free_slots_stack_tensor           [L, slots] i32                 LIFO stack, top = _num_free_slots[l]
  row_seq_lens / seq_id_to_row / free_rows                          host bookkeeping

Slot ids move only through libsvk kernels (decode allocation, compaction); the host
keeps the same integers the reference keeps in Python (`_num_free_slots`, `row_seq_lens`).
In decode, "eviction" rewrites the slot table and the free stack only - K/V payload rows
never move (snapkv.py:1528-1803).
"""

from __future__ import annotations

from collections import deque

import numpy as np
import torch

from ...kernels import h2o_ops
from ...utils.context import get_context
from ...utils.profiler import profiler
from .base import CacheManager, ExplicitKVPayload, LayerBatchStates, PrefillComputeView


class SnapKVCacheManager(CacheManager):
    def __init__(self, config, parallel_context=None):
        super().__init__(config, parallel_context)
        self._uniform_decode_metadata = False
        self._prefill_attn_score_accumulators: dict[tuple[int, int], torch.Tensor] = {}
        self._prefill_score_workspace = None
        self._prefill_context_lens_cpu_by_layer: dict[int, tuple[int, ...]] = {}
        self.allocate_kv_cache()
        # device-resident decode bookkeeping (see `_device_step_params`); SVK_H2O_DEVICE_STATE=0: host-driven steps
        import os
        self._device_step_enabled = os.environ.get("SVK_H2O_DEVICE_STATE", "1") == "1"
        self._dev_row_len = torch.zeros((self.num_kv_layers, self.max_buffer_rows), dtype=torch.int32, device=self.device)
        self._dev_free_ptr = torch.zeros((self.num_kv_layers,), dtype=torch.long, device=self.device)
        self._dev_burst_tickets = torch.zeros((self.num_kv_layers,), dtype=torch.int32, device=self.device)
        self._dev_state_dirty = True
        self._dev_step_cache = None              # (key, SvkH2oDeviceStepArgs, keep-alive tensors)
        # bumped whenever the args struct (whose device pointers a captured hipGraph bakes in by value) is rebuilt: a
        # caller that replays a graph keys it on this, never on object identity (a freed struct's id can be reused)
        self.device_step_generation = 0
        self._device_step = None                 # this step's [args, rows_2d, kv_idx, burst_launched] while active

    # ------------------------------------------------------------------ allocation
    def _resolve_num_slots(self) -> int:
        n = int(getattr(self.config, "num_kvcache_slots", 0) or 0)
        if n > 0:
            return n
        if not torch.cuda.is_available():
            return self.max_buffer_rows * min(self.max_model_len, 8192)
        free, _total = torch.cuda.mem_get_info(self.device)
        per_slot = 2 * self.num_kv_layers * self.num_kv_heads * self.head_dim * 2
        return max(1, int(free * 0.8) // per_slot)

    def allocate_kv_cache(self):
        L, rows = self.num_kv_layers, self.max_buffer_rows
        self.num_slots = self._resolve_num_slots()
        self.config.num_kvcache_slots = self.num_slots
        d = self.device
        self.kv_cache = torch.zeros((2, L, self.num_slots, self.num_kv_heads, self.head_dim),
                                    dtype=torch.bfloat16, device=d)
        self.buffer_req_to_token_slots_tensor = torch.zeros((L, rows, self.max_model_len), dtype=torch.int32, device=d)
        self.buffer_req_to_token_slots = [self.buffer_req_to_token_slots_tensor[i] for i in range(L)]
        self.free_slots_stack_tensor = torch.arange(self.num_slots, dtype=torch.int32, device=d).repeat(L, 1)
        self.free_slots_stack = [self.free_slots_stack_tensor[i] for i in range(L)]
        self._num_free_slots = [self.num_slots for _ in range(L)]
        self._row_seq_lens_all = np.zeros((L, rows), dtype=np.int32)        # one array: per-layer checks are one numpy op
        self.row_seq_lens = [self._row_seq_lens_all[i] for i in range(L)]   # views, the reference's list-of-arrays face
        self.seq_id_to_row = [dict() for _ in range(L)]
        self.free_rows = [deque(range(rows)) for _ in range(L)]
        self._layer_ids_all = torch.arange(L, dtype=torch.int32, device=d)
        self._decode_static_buffers = None

    def permute_free_slots(self, seed: int):
        """Shuffle the initial free stack (benchmarks/tests: makes the KV gather genuinely
        paged).  Only valid while every slot is free."""
        assert all(n == self.num_slots for n in self._num_free_slots)
        g = torch.Generator(device="cpu").manual_seed(int(seed))
        for l in range(self.num_kv_layers):
            self.free_slots_stack_tensor[l].copy_(torch.randperm(self.num_slots, generator=g).to(torch.int32))

    # ------------------------------------------------------------------ accessors
    def get_layer_batch_states(self, layer_idx: int) -> LayerBatchStates:
        return self.layer_batch_states[layer_idx]

    def get_layer_kv_cache(self, layer_idx: int):
        i = self.kv_layer_index(layer_idx)
        return self.kv_cache[0, i], self.kv_cache[1, i]

    def get_layer_buffer_req_to_token_slots(self, layer_idx: int) -> torch.Tensor:
        return self.buffer_req_to_token_slots[self.kv_layer_index(layer_idx)]

    @property
    def num_free_slots(self) -> int:
        return int(min(self._num_free_slots))

    def free_slot_stats(self) -> dict:
        return {"min": int(min(self._num_free_slots)), "max": int(max(self._num_free_slots)), "total": self.num_slots}

    def _get_free_row(self, layer_idx: int, seq_id: int) -> int:
        row = self.seq_id_to_row[layer_idx].get(seq_id)
        if row is None:
            if not self.free_rows[layer_idx]:
                raise RuntimeError(f"No free KV rows: layer={layer_idx} rows={self.max_buffer_rows}")
            row = self.free_rows[layer_idx].popleft()
            self.seq_id_to_row[layer_idx][seq_id] = row
        return int(row)

    def _row_of(self, layer_idx: int, seq) -> int:
        row = self.seq_id_to_row[layer_idx].get(seq.seq_id)
        if row is None:
            raise ValueError(f"unknown seq_id={seq.seq_id} on layer={layer_idx}")
        return int(row)

    # ------------------------------------------------------------------ prefill-side allocation
    def _allocate(self, layer_idx: int, seq_id: int, size: int) -> torch.Tensor:
        """snapkv.py:1319-1340: LIFO pop of stack[ptr-size:ptr] appended to the row."""
        self._dev_state_dirty = True              # host-driven change of rows / pointers: the device copy is stale
        assert self._num_free_slots[layer_idx] >= size, (
            f"Out of KV cache slots: need {size}, free {self._num_free_slots[layer_idx]}")
        row = self._get_free_row(layer_idx, seq_id)
        cur = int(self.row_seq_lens[layer_idx][row])
        if cur + int(size) > self.max_model_len:
            raise RuntimeError("KV row length exceeds max_model_len in _allocate: "
                               f"layer={layer_idx} seq_id={seq_id} row={row} cur_len={cur} size={int(size)} "
                               f"max_model_len={self.max_model_len}")
        ptr = self._num_free_slots[layer_idx]
        select_index = self.free_slots_stack[layer_idx][ptr - size: ptr]
        self._num_free_slots[layer_idx] -= size
        self.buffer_req_to_token_slots[layer_idx][row, cur: cur + size] = select_index
        self.row_seq_lens[layer_idx][row] += size
        return select_index

    def _prepare_prefill(self, seqs):
        """One chunk per sequence appended on every layer (h2o.py:665-748 / snapkv)."""
        d = self.device
        chunk_lens = [int(s.current_chunk_size) for s in seqs]
        total = sum(chunk_lens)
        for layer_idx in self.kv_transformer_layer_indices():
            parts, ctx, rows = [], [], []
            for s, n in zip(seqs, chunk_lens):
                parts.append(self._allocate(layer_idx, s.seq_id, n).clone())
                row = self.seq_id_to_row[layer_idx][s.seq_id]
                rows.append(row)
                ctx.append(int(self.row_seq_lens[layer_idx][row]))
            st = self.layer_batch_states[layer_idx]
            st.slot_mapping = torch.cat(parts) if parts else torch.empty(0, dtype=torch.int32, device=d)
            st.context_lens = torch.tensor(ctx, dtype=torch.int32, device=d)
            st.req_indices = torch.tensor(rows, dtype=torch.int32, device=d)
            st.max_context_len = max(ctx) if ctx else 0
            self._prefill_context_lens_cpu_by_layer[int(layer_idx)] = tuple(ctx)
        cu = np.concatenate(([0], np.cumsum(chunk_lens))).astype(np.int32)
        return torch.from_numpy(cu).to(d), total

    # ------------------------------------------------------------------ prefill token scores (SnapKV)
    def _prefill_score_layer_budget(self, layer_idx: int) -> int | None:
        """snapkv.py:920-933."""
        if self.kv_layer_index(layer_idx) < int(getattr(self.config, "snapkv_num_full_layers", 0) or 0):
            return None
        if self.config.vllm_sparse_method == "snapkv":
            return int(self.config.num_sink_tokens) + int(self.config.decode_keep_tokens) + int(self.config.num_recent_tokens)
        return None

    def _prefill_score_rows(self, layer_idx: int, seqs) -> list[tuple[int, object, int, int]]:
        """snapkv.py:935-1009 (without chain-resume): the window = last `snapkv_window_size` prompt
        tokens must sit inside the current chunk; rows whose prompt fits the budget are skipped."""
        budget = self._prefill_score_layer_budget(layer_idx)
        if budget is None:
            return []
        window = int(getattr(self.config, "snapkv_window_size", 0) or 0)
        if window <= 0:
            return []
        rows = []
        for b_idx, seq in enumerate(seqs):
            prompt_len = int(seq.num_prompt_tokens)
            if prompt_len <= budget:
                continue
            score_end = prompt_len
            score_start = max(0, score_end - min(window, prompt_len))
            chunk_start = int(seq.num_prefilled_tokens)
            chunk_end = chunk_start + int(seq.current_chunk_size)
            if chunk_start <= score_start and chunk_end >= score_end:
                rows.append((b_idx, seq, score_start, score_end))
            elif seq.is_last_chunk_prefill and chunk_start < score_end and chunk_end > score_start:
                raise RuntimeError("SnapKV/PyramidKV prefill score requires the score query window to fit in the final "
                                   f"prefill chunk. layer={layer_idx} seq_id={seq.seq_id} score_range=[{score_start}, {score_end}) "
                                   f"chunk_range=[{chunk_start}, {chunk_end}).")
        return rows

    # ---- scheduler hooks of the SnapKV manager (snapkv.py:806-905, the branches of methods in this build)
    def prefill_batched_tokens_margin(self) -> int:
        return 0

    def remaining_prefill_tokens(self, seq) -> int:
        return int(seq.num_prompt_tokens - seq.num_prefilled_tokens)

    def min_final_prefill_chunk_size(self, seq) -> int:
        """snapkv.py:806-879: a prompt longer than the layer budget is scored with the last `snapkv_window_size` queries, so
        its final chunk must hold that window."""
        if self.config.vllm_sparse_method != "snapkv":
            return 0
        window = int(getattr(self.config, "snapkv_window_size", 0) or 0)
        if window <= 0 or int(getattr(self.config, "snapkv_num_full_layers", 0) or 0) >= int(self.num_kv_layers):
            return 0
        budget = int(self.config.num_sink_tokens) + int(self.config.decode_keep_tokens) + int(self.config.num_recent_tokens)
        if int(seq.num_prompt_tokens) <= budget:
            return 0
        return min(window, int(self.remaining_prefill_tokens(seq)))

    def _prefill_score_initial_value(self) -> float:
        return float("-inf") if getattr(self.config, "sparse_prefill_score_mode", "probability") == "logits" else 0.0

    def _run_prefill_score(self, q, k_cache, step_score, meta, b_start_loc, b_prompt_cache_len, max_score_len,
                           score_starts, score_ends, *, candidate_start: int, num_recent_tokens: int, batch_indices=None,
                           row_stats=None):
        """snapkv.py:1050-1085 -> svk_prefill_score over the view's slot table (`meta` = the AttentionViewMeta of the
        layer's PrefillComputeView)."""
        from ...kernels.prefill_score import PrefillScoreWorkspace, prefill_score_fwd
        if self._prefill_score_workspace is None:
            self._prefill_score_workspace = PrefillScoreWorkspace()
        with profiler.record("prefill_token_score"):
            prefill_score_fwd(q, k_cache, step_score, meta.req_indices, b_start_loc, meta.context_lens, b_prompt_cache_len,
                              int(max_score_len), meta.active_slots, score_starts, score_ends,
                              candidate_start=candidate_start, num_recent_tokens=num_recent_tokens,
                              score_mode=self.config.sparse_prefill_score_mode, workspace=self._prefill_score_workspace,
                              batch_indices=batch_indices, row_stats=row_stats)

    def _get_prefill_attention_score_accumulator(self, layer_idx: int, seq, *, prompt_len: int, device):
        """snapkv.py:1017-1044: the per-(layer, sequence) element-wise-max accumulator; a prompt that starts over
        (num_prefilled_tokens == 0) starts from the mode's neutral value."""
        key = (int(layer_idx), int(seq.seq_id))
        if int(seq.num_prefilled_tokens) == 0:
            self._prefill_attn_score_accumulators.pop(key, None)
        acc = self._prefill_attn_score_accumulators.get(key)
        if acc is None:
            acc = torch.full((int(prompt_len),), self._prefill_score_initial_value(), dtype=torch.float32, device=device)
            self._prefill_attn_score_accumulators[key] = acc
        return acc

    @torch.no_grad()
    def collect_prefill_attention_score(self, layer_idx: int, q: torch.Tensor, view: PrefillComputeView, *,
                                        b_start_loc: torch.Tensor, chunk_lens: torch.Tensor):
        """snapkv.py:1216-1304: score the row with the last-window queries of the final chunk
        (candidates [sink, len - recent)), max-accumulate per (layer, sequence)."""
        ctx = get_context()
        if not ctx.is_prefill:
            return None
        if self.config.vllm_sparse_method != "snapkv":
            return None
        seqs = getattr(ctx, "seqs", None)
        if seqs is None:
            raise RuntimeError("Prefill score collection requires current seqs in context.")
        rows = self._prefill_score_rows(layer_idx, seqs)
        if not rows:
            return None
        if not isinstance(view.payload, ExplicitKVPayload):
            raise TypeError(f"SnapKV prefill scoring requires ExplicitKVPayload, got {type(view.payload).__name__}.")
        meta, payload = view.meta, view.payload
        if int(chunk_lens.ndim) != 1 or int(chunk_lens.shape[0]) != len(seqs):
            raise RuntimeError("SnapKV prefill scoring chunk-length batch mismatch: "
                               f"shape={tuple(chunk_lens.shape)} seqs={len(seqs)}.")
        d = q.device
        ctx_lens = [int(self.row_seq_lens[layer_idx][self.seq_id_to_row[layer_idx][s.seq_id]]) for s in seqs]
        cache_lens = torch.tensor([c - int(s.current_chunk_size) for c, s in zip(ctx_lens, seqs)], dtype=torch.int32, device=d)
        bi = torch.tensor([r[0] for r in rows], dtype=torch.int32, device=d)
        starts = torch.tensor([r[2] for r in rows], dtype=torch.int32, device=d)
        ends = torch.tensor([r[3] for r in rows], dtype=torch.int32, device=d)
        max_ctx = max(ctx_lens[r[0]] for r in rows)
        step = torch.empty((len(rows), max_ctx), dtype=torch.float32, device=d)
        self._run_prefill_score(q, payload.k_cache, step, meta, b_start_loc, cache_lens, max(r[3] - r[2] for r in rows),
                                starts, ends, candidate_start=int(self.config.num_sink_tokens),
                                num_recent_tokens=int(self.config.num_recent_tokens), batch_indices=bi)
        for i, (b_idx, seq, _s, _e) in enumerate(rows):
            n = ctx_lens[b_idx]
            acc = self._get_prefill_attention_score_accumulator(layer_idx, seq, prompt_len=n, device=d)
            torch.maximum(acc[:n], step[i, :n], out=acc[:n])
        return None

    def pop_prefill_attention_score(self, layer_idx: int, seq):
        return self._prefill_attn_score_accumulators.pop((int(layer_idx), int(seq.seq_id)), None)

    def decode_kv_lens_all_layers(self, seqs):
        """[KV layers, len(seqs)] physical row lengths as one numpy gather, for the decode sequences whose row matrix
        `prepare_decode_static` cached this step (None otherwise): per-step policy checks need no Python loop over
        layers x sequences."""
        cached = getattr(self, "_decode_static_rows", None)
        if cached is None or cached[0][0] != tuple(s.seq_id for s in seqs):
            return None
        _, rows_2d, _, kv_idx = cached[:4]
        return self._row_seq_lens_all[kv_idx[:, None], rows_2d]

    def decode_kv_lens_for_layer(self, layer_idx: int, seqs) -> list[int]:
        """snapkv.py:1516-1527."""
        return [int(self.row_seq_lens[layer_idx][self._row_of(layer_idx, s)]) for s in seqs]

    # ------------------------------------------------------------------ decode-side allocation
    def _get_decode_static_buffers(self, graph_batch_size: int):
        """snapkv.py:2698-2750: persistent [L, B] metadata (graph-stable addresses)."""
        buf = self._decode_static_buffers
        if buf is None or buf[0].shape[1] < graph_batch_size:
            d = self.device
            L = self.num_layers
            buf = tuple(torch.zeros((L, graph_batch_size), dtype=torch.int32, device=d) for _ in range(3))
            self._decode_static_buffers = buf
            # the cached device-step arguments hold views of the old buffers: rebuild them (and, through the generation
            # counter, the caller's hipGraph) before the next device-resident step
            self._dev_step_cache = None
            self._dev_state_dirty = True
        return tuple(t[:, :graph_batch_size] for t in buf)

    def _prepare_decode_static_host(self, seqs, input_ids=None, positions=None, slot_mapping=None, context_lens=None,
                                    req_indices=None, *, graph_batch_size: int | None = None):
        """Device-side decode step preparation for all layers in ONE launch
        (h2o.py:256-476 / snapkv.py:2961): every layer pops the window
        [ptr-B, ptr) of its free stack, lane b's slot is appended to row b."""
        real_batch_size = len(seqs)
        if real_batch_size <= 0:
            raise ValueError("Static decode requires a non-empty real decode batch.")
        graph_batch_size = int(graph_batch_size or (input_ids.numel() if input_ids is not None else real_batch_size))
        if real_batch_size > graph_batch_size:
            raise ValueError("Static decode graph batch is smaller than the real decode batch: "
                             f"graph={graph_batch_size}, real={real_batch_size}.")
        layer_ids = [int(l) for l in self.kv_transformer_layer_indices()]
        first = layer_ids[0]
        d = self.device
        L = len(layer_ids)
        # request rows of every layer: cached per batch composition (the dict lookups are the host cost of this
        # function); the first layer's rows are looked up every step and are part of the key
        rows0 = tuple(self._row_of(first, s) for s in seqs)
        key = (tuple(s.seq_id for s in seqs), rows0, tuple(layer_ids))
        cached = getattr(self, "_decode_static_rows", None)
        if cached is None or cached[0] != key:
            rows_2d = np.array([rows0] + [[self._row_of(l, s) for s in seqs] for l in layer_ids[1:]], dtype=np.int64)
            rows_uniform = bool((rows_2d == rows_2d[0]).all())
            kv_idx = np.array([self.kv_layer_index(l) for l in layer_ids], dtype=np.int64)
            cached = (key, rows_2d, rows_uniform, kv_idx,
                      torch.from_numpy(rows_2d[0].astype(np.int32)).to(d),
                      None if rows_uniform else torch.from_numpy(rows_2d.astype(np.int32)).to(d))
            self._decode_static_rows = cached
        _, rows_2d, rows_uniform, kv_idx, rows_gpu, rows_2d_gpu = cached
        cur_2d = self._row_seq_lens_all[kv_idx[:, None], rows_2d]                    # [L, B] lengths before the append
        free_ptrs = [int(self._num_free_slots[l]) for l in layer_ids]
        uniform = rows_uniform and bool((cur_2d == cur_2d[0]).all()) and all(p == free_ptrs[0] for p in free_ptrs[1:])
        max_cur = int(cur_2d.max())
        if max_cur + 1 > self.max_model_len:
            raise RuntimeError(f"KV row length exceeds max_model_len in static decode: max_cur_len={max_cur} "
                               f"max_model_len={self.max_model_len}.")
        static_cap = self._decode_static_max_context_len
        if static_cap is not None and max_cur + 1 > int(static_cap):
            raise RuntimeError("static decode context exceeds the captured graph capacity: "
                               f"next_len={max_cur + 1} static_cap={int(static_cap)}.")
        if min(free_ptrs) < real_batch_size:
            raise RuntimeError(f"Out of KV cache slots in static decode: need={real_batch_size} free={min(free_ptrs)}.")
        if not uniform and not self._supports_nonuniform_decode_layers():
            raise RuntimeError("static decode requires uniform request rows, row lengths and free-stack pointers across "
                               f"KV layers: ptrs={free_ptrs} lens_first_lane={cur_2d[:, 0].tolist()}.")
        sm, cl, ri = self._get_decode_static_buffers(graph_batch_size)
        if uniform:
            cur_gpu = torch.from_numpy(cur_2d[0].astype(np.int32)).to(d, non_blocking=True)
            h2o_ops.decode_alloc_slots(self.buffer_req_to_token_slots_tensor, self.free_slots_stack_tensor,
                                       self._layer_ids_all, rows_gpu, cur_gpu, sm, cl, ri,
                                       free_ptr=free_ptrs[0], batch=real_batch_size)
        else:
            # per-layer rows / lengths / stack pointers: the non-uniform branch of the reference's _prepare_decode
            # (snapkv.py:2656-2673; e.g. snapkv_num_full_layers > 0 leaves the full layers uncompressed)
            if rows_2d_gpu is None:
                rows_2d_gpu = torch.from_numpy(rows_2d.astype(np.int32)).to(d)
            cur_gpu = torch.from_numpy(np.ascontiguousarray(cur_2d, dtype=np.int32)).to(d, non_blocking=True)
            ptr_gpu = torch.tensor(free_ptrs, dtype=torch.long, device=d)
            h2o_ops.decode_alloc_slots(self.buffer_req_to_token_slots_tensor, self.free_slots_stack_tensor,
                                       torch.from_numpy(kv_idx.astype(np.int32)).to(d), rows_2d_gpu, cur_gpu, sm, cl, ri,
                                       free_ptr=min(free_ptrs), batch=real_batch_size, free_ptrs=ptr_gpu)
        for i, l in enumerate(layer_ids):
            self._num_free_slots[l] -= real_batch_size
        self._row_seq_lens_all[kv_idx[:, None], rows_2d] = cur_2d + 1
        for i, l in enumerate(layer_ids):
            st = self.layer_batch_states[l]
            st.slot_mapping, st.context_lens, st.req_indices = sm[i], cl[i], ri[i]
            st.max_context_len = int(static_cap) if static_cap is not None else int(cur_2d[i].max()) + 1
        if slot_mapping is not None:
            slot_mapping.copy_(sm[0])
            context_lens.copy_(cl[0])
            req_indices.copy_(ri[0])
        return input_ids, positions, None

    # ------------------------------------------------------------------ device-resident decode step
    # include/svk.h SvkH2oDeviceStepArgs.  Row lengths and free-stack pointers also live on the device, a decode step
    # allocates from them and runs the method's periodic eviction behind a device-side test, so the step - eviction
    # included - is one hipGraph and uploads nothing.  The host keeps its numpy mirrors in lock-step by the same
    # (deterministic) arithmetic; any host-driven change of rows or pointers marks the device copy stale.  A subclass
    # opts in by returning its parameters from `_device_step_params` (H2O: heavy hitters every `interval` tokens;
    # StreamingLLM: sink + recent window at 2 x (sink + recent)).
    def _device_step_params(self, graph_batch_size: int):
        """-> (budget, trigger_len, recent_count, select_mode, score tensor or None, prefix_count), or None: host-driven
        steps.  Here: SnapKV's decode re-eviction (sparse_controller.py:1104-1223) - a row that reached 2 x decode_keep
        tokens keeps sink ++ top-k of the middle by this step's head-max scores ++ recent - when every layer evicts
        (snapkv_num_full_layers == 0) and no pooling is configured; the scores are the step's scratch rows
        `snapkv_decode_score_tensor` [L, lanes, width], which the controller hands to the attention launches."""
        cfg = self.config
        if cfg.vllm_sparse_method != "snapkv" or int(getattr(cfg, "snapkv_num_full_layers", 0) or 0) > 0:
            return None
        if int(getattr(cfg, "pool_kernel_size", 1) or 1) > 1:
            return None
        sink, keep, recent = int(cfg.num_sink_tokens), int(cfg.decode_keep_tokens), int(cfg.num_recent_tokens)
        budget, trigger = sink + keep + recent, int(2.0 * keep)
        if keep <= 0 or trigger <= budget or trigger > self.max_model_len:
            return None
        if recent < 1:
            # svk_h2o_device_burst needs a recent range of at least one token; the reference's selection handles an empty
            # one (sparse_controller.py:1670-1747), so such a configuration takes the host-driven steps
            return None
        width = max(trigger, int(self._decode_static_max_context_len or 0))
        buf = self.__dict__.get("snapkv_decode_score_tensor")
        if buf is None or buf.shape[1] < int(graph_batch_size) or buf.shape[2] < width:
            buf = self.snapkv_decode_score_tensor = torch.empty((self.num_kv_layers, int(graph_batch_size), width),
                                                                dtype=torch.float32, device=self.device)
        return budget, trigger, recent, 2, buf, sink

    def _on_device_burst(self, seqs, n_rows: int, n_layers: int, dropped_per_row: int) -> None:
        """Counters of a subclass when `n_rows` sequences were evicted on every layer by the in-graph burst."""
        return None

    def _device_step_plan(self, seqs, graph_batch_size: int):
        """-> (args, rows_2d, kv_idx) when this decode step can run from the device-resident state: uniform rows / lengths /
        pointers across the KV layers (H2O's invariant), room in the rows and - after the allocation - still free slots
        (at zero the reference switches to the slot-pressure trigger, h2o.py:1506-1524: that step takes the host path)."""
        params = self._device_step_params(int(graph_batch_size)) if self._device_step_enabled and seqs else None
        if params is None:
            return None
        budget, trigger, recent, select_mode, score_tensor, prefix = params
        layer_ids = [int(l) for l in self.kv_transformer_layer_indices()]
        first = layer_ids[0]
        B = len(seqs)
        rows0 = tuple(self._row_of(first, s) for s in seqs)
        key = (tuple(s.seq_id for s in seqs), rows0, tuple(layer_ids), int(graph_batch_size), budget, trigger, recent, select_mode,
               prefix, 0 if score_tensor is None else score_tensor.data_ptr())
        cache = self._dev_step_cache
        sm, cl, ri = self._get_decode_static_buffers(int(graph_batch_size))
        if cache is not None and cache[2][2].data_ptr() != sm.data_ptr():
            cache = None          # the static step buffers were reallocated (a larger graph batch on the host path)
        if cache is None or cache[0] != key:
            rows_2d = np.array([rows0] + [[self._row_of(l, s) for s in seqs] for l in layer_ids[1:]], dtype=np.int64)
            if not bool((rows_2d == rows_2d[0]).all()) or len(layer_ids) != self.num_kv_layers:
                return None
            kv_idx = np.array([self.kv_layer_index(l) for l in layer_ids], dtype=np.int64)
            d = self.device
            rows_gpu = torch.from_numpy(rows_2d[0].astype(np.int32)).to(d)
            keep = torch.empty((len(layer_ids), B, budget), dtype=torch.long, device=d)
            args = h2o_ops.h2o_device_step_args(
                self.buffer_req_to_token_slots_tensor, self.free_slots_stack_tensor, score_tensor, self._dev_row_len,
                self._dev_free_ptr, rows_gpu, sm, cl, ri, keep, batch=B, budget=budget, recent_count=recent, trigger_len=trigger,
                select_mode=select_mode, prefix_count=prefix, tickets=self._dev_burst_tickets)
            cache = self._dev_step_cache = (key, args, (rows_gpu, keep, sm, cl, ri), rows_2d, kv_idx)
            self.device_step_generation += 1
        _, args, _keepalive, rows_2d, kv_idx = cache
        cur = self._row_seq_lens_all[kv_idx[:, None], rows_2d]
        ptrs = [int(self._num_free_slots[l]) for l in layer_ids]
        if not bool((cur == cur[0]).all()) or any(p != ptrs[0] for p in ptrs[1:]):
            return None
        max_cur = int(cur.max())
        static_cap = self._decode_static_max_context_len
        if (max_cur + 1 > self.max_model_len or max_cur >= trigger or ptrs[0] - B <= 0
                or (static_cap is not None and max_cur + 1 > int(static_cap))):
            return None
        return args, rows_2d, kv_idx

    def _device_state_upload(self):
        self._dev_row_len.copy_(torch.from_numpy(self._row_seq_lens_all), non_blocking=False)
        self._dev_free_ptr.copy_(torch.tensor([int(self._num_free_slots[int(l)]) for l in self.kv_transformer_layer_indices()],
                                              dtype=torch.long))
        # a burst that aborted mid-way (a failed step) would leave a layer's ticket non-zero and that layer would never
        # commit again: every re-upload starts from zeroed tickets
        self._dev_burst_tickets.zero_()
        self._dev_state_dirty = False

    def prepare_decode_static(self, seqs, input_ids=None, positions=None, slot_mapping=None, context_lens=None,
                              req_indices=None, *, graph_batch_size: int | None = None, defer_device_launch: bool = False):
        """h2o.py:256-476 / snapkv.py:2961.  With the device-resident state the allocation is `svk_h2o_device_step_begin`
        (launched here, or by the caller inside its hipGraph when `defer_device_launch`), nothing is uploaded; otherwise the
        host-driven form (`_prepare_decode_static_host`)."""
        real = len(seqs)
        gbs = int(graph_batch_size or (input_ids.numel() if input_ids is not None else real))
        plan = self._device_step_plan(seqs, gbs) if real > 0 and real <= gbs else None
        self._device_step = None
        if plan is None:
            self._dev_state_dirty = True
            return self._prepare_decode_static_host(seqs, input_ids, positions, slot_mapping, context_lens, req_indices,
                                                    graph_batch_size=graph_batch_size)
        args, rows_2d, kv_idx = plan
        if self._dev_state_dirty:
            self._device_state_upload()
        cur0 = self._row_seq_lens_all[kv_idx[0], rows_2d[0]]
        # the host mirrors move by the arithmetic the kernel applies to the device copy
        self._row_seq_lens_all[kv_idx[:, None], rows_2d] += 1
        for l in self.kv_transformer_layer_indices():
            self._num_free_slots[int(l)] -= real
        self._decode_static_rows = ((tuple(s.seq_id for s in seqs),), rows_2d, True, kv_idx)
        sm, cl, ri = self._get_decode_static_buffers(gbs)
        static_cap = self._decode_static_max_context_len
        for i, l in enumerate(self.kv_transformer_layer_indices()):
            st = self.layer_batch_states[l]
            st.slot_mapping, st.context_lens, st.req_indices = sm[i], cl[i], ri[i]
            st.max_context_len = int(static_cap) if static_cap is not None else int(cur0.max()) + 1
        self._device_step = [args, rows_2d, kv_idx, False]
        if not defer_device_launch:
            h2o_ops.h2o_device_step_begin(args)
        if slot_mapping is not None:
            slot_mapping.copy_(sm[0])
            context_lens.copy_(cl[0])
            req_indices.copy_(ri[0])
        return input_ids, positions, None

    def device_step_begin(self):
        """The step's allocation launch, for a caller that took `defer_device_launch` (inside its hipGraph)."""
        if self._device_step is not None:
            h2o_ops.h2o_device_step_begin(self._device_step[0])

    def device_step_burst(self):
        """The step's predicated burst launches (select + compact + commit), inside the caller's hipGraph."""
        if self._device_step is not None:
            h2o_ops.h2o_device_burst(self._device_step[0])
            self._device_step[3] = True

    def device_step_mark_launched(self):
        """A replayed hipGraph carried this step's launches."""
        if self._device_step is not None:
            self._device_step[3] = True

    def _device_step_finish(self, seqs) -> None:
        """Host half of a device-resident step's burst: launch it if the caller has not, and move the mirrors and
        counters (no device value is read)."""
        args, rows_2d, kv_idx, launched = self._device_step
        self._device_step = None
        if not launched:
            h2o_ops.h2o_device_burst(args)
        budget, trigger = int(args.budget), int(args.trigger_len)
        lens = self._row_seq_lens_all[kv_idx[0], rows_2d[0]]
        hit = lens == trigger
        n = int(hit.sum())
        if n == 0:
            return
        self._row_seq_lens_all[kv_idx[:, None], rows_2d[:, hit]] = budget
        for l in self.kv_transformer_layer_indices():
            self._num_free_slots[int(l)] += n * (trigger - budget)
        self._on_device_burst(seqs, n, len(kv_idx), trigger - budget)

    def _supports_nonuniform_decode_layers(self) -> bool:
        """SnapKV-family rows may differ across layers (full layers, per-layer budgets); H2O overrides this to False:
        its decode path needs aligned rows (h2o.py:256-271)."""
        return True

    def _prepare_decode(self, seqs):
        return self.prepare_decode_static(seqs)

    # ------------------------------------------------------------------ release / compaction
    def free_seq(self, seq_id: int):
        """snapkv.py:1489-1514."""
        self._decode_static_rows = None
        self._dev_state_dirty = True
        self._dev_step_cache = None
        for layer_idx in self.kv_transformer_layer_indices():
            row = self.seq_id_to_row[layer_idx].pop(seq_id, None)
            if row is None:
                raise ValueError(f"free_seq: unknown seq_id={seq_id}")
            cur = int(self.row_seq_lens[layer_idx][row])
            if cur > 0:
                ptr = self._num_free_slots[layer_idx]
                self.free_slots_stack[layer_idx][ptr: ptr + cur] = self.buffer_req_to_token_slots[layer_idx][row, :cur]
                self._num_free_slots[layer_idx] += cur
            self.buffer_req_to_token_slots[layer_idx][row, :] = 0
            self.row_seq_lens[layer_idx][row] = 0
            self.free_rows[layer_idx].append(row)
            self._on_row_released(layer_idx, row)

    def _on_row_released(self, layer_idx: int, row: int):
        return None

    def _row_payload_tensor(self):
        """Optional f32 rows compacted together with the slot table (H2O scores)."""
        return None

    def _compact(self, layer_indices, rows_2d: np.ndarray, keep: torch.Tensor, cur_len: int):
        """Uniform-length fused compaction through svk_compact_rows."""
        self._dev_state_dirty = True
        d = self.device
        n_layers, n_lanes, keep_len = keep.shape
        drop = cur_len - keep_len
        free_base = torch.tensor([self._num_free_slots[int(l)] for l in layer_indices], dtype=torch.long, device=d)
        for l in layer_indices:
            if self._num_free_slots[int(l)] + drop * n_lanes > self.num_slots:
                raise RuntimeError("compaction would overflow the free-slot stack: "
                                   f"layer={int(l)} end={self._num_free_slots[int(l)] + drop * n_lanes} capacity={self.num_slots}.")
        h2o_ops.compact_rows(
            self.buffer_req_to_token_slots_tensor, self.free_slots_stack_tensor, keep.contiguous(),
            torch.tensor([self.kv_layer_index(int(l)) for l in layer_indices], dtype=torch.int32, device=d),
            torch.from_numpy(np.ascontiguousarray(rows_2d, dtype=np.int32)).to(d),
            free_base, cur_len=cur_len, row_payload=self._row_payload_tensor())
        for i, l in enumerate(layer_indices):
            self._num_free_slots[int(l)] += drop * n_lanes
            self.row_seq_lens[int(l)][rows_2d[i]] = keep_len

    def _check_keep(self, keep: torch.Tensor, cur_len: int, sorted_: bool) -> torch.Tensor:
        keep = keep.to(device=self.device, dtype=torch.long).contiguous()
        if keep.numel() <= 0 or keep.shape[-1] <= 0:
            raise RuntimeError("free_part_slots got empty keep_indices")
        bounds_ok = ((keep >= 0) & (keep < cur_len)).all()
        torch._assert_async(bounds_ok)            # device-side check, no host sync (snapkv.py:1738-1746)
        if not sorted_:
            keep = torch.sort(keep, dim=-1).values
        return keep

    def free_part_slots(self, layer_idx: int, seq, keep_indices: torch.Tensor, *, keep_indices_sorted: bool = False):
        """snapkv.py:1528-1591."""
        if keep_indices is None:
            return
        self.kv_layer_index(layer_idx)
        self._uniform_decode_metadata = False
        row = self._row_of(layer_idx, seq)
        cur = int(self.row_seq_lens[layer_idx][row])
        keep = self._check_keep(keep_indices.reshape(-1), cur, keep_indices_sorted)
        self._compact([layer_idx], np.array([[row]]), keep.view(1, 1, -1), cur)

    def free_part_slots_batch(self, layer_idx: int, seqs, keep_indices: torch.Tensor, *, keep_indices_sorted: bool = False):
        """snapkv.py:1593-1679."""
        if keep_indices is None or not seqs:
            return
        self.free_part_slots_batch_layers([layer_idx], seqs, keep_indices.unsqueeze(0),
                                          keep_indices_sorted=keep_indices_sorted)

    def free_part_slots_batch_layers(self, layer_indices, seqs, keep_indices: torch.Tensor, *,
                                     keep_indices_sorted: bool = False):
        """snapkv.py:1681-1803: fused over layers when every (layer,row) has one length,
        otherwise the reference's per-row fallback order."""
        if keep_indices is None or not layer_indices or not seqs:
            return
        layer_indices = [int(l) for l in layer_indices]
        for l in layer_indices:
            self.kv_layer_index(l)
        self._uniform_decode_metadata = False
        n_layers, batch = len(layer_indices), len(seqs)
        if keep_indices.dim() != 3 or tuple(keep_indices.shape[:2]) != (n_layers, batch):
            raise RuntimeError("free_part_slots_batch_layers expected keep_indices with shape [layers, batch, keep]: "
                               f"layers={n_layers} batch={batch} keep_shape={tuple(keep_indices.shape)}")
        rows = np.array([[self._row_of(l, s) for s in seqs] for l in layer_indices], dtype=np.int64)
        lens = np.array([[int(self.row_seq_lens[l][r]) for r in rows[i]] for i, l in enumerate(layer_indices)])
        cur = int(lens[0, 0])
        if not np.all(lens == cur):
            for i, l in enumerate(layer_indices):
                for j, s in enumerate(seqs):
                    self.free_part_slots(l, s, keep_indices[i, j], keep_indices_sorted=keep_indices_sorted)
            return
        keep = self._check_keep(keep_indices, cur, keep_indices_sorted)
        self._compact(layer_indices, rows, keep, cur)

    def free_prefix_recent_slots_batch_layers(self, layer_indices, seqs, *, kv_len: int, num_sink_tokens: int,
                                              num_recent_tokens: int):
        """snapkv.py:1805-1896 (StreamingLLM sink + recent window)."""
        if not layer_indices or not seqs:
            return
        layer_indices = [int(l) for l in layer_indices]
        self._uniform_decode_metadata = False
        kv_len = int(kv_len)
        sink_end = min(int(num_sink_tokens), kv_len)
        recent_start = max(sink_end, kv_len - int(num_recent_tokens))
        new_len = sink_end + (kv_len - recent_start)
        if new_len <= 0:
            raise RuntimeError("prefix/recent compaction cannot keep zero tokens.")
        if new_len >= kv_len:
            return
        rows = np.array([[self._row_of(l, s) for s in seqs] for l in layer_indices], dtype=np.int64)
        lens = np.array([[int(self.row_seq_lens[l][r]) for r in rows[i]] for i, l in enumerate(layer_indices)])
        if not np.all(lens == kv_len):
            raise RuntimeError(f"prefix/recent compaction expected uniform row lengths: kv_len={kv_len} observed={lens.tolist()}")
        d = self.device
        keep1 = torch.cat((torch.arange(sink_end, device=d), torch.arange(recent_start, kv_len, device=d)))
        keep = keep1.expand(len(layer_indices), len(seqs), -1).contiguous()
        self._compact(layer_indices, rows, keep, kv_len)
