"""Quest: paged KV cache (16-token pages) + per-page min/max key metadata + query-aware page top-k.

Host mirror of `QuestCacheManager` (engine/cache_manager/quest.py:43-1913) for the hot path: page
allocator (:1210-1360), decode preparation (:1542-1605), metadata maintenance (:1607-1771), decode
view (:1804-1913).  The prefix-cache / offload half of that file is out of scope.

Token slot = page_slot * page_size + offset; one slot/page table for all layers (every layer keeps the
full row), LIFO page stack mirrored on the host like the reference's `free_pages_cpu_stack`.
"""

from __future__ import annotations

from collections import deque

import numpy as np
import torch

from ...kernels import quest_ops
from ...utils.context import get_context
from ...utils.profiler import profiler
from .base import AttentionViewMeta, CacheManager, DecodeComputeView, ExplicitKVPayload, LayerBatchStates


class QuestCacheManager(CacheManager):
    def __init__(self, config, parallel_context=None):
        super().__init__(config, parallel_context)
        self.page_size = int(config.quest_chunk_size)
        self.max_pages_per_row = (self.max_model_len + self.page_size - 1) // self.page_size
        self.allocate_kv_cache()
        d = self.device
        self.free_pages_cpu_stack = np.arange(self.num_pages, dtype=np.int32)
        self._num_free_pages = self.num_pages
        self.buffer_req_to_token_slots = torch.zeros((self.max_buffer_rows, self.max_model_len), dtype=torch.int32, device=d)
        self.buffer_req_to_page_slots = torch.full((self.max_buffer_rows, self.max_pages_per_row), -1, dtype=torch.int32, device=d)
        self.buffer_req_to_page_slots_cpu = np.full((self.max_buffer_rows, self.max_pages_per_row), -1, dtype=np.int32)
        self.seq_id_to_row: dict[int, int] = {}
        self.free_rows = deque(range(self.max_buffer_rows))
        self.row_seq_lens = np.zeros((self.max_buffer_rows,), dtype=np.int32)
        self.layer_batch_state = LayerBatchStates()
        # [2, L, P, Hkv, D]: 0 = max, 1 = min (quest.py:104-113)
        self.metadata_cache = torch.zeros((2, self.num_kv_layers, self.num_pages, self.num_kv_heads, self.head_dim),
                                          dtype=torch.bfloat16, device=d)
        self._static = None
        self._view_bufs: dict[tuple, tuple] = {}
        self._prefill_completed_pages: torch.Tensor | None = None
        # MI355X: device-resident decode bookkeeping.  Row lengths, the page stack and its pointer also live
        # on the device; a decode step allocates from them (`svk_quest_device_step_begin`) and refreshes the min / max rows
        # of the pages it completes behind a device-side test (`svk_quest_device_step_end`), so the step is one hipGraph and
        # uploads nothing.  The host keeps its numpy mirrors in lock-step by the same (deterministic) arithmetic; any
        # host-driven change (prefill allocation, free_seq) marks the device copy stale.  SVK_H2O_DEVICE_STATE=0: host-driven.
        import os
        self._device_step_enabled = os.environ.get("SVK_H2O_DEVICE_STATE", "1") == "1"
        self._dev_row_len = torch.zeros((self.max_buffer_rows,), dtype=torch.int32, device=d)
        self._dev_free_pages = torch.zeros((self.num_pages,), dtype=torch.int32, device=d)
        self._dev_free_page_ptr = torch.zeros((1,), dtype=torch.int32, device=d)
        self._dev_state_dirty = True
        self._dev_step_cache = None
        self.device_step_generation = 0
        self._device_step = None                 # this step's [args, end_launched] while the device-resident step is active
        # MI355X: the decode view is written as page slots (293 instead of 4672 entries per row) and the attention launch
        # addresses it by page (`slot_page_size`): the attended tokens and their order are unchanged
        # (tests/test_gpu_quest.py); False = the reference-shaped token-slot view.
        self.page_slot_view = (self.page_size & (self.page_size - 1)) == 0

    # ------------------------------------------------------------------ allocation
    def allocate_kv_cache(self):
        """quest.py:174-194: one extra min/max summary per physical page."""
        n = int(getattr(self.config, "num_kvcache_slots", 0) or 0)
        if n <= 0:
            if torch.cuda.is_available():
                free, _ = torch.cuda.mem_get_info(self.device)
                per = 2 * self.num_kv_layers * self.num_kv_heads * self.head_dim * 2
                n = int(int(free * 0.8) // int(per * (1.0 + 1.0 / self.page_size)))
            else:
                n = self.max_buffer_rows * min(self.max_model_len, 4096)
        n = (n // self.page_size) * self.page_size
        assert n > 0, "Available memory is insufficient for QuEST paged KV cache"
        self.config.num_kvcache_slots = n
        self.num_slots = n
        self.num_pages = n // self.page_size
        self.kv_cache = torch.zeros((2, self.num_kv_layers, n, self.num_kv_heads, self.head_dim), dtype=torch.bfloat16,
                                    device=self.device)

    def permute_free_pages(self, seed: int):
        assert self._num_free_pages == self.num_pages
        self._dev_state_dirty = True
        self.free_pages_cpu_stack = np.random.default_rng(seed).permutation(self.num_pages).astype(np.int32)

    def get_layer_batch_states(self, layer_idx: int) -> LayerBatchStates:
        return self.layer_batch_state

    def get_layer_kv_cache(self, layer_idx: int):
        i = self.kv_layer_index(layer_idx)
        return self.kv_cache[0, i], self.kv_cache[1, i]

    def get_layer_buffer_req_to_token_slots(self, layer_idx: int) -> torch.Tensor:
        return self.buffer_req_to_token_slots

    @property
    def num_free_slots(self) -> int:
        return int(self._num_free_pages) * self.page_size

    def _get_free_row(self, seq_id: int) -> int:
        row = self.seq_id_to_row.get(seq_id)
        if row is None:
            if not self.free_rows:
                raise RuntimeError("No free QuEST KV rows")
            row = self.free_rows.popleft()
            self.seq_id_to_row[seq_id] = row
        return int(row)

    def _required_new_pages(self, seq_id: int, size: int) -> int:
        row = self.seq_id_to_row.get(seq_id)
        cur = 0 if row is None else int(self.row_seq_lens[row])
        before = (cur + self.page_size - 1) // self.page_size
        after = (cur + int(size) + self.page_size - 1) // self.page_size
        return max(0, after - before)

    # ---- scheduler capacity hooks in page units (quest.py:272-378, without the prefix cache - out of scope here)
    def _ceil_to_page_slots(self, n_tokens: int) -> int:
        n_tokens = int(n_tokens)
        return 0 if n_tokens <= 0 else ((n_tokens + self.page_size - 1) // self.page_size) * self.page_size

    def _partial_page_free_slots(self) -> int:
        total = 0
        for row in self.seq_id_to_row.values():
            off = int(self.row_seq_lens[row]) % self.page_size
            if off:
                total += self.page_size - off
        return total

    def prefill_step_free_slots(self) -> int:
        return int(self.num_free_slots + self._partial_page_free_slots())

    def prefill_step_free_slots_for(self, seq) -> int:
        row = self.seq_id_to_row.get(seq.seq_id)
        partial = 0
        if row is not None:
            off = int(self.row_seq_lens[row]) % self.page_size
            if off:
                partial = self.page_size - off
        return int(self.num_free_slots + partial)

    def prefill_step_reservation_cost(self, seq, scheduled_tokens: int) -> int:
        row = self.seq_id_to_row.get(seq.seq_id)
        cur = 0 if row is None else int(self.row_seq_lens[row])
        remaining, cost = int(scheduled_tokens), 0
        off = cur % self.page_size
        if off:
            take = min(remaining, self.page_size - off)
            cost += take
            remaining -= take
        if remaining > 0:
            cost += self._ceil_to_page_slots(remaining)
        return int(cost)

    def decode_step_free_slots(self) -> int:
        partial_rows = sum(1 for row in self.seq_id_to_row.values() if int(self.row_seq_lens[row]) % self.page_size)
        return int(self._num_free_pages * self.page_size + partial_rows)

    def decode_step_free_slots_for(self, seq) -> int:
        if self._required_new_pages(seq.seq_id, 1) == 0:
            return 1
        return self.page_size if self._num_free_pages > 0 else 0

    def decode_step_reservation_cost(self, seq) -> int:
        return 1 if self._required_new_pages(seq.seq_id, 1) == 0 else self.page_size

    def prompt_admission_free_slots(self) -> int:
        return int(self.num_free_slots)

    def prompt_admission_cost(self, seq) -> int:
        hit = int(getattr(seq, "prefix_cache_hit_len", 0) or 0)
        if hit > 0:
            raise NotImplementedError("QuEST prefix-cache admission is outside this build (SURVEY.md section 2)")
        return self._ceil_to_page_slots(int(seq.num_prompt_tokens))

    def prompt_logical_reservation_cost(self, seq) -> int:
        return int(self.prompt_admission_cost(seq))

    def reserved_prefill_slots(self, waiting_seqs, chunk_prefill_size: int) -> int:
        reserved = 0
        for seq in waiting_seqs:
            if 0 < seq.num_prefilled_tokens < seq.num_prompt_tokens:
                reserved += self._ceil_to_page_slots(int(seq.num_prompt_tokens - seq.num_prefilled_tokens))
        return reserved

    def _pop_pages(self, n: int) -> np.ndarray:
        """LIFO pop, newest first (quest.py:1253-1257 `[ptr-n:ptr][::-1]`)."""
        assert self._num_free_pages >= n, f"Out of QuEST KV pages: need_pages={n}, free_pages={self._num_free_pages}"
        ptr = self._num_free_pages
        out = self.free_pages_cpu_stack[ptr - n: ptr][::-1].copy()
        self._num_free_pages -= n
        return out

    @torch.no_grad()
    def _allocate(self, seq_id: int, size: int) -> torch.Tensor:
        """quest.py:1227-1277 (prefill append of `size` tokens)."""
        self._dev_state_dirty = True
        size = int(size)
        needed = self._required_new_pages(seq_id, size)
        row = self._get_free_row(seq_id)
        cur = int(self.row_seq_lens[row])
        if cur + size > self.max_model_len:
            raise RuntimeError("KV row length exceeds max_model_len in QuEST _allocate: "
                               f"seq_id={seq_id} row={row} cur_len={cur} size={size} max_model_len={self.max_model_len}")
        if needed > 0:
            first = (cur + self.page_size - 1) // self.page_size
            new_pages = self._pop_pages(needed)
            self.buffer_req_to_page_slots_cpu[row, first: first + needed] = new_pages
            self.buffer_req_to_page_slots[row, first: first + needed] = torch.from_numpy(new_pages).to(self.device)
        pos = np.arange(cur, cur + size)
        slots = self.buffer_req_to_page_slots_cpu[row, pos // self.page_size].astype(np.int64) * self.page_size + pos % self.page_size
        slots_gpu = torch.from_numpy(slots.astype(np.int32)).to(self.device)
        self.buffer_req_to_token_slots[row, cur: cur + size] = slots_gpu
        self.row_seq_lens[row] += size
        return slots_gpu

    def _prepare_prefill(self, seqs):
        parts, ctx, rows, completed = [], [], [], []
        for s in seqs:
            n = int(s.current_chunk_size)
            row = self._get_free_row(s.seq_id)
            cur = int(self.row_seq_lens[row])
            parts.append(self._allocate(s.seq_id, n))
            rows.append(row)
            ctx.append(cur + n)
            # pages this chunk completes (quest.py:1652-1685: full pages + completed partial pages)
            first_touched, last_full = cur // self.page_size, (cur + n) // self.page_size
            completed.extend(int(x) for x in self.buffer_req_to_page_slots_cpu[row, first_touched:last_full])
        d = self.device
        st = self.layer_batch_state
        st.slot_mapping = torch.cat(parts) if parts else torch.empty(0, dtype=torch.int32, device=d)
        st.context_lens = torch.tensor(ctx, dtype=torch.int32, device=d)
        st.req_indices = torch.tensor(rows, dtype=torch.int32, device=d)
        st.max_context_len = max(ctx) if ctx else 0
        self._prefill_completed_pages = torch.tensor(completed, dtype=torch.long, device=d) if completed else None

    def _device_state_upload(self):
        self._dev_row_len.copy_(torch.from_numpy(self.row_seq_lens))
        self._dev_free_pages.copy_(torch.from_numpy(self.free_pages_cpu_stack))
        self._dev_free_page_ptr.fill_(int(self._num_free_pages))
        self._dev_state_dirty = False

    @torch.no_grad()
    def prepare_decode_static(self, seqs, input_ids=None, positions=None, slot_mapping=None, context_lens=None,
                              req_indices=None, *, graph_batch_size: int | None = None, defer_device_launch: bool = False):
        """quest.py:1542-1605 + _allocate_batch :1279-1360.  With the device-resident state the allocation is
        `svk_quest_device_step_begin` (launched here, or by the caller inside its hipGraph when `defer_device_launch`) and
        nothing is uploaded; the host pops the same pages from its mirror of the stack."""
        with profiler.record("cache_prepare_decode"):
            B = len(seqs)
            if B <= 0:
                raise ValueError("Static decode requires a non-empty real decode batch.")
            GB = int(graph_batch_size or (slot_mapping.numel() if slot_mapping is not None else B))
            if B > GB:
                raise ValueError(f"Static decode graph batch is smaller than the real decode batch: graph={GB}, real={B}.")
            rows = np.asarray([self._get_free_row(s.seq_id) for s in seqs], dtype=np.int64)
            cur = self.row_seq_lens[rows].copy()
            if int(cur.max()) + 1 > self.max_model_len:
                raise RuntimeError(f"KV row length exceeds max_model_len in QuEST _allocate_batch: max_cur_len={int(cur.max())} "
                                   f"max_model_len={self.max_model_len}")
            d = self.device
            caller_buffers = slot_mapping is not None
            if slot_mapping is None:
                if self._static is None or self._static[0].numel() < GB:
                    self._static = tuple(torch.zeros((GB,), dtype=torch.int32, device=d) for _ in range(3))
                    self._dev_step_cache = None
                slot_mapping, context_lens, req_indices = (t[:GB] for t in self._static)
            self._device_step = None
            use_device = self._device_step_enabled and not caller_buffers
            if use_device:
                # the device copy must be current BEFORE the host mirror moves
                if self._dev_state_dirty:
                    self._device_state_upload()
                key = (tuple(int(r) for r in rows), GB, slot_mapping.data_ptr())
                cache = self._dev_step_cache
                if cache is None or cache[0] != key:
                    rows_gpu = torch.from_numpy(rows.astype(np.int32)).to(d)
                    args = quest_ops.device_step_args(
                        self.buffer_req_to_page_slots, self.buffer_req_to_token_slots, self._dev_row_len, self._dev_free_pages,
                        self._dev_free_page_ptr, rows_gpu, slot_mapping, context_lens, req_indices, self.kv_cache,
                        self.metadata_cache, batch=B, page_size=self.page_size)
                    cache = self._dev_step_cache = (key, args, (rows_gpu, slot_mapping, context_lens, req_indices))
                    self.device_step_generation += 1
                self._device_step = [cache[1], False]
            need = np.nonzero(cur % self.page_size == 0)[0]
            new_pages = np.full((B,), -1, dtype=np.int32)
            if need.size:
                pages = self._pop_pages(int(need.size))
                new_pages[need] = pages
                self.buffer_req_to_page_slots_cpu[rows[need], cur[need] // self.page_size] = pages
            if use_device:
                if not defer_device_launch:
                    quest_ops.device_step_begin(self._device_step[0])
            else:
                self._dev_state_dirty = True
                quest_ops.decode_alloc(self.buffer_req_to_page_slots, self.buffer_req_to_token_slots,
                                       torch.from_numpy(rows.astype(np.int32)).to(d), torch.from_numpy(cur.astype(np.int32)).to(d),
                                       torch.from_numpy(new_pages).to(d), slot_mapping, context_lens, req_indices,
                                       batch=B, page_size=self.page_size)
            self.row_seq_lens[rows] += 1
            st = self.layer_batch_state
            st.slot_mapping, st.context_lens, st.req_indices = slot_mapping, context_lens, req_indices
            cap = self._decode_static_max_context_len
            st.max_context_len = int(cap) if cap is not None else int(cur.max()) + 1
            return input_ids, positions, None

    def device_step_begin(self):
        """The step's allocation launch, for a caller that took `defer_device_launch` (inside its hipGraph)."""
        if self._device_step is not None:
            quest_ops.device_step_begin(self._device_step[0])

    def device_step_mark_launched(self):
        """A replayed hipGraph carried this step's launches."""
        if self._device_step is not None:
            self._device_step[1] = True

    def device_step_burst(self):
        """The step's predicated page min / max refresh (after the layer loop: the completed pages' keys are stored),
        inside the caller's hipGraph."""
        if self._device_step is not None:
            quest_ops.device_step_end(self._device_step[0])
            self._device_step[1] = True

    def _prepare_decode(self, seqs):
        return self.prepare_decode_static(seqs)

    def free_seq(self, seq_id: int):
        """quest.py:1379-1420."""
        self._dev_state_dirty = True
        self._dev_step_cache = None
        row = self.seq_id_to_row.pop(seq_id, None)
        if row is None:
            raise ValueError(f"free_seq: unknown seq_id={seq_id}")
        n = (int(self.row_seq_lens[row]) + self.page_size - 1) // self.page_size
        if n > 0:
            pages = self.buffer_req_to_page_slots_cpu[row, :n].copy()
            self.free_pages_cpu_stack[self._num_free_pages: self._num_free_pages + n] = pages
            self._num_free_pages += n
        self.buffer_req_to_token_slots[row, :] = 0
        self.buffer_req_to_page_slots[row, :] = -1
        self.buffer_req_to_page_slots_cpu[row, :] = -1
        self.row_seq_lens[row] = 0
        self.free_rows.append(row)

    def free_part_slots(self, layer_idx: int, seq, keep_indices, *, keep_indices_sorted: bool = False):
        raise ValueError("QuEST does not physically evict token slots")

    # ------------------------------------------------------------------ metadata
    def save_rope_kv_if_needed(self, layer_idx: int, k: torch.Tensor, v: torch.Tensor):
        super().save_rope_kv_if_needed(layer_idx, k, v)
        self.on_kv_stored(layer_idx, k, self.layer_batch_state.slot_mapping)

    def fused_decode_store_slots(self, layer_idx: int):
        """Decode stores have no side effect here (page min/max is refreshed after the step, `on_forward_end`), and the
        newest token is always the last entry of the attention view (the last page is always attended), so the store can
        ride in the stage-1 launch like for the plain slot-table managers."""
        import os
        if get_context().is_prefill or os.environ.get("SVK_FUSE_DECODE_STORE", "1") != "1":
            return None
        return self.layer_batch_state.slot_mapping

    @torch.no_grad()
    def on_kv_stored(self, layer_idx: int, k: torch.Tensor, slot_mapping: torch.Tensor):
        """quest.py:1607-1685: prefill refreshes the pages completed by this chunk (after the
        store, from the cache); decode metadata is page-level and refreshed in on_forward_end."""
        if not get_context().is_prefill or self._prefill_completed_pages is None:
            return
        with profiler.record("quest_update_metadata"):
            kv_idx = self.kv_layer_index(layer_idx)
            quest_ops.page_minmax(self.kv_cache, self.metadata_cache, self._prefill_completed_pages,
                                  page_size=self.page_size, layers=slice(kv_idx, kv_idx + 1))

    @torch.no_grad()
    def on_forward_end(self, seqs, is_prefill: bool):
        """quest.py:1718-1771: pages completed by this decode step, all layers in one launch."""
        if is_prefill or not seqs:
            return
        if self._device_step is not None:
            # device-resident step: the refresh is the predicated launch of the step (issued here if the caller has not)
            args, launched = self._device_step
            self._device_step = None
            if not launched:
                quest_ops.device_step_end(args)
            return
        pages = []
        for s in seqs:
            row = self.seq_id_to_row.get(s.seq_id)
            if row is None:
                continue
            n = int(self.row_seq_lens[row])
            if n > 0 and n % self.page_size == 0:
                pages.append(int(self.buffer_req_to_page_slots_cpu[row, n // self.page_size - 1]))
        if not pages:
            return
        if min(pages) < 0:
            raise RuntimeError(f"QuEST decode completed a page with an invalid physical page slot: pages={pages}.")
        with profiler.record("quest_update_metadata_decode_pages"):
            quest_ops.page_minmax(self.kv_cache, self.metadata_cache, torch.tensor(pages, dtype=torch.long, device=self.device),
                                  page_size=self.page_size)

    # ------------------------------------------------------------------ decode view
    def _view_buffers(self, batch: int, n_prev: int, keep: int):
        key = (batch, n_prev, keep)
        buf = self._view_bufs.get(key)
        if buf is None:
            d = self.device
            # (rows of 16-byte multiples: the view kernel reads the scores, and writes the view, with vector accesses)
            buf = (torch.empty((batch, (n_prev + 3) // 4 * 4), dtype=torch.float32, device=d)[:, :n_prev],
                   torch.zeros((batch, (keep + 3) // 4 * 4), dtype=torch.int32, device=d)[:, :keep],
                   torch.empty((batch,), dtype=torch.int32, device=d), torch.empty((batch,), dtype=torch.int32, device=d))
            self._view_bufs[key] = buf
        return buf

    @torch.no_grad()
    def build_decode_view(self, layer_idx: int, q: torch.Tensor, active_slots: torch.Tensor, req_indices: torch.Tensor,
                          context_lens: torch.Tensor, *, num_heads: int, num_kv_heads: int, page_slots: bool = False):
        """quest.py:1804-1913.  `page_slots`: the packed view holds page slots (its first ceil(keep / page_size) columns)."""
        if layer_idx < self.config.quest_skip_layers:
            return active_slots, req_indices, context_lens
        token_budget = int(self.config.quest_token_budget)
        if token_budget <= 0:
            return active_slots, req_indices, context_lens
        with profiler.record("quest_build_decode_view_static"):
            kv_idx = self.kv_layer_index(layer_idx)
            page_budget_base = max(3, token_budget // self.page_size)
            max_keep = max(token_budget, page_budget_base * self.page_size, self.page_size)
            max_context_len = self.layer_batch_state.max_context_len
            if max_context_len is None:
                raise RuntimeError("QuEST decode CUDA graph requires max_context_len to be pinned.")
            max_context_len = int(max_context_len)
            if max_context_len <= max_keep:
                return active_slots, req_indices, context_lens
            batch = q.shape[0]
            max_pages = min(self.max_pages_per_row, (max_context_len + self.page_size - 1) // self.page_size)
            prev_budget = min(page_budget_base - 1, max_pages - 1)
            if prev_budget <= 0:
                return active_slots, req_indices, context_lens
            is_long_text = bool(get_context().is_long_text)
            sparse_keep = (prev_budget + 1) * self.page_size
            keep = sparse_keep if is_long_text else max_keep
            scores, packed, lens, lreq = self._view_buffers(batch, max_pages - 1, keep)
            quest_ops.score_pages(q, self.metadata_cache[0, kv_idx], self.metadata_cache[1, kv_idx],
                                  self.buffer_req_to_page_slots, req_indices, context_lens, scores,
                                  page_size=self.page_size, n_prev=max_pages - 1)
            quest_ops.build_view(scores, self.buffer_req_to_page_slots, self.buffer_req_to_token_slots, req_indices,
                                 context_lens, packed, lens, lreq, page_size=self.page_size, n_prev=max_pages - 1,
                                 prev_budget=prev_budget, token_budget=token_budget, page_budget_base=page_budget_base,
                                 max_keep=keep, is_long_text=is_long_text, emit_page_slots=page_slots)
            return packed, lreq, lens

    def build_decode_compute_view(self, layer_idx: int, q: torch.Tensor, selection, *, num_heads: int, num_kv_heads: int):
        """base.py:1162-1206 with the Quest view hook."""
        k_cache, v_cache = self.get_layer_compute_tensors(layer_idx)
        paged = bool(self.page_slot_view and q.is_cuda)
        slots, req, lens = self.build_decode_view(layer_idx, q, self.buffer_req_to_token_slots, selection.req_indices,
                                                  selection.context_lens, num_heads=num_heads, num_kv_heads=num_kv_heads,
                                                  page_slots=paged)
        max_ctx = selection.max_context_len
        metadata = None
        if slots is not self.buffer_req_to_token_slots:
            max_ctx = int(slots.shape[1])          # the view's width in tokens, whichever way its slots are written
            if paged:
                metadata = {"slot_page_size": self.page_size}
        meta = AttentionViewMeta(active_slots=slots, req_indices=req, context_lens=lens, max_context_len=max_ctx,
                                 attn_score=selection.attn_score)
        return DecodeComputeView(meta=meta, payload=ExplicitKVPayload(k_cache=k_cache, v_cache=v_cache, metadata=metadata))
