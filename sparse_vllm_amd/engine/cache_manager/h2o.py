"""H2O physical KV eviction: one cumulative token-importance vector per (layer, row).

Host mirror of `H2OCacheManager` (engine/cache_manager/h2o.py:25-1674).  Differences in
*representation* only (results are the reference's):
  * the reference keeps a Python dict {(layer, seq_id): 1-D tensor} and restacks 28*B
    tensors every token (h2o.py:1009-1037); here the scores live in ONE persistent device
    tensor `h2o_score_tensor[L, rows, max_model_len]` aligned with the slot table, updated
    by the fused decode-score kernel and compacted by the same kernel that compacts the
    slot table;
  * selection, compaction and the score gather are libsvk kernels (bit-exact indices).
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

from ...kernels import h2o_ops
from ...utils.context import get_context
from .base import ExplicitKVPayload, PrefillComputeView
from .snapkv import SnapKVCacheManager


@dataclass(frozen=True)
class _H2ORowRef:
    """Stand-in for an unscheduled active decode row (h2o.py `_H2ORowRef`)."""
    seq_id: int


class H2OCacheManager(SnapKVCacheManager):
    def __init__(self, config, parallel_context=None):
        super().__init__(config, parallel_context)
        self._h2o_active_decode_seq_ids: set[int] = set()
        self._h2o_counters = {
            "intermediate_prefill_evictions": 0,
            "final_prefill_evictions": 0,
            "decode_eviction_bursts": 0,
            "decode_evictions": 0,
            "dropped_tokens": 0,
        }
        self._h2o_final_prefill_workspace: torch.Tensor | None = None
        self.h2o_score_tensor = torch.zeros(
            (self.num_kv_layers, self.max_buffer_rows, self.max_model_len), dtype=torch.float32, device=self.device)

    @property
    def h2o_decode_budget(self) -> int:
        return int(self.config.h2o_decode_budget)

    @property
    def h2o_decode_eviction_interval(self) -> int:
        return int(self.config.h2o_decode_eviction_interval)

    @property
    def h2o_prefill_budget(self) -> int:
        return int(self.config.h2o_prefill_budget)

    def _h2o_budget_partition(self) -> tuple[int, int]:
        budget = self.h2o_decode_budget
        recent_count = max(1, int(budget * float(self.config.h2o_recent_ratio)))
        recent_count = min(recent_count, budget)
        return budget - recent_count, recent_count

    def decode_cuda_graph_context_capacity(self, seqs=None, *, requested_context_capacity: int = 0,
                                           current_context_capacity: int = 0) -> tuple[int, bool]:
        """h2o.py:241-254: graph capacity = the periodic decode peak."""
        capacity = min(self.h2o_decode_budget + self.h2o_decode_eviction_interval, int(self.config.max_model_len))
        return max(1, capacity), False

    def _supports_nonuniform_decode_layers(self) -> bool:
        """H2O appends and evicts one token on every KV layer: rows, lengths and stack pointers stay aligned and the
        static decode path relies on it (h2o.py:256-271, :306-330)."""
        return False

    # ------------------------------------------------------------------ scheduler capacity hooks (h2o.py:73-230)
    # An H2O row never holds more than max(resident, prefill budget) + one chunk during prefill (append, then evict), and
    # budget + interval during decode; the scheduler reserves those physical peaks, not the logical prompt length.
    def _prefill_append_peak(self, resident_len: int, remaining_tokens: int, chunk_prefill_size: int) -> int:
        resident, remaining = int(resident_len), max(0, int(remaining_tokens))
        chunk = max(1, int(chunk_prefill_size))
        return min(resident + remaining, max(resident, self.h2o_prefill_budget) + chunk)

    def _prompt_prefill_peak(self, seq, chunk_prefill_size: int) -> int:
        return self._prefill_append_peak(0, int(seq.num_prompt_tokens), chunk_prefill_size)

    def prompt_admission_cost(self, seq) -> int:
        return self._prompt_prefill_peak(seq, int(self.config.chunk_prefill_size))

    def prompt_admission_free_slots(self) -> int:
        return int(self.num_free_slots)

    def prompt_admission_budgets(self, waiting_seqs, chunk_prefill_size: int) -> dict:
        return {"slots": max(0, int(self.num_free_slots) - int(self.reserved_prefill_slots(waiting_seqs, chunk_prefill_size)))}

    def prompt_admission_costs(self, seq) -> dict:
        return {"slots": self.prompt_admission_cost(seq)}

    def prompt_logical_reservation_cost(self, seq) -> int:
        return self.prompt_admission_cost(seq)

    def reserved_prefill_slots(self, waiting_seqs, chunk_prefill_size: int) -> int:
        """Growth still needed by the prompts that are part-way through prefill, measured on their physical rows."""
        first_layer = int(self.kv_transformer_layer_indices()[0])
        total = 0
        for seq in waiting_seqs:
            done, prompt = int(seq.num_prefilled_tokens), int(seq.num_prompt_tokens)
            if not 0 < done < prompt:
                continue
            have = self._physical_row_len(first_layer, seq)
            total += max(0, self._prefill_append_peak(have, prompt - done, chunk_prefill_size) - have)
        return int(total)

    def prefill_step_free_slots(self) -> int:
        return int(self.num_free_slots)

    def prefill_step_free_slots_for(self, seq) -> int:
        return int(self.num_free_slots)

    def prefill_step_reservation_cost(self, seq, scheduled_tokens: int) -> int:
        return int(scheduled_tokens)

    def decode_step_free_slots(self) -> int:
        return int(self.num_free_slots)

    def decode_step_free_slots_for(self, seq) -> int:
        return int(self.num_free_slots)

    def decode_step_reservation_cost(self, seq) -> int:
        return 1

    def chain_capacity_deficits(self, *, suffix_tokens: int, generation_tokens: int = 0, existing_slots_by_layer=(),
                                outstanding_reserved_slots_by_layer=(), outstanding_reserved_rows: int = 0,
                                needs_resident_row: bool):
        """(required slots per layer, required rows, slot deficits per layer, row deficit) of one chain turn: `suffix_tokens`
        of prefill on top of the existing rows, then `generation_tokens` of decode (the first token is produced by the
        prefill itself)."""
        suffix = max(0, int(suffix_tokens))
        new_kv = max(0, int(generation_tokens) - 1)
        chunk = max(1, int(self.config.chunk_prefill_size))
        trigger = self.h2o_decode_budget + self.h2o_decode_eviction_interval
        layers = self.kv_transformer_layer_indices()

        def at(seq_, i):
            return int(seq_[i]) if i < len(seq_) else 0

        required = []
        for i in range(len(layers)):
            have = at(existing_slots_by_layer, i)
            prefill_peak = self._prefill_append_peak(have, suffix, chunk)
            resident = min(have + suffix, self.h2o_decode_budget) if suffix > 0 else have   # final-prefill compaction
            if new_kv <= 0:
                decode_peak = resident
            elif resident >= trigger:
                decode_peak = resident + 1
            else:
                decode_peak = resident + min(new_kv, trigger - resident)
            required.append(max(0, max(prefill_peak, decode_peak) - have))
        slot_deficits = tuple(
            max(0, need - max(0, int(self._num_free_slots[layer]) - at(outstanding_reserved_slots_by_layer, i)))
            for i, (layer, need) in enumerate(zip(layers, required)))
        rows_needed = 1 if needs_resident_row else 0
        rows_free = max(0, min((len(self.free_rows[layer]) for layer in layers), default=0) - max(0, int(outstanding_reserved_rows)))
        return tuple(required), rows_needed, slot_deficits, max(0, rows_needed - rows_free)

    @staticmethod
    def select_h2o_indices_batch(scores: torch.Tensor, *, budget: int, recent_ratio: float) -> torch.Tensor:
        return h2o_ops.select_h2o_indices_batch(scores, budget=budget, recent_ratio=recent_ratio)

    @staticmethod
    def select_h2o_indices(scores: torch.Tensor, *, budget: int, recent_ratio: float) -> torch.Tensor:
        if scores.dim() != 1:
            raise ValueError(f"H2O scores must be 1D, got shape={tuple(scores.shape)}.")
        return h2o_ops.select_h2o_indices_batch(scores.unsqueeze(0), budget=budget, recent_ratio=recent_ratio)[0]

    # ---- score rows
    def _row_payload_tensor(self):
        return self.h2o_score_tensor

    def _on_row_released(self, layer_idx: int, row: int):
        self.h2o_score_tensor[self.kv_layer_index(layer_idx), row].zero_()

    def _physical_row_len(self, layer_idx: int, seq) -> int:
        row = self.seq_id_to_row[layer_idx].get(int(seq.seq_id))
        if row is None:
            raise RuntimeError(f"H2O physical row is missing: layer={layer_idx} seq_id={seq.seq_id}.")
        return int(self.row_seq_lens[layer_idx][row])

    def h2o_score(self, layer_idx: int, seq_id: int) -> torch.Tensor | None:
        row = self.seq_id_to_row[layer_idx].get(int(seq_id))
        if row is None:
            return None
        return self.h2o_score_tensor[self.kv_layer_index(layer_idx), row, : int(self.row_seq_lens[layer_idx][row])]

    def set_h2o_score(self, layer_idx: int, seq_id: int, score: torch.Tensor):
        row = self.seq_id_to_row[layer_idx][int(seq_id)]
        n = int(score.numel())
        dst = self.h2o_score_tensor[self.kv_layer_index(layer_idx), row]
        dst[:n].copy_(score)
        dst[n:].zero_()

    def update_decode_attention_scores_all_layers(self, layer_indices, seqs, reduced_scores: torch.Tensor) -> bool:
        """h2o.py:957-1038 as a standalone op: cum[:, :, :kv_len] = pad(prev,1) + normalized.
        The decode loop normally never calls this - the accumulation is fused into
        svk_h2o_decode_score_update at `on_layer_attention_end`; kept for API parity."""
        if reduced_scores.dim() != 3:
            raise ValueError("H2O all-layer decode scores must have shape [layers, batch, width], "
                             f"got {tuple(reduced_scores.shape)}.")
        if tuple(reduced_scores.shape[:2]) != (len(layer_indices), len(seqs)):
            raise ValueError("H2O all-layer decode score shape does not match layers and batch: "
                             f"layers={len(layer_indices)} batch={len(seqs)} shape={tuple(reduced_scores.shape)}.")
        uniform = True
        for i, layer_idx in enumerate(layer_indices):
            for b, seq in enumerate(seqs):
                kv_len = self._physical_row_len(layer_idx, seq)
                row = self.seq_id_to_row[layer_idx][seq.seq_id]
                dst = self.h2o_score_tensor[self.kv_layer_index(layer_idx), row]
                dst[kv_len - 1] = 0
                dst[:kv_len] += reduced_scores[i, b, :kv_len]
                uniform &= kv_len == self._physical_row_len(layer_indices[0], seqs[0])
        return uniform

    # ---- prefill score accumulation (h2o.py:750-894)
    def prefill_score_ranges(self, layer_idx: int, seqs):
        """-> [(batch, seq, prompt_cache_len, score_start, score_end)] in compressed physical coordinates."""
        window = int(self.config.h2o_prefill_score_window)
        ranges = []
        for b, seq in enumerate(seqs):
            chunk_len = int(seq.current_chunk_size)
            context_len = self._physical_row_len(layer_idx, seq)
            cache_len = context_len - chunk_len
            if cache_len < 0:
                raise RuntimeError("H2O current chunk exceeds its physical row: "
                                   f"layer={layer_idx} seq_id={seq.seq_id} context={context_len} chunk={chunk_len}.")
            start = max(cache_len, context_len - window) if window > 0 else cache_len
            ranges.append((b, seq, cache_len, start, context_len))
        return ranges

    def _prefill_score_meta_tensors(self, ranges, d):
        """(cache_lens, starts, ends) int32 on the device; rows are uniform across layers: one upload per chunk, not per
        layer (the reference caches the same tensors, snapkv.py:1155-1214 `_cached_prefill_score_metadata_tensors`)."""
        key = tuple((r[2], r[3], r[4]) for r in ranges)
        cached = getattr(self, "_prefill_score_meta", None)
        if cached is None or cached[0] != key or cached[1].device != d:
            meta = torch.tensor([[r[2] for r in ranges], [r[3] for r in ranges], [r[4] for r in ranges]], dtype=torch.int32, device=d)
            cached = (key, meta)
            self._prefill_score_meta = cached
        return key, cached[1][0], cached[1][1], cached[1][2]

    def _prefill_statistics_request(self, layer_idx: int, seqs, d):
        """MI355X: in the probability mode the H2O score of a key is built from softmax rows over ALL causal keys
        (candidate_start = 0, num_recent = 0) - the very statistics the prefill attention of the layer computes - so
        the attention launch is asked to leave the window rows' statistics behind and to zero the step-score rows;
        `collect_prefill_attention_score` then needs one scoring pass instead of three launches (Q.K^T of the window
        once instead of twice).  -> {"request": `score_stats` of kernels.context_attention_fwd, ...} or None."""
        enabled = self.__dict__.get("_prefill_fuse_enabled")
        if enabled is None:
            import os
            enabled = self._prefill_fuse_enabled = (
                self.config.sparse_prefill_score_mode == "probability" and self.head_dim == 128
                and self.num_heads // self.num_kv_heads <= 8 and os.environ.get("SVK_PREFILL_SCORE_FUSE", "1") == "1")
            self._prefill_wpad = {}
        if not enabled or seqs is None:
            return None
        ranges = self.prefill_score_ranges(layer_idx, seqs)
        if not ranges:
            return None
        max_q = max(r[4] - r[3] for r in ranges)
        if max_q <= 0 or max_q > 128:
            return None
        key, cache_lens, starts, _ends = self._prefill_score_meta_tensors(ranges, d)
        wpad = self._prefill_wpad.get(max_q)
        if wpad is None:
            from ...kernels.prefill_score import prefill_score_window_pad
            wpad = self._prefill_wpad[max_q] = prefill_score_window_pad(self.num_heads, self.num_kv_heads, max_q)
        n = len(seqs) * self.num_heads * wpad
        buf = getattr(self, "_prefill_stats_buf", None)
        if buf is None or buf.numel() < n or buf.device != d:
            buf = self._prefill_stats_buf = torch.zeros((n,), dtype=torch.float32, device=d)
        step = torch.empty((len(seqs), max(r[4] for r in ranges)), dtype=torch.float32, device=d)
        # `written_for` is set by the attention backend after its launch: (q storage, q version, q shape)
        return {"request": (buf, starts, wpad, step), "layer": int(layer_idx), "key": key, "row_stats": buf,
                "step": step, "written_for": None, "b_prompt_cache_len": cache_lens}

    def build_prefill_compute_view(self, layer_idx: int, k_current: torch.Tensor, v_current: torch.Tensor,
                                   selection) -> PrefillComputeView:
        """base.py:892-931 (the plain slot table of the physical row).  MI355X: the view's payload also carries the
        request for the attention launch's softmax statistics (see `_prefill_statistics_request`) - an internal detail
        between this manager and the HIP attention backend; a backend that ignores it gets the three-launch scoring."""
        view = super().build_prefill_compute_view(layer_idx, k_current, v_current, selection)
        ctx = get_context()
        if not ctx.is_prefill or not isinstance(view.payload, ExplicitKVPayload) or not view.payload.k_cache.is_cuda:
            return view
        stats = self._prefill_statistics_request(layer_idx, getattr(ctx, "seqs", None), view.payload.k_cache.device)
        if stats is None:
            return view
        md = dict(view.payload.metadata or {})
        md["score_stats"] = stats
        md["b_prompt_cache_len"] = stats["b_prompt_cache_len"]
        payload = ExplicitKVPayload(k_cache=view.payload.k_cache, v_cache=view.payload.v_cache,
                                    backend=view.payload.backend, metadata=md)
        return PrefillComputeView(meta=view.meta, payload=payload)

    @torch.no_grad()
    def collect_prefill_attention_score(self, layer_idx: int, q: torch.Tensor, view: PrefillComputeView, *,
                                        b_start_loc: torch.Tensor, chunk_lens: torch.Tensor):
        """h2o.py:777-894: cum[:len] = expand(prev, len) + W_eff * step_score (logits mode: softmax of the vector
        first), the window = the last min(h2o_prefill_score_window, chunk) queries in compressed physical coordinates,
        candidate_start = 0, num_recent = 0."""
        ctx = get_context()
        if not ctx.is_prefill:
            raise RuntimeError("H2O prefill score collection was called outside prefill.")
        seqs = getattr(ctx, "seqs", None)
        if seqs is None:
            raise RuntimeError("H2O prefill score collection requires current seqs in context.")
        if int(chunk_lens.ndim) != 1 or int(chunk_lens.shape[0]) != len(seqs):
            raise RuntimeError("H2O prefill scoring chunk-length batch mismatch: "
                               f"shape={tuple(chunk_lens.shape)} seqs={len(seqs)}.")
        ranges = self.prefill_score_ranges(layer_idx, seqs)
        if not ranges:
            return None
        if not isinstance(view.payload, ExplicitKVPayload):
            raise TypeError(f"H2O prefill scoring requires ExplicitKVPayload, got {type(view.payload).__name__}.")
        meta, payload = view.meta, view.payload
        context_lens = tuple(int(r[4]) for r in ranges)
        prepared = self._prefill_context_lens_cpu_by_layer.get(int(layer_idx))
        if prepared is None and meta.context_lens.device.type == "cpu":
            prepared = tuple(int(x) for x in meta.context_lens.tolist())
        if prepared is None:
            raise RuntimeError(f"H2O prefill scoring requires CPU context lengths prepared for layer={layer_idx}.")
        if tuple(prepared) != context_lens:
            raise RuntimeError("H2O prefill score view is not in compressed physical coordinates: "
                               f"layer={layer_idx} view={tuple(prepared)} physical={context_lens}.")
        d = q.device
        key, cache_lens, starts, ends = self._prefill_score_meta_tensors(ranges, d)
        max_q = max(r[4] - r[3] for r in ranges)
        stats = (payload.metadata or {}).get("score_stats")
        if (stats is not None and stats["layer"] == int(layer_idx) and stats["key"] == key
                and stats["written_for"] == (q.data_ptr(), q._version, tuple(q.shape))):
            # the attention launch of this layer and chunk, on this very q, left the statistics and the cleared rows:
            # final pass only
            step = stats["step"]
            self._run_prefill_score(q, payload.k_cache, step, meta, b_start_loc, cache_lens, max_q, starts, ends,
                                    candidate_start=0, num_recent_tokens=0, row_stats=stats["row_stats"])
        else:
            step = torch.empty((len(seqs), max(context_lens)), dtype=torch.float32, device=d)
            self._run_prefill_score(q, payload.k_cache, step, meta, b_start_loc, cache_lens, max_q, starts, ends,
                                    candidate_start=0, num_recent_tokens=0)
        kv = self.kv_layer_index(layer_idx)
        for b, seq, cache_len, start, end in ranges:
            row = self.seq_id_to_row[layer_idx][int(seq.seq_id)]
            score_row = step[b, :end]
            if self.config.sparse_prefill_score_mode == "logits":
                score_row = torch.softmax(score_row.float(), dim=0)          # h2o.py:628-655
            self.h2o_score_tensor[kv, row, :end].add_(score_row, alpha=float(end - start))
        return None

    # ---- decode burst (h2o.py:1498-1630)
    def _decode_eviction_groups(self, seqs):
        seq_ids = [int(s.seq_id) for s in seqs]
        if len(seq_ids) != len(set(seq_ids)):
            raise RuntimeError(f"H2O decode eviction received duplicate sequence ids: {seq_ids}.")
        self._h2o_active_decode_seq_ids.update(seq_ids)
        layer_indices = [int(l) for l in self.kv_transformer_layer_indices()]
        budget = self.h2o_decode_budget
        periodic_trigger = budget + self.h2o_decode_eviction_interval
        under_pressure = self.num_free_slots <= 0
        trigger = budget + 1 if under_pressure else periodic_trigger
        if not under_pressure:
            # the common step evicts nothing: one numpy gather over all (layer, sequence) lengths decides it (the loop
            # below is 28 x B Python iterations per decode step)
            kv = self.decode_kv_lens_all_layers(seqs)
            if kv is not None and bool((kv == kv[0]).all()) and not bool((kv[0] >= trigger).any()):
                return layer_indices, {}
        seq_by_id = {int(s.seq_id): s for s in seqs}
        candidate_ids = list(seq_ids)
        if under_pressure:
            candidate_ids.extend(sorted(self._h2o_active_decode_seq_ids.difference(seq_ids)))
        groups: dict[int, list] = {}
        for seq_id in candidate_ids:
            seq = seq_by_id.get(seq_id, _H2ORowRef(seq_id))
            lens = [self._physical_row_len(l, seq) for l in layer_indices]
            if any(x != lens[0] for x in lens[1:]):
                raise RuntimeError("H2O decode eviction requires aligned KV-layer row lengths: "
                                   f"seq_id={seq.seq_id} lengths={lens}.")
            if lens[0] >= trigger:
                groups.setdefault(int(lens[0]), []).append(seq)
        return layer_indices, groups

    def _preflight_decode_eviction_capacity(self, layer_indices, groups) -> None:
        dropped = sum((kv_len - self.h2o_decode_budget) * len(g) for kv_len, g in groups.items())
        for l in layer_indices:
            end = int(self._num_free_slots[l]) + int(dropped)
            if end > self.num_slots:
                raise RuntimeError("H2O decode eviction would overflow the free-slot stack: "
                                   f"layer={l} end={end} capacity={self.num_slots}.")

    def _evict_decode_rows(self, seqs) -> None:
        layer_indices, groups = self._decode_eviction_groups(seqs)
        if not groups:
            return
        self._preflight_decode_eviction_capacity(layer_indices, groups)
        budget = self.h2o_decode_budget
        ratio = float(self.config.h2o_recent_ratio)
        evicted_rows = dropped_tokens = burst_count = 0
        L = len(layer_indices)
        for kv_len, group in groups.items():
            burst_count += len(group)
            rows = np.array([[self.seq_id_to_row[l][int(s.seq_id)] for s in group] for l in layer_indices])
            # scores of every (layer, lane) as one strided [L*n, kv_len] view - no restacking
            kv_layers = torch.tensor([self.kv_layer_index(l) for l in layer_indices], device=self.device)
            rows_gpu = torch.from_numpy(rows).to(self.device)
            scores = self.h2o_score_tensor[kv_layers[:, None], rows_gpu, :kv_len]
            keep = self.select_h2o_indices_batch(scores.view(-1, kv_len), budget=budget,
                                                 recent_ratio=ratio).view(L, len(group), budget)
            self._compact(layer_indices, rows, keep, kv_len)     # slot table + free stack + score rows
            self._uniform_decode_metadata = False
            n = L * len(group)
            evicted_rows += n
            dropped_tokens += (kv_len - budget) * n
        self._h2o_counters["decode_eviction_bursts"] += int(burst_count)
        self._h2o_counters["decode_evictions"] += int(evicted_rows)
        self._h2o_counters["dropped_tokens"] += int(dropped_tokens)

    def evict_after_decode(self, seqs):
        if not seqs:
            return
        if self._device_step is not None:
            self._device_step_finish(seqs)
            return
        self._evict_decode_rows(seqs)

    # ---- prefill-side eviction (h2o.py:1351-1496)
    def _evict_prefill(self, seqs):
        """h2o.py:1351-1480.  Per layer, sequences are visited in batch order (that order fixes the
        free-stack contents); maximal runs of consecutive sequences with the same finality and the
        same physical length are selected/compacted in one launch.  Rows are uniform across layers, so an
        intermediate-chunk run is selected and compacted for ALL layers in one launch each (every layer's free stack
        still sees its own operations in the same order); the final chunk's dense compaction stays per layer."""
        ratio = float(self.config.h2o_recent_ratio)
        layers = [int(l) for l in self.kv_transformer_layer_indices()]
        lens0 = [self._physical_row_len(layers[0], s) for s in seqs]
        uniform = all([self._physical_row_len(l, s) for s in seqs] == lens0 for l in layers[1:])
        layer_sets = [layers] if uniform else [[l] for l in layers]
        for lset in layer_sets:
            first = lset[0]
            i = 0
            while i < len(seqs):
                final = bool(seqs[i].is_last_chunk_prefill)
                kv_len = self._physical_row_len(first, seqs[i])
                j = i + 1
                while (j < len(seqs) and bool(seqs[j].is_last_chunk_prefill) == final
                       and self._physical_row_len(first, seqs[j]) == kv_len):
                    j += 1
                group = seqs[i:j]
                i = j
                budget = self.h2o_decode_budget if final else self.h2o_prefill_budget
                if kv_len <= budget:
                    continue
                rows = np.array([[self.seq_id_to_row[l][int(s.seq_id)] for s in group] for l in lset])     # [L', n]
                rows_gpu = torch.from_numpy(rows).to(self.device)
                if final:
                    for li, l in enumerate(lset):
                        scores = self.h2o_score_tensor[self.kv_layer_index(l), rows_gpu[li], :kv_len]
                        keep = self.select_h2o_indices_batch(scores, budget=budget, recent_ratio=ratio)
                        self._compact_final_prefill_dense_batch(l, group, keep)
                else:
                    kv_idx = torch.tensor([self.kv_layer_index(l) for l in lset], device=self.device)
                    scores = self.h2o_score_tensor[kv_idx[:, None], rows_gpu, :kv_len]                   # [L', n, kv_len]
                    keep = self.select_h2o_indices_batch(scores.reshape(-1, kv_len), budget=budget, recent_ratio=ratio)
                    self._compact(lset, rows, keep.view(len(lset), len(group), -1), kv_len)
                key = "final_prefill_evictions" if final else "intermediate_prefill_evictions"
                self._h2o_counters[key] += len(group) * len(lset)
                self._h2o_counters["dropped_tokens"] += (kv_len - budget) * len(group) * len(lset)

    def evict_after_prefill(self, seqs):
        self._evict_prefill(seqs)
        finals = [s for s in seqs if bool(s.is_last_chunk_prefill)]
        for layer_idx in self.kv_transformer_layer_indices():
            for s in finals:
                kv_len = self._physical_row_len(layer_idx, s)
                if kv_len > self.h2o_decode_budget:
                    raise RuntimeError("H2O final prefill did not compact to the decode budget: "
                                       f"layer={layer_idx} seq_id={s.seq_id} kv_len={kv_len} budget={self.h2o_decode_budget}.")
        if self.num_free_slots <= 0:
            self._evict_decode_rows([])

    def _get_final_prefill_workspace(self, batch_size: int, budget: int) -> torch.Tensor:
        """h2o.py:1065-1113."""
        need = (2, batch_size, budget, self.num_kv_heads, self.head_dim)
        ws = self._h2o_final_prefill_workspace
        if ws is None or ws.shape[1] < batch_size or tuple(ws.shape[2:]) != need[2:]:
            ws = torch.empty(need, dtype=torch.bfloat16, device=self.device)
            self._h2o_final_prefill_workspace = ws
        return ws[:, :batch_size]

    @torch.no_grad()
    def _compact_final_prefill_dense_batch(self, layer_idx: int, seqs, keep_indices: torch.Tensor) -> None:
        """h2o.py:1181-1349: the selected K/V rows MOVE into the `budget` smallest physical
        slots of the row (ascending), the larger slots are released."""
        if not seqs:
            raise RuntimeError("H2O final-prefill dense compaction requires sequences.")
        kv_idx = self.kv_layer_index(layer_idx)
        budget = self.h2o_decode_budget
        batch = len(seqs)
        keep = keep_indices.to(device=self.device, dtype=torch.long).contiguous()
        if keep.dim() != 2 or tuple(keep.shape) != (batch, budget):
            raise RuntimeError("H2O final-prefill keep indices must have shape [batch, decode_budget]: "
                               f"expected={(batch, budget)} got={tuple(keep.shape)}.")
        rows = [self.seq_id_to_row[layer_idx][int(s.seq_id)] for s in seqs]
        if len(rows) != len(set(rows)):
            raise RuntimeError(f"H2O final-prefill dense compaction received duplicate physical rows: rows={rows}.")
        lens = [int(self.row_seq_lens[layer_idx][r]) for r in rows]
        kv_len = lens[0]
        if any(x != kv_len for x in lens[1:]):
            raise RuntimeError("H2O final-prefill dense batch requires uniform physical lengths; "
                               f"layer={layer_idx} lengths={lens}.")
        if kv_len <= budget:
            raise RuntimeError("H2O final-prefill dense compaction requires an over-budget row: "
                               f"layer={layer_idx} kv_len={kv_len} budget={budget}.")
        free_count = (kv_len - budget) * batch
        free_ptr = int(self._num_free_slots[layer_idx])
        if free_ptr + free_count > self.num_slots:
            raise RuntimeError("H2O final-prefill dense compaction would overflow the free-slot stack: "
                               f"layer={layer_idx} ptr={free_ptr} release={free_count} capacity={self.num_slots}.")
        rows_gpu = torch.tensor(rows, dtype=torch.long, device=self.device)
        table = self.buffer_req_to_token_slots[layer_idx]
        old_slots = table[rows_gpu, :kv_len].to(torch.long)
        torch._assert_async(((keep >= 0) & (keep < kv_len)).all())
        if budget > 1:
            torch._assert_async((keep[:, 1:] > keep[:, :-1]).all())
        selected = old_slots.gather(1, keep)
        sorted_old = torch.sort(old_slots, dim=1).values
        dest = sorted_old[:, :budget].contiguous()
        released = sorted_old[:, budget:].reshape(-1)
        k_cache, v_cache = self.get_layer_kv_cache(layer_idx)
        h2o_ops.copy_slots(k_cache, v_cache, selected.reshape(-1).contiguous(), dest.reshape(-1).contiguous(),
                           self._get_final_prefill_workspace(batch, budget))
        self.free_slots_stack[layer_idx][free_ptr: free_ptr + free_count] = released.to(torch.int32)
        self._num_free_slots[layer_idx] = free_ptr + free_count
        table[rows_gpu, :budget] = dest.to(torch.int32)
        table[rows_gpu, budget:kv_len] = 0
        sc = self.h2o_score_tensor[kv_idx]
        kept_scores = sc[rows_gpu, :kv_len].gather(1, keep)
        sc[rows_gpu, :budget] = kept_scores
        sc[rows_gpu, budget:kv_len] = 0
        self.row_seq_lens[layer_idx][rows] = budget
        self._uniform_decode_metadata = False
        self._dev_state_dirty = True

    # ---- device-resident decode step: the machinery lives in SnapKVCacheManager (shared with StreamingLLM); H2O's part
    def _device_step_params(self, graph_batch_size: int):
        """-> (budget, trigger_len, recent_count, select_mode, score tensor, prefix) of the predicated burst (h2o.py:1498-1630)."""
        budget = self.h2o_decode_budget
        recent = min(max(1, int(budget * float(self.config.h2o_recent_ratio))), budget)
        return budget, budget + self.h2o_decode_eviction_interval, recent, 0, self.h2o_score_tensor, 0

    def _on_device_burst(self, seqs, n_rows: int, n_layers: int, dropped_per_row: int) -> None:
        self._h2o_counters["decode_eviction_bursts"] += n_rows
        self._h2o_counters["decode_evictions"] += n_rows * n_layers
        self._h2o_counters["dropped_tokens"] += dropped_per_row * n_rows * n_layers

    def _device_step_finish(self, seqs) -> None:
        self._h2o_active_decode_seq_ids.update(int(s.seq_id) for s in seqs)
        super()._device_step_finish(seqs)

    # ---- lifecycle
    def free_seq(self, seq_id: int):
        self._h2o_active_decode_seq_ids.discard(int(seq_id))
        super().free_seq(int(seq_id))

    def reset_after_warmup(self) -> None:
        self._dev_state_dirty = True
        self.h2o_score_tensor.zero_()
        self._h2o_active_decode_seq_ids.clear()
        for k in self._h2o_counters:
            self._h2o_counters[k] = 0

    def debug_state_summary(self) -> dict:
        return {"h2o": {"counters": dict(self._h2o_counters), "free_slots": self.free_slot_stats()}}
