"""Per-step / per-layer sparse control: score buffers and eviction triggers.

Host mirror of `SparseController` (engine/sparse_controller.py:55-2054) for the
h2o / streamingllm / vanilla decode paths (snapkv and quest hooks are added with their
kernels).  Call order (fixed by ModelRunner.run, model_runner.py:1447-1481):
    prepare_forward -> [per layer: get_decode_selection ... on_layer_attention_end] -> post_forward
"""

from __future__ import annotations

from dataclasses import dataclass

import torch

from ..kernels import h2o_ops
from ..method_registry import normalize_sparse_method
from ..utils.context import get_context
from ..utils.profiler import profiler
from .cache_manager.base import SparseSelection


@dataclass
class LayerBatchSparseState:
    """sparse_controller.py:36-53."""
    context_lens: torch.Tensor | None = None
    req_indices: torch.Tensor | None = None
    max_context_len: int | None = None
    attn_score: torch.Tensor | None = None


class SparseController:
    def __init__(self, config, cache_manager):
        self.config = config
        self.cache_manager = cache_manager
        self.sparse_method = normalize_sparse_method(config.vllm_sparse_method)
        self.num_layers = int(config.num_hidden_layers)
        self.device = cache_manager.device
        self.num_sink = int(config.num_sink_tokens)
        self.num_recent = int(config.num_recent_tokens)
        self.attn_softmax_scale = float(config.head_dim) ** -0.5          # sparse_controller.py:98
        self.snapkv_decode_score_dtype = torch.float32
        self.validate_runtime_invariants = bool(getattr(config, "validate_runtime_invariants", False))
        self.layer_batch_sparse_states = [LayerBatchSparseState() for _ in range(self.num_layers)]
        self._h2o_decode_attn_score_buffers: dict[tuple, torch.Tensor] = {}
        self._fused_h2o_accumulate = True
        self._layer_score_finished = [False] * self.num_layers

    def _is_kv_layer(self, layer_idx: int) -> bool:
        return self.cache_manager.is_full_attention_layer(layer_idx)

    def _h2o_kv_layer_indices(self) -> list[int]:
        return [l for l in range(self.num_layers) if self._is_kv_layer(l)]

    def _needs_attn_score(self, layer_idx: int, is_prefill: bool, seqs=None) -> bool:
        """sparse_controller.py:1963-2034 (h2o: every KV layer in decode, none in prefill)."""
        if self.sparse_method == "h2o":
            return (not is_prefill) and self._is_kv_layer(layer_idx)
        return False

    # ------------------------------------------------------------------ prepare
    def prepare_forward(self, seqs, is_prefill: bool):
        """sparse_controller.py:302-392."""
        for layer_idx in range(self.num_layers):
            st = self.cache_manager.get_layer_batch_states(layer_idx)
            s = self.layer_batch_sparse_states[layer_idx]
            s.context_lens, s.req_indices, s.max_context_len = st.context_lens, st.req_indices, st.max_context_len
            s.attn_score = None
        if not is_prefill and self.sparse_method == "h2o":
            self._prepare_h2o_decode_attn_score_buffer(seqs)

    def _h2o_decode_score_width(self, layer_indices) -> int:
        """sparse_controller.py:401-425."""
        max_len = max(int(self.layer_batch_sparse_states[l].max_context_len) for l in layer_indices)
        if bool(getattr(self.config, "decode_cuda_graph", False)):
            cap = getattr(self.cache_manager, "_decode_static_max_context_len", None)
            if cap is None or int(cap) < max_len:
                raise RuntimeError("H2O decode CUDA graph requires a score capacity covering the current context: "
                                   f"graph_capacity={cap} current={max_len}.")
            return int(cap)
        return int(max_len)

    def _get_h2o_decode_score_buffer(self, num_kv_layers: int, batch_size: int, width: int) -> torch.Tensor:
        """sparse_controller.py:427-461: one contiguous [layers, batch, width] f32 scratch, -1e20."""
        if min(num_kv_layers, batch_size, width) <= 0:
            raise RuntimeError("H2O decode score buffer requires positive dimensions: "
                               f"shape={(num_kv_layers, batch_size, width)}.")
        key = (num_kv_layers, batch_size, width)
        buf = self._h2o_decode_attn_score_buffers.get(key)
        if buf is None:
            buf = torch.empty(key, dtype=self.snapkv_decode_score_dtype, device=self.device)
            self._h2o_decode_attn_score_buffers[key] = buf
        h2o_ops.fill_f32(buf, -1e20)
        return buf

    def _prepare_h2o_decode_attn_score_buffer(self, seqs):
        layer_indices = self._h2o_kv_layer_indices()
        if not layer_indices:
            return
        batch = int(self.layer_batch_sparse_states[layer_indices[0]].context_lens.numel())
        width = self._h2o_decode_score_width(layer_indices)
        reduced = self._get_h2o_decode_score_buffer(len(layer_indices), batch, width)
        for kv_idx, layer_idx in enumerate(layer_indices):
            self.layer_batch_sparse_states[layer_idx].attn_score = reduced[kv_idx]

    # ------------------------------------------------------------------ per layer
    def get_decode_selection(self, layer_idx: int, q: torch.Tensor) -> SparseSelection:
        """sparse_controller.py:881-910 (kind="full": attend the whole physical row)."""
        s = self.layer_batch_sparse_states[layer_idx]
        return SparseSelection(kind="full", req_indices=s.req_indices, context_lens=s.context_lens,
                               max_context_len=s.max_context_len, attn_score=s.attn_score)

    def fused_decode_finish(self, layer_idx: int, mid_o, mid_lse, context_lens, o, block_seq) -> bool:
        """MI355X fusion hook called by the attention backend instead of `flash_decode_stage2`:
        for H2O decode, stage 2 and this layer's `on_layer_attention_end` score epilogue run as
        one launch (svk_h2o_decode_finish).  Returns False when there is nothing to fuse."""
        if self.sparse_method != "h2o" or get_context().is_prefill:
            return False
        s = self.layer_batch_sparse_states[layer_idx]
        if s.attn_score is None or s.attn_score.dim() != 2:
            return False
        cm = self.cache_manager
        cum = cm.h2o_score_tensor[cm.kv_layer_index(layer_idx)] if self._fused_h2o_accumulate else None
        h2o_ops.h2o_decode_finish(mid_o, mid_lse, context_lens, o, block_seq, s.attn_score, self.attn_softmax_scale,
                                  cum_score=cum, b_req_idx=s.req_indices)
        self._layer_score_finished[layer_idx] = True
        return True

    @torch.no_grad()
    def on_layer_attention_end(self, layer_idx: int):
        """sparse_controller.py:748-768.  For H2O decode: in place scale + softmax of the
        head-max raw scores, fused with the cumulative-score accumulation."""
        if not self._is_kv_layer(layer_idx):
            return
        ctx = get_context()
        if ctx.is_prefill or self.sparse_method != "h2o":
            return
        s = self.layer_batch_sparse_states[layer_idx]
        if s.attn_score is None:
            return
        if self._layer_score_finished[layer_idx]:      # already done inside the stage-2 launch
            self._layer_score_finished[layer_idx] = False
            return
        if s.attn_score.dim() != 2:
            raise RuntimeError("SnapKV-family decode attention must write a fused head-reduced [B, L] score tensor: "
                               f"layer={layer_idx} shape={tuple(s.attn_score.shape)}.")
        cm = self.cache_manager
        cum = cm.h2o_score_tensor[cm.kv_layer_index(layer_idx)] if self._fused_h2o_accumulate else None
        h2o_ops.h2o_decode_score_update(s.attn_score, self.attn_softmax_scale, cum_score=cum,
                                        b_req_idx=s.req_indices, b_seqlen=s.context_lens)

    # ------------------------------------------------------------------ post
    @torch.no_grad()
    def post_forward(self, seqs, is_prefill: bool):
        """sparse_controller.py:819-854."""
        if is_prefill:
            if self.sparse_method == "h2o":
                self.cache_manager.evict_after_prefill(seqs)
            elif self.sparse_method == "streamingllm":
                self._streamingllm_prefill_eviction(seqs)
            return
        if self.sparse_method == "h2o":
            self._h2o_decode_eviction(seqs)
        elif self.sparse_method == "streamingllm":
            self._streamingllm_decode_eviction(seqs)

    def _h2o_decode_eviction(self, seqs):
        """sparse_controller.py:1226-1282: scores are already accumulated (fused), evict."""
        with profiler.record("h2o_decode_eviction"):
            if not self._fused_h2o_accumulate:
                layer_indices = self._h2o_kv_layer_indices()
                normalized = torch.stack([self.layer_batch_sparse_states[l].attn_score for l in layer_indices])
                with profiler.record("h2o_decode_score_update"):
                    self.cache_manager.update_decode_attention_scores_all_layers(
                        layer_indices, seqs, normalized[:, : len(seqs)])
            with profiler.record("h2o_decode_compact_total"):
                self.cache_manager.evict_after_decode(seqs)

    # ------------------------------------------------------------------ StreamingLLM
    def _get_streamingllm_budget(self) -> int | None:
        budget = self.num_sink + self.num_recent
        return budget if budget > 0 else None

    def _streamingllm_select_indices(self, kv_len: int) -> torch.Tensor:
        assert kv_len > 0
        sink_end = min(self.num_sink, kv_len)
        recent_start = max(sink_end, kv_len - self.num_recent)
        return torch.cat([torch.arange(sink_end, device=self.device, dtype=torch.long),
                          torch.arange(recent_start, kv_len, device=self.device, dtype=torch.long)])

    def _streamingllm_evict(self, seqs, *, trigger_len: int, budget: int, only_final_prefill: bool):
        cm = self.cache_manager
        layers = [l for l in range(self.num_layers) if self._is_kv_layer(l)]
        pending: dict[tuple, list[int]] = {}
        group_by_key = {}
        for layer_idx in layers:
            by_len: dict[int, list] = {}
            for seq in seqs:
                if only_final_prefill and not seq.is_last_chunk_prefill:
                    continue
                row = cm.seq_id_to_row[layer_idx][seq.seq_id]
                kv_len = int(cm.row_seq_lens[layer_idx][row])
                if kv_len <= budget or kv_len < trigger_len:
                    continue
                by_len.setdefault(kv_len, []).append(seq)
            for kv_len, group in by_len.items():
                key = (tuple(int(s.seq_id) for s in group), int(kv_len))
                pending.setdefault(key, []).append(layer_idx)
                group_by_key[key] = group
        for (ids, kv_len), layer_indices in pending.items():
            cm.free_prefix_recent_slots_batch_layers(layer_indices, group_by_key[(ids, kv_len)], kv_len=kv_len,
                                                     num_sink_tokens=self.num_sink, num_recent_tokens=self.num_recent)

    @torch.no_grad()
    def _streamingllm_prefill_eviction(self, seqs):
        """sparse_controller.py:1470-1556: at the final chunk keep sink + recent."""
        budget = self._get_streamingllm_budget()
        if budget is None:
            return
        with profiler.record("streamingllm_prefill_eviction"):
            self._streamingllm_evict(seqs, trigger_len=0, budget=budget, only_final_prefill=True)

    @torch.no_grad()
    def _streamingllm_decode_eviction(self, seqs):
        """sparse_controller.py:1558-1653: compact when len >= 2 * (sink + recent)."""
        budget = self._get_streamingllm_budget()
        if budget is None:
            return
        with profiler.record("streamingllm_decode_eviction"):
            self._streamingllm_evict(seqs, trigger_len=int(2.0 * budget), budget=budget, only_final_prefill=False)
