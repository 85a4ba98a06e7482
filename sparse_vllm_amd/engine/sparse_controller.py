"""Per-step / per-layer sparse control: score buffers and eviction triggers.

Host mirror of `SparseController` (engine/sparse_controller.py:55-2054) for the
h2o / streamingllm / vanilla decode paths (snapkv and quest hooks are added with their
kernels).  Call order (fixed by ModelRunner.run, model_runner.py:1447-1481):
    prepare_forward -> [per layer: get_decode_selection ... on_layer_attention_end] -> post_forward
"""

from __future__ import annotations

import os
from dataclasses import dataclass

import torch

from ..kernels import deltakv_kernels, h2o_ops
from ..method_registry import normalize_sparse_method
from ..utils.context import get_context
from ..utils.profiler import profiler
from .cache_manager.base import SparseSelection


@dataclass
class LayerBatchSparseState:
    """sparse_controller.py:36-53, field for field."""
    attn_score: torch.Tensor | None = None
    active_indices: torch.Tensor | None = None        # logical indices [B, K]
    active_slots: torch.Tensor | None = None          # physical slots [B, K]
    req_indices: torch.Tensor | None = None
    context_lens: torch.Tensor | None = None
    max_context_len: int | None = None
    active_compressed_indices: torch.Tensor | None = None
    global_req_indices: torch.Tensor | None = None
    deltakv_free_temp_slots: bool = False


def _env_bool(name: str, default: bool) -> bool:
    """sparse_controller.py:18-33."""
    value = os.environ.get(name)
    if value is None:
        return bool(default)
    value = value.strip().lower()
    if value in ("1", "true", "yes", "on"):
        return True
    if value in ("0", "false", "no", "off"):
        return False
    raise ValueError(f"{name} must be one of 1/0, true/false, yes/no, or on/off; got {value!r}.")


class SparseController:
    def __init__(self, config, cache_manager):
        self.config = config
        self.cache_manager = cache_manager
        self.sparse_method = normalize_sparse_method(config.vllm_sparse_method)
        self.num_layers = int(config.num_hidden_layers)
        self.device = cache_manager.device
        self.num_sink = int(config.num_sink_tokens)
        self.num_recent = int(config.num_recent_tokens)
        self.decode_keep_tokens = int(config.decode_keep_tokens)
        self.attn_softmax_scale = float(config.head_dim) ** -0.5          # sparse_controller.py:98
        self._snapkv_decode_reduced_attn_score_buffers: dict[int, torch.Tensor] = {}
        self.snapkv_decode_score_dtype = torch.float32
        self.validate_runtime_invariants = bool(getattr(config, "validate_runtime_invariants", False))
        self.layer_batch_sparse_states = [LayerBatchSparseState() for _ in range(self.num_layers)]
        self._h2o_decode_attn_score_buffers: dict[tuple, torch.Tensor] = {}
        self._fused_h2o_accumulate = True
        # The H2O score epilogue of a layer (scale + softmax of its raw score row, cumulative add) has no consumer before
        # the step's eviction check, so `on_layer_attention_end` only queues it and ALL layers of the step run as ONE
        # launch after the layer loop (`flush_pending_scores`), like the reference's
        # update_decode_attention_scores_all_layers (h2o.py:957-1038): 28 latency-bound row launches become one
        # bandwidth-bound launch.  (Measured against the per-layer forms - inside a fused stage-2 launch, or riding in the
        # next layer's stage-1 launch - it wins at every batch size)
        self._pending_scores: list = []      # (SvkH2oDecodeScoreArgs, keep-alive tensors) of this step's layers
        # MI355X: the H2O raw-score scratch [L, B, W] is not pre-filled with -1e20 every step (the reference's
        # `view.fill_(-1e20)`, sparse_controller.py:427-461): stage 1 STORES the head-max score of every position below a
        # row's length (one owner thread per position, `score_overwrite`) and the score epilogue treats the positions at or
        # beyond the length as -1e20 (`mask_by_len`) - bit-identical results, one launch and B x W x 4 bytes less per
        # layer and step.  SVK_H2O_SCORE_PREFILL=1 restores the reference's fill + max-combine.
        self.decode_scores_overwrite = (self.sparse_method == "h2o" and os.environ.get("SVK_H2O_SCORE_PREFILL", "0") != "1")
        self.is_deltakv_family = self.sparse_method == "deltakv"
        # sparse_controller.py:70-73
        self.dynamic_deltakv_topk_tiebreak = _env_bool("SPARSEVLLM_DELTAKV_DETERMINISTIC_TOPK_TIEBREAK", False)
        self.obs_layer_ids = list(getattr(config, "obs_layer_ids", None) or [])
        self.full_attn_layers = list(getattr(config, "full_attn_layers", None) or [])
        self._decode_attn_score_buffers: dict[int, torch.Tensor] = {}

    def _is_kv_layer(self, layer_idx: int) -> bool:
        return self.cache_manager.is_full_attention_layer(layer_idx)

    def _h2o_kv_layer_indices(self) -> list[int]:
        return [l for l in range(self.num_layers) if self._is_kv_layer(l)]

    def _needs_attn_score(self, layer_idx: int, is_prefill: bool, seqs=None) -> bool:
        """sparse_controller.py:1963-2034 (h2o: every KV layer in decode, none in prefill)."""
        if self.sparse_method == "h2o":
            return (not is_prefill) and self._is_kv_layer(layer_idx)
        if self.is_deltakv_family and layer_idx in self.obs_layer_ids:
            return not is_prefill
        if self.sparse_method == "snapkv":
            if is_prefill:
                return False
            budget = self._get_layer_budget(layer_idx, is_prefill=False)
            if budget is None:
                return False
            trigger_len = self._snapkv_decode_trigger_len(budget)
            if bool(getattr(self.config, "decode_cuda_graph", False)) and getattr(self.cache_manager, "_device_step", None) is None:
                # host-driven steps under graph replay (:1977-1996): a replayed graph cannot start to collect scores on the
                # step that evicts, so every graph family whose capacity reaches the trigger collects them on every step
                # (the device-resident step does the same for another reason: prepare_forward)
                state = self.layer_batch_sparse_states[layer_idx]
                cap = getattr(self.cache_manager, "_decode_static_max_context_len", None)
                cur = state.max_context_len
                if cap is None or (cur is not None and int(cap) < int(cur)):
                    raise RuntimeError("SnapKV decode CUDA graph requires a score capacity covering the "
                                       f"current context: graph_capacity={cap} current={cur}.")
                # (the reference also clamps a short-text family's capacity to the budget, i.e. collects nothing there;
                #  collecting is a superset that changes no result, and synthetic drivers do not classify their batches)
                return int(cap) >= trigger_len and int(cap) > int(budget)
            # eager decode only collects scores on the step that is about to evict (:2003-2007)
            kv_lens = self.cache_manager.decode_kv_lens_for_layer(layer_idx, seqs)
            return any(int(n) >= trigger_len and int(n) > budget for n in kv_lens)
        return False

    def _get_layer_budget(self, layer_idx: int, is_prefill: bool) -> int | None:
        """sparse_controller.py:2036-2048 (snapkv branch)."""
        if self.cache_manager.kv_layer_index(layer_idx) < int(getattr(self.config, "snapkv_num_full_layers", 0) or 0):
            return None
        if self.sparse_method == "snapkv":
            return self.num_sink + self.decode_keep_tokens + self.num_recent
        return None

    def _snapkv_decode_trigger_len(self, budget: int) -> int:
        """sparse_controller.py:2050-2054."""
        return int(2.0 * (int(budget) - self.num_sink - self.num_recent))

    def _get_snapkv_decode_score_buffer(self, layer_idx: int, batch_size: int, max_len: int, *, fill_value: float):
        """sparse_controller.py:691-721."""
        buf = self._snapkv_decode_reduced_attn_score_buffers.get(int(layer_idx))
        if buf is None or buf.shape[0] < batch_size or buf.shape[1] < max_len:
            buf = torch.empty((batch_size, max_len), dtype=torch.float32, device=self.device)
            self._snapkv_decode_reduced_attn_score_buffers[int(layer_idx)] = buf
        view = buf[:batch_size, :max_len]
        view.fill_(fill_value)
        return view

    def _snapkv_select_indices_batch(self, scores: torch.Tensor, kv_len: int, budget: int, *, pool_kernel_size: int = 1):
        """sparse_controller.py:1707-1747 on device: sink ++ topk(middle) ++ recent (ascending; the
        reference's order is unspecified and free_part_slots sorts)."""
        if scores.dim() != 2:
            raise ValueError(f"Expected batched SnapKV scores with shape [B, L], got {tuple(scores.shape)}.")
        assert kv_len > budget
        if int(scores.shape[1]) < int(kv_len):
            raise ValueError(f"SnapKV batched scores are shorter than kv_len: scores={tuple(scores.shape)} kv_len={kv_len}.")
        recent_start = kv_len - self.num_recent
        num_topk = budget - self.num_sink - self.num_recent
        if not (num_topk > 0 and recent_start > self.num_sink):
            num_topk = 0
        if int(pool_kernel_size) > 1 and num_topk > 0:
            mid = torch.nn.functional.max_pool1d(scores[:, None, self.num_sink:recent_start], kernel_size=int(pool_kernel_size),
                                                 padding=int(pool_kernel_size) // 2, stride=1).squeeze(1)
            scores = torch.cat((scores[:, : self.num_sink], mid[:, : recent_start - self.num_sink], scores[:, recent_start:kv_len]), 1)
        scores = scores.float().contiguous() if scores.stride(1) != 1 or scores.dtype != torch.float32 else scores
        return h2o_ops.select_prefix_topk_suffix(scores, kv_len=kv_len, prefix=self.num_sink,
                                                 topk=min(num_topk, max(0, recent_start - self.num_sink)),
                                                 suffix=self.num_recent)

    def _snapkv_select_indices(self, scores: torch.Tensor, kv_len: int, budget: int, *, pool_kernel_size: int = 1):
        return self._snapkv_select_indices_batch(scores[None, :kv_len], kv_len, budget, pool_kernel_size=pool_kernel_size)[0]

    # ------------------------------------------------------------------ prepare
    def prepare_forward(self, seqs, is_prefill: bool):
        """sparse_controller.py:302-392."""
        for layer_idx in range(self.num_layers):
            st = self.cache_manager.get_layer_batch_states(layer_idx)
            s = self.layer_batch_sparse_states[layer_idx]
            s.context_lens, s.req_indices, s.max_context_len = st.context_lens, st.req_indices, st.max_context_len
            s.attn_score = None
            s.active_compressed_indices = None
            s.global_req_indices = st.req_indices
            s.deltakv_free_temp_slots = False
        if not is_prefill and self.is_deltakv_family:
            for layer_idx in self.obs_layer_ids:
                s = self.layer_batch_sparse_states[layer_idx]
                s.attn_score = self._get_decode_attn_score_buffer(
                    layer_idx, int(s.context_lens.numel()), self.cache_manager.num_heads, int(s.max_context_len),
                    fill_value=-1e20)
        if not is_prefill and self.sparse_method == "h2o":
            self._prepare_h2o_decode_attn_score_buffer(seqs)
        if not is_prefill and self.sparse_method == "snapkv" and getattr(self.cache_manager, "_device_step", None) is not None:
            # device-resident step: the re-eviction is decided on the device, so every layer collects the
            # head-max scores every step (what the reference does under its CUDA graphs) into the manager's [L, lanes, W]
            # scratch, one fill for all layers
            buf = self.cache_manager.snapkv_decode_score_tensor
            h2o_ops.fill_f32(buf, -1e20)
            for layer_idx in range(self.num_layers):
                s = self.layer_batch_sparse_states[layer_idx]
                s.attn_score = buf[self.cache_manager.kv_layer_index(layer_idx), : int(s.context_lens.numel())]
        elif not is_prefill and self.sparse_method == "snapkv":
            for layer_idx in range(self.num_layers):
                if self._needs_attn_score(layer_idx, False, seqs):
                    s = self.layer_batch_sparse_states[layer_idx]
                    s.attn_score = self._get_snapkv_decode_score_buffer(
                        layer_idx, int(s.context_lens.numel()), int(s.max_context_len), fill_value=-1e20)

    def _get_decode_attn_score_buffer(self, layer_idx: int, batch_size: int, num_heads: int, max_len: int, *,
                                      fill_value: float) -> torch.Tensor:
        """sparse_controller.py:656-689: stable per-layer [B, H, L] raw-logit buffer, refilled every step."""
        if batch_size <= 0 or num_heads <= 0 or max_len <= 0:
            raise RuntimeError("Decode attention score buffer requires positive shape: "
                               f"layer={layer_idx} batch={batch_size} heads={num_heads} max_len={max_len}.")
        buf = self._decode_attn_score_buffers.get(int(layer_idx))
        fresh = buf is None or buf.shape[0] < batch_size or buf.shape[1] < num_heads or buf.shape[2] < max_len
        if fresh:
            buf = torch.empty((batch_size, num_heads, max_len), dtype=torch.float32, device=self.device)
            self._decode_attn_score_buffers[int(layer_idx)] = buf
        view = buf[:batch_size, :num_heads, :max_len]
        # MI355X: the per-step refill is 29 MB per observation layer at 256 k tokens.  The attention launch overwrites
        # every MAPPED position below a row's length and `_decode_softmax_token_scores` masks by the candidate lengths
        # (never past the length), so a manager that guarantees every position below the length is mapped (raw slot or
        # quantised block: `decode_scores_cover_rows`, DeltaKVCacheManager's full-layer invariant) only needs a new buffer
        # filled; for any other manager an unmapped position would keep a previous step's value, so the reference's
        # per-step refill stays.
        if fresh or not bool(getattr(self.cache_manager, "decode_scores_cover_rows", False)):
            h2o_ops.fill_f32(view, fill_value) if view.is_contiguous() else view.fill_(fill_value)
        return view

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
        if not self.decode_scores_overwrite:
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
    def _build_selection(self, layer_idx: int, *, is_prefill: bool, q: torch.Tensor | None = None) -> SparseSelection:
        """sparse_controller.py:881-935: the logical selection only; cache managers build the physical views.
        kind="full" (attend the whole physical row) for every method of this build except DeltaKV's sparse layers."""
        if not self._is_kv_layer(layer_idx):
            raise RuntimeError(f"layer_idx={layer_idx} is linear_attention and has no KV sparse selection")
        s = self.layer_batch_sparse_states[layer_idx]
        if self.is_deltakv_family and layer_idx not in self.full_attn_layers:
            # :912-935: batch-major view, K selected compressed positions (-1 padded / None = 0)
            chunk_lens = None
            if is_prefill:
                cu = get_context().cu_seqlens_q
                if cu is not None and cu.numel() > 1:
                    chunk_lens = (cu[1:] - cu[:-1]).to(torch.int32)
            return SparseSelection(kind="deltakv", req_indices=s.global_req_indices, context_lens=s.context_lens,
                                   max_context_len=s.max_context_len, attn_score=s.attn_score,
                                   active_compressed_indices=s.active_compressed_indices,
                                   global_req_indices=s.global_req_indices, chunk_lens=chunk_lens,
                                   release_temp_slots=s.deltakv_free_temp_slots)
        req = s.global_req_indices if self.is_deltakv_family else s.req_indices
        return SparseSelection(kind="full", req_indices=req, context_lens=s.context_lens,
                               max_context_len=s.max_context_len, attn_score=s.attn_score,
                               global_req_indices=s.global_req_indices)

    def get_prefill_selection(self, layer_idx: int) -> SparseSelection:
        """sparse_controller.py:957-958."""
        return self._build_selection(layer_idx, is_prefill=True)

    def get_decode_selection(self, layer_idx: int, q: torch.Tensor, active_slots=None, req_indices=None,
                             context_lens=None) -> SparseSelection:
        """sparse_controller.py:960-969."""
        del active_slots, req_indices, context_lens
        return self._build_selection(layer_idx, is_prefill=False, q=q)

    def _h2o_new_slots(self, layer_idx: int):
        """This step's slot_mapping of the layer: -1 marks the padded lanes of a graph-sized batch, whose scores the
        reference drops (`normalized[:, :len(seqs)]`, sparse_controller.py:1226-1282)."""
        sm = self.cache_manager.get_layer_batch_states(layer_idx).slot_mapping
        s = self.layer_batch_sparse_states[layer_idx]
        if sm is None or sm.dtype != torch.int32 or s.attn_score is None or sm.numel() != s.attn_score.shape[0]:
            return None
        return sm

    def flush_pending_scores(self):
        """Run the queued score epilogues of this step's layers as one launch (end of the layer loop - inside the captured
        graph - and before anything reads the scores)."""
        if self._pending_scores:
            batch, self._pending_scores = self._pending_scores, []
            h2o_ops.h2o_decode_score_update_layers([e[0] for e in batch])

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
        if s.attn_score.dim() != 2:
            raise RuntimeError("SnapKV-family decode attention must write a fused head-reduced [B, L] score tensor: "
                               f"layer={layer_idx} shape={tuple(s.attn_score.shape)}.")
        cm = self.cache_manager
        cum = cm.h2o_score_tensor[cm.kv_layer_index(layer_idx)] if self._fused_h2o_accumulate else None
        from ..kernels.gqa_flash_decoding_stage1 import h2o_score_args
        new_slots = self._h2o_new_slots(layer_idx)
        self._pending_scores.append(
            (h2o_score_args(s.attn_score, self.attn_softmax_scale, cum_score=cum, b_req_idx=s.req_indices,
                            b_seqlen=s.context_lens, b_new_slot=new_slots, mask_by_len=self.decode_scores_overwrite),
             (s.attn_score, cum, s.req_indices, s.context_lens, new_slots)))

    def join_side_streams(self):
        """End of the layer loop (inside graph capture when the step is captured): issue the step's score epilogue."""
        self.flush_pending_scores()

    # ------------------------------------------------------------------ DeltaKV query-aware top-k
    @torch.no_grad()
    def on_layer_end(self, layer_idx: int, context=None):
        """sparse_controller.py:971-1052 (DeltaKV decode branch): observation-layer raw logits -> per-head softmax over
        the compressed range -> max over heads (bf16-rounded like the model dtype cast) -> sorted top-k, shared by
        the sparse layers up to the next full-attention layer."""
        context = context or get_context()
        if not self._is_kv_layer(layer_idx) or not self.is_deltakv_family or context.is_prefill:
            return
        if layer_idx not in self.obs_layer_ids:
            return
        with profiler.record("sparse_on_layer_end"):
            state = self.layer_batch_sparse_states[layer_idx]
            if state.attn_score is None:
                raise ValueError("Attn Score hasn't been initialized")
            if state.attn_score.dim() == 3:
                compressed_lens = self.cache_manager.get_compressed_lens(state.req_indices)
                state.attn_score = self._decode_softmax_token_scores(state.attn_score, candidate_start=self.num_sink,
                                                                     candidate_lens=compressed_lens)
            target_layers = []
            for j in range(layer_idx + 1, self.num_layers):
                if j in self.full_attn_layers:
                    break
                target_layers.append(j)
            if not target_layers:
                raise RuntimeError("Dynamic sparse observation layer has no target KV layers: "
                                   f"method={self.sparse_method} observation_layer={layer_idx} "
                                   f"full_attn_layers={self.full_attn_layers}.")
            self._update_dynamic_omnikv_indices(layer_idx, target_layers)

    def _decode_softmax_token_scores(self, scores: torch.Tensor, *, candidate_start: int, candidate_lens: torch.Tensor):
        """sparse_controller.py:255-299 as one fused pass (svk_deltakv_token_scores); values are bf16-representable
        floats (the reference casts to the model dtype), masked entries = finfo(bf16).min."""
        return deltakv_kernels.decode_softmax_token_scores(scores, candidate_start=candidate_start,
                                                           candidate_lens=candidate_lens, scale=self.attn_softmax_scale,
                                                           round_dtype=torch.bfloat16)

    def _update_dynamic_omnikv_indices(self, obs_layer_idx: int, target_layers):
        """sparse_controller.py:1755-1822, :1951-1959 (DeltaKV decode): mask beyond the compressed length with -1e10,
        `topk(k_max, sorted=True)`.  torch leaves the order among equal scores unspecified (the reference notes that
        graph replay and eager disagree, :1803-1806); here equal scores rank by ascending position.  With
        SPARSEVLLM_DELTAKV_DETERMINISTIC_TOPK_TIEBREAK=1 the reference's position key is added first, op by op in fp32
        (:1797-1811: `s + max(|s|, 1) * (pos / n * 1e-6)` — later positions win, and scores below ~1e-4 are
        re-ranked by it), so that mode is bit-identical up to keys that still tie after the fp32 add."""
        obs = self.layer_batch_sparse_states[obs_layer_idx]
        token_scores = obs.attn_score
        search_scores = token_scores[:, self.num_sink:]
        rel_hist_lens = self.cache_manager.get_compressed_lens(obs.req_indices)
        k_max = min(int(self.decode_keep_tokens), int(search_scores.size(1)))
        if self.dynamic_deltakv_topk_tiebreak and search_scores.numel() > 0:
            n = int(search_scores.size(1))
            pos_key = torch.arange(n, device=search_scores.device, dtype=torch.float32) / max(1, n)
            masked = torch.arange(n, device=search_scores.device) >= rel_hist_lens.to(search_scores.device).unsqueeze(1)
            base = search_scores.float().masked_fill(masked, -1e10)
            search_scores = base + base.abs().clamp_min(1.0) * (pos_key.unsqueeze(0) * 1.0e-6)
        if k_max > 0:
            topk_indices = deltakv_kernels.topk_sorted_desc(search_scores, k_max, valid_len=rel_hist_lens,
                                                            masked_value=-1e10)
        else:
            topk_indices = torch.empty((token_scores.shape[0], 0), device=self.device, dtype=torch.int32)
        for l_idx in target_layers:
            t = self.layer_batch_sparse_states[l_idx]
            t.active_compressed_indices = topk_indices
            t.context_lens = obs.context_lens
            t.req_indices = obs.req_indices
            t.global_req_indices = obs.req_indices
            t.deltakv_free_temp_slots = (l_idx == target_layers[-1])

    # ------------------------------------------------------------------ post
    @torch.no_grad()
    def post_forward(self, seqs, is_prefill: bool):
        """sparse_controller.py:819-854."""
        if is_prefill:
            if self.sparse_method == "h2o":
                self.cache_manager.evict_after_prefill(seqs)
            elif self.sparse_method == "streamingllm":
                self._streamingllm_prefill_eviction(seqs)
            elif self.sparse_method == "snapkv":
                self._snapkv_prefill_eviction(seqs)
            elif self.is_deltakv_family:
                # sparse_controller.py:857-866 (on_every_chunk_prefill_end): compress the chunk's raw tail in bulk
                evict = getattr(self.cache_manager, "deltakv_evict", None)
                if evict is not None:
                    evict(seqs)
            return
        if self.sparse_method == "h2o":
            self._h2o_decode_eviction(seqs)
        elif self.is_deltakv_family:
            evict = getattr(self.cache_manager, "deltakv_evict", None)
            if evict is not None:
                evict(seqs)
        elif self.sparse_method == "streamingllm":
            self._streamingllm_decode_eviction(seqs)
        elif self.sparse_method == "snapkv":
            self._snapkv_decode_eviction(seqs)

    # ------------------------------------------------------------------ SnapKV
    @torch.no_grad()
    def _snapkv_prefill_eviction(self, seqs):
        """sparse_controller.py:1059-1102: at the final chunk keep sink ++ topk ++ recent."""
        cm = self.cache_manager
        for layer_idx in range(self.num_layers):
            if not self._is_kv_layer(layer_idx):
                continue
            budget = self._get_layer_budget(layer_idx, is_prefill=True)
            if budget is None:
                continue
            for seq in seqs:
                if not seq.is_last_chunk_prefill:
                    continue
                kv_len = int(seq.num_prefilled_tokens) + int(seq.current_chunk_size)
                if kv_len <= budget:
                    continue
                seq_scores = cm.pop_prefill_attention_score(layer_idx, seq)
                if seq_scores is None:
                    raise RuntimeError("SnapKV/PyramidKV prefill eviction requires prefill attention scores. "
                                       f"method={self.sparse_method} layer={layer_idx} seq_id={seq.seq_id}")
                keep = self._snapkv_select_indices(seq_scores[:kv_len], kv_len, budget,
                                                   pool_kernel_size=int(getattr(self.config, "pool_kernel_size", 1) or 1))
                cm.free_part_slots(layer_idx, seq, keep, keep_indices_sorted=True)

    @torch.no_grad()
    def _snapkv_decode_eviction(self, seqs):
        """sparse_controller.py:1104-1223: when a row reaches 2 x top budget, re-select on this step's
        head-max raw decode scores; equal-length rows are compacted across layers in one launch."""
        cm = self.cache_manager
        if getattr(cm, "_device_step", None) is not None:
            # device-resident step: the re-eviction was the predicated burst of the step's launches; the host only advances
            # its mirrors
            cm._device_step_finish(seqs)
            return
        with profiler.record("snapkv_decode_eviction"):
            pending: dict[tuple, list] = {}
            for layer_idx in range(self.num_layers):
                if not self._is_kv_layer(layer_idx):
                    continue
                scores = self.layer_batch_sparse_states[layer_idx].attn_score
                if scores is None:
                    continue
                budget = self._get_layer_budget(layer_idx, is_prefill=False)
                if budget is None:
                    continue
                trigger_len = self._snapkv_decode_trigger_len(budget)
                kv_lens = cm.decode_kv_lens_for_layer(layer_idx, seqs)
                by_len: dict[int, list] = {}
                for b, (seq, n) in enumerate(zip(seqs, kv_lens)):
                    if n <= budget or n < trigger_len:
                        continue
                    by_len.setdefault(int(n), []).append((b, seq))
                if by_len and scores.dim() != 2:
                    raise RuntimeError("SnapKV/PyramidKV post-forward eviction requires head-reduced [B, L] scores: "
                                       f"layer={layer_idx} shape={tuple(scores.shape)}.")
                for kv_len, group in by_len.items():
                    idx = torch.tensor([b for b, _ in group], dtype=torch.long, device=scores.device)
                    with profiler.record("snapkv_decode_select"):
                        keep = self._snapkv_select_indices_batch(scores.index_select(0, idx)[:, :kv_len].contiguous(),
                                                                 kv_len, budget)
                    group_seqs = [s for _, s in group]
                    if len(group) == 1:
                        # a lone row is compacted at once, batched groups after the layer loop (:1173-1182 vs
                        # :1209-1223): that order fixes the layer's free-stack contents
                        with profiler.record("snapkv_decode_compact"):
                            cm.free_part_slots_batch_layers([layer_idx], group_seqs, keep[None], keep_indices_sorted=True)
                        continue
                    key = (tuple(int(s.seq_id) for s in group_seqs), tuple(keep.shape))
                    pending.setdefault(key, []).append((layer_idx, group_seqs, keep))
            for entries in pending.values():
                layers = [e[0] for e in entries]
                with profiler.record("snapkv_decode_compact_layers"):
                    cm.free_part_slots_batch_layers(layers, entries[0][1], torch.stack([e[2] for e in entries]),
                                                    keep_indices_sorted=True)

    def _h2o_decode_eviction(self, seqs):
        """sparse_controller.py:1226-1282: scores are already accumulated (fused), evict."""
        self.flush_pending_scores()
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
        cand = [s for s in seqs if not only_final_prefill or s.is_last_chunk_prefill]
        if not cand or not layers:
            return
        # the common decode step evicts nothing: decide that with one numpy gather over all (layer, sequence) lengths
        # (at 64 sequences x 28 layers the Python double loop below costs ~0.2 ms, a quarter of the step itself)
        all_lens = getattr(cm, "decode_kv_lens_all_layers", None)
        kv = all_lens(cand) if (all_lens is not None and not only_final_prefill) else None
        if kv is not None and not ((kv > budget) & (kv >= trigger_len)).any():
            return
        pending: dict[tuple, list[int]] = {}
        group_by_key = {}
        for layer_idx in layers:
            by_len: dict[int, list] = {}
            for seq in cand:
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
        cm = self.cache_manager
        if getattr(cm, "_device_step", None) is not None:
            # device-resident step: the window eviction is the predicated burst of the step's launches (inside the graph);
            # the host only advances its mirrors
            cm._device_step_finish(seqs)
            return
        with profiler.record("streamingllm_decode_eviction"):
            self._streamingllm_evict(seqs, trigger_len=int(2.0 * budget), budget=budget, only_final_prefill=False)
