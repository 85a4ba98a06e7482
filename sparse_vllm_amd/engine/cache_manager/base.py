"""Operator surface of the cache managers (mirror of engine/cache_manager/base.py).

Types: LayerBatchStates (:71-81), SparseSelection (:129-143), AttentionViewMeta (:146-155),
ExplicitKVPayload (:158-165), DecodeComputeView / PrefillComputeView (:198-211);
`CacheManager.create` factory (:329-369).  One manager per rank owns every KV / slot /
score tensor for the life of the process; views handed to attention are borrowed.
"""

from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import Any

import torch

from ...utils.context import get_context
from ...method_registry import NATIVE_SPARSE_METHODS, SUPPORTED_SPARSE_METHODS, normalize_sparse_method


@dataclass
class LayerBatchStates:
    slot_mapping: torch.Tensor | None = None
    context_lens: torch.Tensor | None = None
    max_context_len: int | None = None
    req_indices: torch.Tensor | None = None


@dataclass
class SparseSelection:
    kind: str
    req_indices: torch.Tensor
    context_lens: torch.Tensor
    max_context_len: int | None = None
    attn_score: torch.Tensor | None = None
    active_indices: torch.Tensor | None = None
    active_slots: torch.Tensor | None = None
    active_compressed_indices: torch.Tensor | None = None
    global_req_indices: torch.Tensor | None = None
    chunk_lens: torch.Tensor | None = None
    release_temp_slots: bool = False


@dataclass(frozen=True)
class AttentionViewMeta:
    active_slots: torch.Tensor
    req_indices: torch.Tensor
    context_lens: torch.Tensor
    max_context_len: int | None = None
    attn_score: torch.Tensor | None = None
    temp_slots: torch.Tensor | None = None


@dataclass(frozen=True)
class ExplicitKVPayload:
    k_cache: torch.Tensor
    v_cache: torch.Tensor
    backend: str = "dense"
    metadata: dict[str, Any] | None = None


@dataclass(frozen=True)
class DecodeComputeView:
    meta: AttentionViewMeta
    payload: ExplicitKVPayload


@dataclass(frozen=True)
class PrefillComputeView:
    meta: AttentionViewMeta
    payload: ExplicitKVPayload


class CacheManager(ABC):
    """One per rank; owns the physical slots and KV of every layer."""

    validate_runtime_invariants = False

    def __init__(self, config, parallel_context=None):
        self.config = config
        self.validate_runtime_invariants = bool(getattr(config, "validate_runtime_invariants", False))
        self.parallel_context = parallel_context
        self.tp_size = int(getattr(config, "tp_size", 1))
        self.device = torch.device(config.device)
        self.num_layers = int(config.num_hidden_layers)
        self.num_kv_layers = self.num_layers
        self.num_kv_heads = int(config.num_key_value_heads) // self.tp_size
        self.num_heads = int(config.num_attention_heads) // self.tp_size
        self.head_dim = int(config.head_dim)
        self.max_model_len = int(config.max_model_len)
        self.max_buffer_rows = int(config.max_num_seqs_in_gpu)
        self.kv_cache = None
        self._decode_static_max_context_len: int | None = None
        self.layer_batch_states = [LayerBatchStates() for _ in range(self.num_layers)]

    # layout helpers (reference: runtime_layout; Qwen2 has a KV cache on every layer)
    def kv_layer_index(self, layer_idx: int) -> int:
        layer_idx = int(layer_idx)
        if not 0 <= layer_idx < self.num_layers:
            raise ValueError(f"layer {layer_idx} has no KV cache")
        return layer_idx

    def kv_transformer_layer_indices(self) -> tuple[int, ...]:
        return tuple(range(self.num_layers))

    def is_full_attention_layer(self, layer_idx: int) -> bool:
        return 0 <= int(layer_idx) < self.num_layers

    @staticmethod
    def create(config, parallel_context=None) -> "CacheManager":
        """base.py:329-369."""
        sparse_method = normalize_sparse_method(config.vllm_sparse_method)
        if sparse_method not in SUPPORTED_SPARSE_METHODS:
            raise ValueError(f"Unsupported vllm_sparse_method={sparse_method!r}.")
        if sparse_method not in NATIVE_SPARSE_METHODS:
            raise NotImplementedError(
                f"sparse_method={sparse_method!r} is outside the MI355X hot-path build (SURVEY.md section 8); "
                f"native methods: {sorted(m or 'vanilla' for m in NATIVE_SPARSE_METHODS)}.")
        if sparse_method == "streamingllm":
            from .streamingllm import StreamingLLMCacheManager
            return StreamingLLMCacheManager(config, parallel_context)
        if sparse_method == "snapkv":
            from .snapkv import SnapKVCacheManager
            return SnapKVCacheManager(config, parallel_context)
        if sparse_method == "h2o":
            from .h2o import H2OCacheManager
            return H2OCacheManager(config, parallel_context)
        if sparse_method == "quest":
            from .quest import QuestCacheManager
            return QuestCacheManager(config, parallel_context)
        if sparse_method == "deltakv":
            from .deltakv import DeltaKVCacheManager
            return DeltaKVCacheManager(config, parallel_context)
        from .standard import StandardCacheManager
        return StandardCacheManager(config, parallel_context)

    # ---- abstract operator set (base.py:595-614, :981-983, :1228-1231, :1803-1817)
    @abstractmethod
    def allocate_kv_cache(self): ...

    @abstractmethod
    def get_layer_batch_states(self, layer_idx: int) -> LayerBatchStates: ...

    @abstractmethod
    def get_layer_kv_cache(self, layer_idx: int): ...

    @abstractmethod
    def get_layer_buffer_req_to_token_slots(self, layer_idx: int) -> torch.Tensor: ...

    @property
    @abstractmethod
    def num_free_slots(self) -> int: ...

    @abstractmethod
    def free_seq(self, seq_id: int): ...

    @abstractmethod
    def free_part_slots(self, layer_idx: int, seq, keep_indices: torch.Tensor, *, keep_indices_sorted: bool = False): ...

    @abstractmethod
    def _prepare_prefill(self, seqs): ...

    @abstractmethod
    def _prepare_decode(self, seqs): ...

    # ---- shared hooks with reference defaults
    def get_layer_store_view(self, layer_idx: int):
        return self.get_layer_kv_cache(layer_idx)

    def get_layer_compute_tensors(self, layer_idx: int):
        return self.get_layer_kv_cache(layer_idx)

    def get_decode_block_seq(self, layer_idx: int, default: int) -> int:
        return int(default)

    def save_rope_kv_if_needed(self, layer_idx: int, k: torch.Tensor, v: torch.Tensor):
        """base.py:629-694 (_store_layer_kv): scatter this step's K/V rows to their slots.  The model calls this right
        before the layer's `Attention.forward` (models/qwen2.py:126-131).  MI355X: in a decode step whose store may ride
        in the stage-1 launch (`fused_decode_store_slots`) the rows are kept back until that launch takes them
        (`take_deferred_decode_store`); whatever is still held when the next store arrives is written first."""
        if not get_context().is_prefill:
            slots = self.fused_decode_store_slots(layer_idx)
            if slots is not None:
                self.flush_deferred_decode_store()
                self._deferred_decode_store = (int(layer_idx), k, v, slots)
                return
        else:
            self.drop_deferred_decode_store()          # rows of a decode step that never finished: their slots are history
        self._store_rows_now(layer_idx, k, v, self.get_layer_batch_states(layer_idx).slot_mapping)

    def _store_rows_now(self, layer_idx: int, k: torch.Tensor, v: torch.Tensor, slot_mapping: torch.Tensor):
        from ...kernels import store_kvcache
        k_cache, v_cache = self.get_layer_store_view(layer_idx)
        store_kvcache(k, v, k_cache, v_cache, slot_mapping)

    def take_deferred_decode_store(self, layer_idx: int):
        """-> (k, v, slot_mapping) kept back by `save_rope_kv_if_needed` for this layer, or None.  The caller stores them
        (inside its stage-1 launch, or through `store_deferred_decode_rows`)."""
        held = self.__dict__.get("_deferred_decode_store")
        if held is None or held[0] != int(layer_idx):
            return None
        self._deferred_decode_store = None
        return held[1:]

    def store_deferred_decode_rows(self, layer_idx: int, new_kv) -> None:
        k, v, slots = new_kv
        self._store_rows_now(layer_idx, k, v, slots)

    def drop_deferred_decode_store(self) -> None:
        """Forget rows held back by a step that did not finish (a layer raised between `save_rope_kv_if_needed` and the
        launch that takes the rows).  Their slot ids belong to that step: storing them later could write into slots that
        have since been freed and handed to another sequence.  Called at the start of every step."""
        self._deferred_decode_store = None

    def flush_deferred_decode_store(self) -> None:
        """Rows that no attention launch took (a layer left early): store them with a launch of their own."""
        held = self.__dict__.get("_deferred_decode_store")
        if held is not None:
            self._deferred_decode_store = None
            self._store_rows_now(held[0], held[1], held[2], held[3])

    # ------------------------------------------------------------------ scheduler capacity hooks
    # base.py:1290-1393 of the reference: what the scheduler asks a cache manager before it admits a prompt or schedules a
    # prefill chunk / decode token.  Defaults: one persistent slot per token, one shared slot budget.
    def prefill_batched_tokens_margin(self) -> int:
        """base.py:1243-1245: extra headroom the scheduler leaves in max_num_batched_tokens for this manager."""
        return 0

    def remaining_prefill_tokens(self, seq) -> int:
        """base.py:1247-1253."""
        virtual_prefilled = max(int(seq.num_prefilled_tokens), int(getattr(seq, "prefix_cache_hit_len", 0) or 0))
        return int(seq.num_prompt_tokens - virtual_prefilled)

    def min_final_prefill_chunk_size(self, seq) -> int:
        """base.py:1326-1329."""
        return 0

    def prefill_execution_mode(self, seq) -> str:
        """base.py:1255-1258: every method of this build prefills in chunks (full-prompt staging and RawKV offload belong
        to PyramidKV / DeltaKV staging, outside SURVEY section 8)."""
        return "chunked"

    def prefill_batch_compatibility_key(self, seq) -> object:
        """base.py:1285-1288."""
        return None

    def should_schedule_full_prefill(self, seq) -> bool:
        """base.py:1309-1311."""
        return False

    def requires_full_prefill_step(self, seq) -> bool:
        """base.py:1313-1315."""
        return False

    def requires_long_prefill_offload(self, seq) -> bool:
        """base.py:1338-1340."""
        return False

    def reset_prefill_execution_state(self, seq_id: int) -> None:
        """base.py:1277-1280 (nothing sticky without the RawKV offload mode)."""
        return None

    def complete_prefill_execution(self, seq) -> None:
        """base.py:1282-1283."""
        self.reset_prefill_execution_state(int(seq.seq_id))

    def on_prompt_admitted(self, seq, costs: dict) -> None:
        """base.py:1395-1397."""
        return None

    def refresh_prefix_cache_hit(self, seq) -> None:
        """base.py:1399: no prefix cache in this build (SURVEY section 2 marks it out of scope)."""
        return None

    def clear_prefix_cache_hit(self, seq) -> None:
        return None

    def free_slot_stats(self) -> dict:
        """base.py:1453-1455."""
        return {"free_slots": int(self.num_free_slots)}

    def reserved_prefill_slots(self, waiting_seqs, chunk_prefill_size: int) -> int:
        """Slots still owed to prompts that are part-way through their prefill."""
        total = 0
        for seq in waiting_seqs:
            done, prompt = int(seq.num_prefilled_tokens), int(seq.num_prompt_tokens)
            if 0 < done < prompt:
                total += prompt - done
        return total

    def prefill_step_free_slots(self) -> int:
        return int(self.num_free_slots)

    def prefill_step_free_slots_for(self, seq) -> int:
        return int(self.prefill_step_free_slots())

    def prefill_step_reservation_cost(self, seq, scheduled_tokens: int) -> int:
        return int(scheduled_tokens)

    def decode_step_free_slots(self) -> int:
        return int(self.num_free_slots)

    def decode_step_free_slots_for(self, seq) -> int:
        return int(self.decode_step_free_slots())

    def decode_step_reservation_cost(self, seq) -> int:
        return 1

    def prompt_admission_free_slots(self) -> int:
        return int(self.num_free_slots)

    def prompt_admission_cost(self, seq) -> int:
        return int(seq.num_prompt_tokens) - int(getattr(seq, "prefix_cache_hit_len", 0) or 0)

    def prompt_logical_reservation_cost(self, seq) -> int:
        return int(self.prompt_admission_cost(seq))

    def prompt_admission_failure_action(self) -> str:
        return "defer"

    def prompt_admission_budgets(self, waiting_seqs, chunk_prefill_size: int) -> dict:
        headroom = int(self.prompt_admission_free_slots()) - int(self.reserved_prefill_slots(waiting_seqs, chunk_prefill_size))
        return {"slots": max(0, headroom)}

    def prompt_admission_costs(self, seq) -> dict:
        return {"slots": int(self.prompt_admission_cost(seq))}

    def fused_decode_store_slots(self, layer_idx: int):
        """MI355X: slot_mapping of this decode step when the layer's K/V store may ride in the attention launch
        (plain slot-table managers whose store has no side effects), else None -> `save_rope_kv_if_needed`."""
        import os
        if os.environ.get("SVK_FUSE_DECODE_STORE", "1") != "1":
            return None
        if type(self).save_rope_kv_if_needed is not CacheManager.save_rope_kv_if_needed:
            return None          # Quest page metadata, DeltaKV raw/KIVI stores: keep the explicit store
        return self.get_layer_batch_states(layer_idx).slot_mapping

    # ------------------------------------------------------------------ per-layer compute views (base.py:696-734, :838-965)
    def get_layer_compute_view(self, layer_idx: int, active_slots: torch.Tensor, req_indices: torch.Tensor,
                               context_lens: torch.Tensor, selection: SparseSelection | None = None):
        """base.py:696-709 -> (k_cache, v_cache, active_slots, req_indices, context_lens)."""
        k_cache, v_cache = self.get_layer_compute_tensors(layer_idx)
        return k_cache, v_cache, active_slots, req_indices, context_lens

    def get_layer_compute_payload(self, layer_idx: int, active_slots: torch.Tensor, req_indices: torch.Tensor,
                                  context_lens: torch.Tensor, selection: SparseSelection | None = None):
        """base.py:711-734: the decode payload + logical coordinates."""
        k_cache, v_cache, active_slots, req_indices, context_lens = self.get_layer_compute_view(
            layer_idx, active_slots, req_indices, context_lens, selection)
        return ExplicitKVPayload(k_cache=k_cache, v_cache=v_cache), active_slots, req_indices, context_lens

    def has_prefill_staging_view(self, layer_idx: int) -> bool:
        """base.py:1124-1126: whether this prefill layer reads a temporary staging KV view (none in this build)."""
        return False

    def get_prefill_staging_view(self, layer_idx: int):
        """base.py:1128-1133 -> (active_slots, req_indices, context_lens, temp_slots)."""
        raise NotImplementedError

    def has_full_layer_quantized_view(self, layer_idx: int) -> bool:
        """base.py:1135-1137."""
        return False

    def build_full_layer_quantized_view(self, layer_idx: int, req_indices: torch.Tensor, context_lens: torch.Tensor):
        """base.py:1139-1146 -> (active_slots, local_req_indices, context_lens)."""
        raise NotImplementedError

    def _default_active_slots_for_selection(self, layer_idx: int, selection: SparseSelection) -> torch.Tensor:
        """base.py:887-890."""
        if selection.active_slots is not None:
            return selection.active_slots
        return self.get_layer_buffer_req_to_token_slots(layer_idx)

    def get_prefill_compute_view(self, layer_idx: int, k_current: torch.Tensor, v_current: torch.Tensor,
                                 selection: SparseSelection, active_slots: torch.Tensor, req_indices: torch.Tensor,
                                 context_lens: torch.Tensor):
        """base.py:838-856: KV tensors + logical view of the prompt-side attention (the chunk's own K/V are in the cache)."""
        del k_current, v_current
        return self.get_layer_compute_view(layer_idx, active_slots, req_indices, context_lens, selection)

    def get_prefill_compute_payload(self, layer_idx: int, k_current: torch.Tensor, v_current: torch.Tensor,
                                    selection: SparseSelection, active_slots: torch.Tensor, req_indices: torch.Tensor,
                                    context_lens: torch.Tensor):
        """base.py:858-885."""
        k_cache, v_cache, active_slots, req_indices, context_lens = self.get_prefill_compute_view(
            layer_idx, k_current, v_current, selection, active_slots, req_indices, context_lens)
        return ExplicitKVPayload(k_cache=k_cache, v_cache=v_cache), active_slots, req_indices, context_lens

    def build_prefill_compute_view(self, layer_idx: int, k_current: torch.Tensor, v_current: torch.Tensor,
                                   selection: SparseSelection) -> PrefillComputeView:
        """base.py:892-931: staging view, quantised full-layer view or the plain slot table of the selection."""
        temp_slots = None
        if self.has_prefill_staging_view(layer_idx):
            active_slots, req_indices, context_lens, temp_slots = self.get_prefill_staging_view(layer_idx)
        elif self.has_full_layer_quantized_view(layer_idx):
            active_slots, req_indices, context_lens = self.build_full_layer_quantized_view(
                layer_idx, selection.req_indices, selection.context_lens)
        else:
            active_slots = self._default_active_slots_for_selection(layer_idx, selection)
            req_indices = selection.req_indices
            context_lens = selection.context_lens
        payload, active_slots, req_indices, context_lens = self.get_prefill_compute_payload(
            layer_idx, k_current, v_current, selection, active_slots, req_indices, context_lens)
        return PrefillComputeView(
            meta=AttentionViewMeta(active_slots=active_slots, req_indices=req_indices, context_lens=context_lens,
                                   attn_score=selection.attn_score, max_context_len=selection.max_context_len,
                                   temp_slots=temp_slots),
            payload=payload)

    def collect_prefill_attention_score(self, layer_idx: int, q: torch.Tensor, view: PrefillComputeView, *,
                                        b_start_loc: torch.Tensor, chunk_lens: torch.Tensor):
        """base.py:933-944: optional method-owned prefill score collection after the attention output is computed."""
        del layer_idx, q, view, b_start_loc, chunk_lens
        return None

    def record_prefill_query(self, layer_idx: int, q: torch.Tensor, view: PrefillComputeView, *,
                             b_start_loc: torch.Tensor, chunk_lens: torch.Tensor):
        """base.py:946-957: optional method-owned prefill query cache update."""
        del layer_idx, q, view, b_start_loc, chunk_lens
        return None

    def before_prefill_layer_attention(self, layer_idx: int, selection: SparseSelection):
        """base.py:959-965: hook right before a prefill layer's compute view is built (the reference forwards it to
        its prefix-cache coordinator, which this build does not have)."""
        del layer_idx, selection
        return None

    def defer_prefill_eviction(self) -> bool:
        """base.py:967-969."""
        return False

    def pop_prefill_attention_score(self, layer_idx: int, seq):
        """base.py:976-979."""
        del layer_idx, seq
        return None

    def build_decode_compute_view(self, layer_idx: int, q: torch.Tensor, selection: SparseSelection, *,
                                  num_heads: int, num_kv_heads: int) -> DecodeComputeView:
        """base.py:1162-1206: full physical row of every request."""
        k_cache, v_cache = self.get_layer_compute_tensors(layer_idx)
        meta = AttentionViewMeta(
            active_slots=self.get_layer_buffer_req_to_token_slots(layer_idx),
            req_indices=selection.req_indices, context_lens=selection.context_lens,
            max_context_len=selection.max_context_len, attn_score=selection.attn_score)
        return DecodeComputeView(meta=meta, payload=ExplicitKVPayload(k_cache=k_cache, v_cache=v_cache))

    def record_decode_query(self, layer_idx: int, q: torch.Tensor):
        return None

    def on_layer_attention_end(self, layer_idx: int):
        return None

    def release_layer_temp_slots(self, layer_idx: int, temp_slots):
        return None

    def on_forward_end(self, seqs, is_prefill: bool):
        return None
