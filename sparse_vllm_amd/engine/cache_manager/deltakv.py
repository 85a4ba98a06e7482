"""DeltaKV: full-attention layers (raw bf16 or KIVI-int4 blocks) + compressed sparse layers.

Host mirror of the reference's slim DeltaKV runtime for the decode side of the hot path
(SURVEY.md section 8 a20-a25):
  engine/cache_manager/deltakv_base.py        slot pools / maps (:114-200, :1247-1483), prepare_decode_static (:2038-2154),
                                              get_compressed_lens (:2156-2165), build_decode_compute_view (:975-1018)
  engine/cache_manager/deltakv_less_memory.py allocate_kv_cache (:940-1190), _load_residual (:2841-2848),
                                              _deltakv_build_view_and_plan_reconstruct{,_static} (:2850-2934),
                                              KIVI decode view (:2942-2982), get_layer_compute_view (:1344-1402),
                                              deltakv_reconstruct (:4032-4143)
  engine/cache_manager/deltakv_less_memory_cuda_graph.py  _set_postrope_slots (:476-500), get_decode_block_seq (:502-505)

Data model (one pool per layer *kind*, shared slot ids across the layers of a kind):
  full layers    full_kv_cache [2, Lf, full_slots, Hkv, D] bf16, row map `full_layer_slots_map`; with
                 full_layer_kv_quant_bits=4 older tokens live in KIVI blocks (per-channel int4 K with fp32 scale/min,
                 per-token int4 V with bf16 scale/min) addressed by `full_layer_kivi_block_slots_map`.
  sparse layers  deltakv_full_kv_cache [2, Ls, slots, Hkv, D] holds *pre-RoPE* K (sink, cluster centres, raw tail)
                 and per-step post-RoPE reconstructions in temp slots (flagged in `_deltakv_postrope_slot_mask`);
                 every other token is a latent residual (int4 + group scale/min, or bf16) with K father slots.
Sparse-layer decode: static plan -> residual load (dequant + compress_up) -> reconstruct + RoPE write-back into
temp slots -> contiguous attention view (raw K rotated on the way) -> ordinary stage 1 / stage 2.

Compression side: `deltakv_evict` compresses `recent` tokens of a row's raw tail every `recent`
decode steps (centres, causal L2 top-k fathers, compress_down residual, int4 pack) and `_full_layer_kivi_evict` moves
full-layer rows that left the residual window into KIVI blocks.  Prompts arrive either already compressed
(`admit_compressed_row`, synthetic benchmarks) or raw through `_prepare_prefill` + the store hooks, after which the same
`deltakv_evict` / KIVI eviction compress the chunk in bulk.
"""

from __future__ import annotations

from collections import deque

import os

import numpy as np
import torch

from ...kernels import deltakv_kernels as dk
from ...utils.context import get_context
from ...utils.profiler import profiler
from .base import (AttentionViewMeta, CacheManager, DecodeComputeView, ExplicitKVPayload, LayerBatchStates,
                   SparseSelection)


def build_compressor(input_size: int, output_size: int, kind: str, intermediate_size: int, bias: bool) -> torch.nn.Module:
    """utils/compressor.py:36-86: linear | mlp_gelu | mlp_swiglu."""
    if kind == "linear":
        return torch.nn.Linear(input_size, output_size, bias=bias)
    if kind == "mlp_gelu":
        return torch.nn.Sequential(torch.nn.Linear(input_size, intermediate_size, bias=bias), torch.nn.GELU(),
                                   torch.nn.Linear(intermediate_size, output_size, bias=bias))
    if kind == "mlp_swiglu":
        class SwiGLU(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.w12 = torch.nn.Linear(input_size, intermediate_size * 2, bias=bias)
                self.w3 = torch.nn.Linear(intermediate_size, output_size, bias=bias)

            def forward(self, x):
                a, b = self.w12(x).chunk(2, dim=-1)
                return self.w3(torch.nn.functional.silu(a) * b)
        return SwiGLU()
    raise ValueError(f"Unhandled compressor type after normalization: {kind}")


_COMPRESSOR_KINDS = {"": "auto", "auto": "auto", "linear": "linear", "mlp": "mlp_gelu", "gelu": "mlp_gelu",
                     "mlp_gelu": "mlp_gelu", "swiglu": "mlp_swiglu", "mlp_swiglu": "mlp_swiglu"}


def create_compressor(is_down: bool, config) -> torch.nn.Module:
    head_dim, hkv = int(config.head_dim), int(config.num_key_value_heads)
    kv_dim = 2 * head_dim * hkv
    input_size = kv_dim if is_down else int(config.kv_compressed_size)
    output_size = int(config.kv_compressed_size) if is_down else kv_dim
    raw = getattr(config, "compressor_down_type" if is_down else "compressor_up_type", "auto")
    if raw not in _COMPRESSOR_KINDS:
        raise ValueError(f"Unknown compressor type: {raw}. Use auto|linear|mlp_gelu|mlp_swiglu.")
    kind = _COMPRESSOR_KINDS[raw]
    if kind == "auto":
        kind = "mlp_gelu" if bool(config.use_nonlinear_compressor) else "linear"
    inter = int(getattr(config, "compressor_down_intermediate_size" if is_down else "compressor_up_intermediate_size", -1))
    if inter <= 0:
        inter = int(config.compressor_intermediate_size)
    if inter <= 0:
        inter = (input_size + output_size) // 2
    return build_compressor(input_size, output_size, kind, inter, bool(config.compressor_linear_bias))


def rope_cos_sin_cache(max_pos: int, head_dim: int, theta: float, device) -> torch.Tensor:
    """[max_pos, D] f32: cos | sin halves (layers/rotary_embedding.py neox layout)."""
    inv_freq = 1.0 / (float(theta) ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    ang = torch.arange(max_pos, dtype=torch.float32)[:, None] * inv_freq[None, :]
    return torch.cat((ang.cos(), ang.sin()), dim=1).contiguous().to(device)


class _Pool:
    """LIFO free stack of slot ids, mirrored on the host (`free_slots_stack_* [ptr-n:ptr]`)."""

    def __init__(self, n: int, name: str):
        self.stack = np.arange(n, dtype=np.int32)
        self.free = int(n)
        self.size = int(n)
        self.name = name
        self.version = 0          # moves with every change (the device-resident decode step tells its own pops from others)

    def pop(self, n: int) -> np.ndarray:
        n = int(n)
        if self.free < n:
            raise RuntimeError(f"Out of {self.name} slots: need={n} free={self.free}.")
        out = self.stack[self.free - n: self.free].copy()
        self.free -= n
        self.version += 1
        return out

    def push(self, slots: np.ndarray):
        slots = np.asarray(slots, dtype=np.int32).reshape(-1)
        self.stack[self.free: self.free + slots.size] = slots
        self.free += int(slots.size)
        self.version += 1

    def permute(self, seed: int):
        assert self.free == self.size
        self.stack = np.random.default_rng(seed).permutation(self.size).astype(np.int32)
        self.version += 1


class DeltaKVCacheManager(CacheManager):
    def __init__(self, config, parallel_context=None):
        super().__init__(config, parallel_context)
        self.full_attn_layers = sorted(int(x) for x in config.full_attn_layers)
        self.full_layer_ids = [l for l in range(self.num_layers) if l in set(self.full_attn_layers)]
        self.deltakv_layer_ids = [l for l in range(self.num_layers) if l not in set(self.full_attn_layers)]
        self.full_layer_to_idx = {l: i for i, l in enumerate(self.full_layer_ids)}
        self.deltakv_layer_to_idx = {l: i for i, l in enumerate(self.deltakv_layer_ids)}
        self.full_layer_batch_states = LayerBatchStates()
        self.deltakv_layer_batch_states = LayerBatchStates()
        d = self.device
        rows, L = self.max_buffer_rows, self.max_model_len
        self.full_layer_slots_map = torch.zeros((rows, L), dtype=torch.int32, device=d)
        self.sparse_layer_raw_slots_map = torch.full((rows, L), -1, dtype=torch.int32, device=d)
        self.sparse_layer_latent_slots_map = torch.full((rows, L), -1, dtype=torch.int32, device=d)
        self.seq_id_to_row: dict[int, int] = {}
        self.free_rows = deque(range(rows))
        self.row_seq_lens = np.zeros((rows,), dtype=np.int32)
        self.row_deltakv_compressed_lens = np.zeros((rows,), dtype=np.int32)
        self.row_deltakv_compressed_lens_gpu = torch.zeros((rows,), dtype=torch.int32, device=d)
        self.cos_sin_cache = rope_cos_sin_cache(L, self.head_dim, float(config.rope_theta), d)
        self.deltakv_k_norm_weight = None            # Qwen2 has no k_norm; set_model_layers would fill it
        self.deltakv_k_norm_eps = 1e-6
        self.allocate_kv_cache()
        self._init_compressor_modules(config)
        self._deltakv_reset_view_cache()
        self._static: dict[tuple, tuple] = {}
        self._temp_slots_by_shape: dict[tuple, torch.Tensor] = {}
        self._plan_buffers: dict[tuple, tuple] = {}
        self._materialized_view: dict[tuple, tuple] = {}
        self._deltakv_decode_static_compressed_lens = None
        self._deltakv_decode_static_slot_mapping = None
        self._deltakv_decode_static_active_pos = None
        self._pending_raw_store: dict[int, tuple] = {}   # layer -> (k, v) whose store rides in the materialise launch
        self._reset_prefill_staging()
        # MI355X, SURVEY 8(f).2: device-resident decode bookkeeping (`_device_state_sync`); SVK_H2O_DEVICE_STATE=0: every
        # step uploads its rows / lengths / slots (svk_deltakv_decode_alloc)
        import os
        self._device_step_enabled = os.environ.get("SVK_H2O_DEVICE_STATE", "1") == "1"
        self.device_step_generation = 0
        self._device_step = None                 # this step's [args, launched] while the device-resident step is active
        self._dev_step_cache = None
        self._dev_state = None                   # row lengths, the two free stacks and their pointers on the device
        self._dev_stands_for = None              # (full pool version, sparse pool version, row lengths) of that copy

    # ------------------------------------------------------------------ configuration helpers
    def _reset_prefill_staging(self):
        self._deltakv_prefill_staging_active = False
        self._deltakv_prefill_staging_active_slots = None
        self._deltakv_prefill_staging_req_indices = None
        self._deltakv_prefill_staging_context_lens = None

    def _full_layer_kivi_enabled(self) -> bool:
        return bool(self.config.enable_full_layer_kivi_quant) and int(self.config.full_layer_kv_quant_bits or 0) == 4

    def _full_layer_kivi_group_size(self) -> int:
        g = int(self.config.full_layer_kivi_group_size or 32)
        if g % 8 != 0:
            raise ValueError("Full-layer KIVI int4 packing requires group_size divisible by 8; "
                             f"got full_layer_kivi_group_size={g}.")
        if self.head_dim % g != 0:
            raise ValueError("Full-layer KIVI value quantization requires head_dim divisible by group_size; "
                             f"head_dim={self.head_dim}, group_size={g}.")
        return g

    def _sparse_payload_dim(self) -> int:
        return int(self.config.kv_compressed_size)

    def _quant_group_size(self) -> int:
        g = int(self.config.kv_quant_group_size or 0)
        return g if g > 0 else self._sparse_payload_dim()

    def _deltakv_decode_static_max_buffer(self) -> int:
        """deltakv_base.py:2718-2723: decode runs before post-forward compression, so the raw tail can hold one
        recent window plus the next remainder."""
        recent = int(self.config.num_recent_tokens)
        return max(recent + 1, 2 * recent)

    # ------------------------------------------------------------------ storage
    def _prefill_staging_tokens(self) -> int:
        """Raw slots one prefill chunk needs before `deltakv_evict` / the KIVI eviction compress it (the reference
        stages a chunk the same way, deltakv_less_memory.py `_deltakv_full_prefill_*`): one chunk in flight."""
        return int(getattr(self.config, "chunk_prefill_size", 0) or 0)

    def allocate_kv_cache(self):
        """deltakv_less_memory.py:940-1190 (explicit sizes; 0 = derive a synthetic-workload default)."""
        cfg, d = self.config, self.device
        rows, L = self.max_buffer_rows, self.max_model_len
        sink, recent, keep = int(cfg.num_sink_tokens), int(cfg.num_recent_tokens), int(cfg.decode_keep_tokens)
        kivi = self._full_layer_kivi_enabled()
        G = self._full_layer_kivi_group_size() if kivi else 0
        n_sparse = int(cfg.num_kvcache_slots or 0)
        if n_sparse <= 0:
            step = max(1, int(1.0 / max(1e-6, float(cfg.cluster_ratio))))
            centers = -(-L // max(1, recent)) * -(-max(1, recent) // step)      # range(start, end, step) per evicted block
            n_sparse = rows * (sink + 2 * recent + 1 + centers + keep) + 64 + self._prefill_staging_tokens()
        n_latent = int(cfg.deltakv_num_latent_slots or 0) or rows * L
        n_full = int(cfg.deltakv_num_full_layer_slots or 0)
        if n_full <= 0:
            n_full = rows * ((sink + int(cfg.full_layer_kivi_residual_length) + 2 * G + 1) if kivi else L) + 64
            if kivi:
                n_full += self._prefill_staging_tokens()
        n_blocks = int(cfg.deltakv_num_kivi_blocks or 0) or (rows * (L // G + 1) if kivi else 0)
        self.deltakv_full_num_slots, self.deltakv_latent_num_slots = n_sparse, n_latent
        self.full_num_slots, self.full_layer_kivi_num_blocks = n_full, n_blocks
        Lf, Ls, H, D = len(self.full_layer_ids), len(self.deltakv_layer_ids), self.num_kv_heads, self.head_dim
        bf = torch.bfloat16
        self.full_kv_cache = torch.zeros((2, Lf, n_full, H, D), dtype=bf, device=d)
        self.deltakv_full_kv_cache = torch.zeros((2, Ls, n_sparse, H, D), dtype=bf, device=d)
        self.kv_cache = self.deltakv_full_kv_cache
        width = sink + keep + self._deltakv_decode_static_max_buffer()
        self.deltakv_materialized_compute_num_slots = rows * width
        self.deltakv_materialized_kv_cache = torch.zeros((2, rows * width, H, D), dtype=bf, device=d)
        # MI355X: the static-decode reconstruction writes its rows straight into the layer's attention view (include/svk.h
        # `out_k_cache`) - the view kernel then copies the ~150 raw rows only, not the 2048 reconstructed ones out of the
        # scratch slots.  Every sparse layer owns a view for that (the look-ahead reconstruction of the layers behind
        # runs while a layer attends): allocated on first use.  False = scratch slots + full copy (the reference's flow).
        self.recon_into_view = bool(self._RECON_INTO_VIEW_DEFAULT)
        self._layer_views: torch.Tensor | None = None
        self._recon_view_layers: set[int] = set()      # sparse layers (l_idx) whose reconstructed rows of this plan are in their view
        self._view_rest_layers: set[int] = set()       # ... whose raw rows (all but the step's newest) are in their view as well
        self._rotated_store_for: dict[int, dict] = {}  # layer -> the newest row's rotated store for the attention launch
        self._deltakv_postrope_slot_mask = torch.zeros((Ls, n_sparse), dtype=torch.bool, device=d)
        bits = int(cfg.kv_quant_bits or 0)
        payload = self._sparse_payload_dim()
        if bits:
            gs = self._quant_group_size()
            if payload % gs or payload % (32 // bits):
                raise ValueError(f"kv_compressed_size={payload} must be divisible by the quant group size {gs} and {32 // bits}.")
            self.deltakv_latent_cache = torch.zeros((Ls, n_latent, payload // (32 // bits)), dtype=torch.int32, device=d)
            self.deltakv_latent_scales = torch.zeros((Ls, n_latent, payload // gs), dtype=bf, device=d)
            self.deltakv_latent_mins = torch.zeros_like(self.deltakv_latent_scales)
        else:
            self.deltakv_latent_cache = torch.zeros((Ls, n_latent, payload), dtype=bf, device=d)
            self.deltakv_latent_scales = self.deltakv_latent_mins = None
        self.deltakv_latent_to_full_slots = torch.full((Ls, n_latent, int(cfg.deltakv_k_neighbors)), -1, dtype=torch.int32, device=d)
        self.deltakv_slot_to_pos = torch.full((n_sparse,), -1, dtype=torch.int32, device=d)
        self.full_layer_slot_to_pos = torch.full((n_full,), -1, dtype=torch.int32, device=d)
        self._pool_sparse = _Pool(n_sparse, "DeltaKV full cache")
        self._pool_latent = _Pool(n_latent, "DeltaKV latent")
        self._pool_full = _Pool(n_full, "full KV cache")
        if kivi:
            self.full_layer_kivi_key_packed = torch.zeros((Lf, n_blocks, H, D, G // 8), dtype=torch.int32, device=d)
            self.full_layer_kivi_key_scales = torch.zeros((Lf, n_blocks, H, D), dtype=torch.float32, device=d)
            self.full_layer_kivi_key_mins = torch.zeros_like(self.full_layer_kivi_key_scales)
            self.full_layer_kivi_value_packed = torch.zeros((Lf, n_blocks, H, G, D // 8), dtype=torch.int32, device=d)
            self.full_layer_kivi_value_scales = torch.zeros((Lf, n_blocks, H, G, D // G), dtype=bf, device=d)
            self.full_layer_kivi_value_mins = torch.zeros_like(self.full_layer_kivi_value_scales)
            self.full_layer_kivi_block_slots_map = torch.full((rows, L), -1, dtype=torch.int32, device=d)
            self.full_layer_kivi_block_start_pos = torch.full((n_blocks,), -1, dtype=torch.int32, device=d)
            self._pool_kivi = _Pool(n_blocks, "full-layer KIVI block")
            self.row_full_layer_kivi_quantized_lens = np.zeros((rows,), dtype=np.int32)
        else:
            self.full_layer_kivi_key_packed = self.full_layer_kivi_value_packed = None
            self.full_layer_kivi_block_slots_map = self.full_layer_kivi_block_start_pos = None
            self._pool_kivi = None
            self.row_full_layer_kivi_quantized_lens = None
        self.row_kivi_blocks: dict[int, list[int]] = {}
        self.row_latent_slots: dict[int, np.ndarray] = {}
        # sink ++ every centre slot chosen so far, per row (row_deltakv_center_slots, identical for all sparse layers
        # because slot ids are shared between them; deltakv_less_memory.py:3660-3667, :3750-3753)
        self.row_deltakv_center_slots: dict[int, torch.Tensor] = {}

    def _init_compressor_modules(self, config):
        """deltakv_less_memory.py:1190-1199; weights come from `deltakv_path` in the reference (external
        checkpoint, absent here) -> seeded random init, overridable through `load_compressor_state`."""
        self.compress_down, self.compress_up = [], []
        gen_state = torch.random.get_rng_state()
        torch.manual_seed(20260625)
        for _ in self.deltakv_layer_ids:
            self.compress_down.append(create_compressor(True, config).to(device=self.device, dtype=torch.bfloat16))
            self.compress_up.append(create_compressor(False, config).to(device=self.device, dtype=torch.bfloat16))
        torch.random.set_rng_state(gen_state)
        for m in self.compress_down + self.compress_up:
            m.requires_grad_(False)

    def load_compressor_state(self, l_idx: int, *, up: dict | None = None, down: dict | None = None):
        if up is not None:
            self.compress_up[l_idx].load_state_dict(up)
            self.__dict__.pop("_up_stack", None)
        if down is not None:
            self.compress_down[l_idx].load_state_dict(down)

    def permute_free_slots(self, seed: int):
        """Scatter the slot pools (benchmarks / tests: a genuinely paged gather)."""
        self._pool_sparse.permute(seed)
        self._pool_latent.permute(seed + 1)
        self._pool_full.permute(seed + 2)
        if self._pool_kivi is not None:
            self._pool_kivi.permute(seed + 3)

    # ------------------------------------------------------------------ operator surface
    def get_layer_batch_states(self, layer_idx: int) -> LayerBatchStates:
        return self.full_layer_batch_states if layer_idx in self.full_layer_to_idx else self.deltakv_layer_batch_states

    def get_layer_kv_cache(self, layer_idx: int):
        if layer_idx in self.full_layer_to_idx:
            i = self.full_layer_to_idx[layer_idx]
            return self.full_kv_cache[0, i], self.full_kv_cache[1, i]
        i = self.deltakv_layer_to_idx[layer_idx]
        return self.deltakv_full_kv_cache[0, i], self.deltakv_full_kv_cache[1, i]

    def get_layer_buffer_req_to_token_slots(self, layer_idx: int) -> torch.Tensor:
        return self.full_layer_slots_map if layer_idx in self.full_layer_to_idx else self.sparse_layer_raw_slots_map

    @property
    def num_free_slots(self) -> int:
        return min(self._pool_full.free, self._pool_sparse.free)

    def free_slot_stats(self) -> dict:
        out = {"full": self._pool_full.free, "deltakv_full": self._pool_sparse.free, "latent": self._pool_latent.free}
        if self._pool_kivi is not None:
            out["kivi_blocks"] = self._pool_kivi.free
        return out

    def _stores_sparse_raw_kv(self, layer_idx: int) -> bool:
        return layer_idx in self.deltakv_layer_to_idx

    def save_raw_kv_if_needed(self, layer_idx: int, k: torch.Tensor, v: torch.Tensor):
        """deltakv_less_memory.py:1269-1281: sparse layers keep the *pre-RoPE* key."""
        if not self._stores_sparse_raw_kv(layer_idx):
            return
        if self._raw_store_rides_in_view(k):
            # MI355X: one launch less on the per-layer chain of a decode step -- the materialise launch of this layer
            # (get_layer_compute_view) writes the row and reads it from k/v directly
            self._pending_raw_store[layer_idx] = (k, v)
            return
        super().save_rope_kv_if_needed(layer_idx, k, v)

    def _raw_store_rides_in_view(self, k: torch.Tensor) -> bool:
        if os.environ.get("SVK_DELTAKV_FUSE_RAW_STORE", "1") == "0" or get_context().is_prefill:
            return False
        mapping = self.deltakv_layer_batch_states.slot_mapping
        return (mapping is not None and mapping is self._deltakv_decode_static_slot_mapping and k.dim() == 3
                and int(k.shape[0]) == int(mapping.numel()))

    def on_forward_end(self, seqs, is_prefill: bool):
        self._join_recon_stream()
        self._flush_pending_raw_stores()
        self._reset_prefill_staging()
        return super().on_forward_end(seqs, is_prefill)

    # every position below a row's length is mapped on the full layers (a raw slot or a KIVI block: `_prepare_prefill`,
    # `prepare_decode_static` and `_full_layer_kivi_evict` keep full_layer_slots_map / ..._kivi_block_slots_map a partition
    # of [0, len)), so the observation layers' raw-score launch writes every position the score kernels read
    # (SparseController._get_decode_attn_score_buffer)
    decode_scores_cover_rows = True

    # ------------------------------------------------------------------ prompt-side attention view
    @property
    def prefill_attention_view_supported(self) -> bool:
        """Whether THIS prefill step has a prompt-side attention view: first-prefill steps do (below), continuation
        chunks do not (their sparse layers would need the reconstructed view, deltakv_base.py:936-972)."""
        return bool(self._deltakv_prefill_staging_active)

    def has_prefill_staging_view(self, layer_idx: int) -> bool:
        """deltakv_base.py:1020-1024."""
        return bool(self._deltakv_prefill_staging_active and layer_idx in self.deltakv_layer_to_idx)

    def get_prefill_staging_view(self, layer_idx: int):
        """deltakv_base.py:1026-1037 -> (active_slots, req_indices, context_lens, temp_slots)."""
        if not self.has_prefill_staging_view(layer_idx):
            raise NotImplementedError("DeltaKV prefill staging view is not active for this layer.")
        return (self._deltakv_prefill_staging_active_slots, self._deltakv_prefill_staging_req_indices,
                self._deltakv_prefill_staging_context_lens, None)

    def get_prefill_compute_view(self, layer_idx: int, k_current, v_current, selection, active_slots, req_indices,
                                 context_lens):
        """deltakv_base.py:904-934.  A first-prefill step attends the chunk's own post-RoPE K/V (`k_current`,
        `v_current`: the staging view's slots are positions in the step's token order); the full layers of such a step
        hold the whole row raw (no KIVI block exists before the chunk-end eviction), so their slot table is the view."""
        if self.has_prefill_staging_view(layer_idx):
            return k_current, v_current, active_slots, req_indices, context_lens
        if layer_idx in self.full_layer_to_idx and self._deltakv_prefill_staging_active:
            k_cache, v_cache = self.get_layer_kv_cache(layer_idx)
            return k_cache, v_cache, active_slots, req_indices, context_lens
        raise NotImplementedError

    def build_prefill_compute_view(self, layer_idx: int, k_current, v_current, selection):
        """deltakv_base.py:936-972.  Built here: the FIRST prefill step of a prompt (every row starts at length 0 -
        the reference's "full prefill" staging, deltakv_base.py:1852-1993: the sparse layers attend the step's own K/V
        through a staging slot view, the pre-RoPE rows go to the raw slots for the chunk-end compression).  Not built:
        continuation chunks, whose sparse layers attend a reconstructed, RoPE-rotated view
        (`deltakv_reconstruct(chunk_lens=...)`; SURVEY.md section 2 marks it out of scope) and whose full layers have an
        int4 KIVI prefix - the plain slot table is NOT that view (pre-RoPE keys, slot -1 holes), so those are refused
        instead of computing attention over the wrong bytes."""
        if self._deltakv_prefill_staging_active:
            return super().build_prefill_compute_view(layer_idx, k_current, v_current, selection)
        raise NotImplementedError(
            "DeltaKV prefill attention of a continuation chunk needs the reconstructed prefill compute view, which this "
            "build does not have; prefill_chunk(..., outputs=None) runs the store / compression side of the chunk only "
            f"(layer={int(layer_idx)}).")

    def _join_recon_stream(self):
        """Look-ahead reconstructions that no layer waited for (a layer of the group left early, an empty view, an
        exception between layers) are still in flight on the side stream: join it before anything mutates the caches
        it reads (`deltakv_evict`, the next step) - also what lets a hipGraph capture end with no unjoined branch."""
        ahead = self.__dict__.get("_recon_ahead")
        if ahead:
            for side in self.__dict__.get("_recon_streams") or ():
                torch.cuda.current_stream().wait_stream(side)
        if ahead:
            self._recon_ahead = {}

    def _flush_pending_raw_stores(self):
        """Rows whose layer never reached get_layer_compute_view in this step (none in the decode driver's flow)."""
        while self._pending_raw_store:
            layer_idx, (k, v) = self._pending_raw_store.popitem()
            super().save_rope_kv_if_needed(layer_idx, k, v)

    def save_rope_kv_if_needed(self, layer_idx: int, k: torch.Tensor, v: torch.Tensor):
        if self._stores_sparse_raw_kv(layer_idx):
            return None
        return super().save_rope_kv_if_needed(layer_idx, k, v)

    def fused_decode_store_slots(self, layer_idx: int):
        """MI355X: a KIVI full layer's raw store of the step (plain `store_kvcache` into the layer's raw rows, no side
        effect) rides in its stage-1 launch when the wide kernel serves it (`SVK_DELTAKV_FUSE_FULL_STORE=0`: own launch)."""
        if (layer_idx not in self.full_layer_to_idx or not self._full_layer_kivi_enabled() or get_context().is_prefill
                or os.environ.get("SVK_DELTAKV_FUSE_FULL_STORE", "1") != "1"):
            return None
        ok = self.__dict__.get("_kivi_fused_store_ok")
        if ok is None:
            ok = self._kivi_fused_store_ok = dk.kivi_fused_store_supported(
                head_dim=self.head_dim, num_kv_heads=self.num_kv_heads, group_size=self._full_layer_kivi_group_size(),
                block_seq=self.get_decode_block_seq(layer_idx, 256), key_param_dtype=self.full_layer_kivi_key_scales.dtype)
        mapping = self.full_layer_batch_states.slot_mapping
        return mapping if ok and mapping is not None and mapping.dtype == torch.int32 else None

    def get_decode_block_seq(self, layer_idx: int, default: int) -> int:
        if self._full_layer_kivi_enabled() and layer_idx in self.full_layer_to_idx:
            bs = int(self.config.full_layer_kivi_decode_block_seq or default)
            wide = self.__dict__.get("_kivi_wide_ok")
            if wide is None:          # does the wide (whole-block, 16-byte-load) kernel serve this head shape?
                wide = self._kivi_wide_ok = dk.kivi_fused_store_supported(
                    head_dim=self.head_dim, num_kv_heads=self.num_kv_heads, group_size=self._full_layer_kivi_group_size(),
                    block_seq=128, key_param_dtype=self.full_layer_kivi_key_scales.dtype)
            if wide:
                # MI355X launch geometry of the default (whole-block, wide-load) kernel: a workgroup pays ~15 us of
                # prologue (boundary / classification / first-tile loads), so ranges are as long as still fills the
                # chip: two workgroups per CU when that leaves >= 8 tiles per workgroup, else one
                # (tools/kbench_kivi.py: 1 x 256k: 1024 -> 72.7 us, 512 -> 79.4, 256 -> 85.2; 4 x 256k: 2048 -> 180.9
                # us, 1024 -> 191.7, 512 -> 204.7, 256 -> 232.2)
                rows = max(1, int(self.config.max_num_seqs_in_gpu))
                tokens = rows * int(self.max_model_len)
                # Two per CU: the launch must fit the 512 resident slots INCLUDING the three extra workgroups per row
                # (raw / ragged pieces, ~50 us each) - the regular workgroups they displace start when the extras end
                # and finish that much later (4 x 256k: block_seq 2048 = 516 + 12 workgroups 182 us, 2304 = 456 + 12
                # 170 us; 8 x 256k: 4096 -> 318 us, 4480 -> 315 us; tools/kbench_kivi.py after 0.4 s of warm-up - the
                # first configuration timed in a process reads 10-20 % slow).  One per CU: the overflow lands on a
                # CU's second slot, nothing waits.
                # A workgroup is Hkv waves: a tensor-parallel rank with 1 or 2 KV heads gets 4 / Hkv times the workgroups
                # for the same waves per CU (7q/1kv, 4 x 256k: block_seq 640 -> 62 us, 1024 -> 78; 14q/2kv: 1152 -> 99 us,
                # 2304 -> 160; profiles/r03/tp_shapes.txt).
                two = tokens >= 512 * 1024
                scale = max(1, 4 // max(1, int(self.num_kv_heads)))
                per_row = max(1, ((480 * scale) // rows - 3) if two else (256 * scale) // rows)
                bs = max(bs, -(-(-(-int(self.max_model_len) // per_row)) // 128) * 128)
            return bs
        return super().get_decode_block_seq(layer_idx, default)

    def free_part_slots(self, layer_idx: int, seq, keep_indices, *, keep_indices_sorted: bool = False):
        raise ValueError("DeltaKV does not evict through free_part_slots; it compresses the raw tail (deltakv_evict).")

    @torch.no_grad()
    def _prepare_prefill(self, seqs):
        """One prompt chunk per sequence appended RAW on both layer kinds (deltakv_base.py:1881-2036): every token gets a
        full-layer slot and a sparse-layer slot; `SparseController.post_forward` -> `deltakv_evict` then compresses
        whole multiples of `recent` of the raw tail in one go and the KIVI eviction quantises the full-layer groups
        that left the residual window - the prompt's compressed state is produced by the same operators as in decode.
        -> (cu_seqlens_q int32 [B + 1], total tokens)"""
        self._deltakv_reset_view_cache()
        self._reset_prefill_staging()
        d = self.device
        chunk_lens = [int(s.current_chunk_size) for s in seqs]
        rows, ctx, full_parts, sparse_parts = [], [], [], []
        first_prefill = True
        for s, n in zip(seqs, chunk_lens):
            row = self._get_free_row(s.seq_id)
            cur = int(self.row_seq_lens[row])
            first_prefill = first_prefill and cur == 0
            if n <= 0:
                raise ValueError(f"DeltaKV prefill chunk must be positive, got {n} for seq_id={s.seq_id}.")
            if cur + n > self.max_model_len:
                raise RuntimeError(f"KV row length exceeds max_model_len in DeltaKV prefill: cur_len={cur} chunk={n} "
                                   f"max_model_len={self.max_model_len}")
            fs, ss = self._pool_full.pop(n), self._pool_sparse.pop(n)
            fs_gpu, ss_gpu = torch.from_numpy(fs).to(d), torch.from_numpy(ss).to(d)
            pos = torch.arange(cur, cur + n, dtype=torch.int32, device=d)
            self.full_layer_slots_map[row, cur: cur + n] = fs_gpu
            self.full_layer_slot_to_pos[fs_gpu.long()] = pos
            self.sparse_layer_raw_slots_map[row, cur: cur + n] = ss_gpu
            self.deltakv_slot_to_pos[ss_gpu.long()] = pos
            self.row_seq_lens[row] = cur + n
            rows.append(row)
            ctx.append(cur + n)
            full_parts.append(fs_gpu)
            sparse_parts.append(ss_gpu)
        context_lens = torch.tensor(ctx, dtype=torch.int32, device=d)
        req_indices = torch.tensor(rows, dtype=torch.int32, device=d)
        for state, parts in ((self.full_layer_batch_states, full_parts), (self.deltakv_layer_batch_states, sparse_parts)):
            state.slot_mapping = torch.cat(parts)
            state.context_lens, state.req_indices = context_lens, req_indices
            state.max_context_len = max(ctx)
        self._deltakv_decode_static_slot_mapping = None
        self._deltakv_decode_static_compressed_lens = None
        cu = np.concatenate(([0], np.cumsum(chunk_lens))).astype(np.int32)
        if first_prefill and seqs:
            # deltakv_base.py:1965-1993: the staging view of a first-prefill step - row b's positions are the tokens
            # [cu[b], cu[b + 1]) of the step, -1 beyond its chunk
            table = np.full((len(seqs), max(chunk_lens)), -1, dtype=np.int32)
            for b, n in enumerate(chunk_lens):
                table[b, :n] = np.arange(cu[b], cu[b] + n, dtype=np.int32)
            self._deltakv_prefill_staging_active = True
            self._deltakv_prefill_staging_active_slots = torch.from_numpy(table).to(d)
            self._deltakv_prefill_staging_req_indices = torch.arange(len(seqs), dtype=torch.int32, device=d)
            self._deltakv_prefill_staging_context_lens = context_lens
        return torch.from_numpy(cu).to(d), int(sum(chunk_lens))

    def _prepare_decode(self, seqs):
        return self.prepare_decode_static(seqs)

    def _get_free_row(self, seq_id: int) -> int:
        row = self.seq_id_to_row.get(seq_id)
        if row is None:
            if not self.free_rows:
                raise RuntimeError("No free DeltaKV rows")
            row = self.free_rows.popleft()
            self.seq_id_to_row[seq_id] = row
        return int(row)

    def _deltakv_reset_view_cache(self):
        self._deltakv_view_cache_key = None
        self._deltakv_view_cache_value = None

    # ------------------------------------------------------------------ ingest of an already-compressed row
    @torch.no_grad()
    def admit_compressed_row(self, seq, *, total_len: int, compressed_len: int, center_positions, father_center_index,
                             sparse_k_raw, sparse_v, latent, full_k=None, full_v=None, kivi_quantized_end: int = 0,
                             kivi_blocks=None):
        """Install one sequence whose prompt has already been prefilled and compressed.

        positions [0, sink) and [sink+compressed_len, total_len) are raw on the sparse layers; positions
        `center_positions` (inside the compressed range) additionally keep their raw K/V (cluster centres);
        every compressed position p has a latent residual and `K` fathers `father_center_index[p - sink]`
        (indices into sink ++ centres, -1 padded like the reference pads with the first father).
          sparse_k_raw / sparse_v : [Ls, n_raw, Hkv, D] bf16 for the raw positions in ascending position order
          latent : dict(code=[Ls, clen, W] int32, scale=[Ls, clen, g] bf16, mn=...) or dict(dense=[Ls, clen, payload] bf16)
          full_k / full_v : [Lf, n_full_raw, Hkv, D] bf16 post-RoPE rows of the raw full-layer positions
          kivi_blocks : dict(key_packed, key_scales, key_mins, value_packed, value_scales, value_mins) with a
                        leading [Lf, n_blocks] for positions [sink, kivi_quantized_end) when KIVI is enabled."""
        cfg, d = self.config, self.device
        sink = int(cfg.num_sink_tokens)
        row = self._get_free_row(seq.seq_id)
        assert int(self.row_seq_lens[row]) == 0, "row already populated"
        total_len, clen = int(total_len), int(compressed_len)
        if total_len > self.max_model_len:
            raise RuntimeError(f"DeltaKV row length exceeds max_model_len: {total_len} > {self.max_model_len}")
        centers = np.asarray(center_positions, dtype=np.int64)
        raw_pos = np.concatenate((np.arange(min(sink, total_len)), centers, np.arange(sink + clen, total_len))).astype(np.int64)
        assert np.all(np.diff(raw_pos) > 0), "raw positions must be strictly ascending (sink < centres < tail)"
        n_raw = raw_pos.size
        slots = self._pool_sparse.pop(n_raw)
        slots_gpu = torch.from_numpy(slots).to(d)
        pos_gpu = torch.from_numpy(raw_pos).to(d)
        self.sparse_layer_raw_slots_map[row, pos_gpu] = slots_gpu
        self.deltakv_slot_to_pos[slots_gpu.long()] = pos_gpu.to(torch.int32)
        self.deltakv_full_kv_cache[0][:, slots_gpu.long()] = sparse_k_raw.to(d)
        self.deltakv_full_kv_cache[1][:, slots_gpu.long()] = sparse_v.to(d)
        if clen > 0:
            lat = self._pool_latent.pop(clen)
            lat_gpu = torch.from_numpy(lat).to(d)
            self.sparse_layer_latent_slots_map[row, sink: sink + clen] = lat_gpu
            if "dense" in latent:
                self.deltakv_latent_cache[:, lat_gpu.long()] = latent["dense"].to(d)
            else:
                self.deltakv_latent_cache[:, lat_gpu.long()] = latent["code"].to(d)
                self.deltakv_latent_scales[:, lat_gpu.long()] = latent["scale"].to(d)
                self.deltakv_latent_mins[:, lat_gpu.long()] = latent["mn"].to(d)
            # fathers: indices into (sink ++ centres) -> slot ids
            center_slots = np.concatenate((slots[: min(sink, total_len)], slots[min(sink, total_len): min(sink, total_len) + centers.size]))
            fidx = np.asarray(father_center_index, dtype=np.int64)
            fslots = center_slots[np.maximum(fidx, 0)]
            fslots = np.where(fidx >= 0, fslots, fslots[..., :1])
            self.deltakv_latent_to_full_slots[:, lat_gpu.long()] = torch.from_numpy(fslots.astype(np.int32)).to(d)
            self.row_latent_slots[row] = lat
            self.row_deltakv_center_slots[row] = torch.from_numpy(center_slots.astype(np.int32)).to(d)
        # full layers: KIVI blocks cover [sink, quant_end) (deltakv_less_memory.py:3495-3558); the rest stays raw
        qend = int(kivi_quantized_end) if self._full_layer_kivi_enabled() else 0
        qstart = min(sink, total_len)
        if qend <= qstart:
            qend = qstart
        full_pos = np.concatenate((np.arange(qstart), np.arange(qend, total_len))).astype(np.int64)
        fs = self._pool_full.pop(full_pos.size)
        fs_gpu = torch.from_numpy(fs).to(d)
        fpos_gpu = torch.from_numpy(full_pos).to(d)
        if qend > qstart:
            self.full_layer_slots_map[row, qstart:qend] = -1      # the kernel picks the KIVI block for these
        self.full_layer_slots_map[row, fpos_gpu] = fs_gpu
        self.full_layer_slot_to_pos[fs_gpu.long()] = fpos_gpu.to(torch.int32)
        if full_k is not None:
            self.full_kv_cache[0][:, fs_gpu.long()] = full_k.to(d)
            self.full_kv_cache[1][:, fs_gpu.long()] = full_v.to(d)
        if qend > qstart:
            G = self._full_layer_kivi_group_size()
            assert (qend - qstart) % G == 0
            nb = (qend - qstart) // G
            blocks = self._pool_kivi.pop(nb)
            b_gpu = torch.from_numpy(blocks).to(d)
            self.full_layer_kivi_block_slots_map[row, qstart:qend] = b_gpu.repeat_interleave(G)
            self.full_layer_kivi_block_start_pos[b_gpu.long()] = torch.arange(qstart, qend, G, dtype=torch.int32, device=d)
            if kivi_blocks is not None:
                for name in ("key_packed", "key_scales", "key_mins", "value_packed", "value_scales", "value_mins"):
                    getattr(self, f"full_layer_kivi_{name}")[:, b_gpu.long()] = kivi_blocks[name].to(d)
            self.row_kivi_blocks[row] = [int(x) for x in blocks]
            self.row_full_layer_kivi_quantized_lens[row] = qend
        self.row_seq_lens[row] = total_len
        self.row_deltakv_compressed_lens[row] = clen
        self.row_deltakv_compressed_lens_gpu[row] = clen
        return row

    # ------------------------------------------------------------------ decode preparation
    @torch.no_grad()
    def _device_state_sync(self):
        """The device copy of the decode bookkeeping - row lengths, the full-layer and sparse-layer free stacks with their
        pointers (compressed lengths already live in `row_deltakv_compressed_lens_gpu`) - is brought up to the host's
        state when anything but the device-resident step itself has touched it since (compression, admission, free_seq,
        scratch allocations): pool versions and row lengths say so.  Returns the state."""
        st, d = self._dev_state, self.device
        if st is None:
            i32 = torch.int32
            st = self._dev_state = dict(
                row_len=torch.zeros((self.max_buffer_rows,), dtype=i32, device=d),
                full_stack=torch.zeros((self._pool_full.size,), dtype=i32, device=d), full_ptr=torch.zeros((1,), dtype=i32, device=d),
                sparse_stack=torch.zeros((self._pool_sparse.size,), dtype=i32, device=d), sparse_ptr=torch.zeros((1,), dtype=i32, device=d))
            self._dev_step_cache, self._dev_stands_for = None, None
        cur = self._dev_stands_for
        if cur is None or cur[0] != self._pool_full.version or cur[1] != self._pool_sparse.version \
                or not np.array_equal(cur[2], self.row_seq_lens):
            st["row_len"].copy_(torch.from_numpy(self.row_seq_lens))
            for name, pool in (("full", self._pool_full), ("sparse", self._pool_sparse)):
                n = int(pool.free)
                if n:
                    st[name + "_stack"][:n].copy_(torch.from_numpy(pool.stack[:n]))
                st[name + "_ptr"].fill_(n)
            self._dev_stands_for = (self._pool_full.version, self._pool_sparse.version, self.row_seq_lens.copy())
        return st

    def device_step_begin(self):
        """The step's allocation launch, for a caller that took `defer_device_launch` (inside its hipGraph)."""
        if self._device_step is not None:
            dk.deltakv_device_step_begin(self._device_step[0])
            self._device_step[1] = True

    def device_step_mark_launched(self):
        """A replayed hipGraph carried this step's allocation launch."""
        if self._device_step is not None:
            self._device_step[1] = True

    def device_step_burst(self):
        """(protocol of the decode driver: DeltaKV's compression side stays with `deltakv_evict` after the forward)"""

    def prepare_decode_static(self, seqs, input_ids=None, positions=None, slot_mapping=None, context_lens=None,
                              req_indices=None, *, graph_batch_size: int | None = None, defer_device_launch: bool = False):
        """deltakv_base.py:2038-2154: one new raw slot per row in each pool, graph-stable metadata buffers."""
        with profiler.record("cache_prepare_decode"):
            self._deltakv_reset_view_cache()
            self._reset_prefill_staging()
            B = len(seqs)
            if B <= 0:
                raise ValueError("Static DeltaKV decode requires a non-empty real decode batch.")
            GB = int(graph_batch_size or (slot_mapping.numel() if slot_mapping is not None else B))
            if B > GB:
                raise ValueError("Static DeltaKV decode graph batch is smaller than the real decode batch: "
                                 f"graph={GB}, real={B}.")
            d = self.device
            rows = np.asarray([self._get_free_row(s.seq_id) for s in seqs], dtype=np.int64)
            cur = self.row_seq_lens[rows].copy()
            if int(cur.max()) + 1 > self.max_model_len:
                raise RuntimeError(f"KV row length exceeds max_model_len in DeltaKV decode: max_cur_len={int(cur.max())}")
            buf = cur - int(self.config.num_sink_tokens) - self.row_deltakv_compressed_lens[rows]
            if int(buf.max()) + 1 > self._deltakv_decode_static_max_buffer():
                raise RuntimeError("DeltaKV raw tail exceeds the static decode buffer; deltakv_evict (the compression "
                                   "side) must run after every forward (SparseController.post_forward): "
                                   f"tail={int(buf.max()) + 1} max_buffer={self._deltakv_decode_static_max_buffer()}.")
            prev = self._device_step
            if prev is not None and not prev[1]:
                # the previous step's allocation launch never ran (its forward raised, or the caller deferred and never
                # launched): the host mirrors moved, the device copies did not - they stand for nothing until re-uploaded
                self._dev_stands_for = None
            self._device_step = None
            use_device = self._device_step_enabled and slot_mapping is None and context_lens is None and req_indices is None \
                and len(set(rows.tolist())) == B and min(self._pool_full.free, self._pool_sparse.free) >= B
            key = (GB,)
            st = self._static.get(key)
            if st is None:
                st = tuple(torch.zeros((GB,), dtype=torch.int32, device=d) for _ in range(5))
                self._static[key] = st
            own_slot_mapping, own_ctx, own_req, sparse_mapping, compressed = st
            slot_mapping = own_slot_mapping if slot_mapping is None else slot_mapping
            context_lens = own_ctx if context_lens is None else context_lens
            req_indices = own_req if req_indices is None else req_indices
            if use_device:
                # MI355X: the step pops its slots from the device copies of the free stacks, by the arithmetic of the two
                # host pops below (SvkDeltakvDeviceStepArgs) - nothing is uploaded, the launch is a node of the step's graph
                dev = self._device_state_sync()                  # BEFORE the host mirrors move
                ckey = (tuple(int(r) for r in rows), GB)
                cache = self._dev_step_cache
                if cache is None or cache[0] != ckey:
                    rows_gpu = torch.from_numpy(rows.astype(np.int32)).to(d)
                    args = dk.deltakv_device_step_args(
                        rows=rows_gpu, row_len=dev["row_len"], compressed_len=self.row_deltakv_compressed_lens_gpu,
                        full_stack=dev["full_stack"], full_ptr=dev["full_ptr"], sparse_stack=dev["sparse_stack"],
                        sparse_ptr=dev["sparse_ptr"], full_slots_map=self.full_layer_slots_map,
                        full_slot_to_pos=self.full_layer_slot_to_pos, sparse_raw_slots_map=self.sparse_layer_raw_slots_map,
                        sparse_slot_to_pos=self.deltakv_slot_to_pos, context_lens=context_lens, req_indices=req_indices,
                        slot_mapping=slot_mapping, sparse_slot_mapping=sparse_mapping, compressed_lens=compressed, batch=B)
                    cache = self._dev_step_cache = (ckey, args, rows_gpu)
                    self.device_step_generation += 1
                self._device_step = [cache[1], False]
            full_slots = self._pool_full.pop(B)
            sparse_slots = self._pool_sparse.pop(B)
            # the step's host data as ONE upload, the four map scatters and five buffer fills as ONE launch
            # (svk_deltakv_decode_alloc; the reference issues them one by one, each with its own small upload: ~0.45 ms
            # of host-driven copies per step in front of a 2.1 ms graph replay)
            self.row_seq_lens[rows] += 1
            real_lens = self.row_seq_lens[rows]
            if use_device:
                # the device copy moves by exactly these pops and increments when the launch runs
                self._dev_stands_for = (self._pool_full.version, self._pool_sparse.version, self.row_seq_lens.copy())
                if not defer_device_launch:
                    self.device_step_begin()
            else:
                clens = self.row_deltakv_compressed_lens[rows]
                meta = np.stack([rows, cur, full_slots, sparse_slots, clens]).astype(np.int32)
                meta_gpu = torch.from_numpy(meta).to(d, non_blocking=True)
                dk.deltakv_decode_alloc(meta_gpu, batch=B, full_slots_map=self.full_layer_slots_map,
                                        full_slot_to_pos=self.full_layer_slot_to_pos,
                                        sparse_raw_slots_map=self.sparse_layer_raw_slots_map,
                                        sparse_slot_to_pos=self.deltakv_slot_to_pos, context_lens=context_lens,
                                        req_indices=req_indices, slot_mapping=slot_mapping, sparse_slot_mapping=sparse_mapping,
                                        compressed_lens=compressed)
            self._deltakv_decode_static_slot_mapping = sparse_mapping
            self._deltakv_decode_static_compressed_lens = compressed
            cap = self._decode_static_max_context_len
            max_ctx = int(cap) if cap is not None else int(real_lens.max())
            for state, mapping in ((self.full_layer_batch_states, slot_mapping), (self.deltakv_layer_batch_states, sparse_mapping)):
                state.slot_mapping, state.context_lens, state.req_indices = mapping, context_lens, req_indices
                state.max_context_len = max_ctx
            return input_ids, positions, None

    def get_compressed_lens(self, req_indices: torch.Tensor) -> torch.Tensor:
        """deltakv_base.py:2156-2165."""
        c = self._deltakv_decode_static_compressed_lens
        if c is not None and not get_context().is_prefill:
            return c[: req_indices.numel()]
        return self.row_deltakv_compressed_lens_gpu[req_indices.to(torch.long)].to(torch.int32)

    def free_seq(self, seq_id: int):
        """deltakv_base.py:1534-1575."""
        row = self.seq_id_to_row.pop(seq_id, None)
        if row is None:
            raise ValueError(f"free_seq: unknown seq_id={seq_id}")
        n = int(self.row_seq_lens[row])
        if n > 0:
            raw = self.sparse_layer_raw_slots_map[row, :n].cpu().numpy()
            raw = raw[raw >= 0]
            self._pool_sparse.push(raw)
            self.deltakv_slot_to_pos[torch.from_numpy(raw.astype(np.int64)).to(self.device)] = -1
            full = self.full_layer_slots_map[row, :n].cpu().numpy()
            full = full[full >= 0] if self._full_layer_kivi_enabled() else full
            self._pool_full.push(full)
            self.full_layer_slot_to_pos[torch.from_numpy(full.astype(np.int64)).to(self.device)] = -1
        lat = self.row_latent_slots.pop(row, None)
        if lat is not None:
            self._pool_latent.push(lat)
        self.row_deltakv_center_slots.pop(row, None)
        blocks = self.row_kivi_blocks.pop(row, None)
        if blocks:
            self._pool_kivi.push(np.asarray(blocks, dtype=np.int32))
            self.full_layer_kivi_block_start_pos[torch.tensor(blocks, dtype=torch.long, device=self.device)] = -1
            self.full_layer_kivi_block_slots_map[row, :] = -1
            self.row_full_layer_kivi_quantized_lens[row] = 0
        self.full_layer_slots_map[row, :] = 0
        self.sparse_layer_raw_slots_map[row, :] = -1
        self.sparse_layer_latent_slots_map[row, :] = -1
        self.row_seq_lens[row] = 0
        self.row_deltakv_compressed_lens[row] = 0
        self.row_deltakv_compressed_lens_gpu[row] = 0
        self.free_rows.append(row)
        self._deltakv_reset_view_cache()

    # ------------------------------------------------------------------ sparse-layer decode view
    def _ensure_decode_static_temp_slots(self, batch_size: int, k_max: int) -> torch.Tensor:
        """deltakv_less_memory.py:449-470: a persistent [B, K] scratch block reused by every sparse layer."""
        key = (int(batch_size), int(k_max))
        slots = self._temp_slots_by_shape.get(key)
        if slots is None:
            if key[0] == 0 or key[1] == 0:
                slots = torch.empty(key, dtype=torch.int32, device=self.device)
            else:
                slots = torch.from_numpy(self._pool_sparse.pop(key[0] * key[1])).to(self.device).view(*key)
            self._temp_slots_by_shape[key] = slots
        return slots

    def _ensure_decode_static_plan_buffers(self, batch_size: int, k_max: int, max_s: int):
        key = (int(batch_size), int(k_max), int(max_s))
        bufs = self._plan_buffers.get(key)
        if bufs is None:
            d, i32 = self.device, torch.int32
            B, K, S = key
            bufs = (torch.zeros((B, S), dtype=i32, device=d), torch.zeros((B, S), dtype=i32, device=d),
                    torch.arange(B, dtype=i32, device=d), torch.zeros((B,), dtype=i32, device=d),
                    torch.empty((0,), dtype=i32, device=d), torch.zeros((B * K,), dtype=i32, device=d),
                    torch.zeros((B * K,), dtype=i32, device=d), torch.zeros((B * K,), dtype=i32, device=d))
            self._plan_buffers[key] = bufs
        return bufs

    def _deltakv_build_view_and_plan_reconstruct(self, layer_idx: int, active_compressed_indices, req_indices):
        """deltakv_less_memory.py:2850-2882: one plan per observation group (the slot maps are shared by all
        sparse layers), keyed on the stable input addresses."""
        act = active_compressed_indices
        key = (int(req_indices.data_ptr()), int(req_indices.numel()), 0 if act is None else int(act.data_ptr()),
               int(req_indices.numel()) if act is None else int(act.shape[0]), 0 if act is None else int(act.shape[1]))
        if self._deltakv_view_cache_key == key and self._deltakv_view_cache_value is not None:
            return self._deltakv_view_cache_value
        with profiler.record("deltakv_build_view_total"):
            out = self._deltakv_build_view_and_plan_reconstruct_static(layer_idx, act, req_indices)
        self._deltakv_view_cache_key, self._deltakv_view_cache_value = key, out
        return out

    def _deltakv_build_view_and_plan_reconstruct_static(self, layer_idx: int, active_compressed_indices, req_indices):
        """deltakv_less_memory.py:2884-2934."""
        if layer_idx in self.full_layer_to_idx:
            raise ValueError("deltakv_reconstruct should only be called for sparse layers.")
        bsz = int(req_indices.shape[0])
        if active_compressed_indices is None:
            active_compressed_indices = torch.empty((bsz, 0), device=req_indices.device, dtype=torch.int32)
        k_max = int(active_compressed_indices.shape[1])
        context_lens = self.deltakv_layer_batch_states.context_lens
        if context_lens is None:
            raise RuntimeError("DeltaKV static decode context_lens buffer was not initialized.")
        context_lens = context_lens[:bsz]
        compressed_lens = self.get_compressed_lens(req_indices)
        sink = int(self.config.num_sink_tokens)
        max_buffer = self._deltakv_decode_static_max_buffer()
        max_s = sink + k_max + max_buffer
        temp_slots = self._ensure_decode_static_temp_slots(bsz, k_max)
        active_slots, active_pos, local_req, new_context_lens, no_free, recon_pos, recon_latent, recon_out_slot = \
            self._ensure_decode_static_plan_buffers(bsz, k_max, max_s)
        dk.deltakv_static_decode_plan(
            raw_slots_map=self.sparse_layer_raw_slots_map, latent_slots_map=self.sparse_layer_latent_slots_map,
            active_compressed_indices=active_compressed_indices, req_indices=req_indices, context_lens=context_lens,
            compressed_lens=compressed_lens, temp_slots=temp_slots, active_slots_out=active_slots,
            active_pos_out=active_pos, new_context_lens_out=new_context_lens, recon_pos_out=recon_pos,
            recon_latent_out=recon_latent, recon_out_slot_out=recon_out_slot, sink=sink, max_buffer=max_buffer)
        self._deltakv_decode_static_active_pos = active_pos
        return active_slots, local_req, new_context_lens, no_free, recon_pos, recon_latent, recon_out_slot

    def _load_residual(self, l_idx: int, recon_latent: torch.Tensor, bufs=None) -> torch.Tensor:
        """deltakv_less_memory.py:2841-2848: latent gather -> (int4 dequant) -> compress_up (library GEMMs).
        `bufs` = (hidden, delta) caller-owned outputs of the two Linear layers (look-ahead on the side stream)."""
        if int(self.config.kv_quant_bits or 0) == 4:
            cache = self.deltakv_latent_cache[l_idx]
            up = self.compress_up[l_idx]
            fused = self._fused_up_parts(up, cache)
            if fused is not None:
                # gather + dequant + Linear + GELU as one MFMA launch, the second Linear stays a library GEMM
                lin1, lin2 = fused
                h = dk.dequant_linear_act(cache, self.deltakv_latent_scales[l_idx], self.deltakv_latent_mins[l_idx],
                                          self._quant_group_size(), lin1.weight, lin1.bias, activation="gelu",
                                          row_index=recon_latent, out=None if bufs is None else bufs[0])
                if bufs is None or lin2.bias is None:
                    return torch.nn.functional.linear(h, lin2.weight, lin2.bias)
                return torch.addmm(lin2.bias, h, lin2.weight.t(), out=bufs[1])
            # gather + dequant in one launch (row_index = recon_latent, -1 entries read latent 0 and are never written back)
            residual = dk.dequantize_grouped(cache, self.deltakv_latent_scales[l_idx], self.deltakv_latent_mins[l_idx],
                                             self._quant_group_size(), int(cache.shape[-1]) * 8, 4, row_index=recon_latent)
        else:
            residual = self.deltakv_latent_cache[l_idx, recon_latent.clamp_min(0).long()]
        return self.compress_up[l_idx](residual)

    def _recon_view_geometry(self, active_slots: torch.Tensor, recon_latent: torch.Tensor):
        """(view width, first column of the selected block, entries per batch row) when the reconstruction can write into
        the views (int4 latents through the fused two-Linear compress_up: a dense bf16 delta), else None."""
        if not self.recon_into_view or int(self.config.kv_quant_bits or 0) != 4 or active_slots.dim() != 2:
            return None
        ok = self.__dict__.get("_recon_view_ok")
        if ok is None:
            ok = self._recon_view_ok = all(
                (p := self._fused_up_parts(self.compress_up[i], self.deltakv_latent_cache[i])) is not None and p[1].bias is not None
                for i in range(len(self.compress_up)))
        B = int(active_slots.shape[0])
        if not ok or B <= 0 or int(recon_latent.numel()) % B != 0:
            return None
        return int(active_slots.shape[1]), int(self.config.num_sink_tokens), int(recon_latent.numel()) // B

    def _recon_view_out(self, l0: int, l1: int, geom):
        """(out_k [l1 - l0, rows, Hkv, D], out_v, width, offset, entries per row) of the sparse layers [l0, l1), or None."""
        if not self.recon_into_view or geom is None:
            return None
        if self._layer_views is None:
            Ls = len(self.deltakv_layer_ids)
            nbytes = 2 * Ls * int(self.deltakv_materialized_compute_num_slots) * self.num_kv_heads * self.head_dim * 2
            if nbytes > self._LAYER_VIEWS_MAX_BYTES:         # (a view per sparse layer; beyond the budget: scratch slots + copy)
                self.recon_into_view = False
                return None
            self._layer_views = torch.zeros((2, Ls, int(self.deltakv_materialized_compute_num_slots), self.num_kv_heads,
                                             self.head_dim), dtype=torch.bfloat16, device=self.device)
        width, offset, per_row = geom
        return self._layer_views[0, l0:l1], self._layer_views[1, l0:l1], width, offset, per_row

    def _recon_lookahead_buffers(self, l_idx: int, n: int):
        """Caller-owned (hidden, delta) buffers of one sparse layer for the look-ahead reconstruction, or None when the
        layer's compress_up is not the fused two-Linear form."""
        if int(self.config.kv_quant_bits or 0) != 4:
            return None
        fused = self._fused_up_parts(self.compress_up[l_idx], self.deltakv_latent_cache[l_idx])
        if fused is None or fused[1].bias is None:
            return None
        store = self.__dict__.setdefault("_recon_bufs", {})
        cur = store.get(l_idx)
        hid, out = int(fused[0].weight.shape[0]), int(fused[1].weight.shape[0])
        if cur is None or cur[0].shape[0] < n:
            cur = (torch.empty((n, hid), dtype=torch.bfloat16, device=self.device),
                   torch.empty((n, out), dtype=torch.bfloat16, device=self.device))
            store[l_idx] = cur
        return cur[0][:n], cur[1][:n]

    def _reconstruct_layer(self, l_idx: int, recon_pos, recon_latent, recon_out_slot, bufs=None, view_geom=None):
        k_cache, v_cache = self.deltakv_full_kv_cache[0, l_idx], self.deltakv_full_kv_cache[1, l_idx]
        with profiler.record("deltakv_less_memory_reconstruct_load_residual"):
            kv_delta = self._load_residual(l_idx, recon_latent, bufs)
        view = self._recon_view_out(l_idx, l_idx + 1, view_geom) if kv_delta.dtype == torch.bfloat16 else None
        if view is not None:
            self._recon_view_layers.add(int(l_idx))
        with profiler.record("deltakv_less_memory_reconstruct_writeback"):
            # fathers = latent_to_full_slots[l, recon_latent.clamp_min(0)].clamp_min(0), resolved in-kernel
            dk.deltakv_reconstruct_writeback_grouped_heads(
                kv_delta=kv_delta, father_slots=self.deltakv_latent_to_full_slots[l_idx], father_index=recon_latent,
                slot_to_pos=self.deltakv_slot_to_pos,
                out_slots=recon_out_slot, out_pos=recon_pos, cos_sin=self.cos_sin_cache, k_cache=k_cache,
                v_cache=v_cache, k_norm_weight=None if self.deltakv_k_norm_weight is None else self.deltakv_k_norm_weight[l_idx],
                k_norm_eps=float(self.deltakv_k_norm_eps), raw_k_cache=True, store_raw_k=False,
                view_out=None if view is None else (view[0][0], view[1][0], view[2], view[3], view[4]))

    @staticmethod
    def _fused_up_recon_mode() -> str:
        import os
        return os.environ.get("SVK_DELTAKV_FUSED_UP", "1")

    @staticmethod
    def _recon_lookahead_enabled() -> bool:
        import os
        return os.environ.get("SVK_DELTAKV_RECON_AHEAD", "1") == "1"

    def _group_sparse_layers(self, layer_idx: int) -> list[int]:
        """The sparse layers that share the plan of `layer_idx`: the run of sparse layers it starts, up to the next
        full-attention layer."""
        out, l = [], int(layer_idx)
        while l < self.num_layers and l in self.deltakv_layer_to_idx:
            out.append(l)
            l += 1
        return out

    def _reconstruct_group_ahead(self, layer_idx: int, recon_pos, recon_latent, recon_out_slot, view_geom=None,
                                 view_table=None) -> bool:
        """MI355X: the residual load and reconstruction of a sparse layer depend on the plan of its observation group and
        on the layer's own caches, not on the step's activations - so all layers of the group are issued NOW, back to
        back on a side stream (three launches, ~43 us per layer at 2048 tokens), while the main stream walks the layers
        (store, view, attention: ~25 us per layer here, plus the dense model in a real engine) and only waits for the
        layer's event before it builds the view.  Per step of the 256 k configuration: 2.67 -> see DESIGN.md 4.8."""
        layers = self._group_sparse_layers(layer_idx)
        n = int(recon_latent.numel())
        bufs = {l: self._recon_lookahead_buffers(self.deltakv_layer_to_idx[l], n) for l in layers}
        if len(layers) < 2 or any(b is None for b in bufs.values()):
            return False
        sides = self.__dict__.get("_recon_streams")
        if sides is None or len(sides) != self._recon_stream_count():
            sides = self._recon_streams = [torch.cuda.Stream(device=self.device) for _ in range(self._recon_stream_count())]
            self._recon_events = self.__dict__.get("_recon_events") or {}
        main = torch.cuda.current_stream()
        for side in sides:
            side.wait_stream(main)                   # the plan (and everything before it) is complete for the side streams
        # one event per launch group: the walk joins it at the group's first layer, the other layers of the launch group
        # find it joined (every join of a replayed graph is a cross-queue dependency of 6-12 us, trace_step.py)
        self._recon_event_of = {}
        self._recon_joined = set()
        stack = self._stacked_up_weights()
        sub = self._recon_sub_batch(n)
        if stack is not None and sub > 1 and view_table is not None:
            self._views_rest_ahead(layers, view_table, view_geom, sides)
        if stack is None or sub <= 1:
            for i, l in enumerate(layers):
                side = sides[i % len(sides)]
                with torch.cuda.stream(side):
                    self._reconstruct_layer(self.deltakv_layer_to_idx[l], recon_pos, recon_latent, recon_out_slot, bufs[l],
                                            view_geom=view_geom)
                    self._recon_event(l).record(side)
                self._recon_event_of[l] = l
        else:
            # sub-batches of `sub` consecutive layers: one dequant + Linear + GELU launch, one batched GEMM and one
            # reconstruct launch per sub-batch instead of three launches per layer (48 -> ~28 us per layer), small
            # enough that the main stream can start on the group's first layers while the rest is still in flight.
            # The sub-batches alternate between the side streams: at one or a few rows none of the three launches
            # fills the chip, and the chain of launches - not the CUs - is what the walk of the layers waits for.
            sizes, c0, ci = self._recon_sub_batches(n), 0, 0
            # few selected tokens (one row): the residual loads of the WHOLE layer group as one launch in front of the
            # reconstructions - at 2048 tokens a two-layer load is one latency-bound round of workgroups (21 us) and nine
            # layers of it cost little more than two
            load_first = (n < self._RECON_WIDE_TOKENS and len(sides) == 1 and len(layers) > sizes[0]
                          and os.environ.get("SVK_DELTAKV_RECON_LOAD_FIRST", "1") != "0")
            g_idx = [self.deltakv_layer_to_idx[l] for l in layers]
            if load_first:
                with torch.cuda.stream(sides[0]):
                    self._reconstruct_layers_batched(g_idx, stack, recon_pos, recon_latent, recon_out_slot, view_geom=view_geom,
                                                     buf_key="group", phase="load", buf_l0=g_idx[0], buf_layers=len(g_idx))
            while load_first and c0 < len(layers):
                n_c = sizes[min(ci, len(sizes) - 1)]
                chunk = layers[c0: c0 + n_c]
                with torch.cuda.stream(sides[0]):
                    self._reconstruct_layers_batched([self.deltakv_layer_to_idx[l] for l in chunk], stack, recon_pos,
                                                     recon_latent, recon_out_slot, view_geom=view_geom, buf_key="group",
                                                     phase="recon", buf_l0=g_idx[0], buf_layers=len(g_idx))
                    self._recon_event(chunk[0]).record(sides[0])
                for l in chunk:
                    self._recon_event_of[l] = chunk[0]
                c0, ci = c0 + n_c, ci + 1
            while c0 < len(layers):
                n_c = sizes[min(ci, len(sizes) - 1)]
                chunk = layers[c0: c0 + n_c]
                side = sides[ci % len(sides)]
                with torch.cuda.stream(side):
                    self._reconstruct_layers_batched([self.deltakv_layer_to_idx[l] for l in chunk], stack, recon_pos,
                                                     recon_latent, recon_out_slot, view_geom=view_geom,
                                                     buf_key=ci % len(sides))
                    self._recon_event(chunk[0]).record(side)
                for l in chunk:
                    self._recon_event_of[l] = chunk[0]
                c0, ci = c0 + n_c, ci + 1
        self._recon_ahead = {l: True for l in layers}
        return True

    @staticmethod
    def _rotated_store_enabled() -> bool:
        """`SVK_DELTAKV_ROTATED_STORE=0`: the per-layer view launch on the walk (the A/B reference)."""
        return os.environ.get("SVK_DELTAKV_ROTATED_STORE", "1") != "0"

    def _rotated_store_row_lens(self, batch: int):
        lens = self.deltakv_layer_batch_states.context_lens
        if (lens is None or lens.dtype != torch.int32 or not lens.is_contiguous() or int(lens.numel()) < batch
                or os.environ.get("SVK_DELTAKV_ROTATED_POS", "lens") != "lens"):
            return None
        return lens

    def _views_rest_ahead(self, layers, view_table, view_geom, sides) -> None:
        """MI355X: a sparse layer's view launch on the walk (`get_layer_compute_view`: 7 us at one row, 20 us at four beside
        the look-ahead - a chain of dependent loads for a few hundred rows) has nothing left to do once (a) the raw rows of
        the view that exist BEFORE the step - sink + buffered tail - are written for every layer of the group by ONE launch
        at the head of the group (while the walk waits for the first reconstruction anyway), and (b) the step's newest row rides in the
        layer's attention launch (`rotated_store`).  Here: (a)."""
        if not self._rotated_store_enabled() or not self.recon_into_view or view_geom is None:
            return
        active_slots, view_lens = view_table
        new_slots = self.deltakv_layer_batch_states.slot_mapping
        B = int(active_slots.shape[0])
        if (new_slots is None or new_slots is not self._deltakv_decode_static_slot_mapping or new_slots.dtype != torch.int32
                or int(new_slots.numel()) != B or get_context().is_prefill):
            return
        l_idxs = [self.deltakv_layer_to_idx[l] for l in layers]
        l0, l1 = int(l_idxs[0]), int(l_idxs[-1]) + 1
        view = self._recon_view_out(l0, l1, view_geom)
        if view is None or l_idxs != list(range(l0, l1)):
            return
        W = int(active_slots.shape[1])
        total = B * W
        k_max = W - int(self.config.num_sink_tokens) - self._deltakv_decode_static_max_buffer()
        knw = self.deltakv_k_norm_weight
        if knw is not None and (knw.dtype != torch.float32 or knw.stride(-1) != 1):
            knw = self.__dict__.get("_k_norm_weight_f32")
            if knw is None or knw[1] != (self.deltakv_k_norm_weight.data_ptr(), self.deltakv_k_norm_weight._version):
                knw = self._k_norm_weight_f32 = (self.deltakv_k_norm_weight.float().contiguous(),
                                                 (self.deltakv_k_norm_weight.data_ptr(), self.deltakv_k_norm_weight._version))
            knw = knw[0]
        # on the walk's OWN stream: it has nothing to do until the group's first reconstruction lands (one launch group,
        # ~50 us), and in front of the reconstructions on the look-ahead stream the launch would delay all of them
        dk.deltakv_materialize_sparse_view(
            active_slots, view_lens, self.deltakv_slot_to_pos, None,
            self.deltakv_full_kv_cache[0, l0:l1], self.deltakv_full_kv_cache[1, l0:l1],
            view[0][:, :total], view[1][:, :total], self.cos_sin_cache,
            k_norm_weight=None if knw is None else knw[l0:l1], k_norm_eps=float(self.deltakv_k_norm_eps),
            temp_slots=self._ensure_decode_static_temp_slots(B, k_max), temp_offset=int(self.config.num_sink_tokens),
            new_slots=new_slots, skip_temp=True, skip_new=True)
        self._view_rest_layers = set(range(l0, l1))

    def _recon_event(self, layer_idx: int):
        ev = self._recon_events.get(layer_idx)
        if ev is None:
            ev = self._recon_events[layer_idx] = torch.cuda.Event()
        return ev

    _RECON_SUB_BATCHES = [2]
    _RECON_SUB_BATCHES_WIDE = [4]           # ... for launches of at least _RECON_WIDE_TOKENS selected tokens (rows x keep)
    _RECON_WIDE_TOKENS = 4096
    _RECON_STREAMS = 1

    @classmethod
    def _recon_stream_count(cls) -> int:
        import os
        return max(1, int(os.environ.get("SVK_DELTAKV_RECON_STREAMS", cls._RECON_STREAMS)))
    _RECON_INTO_VIEW_DEFAULT = True
    _LAYER_VIEWS_MAX_BYTES = 16 << 30

    @classmethod
    def _recon_sub_batches(cls, n_tokens: int = 0) -> list[int]:
        """Layers per look-ahead launch group (the last value repeats).  At one row (2048 selected tokens) two balances the
        side stream against the main stream's per-layer chain (1 / 2 / 3 / 4 layers -> 1.40 / 1.35 / 1.41-1.48 / 1.42 ms per
        256 k step); at four rows (8192 tokens) the launches fill the chip several times over and fewer, larger ones win
        (2 / 3 / 4 / 5 / 6 layers -> 3.36-3.38 / 3.29-3.32 / 3.24-3.32 / 3.26 / 3.26 ms; at two rows 2 / 4 layers -> 2.04 / 2.00 ms)."""
        env = os.environ.get("SVK_DELTAKV_RECON_SUB")          # developer knob: "1,2" = a first launch group of one layer, then twos
        if env:
            return [max(1, int(x)) for x in env.split(",") if x.strip()]
        return list(cls._RECON_SUB_BATCHES_WIDE if int(n_tokens) >= cls._RECON_WIDE_TOKENS else cls._RECON_SUB_BATCHES)

    @classmethod
    def _recon_sub_batch(cls, n_tokens: int = 0) -> int:
        return max(cls._recon_sub_batches(n_tokens))

    def _stacked_up_weights(self):
        """(W1 [Ls, hid, K], b1 [Ls, hid], W2 [Ls, out, hid], b2 [Ls, out]) of the sparse layers' compress_up modules as
        stacked copies (rebuilt after `load_compressor_state`), or None when a layer is not the fused two-Linear form."""
        parts = [self._fused_up_parts(self.compress_up[i], self.deltakv_latent_cache[i]) for i in range(len(self.compress_up))]
        if any(p is None or p[0].bias is None or p[1].bias is None for p in parts) or int(self.config.kv_quant_bits or 0) != 4:
            return None
        # keyed on the parameters' storage and version counters: load_state_dict / .to() / in-place checkpoint loads
        # rebuild the stacked copies, so this path can never run on weights the per-layer path no longer has
        key = tuple((t.data_ptr(), t._version) for p in parts for t in (p[0].weight, p[0].bias, p[1].weight, p[1].bias))
        st = self.__dict__.get("_up_stack")
        if st is None or st[0] != key:
            # the second Linear's bias rides in the GEMM as one more input feature: W2' = [W2 | b2 | 0 ...] against a hidden
            # row [h | 1 | 0 ...] (kRecondPad columns, so K stays a multiple of the library's 64-wide tiles) - baddbmm would
            # first broadcast-copy the bias into the whole [layers, n, out] output (12 us per launch at 2048 tokens)
            w2 = torch.stack([p[1].weight.detach() for p in parts])
            b2 = torch.stack([p[1].bias.detach() for p in parts])
            w2p = torch.zeros((w2.shape[0], w2.shape[1], w2.shape[2] + self._RECON_PAD), dtype=w2.dtype, device=w2.device)
            w2p[:, :, :w2.shape[2]] = w2
            w2p[:, :, w2.shape[2]] = b2
            st = (key, (torch.stack([p[0].weight.detach() for p in parts]).contiguous(), torch.stack([p[0].bias.detach() for p in parts]).contiguous(),
                        w2p.contiguous(), b2.contiguous()))
            self._up_stack = st
        return st[1]

    _RECON_PAD = 64
    _RECON_GEMM_ROWS = 4096

    def _reconstruct_layers_batched(self, l_idxs, stack, recon_pos, recon_latent, recon_out_slot, view_geom=None, buf_key=0,
                                    phase=None, buf_l0=None, buf_layers=0):
        """Residual load + reconstruction of consecutive sparse layers `l_idxs` (same plan) in three launches.
        `phase`: "load" = the residual load only (hidden rows stay in the buffer, layer `buf_l0` at its row 0, room for
        `buf_layers` layers), "recon" = what follows it for layers whose load an earlier call did; None = both."""
        w1, b1, w2, b2 = stack
        l0, l1 = int(l_idxs[0]), int(l_idxs[-1]) + 1
        assert list(l_idxs) == list(range(l0, l1))
        k, n = l1 - l0, int(recon_latent.numel())
        store = self.__dict__.setdefault("_recon_batch_bufs", {})
        cur = store.get((n, buf_key))          # one buffer pair per (token count, side stream), never freed while in use
        hid = int(w1.shape[1])
        off = 0 if buf_l0 is None else l0 - int(buf_l0)
        need = max(k + off, int(buf_layers))
        if cur is None or cur[0].shape[0] < need:
            if len(store) >= 4 * (self._recon_stream_count() + 1):     # a handful of token counts at most (batch compositions come and go)
                torch.cuda.synchronize(self.device)
                store.clear()
            kb = max(need, self._recon_sub_batch(n))
            hbuf = torch.zeros((kb, n, hid + self._RECON_PAD), dtype=torch.bfloat16, device=self.device)
            hbuf[:, :, hid] = 1.0                                   # the bias feature; the kernel below writes [:, :, :hid] only
            cur = [hbuf, None]                                      # (the delta buffer only if the library GEMM runs)
            store[(n, buf_key)] = cur
        hp = cur[0][off: off + k]
        if phase != "recon":
            dk.dequant_linear_act(self.deltakv_latent_cache[l0:l1], self.deltakv_latent_scales[l0:l1], self.deltakv_latent_mins[l0:l1],
                                  self._quant_group_size(), w1[l0:l1], b1[l0:l1], activation="gelu", row_index=recon_latent,
                                  out=hp[:, :, :hid], layers=True)
        if phase == "load":
            return
        knw = self.deltakv_k_norm_weight
        view = self._recon_view_out(l0, l1, view_geom)
        if view is not None:
            self._recon_view_layers.update(range(l0, l1))
        if self._fused_up_recon_mode() != "0" and dk.deltakv_up_reconstruct_supported(
                head_dim=self.head_dim, num_kv_heads=self.num_kv_heads,
                k_fathers=int(self.deltakv_latent_to_full_slots.shape[-1]), hidden_features=hid):
            # MI355X: second Linear + reconstruction in one hand-written MFMA launch, the delta rows stay in LDS
            # (include/svk.h SvkDeltakvUpReconArgs).  Per two layers alone: 30 against 35 us at 2048
            # selected tokens, 98 against 110 us at 4 x 2048; in the step 1.496 against 1.501 ms at one row (inside the
            # noise) and 3.48-3.51 against 3.58-3.59 ms at four.  `SVK_DELTAKV_FUSED_UP=0`: the library GEMM and the
            # reconstruct launch below (what shapes the fused launch does not serve - head_dim 64, more than four
            # fathers - always take)
            dk.deltakv_up_reconstruct_layers(
                hp[:, :, :hid], w2[l0:l1, :, :hid], b2[l0:l1], self.deltakv_latent_to_full_slots[l0:l1], recon_latent,
                self.deltakv_slot_to_pos, recon_out_slot, recon_pos, self.cos_sin_cache, self.deltakv_full_kv_cache[0, l0:l1],
                self.deltakv_full_kv_cache[1, l0:l1], k_norm_weight=None if knw is None else knw[l0:l1].float().contiguous(),
                k_norm_eps=float(self.deltakv_k_norm_eps), view_out=view)
            return
        # (row chunks: this image's hipBLASLt faults inside the batched bf16 GEMM at 8192 rows - plain
        #  torch.bmm([2, 8192, 2112] x [2, 2112, 1024]), tools/probe_bmm2.py; 4096 rows and the per-layer mm are fine)
        if cur[1] is None:
            cur[1] = torch.empty((cur[0].shape[0], n, int(w2.shape[1])), dtype=torch.bfloat16, device=self.device)
        delta = cur[1][:k]
        for c0 in range(0, n, self._RECON_GEMM_ROWS):
            c1 = min(n, c0 + self._RECON_GEMM_ROWS)
            torch.bmm(hp[:, c0:c1], w2[l0:l1].transpose(1, 2), out=delta[:, c0:c1])
        dk.deltakv_reconstruct_writeback_layers(
            delta, self.deltakv_latent_to_full_slots[l0:l1], recon_latent, self.deltakv_slot_to_pos, recon_out_slot, recon_pos,
            self.cos_sin_cache, self.deltakv_full_kv_cache[0, l0:l1], self.deltakv_full_kv_cache[1, l0:l1],
            k_norm_weight=None if knw is None else knw[l0:l1].float().contiguous(), k_norm_eps=float(self.deltakv_k_norm_eps),
            raw_k_cache=True, store_raw_k=False, view_out=view)

    @staticmethod
    def _fused_up_parts(up, cache):
        """(Linear, Linear) of an `mlp_gelu` compress_up the fused dequant+Linear+GELU kernel can serve, else None."""
        import os
        if not isinstance(up, torch.nn.Sequential) or len(up) != 3:
            return None
        lin1, act, lin2 = up[0], up[1], up[2]
        if not (isinstance(lin1, torch.nn.Linear) and isinstance(lin2, torch.nn.Linear) and isinstance(act, torch.nn.GELU)):
            return None
        if getattr(act, "approximate", "none") != "none" or lin1.weight.dtype != torch.bfloat16:
            return None
        k = int(lin1.weight.shape[1])
        if k % 32 != 0 or k > 512 or int(cache.shape[-1]) * 8 != k or not lin1.weight.is_contiguous():
            return None
        return lin1, lin2

    def _set_postrope_slots(self, layer_idx: int, slots: torch.Tensor):
        """deltakv_less_memory_cuda_graph.py:476-500 (-1 entries fall on a dummy slot that no row owns)."""
        mask = self._deltakv_postrope_slot_mask[self.deltakv_layer_to_idx[int(layer_idx)]]
        mask.zero_()
        if slots.numel():
            dummy = self._postrope_dummy_slot()
            safe = torch.where(slots >= 0, slots, dummy.expand_as(slots)).long()
            mask.index_fill_(0, safe, True)

    def _postrope_dummy_slot(self) -> torch.Tensor:
        s = getattr(self, "_deltakv_postrope_dummy_slot", None)
        if s is None:
            s = torch.from_numpy(self._pool_sparse.pop(1)).to(self.device)
            self._deltakv_postrope_dummy_slot = s
        return s

    @torch.no_grad()
    def deltakv_reconstruct(self, layer_idx: int, active_compressed_indices, context_lens, req_indices, chunk_lens=None,
                            return_reconstruct_temp_slots: bool = True):
        """deltakv_less_memory.py:4032-4143 (static decode branch)."""
        del context_lens, chunk_lens, return_reconstruct_temp_slots
        with profiler.record("deltakv_less_memory_reconstruct_total"):
            plan_before = self._deltakv_view_cache_value
            active_slots, local_req, new_context_lens, temp_slots, recon_pos, recon_latent, recon_out_slot = \
                self._deltakv_build_view_and_plan_reconstruct(layer_idx, active_compressed_indices, req_indices)
            fresh_plan = self._deltakv_view_cache_value is not plan_before
            l_idx = self.deltakv_layer_to_idx[layer_idx]
            if fresh_plan:
                self._recon_view_layers = set()
                self._view_rest_layers = set()
            if recon_latent.numel() > 0:
                geom = self._recon_view_geometry(active_slots, recon_latent)
                ahead = self.__dict__.get("_recon_ahead") or {}
                if fresh_plan:
                    ahead = self._recon_ahead = {}
                    if self._recon_lookahead_enabled():
                        self._reconstruct_group_ahead(layer_idx, recon_pos, recon_latent, recon_out_slot, view_geom=geom,
                                                      view_table=(active_slots, new_context_lens))
                        ahead = self._recon_ahead
                if ahead.pop(int(layer_idx), False):
                    owner = self.__dict__.get("_recon_event_of", {}).get(int(layer_idx), int(layer_idx))
                    if owner not in self._recon_joined:
                        torch.cuda.current_stream().wait_event(self._recon_events[owner])
                        self._recon_joined.add(owner)
                else:
                    self._reconstruct_layer(l_idx, recon_pos, recon_latent, recon_out_slot, view_geom=geom)
            # static decode: the post-RoPE slots of this layer are exactly the reconstruct scratch slots the plan put
            # into the view, so the attention view identifies them positionally (no per-layer mask maintenance;
            # `_set_postrope_slots` remains for callers that want the reference's mask)
            return active_slots, local_req, new_context_lens, torch.empty((0,), device=self.device, dtype=torch.int32)

    def _ensure_materialized_sparse_view(self, batch_size: int, width: int):
        """deltakv_less_memory.py:1309-1342: the attention view addresses the compact copy as arange(B*W)."""
        total = int(batch_size) * int(width)
        if total > int(self.deltakv_materialized_compute_num_slots):
            raise RuntimeError("DeltaKV materialized sparse workspace is too small: "
                               f"need={total} capacity={int(self.deltakv_materialized_compute_num_slots)} "
                               f"batch={batch_size} width={width}. Increase max_num_batched_tokens or reduce decode keep tokens.")
        key = (int(batch_size), int(width))
        bufs = self._materialized_view.get(key)
        if bufs is None:
            d = self.device
            bufs = (torch.arange(max(1, total), dtype=torch.int32, device=d).view(max(1, batch_size), max(1, width)),
                    torch.arange(max(1, batch_size), dtype=torch.int32, device=d))
            self._materialized_view[key] = bufs
        return bufs[0][:batch_size, :width], bufs[1][:batch_size]

    def get_layer_compute_view(self, layer_idx: int, active_slots, req_indices, context_lens, selection=None):
        """deltakv_less_memory.py:1344-1402."""
        if active_slots.dim() != 2:
            raise RuntimeError("DeltaKV sparse materialization expects a 2D active slot table, "
                               f"got shape={tuple(active_slots.shape)}.")
        B, W = int(active_slots.shape[0]), int(active_slots.shape[1])
        total = B * W
        local_active, local_req = self._ensure_materialized_sparse_view(B, W)
        l_idx = self.deltakv_layer_to_idx[layer_idx]
        in_view = l_idx in self._recon_view_layers and self._layer_views is not None    # this plan's reconstruction is already there
        if in_view:
            k_out, v_out = self._layer_views[0, l_idx, :total], self._layer_views[1, l_idx, :total]
        else:
            k_out, v_out = self.deltakv_materialized_kv_cache[0, :total], self.deltakv_materialized_kv_cache[1, :total]
        if total == 0:
            return k_out, v_out, local_active, local_req, context_lens
        new_k = new_v = new_slots = None
        pending = self._pending_raw_store.pop(layer_idx, None)
        if pending is not None:
            new_k, new_v = pending
            new_slots = self.deltakv_layer_batch_states.slot_mapping
            if int(new_slots.numel()) != B:
                raise RuntimeError(f"DeltaKV fused raw store: {int(new_slots.numel())} slots for a view of {B} rows")
        self._rotated_store_for.pop(layer_idx, None)
        if in_view and new_slots is not None and l_idx in self._view_rest_layers and l_idx in self._recon_view_layers:
            # every row of this view but the step's newest is (being) written on the look-ahead stream, whose event the
            # caller has waited for; the newest row rides in the attention launch: no launch here
            knw = None if self.deltakv_k_norm_weight is None else self.deltakv_k_norm_weight[l_idx]
            if knw is not None and knw.dtype != torch.float32:
                knw = self._k_norm_weight_f32[0][l_idx]
            self._rotated_store_for[layer_idx] = dict(
                new_kv=(new_k, new_v, new_slots),
                args=dict(raw_k=self.deltakv_full_kv_cache[0, l_idx], raw_v=self.deltakv_full_kv_cache[1, l_idx],
                          slot_to_pos=self.deltakv_slot_to_pos, cos_sin=self.cos_sin_cache, k_norm_weight=knw,
                          k_norm_eps=float(self.deltakv_k_norm_eps),
                          # the step's allocation wrote slot_to_pos[new slot] = row length - 1 (deltakv_base.py:2098-2113)
                          row_lens=self._rotated_store_row_lens(B)))
            return k_out, v_out, local_active, local_req, context_lens
        with profiler.record("deltakv_materialize_sparse_view"):
            k_max = W - int(self.config.num_sink_tokens) - self._deltakv_decode_static_max_buffer()
            dk.deltakv_materialize_sparse_view(
                active_slots, context_lens, self.deltakv_slot_to_pos, None,
                self.deltakv_full_kv_cache[0, l_idx], self.deltakv_full_kv_cache[1, l_idx], k_out, v_out,
                self.cos_sin_cache,
                k_norm_weight=None if self.deltakv_k_norm_weight is None else self.deltakv_k_norm_weight[l_idx],
                k_norm_eps=float(self.deltakv_k_norm_eps),
                temp_slots=self._ensure_decode_static_temp_slots(B, k_max), temp_offset=int(self.config.num_sink_tokens),
                new_k=new_k, new_v=new_v, new_slots=new_slots, skip_temp=in_view)
        return k_out, v_out, local_active, local_req, context_lens

    def build_decode_compute_view(self, layer_idx: int, q: torch.Tensor, selection: SparseSelection, *, num_heads: int,
                                  num_kv_heads: int) -> DecodeComputeView:
        """deltakv_less_memory.py:2942-2995 (KIVI full layers) + deltakv_base.py:975-1018 (sparse layers)."""
        if layer_idx in self.full_layer_to_idx:
            l_idx = self.full_layer_to_idx[layer_idx]
            meta = AttentionViewMeta(active_slots=self.full_layer_slots_map, req_indices=selection.req_indices,
                                     context_lens=selection.context_lens, attn_score=selection.attn_score,
                                     max_context_len=selection.max_context_len)
            if self._full_layer_kivi_enabled():
                return DecodeComputeView(meta=meta, payload=ExplicitKVPayload(
                    k_cache=self.full_kv_cache[0, l_idx], v_cache=self.full_kv_cache[1, l_idx], backend="full_layer_kivi",
                    metadata={"kivi_block_slots_map": self.full_layer_kivi_block_slots_map,
                              "kivi_block_start_pos": self.full_layer_kivi_block_start_pos,
                              "key_packed": self.full_layer_kivi_key_packed[l_idx],
                              "key_scales": self.full_layer_kivi_key_scales[l_idx],
                              "key_mins": self.full_layer_kivi_key_mins[l_idx],
                              "value_packed": self.full_layer_kivi_value_packed[l_idx],
                              "value_scales": self.full_layer_kivi_value_scales[l_idx],
                              "value_mins": self.full_layer_kivi_value_mins[l_idx],
                              "group_size": self._full_layer_kivi_group_size(),
                              "block_n": int(self.config.full_layer_kivi_decode_block_n or 16),
                              "num_warps": int(self.config.full_layer_kivi_decode_num_warps or 2),
                              "num_stages": int(self.config.full_layer_kivi_decode_num_stages or 3)}))
            return DecodeComputeView(meta=meta, payload=ExplicitKVPayload(k_cache=self.full_kv_cache[0, l_idx],
                                                                          v_cache=self.full_kv_cache[1, l_idx]))
        if selection.kind != "deltakv":
            raise RuntimeError(f"sparse DeltaKV layer {layer_idx} needs a 'deltakv' selection, got {selection.kind!r}")
        active_slots, local_req, context_lens, temp_slots = self.deltakv_reconstruct(
            layer_idx=layer_idx, active_compressed_indices=selection.active_compressed_indices,
            context_lens=selection.context_lens, req_indices=selection.req_indices, chunk_lens=selection.chunk_lens,
            return_reconstruct_temp_slots=selection.release_temp_slots)
        k_cache, v_cache, active_slots, req_indices, context_lens = self.get_layer_compute_view(
            layer_idx, active_slots, local_req, context_lens, selection)
        rotated = self._rotated_store_for.pop(layer_idx, None)
        return DecodeComputeView(
            meta=AttentionViewMeta(active_slots=active_slots, req_indices=req_indices, context_lens=context_lens,
                                   attn_score=selection.attn_score, max_context_len=selection.max_context_len,
                                   temp_slots=temp_slots),
            payload=ExplicitKVPayload(k_cache=k_cache, v_cache=v_cache,
                                      **({} if rotated is None else {"metadata": {"rotated_store": rotated}})))

    def release_layer_temp_slots(self, layer_idx: int, temp_slots):
        """Static decode keeps its reconstruct scratch for the life of the graph (deltakv_less_memory.py:449-470)."""
        return None

    # ------------------------------------------------------------------ compression side
    def _deltakv_base_cluster_step(self) -> int:
        """deltakv_base.py:255-260."""
        ratio = float(self.config.cluster_ratio or 0.0)
        if ratio <= 0.0:
            raise ValueError(f"DeltaKV cluster_ratio must be > 0, got {ratio}.")
        return max(1, int(1.0 / max(1e-6, ratio)))

    @staticmethod
    def _metric_l2(kv_states: torch.Tensor, all_centers: torch.Tensor) -> torch.Tensor:
        """deltakv_base.py:2168-2190: ranking score 2*dot(a, b) - ||b||^2, the [N, M] matrix kept in bf16 (library GEMM)."""
        dot = torch.matmul(kv_states, all_centers.transpose(0, 1))
        b_norm = (all_centers * all_centers).sum(dim=1, dtype=torch.float32).to(dot.dtype)
        return dot.mul(2.0).sub_(b_norm.unsqueeze(0))

    @staticmethod
    def _fused_cluster_enabled() -> bool:
        import os
        return os.environ.get("SVK_DELTAKV_FUSED_CLUSTER", "1") != "0"

    def _gather_raw_kv_rows(self, l_idx: int, slots: torch.Tensor) -> torch.Tensor:
        """[n] slots -> [n, 2*Hkv*D] concat(K_raw, V) rows of sparse layer l_idx (deltakv_less_memory.py:2708-2717)."""
        half = self.num_kv_heads * self.head_dim
        idx = slots.long()
        return torch.cat((self.deltakv_full_kv_cache[0, l_idx, idx].reshape(-1, half),
                          self.deltakv_full_kv_cache[1, l_idx, idx].reshape(-1, half)), dim=-1)

    def _cluster_compress(self, l_idx: int, kv_block: torch.Tensor, all_center_slots: torch.Tensor, m0: int,
                          new_center_rel: torch.Tensor):
        """deltakv_less_memory.py:2719-2802: -> (father slots [n, K] int32, base_kv [n, kv_dim] bf16).  The centre rows
        are gathered once for the ranking GEMM; top-k with the causal mask over the block's own centres and the mean of
        the father rows are HIP kernels that read the layer cache directly."""
        k_neighbors = int(self.config.deltakv_k_neighbors)
        m = int(all_center_slots.numel())
        if m == 0:
            raise RuntimeError("DeltaKV less-memory: no available reference centers.")
        k_eff = min(k_neighbors, m)
        if self._fused_cluster_enabled() and kv_block.is_cuda and dk.cluster_l2_topk_supported(
                num_kv_heads=self.num_kv_heads, head_dim=self.head_dim, dtype=kv_block.dtype, rows=int(kv_block.shape[0])):
            # MI355X: ranking product + mask + top-k in one MFMA launch over the layer caches; neither the gathered centre
            # matrix nor the [n, m] scores exist (`SVK_DELTAKV_FUSED_CLUSTER=0`: the library GEMM + svk_cluster_topk below)
            topk = dk.cluster_l2_topk(kv_block, self.deltakv_full_kv_cache[0, l_idx], self.deltakv_full_kv_cache[1, l_idx],
                                      all_center_slots, m0=m0, new_center_rel=new_center_rel, k=k_eff)
        else:
            centers = self._gather_raw_kv_rows(l_idx, all_center_slots)
            scores = self._metric_l2(kv_block, centers)
            topk = dk.cluster_topk(scores, m0=m0, new_center_rel=new_center_rel, k=k_eff)
        base, fathers = dk.gather_mean_fathers(self.deltakv_full_kv_cache[0, l_idx], self.deltakv_full_kv_cache[1, l_idx],
                                               all_center_slots, topk, k_out=k_neighbors)
        return fathers, base

    def _store_residual(self, l_idx: int, latent_slots: torch.Tensor, residual: torch.Tensor):
        """deltakv_less_memory.py:2157-2179: int4 group quantise + pack straight into the latent caches."""
        if latent_slots.numel() != residual.shape[0]:
            raise RuntimeError("DeltaKV less-memory latent residual store shape mismatch: "
                               f"slots={int(latent_slots.numel())}, residual_rows={int(residual.shape[0])}.")
        if int(self.config.kv_quant_bits or 0) == 4:
            dk.triton_quantize_and_pack_2d_int4_grouped(
                residual, self._quant_group_size(),
                out=(self.deltakv_latent_cache[l_idx], self.deltakv_latent_scales[l_idx], self.deltakv_latent_mins[l_idx]),
                dst_rows=latent_slots)
        else:
            self.deltakv_latent_cache[l_idx, latent_slots.long()] = residual.to(self.deltakv_latent_cache.dtype)

    def _deltakv_store_layer_latent(self, *, l_idx: int, latent_slots: torch.Tensor, kv_block: torch.Tensor, base_kv: torch.Tensor):
        """deltakv_less_memory.py:2181-2240 (store_all): residual = down(kv) - down(base), both through one library GEMM
        batch when the block is small (:2232-2235)."""
        down = self.compress_down[l_idx]
        n = int(kv_block.shape[0])
        if n <= 1024:
            enc = down(torch.cat((kv_block, base_kv), dim=0))
            residual = enc[:n]
            residual.sub_(enc[n:])
        else:
            residual = down(kv_block)
            residual.sub_(down(base_kv))
        self._store_residual(l_idx, latent_slots, residual)

    @torch.no_grad()
    def deltakv_evict(self, seqs):
        """deltakv_less_memory.py:3602-3780: when a row's raw tail exceeds `recent`, compress whole multiples of `recent`:
        centres every int(1/cluster_ratio) tokens of the evicted block keep their raw K/V, every evicted token gets a
        latent residual against the mean of its K nearest (L2) causal centres, non-centre raw slots are released."""
        with profiler.record("deltakv_less_memory_evict_total"):
            if not self.deltakv_layer_ids:
                return
            self._join_recon_stream()
            self._flush_pending_raw_stores()
            d = self.device
            sink, recent = int(self.config.num_sink_tokens), int(self.config.num_recent_tokens)
            step = self._deltakv_base_cluster_step()
            for seq in seqs:
                row = self.seq_id_to_row.get(seq.seq_id)
                if row is None:
                    continue
                total_len = int(self.row_seq_lens[row])
                clen = int(self.row_deltakv_compressed_lens[row])
                start = sink + clen
                buffer_len = total_len - start
                if buffer_len <= recent:
                    continue
                evict_len = ((buffer_len - recent) // recent) * recent
                if evict_len <= 0:
                    continue
                end = start + evict_len
                raw_block = self.sparse_layer_raw_slots_map[row, start:end].clone()
                center_rel_np = np.arange(0, evict_len, step, dtype=np.int32)          # range(start, end, step) - start
                center_rel = torch.from_numpy(center_rel_np).to(d)
                new_center_slots = raw_block[center_rel.long()].contiguous()
                existing = self.row_deltakv_center_slots.get(row)
                if existing is None:
                    existing = self.sparse_layer_raw_slots_map[row, :sink].clone()
                all_centers = torch.cat((existing, new_center_slots)).contiguous()
                latent_np = self._pool_latent.pop(evict_len)
                latent_slots = torch.from_numpy(latent_np).to(d)
                self.sparse_layer_latent_slots_map[row, start:end] = latent_slots
                prev = self.row_latent_slots.get(row)
                self.row_latent_slots[row] = latent_np if prev is None else np.concatenate((prev, latent_np))
                for l_idx in range(len(self.deltakv_layer_ids)):
                    kv_block = self._gather_raw_kv_rows(l_idx, raw_block)
                    fathers, base = self._cluster_compress(l_idx, kv_block, all_centers, int(existing.numel()), center_rel)
                    self.deltakv_latent_to_full_slots[l_idx, latent_slots.long()] = fathers
                    self._deltakv_store_layer_latent(l_idx=l_idx, latent_slots=latent_slots, kv_block=kv_block, base_kv=base)
                self.row_deltakv_center_slots[row] = all_centers
                is_center = np.zeros((evict_len,), dtype=bool)
                is_center[center_rel_np] = True
                raw_np = raw_block.cpu().numpy()
                free_np = raw_np[~is_center]
                self._pool_sparse.push(free_np)
                free_gpu = torch.from_numpy(free_np.astype(np.int64)).to(d)
                self.deltakv_slot_to_pos[free_gpu] = -1
                pos_free = torch.from_numpy((np.arange(start, end)[~is_center]).astype(np.int64)).to(d)
                self.sparse_layer_raw_slots_map[row, pos_free] = -1
                self.row_deltakv_compressed_lens[row] += evict_len
                self.row_deltakv_compressed_lens_gpu[row] += evict_len
            self._full_layer_kivi_evict(seqs)
            self._deltakv_reset_view_cache()

    @torch.no_grad()
    def _full_layer_kivi_evict(self, seqs):
        """deltakv_less_memory.py:3495-3558: quantise whole groups of the full layers' raw rows that fell out of the
        residual window into KIVI blocks and release their raw slots."""
        if not self._full_layer_kivi_enabled() or not self.full_layer_ids:
            return
        d = self.device
        G = self._full_layer_kivi_group_size()
        residual_length = int(self.config.full_layer_kivi_residual_length or G)
        sink = int(self.config.num_sink_tokens)
        for seq in seqs:
            row = self.seq_id_to_row.get(seq.seq_id)
            if row is None:
                continue
            total_len = int(self.row_seq_lens[row])
            quant_rel_end = ((max(0, total_len - sink) - residual_length) // G) * G
            if quant_rel_end <= 0:
                continue
            quant_end = sink + quant_rel_end
            quant_start = max(sink, int(self.row_full_layer_kivi_quantized_lens[row] or sink))
            quant_start = sink + (((quant_start - sink) + G - 1) // G) * G
            if quant_end <= quant_start:
                continue
            slots = self.full_layer_slots_map[row, quant_start:quant_end].contiguous()
            num_blocks = (quant_end - quant_start) // G
            blocks_np = self._pool_kivi.pop(num_blocks)
            block_slots = torch.from_numpy(blocks_np).to(d)
            raw = slots.view(num_blocks, G)
            for l_idx in range(len(self.full_layer_ids)):
                dk.kivi_store_blocks(k_cache=self.full_kv_cache[0, l_idx], v_cache=self.full_kv_cache[1, l_idx], raw_slots=raw,
                                     block_slots=block_slots, key_packed=self.full_layer_kivi_key_packed[l_idx],
                                     key_scales=self.full_layer_kivi_key_scales[l_idx],
                                     key_mins=self.full_layer_kivi_key_mins[l_idx],
                                     value_packed=self.full_layer_kivi_value_packed[l_idx],
                                     value_scales=self.full_layer_kivi_value_scales[l_idx],
                                     value_mins=self.full_layer_kivi_value_mins[l_idx], group_size=G)
            self.full_layer_kivi_block_slots_map[row, quant_start:quant_end] = block_slots.repeat_interleave(G)
            self.full_layer_kivi_block_start_pos[block_slots.long()] = torch.arange(quant_start, quant_end, G, dtype=torch.int32, device=d)
            slots_np = slots.cpu().numpy()
            if (slots_np < 0).any():
                raise RuntimeError("Full-layer KIVI expects raw full-layer slots for the quantized block.")
            self._pool_full.push(slots_np)
            self.full_layer_slot_to_pos[slots.long()] = -1
            self.full_layer_slots_map[row, quant_start:quant_end] = -1
            self.row_kivi_blocks.setdefault(row, []).extend(int(x) for x in blocks_np)
            self.row_full_layer_kivi_quantized_lens[row] = quant_end

