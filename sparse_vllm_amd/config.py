"""Hot-path configuration: the knob names, defaults and validation of the reference.

Mirrors, for the methods of this build only,
  configs/groups.py:40-70      SparseMethodConfig defaults
  configs/sparse.py:33-127     _normalize_quest / _normalize_h2o / _normalize_sparse_prefill_score
  configs/runtime_params.py:15-198  public aliases, legacy-name rejection
so a kwargs dict written for `sparsevllm.LLM(model, **kwargs)` configures this path the
same way.  Everything unrelated to the sparse attention path is out of scope.
"""

from __future__ import annotations

from dataclasses import dataclass, field, fields
from types import SimpleNamespace
from typing import Any

from .method_registry import (
    SUPPORTED_SPARSE_METHODS,
    normalize_sparse_method,
    resolve_prefill_schedule_policy,
)

_COMMON_ALIASES = {
    "sink_keep_tokens": "num_sink_tokens",
    "recent_keep_tokens": "num_recent_tokens",
    "full_attention_layers": "full_attn_layers",
    "deltakv_center_ratio": "cluster_ratio",
    "deltakv_latent_dim": "kv_compressed_size",
    "deltakv_latent_quant_bits": "kv_quant_bits",
    "deltakv_latent_quant_group_size": "kv_quant_group_size",
    "engine_prefill_chunk_size": "chunk_prefill_size",
    "deltakv_neighbor_count": "deltakv_k_neighbors",
}

_LEGACY_RUNTIME_KEYS = {
    "model_cls": "sparse_method",
    "vllm_sparse_method": "sparse_method",
    "compressor_path": "deltakv_checkpoint_path",
    "deltakv_path": "deltakv_checkpoint_path",
    "num_top_tokens": "decode_keep_tokens",
    "num_top_tokens_in_prefill": "removed; use decode_keep_tokens",
    "prefill_keep_tokens": "removed; use decode_keep_tokens",
    "num_sink_tokens": "sink_keep_tokens",
    "num_recent_tokens": "recent_keep_tokens",
    "tail_token_size": "recent_keep_tokens",
    "quest_token_budget": ("removed; Sparse-vLLM QuEST derives it from sink_keep_tokens + "
                           "decode_keep_tokens + recent_keep_tokens"),
    "full_attn_layers": "full_attention_layers",
    "k_neighbors": "deltakv_neighbor_count",
    "deltakv_k_neighbors": "deltakv_neighbor_count",
    "seq_chunk_size": "removed; use deltakv_neighbor_count for cluster reference top-k",
    "compressor_token_group_size": "removed; use deltakv_neighbor_count for cluster reference top-k",
    "ref_mode": "removed; cluster_e2e_big always uses cluster-derived references",
    "cluster_ratio": "deltakv_center_ratio",
    "kv_compressed_size": "deltakv_latent_dim",
    "kv_quant_bits": "deltakv_latent_quant_bits",
    "kv_quant_group_size": "deltakv_latent_quant_group_size",
    "chunk_prefill_size": "engine_prefill_chunk_size",
    "model_prefill_chunk_size": "engine_prefill_chunk_size",
    "sparsevllm_prefill_chunk_size": "engine_prefill_chunk_size",
    "chunk_prefill_accel_omnikv": "removed; OmniKV prefill routing is runtime-owned",
    "deltakv_visual_compress_only": "visual_token_prune_only",
    "deltakv_visual_keep_ratio": "visual_token_keep_ratio",
}


def normalize_runtime_params(params: dict[str, Any] | None) -> dict[str, Any]:
    """Public kwargs -> native field names (runtime_params.py:150-198).  Legacy names raise
    ValueError exactly like the reference's API boundary (:126-136)."""
    out = dict(params or {})
    found = sorted(k for k in out if k in _LEGACY_RUNTIME_KEYS)
    if found:
        details = ", ".join(f"`{k}` -> `{_LEGACY_RUNTIME_KEYS[k]}`" for k in found)
        raise ValueError("Legacy runtime parameter names are no longer accepted. "
                         f"Use the new semantic names instead: {details}.")
    if "sparse_method" in out:
        out["vllm_sparse_method"] = normalize_sparse_method(out.pop("sparse_method"))
    if "deltakv_checkpoint_path" in out:
        out["deltakv_path"] = out.pop("deltakv_checkpoint_path")
    for alias, target in _COMMON_ALIASES.items():
        if alias not in out:
            continue
        value = out.pop(alias)
        if target in out and out[target] != value:
            raise ValueError(f"Conflicting runtime parameters: `{alias}`={value!r} maps to `{target}`, "
                             f"but `{target}`={out[target]!r} was also provided.")
        out[target] = value
    v = out.get("decode_keep_tokens")
    if isinstance(v, float) and v <= 1.0:
        raise ValueError(f"Sparse-vLLM `decode_keep_tokens` must be an explicit token count, got ratio-style value {v!r}. "
                         "Convert the ratio using the target context length before running Sparse-vLLM.")
    return out


@dataclass(kw_only=True)
class Config:
    """The subset of the reference `Config` the sparse attention path reads."""

    # model shape (reference: hf_config / runtime_layout)
    num_hidden_layers: int = 28
    num_attention_heads: int = 28
    num_key_value_heads: int = 4
    head_dim: int = 128
    # capacity
    max_model_len: int = 32768
    max_num_seqs_in_gpu: int = 64
    num_kvcache_slots: int = 0                 # 0 -> derived by the cache manager
    chunk_prefill_size: int = 8192
    prefill_schedule_policy: str | None = None
    decode_cuda_graph: bool = False
    validate_runtime_invariants: bool = False
    enable_profiler: bool = False
    device: str = "cuda:0"
    tp_size: int = 1
    # sparse methods (configs/groups.py:44-66)
    vllm_sparse_method: str = ""
    num_sink_tokens: int = 64
    num_recent_tokens: int = 512
    decode_keep_tokens: int = 4096
    quest_chunk_size: int = 16
    quest_token_budget: int = field(default=0, init=False)
    quest_skip_layers: int = 2
    snapkv_window_size: int = 32
    snapkv_num_full_layers: int = 0
    pool_kernel_size: int = 1            # configs/groups.py:129 (max_pool1d over the SnapKV middle scores)
    sparse_prefill_score_mode: str = "probability"
    sparse_attn_score_dtype: str = "float32"
    h2o_decode_budget: int = 4096
    h2o_decode_eviction_interval: int = 128
    h2o_prefill_budget: int = 8192
    h2o_recent_ratio: float = 0.5
    h2o_prefill_score_window: int = 128
    # DeltaKV (configs/groups.py:50-52,104-148; configs/delta.py)
    full_attn_layers: str | list[int] = "0"
    obs_layer_ids: list[int] = field(default=None, init=False)
    rope_theta: float = 1000000.0
    deltakv_path: str | None = None
    allow_missing_deltakv_path: bool = False
    deltakv_k_neighbors: int = 4
    cluster_ratio: float = 0.1
    kv_compressed_size: int = 128
    kv_quant_bits: int = 4
    kv_quant_group_size: int = 0
    full_layer_kv_quant_bits: int = 0
    full_layer_kivi_group_size: int = 32
    full_layer_kivi_residual_length: int = 32
    full_layer_kivi_decode_block_seq: int = 256
    full_layer_kivi_decode_block_n: int = 16
    full_layer_kivi_decode_num_warps: int = 2
    full_layer_kivi_decode_num_stages: int = 3
    enable_full_layer_kivi_quant: bool = True
    use_compression: bool = True
    use_nonlinear_compressor: bool = True
    compressor_intermediate_size: int = 2048
    compressor_linear_bias: bool = True
    compressor_down_type: str = "auto"
    compressor_up_type: str = "auto"
    compressor_down_intermediate_size: int = -1
    compressor_up_intermediate_size: int = -1
    deltakv_full_pool_reserve_ratio: float = 0.1
    deltakv_sparse_decode_backend: str = "custom"
    deltakv_num_latent_slots: int = 0          # 0 -> derived by the cache manager
    deltakv_num_full_layer_slots: int = 0
    deltakv_num_kivi_blocks: int = 0

    def __post_init__(self):
        self.vllm_sparse_method = normalize_sparse_method(self.vllm_sparse_method)
        if self.vllm_sparse_method not in SUPPORTED_SPARSE_METHODS:
            supported = ", ".join(repr(m) for m in sorted(SUPPORTED_SPARSE_METHODS) if m)
            raise ValueError(f"Unsupported vllm_sparse_method={self.vllm_sparse_method!r}. "
                             f"Supported methods: '', {supported}.")
        self.prefill_schedule_policy = resolve_prefill_schedule_policy(self.vllm_sparse_method,
                                                                       self.prefill_schedule_policy)
        self._normalize_quest()
        self._normalize_sparse_prefill_score()
        if self.vllm_sparse_method == "h2o":
            self._normalize_h2o()
        self._normalize_sparse_layout()
        if self.vllm_sparse_method == "deltakv":
            self._normalize_deltakv()
        if self.num_attention_heads % self.num_key_value_heads:
            raise ValueError("num_attention_heads must be divisible by num_key_value_heads")
        if self.num_key_value_heads % self.tp_size or self.num_attention_heads % self.tp_size:
            raise ValueError(f"attention heads ({self.num_attention_heads}/{self.num_key_value_heads}) must be "
                             f"divisible by tp_size={self.tp_size}")

    # configs/sparse.py:33-60
    def _normalize_quest(self):
        if self.quest_chunk_size <= 0:
            raise ValueError("quest_chunk_size 必须 > 0")
        self.quest_token_budget = 0
        if self.vllm_sparse_method == "quest":
            for name in ("num_sink_tokens", "decode_keep_tokens", "num_recent_tokens"):
                value = getattr(self, name)
                if isinstance(value, bool) or not isinstance(value, int) or value < 0:
                    raise ValueError(f"QuEST {name} must be a non-negative integer, got {value!r}.")
            self.quest_token_budget = self.num_sink_tokens + self.decode_keep_tokens + self.num_recent_tokens
            if self.quest_token_budget <= 0:
                raise ValueError("QuEST derived token budget must be > 0: num_sink_tokens + decode_keep_tokens + "
                                 f"num_recent_tokens = {self.quest_token_budget}.")
        if self.quest_skip_layers < 0:
            raise ValueError("quest_skip_layers 不能 < 0")

    # configs/sparse.py:62-98
    def _normalize_h2o(self):
        for name in ("h2o_decode_budget", "h2o_decode_eviction_interval"):
            v = getattr(self, name)
            if isinstance(v, bool) or int(v) != v or int(v) <= 0:
                raise ValueError(f"{name} must be a positive integer, got {v!r}.")
            setattr(self, name, int(v))
        self.h2o_prefill_budget = int(self.h2o_prefill_budget)
        if self.h2o_prefill_budget < self.h2o_decode_budget:
            raise ValueError("h2o_prefill_budget must be >= h2o_decode_budget, "
                             f"got prefill={self.h2o_prefill_budget} decode={self.h2o_decode_budget}.")
        decode_peak = self.h2o_decode_budget + self.h2o_decode_eviction_interval
        if decode_peak % 64 != 0:
            raise ValueError("h2o_decode_budget + h2o_decode_eviction_interval must be divisible by 64 for the "
                             f"scored decode kernel, got {self.h2o_decode_budget} + "
                             f"{self.h2o_decode_eviction_interval} = {decode_peak}.")
        self.h2o_recent_ratio = float(self.h2o_recent_ratio)
        if not 0.0 < self.h2o_recent_ratio < 1.0:
            raise ValueError(f"h2o_recent_ratio must be in (0, 1), got {self.h2o_recent_ratio}.")
        self.h2o_prefill_score_window = int(self.h2o_prefill_score_window)
        if self.sparse_prefill_score_mode == "logits":
            if self.h2o_prefill_score_window < 0:
                raise ValueError("h2o_prefill_score_window must be non-negative in logits mode (0 means the full "
                                 f"chunk), got {self.h2o_prefill_score_window}.")
        elif not 1 <= self.h2o_prefill_score_window <= 128:
            raise ValueError("h2o_prefill_score_window must be in [1, 128] because the prefill score kernel supports "
                             f"at most 128 query tokens, got {self.h2o_prefill_score_window}.")

    # configs/sparse.py:33-37, :240-257 (every layer of this build's models has a KV cache)
    def _normalize_sparse_layout(self):
        if isinstance(self.full_attn_layers, str):
            layers = self.full_attn_layers.strip()
            self.full_attn_layers = [] if not layers else [int(x) for x in layers.split(",")]
        self.full_attn_layers = [int(x) for x in self.full_attn_layers]
        unknown = sorted(l for l in self.full_attn_layers if not 0 <= l < self.num_hidden_layers)
        if unknown and self.vllm_sparse_method in {"omnikv", "deltakv"}:
            raise ValueError("full_attn_layers must contain KV/full-attention layer indices for "
                             f"{self.vllm_sparse_method}; non-KV layers={unknown}.")
        configured = set(self.full_attn_layers)
        self.obs_layer_ids = [l for l in self.full_attn_layers
                              if 0 <= l and l + 1 < self.num_hidden_layers and (l + 1) not in configured]

    # configs/delta.py:43-160 (normalize_deltakv_storage + validate_deltakv_runtime, slim runtime rules)
    def _normalize_deltakv(self):
        for attr in ("compressor_down_type", "compressor_up_type"):
            v = getattr(self, attr)
            v = "auto" if v is None else str(v).strip().lower()
            setattr(self, attr, v or "auto")
        if not self.use_compression:
            raise ValueError("DeltaKV runtime is compressor-only; set use_compression=True.")
        if self.deltakv_path is None and not self.allow_missing_deltakv_path:
            raise ValueError("DeltaKV requires deltakv_path for compressor sparse layers. "
                             "Set allow_missing_deltakv_path=True only for construction-only tests.")
        self.kv_quant_bits = int(self.kv_quant_bits or 0)
        if self.kv_quant_bits not in (0, 4):
            raise ValueError("DeltaKV slim runtime supports sparse compressor residual bits 0 or 4 only, "
                             f"got kv_quant_bits={self.kv_quant_bits}.")
        self.full_layer_kv_quant_bits = int(self.full_layer_kv_quant_bits or 0)
        if self.full_layer_kv_quant_bits not in (0, 4):
            raise ValueError("DeltaKV slim runtime supports full-layer storage bits 0 or 4 only, "
                             f"got full_layer_kv_quant_bits={self.full_layer_kv_quant_bits}.")
        self.kv_quant_group_size = int(self.kv_quant_group_size or 0)
        if self.kv_quant_group_size < 0:
            raise ValueError(f"kv_quant_group_size must be >= 0, got {self.kv_quant_group_size}.")
        self.full_layer_kivi_group_size = int(self.full_layer_kivi_group_size or 32)
        if self.full_layer_kivi_group_size <= 0:
            raise ValueError(f"full_layer_kivi_group_size must be > 0, got {self.full_layer_kivi_group_size}.")
        self.full_layer_kivi_residual_length = int(self.full_layer_kivi_residual_length or self.full_layer_kivi_group_size)
        if self.full_layer_kivi_residual_length <= 0:
            raise ValueError(f"full_layer_kivi_residual_length must be > 0, got {self.full_layer_kivi_residual_length}.")
        bs = int(self.full_layer_kivi_decode_block_seq or 256)
        if bs <= 0 or bs % 16:
            raise ValueError(f"full_layer_kivi_decode_block_seq must be a positive multiple of 16, got {bs}.")
        self.full_layer_kivi_decode_block_seq = bs
        if not 0.0 <= float(self.deltakv_full_pool_reserve_ratio) < 1.0:
            raise ValueError("deltakv_full_pool_reserve_ratio must be in [0, 1), "
                             f"got {self.deltakv_full_pool_reserve_ratio}.")
        if int(self.deltakv_k_neighbors) <= 0:
            raise ValueError(f"deltakv_k_neighbors must be > 0, got {self.deltakv_k_neighbors}.")
        backend = str(self.deltakv_sparse_decode_backend or "auto").strip().lower()
        if backend not in {"auto", "custom", "fa2"}:
            raise ValueError("deltakv_sparse_decode_backend must be one of 'auto', 'custom', or 'fa2', "
                             f"got {self.deltakv_sparse_decode_backend!r}.")
        if backend == "fa2":
            raise ValueError("deltakv_sparse_decode_backend='fa2' requires the flash_attn package; "
                             "use 'custom' or leave it as 'auto' when flash_attn is not installed.")
        self.deltakv_sparse_decode_backend = "custom"
        if not self.full_attn_layers:
            raise ValueError("DeltaKV needs at least one full-attention (observation) layer in full_attn_layers.")

    # configs/sparse.py:101-127
    def _normalize_sparse_prefill_score(self):
        mode = str(self.sparse_prefill_score_mode).strip().lower()
        allowed = {"probability", "logits"}
        if mode not in allowed:
            raise ValueError(f"sparse_prefill_score_mode must be one of {sorted(allowed)}, got "
                             f"{self.sparse_prefill_score_mode!r}.")
        if mode != "probability" and self.vllm_sparse_method not in {"snapkv", "pyramidkv", "h2o"}:
            raise ValueError("sparse_prefill_score_mode='logits' only applies to SnapKV/PyramidKV/H2O, got "
                             f"method={self.vllm_sparse_method!r}.")
        if mode == "logits" and self.sparse_attn_score_dtype != "float32":
            raise ValueError("sparse_prefill_score_mode='logits' requires sparse_attn_score_dtype='float32', got "
                             f"{self.sparse_attn_score_dtype!r}.")
        self.sparse_prefill_score_mode = mode

    @property
    def hf_config(self):
        return SimpleNamespace(num_hidden_layers=self.num_hidden_layers, num_attention_heads=self.num_attention_heads,
                               num_key_value_heads=self.num_key_value_heads, head_dim=self.head_dim,
                               rope_theta=self.rope_theta, torch_dtype="bfloat16")

    @classmethod
    def from_kwargs(cls, **kwargs) -> "Config":
        """`LLM(model, **kwargs)`-style construction: aliases applied, unknown keys fatal
        (engine/llm_engine.py:221-232)."""
        native = normalize_runtime_params(kwargs)
        known = {f.name for f in fields(cls) if f.init}
        unknown = sorted(k for k in native if k not in known)
        if unknown:
            raise ValueError(f"Unknown config keys: {unknown}")
        return cls(**native)
