"""Sparse-method names, aliases and prefill policies of the hot path.

Host-side mirror of the reference's `sparsevllm/method_registry.py` (aliases :19-35,
canonical names :37-48, policy table :211-237, `normalize_sparse_method` :240-244,
policy resolution :317-348) so that `sparse_method=` strings written for the reference
select the same method here.  Methods outside this build's scope (SURVEY.md section 8:
omnikv, rkv, skipkv, pyramidkv) are *known names* that `CacheManager.create` refuses.
"""

from __future__ import annotations

from dataclasses import dataclass
from enum import Enum, auto

PREFILL_POLICY_ALL_CHUNKED = "all_chunked"
PREFILL_POLICY_LONG_BS1FULL_SHORT_BATCH = "long_bs1full_short_batch"
PREFILL_POLICY_AUTO = "auto"

SUPPORTED_PREFILL_POLICIES = {PREFILL_POLICY_ALL_CHUNKED, PREFILL_POLICY_LONG_BS1FULL_SHORT_BATCH}

METHOD_ALIASES = {
    None: "",
    "": "",
    "vanilla": "",
    "attention-sink": "streamingllm",
    "attention_sink": "streamingllm",
    "r-kv": "rkv",
    "r_kv": "rkv",
    "skip-kv": "skipkv",
    "skip_kv": "skipkv",
    "deltakv-less-memory": "deltakv",
    "deltakv_less_memory": "deltakv",
    "deltakv-less-memory-cudagraph": "deltakv",
    "deltakv_less_memory_cudagraph": "deltakv",
}

CANONICAL_SPARSE_METHODS = {"", "streamingllm", "snapkv", "h2o", "pyramidkv", "omnikv", "quest", "rkv", "skipkv",
                            "deltakv"}

# what this build provides natively on MI355X (the north-star path)
NATIVE_SPARSE_METHODS = {"", "streamingllm", "snapkv", "h2o", "quest", "deltakv"}

SUPPORTED_SPARSE_METHODS = set(CANONICAL_SPARSE_METHODS)
SUPPORTED_SPARSE_METHOD_ALIASES = {str(k) for k in METHOD_ALIASES if k is not None and str(k)}

# method_registry.py:53-63
PREFIX_CACHE_SUPPORTED_METHODS = {"", "streamingllm", "omnikv", "quest", "snapkv", "h2o", "pyramidkv", "rkv", "skipkv"}

# method_registry.py:196-209
DECODE_CUDA_GRAPH_SUPPORTED_METHODS = set(CANONICAL_SPARSE_METHODS)
TP_DECODE_CUDA_GRAPH_SUPPORTED_METHODS = {"", "streamingllm", "snapkv", "h2o", "pyramidkv", "omnikv", "quest", "rkv",
                                          "skipkv"}


class ParallelMode(str, Enum):
    """distributed/topology.py:7-9 (only the dense-model mode exists in this build)."""
    STANDARD = "standard"
    OUTER_TP_MOE = "outer_tp_moe_tp_ep"


@dataclass(frozen=True)
class ParallelTopology:
    """distributed/topology.py:12-73, the fields `validate_model_runtime_compatibility` reads."""
    tensor_parallel_size: int = 1
    expert_parallel_size: int = 1
    data_parallel_size: int = 1
    mode: ParallelMode = ParallelMode.STANDARD

    def __post_init__(self):
        for name in ("tensor_parallel_size", "expert_parallel_size", "data_parallel_size"):
            object.__setattr__(self, name, int(getattr(self, name)))
        object.__setattr__(self, "mode", ParallelMode(self.mode))
        sizes = (self.tensor_parallel_size, self.expert_parallel_size, self.data_parallel_size)
        if any(n <= 0 for n in sizes):
            raise ValueError(f"Parallel sizes must be positive, got TP={sizes[0]}, EP={sizes[1]}, DP={sizes[2]}.")


class AttentionScoreKind(Enum):
    """operators/attention_capabilities.py:12-15."""
    NONE = auto()
    RAW_QK_PER_HEAD = auto()
    RAW_QK_REDUCED = auto()


class PrefillScoreCollectionKind(Enum):
    """method_registry.py:83-85."""
    NONE = auto()
    METHOD_OWNED_POSTHOC_REDUCED = auto()


@dataclass(frozen=True)
class SparsePrefillAttentionContract:
    """method_registry.py:88-91."""
    main_score_kind: AttentionScoreKind
    score_collection: PrefillScoreCollectionKind


@dataclass(frozen=True)
class ModelRuntimeCompatibility:
    """method_registry.py:76-80."""
    sparse_methods: frozenset
    prefix_cache_methods: frozenset
    decode_cuda_graph_methods: frozenset = frozenset()


# method_registry.py:119-123, :181-194 - the dense model families; the MoE / Gemma rows belong to model families outside
# this build and resolve to NotImplementedError like any unknown model type
DENSE_MODEL_COMPATIBILITY = ModelRuntimeCompatibility(
    sparse_methods=frozenset(CANONICAL_SPARSE_METHODS),
    prefix_cache_methods=frozenset(PREFIX_CACHE_SUPPORTED_METHODS),
    decode_cuda_graph_methods=frozenset(CANONICAL_SPARSE_METHODS))

MODEL_RUNTIME_COMPATIBILITY = {(m, ParallelMode.STANDARD): DENSE_MODEL_COMPATIBILITY
                               for m in ("qwen2", "qwen3", "qwen3_5", "llama")}

_DEFAULT_PREFILL_POLICY_BY_METHOD = {
    "": PREFILL_POLICY_ALL_CHUNKED,
    "streamingllm": PREFILL_POLICY_ALL_CHUNKED,
    "snapkv": PREFILL_POLICY_ALL_CHUNKED,
    "h2o": PREFILL_POLICY_ALL_CHUNKED,
    "pyramidkv": PREFILL_POLICY_LONG_BS1FULL_SHORT_BATCH,
    "omnikv": PREFILL_POLICY_ALL_CHUNKED,
    "quest": PREFILL_POLICY_ALL_CHUNKED,
    "rkv": PREFILL_POLICY_ALL_CHUNKED,
    "skipkv": PREFILL_POLICY_ALL_CHUNKED,
    "deltakv": PREFILL_POLICY_LONG_BS1FULL_SHORT_BATCH,
}

PREFILL_POLICY_BY_METHOD = {
    **_DEFAULT_PREFILL_POLICY_BY_METHOD,
    **{alias: _DEFAULT_PREFILL_POLICY_BY_METHOD[target]
       for alias, target in METHOD_ALIASES.items() if alias},
}

_PREFILL_POSTHOC_SCORE_METHODS = frozenset({"snapkv", "pyramidkv", "h2o", "rkv"})


def normalize_sparse_method(method: str | None) -> str:
    if method is None:
        return ""
    normalized = str(method).strip().lower()
    return METHOD_ALIASES.get(normalized, normalized)


def is_deltakv_method(method: str | None) -> bool:
    return normalize_sparse_method(method) == "deltakv"


def is_decode_cuda_graph_supported(method: str | None) -> bool:
    return normalize_sparse_method(method) in DECODE_CUDA_GRAPH_SUPPORTED_METHODS


def is_tp_decode_cuda_graph_supported(method: str | None) -> bool:
    return normalize_sparse_method(method) in TP_DECODE_CUDA_GRAPH_SUPPORTED_METHODS


def sparse_prefill_attention_contract(method: str | None) -> SparsePrefillAttentionContract:
    """method_registry.py:99-113: the prefill attention itself never returns scores (`main_score_kind` NONE); the
    SnapKV family / H2O / R-KV collect their own reduced scores after it."""
    normalized = normalize_sparse_method(method)
    if normalized not in CANONICAL_SPARSE_METHODS:
        raise ValueError(f"Unknown sparse method {normalized!r}.")
    collection = (PrefillScoreCollectionKind.METHOD_OWNED_POSTHOC_REDUCED
                  if normalized in _PREFILL_POSTHOC_SCORE_METHODS else PrefillScoreCollectionKind.NONE)
    return SparsePrefillAttentionContract(main_score_kind=AttentionScoreKind.NONE, score_collection=collection)


def needs_prefill_posthoc_scores(method: str | None) -> bool:
    return (sparse_prefill_attention_contract(method).score_collection
            is PrefillScoreCollectionKind.METHOD_OWNED_POSTHOC_REDUCED)


def _method_list(methods) -> str:
    return ", ".join("'vanilla'" if m == "" else repr(m) for m in sorted(methods))


def validate_model_runtime_compatibility(*, model_type: str, sparse_method: str | None, topology: ParallelTopology,
                                         decode_cuda_graph: bool, enable_prefix_caching: bool) -> ModelRuntimeCompatibility:
    """method_registry.py:271-314 (check order and messages kept: graph, then method, then prefix cache)."""
    model_type = str(model_type or "").strip().lower()
    method = normalize_sparse_method(sparse_method)
    compatibility = MODEL_RUNTIME_COMPATIBILITY.get((model_type, topology.mode))
    if compatibility is None:
        raise NotImplementedError(f"Unsupported Sparse-vLLM model_type={model_type!r} with "
                                  f"parallel mode={topology.mode.value!r}.")
    if bool(decode_cuda_graph) and method not in compatibility.decode_cuda_graph_methods:
        raise ValueError(f"{model_type} v1 decode_cuda_graph is validated only for "
                         f"{_method_list(compatibility.decode_cuda_graph_methods)}; got method={method!r}.")
    if method not in compatibility.sparse_methods:
        raise ValueError(f"Unsupported {model_type} {topology.mode.value} sparse method {method!r}; "
                         f"validated methods: {_method_list(compatibility.sparse_methods)}.")
    if bool(enable_prefix_caching) and method not in compatibility.prefix_cache_methods:
        raise ValueError(f"{model_type} prefix caching is validated only for "
                         f"{_method_list(compatibility.prefix_cache_methods)}; got method={method!r}.")
    return compatibility


def get_default_prefill_schedule_policy(method: str | None) -> str:
    normalized = normalize_sparse_method(method)
    if normalized not in _DEFAULT_PREFILL_POLICY_BY_METHOD:
        supported = ", ".join(repr(name) for name in sorted(CANONICAL_SPARSE_METHODS) if name)
        aliases = ", ".join(repr(name) for name in sorted(SUPPORTED_SPARSE_METHOD_ALIASES))
        raise ValueError(
            f"Unsupported vllm_sparse_method={method!r}. Supported methods: '', {supported}. "
            f"Supported aliases: {aliases}.")
    return _DEFAULT_PREFILL_POLICY_BY_METHOD[normalized]


def resolve_prefill_schedule_policy(method: str | None, policy: str | None) -> str:
    default_policy = get_default_prefill_schedule_policy(method)
    if policy is None:
        return default_policy
    requested = str(policy).strip().lower()
    if requested in {"", PREFILL_POLICY_AUTO}:
        return default_policy
    if requested not in SUPPORTED_PREFILL_POLICIES:
        supported = ", ".join(repr(name) for name in sorted(SUPPORTED_PREFILL_POLICIES))
        raise ValueError(
            f"Unsupported prefill_schedule_policy={policy!r}. Supported policies: {supported}, "
            f"or {PREFILL_POLICY_AUTO!r}.")
    if requested != default_policy:
        raise ValueError(
            "prefill_schedule_policy must match the registry default for reproducibility. "
            f"method={normalize_sparse_method(method)!r} requested={requested!r} default={default_policy!r}.")
    return requested
