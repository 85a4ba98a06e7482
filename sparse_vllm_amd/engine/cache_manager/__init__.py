from .base import (  # noqa: F401
    AttentionViewMeta,
    CacheManager,
    DecodeComputeView,
    ExplicitKVPayload,
    LayerBatchStates,
    PrefillComputeView,
    SparseSelection,
)
