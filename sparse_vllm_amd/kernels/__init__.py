"""Python wrappers with the reference's kernel names/argument order
(src/sparsevllm/kernels/triton/*), each a thin call into libsvk.so."""

from .gqa_flash_decoding_stage1 import flash_decode_stage1, flash_decode_stage1_with_score  # noqa: F401
from .flash_decoding_stage2 import flash_decode_stage2  # noqa: F401
from .store_kvcache import store_kvcache  # noqa: F401
