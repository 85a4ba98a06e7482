"""StreamingLLM / attention-sink cache manager (mirror of engine/cache_manager/streamingllm.py:10-51):
SnapKV slot bookkeeping + the fixed sink/recent window scheduling margins."""

from __future__ import annotations

from .snapkv import SnapKVCacheManager


class StreamingLLMCacheManager(SnapKVCacheManager):
    def __init__(self, config, parallel_context=None):
        super().__init__(config, parallel_context)
        self._uniform_decode_metadata = True

    def prefill_batched_tokens_margin(self) -> int:
        return int(self.config.num_recent_tokens)

    def remaining_prefill_tokens(self, seq) -> int:
        remaining = int(seq.num_prompt_tokens - seq.num_prefilled_tokens)
        recent = int(self.config.num_recent_tokens)
        if recent > 0 and remaining > recent:
            return remaining - recent
        return remaining

    def free_prefix_recent_slots_batch_layers(self, layer_indices, seqs, *, kv_len, num_sink_tokens, num_recent_tokens):
        super().free_prefix_recent_slots_batch_layers(layer_indices, seqs, kv_len=kv_len,
                                                      num_sink_tokens=num_sink_tokens,
                                                      num_recent_tokens=num_recent_tokens)
        if layer_indices and len(layer_indices) == self.num_layers:
            self._uniform_decode_metadata = True
