"""StreamingLLM / attention-sink cache manager (mirror of engine/cache_manager/streamingllm.py:10-51).

Physical bookkeeping is SnapKV's (per-layer slot table + LIFO free stack, `svk_compact_rows` for the sink + recent
window shift); what differs is scheduling headroom: the window policy is the same on every layer, so the decode
metadata stays layer-uniform, and the last `num_recent_tokens` of a prompt are not counted as prefill work that needs
its own batch budget."""

from __future__ import annotations

from .snapkv import SnapKVCacheManager


class StreamingLLMCacheManager(SnapKVCacheManager):
    def __init__(self, config, parallel_context=None):
        super().__init__(config, parallel_context)
        self._uniform_decode_metadata = True            # same window on all layers, before and after compaction

    def _window(self) -> int:
        return int(self.config.num_recent_tokens)

    def prefill_batched_tokens_margin(self) -> int:
        """Extra batched-token budget the scheduler grants a prefill step of this method."""
        return self._window()

    def remaining_prefill_tokens(self, seq) -> int:
        """Prompt tokens still to be scheduled, less the recent window when more than a window is left."""
        left = int(seq.num_prompt_tokens) - int(seq.num_prefilled_tokens)
        window = self._window()
        return left - window if 0 < window < left else left

    def free_prefix_recent_slots_batch_layers(self, layer_indices, seqs, *, kv_len, num_sink_tokens, num_recent_tokens):
        super().free_prefix_recent_slots_batch_layers(layer_indices, seqs, kv_len=kv_len, num_sink_tokens=num_sink_tokens,
                                                      num_recent_tokens=num_recent_tokens)
        covers_every_layer = bool(layer_indices) and len(layer_indices) == self.num_layers
        if covers_every_layer:
            self._uniform_decode_metadata = True
