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

    def _device_step_params(self, graph_batch_size: int):
        """MI355X: the decode-time window eviction - keep [0, sink) and the `recent` newest tokens of a row
        that holds 2 x (sink + recent) (sparse_controller.py:1558-1668, snapkv.py:1805-1896) - as the predicated burst of
        the device-resident step: same slot tables, free stacks and lengths as the host-driven steps, bit for bit
        (tests/test_gpu_streamingllm_e2e.py), no per-step upload, no host decision."""
        sink, recent = int(self.config.num_sink_tokens), int(self.config.num_recent_tokens)
        budget = sink + recent
        if recent <= 0 or budget <= 0:
            return None
        return budget, 2 * budget, recent, 1, None, sink

    def _on_device_burst(self, seqs, n_rows: int, n_layers: int, dropped_per_row: int) -> None:
        self._uniform_decode_metadata = True          # the window moved on every layer alike

    def decode_cuda_graph_context_capacity(self, seqs=None, *, requested_context_capacity: int = 0,
                                           current_context_capacity: int = 0) -> tuple[int, bool]:
        """Graph context capacity = the physical decode peak of a sink + recent row, like H2O's hook (h2o.py:241-254):
        a row is compacted back to sink + recent once it holds 2 x (sink + recent) tokens
        (sparse_controller.py:1558-1653), so no decode step attends over more than that.  The reference leaves this
        method on the runner's default (the logical context bucket, decode_cuda_graph.py:172-203), which sizes the
        split-KV grid for 32 k tokens when a row holds 576..1152: on MI355X that is one workgroup per row and
        33.9 us per layer instead of 17 (B=64).  Rows admitted longer than the peak keep their own length."""
        budget = int(self.config.num_sink_tokens) + int(self.config.num_recent_tokens)
        if budget <= 0:
            return max(1, int(self.config.max_model_len)), False
        longest = 0
        for seq in seqs or []:
            row = self.seq_id_to_row[0].get(seq.seq_id)
            if row is not None:
                longest = max(longest, int(self.row_seq_lens[0][row]) + 1)
        return max(1, min(max(2 * budget, longest), int(self.config.max_model_len))), False

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
