"""Wall-clock section profiler with the reference's hot-path labels (utils/profiler.py:10-89):
cache_prepare_decode, decode_attention_stage1_{full,sparse}, decode_attention_stage2_*,
h2o_decode_eviction, h2o_decode_score_update, h2o_decode_compact_total, h2o_decode_burst_select,
h2o_decode_burst_compact_layers, streamingllm_decode_eviction, quest_build_decode_view_static ..."""

from __future__ import annotations

import os
import time
from collections import defaultdict
from contextlib import contextmanager

import torch


class Profiler:
    def __init__(self):
        self.enabled = False
        self.totals = defaultdict(float)
        self.counts = defaultdict(int)

    @contextmanager
    def record(self, name: str):
        if not self.enabled or (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
            yield
            return
        sync = os.environ.get("SPARSEVLLM_SYNC_DEVICE", "0") == "1" and torch.cuda.is_available()
        if sync:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        try:
            yield
        finally:
            if sync:
                torch.cuda.synchronize()
            self.totals[name] += time.perf_counter() - t0
            self.counts[name] += 1

    def summary(self) -> dict:
        return {k: {"total_s": v, "calls": self.counts[k]} for k, v in sorted(self.totals.items())}

    def reset(self):
        self.totals.clear()
        self.counts.clear()


profiler = Profiler()
