"""Dense (vanilla) cache manager: same token-granular slot table, no eviction
(minimal mirror of engine/cache_manager/standard.py `StandardCacheManager`; the radix prefix
cache half of that file is out of scope)."""

from __future__ import annotations

from .snapkv import SnapKVCacheManager


class StandardCacheManager(SnapKVCacheManager):
    def free_part_slots(self, layer_idx, seq, keep_indices, *, keep_indices_sorted=False):
        raise RuntimeError("vanilla attention never evicts KV slots")
