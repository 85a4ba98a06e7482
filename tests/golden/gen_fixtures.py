#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE itself (build container only).

Usage (from the repo root, in the build container where /root/reference exists):

    python tests/golden/gen_fixtures.py [group ...]

The reference is imported from /root/reference/src with a stand-in `loguru`
logger (its only missing dependency on this path; the shim is written to a temp
dir and contains no reference code).  Torch-level functions run natively on CPU;
Triton kernels run under TRITON_INTERPRET=1 on float32 tensors that hold
bf16-representable values (bf16 tensors are unsupported by the interpreter).

Only the resulting *data* (inputs + expected outputs, .npz) is committed under
tests/golden/.  Nothing of the reference travels to the GPU box.
"""

from __future__ import annotations

import os
import sys
import tempfile
from types import SimpleNamespace

HERE = os.path.dirname(os.path.abspath(__file__))

_SHIM = '''
import sys
class _L:
    def debug(self, *a, **k): pass
    def info(self, *a, **k): pass
    def warning(self, m, *a, **k): pass
    def error(self, m, *a, **k): sys.stderr.write(f"[E] {m}\\n")
    exception = error
    def remove(self, *a, **k): pass
    def add(self, *a, **k): return 0
    def bind(self, **k): return self
    def opt(self, **k): return self
    def log(self, *a, **k): pass
logger = _L()
'''


def _bootstrap():
    shim_dir = tempfile.mkdtemp(prefix="svk_shim_")
    os.makedirs(os.path.join(shim_dir, "loguru"))
    with open(os.path.join(shim_dir, "loguru", "__init__.py"), "w") as f:
        f.write(_SHIM)
    os.environ.setdefault("SPARSEVLLM_PLATFORM", "cpu")
    os.environ.setdefault("TRITON_INTERPRET", "1")
    sys.dont_write_bytecode = True
    sys.path[:0] = ["/root/reference/src", "/root/reference", shim_dir]


_bootstrap()

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.set_num_threads(4)


def bf16f(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


def bits(t: torch.Tensor) -> np.ndarray:
    """float32 tensor holding bf16 values -> uint16 bit patterns (compact storage)."""
    return (t.contiguous().view(torch.int32).numpy().view(np.uint32) >> 16).astype(np.uint16)


def save(name: str, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


# ----------------------------------------------------------------------------------
# decode attention (Triton interpreter)
# ----------------------------------------------------------------------------------

def gen_decode():
    from sparsevllm.kernels.triton.gqa_flash_decoding_stage1 import (
        flash_decode_stage1, flash_decode_stage1_with_score)
    from sparsevllm.kernels.triton.flash_decoding_stage2 import flash_decode_stage2

    out = {}
    cases = [
        # name, B, Hq, Hkv, D, slots, lens, rows, maxlen_table, block_seq, score mode
        ("a", 2, 28, 4, 128, 224, (150, 37), (1, 3), 160, 64, "2d"),
        ("b", 3, 8, 2, 64, 512, (200, 129, 1), (0, 2, 1), 256, 32, "3d"),
        ("c", 2, 28, 4, 128, 272, (97, 160), (2, 0), 160, 256, "none"),
        ("d", 2, 14, 2, 64, 300, (64, 48), (1, 0), 128, 16, "2d"),
    ]
    g = torch.Generator().manual_seed(20260625)
    for (name, B, Hq, Hkv, D, slots, lens, rows, cap, block_seq, mode) in cases:
        q = bf16f(torch.randn(B, Hq, D, generator=g) * 0.5)
        k = bf16f(torch.randn(slots, Hkv, D, generator=g) * 0.5)
        v = bf16f(torch.randn(slots, Hkv, D, generator=g) * 0.5)
        req = torch.zeros(max(rows) + 1, cap, dtype=torch.int32)
        perm = torch.randperm(slots, generator=g).to(torch.int32)
        off = 0
        for b in range(B):
            req[rows[b], : lens[b]] = perm[off: off + lens[b]]
            off += lens[b]
        bidx = torch.tensor(rows, dtype=torch.int32)
        blen = torch.tensor(lens, dtype=torch.int32)
        max_len = max(lens)
        nblk = (max_len + block_seq - 1) // block_seq
        mid = torch.zeros(B, Hq, nblk, D)
        lse = torch.zeros(B, Hq, nblk)
        if mode == "2d":
            score = torch.full((B, max_len), -1e20)
            flash_decode_stage1_with_score(q, k, v, req, bidx, blen, max_len, mid, lse, score, block_seq)
        elif mode == "3d":
            score = torch.full((B, Hq, max_len), -1e20)
            flash_decode_stage1_with_score(q, k, v, req, bidx, blen, max_len, mid, lse, score, block_seq)
        else:
            score = torch.zeros(1)
            flash_decode_stage1(q, k, v, req, bidx, blen, max_len, mid, lse, block_seq)
        o = torch.empty_like(q)
        flash_decode_stage2(mid, lse, blen, o, block_seq)
        out.update({
            f"{name}_q": bits(q), f"{name}_k": bits(k), f"{name}_v": bits(v),
            f"{name}_req": req.numpy(), f"{name}_bidx": bidx.numpy(), f"{name}_blen": blen.numpy(),
            f"{name}_meta": np.array([max_len, block_seq, {"none": 0, "2d": 2, "3d": 3}[mode]], dtype=np.int64),
            f"{name}_mid_o": mid.numpy(), f"{name}_mid_lse": lse.numpy(),
            f"{name}_score": score.numpy(), f"{name}_o": o.numpy(),
        })
    save("decode_attention", **out)


# ----------------------------------------------------------------------------------
# H2O selection / scores (torch on CPU)
# ----------------------------------------------------------------------------------

def gen_h2o_select():
    from sparsevllm.engine.cache_manager.h2o import H2OCacheManager as M

    out = {}
    g = torch.Generator().manual_seed(7)
    # known answer from the reference's own unit test (tests/test_h2o_cache_manager.py:290-293)
    ka = torch.tensor([1.0, 9.0, 2.0, 8.0, 3.0, 0.0, 0.0, 0.0])
    out["ka_scores"] = ka.numpy()
    out["ka_keep"] = M.select_h2o_indices(ka, budget=4, recent_ratio=0.5).numpy()
    cases = [
        ("u", torch.rand(6, 300, generator=g), 64, 0.5),
        ("ties", torch.randint(0, 7, (5, 257), generator=g).float() / 4.0, 100, 0.5),
        ("short", torch.rand(3, 40, generator=g), 64, 0.5),
        ("ratio", torch.rand(4, 999, generator=g), 333, 0.1),
        ("zeros", torch.cat([torch.zeros(2, 50), -torch.zeros(2, 50), torch.rand(2, 100, generator=g)], 1), 120, 0.25),
        ("softmaxlike", torch.softmax(torch.randn(4, 4224, generator=g) * 3, -1) + torch.rand(4, 4224, generator=g), 4096, 0.5),
        ("allrecent", torch.rand(2, 10, generator=g), 1, 0.5),
    ]
    for name, s, budget, ratio in cases:
        keep = M.select_h2o_indices_batch(s, budget=budget, recent_ratio=ratio)
        keep1 = torch.stack([M.select_h2o_indices(r, budget=budget, recent_ratio=ratio) for r in s])
        assert torch.equal(keep, keep1)
        out[f"{name}_scores"] = s.numpy()
        out[f"{name}_cfg"] = np.array([budget, ratio], dtype=np.float64)
        out[f"{name}_keep"] = keep.numpy()
    save("h2o_select", **out)


def gen_h2o_scores():
    from sparsevllm.engine.cache_manager.h2o import H2OCacheManager as M

    out = {}
    g = torch.Generator().manual_seed(11)
    # decode normalisation exactly as sparse_controller.py:766-767
    B, W, D = 3, 200, 128
    raw = torch.full((B, W), -1e20)
    lens = [200, 150, 7]
    for b, L in enumerate(lens):
        raw[b, :L] = torch.randn(L, generator=g) * 6
    out["norm_raw"] = raw.numpy().copy()
    x = raw.clone()
    x.mul_(float(D) ** -0.5)
    torch.softmax(x, dim=-1, out=x)
    out["norm_out"] = x.numpy()
    out["norm_meta"] = np.array([D], dtype=np.int64)

    # accumulate (h2o.py:610-626) incl. the reference unit-test numbers (:400-452 style)
    prev = torch.rand(37, generator=g)
    step = torch.rand(64, generator=g)
    out["acc_prev"] = prev.numpy()
    out["acc_step"] = step.numpy()
    out["acc_out"] = M._accumulate_score(prev, step, new_len=50, weight=128.0).numpy()
    out["acc_out_none"] = M._accumulate_score(None, step, new_len=20, weight=3.0).numpy()

    # logits-mode normalisation (h2o.py:628-655)
    lg = torch.randn(80, generator=g) * 4
    lg[10:20] = float("-inf")
    out["logit_in"] = lg.numpy()
    out["logit_out"] = M._normalize_logit_prefill_score(lg, new_len=64).numpy()
    save("h2o_scores", **out)


# ----------------------------------------------------------------------------------
# manager-level state transitions (slot table / free stack / scores)
# ----------------------------------------------------------------------------------

def _make_manager(lengths_by_layer, *, cap=96, nslots=1024, budget=8, interval=4, prefill_budget=16,
                  heads=2, dim=4, slot_seed=3, free_ptr=None, cls_name="H2OCacheManager"):
    """Hand-built H2OCacheManager (no engine), same attribute recipe the reference's
    unit tests use (tests/test_h2o_cache_manager.py:49-138), with a *shared* layer-major
    slot table / free-stack tensor so the fused batch-layers paths are exercised."""
    from sparsevllm.engine.cache_manager.h2o import H2OCacheManager
    from sparsevllm.engine.cache_manager.snapkv import SnapKVCacheManager
    from sparsevllm.method_registry import PREFILL_POLICY_ALL_CHUNKED

    L = len(lengths_by_layer)
    B = len(lengths_by_layer[0])
    m = object.__new__({"H2OCacheManager": H2OCacheManager, "SnapKVCacheManager": SnapKVCacheManager}[cls_name])
    m.device = torch.device("cpu")
    m.num_layers = L
    m.num_kv_layers = L
    m.runtime_layout = SimpleNamespace(
        kv_idx_to_layer_idx=tuple(range(L)),
        kv_layer_index=lambda layer: int(layer),
        is_full_attention=lambda layer: 0 <= int(layer) < L,
    )
    m.config = SimpleNamespace(
        vllm_sparse_method="h2o", h2o_decode_budget=budget, h2o_decode_eviction_interval=interval,
        h2o_prefill_budget=prefill_budget, h2o_recent_ratio=0.5, h2o_prefill_score_window=4,
        sparse_prefill_score_mode="probability", max_model_len=cap, snapkv_window_size=4,
        snapkv_num_full_layers=0, pyramid_layer_ratios=None,
        prefill_schedule_policy=PREFILL_POLICY_ALL_CHUNKED, chunk_prefill_size=4,
        validate_runtime_invariants=False, num_kvcache_slots=nslots,
    )
    m.validate_runtime_invariants = False
    m.max_model_len = cap
    m.num_kv_heads = heads
    m.head_dim = dim
    m.hf_config = SimpleNamespace(torch_dtype=torch.float32)
    m._h2o_scores = {}
    m._h2o_active_decode_seq_ids = set()
    m._h2o_counters = {k: 0 for k in ("intermediate_prefill_evictions", "final_prefill_evictions",
                                      "decode_eviction_bursts", "decode_evictions", "dropped_tokens")}
    m._h2o_final_prefill_workspace = None
    m._uniform_decode_metadata = False
    m.seq_id_to_row = [{i: i for i in range(B)} for _ in range(L)]
    m.row_seq_lens = [np.asarray(x, dtype=np.int32) for x in lengths_by_layer]
    m.buffer_req_to_token_slots_tensor = torch.zeros((L, B, cap), dtype=torch.int32)
    m.buffer_req_to_token_slots = [m.buffer_req_to_token_slots_tensor[i] for i in range(L)]
    g = torch.Generator().manual_seed(slot_seed)
    m.free_slots_stack_tensor = torch.zeros((L, nslots), dtype=torch.int32)
    m.free_slots_stack = [m.free_slots_stack_tensor[i] for i in range(L)]
    m._num_free_slots = []
    for li, lengths in enumerate(lengths_by_layer):
        perm = torch.randperm(nslots, generator=g).to(torch.int32)
        used = 0
        for r, n in enumerate(lengths):
            m.buffer_req_to_token_slots[li][r, :n] = perm[used: used + n]
            used += n
        nfree = nslots - used if free_ptr is None else free_ptr
        m.free_slots_stack[li][:nfree] = perm[used: used + nfree]
        m._num_free_slots.append(int(nfree))
    m.kv_cache = torch.zeros((2, L, nslots, heads, dim), dtype=torch.float32)
    m.attention_cache_storage = None
    return m


def _state(m):
    return dict(
        slot_table=m.buffer_req_to_token_slots_tensor.numpy().copy(),
        free_stack=m.free_slots_stack_tensor.numpy().copy(),
        free_ptr=np.asarray(m._num_free_slots, dtype=np.int64),
        row_len=np.stack(m.row_seq_lens).astype(np.int32),
    )


def _put(out, prefix, st):
    for k_, v_ in st.items():
        out[f"{prefix}_{k_}"] = v_


def gen_compaction():
    from sparsevllm.engine.cache_manager.snapkv import SnapKVCacheManager
    from sparsevllm.engine.cache_manager.h2o import H2OCacheManager

    out = {}
    g = torch.Generator().manual_seed(5)
    seqs = lambda n: [SimpleNamespace(seq_id=i, is_last_chunk_prefill=True) for i in range(n)]

    # (1) fused batch-layers compaction, uniform lengths
    L, B, cur, K = 3, 4, 40, 25
    m = _make_manager([[cur] * B] * L)
    keep = torch.stack([torch.stack([torch.randperm(cur, generator=g)[:K] for _ in range(B)]) for _ in range(L)])
    _put(out, "bl_before", _state(m))
    out["bl_keep"] = keep.numpy()
    SnapKVCacheManager.free_part_slots_batch_layers(m, list(range(L)), seqs(B), keep, keep_indices_sorted=False)
    _put(out, "bl_after", _state(m))

    # (2) non-uniform lengths -> per-row fallback order
    m = _make_manager([[30, 22, 30], [30, 22, 30]], cls_name="SnapKVCacheManager")
    keep = torch.stack([torch.stack([torch.sort(torch.randperm(22, generator=g)[:10]).values for _ in range(3)]) for _ in range(2)])
    _put(out, "nu_before", _state(m))
    out["nu_keep"] = keep.numpy()
    SnapKVCacheManager.free_part_slots_batch_layers(m, [0, 1], seqs(3), keep, keep_indices_sorted=True)
    _put(out, "nu_after", _state(m))

    # (3) StreamingLLM sink+recent compaction
    m = _make_manager([[50] * 3] * 2)
    _put(out, "sr_before", _state(m))
    out["sr_cfg"] = np.array([50, 4, 16], dtype=np.int64)
    SnapKVCacheManager.free_prefix_recent_slots_batch_layers(
        m, [0, 1], seqs(3), kv_len=50, num_sink_tokens=4, num_recent_tokens=16)
    _put(out, "sr_after", _state(m))

    # (4) final-prefill dense compaction (K/V payload moves)
    budget = 8
    m = _make_manager([[20] * 3], budget=budget, heads=2, dim=4)
    m.kv_cache.copy_(torch.randn(m.kv_cache.shape, generator=g))
    keep = torch.stack([torch.sort(torch.randperm(20, generator=g)[:budget]).values for _ in range(3)])
    _put(out, "fp_before", _state(m))
    out["fp_keep"] = keep.numpy()
    out["fp_kv_before"] = m.kv_cache.numpy().copy()
    m.get_layer_kv_cache = lambda layer_idx: (m.kv_cache[0, layer_idx], m.kv_cache[1, layer_idx])
    H2OCacheManager._compact_final_prefill_dense_batch(m, 0, seqs(3), keep)
    _put(out, "fp_after", _state(m))
    out["fp_kv_after"] = m.kv_cache.numpy().copy()
    out["fp_cfg"] = np.array([budget], dtype=np.int64)
    save("compaction", **out)


def gen_h2o_burst():
    """Whole decode bursts through H2OCacheManager._evict_decode_rows (h2o.py:1558-1625)
    incl. the slot-pressure trigger and unscheduled active rows (h2o.py:1498-1538)."""
    out = {}
    g = torch.Generator().manual_seed(9)
    budget, interval = 8, 4

    def run(tag, lens, scheduled, active, free_ptr):
        L = 2
        m = _make_manager([list(lens)] * L, budget=budget, interval=interval, free_ptr=free_ptr)
        for l in range(L):
            for r, n in enumerate(lens):
                # coarse values -> guaranteed ties
                m._h2o_scores[(l, r)] = (torch.randint(0, 6, (n,), generator=g).float() / 3.0
                                         + (torch.rand(n, generator=g) > 0.7).float() * torch.rand(n, generator=g))
        m._h2o_active_decode_seq_ids = set(active)
        _put(out, f"{tag}_before", _state(m))
        for l in range(L):
            for r, n in enumerate(lens):
                out[f"{tag}_score_before_{l}_{r}"] = m._h2o_scores[(l, r)].numpy().copy()
        m._evict_decode_rows([SimpleNamespace(seq_id=i) for i in scheduled])
        _put(out, f"{tag}_after", _state(m))
        for l in range(L):
            for r, n in enumerate(lens):
                out[f"{tag}_score_after_{l}_{r}"] = m._h2o_scores[(l, r)].numpy().copy()
        out[f"{tag}_cfg"] = np.array([budget, interval, free_ptr], dtype=np.int64)
        out[f"{tag}_scheduled"] = np.array(scheduled, dtype=np.int64)
        out[f"{tag}_active"] = np.array(sorted(active), dtype=np.int64)
        out[f"{tag}_counters"] = np.array([m._h2o_counters[k] for k in
                                           ("decode_eviction_bursts", "decode_evictions", "dropped_tokens")], dtype=np.int64)

    run("periodic", (12, 12, 9, 12), [0, 1, 2, 3], {0, 1, 2, 3}, 100)
    run("pressure", (12, 9, 10, 11), [0, 1], {0, 1, 2, 3}, 0)
    run("noop", (11, 10, 9, 8), [0, 1, 2, 3], {0, 1, 2, 3}, 50)
    save("h2o_burst", **out)


def gen_quest():
    """QuestCacheManager (hand-built, SURVEY.md appendix A recipe): page metadata after a decode
    step completes pages (on_forward_end), page scores (_score_pages_batched) and the packed decode
    view (build_decode_view) in bf16 on CPU."""
    from collections import deque
    from sparsevllm.engine.cache_manager.base import LayerBatchStates
    from sparsevllm.engine.cache_manager.quest import QuestCacheManager
    from sparsevllm.utils.context import set_context, reset_context

    out = {}
    g = torch.Generator().manual_seed(21)
    page, Hkv, D, Hq, L = 16, 2, 64, 14, 1
    rows, max_model_len = 3, 512
    num_pages = 100
    m = object.__new__(QuestCacheManager)
    m.device = torch.device("cpu")
    m.page_size = page
    m.max_model_len = max_model_len
    m.max_pages_per_row = max_model_len // page
    m.head_dim, m.num_kv_heads, m.num_kv_layers, m.num_layers = D, Hkv, L, L
    m.runtime_layout = SimpleNamespace(kv_layer_index=lambda l: int(l), kv_idx_to_layer_idx=tuple(range(L)))
    m.page_offsets_i32 = torch.arange(page, dtype=torch.int32)
    m.page_offsets_i64 = m.page_offsets_i32.to(torch.int64)
    m.num_pages = num_pages
    m.kv_cache = (torch.randn(2, L, num_pages * page, Hkv, D, generator=g) * 0.5).to(torch.bfloat16)
    m.metadata_cache = torch.zeros(2, L, num_pages, Hkv, D, dtype=torch.bfloat16)
    m.buffer_req_to_token_slots = torch.zeros(rows, max_model_len, dtype=torch.int32)
    m.buffer_req_to_page_slots = torch.full((rows, m.max_pages_per_row), -1, dtype=torch.int32)
    m.buffer_req_to_page_slots_cpu = np.full((rows, m.max_pages_per_row), -1, dtype=np.int32)
    m.enable_prefix_caching = False
    m.prefix_offload_controller = None
    m.seq_id_to_row = {}
    m.row_seq_lens = np.zeros((rows,), dtype=np.int32)
    m.layer_batch_state = LayerBatchStates()
    m.config = SimpleNamespace(quest_skip_layers=0, quest_token_budget=160)
    # rows: lengths 480 (full pages), 331 (partial last page), 96 (short -> dense in mixed mode)
    lens = [480, 331, 96]
    perm = torch.randperm(num_pages, generator=g).tolist()
    used = 0
    for r, n in enumerate(lens):
        m.seq_id_to_row[r] = r
        npg = (n + page - 1) // page
        ps = perm[used: used + npg]
        used += npg
        m.buffer_req_to_page_slots[r, :npg] = torch.tensor(ps, dtype=torch.int32)
        m.buffer_req_to_page_slots_cpu[r, :npg] = np.asarray(ps, dtype=np.int32)
        pos = torch.arange(n)
        m.buffer_req_to_token_slots[r, :n] = (torch.tensor(ps)[pos // page] * page + pos % page).to(torch.int32)
        m.row_seq_lens[r] = n
    out["kv_k"] = bits(m.kv_cache[0].float())
    out["page_table"] = m.buffer_req_to_page_slots.numpy().copy()
    out["token_table"] = m.buffer_req_to_token_slots.numpy().copy()
    out["lens"] = np.asarray(lens, dtype=np.int32)
    # metadata of every complete page through the reference's own decode-completion hook: pretend each
    # row just completed each of its full pages in turn
    class _Base:  # CacheManager.on_forward_end default is a no-op for this purpose
        pass
    seqs = [SimpleNamespace(seq_id=r) for r in range(rows)]
    real_lens = m.row_seq_lens.copy()
    import sparsevllm.engine.cache_manager.quest as qmod
    orig_super = qmod.CacheManager.on_forward_end
    qmod.CacheManager.on_forward_end = lambda self, seqs, is_prefill: None
    m._poll_prefix_offload = lambda: None
    try:
        max_full = max(n // page for n in lens)
        for pg in range(1, max_full + 1):
            for r, n in enumerate(lens):
                m.row_seq_lens[r] = pg * page if pg * page <= n else 1   # len%16==0 -> page pg-1 "completed"
            QuestCacheManager.on_forward_end(m, seqs, False)
    finally:
        qmod.CacheManager.on_forward_end = orig_super
    m.row_seq_lens[:] = real_lens
    out["metadata"] = np.stack([bits(m.metadata_cache[0].float()), bits(m.metadata_cache[1].float())])

    q = (torch.randn(rows, Hq, D, generator=g) * 0.5).to(torch.bfloat16)
    out["q"] = bits(q.float())
    ctx_lens = torch.tensor(lens, dtype=torch.int32)
    req = torch.arange(rows, dtype=torch.int32)
    for tag, long_text in (("long", True), ("mixed", False)):
        reset_context()
        set_context(False, is_long_text=long_text)
        m.layer_batch_state.max_context_len = max(lens)
        active = m.buffer_req_to_token_slots
        packed, ridx, clens = QuestCacheManager.build_decode_view(
            m, 0, q, active, req, ctx_lens, num_heads=Hq, num_kv_heads=Hkv)
        out[f"{tag}_packed"] = packed.numpy()
        out[f"{tag}_req"] = ridx.numpy()
        out[f"{tag}_lens"] = clens.numpy()
    reset_context()
    # raw page scores of layer 2 for all previous pages of each row
    max_pages = (max(lens) + page - 1) // page
    prev = m.buffer_req_to_page_slots[:, : max_pages - 1].long().clamp_min(0)
    pmax = m.metadata_cache[0, 0].index_select(0, prev.reshape(-1)).view(rows, max_pages - 1, Hkv, D).permute(0, 2, 1, 3)
    pmin = m.metadata_cache[1, 0].index_select(0, prev.reshape(-1)).view(rows, max_pages - 1, Hkv, D).permute(0, 2, 1, 3)
    sc = QuestCacheManager._score_pages_batched(q, pmax, pmin, Hkv)
    out["page_scores"] = sc.float().numpy()
    out["cfg"] = np.array([page, 160, 0, max(lens), m.max_pages_per_row, Hkv, 0], dtype=np.int64)
    save("quest", **out)


def gen_prefill_score():
    """prefill_score_fwd (both modes) under the Triton interpreter, fp32 tensors holding bf16 values."""
    from sparsevllm.kernels.triton.prefill_score import prefill_score_fwd

    out = {}
    g = torch.Generator().manual_seed(33)
    cases = [
        # name, Hq, Hkv, D, [(cache_len, chunk_len)], window, candidate_start, num_recent, mode
        ("p1", 8, 2, 64, [(96, 40), (0, 57)], 32, 0, 0, "probability"),
        ("p2", 28, 4, 128, [(70, 50)], 32, 4, 16, "probability"),       # SnapKV-style sink/recent carve-out
        ("l1", 8, 2, 64, [(96, 40), (10, 30)], 40, 0, 0, "logits"),
        ("p3", 14, 2, 64, [(0, 20)], 128, 0, 0, "probability"),          # window longer than the chunk
    ]
    for name, Hq, Hkv, D, seqs_cfg, window, cstart, nrecent, mode in cases:
        nb = len(seqs_cfg)
        ctx = [c + n for c, n in seqs_cfg]
        slots = sum(ctx) + 32
        k = bf16f(torch.randn(slots, Hkv, D, generator=g) * 0.5)
        total_q = sum(n for _, n in seqs_cfg)
        q = bf16f(torch.randn(total_q, Hq, D, generator=g) * 0.5)
        cap = max(ctx) + 9
        req = torch.zeros(nb + 1, cap, dtype=torch.int32)
        perm = torch.randperm(slots, generator=g).to(torch.int32)
        rows = list(range(nb, 0, -1))           # rows nb..1 (not identity)
        off = 0
        for i, L in enumerate(ctx):
            req[rows[i], :L] = perm[off: off + L]
            off += L
        b_req = torch.tensor(rows, dtype=torch.int32)
        b_start = torch.tensor(np.concatenate(([0], np.cumsum([n for _, n in seqs_cfg])[:-1])), dtype=torch.int32)
        b_seq = torch.tensor(ctx, dtype=torch.int32)
        b_cache = torch.tensor([c for c, _ in seqs_cfg], dtype=torch.int32)
        q_end = torch.tensor(ctx, dtype=torch.int32)
        q_start = torch.tensor([max(L - window, c) for L, (c, n) in zip(ctx, seqs_cfg)], dtype=torch.int32)
        max_q = int((q_end - q_start).max())
        Lc = max(ctx)
        score = torch.empty(nb, Lc)
        prefill_score_fwd(q, k, score, b_req, b_start, b_seq, b_cache, max_q, req, q_start, q_end,
                          candidate_start=cstart, num_recent_tokens=nrecent, score_mode=mode)
        out.update({f"{name}_q": bits(q), f"{name}_k": bits(k), f"{name}_req": req.numpy(),
                    f"{name}_b_req": b_req.numpy(), f"{name}_b_start": b_start.numpy(), f"{name}_b_seq": b_seq.numpy(),
                    f"{name}_b_cache": b_cache.numpy(), f"{name}_q_start": q_start.numpy(), f"{name}_q_end": q_end.numpy(),
                    f"{name}_cfg": np.array([max_q, cstart, nrecent, 1 if mode == "logits" else 0], dtype=np.int64),
                    f"{name}_score": score.numpy()})
    save("prefill_score", **out)


def gen_deltakv():
    """DeltaKV decode-side Triton kernels under the interpreter.  Their Python wrappers assert
    `.is_cuda`, so the @triton.jit kernels are launched directly with the wrapper's grid / constexprs
    (deltakv_kernels.py:3854-3942, :2909-3012, :3344-3450; quant.py:160-216)."""
    from sparsevllm.kernels.triton import deltakv_kernels as dk
    from sparsevllm.kernels.triton import quant as qk

    out = {}
    g = torch.Generator().manual_seed(44)
    # ---- static decode plan
    B, K, sink, max_buffer, rows, max_pos = 4, 12, 3, 6, 6, 64
    S = sink + K + max_buffer
    raw = torch.randint(-1, 500, (rows, max_pos), generator=g, dtype=torch.int32)
    raw[:, :sink] = torch.randint(0, 500, (rows, sink), generator=g, dtype=torch.int32)
    lat = torch.randint(-1, 300, (rows, max_pos), generator=g, dtype=torch.int32)
    lat[lat % 3 == 0] = -1
    active = torch.randint(-2, 40, (B, K), generator=g, dtype=torch.int32)
    req = torch.tensor([4, 0, 2, 5], dtype=torch.int32)
    ctx = torch.tensor([50, 30, 9, 64], dtype=torch.int32)
    clen = torch.tensor([40, 5, 0, 55], dtype=torch.int32)
    temp = torch.randint(600, 700, (B, K), generator=g, dtype=torch.int32)
    a_slots = torch.zeros(B, S, dtype=torch.int32); a_pos = torch.zeros(B, S, dtype=torch.int32)
    new_len = torch.zeros(B, dtype=torch.int32)
    r_pos = torch.zeros(B * K, dtype=torch.int32); r_lat = torch.zeros(B * K, dtype=torch.int32); r_out = torch.zeros(B * K, dtype=torch.int32)
    block_m = 1 << max(0, (S - 1).bit_length())
    dk._deltakv_static_decode_plan_kernel[(B,)](
        raw, lat, active, req, ctx, clen, temp, a_slots, a_pos, new_len, r_pos, r_lat, r_out,
        raw.stride(0), raw.stride(1), lat.stride(0), lat.stride(1), active.stride(0), active.stride(1),
        temp.stride(0), temp.stride(1), a_slots.stride(0), a_slots.stride(1), a_pos.stride(0), a_pos.stride(1),
        SINK=sink, K_MAX=K, MAX_BUFFER=max_buffer, MAX_S=S, MAX_POS=max_pos - 1, BLOCK_M=block_m)
    out.update(plan_raw=raw.numpy(), plan_lat=lat.numpy(), plan_active=active.numpy(), plan_req=req.numpy(),
               plan_ctx=ctx.numpy(), plan_clen=clen.numpy(), plan_temp=temp.numpy(),
               plan_cfg=np.array([sink, max_buffer], dtype=np.int64), plan_slots=a_slots.numpy(), plan_pos=a_pos.numpy(),
               plan_new_len=new_len.numpy(), plan_rpos=r_pos.numpy(), plan_rlat=r_lat.numpy(), plan_rout=r_out.numpy())

    # ---- reconstruct + RoPE write-back: dense delta and packed int2 / int4 residuals
    Hkv, D, slots, N, Kf, max_p = 2, 64, 96, 10, 4, 128
    Dtot = Hkv * D
    kc0 = bf16f(torch.randn(slots, Hkv, D, generator=g) * 0.5)
    vc0 = bf16f(torch.randn(slots, Hkv, D, generator=g) * 0.5)
    inv_freq = 1.0 / (10000 ** (torch.arange(0, D // 2).float() / (D // 2)))
    ang = torch.arange(max_p).float()[:, None] * inv_freq[None, :]
    cos_sin = torch.cat((ang.cos(), ang.sin()), dim=1).contiguous()
    fathers = torch.randint(0, 64, (N, Kf), generator=g, dtype=torch.int32)
    slot_to_pos = torch.randint(0, max_p, (slots,), generator=g, dtype=torch.int32)
    out_slots = torch.arange(70, 70 + N, dtype=torch.int32)
    out_pos = torch.randint(0, max_p, (N,), generator=g, dtype=torch.int32)
    out_slots_d = out_slots.clone(); out_slots_d[3] = -1          # dense form: entry 3 skipped
    knorm = torch.rand(D, generator=g) + 0.5
    delta = bf16f(torch.randn(N, 2 * Dtot, generator=g) * 0.3)
    out.update(rc_k=bits(kc0), rc_v=bits(vc0), rc_cos_sin=cos_sin.numpy(), rc_fathers=fathers.numpy(),
               rc_slot_to_pos=slot_to_pos.numpy(), rc_out_slots=out_slots.numpy(), rc_out_slots_dense=out_slots_d.numpy(),
               rc_out_pos=out_pos.numpy(), rc_delta=bits(delta), rc_knorm=knorm.numpy())
    for tag, raw_k, use_norm in (("dense", False, False), ("dense_norm_raw", True, True)):
        kc, vc = kc0.clone(), vc0.clone()
        dk._deltakv_reconstruct_writeback_grouped_heads_kernel[(N, 1)](
            delta, fathers, slot_to_pos, out_slots_d, out_pos, cos_sin, kc, vc, kc, vc, knorm if use_norm else cos_sin,
            delta.stride(0), delta.stride(1), fathers.stride(0), fathers.stride(1), cos_sin.stride(0), cos_sin.stride(1),
            kc.stride(0), kc.stride(1), kc.stride(2), vc.stride(0), vc.stride(1), vc.stride(2),
            kc.stride(0), kc.stride(1), kc.stride(2), vc.stride(0), vc.stride(1), vc.stride(2),
            (knorm if use_norm else cos_sin).stride(0),
            D=Dtot, HEAD_DIM=D, HD2=D // 2, NUM_KV_HEADS=Hkv, K=Kf, HEADS_PER_PROG=2, USE_PRE_ROPE_K=False,
            USE_REF_V=False, APPLY_K_NORM=use_norm, K_NORM_EPS=1e-6, RAW_K_CACHE=raw_k, STORE_RAW_K=False)
        out[f"rc_{tag}_k"] = kc.numpy(); out[f"rc_{tag}_v"] = vc.numpy()
    n_lat = 24
    latent_slots = torch.randint(0, n_lat, (N,), generator=g, dtype=torch.int32)
    latent_slots[5] = -1                                           # quantised form: entry 5 skipped
    out["rc_latent_slots"] = latent_slots.numpy()
    for qbits, group in ((4, 32), (2, 32), (2, 2 * Dtot)):
        fpi = 32 // qbits
        codes = torch.randint(0, 2 ** 31 - 1, (n_lat, 2 * Dtot // fpi), generator=g, dtype=torch.int32)
        codes = codes ^ (torch.randint(0, 2, codes.shape, generator=g, dtype=torch.int32) << 31)
        ngroups = 2 * Dtot // group
        scale = torch.rand(n_lat, ngroups, generator=g) * 0.05
        mn = -torch.rand(n_lat, ngroups, generator=g) * 0.3
        kc, vc = kc0.clone(), vc0.clone()
        dk._deltakv_less_memory_reconstruct_writeback_kernel[(N, 1)](
            codes, scale, mn, latent_slots, fathers, slot_to_pos, out_slots, out_pos, cos_sin, kc, vc, cos_sin,
            codes.stride(0), codes.stride(1), scale.stride(0), scale.stride(1), mn.stride(0), mn.stride(1),
            fathers.stride(0), fathers.stride(1), cos_sin.stride(0), cos_sin.stride(1),
            kc.stride(0), kc.stride(1), kc.stride(2), vc.stride(0), vc.stride(1), vc.stride(2), cos_sin.stride(0),
            D=Dtot, HEAD_DIM=D, HD2=D // 2, NUM_KV_HEADS=Hkv, K=Kf, HEADS_PER_PROG=2, BITS=qbits, FEAT_PER_INT=fpi,
            QUANT_MASK=(1 << qbits) - 1, GROUP_SIZE=group, RAW_K_CACHE=False, STORE_RAW_K=False, APPLY_K_NORM=False,
            K_NORM_EPS=1e-6)
        t = f"q{qbits}g{group}"
        out.update({f"rc_{t}_codes": codes.numpy(), f"rc_{t}_scale": scale.numpy(), f"rc_{t}_mn": mn.numpy(),
                    f"rc_{t}_k": kc.numpy(), f"rc_{t}_v": vc.numpy()})
    # ---- 2-D int4 grouped quantise + dequantise (quant.py:28-216), fp32 data
    data = torch.randn(6, 64, generator=g)
    code = torch.empty((6, 8), dtype=torch.int32); sc = torch.empty((6, 2)); mnq = torch.empty((6, 2))
    qk._quantize_pack_2d_int4_grouped_kernel[(6, 2)](data, code, sc, mnq, data.stride(0), data.stride(1), code.stride(0),
                                                     code.stride(1), sc.stride(0), sc.stride(1), D=64, GROUP_SIZE=32,
                                                     PACKS_PER_GROUP=4, BLOCK_G=32)
    deq = torch.empty((6, 64))
    qk._dequantize_2d_int4_grouped_kernel[(6,)](code, sc, mnq, deq, code.stride(0), code.stride(1), sc.stride(0),
                                                sc.stride(1), deq.stride(0), deq.stride(1), D=64, GROUP_SIZE=32, BLOCK_D=64)
    out.update(q4_data=data.numpy(), q4_code=code.numpy(), q4_scale=sc.numpy(), q4_mn=mnq.numpy(), q4_deq=deq.numpy())
    save("deltakv", **out)


def gen_kivi():
    """_full_layer_kivi_flash_decode_stage1_kernel under the interpreter (fp32 tensors holding bf16 values)."""
    from sparsevllm.kernels.triton import deltakv_kernels as dk

    out = {}
    g = torch.Generator().manual_seed(55)
    B, Hq, Hkv, D, G = 2, 8, 2, 64, 32
    rows, max_len = 3, 160
    lens = [150, 70]
    req = torch.tensor([2, 0], dtype=torch.int32)
    n_blocks, n_raw = 8, 96
    raw_k = bf16f(torch.randn(n_raw, Hkv, D, generator=g) * 0.5)
    raw_v = bf16f(torch.randn(n_raw, Hkv, D, generator=g) * 0.5)
    key_packed = torch.randint(-2 ** 31, 2 ** 31 - 1, (n_blocks, Hkv, D, G // 8), generator=g, dtype=torch.int64).to(torch.int32)
    value_packed = torch.randint(-2 ** 31, 2 ** 31 - 1, (n_blocks, Hkv, G, D // 8), generator=g, dtype=torch.int64).to(torch.int32)
    key_scales = bf16f(torch.rand(n_blocks, Hkv, D, generator=g) * 0.08)
    key_mins = bf16f(-torch.rand(n_blocks, Hkv, D, generator=g) * 0.6)
    value_scales = bf16f(torch.rand(n_blocks, Hkv, G, D // G, generator=g) * 0.08)
    value_mins = bf16f(-torch.rand(n_blocks, Hkv, G, D // G, generator=g) * 0.6)
    raw_map = torch.full((rows, max_len), -1, dtype=torch.int32)
    blk_map = torch.full((rows, max_len), -1, dtype=torch.int32)
    blk_start = torch.zeros(n_blocks, dtype=torch.int32)
    # row 2 (lane 0): sink 8 raw, 4 KIVI blocks (tokens 8..135), raw tail; row 0 (lane 1): sink 8 raw, 1 block, raw tail
    perm = torch.randperm(n_raw, generator=g).to(torch.int32)
    used = 0
    def put_raw(r, a, b_):
        nonlocal used
        raw_map[r, a:b_] = perm[used: used + (b_ - a)]
        used += b_ - a
    put_raw(2, 0, 8)
    for i, bslot in enumerate([5, 1, 6, 2]):
        blk_map[2, 8 + 32 * i: 8 + 32 * (i + 1)] = bslot
        blk_start[bslot] = 8 + 32 * i
    put_raw(2, 136, 150)
    put_raw(0, 0, 8)
    blk_map[0, 8:40] = 4
    blk_start[4] = 8
    put_raw(0, 40, 70)
    q = bf16f(torch.randn(B, Hq, D, generator=g) * 0.5)
    clens = torch.tensor(lens, dtype=torch.int32)
    block_seq = 64
    nblk = (max(lens) + block_seq - 1) // block_seq
    mid = torch.zeros(B, Hq, nblk, D); lse = torch.zeros(B, Hq, nblk)
    score = torch.full((B, Hq, max(lens)), -1e20)
    grid = (B, Hkv, nblk)
    dk._full_layer_kivi_flash_decode_stage1_kernel[grid](
        q, raw_k, raw_v, raw_map, blk_map, blk_start, key_packed, key_scales, key_mins, value_packed, value_scales,
        value_mins, req, clens, mid, lse, score,
        q.stride(0), q.stride(1), q.stride(2), raw_k.stride(0), raw_k.stride(1), raw_k.stride(2),
        raw_v.stride(0), raw_v.stride(1), raw_v.stride(2), raw_map.stride(0), raw_map.stride(1),
        blk_map.stride(0), blk_map.stride(1),
        key_packed.stride(0), key_packed.stride(1), key_packed.stride(2), key_packed.stride(3),
        key_scales.stride(0), key_scales.stride(1), key_scales.stride(2),
        value_packed.stride(0), value_packed.stride(1), value_packed.stride(2), value_packed.stride(3),
        value_scales.stride(0), value_scales.stride(1), value_scales.stride(2), value_scales.stride(3),
        mid.stride(0), mid.stride(1), mid.stride(2), mid.stride(3), lse.stride(0), lse.stride(1), lse.stride(2),
        score.stride(0), score.stride(1), score.stride(2),
        sm_scale=1.0 / (D ** 0.5), gqa_group_size=Hq // Hkv, Q_HEAD_NUM=16, BLOCK_SEQ=block_seq, BLOCK_DMODEL=D,
        BLOCK_N=16, GROUP_SIZE=G, FEAT_PER_INT=8, QUANT_MASK=15, STORE_SCORE=True)
    out.update(q=bits(q), raw_k=bits(raw_k), raw_v=bits(raw_v), raw_map=raw_map.numpy(), blk_map=blk_map.numpy(),
               blk_start=blk_start.numpy(), key_packed=key_packed.numpy(), key_scales=bits(key_scales),
               key_mins=bits(key_mins), value_packed=value_packed.numpy(), value_scales=bits(value_scales),
               value_mins=bits(value_mins), req=req.numpy(), lens=clens.numpy(),
               cfg=np.array([G, block_seq, max(lens)], dtype=np.int64), mid_o=mid.numpy(), mid_lse=lse.numpy(),
               score=score.numpy())
    # ---- one LONG row (8259 tokens, one KV head, 7 query heads of 128 dims) so that the decode kernel is pinned by the
    # reference above toy size; the inputs come from a seed shared with the tests (tests/golden_inputs.py), only the
    # reference's partials and raw scores are stored
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import golden_inputs
    d = golden_inputs.kivi_long_row_inputs()
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x))
    q, raw_k, raw_v = T(d["q"]), T(d["raw_k"]), T(d["raw_v"])
    raw_map, blk_map, blk_start = T(d["raw_map"]), T(d["blk_map"]), T(d["blk_start"])
    key_packed, value_packed = T(d["key_packed"]), T(d["value_packed"])
    key_scales, key_mins, value_scales, value_mins = T(d["key_scales"]), T(d["key_mins"]), T(d["value_scales"]), T(d["value_mins"])
    req, clens = T(d["req"]), T(d["lens"])
    Hq, Hkv, D, G, length = d["Hq"], d["Hkv"], d["D"], d["G"], d["length"]
    block_seq = 1024
    nblk = (length + block_seq - 1) // block_seq
    mid = torch.zeros(1, Hq, nblk, D); lse = torch.zeros(1, Hq, nblk)
    score = torch.full((1, Hq, length), -1e20)
    dk._full_layer_kivi_flash_decode_stage1_kernel[(1, Hkv, nblk)](
        q, raw_k, raw_v, raw_map, blk_map, blk_start, key_packed, key_scales, key_mins, value_packed, value_scales,
        value_mins, req, clens, mid, lse, score,
        q.stride(0), q.stride(1), q.stride(2), raw_k.stride(0), raw_k.stride(1), raw_k.stride(2),
        raw_v.stride(0), raw_v.stride(1), raw_v.stride(2), raw_map.stride(0), raw_map.stride(1),
        blk_map.stride(0), blk_map.stride(1),
        key_packed.stride(0), key_packed.stride(1), key_packed.stride(2), key_packed.stride(3),
        key_scales.stride(0), key_scales.stride(1), key_scales.stride(2),
        value_packed.stride(0), value_packed.stride(1), value_packed.stride(2), value_packed.stride(3),
        value_scales.stride(0), value_scales.stride(1), value_scales.stride(2), value_scales.stride(3),
        mid.stride(0), mid.stride(1), mid.stride(2), mid.stride(3), lse.stride(0), lse.stride(1), lse.stride(2),
        score.stride(0), score.stride(1), score.stride(2),
        sm_scale=1.0 / (D ** 0.5), gqa_group_size=Hq // Hkv, Q_HEAD_NUM=16, BLOCK_SEQ=block_seq, BLOCK_DMODEL=D,
        BLOCK_N=16, GROUP_SIZE=G, FEAT_PER_INT=8, QUANT_MASK=15, STORE_SCORE=True)
    out.update(long_cfg=np.array([G, block_seq, length], dtype=np.int64), long_mid_o=mid.numpy(), long_mid_lse=lse.numpy(),
               long_score=score.numpy())
    save("kivi", **out)


def gen_deltakv_view():
    """`deltakv_materialize_sparse_view` (deltakv_kernels.py:3489-3585, kernel :3588-3693): the attention-facing
    contiguous copy of a sparse layer's active slots (raw pre-RoPE K -> optional k-norm -> RoPE at slot_to_pos;
    slots flagged post-RoPE are copied).  The wrapper asserts `.is_cuda`, so the jit kernel is launched directly."""
    from sparsevllm.kernels.triton import deltakv_kernels as dk

    g = torch.Generator().manual_seed(45)
    Hkv, D, slots, B, W, max_p, BN = 2, 64, 80, 3, 11, 96, 16
    kc = bf16f(torch.randn(slots, Hkv, D, generator=g) * 0.5)
    vc = bf16f(torch.randn(slots, Hkv, D, generator=g) * 0.5)
    inv_freq = 1.0 / (10000 ** (torch.arange(0, D // 2).float() / (D // 2)))
    ang = torch.arange(max_p).float()[:, None] * inv_freq[None, :]
    cos_sin = torch.cat((ang.cos(), ang.sin()), dim=1).contiguous()
    active = torch.randint(0, slots, (B, W), generator=g, dtype=torch.int32)
    active[1, 7:] = -1                                   # padding beyond a short row
    lens = torch.tensor([W, 7, W - 2], dtype=torch.int32)
    slot_to_pos = torch.randint(0, max_p, (slots,), generator=g, dtype=torch.int32)
    slot_to_pos[5] = -1                                  # unknown position -> clamped to 0
    post = torch.rand(slots, generator=g) < 0.3
    knorm = torch.rand(D, generator=g) + 0.5
    out = dict(k=bits(kc), v=bits(vc), cos_sin=cos_sin.numpy(), active=active.numpy(), lens=lens.numpy(),
               slot_to_pos=slot_to_pos.numpy(), post=post.numpy(), knorm=knorm.numpy())
    total = B * W
    for tag, use_norm, use_mask in (("plain", False, False), ("norm_mask", True, True)):
        ok = torch.zeros(total, Hkv, D); ov = torch.zeros(total, Hkv, D)
        nw = knorm if use_norm else cos_sin
        pm = post if use_mask else slot_to_pos
        dk._deltakv_materialize_sparse_view_block_kernel[((total + BN - 1) // BN, Hkv)](
            active, lens, slot_to_pos, pm, kc, vc, ok, ov, cos_sin, nw,
            active.stride(0), active.stride(1), kc.stride(0), kc.stride(1), kc.stride(2),
            vc.stride(0), vc.stride(1), vc.stride(2), ok.stride(0), ok.stride(1), ok.stride(2),
            ov.stride(0), ov.stride(1), ov.stride(2), cos_sin.stride(0), cos_sin.stride(1), nw.stride(0),
            TOTAL=total, WIDTH=W, BLOCK_N=BN, NUM_SLOTS=slots, HEAD_DIM=D, HD2=D // 2, NUM_KV_HEADS=Hkv,
            APPLY_K_NORM=use_norm, HAS_POSTROPE_MASK=use_mask, K_NORM_EPS=1e-6)
        out[f"{tag}_k"] = ok.numpy(); out[f"{tag}_v"] = ov.numpy()
    save("deltakv_view", **out)


def gen_deltakv_compress():
    """DeltaKV compression side: residual quantiser (quant.py:29-117), KIVI block quantiser
    (quant.py:243-301 as used by _store_full_layer_kivi_blocks, deltakv_less_memory.py:1741-1780) and the
    L2 top-k father assignment `_cluster_compress` (deltakv_less_memory.py:2719-2802).  Low-precision inputs are
    fp16 (the interpreter cannot do bf16): they pin the per-operation rounding structure, which the oracle / HIP
    path then apply with bf16 rounding."""
    import contextlib
    from sparsevllm.kernels.triton import quant as qk
    from sparsevllm.engine.cache_manager.deltakv_less_memory import DeltaKVLessMemoryCacheManager as M

    g = torch.Generator().manual_seed(46)
    out = {}
    # ---- 2-D grouped int4 residual quantiser, fp32 and fp16 data
    for tag, dt in (("f32", torch.float32), ("f16", torch.float16)):
        n, d, group = 9, 64, 16
        data = (torch.randn(n, d, generator=g) * 0.7).to(dt)
        data[2] = 0.25                                            # constant row: scale 0
        code = torch.zeros(n, d // 8, dtype=torch.int32)
        scale = torch.zeros(n, d // group, dtype=dt); mn = torch.zeros(n, d // group, dtype=dt)
        qk._quantize_pack_2d_int4_grouped_kernel[(n, d // group)](
            data, code, scale, mn, data.stride(0), data.stride(1), code.stride(0), code.stride(1), scale.stride(0),
            scale.stride(1), D=d, GROUP_SIZE=group, PACKS_PER_GROUP=group // 8, BLOCK_G=16)
        out.update({f"q2d_{tag}_data": data.float().numpy(), f"q2d_{tag}_code": code.numpy(),
                    f"q2d_{tag}_scale": scale.float().numpy(), f"q2d_{tag}_mn": mn.float().numpy()})
    # ---- KIVI quantiser along the last dim (torch arithmetic + Triton min/max and pack), fp16 data
    saved = torch.cuda.device
    torch.cuda.device = lambda *_a, **_k: contextlib.nullcontext()
    try:
        for tag, dt in (("f16", torch.float16), ("f32", torch.float32)):
            blocks, H, D, G = 3, 2, 16, 32
            k = (torch.randn(blocks, G, H, D, generator=g) * 0.6).to(dt)
            v = (torch.randn(blocks, G, H, D, generator=g) * 0.6).to(dt)
            ks = k.permute(0, 2, 3, 1).contiguous()
            pk, sk, mk = qk.triton_quantize_and_pack_along_last_dim(ks, G, 4)
            vs = v.permute(0, 2, 1, 3).contiguous()
            pv, sv, mv = qk.triton_quantize_and_pack_along_last_dim(vs, 16, 4)
            out.update({f"kivi_{tag}_k": k.float().numpy(), f"kivi_{tag}_v": v.float().numpy(),
                        f"kivi_{tag}_kcode": pk.numpy(), f"kivi_{tag}_kscale": sk.squeeze(-1).float().numpy(),
                        f"kivi_{tag}_kmn": mk.squeeze(-1).float().numpy(), f"kivi_{tag}_vcode": pv.numpy(),
                        f"kivi_{tag}_vscale": sv.float().numpy(), f"kivi_{tag}_vmn": mv.float().numpy()})
    finally:
        torch.cuda.device = saved
    # ---- father assignment: L2 ranking score in the storage dtype, causal over the block's own centres, top-k, mean
    m = object.__new__(M)
    Hkv, D, slots = 2, 16, 64
    m.num_kv_heads, m.head_dim, m.device = Hkv, D, torch.device("cpu")
    m.hf_config = SimpleNamespace(torch_dtype=torch.bfloat16)
    m.config = SimpleNamespace(deltakv_k_neighbors=3, cluster_metric="l2", deltakv_cluster_gather_chunk_size=16384)
    m.deltakv_layer_to_idx = {1: 0}
    m.deltakv_full_kv_cache = (torch.randn(2, 1, slots, Hkv, D, generator=g) * 0.5).to(torch.bfloat16)
    m.deltakv_slot_to_pos = torch.arange(slots, dtype=torch.int32)
    m._is_stream_capturing = lambda: False
    m._prefill_pre_rope_stage_active = lambda: False
    n = 24
    kv = (torch.randn(1, n, 2 * Hkv * D, generator=g) * 0.5).to(torch.bfloat16)
    existing = torch.tensor([3, 9, 17, 40, 41], dtype=torch.int32)
    rel = torch.tensor([0, 6, 12, 18], dtype=torch.long)
    topk, base = m._cluster_compress(layer_idx=1, kv_states=kv, existing_center_slots=existing, cluster_step=6,
                                     new_center_rel=rel, validate_centers=False)
    out.update(cc_cache_k=bits(m.deltakv_full_kv_cache[0, 0].float()), cc_cache_v=bits(m.deltakv_full_kv_cache[1, 0].float()),
               cc_kv=bits(kv[0].float()), cc_existing=existing.numpy(), cc_rel=rel.numpy().astype(np.int32),
               cc_topk=topk.numpy(), cc_base=bits(base[0].float()), cc_k=np.array([3], dtype=np.int64))
    # ---- the same through IRREGULAR centre positions (BASELINE configs[4] is "dynamic-stride": explicit positions instead
    # of range(0, n, step); the reference's own kernel test uses these six, tests/test_deltakv_less_memory_kernel.py:129-160)
    m.config = SimpleNamespace(deltakv_k_neighbors=4, cluster_metric="l2", deltakv_cluster_gather_chunk_size=16384)
    n = 37
    kv = (torch.randn(1, n, 2 * Hkv * D, generator=g) * 0.5).to(torch.bfloat16)
    existing = torch.tensor([7, 22, 61], dtype=torch.int32)
    rel = torch.tensor([0, 5, 11, 18, 27, 35], dtype=torch.long)
    topk, base = m._cluster_compress(layer_idx=1, kv_states=kv, existing_center_slots=existing, cluster_step=10,
                                     new_center_rel=rel, validate_centers=False)
    out.update(ci_cache_k=out["cc_cache_k"], ci_cache_v=out["cc_cache_v"], ci_kv=bits(kv[0].float()), ci_existing=existing.numpy(),
               ci_rel=rel.numpy().astype(np.int32), ci_topk=topk.numpy(), ci_base=bits(base[0].float()),
               ci_k=np.array([4], dtype=np.int64))
    save("deltakv_compress", **out)


def gen_prefill_attention():
    """`context_attention_fwd(attn_score=None)` (context_flashattention_nopad.py:242-276) under the interpreter on fp32
    tensors holding bf16 values; two sequences of one chunked-prefill step, one with a cached prefix."""
    from sparsevllm.kernels.triton.context_flashattention_nopad import context_attention_fwd

    g = torch.Generator().manual_seed(47)
    Hq, Hkv, D, slots = 4, 2, 64, 160
    chunk, pc = [37, 20], [11, 0]
    T = sum(chunk)
    q = bf16f(torch.randn(T, Hq, D, generator=g) * 0.5)
    kc = bf16f(torch.randn(slots, Hkv, D, generator=g) * 0.5)
    vc = bf16f(torch.randn(slots, Hkv, D, generator=g) * 0.5)
    table = torch.randperm(slots, generator=g)[: 3 * 50].to(torch.int32).view(3, 50)
    req = torch.tensor([2, 0], dtype=torch.int32)
    start = torch.tensor([0, chunk[0]], dtype=torch.int32)
    seq_len = torch.tensor([chunk[0] + pc[0], chunk[1] + pc[1]], dtype=torch.int32)
    pcl = torch.tensor(pc, dtype=torch.int32)
    o = torch.zeros(T, Hq, D)
    context_attention_fwd(q, kc, vc, o, req, start, seq_len, pcl, max(chunk), table)
    out = dict(q=bits(q), k=bits(kc), v=bits(vc), table=table.numpy(), req=req.numpy(), start=start.numpy(),
               seq_len=seq_len.numpy(), pcl=pcl.numpy(), o=o.numpy())
    # the score-collecting forms (`attn_score=`, :82-240): chunks longer than one 128-row query block
    chunk2, pc2 = [150, 40, 129], [11, 0, 70]
    T2 = sum(chunk2)
    q2 = bf16f(torch.randn(T2, Hq, D, generator=g) * 0.5)
    k2 = bf16f(torch.randn(400, Hkv, D, generator=g) * 0.5)
    table2 = torch.randperm(400, generator=g)[: 3 * 130].to(torch.int32).view(3, 130)      # (row width < a context: see lens)
    table2 = torch.cat([table2, torch.randperm(400, generator=g)[: 3 * 100].to(torch.int32).view(3, 100)], dim=1)
    req2 = torch.tensor([1, 2, 0], dtype=torch.int32)
    start2 = torch.tensor([0, chunk2[0], chunk2[0] + chunk2[1]], dtype=torch.int32)
    seq2 = torch.tensor([c + p_ for c, p_ in zip(chunk2, pc2)], dtype=torch.int32)
    pcl2 = torch.tensor(pc2, dtype=torch.int32)
    L2 = int(seq2.max())
    s3 = torch.zeros(3, Hq, L2)
    o3 = torch.zeros(T2, Hq, D)
    context_attention_fwd(q2, k2, k2, o3, req2, start2, seq2, pcl2, max(chunk2), table2, attn_score=s3)
    s2 = torch.full((3, L2), -1.0e20)
    o2 = torch.zeros(T2, Hq, D)
    context_attention_fwd(q2, k2, k2, o2, req2, start2, seq2, pcl2, max(chunk2), table2, attn_score=s2)
    assert torch.equal(o2, o3)
    out.update(s_q=bits(q2), s_k=bits(k2), s_table=table2.numpy(), s_req=req2.numpy(), s_start=start2.numpy(),
               s_seq_len=seq2.numpy(), s_pcl=pcl2.numpy(), s_score3=s3.numpy(), s_score2=s2.numpy())
    save("prefill_attention", **out)


def gen_h2o_capacity():
    """Scheduler capacity hooks of H2OCacheManager (h2o.py:73-230) on hand-built managers: one case per row of `cases`,
    every hook's answer stored as int64 arrays."""
    from collections import deque
    cases = []
    rng = np.random.default_rng(11)
    for case in range(12):
        L = int(rng.integers(1, 4))
        B = int(rng.integers(2, 5))
        budget = int(rng.choice([8, 16, 32]))
        interval = int(rng.choice([4, 8]))
        prefill_budget = int(rng.choice([16, 24, 48]))
        chunk = int(rng.choice([4, 8, 16]))
        lens = [[int(x) for x in rng.integers(0, budget + interval, B)] for _ in range(L)]
        for li in range(1, L):
            lens[li] = list(lens[0])                       # uniform rows across layers like the real manager
        free_ptr = int(rng.integers(0, 60))
        m = _make_manager(lens, cap=128, nslots=512, budget=budget, interval=interval, prefill_budget=prefill_budget,
                          free_ptr=free_ptr)
        m.config.chunk_prefill_size = chunk
        m.free_rows = [deque(range(int(rng.integers(0, 3)))) for _ in range(L)]
        seqs = []
        for r in range(B):
            prompt = int(rng.integers(1, 200))
            done = int(rng.integers(0, prompt + 1))
            seqs.append(SimpleNamespace(seq_id=r, num_prompt_tokens=prompt, num_prefilled_tokens=done, prefix_cache_hit_len=0))
        waiting = deque(seqs)
        out = dict(L=L, B=B, budget=budget, interval=interval, prefill_budget=prefill_budget, chunk=chunk,
                   lens=np.asarray(lens, np.int64), free=np.asarray(m._num_free_slots, np.int64),
                   free_rows=np.asarray([len(x) for x in m.free_rows], np.int64),
                   prompt=np.asarray([s_.num_prompt_tokens for s_ in seqs], np.int64),
                   done=np.asarray([s_.num_prefilled_tokens for s_ in seqs], np.int64))
        out["admission_cost"] = np.asarray([m.prompt_admission_cost(s_) for s_ in seqs], np.int64)
        out["logical_cost"] = np.asarray([m.prompt_logical_reservation_cost(s_) for s_ in seqs], np.int64)
        out["admission_free"] = np.int64(m.prompt_admission_free_slots())
        out["reserved"] = np.int64(m.reserved_prefill_slots(waiting, chunk))
        out["budget_slots"] = np.int64(m.prompt_admission_budgets(waiting, chunk)["slots"])
        out["costs_slots"] = np.asarray([m.prompt_admission_costs(s_)["slots"] for s_ in seqs], np.int64)
        out["prefill_free"] = np.int64(m.prefill_step_free_slots())
        out["prefill_free_for"] = np.asarray([m.prefill_step_free_slots_for(s_) for s_ in seqs], np.int64)
        out["prefill_cost"] = np.asarray([m.prefill_step_reservation_cost(s_, 7 + i) for i, s_ in enumerate(seqs)], np.int64)
        out["decode_free"] = np.int64(m.decode_step_free_slots())
        out["decode_cost"] = np.asarray([m.decode_step_reservation_cost(s_) for s_ in seqs], np.int64)
        chain = []
        for suffix, gen_t, need_row in ((0, 0, False), (5, 1, True), (40, 30, True), (0, 25, False), (300, 2, True)):
            existing = tuple(int(x) for x in rng.integers(0, budget + interval, L))
            reserved = tuple(int(x) for x in rng.integers(0, 20, L))
            req, rows_req, deficits, row_def = m.chain_capacity_deficits(
                suffix_tokens=suffix, generation_tokens=gen_t, existing_slots_by_layer=existing,
                outstanding_reserved_slots_by_layer=reserved, outstanding_reserved_rows=int(rng.integers(0, 2)),
                needs_resident_row=need_row)
            chain.append(np.concatenate([[suffix, gen_t, int(need_row)], existing, reserved, req, [rows_req], deficits, [row_def]]))
        out["chain"] = np.asarray(chain, np.int64)
        cases.append(out)
    flat = {}
    for i, c in enumerate(cases):
        for k, v in c.items():
            flat[f"c{i}_{k}"] = np.asarray(v)
    flat["n_cases"] = np.int64(len(cases))
    save("h2o_capacity", **flat)


def gen_decode_alloc():
    """Decode slot allocation: (1) SnapKVCacheManager._prepare_decode on NON-uniform layers (per-layer rows /
    lengths / stack pointers, snapkv.py:2656-2673), (2) H2OCacheManager.prepare_decode_static with a graph batch
    larger than the real batch (padded lanes: slot -1, lane 0's metadata, h2o.py:419-437), two consecutive steps."""
    from sparsevllm.engine.cache_manager.snapkv import SnapKVCacheManager
    from sparsevllm.engine.cache_manager.h2o import H2OCacheManager

    out = {}
    seqs = [SimpleNamespace(seq_id=i, decode_input_token=7 + i, decode_input_position=40 + i) for i in range(3)]
    states = lambda L: [SimpleNamespace(slot_mapping=None, context_lens=None, req_indices=None, max_context_len=0)
                        for _ in range(L)]

    # (1) layer 0 is a "full" layer (long rows, fewer free slots), layers 1-2 are compressed
    m = _make_manager([[30, 22, 27], [12, 9, 12], [12, 9, 11]], cls_name="SnapKVCacheManager")
    m._num_free_slots[0] -= 5
    m._num_free_slots[2] -= 1
    m.layer_batch_states = states(3)
    _put(out, "nu_before", _state(m))
    for step in range(2):
        SnapKVCacheManager._prepare_decode(m, seqs)
        out[f"nu{step}_slot_mapping"] = np.stack([st.slot_mapping.numpy().copy() for st in m.layer_batch_states])
        out[f"nu{step}_context_lens"] = np.stack([st.context_lens.numpy().copy() for st in m.layer_batch_states])
        out[f"nu{step}_req_indices"] = np.stack([st.req_indices.numpy().copy() for st in m.layer_batch_states])
        out[f"nu{step}_max_context_len"] = np.array([int(st.max_context_len) for st in m.layer_batch_states])
        _put(out, f"nu{step}_after", _state(m))

    # (2) real batch 3 in a graph batch of 6
    m = _make_manager([[30, 22, 27], [30, 22, 27]], cls_name="H2OCacheManager")
    m.layer_batch_states = states(2)
    m._decode_static_state_binding_key = None
    m._decode_static_buffers = {}
    m._decode_static_max_context_len = None
    m.validate_decode_cuda_graph_slot_mappings = lambda: None
    GB = 6
    _put(out, "pad_before", _state(m))
    for step in range(2):
        ii, pp = torch.zeros(GB, dtype=torch.int64), torch.zeros(GB, dtype=torch.int64)
        sm, cl, ri = (torch.zeros(GB, dtype=torch.int32) for _ in range(3))
        H2OCacheManager.prepare_decode_static(m, seqs, ii, pp, sm, cl, ri)
        out[f"pad{step}_slot_mapping"] = np.stack([st.slot_mapping.numpy().copy() for st in m.layer_batch_states])
        out[f"pad{step}_context_lens"] = np.stack([st.context_lens.numpy().copy() for st in m.layer_batch_states])
        out[f"pad{step}_req_indices"] = np.stack([st.req_indices.numpy().copy() for st in m.layer_batch_states])
        out[f"pad{step}_first"] = np.stack([sm.numpy(), cl.numpy(), ri.numpy()])
        out[f"pad{step}_input_ids"] = ii.numpy()
        out[f"pad{step}_positions"] = pp.numpy()
        _put(out, f"pad{step}_after", _state(m))
    save("decode_alloc", **out)


# ----------------------------------------------------------------------------------
# SnapKV / StreamingLLM selection (torch on CPU) - SparseController functions
# ----------------------------------------------------------------------------------

def _make_controller(*, sink, recent, keep, method="snapkv", cache_manager=None, num_layers=1,
                     model_dtype=torch.bfloat16, head_dim=128):
    """Hand-built SparseController (no engine): the attribute set the selection functions read
    (engine/sparse_controller.py:59-152)."""
    from sparsevllm.engine.sparse_controller import LayerBatchSparseState, SparseController
    c = object.__new__(SparseController)
    c.sparse_method = method
    c.is_deltakv_family = method.startswith("deltakv")
    c.device = torch.device("cpu")
    c.num_sink, c.num_recent, c.decode_keep_tokens = int(sink), int(recent), int(keep)
    c.num_layers = num_layers
    c.config = SimpleNamespace(hf_config=SimpleNamespace(torch_dtype=model_dtype, num_hidden_layers=num_layers),
                               pyramid_layer_ratios=None, snapkv_num_full_layers=0, pool_kernel_size=1,
                               runtime_layout=None, decode_cuda_graph=False)
    c.cache_manager = cache_manager
    c.attn_softmax_scale = float(head_dim) ** -0.5
    c.snapkv_decode_score_dtype = torch.float32
    c.layer_batch_sparse_states = {i: LayerBatchSparseState() for i in range(num_layers)}
    c.dynamic_deltakv_topk_tiebreak = False
    c.debug_dynamic_selection = {}
    c.debug_dynamic_selection_detail = False
    c.validate_runtime_invariants = False
    return c


def _pooled(scores, sink, recent_start, pool):
    import torch.nn.functional as F
    mid = scores[sink:recent_start]
    if pool > 1:
        mid = F.max_pool1d(mid[None, None, :], kernel_size=pool, padding=pool // 2, stride=1)[0, 0]
    return mid


def gen_snapkv_select():
    """SparseController._snapkv_select_indices{,_batch} (sparse_controller.py:1670-1747) with pool_kernel_size 1 and
    > 1, _streamingllm_select_indices (:1661-1668), _snapkv_decode_trigger_len (:2050-2054), _get_layer_budget."""
    out = {}
    cases = [
        # name, kv_len, sink, recent, keep(top), pool, batch, want a tie-free threshold?
        ("a", 90, 4, 8, 20, 1, 3, True),
        ("b", 300, 64, 32, 100, 1, 2, True),
        ("c", 200, 4, 16, 40, 5, 3, True),       # pooled, seed searched so the k-th / (k+1)-th pooled values differ
        ("d", 200, 4, 16, 40, 7, 2, False),      # pooled, ties at the threshold (the reference's pick is recorded)
        ("e", 64, 8, 8, 60, 1, 2, True),         # num_topk > middle length -> clamp
        ("f", 40, 30, 12, 0, 1, 2, True),        # recent_start <= sink -> sink ++ recent only (budget 42 > kv_len is
                                                  # not allowed; use keep=0 with kv_len>budget handled below)
        ("g", 5000, 64, 512, 1024, 1, 2, True),
        ("h", 4300, 64, 512, 1024, 5, 1, True),
    ]
    names = []
    for name, kv_len, sink, recent, keep, pool, batch, tie_free in cases:
        budget = sink + keep + recent
        if kv_len <= budget:
            continue
        c = _make_controller(sink=sink, recent=recent, keep=keep)
        recent_start = kv_len - recent
        num_topk = min(keep, max(0, recent_start - sink))
        seed = 100
        while True:
            g = torch.Generator().manual_seed(seed)
            scores = torch.rand(batch, kv_len + 3, generator=g)           # wider than kv_len like the decode buffer
            ok = True
            if num_topk > 0 and recent_start > sink:
                for b in range(batch):
                    mid = _pooled(scores[b, :kv_len], sink, recent_start, pool)
                    srt = torch.sort(mid, descending=True).values
                    tied = num_topk < mid.numel() and bool(srt[num_topk - 1] == srt[num_topk])
                    ok &= (not tied) if tie_free else True
                if not tie_free:
                    mid = _pooled(scores[0, :kv_len], sink, recent_start, pool)
                    srt = torch.sort(mid, descending=True).values
                    ok = num_topk < mid.numel() and bool(srt[num_topk - 1] == srt[num_topk])
            if ok:
                break
            seed += 1
        keep_b = c._snapkv_select_indices_batch(scores[:, :kv_len], kv_len, budget, pool_kernel_size=pool)
        keep_s = torch.stack([c._snapkv_select_indices(scores[b, :kv_len], kv_len, budget, pool_kernel_size=pool)
                              for b in range(batch)])
        out[f"{name}_scores"] = scores.numpy()
        out[f"{name}_cfg"] = np.array([kv_len, sink, recent, keep, pool, budget, int(tie_free)], dtype=np.int64)
        out[f"{name}_keep_batch"] = keep_b.numpy().astype(np.int64)
        out[f"{name}_keep_scalar"] = keep_s.numpy().astype(np.int64)
        names.append(name)
    out["names"] = np.array(names)
    # trigger lengths / budgets / StreamingLLM keep sets
    trig = []
    for sink, recent, keep in [(4, 8, 20), (64, 512, 4096), (0, 0, 7), (64, 32, 100)]:
        c = _make_controller(sink=sink, recent=recent, keep=keep)
        budget = c._get_layer_budget(0, is_prefill=False)
        trig.append([sink, recent, keep, budget, c._snapkv_decode_trigger_len(budget)])
    out["trigger"] = np.array(trig, dtype=np.int64)
    sl = []
    for sink, recent, kv_len in [(4, 16, 50), (64, 512, 1152), (4, 16, 10), (4, 16, 3), (0, 8, 20), (8, 0, 20)]:
        c = _make_controller(sink=sink, recent=recent, keep=0, method="streamingllm")
        idx = c._streamingllm_select_indices(kv_len).numpy().astype(np.int64)
        out[f"sl_{sink}_{recent}_{kv_len}"] = idx
        sl.append([sink, recent, kv_len, c._get_streamingllm_budget() or -1])
    out["sl_cases"] = np.array(sl, dtype=np.int64)
    save("snapkv_select", **out)


def _make_snapkv_manager(lengths_by_layer, *, cap, nslots, sink, recent, keep, window, heads, dim, slot_seed=11,
                         mode="probability"):
    m = _make_manager(lengths_by_layer, cap=cap, nslots=nslots, heads=heads, dim=dim, slot_seed=slot_seed,
                      cls_name="SnapKVCacheManager")
    m.config.vllm_sparse_method = "snapkv"
    m.config.num_sink_tokens, m.config.num_recent_tokens, m.config.decode_keep_tokens = sink, recent, keep
    m.config.snapkv_window_size = window
    m.config.sparse_prefill_score_mode = mode
    m.config.sparse_attn_score_dtype = "float32"
    m.config.pool_kernel_size = 1
    m._prefill_attn_score_accumulators = {}
    m.kv_layer_index = lambda layer: int(layer)
    return m


def gen_snapkv_e2e():
    """The SnapKV prompt path of the reference end to end on CPU: collect_prefill_attention_score
    (snapkv.py:1216-1304; the real prefill_score_fwd under the Triton interpreter, candidate_start = sink,
    num_recent = recent, elementwise-max accumulator :1017-1044,:1299-1303) -> _snapkv_prefill_eviction
    (sparse_controller.py:1059-1102) -> free_part_slots; then _snapkv_decode_eviction (:1104-1223) on a
    decode-time score buffer (single-row and batched groups, fused batch-layers compaction)."""
    from sparsevllm.engine.cache_manager.base import AttentionViewMeta, ExplicitKVPayload, PrefillComputeView
    from sparsevllm.engine.cache_manager.snapkv import SnapKVCacheManager
    from sparsevllm.utils.context import set_context

    out = {}
    g = torch.Generator().manual_seed(77)
    # ---------------- prefill: two prompts in one final chunk, one of them under budget (not scored, not evicted)
    L, Hq, Hkv, D = 2, 4, 2, 64
    prompts = (90, 30, 77)
    tot = sum(prompts)
    q = bf16f(torch.randn(L, tot, Hq, D, generator=g) * 0.5)
    kc = bf16f(torch.randn(L, 256, Hkv, D, generator=g) * 0.5)
    out["p_q"] = bits(q)
    out["p_k"] = bits(kc)
    for tag, mode, pool in (("pf", "probability", 1), ("pl", "logits", 1), ("pp", "probability", 5)):
        sink, recent, keep, window = 4, 8, 20, 8
        budget = sink + keep + recent
        m = _make_snapkv_manager([[n for n in prompts]] * L, cap=128, nslots=256, sink=sink, recent=recent, keep=keep,
                                 window=window, heads=Hkv, dim=D, mode=mode)
        m.config.pool_kernel_size = pool
        c = _make_controller(sink=sink, recent=recent, keep=keep, cache_manager=m, num_layers=L)
        c.config.pool_kernel_size = pool
        seqs = [SimpleNamespace(seq_id=i, num_prompt_tokens=n, num_prefilled_tokens=0, current_chunk_size=n,
                                is_last_chunk_prefill=True, chain_status="", is_recompute_replay=False,
                                chain_reused_tokens=0) for i, n in enumerate(prompts)]
        set_context(True, seqs=seqs)
        starts = torch.tensor(np.concatenate(([0], np.cumsum(prompts)[:-1])), dtype=torch.int32)
        chunk_lens = torch.tensor(prompts, dtype=torch.int32)
        m._prefill_context_lens_cpu_by_layer = {l: tuple(prompts) for l in range(L)}
        _put(out, f"{tag}_before", _state(m))
        out[f"{tag}_cfg"] = np.array([sink, recent, keep, window, budget, 1 if mode == "logits" else 0, pool], dtype=np.int64)
        out[f"{tag}_prompts"] = np.array(prompts, dtype=np.int64)
        for l in range(L):
            meta = AttentionViewMeta(active_slots=m.buffer_req_to_token_slots[l],
                                     req_indices=torch.arange(len(prompts), dtype=torch.int32),
                                     context_lens=torch.tensor(prompts, dtype=torch.int32), max_context_len=max(prompts))
            view = PrefillComputeView(meta=meta, payload=ExplicitKVPayload(k_cache=kc[l], v_cache=kc[l]))
            SnapKVCacheManager.collect_prefill_attention_score(m, l, q[l], view, b_start_loc=starts, chunk_lens=chunk_lens)
            for i in range(len(prompts)):
                acc = m._prefill_attn_score_accumulators.get((l, i))
                if acc is not None:
                    out[f"{tag}_acc_{l}_{i}"] = acc.numpy().copy()
        out[f"{tag}_scored"] = np.array(sorted(k_[1] for k_ in m._prefill_attn_score_accumulators if k_[0] == 0), dtype=np.int64)
        c._snapkv_prefill_eviction(seqs)
        _put(out, f"{tag}_after", _state(m))
    # ---------------- accumulator across a recompute replay: a second collect on the same prompt keeps the max
    # (num_prefilled_tokens != 0 keeps the accumulator; == 0 resets it)
    # pinned through _get_prefill_attention_score_accumulator directly
    m = _make_snapkv_manager([[10]], cap=16, nslots=32, sink=1, recent=1, keep=2, window=2, heads=1, dim=4)
    for mode in ("probability", "logits"):
        m.config.sparse_prefill_score_mode = mode
        m._prefill_attn_score_accumulators = {}
        seq = SimpleNamespace(seq_id=0, num_prefilled_tokens=0)
        a0 = m._get_prefill_attention_score_accumulator(0, seq, prompt_len=10, device=torch.device("cpu"))
        out[f"acc_init_{mode}"] = a0.numpy().copy()
        s1 = torch.rand(10, generator=g) - 0.5
        torch.maximum(a0, s1, out=a0)
        seq.num_prefilled_tokens = 4
        a1 = m._get_prefill_attention_score_accumulator(0, seq, prompt_len=10, device=torch.device("cpu"))
        s2 = torch.rand(10, generator=g) - 0.5
        torch.maximum(a1, s2, out=a1)
        out[f"acc_steps_{mode}"] = np.stack([s1.numpy(), s2.numpy()])
        out[f"acc_final_{mode}"] = a1.numpy().copy()
        seq.num_prefilled_tokens = 0
        a2 = m._get_prefill_attention_score_accumulator(0, seq, prompt_len=10, device=torch.device("cpu"))
        out[f"acc_reset_{mode}"] = a2.numpy().copy()

    # ---------------- decode re-eviction
    sink, recent, keep = 4, 8, 20
    budget = sink + keep + recent                 # 32, trigger 40
    for tag, lens in (("dg", (40, 40, 33, 40)), ("ds", (41, 35, 40, 12)), ("dn", (39, 38, 33, 12)),
                      ("dm", (40, 41, 40, 12)), ("dq", (41, 40, 40, 41))):
        L = 3
        m = _make_snapkv_manager([list(lens)] * L, cap=64, nslots=300, sink=sink, recent=recent, keep=keep, window=8,
                                 heads=2, dim=4)
        m.decode_kv_lens_for_layer = lambda layer_idx, seqs_, _m=m: [int(_m.row_seq_lens[layer_idx][s.seq_id]) for s in seqs_]
        c = _make_controller(sink=sink, recent=recent, keep=keep, cache_manager=m, num_layers=L)
        seqs = [SimpleNamespace(seq_id=i) for i in range(len(lens))]
        set_context(False, seqs=seqs)
        _put(out, f"{tag}_before", _state(m))
        for l in range(L):
            st = c.layer_batch_sparse_states[l]
            st.attn_score = torch.rand(len(lens), max(lens), generator=g)
            st.max_context_len = max(lens)
            st.context_lens = torch.tensor(lens, dtype=torch.int32)
            out[f"{tag}_score_{l}"] = st.attn_score.numpy().copy()
        c._snapkv_decode_eviction(seqs)
        _put(out, f"{tag}_after", _state(m))
        out[f"{tag}_cfg"] = np.array([sink, recent, keep, budget, c._snapkv_decode_trigger_len(budget)], dtype=np.int64)
    save("snapkv_e2e", **out)


# ----------------------------------------------------------------------------------
# DeltaKV observation-layer token scores + sorted top-k (torch on CPU)
# ----------------------------------------------------------------------------------

def gen_deltakv_topk():
    """SparseController._decode_softmax_token_scores (sparse_controller.py:255-299) and the decode branch of
    _update_dynamic_omnikv_indices (:1755-1822, :1951-1959): masked topk(sorted=True) with and without the
    deterministic tie-break key (:1797-1811)."""
    from sparsevllm.utils.context import set_context
    out = {}
    g = torch.Generator().manual_seed(4242)
    cases = [
        # name, B, H, L, sink, recent, keep, compressed lens, model dtype
        ("a", 3, 28, 700, 8, 16, 128, (600, 130, 40), torch.bfloat16),
        ("b", 2, 8, 300, 4, 8, 64, (280, 0), torch.float32),
        ("c", 2, 28, 5000, 8, 128, 2048, (4800, 2050), torch.bfloat16),
        ("d", 2, 4, 200, 8, 16, 512, (150, 100), torch.bfloat16),         # keep > candidates
        # (a float16 model dtype cannot run this path in the reference: masked_fill_(-1e10) overflows c10::Half)
    ]
    names = []
    for name, B, H, L, sink, recent, keep, clens, mdt in cases:
        raw = bf16f(torch.randn(B, H, L, generator=g) * 6.0)
        clen_t = torch.tensor(clens, dtype=torch.int32)
        cm = SimpleNamespace(get_compressed_lens=lambda req, _c=clen_t: _c.clone())
        res = {}
        for tb in (False, True):
            c = _make_controller(sink=sink, recent=recent, keep=keep, method="deltakv-triton-v4", cache_manager=cm,
                                 num_layers=3, model_dtype=mdt)
            c.dynamic_deltakv_topk_tiebreak = tb
            set_context(False, is_long_text=True)
            ts = c._decode_softmax_token_scores(raw.clone(), candidate_start=sink, candidate_lens=clen_t)
            st = c.layer_batch_sparse_states[0]
            st.attn_score = ts.clone()
            st.context_lens = torch.tensor([sink + cl + recent for cl in clens], dtype=torch.int32)
            st.req_indices = torch.arange(B, dtype=torch.int32)
            c._update_dynamic_omnikv_indices(0, [1, 2])
            res[tb] = (ts, c.layer_batch_sparse_states[1].active_compressed_indices,
                       st.attn_score.clone())
            assert c.layer_batch_sparse_states[2].active_compressed_indices is res[tb][1]
            assert c.layer_batch_sparse_states[2].deltakv_free_temp_slots and not c.layer_batch_sparse_states[1].deltakv_free_temp_slots
        ts = res[False][0]
        out[f"{name}_raw"] = bits(raw)
        out[f"{name}_cfg"] = np.array([sink, recent, keep, {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[mdt]], dtype=np.int64)
        out[f"{name}_clens"] = np.array(clens, dtype=np.int64)
        out[f"{name}_token_scores"] = ts.float().numpy()
        out[f"{name}_topk"] = res[False][1].numpy()
        out[f"{name}_topk_tiebreak"] = res[True][1].numpy()
        # the masked search scores the reference's topk saw (the obs state's buffer is masked in place)
        out[f"{name}_search_masked"] = res[False][2][:, sink:].float().numpy()
        names.append(name)
    out["names"] = np.array(names)
    save("deltakv_topk", **out)


# ----------------------------------------------------------------------------------
# method surface (pure Python tables)
# ----------------------------------------------------------------------------------

def _call(fn, *a, **k):
    try:
        return {"ok": fn(*a, **k)}
    except Exception as e:          # the exception class and text are part of the surface
        return {"err": type(e).__name__, "msg": str(e)}


def gen_method_surface():
    """method_registry.py (aliases, policies, contracts, compatibility) and configs/runtime_params.py
    (public aliases, legacy-name rejection) as a JSON table of inputs -> outputs / error texts."""
    import json
    from sparsevllm import method_registry as mr
    from sparsevllm.configs.runtime_params import normalize_runtime_params
    from sparsevllm.distributed.topology import ParallelTopology

    names = [None, "", " H2O ", "vanilla", "Vanilla", "attention-sink", "attention_sink", "r-kv", "r_kv", "skip-kv", "skip_kv",
             "deltakv", "deltakv-less-memory", "deltakv_less_memory", "deltakv-less-memory-cudagraph",
             "deltakv_less_memory_cudagraph", "streamingllm", "snapkv", "h2o", "pyramidkv", "omnikv", "quest", "rkv",
             "skipkv", "nope", "deltakv-triton-v4", 7]
    policies = [None, "", "auto", "AUTO", "all_chunked", " long_bs1full_short_batch ", "bogus"]
    out = {"names": [n if n is None or isinstance(n, str) else {"int": n} for n in names], "policies": policies}
    rows = []
    compat_tables = {}
    for n in names:
        row = {
            "normalize": _call(mr.normalize_sparse_method, n),
            "default_policy": _call(mr.get_default_prefill_schedule_policy, n),
            "resolve": [_call(mr.resolve_prefill_schedule_policy, n, p) for p in policies],
            "is_deltakv": _call(mr.is_deltakv_method, n),
            "graph": _call(mr.is_decode_cuda_graph_supported, n),
            "tp_graph": _call(mr.is_tp_decode_cuda_graph_supported, n),
        }
        c = _call(mr.sparse_prefill_attention_contract, n)
        if "ok" in c:
            c = {"ok": [c["ok"].main_score_kind.name, c["ok"].score_collection.name]}
        row["contract"] = c
        compat = []
        for model_type in ("qwen2", "Llama", "qwen3_moe", "mystery"):
            for graph in (False, True):
                for prefix in (False, True):
                    r = _call(mr.validate_model_runtime_compatibility, model_type=model_type, sparse_method=n,
                              topology=ParallelTopology(1, 1, 1), decode_cuda_graph=graph, enable_prefix_caching=prefix)
                    if "ok" in r:
                        lists = [sorted(r["ok"].sparse_methods), sorted(r["ok"].prefix_cache_methods),
                                 sorted(r["ok"].decode_cuda_graph_methods)]
                        key = json.dumps(lists)
                        if key not in compat_tables:
                            compat_tables[key] = f"T{len(compat_tables)}"
                        r = {"ok": compat_tables[key]}
                    compat.append([model_type, graph, prefix, r])
        row["compat"] = compat
        rows.append(row)
    out["rows"] = rows
    out["compat_tables"] = {v: json.loads(k) for k, v in compat_tables.items()}
    out["tables"] = {
        "METHOD_ALIASES": {("<None>" if k is None else k): v for k, v in mr.METHOD_ALIASES.items()},
        "CANONICAL_SPARSE_METHODS": sorted(mr.CANONICAL_SPARSE_METHODS),
        "SUPPORTED_SPARSE_METHODS": sorted(mr.SUPPORTED_SPARSE_METHODS),
        "SUPPORTED_SPARSE_METHOD_ALIASES": sorted(mr.SUPPORTED_SPARSE_METHOD_ALIASES),
        "PREFIX_CACHE_SUPPORTED_METHODS": sorted(mr.PREFIX_CACHE_SUPPORTED_METHODS),
        "DECODE_CUDA_GRAPH_SUPPORTED_METHODS": sorted(mr.DECODE_CUDA_GRAPH_SUPPORTED_METHODS),
        "TP_DECODE_CUDA_GRAPH_SUPPORTED_METHODS": sorted(mr.TP_DECODE_CUDA_GRAPH_SUPPORTED_METHODS),
        "PREFILL_POLICY_BY_METHOD": dict(mr.PREFILL_POLICY_BY_METHOD),
        "SUPPORTED_PREFILL_POLICIES": sorted(mr.SUPPORTED_PREFILL_POLICIES),
    }
    kw_cases = [
        {},
        {"sparse_method": "h2o", "h2o_decode_budget": 4096},
        {"sparse_method": "vanilla"},
        {"sparse_method": "attention-sink", "sink_keep_tokens": 4, "recent_keep_tokens": 16},
        {"sparse_method": "deltakv", "deltakv_checkpoint_path": "/x", "deltakv_center_ratio": 0.03, "deltakv_latent_dim": 256,
         "deltakv_latent_quant_bits": 4, "deltakv_latent_quant_group_size": 32, "deltakv_neighbor_count": 4,
         "full_attention_layers": "0,1,2", "engine_prefill_chunk_size": 8192},
        {"sparse_method": "quest", "decode_keep_tokens": 0.5},
        {"sparse_method": "quest", "decode_keep_tokens": 4096, "quest_chunk_size": 16},
        {"sink_keep_tokens": 4, "num_sink_tokens": 4},
        {"vllm_sparse_method": "h2o"},
        {"num_top_tokens": 3, "chunk_prefill_size": 5},
        {"seq_chunk_size": 1},
        {"deltakv_visual_keep_ratio": 0.5, "ref_mode": "x"},
        {"quest_token_budget": 100},
        {"k_neighbors": 4, "cluster_ratio": 0.1, "kv_quant_bits": 4},
    ]
    res = []
    for kw in kw_cases:
        r = _call(normalize_runtime_params, dict(kw), backend="sparsevllm")
        if "ok" in r:
            r = {"ok": r["ok"].infer_config}
        res.append([kw, r])
    # (every alias target is itself a legacy key, so the "conflicting alias" branch of _set_alias is unreachable from
    # the public API: the legacy-name rejection fires first - see the {"sink_keep_tokens", "num_sink_tokens"} case)
    out["runtime_params"] = res
    path = os.path.join(HERE, "method_surface.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def gen_capacity_others():
    """Scheduler capacity hooks of the SnapKV / StreamingLLM / QuEST managers (snapkv.py:761-905, streamingllm.py:24-32,
    quest.py:272-378 without a prefix cache, base.py:1242-1397 defaults) on hand-built managers, as JSON."""
    import json
    from collections import deque
    from sparsevllm.engine.cache_manager.quest import QuestCacheManager
    from sparsevllm.engine.cache_manager.snapkv import SnapKVCacheManager
    from sparsevllm.engine.cache_manager.streamingllm import StreamingLLMCacheManager
    rng = np.random.default_rng(23)
    hooks = ("prompt_admission_cost", "prompt_logical_reservation_cost", "prefill_step_free_slots_for",
             "decode_step_free_slots_for", "decode_step_reservation_cost", "remaining_prefill_tokens",
             "min_final_prefill_chunk_size")
    out = []

    def record(kind, m, seqs, extra):
        waiting = deque(seqs)
        rec = dict(kind=kind, **extra,
                   seqs=[[int(s.seq_id), int(s.num_prompt_tokens), int(s.num_prefilled_tokens)] for s in seqs])
        for h in hooks:
            rec[h] = [int(getattr(m, h)(s)) for s in seqs]
        rec["prefill_step_reservation_cost"] = [int(m.prefill_step_reservation_cost(s, 5 + 7 * i)) for i, s in enumerate(seqs)]
        rec["prompt_admission_free_slots"] = int(m.prompt_admission_free_slots())
        rec["prefill_step_free_slots"] = int(m.prefill_step_free_slots())
        rec["decode_step_free_slots"] = int(m.decode_step_free_slots())
        rec["reserved_prefill_slots"] = int(m.reserved_prefill_slots(waiting, 8))
        rec["prompt_admission_budgets"] = {k: int(v) for k, v in m.prompt_admission_budgets(waiting, 8).items()}
        rec["prompt_admission_costs"] = [{k: int(v) for k, v in m.prompt_admission_costs(s).items()} for s in seqs]
        rec["prefill_batched_tokens_margin"] = int(m.prefill_batched_tokens_margin())
        rec["prompt_admission_failure_action"] = m.prompt_admission_failure_action()
        out.append(rec)

    def mkseqs(n):
        res = []
        for r in range(n):
            prompt = int(rng.integers(1, 300))
            done = int(rng.choice([0, prompt, int(rng.integers(0, prompt + 1))]))
            res.append(SimpleNamespace(seq_id=r, num_prompt_tokens=prompt, num_prefilled_tokens=done, prefix_cache_hit_len=0,
                                       chain_status="", is_recompute_replay=False, chain_reused_tokens=0))
        return res

    for case in range(6):
        B = int(rng.integers(2, 5))
        lens = [int(x) for x in rng.integers(0, 90, B)]
        free_ptr = int(rng.integers(0, 80))
        sink, recent, keep, window = int(rng.choice([4, 64])), int(rng.choice([8, 32])), int(rng.choice([20, 100])), int(rng.choice([0, 8, 32]))
        full_layers = int(rng.choice([0, 0, 2]))
        for kind, cls in (("snapkv", SnapKVCacheManager), ("streamingllm", StreamingLLMCacheManager)):
            m = _make_manager([lens, lens], cap=128, nslots=512, free_ptr=free_ptr, cls_name="SnapKVCacheManager")
            m.__class__ = cls
            m.config.vllm_sparse_method = kind
            m.config.num_sink_tokens, m.config.num_recent_tokens, m.config.decode_keep_tokens = sink, recent, keep
            m.config.snapkv_window_size = window
            m.config.snapkv_num_full_layers = full_layers
            m.config.pyramid_layer_ratios = None
            m.kv_layer_index = lambda layer: int(layer)
            m.kv_transformer_layer_indices = lambda: (0, 1)
            m.chain_physical_kv_len = lambda layer, seq_id: 0
            record(kind, m, mkseqs(B), dict(lens=lens, free=[int(x) for x in m._num_free_slots], sink=sink, recent=recent,
                                            keep=keep, window=window, full_layers=full_layers))
    for case in range(6):
        B = int(rng.integers(2, 5))
        page = int(rng.choice([4, 16]))
        m = object.__new__(QuestCacheManager)
        m.page_size = page
        m.prefix_cache = None
        m.row_seq_lens = np.asarray([int(x) for x in rng.integers(0, 100, B + 1)], dtype=np.int32)
        m.seq_id_to_row = {r: r for r in range(B - 1)}          # the last sequence has no row yet
        m._num_free_pages = int(rng.integers(0, 9))
        m.config = SimpleNamespace(vllm_sparse_method="quest")
        record("quest", m, mkseqs(B), dict(page=page, lens=[int(x) for x in m.row_seq_lens], free_pages=int(m._num_free_pages),
                                           rows={str(k): int(v) for k, v in m.seq_id_to_row.items()}))
    path = os.path.join(HERE, "capacity_others.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


# ----------------------------------------------------------------------------------
# Attention.forward prefill branch: hook order + arguments (layers/attention.py:88-161)
# ----------------------------------------------------------------------------------
def gen_attention_hooks():
    """(A) The reference's `Attention.forward` prefill branch run on recording stand-ins of the cache manager / sparse
    controller (tests/hook_trace.py - shared with the test of this build's mirror) under the reference's fake-attention
    switches: the hook sequence, argument shapes / values, results and exception texts.  (B) The same branch over a REAL
    hand-built H2OCacheManager for two prompt chunks x two layers with `_run_prefill_score` replaced by a deterministic
    stand-in (as tests/test_h2o_cache_manager.py:418-448 does): what the scoring launch is asked for and the cumulative
    H2O score rows afterwards."""
    import json
    sys.path.insert(0, os.path.dirname(HERE))
    import hook_trace as ht
    from sparsevllm.engine.cache_manager import base as rb
    from sparsevllm.layers.attention import Attention
    from sparsevllm.utils.context import get_context, set_context

    types = SimpleNamespace(SparseSelection=rb.SparseSelection, AttentionViewMeta=rb.AttentionViewMeta,
                            ExplicitKVPayload=rb.ExplicitKVPayload, PrefillComputeView=rb.PrefillComputeView,
                            DecodeComputeView=rb.DecodeComputeView)

    def install(is_prefill, cu, cm, sc, layer, seqs=None):
        set_context(is_prefill, cu_seqlens_q=cu, cache_manager=cm, seqs=seqs)
        ctx = get_context()
        ctx.sparse_controller = sc
        ctx.now_layer_idx = layer

    out = {"cases": {c["name"]: ht.run_case(c, attention_cls=Attention, types=types, install_context=install)
                     for c in ht.CASES}}
    # (C) the decode branch (layers/attention.py:162-250) on recording stand-ins, and the backend's own slot check
    # (layers/attention_backend.py:397-439) on hand-built views
    out["decode_cases"] = {c["name"]: ht.run_decode_case(c, attention_cls=Attention, types=types, install_context=install)
                           for c in ht.DECODE_CASES}
    backend = Attention(4, 4, 0.5, 2).attention_backend
    out["bounds_cases"] = {c["name"]: ht.run_bounds_case(c, backend=backend, types=types) for c in ht.BOUNDS_CASES}

    # ---- (B)
    F = ht.H2O_FLOW
    L, B = F["layers"], len(F["chunks"][0])
    m = _make_manager([[0] * B] * L, cap=96, nslots=256, budget=48, interval=16, prefill_budget=64, heads=F["kv_heads"],
                      dim=F["dim"])
    m.config.h2o_prefill_score_window = F["window"]
    m._pyramidkv_prefill_staging_active = False          # PyramidKV staging (out of scope) is off: snapkv.py:533-539
    c = _make_controller(sink=0, recent=0, keep=0, method="h2o", cache_manager=m, num_layers=L)
    calls = []
    m._run_prefill_score = ht.fake_prefill_score_fn(calls)
    os.environ["SPARSEVLLM_FAKE_ATTENTION"] = "1"
    os.environ["SPARSEVLLM_ALLOW_FAKE_ATTENTION"] = "1"
    try:
        attn = Attention(F["heads"], F["dim"], F["dim"] ** -0.5, F["kv_heads"])
        done = [0] * B
        for chunk in F["chunks"]:
            seqs = [SimpleNamespace(seq_id=i, num_prompt_tokens=sum(ch[i] for ch in F["chunks"]), num_prefilled_tokens=done[i],
                                    current_chunk_size=n) for i, n in enumerate(chunk)]
            cu = torch.tensor(np.concatenate(([0], np.cumsum(chunk))), dtype=torch.int32)
            for l in range(L):
                for i, n in enumerate(chunk):          # the chunk is appended to the physical row (slots are irrelevant here)
                    m.row_seq_lens[l][i] += n
                ctx_lens = [int(m.row_seq_lens[l][i]) for i in range(B)]
                st = c.layer_batch_sparse_states[l]
                st.context_lens = torch.tensor(ctx_lens, dtype=torch.int32)
                st.req_indices = torch.arange(B, dtype=torch.int32)
                st.global_req_indices = st.req_indices
                st.max_context_len = max(ctx_lens)
                st.attn_score = None
            m._prefill_context_lens_cpu_by_layer = {l: tuple(int(x) for x in m.row_seq_lens[l][:B]) for l in range(L)}
            n_tok = int(cu[-1])
            q = torch.zeros((n_tok, F["heads"], F["dim"]), dtype=torch.bfloat16)
            kv = torch.zeros((n_tok, F["kv_heads"], F["dim"]), dtype=torch.bfloat16)
            for l in range(L):
                install(True, cu, m, c, l, seqs=seqs)
                attn(q, kv, kv)
            for i, n in enumerate(chunk):
                done[i] += n
    finally:
        os.environ.pop("SPARSEVLLM_FAKE_ATTENTION", None)
        os.environ.pop("SPARSEVLLM_ALLOW_FAKE_ATTENTION", None)
    out["h2o_flow"] = {"calls": calls,
                       "scores": {f"{l}_{i}": [float(x) for x in m._h2o_scores[(l, i)].tolist()] for l in range(L) for i in range(B)}}
    path = os.path.join(HERE, "attention_hooks.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def _registry_namespace():
    """The reference's operator-registry / platform modules as the namespace tests/registry_scenarios.py drives."""
    import sparsevllm.operators.decode_attention as da
    import sparsevllm.operators.registry as reg
    import sparsevllm.platforms as platforms
    from sparsevllm.platforms import interface as pi
    return SimpleNamespace(
        OpRegistry=reg.OpRegistry, OpResolver=reg.OpResolver, SupportResult=reg.SupportResult,
        record_operator_binding=reg.record_operator_binding, operator_runtime_stats=reg.operator_runtime_stats,
        runtime_version_at_least=reg.runtime_version_at_least, bindings=reg._OPERATOR_BINDINGS,
        DeviceCaps=pi.DeviceCaps, PlatformEnum=pi.PlatformEnum, AllocatorStats=pi.AllocatorStats, Platform=pi.Platform,
        platforms=platforms, DecodeAttentionLaunchSpec=da.DecodeAttentionLaunchSpec,
        DecodeAttentionLaunchProvider=da.DecodeAttentionLaunchProvider,
        DefaultGqaDecodeLaunchProvider=da.DefaultGqaDecodeLaunchProvider,
        DECODE_ATTENTION_LAUNCH_REGISTRY=da.DECODE_ATTENTION_LAUNCH_REGISTRY,
        PreparedDecodeAttentionLaunchOp=da.PreparedDecodeAttentionLaunchOp,
        prepare_decode_attention_launch_op=da.prepare_decode_attention_launch_op)


def gen_operator_registry():
    """The reference's OpRegistry / OpResolver / SupportResult / DeviceCaps / Platform / decode-attention-launch family
    answering tests/registry_scenarios.py (the cases of its own tests/test_operator_registry.py and tests/test_platforms.py,
    plus field tables of the boundary dataclasses)."""
    import json
    sys.path.insert(0, os.path.dirname(HERE))
    import registry_scenarios as rs
    out = rs.run_all(_registry_namespace())
    path = os.path.join(HERE, "operator_registry.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def gen_step_planner():
    """The reference's `Scheduler.schedule` (engine/scheduler.py:398-792) answering tests/planner_scenarios.py: scripted
    MemoryOracle, plain sequences; per step which sequences run with which chunk sizes, the queue orders, exception texts."""
    import json
    sys.path.insert(0, os.path.dirname(HERE))
    import planner_scenarios as ps
    from sparsevllm.engine.scheduler import Scheduler
    out = ps.run_all(SimpleNamespace(make=lambda cfg, oracle: Scheduler(cfg, oracle)))
    path = os.path.join(HERE, "step_planner.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


# ----------------------------------------------------------------------------------
# planned run: the reference's Scheduler over its own cache managers through the engine's step order (capacity only)
# ----------------------------------------------------------------------------------
def _planned_run_loop(sc, sch, m, seqs, *, prepare_prefill, prepare_decode, after_prefill, after_decode, snapshot):
    base = seqs[0].seq_id
    trace = []
    for s in seqs:
        sch.add(s)
    while not sch.is_finished():
        assert len(trace) < 2000
        chosen, is_prefill, preempted = sch.schedule()
        assert chosen and not preempted
        rec = dict(prefill=bool(is_prefill), seqs=[[s.seq_id - base, int(s.current_chunk_size) if is_prefill else 1] for s in chosen])
        if is_prefill:
            prepare_prefill(chosen)
            after_prefill(chosen)
        else:
            prepare_decode(chosen)
            after_decode(chosen)
        sch.postprocess(chosen, [0] * len(chosen), is_prefill)
        rec.update(snapshot(base))                    # before the finished rows are released
        rec["finished"] = [s.seq_id - base for s in chosen if s.is_finished]
        for s in chosen:
            if s.is_finished:
                m.free_seq(s.seq_id)
        rec["waiting"] = [s.seq_id - base for s in sch.waiting]
        rec["decoding"] = [s.seq_id - base for s in sch.decoding]
        rec["deferred"] = sorted(int(x) - base for x in sch._admission_defer_warned_seq_ids)
        trace.append(rec)
    return trace


def gen_planned_run():
    """tests/planned_run_scenarios.py through the reference: `Scheduler.schedule / postprocess` over the reference's
    H2OCacheManager (`_prepare_prefill`, `evict_after_prefill`, `_prepare_decode`, `evict_after_decode`, `free_seq`;
    arbitrary scores - counts do not depend on them) and QuestCacheManager (`_prepare_prefill`, `_prepare_decode`,
    `free_seq`; page tables recorded) on CPU."""
    import json
    from collections import deque
    sys.path.insert(0, os.path.dirname(HERE))
    import planned_run_scenarios as prs
    from sparsevllm.engine.cache_manager.base import LayerBatchStates
    from sparsevllm.engine.cache_manager.h2o import H2OCacheManager
    from sparsevllm.engine.cache_manager.quest import QuestCacheManager
    from sparsevllm.engine.scheduler import Scheduler
    from sparsevllm.engine.sequence import Sequence
    from sparsevllm.sampling_params import SamplingParams
    out = {}

    def sched_cfg(sc):
        return SimpleNamespace(prefill_schedule_policy="all_chunked", eos=-1, eos_token_ids=(), num_sink_tokens=sc["sink"],
                               num_recent_tokens=sc["recent"], decode_keep_tokens=sc["keep"], snapkv_window_size=sc.get("window", 4),
                               vllm_sparse_method=sc["method"], **sc["planner"])

    def mkseqs(sc):
        return [Sequence(list(range(n)), SamplingParams(max_tokens=g, ignore_eos=True)) for n, g in zip(sc["prompts"], sc["gens"])]

    # ---- H2O
    sc = prs.H2O
    L = sc["layers"]
    m = _make_manager([[0] * sc["rows"]] * L, cap=sc["max_model_len"], nslots=sc["slots"], budget=sc["budget"],
                      interval=sc["interval"], prefill_budget=sc["prefill_budget"])
    m.seq_id_to_row = [dict() for _ in range(L)]
    m.free_rows = [deque(range(sc["rows"])) for _ in range(L)]
    m.config.chunk_prefill_size = sc["planner"]["chunk_prefill_size"]
    m.config.h2o_recent_ratio = sc["recent_ratio"]
    m.layer_batch_states = [SimpleNamespace(slot_mapping=None, context_lens=None, req_indices=None, max_context_len=0) for _ in range(L)]
    # what `free_seq` touches besides the slot bookkeeping (snapkv.py:1489-1514): none of it exists without PyramidKV staging
    m._prefill_attn_score_accumulators = {}
    m._pyramidkv_clear_long_prefill_offload_prefetch = lambda: None
    m._pyramidkv_long_prefill_offload_kind = lambda: "none"
    m.raw_kv_offload_buffer = SimpleNamespace(release_layer=lambda **k: None)
    seqs = mkseqs(sc)

    def fake_scores(chosen):
        for l in m.kv_transformer_layer_indices():
            for s in chosen:
                n = int(m.row_seq_lens[l][m.seq_id_to_row[l][s.seq_id]])
                g = torch.Generator().manual_seed(1000 * l + 7 * s.seq_id + n)
                m._h2o_scores[m._score_key(l, s.seq_id)] = torch.rand(n, generator=g)

    def snap_h2o(base):
        lens = {}
        for s in seqs:
            if s.seq_id in m.seq_id_to_row[0]:
                per_layer = [int(m.row_seq_lens[l][m.seq_id_to_row[l][s.seq_id]]) for l in range(L)]
                assert len(set(per_layer)) == 1
                lens[str(s.seq_id - base)] = per_layer[0]
        return dict(free=[int(x) for x in m._num_free_slots], lens=lens, counters={k: int(v) for k, v in m._h2o_counters.items()})

    trace = _planned_run_loop(
        sc, Scheduler(sched_cfg(sc), m), m, seqs,
        prepare_prefill=lambda ch: m._prepare_prefill(ch),
        prepare_decode=lambda ch: H2OCacheManager._prepare_decode(m, ch),
        after_prefill=lambda ch: (fake_scores(ch), m.evict_after_prefill(ch)),
        after_decode=lambda ch: (fake_scores(ch), m.evict_after_decode(ch)), snapshot=snap_h2o)
    out["h2o"] = dict(trace=trace)

    # ---- Quest
    sc = prs.QUEST
    page, rows, n_pages = sc["page"], sc["rows"], sc["pages"]
    q = object.__new__(QuestCacheManager)
    q.device = torch.device("cpu")
    q.page_size, q.max_model_len = page, sc["max_model_len"]
    q.max_pages_per_row = sc["max_model_len"] // page
    q.num_pages = n_pages
    q.num_layers = q.num_kv_layers = sc["layers"]
    q.buffer_req_to_token_slots = torch.zeros(rows, sc["max_model_len"], dtype=torch.int32)
    q.buffer_req_to_page_slots = torch.full((rows, q.max_pages_per_row), -1, dtype=torch.int32)
    q.buffer_req_to_page_slots_cpu = np.full((rows, q.max_pages_per_row), -1, dtype=np.int32)
    perm = torch.randperm(n_pages, generator=torch.Generator().manual_seed(5)).to(torch.int32)
    q.free_pages_stack = perm.clone()
    q.free_pages_cpu_stack = perm.numpy().copy()
    q._num_free_pages = n_pages
    q.page_offsets_i32 = torch.arange(page, dtype=torch.int32)
    q.page_offsets_i64 = q.page_offsets_i32.to(torch.int64)
    q.enable_prefix_caching, q.prefix_cache, q.prefix_offload_controller = False, None, None
    q.seq_id_to_row, q.free_rows = {}, deque(range(rows))
    q.row_seq_lens = np.zeros((rows,), dtype=np.int32)
    q.layer_batch_state = LayerBatchStates()
    q.seq_id_to_cached_pages, q.seq_id_to_prefix_blocks, q.seq_id_to_materialized_blocks = {}, {}, {}
    q.prefix_runtime_states, q.pending_prefix_blocks = {}, {}
    q.config = SimpleNamespace(vllm_sparse_method="quest", quest_skip_layers=sc["skip_layers"], quest_token_budget=sc["token_budget"])
    for name in ("_poll_prefix_offload", "_attach_prefix_cache_if_needed", "_record_prefix_materialization",
                 "_release_prefix_blocks", "_schedule_write_through_prefix_blocks", "_evict_prefix_cache_until_free"):
        setattr(q, name, lambda *a, **k: None)
    qseqs = mkseqs(sc)

    def snap_quest(base):
        lens, tables = {}, {}
        for s in qseqs:
            r = q.seq_id_to_row.get(s.seq_id)
            if r is not None:
                n = int(q.row_seq_lens[r])
                lens[str(s.seq_id - base)] = n
                tables[str(s.seq_id - base)] = [int(x) for x in q.buffer_req_to_page_slots_cpu[r, : (n + page - 1) // page]]
        return dict(free_pages=int(q._num_free_pages), free_slots=int(q.num_free_slots), lens=lens, page_tables=tables)

    trace = _planned_run_loop(
        sc, Scheduler(sched_cfg(sc), q), q, qseqs,
        prepare_prefill=lambda ch: q._prepare_prefill(ch), prepare_decode=lambda ch: q._prepare_decode(ch),
        after_prefill=lambda ch: None, after_decode=lambda ch: None, snapshot=snap_quest)
    out["quest"] = dict(trace=trace, free_pages_stack=[int(x) for x in perm])

    # ---- StreamingLLM (sink + recent window): the controller's eviction functions over the SnapKV-family manager
    from sparsevllm.engine.cache_manager.snapkv import SnapKVCacheManager
    from sparsevllm.engine.cache_manager.streamingllm import StreamingLLMCacheManager
    from sparsevllm.utils.context import set_context
    sc = prs.STREAMINGLLM
    L = sc["layers"]
    sm = _make_snapkv_manager([[0] * sc["rows"]] * L, cap=sc["max_model_len"], nslots=sc["slots"], sink=sc["sink"],
                              recent=sc["recent"], keep=sc["keep"], window=4, heads=2, dim=4)
    sm.__class__ = StreamingLLMCacheManager
    sm.config.vllm_sparse_method = "streamingllm"
    sm.config.chunk_prefill_size = sc["planner"]["chunk_prefill_size"]
    sm.seq_id_to_row = [dict() for _ in range(L)]
    sm.free_rows = [deque(range(sc["rows"])) for _ in range(L)]
    sm.layer_batch_states = [SimpleNamespace(slot_mapping=None, context_lens=None, req_indices=None, max_context_len=0) for _ in range(L)]
    sm._pyramidkv_clear_long_prefill_offload_prefetch = lambda: None
    sm._pyramidkv_long_prefill_offload_kind = lambda: "none"
    sm.raw_kv_offload_buffer = SimpleNamespace(release_layer=lambda **k: None)
    ctl = _make_controller(sink=sc["sink"], recent=sc["recent"], keep=sc["keep"], method="streamingllm", cache_manager=sm, num_layers=L)
    sseqs = mkseqs(sc)
    initial_stack = sm.free_slots_stack_tensor.numpy().copy()

    def sync_states(is_prefill):                                  # what SparseController.prepare_forward does for these fields
        for l in range(L):
            st, bs = ctl.layer_batch_sparse_states[l], sm.layer_batch_states[l]
            st.context_lens = bs.context_lens.clone() if is_prefill else bs.context_lens
            st.max_context_len, st.req_indices = bs.max_context_len, bs.req_indices

    def s_prefill(ch):
        set_context(True, seqs=ch)
        sm._prepare_prefill(ch)
        sync_states(True)

    def s_decode(ch):
        set_context(False, seqs=ch)
        SnapKVCacheManager._prepare_decode(sm, ch)
        sync_states(False)

    def snap_sllm(base):
        lens, tables = {}, {}
        for s_ in sseqs:
            if s_.seq_id in sm.seq_id_to_row[0]:
                rows_l = [sm.seq_id_to_row[l][s_.seq_id] for l in range(L)]
                n = int(sm.row_seq_lens[0][rows_l[0]])
                assert all(int(sm.row_seq_lens[l][rows_l[l]]) == n for l in range(L))
                lens[str(s_.seq_id - base)] = n
                tables[str(s_.seq_id - base)] = [[int(x) for x in sm.buffer_req_to_token_slots[l][rows_l[l], :n]] for l in range(L)]
        import zlib
        return dict(free=[int(x) for x in sm._num_free_slots], lens=lens, slot_tables=tables,
                    free_stack_crc=[int(zlib.crc32(sm.free_slots_stack[l][: int(sm._num_free_slots[l])].numpy().astype(np.int32).tobytes()))
                                    for l in range(L)])

    trace = _planned_run_loop(
        sc, Scheduler(sched_cfg(sc), sm), sm, sseqs, prepare_prefill=s_prefill, prepare_decode=s_decode,
        after_prefill=lambda ch: ctl._streamingllm_prefill_eviction(ch), after_decode=lambda ch: ctl._streamingllm_decode_eviction(ch),
        snapshot=snap_sllm)
    out["streamingllm"] = dict(trace=trace, initial_free_stack=[[int(x) for x in initial_stack[l]] for l in range(L)])

    # ---- SnapKV: final-chunk selection and decode re-eviction through the controller, arbitrary scores (counts only)
    sc = prs.SNAPKV
    L = sc["layers"]
    km = _make_snapkv_manager([[0] * sc["rows"]] * L, cap=sc["max_model_len"], nslots=sc["slots"], sink=sc["sink"],
                              recent=sc["recent"], keep=sc["keep"], window=sc["window"], heads=2, dim=4)
    km.config.chunk_prefill_size = sc["planner"]["chunk_prefill_size"]
    km.seq_id_to_row = [dict() for _ in range(L)]
    km.free_rows = [deque(range(sc["rows"])) for _ in range(L)]
    km.layer_batch_states = [SimpleNamespace(slot_mapping=None, context_lens=None, req_indices=None, max_context_len=0) for _ in range(L)]
    km._pyramidkv_clear_long_prefill_offload_prefetch = lambda: None
    km._pyramidkv_long_prefill_offload_kind = lambda: "none"
    km.raw_kv_offload_buffer = SimpleNamespace(release_layer=lambda **k: None)
    km.decode_kv_lens_for_layer = lambda layer_idx, seqs_, _m=km: [int(_m.row_seq_lens[layer_idx][_m.seq_id_to_row[layer_idx][s_.seq_id]])
                                                                  for s_ in seqs_]
    kctl = _make_controller(sink=sc["sink"], recent=sc["recent"], keep=sc["keep"], method="snapkv", cache_manager=km, num_layers=L)
    kseqs = mkseqs(sc)
    kg = torch.Generator().manual_seed(5)

    def k_sync(is_prefill):
        for l in range(L):
            st, bs = kctl.layer_batch_sparse_states[l], km.layer_batch_states[l]
            st.context_lens = bs.context_lens.clone() if is_prefill else bs.context_lens
            st.max_context_len, st.req_indices = bs.max_context_len, bs.req_indices

    def k_prefill(ch):
        set_context(True, seqs=ch)
        km._prepare_prefill(ch)
        k_sync(True)

    def k_decode(ch):
        set_context(False, seqs=ch)
        SnapKVCacheManager._prepare_decode(km, ch)
        k_sync(False)

    def k_after_prefill(ch):
        for l in range(L):
            for s_ in ch:
                if s_.is_last_chunk_prefill:
                    n = int(s_.num_prefilled_tokens) + int(s_.current_chunk_size)
                    km._prefill_attn_score_accumulators[(l, s_.seq_id)] = torch.rand(n, generator=kg)
        kctl._snapkv_prefill_eviction(ch)

    def k_after_decode(ch):
        for l in range(L):
            st = kctl.layer_batch_sparse_states[l]
            st.attn_score = torch.rand(len(ch), int(st.max_context_len), generator=kg)
        kctl._snapkv_decode_eviction(ch)

    def snap_snapkv(base):
        lens = {}
        for s_ in kseqs:
            if s_.seq_id in km.seq_id_to_row[0]:
                per_layer = [int(km.row_seq_lens[l][km.seq_id_to_row[l][s_.seq_id]]) for l in range(L)]
                assert len(set(per_layer)) == 1
                lens[str(s_.seq_id - base)] = per_layer[0]
        return dict(free=[int(x) for x in km._num_free_slots], lens=lens)

    trace = _planned_run_loop(sc, Scheduler(sched_cfg(sc), km), km, kseqs, prepare_prefill=k_prefill, prepare_decode=k_decode,
                              after_prefill=k_after_prefill, after_decode=k_after_decode, snapshot=snap_snapkv)
    out["snapkv"] = dict(trace=trace)

    for name, t in out.items():
        kinds = [r["prefill"] for r in t["trace"]]
        print(name, "steps", len(kinds), "prefill", sum(kinds), "deferred steps", sum(1 for r in t["trace"] if r["deferred"]))
    path = os.path.join(HERE, "planned_run.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=None, sort_keys=True, separators=(",", ":"))
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


GROUPS = {
    "decode_alloc": gen_decode_alloc,
    "decode": gen_decode,
    "h2o_select": gen_h2o_select,
    "h2o_scores": gen_h2o_scores,
    "compaction": gen_compaction,
    "h2o_burst": gen_h2o_burst,
    "h2o_capacity": gen_h2o_capacity,
    "quest": gen_quest,
    "prefill_score": gen_prefill_score,
    "deltakv": gen_deltakv,
    "kivi": gen_kivi,
    "deltakv_view": gen_deltakv_view,
    "deltakv_compress": gen_deltakv_compress,
    "prefill_attention": gen_prefill_attention,
    "snapkv_select": gen_snapkv_select,
    "snapkv_e2e": gen_snapkv_e2e,
    "deltakv_topk": gen_deltakv_topk,
    "method_surface": gen_method_surface,
    "capacity_others": gen_capacity_others,
    "attention_hooks": gen_attention_hooks,
    "operator_registry": gen_operator_registry,
    "step_planner": gen_step_planner,
    "planned_run": gen_planned_run,
}


def main(argv):
    names = argv or list(GROUPS)
    for n in names:
        GROUPS[n]()


if __name__ == "__main__":
    main(sys.argv[1:])
