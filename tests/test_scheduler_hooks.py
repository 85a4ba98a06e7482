"""Scheduler capacity hooks of the H2O cache manager (reference h2o.py:73-230, base.py:1290-1393)
against answers recorded from the reference itself on hand-built managers (tests/golden/gen_fixtures.py h2o_capacity).
Host logic only: no GPU, no kernels."""

from collections import deque
from types import SimpleNamespace

import numpy as np


def _manager(g, i):
    from sparse_vllm_amd.engine.cache_manager.h2o import H2OCacheManager
    L, B = int(g[f"c{i}_L"]), int(g[f"c{i}_B"])
    m = object.__new__(H2OCacheManager)
    m.num_layers = L
    m.config = SimpleNamespace(h2o_decode_budget=int(g[f"c{i}_budget"]), h2o_decode_eviction_interval=int(g[f"c{i}_interval"]),
                               h2o_prefill_budget=int(g[f"c{i}_prefill_budget"]), h2o_recent_ratio=0.5,
                               chunk_prefill_size=int(g[f"c{i}_chunk"]))
    m._num_free_slots = [int(x) for x in g[f"c{i}_free"]]
    m.row_seq_lens = [np.asarray(x, dtype=np.int32) for x in g[f"c{i}_lens"]]
    m.seq_id_to_row = [{r: r for r in range(B)} for _ in range(L)]
    m.free_rows = [deque(range(int(n))) for n in g[f"c{i}_free_rows"]]
    return m, L, B


def test_h2o_capacity_hooks_match_reference(golden):
    g = golden("h2o_capacity")
    for i in range(int(g["n_cases"])):
        m, L, B = _manager(g, i)
        chunk = int(g[f"c{i}_chunk"])
        seqs = [SimpleNamespace(seq_id=r, num_prompt_tokens=int(g[f"c{i}_prompt"][r]), num_prefilled_tokens=int(g[f"c{i}_done"][r]),
                                prefix_cache_hit_len=0) for r in range(B)]
        waiting = deque(seqs)
        assert [m.prompt_admission_cost(s) for s in seqs] == list(g[f"c{i}_admission_cost"])
        assert [m.prompt_logical_reservation_cost(s) for s in seqs] == list(g[f"c{i}_logical_cost"])
        assert m.prompt_admission_free_slots() == int(g[f"c{i}_admission_free"])
        assert m.reserved_prefill_slots(waiting, chunk) == int(g[f"c{i}_reserved"])
        assert m.prompt_admission_budgets(waiting, chunk) == {"slots": int(g[f"c{i}_budget_slots"])}
        assert [m.prompt_admission_costs(s)["slots"] for s in seqs] == list(g[f"c{i}_costs_slots"])
        assert m.prefill_step_free_slots() == int(g[f"c{i}_prefill_free"])
        assert [m.prefill_step_free_slots_for(s) for s in seqs] == list(g[f"c{i}_prefill_free_for"])
        assert [m.prefill_step_reservation_cost(s, 7 + j) for j, s in enumerate(seqs)] == list(g[f"c{i}_prefill_cost"])
        assert m.decode_step_free_slots() == int(g[f"c{i}_decode_free"])
        assert [m.decode_step_reservation_cost(s) for s in seqs] == list(g[f"c{i}_decode_cost"])
        assert m.prompt_admission_failure_action() == "defer"
        # chain turns: row = [suffix, gen, needs_row, existing[L], reserved[L]] -> [required[L], rows, deficits[L], row deficit]
        rng = np.random.default_rng(0)
        for row in g[f"c{i}_chain"]:
            row = [int(x) for x in row]
            suffix, gen_t, need_row = row[0], row[1], bool(row[2])
            existing, reserved = tuple(row[3:3 + L]), tuple(row[3 + L:3 + 2 * L])
            want_req, want_rows = tuple(row[3 + 2 * L:3 + 3 * L]), row[3 + 3 * L]
            want_def, want_row_def = tuple(row[4 + 3 * L:4 + 4 * L]), row[4 + 4 * L]
            # outstanding_reserved_rows was drawn in {0, 1} by the generator and is not stored: the recorded row deficit
            # must be reproduced by one of the two
            got = [m.chain_capacity_deficits(suffix_tokens=suffix, generation_tokens=gen_t, existing_slots_by_layer=existing,
                                             outstanding_reserved_slots_by_layer=reserved, outstanding_reserved_rows=k,
                                             needs_resident_row=need_row) for k in (0, 1)]
            assert all(x[0] == want_req and x[1] == want_rows and x[2] == want_def for x in got)
            assert want_row_def in (got[0][3], got[1][3])
        del rng


def test_base_capacity_hooks_defaults():
    """base.py:1290-1393 defaults: one persistent slot per token, one shared slot budget."""
    from sparse_vllm_amd.engine.cache_manager.base import CacheManager

    class _Plain:                                       # the hooks only need `num_free_slots`
        num_free_slots = 37
    for name in ("reserved_prefill_slots", "prefill_step_free_slots", "prefill_step_free_slots_for", "prefill_step_reservation_cost",
                 "decode_step_free_slots", "decode_step_free_slots_for", "decode_step_reservation_cost", "prompt_admission_free_slots",
                 "prompt_admission_cost", "prompt_logical_reservation_cost", "prompt_admission_budgets", "prompt_admission_costs"):
        setattr(_Plain, name, getattr(CacheManager, name))
    m = _Plain()
    seqs = deque([SimpleNamespace(seq_id=0, num_prompt_tokens=30, num_prefilled_tokens=10, prefix_cache_hit_len=4),
                  SimpleNamespace(seq_id=1, num_prompt_tokens=9, num_prefilled_tokens=0, prefix_cache_hit_len=0),
                  SimpleNamespace(seq_id=2, num_prompt_tokens=9, num_prefilled_tokens=9, prefix_cache_hit_len=0)])
    assert m.reserved_prefill_slots(seqs, 8) == 20
    assert m.prompt_admission_budgets(seqs, 8) == {"slots": 17}
    assert m.prompt_admission_cost(seqs[0]) == 26 and m.prompt_admission_costs(seqs[1]) == {"slots": 9}
    assert m.prefill_step_free_slots() == 37 == m.decode_step_free_slots_for(seqs[0])
    assert m.prefill_step_reservation_cost(seqs[0], 5) == 5 and m.decode_step_reservation_cost(seqs[0]) == 1


def test_mi355x_decode_launch_geometry():
    """`Mi355xDecodeLaunchProvider` (registered in the reference's decode-attention-launch family,
    operators/decode_attention.py:13-158): the largest 16-aligned BLOCK_SEQ that keeps one workgroup per CU and - for 1- and
    2-KV-head tensor-parallel ranks, whose workgroups are one or two waves - one wave per SIMD; on any other device the
    reference's default provider answers and keeps the caller's value."""
    import torch
    from sparse_vllm_amd.operators.decode_attention import (DECODE_ATTENTION_LAUNCH_REGISTRY, DecodeAttentionLaunchSpec,
                                                            PreparedDecodeAttentionLaunchOp)
    from sparse_vllm_amd.operators.registry import OpResolver
    from sparse_vllm_amd.platforms.interface import DeviceCaps, PlatformEnum
    caps = DeviceCaps(platform=PlatformEnum.ROCM, device_type="cuda", device_index=0, device_name="AMD Instinct MI355X",
                      arch="gfx950", num_cus=256)

    def prepared(hq, hkv, caps_):
        spec = DecodeAttentionLaunchSpec(hq, hkv, 128, torch.bfloat16)
        return PreparedDecodeAttentionLaunchOp(spec, OpResolver(DECODE_ATTENTION_LAUNCH_REGISTRY).resolve(spec, caps_).provider)

    def cfg(hq, hkv, batch, length):
        op = prepared(hq, hkv, caps)
        assert op.name == "mi355x_hip_gqa" and op.accepts_batch_size
        return op.launch_config(block_seq=256, max_context_len=length, requires_attention_scores=True, batch_size=batch)[0]
    assert cfg(28, 4, 256, 4224) == 4224           # one block per sequence: stage 1 writes the output itself
    assert cfg(28, 4, 128, 4224) == 2112
    assert cfg(28, 4, 64, 4224) == 1056
    assert cfg(28, 4, 1, 4224) == 32               # floor: MIN_BLOCK_SEQ (one 32-token tile)
    assert cfg(32, 8, 256, 4224) == 4224           # 8 waves per workgroup: still one workgroup per CU
    assert cfg(14, 2, 256, 4224) == 2112           # two waves per workgroup -> 512 workgroups
    assert cfg(7, 1, 256, 4224) == 1056            # one wave per workgroup -> 1024 workgroups
    assert cfg(7, 1, 64, 4224) == 272
    assert cfg(28, 4, 4, 4672) == 96               # short blocks are whole 32-token tiles (Quest view: 73 -> 80 -> 96)
    assert cfg(28, 4, 8, 4672) == 160
    assert cfg(28, 4, 64, 1152) == 288
    for hkv in (1, 2, 4, 8):
        assert cfg(8 * hkv // hkv * hkv if hkv > 1 else 7, hkv, 16, 8192) % 16 == 0
    other = DeviceCaps(platform=PlatformEnum.ROCM, device_type="cuda", device_index=0, device_name="AMD Instinct MI300X",
                       arch="gfx942", num_cus=304)
    op = prepared(28, 4, other)
    assert op.name == "default_gqa" and not op.accepts_batch_size
    assert op.launch_config(block_seq=256, max_context_len=4224, requires_attention_scores=False) == (256, 16, 2)


def test_snapkv_streamingllm_quest_capacity_hooks_match_reference():
    """Scheduler capacity hooks of the SnapKV / StreamingLLM / QuEST managers (snapkv.py:761-905, streamingllm.py:24-32,
    quest.py:272-378 without a prefix cache, base.py:1242-1397) against answers recorded from the reference's own hooks
    on hand-built managers (tests/golden/gen_fixtures.py capacity_others)."""
    import json
    import os
    from sparse_vllm_amd.engine.cache_manager.quest import QuestCacheManager
    from sparse_vllm_amd.engine.cache_manager.snapkv import SnapKVCacheManager
    from sparse_vllm_amd.engine.cache_manager.streamingllm import StreamingLLMCacheManager
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "capacity_others.json")) as f:
        cases = json.load(f)
    assert {c["kind"] for c in cases} == {"snapkv", "streamingllm", "quest"}
    for c in cases:
        seqs = [SimpleNamespace(seq_id=a, num_prompt_tokens=b, num_prefilled_tokens=d, prefix_cache_hit_len=0) for a, b, d in c["seqs"]]
        if c["kind"] == "quest":
            m = object.__new__(QuestCacheManager)
            m.page_size = c["page"]
            m.row_seq_lens = np.asarray(c["lens"], dtype=np.int32)
            m.seq_id_to_row = {int(k): v for k, v in c["rows"].items()}
            m._num_free_pages = c["free_pages"]
        else:
            m = object.__new__({"snapkv": SnapKVCacheManager, "streamingllm": StreamingLLMCacheManager}[c["kind"]])
            m.num_layers = m.num_kv_layers = 2
            m.config = SimpleNamespace(vllm_sparse_method=c["kind"], num_sink_tokens=c["sink"], num_recent_tokens=c["recent"],
                                       decode_keep_tokens=c["keep"], snapkv_window_size=c["window"],
                                       snapkv_num_full_layers=c["full_layers"])
            m._num_free_slots = list(c["free"])
            m.row_seq_lens = [np.asarray(c["lens"], dtype=np.int32) for _ in range(2)]
            m.seq_id_to_row = [{r: r for r in range(len(c["lens"]))} for _ in range(2)]
        waiting = deque(seqs)
        for h in ("prompt_admission_cost", "prompt_logical_reservation_cost", "prefill_step_free_slots_for",
                  "decode_step_free_slots_for", "decode_step_reservation_cost", "remaining_prefill_tokens",
                  "min_final_prefill_chunk_size"):
            assert [int(getattr(m, h)(s)) for s in seqs] == c[h], (c["kind"], h)
        assert [int(m.prefill_step_reservation_cost(s, 5 + 7 * i)) for i, s in enumerate(seqs)] == c["prefill_step_reservation_cost"]
        assert int(m.prompt_admission_free_slots()) == c["prompt_admission_free_slots"]
        assert int(m.prefill_step_free_slots()) == c["prefill_step_free_slots"]
        assert int(m.decode_step_free_slots()) == c["decode_step_free_slots"]
        assert int(m.reserved_prefill_slots(waiting, 8)) == c["reserved_prefill_slots"]
        assert {k: int(v) for k, v in m.prompt_admission_budgets(waiting, 8).items()} == c["prompt_admission_budgets"]
        assert [{k: int(v) for k, v in m.prompt_admission_costs(s).items()} for s in seqs] == c["prompt_admission_costs"]
        assert int(m.prefill_batched_tokens_margin()) == c["prefill_batched_tokens_margin"]
        assert m.prompt_admission_failure_action() == c["prompt_admission_failure_action"]


def test_deltakv_first_prefill_staging_view_bookkeeping():
    """Host side of the DeltaKV prompt attention view (deltakv_base.py:1852-1993, :1020-1037): a first-prefill step - every
    row starts at length 0 - marks the sparse layers as staged with slots = the step's token order (-1 beyond a chunk);
    the flag lives for that step; a continuation chunk has no view and `build_prefill_compute_view` refuses it."""
    import pytest
    from sparse_vllm_amd.config import Config
    from sparse_vllm_amd.engine.cache_manager.base import CacheManager
    from sparse_vllm_amd.engine.sequence import Sequence
    conf = Config.from_kwargs(
        sparse_method="deltakv", num_hidden_layers=4, full_attention_layers="0,2", num_attention_heads=8,
        num_key_value_heads=2, head_dim=64, max_model_len=512, max_num_seqs_in_gpu=3, sink_keep_tokens=4,
        recent_keep_tokens=8, decode_keep_tokens=12, deltakv_neighbor_count=2, deltakv_latent_dim=32,
        deltakv_latent_quant_bits=4, deltakv_latent_quant_group_size=16, deltakv_center_ratio=0.25,
        allow_missing_deltakv_path=True, compressor_up_type="linear", full_layer_kv_quant_bits=4,
        full_layer_kivi_decode_block_seq=64, rope_theta=10000.0, engine_prefill_chunk_size=256, num_kvcache_slots=4096,
        device="cpu")
    cm = CacheManager.create(conf, None)
    assert not cm.prefill_attention_view_supported
    s1, s2 = Sequence(num_prompt_tokens=80), Sequence(num_prompt_tokens=17)
    s1.current_chunk_size, s2.current_chunk_size = 40, 17
    cu, total = cm._prepare_prefill([s1, s2])
    assert cu.tolist() == [0, 40, 57] and total == 57
    assert cm.prefill_attention_view_supported
    assert [cm.has_prefill_staging_view(l) for l in range(4)] == [False, True, False, True]      # sparse layers only
    slots, req, ctx, temp = cm.get_prefill_staging_view(1)
    assert temp is None and req.tolist() == [0, 1] and ctx.tolist() == [40, 17] and tuple(slots.shape) == (2, 40)
    assert slots[0].tolist() == list(range(40)) and slots[1, :17].tolist() == list(range(40, 57))
    assert (slots[1, 17:] == -1).all()
    with pytest.raises(NotImplementedError):
        cm.get_prefill_staging_view(0)
    cm.on_forward_end([s1, s2], True)
    assert not cm.prefill_attention_view_supported and not cm.has_prefill_staging_view(1)
    s1.num_prefilled_tokens, s1.current_chunk_size = 40, 40
    cm._prepare_prefill([s1])                            # continuation chunk: the row holds 40 tokens
    assert not cm.prefill_attention_view_supported
    with pytest.raises(NotImplementedError, match="reconstructed prefill compute view"):
        cm.build_prefill_compute_view(1, None, None, None)


def test_capture_without_gc_restores_the_collector():
    """`capture_without_gc` (engine/decode_driver.py): the cycle collector is off inside the block - a captured hipGraph
    reachable only from dead cycles must not be destroyed inside another capture - and back afterwards, also on errors."""
    import gc
    import pytest
    from sparse_vllm_amd.engine.decode_driver import capture_without_gc
    assert gc.isenabled()
    with capture_without_gc():
        assert not gc.isenabled()
    assert gc.isenabled()
    with pytest.raises(RuntimeError):
        with capture_without_gc():
            raise RuntimeError("capture failed")
    assert gc.isenabled()
    gc.disable()
    try:
        with capture_without_gc():
            assert not gc.isenabled()
        assert not gc.isenabled()              # a caller that runs with the collector off keeps it off
    finally:
        gc.enable()
