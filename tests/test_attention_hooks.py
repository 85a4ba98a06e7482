"""CPU: both branches of this build's `Attention.forward` against the REFERENCE's (layers/attention.py:88-161 prefill,
:162-250 decode) and the backend's decode slot check (layers/attention_backend.py:397-439).

tests/golden/attention_hooks.json was produced by tests/golden/gen_fixtures.py (`attention_hooks`) running the
reference's `Attention.forward` over the recording stand-ins of tests/hook_trace.py; the same stand-ins driven through
`sparse_vllm_amd.layers.attention.Attention` must leave the identical trace: hook order, argument shapes and values,
results, exception classes and texts ("call order per layer is fixed by Attention.forward").
No kernel runs here: the reference's fake-attention switches (SPARSEVLLM_FAKE_ATTENTION + ..._ALLOW_...) stand in for the
launch, and in the H2O flow the scoring launch is replaced the way the reference's own unit tests replace it.
"""

import json
import os
from types import SimpleNamespace

import numpy as np
import pytest

import hook_trace as ht

torch = pytest.importorskip("torch")

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def fixture():
    with open(os.path.join(HERE, "golden", "attention_hooks.json")) as f:
        return json.load(f)


def _types():
    from sparse_vllm_amd.engine.cache_manager import base as b
    return SimpleNamespace(SparseSelection=b.SparseSelection, AttentionViewMeta=b.AttentionViewMeta,
                           ExplicitKVPayload=b.ExplicitKVPayload, PrefillComputeView=b.PrefillComputeView,
                           DecodeComputeView=b.DecodeComputeView)


def _install(is_prefill, cu, cm, sc, layer, seqs=None):
    from sparse_vllm_amd.utils.context import set_context
    ctx = set_context(is_prefill, cu_seqlens_q=cu, cache_manager=cm, sparse_controller=sc)
    ctx.now_layer_idx = layer
    ctx.seqs = seqs
    return ctx


@pytest.mark.parametrize("case", ht.CASES, ids=[c["name"] for c in ht.CASES])
def test_prefill_hook_trace_equals_reference(fixture, case):
    from sparse_vllm_amd.layers.attention import Attention
    got = ht.run_case(case, attention_cls=Attention, types=_types(), install_context=_install)
    ref = fixture["cases"][case["name"]]
    assert json.loads(json.dumps(got["trace"])) == ref["trace"]
    assert json.loads(json.dumps(got["result"])) == ref["result"]


def test_every_reference_case_is_replayed(fixture):
    assert sorted(fixture["cases"]) == sorted(c["name"] for c in ht.CASES)
    assert sorted(fixture["decode_cases"]) == sorted(c["name"] for c in ht.DECODE_CASES)
    assert sorted(fixture["bounds_cases"]) == sorted(c["name"] for c in ht.BOUNDS_CASES)


@pytest.mark.parametrize("case", ht.DECODE_CASES, ids=[c["name"] for c in ht.DECODE_CASES])
def test_decode_hook_trace_equals_reference(fixture, case):
    """The DECODE branch (layers/attention.py:162-250): get_decode_selection -> build_decode_compute_view (TypeError for a
    payload that is not explicit KV) -> static capacity / slot-table clamp / SVLLM_DEBUG_DECODE_BOUNDS -> get_decode_block_seq
    -> the launch provider asked with exactly the reference's three keywords -> run_decode with exactly the reference's
    keywords and workspace shapes -> record_decode_query -> the two on_layer_attention_end hooks; temp slots released in
    `finally` on every path."""
    from sparse_vllm_amd.layers.attention import Attention
    got = ht.run_decode_case(case, attention_cls=Attention, types=_types(), install_context=_install)
    ref = fixture["decode_cases"][case["name"]]
    assert json.loads(json.dumps(got["trace"])) == ref["trace"]
    assert json.loads(json.dumps(got["result"])) == ref["result"]


@pytest.mark.parametrize("case", ht.BOUNDS_CASES, ids=[c["name"] for c in ht.BOUNDS_CASES])
def test_decode_bounds_checker_equals_reference(fixture, case):
    """`_debug_check_decode_bounds` (layers/attention_backend.py:397-439): same verdicts, same messages."""
    from sparse_vllm_amd.layers.attention import HipAttentionBackend
    got = ht.run_bounds_case(case, backend=HipAttentionBackend(), types=_types())
    assert json.loads(json.dumps(got)) == fixture["bounds_cases"][case["name"]]


def test_h2o_prefill_flow_through_attention_layer_equals_reference(fixture):
    """A real H2OCacheManager (host state on CPU tensors) + SparseController driven chunk by chunk through
    `_prepare_prefill -> prepare_forward -> Attention.forward`: what `_run_prefill_score` is asked for (window in
    compressed physical coordinates, candidate_start 0, num_recent 0, the view's slot table / lengths / rows) and the
    cumulative score rows after two chunks equal the reference's (h2o.py:750-894)."""
    from sparse_vllm_amd.config import Config
    from sparse_vllm_amd.engine.cache_manager.base import CacheManager
    from sparse_vllm_amd.engine.sequence import Sequence
    from sparse_vllm_amd.engine.sparse_controller import SparseController
    from sparse_vllm_amd.layers.attention import Attention
    F = ht.H2O_FLOW
    L, B = F["layers"], len(F["chunks"][0])
    conf = Config.from_kwargs(sparse_method="h2o", num_hidden_layers=L, max_model_len=96, max_num_seqs_in_gpu=B,
                              num_kvcache_slots=256, h2o_decode_budget=48, h2o_decode_eviction_interval=16,
                              h2o_prefill_budget=64, h2o_prefill_score_window=F["window"], num_attention_heads=F["heads"],
                              num_key_value_heads=F["kv_heads"], head_dim=F["dim"], device="cpu")
    cm = CacheManager.create(conf)
    sc = SparseController(conf, cm)
    calls = []
    cm._run_prefill_score = ht.fake_prefill_score_fn(calls)
    attn = Attention(F["heads"], F["dim"], F["dim"] ** -0.5, F["kv_heads"])
    seqs = [Sequence(num_prompt_tokens=sum(ch[i] for ch in F["chunks"])) for i in range(B)]
    for i, s in enumerate(seqs):
        s.seq_id = i
    saved = {k: os.environ.get(k) for k in ("SPARSEVLLM_FAKE_ATTENTION", "SPARSEVLLM_ALLOW_FAKE_ATTENTION")}
    os.environ["SPARSEVLLM_FAKE_ATTENTION"] = "1"
    os.environ["SPARSEVLLM_ALLOW_FAKE_ATTENTION"] = "1"
    try:
        for chunk in F["chunks"]:
            for s, n in zip(seqs, chunk):
                s.current_chunk_size = n
            cu, total = cm._prepare_prefill(seqs)
            ctx = _install(True, cu, cm, sc, 0, seqs=seqs)
            sc.prepare_forward(seqs, True)
            q = torch.zeros((total, F["heads"], F["dim"]), dtype=torch.bfloat16)
            kv = torch.zeros((total, F["kv_heads"], F["dim"]), dtype=torch.bfloat16)
            for l in range(L):
                ctx.now_layer_idx = l
                o = attn(q, kv, kv)
                assert o.shape == q.shape and (o == 0).all()
            for s, n in zip(seqs, chunk):
                s.num_prefilled_tokens += n
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    ref = fixture["h2o_flow"]
    assert len(calls) == len(ref["calls"]) == L * len(F["chunks"])
    for got, want in zip(calls, ref["calls"]):
        got = dict(got)
        want = dict(want)
        # product pool sizes differ from the hand-built reference manager only in the slot-table row count
        for k in ("k_cache", "active_slots"):
            got.pop(k), want.pop(k)
        assert got == want
    for l in range(L):
        for i in range(B):
            np.testing.assert_array_equal(cm.h2o_score(l, i).numpy(), np.asarray(ref["scores"][f"{l}_{i}"], np.float32))


def test_collect_signature_is_the_reference_one():
    """(layer_idx, q, view, *, b_start_loc, chunk_lens) on the base class and on every manager that overrides it."""
    import inspect
    from sparse_vllm_amd.engine.cache_manager.base import CacheManager
    from sparse_vllm_amd.engine.cache_manager.h2o import H2OCacheManager
    from sparse_vllm_amd.engine.cache_manager.snapkv import SnapKVCacheManager
    for cls in (CacheManager, SnapKVCacheManager, H2OCacheManager):
        for name in ("collect_prefill_attention_score", "record_prefill_query"):
            sig = inspect.signature(getattr(cls, name))
            params = list(sig.parameters.values())
            assert [p.name for p in params] == ["self", "layer_idx", "q", "view", "b_start_loc", "chunk_lens"]
            assert [p.kind for p in params[-2:]] == [inspect.Parameter.KEYWORD_ONLY] * 2
    sig = inspect.signature(CacheManager.build_prefill_compute_view)
    assert list(sig.parameters) == ["self", "layer_idx", "k_current", "v_current", "selection"]
    sig = inspect.signature(CacheManager.before_prefill_layer_attention)
    assert list(sig.parameters) == ["self", "layer_idx", "selection"]
