"""CPU: the method surface (`sparse_vllm_amd.method_registry`, `config.normalize_runtime_params`) against a table
generated from the reference's own functions (tests/golden/gen_fixtures.py `method_surface`; method_registry.py:19-348,
configs/runtime_params.py:15-198): every alias x policy, the prefill-score contract, graph / prefix-cache support,
model-runtime compatibility for the dense model families, public kwargs aliases and legacy-name rejection - return
values, exception classes and messages."""

import json
import os

import pytest

from sparse_vllm_amd import method_registry as mr
from sparse_vllm_amd.config import normalize_runtime_params

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def table():
    with open(os.path.join(HERE, "golden", "method_surface.json")) as f:
        return json.load(f)


def _call(fn, *a, **k):
    try:
        return {"ok": fn(*a, **k)}
    except Exception as e:
        return {"err": type(e).__name__, "msg": str(e)}


def _name(n):
    return n["int"] if isinstance(n, dict) else n


def test_constant_tables(table):
    t = table["tables"]
    assert {("<None>" if k is None else k): v for k, v in mr.METHOD_ALIASES.items()} == t["METHOD_ALIASES"]
    for key in ("CANONICAL_SPARSE_METHODS", "SUPPORTED_SPARSE_METHODS", "SUPPORTED_SPARSE_METHOD_ALIASES",
                "PREFIX_CACHE_SUPPORTED_METHODS", "DECODE_CUDA_GRAPH_SUPPORTED_METHODS",
                "TP_DECODE_CUDA_GRAPH_SUPPORTED_METHODS", "SUPPORTED_PREFILL_POLICIES"):
        assert sorted(getattr(mr, key)) == t[key], key
    assert dict(mr.PREFILL_POLICY_BY_METHOD) == t["PREFILL_POLICY_BY_METHOD"]


def test_names_policies_contract_and_support_flags(table):
    policies = table["policies"]
    for n, row in zip(table["names"], table["rows"]):
        n = _name(n)
        assert _call(mr.normalize_sparse_method, n) == row["normalize"], n
        assert _call(mr.get_default_prefill_schedule_policy, n) == row["default_policy"], n
        for p, ref in zip(policies, row["resolve"]):
            assert _call(mr.resolve_prefill_schedule_policy, n, p) == ref, (n, p)
        assert _call(mr.is_deltakv_method, n) == row["is_deltakv"]
        assert _call(mr.is_decode_cuda_graph_supported, n) == row["graph"]
        assert _call(mr.is_tp_decode_cuda_graph_supported, n) == row["tp_graph"]
        c = _call(mr.sparse_prefill_attention_contract, n)
        if "ok" in c:
            c = {"ok": [c["ok"].main_score_kind.name, c["ok"].score_collection.name]}
        assert c == row["contract"], n


def test_model_runtime_compatibility_dense_models(table):
    """Dense model families (qwen2, llama ...) x every name x graph x prefix-cache flags.  The reference's MoE / Gemma
    rows are other model families (out of scope): here they are unknown model types and raise the
    same NotImplementedError an unknown type raises in the reference."""
    tabs = table["compat_tables"]
    for n, row in zip(table["names"], table["rows"]):
        n = _name(n)
        for model_type, graph, prefix, ref in row["compat"]:
            got = _call(mr.validate_model_runtime_compatibility, model_type=model_type, sparse_method=n,
                        topology=mr.ParallelTopology(1, 1, 1), decode_cuda_graph=graph, enable_prefix_caching=prefix)
            if model_type == "qwen3_moe":
                assert got["err"] == "NotImplementedError" and "qwen3_moe" in got["msg"]
                continue
            if "ok" in got:
                lists = [sorted(got["ok"].sparse_methods), sorted(got["ok"].prefix_cache_methods),
                         sorted(got["ok"].decode_cuda_graph_methods)]
                assert "ok" in ref and tabs[ref["ok"]] == lists, (n, model_type, graph, prefix)
            else:
                assert got == ref, (n, model_type, graph, prefix)


def test_runtime_params_aliases_and_legacy_rejection(table):
    for kw, ref in table["runtime_params"]:
        got = _call(normalize_runtime_params, dict(kw))
        if "ok" in ref:
            exp = dict(ref["ok"])
            if "vllm_sparse_method" in exp:
                # the reference maps the remaining aliases when Config normalises the method; this mirror does it here
                exp["vllm_sparse_method"] = mr.normalize_sparse_method(exp["vllm_sparse_method"])
            assert got == {"ok": exp}, kw
        else:
            assert got == ref, kw
