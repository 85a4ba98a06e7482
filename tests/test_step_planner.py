"""CPU: `sparse_vllm_amd.engine.step_planner.StepPlanner` against the REFERENCE's `Scheduler.schedule`
(engine/scheduler.py:398-792) on the scenarios of tests/planner_scenarios.py - tests/golden/step_planner.json holds what the
reference scheduler did (tests/golden/gen_fixtures.py `step_planner`): per step which sequences run, with which chunk
sizes, the queue orders afterwards, and the exception class + text where the reference fails fast.  Where the reference
preempts, the planner raises `PreemptionRequired` for the same victim and leaves its queues untouched (SURVEY 8(f).4,
bounded: no preemption)."""

import json
import os
from types import SimpleNamespace

import pytest

import planner_scenarios as ps

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def ref():
    with open(os.path.join(HERE, "golden", "step_planner.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def got():
    from sparse_vllm_amd.engine.step_planner import StepPlanner
    return json.loads(json.dumps(ps.run_all(SimpleNamespace(make=lambda cfg, oracle: StepPlanner(cfg, oracle)))))


def _names():
    with open(os.path.join(HERE, "golden", "step_planner.json")) as f:
        return sorted(json.load(f))


@pytest.mark.parametrize("name", _names())
def test_plan_equals_reference_schedule(ref, got, name):
    want, have = ref[name], got[name]
    # the texts name the oracle's class; the bug-check message for a finished prompt in `waiting` is not compared (Chinese
    # text in the reference, never reached by a consistent caller)
    assert have["trace"] == want["trace"]
    assert have["admitted"] == want["admitted"] and have["completed"] == want["completed"]


def test_every_scenario_is_pinned(ref, got):
    assert sorted(ref) == sorted(got) and len(ref) >= 35
    kinds = {"err": 0, "preempt": 0, "prefill": 0, "decode": 0}
    for sc in ref.values():
        for step in sc["trace"]:
            if "err" in step:
                kinds["err"] += 1
            elif "preempt" in step:
                kinds["preempt"] += 1
            elif step["prefill"]:
                kinds["prefill"] += 1
            else:
                kinds["decode"] += 1
    assert kinds["err"] >= 6 and kinds["preempt"] >= 2 and kinds["prefill"] >= 40 and kinds["decode"] >= 15, kinds


def test_preemption_request_leaves_queues_untouched():
    from sparse_vllm_amd.engine.step_planner import PreemptionRequired, StepPlanner
    oracle = ps.ScriptedOracle(decode_free=4, per_seq_decode_free={0: 0, 1: 0})
    p = StepPlanner(ps._cfg(None), oracle)
    rows = [ps.PlanSeq(i, 3, prefilled=3, completion=1) for i in range(2)]
    p.decoding.extend(rows)
    with pytest.raises(PreemptionRequired) as e:
        p.schedule()
    assert e.value.victim is rows[0] and list(p.decoding) == rows and not p.waiting


def test_planner_over_a_real_cache_manager():
    """The hooks of this build's managers are the oracle: an H2O manager (CPU host state) plans chunked prefill of two
    prompts against its prefill budget arithmetic and then decodes them."""
    from sparse_vllm_amd.config import Config
    from sparse_vllm_amd.engine.cache_manager.base import CacheManager
    from sparse_vllm_amd.engine.sequence import Sequence
    from sparse_vllm_amd.engine.step_planner import StepPlanner
    conf = Config.from_kwargs(sparse_method="h2o", num_hidden_layers=2, max_model_len=256, max_num_seqs_in_gpu=4,
                              num_kvcache_slots=512, h2o_decode_budget=48, h2o_decode_eviction_interval=16,
                              h2o_prefill_budget=64, num_attention_heads=4, num_key_value_heads=2, head_dim=4, device="cpu")
    cm = CacheManager.create(conf)
    cfg = SimpleNamespace(max_num_seqs_in_batch=4, max_num_batched_tokens=64, max_decoding_seqs=8, chunk_prefill_size=32,
                          num_sink_tokens=conf.num_sink_tokens, num_recent_tokens=conf.num_recent_tokens,
                          decode_keep_tokens=conf.decode_keep_tokens, vllm_sparse_method="h2o")
    p = StepPlanner(cfg, cm)
    seqs = [Sequence(num_prompt_tokens=70), Sequence(num_prompt_tokens=20)]
    for s in seqs:
        s.num_completion_tokens = 0
        s.prefix_cache_hit_len = 0
        p.add(s)
    plan = []
    for _ in range(6):
        chosen, is_prefill, _ = p.schedule()
        plan.append((is_prefill, [(s.seq_id - seqs[0].seq_id, s.current_chunk_size if is_prefill else 1) for s in chosen]))
        if is_prefill:
            cm._prepare_prefill(chosen)                        # the chunk's slots are taken, as the engine's step would
            p.after_prefill(chosen)
        else:
            break
    assert plan[0] == (True, [(0, 32), (1, 20)])            # 64-token step: a 32-token chunk + the whole short prompt
    assert plan[1] == (True, [(0, 32)]) and plan[2] == (True, [(0, 6)])
    assert plan[3][0] is False and sorted(i for i, _ in plan[3][1]) == [0, 1]
