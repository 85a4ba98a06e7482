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


def test_planner_over_a_quest_manager_reproduces_the_reference_plan():
    """tests/golden/planned_run.json `quest` (the reference's Scheduler over the reference's QuestCacheManager,
    tests/planned_run_scenarios.py) against `StepPlanner` over this build's QuestCacheManager, host side only (paging is host
    arithmetic; the GPU twin, tests/test_gpu_planned_run.py, also executes every step): per step the same sequences and
    chunk sizes, queues, deferred prompts, finished rows, free pages and - bit for bit - the same page tables."""
    import numpy as np
    import planned_run_scenarios as prs
    from sparse_vllm_amd.config import Config
    from sparse_vllm_amd.engine.cache_manager.base import CacheManager
    from sparse_vllm_amd.engine.sequence import Sequence
    from sparse_vllm_amd.engine.step_planner import StepPlanner
    with open(os.path.join(HERE, "golden", "planned_run.json")) as f:
        want = json.load(f)["quest"]
    sc = prs.QUEST
    conf = Config.from_kwargs(sparse_method="quest", num_hidden_layers=sc["layers"], max_model_len=sc["max_model_len"],
                              max_num_seqs_in_gpu=sc["rows"], num_kvcache_slots=sc["pages"] * sc["page"],
                              sink_keep_tokens=sc["sink"], recent_keep_tokens=sc["recent"], decode_keep_tokens=sc["keep"],
                              quest_skip_layers=sc["skip_layers"], num_attention_heads=4, num_key_value_heads=2, head_dim=4,
                              engine_prefill_chunk_size=sc["planner"]["chunk_prefill_size"], device="cpu")
    cm = CacheManager.create(conf)
    cm.free_pages_cpu_stack = np.asarray(want["free_pages_stack"], dtype=np.int32)
    cfg = SimpleNamespace(num_sink_tokens=sc["sink"], num_recent_tokens=sc["recent"], decode_keep_tokens=sc["keep"],
                          vllm_sparse_method="quest", **sc["planner"])
    p = StepPlanner(cfg, cm)
    seqs = [Sequence(num_prompt_tokens=n, max_tokens=g) for n, g in zip(sc["prompts"], sc["gens"])]
    base = seqs[0].seq_id
    for s in seqs:
        p.add(s)
    step, page = 0, sc["page"]
    while not p.is_finished():
        chosen, is_prefill, _ = p.schedule()
        w = want["trace"][step]
        assert is_prefill == w["prefill"]
        assert [[s.seq_id - base, int(s.current_chunk_size) if is_prefill else 1] for s in chosen] == w["seqs"], step
        if is_prefill:
            cm._prepare_prefill(chosen)
        else:
            for s in chosen:
                cm._allocate(s.seq_id, 1)
        finished = p.postprocess(chosen, [0] * len(chosen), is_prefill)
        tables = {str(s.seq_id - base): [int(x) for x in cm.buffer_req_to_page_slots_cpu[
            cm.seq_id_to_row[s.seq_id], : (int(cm.row_seq_lens[cm.seq_id_to_row[s.seq_id]]) + page - 1) // page]]
                  for s in seqs if s.seq_id in cm.seq_id_to_row}
        assert tables == w["page_tables"], step
        assert int(cm._num_free_pages) == w["free_pages"] and int(cm.num_free_slots) == w["free_slots"]
        assert sorted(s.seq_id - base for s in finished) == w["finished"]
        for s in finished:
            cm.free_seq(s.seq_id)
        assert [s.seq_id - base for s in p.waiting] == w["waiting"] and [s.seq_id - base for s in p.decoding] == w["decoding"]
        assert sorted(x - base for x in p._defer_noted) == w["deferred"], step
        step += 1
    assert step == len(want["trace"]) and any(r["deferred"] for r in want["trace"])
    assert cm._num_free_pages == cm.num_pages


def test_postprocess_moves_progress_tokens_and_finished_rows():
    """`StepPlanner.postprocess` = the queue effects of `Scheduler.postprocess` (scheduler.py:794-870) for this build's
    sequences: a prompt that finishes its last chunk takes its first token and joins `decoding`; an unfinished one returns to
    the head of `waiting`; a row whose generation budget is used up leaves `decoding` and is returned for release - also when
    the budget is one token (finished by the prefill step itself)."""
    from sparse_vllm_amd.engine.sequence import Sequence
    from sparse_vllm_amd.engine.step_planner import StepPlanner
    oracle = ps.ScriptedOracle()
    p = StepPlanner(ps._cfg(None, chunk=4, max_tokens=16), oracle)
    a, b, c = Sequence(num_prompt_tokens=6, max_tokens=2), Sequence(num_prompt_tokens=3, max_tokens=1), Sequence(num_prompt_tokens=4, max_tokens=3)
    for s in (a, b, c):
        p.add(s)
    chosen, is_prefill, _ = p.schedule()
    assert is_prefill and [(s.seq_id, s.current_chunk_size) for s in chosen] == [(a.seq_id, 4), (b.seq_id, 3), (c.seq_id, 4)]
    finished = p.postprocess(chosen, [7, 8, 9], True)
    assert finished == [b] and b.num_completion_tokens == 1 and b.last_token == 8 and b.num_tokens == 4     # budget of one token
    assert list(p.waiting) == [a] and a.num_prefilled_tokens == 4 and a.num_completion_tokens == 0
    assert list(p.decoding) == [c] and c.num_completion_tokens == 1 and c.num_tokens == 5
    assert oracle.completed == [b.seq_id, c.seq_id]
    chosen, is_prefill, _ = p.schedule()                       # the rest of `a`
    assert is_prefill and chosen == [a] and a.current_chunk_size == 2
    assert p.postprocess(chosen, [1], True) == [] and list(p.decoding) == [c, a] and not p.waiting
    # decode: short rows first (a holds 7 tokens > sink + keep + recent = 6, c holds 5), so c decodes alone until it is done
    chosen, is_prefill, _ = p.schedule()
    assert not is_prefill and chosen == [c]
    assert p.postprocess(chosen, [0], False) == [] and c.num_completion_tokens == 2
    chosen, _, _ = p.schedule()
    assert chosen == [c] and p.postprocess(chosen, [0], False) == [c] and list(p.decoding) == [a]      # 3 of 3
    chosen, is_prefill, _ = p.schedule()
    assert not is_prefill and chosen == [a]
    assert p.postprocess(chosen, [0], False) == [a] and p.is_finished()                                 # 2 of 2
