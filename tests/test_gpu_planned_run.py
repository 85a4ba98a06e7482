"""SURVEY 8(f).4 with a consumer: waiting prompts -> `StepPlanner` -> planned prefill chunks -> decode, executed on the GPU
by `SparseDecodeDriver.run` (engine/decode_driver.py) under a KV pool small enough that admission defers, for H2O, Quest,
SnapKV and StreamingLLM (whose slot tables and free stacks, having no scores in them, must equal the REFERENCE run's bit for bit).
Two independent checks of every step:

* the PLAN (which sequences ran, chunk sizes, queue orders, deferred prompts, finished rows, every row's physical length,
  free capacity, H2O eviction counters, Quest page tables) equals tests/golden/planned_run.json - what the REFERENCE's
  `Scheduler` (engine/scheduler.py:398-870) did over the reference's own cache managers on the same prompt set
  (tests/golden/gen_fixtures.py `planned_run`, tests/planned_run_scenarios.py);
* the STATE AND OUTPUTS of the executed step equal the numpy oracle chained step by step on a mirror of the device state:
  slot tables / free stacks / lengths bit-exact, H2O cumulative scores and attention outputs within the path's tolerances
  (2e-2 outputs, the reference's bar for decode partials; scores rtol 2e-2 / atol 2e-3 in prefill, 1e-4 / 1e-6 in decode).

Decode steps run through `prepare_decode_static` and hipGraph replay whenever the decode batch repeats.
"""

import json
import os
from types import SimpleNamespace

import numpy as np
import pytest

import planned_run_scenarios as prs
from oracle import bf16_round
from oracle import decode_attention as oda
from oracle import h2o as oh
from oracle import prefill_attention as opa
from oracle import prefill_score as ops
from oracle import quest as oq

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

HERE = os.path.dirname(os.path.abspath(__file__))
Hq, Hkv, D = 28, 4, 128
TOL = 2e-2


def _f(t):
    return t.float().cpu().numpy()


@pytest.fixture(scope="module")
def ref_plan():
    with open(os.path.join(HERE, "golden", "planned_run.json")) as f:
        return json.load(f)


class StepInputs:
    """Stand-in for the layers around the attention path: seeded random q / k / v per step.  Decode steps reuse one set of
    buffers per batch size (stable addresses: the driver captures the step as a hipGraph once a batch repeats)."""

    def __init__(self, layers, device):
        self.L, self.device, self.g, self.decode_bufs = layers, device, torch.Generator().manual_seed(77), {}

    def _draw(self, n, heads):
        return (torch.randn(self.L, n, heads, D, generator=self.g) * 0.4).to(torch.bfloat16)

    def __call__(self, is_prefill, seqs, n):
        q, k, v = self._draw(n, Hq), self._draw(n, Hkv), self._draw(n, Hkv)
        if is_prefill:
            q, k, v = (t.to(self.device) for t in (q, k, v))
            return q, k, v, torch.zeros_like(q)
        bufs = self.decode_bufs.get(n)
        if bufs is None:
            bufs = self.decode_bufs[n] = tuple(torch.zeros_like(t, device=self.device) for t in (q, k, v, q))
        for dst, src in zip(bufs[:3], (q, k, v)):
            dst.copy_(src)
        return bufs


def _planner(sc, cm):
    from sparse_vllm_amd.engine.step_planner import StepPlanner
    cfg = SimpleNamespace(num_sink_tokens=sc["sink"], num_recent_tokens=sc["recent"], decode_keep_tokens=sc["keep"],
                          vllm_sparse_method=sc["method"], **sc["planner"])
    return StepPlanner(cfg, cm)


def _prompts(sc):
    from sparse_vllm_amd.engine.sequence import Sequence
    return [Sequence(num_prompt_tokens=n, max_tokens=g) for n, g in zip(sc["prompts"], sc["gens"])]


def _check_plan_record(step, rec, want, planner, base, lens, extra):
    """One executed step against the reference scheduler's record of the same step."""
    where = f"step {step}"
    assert rec["prefill"] == want["prefill"], where
    assert [[s.seq_id - base, c] for s, c in zip(rec["seqs"], rec["chunks"])] == want["seqs"], where
    assert sorted(s.seq_id - base for s in rec["finished"]) == sorted(want["finished"]), where
    assert [s.seq_id - base for s in planner.waiting] == want["waiting"], where
    assert [s.seq_id - base for s in planner.decoding] == want["decoding"], where
    assert sorted(x - base for x in planner._defer_noted) == want["deferred"], where
    assert {str(k): int(v) for k, v in lens.items()} == want["lens"], where
    for key, value in extra.items():
        assert value == want[key], f"{where}: {key} {value} != {want[key]}"


# ------------------------------------------------------------------------------------------------------------ H2O
def test_h2o_planned_run_matches_reference_plan_and_oracle(ref_plan):
    from sparse_vllm_amd.config import Config
    from sparse_vllm_amd.engine.decode_driver import SparseDecodeDriver
    sc = prs.H2O
    L, budget, interval, pre_budget, window = sc["layers"], sc["budget"], sc["interval"], sc["prefill_budget"], sc["window"]
    conf = Config.from_kwargs(sparse_method="h2o", num_hidden_layers=L, max_model_len=sc["max_model_len"],
                              max_num_seqs_in_gpu=sc["rows"], num_kvcache_slots=sc["slots"], h2o_decode_budget=budget,
                              h2o_decode_eviction_interval=interval, h2o_prefill_budget=pre_budget,
                              h2o_prefill_score_window=window, h2o_recent_ratio=sc["recent_ratio"],
                              engine_prefill_chunk_size=sc["planner"]["chunk_prefill_size"],
                              sink_keep_tokens=sc["sink"], recent_keep_tokens=sc["recent"], decode_keep_tokens=sc["keep"])
    drv = SparseDecodeDriver(conf)
    cm = drv.cache_manager
    cm.permute_free_slots(2)
    drv.enable_decode_graph()
    planner = _planner(sc, cm)
    seqs = _prompts(sc)
    base = seqs[0].seq_id
    want = ref_plan["h2o"]["trace"]

    st = oh.SlotState(cm.buffer_req_to_token_slots_tensor.cpu().numpy().copy(), cm.free_slots_stack_tensor.cpu().numpy().copy(),
                      np.asarray(cm._num_free_slots, dtype=np.int64), np.stack(cm.row_seq_lens).astype(np.int32))
    kc, vc = _f(cm.kv_cache[0]).copy(), _f(cm.kv_cache[1]).copy()
    active_decode: set[int] = set()
    seen = dict(steps=0, graph_replays=0)

    def row_of(s):
        return cm.seq_id_to_row[0][s.seq_id]

    def oracle_prefill(rec):
        active, chunk_lens = rec["seqs"], rec["chunks"]
        arows = [row_of(s) for s in active]
        qn, kn, vn, got_o = _f(rec["q"]), _f(rec["k"]), _f(rec["v"]), _f(rec["outputs"])
        starts = np.concatenate(([0], np.cumsum(chunk_lens)[:-1])).astype(np.int32)
        # `postprocess` has already moved the progress: the chunk was final iff the prompt is complete now
        finals = [s.num_prefilled_tokens >= s.num_prompt_tokens for s in active]
        for l in range(L):
            ctx, cache = [], []
            for b, (r, n) in enumerate(zip(arows, chunk_lens)):
                prev = int(st.row_len[l, r])
                new = oh.allocate(st, l, r, n)
                sl = slice(int(starts[b]), int(starts[b]) + n)
                kc[l][new], vc[l][new] = kn[l][sl], vn[l][sl]
                ctx.append(prev + n)
                cache.append(prev)
            ctx, cache = np.array(ctx, np.int32), np.array(cache, np.int32)
            ref_o = opa.context_attention_fwd(qn[l], kc[l], vc[l], np.array(arows, np.int32), starts, ctx, cache, st.slot_table[l])
            np.testing.assert_allclose(got_o[l], bf16_round(ref_o), rtol=TOL, atol=TOL, err_msg=f"prefill layer {l} step {seen['steps']}")
            qs = np.maximum(cache, ctx - window).astype(np.int32)
            stepsc = np.empty((len(active), int(ctx.max())), np.float32)
            ops.prefill_score_fwd(qn[l], kc[l], stepsc, np.array(arows, np.int32), starts, ctx, cache, int((ctx - qs).max()),
                                  st.slot_table[l], qs, ctx, score_mode="probability")
            for b, r in enumerate(arows):
                st.scores[(l, r)] = oh.accumulate_score(st.scores.get((l, r)), stepsc[b], new_len=int(ctx[b]),
                                                        weight=float(ctx[b] - qs[b]))
        for l in range(L):
            for r, fin in zip(arows, finals):
                n = int(st.row_len[l, r])
                cap = budget if fin else pre_budget
                if n <= cap:
                    continue
                keep = oh.select_h2o_indices(st.scores[(l, r)], budget=cap, recent_ratio=sc["recent_ratio"])
                kept = st.scores[(l, r)][keep]
                if fin:
                    oh.compact_final_prefill_dense_batch(st, l, [r], keep[None, :], cap, kc[l], vc[l])
                else:
                    oh.free_part_slots(st, l, r, keep, keep_sorted=True)
                st.scores[(l, r)] = kept
        return 2e-2, 2e-3

    def oracle_decode(rec):
        rows = [row_of(s) for s in rec["seqs"]]
        B = len(rows)
        active_decode.update(rows)
        qn, kn, vn, got_o = _f(rec["q"]), _f(rec["k"]), _f(rec["v"]), _f(rec["outputs"])
        new_slots = oh.decode_allocate_batch_layers(st, range(L), rows)
        lens = np.array([st.row_len[0, r] for r in rows], dtype=np.int32)
        for l in range(L):
            kc[l][new_slots[l]], vc[l][new_slots[l]] = kn[l], vn[l]
            W = int(lens.max())
            raw = np.full((B, W), -1e20, dtype=np.float32)
            mid, lse = oda.flash_decode_stage1(qn[l], kc[l], vc[l], st.slot_table[l], np.array(rows, np.int32), lens, W, 64,
                                               attn_score=raw)
            o = oda.flash_decode_stage2(mid, lse, lens, 64)
            np.testing.assert_allclose(got_o[l], bf16_round(o), rtol=TOL, atol=TOL, err_msg=f"decode layer {l} step {seen['steps']}")
            norm = oda.h2o_normalize_decode_scores(raw, D)
            for b, r in enumerate(rows):
                st.scores[(l, r)] = oh.update_decode_scores(st.scores[(l, r)], norm[b], int(lens[b]))
        live = {r: int(st.row_len[0, r]) for r in active_decode}
        groups = oh.decode_eviction_groups(live, rows, sorted(active_decode), budget=budget, interval=interval,
                                           num_free_slots=int(st.free_ptr.min()))
        if groups:
            oh.evict_decode_rows(st, range(L), groups, budget=budget, recent_ratio=sc["recent_ratio"])
        return 1e-4, 1e-6

    def on_step(rec):
        torch.cuda.synchronize()
        rtol, atol = (oracle_prefill if rec["prefill"] else oracle_decode)(rec)
        # ---- device state == oracle mirror
        np.testing.assert_array_equal(np.stack(cm.row_seq_lens), st.row_len)
        np.testing.assert_array_equal(np.asarray(cm._num_free_slots), st.free_ptr)
        tab, stack, score = (t.cpu().numpy() for t in (cm.buffer_req_to_token_slots_tensor, cm.free_slots_stack_tensor, cm.h2o_score_tensor))
        live_rows = sorted(set(cm.seq_id_to_row[0].values()))
        for l in range(L):
            for r in live_rows:
                n = int(st.row_len[l, r])
                np.testing.assert_array_equal(tab[l, r, :n], st.slot_table[l, r, :n], err_msg=f"slot table step {seen['steps']} layer {l} row {r}")
                assert (tab[l, r, n:] == 0).all()
                if (l, r) in st.scores:
                    np.testing.assert_allclose(score[l, r, :n], st.scores[(l, r)], rtol=rtol, atol=atol)
            p = int(st.free_ptr[l])
            np.testing.assert_array_equal(stack[l, :p], st.free_stack[l, :p])
        # ---- the plan == the reference scheduler's
        lens = {s.seq_id - base: int(cm.row_seq_lens[0][row_of(s)]) for s in seqs if s.seq_id in cm.seq_id_to_row[0]}
        _check_plan_record(seen["steps"], rec, want[seen["steps"]], planner, base, lens,
                           dict(free=[int(x) for x in cm._num_free_slots], counters={k: int(v) for k, v in cm._h2o_counters.items()}))
        for s in rec["finished"]:
            r = row_of(s)
            active_decode.discard(r)
            oh.free_seq(st, range(L), r)
        seen["steps"] += 1

    plan = drv.run(planner, seqs, StepInputs(L, drv.device), on_step=on_step)
    assert len(plan) == len(want) == seen["steps"]
    # what the scenario was built to contain
    assert any(r["deferred"] for r in want), "admission never deferred"
    assert sum(r["prefill"] for r in want) >= 10 and any(len(r["seqs"]) >= 3 for r in want if r["prefill"])
    assert cm._h2o_counters["intermediate_prefill_evictions"] > 0 and cm._h2o_counters["decode_eviction_bursts"] > 0
    assert drv.graph_stats["replayed"] >= 10, drv.graph_stats
    # everything was returned: all rows free, every slot back on the stacks
    assert all(n == cm.num_slots for n in cm._num_free_slots) and not cm.seq_id_to_row[0]
    np.testing.assert_array_equal(_f(cm.kv_cache[0]), kc)
    np.testing.assert_array_equal(_f(cm.kv_cache[1]), vc)


# ------------------------------------------------------------------------------------------------------------ Quest
def test_quest_planned_run_matches_reference_plan_and_oracle(ref_plan):
    from sparse_vllm_amd.config import Config
    from sparse_vllm_amd.engine.decode_driver import SparseDecodeDriver
    sc = prs.QUEST
    L, page, skip = sc["layers"], sc["page"], sc["skip_layers"]
    conf = Config.from_kwargs(sparse_method="quest", num_hidden_layers=L, max_model_len=sc["max_model_len"],
                              max_num_seqs_in_gpu=sc["rows"], num_kvcache_slots=sc["pages"] * page, sink_keep_tokens=sc["sink"],
                              recent_keep_tokens=sc["recent"], decode_keep_tokens=sc["keep"], quest_skip_layers=skip,
                              engine_prefill_chunk_size=sc["planner"]["chunk_prefill_size"])
    assert conf.quest_token_budget == sc["token_budget"]
    drv = SparseDecodeDriver(conf)
    cm = drv.cache_manager
    assert cm.num_pages == sc["pages"] and cm.page_size == page
    cm.free_pages_cpu_stack = np.asarray(ref_plan["quest"]["free_pages_stack"], dtype=np.int32)   # the reference run's stack
    cm._dev_state_dirty = True
    drv.enable_decode_graph()
    planner = _planner(sc, cm)
    seqs = _prompts(sc)
    base = seqs[0].seq_id
    want = ref_plan["quest"]["trace"]
    kc, vc = _f(cm.kv_cache[0]).copy(), _f(cm.kv_cache[1]).copy()          # mirror written through the REFERENCE's page tables
    seen = dict(steps=0, sparse_views=0, dense_views=0)

    def token_slots(pages, lo, hi):
        pos = np.arange(lo, hi)
        return np.asarray(pages, dtype=np.int64)[pos // page] * page + pos % page

    def on_step(rec):
        torch.cuda.synchronize()
        step, w = seen["steps"], want[seen["steps"]]
        rows = np.array([cm.seq_id_to_row[s.seq_id] for s in rec["seqs"]], np.int32)
        lens_now = cm.row_seq_lens[rows].astype(np.int32)
        qn, kn, vn, got_o = _f(rec["q"]), _f(rec["k"]), _f(rec["v"]), _f(rec["outputs"])
        # ---- plan + paging against the reference run
        lens, tables = {}, {}
        for s in seqs:
            r = cm.seq_id_to_row.get(s.seq_id)
            if r is not None:
                n = int(cm.row_seq_lens[r])
                lens[s.seq_id - base] = n
                tables[str(s.seq_id - base)] = [int(x) for x in cm.buffer_req_to_page_slots_cpu[r, : (n + page - 1) // page]]
        _check_plan_record(step, rec, w, planner, base, lens,
                           dict(free_pages=int(cm._num_free_pages), free_slots=int(cm.num_free_slots), page_tables=tables))
        ttab, ptab = cm.buffer_req_to_token_slots.cpu().numpy(), cm.buffer_req_to_page_slots.cpu().numpy()
        np.testing.assert_array_equal(ptab, cm.buffer_req_to_page_slots_cpu)
        for key, pages in w["page_tables"].items():
            r = cm.seq_id_to_row[base + int(key)]
            n = w["lens"][key]
            np.testing.assert_array_equal(ttab[r, :n], token_slots(pages, 0, n))
        # ---- the step's K/V through the reference's tables into the mirror; the device cache must hold the same bytes
        off = 0
        for s, c, r, n in zip(rec["seqs"], rec["chunks"], rows, lens_now):
            dst = token_slots(w["page_tables"][str(s.seq_id - base)], n - c, n)
            for l in range(L):
                kc[l][dst], vc[l][dst] = kn[l][off: off + c], vn[l][off: off + c]
            off += c
        np.testing.assert_array_equal(_f(cm.kv_cache[0]), kc)
        np.testing.assert_array_equal(_f(cm.kv_cache[1]), vc)
        # ---- min / max metadata of every complete page of the scheduled rows: exact
        md = _f(cm.metadata_cache)
        for l in range(L):
            for r, n in zip(rows, lens_now):
                full = ptab[r, : n // page].astype(np.int64)
                pmax, pmin = oq.page_minmax(kc[l], full, page)
                np.testing.assert_array_equal(md[0, l][full], pmax)
                np.testing.assert_array_equal(md[1, l][full], pmin)
        # ---- attention outputs
        if rec["prefill"]:
            chunk = np.array(rec["chunks"], np.int32)
            starts = np.concatenate(([0], np.cumsum(chunk)[:-1])).astype(np.int32)
            for l in range(L):
                ref_o = opa.context_attention_fwd(qn[l], kc[l], vc[l], rows, starts, lens_now, lens_now - chunk, ttab)
                np.testing.assert_allclose(got_o[l], bf16_round(ref_o), rtol=TOL, atol=TOL, err_msg=f"prefill layer {l} step {step}")
        else:
            long_text = planner._long_text_threshold() < int(rec["seqs"][0].num_tokens) - 1      # as scheduled (before the append)
            assert bool(drv.is_long_text) == long_text
            max_ctx = int(cm.layer_batch_state.max_context_len)
            for l in range(L):
                res = None
                if l >= skip:
                    res = oq.build_decode_view(qn[l], md[0, l], md[1, l], ttab, ptab, rows, lens_now, page_size=page,
                                               token_budget=sc["token_budget"], max_context_len=max_ctx,
                                               max_pages_per_row=cm.max_pages_per_row, num_kv_heads=Hkv, is_long_text=long_text)
                if res is None:
                    mid, lse = oda.flash_decode_stage1(qn[l], kc[l], vc[l], ttab, rows, lens_now, int(lens_now.max()), 64)
                    o = oda.flash_decode_stage2(mid, lse, lens_now, 64)
                    seen["dense_views"] += 1
                else:
                    packed, lreq, llens, _ = res
                    mid, lse = oda.flash_decode_stage1(qn[l], kc[l], vc[l], packed, lreq, llens, int(packed.shape[1]), 64)
                    o = oda.flash_decode_stage2(mid, lse, llens, 64)
                    seen["sparse_views"] += int((llens < lens_now).any())
                np.testing.assert_allclose(got_o[l], bf16_round(o), rtol=3e-2, atol=3e-2, err_msg=f"decode layer {l} step {step}")
        seen["steps"] += 1

    plan = drv.run(planner, seqs, StepInputs(L, drv.device), on_step=on_step)
    assert len(plan) == len(want) == seen["steps"]
    assert any(r["deferred"] for r in want), "admission never deferred"
    assert seen["sparse_views"] > 0 and seen["dense_views"] > 0             # long rows attended through a query-aware view
    assert drv.graph_stats["replayed"] >= 10, drv.graph_stats
    assert cm._num_free_pages == cm.num_pages and not cm.seq_id_to_row


# ------------------------------------------------------------------------------------------------------------ StreamingLLM
def test_streamingllm_planned_run_reproduces_the_reference_bit_for_bit(ref_plan):
    """Sink + recent window eviction has no scores in it: from the reference run's initial free stack the GPU run must
    reproduce the reference's plan AND its slot tables, free-stack contents (crc of the live part, order included) and row
    lengths after every step - prefill chunks, final-chunk eviction, decode re-eviction at twice the budget, rows released -
    while the attention outputs are checked against the oracle over those tables."""
    import zlib
    from sparse_vllm_amd.config import Config
    from sparse_vllm_amd.engine.decode_driver import SparseDecodeDriver
    sc = prs.STREAMINGLLM
    L = sc["layers"]
    conf = Config.from_kwargs(sparse_method="streamingllm", num_hidden_layers=L, max_model_len=sc["max_model_len"],
                              max_num_seqs_in_gpu=sc["rows"], num_kvcache_slots=sc["slots"], sink_keep_tokens=sc["sink"],
                              recent_keep_tokens=sc["recent"], engine_prefill_chunk_size=sc["planner"]["chunk_prefill_size"])
    drv = SparseDecodeDriver(conf)
    cm = drv.cache_manager
    ref = ref_plan["streamingllm"]
    cm.free_slots_stack_tensor.copy_(torch.tensor(ref["initial_free_stack"], dtype=torch.int32))
    cm._dev_state_dirty = True
    drv.enable_decode_graph()
    planner = _planner(sc, cm)
    seqs = _prompts(sc)
    base = seqs[0].seq_id
    want = ref["trace"]
    kc, vc = _f(cm.kv_cache[0]).copy(), _f(cm.kv_cache[1]).copy()
    prev_tables: dict[int, list] = {}                 # seq index -> per-layer slot lists after its last step (the reference's)
    seen = dict(steps=0, evicting_decode_steps=0)

    def on_step(rec):
        torch.cuda.synchronize()
        step, w = seen["steps"], want[seen["steps"]]
        qn, kn, vn, got_o = _f(rec["q"]), _f(rec["k"]), _f(rec["v"]), _f(rec["outputs"])
        ids = [s.seq_id - base for s in rec["seqs"]]
        chunk = np.array(rec["chunks"], np.int32)
        starts = np.concatenate(([0], np.cumsum(chunk)[:-1])).astype(np.int32)
        # ---- the rows as the attention saw them: the reference's table of the previous step + this step's new slots
        for l in range(L):
            new_slots = cm.layer_batch_states[l].slot_mapping.cpu().numpy()[: int(chunk.sum())]
            width = max(len(prev_tables.get(i, [[]] * L)[l]) + int(c) for i, c in zip(ids, chunk))
            table = np.zeros((len(ids), width), np.int32)
            lens = np.zeros(len(ids), np.int32)
            for b, (i, c) in enumerate(zip(ids, chunk)):
                old = prev_tables.get(i, [[]] * L)[l]
                row = list(old) + [int(x) for x in new_slots[starts[b]: starts[b] + c]]
                table[b, : len(row)], lens[b] = row, len(row)
                kc[l][row[len(old):]] = kn[l][starts[b]: starts[b] + c]
                vc[l][row[len(old):]] = vn[l][starts[b]: starts[b] + c]
            req = np.arange(len(ids), dtype=np.int32)
            if rec["prefill"]:
                ref_o = opa.context_attention_fwd(qn[l], kc[l], vc[l], req, starts, lens, lens - chunk, table)
            else:
                mid, lse = oda.flash_decode_stage1(qn[l], kc[l], vc[l], table, req, lens, int(lens.max()), 64)
                ref_o = oda.flash_decode_stage2(mid, lse, lens, 64)
            np.testing.assert_allclose(got_o[l], bf16_round(ref_o), rtol=TOL, atol=TOL, err_msg=f"layer {l} step {step}")
        # ---- plan, lengths, slot tables and free stacks == the reference run
        live = {s.seq_id - base: s for s in seqs if s.seq_id in cm.seq_id_to_row[0]}
        lens_now = {i: int(cm.row_seq_lens[0][cm.seq_id_to_row[0][s.seq_id]]) for i, s in live.items()}
        tab = cm.buffer_req_to_token_slots_tensor.cpu().numpy()
        stack = cm.free_slots_stack_tensor.cpu().numpy()
        tables = {str(i): [[int(x) for x in tab[l, cm.seq_id_to_row[l][s.seq_id], : lens_now[i]]] for l in range(L)] for i, s in live.items()}
        crc = [int(zlib.crc32(stack[l, : int(cm._num_free_slots[l])].astype(np.int32).tobytes())) for l in range(L)]
        _check_plan_record(step, rec, w, planner, base, lens_now,
                           dict(free=[int(x) for x in cm._num_free_slots], slot_tables=tables, free_stack_crc=crc))
        if not rec["prefill"] and any(w["lens"][str(i)] < (prev_tables.get(i) and len(prev_tables[i][0]) or 0) for i in ids):
            seen["evicting_decode_steps"] += 1
        for i in live:
            prev_tables[i] = w["slot_tables"][str(i)]
        for s in rec["finished"]:
            prev_tables.pop(s.seq_id - base, None)
        np.testing.assert_array_equal(_f(cm.kv_cache[0]), kc)
        np.testing.assert_array_equal(_f(cm.kv_cache[1]), vc)
        seen["steps"] += 1

    plan = drv.run(planner, seqs, StepInputs(L, drv.device), on_step=on_step)
    assert len(plan) == len(want) == seen["steps"]
    assert any(r["deferred"] for r in want) and seen["evicting_decode_steps"] >= 3
    assert drv.graph_stats["replayed"] >= 10, drv.graph_stats
    assert all(n == cm.num_slots for n in cm._num_free_slots) and not cm.seq_id_to_row[0]


# ------------------------------------------------------------------------------------------------------------ SnapKV
def test_snapkv_planned_run_matches_reference_plan_and_oracle(ref_plan):
    """SnapKV through the engine loop: whole prompts resident until their final chunk (which must hold the score window:
    `min_final_prefill_chunk_size`), final-chunk selection sink + top-k of the window's scores + recent, decode re-eviction at
    twice the top budget from the step's head-max raw scores.  Plan, queues, deferred prompts, lengths and free counts equal
    the reference run's; slot tables, free stacks and outputs equal the chained oracle's (`oracle/snapkv.py`)."""
    from oracle import snapkv as osk
    from sparse_vllm_amd.config import Config
    from sparse_vllm_amd.engine.decode_driver import SparseDecodeDriver
    sc = prs.SNAPKV
    L, sink, recent, keep, window = sc["layers"], sc["sink"], sc["recent"], sc["keep"], sc["window"]
    budget = sink + keep + recent
    conf = Config.from_kwargs(sparse_method="snapkv", num_hidden_layers=L, max_model_len=sc["max_model_len"],
                              max_num_seqs_in_gpu=sc["rows"], num_kvcache_slots=sc["slots"], sink_keep_tokens=sink,
                              recent_keep_tokens=recent, decode_keep_tokens=keep, snapkv_window_size=window,
                              engine_prefill_chunk_size=sc["planner"]["chunk_prefill_size"])
    drv = SparseDecodeDriver(conf)
    cm = drv.cache_manager
    cm.permute_free_slots(6)
    drv.enable_decode_graph()
    planner = _planner(sc, cm)
    seqs = _prompts(sc)
    base = seqs[0].seq_id
    want = ref_plan["snapkv"]["trace"]
    st = oh.SlotState(cm.buffer_req_to_token_slots_tensor.cpu().numpy().copy(), cm.free_slots_stack_tensor.cpu().numpy().copy(),
                      np.asarray(cm._num_free_slots, dtype=np.int64), np.stack(cm.row_seq_lens).astype(np.int32))
    kc, vc = _f(cm.kv_cache[0]).copy(), _f(cm.kv_cache[1]).copy()
    acc: dict = {}                                   # (layer, row) -> element-wise max of the window scores
    seen = dict(steps=0, decode_evictions=0, prefill_evictions=0)

    def row_of(s):
        return cm.seq_id_to_row[0][s.seq_id]

    def on_step(rec):
        torch.cuda.synchronize()
        step = seen["steps"]
        active, chunk = rec["seqs"], np.array(rec["chunks"], np.int32)
        rows = [row_of(s) for s in active]
        qn, kn, vn, got_o = _f(rec["q"]), _f(rec["k"]), _f(rec["v"]), _f(rec["outputs"])
        req = np.array(rows, np.int32)
        if rec["prefill"]:
            starts = np.concatenate(([0], np.cumsum(chunk)[:-1])).astype(np.int32)
            finals = [s.num_prefilled_tokens >= s.num_prompt_tokens for s in active]         # (postprocess has run)
            done = [s.num_prefilled_tokens - int(c) for s, c in zip(active, chunk)]
            scored = osk.prefill_score_rows([s.num_prompt_tokens for s in active], done, chunk, budget=budget, window=window)
            for l in range(L):
                ctx = []
                for b, (r, n) in enumerate(zip(rows, chunk)):
                    new = oh.allocate(st, l, r, int(n))
                    kc[l][new], vc[l][new] = kn[l][starts[b]: starts[b] + n], vn[l][starts[b]: starts[b] + n]
                    ctx.append(int(st.row_len[l, r]))
                ctx = np.array(ctx, np.int32)
                ref_o = opa.context_attention_fwd(qn[l], kc[l], vc[l], req, starts, ctx, ctx - chunk, st.slot_table[l])
                np.testing.assert_allclose(got_o[l], bf16_round(ref_o), rtol=TOL, atol=TOL, err_msg=f"prefill layer {l} step {step}")
                if scored:
                    sb = [b for b, _, _ in scored]
                    qs = np.array([s0 for _, s0, _ in scored], np.int32)
                    qe = np.array([e0 for _, _, e0 in scored], np.int32)
                    scs = np.empty((len(sb), int(ctx[sb].max())), np.float32)
                    ops.prefill_score_fwd(qn[l], kc[l], scs, req[sb], starts[sb], ctx[sb], (ctx - chunk)[sb], int((qe - qs).max()),
                                          st.slot_table[l], qs, qe, candidate_start=sink, num_recent_tokens=recent)
                    for j, b in enumerate(sb):
                        acc[(l, rows[b])] = osk.accumulate_prefill_score(acc.get((l, rows[b])), scs[j, : ctx[b]], mode="probability")
            before = int(st.free_ptr[0])
            osk.snapkv_prefill_eviction(st, range(L), rows, [int(st.row_len[0, r]) for r in rows], finals, acc, sink=sink,
                                        recent=recent, keep=keep)
            seen["prefill_evictions"] += int(int(st.free_ptr[0]) != before)
            for r, fin in zip(rows, finals):
                if fin:
                    for l in range(L):
                        acc.pop((l, r), None)
        else:
            new_slots = oh.decode_allocate_batch_layers(st, range(L), rows)
            lens = np.array([st.row_len[0, r] for r in rows], dtype=np.int32)
            raws = {}
            for l in range(L):
                kc[l][new_slots[l]], vc[l][new_slots[l]] = kn[l], vn[l]
                raw = np.full((len(rows), int(lens.max())), -1e20, dtype=np.float32)
                mid, lse = oda.flash_decode_stage1(qn[l], kc[l], vc[l], st.slot_table[l], req, lens, int(lens.max()), 64, attn_score=raw)
                o = oda.flash_decode_stage2(mid, lse, lens, 64)
                np.testing.assert_allclose(got_o[l], bf16_round(o), rtol=TOL, atol=TOL, err_msg=f"decode layer {l} step {step}")
                raws[l] = raw
            before = int(st.free_ptr[0])
            osk.snapkv_decode_eviction(st, range(L), rows, raws, sink=sink, recent=recent, keep=keep)
            seen["decode_evictions"] += int(int(st.free_ptr[0]) != before)
        # ---- device state == oracle mirror, plan == reference
        np.testing.assert_array_equal(np.stack(cm.row_seq_lens), st.row_len)
        np.testing.assert_array_equal(np.asarray(cm._num_free_slots), st.free_ptr)
        tab, stack = cm.buffer_req_to_token_slots_tensor.cpu().numpy(), cm.free_slots_stack_tensor.cpu().numpy()
        for l in range(L):
            for r in sorted(set(cm.seq_id_to_row[0].values())):
                n = int(st.row_len[l, r])
                np.testing.assert_array_equal(tab[l, r, :n], st.slot_table[l, r, :n], err_msg=f"slot table step {step} layer {l} row {r}")
            p = int(st.free_ptr[l])
            np.testing.assert_array_equal(stack[l, :p], st.free_stack[l, :p])
        lens_now = {s.seq_id - base: int(cm.row_seq_lens[0][row_of(s)]) for s in seqs if s.seq_id in cm.seq_id_to_row[0]}
        _check_plan_record(step, rec, want[step], planner, base, lens_now, dict(free=[int(x) for x in cm._num_free_slots]))
        for s in rec["finished"]:
            oh.free_seq(st, range(L), row_of(s))
        seen["steps"] += 1

    plan = drv.run(planner, seqs, StepInputs(L, drv.device), on_step=on_step)
    assert len(plan) == len(want) == seen["steps"]
    assert any(r["deferred"] for r in want) and seen["prefill_evictions"] >= 4 and seen["decode_evictions"] >= 6
    assert drv.graph_stats["replayed"] >= 10, drv.graph_stats
    assert all(n == cm.num_slots for n in cm._num_free_slots) and not cm.seq_id_to_row[0]
    np.testing.assert_array_equal(_f(cm.kv_cache[0]), kc)
    np.testing.assert_array_equal(_f(cm.kv_cache[1]), vc)
