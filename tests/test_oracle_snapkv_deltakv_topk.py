"""CPU: the oracle's SnapKV / StreamingLLM selection, prefill accumulator, decode re-eviction and the DeltaKV
observation-layer token scores + sorted top-k against fixtures produced by the reference's own functions
(tests/golden/gen_fixtures.py groups snapkv_select, snapkv_e2e, deltakv_topk)."""

import numpy as np

from oracle import bf16_bits_to_f32
from oracle import deltakv as od
from oracle import h2o as oh
from oracle import prefill_score as ops
from oracle import snapkv as osk


def _state(g, prefix):
    return oh.SlotState(g[f"{prefix}_slot_table"].copy(), g[f"{prefix}_free_stack"].copy(),
                        g[f"{prefix}_free_ptr"].copy(), g[f"{prefix}_row_len"].copy())


def _assert_state(st, g, prefix):
    np.testing.assert_array_equal(st.row_len, g[f"{prefix}_row_len"])
    np.testing.assert_array_equal(st.free_ptr, g[f"{prefix}_free_ptr"])
    np.testing.assert_array_equal(st.slot_table, g[f"{prefix}_slot_table"])
    for l in range(st.free_ptr.size):
        p = int(st.free_ptr[l])
        np.testing.assert_array_equal(st.free_stack[l, :p], g[f"{prefix}_free_stack"][l, :p])


def test_snapkv_select_vs_reference(golden):
    g = golden("snapkv_select")
    identical = 0
    for name in g["names"]:
        kv_len, sink, recent, keep, pool, budget, tie_free = (int(x) for x in g[f"{name}_cfg"])
        scores = g[f"{name}_scores"]
        for b in range(scores.shape[0]):
            mine = osk.snapkv_select_indices(scores[b, :kv_len], kv_len, budget, sink=sink, recent=recent, pool=pool)
            assert mine.size == min(budget, sink + recent + max(0, kv_len - recent - sink))
            for ref in (g[f"{name}_keep_batch"][b], g[f"{name}_keep_scalar"][b]):
                same = osk.check_keep_set(scores[b, :kv_len], kv_len, budget, mine, ref, sink=sink, recent=recent, pool=pool)
                if tie_free:
                    assert same, f"case {name} row {b}: tie-free threshold but the sets differ"
                identical += same
    assert identical >= 20


def test_snapkv_budget_trigger_streamingllm_vs_reference(golden):
    g = golden("snapkv_select")
    for sink, recent, keep, budget, trig in g["trigger"]:
        assert osk.snapkv_layer_budget(sink, keep, recent) == budget
        assert osk.snapkv_decode_trigger_len(budget, sink, recent) == trig
    for sink, recent, kv_len, budget in g["sl_cases"]:
        np.testing.assert_array_equal(osk.streamingllm_select_indices(int(kv_len), int(sink), int(recent)),
                                      g[f"sl_{sink}_{recent}_{kv_len}"])
        assert (osk.streamingllm_budget(sink, recent) or -1) == budget


def test_max_pool1d_matches_torch():
    import torch
    rng = np.random.default_rng(0)
    for n, k in ((17, 3), (50, 5), (9, 7), (6, 9), (12, 4)):
        x = rng.standard_normal((2, n)).astype(np.float32)
        ref = torch.nn.functional.max_pool1d(torch.from_numpy(x)[:, None, :], kernel_size=k, padding=k // 2, stride=1)[:, 0].numpy()
        np.testing.assert_array_equal(osk.max_pool1d_same(x, k), ref)


def test_snapkv_prefill_collect_and_eviction_vs_reference(golden):
    """collect_prefill_attention_score (real prefill_score_fwd under the interpreter) -> accumulator -> selection ->
    free_part_slots, for probability / logits scores and a pooled selection."""
    g = golden("snapkv_e2e")
    q, k = bf16_bits_to_f32(g["p_q"]), bf16_bits_to_f32(g["p_k"])
    for tag in ("pf", "pl", "pp"):
        sink, recent, keep, window, budget, logits, pool = (int(x) for x in g[f"{tag}_cfg"])
        mode = "logits" if logits else "probability"
        prompts = [int(x) for x in g[f"{tag}_prompts"]]
        st = _state(g, f"{tag}_before")
        L = st.row_len.shape[0]
        rows = osk.prefill_score_rows(prompts, [0] * len(prompts), prompts, budget=budget, window=window)
        np.testing.assert_array_equal([r[0] for r in rows], g[f"{tag}_scored"])
        starts = np.concatenate(([0], np.cumsum(prompts)[:-1])).astype(np.int32)
        ctx = np.array(prompts, np.int32)
        scores = {}
        for l in range(L):
            bidx = np.array([r[0] for r in rows], np.int32)
            step = np.empty((len(rows), max(prompts[b] for b in bidx)), np.float32)
            ops.prefill_score_fwd(q[l], k[l], step, np.arange(len(prompts), dtype=np.int32), starts, ctx,
                                  np.zeros(len(prompts), np.int32), max(r[2] - r[1] for r in rows), st.slot_table[l],
                                  np.array([r[1] for r in rows], np.int32), np.array([r[2] for r in rows], np.int32),
                                  candidate_start=sink, num_recent_tokens=recent, score_mode=mode, batch_indices=bidx)
            for i, (b, _, _) in enumerate(rows):
                acc = osk.accumulate_prefill_score(None, step[i, :prompts[b]], mode=mode)
                ref = g[f"{tag}_acc_{l}_{b}"]
                fin = np.isfinite(ref)
                np.testing.assert_array_equal(np.isfinite(acc), fin)
                np.testing.assert_allclose(acc[fin], ref[fin], rtol=2e-5, atol=1e-6)
                scores[(l, b)] = ref           # select on the reference's own scores: the state must then be bit-exact
        osk.snapkv_prefill_eviction(st, range(L), range(len(prompts)), prompts, [True] * len(prompts), scores,
                                    sink=sink, recent=recent, keep=keep, pool=pool)
        if pool == 1:
            _assert_state(st, g, f"{tag}_after")
        else:
            # pooled scores tie at the threshold: lengths / counters exact, rows equal as sets up to those ties
            np.testing.assert_array_equal(st.row_len, g[f"{tag}_after_row_len"])
            np.testing.assert_array_equal(st.free_ptr, g[f"{tag}_after_free_ptr"])
            before = g[f"{tag}_before_slot_table"]
            for l in range(L):
                for b, _, _ in rows:
                    n = prompts[b]
                    pos_of = {int(s): i for i, s in enumerate(before[l, b, :n])}
                    got = np.array([pos_of[int(s)] for s in st.slot_table[l, b, :budget]])
                    ref = np.array([pos_of[int(s)] for s in g[f"{tag}_after_slot_table"][l, b, :budget]])
                    osk.check_keep_set(scores[(l, b)], n, budget, got, ref, sink=sink, recent=recent, pool=pool)


def test_snapkv_accumulator_lifecycle_vs_reference(golden):
    g = golden("snapkv_e2e")
    for mode in ("probability", "logits"):
        init = g[f"acc_init_{mode}"]
        assert (init == osk.prefill_score_initial_value(mode)).all()
        s1, s2 = g[f"acc_steps_{mode}"]
        acc = osk.accumulate_prefill_score(None, s1, mode=mode)
        acc = osk.accumulate_prefill_score(acc, s2, mode=mode)      # a later chunk keeps the running maximum
        np.testing.assert_array_equal(acc, g[f"acc_final_{mode}"])
        np.testing.assert_array_equal(g[f"acc_reset_{mode}"], init)    # num_prefilled_tokens == 0 starts over


def test_snapkv_decode_eviction_vs_reference(golden):
    g = golden("snapkv_e2e")
    for tag in ("dg", "ds", "dn", "dm", "dq"):
        sink, recent, keep, budget, trigger = (int(x) for x in g[f"{tag}_cfg"])
        assert osk.snapkv_decode_trigger_len(budget, sink, recent) == trigger
        st = _state(g, f"{tag}_before")
        L, R = st.row_len.shape
        osk.snapkv_decode_eviction(st, range(L), list(range(R)), {l: g[f"{tag}_score_{l}"] for l in range(L)},
                                   sink=sink, recent=recent, keep=keep)
        _assert_state(st, g, f"{tag}_after")
        if tag == "dn":
            np.testing.assert_array_equal(st.row_len, g[f"{tag}_before_row_len"])


def test_deltakv_token_scores_and_topk_vs_reference(golden):
    g = golden("deltakv_topk")
    exact = total = 0
    for name in g["names"]:
        sink, recent, keep, dt = (int(x) for x in g[f"{name}_cfg"])
        dtype = {0: "float32", 1: "bfloat16"}[dt]
        raw = bf16_bits_to_f32(g[f"{name}_raw"])
        clens = g[f"{name}_clens"]
        ref_ts = g[f"{name}_token_scores"]
        ts = od.token_scores_full(raw, candidate_start=sink, candidate_lens=clens, scale=128 ** -0.5, model_dtype=dtype)
        fill = ref_ts.min()
        np.testing.assert_array_equal(ts == fill, ref_ts == fill)             # same masked positions, same fill value
        valid = ref_ts != fill
        # fp32 softmax in another summation order, then one bf16 rounding: at most one bf16 ulp apart
        np.testing.assert_allclose(ts[valid], ref_ts[valid], rtol=2.0 ** -7 if dtype == "bfloat16" else 2e-6, atol=1e-12)
        if dtype == "bfloat16":
            assert (ts[valid] == ref_ts[valid]).mean() > 0.98
        # the masked search row the reference's topk saw
        keys = od.dynamic_topk_keys(ref_ts, sink=sink, compressed_lens=clens, model_dtype=dtype)
        np.testing.assert_array_equal(keys, g[f"{name}_search_masked"])
        for tb, ref_idx in ((False, g[f"{name}_topk"]), (True, g[f"{name}_topk_tiebreak"])):
            keys = od.dynamic_topk_keys(ref_ts, sink=sink, compressed_lens=clens, tiebreak=tb, model_dtype=dtype)
            mine = od.dynamic_topk_indices(ref_ts, sink=sink, compressed_lens=clens, keep=keep, tiebreak=tb, model_dtype=dtype)
            assert mine.shape == ref_idx.shape and mine.dtype == ref_idx.dtype
            for b in range(mine.shape[0]):
                same = od.check_sorted_topk(keys[b], mine[b], ref_idx[b])
                total += 1
                exact += same
                # (even with the tie-break key neighbours can still tie after the fp32 add - fixture a, row 1 -
                # so identity is not required; check_sorted_topk is the contract)
    assert exact >= 4 and total == 18
