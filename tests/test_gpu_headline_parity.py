"""Oracle-chained end-to-end parity at the sizes BASELINE.json quotes.

* H2O at configs[2]: decode_budget 4096, eviction interval 128, 28 q / 4 kv heads x 128, max_model_len 131072 (the
  slot-table row stride), permuted slot pool, three rows of which one is five tokens behind (so the bursts of one
  interval split over two steps), two layers, 140 decode steps replayed as hipGraphs with the bookkeeping resident on
  the device.  After EVERY step the pinned numpy oracle (oracle/h2o.py + oracle/decode_attention.py, chained exactly as
  tests/test_gpu_h2o_e2e.py does at toy budgets) must agree: slot tables, free-stack contents and order, row lengths
  bit-exact; cumulative scores rtol 1e-4; attention outputs 2e-2.  A selection mismatch prints the |score - threshold|
  of the disagreeing tokens (SURVEY 7 "hard parts" (2)).
* Quest at configs[3]: 4 x 131 072 tokens, one sparse layer: page scores against oracle/quest.py (bf16-valued, compared
  exactly up to the bf16 rounding of near-tie sums), the decode view as a SET of pages checked against the oracle's
  scores with `check_topk_set`.
* DeltaKV at configs[4]: one observation layer at 262 152 tokens: `oracle.deltakv.token_scores_full` +
  `dynamic_topk_indices`, compared with `check_sorted_topk`.
"""

import os

import numpy as np
import pytest

from oracle import bf16_round
from oracle import decode_attention as oda
from oracle import deltakv as odk
from oracle import h2o as oh
from oracle import quest as oq

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _bf(t):
    return t.float().cpu().numpy()


def _explain_selection_mismatch(old_row, got_row, ref_row, scores, budget, recent_ratio=0.5):
    """Both rows were compacted from `old_row` (length n).  Which positions do the two keep sets disagree on, and how far
    are their cumulative scores from the selection threshold (the heavy_count-th largest candidate score)?"""
    n = old_row.shape[0]
    pos_of = {int(s): i for i, s in enumerate(old_row)}
    got = np.array(sorted(pos_of[int(s)] for s in got_row if int(s) in pos_of))
    ref = np.array(sorted(pos_of[int(s)] for s in ref_row))
    heavy, recent = oh.h2o_budget_partition(budget, recent_ratio)
    cand = scores[: n - recent]
    thr = np.sort(cand)[::-1][heavy - 1]
    diff = np.setxor1d(got, ref)
    lines = [f"  pos {int(p)}: score {float(scores[p])!r}  |score - thr| = {abs(float(scores[p]) - float(thr)):.3e}"
             for p in diff[:16]]
    return (f"keep sets differ on {diff.size} positions (threshold = {float(thr)!r}, heavy {heavy} of {cand.size} candidates)\n"
            + "\n".join(lines))


def test_h2o_headline_config_chained_oracle_under_graph_replay():
    from sparse_vllm_amd.config import Config
    from tools.synthetic import SyntheticDecodeDriver as SparseDecodeDriver
    B, L, budget, interval, Hq, Hkv, D = 3, 2, 4096, 128, 28, 4, 128
    behind, steps, block_seq = 5, 140, 256
    conf = Config.from_kwargs(sparse_method="h2o", num_hidden_layers=L, num_attention_heads=Hq, num_key_value_heads=Hkv,
                              head_dim=D, max_model_len=131072, max_num_seqs_in_gpu=B + 1,
                              num_kvcache_slots=B * (budget + interval) + 4096, h2o_decode_budget=budget,
                              h2o_decode_eviction_interval=interval, h2o_prefill_budget=2 * budget)
    drv = SparseDecodeDriver(conf)
    cm = drv.cache_manager
    device_state = os.environ.get("SVK_H2O_DEVICE_STATE", "1") != "0"          # (the opt-out knob: host-driven steps)
    assert cm._device_step_enabled == device_state, "the headline path keeps its bookkeeping on the device"
    cm.permute_free_slots(20260625)
    seqs = drv.admit_resident_rows(B, budget, logical_len=131072, seed=5)
    # the last row is `behind` tokens behind the others: its burst falls on a later step of the interval
    for l in range(L):
        keep = torch.arange(budget - behind, device=drv.device)
        cm.free_part_slots(l, seqs[-1], keep, keep_indices_sorted=True)
    rows = [cm.seq_id_to_row[0][s.seq_id] for s in seqs]
    drv.enable_decode_graph()

    # ---- mirror the initial device state into the oracle
    torch.cuda.synchronize()
    st = oh.SlotState(cm.buffer_req_to_token_slots_tensor.cpu().numpy().copy(),
                      cm.free_slots_stack_tensor.cpu().numpy().copy(),
                      np.asarray(cm._num_free_slots, dtype=np.int64).copy(),
                      np.stack(cm.row_seq_lens).astype(np.int32).copy())
    kc, vc = _bf(cm.kv_cache[0]).copy(), _bf(cm.kv_cache[1]).copy()
    for l in range(L):
        for r in rows:
            n = int(st.row_len[l, r])
            st.scores[(l, r)] = cm.h2o_score_tensor[l, r, :n].cpu().numpy().copy()
    assert sorted(int(st.row_len[0, r]) for r in rows) == [budget - behind, budget, budget]

    q, k, v = drv.random_step_inputs(seed=1)            # fixed buffers (graph replay); refilled in place per step
    outs = torch.zeros((L, B, Hq, D), dtype=torch.bfloat16, device=drv.device)
    rows_np = np.array(rows, np.int32)
    burst_steps, worst_score_err, mismatches = [], 0.0, 0
    for step in range(steps):
        q2, k2, v2 = drv.random_step_inputs(seed=1000 + step)
        q.copy_(q2), k.copy_(k2), v.copy_(v2)
        drv.step(q, k, v, outputs=outs)
        torch.cuda.synchronize()

        # ---- oracle step
        new_slots = oh.decode_allocate_batch_layers(st, range(L), rows)
        lens = np.array([st.row_len[0, r] for r in rows], dtype=np.int32)
        qn, kn, vn = _bf(q), _bf(k), _bf(v)
        W = int(lens.max())
        for l in range(L):
            kc[l][new_slots[l]] = kn[l]
            vc[l][new_slots[l]] = vn[l]
            raw = np.full((B, W), -1e20, dtype=np.float32)
            mid, lse = oda.flash_decode_stage1(qn[l], kc[l], vc[l], st.slot_table[l], rows_np, lens, W, block_seq,
                                               attn_score=raw)
            o = oda.flash_decode_stage2(mid, lse, lens, block_seq)
            np.testing.assert_allclose(_bf(outs[l]), bf16_round(o), rtol=2e-2, atol=2e-2, err_msg=f"step {step} layer {l}")
            norm = oda.h2o_normalize_decode_scores(raw, D)
            for b, r in enumerate(rows):
                st.scores[(l, r)] = oh.update_decode_scores(st.scores[(l, r)], norm[b], int(lens[b]))
        row_lens = {r: int(st.row_len[0, r]) for r in rows}
        groups = oh.decode_eviction_groups(row_lens, rows, rows, budget=budget, interval=interval,
                                           num_free_slots=int(st.free_ptr.min()))
        pre = None
        if groups:
            burst_steps.append(step)
            pre = {(l, r): (st.slot_table[l, r, : row_lens[r]].copy(), st.scores[(l, r)].copy())
                   for l in range(L) for g in groups.values() for r in g}
            oh.evict_decode_rows(st, range(L), groups, budget=budget, recent_ratio=0.5)

        # ---- compare the full state: host mirrors, device copies, tables, stacks, scores
        np.testing.assert_array_equal(np.stack(cm.row_seq_lens), st.row_len, err_msg=f"step {step}")
        np.testing.assert_array_equal(np.asarray(cm._num_free_slots), st.free_ptr, err_msg=f"step {step}")
        if device_state:
            np.testing.assert_array_equal(cm._dev_row_len.cpu().numpy(), st.row_len, err_msg=f"device row lengths, step {step}")
            np.testing.assert_array_equal(cm._dev_free_ptr.cpu().numpy().reshape(-1), st.free_ptr)
        if True:
            for l in range(L):
                p = int(st.free_ptr[l])
                stack = cm.free_slots_stack_tensor[l, :p].cpu().numpy()
                for r in rows:
                    n = int(st.row_len[l, r])
                    tab = cm.buffer_req_to_token_slots_tensor[l, r, : n + 256].cpu().numpy()
                    if not np.array_equal(tab[:n], st.slot_table[l, r, :n]):
                        mismatches += 1
                        why = ""
                        if pre is not None and (l, r) in pre:
                            why = _explain_selection_mismatch(pre[(l, r)][0], tab[:n], st.slot_table[l, r, :n],
                                                              pre[(l, r)][1], budget)
                        raise AssertionError(f"slot table diverged at step {step} layer {l} row {r}\n{why}")
                    assert (tab[n:] == 0).all(), f"row tail not zeroed at step {step} layer {l} row {r}"
                np.testing.assert_array_equal(stack, st.free_stack[l, :p], err_msg=f"free stack, step {step} layer {l}")
        for l in range(L):
            for r in rows:
                n = int(st.row_len[l, r])
                got = cm.h2o_score_tensor[l, r, :n].cpu().numpy()
                np.testing.assert_allclose(got, st.scores[(l, r)], rtol=1e-4, atol=1e-6, err_msg=f"scores step {step} layer {l} row {r}")
                worst_score_err = max(worst_score_err, float(np.abs(got - st.scores[(l, r)]).max()))

    # two rows burst when they reach budget + interval, the late row `behind` steps after them
    assert burst_steps == [interval - 1, interval - 1 + behind], burst_steps
    assert cm._h2o_counters["decode_eviction_bursts"] == B
    if device_state:
        assert drv.graph_stats["replayed"] >= steps - 8, drv.graph_stats   # the device-resident step is one graph throughout
    assert mismatches == 0
    print(f"headline H2O chain: {steps} steps, bursts at {burst_steps}, worst |score err| {worst_score_err:.3e}")


def test_quest_full_size_against_pinned_oracle():
    """4 sequences x 131 072 tokens (8191 previous pages of 16), token budget 4672, Qwen2.5-7B heads, one sparse layer:
    `oracle.quest.score_pages_batched` for the scores, `oracle.quest.check_topk_set` on the oracle's scores for the view."""
    from sparse_vllm_amd.kernels import quest_ops
    d = torch.device("cuda:0")
    Hq, Hkv, D, ps, B, budget = 28, 4, 128, 16, 4, 4672
    lens_l = [131072, 131072 - 3, 90001, 131072 - 16 * 700]
    ctx = max(lens_l)
    pages = (ctx + ps - 1) // ps
    n_prev = pages - 1
    prev_budget = budget // ps - 1
    gen = torch.Generator(device=d).manual_seed(11)
    pool = pages * B + 5
    pmax = (torch.randn(pool, Hkv, D, device=d, generator=gen) * 0.5 + 1).bfloat16()
    pmin = (torch.randn(pool, Hkv, D, device=d, generator=gen) * 0.5 - 1).bfloat16()
    q = (torch.randn(B, Hq, D, device=d, generator=gen) * 0.5).bfloat16()
    ptab = torch.stack([torch.randperm(pool, device=d, generator=gen)[:pages] for _ in range(B)]).to(torch.int32)
    ttab = torch.zeros(B, ctx, dtype=torch.int32, device=d)
    req = torch.arange(B, dtype=torch.int32, device=d)
    lens = torch.tensor(lens_l, dtype=torch.int32, device=d)
    keep = (prev_budget + 1) * ps
    scores = torch.full((B, n_prev), 7.0, dtype=torch.float32, device=d)
    packed = torch.full((B, keep), -5, dtype=torch.int32, device=d)
    ll = torch.zeros(B, dtype=torch.int32, device=d)
    lr = torch.zeros(B, dtype=torch.int32, device=d)
    quest_ops.score_pages(q, pmax, pmin, ptab, req, lens, scores, page_size=ps, n_prev=n_prev)
    quest_ops.build_view(scores, ptab, ttab, req, lens, packed, ll, lr, page_size=ps, n_prev=n_prev, prev_budget=prev_budget,
                         token_budget=budget, page_budget_base=budget // ps, max_keep=keep, is_long_text=True)
    torch.cuda.synchronize()

    # ---- the pinned oracle on the same inputs (quest.py:1773-1913)
    ref = oq.build_decode_view(_bf(q), _bf(pmax), _bf(pmin), np.zeros((B, 1), np.int32), ptab.cpu().numpy(),
                               np.arange(B), np.array(lens_l), page_size=ps, token_budget=budget, max_context_len=ctx,
                               max_pages_per_row=pages, num_kv_heads=Hkv, is_long_text=True)
    assert ref is not None
    ref_packed, _, ref_lens, info = ref
    assert info["prev_budget"] == prev_budget and info["page_scores"].shape == (B, n_prev)
    got_scores = scores.cpu().numpy()
    got_packed = packed.cpu().numpy()
    np.testing.assert_array_equal(ll.cpu().numpy(), ref_lens)
    ptab_np = ptab.cpu().numpy()
    n_exact = n_total = 0
    for b in range(B):
        valid = info["valid"][b]
        rs, gs = info["page_scores"][b], got_scores[b]
        assert np.isneginf(gs[~valid]).all()
        # bf16-valued scores: a different summation order inside the fp32 dot can move a sum across a bf16 rounding
        # boundary, i.e. by at most one bf16 ulp of the score (2^-8 relative) + one ulp of each bf16 addend
        np.testing.assert_allclose(gs[valid], rs[valid], rtol=2 ** -6, atol=2 ** -6)
        n_exact += int((gs[valid] == rs[valid]).sum())
        n_total += int(valid.sum())
        # the view: ascending logical pages, each chosen once, the last page closing it; a valid top-k set of the
        # ORACLE's scores up to that one-ulp freedom at the threshold
        page_of = got_packed[b].reshape(-1, ps)
        assert (page_of % ps == np.arange(ps)).all()
        inv = np.full((pool,), -1, np.int64)
        inv[ptab_np[b]] = np.arange(pages)
        logical = inv[page_of[:, 0] // ps]
        assert (logical >= 0).all() and int(logical[-1]) == int(info["num_pages"][b]) - 1
        sel = logical[:-1]
        assert (sel[1:] > sel[:-1]).all() and valid[sel].all()
        oq.check_topk_set(np.where(valid, gs, -np.inf), sel, prev_budget)                 # exact on the product's own scores
        thr = np.sort(rs[valid])[::-1][prev_budget - 1]
        oq.check_topk_set(np.where(valid, rs, -np.inf), sel, prev_budget, atol=float(abs(thr)) * 2 ** -6 + 2 ** -6)
        # and where product and oracle agree on every score around the threshold the page sets are identical
        ref_sel = info["selected_pages"][b][:-1]
        differ = np.setxor1d(sel, ref_sel)
        assert (np.abs(rs[differ] - thr) <= abs(thr) * 2 ** -6 + 2 ** -6).all(), "page sets differ away from the threshold"
    assert n_exact / n_total > 0.98, f"only {n_exact}/{n_total} page scores identical to the oracle's bf16 values"


def test_deltakv_observation_chain_full_size_against_pinned_oracle():
    """One observation layer at BASELINE.json configs[4] size: raw logits [1, 28, 262 152] -> `token_scores_full`
    (sparse_controller.py:255-299) -> `dynamic_topk_indices` (:1784-1822, both tie rules) from the pinned oracle."""
    from sparse_vllm_amd.kernels.deltakv_kernels import decode_softmax_token_scores, topk_sorted_desc
    d = torch.device("cuda:0")
    B, H, L, sink, k = 1, 28, 262152, 8, 2048
    gen = torch.Generator(device=d).manual_seed(5)
    raw = torch.randn(B, H, L, device=d, generator=gen) * 6.0                              # peaky rows
    clen = torch.tensor([L - sink - 137], dtype=torch.int32, device=d)
    scale = 128 ** -0.5
    got = decode_softmax_token_scores(raw, candidate_start=sink, candidate_lens=clen, scale=scale, round_dtype=torch.bfloat16)
    torch.cuda.synchronize()
    n = int(clen[0])
    ref = odk.token_scores_full(raw.cpu().numpy(), candidate_start=sink, candidate_lens=[n], scale=scale)
    g = got.float().cpu().numpy()
    fill = float(torch.finfo(torch.bfloat16).min)
    assert (g[0, :sink] == fill).all() and (g[0, sink + n:] == fill).all()
    assert (ref[0, :sink] == np.float32(fill)).all()
    # bf16(p): the two fp32 softmax evaluations differ in their last bits, which can move p across a bf16 rounding boundary
    np.testing.assert_allclose(g[0, sink: sink + n], ref[0, sink: sink + n], rtol=2 ** -7, atol=1e-12)
    same = float((g[0, sink: sink + n] == ref[0, sink: sink + n]).mean())
    assert same > 0.99, same
    for tiebreak in (False, True):
        # the top-k of the PRODUCT's scores against the oracle's top-k of the same scores: isolates the selection
        keys = odk.dynamic_topk_keys(g, sink=sink, compressed_lens=[n], tiebreak=tiebreak)
        ref_idx = odk.dynamic_topk_indices(g, sink=sink, compressed_lens=[n], keep=k, tiebreak=tiebreak)
        search = got[:, sink:]
        if tiebreak:        # the product's key construction (SparseController._update_dynamic_omnikv_indices)
            m = search.size(1)
            pos_key = torch.arange(m, device=d, dtype=torch.float32) / max(1, m)
            base = search.float().masked_fill(torch.arange(m, device=d) >= clen.unsqueeze(1), -1e10)
            search = base + base.abs().clamp_min(1.0) * (pos_key.unsqueeze(0) * 1.0e-6)
        idx = topk_sorted_desc(search, k, valid_len=clen, masked_value=-1e10)[0].cpu().numpy()
        odk.check_sorted_topk(keys[0], idx, ref_idx[0])
        assert (idx >= 0).all() and (idx < n).all()
