"""Chunked prefill of the sparse side on the GPU (H2O score accumulation + intermediate / final
eviction; SnapKV final-chunk selection) against the oracle driven chunk by chunk on the same inputs."""

import numpy as np
import pytest

from oracle import bf16_round
from oracle import h2o as oh
from oracle import prefill_attention as opa
from oracle import prefill_score as ops

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _f(t):
    return t.float().cpu().numpy()


def _mirror_state(cm):
    return oh.SlotState(cm.buffer_req_to_token_slots_tensor.cpu().numpy().copy(),
                        cm.free_slots_stack_tensor.cpu().numpy().copy(),
                        np.asarray(cm._num_free_slots, dtype=np.int64), np.stack(cm.row_seq_lens).astype(np.int32))


def _assert_state(cm, st, rows):
    np.testing.assert_array_equal(np.stack(cm.row_seq_lens), st.row_len)
    np.testing.assert_array_equal(np.asarray(cm._num_free_slots), st.free_ptr)
    tab = cm.buffer_req_to_token_slots_tensor.cpu().numpy()
    stack = cm.free_slots_stack_tensor.cpu().numpy()
    for l in range(tab.shape[0]):
        for r in rows:
            n = int(st.row_len[l, r])
            np.testing.assert_array_equal(tab[l, r, :n], st.slot_table[l, r, :n])
            assert (tab[l, r, n:] == 0).all()
        p = int(st.free_ptr[l])
        np.testing.assert_array_equal(stack[l, :p], st.free_stack[l, :p])


@pytest.mark.parametrize("prompts,mode", [((300, 300), "probability"), ((300, 217), "probability"),
                                          ((260, 260), "logits")])
def test_h2o_chunked_prefill_matches_oracle(prompts, mode):
    from sparse_vllm_amd.config import Config
    from tools.synthetic import SyntheticDecodeDriver as SparseDecodeDriver
    from sparse_vllm_amd.engine.sequence import Sequence
    L, Hq, Hkv, D = 2, 28, 4, 128
    chunk, pre_budget, dec_budget, window = 64, 128, 64, 16
    conf = Config.from_kwargs(sparse_method="h2o", num_hidden_layers=L, max_model_len=512, max_num_seqs_in_gpu=3,
                              num_kvcache_slots=700, h2o_decode_budget=dec_budget, h2o_decode_eviction_interval=64,
                              h2o_prefill_budget=pre_budget, h2o_prefill_score_window=window,
                              engine_prefill_chunk_size=chunk, sparse_prefill_score_mode=mode)
    drv = SparseDecodeDriver(conf)
    cm = drv.cache_manager
    cm.permute_free_slots(2)
    seqs = [Sequence(num_prompt_tokens=n) for n in prompts]
    st = _mirror_state(cm)
    kc = _f(cm.kv_cache[0]).copy()
    vc = _f(cm.kv_cache[1]).copy()
    g = torch.Generator().manual_seed(1)
    rows = None
    step = 0
    while any(s.num_prefilled_tokens < s.num_prompt_tokens for s in seqs):
        active = [s for s in seqs if s.num_prefilled_tokens < s.num_prompt_tokens]
        for s in active:
            s.current_chunk_size = min(chunk, s.num_prompt_tokens - s.num_prefilled_tokens)
        tot = sum(s.current_chunk_size for s in active)
        q = (torch.randn(L, tot, Hq, D, generator=g) * 0.4).to(torch.bfloat16).to(drv.device)
        k = (torch.randn(L, tot, Hkv, D, generator=g) * 0.4).to(torch.bfloat16).to(drv.device)
        v = (torch.randn(L, tot, Hkv, D, generator=g) * 0.4).to(torch.bfloat16).to(drv.device)
        finals = [bool(s.is_last_chunk_prefill) for s in active]
        chunk_lens = [s.current_chunk_size for s in active]
        outs = torch.zeros_like(q)
        drv.prefill_chunk(active, q, k, v, outputs=outs)
        torch.cuda.synchronize()
        got_o = _f(outs)
        rows_all = {s.seq_id: cm.seq_id_to_row[0][s.seq_id] for s in seqs if s.seq_id in cm.seq_id_to_row[0]}
        arows = [rows_all[s.seq_id] for s in active]
        # ---------------- oracle chunk
        qn, kn, vn = _f(q), _f(k), _f(v)
        starts = np.concatenate(([0], np.cumsum(chunk_lens)[:-1])).astype(np.int32)
        for l in range(L):
            ctx, cache = [], []
            for s, r, n in zip(active, arows, chunk_lens):
                prev = int(st.row_len[l, r])
                new = oh.allocate(st, l, r, n)
                sl = slice(int(starts[active.index(s)]), int(starts[active.index(s)]) + n)
                kc[l][new] = kn[l][sl]
                vc[l][new] = vn[l][sl]
                ctx.append(prev + n)
                cache.append(prev)
            ctx = np.array(ctx, np.int32)
            cache = np.array(cache, np.int32)
            # the chunk's causal attention over the (already compressed) physical row: cached prefix + this chunk
            ref_o = opa.context_attention_fwd(qn[l], kc[l], vc[l], np.array(arows, np.int32), starts, ctx, cache, st.slot_table[l])
            np.testing.assert_allclose(got_o[l], bf16_round(ref_o), rtol=2e-2, atol=2e-2, err_msg=f"prefill attention layer {l} step {step}")
            qs = np.maximum(cache, ctx - window).astype(np.int32)
            stepsc = np.empty((len(active), int(ctx.max())), np.float32)
            ops.prefill_score_fwd(qn[l], kc[l], stepsc, np.array(arows, np.int32), starts, ctx, cache,
                                  int((ctx - qs).max()), st.slot_table[l], qs, ctx, score_mode=mode)
            for b, (s, r) in enumerate(zip(active, arows)):
                row_sc = stepsc[b]
                if mode == "logits":
                    row_sc = oh.normalize_logit_prefill_score(row_sc, new_len=int(ctx[b]))
                st.scores[(l, r)] = oh.accumulate_score(st.scores.get((l, r)), row_sc, new_len=int(ctx[b]),
                                                        weight=float(ctx[b] - qs[b]))
        for l in range(L):
            for s, r, fin in zip(active, arows, finals):
                n = int(st.row_len[l, r])
                budget = dec_budget if fin else pre_budget
                if n <= budget:
                    continue
                keep = oh.select_h2o_indices(st.scores[(l, r)], budget=budget, recent_ratio=0.5)
                kept = st.scores[(l, r)][keep]
                if fin:
                    oh.compact_final_prefill_dense_batch(st, l, [r], keep[None, :], budget, kc[l], vc[l])
                else:
                    oh.free_part_slots(st, l, r, keep, keep_sorted=True)
                st.scores[(l, r)] = kept
        # ---------------- compare
        _assert_state(cm, st, list(rows_all.values()))
        sc = cm.h2o_score_tensor.cpu().numpy()
        for l in range(L):
            for r in rows_all.values():
                n = int(st.row_len[l, r])
                if (l, r) in st.scores:
                    np.testing.assert_allclose(sc[l, r, :n], st.scores[(l, r)], rtol=2e-2, atol=2e-3)
                assert (sc[l, r, n:] == 0).all()
        step += 1
    # after the final chunks every row holds exactly the decode budget and the K/V payload of the kept
    # tokens sits in the row's smallest slots (h2o.py:1181-1349): compare the cache bytes
    np.testing.assert_array_equal(_f(cm.kv_cache[0]), kc)
    np.testing.assert_array_equal(_f(cm.kv_cache[1]), vc)
    for l in range(L):
        for r in rows_all.values():
            assert int(st.row_len[l, r]) == dec_budget
            sl = st.slot_table[l, r, :dec_budget]
            assert (np.diff(sl) > 0).all()
    assert cm._h2o_counters["final_prefill_evictions"] == L * len(prompts)


def test_snapkv_final_chunk_selection_and_decode_eviction():
    from sparse_vllm_amd.config import Config
    from tools.synthetic import SyntheticDecodeDriver as SparseDecodeDriver
    from sparse_vllm_amd.engine.sequence import Sequence
    from oracle import decode_attention as oda
    L, Hq, Hkv, D = 2, 28, 4, 128
    sink, recent, keep_top, window = 4, 8, 20, 8
    budget = sink + keep_top + recent
    conf = Config.from_kwargs(sparse_method="snapkv", num_hidden_layers=L, max_model_len=256, max_num_seqs_in_gpu=2,
                              num_kvcache_slots=400, sink_keep_tokens=sink, recent_keep_tokens=recent,
                              decode_keep_tokens=keep_top, snapkv_window_size=window, engine_prefill_chunk_size=96)
    drv = SparseDecodeDriver(conf)
    cm = drv.cache_manager
    cm.permute_free_slots(6)
    seqs = [Sequence(num_prompt_tokens=90), Sequence(num_prompt_tokens=90)]
    for s in seqs:
        s.current_chunk_size = 90
    g = torch.Generator().manual_seed(3)
    mk = lambda n, h: (torch.randn(L, n, h, D, generator=g) * 0.4).to(torch.bfloat16).to(drv.device)
    q, k, v = mk(180, Hq), mk(180, Hkv), mk(180, Hkv)
    st = _mirror_state(cm)
    drv.prefill_chunk(seqs, q, k, v)
    torch.cuda.synchronize()
    rows = [cm.seq_id_to_row[0][s.seq_id] for s in seqs]
    qn, kn = _f(q), _f(k)
    kc = np.zeros_like(_f(cm.kv_cache[0]))
    for l in range(L):
        for b, r in enumerate(rows):
            new = oh.allocate(st, l, r, 90)
            kc[l][new] = kn[l][b * 90:(b + 1) * 90]
    for l in range(L):
        ctx = np.array([90, 90], np.int32)
        sc = np.empty((2, 90), np.float32)
        ops.prefill_score_fwd(qn[l], kc[l], sc, np.array(rows, np.int32), np.array([0, 90], np.int32), ctx,
                              np.zeros(2, np.int32), window, st.slot_table[l], ctx - window, ctx,
                              candidate_start=sink, num_recent_tokens=recent)
        for b, r in enumerate(rows):
            mid = sc[b, sink:90 - recent]
            order = np.argsort(-mid, kind="stable")[:keep_top]
            keep = np.concatenate((np.arange(sink), np.sort(order) + sink, np.arange(90 - recent, 90)))
            got_row = cm.buffer_req_to_token_slots_tensor[l, r, :budget].cpu().numpy()
            exp_row = st.slot_table[l, r, :90][keep]
            if not np.array_equal(got_row, exp_row):
                # only acceptable explanation: a tie (within float noise) at the top-k boundary
                thr = np.sort(mid)[::-1][keep_top - 1]
                diff = np.setxor1d(got_row, exp_row)
                pos = [int(np.nonzero(st.slot_table[l, r, :90] == s_)[0][0]) for s_ in diff]
                assert all(abs(sc[b, p] - thr) < 1e-4 for p in pos), "SnapKV selection differs beyond score noise"
            oh.free_part_slots(st, l, r, keep, keep_sorted=True)
    assert (np.stack(cm.row_seq_lens)[:, rows] == budget).all()
    # decode until the re-eviction trigger (2 x top budget = 40 tokens): rows 32 -> 40 -> 32
    for s in seqs:
        s.num_tokens = 90
    drv.seqs = seqs
    lens_seen = []
    for step in range(10):
        qd, kd, vd = drv.random_step_inputs(seed=50 + step)
        drv.step(qd, kd, vd)
        lens_seen.append(int(cm.row_seq_lens[0][rows[0]]))
    assert max(lens_seen) == 2 * keep_top - 1 or 2 * keep_top in lens_seen or budget in lens_seen
    assert lens_seen[-1] < 2 * keep_top and budget in lens_seen[1:]
