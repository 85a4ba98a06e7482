"""GPU parity tests for the Quest path (page metadata, page scores, top-k view, paged decode)."""

import numpy as np
import pytest

from oracle import bf16_bits_to_f32, bf16_round, f32_to_bf16_bits
from oracle import decode_attention as oda
from oracle import quest as oq

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev():
    return torch.device("cuda:0")


def to_bf16(x):
    return torch.from_numpy(f32_to_bf16_bits(x).view(np.int16).copy()).to(dev()).view(torch.bfloat16)


def _fixture(golden):
    g = golden("quest")
    page, budget, layer, max_ctx, mppr, Hkv, _ = (int(x) for x in g["cfg"])
    k = bf16_bits_to_f32(g["kv_k"])
    meta = bf16_bits_to_f32(g["metadata"])
    q = bf16_bits_to_f32(g["q"])
    return g, page, budget, layer, max_ctx, mppr, Hkv, k, meta, q


def test_page_minmax_golden(golden):
    from sparse_vllm_amd.kernels.quest_ops import page_minmax
    g, page, budget, layer, max_ctx, mppr, Hkv, k, meta, q = _fixture(golden)
    L, slots, H, D = k.shape
    kv = torch.zeros((2, L, slots, H, D), dtype=torch.bfloat16, device=dev())
    kv[0] = to_bf16(k)
    md = torch.zeros((2, L, slots // page, H, D), dtype=torch.bfloat16, device=dev())
    pages = np.concatenate([g["page_table"][r, : n // page] for r, n in enumerate(g["lens"])]).astype(np.int64)
    page_minmax(kv, md, torch.from_numpy(pages).to(dev()), page_size=page)
    got = md.float().cpu().numpy()
    np.testing.assert_array_equal(got[:, :, pages], meta[:, :, pages])
    untouched = np.setdiff1d(np.arange(slots // page), pages)
    assert (got[:, :, untouched] == 0).all()


def test_score_pages_and_view_golden(golden):
    from sparse_vllm_amd.kernels.quest_ops import build_view, score_pages
    g, page, budget, layer, max_ctx, mppr, Hkv, k, meta, q = _fixture(golden)
    lens = g["lens"]
    B = len(lens)
    d = dev()
    max_pages = (max_ctx + page - 1) // page
    n_prev = max_pages - 1
    ptab = torch.from_numpy(g["page_table"].copy()).to(d)
    ttab = torch.from_numpy(g["token_table"].copy()).to(d)
    req = torch.arange(B, dtype=torch.int32, device=d)
    cl = torch.from_numpy(lens.copy()).to(d)
    scores = torch.zeros((B, n_prev), dtype=torch.float32, device=d)
    score_pages(to_bf16(q), to_bf16(meta[0, layer]), to_bf16(meta[1, layer]), ptab, req, cl, scores, page_size=page,
                n_prev=n_prev)
    got = scores.cpu().numpy()
    ref = g["page_scores"]
    num_pages = (lens + page - 1) // page
    valid = np.arange(n_prev)[None, :] < (num_pages - 1)[:, None]
    assert np.isneginf(got[~valid]).all()
    # bf16-valued scores; the MFMA's fp32 summation order may flip a rounding: <= 1 bf16 ulp, rarely
    np.testing.assert_allclose(got[valid], ref[valid], rtol=2 ** -7, atol=1e-6)
    assert (got[valid] != ref[valid]).mean() < 0.05
    # the view kernel on the REFERENCE's scores must make the oracle's deterministic choice exactly
    page_budget_base = max(3, budget // page)
    prev_budget = min(page_budget_base - 1, max_pages - 1)
    for long_text in (True, False):
        keep = (prev_budget + 1) * page if long_text else max(budget, page_budget_base * page, page)
        packed = torch.zeros((B, keep), dtype=torch.int32, device=d)
        ll = torch.zeros((B,), dtype=torch.int32, device=d)
        lr = torch.zeros((B,), dtype=torch.int32, device=d)
        sc_in = torch.from_numpy(np.where(valid, ref, -np.inf).astype(np.float32)).to(d)
        build_view(sc_in, ptab, ttab, req, cl, packed, ll, lr, page_size=page, n_prev=n_prev, prev_budget=prev_budget,
                   token_budget=budget, page_budget_base=page_budget_base, max_keep=keep, is_long_text=long_text)
        exp_packed, exp_req, exp_lens, info = oq.build_decode_view(
            q, meta[0, layer], meta[1, layer], g["token_table"], g["page_table"], np.arange(B, dtype=np.int32), lens,
            page_size=page, token_budget=budget, max_context_len=max_ctx, max_pages_per_row=mppr, num_kv_heads=Hkv,
            is_long_text=long_text)
        np.testing.assert_array_equal(ll.cpu().numpy(), exp_lens)
        np.testing.assert_array_equal(lr.cpu().numpy(), exp_req)
        gp = packed.cpu().numpy()
        for b in range(B):
            if valid[b].sum() < prev_budget and long_text:
                continue      # undefined in the reference too (-inf pages)
            np.testing.assert_array_equal(gp[b, : exp_lens[b]], exp_packed[b, : exp_lens[b]])
        # and the reference's own choice differs only by boundary ties
        tag = "long" if long_text else "mixed"
        np.testing.assert_array_equal(ll.cpu().numpy(), g[f"{tag}_lens"])


@pytest.mark.parametrize("dist", ["normal", "normal_bf16", "uniform_one_binade", "narrow", "all_equal", "two_values",
                                  "with_neg_inf", "tiny_probabilities"])
@pytest.mark.parametrize("shape", [dict(n_prev=8191, prev_budget=291), dict(n_prev=2047, prev_budget=255),
                                   dict(n_prev=20000, prev_budget=1023), dict(n_prev=700, prev_budget=699),
                                   dict(n_prev=8193, prev_budget=291), dict(n_prev=32768, prev_budget=4000)])
def test_build_view_selection_exact_over_score_distributions(dist, shape):
    """The page top-k of `svk_quest_build_view` == the first prev_budget entries of a stable descending argsort, as a
    set written in ascending page order, for score rows that drive the select through each of its routes: a handful
    of candidates in the k-th score's leading-bits bin (counted), hundreds (radix passes over the candidate list),
    more than the list holds (radix passes over all keys), all keys equal (no pass at all)."""
    from sparse_vllm_amd.kernels.quest_ops import build_view
    n_prev, prev_budget = shape["n_prev"], shape["prev_budget"]
    page, B = 16, 3
    rng = np.random.default_rng(n_prev + len(dist))
    if dist == "normal":
        sc = rng.standard_normal((B, n_prev)).astype(np.float32) * 3
    elif dist == "normal_bf16":
        sc = bf16_round(rng.standard_normal((B, n_prev)).astype(np.float32) * 3)
    elif dist == "uniform_one_binade":
        sc = (1.0 + rng.random((B, n_prev))).astype(np.float32)
    elif dist == "narrow":
        sc = (1.0 + rng.random((B, n_prev)) * 2.0 ** -14).astype(np.float32)
    elif dist == "all_equal":
        sc = np.full((B, n_prev), 0.25, dtype=np.float32)
    elif dist == "two_values":
        sc = rng.choice(np.array([1.5, -2.0], dtype=np.float32), (B, n_prev))
    elif dist == "with_neg_inf":
        sc = bf16_round(rng.standard_normal((B, n_prev)).astype(np.float32))
        sc[:, n_prev - n_prev // 3:] = -np.inf
        sc[1, prev_budget // 2:] = -np.inf
    else:
        sc = (rng.random((B, n_prev)) ** 8 * 1e-3).astype(np.float32)
    n_pages = n_prev + 1
    ptab = np.stack([rng.permutation(n_pages * B)[:n_pages] for _ in range(B)]).astype(np.int32)
    d = dev()
    lens = np.full(B, n_pages * page - 5, dtype=np.int32)
    keep = (prev_budget + 1) * page
    packed = torch.zeros((B, keep), dtype=torch.int32, device=d)
    ll = torch.zeros((B,), dtype=torch.int32, device=d)
    lr = torch.zeros((B,), dtype=torch.int32, device=d)
    ttab = torch.zeros((B, 16), dtype=torch.int32, device=d)
    build_view(torch.from_numpy(sc).to(d), torch.from_numpy(ptab).to(d), ttab, torch.arange(B, dtype=torch.int32, device=d),
               torch.from_numpy(lens).to(d), packed, ll, lr, page_size=page, n_prev=n_prev, prev_budget=prev_budget,
               token_budget=prev_budget * page + page, page_budget_base=prev_budget + 1, max_keep=keep, is_long_text=True)
    got = packed.cpu().numpy()
    for b in range(B):
        order = np.sort(np.argsort(-sc[b], kind="stable")[:prev_budget])
        exp_pages = np.concatenate((ptab[b, order], ptab[b, n_pages - 1:n_pages]))
        exp = (exp_pages[:, None].astype(np.int64) * page + np.arange(page)[None, :]).reshape(-1)
        np.testing.assert_array_equal(got[b], exp)
    np.testing.assert_array_equal(ll.cpu().numpy(), prev_budget * page + (lens - n_prev * page))


def test_quest_scores_qwen7b_shape_vs_oracle():
    from sparse_vllm_amd.kernels.quest_ops import page_minmax, score_pages
    rng = np.random.default_rng(4)
    page, Hq, Hkv, D, B = 16, 28, 4, 128, 3
    lens = np.array([4096 + 160, 2000, 333], dtype=np.int32)
    n_pages_tot = int(((lens + page - 1) // page).sum()) + 8
    k = bf16_round((rng.standard_normal((n_pages_tot * page, Hkv, D)) * 0.5).astype(np.float32))
    q = bf16_round((rng.standard_normal((B, Hq, D)) * 0.5).astype(np.float32))
    perm = rng.permutation(n_pages_tot)
    max_pages = int((lens.max() + page - 1) // page)
    ptab = np.full((B, max_pages + 3), -1, dtype=np.int32)
    off = 0
    for b, n in enumerate(lens):
        npg = (n + page - 1) // page
        ptab[b, :npg] = perm[off: off + npg]
        off += npg
    d = dev()
    kv = torch.zeros((2, 1, n_pages_tot * page, Hkv, D), dtype=torch.bfloat16, device=d)
    kv[0, 0] = to_bf16(k)
    md = torch.zeros((2, 1, n_pages_tot, Hkv, D), dtype=torch.bfloat16, device=d)
    full = np.concatenate([ptab[b, : n // page] for b, n in enumerate(lens)]).astype(np.int64)
    page_minmax(kv, md, torch.from_numpy(full).to(d), page_size=page)
    pmax, pmin = oq.page_minmax(k, full, page)
    np.testing.assert_array_equal(md[0, 0].float().cpu().numpy()[full], pmax)
    np.testing.assert_array_equal(md[1, 0].float().cpu().numpy()[full], pmin)
    n_prev = max_pages - 1
    scores = torch.zeros((B, n_prev), dtype=torch.float32, device=d)
    score_pages(to_bf16(q), md[0, 0], md[1, 0], torch.from_numpy(ptab).to(d), torch.arange(B, dtype=torch.int32, device=d),
                torch.from_numpy(lens).to(d), scores, page_size=page, n_prev=n_prev)
    mmax = md[0, 0].float().cpu().numpy()
    mmin = md[1, 0].float().cpu().numpy()
    prev = np.maximum(ptab[:, :n_prev], 0).astype(np.int64)
    ref = oq.score_pages_batched(q, mmax[prev].transpose(0, 2, 1, 3), mmin[prev].transpose(0, 2, 1, 3), Hkv)
    valid = np.arange(n_prev)[None, :] < ((lens + page - 1) // page - 1)[:, None]
    got = scores.cpu().numpy()
    np.testing.assert_allclose(got[valid], ref[valid], rtol=2 ** -7, atol=1e-6)
    assert (got[valid] != ref[valid]).mean() < 0.05
    assert np.isneginf(got[~valid]).all()


def test_quest_decode_steps_match_oracle():
    """Quest through the operator surface: paged allocation, metadata refresh when a page completes,
    query-aware view, unscored decode over the packed table."""
    from sparse_vllm_amd.config import Config
    from tools.synthetic import SyntheticDecodeDriver as SparseDecodeDriver
    from sparse_vllm_amd.utils.context import get_context
    B, L, page = 2, 3, 16
    conf = Config.from_kwargs(sparse_method="quest", num_hidden_layers=L, max_model_len=1024, max_num_seqs_in_gpu=B,
                              num_kvcache_slots=B * 1024 + 64, sink_keep_tokens=16, recent_keep_tokens=16,
                              decode_keep_tokens=96, quest_skip_layers=1)
    assert conf.quest_token_budget == 128
    drv = SparseDecodeDriver(conf)
    cm = drv.cache_manager
    cm.permute_free_pages(5)
    start = 500
    seqs = drv.admit_resident_rows(B, start, seed=9)
    rows = [cm.seq_id_to_row[s.seq_id] for s in seqs]
    outs = torch.zeros((L, B, 28, 128), dtype=torch.bfloat16, device=drv.device)
    budget = 128
    for step in range(20):       # crosses a page boundary at len 512
        q, k, v = drv.random_step_inputs(seed=300 + step)
        import sparse_vllm_amd.utils.context as ctxmod
        drv.step(q, k, v, outputs=outs)
        torch.cuda.synchronize()
        lens = cm.row_seq_lens[rows].astype(np.int32)
        assert (lens == start + step + 1).all()
        ttab = cm.buffer_req_to_token_slots.cpu().numpy()
        ptab = cm.buffer_req_to_page_slots.cpu().numpy()
        # paging invariant: slot = page_slot*16 + offset
        for b, r in enumerate(rows):
            pos = np.arange(lens[b])
            np.testing.assert_array_equal(ttab[r, : lens[b]], ptab[r, pos // page] * page + pos % page)
        kc = cm.kv_cache[0].float().cpu().numpy()
        vc = cm.kv_cache[1].float().cpu().numpy()
        md = cm.metadata_cache.float().cpu().numpy()
        qn = q.float().cpu().numpy()
        for l in range(L):
            if l < conf.quest_skip_layers:
                mid, lse = oda.flash_decode_stage1(qn[l], kc[l], vc[l], ttab, np.array(rows, np.int32), lens,
                                                   int(lens.max()), 64)
                o = oda.flash_decode_stage2(mid, lse, lens, 64)
            else:
                # metadata of complete pages must be exact
                for b, r in enumerate(rows):
                    full = ptab[r, : lens[b] // page].astype(np.int64)
                    # the page completed by THIS step is refreshed after the forward (quest.py:1718-1771)
                    pmax, pmin = oq.page_minmax(kc[l], full, page)
                    np.testing.assert_array_equal(md[0, l][full], pmax)
                    np.testing.assert_array_equal(md[1, l][full], pmin)
                res = oq.build_decode_view(qn[l], md[0, l], md[1, l], ttab, ptab, np.array(rows, np.int32), lens,
                                           page_size=page, token_budget=budget, max_context_len=int(lens.max()),
                                           max_pages_per_row=cm.max_pages_per_row, num_kv_heads=4, is_long_text=False)
                packed, lreq, llens, info = res
                mid, lse = oda.flash_decode_stage1(qn[l], kc[l], vc[l], packed, lreq, llens, int(packed.shape[1]), 64)
                o = oda.flash_decode_stage2(mid, lse, llens, 64)
            np.testing.assert_allclose(outs[l].float().cpu().numpy(), bf16_round(o), rtol=3e-2, atol=3e-2)


@pytest.mark.parametrize("dist", ["normal_bf16", "normal", "all_equal", "with_neg_inf"])
@pytest.mark.parametrize("shape", [dict(n_prev=8191, prev_budget=291), dict(n_prev=2047, prev_budget=255),
                                   dict(n_prev=700, prev_budget=699), dict(n_prev=40000, prev_budget=300)])
def test_page_slot_view_equals_token_slot_view(dist, shape):
    """`SvkQuestBuildViewArgs.emit_page_slots`: the view written as page slots, expanded to page_slot * 16 + offset, is the
    token-slot view - rows that are sparse, sparse with a short last page, and dense (short)."""
    from sparse_vllm_amd.kernels.quest_ops import build_view
    n_prev, kb = shape["n_prev"], shape["prev_budget"]
    page, B = 16, 4
    rng = np.random.default_rng(n_prev * 3 + len(dist))
    if dist == "normal_bf16":
        sc = bf16_round(rng.standard_normal((B, n_prev)).astype(np.float32) * 3)
    elif dist == "normal":
        sc = rng.standard_normal((B, n_prev)).astype(np.float32) * 3
    elif dist == "all_equal":
        sc = np.full((B, n_prev), 0.25, dtype=np.float32)
    else:
        sc = bf16_round(rng.standard_normal((B, n_prev)).astype(np.float32))
        sc[:, n_prev - n_prev // 3:] = -np.inf
        sc[1, max(kb, n_prev // 2):] = -np.inf
    n_pages = n_prev + 1
    ptab = np.stack([rng.permutation(n_pages * B)[:n_pages] for _ in range(B)]).astype(np.int32)
    ttab = (ptab[:, :, None].astype(np.int64) * page + np.arange(page)[None, None, :]).reshape(B, -1).astype(np.int32)
    lens = np.array([n_pages * page, n_pages * page - 9, n_pages * page - 15, min(n_pages, kb) * page - 3], dtype=np.int32)
    token_budget, base = (kb + 1) * page, kb + 1
    keep = (kb + 1) * page
    d = dev()
    t_sc, t_pt, t_tt = torch.from_numpy(sc).to(d), torch.from_numpy(ptab).to(d), torch.from_numpy(ttab).to(d)
    req, t_len = torch.arange(B, dtype=torch.int32, device=d), torch.from_numpy(lens).to(d)

    def run(emit_pages):
        packed = torch.full((B, keep), -7, dtype=torch.int32, device=d)
        ll = torch.zeros((B,), dtype=torch.int32, device=d)
        lr = torch.zeros((B,), dtype=torch.int32, device=d)
        build_view(t_sc, t_pt, t_tt, req, t_len, packed, ll, lr, page_size=page, n_prev=n_prev, prev_budget=kb,
                   token_budget=token_budget, page_budget_base=base, max_keep=keep, is_long_text=False,
                   emit_page_slots=emit_pages)
        torch.cuda.synchronize()
        return packed.cpu().numpy(), ll.cpu().numpy()

    ref, ref_lens = run(False)
    assert ref_lens[3] == lens[3] and (ref_lens[:3] < lens[:3]).all()          # row 3 is dense, the others sparse
    got, got_lens = run(True)
    np.testing.assert_array_equal(got_lens, ref_lens)
    for b in range(B):
        n = int(ref_lens[b])
        n_pg = (n + page - 1) // page
        tok = (got[b, :n_pg, None].astype(np.int64) * page + np.arange(page)[None, :]).reshape(-1)[:n]
        np.testing.assert_array_equal(tok, ref[b, :n])


@pytest.mark.parametrize("cfg", [dict(D=128, Hq=28, Hkv=4, block_seq=256, scored=False),
                                 dict(D=64, Hq=16, Hkv=8, block_seq=128, scored=False),
                                 dict(D=128, Hq=32, Hkv=8, block_seq=256, scored=True)])
def test_stage1_page_slot_addressing_equals_token_slots(cfg):
    """`SvkFlashDecodeStage1Args.slot_page_size`: the launch over a table of page slots == the launch over the token slots
    page_slot * 16 + offset, bit for bit (partials, log-sum-exps, decode scores), ragged lengths."""
    from sparse_vllm_amd.kernels.gqa_flash_decoding_stage1 import _launch
    D, Hq, Hkv, bs = cfg["D"], cfg["Hq"], cfg["Hkv"], cfg["block_seq"]
    page, B, n_pages = 16, 4, 300
    rng = np.random.default_rng(D + Hq)
    d = dev()
    ptab = rng.permutation(B * n_pages + 5)[: B * n_pages].reshape(B, n_pages).astype(np.int32)
    ttab = (ptab[:, :, None].astype(np.int64) * page + np.arange(page)[None, None, :]).reshape(B, -1).astype(np.int32)
    lens = np.array([n_pages * page, n_pages * page - 13, 17, 1], dtype=np.int32)
    slots = (B * n_pages + 5) * page
    kc = torch.randn((slots, Hkv, D), device=d).to(torch.bfloat16)
    vc = torch.randn((slots, Hkv, D), device=d).to(torch.bfloat16)
    q = torch.randn((B, Hq, D), device=d).to(torch.bfloat16)
    req = torch.tensor([2, 0, 3, 1], dtype=torch.int32, device=d)
    t_len = torch.from_numpy(lens).to(d)
    L = n_pages * page
    nblk = (L + bs - 1) // bs
    outs = []
    for tab, sps in ((ttab, 0), (ptab, page)):
        rows = np.zeros_like(tab)
        rows[req.cpu().numpy()] = tab
        mid = torch.full((B, Hq, nblk, D), 3.0, dtype=torch.float32, device=d)
        lse = torch.full((B, Hq, nblk), 3.0, dtype=torch.float32, device=d)
        score = torch.full((B, L), -1e20, dtype=torch.float32, device=d) if cfg["scored"] else None
        _launch(q, kc, vc, torch.from_numpy(rows).to(d), req, t_len, L, mid, lse, score, bs, None, slot_page_size=sps)
        torch.cuda.synchronize()
        outs.append((mid.view(torch.int32).cpu().numpy(), lse.view(torch.int32).cpu().numpy(),
                     None if score is None else score.view(torch.int32).cpu().numpy()))
    np.testing.assert_array_equal(outs[1][0], outs[0][0])
    np.testing.assert_array_equal(outs[1][1], outs[0][1])
    if cfg["scored"]:
        np.testing.assert_array_equal(outs[1][2], outs[0][2])


def _run_quest(device_state: bool, graph: bool, steps: int, *, ragged: bool = False, sync_debug: bool = False,
               reference_shaped: bool = False):
    from sparse_vllm_amd.config import Config
    from tools.synthetic import SyntheticDecodeDriver as SparseDecodeDriver
    B, L = 4, 3
    conf = Config.from_kwargs(sparse_method="quest", num_hidden_layers=L, max_model_len=1024, max_num_seqs_in_gpu=B,
                              num_kvcache_slots=B * 1024 + 160, sink_keep_tokens=16, recent_keep_tokens=16,
                              decode_keep_tokens=96, quest_skip_layers=1)
    drv = SparseDecodeDriver(conf)
    cm = drv.cache_manager
    cm._device_step_enabled = device_state
    if reference_shaped:
        cm.page_slot_view = False
    cm.permute_free_pages(5)
    drv.admit_resident_rows(B, 500, seed=9)
    if ragged:
        # rows of different phase inside their pages: new pages / completed pages hit subsets of the batch
        for i, s in enumerate(drv.seqs):
            if i:
                cm._allocate(s.seq_id, 3 * i + 1)
    if graph:
        drv.enable_decode_graph()
    q, k, v = drv.random_step_inputs(seed=3)
    o = torch.zeros((L, B, 28, 128), dtype=torch.bfloat16, device=drv.device)
    used_device = 0
    for i in range(steps):
        if sync_debug and i >= 4:
            torch.cuda.set_sync_debug_mode("error")
        try:
            drv.step(q, k, v, outputs=o)
        finally:
            torch.cuda.set_sync_debug_mode("default")
        used_device += int(cm._dev_step_cache is not None and not cm._dev_state_dirty)
    torch.cuda.synchronize()
    return dict(o=o.view(torch.int16).cpu().numpy().copy(), ttab=cm.buffer_req_to_token_slots.cpu().numpy().copy(),
                ptab=cm.buffer_req_to_page_slots.cpu().numpy().copy(), ptab_cpu=cm.buffer_req_to_page_slots_cpu.copy(),
                md=cm.metadata_cache.view(torch.int16).cpu().numpy().copy(), lens=cm.row_seq_lens.copy(),
                nfree=int(cm._num_free_pages), stack=cm.free_pages_cpu_stack.copy(),
                dev_lens=cm._dev_row_len.cpu().numpy().copy(), dev_stack=cm._dev_free_pages.cpu().numpy().copy(),
                dev_ptr=int(cm._dev_free_page_ptr.item()), used_device=used_device)


@pytest.mark.parametrize("ragged", [False, True])
def test_quest_device_resident_steps_equal_host_driven_steps(ragged):
    """SURVEY 8(f).2 for Quest: row lengths and the page stack on the device, the allocation
    (`svk_quest_device_step_begin`) and the predicated min / max refresh of completed pages (`svk_quest_device_step_end`) as
    launches of the step.  Against the host-driven steps (host page pops + uploads, host scan for completed pages) over
    >= 3 page boundaries: token / page tables, metadata, lengths and outputs bit-identical, eager and under hipGraph
    replay; the device copies equal the host mirrors; rows that cross page boundaries at different steps included."""
    steps = 3 * 16 + 7
    ref = _run_quest(False, False, steps, ragged=ragged)
    assert ref["used_device"] == 0
    for graph in (False, True):
        got = _run_quest(True, graph, steps, ragged=ragged)
        assert got["used_device"] >= steps - 2
        for key in ("o", "ttab", "ptab", "md", "lens", "ptab_cpu"):
            np.testing.assert_array_equal(got[key], ref[key], err_msg=f"{key} graph={graph}")
        assert got["nfree"] == ref["nfree"]
        np.testing.assert_array_equal(got["stack"][: got["nfree"]], ref["stack"][: ref["nfree"]])
        np.testing.assert_array_equal(got["dev_lens"], got["lens"])
        assert got["dev_ptr"] == got["nfree"]
        np.testing.assert_array_equal(got["dev_stack"][: got["nfree"]], got["stack"][: got["nfree"]])
        np.testing.assert_array_equal(got["ptab"], got["ptab_cpu"])


@pytest.mark.parametrize("graph", [False, True])
def test_quest_steps_with_page_slot_view_equal_reference_shaped_steps(graph):
    """QuestCacheManager.page_slot_view (view written as page slots, attention addressed by page): the same outputs, tables
    and metadata as with the reference-shaped token-slot view, eager and under replay."""
    steps = 2 * 16 + 5
    ref = _run_quest(True, graph, steps, ragged=True, reference_shaped=True)
    got = _run_quest(True, graph, steps, ragged=True)
    for key in ("o", "ttab", "ptab", "md", "lens"):
        np.testing.assert_array_equal(got[key], ref[key], err_msg=key)


def test_quest_device_resident_step_needs_no_host_sync():
    """Three page boundaries under hipGraph replay with torch's sync debug mode "error": no host <-> device
    synchronisation, no upload."""
    got = _run_quest(True, True, 3 * 16 + 8, sync_debug=True)
    assert got["used_device"] >= 3 * 16
