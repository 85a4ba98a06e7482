"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the
C ABI (ctypes -> libsvk.so), against (1) the committed golden vectors produced by the
reference itself and (2) the numpy oracle on seeded inputs.

Tolerances
  * raw token scores (fp32 accumulate of exact bf16 products): atol 1e-4 + rtol 1e-5
    (only the summation order differs from the oracle / the reference's tl.dot);
  * attention partials/outputs: rtol = atol = 2e-2, the reference's own bar for these
    kernels (tests/test_prefill_score_kernel.py:282-284 of the reference); observed error is
    ~1e-3 (bf16 P rounding + bf16 output);
  * every integer result (indices, slot tables, free stacks, lengths): bit-exact.
"""

import numpy as np
import pytest

from oracle import bf16_bits_to_f32, bf16_round, f32_to_bf16_bits
from oracle import decode_attention as oda
from oracle import h2o as oh

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

SCORE_ATOL, SCORE_RTOL = 1e-4, 1e-5
ATTN_TOL = 2e-2


def dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda:0")


def to_bf16(x_f32: np.ndarray):
    bits = f32_to_bf16_bits(x_f32).view(np.int16)
    return torch.from_numpy(bits.copy()).to(dev()).view(torch.bfloat16)


def run_decode(q, k, v, req, bidx, blen, max_len, block_seq, mode, score_init=-1e20):
    from sparse_vllm_amd.kernels import flash_decode_stage1, flash_decode_stage1_with_score, flash_decode_stage2
    B, Hq, D = q.shape
    nblk = (max_len + block_seq - 1) // block_seq
    tq, tk, tv = to_bf16(q), to_bf16(k), to_bf16(v)
    treq = torch.from_numpy(req).to(dev())
    tb = torch.from_numpy(bidx).to(dev())
    tl = torch.from_numpy(blen).to(dev())
    mid = torch.full((B, Hq, nblk, D), 7.0, dtype=torch.float32, device=dev())
    lse = torch.full((B, Hq, nblk), 7.0, dtype=torch.float32, device=dev())
    score = None
    if mode == 2:
        score = torch.full((B, max_len), score_init, dtype=torch.float32, device=dev())
        flash_decode_stage1_with_score(tq, tk, tv, treq, tb, tl, max_len, mid, lse, score, block_seq)
    elif mode == 3:
        score = torch.full((B, Hq, max_len), score_init, dtype=torch.float32, device=dev())
        flash_decode_stage1_with_score(tq, tk, tv, treq, tb, tl, max_len, mid, lse, score, block_seq)
    else:
        flash_decode_stage1(tq, tk, tv, treq, tb, tl, max_len, mid, lse, block_seq)
    o = torch.empty_like(tq)
    flash_decode_stage2(mid, lse, tl, o, block_seq)
    torch.cuda.synchronize()
    return (mid.cpu().numpy(), lse.cpu().numpy(), None if score is None else score.cpu().numpy(),
            o.float().cpu().numpy())


@pytest.mark.parametrize("case", ["a", "b", "c", "d"])
def test_decode_vs_reference_golden(golden, case):
    g = golden("decode_attention")
    q, k, v = (bf16_bits_to_f32(g[f"{case}_{n}"]) for n in "qkv")
    max_len, block_seq, mode = (int(x) for x in g[f"{case}_meta"])
    mid, lse, score, o = run_decode(q, k, v, g[f"{case}_req"], g[f"{case}_bidx"], g[f"{case}_blen"], max_len,
                                    block_seq, mode)
    np.testing.assert_allclose(lse, g[f"{case}_mid_lse"], rtol=ATTN_TOL, atol=ATTN_TOL)
    np.testing.assert_allclose(mid, g[f"{case}_mid_o"], rtol=ATTN_TOL, atol=ATTN_TOL)
    np.testing.assert_allclose(o, g[f"{case}_o"], rtol=ATTN_TOL, atol=ATTN_TOL)
    if score is not None:
        np.testing.assert_allclose(score, g[f"{case}_score"], rtol=SCORE_RTOL, atol=SCORE_ATOL)


def _rand_case(seed, B, Hq, Hkv, D, lens, block_seq, cap=None, scale=0.3):
    rng = np.random.default_rng(seed)
    lens = np.asarray(lens, dtype=np.int32)
    cap = cap or int(max(lens) + 7)
    slots = int(lens.sum() + 64)
    q = bf16_round((rng.standard_normal((B, Hq, D)) * scale).astype(np.float32))
    k = bf16_round((rng.standard_normal((slots, Hkv, D)) * scale).astype(np.float32))
    v = bf16_round((rng.standard_normal((slots, Hkv, D)) * scale).astype(np.float32))
    perm = rng.permutation(slots).astype(np.int32)
    rows = rng.permutation(B + 3)[:B].astype(np.int32)
    req = np.zeros((B + 3, cap), dtype=np.int32)
    off = 0
    for b in range(B):
        req[rows[b], : lens[b]] = perm[off: off + lens[b]]
        off += lens[b]
    return q, k, v, req, rows, lens


@pytest.mark.parametrize("mode", [0, 2, 3])
@pytest.mark.parametrize("shape", [
    (3, 28, 4, 128, (4224, 4097, 1), 256),     # Qwen2.5-7B heads, H2O row lengths, 1-token row
    (2, 28, 4, 128, (1000, 31), 128),
    (2, 14, 2, 64, (777, 300), 64),             # Qwen2.5-0.5B heads
    (2, 32, 8, 128, (513, 512), 512),           # G=4, 8 kv heads
    (1, 8, 8, 64, (100,), 48),                  # MHA-as-GQA G=1, block_seq not multiple of 32
    (3, 7, 1, 128, (2113, 640, 5), 272),        # one TP=4 rank of Qwen2.5-7B: 1 KV head, G=7
    (2, 14, 2, 128, (1500, 1499), 512),         # TP=2 rank: 2 KV heads
    (2, 8, 1, 128, (300, 257), 64),             # Llama-3.1-8B at TP=8: 1 KV head, G=8
    (2, 10, 2, 128, (95, 400), 96),             # G=5
    (2, 12, 4, 128, (333, 64), 176),            # G=3
])
def test_decode_vs_oracle(shape, mode):
    B, Hq, Hkv, D, lens, block_seq = shape
    q, k, v, req, rows, lens = _rand_case(1234 + B + Hq + mode, B, Hq, Hkv, D, lens, block_seq)
    max_len = int(max(lens))
    score_ref = None
    if mode == 2:
        score_ref = np.full((B, max_len), -1e20, dtype=np.float32)
    elif mode == 3:
        score_ref = np.full((B, Hq, max_len), -1e20, dtype=np.float32)
    mid_ref, lse_ref = oda.flash_decode_stage1(q, k, v, req, rows, lens, max_len, block_seq, attn_score=score_ref)
    o_ref = oda.flash_decode_stage2(mid_ref, lse_ref, lens, block_seq)
    mid, lse, score, o = run_decode(q, k, v, req, rows, lens, max_len, block_seq, mode)
    np.testing.assert_allclose(lse, lse_ref, rtol=ATTN_TOL, atol=ATTN_TOL)
    np.testing.assert_allclose(mid, mid_ref, rtol=ATTN_TOL, atol=ATTN_TOL)
    np.testing.assert_allclose(o, bf16_round(o_ref), rtol=ATTN_TOL, atol=ATTN_TOL)
    if score is not None:
        np.testing.assert_allclose(score, score_ref, rtol=SCORE_RTOL, atol=SCORE_ATOL)
    # tighter, informational bound on the real error
    assert np.abs(o - o_ref).max() < 5e-3


def test_decode_score_combines_with_existing_buffer_like_atomic_max():
    B, Hq, Hkv, D, lens, block_seq = 2, 28, 4, 128, (300, 200), 64
    q, k, v, req, rows, lens = _rand_case(5, B, Hq, Hkv, D, lens, block_seq)
    ref = np.full((B, 300), 1.0, dtype=np.float32)        # pre-existing content above most logits
    oda.flash_decode_stage1(q, k, v, req, rows, lens, 300, block_seq, attn_score=ref)
    _, _, score, _ = run_decode(q, k, v, req, rows, lens, 300, block_seq, 2, score_init=1.0)
    np.testing.assert_allclose(score, ref, rtol=SCORE_RTOL, atol=SCORE_ATOL)
    assert (score[1, 200:] == 1.0).all()


def test_decode_rejects_bad_layout():
    from sparse_vllm_amd.kernels import flash_decode_stage1
    d = dev()
    q = torch.zeros(1, 28, 128, dtype=torch.bfloat16, device=d)
    kc = torch.zeros(8, 4, 128, dtype=torch.bfloat16, device=d)
    req = torch.zeros(1, 8, dtype=torch.int32, device=d)
    one = torch.ones(1, dtype=torch.int32, device=d)
    mid = torch.zeros(1, 28, 1, 128, device=d)
    lse = torch.zeros(1, 28, 1, device=d)
    with pytest.raises(AssertionError):
        flash_decode_stage1(q, kc, kc, req, one, one, 8, mid, lse, 24)       # BLOCK_SEQ % BLOCK_N
    with pytest.raises(AssertionError):
        flash_decode_stage1(q, kc, kc[:, :, ::2], req, one, one, 8, mid, lse, 32)


# ------------------------------------------------------------------ H2O score update
def test_h2o_score_normalise_golden(golden):
    from sparse_vllm_amd.kernels.h2o_ops import h2o_decode_score_update
    g = golden("h2o_scores")
    x = torch.from_numpy(g["norm_raw"].copy()).to(dev())
    h2o_decode_score_update(x, float(int(g["norm_meta"][0])) ** -0.5)
    out = x.cpu().numpy()
    np.testing.assert_allclose(out, g["norm_out"], rtol=1e-5, atol=1e-10)
    assert (out[2, 7:] == 0).all()


def test_h2o_score_update_accumulates_like_reference():
    from sparse_vllm_amd.kernels.h2o_ops import h2o_decode_score_update
    rng = np.random.default_rng(3)
    B, W, rows, cap = 5, 4224, 9, 4300
    lens = np.array([4224, 4100, 4097, 1, 2000], dtype=np.int32)
    raw = np.full((B, W), -1e20, dtype=np.float32)
    for b, L in enumerate(lens):
        raw[b, :L] = rng.standard_normal(L).astype(np.float32) * 5
    ridx = rng.permutation(rows)[:B].astype(np.int32)
    cum = rng.random((rows, cap)).astype(np.float32)
    for b, L in enumerate(lens):
        cum[ridx[b], L - 1:] = 123.0          # stale garbage where the new token lands must not matter
    norm_ref = oda.h2o_normalize_decode_scores(raw, 128)
    cum_ref = cum.copy()
    for b, L in enumerate(lens):
        cum_ref[ridx[b], :L] = oh.update_decode_scores(cum[ridx[b], : L - 1], norm_ref[b], int(L))
    x = torch.from_numpy(raw.copy()).to(dev())
    tc = torch.from_numpy(cum.copy()).to(dev())
    h2o_decode_score_update(x, 128 ** -0.5, cum_score=tc, b_req_idx=torch.from_numpy(ridx).to(dev()),
                            b_seqlen=torch.from_numpy(lens).to(dev()))
    np.testing.assert_allclose(x.cpu().numpy(), norm_ref, rtol=1e-5, atol=1e-10)
    np.testing.assert_allclose(tc.cpu().numpy(), cum_ref, rtol=1e-5, atol=1e-9)
    # untouched beyond each row's length
    got = tc.cpu().numpy()
    for b, L in enumerate(lens):
        assert (got[ridx[b], L:] == 123.0).all()


# ------------------------------------------------------------------ H2O selection (bit-exact)
@pytest.mark.parametrize("case", ["u", "ties", "short", "ratio", "zeros", "softmaxlike", "allrecent"])
def test_select_golden(golden, case):
    from sparse_vllm_amd.kernels.h2o_ops import select_h2o_indices_batch
    g = golden("h2o_select")
    budget, ratio = g[f"{case}_cfg"]
    s = torch.from_numpy(g[f"{case}_scores"].copy()).to(dev())
    keep = select_h2o_indices_batch(s, budget=int(budget), recent_ratio=float(ratio))
    np.testing.assert_array_equal(keep.cpu().numpy(), g[f"{case}_keep"])


def test_select_known_answer_and_errors():
    from sparse_vllm_amd.kernels.h2o_ops import select_h2o_indices_batch
    s = torch.tensor([[1, 9, 2, 8, 3, 0, 0, 0]], dtype=torch.float32, device=dev())
    assert select_h2o_indices_batch(s, budget=4, recent_ratio=0.5).cpu().tolist() == [[1, 3, 6, 7]]
    with pytest.raises(ValueError):
        select_h2o_indices_batch(s[0], budget=4, recent_ratio=0.5)
    with pytest.raises(ValueError):
        select_h2o_indices_batch(s, budget=0, recent_ratio=0.5)
    with pytest.raises(ValueError):
        select_h2o_indices_batch(s, budget=4, recent_ratio=1.0)


@pytest.mark.parametrize("cfg", [
    (224, 4224, 4096, 0.5, "uniform"),      # decode burst at BASELINE size: 28 layers x 8 seqs
    (56, 16384, 8192, 0.5, "uniform"),      # intermediate prefill eviction
    (28, 16384, 4096, 0.5, "ties"),         # final prefill with massive ties
    (16, 4224, 4096, 0.5, "adversarial"),   # every candidate equal / negative zeros / denormals
    (8, 5000, 4096, 0.03, "ties"),
])
def test_select_vs_oracle_full_size(cfg):
    from sparse_vllm_amd.kernels.h2o_ops import select_h2o_indices_batch
    rows, kv_len, budget, ratio, kind = cfg
    rng = np.random.default_rng(rows + kv_len)
    if kind == "uniform":
        s = rng.random((rows, kv_len)).astype(np.float32)
    elif kind == "ties":
        s = (rng.integers(0, 50, (rows, kv_len)) / 64.0).astype(np.float32)
    else:
        s = np.zeros((rows, kv_len), dtype=np.float32)
        s[1] = -0.0
        s[2, ::2] = -0.0
        s[3] = 1e-42                       # denormal
        s[4] = rng.choice(np.array([0.0, -0.0, 1e-45, -1e-45, 1.0, -1.0], dtype=np.float32), kv_len)
        s[5:] = rng.standard_normal((rows - 5, kv_len)).astype(np.float32).round(1)
    ref = oh.select_h2o_indices_batch(s, budget=budget, recent_ratio=ratio)
    got = select_h2o_indices_batch(torch.from_numpy(s).to(dev()), budget=budget, recent_ratio=ratio).cpu().numpy()
    np.testing.assert_array_equal(got, ref)
    # size-independent properties
    assert (np.diff(got, axis=1) > 0).all()
    rc = max(1, int(budget * ratio))
    assert (got[:, -rc:] == np.arange(kv_len - rc, kv_len)).all()


# ------------------------------------------------------------------ compaction (bit-exact)
def _run_compact(st: oh.SlotState, layers, rows_per_layer, keep, cur_len, payload=None):
    from sparse_vllm_amd.kernels.h2o_ops import compact_rows
    d = dev()
    tab = torch.from_numpy(st.slot_table.copy()).to(d)
    stack = torch.from_numpy(st.free_stack.copy()).to(d)
    pay = None if payload is None else torch.from_numpy(payload.copy()).to(d)
    compact_rows(tab, stack, torch.from_numpy(np.ascontiguousarray(keep)).to(d),
                 torch.tensor(layers, dtype=torch.int32, device=d),
                 torch.from_numpy(np.ascontiguousarray(rows_per_layer, dtype=np.int32)).to(d),
                 torch.from_numpy(st.free_ptr[layers].astype(np.int64)).to(d), cur_len=cur_len, row_payload=pay)
    torch.cuda.synchronize()
    return tab.cpu().numpy(), stack.cpu().numpy(), None if pay is None else pay.cpu().numpy()


def test_compact_rows_golden(golden):
    g = golden("compaction")
    st = oh.SlotState(g["bl_before_slot_table"], g["bl_before_free_stack"], g["bl_before_free_ptr"],
                      g["bl_before_row_len"])
    keep = np.sort(g["bl_keep"], axis=2)         # reference sorts when keep_indices_sorted=False
    L, B, K = keep.shape
    tab, stack, _ = _run_compact(st, list(range(L)), np.tile(np.arange(B), (L, 1)), keep, 40)
    np.testing.assert_array_equal(tab, g["bl_after_slot_table"])
    for l in range(L):
        p = int(g["bl_after_free_ptr"][l])
        np.testing.assert_array_equal(stack[l, :p], g["bl_after_free_stack"][l, :p])


def test_compact_streamingllm_golden(golden):
    g = golden("compaction")
    st = oh.SlotState(g["sr_before_slot_table"], g["sr_before_free_stack"], g["sr_before_free_ptr"],
                      g["sr_before_row_len"])
    kv_len, sink, recent = (int(x) for x in g["sr_cfg"])
    keep1 = np.concatenate((np.arange(sink), np.arange(kv_len - recent, kv_len))).astype(np.int64)
    keep = np.broadcast_to(keep1, (2, 3, keep1.size)).copy()
    tab, stack, _ = _run_compact(st, [0, 1], np.tile(np.arange(3), (2, 1)), keep, kv_len)
    np.testing.assert_array_equal(tab, g["sr_after_slot_table"])
    for l in range(2):
        p = int(g["sr_after_free_ptr"][l])
        np.testing.assert_array_equal(stack[l, :p], g["sr_after_free_stack"][l, :p])


def test_compact_rows_vs_oracle_h2o_burst_size():
    rng = np.random.default_rng(77)
    L, rows, cap, B = 28, 12, 4352, 8
    cur, budget = 4224, 4096
    nslots = rows * cap
    st = oh.make_slot_state(L, rows, cap, nslots, permute_seed=1)
    lane_rows = np.stack([rng.permutation(rows)[:B] for _ in range(L)])
    for l in range(L):
        for r in range(rows):
            oh.allocate(st, l, r, cur if r in lane_rows[l] else 100)
    scores = rng.random((L, rows, cap)).astype(np.float32)
    sel = np.stack([oh.select_h2o_indices_batch(scores[l, lane_rows[l], :cur], budget=budget, recent_ratio=0.5)
                    for l in range(L)])
    ref = st.copy()
    pay_ref = scores.copy()
    for l in range(L):
        oh.free_part_slots_batch_layers(ref, [l], list(lane_rows[l]), sel[l:l + 1], keep_sorted=True)
        for j, r in enumerate(lane_rows[l]):
            pay_ref[l, r, :budget] = scores[l, r, sel[l, j]]
            pay_ref[l, r, budget:cur] = 0
    tab, stack, pay = _run_compact(st, list(range(L)), lane_rows, sel, cur, payload=scores)
    np.testing.assert_array_equal(tab, ref.slot_table)
    np.testing.assert_array_equal(pay, pay_ref)
    for l in range(L):
        p = int(ref.free_ptr[l])
        np.testing.assert_array_equal(stack[l, :p], ref.free_stack[l, :p])
    # conservation: every slot is either in a row or in the free stack exactly once
    for l in range(L):
        used = np.concatenate([tab[l, r, : ref.row_len[l, r]] for r in range(rows)])
        allv = np.concatenate([used, stack[l, : int(ref.free_ptr[l])]])
        assert np.array_equal(np.sort(allv), np.arange(nslots))


# ------------------------------------------------------------------ KV payload
def test_store_kvcache_and_copy_slots_exact():
    from sparse_vllm_amd.kernels import store_kvcache
    from sparse_vllm_amd.kernels.h2o_ops import copy_slots
    d = dev()
    g = torch.Generator(device="cpu").manual_seed(0)
    n, H, D, slots = 300, 4, 128, 1000
    key = torch.randn(n, H, D, generator=g).bfloat16().to(d)
    val = torch.randn(n, H, D, generator=g).bfloat16().to(d)
    kc = torch.zeros(slots, H, D, dtype=torch.bfloat16, device=d)
    vc = torch.zeros_like(kc)
    sm = torch.randperm(slots, generator=g)[:n].to(torch.int32)
    sm[::7] = -1
    store_kvcache(key, val, kc, vc, sm.to(d))
    ek = torch.zeros_like(kc)
    ev = torch.zeros_like(vc)
    m = (sm != -1)
    ek[sm[m].long().to(d)] = key[m.to(d)]
    ev[sm[m].long().to(d)] = val[m.to(d)]
    assert torch.equal(kc, ek) and torch.equal(vc, ev)
    # strided source rows (q/k/v split of a fused projection) must work too
    fused = torch.randn(n, 3 * H * D, generator=g).bfloat16().to(d)
    k2 = fused[:, H * D: 2 * H * D].view(n, H, D)
    v2 = fused[:, 2 * H * D:].view(n, H, D)
    store_kvcache(k2, v2, kc, vc, sm.to(d))
    ek[sm[m].long().to(d)] = k2[m.to(d)]
    assert torch.equal(kc, ek)
    # overlapping slot move == gather-then-scatter
    src = torch.randperm(slots, generator=g)[:128].long()
    dst = torch.sort(src).values
    ws = torch.empty(2, 128, H, D, dtype=torch.bfloat16, device=d)
    ek2, ev2 = kc.clone(), vc.clone()
    ek2[dst.to(d)] = kc[src.to(d)]
    ev2[dst.to(d)] = vc[src.to(d)]
    copy_slots(kc, vc, src.to(d), dst.to(d), ws)
    assert torch.equal(kc, ek2) and torch.equal(vc, ev2)


def test_decode_alloc_slots_matches_oracle():
    from sparse_vllm_amd.kernels.h2o_ops import decode_alloc_slots
    d = dev()
    L, rows, cap, nslots, B, GB = 4, 6, 64, 512, 3, 5
    st = oh.make_slot_state(L, rows, cap, nslots, permute_seed=4)
    lens = {0: 10, 2: 33, 5: 7}
    for l in range(L):
        for r, n in lens.items():
            oh.allocate(st, l, r, n)
    lane_rows = [5, 0, 2]
    tab = torch.from_numpy(st.slot_table.copy()).to(d)
    stack = torch.from_numpy(st.free_stack.copy()).to(d)
    sm = torch.zeros(L, GB, dtype=torch.int32, device=d)
    cl = torch.zeros_like(sm)
    ri = torch.zeros_like(sm)
    free_ptr = int(st.free_ptr[0])
    decode_alloc_slots(tab, stack, torch.arange(L, dtype=torch.int32, device=d),
                       torch.tensor(lane_rows, dtype=torch.int32, device=d),
                       torch.tensor([lens[r] for r in lane_rows], dtype=torch.int32, device=d),
                       sm, cl, ri, free_ptr=free_ptr, batch=B)
    new = oh.decode_allocate_batch_layers(st, range(L), lane_rows)
    np.testing.assert_array_equal(tab.cpu().numpy(), st.slot_table)
    np.testing.assert_array_equal(sm.cpu().numpy()[:, :B], new)
    assert (sm.cpu().numpy()[:, B:] == -1).all()
    np.testing.assert_array_equal(cl.cpu().numpy()[0], [8, 11, 34, 8, 8])
    np.testing.assert_array_equal(ri.cpu().numpy()[0], [5, 0, 2, 5, 5])


@pytest.mark.parametrize("mode", [0, 2, 3])
@pytest.mark.parametrize("shape", [
    dict(B=3, Hq=28, Hkv=4, D=128, lens=[4224, 4100, 17], block_seq=1056),
    dict(B=5, Hq=14, Hkv=2, D=64, lens=[1, 16, 33, 256, 300], block_seq=64),
    dict(B=2, Hq=8, Hkv=8, D=128, lens=[512, 511], block_seq=256),
])
def test_decode_with_fused_store_equals_store_then_decode(shape, mode):
    """`new_kv=(k, v, slot_mapping)`: the K/V rows of the newest token are written by the stage-1 launch itself.
    Cache contents, partials, lse and scores must be identical to svk_store_kvcache followed by the plain launch;
    a lane with slot -1 (padded graph lane) stores nothing."""
    import os
    from sparse_vllm_amd.kernels import flash_decode_stage1, flash_decode_stage1_with_score, store_kvcache
    B, Hq, Hkv, D, lens, block_seq = (shape[k] for k in ("B", "Hq", "Hkv", "D", "lens", "block_seq"))
    q, k, v, req, bidx, blen = _rand_case(31 + mode, B, Hq, Hkv, D, lens, block_seq)
    max_len = int(max(lens))
    rng = np.random.default_rng(5)
    nk = bf16_round((rng.standard_normal((B, Hkv, D)) * 0.3).astype(np.float32))
    nv = bf16_round((rng.standard_normal((B, Hkv, D)) * 0.3).astype(np.float32))
    slot_mapping = np.array([req[bidx[b], blen[b] - 1] for b in range(B)], dtype=np.int32)
    slot_mapping[B - 1] = -1                                    # padded lane: nothing stored
    nblk = (max_len + block_seq - 1) // block_seq
    tq, tnk, tnv = to_bf16(q), to_bf16(nk), to_bf16(nv)
    treq, tb, tl = (torch.from_numpy(x).to(dev()) for x in (req, bidx, blen))
    tsm = torch.from_numpy(slot_mapping).to(dev())

    def run(fused):
        tk, tv = to_bf16(k), to_bf16(v)
        mid = torch.full((B, Hq, nblk, D), 7.0, dtype=torch.float32, device=dev())
        lse = torch.full((B, Hq, nblk), 7.0, dtype=torch.float32, device=dev())
        score = None
        if mode == 2:
            score = torch.full((B, max_len), -1e20, dtype=torch.float32, device=dev())
        elif mode == 3:
            score = torch.full((B, Hq, max_len), -1e20, dtype=torch.float32, device=dev())
        new_kv = (tnk, tnv, tsm) if fused else None
        if not fused:
            store_kvcache(tnk, tnv, tk, tv, tsm)
        if score is not None:
            flash_decode_stage1_with_score(tq, tk, tv, treq, tb, tl, max_len, mid, lse, score, block_seq, new_kv=new_kv)
        else:
            flash_decode_stage1(tq, tk, tv, treq, tb, tl, max_len, mid, lse, block_seq, new_kv=new_kv)
        torch.cuda.synchronize()
        out = [tk.view(torch.int16).cpu().numpy(), tv.view(torch.int16).cpu().numpy(), mid.cpu().numpy(), lse.cpu().numpy()]
        if score is not None:
            out.append(score.cpu().numpy())
        return out

    ref, got = run(False), run(True)
    for x, y in zip(ref[:2], got[:2]):
        np.testing.assert_array_equal(x, y)                     # cache bytes
    # the padded lane reads a row another lane may be writing: compare the real lanes only
    for x, y in zip(ref[2:], got[2:]):
        np.testing.assert_array_equal(x[:B - 1], y[:B - 1])


@pytest.mark.parametrize("mode", [0, 2, 3])
@pytest.mark.parametrize("shape", [
    dict(B=3, Hq=28, Hkv=4, D=128, lens=[4224, 4100, 17], block_seq=4224),
    dict(B=5, Hq=14, Hkv=2, D=64, lens=[1, 16, 33, 256, 300], block_seq=304),
    dict(B=2, Hq=8, Hkv=8, D=128, lens=[512, 511], block_seq=512),
    dict(B=4, Hq=7, Hkv=1, D=128, lens=[700, 0, 64, 1], block_seq=704),
])
def test_single_block_decode_writes_output_itself(shape, mode):
    """`direct_out`: with one block per sequence the stage-1 launch writes the attention output (the stage-2 merge of
    a single partial is the identity + bf16 rounding).  Output, lse and scores bit-identical to stage 1 + stage 2,
    with and without the fused store; more than one block per row is refused."""
    import os
    from sparse_vllm_amd.kernels import flash_decode_stage1, flash_decode_stage1_with_score, flash_decode_stage2
    from sparse_vllm_amd.kernels.gqa_flash_decoding_stage1 import direct_out_supported
    if os.environ.get("SVK_DECODE_DIRECT_OUT", "1") == "0":
        pytest.skip("direct output switched off by SVK_DECODE_DIRECT_OUT=0")
    B, Hq, Hkv, D, lens, block_seq = (shape[k] for k in ("B", "Hq", "Hkv", "D", "lens", "block_seq"))
    q, k, v, req, bidx, blen = _rand_case(77 + mode, B, Hq, Hkv, D, [max(n, 1) for n in lens], block_seq)
    blen = np.array(lens, dtype=np.int32)
    max_len = int(max(lens))
    assert direct_out_supported(max_len, block_seq) and not direct_out_supported(block_seq + 1, block_seq)
    tq, tk, tv = to_bf16(q), to_bf16(k), to_bf16(v)
    treq, tb, tl = (torch.from_numpy(x).to(dev()) for x in (req, bidx, blen))
    rng = np.random.default_rng(6)
    nk, nv = (to_bf16(bf16_round((rng.standard_normal((B, Hkv, D)) * 0.3).astype(np.float32))) for _ in range(2))
    sm = torch.from_numpy(np.array([req[bidx[b], max(blen[b], 1) - 1] for b in range(B)], dtype=np.int32)).to(dev())

    def run(direct, store):
        kc, vc = tk.clone(), tv.clone()
        mid = torch.full((B, Hq, 1, D), 7.0, dtype=torch.float32, device=dev())
        lse = torch.full((B, Hq, 1), 7.0, dtype=torch.float32, device=dev())
        o = torch.full((B, Hq, D), 3.0, dtype=torch.bfloat16, device=dev())
        score = None
        if mode == 2:
            score = torch.full((B, max_len), -1e20, dtype=torch.float32, device=dev())
        elif mode == 3:
            score = torch.full((B, Hq, max_len), -1e20, dtype=torch.float32, device=dev())
        kw = dict(new_kv=(nk, nv, sm) if store else None, direct_out=o if direct else None)
        if score is not None:
            flash_decode_stage1_with_score(tq, kc, vc, treq, tb, tl, max_len, mid, lse, score, block_seq, **kw)
        else:
            flash_decode_stage1(tq, kc, vc, treq, tb, tl, max_len, mid, lse, block_seq, **kw)
        if not direct:
            flash_decode_stage2(mid, lse, tl, o, block_seq)
        else:
            assert float(mid.min()) == 7.0                      # the partials were not written
        torch.cuda.synchronize()
        return o, lse, score

    for store in (False, True):
        (o1, l1, s1), (o2, l2, s2) = run(False, store), run(True, store)
        live = torch.from_numpy(blen > 0).to(dev())             # an empty row: NaN from the merge, zeros from the direct write
        assert torch.equal(o1[live].view(torch.int16), o2[live].view(torch.int16))
        assert torch.equal(l1, l2)
        if s1 is not None:
            assert torch.equal(s1, s2)
        if not bool(live.all()):
            assert float(o2[~live].float().abs().max()) == 0.0
    with pytest.raises(ValueError):
        mid = torch.zeros((B, Hq, 2, D), dtype=torch.float32, device=dev())
        lse = torch.zeros((B, Hq, 2), dtype=torch.float32, device=dev())
        flash_decode_stage1(tq, tk, tv, treq, tb, tl, max_len, mid, lse, -(-(-(-max_len // 2)) // 16) * 16,
                            direct_out=torch.empty((B, Hq, D), dtype=torch.bfloat16, device=dev()))


@pytest.mark.parametrize("case", [dict(B=1, Hq=28, D=128, nblk=64, block_seq=66), dict(B=3, Hq=28, D=128, nblk=700, block_seq=256),
                                  dict(B=2, Hq=14, D=64, nblk=33, block_seq=128), dict(B=4, Hq=8, D=128, nblk=3, block_seq=2048, cap=1025)])
def test_split_kv_merge_many_partials_vs_float64(case):
    """Stage-2 merge (max-first form; 256- or 1024-thread workgroups by workspace capacity) against a float64
    log-sum-exp merge: bf16 outputs within one bf16 ulp of the exact result, ragged partial counts per lane."""
    from sparse_vllm_amd.kernels import flash_decode_stage2
    B, Hq, D, nblk, block_seq = (case[k] for k in ("B", "Hq", "D", "nblk", "block_seq"))
    cap = case.get("cap", nblk)
    d = dev()
    g = torch.Generator(device="cpu").manual_seed(nblk)
    lens = torch.randint(1, nblk * block_seq + 1, (B,), generator=g).to(torch.int32)
    lens[0] = nblk * block_seq
    mid = torch.randn(B, Hq, cap, D, generator=g)
    lse = torch.randn(B, Hq, cap, generator=g) * 4
    ref = np.zeros((B, Hq, D))
    for b in range(B):
        n = (int(lens[b]) + block_seq - 1) // block_seq
        w = np.exp(lse[b, :, :n].double().numpy() - lse[b, :, :n].double().numpy().max(axis=1, keepdims=True))
        ref[b] = (w[:, :, None] * mid[b, :, :n].double().numpy()).sum(axis=1) / w.sum(axis=1, keepdims=True)
    o1 = torch.empty(B, Hq, D, dtype=torch.bfloat16, device=d)
    flash_decode_stage2(mid.to(d), lse.to(d), lens.to(d), o1, block_seq)
    torch.cuda.synchronize()
    np.testing.assert_allclose(o1.float().cpu().numpy(), ref, rtol=2 ** -7, atol=1e-3)


@pytest.mark.parametrize("case", [dict(B=1, Hq=28, D=128, nblk=260, block_seq=1024, extra=3), dict(B=4, Hq=28, D=128, nblk=300, block_seq=32),
                                  dict(B=2, Hq=14, D=64, nblk=1024, block_seq=16), dict(B=3, Hq=7, D=128, nblk=257, block_seq=64),
                                  dict(B=2, Hq=28, D=128, nblk=520, block_seq=512, extra=3)])
def test_two_level_split_kv_merge(case, monkeypatch):
    """Launches with more than 256 partials per row: the two-level merge (32 partials per first-level workgroup, the last
    arriver of a (row, head) merges the second level and resets the ticket) against the float64 merge and against the
    one-level kernel (`SVK_STAGE2_SPLIT=0`); ragged rows (rows with 1, 31, 32, 33 partials beside full ones: some first-level
    workgroups have nothing to do, single-group rows skip the second level), extra partials, three launches in a row on
    the same workspace (tickets self-clean: identical bits every time)."""
    from sparse_vllm_amd.kernels import flash_decode_stage2
    monkeypatch.setenv("SVK_STAGE2_SPLIT", "1")
    B, Hq, D, nblk, block_seq = (case[k] for k in ("B", "Hq", "D", "nblk", "block_seq"))
    extra = case.get("extra", 0)
    d = dev()
    g = torch.Generator(device="cpu").manual_seed(nblk + B)
    reg = nblk - extra
    lens = torch.tensor([reg * block_seq] + [int(x) for x in (torch.tensor([1, 31, 33, 32])[: B - 1] * block_seq - 3).clamp_min(1)],
                        dtype=torch.int32)[:B]
    mid = torch.randn(B, Hq, nblk, D, generator=g)
    lse = torch.randn(B, Hq, nblk, generator=g) * 4
    ref = np.zeros((B, Hq, D))
    for b in range(B):
        n = (int(lens[b]) + block_seq - 1) // block_seq + extra
        w = np.exp(lse[b, :, :n].double().numpy() - lse[b, :, :n].double().numpy().max(axis=1, keepdims=True))
        ref[b] = (w[:, :, None] * mid[b, :, :n].double().numpy()).sum(axis=1) / w.sum(axis=1, keepdims=True)
    md, ld, nd = mid.to(d), lse.to(d), lens.to(d)
    outs = []
    for _ in range(3):
        o = torch.full((B, Hq, D), 7.0, dtype=torch.bfloat16, device=d)
        flash_decode_stage2(md, ld, nd, o, block_seq, extra_partials=extra)
        torch.cuda.synchronize()
        outs.append(o.view(torch.int16).cpu().numpy().copy())
        np.testing.assert_allclose(o.float().cpu().numpy(), ref, rtol=2 ** -7, atol=1e-3)
    assert (outs[0] == outs[1]).all() and (outs[1] == outs[2]).all()
    from sparse_vllm_amd.kernels import flash_decoding_stage2 as mod
    assert mod._SPLIT_WS, "the two-level form was not taken"
    monkeypatch.setenv("SVK_STAGE2_SPLIT", "0")
    o1 = torch.empty((B, Hq, D), dtype=torch.bfloat16, device=d)
    flash_decode_stage2(md, ld, nd, o1, block_seq, extra_partials=extra)
    torch.cuda.synchronize()
    np.testing.assert_allclose(o1.float().cpu().numpy(), ref, rtol=2 ** -7, atol=1e-3)
    # two summation orders of the same fp32 terms: bf16 outputs agree to an ulp
    np.testing.assert_allclose(o1.float().cpu().numpy(), outs[0].view(np.int16).astype(np.int32).__lshift__(16).view(np.float32).reshape(B, Hq, D),
                               rtol=2 ** -7, atol=1e-3)
