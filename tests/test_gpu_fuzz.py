"""Seeded random-shape sweeps of the three attention-family kernels against the oracle (`-m gpu`): batch, lengths,
block sizes, head shapes, windows and ranges are drawn at random so that tile / block / window boundaries fall in
places the hand-written cases do not name (lengths one off a tile, rows shorter than a block, blocks that end inside
the candidate range, one-token chunks ...).  Tolerances are those of the named tests."""

import os

import numpy as np
import pytest

from oracle import bf16_bits_to_f32, bf16_round, f32_to_bf16_bits
from oracle import decode_attention as oda
from oracle import prefill_attention as opa
from oracle import prefill_score as ops

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

# SVK_FUZZ_SCALE=k multiplies the number of seeds (a one-off deeper sweep; the default keeps the suite short)
_SCALE = max(1, int(os.environ.get("SVK_FUZZ_SCALE", "1")))

HEADS = [(28, 4, 128), (7, 1, 128), (32, 8, 128), (14, 2, 64), (8, 8, 64), (16, 2, 128), (12, 4, 128), (10, 2, 128)]


def dev():
    return torch.device("cuda:0")


def tb(x_f32):
    return torch.from_numpy(f32_to_bf16_bits(x_f32).view(np.int16).copy()).to(dev()).view(torch.bfloat16)


def ti(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev())


@pytest.mark.parametrize("seed", range(24 * _SCALE))
def test_fuzz_decode_stage1(seed):
    from sparse_vllm_amd.kernels import flash_decode_stage1, flash_decode_stage1_with_score, flash_decode_stage2
    rng = np.random.default_rng(1000 + seed)
    Hq, Hkv, D = HEADS[seed % len(HEADS)]
    B = int(rng.integers(1, 5))
    special = [1, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 129, 255, 257]
    lens = np.array([int(rng.choice(special)) if rng.random() < 0.4 else int(rng.integers(1, 700)) for _ in range(B)], np.int32)
    block_seq = int(rng.choice([16, 32, 48, 64, 80, 128, 256, 512]))
    mode = int(rng.choice([0, 2, 3]))
    max_len = int(lens.max()) + int(rng.integers(0, 3)) * 16            # graph-style capacity beyond the longest row
    rows_n = B + 2
    slots = rows_n * max_len + 13
    q = bf16_round((rng.standard_normal((B, Hq, D)) * 0.5).astype(np.float32))
    k = bf16_round((rng.standard_normal((slots, Hkv, D)) * 0.5).astype(np.float32))
    v = bf16_round((rng.standard_normal((slots, Hkv, D)) * 0.5).astype(np.float32))
    table = rng.permutation(slots)[: rows_n * max_len].reshape(rows_n, max_len).astype(np.int32)
    rows = rng.permutation(rows_n)[:B].astype(np.int32)
    nblk = (max_len + block_seq - 1) // block_seq
    mid = torch.full((B, Hq, nblk, D), 7.0, dtype=torch.float32, device=dev())
    lse = torch.full((B, Hq, nblk), 7.0, dtype=torch.float32, device=dev())
    score = score_ref = None
    if mode == 2:
        score = torch.full((B, max_len), -1e20, dtype=torch.float32, device=dev())
        score_ref = np.full((B, max_len), -1e20, np.float32)
    elif mode == 3:
        score = torch.full((B, Hq, max_len), -1e20, dtype=torch.float32, device=dev())
        score_ref = np.full((B, Hq, max_len), -1e20, np.float32)
    if score is None:
        flash_decode_stage1(tb(q), tb(k), tb(v), ti(table), ti(rows), ti(lens), max_len, mid, lse, block_seq)
    else:
        flash_decode_stage1_with_score(tb(q), tb(k), tb(v), ti(table), ti(rows), ti(lens), max_len, mid, lse, score, block_seq)
    o = torch.empty((B, Hq, D), dtype=torch.bfloat16, device=dev())
    flash_decode_stage2(mid, lse, ti(lens), o, block_seq)
    from sparse_vllm_amd.kernels.gqa_flash_decoding_stage1 import direct_out_supported
    if nblk == 1 and direct_out_supported(max_len, block_seq):
        # single-block launch: stage 1 writes the output itself, bit-identical to the two launches above
        o2 = torch.empty_like(o)
        lse2 = torch.full_like(lse, 7.0)
        if score is None:
            flash_decode_stage1(tb(q), tb(k), tb(v), ti(table), ti(rows), ti(lens), max_len, mid, lse2, block_seq, direct_out=o2)
        else:
            score2 = torch.full_like(score, -1e20)
            flash_decode_stage1_with_score(tb(q), tb(k), tb(v), ti(table), ti(rows), ti(lens), max_len, mid, lse2, score2,
                                           block_seq, direct_out=o2)
            assert torch.equal(score2, score)
        assert torch.equal(o2.view(torch.int16), o.view(torch.int16)) and torch.equal(lse2, lse)
    torch.cuda.synchronize()
    mid_ref, lse_ref = oda.flash_decode_stage1(q, k, v, table, rows, lens, max_len, block_seq, attn_score=score_ref)
    o_ref = oda.flash_decode_stage2(mid_ref, lse_ref, lens, block_seq)
    np.testing.assert_allclose(o.float().cpu().numpy(), bf16_round(o_ref), rtol=2e-2, atol=2e-2)
    if score is not None:
        np.testing.assert_allclose(score.cpu().numpy(), score_ref, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("seed", range(16 * _SCALE))
def test_fuzz_prefill_attention(seed):
    from sparse_vllm_amd.kernels.context_flashattention_nopad import context_attention_fwd
    rng = np.random.default_rng(2000 + seed)
    Hq, Hkv, D = HEADS[seed % len(HEADS)]
    B = int(rng.integers(1, 4))
    special = [1, 31, 32, 33, 63, 64, 65, 96, 128, 129]
    chunks = [int(rng.choice(special)) if rng.random() < 0.5 else int(rng.integers(1, 300)) for _ in range(B)]
    pcs = [0 if rng.random() < 0.3 else int(rng.integers(1, 400)) for _ in range(B)]
    T = sum(chunks)
    width = max(c + p for c, p in zip(chunks, pcs)) + 2
    rows_n = B + 1
    slots = rows_n * width + 7
    q = bf16_round((rng.standard_normal((T, Hq, D)) * 0.5).astype(np.float32))
    k = bf16_round((rng.standard_normal((slots, Hkv, D)) * 0.5).astype(np.float32))
    v = bf16_round((rng.standard_normal((slots, Hkv, D)) * 0.5).astype(np.float32))
    table = rng.permutation(slots)[: rows_n * width].reshape(rows_n, width).astype(np.int32)
    req = rng.permutation(rows_n)[:B].astype(np.int32)
    start = np.concatenate(([0], np.cumsum(chunks)[:-1])).astype(np.int32)
    seq_len = np.array([c + p for c, p in zip(chunks, pcs)], np.int32)
    pcl = np.array(pcs, np.int32)
    o = torch.zeros((T, Hq, D), dtype=torch.bfloat16, device=dev())
    context_attention_fwd(tb(q), tb(k), tb(v), o, ti(req), ti(start), ti(seq_len), ti(pcl), max(chunks), ti(table))
    torch.cuda.synchronize()
    ref = opa.context_attention_fwd(q, k, v, req, start, seq_len, pcl, table)
    np.testing.assert_allclose(o.float().cpu().numpy(), bf16_round(ref), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("seed", range(16 * _SCALE))
def test_fuzz_prefill_score(seed):
    from sparse_vllm_amd.kernels.prefill_score import prefill_score_fwd
    rng = np.random.default_rng(3000 + seed)
    Hq, Hkv, D = HEADS[seed % len(HEADS)]
    nb = int(rng.integers(1, 4))
    mode = "logits" if seed % 4 == 3 else "probability"
    seqs = [(0 if rng.random() < 0.3 else int(rng.integers(1, 500)), int(rng.integers(1, 300))) for _ in range(nb)]
    window = int(rng.choice([1, 16, 17, 31, 32, 33, 64, 100, 128]))
    cstart = int(rng.choice([0, 0, 3, 64, 130]))
    nrecent = int(rng.choice([0, 0, 5, 40, 200]))
    ctx = [c + n for c, n in seqs]
    slots = sum(ctx) + 33
    k = bf16_round((rng.standard_normal((slots, Hkv, D)) * 0.4).astype(np.float32))
    q = bf16_round((rng.standard_normal((sum(n for _, n in seqs), Hq, D)) * 0.4).astype(np.float32))
    req = np.zeros((nb + 1, max(ctx) + 3), dtype=np.int32)
    perm = rng.permutation(slots).astype(np.int32)
    rows = rng.permutation(nb + 1)[:nb].astype(np.int32)
    off = 0
    for i, L in enumerate(ctx):
        req[rows[i], :L] = perm[off: off + L]
        off += L
    b_start = np.concatenate(([0], np.cumsum([n for _, n in seqs])[:-1])).astype(np.int32)
    b_seq = np.array(ctx, np.int32)
    b_cache = np.array([c for c, _ in seqs], np.int32)
    qe = b_seq.copy()
    qs = np.array([max(L - window, c) for L, (c, n) in zip(ctx, seqs)], np.int32)
    max_q = int((qe - qs).max())
    ref = np.empty((nb, max(ctx)), np.float32)
    ops.prefill_score_fwd(q, k, ref, rows, b_start, b_seq, b_cache, max_q, req, qs, qe, candidate_start=cstart,
                          num_recent_tokens=nrecent, score_mode=mode)
    out = torch.full(ref.shape, 777.0, dtype=torch.float32, device=dev())
    prefill_score_fwd(tb(q), tb(k), out, ti(rows), ti(b_start), ti(b_seq), ti(b_cache), max_q, ti(req), ti(qs), ti(qe),
                      candidate_start=cstart, num_recent_tokens=nrecent, score_mode=mode)
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    if mode == "logits":
        assert np.array_equal(np.isneginf(out), np.isneginf(ref))
        fin = np.isfinite(ref)
        np.testing.assert_allclose(out[fin], ref[fin], rtol=1e-4, atol=2e-3)
    else:
        np.testing.assert_allclose(out, ref, rtol=2e-2, atol=2e-4)


@pytest.mark.parametrize("seed", range(12 * _SCALE))
def test_fuzz_decode_stage1_page_slot_addressing(seed):
    """`slot_page_size`: over random head shapes, batch sizes, ragged lengths, block sizes, page sizes and score modes the
    launch over a table of page slots equals the launch over the expanded token slots bit for bit."""
    from sparse_vllm_amd.kernels.gqa_flash_decoding_stage1 import _launch
    rng = np.random.default_rng(5000 + seed)
    Hq, Hkv, D = HEADS[seed % len(HEADS)]
    B = int(rng.integers(1, 5))
    page = int(rng.choice([4, 8, 16, 32]))
    n_pages = int(rng.integers(2, 40))
    L = n_pages * page
    lens = np.array([int(rng.integers(1, L + 1)) for _ in range(B)], np.int32)
    lens[int(rng.integers(0, B))] = L
    block_seq = int(rng.choice([16, 32, 64, 96, 128, 256]))
    mode = int(rng.choice([0, 2, 3]))
    rows_n = B + 1
    total_pages = rows_n * n_pages + 3
    ptab = rng.permutation(total_pages)[: rows_n * n_pages].reshape(rows_n, n_pages).astype(np.int32)
    ttab = (ptab[:, :, None].astype(np.int64) * page + np.arange(page)[None, None, :]).reshape(rows_n, -1).astype(np.int32)
    rows = rng.permutation(rows_n)[:B].astype(np.int32)
    slots = total_pages * page
    q = tb((rng.standard_normal((B, Hq, D)) * 0.5).astype(np.float32))
    k = tb((rng.standard_normal((slots, Hkv, D)) * 0.5).astype(np.float32))
    v = tb((rng.standard_normal((slots, Hkv, D)) * 0.5).astype(np.float32))
    nblk = (L + block_seq - 1) // block_seq
    outs = []
    for tab, sps in ((ttab, 0), (ptab, page)):
        mid = torch.full((B, Hq, nblk, D), 7.0, dtype=torch.float32, device=dev())
        lse = torch.full((B, Hq, nblk), 7.0, dtype=torch.float32, device=dev())
        score = None
        if mode == 2:
            score = torch.full((B, L), -1e20, dtype=torch.float32, device=dev())
        elif mode == 3:
            score = torch.full((B, Hq, L), -1e20, dtype=torch.float32, device=dev())
        _launch(q, k, v, ti(tab), ti(rows), ti(lens), L, mid, lse, score, block_seq, None, slot_page_size=sps)
        torch.cuda.synchronize()
        outs.append((mid.view(torch.int32).cpu().numpy(), lse.view(torch.int32).cpu().numpy(),
                     None if score is None else score.view(torch.int32).cpu().numpy()))
    np.testing.assert_array_equal(outs[1][0], outs[0][0])
    np.testing.assert_array_equal(outs[1][1], outs[0][1])
    if mode:
        np.testing.assert_array_equal(outs[1][2], outs[0][2])


@pytest.mark.parametrize("seed", range(16 * _SCALE))
def test_fuzz_quest_build_view(seed):
    """svk_quest_build_view over random row lengths, budgets, score distributions (fp32 / bf16-valued, one sign or both,
    narrow bands, heavy ties, -inf tails), page-slot and token-slot views: the selected pages are the first prev_budget of
    a stable descending argsort, written in ascending page order, followed by the last page."""
    from sparse_vllm_amd.kernels.quest_ops import build_view
    rng = np.random.default_rng(9000 + seed)
    page = 16
    n_prev = int(rng.choice([int(rng.integers(3, 300)), int(rng.integers(300, 5000)), int(rng.integers(5000, 40000))]))
    kb = int(rng.integers(1, min(n_prev, 2000) + 1))
    B = int(rng.integers(1, 4))
    kind = seed % 8
    sc = rng.standard_normal((B, n_prev)).astype(np.float32) * 3
    if kind == 1:
        sc = bf16_round(sc)
    elif kind == 2:
        sc = bf16_round(np.abs(sc) + 40.0)                       # positive narrow band: <= 8 varying key bits
    elif kind == 3:
        sc = bf16_round(-np.abs(sc) - 1.0)
    elif kind == 4:
        sc = rng.choice(np.array([0.5, -0.5, 2.0], dtype=np.float32), (B, n_prev))
    elif kind == 5:
        sc = bf16_round(sc)
        sc[:, n_prev - n_prev // 4:] = -np.inf
    elif kind == 6:
        sc = (1.0 + rng.random((B, n_prev)) * 2.0 ** -12).astype(np.float32)
    elif kind == 7:
        sc = bf16_round(sc * 1e-3)
    n_pages = n_prev + 1
    ptab = np.stack([rng.permutation(n_pages * B + 5)[:n_pages] for _ in range(B)]).astype(np.int32)
    lens = (n_pages * page - rng.integers(0, page, size=B)).astype(np.int32)
    paged = bool(seed & 1)
    keep = (kb + 1) * page
    d = dev()
    packed = torch.full((B, keep), -3, dtype=torch.int32, device=d)
    ll = torch.zeros((B,), dtype=torch.int32, device=d)
    lr = torch.zeros((B,), dtype=torch.int32, device=d)
    ttab = torch.zeros((B, 16), dtype=torch.int32, device=d)
    build_view(ti(sc), ti(ptab), ttab, torch.arange(B, dtype=torch.int32, device=d), ti(lens), packed, ll, lr, page_size=page,
               n_prev=n_prev, prev_budget=kb, token_budget=keep, page_budget_base=kb + 1, max_keep=keep, is_long_text=True,
               emit_page_slots=paged)
    got = packed.cpu().numpy()
    for b in range(B):
        order = np.sort(np.argsort(-sc[b], kind="stable")[:kb])
        exp_pages = np.concatenate((ptab[b, order], ptab[b, n_pages - 1:n_pages]))
        if paged:
            np.testing.assert_array_equal(got[b, : kb + 1], exp_pages)
        else:
            exp = (exp_pages[:, None].astype(np.int64) * page + np.arange(page)[None, :]).reshape(-1)
            np.testing.assert_array_equal(got[b], exp)
    np.testing.assert_array_equal(ll.cpu().numpy(), kb * page + (lens - n_prev * page))
