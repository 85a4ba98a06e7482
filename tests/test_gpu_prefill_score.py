"""GPU parity of svk_prefill_score (MFMA Q.K^T) vs the reference-generated fixtures and the oracle.
Tolerance: rtol 2e-2 / atol 2e-4 on probabilities (the reference's kernel-test bar is rtol=atol=2e-2,
tests/test_prefill_score_kernel.py:282-284); logits: atol 2e-3 (fp32 accumulate of bf16 products)."""

import numpy as np
import pytest

from oracle import bf16_bits_to_f32, bf16_round, f32_to_bf16_bits
from oracle import prefill_score as ops

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev():
    return torch.device("cuda:0")


def to_bf16(x):
    return torch.from_numpy(f32_to_bf16_bits(x).view(np.int16).copy()).to(dev()).view(torch.bfloat16)


def run(q, k, shape, b_req, b_start, b_seq, b_cache, max_q, req, qs, qe, cstart, nrecent, mode, batch_indices=None):
    from sparse_vllm_amd.kernels.prefill_score import prefill_score_fwd
    d = dev()
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(d)
    out = torch.full(shape, 777.0, dtype=torch.float32, device=d)
    prefill_score_fwd(to_bf16(q), to_bf16(k), out, t(b_req), t(b_start), t(b_seq), t(b_cache), max_q, t(req), t(qs), t(qe),
                      candidate_start=cstart, num_recent_tokens=nrecent, score_mode=mode,
                      batch_indices=None if batch_indices is None else t(batch_indices))
    torch.cuda.synchronize()
    return out.cpu().numpy()


def compare(out, ref, logits):
    if logits:
        assert np.array_equal(np.isneginf(out), np.isneginf(ref))
        fin = np.isfinite(ref)
        np.testing.assert_allclose(out[fin], ref[fin], rtol=1e-4, atol=2e-3)
    else:
        np.testing.assert_allclose(out, ref, rtol=2e-2, atol=2e-4)
        assert np.abs(out - ref).max() < 1e-3


@pytest.mark.parametrize("case", ["p1", "p2", "l1", "p3"])
def test_prefill_score_golden(golden, case):
    g = golden("prefill_score")
    q, k = bf16_bits_to_f32(g[f"{case}_q"]), bf16_bits_to_f32(g[f"{case}_k"])
    max_q, cstart, nrecent, is_logits = (int(x) for x in g[f"{case}_cfg"])
    ref = g[f"{case}_score"]
    out = run(q, k, ref.shape, g[f"{case}_b_req"], g[f"{case}_b_start"], g[f"{case}_b_seq"], g[f"{case}_b_cache"], max_q,
              g[f"{case}_req"], g[f"{case}_q_start"], g[f"{case}_q_end"], cstart, nrecent,
              "logits" if is_logits else "probability")
    compare(out, ref, bool(is_logits))


@pytest.mark.parametrize("cfg", [
    # Hq, Hkv, D, [(cache, chunk)], window, cstart, nrecent, mode
    (28, 4, 128, [(3000, 1096)], 128, 0, 0, "probability"),            # H2O chunk: window 128, GQA 7
    (28, 4, 128, [(500, 700), (0, 333)], 32, 64, 512, "probability"),  # SnapKV: window 32, sink 64, recent 512
    (28, 4, 128, [(1000, 600)], 128, 0, 0, "logits"),
    (14, 2, 64, [(300, 300), (10, 100), (0, 50)], 100, 0, 0, "probability"),
    (8, 8, 64, [(100, 200)], 200, 0, 0, "logits"),                      # logits window > 128
    # head_dim 128 (LDS-shared K tile kernel): group sizes 1 / 2 / 4 / 5 / 8, windows that are not a multiple of 32,
    # candidate ranges that start / end inside a 128-key block, a window longer than the chunk, logits windows up to 128
    (32, 8, 128, [(777, 300)], 100, 0, 0, "probability"),
    (8, 8, 128, [(130, 190), (0, 17)], 17, 3, 40, "probability"),
    (16, 8, 128, [(255, 129), (640, 1)], 128, 0, 0, "probability"),
    (10, 2, 128, [(64, 65)], 90, 70, 30, "probability"),
    (16, 2, 128, [(511, 513)], 128, 130, 200, "probability"),
    (32, 8, 128, [(300, 260), (5, 130)], 128, 0, 0, "logits"),
    (10, 2, 128, [(90, 70)], 48, 16, 0, "logits"),
])
def test_prefill_score_vs_oracle(cfg):
    Hq, Hkv, D, seqs, window, cstart, nrecent, mode = cfg
    rng = np.random.default_rng(Hq + len(seqs) + window)
    ctx = [c + n for c, n in seqs]
    nb = len(seqs)
    slots = sum(ctx) + 64
    k = bf16_round((rng.standard_normal((slots, Hkv, D)) * 0.4).astype(np.float32))
    q = bf16_round((rng.standard_normal((sum(n for _, n in seqs), Hq, D)) * 0.4).astype(np.float32))
    req = np.zeros((nb + 2, max(ctx) + 5), dtype=np.int32)
    perm = rng.permutation(slots).astype(np.int32)
    rows = rng.permutation(nb + 2)[:nb].astype(np.int32)
    off = 0
    for i, L in enumerate(ctx):
        req[rows[i], :L] = perm[off: off + L]
        off += L
    b_start = np.concatenate(([0], np.cumsum([n for _, n in seqs])[:-1])).astype(np.int32)
    b_seq = np.array(ctx, dtype=np.int32)
    b_cache = np.array([c for c, _ in seqs], dtype=np.int32)
    qe = b_seq.copy()
    qs = np.array([max(L - window, c) for L, (c, n) in zip(ctx, seqs)], dtype=np.int32)
    max_q = int((qe - qs).max())
    ref = np.empty((nb, max(ctx)), dtype=np.float32)
    ops.prefill_score_fwd(q, k, ref, rows, b_start, b_seq, b_cache, max_q, req, qs, qe, candidate_start=cstart,
                          num_recent_tokens=nrecent, score_mode=mode)
    out = run(q, k, ref.shape, rows, b_start, b_seq, b_cache, max_q, req, qs, qe, cstart, nrecent, mode)
    compare(out, ref, mode == "logits")
    # with batch_indices: ranges in reverse order of the sequences
    if nb > 1:
        bi = np.arange(nb - 1, -1, -1).astype(np.int32)
        ref2 = np.empty_like(ref)
        ops.prefill_score_fwd(q, k, ref2, rows, b_start, b_seq, b_cache, max_q, req, qs[bi], qe[bi], candidate_start=cstart,
                              num_recent_tokens=nrecent, score_mode=mode, batch_indices=bi)
        out2 = run(q, k, ref.shape, rows, b_start, b_seq, b_cache, max_q, req, qs[bi], qe[bi], cstart, nrecent, mode, bi)
        compare(out2, ref2, mode == "logits")


def test_prefill_score_validation():
    from sparse_vllm_amd.kernels.prefill_score import prefill_score_fwd
    d = dev()
    q = torch.zeros(4, 8, 64, dtype=torch.bfloat16, device=d)
    k = torch.zeros(16, 2, 64, dtype=torch.bfloat16, device=d)
    z = torch.zeros(1, dtype=torch.int32, device=d)
    sc = torch.zeros(1, 8, device=d)
    req = torch.zeros(1, 8, dtype=torch.int32, device=d)
    with pytest.raises(ValueError, match="score_mode"):
        prefill_score_fwd(q, k, sc, z, z, z, z, 4, req, z, z, score_mode="softmax")
    with pytest.raises(ValueError, match="too large"):
        prefill_score_fwd(q, k, sc, z, z, z, z, 200, req, z, z)
    with pytest.raises(ValueError, match="one row per score range"):
        prefill_score_fwd(q, k, torch.zeros(2, 8, device=d), z, z, z, z, 4, req, z, z)


@pytest.mark.parametrize("cfg", [
    # Hq, Hkv, [(cache, chunk)], window
    (28, 4, [(3000, 1096)], 128),                     # an H2O chunk: window 128, GQA 7
    (28, 4, [(500, 700), (0, 333), (64, 40)], 128),   # ragged batch, a chunk shorter than the window
    (32, 8, [(100, 300), (700, 65)], 48),             # Llama heads, a window that is not a power of two
    (28, 4, [(0, 20)], 16),
])
def test_prefill_score_from_attention_statistics(cfg):
    """MI355X fusion: context_attention_fwd(score_stats=...) leaves the score window's softmax statistics and a cleared
    score row, prefill_score_fwd(row_stats=...) runs its final pass only.  Same scores as the stand-alone three-launch
    form (the row statistics are the same sums in another order: rtol 1e-4) and as the oracle (tolerance of this file)."""
    import os
    from sparse_vllm_amd.kernels.context_flashattention_nopad import context_attention_fwd
    from sparse_vllm_amd.kernels.prefill_score import prefill_score_fwd, prefill_score_window_pad
    Hq, Hkv, seqs_cfg, window = cfg
    D = 128
    rng = np.random.default_rng(Hq + window)
    nb = len(seqs_cfg)
    ctx = [c + n for c, n in seqs_cfg]
    slots = sum(ctx) + 50
    d = dev()
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(d)
    k = bf16_round((rng.standard_normal((slots, Hkv, D)) * 0.5).astype(np.float32))
    v = bf16_round((rng.standard_normal((slots, Hkv, D)) * 0.5).astype(np.float32))
    total_q = sum(n for _, n in seqs_cfg)
    q = bf16_round((rng.standard_normal((total_q, Hq, D)) * 0.5).astype(np.float32))
    cap = max(ctx) + 7
    req = np.zeros((nb + 1, cap), np.int32)
    perm = rng.permutation(slots).astype(np.int32)
    rows, off = list(range(nb, 0, -1)), 0
    for i, L in enumerate(ctx):
        req[rows[i], :L] = perm[off: off + L]
        off += L
    b_req = np.array(rows, np.int32)
    b_start = np.concatenate(([0], np.cumsum([n for _, n in seqs_cfg])[:-1])).astype(np.int32)
    b_seq = np.array(ctx, np.int32)
    b_cache = np.array([c for c, _ in seqs_cfg], np.int32)
    qe = b_seq.copy()
    qs = np.array([max(L - window, c) for L, (c, n) in zip(ctx, seqs_cfg)], np.int32)
    max_q = int((qe - qs).max())
    tq, tk, tv = to_bf16(q), to_bf16(k), to_bf16(v)
    # stand-alone
    ref3 = torch.full((nb, max(ctx)), 7.0, dtype=torch.float32, device=d)
    prefill_score_fwd(tq, tk, ref3, t(b_req), t(b_start), t(b_seq), t(b_cache), max_q, t(req), t(qs), t(qe))
    # fused: attention leaves the statistics
    wpad = prefill_score_window_pad(Hq, Hkv, max_q)
    stats = torch.full((nb * Hq * wpad,), float("nan"), dtype=torch.float32, device=d)
    got = torch.full((nb, max(ctx)), 7.0, dtype=torch.float32, device=d)
    o = torch.empty_like(tq)
    context_attention_fwd(tq, tk, tv, o, t(b_req), t(b_start), t(b_seq), t(b_cache), max(n for _, n in seqs_cfg), t(req),
                          score_stats=(stats, t(qs), wpad, got))
    torch.cuda.synchronize()
    assert not got.any()                                                        # cleared by the attention launch
    st = stats.view(nb, Hq, wpad).cpu().numpy()
    for i in range(nb):
        assert np.isfinite(st[i, :, : qe[i] - qs[i]]).all() and np.isnan(st[i, :, qe[i] - qs[i]:]).all()
    prefill_score_fwd(tq, tk, got, t(b_req), t(b_start), t(b_seq), t(b_cache), max_q, t(req), t(qs), t(qe), row_stats=stats)
    torch.cuda.synchronize()
    a, b = got.cpu().numpy(), ref3.cpu().numpy()
    np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-8)
    out = np.empty_like(b)
    ops.prefill_score_fwd(q, k, out, b_req, b_start, b_seq, b_cache, max_q, req, qs, qe)
    compare(a, out, False)
    # the statistics only stand for softmax rows over all causal keys
    with pytest.raises(ValueError, match="candidate_start = 0"):
        prefill_score_fwd(tq, tk, got, t(b_req), t(b_start), t(b_seq), t(b_cache), max_q, t(req), t(qs), t(qe),
                          candidate_start=4, row_stats=stats)
