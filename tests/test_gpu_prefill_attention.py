"""GPU parity of the chunked-prefill causal attention through the C ABI.

Tolerance: rtol = atol = 2e-2 on the bf16 outputs, the reference's own bar for its attention kernels
(tests/test_prefill_score_kernel.py:282-284); observed ~3e-3 (bf16 rounding of P and of the output)."""

import numpy as np
import pytest

from oracle import bf16_bits_to_f32, bf16_round, f32_to_bf16_bits
from oracle import prefill_attention as opa

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

TOL = 2e-2


def dev():
    return torch.device("cuda:0")


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev())


def bfb(bits):
    return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16).copy()).to(dev()).view(torch.bfloat16)


def run(q_bits, k_bits, v_bits, req, start, seq_len, pcl, table, max_input_len):
    from sparse_vllm_amd.kernels.context_flashattention_nopad import context_attention_fwd
    q = bfb(q_bits)
    o = torch.full(q.shape, 7.0, dtype=torch.bfloat16, device=dev())
    context_attention_fwd(q, bfb(k_bits), bfb(v_bits), o, t(req), t(start), t(seq_len), t(pcl), int(max_input_len), t(table))
    torch.cuda.synchronize()
    return o.float().cpu().numpy()


def test_prefill_attention_golden(golden):
    g = golden("prefill_attention")
    o = run(g["q"], g["k"], g["v"], g["req"], g["start"], g["seq_len"], g["pcl"], g["table"], 37)
    np.testing.assert_allclose(o, g["o"], rtol=TOL, atol=TOL)


@pytest.mark.parametrize("Hq,Hkv,D,chunks,pcs", [
    (28, 4, 128, [300, 33, 1], [200, 0, 64]),          # Qwen2.5-7B heads: chunk with a cached prefix, a fresh prompt, a 1-token chunk
    (32, 8, 128, [129], [0]),                          # Llama-3.1-8B heads
    (14, 2, 64, [70, 64], [5, 130]),                   # Qwen2.5-0.5B heads
    (8, 8, 64, [40], [17]),                            # MHA
])
def test_prefill_attention_random(Hq, Hkv, D, chunks, pcs):
    rng = np.random.default_rng(Hq + D + len(chunks))
    B = len(chunks)
    T = sum(chunks)
    width = max(c + p for c, p in zip(chunks, pcs)) + 3
    rows = B + 2
    slots = rows * width + 11
    f2b = f32_to_bf16_bits
    q = f2b((rng.standard_normal((T, Hq, D)) * 0.5).astype(np.float32))
    k = f2b((rng.standard_normal((slots, Hkv, D)) * 0.5).astype(np.float32))
    v = f2b((rng.standard_normal((slots, Hkv, D)) * 0.5).astype(np.float32))
    table = rng.permutation(slots)[: rows * width].reshape(rows, width).astype(np.int32)
    req = rng.permutation(rows)[:B].astype(np.int32)
    start = np.concatenate(([0], np.cumsum(chunks)[:-1])).astype(np.int32)
    seq_len = np.array([c + p for c, p in zip(chunks, pcs)], np.int32)
    pcl = np.array(pcs, np.int32)
    o = run(q, k, v, req, start, seq_len, pcl, table, max(chunks))
    f = bf16_bits_to_f32
    ref = opa.context_attention_fwd(f(q), f(k), f(v), req, start, seq_len, pcl, table)
    np.testing.assert_allclose(o, bf16_round(ref), rtol=TOL, atol=TOL)
    dense = opa.context_attention_dense(f(q), f(k), f(v), req, start, seq_len, pcl, table)
    np.testing.assert_allclose(o, dense, rtol=TOL, atol=TOL)


def test_prefill_attention_rows_outside_the_chunks_untouched():
    """max_input_len larger than a sequence's chunk: the extra query blocks must not write."""
    rng = np.random.default_rng(3)
    Hq, Hkv, D = 8, 2, 64
    chunks, pcs = [5, 90], [0, 0]
    T = sum(chunks)
    f2b = f32_to_bf16_bits
    q = f2b(rng.standard_normal((T + 4, Hq, D)).astype(np.float32))      # 4 trailing tokens belong to nobody
    k = f2b(rng.standard_normal((300, Hkv, D)).astype(np.float32))
    v = f2b(rng.standard_normal((300, Hkv, D)).astype(np.float32))
    table = rng.permutation(300)[:200].reshape(2, 100).astype(np.int32)
    o = run(q, k, v, np.array([0, 1], np.int32), np.array([0, 5], np.int32), np.array(chunks, np.int32), np.array(pcs, np.int32),
            table, 90)
    assert (o[T:] == 7.0).all()
    assert np.isfinite(o[:T]).all() and not (o[:T] == 7.0).all()


def to_bf16(x_f32):
    return bfb(f32_to_bf16_bits(x_f32))


def _score_case(g, Hq_first=True):
    f = bf16_bits_to_f32
    return (f(g["s_q"]), f(g["s_k"]), g["s_req"], g["s_start"], g["s_seq_len"], g["s_pcl"], g["s_table"])


def test_context_attention_score_forms_golden_and_random(golden):
    """`context_attention_fwd(attn_score=...)` (context_flashattention_nopad.py:82-240): the 3-D form adds per-head sums of
    the raw logits over the chunk's query rows, the 2-D form takes the head / 128-row-block maximum of their means.
    Against the interpreter fixture (fp32 sums in another order: rtol 1e-4 / atol 1e-3 on sums of ~100 products of
    magnitude <~ 10) and the oracle at Qwen2.5-7B heads with several query blocks."""
    from oracle import prefill_attention as opa
    from sparse_vllm_amd.kernels.context_flashattention_nopad import context_attention_fwd
    g = golden("prefill_attention")
    q, k, req, start, seq_len, pcl, table = _score_case(g)
    d = dev()
    tq, tk = to_bf16(q), to_bf16(k)

    def run(q_, k_, req_, start_, seq_, pcl_, table_, score):
        o = torch.empty_like(q_)
        context_attention_fwd(q_, k_, k_, o, t(req_), t(start_), t(seq_), t(pcl_), int((seq_ - pcl_).max()), t(table_), attn_score=score)
        torch.cuda.synchronize()
        return o

    s3 = torch.zeros(g["s_score3"].shape, dtype=torch.float32, device=d)
    o3 = run(tq, tk, req, start, seq_len, pcl, table, s3)
    np.testing.assert_allclose(s3.cpu().numpy(), g["s_score3"], rtol=1e-4, atol=1e-3)
    s2 = torch.full(g["s_score2"].shape, -1.0e20, dtype=torch.float32, device=d)
    o2 = run(tq, tk, req, start, seq_len, pcl, table, s2)
    np.testing.assert_allclose(s2.cpu().numpy(), g["s_score2"], rtol=1e-4, atol=1e-4)
    assert torch.equal(o2, o3)                                     # the attention output does not depend on the score form
    o0 = torch.empty_like(tq)
    context_attention_fwd(tq, tk, tk, o0, t(req), t(start), t(seq_len), t(pcl), int((seq_len - pcl).max()), t(table))
    assert torch.equal(o0, o3)
    # accumulate semantics: a second call adds to the 3-D buffer and leaves a saturated 2-D buffer alone
    run(tq, tk, req, start, seq_len, pcl, table, s3)
    np.testing.assert_allclose(s3.cpu().numpy(), 2 * g["s_score3"], rtol=1e-4, atol=2e-3)
    # Qwen2.5-7B heads, three query blocks + a cached prefix
    rng = np.random.default_rng(12)
    Hq, Hkv, D = 28, 4, 128
    chunk, pc = [300, 77], [500, 0]
    L = [c + p for c, p in zip(chunk, pc)]
    slots = sum(L) + 9
    qn = bf16_round((rng.standard_normal((sum(chunk), Hq, D)) * 0.4).astype(np.float32))
    kn = bf16_round((rng.standard_normal((slots, Hkv, D)) * 0.4).astype(np.float32))
    tab = np.zeros((2, max(L) + 3), np.int32)
    perm = rng.permutation(slots).astype(np.int32)
    tab[0, :L[0]], tab[1, :L[1]] = perm[:L[0]], perm[L[0]:L[0] + L[1]]
    reqn, startn = np.array([0, 1], np.int32), np.array([0, chunk[0]], np.int32)
    seqn, pcn = np.array(L, np.int32), np.array(pc, np.int32)
    for dim in (3, 2):
        shape = (2, Hq, max(L)) if dim == 3 else (2, max(L))
        init = 0.0 if dim == 3 else -1.0e20
        got = torch.full(shape, init, dtype=torch.float32, device=d)
        run(to_bf16(qn), to_bf16(kn), reqn, startn, seqn, pcn, tab, got)
        ref = opa.context_attention_scores(qn, kn, reqn, startn, seqn, pcn, tab, np.full(shape, init, np.float32))
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-4, atol=2e-3)
    with pytest.raises(ValueError, match="attn_score must be"):
        run(tq, tk, req, start, seq_len, pcl, table, torch.zeros((3,), device=d))
