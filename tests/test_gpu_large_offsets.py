"""K/V tensors that span more than 4 GiB take the 64-bit row addressing of the decode and prefill attention kernels
(below 4 GiB a row address is the tensor base + a 32-bit byte offset).  The same rows are run once from compact tensors
and once from a strided view whose slot stride pushes the span past 4 GiB: the kernels do the same arithmetic on the
same bytes, so the results must be bit-identical.  Also: GQA group sizes 1..8 of the head_dim-128 prefill kernel."""

import numpy as np
import pytest

from oracle import bf16_round, bf16_bits_to_f32, f32_to_bf16_bits
from oracle import prefill_attention as opa

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

TOL = 2e-2


def dev():
    return torch.device("cuda:0")


def _spread(x, stride_elems):
    """[slots, Hkv, D] bf16 -> the same values as a view with slot stride `stride_elems` (span > 4 GiB)."""
    slots, hkv, d = x.shape
    big = torch.zeros((slots, stride_elems), dtype=torch.bfloat16, device=x.device)
    big[:, : hkv * d] = x.reshape(slots, hkv * d)
    view = big[:, : hkv * d].view(slots, hkv, d)
    assert view.stride(0) == stride_elems and slots * stride_elems * 2 > (1 << 32)
    return view


@pytest.mark.parametrize("mode", [0, 2])
def test_decode_stage1_64bit_offsets_equal_32bit(mode):
    from sparse_vllm_amd.kernels.gqa_flash_decoding_stage1 import flash_decode_stage1, flash_decode_stage1_with_score
    torch.manual_seed(1)
    B, Hq, Hkv, D, L, block_seq = 3, 28, 4, 128, 300, 64
    slots = B * L + 50
    q = (torch.randn(B, Hq, D, device=dev()) * 0.4).bfloat16()
    k = (torch.randn(slots, Hkv, D, device=dev()) * 0.4).bfloat16()
    v = (torch.randn(slots, Hkv, D, device=dev()) * 0.4).bfloat16()
    table = torch.randperm(slots, device=dev())[: B * L].to(torch.int32).view(B, L)
    req = torch.arange(B, dtype=torch.int32, device=dev())
    lens = torch.tensor([L, L - 37, 65], dtype=torch.int32, device=dev())
    nblk = (L + block_seq - 1) // block_seq
    outs = []
    for spread in (False, True):
        kk, vv = (k, v) if not spread else (_spread(k, 2_400_000), _spread(v, 2_400_000))
        mid = torch.zeros(B, Hq, nblk, D, dtype=torch.float32, device=dev())
        lse = torch.zeros(B, Hq, nblk, dtype=torch.float32, device=dev())
        if mode == 0:
            flash_decode_stage1(q, kk, vv, table, req, lens, L, mid, lse, block_seq)
            sc = None
        else:
            sc = torch.full((B, L), -1e20, dtype=torch.float32, device=dev())
            flash_decode_stage1_with_score(q, kk, vv, table, req, lens, L, mid, lse, sc, block_seq)
        torch.cuda.synchronize()
        outs.append((mid.cpu().numpy(), lse.cpu().numpy(), None if sc is None else sc.cpu().numpy()))
        del kk, vv
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    if mode:
        np.testing.assert_array_equal(outs[0][2], outs[1][2])
        assert np.isfinite(outs[0][2][0]).all()


@pytest.mark.parametrize("Hq,Hkv,D", [(28, 4, 128), (14, 2, 64)])
def test_prefill_attention_64bit_offsets_equal_32bit(Hq, Hkv, D):
    from sparse_vllm_amd.kernels.context_flashattention_nopad import context_attention_fwd
    torch.manual_seed(2)
    chunks, pcs = [150, 40], [210, 0]
    T = sum(chunks)
    width = 400
    slots = 2 * width + 30
    q = (torch.randn(T, Hq, D, device=dev()) * 0.4).bfloat16()
    k = (torch.randn(slots, Hkv, D, device=dev()) * 0.4).bfloat16()
    v = (torch.randn(slots, Hkv, D, device=dev()) * 0.4).bfloat16()
    table = torch.randperm(slots, device=dev())[: 2 * width].to(torch.int32).view(2, width)
    req = torch.tensor([1, 0], dtype=torch.int32, device=dev())
    start = torch.tensor([0, chunks[0]], dtype=torch.int32, device=dev())
    seq_len = torch.tensor([c + p for c, p in zip(chunks, pcs)], dtype=torch.int32, device=dev())
    pcl = torch.tensor(pcs, dtype=torch.int32, device=dev())
    outs = []
    for spread in (False, True):
        kk, vv = (k, v) if not spread else (_spread(k, 2_700_000), _spread(v, 2_700_000))
        o = torch.zeros_like(q)
        context_attention_fwd(q, kk, vv, o, req, start, seq_len, pcl, max(chunks), table)
        torch.cuda.synchronize()
        outs.append(o.float().cpu().numpy())
        del kk, vv
    np.testing.assert_array_equal(outs[0], outs[1])
    assert np.isfinite(outs[0]).all() and np.abs(outs[0]).max() > 0


@pytest.mark.parametrize("Hq,Hkv", [(4, 4), (8, 4), (10, 2), (12, 2), (16, 2), (3, 1), (6, 1)])
def test_prefill_attention_group_sizes_head_dim_128(Hq, Hkv):
    """GQA groups 1, 2, 5, 6, 8, 3, 6 waves per workgroup (the LDS-shared tile kernel deals its DMA groups by wave count)."""
    from sparse_vllm_amd.kernels.context_flashattention_nopad import context_attention_fwd
    rng = np.random.default_rng(Hq * 16 + Hkv)
    D = 128
    chunks, pcs = [97, 260], [333, 0]
    B, T = len(chunks), sum(chunks)
    width = 450
    slots = B * width + 9
    f2b, f = f32_to_bf16_bits, bf16_bits_to_f32
    q = f2b((rng.standard_normal((T, Hq, D)) * 0.5).astype(np.float32))
    k = f2b((rng.standard_normal((slots, Hkv, D)) * 0.5).astype(np.float32))
    v = f2b((rng.standard_normal((slots, Hkv, D)) * 0.5).astype(np.float32))
    table = rng.permutation(slots)[: B * width].reshape(B, width).astype(np.int32)
    req = np.array([1, 0], np.int32)
    start = np.array([0, chunks[0]], np.int32)
    seq_len = np.array([c + p for c, p in zip(chunks, pcs)], np.int32)
    pcl = np.array(pcs, np.int32)

    def tb(bits):
        return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16).copy()).to(dev()).view(torch.bfloat16)

    def ti(x):
        return torch.from_numpy(np.ascontiguousarray(x)).to(dev())

    o = torch.zeros((T, Hq, D), dtype=torch.bfloat16, device=dev())
    context_attention_fwd(tb(q), tb(k), tb(v), o, ti(req), ti(start), ti(seq_len), ti(pcl), max(chunks), ti(table))
    torch.cuda.synchronize()
    ref = opa.context_attention_fwd(f(q), f(k), f(v), req, start, seq_len, pcl, table)
    np.testing.assert_allclose(o.float().cpu().numpy(), bf16_round(ref), rtol=TOL, atol=TOL)
