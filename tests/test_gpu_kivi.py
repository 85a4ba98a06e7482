"""GPU parity of the KIVI-int4 full-layer decode stage 1 through the C ABI.

Tolerances: vs the oracle with bf16-rounded dequantised K/V (the real-model data flow, `.to(q.dtype)`):
partials rtol = atol = 2e-2 (the reference's own bar for decode partials), raw scores atol 1e-4 + rtol 1e-5
(fp32 accumulation of exact bf16 products).  The committed fixture was produced by the reference kernel
under the Triton interpreter on fp32 tensors, where dequantised K/V are NOT rounded to bf16, so against it
scores get the bf16 rounding slack of K (2^-8 relative per element)."""

import numpy as np
import pytest

from oracle import bf16_bits_to_f32, bf16_round, f32_to_bf16_bits
from oracle import deltakv as od
from oracle import kivi as ok

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ATTN_TOL = 2e-2


def dev():
    return torch.device("cuda:0")


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev())


def bf(x_bits):
    return torch.from_numpy(np.ascontiguousarray(x_bits).view(np.int16).copy()).to(dev()).view(torch.bfloat16)


def kparam(x):
    """per-channel key scale/min: fp32 (what the reference manager stores) or bf16 bit patterns"""
    return t(x) if x.dtype == np.float32 else bf(x)


def kparam_f32(x):
    return x if x.dtype == np.float32 else bf16_bits_to_f32(x)


def run_gpu(inp_bits, maps, *, max_len, G, block_seq, score_shape=None):
    from sparse_vllm_amd.kernels.deltakv_kernels import full_layer_kivi_flash_decode_stage1
    q = bf(inp_bits["q"])
    B, Hq, D = q.shape
    nblk = (max_len + block_seq - 1) // block_seq
    mid = torch.full((B, Hq, nblk, D), 7.0, dtype=torch.float32, device=dev())
    lse = torch.full((B, Hq, nblk), 7.0, dtype=torch.float32, device=dev())
    score = None if score_shape is None else torch.full(score_shape, -1e20, dtype=torch.float32, device=dev())
    full_layer_kivi_flash_decode_stage1(
        q=q, raw_k=bf(inp_bits["raw_k"]), raw_v=bf(inp_bits["raw_v"]), raw_slots_map=t(maps["raw_map"]),
        kivi_block_slots_map=t(maps["blk_map"]), kivi_block_start_pos=t(maps["blk_start"]),
        key_packed=t(inp_bits["key_packed"]), key_scales=kparam(inp_bits["key_scales"]), key_mins=kparam(inp_bits["key_mins"]),
        value_packed=t(inp_bits["value_packed"]), value_scales=bf(inp_bits["value_scales"]),
        value_mins=bf(inp_bits["value_mins"]), req_indices=t(maps["req"]), context_lens=t(maps["lens"]),
        max_len_in_batch=max_len, mid_out=mid, mid_out_logsumexp=lse, group_size=G, block_seq=block_seq,
        attn_score=score)
    torch.cuda.synchronize()
    return mid.cpu().numpy(), lse.cpu().numpy(), None if score is None else score.cpu().numpy()


def run_oracle(inp_bits, maps, *, max_len, G, block_seq, score_shape=None):
    f = bf16_bits_to_f32
    score = None if score_shape is None else np.full(score_shape, -1e20, np.float32)
    mid, lse = ok.full_layer_kivi_flash_decode_stage1(
        q=f(inp_bits["q"]), raw_k=f(inp_bits["raw_k"]), raw_v=f(inp_bits["raw_v"]), raw_slots_map=maps["raw_map"],
        kivi_block_slots_map=maps["blk_map"], kivi_block_start_pos=maps["blk_start"], key_packed=inp_bits["key_packed"],
        key_scales=kparam_f32(inp_bits["key_scales"]), key_mins=kparam_f32(inp_bits["key_mins"]),
        value_packed=inp_bits["value_packed"], value_scales=f(inp_bits["value_scales"]), value_mins=f(inp_bits["value_mins"]), req_indices=maps["req"],
        context_lens=maps["lens"], max_len_in_batch=max_len, group_size=G, block_seq=block_seq, attn_score=score)
    return mid, lse, score


def valid_blocks(lens, block_seq, nblk):
    return (np.arange(nblk)[None, :] * block_seq) < np.asarray(lens)[:, None]


def compare(got, ref, lens, block_seq, score_tol=(1e-5, 1e-4)):
    """The kernel aligns its workgroup ranges to KIVI block boundaries (workgroup i covers
    [i*BS + f(i*BS), (i+1)*BS + f((i+1)*BS)), f = distance to the next block boundary), so partial i is not the
    reference's partial i; what stage 2 merges out of them is.  Compare the merged outputs, the neutral tail partials
    and the position-indexed raw scores."""
    from oracle import decode_attention as oda
    mid, lse, score = got
    mid_r, lse_r, score_r = ref
    lens = np.asarray(lens, np.int32)
    o = oda.flash_decode_stage2(mid, lse, lens, block_seq)
    o_r = oda.flash_decode_stage2(mid_r, lse_r, lens, block_seq)
    np.testing.assert_allclose(o, o_r, rtol=ATTN_TOL, atol=ATTN_TOL)
    vb = valid_blocks(lens, block_seq, mid.shape[2])
    for b in range(mid.shape[0]):
        assert np.all(np.isneginf(lse[b][:, ~vb[b]])) and not mid[b][:, ~vb[b]].any()
        assert np.isfinite(mid[b][:, vb[b]]).all()
    if score is not None:
        np.testing.assert_allclose(score, score_r, rtol=score_tol[0], atol=score_tol[1])


def test_kivi_stage1_golden(golden):
    g = golden("kivi")
    G, block_seq, max_len = (int(x) for x in g["cfg"])
    maps = dict(raw_map=g["raw_map"], blk_map=g["blk_map"], blk_start=g["blk_start"], req=g["req"], lens=g["lens"])
    got = run_gpu(g, maps, max_len=max_len, G=G, block_seq=block_seq, score_shape=g["score"].shape)
    # reference fixture (no bf16 rounding of dequantised K/V inside the interpreter run)
    compare(got, (g["mid_o"], g["mid_lse"], g["score"]), g["lens"], block_seq, score_tol=(2e-2, 5e-2))
    # oracle with the real bf16 data flow: tight
    compare(got, run_oracle(g, maps, max_len=max_len, G=G, block_seq=block_seq, score_shape=g["score"].shape),
            g["lens"], block_seq)


@pytest.mark.parametrize("block_seq,spare", [(1024, 0), (1024, 3), (2048, 3), (256, 0)])
def test_kivi_stage1_long_row_golden(golden, block_seq, spare):
    """The reference's kernel on one 8259-token row (tests/golden/kivi.npz `long_*`, inputs from the shared seed): merged
    output against the merge of the reference's partials, raw scores position by position - with the reference's
    block_seq and with others (any partition of the row merges to the same output), with and without the extra
    workgroups; this shape (head_dim 128, one KV head, fp32 key parameters) takes the wide kernel."""
    import golden_inputs
    from oracle import decode_attention as oda
    from sparse_vllm_amd.kernels.deltakv_kernels import full_layer_kivi_flash_decode_stage1
    from sparse_vllm_amd.kernels.flash_decoding_stage2 import flash_decode_stage2
    g = golden("kivi")
    G, ref_block_seq, length = (int(x) for x in g["long_cfg"])
    d = golden_inputs.kivi_long_row_inputs()
    f2b = f32_to_bf16_bits
    q = bf(f2b(d["q"]))
    nblk = (length + block_seq - 1) // block_seq
    mid = torch.full((1, d["Hq"], nblk + spare, d["D"]), 7.0, dtype=torch.float32, device=dev())
    lse = torch.full((1, d["Hq"], nblk + spare), 7.0, dtype=torch.float32, device=dev())
    score = torch.full((1, d["Hq"], length), -1e20, dtype=torch.float32, device=dev())
    lens = t(d["lens"])
    extra = full_layer_kivi_flash_decode_stage1(
        q=q, raw_k=bf(f2b(d["raw_k"])), raw_v=bf(f2b(d["raw_v"])), raw_slots_map=t(d["raw_map"]),
        kivi_block_slots_map=t(d["blk_map"]), kivi_block_start_pos=t(d["blk_start"]), key_packed=t(d["key_packed"]),
        key_scales=t(d["key_scales"]), key_mins=t(d["key_mins"]), value_packed=t(d["value_packed"]),
        value_scales=bf(f2b(d["value_scales"])), value_mins=bf(f2b(d["value_mins"])), req_indices=t(d["req"]), context_lens=lens,
        max_len_in_batch=length, mid_out=mid, mid_out_logsumexp=lse, group_size=G, block_seq=block_seq, attn_score=score,
        extra_partial_slots=spare)
    assert extra == (3 if spare and block_seq % 128 == 0 else 0)
    o = torch.empty((1, d["Hq"], d["D"]), dtype=torch.bfloat16, device=dev())
    flash_decode_stage2(mid, lse, lens, o, block_seq, extra_partials=extra)
    torch.cuda.synchronize()
    o_ref = oda.flash_decode_stage2(g["long_mid_o"], g["long_mid_lse"], d["lens"], ref_block_seq)
    # the interpreter run keeps the dequantised K / V in fp32 (no `.to(q.dtype)` rounding): the fixture tolerance of
    # test_kivi_stage1_golden
    np.testing.assert_allclose(o.float().cpu().numpy(), o_ref, rtol=ATTN_TOL, atol=ATTN_TOL)
    np.testing.assert_allclose(score.cpu().numpy(), g["long_score"], rtol=2e-2, atol=5e-2)
    # and the oracle with the real bf16 data flow: tight
    sc_o = np.full(g["long_score"].shape, -1e20, np.float32)
    mid_o, lse_o = ok.full_layer_kivi_flash_decode_stage1(
        q=d["q"], raw_k=d["raw_k"], raw_v=d["raw_v"], raw_slots_map=d["raw_map"], kivi_block_slots_map=d["blk_map"],
        kivi_block_start_pos=d["blk_start"], key_packed=d["key_packed"], key_scales=d["key_scales"], key_mins=d["key_mins"],
        value_packed=d["value_packed"], value_scales=d["value_scales"], value_mins=d["value_mins"], req_indices=d["req"],
        context_lens=d["lens"], max_len_in_batch=length, group_size=G, block_seq=ref_block_seq, attn_score=sc_o)
    np.testing.assert_allclose(score.cpu().numpy(), sc_o, rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(o.float().cpu().numpy(), bf16_round(oda.flash_decode_stage2(mid_o, lse_o, d["lens"], ref_block_seq)),
                               rtol=ATTN_TOL, atol=ATTN_TOL)


def make_case(rng, *, B, Hq, Hkv, D, G, lens, rows, raw_tail, sink, key_f32=False):
    """Rows = [sink raw tokens | KIVI blocks of G tokens | raw tail]; block slots and raw slots scattered."""
    f2b = f32_to_bf16_bits
    max_len = int(max(lens))
    width = max_len + 5
    n_blocks_row = [max(0, (int(n) - sink - raw_tail)) // G for n in lens]
    n_blocks = sum(n_blocks_row) + 3
    n_raw = sum(int(n) - nb * G for n, nb in zip(lens, n_blocks_row)) + 7
    raw_map = np.full((rows, width), -1, np.int32)
    blk_map = np.full((rows, width), -1, np.int32)
    blk_start = np.zeros(n_blocks, np.int32)
    req = rng.permutation(rows)[:B].astype(np.int32)
    raw_perm = rng.permutation(n_raw).astype(np.int32)
    blk_perm = rng.permutation(n_blocks).astype(np.int32)
    ru = bu = 0
    for b in range(B):
        r, n, nb = int(req[b]), int(lens[b]), n_blocks_row[b]
        s = min(sink, n)
        raw_map[r, :s] = raw_perm[ru: ru + s]; ru += s
        for i in range(nb):
            bs = int(blk_perm[bu]); bu += 1
            blk_map[r, s + i * G: s + (i + 1) * G] = bs
            blk_start[bs] = s + i * G
        tail0 = s + nb * G
        raw_map[r, tail0:n] = raw_perm[ru: ru + n - tail0]; ru += n - tail0
    kdata = rng.standard_normal((n_blocks, Hkv, D, G)).astype(np.float32)
    vdata = rng.standard_normal((n_blocks, Hkv, G, D)).astype(np.float32)
    kc, ks, km = od.quantize_pack_grouped(kdata.reshape(-1, G), G, 4)
    vc, vs, vm = od.quantize_pack_grouped(vdata.reshape(-1, D), G, 4)
    bits = dict(
        q=f2b(rng.standard_normal((B, Hq, D)).astype(np.float32)),
        raw_k=f2b(rng.standard_normal((n_raw, Hkv, D)).astype(np.float32)),
        raw_v=f2b(rng.standard_normal((n_raw, Hkv, D)).astype(np.float32)),
        key_packed=kc.reshape(n_blocks, Hkv, D, G // 8).astype(np.int32),
        key_scales=ks.reshape(n_blocks, Hkv, D).astype(np.float32) if key_f32 else f2b(ks.reshape(n_blocks, Hkv, D)),
        key_mins=km.reshape(n_blocks, Hkv, D).astype(np.float32) if key_f32 else f2b(km.reshape(n_blocks, Hkv, D)),
        value_packed=vc.reshape(n_blocks, Hkv, G, D // 8).astype(np.int32),
        value_scales=f2b(vs.reshape(n_blocks, Hkv, G, D // G)), value_mins=f2b(vm.reshape(n_blocks, Hkv, G, D // G)))
    maps = dict(raw_map=raw_map, blk_map=blk_map, blk_start=blk_start, req=req, lens=np.asarray(lens, np.int32))
    return bits, maps, max_len


@pytest.mark.parametrize("Hq,Hkv,D,G,block_seq,with_score,key_f32", [
    (28, 4, 128, 32, 128, True, True),       # Qwen2.5-7B heads (paper config), observation layer, fp32 key params
    (28, 4, 128, 32, 64, False, False),
    (32, 8, 128, 32, 96, True, True),        # Llama-3.1-8B heads
    (16, 2, 64, 32, 48, True, False),
    (8, 8, 64, 64, 32, False, True),         # MHA, one group per head
])
def test_kivi_stage1_random(Hq, Hkv, D, G, block_seq, with_score, key_f32):
    rng = np.random.default_rng(Hq * 131 + block_seq)
    lens = [517, 64, 9, 300]
    bits, maps, max_len = make_case(rng, B=4, Hq=Hq, Hkv=Hkv, D=D, G=G, lens=lens, rows=6, raw_tail=40, sink=8,
                                     key_f32=key_f32)
    shape = (4, Hq, max_len) if with_score else None
    got = run_gpu(bits, maps, max_len=max_len, G=G, block_seq=block_seq, score_shape=shape)
    ref = run_oracle(bits, maps, max_len=max_len, G=G, block_seq=block_seq, score_shape=shape)
    compare(got, ref, lens, block_seq)


@pytest.mark.parametrize("sink,raw_tail,lens,block_seq", [
    (8, 40, [1500, 777, 136, 8], 512),       # sink tile = 8 raw rows + 120 quantised tokens: fast pass + raw pass
    (32, 64, [1500, 1181], 256),             # whole raw block in front, 64..95 raw rows behind, ragged ends
    (0, 0, [1024, 515], 256),                # no raw row at all but a ragged end
    (4, 21, [700, 300], 128),                # groups not aligned to 8 tokens: the per-token fallback
    (8, 200, [1000, 420], 1024),             # raw tail longer than a tile: pure raw tiles
])
def test_kivi_stage1_tile_passes(sink, raw_tail, lens, block_seq):
    """Tiles are split into a pass over the 8-aligned quantised groups (word loads), a pass over the raw rows
    (vector loads) and the per-token fallback; every combination must give the oracle's merged output and scores."""
    rng = np.random.default_rng(sink * 1000 + raw_tail)
    Hq, Hkv, D, G = 28, 4, 128, 32
    B = len(lens)
    bits, maps, max_len = make_case(rng, B=B, Hq=Hq, Hkv=Hkv, D=D, G=G, lens=lens, rows=B + 1, raw_tail=raw_tail,
                                     sink=sink, key_f32=True)
    shape = (B, Hq, max_len)
    got = run_gpu(bits, maps, max_len=max_len, G=G, block_seq=block_seq, score_shape=shape)
    ref = run_oracle(bits, maps, max_len=max_len, G=G, block_seq=block_seq, score_shape=shape)
    compare(got, ref, lens, block_seq)


@pytest.mark.parametrize("block_seq", [128, 256, 1024])
def test_kivi_stage1_irregular_maps(block_seq):
    """The maps are per token (deltakv_kernels.py:806-828): a raw token in the middle of a quantised run, a block whose
    recorded start is shifted (tokens in front of it fall outside and are skipped, :821-823), a block that is only half
    mapped, blocks out of position order.  The wide whole-block path must reject exactly those tiles and the narrow
    paths must give the oracle's result - scores position by position, merged outputs."""
    rng = np.random.default_rng(block_seq)
    Hq, Hkv, D, G = 28, 4, 128, 32
    lens = [1500, 1100, 700]
    B = len(lens)
    bits, maps, max_len = make_case(rng, B=B, Hq=Hq, Hkv=Hkv, D=D, G=G, lens=lens, rows=B + 1, raw_tail=40, sink=8, key_f32=True)
    raw_map, blk_map, blk_start = maps["raw_map"], maps["blk_map"], maps["blk_start"]
    r0, r1, r2 = (int(x) for x in maps["req"])
    spare_raw = int(raw_map.max()) - 3                    # make_case leaves 7 unused raw slots at the top
    # row 0: a raw token inside block 5 of the row, and block 9 recorded 3 tokens late
    p = 8 + 5 * G + 11
    raw_map[r0, p] = spare_raw
    blk_map[r0, p] = -1
    blk_start[blk_map[r0, 8 + 9 * G]] += 3
    # row 1: the second half of block 3 unmapped (neither raw nor quantised: invalid tokens), blocks 12 and 13 swapped
    blk_map[r1, 8 + 3 * G + 16: 8 + 4 * G] = -1
    a, b_ = int(blk_map[r1, 8 + 12 * G]), int(blk_map[r1, 8 + 13 * G])
    blk_map[r1, 8 + 12 * G: 8 + 13 * G] = b_
    blk_map[r1, 8 + 13 * G: 8 + 14 * G] = a
    blk_start[a], blk_start[b_] = blk_start[b_], blk_start[a]
    # row 2 stays regular
    shape = (B, Hq, max_len)
    got = run_gpu(bits, maps, max_len=max_len, G=G, block_seq=block_seq, score_shape=shape)
    ref = run_oracle(bits, maps, max_len=max_len, G=G, block_seq=block_seq, score_shape=shape)
    compare(got, ref, lens, block_seq)
    # skipped tokens keep the score buffer's fill value in both
    assert (got[2][0, :, 8 + 9 * G: 8 + 9 * G + 3] == np.float32(-1e20)).all()


def run_gpu_extra(inp_bits, maps, *, max_len, G, block_seq, score_shape=None):
    """The launch with three spare partial slots per row + the product's stage 2 told about them: merged bf16 output."""
    from sparse_vllm_amd.kernels.deltakv_kernels import full_layer_kivi_flash_decode_stage1
    from sparse_vllm_amd.kernels.flash_decoding_stage2 import flash_decode_stage2
    q = bf(inp_bits["q"])
    B, Hq, D = q.shape
    nblk = (max_len + block_seq - 1) // block_seq
    mid = torch.full((B, Hq, nblk + 3, D), 7.0, dtype=torch.float32, device=dev())
    lse = torch.full((B, Hq, nblk + 3), 7.0, dtype=torch.float32, device=dev())
    score = None if score_shape is None else torch.full(score_shape, -1e20, dtype=torch.float32, device=dev())
    extra = full_layer_kivi_flash_decode_stage1(
        q=q, raw_k=bf(inp_bits["raw_k"]), raw_v=bf(inp_bits["raw_v"]), raw_slots_map=t(maps["raw_map"]),
        kivi_block_slots_map=t(maps["blk_map"]), kivi_block_start_pos=t(maps["blk_start"]),
        key_packed=t(inp_bits["key_packed"]), key_scales=kparam(inp_bits["key_scales"]), key_mins=kparam(inp_bits["key_mins"]),
        value_packed=t(inp_bits["value_packed"]), value_scales=bf(inp_bits["value_scales"]),
        value_mins=bf(inp_bits["value_mins"]), req_indices=t(maps["req"]), context_lens=t(maps["lens"]),
        max_len_in_batch=max_len, mid_out=mid, mid_out_logsumexp=lse, group_size=G, block_seq=block_seq,
        attn_score=score, extra_partial_slots=3)
    o = torch.full((B, Hq, D), 7.0, dtype=torch.bfloat16, device=dev())
    flash_decode_stage2(mid, lse, t(maps["lens"]), o, block_seq, extra_partials=extra)
    torch.cuda.synchronize()
    return extra, o.float().cpu().numpy(), mid.cpu().numpy(), lse.cpu().numpy(), None if score is None else score.cpu().numpy()


@pytest.mark.parametrize("sink,raw_tail,lens,block_seq,irregular", [
    (8, 40, [1500, 777, 136, 8], 512, False),
    (32, 64, [1500, 1181], 256, False),
    (0, 0, [1024, 515], 256, False),
    (4, 21, [700, 300], 128, False),
    (8, 200, [1000, 420], 1024, False),
    (8, 600, [3000, 2100, 5], 1024, False),      # raw tail longer than the tail scan window
    (8, 40, [4100, 2048 + 8 + 40, 1100], 1024, False),
    (8, 40, [1500, 1100, 700], 256, True),
    (200, 40, [1500, 900], 128, False),          # raw head longer than the head scan window
])
def test_kivi_stage1_extra_partials(sink, raw_tail, lens, block_seq, irregular):
    """With three spare partial slots per row the wide launch hands the raw sink tokens, the raw residual tail and the
    ragged quantised piece in front of it to extra workgroups (partials nblk_row .. nblk_row + 2) and leaves the
    regular partials past a row's length unwritten; stage 2 merges the extras.  Same merged output (against the oracle's
    partials merged by the oracle, bf16 output tolerance) and the same position-indexed raw scores as the plain launch."""
    from oracle import decode_attention as oda
    rng = np.random.default_rng(sink * 1000 + raw_tail + block_seq)
    Hq, Hkv, D, G = 28, 4, 128, 32
    B = len(lens)
    bits, maps, max_len = make_case(rng, B=B, Hq=Hq, Hkv=Hkv, D=D, G=G, lens=lens, rows=B + 1, raw_tail=raw_tail,
                                     sink=sink, key_f32=True)
    if irregular:
        raw_map, blk_map, blk_start = maps["raw_map"], maps["blk_map"], maps["blk_start"]
        r0, r1 = int(maps["req"][0]), int(maps["req"][1])
        p = 8 + 5 * G + 11
        raw_map[r0, p] = int(raw_map.max()) - 3
        blk_map[r0, p] = -1
        blk_map[r1, 8 + 3 * G + 16: 8 + 4 * G] = -1
    shape = (B, Hq, max_len)
    extra, o, mid, lse, score = run_gpu_extra(bits, maps, max_len=max_len, G=G, block_seq=block_seq, score_shape=shape)
    assert extra == 3                                                           # this launch takes the wide kernel
    mid_r, lse_r, score_r = run_oracle(bits, maps, max_len=max_len, G=G, block_seq=block_seq, score_shape=shape)
    o_r = oda.flash_decode_stage2(mid_r, lse_r, np.asarray(lens, np.int32), block_seq)
    np.testing.assert_allclose(o, bf16_round(o_r), rtol=ATTN_TOL, atol=ATTN_TOL)
    np.testing.assert_allclose(score, score_r, rtol=1e-5, atol=1e-4)
    # the plain launch of the same inputs: same scores bit for bit (same tiles, only their workgroup differs)
    _, _, score_plain = run_gpu(bits, maps, max_len=max_len, G=G, block_seq=block_seq, score_shape=shape)
    for b, n in enumerate(lens):
        nb_row = (n + block_seq - 1) // block_seq
        if extra:
            assert np.isfinite(mid[b][:, :nb_row + 3]).all()                    # regular + extra partials written
            assert (mid[b][:, nb_row + 3:] == 7.0).all()                        # nothing past them
        np.testing.assert_array_equal(score[b, :, :n], score_plain[b, :, :n])
    # launches the wide kernel does not serve take no extras (bf16 key parameters here)
    bits16, maps16, ml16 = make_case(rng, B=1, Hq=Hq, Hkv=Hkv, D=D, G=G, lens=[300], rows=2, raw_tail=40, sink=8, key_f32=False)
    assert run_gpu_extra(bits16, maps16, max_len=ml16, G=G, block_seq=128)[0] == 0


@pytest.mark.parametrize("spare", [3, 0])
def test_kivi_stage1_fused_raw_store_equals_store_then_launch(spare):
    """The step's raw store of the layer inside the launch (`new_kv`): the workgroup that owns position len - 1 writes the
    new rows before it reads them.  Against store_kvcache followed by the plain launch: raw caches, partials, scores and
    merged outputs bit-identical, padded lanes (slot -1) untouched; with the extra workgroups and with the last-workgroup
    form."""
    from sparse_vllm_amd.kernels import store_kvcache
    from sparse_vllm_amd.kernels.deltakv_kernels import full_layer_kivi_flash_decode_stage1, kivi_fused_store_supported
    from sparse_vllm_amd.kernels.flash_decoding_stage2 import flash_decode_stage2
    rng = np.random.default_rng(77 + spare)
    Hq, Hkv, D, G, block_seq = 28, 4, 128, 32, 256
    lens = [1500, 777, 136, 9]
    B = len(lens)
    bits, maps, max_len = make_case(rng, B=B, Hq=Hq, Hkv=Hkv, D=D, G=G, lens=lens, rows=B + 1, raw_tail=40, sink=8, key_f32=True)
    assert kivi_fused_store_supported(head_dim=D, num_kv_heads=Hkv, group_size=G, block_seq=block_seq, key_param_dtype=torch.float32)
    new_k = bf(f32_to_bf16_bits(rng.standard_normal((B, Hkv, D)).astype(np.float32)))
    new_v = bf(f32_to_bf16_bits(rng.standard_normal((B, Hkv, D)).astype(np.float32)))
    # the newest token of row b lives in raw slot raw_map[req[b], len - 1]; lane 3 is a padded graph lane (slot -1)
    slots_np = np.array([maps["raw_map"][maps["req"][b], lens[b] - 1] for b in range(B)], np.int32)
    slots_np[3] = -1
    slots = t(slots_np)

    def launch(fused):
        raw_k, raw_v = bf(bits["raw_k"]), bf(bits["raw_v"])
        if not fused:
            store_kvcache(new_k, new_v, raw_k, raw_v, slots)
        nblk = (max_len + block_seq - 1) // block_seq
        mid = torch.full((B, Hq, nblk + spare, D), 7.0, dtype=torch.float32, device=dev())
        lse = torch.full((B, Hq, nblk + spare), 7.0, dtype=torch.float32, device=dev())
        score = torch.full((B, Hq, max_len), -1e20, dtype=torch.float32, device=dev())
        extra = full_layer_kivi_flash_decode_stage1(
            q=bf(bits["q"]), raw_k=raw_k, raw_v=raw_v, raw_slots_map=t(maps["raw_map"]), kivi_block_slots_map=t(maps["blk_map"]),
            kivi_block_start_pos=t(maps["blk_start"]), key_packed=t(bits["key_packed"]), key_scales=kparam(bits["key_scales"]),
            key_mins=kparam(bits["key_mins"]), value_packed=t(bits["value_packed"]), value_scales=bf(bits["value_scales"]),
            value_mins=bf(bits["value_mins"]), req_indices=t(maps["req"]), context_lens=t(maps["lens"]), max_len_in_batch=max_len,
            mid_out=mid, mid_out_logsumexp=lse, group_size=G, block_seq=block_seq, attn_score=score, extra_partial_slots=spare,
            new_kv=(new_k, new_v, slots) if fused else None)
        o = torch.full((B, Hq, D), 7.0, dtype=torch.bfloat16, device=dev())
        flash_decode_stage2(mid, lse, t(maps["lens"]), o, block_seq, extra_partials=extra)
        torch.cuda.synchronize()
        return raw_k, raw_v, mid, lse, score, o

    ref, got = launch(False), launch(True)
    for x, y in zip(ref, got):
        assert torch.equal(x.view(torch.int16) if x.dtype == torch.bfloat16 else x, y.view(torch.int16) if y.dtype == torch.bfloat16 else y)
    # the new rows are in the cache, and they are what the launch saw (the output changes when the new value changes)
    assert torch.equal(got[0][int(slots_np[0])], new_k[0]) and torch.equal(got[1][int(slots_np[1])], new_v[1])


def test_kivi_stage1_all_raw_matches_plain_stage1():
    """With no KIVI block the kernel is the ordinary slot-table decode: compare with svk_flash_decode_stage1.
    The two kernels tile the row differently (128 vs 32 tokens per online-softmax step), so P is rounded to bf16
    against different running maxima: partials agree to bf16 precision of P (2^-9 relative per term), not bitwise."""
    from sparse_vllm_amd.kernels.gqa_flash_decoding_stage1 import flash_decode_stage1
    rng = np.random.default_rng(5)
    B, Hq, Hkv, D, G, block_seq = 3, 28, 4, 128, 32, 64
    lens = [200, 33, 128]
    bits, maps, max_len = make_case(rng, B=B, Hq=Hq, Hkv=Hkv, D=D, G=G, lens=lens, rows=3, raw_tail=10**6, sink=0)
    assert (maps["blk_map"] < 0).all()
    mid, lse, _ = run_gpu(bits, maps, max_len=max_len, G=G, block_seq=block_seq)
    nblk = (max_len + block_seq - 1) // block_seq
    mid2 = torch.zeros((B, Hq, nblk, D), dtype=torch.float32, device=dev())
    lse2 = torch.zeros((B, Hq, nblk), dtype=torch.float32, device=dev())
    flash_decode_stage1(bf(bits["q"]), bf(bits["raw_k"]), bf(bits["raw_v"]), t(maps["raw_map"]), t(maps["req"]),
                        t(maps["lens"]), max_len, mid2, lse2, block_seq)
    torch.cuda.synchronize()
    vb = valid_blocks(lens, block_seq, nblk)
    for b in range(B):
        np.testing.assert_allclose(mid[b][:, vb[b]], mid2.cpu().numpy()[b][:, vb[b]], rtol=5e-3, atol=5e-3)
        np.testing.assert_allclose(lse[b][:, vb[b]], lse2.cpu().numpy()[b][:, vb[b]], rtol=1e-6, atol=1e-6)


def test_kivi_stage1_validation():
    from sparse_vllm_amd.kernels.deltakv_kernels import full_layer_kivi_flash_decode_stage1
    rng = np.random.default_rng(1)
    bits, maps, max_len = make_case(rng, B=1, Hq=8, Hkv=2, D=64, G=32, lens=[80], rows=1, raw_tail=8, sink=8)
    kw = dict(q=bf(bits["q"]), raw_k=bf(bits["raw_k"]), raw_v=bf(bits["raw_v"]), raw_slots_map=t(maps["raw_map"]),
              kivi_block_slots_map=t(maps["blk_map"]), kivi_block_start_pos=t(maps["blk_start"]),
              key_packed=t(bits["key_packed"]), key_scales=bf(bits["key_scales"]), key_mins=bf(bits["key_mins"]),
              value_packed=t(bits["value_packed"]), value_scales=bf(bits["value_scales"]), value_mins=bf(bits["value_mins"]),
              req_indices=t(maps["req"]), context_lens=t(maps["lens"]), max_len_in_batch=max_len,
              mid_out=torch.zeros((1, 8, 2, 64), device=dev()), mid_out_logsumexp=torch.zeros((1, 8, 2), device=dev()),
              group_size=32, block_seq=64)
    with pytest.raises(ValueError, match="block_seq must be a positive multiple of 16"):
        full_layer_kivi_flash_decode_stage1(**{**kw, "block_seq": 40})
    with pytest.raises(ValueError, match="Invalid KIVI group_size"):
        full_layer_kivi_flash_decode_stage1(**{**kw, "group_size": 48})
    with pytest.raises(ValueError, match="exceeds map width"):
        full_layer_kivi_flash_decode_stage1(**{**kw, "max_len_in_batch": 4096})
    with pytest.raises(ValueError, match="rank-3 attention scores only"):
        full_layer_kivi_flash_decode_stage1(**{**kw, "attn_score": torch.zeros((1, 80), device=dev())})
    full_layer_kivi_flash_decode_stage1(**{**kw, "max_len_in_batch": 0})      # no-op like the reference
