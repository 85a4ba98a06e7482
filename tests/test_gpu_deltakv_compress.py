"""GPU parity of the DeltaKV compression-side kernels through the C ABI.

Integer results (codes, father sets given the scores) are bit-exact against the oracle / the reference fixtures; scale and
min are values of the storage dtype and must match exactly as well (all arithmetic is element-wise and ordered)."""

import numpy as np
import pytest

from oracle import bf16_bits_to_f32, bf16_round, f32_to_bf16_bits
from oracle import deltakv_compress as oc
from oracle.quest import check_topk_set

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev():
    return torch.device("cuda:0")


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev())


def bf(x_f32):
    return torch.from_numpy(f32_to_bf16_bits(x_f32).view(np.int16).copy()).to(dev()).view(torch.bfloat16)


def test_quantize_pack_fixtures_and_bf16(golden):
    from sparse_vllm_amd.kernels.deltakv_kernels import triton_quantize_and_pack_2d_int4_grouped as qp
    g = golden("deltakv_compress")
    for tag, dt in (("f32", torch.float32), ("f16", torch.float16)):
        code, scale, mn = qp(t(g[f"q2d_{tag}_data"]).to(dt), 16)
        np.testing.assert_array_equal(code.cpu().numpy(), g[f"q2d_{tag}_code"])
        np.testing.assert_array_equal(scale.float().cpu().numpy(), g[f"q2d_{tag}_scale"])
        np.testing.assert_array_equal(mn.float().cpu().numpy(), g[f"q2d_{tag}_mn"])
    rng = np.random.default_rng(0)
    x = bf16_round((rng.standard_normal((300, 256)) * 0.4).astype(np.float32))
    x[7] = 0.125
    code, scale, mn = qp(bf(x), 32)
    rc, rs, rm = oc.quantize_pack_2d(x, 32, 4, oc.rounding("bf16"))
    np.testing.assert_array_equal(code.cpu().numpy(), rc)
    np.testing.assert_array_equal(scale.float().cpu().numpy(), rs)
    np.testing.assert_array_equal(mn.float().cpu().numpy(), rm)
    # scattered store into caller-owned caches (the fused latent-cache update)
    dst = torch.from_numpy(rng.permutation(500)[:300].astype(np.int32)).to(dev())
    cc = torch.zeros((500, 32), dtype=torch.int32, device=dev())
    cs = torch.zeros((500, 8), dtype=torch.bfloat16, device=dev())
    cm = torch.zeros_like(cs)
    qp(bf(x), 32, out=(cc, cs, cm), dst_rows=dst)
    torch.cuda.synchronize()
    assert torch.equal(cc[dst.long()], code) and torch.equal(cs[dst.long()], scale) and torch.equal(cm[dst.long()], mn)
    untouched = torch.ones(500, dtype=torch.bool, device=dev())
    untouched[dst.long()] = False
    assert not cc[untouched].any()
    with pytest.raises(ValueError, match="positive multiple of 8"):
        qp(bf(x), 12)
    with pytest.raises(ValueError, match="divisible by group_size"):
        qp(bf(x), 48)


@pytest.mark.parametrize("H,D,G,key_f32", [(4, 128, 32, True), (2, 64, 32, False), (8, 128, 32, True)])
def test_kivi_store_blocks_vs_oracle(H, D, G, key_f32):
    from sparse_vllm_amd.kernels.deltakv_kernels import kivi_store_blocks
    rng = np.random.default_rng(H + D)
    slots, blocks, nblk_total = 700, 9, 20
    k = bf16_round((rng.standard_normal((slots, H, D)) * 0.5).astype(np.float32))
    v = bf16_round((rng.standard_normal((slots, H, D)) * 0.5).astype(np.float32))
    k[5] = 0.0
    raw = rng.permutation(slots)[: blocks * G].reshape(blocks, G).astype(np.int32)
    dst = rng.permutation(nblk_total)[:blocks].astype(np.int32)
    z = lambda *shape, dt=torch.int32: torch.zeros(shape, dtype=dt, device=dev())
    out = dict(key_packed=z(nblk_total, H, D, G // 8), key_scales=z(nblk_total, H, D, dt=torch.float32 if key_f32 else torch.bfloat16),
               key_mins=z(nblk_total, H, D, dt=torch.float32 if key_f32 else torch.bfloat16), value_packed=z(nblk_total, H, G, D // 8),
               value_scales=z(nblk_total, H, G, D // G, dt=torch.bfloat16), value_mins=z(nblk_total, H, G, D // G, dt=torch.bfloat16))
    kivi_store_blocks(k_cache=bf(k), v_cache=bf(v), raw_slots=t(raw), block_slots=t(dst), group_size=G, **out)
    torch.cuda.synchronize()
    ref = oc.kivi_quantize_blocks(k[raw], v[raw], G, oc.rounding("bf16"))
    for name, r in ref.items():
        got = out[name][t(dst).long()]
        got = got.cpu().numpy() if got.dtype == torch.int32 else got.float().cpu().numpy()
        np.testing.assert_array_equal(got, r, err_msg=name)
    # and the stored blocks decode back to within one quantisation step through the decode kernel's formula
    deq = ref["key_scales"][..., None] * 15 + ref["key_mins"][..., None]
    assert np.isfinite(deq).all()


@pytest.mark.parametrize("tag", ["cc", "ci"])
def test_cluster_topk_and_gather_mean(golden, tag):
    """`cc`: regular centre stride; `ci`: irregular centre positions [0, 5, 11, 18, 27, 35] through the reference's
    `_cluster_compress` (dynamic-stride form of BASELINE configs[4])."""
    from sparse_vllm_amd.kernels.deltakv_kernels import cluster_topk, gather_mean_fathers
    g0 = golden("deltakv_compress")
    g = {k_[3:]: g0[k_] for k_ in g0.files if k_.startswith(tag + "_")}
    g = {"cc_" + k_: v_ for k_, v_ in g.items()}
    k = int(g["cc_k"][0])
    ck, cv = bf16_bits_to_f32(g["cc_cache_k"]), bf16_bits_to_f32(g["cc_cache_v"])
    cache = np.concatenate((ck.reshape(64, -1), cv.reshape(64, -1)), axis=1)
    kv = bf16_bits_to_f32(g["cc_kv"])
    scores, topk_ref, base_ref, allc = oc.cluster_compress(kv, cache[g["cc_existing"]], g["cc_rel"], k)
    # scores as the library GEMM path would hand them over: bf16 [n, m] before masking
    raw_scores = oc.l2_scores(kv, allc, oc.rounding("bf16"))
    m0 = len(g["cc_existing"])
    got = cluster_topk(bf(raw_scores), m0=m0, new_center_rel=t(g["cc_rel"]), k=k).cpu().numpy()
    np.testing.assert_array_equal(got, topk_ref)                   # same scores -> same (score desc, index asc) order
    for r in range(kv.shape[0]):
        check_topk_set(scores[r], g["cc_topk"][r], k, atol=float(np.abs(scores[r][np.isfinite(scores[r])]).max() * 2.0 ** -7))
    # gather-mean: the new centres of the fixture are rows of `kv`, which the manager reads from the cache by slot:
    # put them into spare cache rows
    new_slots = np.arange(50, 50 + len(g["cc_rel"]), dtype=np.int32)
    H, D = ck.shape[1], ck.shape[2]
    ck2, cv2 = ck.copy(), cv.copy()
    ck2[new_slots] = kv[g["cc_rel"], : H * D].reshape(-1, H, D)
    cv2[new_slots] = kv[g["cc_rel"], H * D:].reshape(-1, H, D)
    center_slots = np.concatenate((g["cc_existing"], new_slots)).astype(np.int32)
    base, fathers = gather_mean_fathers(bf(ck2), bf(cv2), t(center_slots), t(got), k_out=5)
    np.testing.assert_array_equal(fathers.cpu().numpy()[:, :k], center_slots[got])
    np.testing.assert_array_equal(fathers.cpu().numpy()[:, k:], np.repeat(center_slots[got][:, :1], 5 - k, axis=1))
    np.testing.assert_allclose(base.float().cpu().numpy(), base_ref, rtol=2 ** -7, atol=1e-6)
    np.testing.assert_allclose(base.float().cpu().numpy(), bf16_bits_to_f32(g["cc_base"]), rtol=2 ** -6, atol=1e-5)
    # wide rows, ties and the all-masked corner
    rng = np.random.default_rng(2)
    n, m0, mn_ = 128, 4000, 4
    sc = bf16_round(rng.standard_normal((n, m0 + mn_)).astype(np.float32))
    sc[:, 10:20] = 3.0                                              # ties -> lower column first
    rel = np.array([0, 33, 66, 99], np.int32)
    got = cluster_topk(bf(sc), m0=m0, new_center_rel=t(rel), k=4, row_offset=0).cpu().numpy()
    masked = sc.copy()
    masked[:, m0:] = np.where(rel[None, :] <= np.arange(n)[:, None], sc[:, m0:], -np.inf)
    np.testing.assert_array_equal(got, np.argsort(-masked, axis=1, kind="stable")[:, :4])


def _cluster_case(rng, *, rows, m0, rel, H=4, D=128, slots=None):
    """Centre rows live in the layer caches (as in the manager): existing centres in random slots, the block's own centres =
    rows `rel` of the token block stored in further slots."""
    m_new = len(rel)
    slots = slots or (m0 + m_new + 50)
    ck = bf16_round((rng.standard_normal((slots, H, D)) * 0.5).astype(np.float32))
    cv = bf16_round((rng.standard_normal((slots, H, D)) * 0.5).astype(np.float32))
    perm = rng.permutation(slots).astype(np.int32)
    center_slots = perm[: m0 + m_new].copy()
    kv = bf16_round((rng.standard_normal((rows, 2 * H * D)) * 0.5).astype(np.float32))
    # tokens near some centre, so that the ranking has structure (and near-ties at the boundary)
    near = rng.integers(0, max(m0, 1), rows)
    if m0:
        cat = np.concatenate((ck[center_slots[:m0]].reshape(m0, -1), cv[center_slots[:m0]].reshape(m0, -1)), axis=1)
        kv = bf16_round(0.7 * cat[near] + 0.3 * kv)
    for j, r in enumerate(rel):
        ck[center_slots[m0 + j]] = kv[r, : H * D].reshape(H, D)
        cv[center_slots[m0 + j]] = kv[r, H * D:].reshape(H, D)
    existing = np.concatenate((ck[center_slots[:m0]].reshape(m0, H * D), cv[center_slots[:m0]].reshape(m0, H * D)), axis=1)
    return kv, ck, cv, center_slots, existing


@pytest.mark.parametrize("cfg", [
    dict(rows=128, m0=4000, rel=[0, 33, 66, 99], k=4),                       # a decode-time eviction at the paper shape: centre splits
    dict(rows=128, m0=8, rel=list(range(0, 128, 33)), k=4),                  # a row's first eviction: sink centres only
    dict(rows=300, m0=0, rel=list(range(0, 300, 10)), k=4),                  # no existing centres: early tokens see fewer than k
    dict(rows=2048, m0=40, rel=list(range(0, 2048, 97)), k=8),               # one split, k = 8
    dict(rows=77, m0=513, rel=[0, 5, 11, 18, 27, 35, 76], k=3),              # ragged rows / tiles, k below the kernel's list
    dict(rows=64, m0=300, rel=[0, 31], k=4, H=2, D=64),                      # kv_dim 256
])
def test_fused_cluster_l2_topk_vs_oracle(cfg):
    """svk_cluster_l2_topk (ranking product + causal mask + top-k in one MFMA launch, no [rows, m] matrix) against
    oracle.deltakv_compress.cluster_compress (numpy restatement of `_cluster_compress`, deltakv_less_memory.py:2719-2802,
    `_metric_l2` deltakv_base.py:2168-2190).  The two differ by the fp32 summation order inside the dot product and the
    norm, i.e. a score may sit one bf16 ulp away: every row's result must be a valid top-k of the oracle's scores within
    that ulp (ties and near-ties at the boundary may go either way), nearly all rows must be identical, and the launch must
    agree the same way with this build's library-GEMM path (`torch.matmul` + svk_cluster_topk)."""
    from sparse_vllm_amd.kernels import deltakv_kernels as dk
    rows, m0, rel, k = cfg["rows"], cfg["m0"], np.asarray(cfg["rel"], np.int32), cfg["k"]
    H, D = cfg.get("H", 4), cfg.get("D", 128)
    rng = np.random.default_rng(rows + m0)
    kv, ck, cv, center_slots, existing = _cluster_case(rng, rows=rows, m0=m0, rel=rel, H=H, D=D)
    assert dk.cluster_l2_topk_supported(num_kv_heads=H, head_dim=D, dtype=torch.bfloat16)
    got = dk.cluster_l2_topk(bf(kv), bf(ck), bf(cv), t(center_slots), m0=m0, new_center_rel=t(rel), k=k).cpu().numpy()
    scores, topk_ref, _, allc = oc.cluster_compress(kv, existing, rel, k)
    assert got.shape == topk_ref.shape == (rows, k)
    same = 0
    for r in range(rows):
        finite = scores[r][np.isfinite(scores[r])]
        n_valid = int(np.isfinite(scores[r]).sum())
        if n_valid >= k:
            check_topk_set(scores[r], got[r], k, atol=float(np.abs(finite).max() * 2.0 ** -7))
            assert np.isfinite(scores[r][got[r]]).all()
        else:
            # fewer visible centres than k: the visible ones first, then masked columns by ascending index (svk_cluster_topk's order)
            assert set(got[r][:n_valid]) == set(np.nonzero(np.isfinite(scores[r]))[0])
            np.testing.assert_array_equal(got[r][n_valid:], np.nonzero(~np.isfinite(scores[r]))[0][: k - n_valid])
        same += int(np.array_equal(got[r], topk_ref[r]))
    assert same >= 0.97 * rows, (same, rows)
    # the library path of this build on the same inputs
    centers = torch.cat((bf(ck)[t(center_slots).long()].reshape(m0 + len(rel), -1), bf(cv)[t(center_slots).long()].reshape(m0 + len(rel), -1)), dim=1)
    dot = torch.matmul(bf(kv), centers.t())
    lib_scores = dot.mul(2.0).sub_((centers * centers).sum(dim=1, dtype=torch.float32).to(dot.dtype).unsqueeze(0))
    lib = dk.cluster_topk(lib_scores, m0=m0, new_center_rel=t(rel), k=k).cpu().numpy()
    assert (lib == got).all(axis=1).mean() >= 0.97


@pytest.mark.parametrize("tag", ["cc", "ci"])
def test_fused_cluster_l2_topk_reference_fixtures(golden, tag):
    """The reference's `_cluster_compress` fixtures (regular and irregular centre positions) through the fused launch: the
    father sets are valid top-k sets of the oracle's scores within one bf16 ulp, as for the reference's own output."""
    from sparse_vllm_amd.kernels import deltakv_kernels as dk
    g0 = golden("deltakv_compress")
    g = {k_[3:]: g0[k_] for k_ in g0.files if k_.startswith(tag + "_")}
    k = int(g["k"][0])
    ck, cv = bf16_bits_to_f32(g["cache_k"]), bf16_bits_to_f32(g["cache_v"])
    kv = bf16_bits_to_f32(g["kv"])
    H, D = ck.shape[1], ck.shape[2]
    if not dk.cluster_l2_topk_supported(num_kv_heads=H, head_dim=D, dtype=torch.bfloat16):
        pytest.skip("fixture shape is not served by the fused launch")
    rel = g["rel"].astype(np.int32)
    new_slots = np.arange(50, 50 + len(rel), dtype=np.int32)
    ck2, cv2 = ck.copy(), cv.copy()
    ck2[new_slots] = kv[rel, : H * D].reshape(-1, H, D)
    cv2[new_slots] = kv[rel, H * D:].reshape(-1, H, D)
    center_slots = np.concatenate((g["existing"], new_slots)).astype(np.int32)
    cache = np.concatenate((ck.reshape(ck.shape[0], -1), cv.reshape(cv.shape[0], -1)), axis=1)
    scores, topk_ref, _, _ = oc.cluster_compress(kv, cache[g["existing"]], rel, k)
    got = dk.cluster_l2_topk(bf(kv), bf(ck2), bf(cv2), t(center_slots), m0=len(g["existing"]), new_center_rel=t(rel), k=k).cpu().numpy()
    for r in range(kv.shape[0]):
        tol = float(np.abs(scores[r][np.isfinite(scores[r])]).max() * 2.0 ** -7)
        if int(np.isfinite(scores[r]).sum()) >= k:
            check_topk_set(scores[r], got[r], k, atol=tol)
            check_topk_set(scores[r], g["topk"][r], k, atol=tol)          # the reference's own choice passes the same check
    assert (got == topk_ref).all(axis=1).mean() >= 0.9
