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
