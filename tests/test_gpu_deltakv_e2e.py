"""End-to-end DeltaKV decode on the GPU through the reference's operator surface (CacheManager.create ->
prepare_decode_static -> Attention.forward per layer -> SparseController.on_layer_end) against the numpy oracle
chained step by step on a mirror of the device state:

  full (observation) layers : KIVI-int4 or raw decode stage 1 with raw 3-D scores -> stage 2
  query-aware top-k         : per-head softmax over the compressed range, max over heads, bf16, sorted top-k
  sparse layers             : static plan (bit-exact) -> latent dequant -> compress_up -> reconstruct + RoPE ->
                              attention view (RoPE on the fly) -> decode attention

Tolerances: attention outputs rtol = atol = 2e-2 (the reference's bar for decode partials); token scores 1 bf16 ulp;
the GPU's top-k must be a valid top-k set of the oracle's scores up to that ulp (ties are order-free)."""

import numpy as np
import pytest

from oracle import bf16_bits_to_f32, bf16_round
from oracle import decode_attention as oda
from oracle import deltakv as od
from oracle import kivi as ok
from oracle.quest import check_topk_set

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

TOL = 2e-2


def f32(t):
    return t.float().cpu().numpy()


def linear_up(mod, x):
    """compress_up on bf16-valued fp32 arrays (fp32 accumulation, bf16 outputs like the torch bf16 modules)."""
    def lin(layer, v):
        y = v @ f32(layer.weight).T
        if layer.bias is not None:
            y = y + f32(layer.bias)
        return bf16_round(y.astype(np.float32))
    if isinstance(mod, torch.nn.Linear):
        return lin(mod, x)
    h = lin(mod[0], x)
    h = bf16_round((0.5 * h * (1.0 + np.vectorize(__import__("math").erf)(h / np.sqrt(2.0)))).astype(np.float32))
    return lin(mod[2], h)


@pytest.mark.parametrize("cfg", [
    dict(layers=5, full="0,3", kivi=True, up="linear", bits=4, lens=[148, 92, 61], Hq=8, Hkv=2, D=64),
    dict(layers=6, full="0,1,4", kivi=False, up="mlp_gelu", bits=0, lens=[116, 44], Hq=28, Hkv=4, D=128),
    # the published form: int4 latents through the two-Linear GELU compress_up - the fused residual load, the look-ahead
    # reconstruction in groups of two layers on the side stream and the reconstruction straight into the layers' views
    dict(layers=8, full="0,4", kivi=True, up="mlp_gelu", bits=4, lens=[148, 92], Hq=8, Hkv=2, D=64),
    # BASELINE configs[4]'s head shape (Qwen2.5-7B: 28 / 4 heads of 128), int4 latents, two-Linear GELU compress_up with a
    # hidden width the fused launch serves (a multiple of 64), four fathers, look-ahead on: the DEFAULT reconstruction here
    # is `svk_deltakv_up_reconstruct` (second Linear + reconstruction in one launch) and the test asserts that it ran
    dict(layers=7, full="0,3", kivi=True, up="mlp_gelu", bits=4, lens=[180, 97], Hq=28, Hkv=4, D=128, latent=64, group=32,
         inter=128, Kf=4, keep=40, fused=True),
])
def test_deltakv_decode_steps_match_oracle(cfg, monkeypatch):
    from sparse_vllm_amd.config import Config
    from sparse_vllm_amd.kernels import deltakv_kernels as dk
    from tools.synthetic import SyntheticDecodeDriver as SparseDecodeDriver
    B, L = len(cfg["lens"]), cfg["layers"]
    sink, recent, keep, Kf = 4, 8, cfg.get("keep", 12), cfg.get("Kf", 2)
    group = cfg.get("group", 16)
    fused_calls = {"n": 0, "layers": 0}
    fused_launch = dk.deltakv_up_reconstruct_layers

    def counted(hidden, *a, **k):
        fused_calls["n"] += 1
        fused_calls["layers"] += int(hidden.shape[0])
        return fused_launch(hidden, *a, **k)
    monkeypatch.setattr(dk, "deltakv_up_reconstruct_layers", counted)
    monkeypatch.delenv("SVK_DELTAKV_FUSED_UP", raising=False)       # the defaults are what is under test
    monkeypatch.delenv("SVK_DELTAKV_RECON_AHEAD", raising=False)
    conf = Config.from_kwargs(
        sparse_method="deltakv", num_hidden_layers=L, full_attention_layers=cfg["full"], num_attention_heads=cfg["Hq"],
        num_key_value_heads=cfg["Hkv"], head_dim=cfg["D"], max_model_len=256, max_num_seqs_in_gpu=B + 1,
        sink_keep_tokens=sink, recent_keep_tokens=recent, decode_keep_tokens=keep, deltakv_neighbor_count=Kf,
        deltakv_latent_dim=cfg.get("latent", 32), deltakv_latent_quant_bits=cfg["bits"],
        deltakv_latent_quant_group_size=group,
        deltakv_center_ratio=0.25, allow_missing_deltakv_path=True, compressor_up_type=cfg["up"],
        compressor_intermediate_size=cfg.get("inter", 48), full_layer_kv_quant_bits=4 if cfg["kivi"] else 0,
        full_layer_kivi_decode_block_seq=64, rope_theta=10000.0)
    drv = SparseDecodeDriver(conf)
    cm, sc = drv.cache_manager, drv.sparse_controller
    cm.permute_free_slots(7)
    seqs = drv.admit_compressed_rows(B, cfg["lens"], seed=3)
    rows = np.array([cm.seq_id_to_row[s.seq_id] for s in seqs])
    Hq, Hkv, D = cfg["Hq"], cfg["Hkv"], cfg["D"]
    scale = D ** -0.5
    max_buffer = cm._deltakv_decode_static_max_buffer()
    assert max_buffer == 2 * recent

    outs = torch.zeros((L, B, Hq, D), dtype=torch.bfloat16, device=drv.device)
    n_evictions = 0
    for step in range(cfg.get("steps", 28)):    # the raw tail cycles recent .. 2*recent: a compression every `recent` steps
        # ---- mirror of the device state before the step (compression at the end of the previous step changed it)
        full_k, full_v = f32(cm.full_kv_cache[0]).copy(), f32(cm.full_kv_cache[1]).copy()
        sp_k, sp_v = f32(cm.deltakv_full_kv_cache[0]).copy(), f32(cm.deltakv_full_kv_cache[1]).copy()
        cos_sin = cm.cos_sin_cache.cpu().numpy()
        lat_map = cm.sparse_layer_latent_slots_map.cpu().numpy()
        fathers = cm.deltakv_latent_to_full_slots.cpu().numpy()
        if cfg["bits"]:
            lat_code = cm.deltakv_latent_cache.cpu().numpy()
            lat_scale, lat_mn = f32(cm.deltakv_latent_scales), f32(cm.deltakv_latent_mins)
        else:
            lat_dense = f32(cm.deltakv_latent_cache)
        clens = cm.row_deltakv_compressed_lens[rows].copy()
        if cfg["kivi"]:
            kv = {n: getattr(cm, f"full_layer_kivi_{n}") for n in ("key_packed", "key_scales", "key_mins", "value_packed",
                                                                    "value_scales", "value_mins")}
            kv = {n: (t.cpu().numpy() if t.dtype in (torch.int32, torch.float32) else f32(t)) for n, t in kv.items()}
            blk_map = cm.full_layer_kivi_block_slots_map.cpu().numpy()
            blk_start = cm.full_layer_kivi_block_start_pos.cpu().numpy()
        clens_before = clens.copy()
        full_map = cm.full_layer_slots_map.cpu().numpy()
        raw_map = cm.sparse_layer_raw_slots_map.cpu().numpy()
        slot_to_pos = cm.deltakv_slot_to_pos.cpu().numpy()
        q, k, v = drv.random_step_inputs(seed=50 + step)
        drv.step(q, k, v, outputs=outs)
        torch.cuda.synchronize()
        got = f32(outs)
        qf, kf, vf = f32(q), f32(k), f32(v)
        lens = cm.row_seq_lens[rows].copy()
        # maps as the attention saw them = the pre-step maps + this step's new token (the newest token is never
        # compressed in the same step, so its slots are still in the post-step maps)
        new_full = cm.full_layer_slots_map.cpu().numpy()[rows, lens - 1]
        new_sparse = cm.sparse_layer_raw_slots_map.cpu().numpy()[rows, lens - 1]
        assert (new_full >= 0).all() and (new_sparse >= 0).all()
        np.testing.assert_array_equal(cm.deltakv_slot_to_pos.cpu().numpy()[new_sparse], lens - 1)
        full_map[rows, lens - 1] = new_full
        raw_map[rows, lens - 1] = new_sparse
        slot_to_pos[new_sparse] = lens - 1
        active = None
        for l in range(L):
            if l in cm.full_layer_to_idx:
                i = cm.full_layer_to_idx[l]
                full_k[i][new_full], full_v[i][new_full] = kf[l], vf[l]
                max_len = int(lens.max())
                score = np.full((B, Hq, max_len), -1e20, np.float32)
                if cfg["kivi"]:
                    mid, lse = ok.full_layer_kivi_flash_decode_stage1(
                        q=qf[l], raw_k=full_k[i], raw_v=full_v[i], raw_slots_map=full_map, kivi_block_slots_map=blk_map,
                        kivi_block_start_pos=blk_start, key_packed=kv["key_packed"][i], key_scales=kv["key_scales"][i],
                        key_mins=kv["key_mins"][i], value_packed=kv["value_packed"][i], value_scales=kv["value_scales"][i],
                        value_mins=kv["value_mins"][i], req_indices=rows, context_lens=lens, max_len_in_batch=max_len,
                        group_size=32, block_seq=64, attn_score=score)
                else:
                    mid, lse = oda.flash_decode_stage1(qf[l], full_k[i], full_v[i], full_map, rows.astype(np.int32),
                                                       lens.astype(np.int32), max_len, 64, attn_score=score)
                ref = oda.flash_decode_stage2(mid, lse, lens.astype(np.int32), 64)
                np.testing.assert_allclose(got[l], bf16_round(ref), rtol=TOL, atol=TOL, err_msg=f"full layer {l} step {step}")
                if l in sc.obs_layer_ids:
                    tok = bf16_round(od.decode_softmax_token_scores(score, sink=sink, compressed_lens=clens, scale=scale))
                    st = sc.layer_batch_sparse_states[l + 1]
                    active = st.active_compressed_indices.cpu().numpy()
                    k_max = min(keep, int(lens.max()) - sink)
                    assert active.shape == (B, k_max)
                    gpu_tok = f32(sc.layer_batch_sparse_states[l].attn_score)[:, sink:]
                    for b in range(B):
                        c = int(clens[b])
                        np.testing.assert_allclose(gpu_tok[b, :c], tok[b, :c], rtol=2 ** -7, atol=1e-30)
                        n_valid = min(c, k_max)
                        if n_valid:
                            # a valid top-k of the GPU's own (bf16) scores; sorted descending, ties by position
                            check_topk_set(gpu_tok[b, :c], active[b, :n_valid], n_valid)
                            vals = gpu_tok[b, active[b, :n_valid]]
                            assert np.all(np.diff(vals) <= 0)
                continue
            # ---- sparse layer
            i = cm.deltakv_layer_to_idx[l]
            sp_k[i][new_sparse], sp_v[i][new_sparse] = kf[l], vf[l]
            assert active is not None
            temp = cm._temp_slots_by_shape[(B, active.shape[1])].cpu().numpy()
            plan = od.static_decode_plan(raw_map, lat_map, active, rows, lens, clens, temp, sink=sink, max_buffer=max_buffer)
            rl, ro, rp = plan["recon_latent"], plan["recon_out_slot"], plan["recon_pos"]
            need = rl >= 0
            safe_l = np.maximum(rl, 0)
            if cfg["bits"]:
                x = bf16_round(od.dequantize_grouped(lat_code[i][safe_l], lat_scale[i][safe_l], lat_mn[i][safe_l], group, 4))
            else:
                x = lat_dense[i][safe_l]
            delta = linear_up(cm.compress_up[i], x)
            od.reconstruct_writeback(sp_k[i], sp_v[i], father_slots=np.maximum(fathers[i][safe_l], 0), slot_to_pos=slot_to_pos,
                                     out_slots=ro, out_pos=rp, cos_sin=cos_sin, delta=delta, raw_k_cache=True)
            post = np.zeros(sp_k[i].shape[0], bool)
            post[ro[need]] = True
            vk, vv = od.materialize_sparse_view(plan["active_slots"], slot_to_pos, sp_k[i], sp_v[i], cos_sin, postrope_mask=post)
            W = plan["active_slots"].shape[1]
            table = np.arange(B * W, dtype=np.int32).reshape(B, W)
            ref, _ = oda.decode_attention_dense(qf[l], vk, vv, table, np.arange(B, dtype=np.int32), plan["new_context_lens"])
            np.testing.assert_allclose(got[l], bf16_round(ref), rtol=TOL, atol=TOL, err_msg=f"sparse layer {l} step {step}")
        # ---- the last observation group's plan is still in the device buffers: bit-exact
        S = sink + active.shape[1] + max_buffer
        a_slots, a_pos, _, new_len, _, r_pos, r_lat, r_out = (t.cpu().numpy() for t in cm._plan_buffers[(B, active.shape[1], S)])
        np.testing.assert_array_equal(a_slots, plan["active_slots"])
        np.testing.assert_array_equal(a_pos, plan["active_pos"])
        np.testing.assert_array_equal(new_len, plan["new_context_lens"])
        np.testing.assert_array_equal(r_pos, plan["recon_pos"])
        np.testing.assert_array_equal(r_lat, plan["recon_latent"])
        np.testing.assert_array_equal(r_out, plan["recon_out_slot"])

        # ---- compression side bookkeeping after the step (deltakv_evict / KIVI evict ran in post_forward)
        after = cm.row_deltakv_compressed_lens[rows]
        n_evictions += int((after != clens_before).sum())
        tails = cm.row_seq_lens[rows] - sink - after
        assert ((tails >= min(recent, int(tails.max()))) | (after == 0)).all() and (tails < 2 * recent).all()
        raw_after = cm.sparse_layer_raw_slots_map.cpu().numpy()
        lat_after = cm.sparse_layer_latent_slots_map.cpu().numpy()
        for bi, r in enumerate(rows):
            n_tot, c = int(cm.row_seq_lens[r]), int(after[bi])
            assert (lat_after[r, sink:sink + c] >= 0).all() and (lat_after[r, sink + c:n_tot] < 0).all()
            assert (raw_after[r, :sink] >= 0).all() and (raw_after[r, sink + c:n_tot] >= 0).all()
            centres = raw_after[r, sink:sink + c] >= 0
            assert centres.sum() == len(cm.row_deltakv_center_slots.get(int(r), [0] * sink)) - sink or c == 0
    assert n_evictions >= 2 * B - 1
    n_steps = cfg.get("steps", 28)
    if cfg.get("fused"):
        # every sparse layer of every step went through the fused second-Linear + reconstruction launch
        assert fused_calls["layers"] == n_steps * len(cm.deltakv_layer_ids), fused_calls
    else:
        assert fused_calls["n"] == 0, fused_calls          # (shapes the fused launch does not serve)


def test_deltakv_free_seq_returns_every_slot():
    from sparse_vllm_amd.config import Config
    from tools.synthetic import SyntheticDecodeDriver as SparseDecodeDriver
    conf = Config.from_kwargs(
        sparse_method="deltakv", num_hidden_layers=3, full_attention_layers="0", num_attention_heads=8,
        num_key_value_heads=2, head_dim=64, max_model_len=256, max_num_seqs_in_gpu=3, sink_keep_tokens=4,
        recent_keep_tokens=8, decode_keep_tokens=12, deltakv_neighbor_count=2, deltakv_latent_dim=32,
        deltakv_latent_quant_group_size=16, deltakv_center_ratio=0.25, allow_missing_deltakv_path=True,
        compressor_up_type="linear", full_layer_kv_quant_bits=4)
    drv = SparseDecodeDriver(conf)
    cm = drv.cache_manager
    before = cm.free_slot_stats()
    seqs = drv.admit_compressed_rows(2, [150, 80], seed=1)
    q, k, v = drv.random_step_inputs(seed=9)
    drv.step(q, k, v)
    torch.cuda.synchronize()
    held = cm.free_slot_stats()
    assert held["latent"] < before["latent"] and held["kivi_blocks"] < before["kivi_blocks"]
    for s in seqs:
        cm.free_seq(s.seq_id)
    after = cm.free_slot_stats()
    # reconstruct scratch (+ the mask's dummy slot when the reference-style mask path was used)
    scratch = sum(t.numel() for t in cm._temp_slots_by_shape.values()) + (getattr(cm, "_deltakv_postrope_dummy_slot", None) is not None)
    assert after["full"] == before["full"] and after["latent"] == before["latent"]
    assert after["kivi_blocks"] == before["kivi_blocks"]
    assert after["deltakv_full"] == before["deltakv_full"] - scratch
    with pytest.raises(ValueError, match="unknown seq_id"):
        cm.free_seq(seqs[0].seq_id)


@pytest.mark.parametrize("graph", [False, True])
def test_reconstruction_into_the_views_equals_scratch_slots_then_copy(graph):
    """DeltaKVCacheManager.recon_into_view (SvkDeltakvReconstructArgs.out_k_cache + SvkDeltakvMaterializeArgs.skip_temp):
    the reconstructed rows written straight into each layer's attention view, with the view kernel copying the raw rows
    only, give the decode outputs of the reference's flow - scratch slots in the cache, then the whole view copied - bit
    for bit, over compression events, eager and under hipGraph replay (look-ahead reconstruction on the side stream)."""
    from sparse_vllm_amd.config import Config
    from tools.synthetic import SyntheticDecodeDriver as SparseDecodeDriver
    L, B, Hq, D = 8, 3, 8, 64

    def run(into_view):
        conf = Config.from_kwargs(
            sparse_method="deltakv", num_hidden_layers=L, full_attention_layers="0,4", num_attention_heads=Hq,
            num_key_value_heads=2, head_dim=D, max_model_len=256, max_num_seqs_in_gpu=B + 1, sink_keep_tokens=4,
            recent_keep_tokens=8, decode_keep_tokens=12, deltakv_neighbor_count=2, deltakv_latent_dim=32,
            deltakv_latent_quant_bits=4, deltakv_latent_quant_group_size=16, deltakv_center_ratio=0.25,
            allow_missing_deltakv_path=True, compressor_up_type="mlp_gelu", compressor_intermediate_size=48,
            full_layer_kv_quant_bits=4, full_layer_kivi_decode_block_seq=64, rope_theta=10000.0)
        drv = SparseDecodeDriver(conf)
        cm = drv.cache_manager
        cm.recon_into_view = into_view
        cm.permute_free_slots(7)
        drv.admit_compressed_rows(B, [148, 92, 61], seed=3)
        if graph:
            drv.enable_decode_graph()
        outs = torch.zeros((L, B, Hq, D), dtype=torch.bfloat16, device=drv.device)
        got = []
        for step in range(20):                        # two compression events per row
            q, k, v = drv.random_step_inputs(seed=70 + step)
            drv.step(q, k, v, outputs=outs)
            torch.cuda.synchronize()
            got.append(outs.view(torch.int16).cpu().numpy().copy())
        used = cm._layer_views is not None
        return np.stack(got), used

    ref, used_ref = run(False)
    new, used_new = run(True)
    assert used_new and not used_ref                  # the run under test really took the view path
    np.testing.assert_array_equal(new, ref)


def _run_bookkeeping(device_state: bool, graph: bool, steps: int, *, sync_debug_from: int | None = None, recent: int = 8,
                     kivi: bool = True):
    from sparse_vllm_amd.config import Config
    from tools.synthetic import SyntheticDecodeDriver as SparseDecodeDriver
    L, B, Hq, D = 6, 3, 8, 64
    conf = Config.from_kwargs(
        sparse_method="deltakv", num_hidden_layers=L, full_attention_layers="0,3", num_attention_heads=Hq,
        num_key_value_heads=2, head_dim=D, max_model_len=256, max_num_seqs_in_gpu=B + 1, sink_keep_tokens=4,
        recent_keep_tokens=recent, decode_keep_tokens=12, deltakv_neighbor_count=2, deltakv_latent_dim=32,
        deltakv_latent_quant_bits=4, deltakv_latent_quant_group_size=16, deltakv_center_ratio=0.25,
        allow_missing_deltakv_path=True, compressor_up_type="mlp_gelu", compressor_intermediate_size=48,
        full_layer_kv_quant_bits=4 if kivi else 0, full_layer_kivi_decode_block_seq=64, rope_theta=10000.0)
    drv = SparseDecodeDriver(conf)
    cm = drv.cache_manager
    cm._device_step_enabled = device_state
    cm.permute_free_slots(11)
    drv.admit_compressed_rows(B, [148, 92, 61], seed=5)
    if graph:
        drv.enable_decode_graph()
    outs = torch.zeros((L, B, Hq, D), dtype=torch.bfloat16, device=drv.device)
    got, used_device, compared = [], 0, 0
    for step in range(steps):
        q, k, v = drv.random_step_inputs(seed=40 + step)
        if sync_debug_from is not None and step >= sync_debug_from:
            torch.cuda.set_sync_debug_mode("error")
        try:
            drv.step(q, k, v, outputs=outs)
        finally:
            torch.cuda.set_sync_debug_mode("default")
        torch.cuda.synchronize()
        got.append(outs.view(torch.int16).cpu().numpy().copy())
        if cm._device_step is not None:
            used_device += 1
            stands = cm._dev_stands_for
            if stands[0] == cm._pool_full.version and stands[1] == cm._pool_sparse.version:
                # nothing but the step touched the bookkeeping: the device copies ARE the host mirrors
                dev = cm._dev_state
                np.testing.assert_array_equal(dev["row_len"].cpu().numpy(), cm.row_seq_lens)
                for name, pool in (("full", cm._pool_full), ("sparse", cm._pool_sparse)):
                    assert int(dev[name + "_ptr"].item()) == pool.free
                    np.testing.assert_array_equal(dev[name + "_stack"][: pool.free].cpu().numpy(), pool.stack[: pool.free])
                compared += 1
    n = int(cm.row_seq_lens.max())
    return dict(o=np.stack(got), full_map=cm.full_layer_slots_map[:, :n].cpu().numpy().copy(),
                sparse_map=cm.sparse_layer_raw_slots_map[:, :n].cpu().numpy().copy(),
                full_pos=cm.full_layer_slot_to_pos.cpu().numpy().copy(), sparse_pos=cm.deltakv_slot_to_pos.cpu().numpy().copy(),
                lens=cm.row_seq_lens.copy(), clens=cm.row_deltakv_compressed_lens.copy(),
                full_free=cm._pool_full.stack[: cm._pool_full.free].copy(),
                sparse_free=cm._pool_sparse.stack[: cm._pool_sparse.free].copy(), used_device=used_device, compared=compared)


def test_deltakv_device_resident_steps_equal_host_driven_steps():
    """SURVEY 8(f).2 for DeltaKV: row lengths and the two raw-slot free stacks on the device, the step's allocation
    (`svk_deltakv_device_step_begin`) a launch of the step instead of an upload + `svk_deltakv_decode_alloc`.  Against the
    host-driven steps (deltakv_base.py:2038-2154) over several compression events per row: outputs, slot maps, position
    maps, free stacks (content and order), lengths bit-identical, eager and under hipGraph replay; whenever only the step
    touched the bookkeeping the device copies equal the host mirrors."""
    steps = 26
    ref = _run_bookkeeping(False, False, steps)
    assert ref["used_device"] == 0
    assert int(ref["clens"].max()) > 148 - 4 - 8               # compression did happen during the run
    for graph in (False, True):
        got = _run_bookkeeping(True, graph, steps)
        assert got["used_device"] == steps and got["compared"] >= steps // 3
        for key in ("o", "full_map", "sparse_map", "full_pos", "sparse_pos", "lens", "clens", "full_free", "sparse_free"):
            np.testing.assert_array_equal(got[key], ref[key], err_msg=f"{key} graph={graph}")


def test_deltakv_device_resident_step_uploads_nothing():
    """Between compression events (the sparse layers' `deltakv_evict`, the full layers' KIVI block quantisation every
    group of tokens - both host-driven) a replayed DeltaKV step is a graph launch plus numpy arithmetic on the host
    mirrors: torch's sync debug mode "error" around the steps of a window long enough not to compress, raw full layers."""
    got = _run_bookkeeping(True, True, 20, sync_debug_from=4, recent=64, kivi=False)
    assert got["used_device"] == 20 and got["compared"] >= 16


def test_fused_second_linear_and_reconstruction_in_the_decode_steps(monkeypatch):
    """`svk_deltakv_up_reconstruct` inside the manager's look-ahead (SVK_DELTAKV_FUSED_UP=always) against the library GEMM +
    reconstruct launches (=0) over decode steps with compression events, head_dim 128: the delta rows differ by the fp32
    summation order of the second Linear only, the attention outputs agree within the bf16 tolerance of the path and
    nearly all of them exactly; the fused launch really ran."""
    from sparse_vllm_amd.config import Config
    from sparse_vllm_amd.kernels import deltakv_kernels as dk
    from tools.synthetic import SyntheticDecodeDriver as SparseDecodeDriver
    L, B, Hq, Hkv, D = 6, 2, 8, 4, 128
    calls = {"n": 0}
    orig = dk.deltakv_up_reconstruct_layers

    def counted(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)
    monkeypatch.setattr(dk, "deltakv_up_reconstruct_layers", counted)

    def run(mode):
        monkeypatch.setenv("SVK_DELTAKV_FUSED_UP", mode)
        monkeypatch.setenv("SVK_DELTAKV_RECON_AHEAD", "1")        # the fused launch lives in the look-ahead's layer batches
        conf = Config.from_kwargs(
            sparse_method="deltakv", num_hidden_layers=L, full_attention_layers="0,3", num_attention_heads=Hq,
            num_key_value_heads=Hkv, head_dim=D, max_model_len=512, max_num_seqs_in_gpu=B + 1, sink_keep_tokens=4,
            recent_keep_tokens=16, decode_keep_tokens=160, deltakv_neighbor_count=4, deltakv_latent_dim=64,
            deltakv_latent_quant_bits=4, deltakv_latent_quant_group_size=32, deltakv_center_ratio=0.1,
            allow_missing_deltakv_path=True, compressor_up_type="mlp_gelu", compressor_intermediate_size=128,
            full_layer_kv_quant_bits=0, rope_theta=10000.0)
        drv = SparseDecodeDriver(conf)
        drv.cache_manager.permute_free_slots(5)
        drv.admit_compressed_rows(B, [300, 211], seed=2)
        outs = torch.zeros((L, B, Hq, D), dtype=torch.bfloat16, device=drv.device)
        got = []
        for step in range(20):
            q, k, v = drv.random_step_inputs(seed=90 + step)
            drv.step(q, k, v, outputs=outs)
            torch.cuda.synchronize()
            got.append(outs.float().cpu().numpy().copy())
        return np.stack(got)

    ref = run("0")
    assert calls["n"] == 0
    new = run("always")
    assert calls["n"] >= 20                      # two sparse groups of two layers per step
    np.testing.assert_allclose(new, ref, rtol=2e-2, atol=2e-2)
    assert float((new == ref).mean()) > 0.9


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("shape", [dict(Hq=8, Hkv=2, D=64, inter=48, latent=32, group=16, lens=[148, 92, 61], keep=12, knorm=False),
                                   dict(Hq=28, Hkv=4, D=128, inter=128, latent=64, group=32, lens=[300, 211, 97], keep=160, knorm=True)])
def test_views_without_a_launch_on_the_walk_equal_the_view_launch_per_layer(shape, graph, monkeypatch):
    """`SVK_DELTAKV_ROTATED_STORE` (default on): a sparse layer's view is complete without a launch of its own on the walk -
    its raw rows are written for the whole layer group by one launch in front of the look-ahead reconstructions, the
    step's newest row (raw into the layer cache, k-normed + rotated into the view) rides in the attention launch.  Against
    the per-layer view launch (=0, deltakv_less_memory.py:1344-1402 with the riding store): outputs of every layer and the
    sparse layers' raw caches bit-identical over steps with compression events, eager and replayed; the view launches per
    step drop from one per sparse layer to one per layer group."""
    from sparse_vllm_amd.config import Config
    from sparse_vllm_amd.kernels import deltakv_kernels as dk
    from tools.synthetic import SyntheticDecodeDriver as SparseDecodeDriver
    L, B = 8, 3
    Hq, Hkv, D = shape["Hq"], shape["Hkv"], shape["D"]
    calls = {"n": 0}
    orig = dk.deltakv_materialize_sparse_view

    def counted(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)
    monkeypatch.setattr(dk, "deltakv_materialize_sparse_view", counted)

    def run(mode):
        monkeypatch.setenv("SVK_DELTAKV_ROTATED_STORE", mode)
        monkeypatch.setenv("SVK_DELTAKV_RECON_AHEAD", "1")        # the path under test lives behind the look-ahead ...
        monkeypatch.setenv("SVK_DELTAKV_FUSE_RAW_STORE", "1")     # ... and the riding raw store
        conf = Config.from_kwargs(
            sparse_method="deltakv", num_hidden_layers=L, full_attention_layers="0,4", num_attention_heads=Hq,
            num_key_value_heads=Hkv, head_dim=D, max_model_len=512, max_num_seqs_in_gpu=B + 1, sink_keep_tokens=4,
            recent_keep_tokens=8, decode_keep_tokens=shape["keep"], deltakv_neighbor_count=2, deltakv_latent_dim=shape["latent"],
            deltakv_latent_quant_bits=4, deltakv_latent_quant_group_size=shape["group"], deltakv_center_ratio=0.25,
            allow_missing_deltakv_path=True, compressor_up_type="mlp_gelu", compressor_intermediate_size=shape["inter"],
            full_layer_kv_quant_bits=4, full_layer_kivi_decode_block_seq=64, rope_theta=10000.0)
        drv = SparseDecodeDriver(conf)
        cm = drv.cache_manager
        if shape["knorm"]:
            gen = torch.Generator().manual_seed(4)
            cm.deltakv_k_norm_weight = torch.rand((len(cm.deltakv_layer_ids), D), generator=gen).add(0.5).to(drv.device)
        cm.permute_free_slots(7)
        drv.admit_compressed_rows(B, shape["lens"], seed=3)
        if graph:
            drv.enable_decode_graph()
        outs = torch.zeros((L, B, Hq, D), dtype=torch.bfloat16, device=drv.device)
        got = []
        calls["n"] = 0
        for step in range(20):                        # two compression events per row
            q, k, v = drv.random_step_inputs(seed=70 + step)
            drv.step(q, k, v, outputs=outs)
            torch.cuda.synchronize()
            got.append(outs.view(torch.int16).cpu().numpy().copy())
        raw = cm.deltakv_full_kv_cache.view(torch.int16).cpu().numpy().copy()
        n_sparse, groups = len(cm.deltakv_layer_ids), 2
        return np.stack(got), raw, calls["n"], n_sparse, groups

    ref, raw_ref, n_ref, n_sparse, groups = run("0")
    new, raw_new, n_new, _, _ = run("1")
    np.testing.assert_array_equal(new, ref)
    # the reconstruct-scratch slots of the raw caches are written by neither flow when the views take the reconstruction;
    # everything the steps stored must agree
    np.testing.assert_array_equal(raw_new, raw_ref)
    if not graph:
        assert n_ref == 20 * n_sparse and n_new == 20 * groups, (n_ref, n_new)
    else:
        assert n_new < n_ref


def test_lookahead_launch_group_sizes_do_not_change_the_steps(monkeypatch):
    """The look-ahead reconstructs the sparse layers of a layer group in launch groups of 2 layers (4 from 4096 selected
    tokens on: `_recon_sub_batches`).  Whatever the grouping - one layer per launch, twos, a whole group at once, mixed - the
    decode steps are the same bit for bit (groups of 5 and 3 sparse layers, compression events, replayed)."""
    from sparse_vllm_amd.config import Config
    from sparse_vllm_amd.engine.cache_manager.deltakv import DeltaKVCacheManager
    from tools.synthetic import SyntheticDecodeDriver as SparseDecodeDriver
    L, B, Hq, Hkv, D = 10, 2, 8, 4, 128

    def run(sizes, wide_from):
        monkeypatch.setenv("SVK_DELTAKV_RECON_AHEAD", "1")
        monkeypatch.delenv("SVK_DELTAKV_RECON_SUB", raising=False)
        monkeypatch.setattr(DeltaKVCacheManager, "_RECON_SUB_BATCHES", list(sizes))
        monkeypatch.setattr(DeltaKVCacheManager, "_RECON_SUB_BATCHES_WIDE", [4])
        monkeypatch.setattr(DeltaKVCacheManager, "_RECON_WIDE_TOKENS", wide_from)
        conf = Config.from_kwargs(
            sparse_method="deltakv", num_hidden_layers=L, full_attention_layers="0,6", num_attention_heads=Hq,
            num_key_value_heads=Hkv, head_dim=D, max_model_len=512, max_num_seqs_in_gpu=B + 1, sink_keep_tokens=4,
            recent_keep_tokens=16, decode_keep_tokens=160, deltakv_neighbor_count=4, deltakv_latent_dim=64,
            deltakv_latent_quant_bits=4, deltakv_latent_quant_group_size=32, deltakv_center_ratio=0.1,
            allow_missing_deltakv_path=True, compressor_up_type="mlp_gelu", compressor_intermediate_size=128,
            full_layer_kv_quant_bits=0, rope_theta=10000.0)
        drv = SparseDecodeDriver(conf)
        drv.cache_manager.permute_free_slots(5)
        drv.admit_compressed_rows(B, [300, 211], seed=2)
        drv.enable_decode_graph()
        outs = torch.zeros((L, B, Hq, D), dtype=torch.bfloat16, device=drv.device)
        got = []
        for step in range(20):
            q, k, v = drv.random_step_inputs(seed=90 + step)
            drv.step(q, k, v, outputs=outs)
            torch.cuda.synchronize()
            got.append(outs.view(torch.int16).cpu().numpy().copy())
        return np.stack(got)

    ref = run([2], 1 << 30)
    for sizes, wide_from in (([2], 0), ([8], 1 << 30), ([3, 2], 1 << 30), ([1, 4], 1 << 30)):
        np.testing.assert_array_equal(run(sizes, wide_from), ref, err_msg=f"sizes={sizes} wide_from={wide_from}")
    # one layer per launch everywhere is the per-layer path (library GEMMs for compress_up): the second Linear's fp32
    # summation order differs, a handful of outputs land on the neighbouring bf16 value
    one = run([1], 1 << 30)
    assert float((one == ref).mean()) > 0.99

    def as_float(a):
        return torch.from_numpy(a.copy()).view(torch.bfloat16).float().numpy()
    np.testing.assert_allclose(as_float(one), as_float(ref), rtol=2e-2, atol=2e-2)
