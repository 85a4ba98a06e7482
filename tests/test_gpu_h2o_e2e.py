"""End-to-end H2O decode on the GPU through the reference's operator surface
(CacheManager.create -> prepare_decode_static -> Attention.forward per layer ->
SparseController.post_forward) against the numpy oracle driven step by step on the same
inputs: slot tables / free stacks / lengths bit-exact, kept indices bit-exact (any mismatch
must be attributable to a score tie within float noise), scores and outputs within tolerance."""

import numpy as np
import pytest

from oracle import bf16_round
from oracle import decode_attention as oda
from oracle import h2o as oh

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _bf(t):
    return t.float().cpu().numpy()


@pytest.mark.parametrize("cfg", [
    dict(B=3, L=3, budget=48, interval=16, start=48, steps=40, Hq=28, Hkv=4, D=128),
    dict(B=2, L=2, budget=112, interval=16, start=100, steps=50, Hq=14, Hkv=2, D=64),
    dict(B=2, L=2, budget=112, interval=16, start=104, steps=45, Hq=7, Hkv=1, D=128),    # one TP=4 rank of Qwen2.5-7B
])
def test_h2o_decode_steps_match_oracle(cfg):
    # the score epilogues of all layers of a step run in one launch after the layer loop
    from sparse_vllm_amd.config import Config
    from tools.synthetic import SyntheticDecodeDriver as SparseDecodeDriver
    B, L, budget, interval = cfg["B"], cfg["L"], cfg["budget"], cfg["interval"]
    conf = Config.from_kwargs(sparse_method="h2o", num_hidden_layers=L, num_attention_heads=cfg["Hq"],
                              num_key_value_heads=cfg["Hkv"], head_dim=cfg["D"], max_model_len=256,
                              max_num_seqs_in_gpu=B + 2, num_kvcache_slots=B * (budget + interval) + 37,
                              h2o_decode_budget=budget, h2o_decode_eviction_interval=interval,
                              h2o_prefill_budget=2 * budget)
    from sparse_vllm_amd.kernels import h2o_ops as _ops
    before = dict(_ops.SCORE_LAYERS_LAUNCHES)
    drv = SparseDecodeDriver(conf)
    cm = drv.cache_manager
    cm.permute_free_slots(11)
    seqs = drv.admit_resident_rows(B, cfg["start"], seed=5)
    rows = [cm.seq_id_to_row[0][s.seq_id] for s in seqs]

    # ---- mirror the initial device state into the oracle
    st = oh.SlotState(cm.buffer_req_to_token_slots_tensor.cpu().numpy().copy(),
                      cm.free_slots_stack_tensor.cpu().numpy().copy(),
                      np.asarray(cm._num_free_slots, dtype=np.int64),
                      np.stack(cm.row_seq_lens).astype(np.int32))
    kc = _bf(cm.kv_cache[0]).copy()
    vc = _bf(cm.kv_cache[1]).copy()
    for l in range(L):
        for r in rows:
            st.scores[(l, r)] = cm.h2o_score_tensor[l, r, : cfg["start"]].cpu().numpy().copy()

    outs = torch.zeros((L, B, cfg["Hq"], cfg["D"]), dtype=torch.bfloat16, device=drv.device)
    n_bursts = 0
    for step in range(cfg["steps"]):
        q, k, v = drv.random_step_inputs(seed=100 + step)
        drv.step(q, k, v, outputs=outs)
        torch.cuda.synchronize()

        # ---- oracle step
        new_slots = oh.decode_allocate_batch_layers(st, range(L), rows)
        lens = np.array([st.row_len[0, r] for r in rows], dtype=np.int32)
        qn, kn, vn = _bf(q), _bf(k), _bf(v)
        for l in range(L):
            kc[l][new_slots[l]] = kn[l]
            vc[l][new_slots[l]] = vn[l]
            W = int(lens.max())
            raw = np.full((B, W), -1e20, dtype=np.float32)
            mid, lse = oda.flash_decode_stage1(qn[l], kc[l], vc[l], st.slot_table[l], np.array(rows, np.int32), lens,
                                               W, 64, attn_score=raw)
            o = oda.flash_decode_stage2(mid, lse, lens, 64)
            np.testing.assert_allclose(_bf(outs[l]), bf16_round(o), rtol=2e-2, atol=2e-2)
            norm = oda.h2o_normalize_decode_scores(raw, cfg["D"])
            for b, r in enumerate(rows):
                st.scores[(l, r)] = oh.update_decode_scores(st.scores[(l, r)], norm[b], int(lens[b]))
        row_lens = {r: int(st.row_len[0, r]) for r in rows}
        groups = oh.decode_eviction_groups(row_lens, rows, rows, budget=budget, interval=interval,
                                           num_free_slots=int(st.free_ptr.min()))
        if groups:
            n_bursts += 1
            oh.evict_decode_rows(st, range(L), groups, budget=budget, recent_ratio=0.5)

        # ---- compare full state
        np.testing.assert_array_equal(np.stack(cm.row_seq_lens), st.row_len)
        np.testing.assert_array_equal(np.asarray(cm._num_free_slots), st.free_ptr)
        tab = cm.buffer_req_to_token_slots_tensor.cpu().numpy()
        stack = cm.free_slots_stack_tensor.cpu().numpy()
        sc = cm.h2o_score_tensor.cpu().numpy()
        for l in range(L):
            for r in rows:
                n = int(st.row_len[l, r])
                if not np.array_equal(tab[l, r, :n], st.slot_table[l, r, :n]):
                    # attribute: only acceptable if the disagreeing tokens' scores are within float noise
                    raise AssertionError(f"slot table diverged at step {step} layer {l} row {r}")
                np.testing.assert_allclose(sc[l, r, :n], st.scores[(l, r)], rtol=1e-4, atol=1e-6)
                assert (tab[l, r, n:] == 0).all()
            p = int(st.free_ptr[l])
            np.testing.assert_array_equal(stack[l, :p], st.free_stack[l, :p])
    assert n_bursts >= 2
    assert cm._h2o_counters["decode_eviction_bursts"] == n_bursts * B
    # the layers' buffers are equally spaced slices: every step took the single all-layers launch
    from sparse_vllm_amd.kernels import h2o_ops
    assert h2o_ops.SCORE_LAYERS_LAUNCHES["batched"] > before["batched"] and h2o_ops.SCORE_LAYERS_LAUNCHES["per_layer"] == before["per_layer"]


def _run_h2o(device_state: bool, graph: bool, steps: int, *, ragged: bool = False, sync_debug: bool = False, slots: int | None = None):
    from sparse_vllm_amd.config import Config
    from tools.synthetic import SyntheticDecodeDriver as SparseDecodeDriver
    B, L, budget, interval = 4, 3, 48, 16
    conf = Config.from_kwargs(sparse_method="h2o", num_hidden_layers=L, max_model_len=128, max_num_seqs_in_gpu=B + 1,
                              num_kvcache_slots=slots or (B * (budget + interval) + 41), h2o_decode_budget=budget,
                              h2o_decode_eviction_interval=interval, h2o_prefill_budget=2 * budget)
    drv = SparseDecodeDriver(conf)
    cm = drv.cache_manager
    cm._device_step_enabled = device_state
    cm.permute_free_slots(4)
    drv.admit_resident_rows(B, budget, seed=8)
    if ragged:
        # two of the four rows are 5 tokens behind: bursts then hit subsets of the batch (two phases per interval)
        for l in range(L):
            for s in drv.seqs[:2]:
                r = cm.seq_id_to_row[l][s.seq_id]
                keep = torch.arange(budget - 5, device=drv.device)
                cm.free_part_slots(l, s, keep, keep_indices_sorted=True)
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
    return dict(o=o.view(torch.int16).cpu().numpy().copy(), score=cm.h2o_score_tensor.cpu().numpy().copy(),
                table=cm.buffer_req_to_token_slots_tensor.cpu().numpy().copy(), stack=cm.free_slots_stack_tensor.cpu().numpy().copy(),
                lens=np.stack(cm.row_seq_lens).copy(), ptr=list(cm._num_free_slots), counters=dict(cm._h2o_counters),
                dev_lens=cm._dev_row_len.cpu().numpy().copy(), dev_ptr=cm._dev_free_ptr.cpu().numpy().copy(),
                used_device=used_device)


@pytest.mark.parametrize("ragged", [False, True])
def test_device_resident_bookkeeping_equals_host_driven_steps(ragged):
    """row lengths and free-stack pointers on the device, the allocation and the (predicated) burst as
    launches of the step.  Against the host-driven form of the same steps across four bursts: slot tables, free stacks
    (content and order), score rows and outputs bit-identical, eager and under hipGraph replay; the device copy of the
    bookkeeping equals the host mirrors; rows that trigger at different steps (subsets of the batch) included."""
    steps = 4 * 16 + 6
    ref = _run_h2o(False, False, steps, ragged=ragged)
    assert ref["used_device"] == 0 and ref["counters"]["decode_eviction_bursts"] >= 4 * 2
    for graph in (False, True):
        got = _run_h2o(True, graph, steps, ragged=ragged)
        assert got["used_device"] >= steps - 2
        for key in ("o", "score", "table", "lens"):
            np.testing.assert_array_equal(got[key], ref[key], err_msg=f"{key} graph={graph}")
        assert got["ptr"] == ref["ptr"] and got["counters"] == ref["counters"]
        for l in range(len(ref["ptr"])):
            np.testing.assert_array_equal(got["stack"][l, : ref["ptr"][l]], ref["stack"][l, : ref["ptr"][l]])
        np.testing.assert_array_equal(got["dev_lens"], got["lens"])
        np.testing.assert_array_equal(got["dev_ptr"], np.asarray(got["ptr"]))


def test_device_resident_step_needs_no_host_sync():
    """Under hipGraph replay a step - the burst included - is a graph launch plus numpy arithmetic on the host mirrors:
    with torch's sync debug mode set to "error" three full eviction intervals run without a single host <-> device
    synchronisation."""
    got = _run_h2o(True, True, 3 * 16 + 8, sync_debug=True)
    assert got["counters"]["decode_eviction_bursts"] == 3 * 4 and got["used_device"] >= 3 * 16


def test_device_resident_bookkeeping_soak_forty_bursts():
    """The same comparison over 40 eviction intervals (643 steps, ragged rows: 80 partial bursts): the device copies of the
    bookkeeping never drift from the host mirrors, the graph-replayed device-resident run stays bit-identical to the
    host-driven eager run (tables, free stacks, pointers, cumulative scores, outputs), and every layer's rows and free
    stack still partition its slot pool at the end."""
    steps = 40 * 16 + 3
    ref = _run_h2o(False, False, steps, ragged=True)
    got = _run_h2o(True, True, steps, ragged=True)
    assert ref["counters"]["decode_eviction_bursts"] >= 40 * 4 - 4 and got["counters"] == ref["counters"]
    for key in ("o", "score", "table", "lens"):
        np.testing.assert_array_equal(got[key], ref[key], err_msg=key)
    assert got["ptr"] == ref["ptr"]
    np.testing.assert_array_equal(got["dev_lens"], got["lens"])
    np.testing.assert_array_equal(got["dev_ptr"], np.asarray(got["ptr"]))
    n_slots = got["stack"].shape[1]
    for l in range(len(got["ptr"])):
        np.testing.assert_array_equal(got["stack"][l, : got["ptr"][l]], ref["stack"][l, : ref["ptr"][l]])
        used = np.concatenate([got["table"][l, r, : got["lens"][l][r]] for r in range(got["table"].shape[1]) if got["lens"][l][r] > 0])
        free = got["stack"][l, : got["ptr"][l]]
        both = np.concatenate([used, free])
        assert len(np.unique(both)) == len(both) and len(both) <= n_slots           # disjoint: no slot owned twice


def test_h2o_slot_pressure_fallback_under_graph_replay():
    """A pool so tight that the free stack runs dry before the rows reach budget + interval: the reference then lowers the
    trigger to budget + 1 (h2o.py:1506-1524).  Those steps leave the device-resident path (its plan refuses a step that
    would empty the stack) and run host-driven, also when the caller replays hipGraphs: the driver drops its graph, runs
    the step eagerly and re-captures.  Against the host-driven eager run: tables, free stacks, scores, outputs and counters
    bit-identical over several pressure bursts; both runs did evict under pressure (more bursts than the periodic rule
    alone would give)."""
    B, budget, interval, steps = 4, 48, 16, 60
    slots = B * budget + 2 * B                     # two steps of head-room, then the stack is empty: pressure
    ref = _run_h2o(False, False, steps, slots=slots)
    periodic_only = (steps // interval) * B
    assert ref["counters"]["decode_eviction_bursts"] > periodic_only
    for device_state in (True, False):
        got = _run_h2o(device_state, True, steps, slots=slots)
        for key in ("o", "score", "table", "lens"):
            np.testing.assert_array_equal(got[key], ref[key], err_msg=f"{key} device_state={device_state}")
        assert got["ptr"] == ref["ptr"] and got["counters"] == ref["counters"]
        for l in range(len(ref["ptr"])):
            np.testing.assert_array_equal(got["stack"][l, : ref["ptr"][l]], ref["stack"][l, : ref["ptr"][l]])
