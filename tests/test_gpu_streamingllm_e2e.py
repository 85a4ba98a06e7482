"""End-to-end StreamingLLM decode on the GPU through the reference's operator surface (CacheManager.create ->
prepare_decode_static -> Attention.forward per layer -> SparseController.post_forward) against the numpy oracle
driven step by step on the same inputs: attention outputs within tolerance, slot tables / free stacks / lengths
bit-exact across sink + recent compactions (sparse_controller.py:1558-1653, snapkv.py:1805-1896), eager and under
hipGraph replay with the physical-peak context capacity of StreamingLLMCacheManager."""

import os

import numpy as np
import pytest

from oracle import bf16_round
from oracle import decode_attention as oda
from oracle import h2o as oh

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _bf(t):
    return t.float().cpu().numpy()


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("cfg", [
    dict(B=3, L=3, sink=4, recent=12, start=16, steps=56, Hq=28, Hkv=4, D=128),
    dict(B=2, L=2, sink=8, recent=24, start=40, steps=96, Hq=14, Hkv=2, D=64),
])
def test_streamingllm_decode_steps_match_oracle(cfg, graph):
    from sparse_vllm_amd.config import Config
    from tools.synthetic import SyntheticDecodeDriver as SparseDecodeDriver
    B, L, sink, recent = cfg["B"], cfg["L"], cfg["sink"], cfg["recent"]
    budget = sink + recent
    conf = Config.from_kwargs(sparse_method="streamingllm", sink_keep_tokens=sink, recent_keep_tokens=recent,
                              num_hidden_layers=L, num_attention_heads=cfg["Hq"], num_key_value_heads=cfg["Hkv"],
                              head_dim=cfg["D"], max_model_len=256, max_num_seqs_in_gpu=B + 1,
                              num_kvcache_slots=B * 2 * budget + 19)
    drv = SparseDecodeDriver(conf)
    cm = drv.cache_manager
    cm.permute_free_slots(3)
    seqs = drv.admit_resident_rows(B, cfg["start"], logical_len=200, seed=9)
    rows = [cm.seq_id_to_row[0][s.seq_id] for s in seqs]
    if graph:
        drv.enable_decode_graph()
        # the physical decode peak, not the logical context (200) or the table width (256)
        assert cm._decode_static_max_context_len == max(2 * budget, cfg["start"] + 1)

    st = oh.SlotState(cm.buffer_req_to_token_slots_tensor.cpu().numpy().copy(),
                      cm.free_slots_stack_tensor.cpu().numpy().copy(),
                      np.asarray(cm._num_free_slots, dtype=np.int64),
                      np.stack(cm.row_seq_lens).astype(np.int32))
    kc = _bf(cm.kv_cache[0]).copy()
    vc = _bf(cm.kv_cache[1]).copy()
    outs = torch.zeros((L, B, cfg["Hq"], cfg["D"]), dtype=torch.bfloat16, device=drv.device)
    compactions = 0
    for step in range(cfg["steps"]):
        q, k, v = drv.random_step_inputs(seed=300 + step)
        drv.step(q, k, v, outputs=outs)
        torch.cuda.synchronize()

        new_slots = oh.decode_allocate_batch_layers(st, range(L), rows)
        lens = np.array([st.row_len[0, r] for r in rows], dtype=np.int32)
        qn, kn, vn = _bf(q), _bf(k), _bf(v)
        for l in range(L):
            kc[l][new_slots[l]] = kn[l]
            vc[l][new_slots[l]] = vn[l]
            mid, lse = oda.flash_decode_stage1(qn[l], kc[l], vc[l], st.slot_table[l], np.array(rows, np.int32), lens,
                                               int(lens.max()), 64)
            o = oda.flash_decode_stage2(mid, lse, lens, 64)
            np.testing.assert_allclose(_bf(outs[l]), bf16_round(o), rtol=2e-2, atol=2e-2)
        kv_len = int(lens[0])
        assert (lens == kv_len).all()
        if kv_len >= 2 * budget:
            compactions += 1
            oh.free_prefix_recent_slots(st, range(L), rows, kv_len=kv_len, prefix_tokens=sink, recent_tokens=recent)

        np.testing.assert_array_equal(np.stack(cm.row_seq_lens), st.row_len)
        np.testing.assert_array_equal(np.asarray(cm._num_free_slots), st.free_ptr)
        tab = cm.buffer_req_to_token_slots_tensor.cpu().numpy()
        stack = cm.free_slots_stack_tensor.cpu().numpy()
        for l in range(L):
            for r in rows:
                n = int(st.row_len[l, r])
                np.testing.assert_array_equal(tab[l, r, :n], st.slot_table[l, r, :n])
                assert (tab[l, r, n:] == 0).all()
            p = int(st.free_ptr[l])
            np.testing.assert_array_equal(stack[l, :p], st.free_stack[l, :p])
    assert compactions >= 3
    # the steps ran from the device-resident bookkeeping, whose copies equal the host mirrors
    if os.environ.get("SVK_H2O_DEVICE_STATE", "1") != "0":          # (the opt-out knob runs the host-driven steps)
        assert cm._dev_step_cache is not None
        np.testing.assert_array_equal(cm._dev_row_len.cpu().numpy(), np.stack(cm.row_seq_lens))
        np.testing.assert_array_equal(cm._dev_free_ptr.cpu().numpy(), np.asarray(cm._num_free_slots))


def _run_window(device_state: bool, graph: bool, steps: int, *, ragged: bool = False, sync_debug: bool = False):
    from sparse_vllm_amd.config import Config
    from tools.synthetic import SyntheticDecodeDriver as SparseDecodeDriver
    B, L, sink, recent = 4, 3, 4, 12
    budget = sink + recent
    conf = Config.from_kwargs(sparse_method="streamingllm", sink_keep_tokens=sink, recent_keep_tokens=recent,
                              num_hidden_layers=L, max_model_len=128, max_num_seqs_in_gpu=B + 1,
                              num_kvcache_slots=B * 2 * budget + 23)
    drv = SparseDecodeDriver(conf)
    cm = drv.cache_manager
    cm._device_step_enabled = device_state
    cm.permute_free_slots(4)
    drv.admit_resident_rows(B, budget + 3, logical_len=100, seed=8)
    if ragged:
        # two of the four rows are 5 tokens behind: the window then moves on subsets of the batch
        for l in range(L):
            for s in drv.seqs[:2]:
                cm.free_part_slots(l, s, torch.arange(budget - 2, device=drv.device), keep_indices_sorted=True)
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
    return dict(o=o.view(torch.int16).cpu().numpy().copy(), table=cm.buffer_req_to_token_slots_tensor.cpu().numpy().copy(),
                stack=cm.free_slots_stack_tensor.cpu().numpy().copy(), lens=np.stack(cm.row_seq_lens).copy(),
                ptr=list(cm._num_free_slots), dev_lens=cm._dev_row_len.cpu().numpy().copy(),
                dev_ptr=cm._dev_free_ptr.cpu().numpy().copy(), used_device=used_device)


@pytest.mark.parametrize("ragged", [False, True])
def test_streamingllm_device_resident_steps_equal_host_driven_steps(ragged):
    """SURVEY 8(f).2 for StreamingLLM: row lengths / free-stack pointers on the device, the allocation and the predicated
    window eviction (`SVK_DEVICE_SELECT_WINDOW`) as launches of the step.  Against the host-driven form
    (SparseController._streamingllm_decode_eviction -> free_prefix_recent_slots_batch_layers) across >= 4 window moves:
    slot tables, free stacks (content and order), lengths and outputs bit-identical, eager and under hipGraph replay;
    the device copies of the bookkeeping equal the host mirrors; rows that trigger at different steps included."""
    steps = 4 * 16 + 9
    ref = _run_window(False, False, steps, ragged=ragged)
    assert ref["used_device"] == 0
    for graph in (False, True):
        got = _run_window(True, graph, steps, ragged=ragged)
        assert got["used_device"] >= steps - 2
        for key in ("o", "table", "lens"):
            np.testing.assert_array_equal(got[key], ref[key], err_msg=f"{key} graph={graph}")
        assert got["ptr"] == ref["ptr"]
        for l in range(len(ref["ptr"])):
            np.testing.assert_array_equal(got["stack"][l, : ref["ptr"][l]], ref["stack"][l, : ref["ptr"][l]])
        np.testing.assert_array_equal(got["dev_lens"], got["lens"])
        np.testing.assert_array_equal(got["dev_ptr"], np.asarray(got["ptr"]))


def test_streamingllm_device_resident_step_needs_no_host_sync():
    """Under hipGraph replay a step - the window eviction included - is a graph launch plus numpy arithmetic on the host
    mirrors: three full window cycles under torch's sync debug mode "error"."""
    got = _run_window(True, True, 3 * 16 + 8, sync_debug=True)
    assert got["used_device"] >= 3 * 16
    assert int(got["lens"].max()) < 2 * 16          # the window did move
