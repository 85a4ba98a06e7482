"""GPU: round-5 host paths on real managers.

* `SVLLM_DEBUG_DECODE_BOUNDS=1` (both checks of the reference: layers/attention.py:196-211 and
  layers/attention_backend.py:397-439) over eager decode steps of every method - no false positive on the token-slot
  views, on Quest's page-slot view (slot ids are PAGE slots there) or on the KIVI full layers (skipped like the
  reference skips them), outputs identical to the unchecked run; a corrupted slot table raises the reference's message.
* the model calls `save_rope_kv_if_needed` and then the attention layer (models/qwen2.py:126-131): with the store riding
  in stage 1 (`take_deferred_decode_store`) and with `SVK_FUSE_DECODE_STORE=0` the caches and outputs are bit-identical;
  rows no launch took are flushed.
* `tools/e2e_decoder.py` on a two-layer model: the end-to-end leg's driver runs, graph replay and eager launch agree on the
  greedy tokens of a short run."""

import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QWEN = dict(num_attention_heads=28, num_key_value_heads=4, head_dim=128)


def _drivers():
    from sparse_vllm_amd.config import Config
    from tools.synthetic import SyntheticDecodeDriver as Drv
    out = {}
    cfg = Config.from_kwargs(sparse_method="h2o", num_hidden_layers=2, max_model_len=512, max_num_seqs_in_gpu=3,
                             num_kvcache_slots=1024, h2o_decode_budget=192, h2o_decode_eviction_interval=64, h2o_prefill_budget=256, **QWEN)
    d = Drv(cfg); d.cache_manager.permute_free_slots(3); d.admit_resident_rows(3, 200, seed=1); out["h2o"] = d
    cfg = Config.from_kwargs(sparse_method="quest", num_hidden_layers=3, sink_keep_tokens=16, decode_keep_tokens=64, recent_keep_tokens=32,
                             max_model_len=1024, max_num_seqs_in_gpu=2, num_kvcache_slots=2 * 1024, **QWEN)
    d = Drv(cfg); d.cache_manager.permute_free_pages(1); d.admit_resident_rows(2, 700, seed=2); out["quest"] = d
    cfg = Config.from_kwargs(sparse_method="streamingllm", num_hidden_layers=2, sink_keep_tokens=8, recent_keep_tokens=40, max_model_len=256,
                             max_num_seqs_in_gpu=2, num_kvcache_slots=512, **QWEN)
    d = Drv(cfg); d.cache_manager.permute_free_slots(2); d.admit_resident_rows(2, 48, logical_len=200, seed=3); out["streamingllm"] = d
    cfg = Config.from_kwargs(sparse_method="deltakv", num_hidden_layers=4, full_attention_layers="0,2", sink_keep_tokens=8,
                             recent_keep_tokens=32, decode_keep_tokens=64, deltakv_neighbor_count=4, deltakv_latent_dim=256,
                             deltakv_latent_quant_bits=4, deltakv_latent_quant_group_size=32, deltakv_center_ratio=0.1,
                             allow_missing_deltakv_path=True, compressor_intermediate_size=256, full_layer_kv_quant_bits=4,
                             max_model_len=1024, max_num_seqs_in_gpu=2, **QWEN)
    d = Drv(cfg); d.cache_manager.permute_free_slots(1); d.admit_compressed_rows(2, [8 + 32 * 9, 8 + 32 * 6 + 5], seed=4); out["deltakv"] = d
    return out


@pytest.mark.parametrize("method", ["h2o", "quest", "streamingllm", "deltakv"])
def test_debug_decode_bounds_on_real_managers(method, monkeypatch):
    outs = {}
    for checked in (False, True):
        monkeypatch.setenv("SVLLM_DEBUG_DECODE_BOUNDS", "1" if checked else "0")
        drv = _drivers()[method]
        cm = drv.cache_manager
        o = torch.zeros((cm.num_layers, len(drv.seqs), 28, 128), dtype=torch.bfloat16, device=drv.device)
        for step in range(3):
            q, k, v = drv.random_step_inputs(seed=10 + step)
            drv.step(q, k, v, outputs=o)
        torch.cuda.synchronize()
        outs[checked] = o.view(torch.int16).cpu().numpy().copy()
        del drv
    np.testing.assert_array_equal(outs[False], outs[True])


def test_debug_decode_bounds_catches_a_corrupted_slot_table(monkeypatch):
    monkeypatch.setenv("SVLLM_DEBUG_DECODE_BOUNDS", "1")
    drv = _drivers()["h2o"]
    cm = drv.cache_manager
    row = cm.seq_id_to_row[1][drv.seqs[1].seq_id]
    cm.buffer_req_to_token_slots_tensor[1, row, 17] = 10 ** 6            # a wild slot id inside the visible range of layer 1
    q, k, v = drv.random_step_inputs(seed=5)
    with pytest.raises(RuntimeError, match=r"decode physical slot out of bounds before attention: batch=1 req_row=\d+ pos=17 slot=1000000"):
        drv.step(q, k, v)
    monkeypatch.setenv("SVLLM_DEBUG_DECODE_BOUNDS", "0")                # unchecked, the kernels would have trusted it


@pytest.mark.parametrize("method", ["h2o", "quest", "streamingllm", "deltakv"])
@pytest.mark.parametrize("graph", [False, True])
def test_device_side_slot_guard_on_real_managers(method, graph, monkeypatch):
    """`SVLLM_DEBUG_DECODE_BOUNDS=device` (svk_check_slot_table): the reference's bounds check as a launch of the step - no
    synchronisation, captured into the step's hipGraph.  Clean steps of every method leave the record clean and the outputs
    bit-identical to the unchecked run; a slot table corrupted behind the captured graph's back is reported with the
    reference's message and coordinates by `raise_if_slot_check_failed` (the read is where the caller synchronises anyway)."""
    from sparse_vllm_amd.kernels.store_kvcache import raise_if_slot_check_failed, slot_check_status
    outs = {}
    for mode in ("0", "device"):
        monkeypatch.setenv("SVLLM_DEBUG_DECODE_BOUNDS", mode)
        drv = _drivers()[method]
        cm = drv.cache_manager
        if graph:
            drv.enable_decode_graph()
        o = torch.zeros((cm.num_layers, len(drv.seqs), 28, 128), dtype=torch.bfloat16, device=drv.device)
        q, k, v = drv.random_step_inputs(seed=20)
        for step in range(4):
            drv.step(q, k, v, outputs=o)
        torch.cuda.synchronize()
        outs[mode] = o.view(torch.int16).cpu().numpy().copy()
        if mode == "device":
            raise_if_slot_check_failed(drv.device)                                 # clean
            assert int(slot_check_status(drv.device)[0]) == 0
            if method == "h2o":
                row = cm.seq_id_to_row[1][drv.seqs[1].seq_id]
                wild = int(cm.kv_cache.shape[2]) + 3                               # just past the pool: recorded, and harmless to read
                saved = int(cm.buffer_req_to_token_slots_tensor[1, row, 17])
                cm.buffer_req_to_token_slots_tensor[1, row, 17] = wild
                drv.step(q, k, v, outputs=o)                                       # (a replayed graph when `graph`)
                cm.buffer_req_to_token_slots_tensor[1, row, 17] = saved
                with pytest.raises(RuntimeError, match=rf"decode physical slot out of bounds before attention: batch=1 req_row={row} pos=17 slot={wild}"):
                    raise_if_slot_check_failed(drv.device)
                raise_if_slot_check_failed(drv.device)                             # the record was cleared
        del drv
    np.testing.assert_array_equal(outs["0"], outs["device"])


def test_slot_guard_kinds_and_page_slots():
    """svk_check_slot_table at kernel level: a request row outside the table, a visible length beyond the table's width, a
    negative slot, page-slot tables (Quest's view: ids are PAGE slots, lengths in tokens); the first violation stays."""
    from sparse_vllm_amd.kernels.store_kvcache import check_slot_table_async
    d = torch.device("cuda:0")
    tab = torch.arange(4 * 64, dtype=torch.int32, device=d).reshape(4, 64) % 100
    i32 = lambda *x: torch.tensor(x, dtype=torch.int32, device=d)

    def run(tab, req, lens, cap, page=0):
        st = torch.zeros(8, dtype=torch.int32, device=d)
        check_slot_table_async(tab, req, lens, slot_cap=cap, slot_page_size=page, status=st)
        return st.cpu().tolist()[:6]

    assert run(tab, i32(0, 3), i32(64, 10), 100)[0] == 0
    assert run(tab, i32(0, 4), i32(64, 10), 100)[:3] == [1, 1, 4]                          # row 4 of a 4-row table
    assert run(tab, i32(0, 3), i32(64, 65), 100)[:3] == [2, 1, 3] and run(tab, i32(0, 3), i32(64, 65), 100)[5] == 65
    bad = tab.clone(); bad[2, 5] = -1
    assert run(bad, i32(2), i32(6), 100) == [3, 0, 2, 5, -1, 6]
    assert run(bad, i32(2), i32(5), 100)[0] == 0                                           # position 5 is not visible at length 5
    bad = tab.clone(); bad[1, 9] = 100
    assert run(bad, i32(1), i32(64), 100)[:5] == [3, 0, 1, 9, 100]
    # page slots: 16-token pages, 10 pages visible at length 150; the table's width is in pages
    pages = (torch.arange(4 * 16, dtype=torch.int32, device=d).reshape(4, 16)) % 30
    assert run(pages, i32(0, 1), i32(150, 256), 30, page=16)[0] == 0
    assert run(pages, i32(0, 1), i32(150, 257), 30, page=16)[:3] == [2, 1, 1]              # 17 pages > width 16
    bad = pages.clone(); bad[0, 9] = 30
    assert run(bad, i32(0), i32(150), 30, page=16)[:5] == [3, 0, 0, 9, 30]
    assert run(bad, i32(0), i32(144), 30, page=16)[0] == 0                                 # 9 pages visible


@pytest.mark.parametrize("method", ["h2o", "quest", "deltakv"])
def test_store_riding_in_stage1_equals_store_then_launch(method, monkeypatch):
    res = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("SVK_FUSE_DECODE_STORE", fuse)
        monkeypatch.setenv("SVK_DELTAKV_FUSE_FULL_STORE", fuse)
        drv = _drivers()[method]
        cm = drv.cache_manager
        taken = []
        orig = cm.take_deferred_decode_store
        cm.take_deferred_decode_store = lambda l, _o=orig, _t=taken: (_t.append(l), _o(l))[1]
        o = torch.zeros((cm.num_layers, len(drv.seqs), 28, 128), dtype=torch.bfloat16, device=drv.device)
        for step in range(3):
            q, k, v = drv.random_step_inputs(seed=20 + step)
            drv.step(q, k, v, outputs=o)
        torch.cuda.synchronize()
        assert cm.__dict__.get("_deferred_decode_store") is None          # nothing left behind after a step
        caches = [t for t in (getattr(cm, "kv_cache", None), getattr(cm, "full_kv_cache", None)) if t is not None]
        res[fuse] = (o.view(torch.int16).cpu().numpy().copy(), [c.view(torch.int16).cpu().numpy().copy() for c in caches], len(taken))
        del drv
    np.testing.assert_array_equal(res["1"][0], res["0"][0])
    for a, b in zip(res["1"][1], res["0"][1]):
        np.testing.assert_array_equal(a, b)
    assert res["1"][2] > 0


def test_flush_of_rows_no_launch_took(monkeypatch):
    monkeypatch.setenv("SVK_FUSE_DECODE_STORE", "1")
    drv = _drivers()["h2o"]
    cm = drv.cache_manager
    from sparse_vllm_amd.utils.context import set_context
    cm.prepare_decode_static(drv.seqs)
    set_context(False, cache_manager=cm, sparse_controller=drv.sparse_controller)
    q, k, v = drv.random_step_inputs(seed=7)
    cm.save_rope_kv_if_needed(0, k[0], v[0])
    assert cm._deferred_decode_store is not None
    slots = cm.get_layer_batch_states(0).slot_mapping.long()
    cm.save_rope_kv_if_needed(1, k[1], v[1])                            # the next store arrives: layer 0's rows are written first
    torch.cuda.synchronize()
    assert torch.equal(cm.kv_cache[0, 0][slots], k[0]) and torch.equal(cm.kv_cache[1, 0][slots], v[0])
    cm.flush_deferred_decode_store()
    torch.cuda.synchronize()
    slots1 = cm.get_layer_batch_states(1).slot_mapping.long()
    assert torch.equal(cm.kv_cache[0, 1][slots1], k[1]) and cm._deferred_decode_store is None


def test_e2e_decoder_two_layer_smoke():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "e2e_decoder.py"), "--layers", "2", "--batches", "2", "--steps", "6",
                        "--warmup", "3", "--modes", "graph,eager"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert [l["launch"] for l in lines] == ["hipGraph replay", "eager"] and all("error" not in l for l in lines)
    g, e = lines
    assert g["batch_per_group"] == 2 and g["tp"] == 1 and g["layers"] == 2 and g["value"] > 0
    assert g["graph_steps"]["replayed"] >= 6
    assert g["sample_tokens"] == e["sample_tokens"]                      # same greedy tokens, replayed or eagerly launched
