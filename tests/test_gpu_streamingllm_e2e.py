"""End-to-end StreamingLLM decode on the GPU through the reference's operator surface (CacheManager.create ->
prepare_decode_static -> Attention.forward per layer -> SparseController.post_forward) against the numpy oracle
driven step by step on the same inputs: attention outputs within tolerance, slot tables / free stacks / lengths
bit-exact across sink + recent compactions (sparse_controller.py:1558-1653, snapkv.py:1805-1896), eager and under
hipGraph replay with the physical-peak context capacity of StreamingLLMCacheManager."""

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
    dict(B=3, L=3, sink=4, recent=12, start=16, steps=40, Hq=28, Hkv=4, D=128),
    dict(B=2, L=2, sink=8, recent=24, start=40, steps=60, Hq=14, Hkv=2, D=64),
])
def test_streamingllm_decode_steps_match_oracle(cfg, graph):
    from sparse_vllm_amd.config import Config
    from sparse_vllm_amd.engine.decode_driver import SparseDecodeDriver
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
    assert compactions >= 1
