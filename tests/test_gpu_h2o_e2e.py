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
@pytest.mark.parametrize("defer", ["0", "1", "end"])
def test_h2o_decode_steps_match_oracle(cfg, defer, monkeypatch):
    # the score epilogue of a layer either runs in the fused finish launch ("0") or rides in the next layer's stage-1
    # launch ("1"), or all layers of the step run in one launch after the layer loop ("end", the default)
    monkeypatch.setenv("SVK_H2O_DEFER_SCORE", defer)
    from sparse_vllm_amd.config import Config
    from sparse_vllm_amd.engine.decode_driver import SparseDecodeDriver
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
    if defer == "end":
        # the layers' buffers are equally spaced slices: every step took the single all-layers launch
        from sparse_vllm_amd.kernels import h2o_ops
        assert h2o_ops.SCORE_LAYERS_LAUNCHES["batched"] > before["batched"] and h2o_ops.SCORE_LAYERS_LAUNCHES["per_layer"] == before["per_layer"]
