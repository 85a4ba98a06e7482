"""BASELINE.json configs[0] on the GPU: `sparse_method=""` (vanilla, StandardCacheManager) with Qwen2.5-0.5B head shapes
(14 query / 2 KV heads x 64), a 2 k-token prompt prefilled in chunks and then decoded, through the reference's operator
surface (CacheManager.create -> _prepare_prefill / prepare_decode_static -> Attention.forward per layer -> post_forward)
against the numpy oracle chained over a mirror of the slot state: chunk attention and decode outputs within the
attention tolerance (rtol = atol = 2e-2), slot tables / free stacks / lengths bit-exact, nothing ever evicted, and
free_seq returns every slot (engine/cache_manager/standard.py, snapkv.py:1319-1340, :1489-1514)."""

import numpy as np
import pytest

from oracle import bf16_round
from oracle import decode_attention as oda
from oracle import h2o as oh
from oracle import prefill_attention as opa

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _f(t):
    return t.float().cpu().numpy()


@pytest.mark.parametrize("graph", [False, True])
def test_vanilla_prefill_then_decode_matches_oracle(graph):
    from sparse_vllm_amd.config import Config
    from sparse_vllm_amd.engine.cache_manager.standard import StandardCacheManager
    from tools.synthetic import SyntheticDecodeDriver as SparseDecodeDriver
    from sparse_vllm_amd.engine.sequence import Sequence
    L, Hq, Hkv, D = 2, 14, 2, 64
    prompts, chunk, steps = (2048, 700), 512, 12
    conf = Config.from_kwargs(sparse_method="vanilla", num_hidden_layers=L, num_attention_heads=Hq, num_key_value_heads=Hkv,
                              head_dim=D, max_model_len=2048 + 64, max_num_seqs_in_gpu=3, num_kvcache_slots=3000,
                              engine_prefill_chunk_size=chunk)
    assert conf.vllm_sparse_method == ""
    drv = SparseDecodeDriver(conf)
    cm = drv.cache_manager
    assert type(cm) is StandardCacheManager
    cm.permute_free_slots(5)
    seqs = [Sequence(num_prompt_tokens=n) for n in prompts]
    st = oh.SlotState(cm.buffer_req_to_token_slots_tensor.cpu().numpy().copy(), cm.free_slots_stack_tensor.cpu().numpy().copy(),
                      np.asarray(cm._num_free_slots, dtype=np.int64), np.stack(cm.row_seq_lens).astype(np.int32))
    kc, vc = _f(cm.kv_cache[0]).copy(), _f(cm.kv_cache[1]).copy()
    g = torch.Generator().manual_seed(11)
    mk = lambda n, h: (torch.randn(L, n, h, D, generator=g) * 0.4).to(torch.bfloat16).to(drv.device)

    def check_state():
        np.testing.assert_array_equal(np.stack(cm.row_seq_lens), st.row_len)
        np.testing.assert_array_equal(np.asarray(cm._num_free_slots), st.free_ptr)
        np.testing.assert_array_equal(cm.buffer_req_to_token_slots_tensor.cpu().numpy(), st.slot_table)
        stack = cm.free_slots_stack_tensor.cpu().numpy()
        for l in range(L):
            p = int(st.free_ptr[l])
            np.testing.assert_array_equal(stack[l, :p], st.free_stack[l, :p])

    # ---------------- chunked prefill
    while any(s.num_prefilled_tokens < s.num_prompt_tokens for s in seqs):
        active = [s for s in seqs if s.num_prefilled_tokens < s.num_prompt_tokens]
        for s in active:
            s.current_chunk_size = min(chunk, s.num_prompt_tokens - s.num_prefilled_tokens)
        lens = [s.current_chunk_size for s in active]
        tot = sum(lens)
        q, k, v = mk(tot, Hq), mk(tot, Hkv), mk(tot, Hkv)
        outs = torch.zeros_like(q)
        drv.prefill_chunk(active, q, k, v, outputs=outs)
        torch.cuda.synchronize()
        rows = [cm.seq_id_to_row[0][s.seq_id] for s in active]
        starts = np.concatenate(([0], np.cumsum(lens)[:-1])).astype(np.int32)
        qn, kn, vn = _f(q), _f(k), _f(v)
        for l in range(L):
            ctx, cache = [], []
            for s, r, n, s0 in zip(active, rows, lens, starts):
                prev = int(st.row_len[l, r])
                new = oh.allocate(st, l, r, n)
                kc[l][new] = kn[l][s0:s0 + n]
                vc[l][new] = vn[l][s0:s0 + n]
                ctx.append(prev + n)
                cache.append(prev)
            ref = opa.context_attention_fwd(qn[l], kc[l], vc[l], np.array(rows, np.int32), starts, np.array(ctx, np.int32),
                                            np.array(cache, np.int32), st.slot_table[l])
            np.testing.assert_allclose(_f(outs[l]), bf16_round(ref), rtol=2e-2, atol=2e-2)
        check_state()
    np.testing.assert_array_equal(_f(cm.kv_cache[0]), kc)            # store_kvcache: exact copies
    np.testing.assert_array_equal(_f(cm.kv_cache[1]), vc)

    # ---------------- decode
    for s in seqs:
        s.num_tokens = s.num_prompt_tokens
    drv.seqs = seqs
    if graph:
        drv.enable_decode_graph()
    rows = [cm.seq_id_to_row[0][s.seq_id] for s in seqs]
    B = len(seqs)
    outs = torch.zeros((L, B, Hq, D), dtype=torch.bfloat16, device=drv.device)
    for step in range(steps):
        q, k, v = drv.random_step_inputs(seed=100 + step, scale=0.4)
        drv.step(q, k, v, outputs=outs)
        torch.cuda.synchronize()
        new_slots = oh.decode_allocate_batch_layers(st, range(L), rows)
        lens = np.array([st.row_len[0, r] for r in rows], dtype=np.int32)
        qn, kn, vn = _f(q), _f(k), _f(v)
        for l in range(L):
            kc[l][new_slots[l]] = kn[l]
            vc[l][new_slots[l]] = vn[l]
            mid, lse = oda.flash_decode_stage1(qn[l], kc[l], vc[l], st.slot_table[l], np.array(rows, np.int32), lens,
                                               int(lens.max()), 256)
            o = oda.flash_decode_stage2(mid, lse, lens, 256)
            np.testing.assert_allclose(_f(outs[l]), bf16_round(o), rtol=2e-2, atol=2e-2)
        check_state()
    assert [int(x) for x in st.row_len[0, rows]] == [n + steps for n in prompts]      # vanilla never evicts
    with pytest.raises(RuntimeError):
        cm.free_part_slots(0, seqs[0], torch.arange(4, device=drv.device))
    # ---------------- release: every slot returns to the free stack
    for s, r in zip(seqs, rows):
        cm.free_seq(s.seq_id)
        oh.free_seq(st, range(L), r)
    check_state()
    assert all(int(n) == cm.num_slots for n in cm._num_free_slots)
    stack = cm.free_slots_stack_tensor.cpu().numpy()
    for l in range(L):
        assert np.array_equal(np.sort(stack[l]), np.arange(cm.num_slots))
