"""DeltaKV prompts compressed natively: raw K/V of a prompt chunk goes through `_prepare_prefill` + the store hooks and
`SparseController.post_forward(prefill)` -> `deltakv_evict` (bulk: several `recent` blocks at once) + the KIVI
eviction (deltakv_less_memory.py:3602-3780, :3495-3558), against the oracle's cluster compression chained on the same
rows: fathers are a valid causal top-k of the oracle's L2 scores, the latent decodes to compress_down(kv - mean) within
the int4 step, the KIVI blocks are the oracle's bit for bit, and decode runs on from the produced state."""

import numpy as np
import pytest

from oracle import bf16_round
from oracle import deltakv as od
from oracle import deltakv_compress as oc
from oracle.quest import check_topk_set

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def f32(t):
    return t.float().cpu().numpy()


def _driver(bits, kivi, chunk):
    from sparse_vllm_amd.config import Config
    from tools.synthetic import SyntheticDecodeDriver as SparseDecodeDriver
    conf = Config.from_kwargs(
        sparse_method="deltakv", num_hidden_layers=4, full_attention_layers="0,2", num_attention_heads=8,
        num_key_value_heads=2, head_dim=64, max_model_len=512, max_num_seqs_in_gpu=3, sink_keep_tokens=4,
        recent_keep_tokens=8, decode_keep_tokens=12, deltakv_neighbor_count=2, deltakv_latent_dim=32,
        deltakv_latent_quant_bits=bits, deltakv_latent_quant_group_size=16, deltakv_center_ratio=0.25,
        allow_missing_deltakv_path=True, compressor_up_type="linear", full_layer_kv_quant_bits=4 if kivi else 0,
        full_layer_kivi_decode_block_seq=64, rope_theta=10000.0, engine_prefill_chunk_size=chunk)
    return SparseDecodeDriver(conf)


@pytest.mark.parametrize("bits,kivi", [(4, True), (0, False)])
def test_whole_prompt_chunk_is_compressed_like_the_oracle(bits, kivi):
    from sparse_vllm_amd.engine.sequence import Sequence
    drv = _driver(bits, kivi, 256)
    cm = drv.cache_manager
    cm.permute_free_slots(5)
    L, Hkv, D, sink, recent, step, Kf = 4, 2, 64, 4, 8, 4, 2
    n = 4 + 8 * 9 + 5                                    # 81 tokens: 9 blocks leave the raw tail, 8 + 5 stay
    seq = Sequence(num_prompt_tokens=n)
    seq.current_chunk_size = n
    g = torch.Generator().manual_seed(11)
    mk = lambda h: (torch.randn(L, n, h, D, generator=g) * 0.5).to(torch.bfloat16).to(drv.device)
    q, k, v = mk(8), mk(Hkv), mk(Hkv)
    drv.prefill_chunk([seq], q, k, v)
    torch.cuda.synchronize()
    row = cm.seq_id_to_row[seq.seq_id]
    evict_len = ((n - sink - recent) // recent) * recent
    assert evict_len == 64 and int(cm.row_deltakv_compressed_lens[row]) == evict_len and int(cm.row_seq_lens[row]) == n
    raw_map = cm.sparse_layer_raw_slots_map[row].cpu().numpy()
    lat_map = cm.sparse_layer_latent_slots_map[row].cpu().numpy()
    rel = np.arange(0, evict_len, step)
    is_center = np.zeros(evict_len, bool)
    is_center[rel] = True
    assert (raw_map[:sink] >= 0).all() and (raw_map[sink + evict_len:n] >= 0).all()
    np.testing.assert_array_equal(raw_map[sink:sink + evict_len] >= 0, is_center)
    assert (lat_map[sink:sink + evict_len] >= 0).all() and (lat_map[sink + evict_len:n] < 0).all()
    center_slots = np.concatenate((raw_map[:sink], raw_map[sink:sink + evict_len][is_center]))
    np.testing.assert_array_equal(cm.row_deltakv_center_slots[row].cpu().numpy(), center_slots)
    lat = lat_map[sink:sink + evict_len]
    kf, vf = f32(k), f32(v)
    for l in cm.deltakv_layer_ids:
        i = cm.deltakv_layer_to_idx[l]
        rows_kv = np.concatenate((kf[l].reshape(n, -1), vf[l].reshape(n, -1)), axis=1)      # [n, 2*Hkv*D] concat(K_raw, V)
        block = rows_kv[sink:sink + evict_len]
        scores, topk_ref, base_ref, allc = oc.cluster_compress(block, rows_kv[:sink], rel, Kf)
        fathers = cm.deltakv_latent_to_full_slots[i][torch.from_numpy(lat).long().to(drv.device)].cpu().numpy()
        col_of = {int(s): c for c, s in enumerate(center_slots)}
        got_cols = np.vectorize(col_of.__getitem__)(fathers)
        for r in range(evict_len):
            finite = np.isfinite(scores[r])
            check_topk_set(scores[r], got_cols[r], Kf, atol=float(np.abs(scores[r][finite]).max() * 2.0 ** -7))
        # residual = down(kv) - down(mean of the GPU's own fathers): recompute with the fathers the GPU chose
        base = bf16_round(allc[got_cols].astype(np.float32).mean(axis=1))
        # compress_down itself is a library GEMM chain (not under test): evaluate the module on the oracle's rows
        down = cm.compress_down[i]
        enc = lambda x: f32(down(torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(torch.bfloat16).to(drv.device)))
        residual = bf16_round(enc(block) - enc(base))
        if bits:
            code = cm.deltakv_latent_cache[i][torch.from_numpy(lat).long().to(drv.device)].cpu().numpy()
            sc = f32(cm.deltakv_latent_scales[i][torch.from_numpy(lat).long().to(drv.device)])
            mn = f32(cm.deltakv_latent_mins[i][torch.from_numpy(lat).long().to(drv.device)])
            deq = od.dequantize_grouped(code, sc, mn, 16, 4)
            # within one quantisation step of the oracle residual (+ bf16 noise of the two library GEMMs)
            stepsz = np.repeat(sc, 16, axis=1)
            assert np.all(np.abs(deq - residual) <= 0.75 * stepsz + 0.02 * np.abs(residual).max())
        else:
            got = f32(cm.deltakv_latent_cache[i][torch.from_numpy(lat).long().to(drv.device)])
            np.testing.assert_allclose(got, residual, rtol=3e-2, atol=0.02 * np.abs(residual).max())
    if kivi:
        G = 32
        qend = int(cm.row_full_layer_kivi_quantized_lens[row])
        assert qend == sink + ((n - sink - int(drv.config.full_layer_kivi_residual_length)) // G) * G and qend > sink
        blk = cm.full_layer_kivi_block_slots_map[row].cpu().numpy()
        nb = (qend - sink) // G
        ids = blk[sink:qend:G]
        assert len(set(ids.tolist())) == nb and (blk[qend:n] < 0).all()
        for l in cm.full_layer_ids:
            i = cm.full_layer_to_idx[l]
            ref = oc.kivi_quantize_blocks(kf[l][sink:qend].reshape(nb, G, Hkv, D), vf[l][sink:qend].reshape(nb, G, Hkv, D),
                                          G, oc.rounding("bf16"))
            for name, r in ref.items():
                t_ = getattr(cm, f"full_layer_kivi_{name}")[i][torch.from_numpy(ids).long().to(drv.device)]
                got = t_.cpu().numpy() if t_.dtype == torch.int32 else t_.float().cpu().numpy()
                np.testing.assert_array_equal(got, r, err_msg=f"layer {l} {name}")
    # decode goes on from the natively compressed state
    seq.num_tokens = n
    seq.num_prefilled_tokens = n
    drv.seqs = [seq]
    outs = torch.zeros((L, 1, 8, D), dtype=torch.bfloat16, device=drv.device)
    for stp in range(10):
        qd, kd, vd = drv.random_step_inputs(seed=70 + stp)
        drv.step(qd, kd, vd, outputs=outs)
    torch.cuda.synchronize()
    assert torch.isfinite(outs.float()).all()
    assert int(cm.row_deltakv_compressed_lens[row]) == ((n + 10 - sink - recent) // recent) * recent


def test_chunked_prompt_reaches_the_same_compressed_length_and_frees_every_slot():
    from sparse_vllm_amd.engine.sequence import Sequence
    drv = _driver(4, True, 48)
    cm = drv.cache_manager
    free0 = cm.free_slot_stats()
    L, Hkv, D, sink, recent = 4, 2, 64, 4, 8
    lens = [150, 97]
    seqs = [Sequence(num_prompt_tokens=x) for x in lens]
    g = torch.Generator().manual_seed(2)
    done = [0, 0]
    while any(d_ < x for d_, x in zip(done, lens)):
        batch = [(s, min(48, x - d_)) for s, x, d_ in zip(seqs, lens, done) if d_ < x]
        for s, c in batch:
            s.current_chunk_size = c
        tot = sum(c for _, c in batch)
        mk = lambda h: (torch.randn(L, tot, h, D, generator=g) * 0.5).to(torch.bfloat16).to(drv.device)
        drv.prefill_chunk([s for s, _ in batch], mk(8), mk(Hkv), mk(Hkv))
        for s, c in batch:
            done[seqs.index(s)] += c
        for s, d_ in zip(seqs, done):
            if s.seq_id in cm.seq_id_to_row:
                r = cm.seq_id_to_row[s.seq_id]
                tail = int(cm.row_seq_lens[r]) - sink - int(cm.row_deltakv_compressed_lens[r])
                assert int(cm.row_seq_lens[r]) == d_ and (tail <= 2 * recent - 1 or d_ <= sink + recent)
    for s, x in zip(seqs, lens):
        r = cm.seq_id_to_row[s.seq_id]
        assert int(cm.row_deltakv_compressed_lens[r]) == ((x - sink - recent) // recent) * recent
    for s in seqs:
        cm.free_seq(s.seq_id)
    assert cm.free_slot_stats() == free0


def _compressed_state(cm):
    """Everything the compression side of a prompt leaves behind (slot ids included: same pools, same seeds)."""
    keys = ("sparse_layer_raw_slots_map", "sparse_layer_latent_slots_map", "full_layer_slots_map", "deltakv_latent_cache",
            "deltakv_latent_scales", "deltakv_latent_mins", "deltakv_latent_to_full_slots", "deltakv_full_kv_cache",
            "full_kv_cache", "full_layer_kivi_key_packed", "full_layer_kivi_value_packed", "full_layer_kivi_block_slots_map")
    out = {}
    for k in keys:
        t = getattr(cm, k, None)
        if isinstance(t, torch.Tensor):
            out[k] = (t.view(torch.int16) if t.dtype == torch.bfloat16 else t).cpu().numpy().copy()
    out["lens"] = cm.row_seq_lens.copy()
    out["clens"] = cm.row_deltakv_compressed_lens.copy()
    return out


@pytest.mark.parametrize("bits,kivi", [(4, True), (0, False)])
def test_first_prefill_step_attends_the_chunk_through_the_staging_view(bits, kivi):
    """deltakv_base.py:936-972 / :1852-1993 ("full prefill" staging): in the first prefill step of a prompt the sparse layers
    attend the step's own post-RoPE K/V through the staging slot view and the full layers their raw rows; the pre-RoPE
    keys go to the raw slots and the chunk end compresses them.  Two prompts of different lengths in one step: outputs of
    every layer against the float64 causal attention of the oracle (rtol = atol = 2e-2, the reference's bar), and the
    compressed state bit-identical to the store-only run of the same prompts (the attention touches nothing)."""
    from oracle import prefill_attention as opa
    from sparse_vllm_amd.engine.sequence import Sequence
    L, Hq, Hkv, D = 4, 8, 2, 64
    lens = [4 + 8 * 9 + 5, 37]
    n = sum(lens)
    g = torch.Generator().manual_seed(21)
    mk = lambda h: (torch.randn(L, n, h, D, generator=g) * 0.5).to(torch.bfloat16)
    q, k_rope, k_raw, v = mk(Hq), mk(Hkv), mk(Hkv), mk(Hkv)
    states = []
    for with_outputs in (True, False):
        drv = _driver(bits, kivi, 256)
        cm = drv.cache_manager
        cm.permute_free_slots(5)
        seqs = []
        for m in lens:
            s = Sequence(num_prompt_tokens=m)
            s.current_chunk_size = m
            seqs.append(s)
        dev_ = lambda t: t.to(drv.device)
        out = torch.zeros((L, n, Hq, D), dtype=torch.bfloat16, device=drv.device) if with_outputs else None
        drv.prefill_chunk(seqs, dev_(q), dev_(k_rope), dev_(v), outputs=out, k_raw=dev_(k_raw))
        torch.cuda.synchronize()
        assert not cm.prefill_attention_view_supported           # the staging view lives for one step
        states.append(_compressed_state(cm))
        if with_outputs:
            cu = np.concatenate(([0], np.cumsum(lens)))
            table = np.full((2, max(lens)), -1, dtype=np.int64)
            for b, m in enumerate(lens):
                table[b, :m] = np.arange(cu[b], cu[b] + m)
            for l in range(L):
                ref = opa.context_attention_dense(f32(q[l]), f32(k_rope[l]), f32(v[l]), np.arange(2), cu[:2], np.array(lens),
                                                  np.zeros(2, np.int64), table)
                np.testing.assert_allclose(f32(out[l]), ref, rtol=2e-2, atol=2e-2, err_msg=f"layer {l}")
            row = cm.seq_id_to_row[seqs[0].seq_id]
            assert int(cm.row_deltakv_compressed_lens[row]) == 64       # ... and the chunk end still compressed the prompt
    for key in states[0]:
        np.testing.assert_array_equal(states[0][key], states[1][key], err_msg=key)


def test_prefill_attention_of_a_continuation_chunk_is_refused():
    """A continuation chunk's sparse layers attend a reconstructed, RoPE-rotated view in the reference
    (deltakv_base.py:936-972 -> deltakv_reconstruct(chunk_lens=...)), which this build does not have; the plain slot table
    holds pre-RoPE keys / slot -1 holes, so asking for its outputs must fail loudly instead of reading the wrong bytes.
    The first chunk of the same prompt does have a view (the staging view above)."""
    from sparse_vllm_amd.engine.sequence import Sequence
    drv = _driver(4, True, 256)
    seq = Sequence(num_prompt_tokens=80)
    seq.current_chunk_size = 40
    g = torch.Generator().manual_seed(1)
    mk = lambda h: (torch.randn(4, 40, h, 64, generator=g) * 0.5).to(torch.bfloat16).to(drv.device)
    q, k, v = mk(8), mk(2), mk(2)
    drv.prefill_chunk([seq], q, k, v, outputs=torch.zeros_like(q))
    seq.current_chunk_size = 40
    with pytest.raises(NotImplementedError, match="reconstructed prefill compute view"):
        drv.prefill_chunk([seq], q, k, v, outputs=torch.zeros_like(q))
    drv2 = _driver(4, True, 256)
    seq2 = Sequence(num_prompt_tokens=80)
    seq2.current_chunk_size = 40
    drv2.prefill_chunk([seq2], q, k, v)                      # store-only: fine
    seq2.current_chunk_size = 40
    drv2.prefill_chunk([seq2], q, k, v)
