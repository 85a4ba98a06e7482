"""CPU, two gloo ranks: the tensor-parallel sharding of the end-to-end decode leg (tools/e2e_decoder.py).

A 2-layer toy of the Qwen2.5 shape family (4 q / 2 kv heads, vocab-sharded head, QKV bias) is decoded for several greedy
steps by ONE rank with tp=1 and by TWO gloo ranks with tp=2 (heads, MLP columns and vocab rows sharded; all-reduce after
o_proj and down_proj, all-gather of the per-shard (max logit, index) pairs - the collectives that run over RCCL / xGMI on
a GPU node).  Same seed, same weights: the two runs must produce the same token sequence and the same winning logits (to
fp32 summation-order noise).  The attention itself is a stand-in here (each query head reads its KV head's current value
row); on a GPU the hook is this build's `Attention` (tools/e2e_decoder.py `_build_sparse_driver`)."""

import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOY = dict(hidden=64, layers=2, q_heads=4, kv_heads=2, head_dim=16, inter=96, vocab=128, rope_theta=1e4, rms_eps=1e-6)
STEPS, BATCH = 6, 3


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _decode(model, tp_rank, tp_size):
    tokens = torch.tensor([5, 77, 120])
    positions = torch.tensor([3, 9, 200])
    G = TOY["q_heads"] // TOY["kv_heads"]

    def attention(layer, q, k, v):              # stand-in: head h reads v of its KV head (no cache on CPU)
        return v.repeat_interleave(G, dim=1) + 0.01 * q

    out = []
    for _ in range(STEPS):
        tokens = model.step(tokens, positions, attention)
        positions = positions + 1
        out.append(tokens.tolist())
    return out


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tools.e2e_decoder import QwenShapedDecoder
    model = QwenShapedDecoder(TOY, tp_rank=rank, tp_size=world, device="cpu", dtype=torch.float32, seed=3, max_positions=256)
    q.put((rank, _decode(model, rank, world), model.weight_bytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_tp2_on_gloo_equals_tp1():
    sys.path.insert(0, ROOT)
    from tools.e2e_decoder import QwenShapedDecoder
    single = QwenShapedDecoder(TOY, device="cpu", dtype=torch.float32, seed=3, max_positions=256)
    want = _decode(single, 0, 1)
    assert len({tuple(t) for t in want}) > 1                       # the toy does move between tokens

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, t0, w0), (_, t1, w1) = got
    assert t0 == t1 == want                                         # both ranks agree with each other and with tp=1
    # a rank holds half of every sharded matrix (norm weights are replicated)
    norms = (2 * TOY["layers"] + 1) * TOY["hidden"] * 4
    assert w0 == w1 and abs((w0 - norms) * 2 - (single.weight_bytes() - norms)) == 0


def test_shape_checks_and_bound():
    sys.path.insert(0, ROOT)
    from tools.e2e_decoder import QWEN25_7B, QwenShapedDecoder, bound_ms
    with pytest.raises(ValueError, match="not divisible"):
        QwenShapedDecoder(TOY, tp_size=3, device="cpu", dtype=torch.float32)
    assert QWEN25_7B["q_heads"] % 4 == 0 and QWEN25_7B["kv_heads"] % 4 == 0 and QWEN25_7B["inter"] % 4 == 0 and QWEN25_7B["vocab"] % 4 == 0
    # B=1 -> 1.79 ms, B=64 -> 3.71 ms
    assert abs(bound_ms(1, 1) - 1.79) < 0.01 and abs(bound_ms(64, 1) - 3.71) < 0.01
    # parameter count of the real shape: 7.07 B read per step, without building it
    s = QWEN25_7B
    per_layer = (s["q_heads"] + 2 * s["kv_heads"]) * s["head_dim"] * s["hidden"] + s["hidden"] * s["q_heads"] * s["head_dim"] \
        + 3 * s["inter"] * s["hidden"]
    assert abs((per_layer * s["layers"] + s["vocab"] * s["hidden"]) / 1e9 - 7.07) < 0.01
