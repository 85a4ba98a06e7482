"""N>1 path on CPU: two gloo ranks shard the sequences and agree on (sum tokens, max time)."""

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sparse_vllm_amd.replicas import aggregate_throughput, shard_sequences
    mine = shard_sequences(11, rank, world)
    tokens, seconds = aggregate_throughput(len(mine) * 10, 1.0 + rank)
    out.put((rank, mine, tokens, seconds))
    dist.barrier()
    dist.destroy_process_group()


def test_two_replicas_shard_and_aggregate():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, tok0, t0), (r1, s1, tok1, t1) = res
    assert s0 == [0, 2, 4, 6, 8, 10] and s1 == [1, 3, 5, 7, 9]
    assert sorted(s0 + s1) == list(range(11))                  # disjoint cover, no collective on the data path
    assert tok0 == tok1 == 110 and t0 == t1 == 2.0             # SUM of tokens, MAX of time


def test_single_process_is_identity():
    from sparse_vllm_amd.replicas import aggregate_throughput, shard_sequences
    assert aggregate_throughput(5, 0.5) == (5, 0.5)
    assert shard_sequences(4, 0, 1) == [0, 1, 2, 3]
    with pytest.raises(ValueError):
        shard_sequences(4, 2, 2)
