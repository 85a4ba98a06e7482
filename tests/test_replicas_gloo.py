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


def test_bench_spawn_path_two_ranks_on_cpu():
    """`bench.py --gpus 2` started without a launcher spawns `torch.distributed.run` itself (one rank per GPU on a real
    node); with `--stub-driver` the same code path runs here on two gloo ranks whose step is a sleep of rank + 1 ms, so the
    launch, the rendezvous on 127.0.0.1, the barrier-bracketed timed region and the aggregation (SUM of tokens, MAX of time)
    are exercised end to end and the one JSON line rank 0 prints is checked."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    steps, batch = 6, 5
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", str(steps), "--warmup", "1",
                          "--batch", str(batch), "--stub-driver"], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout                         # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == steps and out["scaling"] == "weak" and out["higher_is_better"] is True
    assert out["config"]["seqs_per_gpu"] == batch and out["config"]["global_batch"] == 2 * batch
    assert "replicas x2" in out["config"]["parallelism"]
    # MAX over ranks: rank 1 sleeps 2 ms per step; SUM of tokens: 2 * batch * steps
    assert out["ms_per_step"] >= 2.0
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 * steps - 2 * batch * steps) < 1e-6 * 2 * batch * steps
    assert "roofline" not in out and "cpu_baseline" not in out  # N > 1: neither leg runs
    # what every replica measured on its own clock: both ranks reported, rank 1 (2 ms sleeps) is the slower one
    assert out["ranks_seen"] == 2 and len(out["per_rank_ms_per_step"]) == 2
    assert out["per_rank_ms_per_step"][1] >= 2.0 > 0.0 and out["per_rank_ms_per_step"][0] >= 1.0
    assert abs(max(out["per_rank_ms_per_step"]) - out["ms_per_step"]) < 0.5
    assert out["per_gpu_tokens_per_s"]["min"] <= out["per_gpu_tokens_per_s"]["max"]


def test_bench_refuses_mismatched_world_size():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--stub-driver"], capture_output=True,
                         text=True, timeout=300, env=env)
    assert res.returncode == 2 and "does not match WORLD_SIZE" in res.stderr
