"""Sequence-sharded replicas: the only multi-GPU form of this path.

Every hot-path structure is per (sequence row, layer) and the reference has no data parallelism
for Qwen2 (docs/en/features/supported-models.md:15), so N GPUs = N independent
replicas, one process per GPU, each with its own cache manager.  There is NO data-path collective;
`torch.distributed` (RCCL on GPUs, gloo in the CPU tests) is used only to agree on the timing:
barrier, MAX over ranks of the elapsed time, SUM of processed tokens.
"""

from __future__ import annotations

import torch
import torch.distributed as dist


def shard_sequences(num_seqs: int, rank: int, world_size: int) -> list[int]:
    """Round-robin partition of request ids over replicas (ids rank, rank+W, ...)."""
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of {world_size}")
    return list(range(rank, int(num_seqs), world_size))


def aggregate_throughput(local_tokens: int, local_seconds: float, *, device="cpu", group=None) -> tuple[int, float]:
    """-> (total tokens over all ranks, max elapsed seconds over ranks).  Single process: identity."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return int(local_tokens), float(local_seconds)
    t = torch.tensor([float(local_seconds)], dtype=torch.float64, device=device)
    n = torch.tensor([int(local_tokens)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(n, op=dist.ReduceOp.SUM, group=group)
    return int(n.item()), float(t.item())


def gather_per_rank(local_value: float, *, device="cpu", group=None) -> list[float]:
    """-> every rank's value, rank order, on every rank (one all-gather of a float64).  Single process: [value].  The
    length of the list is the number of ranks that really took part - bench.py reports it as `ranks_seen`."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [float(local_value)]
    mine = torch.tensor([float(local_value)], dtype=torch.float64, device=device)
    parts = [torch.zeros_like(mine) for _ in range(dist.get_world_size(group))]
    dist.all_gather(parts, mine, group=group)
    return [float(p.item()) for p in parts]
