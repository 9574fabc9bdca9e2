"""Batch sharding across ranks (one process per GPU).  The path has no exchange step: every QP is
independent (all solver state is per instance, /root/reference/src/fcc_qp.hpp:130-171), so ranks
only meet at the timing barriers and at the MAX reduction of the elapsed time."""
from __future__ import annotations

from typing import Tuple


def shard_bounds(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous split of `total` instances: rank g owns [g*total/world, (g+1)*total/world)."""
    return (total * rank) // world, (total * (rank + 1)) // world


#: the global batches bench.py times split over the ranks beside its weak-scaling headline (label, instances):
#: the metric's batch (4096 / G per GPU) and BASELINE configs[4] (65536: 8192 per GPU on 8 GPUs)
SHARDED_RECORDS = (("strong_batch4096", 4096), ("config4_65536", 65536))


def shard_plan(total: int, world: int, rank: int, base: int = 1000, tile_above: int = 2048, tile_from: int = 256):
    """What rank `rank` solves of ONE global batch of `total` instances (instance i: generator seed base + i):
    (lo, hi, seed0, distinct) -- its contiguous range, the seed of its first instance, and how many distinct
    instances it generates (None: all of them; shards above `tile_above` are tiled from their first `tile_from`)."""
    lo, hi = shard_bounds(total, world, rank)
    return lo, hi, base + lo, (None if hi - lo <= tile_above else tile_from)


def rank_seed0(rank: int, per_rank: int, base: int = 1000) -> int:
    """First generator seed of a rank under weak scaling (instance i of the global batch: base + i)."""
    return base + rank * per_rank


def max_over_ranks(value: float, device=None) -> float:
    """MAX of a python float over all ranks (identity without an initialised process group)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def barrier() -> None:
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
