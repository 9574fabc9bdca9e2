"""Batch sharding across ranks (one process per GPU).  The path has no exchange step: every QP is
independent (all solver state is per instance, /root/reference/src/fcc_qp.hpp:130-171), so ranks
only meet at the timing barriers and at the MAX reduction of the elapsed time."""
from __future__ import annotations

from typing import Tuple


def shard_bounds(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous split of `total` instances: rank g owns [g*total/world, (g+1)*total/world)."""
    return (total * rank) // world, (total * (rank + 1)) // world


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
