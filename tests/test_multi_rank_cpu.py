"""The N > 1 path of bench.py on CPU: world_size 2 over gloo.  There is no data-path collective
to test (the batch shards embarrassingly); what the ranks share is the shard arithmetic, the
barrier and the MAX / SUM reductions behind `value`."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fcc_qp_amd import problems as P
from fcc_qp_amd import sharding as S


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        per_rank = 8
        seed0 = S.rank_seed0(rank, per_rank)
        B = P.workload("W-easy", per_rank, P.WALKING, seed0=seed0)
        lo, hi = S.shard_bounds(world * per_rank, world, rank)
        S.barrier()
        elapsed = 0.25 * (rank + 1)                     # rank 1 is "slower"
        worst = S.max_over_ranks(elapsed)
        solved = S.sum_over_ranks(float(per_rank * 3))
        q.put((rank, seed0, lo, hi, float(B.b[0, 0]), worst, solved))
    finally:
        dist.destroy_process_group()


def test_two_ranks_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, lo0, hi0, b0, w0, n0), (r1, s1, lo1, hi1, b1, w1, n1) = out
    assert (s0, s1) == (1000, 1008)                     # weak scaling: disjoint seeds per rank
    assert (lo0, hi0, lo1, hi1) == (0, 8, 8, 16)        # contiguous split of the global batch
    assert b0 != b1                                     # ... hence different instances
    assert w0 == w1 == 0.5                              # MAX over ranks
    assert n0 == n1 == 48.0                             # SUM over ranks (units all ranks processed)


def test_single_process_helpers_are_identity():
    assert S.max_over_ranks(1.5) == 1.5 and S.sum_over_ranks(2.0) == 2.0
    S.barrier()
    assert S.shard_bounds(4096, 8, 3) == (1536, 2048)
    assert [S.shard_bounds(10, 3, r) for r in range(3)] == [(0, 3), (3, 6), (6, 10)]
