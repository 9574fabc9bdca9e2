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
        # the two global batches bench.py also times split over the ranks (strong_batch4096, config4_65536)
        plans = [S.shard_plan(gb, world, rank) for _, gb in S.SHARDED_RECORDS]
        covered = [S.sum_over_ranks(float(p[1] - p[0])) for p in plans]
        first = P.workload("W-sat", 2, P.WALKING, seed0=plans[0][2])           # the first instances of this rank's shard
        q.put((rank, seed0, lo, hi, float(B.b[0, 0]), worst, solved, plans, covered, float(first.b[0, 0])))
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
    (r0, s0, lo0, hi0, b0, w0, n0, pl0, cov0, f0), (r1, s1, lo1, hi1, b1, w1, n1, pl1, cov1, f1) = out
    assert (s0, s1) == (1000, 1008)                     # weak scaling: disjoint seeds per rank
    assert (lo0, hi0, lo1, hi1) == (0, 8, 8, 16)        # contiguous split of the global batch
    assert b0 != b1                                     # ... hence different instances
    assert w0 == w1 == 0.5                              # MAX over ranks
    assert n0 == n1 == 48.0                             # SUM over ranks (units all ranks processed)
    # one global batch of 4096 / 65536 split over the two ranks: contiguous, complete, seeds = 1000 + global index
    assert pl0 == [(0, 2048, 1000, None), (0, 32768, 1000, 256)]
    assert pl1 == [(2048, 4096, 3048, None), (32768, 65536, 33768, 256)]
    assert cov0 == cov1 == [4096.0, 65536.0]
    assert f0 != f1                                     # rank 1's shard starts at instance 2048, not at 0


def test_shard_plans_partition_the_global_batches():
    """strong_batch4096 / config4_65536 at the world sizes the driver runs: every instance exactly once, in order."""
    for _, total in S.SHARDED_RECORDS:
        for world in (1, 2, 4, 8):
            plans = [S.shard_plan(total, world, r) for r in range(world)]
            assert plans[0][0] == 0 and plans[-1][1] == total
            assert all(plans[r][1] == plans[r + 1][0] for r in range(world - 1))
            assert all(p[1] - p[0] == total // world and p[2] == 1000 + p[0] for p in plans)
    assert S.shard_plan(4096, 8, 3) == (1536, 2048, 2536, None)          # 512 per GPU: the launch-bound case of SURVEY 8e
    assert S.shard_plan(65536, 8, 7) == (57344, 65536, 58344, 256)       # 8192 per GPU


def test_single_process_helpers_are_identity():
    assert S.max_over_ranks(1.5) == 1.5 and S.sum_over_ranks(2.0) == 2.0
    S.barrier()
    assert S.shard_bounds(4096, 8, 3) == (1536, 2048)
    assert [S.shard_bounds(10, 3, r) for r in range(3)] == [(0, 3), (3, 6), (6, 10)]
