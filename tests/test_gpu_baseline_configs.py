"""One test per entry of BASELINE.json `configs` (the headline metric itself is bench.py's job)."""
import numpy as np
import pytest

from fcc_qp_amd import problems as P
from tests import capi as CAPI
from tests.helpers import Z_TOL, assert_parity, oracle_batch

pytestmark = pytest.mark.gpu


def test_config0_single_instance_via_pybind(oracle):
    """configs[0]: single walking-shape QP through the pybind FCCQP class (reference call sequence)."""
    from fcc_qp import FCCQP, FCCQPOptions
    qp = P.osc_qp(1000, P.WALKING)
    s = FCCQP(60, 38, 12, 38)
    o = FCCQPOptions(); o.rho = 5e-5; o.eps_fcone = 1e-6; o.eps_bound = 1e-6; o.max_iter = 100
    s.set_options(o); s.set_warm_start(False)
    s.Solve(qp["Q"], qp["b"], qp["A_eq"], qp["b_eq"], qp["friction_coeffs"], qp["lb"], qp["ub"])
    r = s.GetSolution()
    ref = oracle.OracleFCCQP(60, 38, 12, 38)
    op = oracle.FCCQPOptions(); op.__dict__.update(P.HARNESS_OPTIONS); ref.set_options(op)
    ref.Solve(**qp); rr = ref.GetSolution()
    assert np.max(np.abs(r.z - rr.z)) <= Z_TOL * (1 + np.max(np.abs(rr.z)))
    assert r.details.n_iter == rr.details.n_iter


def test_config1_batch_1024_replicated(oracle):
    """configs[1]: one W-sat instance copied 1024 times into distinct buffers, fp64, one QP per
    wavefront: every copy gives bit-identical results, equal to the oracle's within tolerance."""
    one = P.workload("W-sat", 1, P.WALKING)
    B = P.Batch(*(np.ascontiguousarray(np.repeat(a, 1024, axis=0)) for a in one.arrays()))
    s = CAPI.Solver(*P.WALKING.ctor_args, batch=1024); s.set_options(**P.HARNESS_OPTIONS)
    s.solve_ok(*B.arrays()); z, d = s.solution()
    assert s.kernel() == CAPI.KERNEL_FUSED
    assert np.all(z == z[0]) and len({x.n_iter for x in d}) == 1
    z_ref, it_ref, _ = oracle_batch(oracle, P.WALKING.ctor_args, P.HARNESS_OPTIONS, one)
    assert_parity(z[:1], [d[0].n_iter], z_ref, it_ref, tol=1e-9, what="replicated")


def test_config2_full_log_as_one_cold_batch_fp32_storage(oracle):
    """configs[2]: a whole (synthetic) walking log streamed as ONE cold-start batch from float32
    arrays.  fp32 is the STORAGE type (halves the HBM / PCIe bytes of the call); the kernels widen on
    load and compute in fp64 -- plain fp32 arithmetic cannot hold the 1e-6 tolerance on K_rho
    (cond ~1e7..1e9) -- so parity is checked at the fp64 tolerance against the
    oracle run on the same float-rounded problem."""
    qps = P.gait_sequence(800, P.WALKING, seed0=11000)      # two gait cycles of the synthetic log
    B = P.stack(qps)
    f32 = [np.ascontiguousarray(a.astype(np.float32)) for a in B.arrays()]
    wide = P.Batch(*(a.astype(np.float64) for a in f32))
    s = CAPI.Solver(*P.WALKING.ctor_args, batch=B.batch); s.set_options(**P.HARNESS_OPTIONS)
    s.solve_f32_ok(*f32); z, d = s.solution()
    assert s.kernel() == CAPI.KERNEL_FUSED
    assert s.algorithmic_bytes() == B.batch * (4 * 6102 + 8 * 60)
    idx = np.arange(0, B.batch, 10)
    sub = P.Batch(*(a[idx] for a in wide.arrays()))
    z_ref, it_ref, _ = oracle_batch(oracle, P.WALKING.ctor_args, P.HARNESS_OPTIONS, sub)
    assert_parity(z[idx], [d[i].n_iter for i in idx], z_ref, it_ref, tol=Z_TOL, what="fp32 log batch")
    it = np.array([x.n_iter for x in d])
    assert it.min() == 0 and it.max() == 100               # the floor-plus-spikes shape of fccqp.pdf Fig. 1
    # how far the float-rounded problem's solution is from the fp64 problem's (information, loose bound)
    s.solve_ok(*B.arrays()); z64, _ = s.solution()
    assert np.max(np.abs(z - z64)) <= 1e-2 * (1 + np.max(np.abs(z64)))


def test_config2_full_log_as_one_cold_batch_fp64(oracle):
    """configs[2] with fp64 storage: a whole (synthetic) log solved as ONE cold-start batch."""
    qps = P.gait_sequence(800, P.WALKING, seed0=11000)      # two gait cycles of the synthetic log
    B = P.stack(qps)
    s = CAPI.Solver(*P.WALKING.ctor_args, batch=B.batch); s.set_options(**P.HARNESS_OPTIONS)
    s.solve_ok(*B.arrays()); z, d = s.solution()
    idx = np.arange(0, B.batch, 10)
    sub = P.Batch(*(a[idx] for a in B.arrays()))
    z_ref, it_ref, _ = oracle_batch(oracle, P.WALKING.ctor_args, P.HARNESS_OPTIONS, sub)
    assert_parity(z[idx], [d[i].n_iter for i in idx], z_ref, it_ref, tol=Z_TOL, what="log batch")
    it = np.array([x.n_iter for x in d])
    assert it.min() == 0 and it.max() == 100               # the floor-plus-spikes shape of fccqp.pdf Fig. 1


def test_config3_large_shape_batch_4096(oracle):
    """configs[3]: 36-DoF / 8-contact shape FCCQP(114, 60, 24, 66), batch 4096 (generic kernel: the
    122 KB packed KKT takes a whole CU's LDS).  Full batch through properties, a sample vs the oracle."""
    nb = 4096
    B = P.workload("W-mix", nb, P.LARGE, distinct=64)
    s = CAPI.Solver(*P.LARGE.ctor_args, batch=nb); s.set_options(**P.PAPER_OPTIONS)
    s.solve_ok(*B.arrays()); z, d = s.solution()
    it = np.array([x.n_iter for x in d])
    assert np.all(np.isfinite(z))
    for r in range(1, nb // 64):
        assert np.array_equal(z[:64], z[64 * r:64 * (r + 1)]) and np.array_equal(it[:64], it[64 * r:64 * (r + 1)])
    res = np.einsum("bij,bj->bi", B.A_eq[:64], z[:64]) - B.b_eq[:64]
    assert np.max(np.abs(res)) <= 1e-7 * (1 + np.max(np.abs(B.b_eq)))
    sl = B.slice(0, 32)
    z_ref, it_ref, _ = oracle_batch(oracle, P.LARGE.ctor_args, P.PAPER_OPTIONS, sl)
    assert_parity(z[:32], it[:32], z_ref, it_ref, tol=Z_TOL, what="large 4096")


def test_config4_one_gpu_shard_of_65536(oracle):
    """configs[4]: batch 65536 sharded over 8 GPUs = 8192 instances per GPU, no collective.  One
    shard (rank 3's seeds) on this GPU: deterministic, permutation-equivariant, sample vs oracle."""
    from fcc_qp_amd import sharding as S
    per = 65536 // 8
    lo, hi = S.shard_bounds(65536, 8, 3)
    assert (lo, hi) == (3 * per, 4 * per)
    B = P.workload("W-mix", per, P.WALKING, seed0=1000 + lo, distinct=128)
    s = CAPI.Solver(*P.WALKING.ctor_args, batch=per); s.set_options(**P.HARNESS_OPTIONS)
    s.solve_ok(*B.arrays()); z, d = s.solution()
    s.solve_ok(*B.arrays()); z2, _ = s.solution()
    assert np.array_equal(z, z2)                                         # run-to-run deterministic
    perm = np.random.default_rng(3).permutation(per)
    s.solve_ok(*(a[perm] for a in B.arrays())); zp, _ = s.solution()
    assert np.array_equal(zp, z[perm])
    z_ref, it_ref, _ = oracle_batch(oracle, P.WALKING.ctor_args, P.HARNESS_OPTIONS, B.slice(0, 128))
    assert_parity(z[:128], [x.n_iter for x in d[:128]], z_ref, it_ref, tol=1e-9, what="shard sample")
