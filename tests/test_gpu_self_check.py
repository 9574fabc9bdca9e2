"""The equality-residual self-check behind a solve (fccqp_set_self_check, fcc_qp_amd/csrc/fccqp_check.hip) and the
timing accessor that says how a time was measured (fccqp_last_timing).  Through the C ABI and through pybind."""
import numpy as np
import pytest

from fcc_qp_amd import problems as P
from tests import capi as CAPI

pytestmark = pytest.mark.gpu

KERNELS = [CAPI.KERNEL_GENERIC, CAPI.KERNEL_AUTO]


def _flags(solver):
    _, d = solver.solution()
    return np.array([x.flags for x in d])


@pytest.mark.parametrize("kernel", KERNELS)
def test_self_check_passes_sound_solves_and_catches_an_unsatisfiable_row(kernel):
    """A_eq z = b_eq holds at every exit (each primal iterate is the head of a KKT solve, fcc_qp.cpp:86-87): the check
    is silent on W-mix, also on an instance with an all-zero row and b_eq = 0 there (dropped pivot, consistent), and
    raises FLAG_EQ_RESIDUAL on exactly the instance whose all-zero row asks for b_eq = 1."""
    base = P.workload("W-mix", 8, P.WALKING)
    qps = [base.instance(i) for i in range(8)]
    for i, rhs in ((2, 0.0), (5, 1.0)):
        q = {k: np.array(v, copy=True) for k, v in qps[i].items()}
        q["A_eq"][7] = 0.0
        q["b_eq"][7] = rhs
        qps[i] = q
    B = P.stack(qps)
    s = CAPI.Solver(*P.WALKING.ctor_args, batch=8); s.set_options(**P.HARNESS_OPTIONS); s.set_kernel(kernel)
    s.solve_ok(*B.arrays())
    assert not np.any(_flags(s) & CAPI.FLAG_EQ_RESIDUAL), "off by default at this size"
    z0, _ = s.solution()
    s.set_self_check(True)
    s.solve_ok(*B.arrays())
    fl = _flags(s)
    assert [bool(f & CAPI.FLAG_EQ_RESIDUAL) for f in fl] == [i == 5 for i in range(8)], fl
    z1, _ = s.solution()
    assert np.array_equal(z0, z1), "the check only reads"
    r = np.einsum("bij,bj->bi", B.A_eq, z1) - B.b_eq
    assert abs(r[5, 7] + 1.0) < 1e-12 and np.max(np.abs(np.delete(r, 5, axis=0))) < 1e-6
    # device-pointer solves and warm starts take the same path
    s.set_warm_start(True)
    s.solve_ok(*B.arrays())
    assert [bool(f & CAPI.FLAG_EQ_RESIDUAL) for f in _flags(s)] == [i == 5 for i in range(8)]


def test_self_check_catches_a_non_finite_result():
    base = P.workload("W-sat", 4, P.WALKING)
    qps = [base.instance(i) for i in range(4)]
    q = {k: np.array(v, copy=True) for k, v in qps[1].items()}
    q["b"][3] = np.nan
    qps[1] = q
    B = P.stack(qps)
    s = CAPI.Solver(*P.WALKING.ctor_args, batch=4); s.set_options(**P.HARNESS_OPTIONS); s.set_self_check(True)
    s.solve_ok(*B.arrays())
    z, _ = s.solution()
    assert not np.all(np.isfinite(z[1])) and np.all(np.isfinite(np.delete(z, 1, axis=0)))
    assert [bool(f & CAPI.FLAG_EQ_RESIDUAL) for f in _flags(s)] == [False, True, False, False]


@pytest.mark.parametrize("dims", [(180, 80, 12, 100), (300, 212, 9, 100)])      # N = 260, 512: the eight-slot kernel
def test_self_check_is_on_by_default_beyond_256_rows_and_silent(dims):
    from tests.test_gpu_parity import _random_qp
    n, m, nc, s0 = dims
    rng = np.random.default_rng(sum(dims) + 3)
    B = P.stack([_random_qp(rng, n, m, nc, s0) for _ in range(3)])
    s = CAPI.Solver(*dims, batch=3); s.set_options(max_iter=25, rho=0.1, eps_fcone=1e-7, eps_bound=1e-7)
    s.solve_ok(*B.arrays())
    z, _ = s.solution()
    assert not np.any(_flags(s) & CAPI.FLAG_EQ_RESIDUAL)
    r = np.einsum("bij,bj->bi", B.A_eq, z) - B.b_eq
    assert np.max(np.abs(r)) < 1e-8 * (1 + np.max(np.abs(B.b_eq)))
    # an unsatisfiable row is reported without asking
    q = {k: np.array(v, copy=True) for k, v in B.instance(1).items()}
    q["A_eq"][m - 1] = 0.0
    q["b_eq"][m - 1] = 2.0
    B2 = P.stack([B.instance(0), q, B.instance(2)])
    s.solve_ok(*B2.arrays())
    assert [bool(f & CAPI.FLAG_EQ_RESIDUAL) for f in _flags(s)] == [False, True, False]


def test_last_timing_says_how_it_was_measured():
    """Small synchronous host-pointer solves record no events (host clock); device-pointer / large solves are timed by
    a HIP event pair; with kernel timing switched off nothing is measured -- never two kinds under one name."""
    import fcc_qp_amd as F
    one = CAPI.Solver(*P.WALKING.ctor_args, batch=1); one.set_options(**P.HARNESS_OPTIONS)
    B1 = P.workload("W-sat", 1, P.WALKING)
    one.solve_ok(*B1.arrays())
    ms, by = one.last_timing()
    assert by == CAPI.TIMING_HOST_CLOCK and 0.0 < ms < 50.0
    assert one.kernel_ms() == pytest.approx(ms)
    big = CAPI.Solver(*P.WALKING.ctor_args, batch=512); big.set_options(**P.HARNESS_OPTIONS)
    Bb = P.workload("W-sat", 512, P.WALKING, distinct=32)
    big.solve_ok(*Bb.arrays())
    ms, by = big.last_timing()
    assert by == CAPI.TIMING_DEVICE_EVENTS and 0.0 < ms < 50.0
    # through pybind: names instead of codes
    fb = F.FCCQPBatch(*P.WALKING.ctor_args, 512)
    fb.Solve(*Bb.arrays())
    ms, how = fb.LastTiming()
    assert how == "device_events" and ms > 0.0
    import torch
    dev = torch.device("cuda", 0)
    tens = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in Bb.arrays()]
    fb.set_kernel_timing(False)
    F.solve_tensors(fb, *tens)
    torch.cuda.synchronize()
    ms, how = fb.LastTiming()
    assert how == "none" and ms == 0.0
    small = F.FCCQPBatch(*P.WALKING.ctor_args, 2)
    small.Solve(*P.workload("W-sat", 2, P.WALKING).arrays())
    assert small.LastTiming()[1] == "host_clock"
    assert F.FLAG_EQ_RESIDUAL == CAPI.FLAG_EQ_RESIDUAL == 32


def test_presolve_path_is_reported():
    """One-wave kernel, cold start: the presolve's system (I - rho G) x0 = x_c is solved by conjugate gradients
    (FLAG_PRESOLVE_CG) -- on W-sat, W-mix and the semi-definite W-psd alike --, EXCEPT where CG declines and the direct
    factorisation of W takes over with its own rules: a variable without cost or constraint (singular diagonal -> the
    COD pass, as the reference's LDLT failure), and a rho so large that W is too ill-conditioned to converge in
    n - m + 10 steps.  Warm starts skip the presolve; the generic kernel has no CG path.  z agrees between the two
    presolves to rounding."""
    from tests.test_gpu_parity import _dead_variable
    shape = P.WALKING
    for wl in ("W-sat", "W-mix", "W-psd"):
        B = P.workload(wl, 16, shape)
        s = CAPI.Solver(*shape.ctor_args, batch=16); s.set_options(**P.HARNESS_OPTIONS)
        s.solve_ok(*B.arrays())
        fl = _flags(s)
        assert np.all(fl & CAPI.FLAG_PRESOLVE_RAN) and np.all(fl & CAPI.FLAG_PRESOLVE_CG), (wl, fl)
        z_cg, _ = s.solution()
        s.set_kernel(CAPI.KERNEL_GENERIC)
        s.solve_ok(*B.arrays())
        assert not np.any(_flags(s) & CAPI.FLAG_PRESOLVE_CG)
        z_dir, _ = s.solution()
        assert np.max(np.abs(z_cg - z_dir)) <= 1e-9 * (1 + np.max(np.abs(z_dir))), wl
        s.set_kernel(CAPI.KERNEL_AUTO)
        s.set_warm_start(True)
        s.solve_ok(*B.arrays())
        assert not np.any(_flags(s) & (CAPI.FLAG_PRESOLVE_RAN | CAPI.FLAG_PRESOLVE_CG))
    # CG declines: dead variables (instances 1 and 4), the rest of the batch is unaffected
    base = P.workload("W-sat", 6, shape)
    qps = [base.instance(i) for i in range(6)]
    qps[1] = _dead_variable(qps[1], 25); qps[4] = _dead_variable(qps[4], 31)
    s = CAPI.Solver(*shape.ctor_args, batch=6); s.set_options(**P.HARNESS_OPTIONS)
    s.solve_ok(*P.stack(qps).arrays())
    fl = _flags(s)
    assert [bool(f & CAPI.FLAG_PRESOLVE_CG) for f in fl] == [True, False, True, True, False, True]
    assert [bool(f & CAPI.FLAG_PRESOLVE_COD) for f in fl] == [False, True, False, False, True, False]
    # CG does not converge: rho = 100 (cond(W) ~ 1e8) -> the factorisation of W, no flag, same accuracy as before
    s2 = CAPI.Solver(*shape.ctor_args, batch=6); s2.set_options(max_iter=0, rho=100.0, eps_fcone=1e-6, eps_bound=1e-6)
    s2.solve_ok(*base.arrays())
    fl = _flags(s2)
    assert np.all(fl & CAPI.FLAG_PRESOLVE_RAN) and not np.any(fl & CAPI.FLAG_PRESOLVE_CG), fl
