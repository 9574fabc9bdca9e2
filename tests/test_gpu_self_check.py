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
