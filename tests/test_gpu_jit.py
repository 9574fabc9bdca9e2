"""Run-time shape specialisation on the GPU: a shape that is NOT compiled into the library gets its specialised kernel
from fccqp_create (hiprtc in process, fcc_qp_amd/csrc/fccqp_jit.cpp), as the reference's constructor takes any sizes at
run time (/root/reference/src/fcc_qp.cpp:24-55).  Checked against the oracle through the C ABI."""
import time

import numpy as np
import pytest

from fcc_qp_amd import problems as P
from tests import capi as CAPI
from tests.test_gpu_parity import _random_qp

pytestmark = pytest.mark.gpu
OPTS = dict(max_iter=60, rho=0.1, eps_fcone=1e-7, eps_bound=1e-7)


class _Opt:
    max_iter, rho, eps_fcone, eps_bound = (OPTS[k] for k in ("max_iter", "rho", "eps_fcone", "eps_bound"))


@pytest.fixture
def jit_on():
    L = CAPI.lib()
    L.fccqp_set_jit(1)
    yield L
    L.fccqp_set_jit(0)


def _batch(dims, k=8):
    rng = np.random.default_rng(sum(dims) + 17)
    return P.stack([_random_qp(rng, *dims) for _ in range(k)])


def _rel(z, z_ref):
    return float(np.max(np.abs(z - z_ref), axis=1).max() / (1.0 + np.max(np.abs(z_ref))))


def test_unlisted_shape_gets_the_one_wave_kernel_and_the_second_create_hits_the_cache(oracle, jit_on):
    L = jit_on
    dims = (44, 30, 6, 30)
    assert L.fccqp_kernel_origin_for_shape(*dims) == CAPI.ORIGIN_JIT
    B = _batch(dims)
    z_ref, it_ref, _, _ = oracle.solve_batch(dims, OPTS, *B.arrays(), nthreads=2)
    t0 = time.perf_counter()
    s = CAPI.Solver(*dims, batch=B.batch)
    t_first = time.perf_counter() - t0
    assert s.kernel() == CAPI.KERNEL_FUSED and s.kernel_origin() == CAPI.ORIGIN_JIT and s.last_error() == ""
    s.set_options(**OPTS)
    s.solve_ok(*B.arrays())
    z, d = s.solution()
    assert s.kernel() == CAPI.KERNEL_FUSED
    assert _rel(z, z_ref) <= 1e-9 and [x.n_iter for x in d] == list(it_ref)
    # a second handle after the in-memory copies are forgotten: the disk cache, no compiler
    L.fccqp_debug_jit_forget()
    t0 = time.perf_counter()
    s2 = CAPI.Solver(*dims, batch=B.batch)
    t_second = time.perf_counter() - t0
    assert s2.kernel_origin() == CAPI.ORIGIN_JIT
    assert t_second < 0.1, f"second create took {t_second:.3f} s (first: {t_first:.2f} s)"
    s2.set_options(**OPTS)
    s2.solve_ok(*B.arrays())
    z2, _ = s2.solution()
    assert np.array_equal(z2, z)
    # warm start on the compiled kernel (the reference loop's second step, fcc_qp_test.py:86-89)
    orcs = [oracle.OracleFCCQP(*dims) for _ in range(B.batch)]
    for i, o in enumerate(orcs):
        o.set_options(_Opt()); o.Solve(**B.instance(i)); o.set_warm_start(True); o.Solve(**B.instance(i))
    s.set_warm_start(True)
    s.solve_ok(*B.arrays())
    zw, dw = s.solution()
    zw_ref = np.stack([o.GetSolution().z for o in orcs])
    assert _rel(zw, zw_ref) <= 1e-9
    # the generic kernel on the same handle agrees (and is what FCCQP_JIT=0 would have used)
    s.set_warm_start(False); s.set_kernel(CAPI.KERNEL_GENERIC)
    s.solve_ok(*B.arrays())
    zg, _ = s.solution()
    assert s.kernel() == CAPI.KERNEL_GENERIC and _rel(zg, z) <= 1e-9
    # the two-launch form is only built ahead of time
    s.set_kernel(CAPI.KERNEL_FUSED_SPLIT)
    assert s.solve(*B.arrays()) == CAPI.ERR_UNSUPPORTED


def test_unlisted_shape_above_64_variables_gets_the_cooperative_kernel(oracle, jit_on):
    dims = (100, 50, 18, 60)
    B = _batch(dims, 6)
    z_ref, it_ref, _, _ = oracle.solve_batch(dims, OPTS, *B.arrays(), nthreads=2)
    s = CAPI.Solver(*dims, batch=B.batch)
    assert s.kernel() == CAPI.KERNEL_COOP and s.kernel_origin() == CAPI.ORIGIN_JIT, s.last_error()
    s.set_options(**OPTS)
    s.solve_ok(*B.arrays())
    z, d = s.solution()
    assert s.kernel() == CAPI.KERNEL_COOP
    assert _rel(z, z_ref) <= 1e-9 and [x.n_iter for x in d] == list(it_ref)


def test_fp32_storage_on_a_run_time_compiled_kernel(jit_on):
    """The float instantiation is compiled on the first fp32-storage solve; results are bit-identical to the fp64
    entry point on the widened values (BASELINE configs[2]'s contract, tests/test_gpu_parity.py)."""
    dims = (47, 30, 9, 30)
    B = _batch(dims)
    f32 = [a.astype(np.float32) for a in B.arrays()]
    s = CAPI.Solver(*dims, batch=B.batch); s.set_options(**OPTS)
    assert s.kernel_origin() == CAPI.ORIGIN_JIT
    s.solve_f32_ok(*f32)
    z32, d32 = s.solution()
    assert s.kernel() == CAPI.KERNEL_FUSED, s.last_error()
    s.solve_ok(*(a.astype(np.float64) for a in f32))
    z64, d64 = s.solution()
    assert np.array_equal(z32, z64) and [x.n_iter for x in d32] == [x.n_iter for x in d64]


def test_shapes_without_a_run_time_kernel_fall_back_to_the_generic_one(oracle, jit_on):
    for dims in ((130, 40, 6, 100), (30, 12, 0, 0)):       # more than 128 variables; no contacts
        s = CAPI.Solver(*dims, batch=2)
        assert s.kernel() == CAPI.KERNEL_GENERIC and s.kernel_origin() == CAPI.ORIGIN_GENERIC_ONLY and s.last_error() == ""


def test_switched_off_means_generic(jit_on):
    L = jit_on
    L.fccqp_set_jit(0)
    s = CAPI.Solver(44, 30, 6, 30, batch=2)
    assert s.kernel() == CAPI.KERNEL_GENERIC and s.kernel_origin() == CAPI.ORIGIN_GENERIC_ONLY


def test_python_class_of_the_reference_takes_any_sizes(oracle, jit_on):
    """FCCQP(num_vars, num_equality_constraints, nc, lambda_c_start) from Python, sizes the library was not built for."""
    import fcc_qp_amd as F
    dims = (47, 30, 9, 30)
    qp = _random_qp(np.random.default_rng(3), *dims)
    s = F.FCCQP(*dims)
    o = F.FCCQPOptions(); o.max_iter, o.rho, o.eps_fcone, o.eps_bound = 60, 0.1, 1e-7, 1e-7
    s.set_options(o)
    s.Solve(qp["Q"], qp["b"], qp["A_eq"], qp["b_eq"], qp["friction_coeffs"], qp["lb"], qp["ub"])
    r = s.GetSolution()
    ref = oracle.OracleFCCQP(*dims); ref.set_options(_Opt()); ref.Solve(**qp)
    sol = ref.GetSolution()
    assert np.max(np.abs(np.asarray(r.z) - sol.z)) <= 1e-9 * (1 + np.max(np.abs(sol.z))) and r.details.n_iter == sol.details.n_iter
