"""The two-launch form of the cooperative kernel (FCCQP_KERNEL_COOP_SPLIT: coop_kernel<.., SPLIT> up to the presolve at
one QP per compute unit, then coop_admm_kernel -- ADMM iterations, metrics, write-back -- at two or three QPs per compute
unit, fccqp_coop_impl.h) against the one-launch cooperative kernel and the oracle, through the C ABI.

The second launch runs the same instruction sequence on the same data as the tail of the one-launch kernel, so every
output must be BIT-IDENTICAL: z, duals, n_iter, flags, the four metrics.  Covered: cold solves, warm-start chains (the
state left by one form continued by the other), option sets that stop at intermediate iterations, max_iter = 0, fp32
storage, instances that leave for the COD pass between the two launches, a batch of several rounds, what AUTO picks
by batch size, and shapes that have no two-launch form.
Reference behaviour: FCCQP::Solve / DoADMM, /root/reference/src/fcc_qp.cpp:57-191."""
import numpy as np
import pytest

from fcc_qp_amd import problems as P
from tests import capi as CAPI
from tests.helpers import Z_TOL_TIGHT, assert_parity, oracle_batch
from tests.test_gpu_coop_kernel import gpu, random_batch

pytestmark = pytest.mark.gpu

LARGE = P.LARGE.ctor_args           # (114, 60, 24, 66): BASELINE configs[3]
FIELDS = ("admm_residual_bounds", "admm_residual_friction_cone", "bounds_viol", "friction_cone_viol", "n_iter", "solve_status", "flags")


def assert_same_bits(one, two, what):
    for step, ((z1, it1, d1, st1), (z2, it2, d2, st2)) in enumerate(zip(one, two)):
        assert np.array_equal(z1, z2), f"{what}: z differs at step {step} (max {np.max(np.abs(z1 - z2)):.3e})"
        assert np.array_equal(it1, it2), f"{what}: n_iter differs at step {step}"
        for a, b in zip(st1, st2):
            assert np.array_equal(a, b), f"{what}: warm-start state differs at step {step}"
        for i, (a, b) in enumerate(zip(d1, d2)):
            for f in FIELDS:
                va, vb = getattr(a, f), getattr(b, f)
                assert va == vb or (va != va and vb != vb), (what, step, i, f, va, vb)


@pytest.mark.parametrize("workload", ["W-sat", "W-mix", "W-psd"])
def test_two_launches_equal_one_launch_and_the_oracle(oracle, workload):
    B = P.workload(workload, 40, P.LARGE)
    one = gpu(LARGE, P.HARNESS_OPTIONS, B, CAPI.KERNEL_COOP, warm_steps=2)
    two = gpu(LARGE, P.HARNESS_OPTIONS, B, CAPI.KERNEL_COOP_SPLIT, warm_steps=2)
    assert_same_bits(one, two, workload)
    z_ref, it_ref, _ = oracle_batch(oracle, LARGE, P.HARNESS_OPTIONS, B)
    assert_parity(two[0][0], two[0][1], z_ref, it_ref, tol=Z_TOL_TIGHT, what=f"coop_split {workload}")


@pytest.mark.parametrize("options", [dict(max_iter=80, rho=1e-2, eps_fcone=1e-7, eps_bound=1e-7),
                                     dict(max_iter=0, rho=5e-5, eps_fcone=1e-6, eps_bound=1e-6),
                                     dict(max_iter=1, rho=1e-3, eps_fcone=1e-3, eps_bound=1e-3),
                                     dict(max_iter=300, rho=0.1, eps_fcone=1e-3, eps_bound=1e-3)])
@pytest.mark.parametrize("dims", [LARGE, (70, 20, 9, 40)], ids=["n114_eight_waves", "n70_four_waves"])
def test_option_sets_random_population(oracle, options, dims):
    """Both listed shapes with more than 64 variables: FCCQP(114, 60, 24, 66) (eight waves in the first launch, 29 / 29 /
    28 / 28 variables per ADMM wave) and FCCQP(70, 20, 9, 40) (four waves, ragged second row half, three contacts)."""
    B = random_batch(dims, 24, seed=17 + options["max_iter"])
    one = gpu(dims, options, B, CAPI.KERNEL_COOP, warm_steps=1)
    two = gpu(dims, options, B, CAPI.KERNEL_COOP_SPLIT, warm_steps=1)
    assert_same_bits(one, two, str(options))
    assert all(x.flags & CAPI.FLAG_PRESOLVE_CG for x in two[0][2])
    z_ref, it_ref, _ = oracle_batch(oracle, dims, options, B)
    assert_parity(two[0][0], two[0][1], z_ref, it_ref, tol=Z_TOL_TIGHT, what=f"coop_split {dims} {options}")


@pytest.mark.parametrize("rho", [100.0, 1e4])
def test_declined_conjugate_gradients_go_through_the_third_launch(rho):
    """cond(W) ~ rho / lambda_min of the reduced Hessian: at rho = 100 and beyond conjugate gradients do not converge in
    n - m + 10 steps on the OSC workloads and decline in the second launch; the third launch (the whole-solve kernel behind
    its gate) takes the direct factorisation of W for exactly those instances.  (No oracle here: this is beyond the
    Woodbury presolve's own accuracy envelope, tests/test_gpu_parity.py; what is checked is that the two forms agree.)"""
    options = dict(max_iter=20, rho=rho, eps_fcone=1e-6, eps_bound=1e-6)
    B = P.workload("W-sat", 36, P.LARGE)
    one = gpu(LARGE, options, B, CAPI.KERNEL_COOP, warm_steps=1)
    two = gpu(LARGE, options, B, CAPI.KERNEL_COOP_SPLIT, warm_steps=1)
    assert_same_bits(one, two, f"rho={rho}")
    cg = [bool(x.flags & CAPI.FLAG_PRESOLVE_CG) for x in two[0][2]]
    assert not all(cg), "expected conjugate gradients to decline on some instance"
    assert all(x.flags & CAPI.FLAG_PRESOLVE_RAN for x in two[0][2])
    assert np.all(np.isfinite(two[0][0]))


def test_fp32_storage_inputs():
    B = random_batch(LARGE, 16, seed=5)
    B32 = P.Batch(*[np.asarray(a, dtype=np.float32).astype(np.float64) for a in B.arrays()])
    one = gpu(LARGE, P.HARNESS_OPTIONS, B32, CAPI.KERNEL_COOP, f32=True)
    two = gpu(LARGE, P.HARNESS_OPTIONS, B32, CAPI.KERNEL_COOP_SPLIT, f32=True)
    ref = gpu(LARGE, P.HARNESS_OPTIONS, B32, CAPI.KERNEL_COOP_SPLIT)
    assert_same_bits(one, two, "f32 storage")
    assert_same_bits(ref, two, "f32 storage vs f64 on the widened values")


def test_the_state_of_one_form_is_continued_by_the_other():
    B = P.workload("W-mix", 20, P.LARGE)
    s = CAPI.Solver(*LARGE, batch=B.batch); s.set_options(**P.HARNESS_OPTIONS)
    t = CAPI.Solver(*LARGE, batch=B.batch); t.set_options(**P.HARNESS_OPTIONS); t.set_kernel(CAPI.KERNEL_COOP)
    for step, k in enumerate((CAPI.KERNEL_COOP_SPLIT, CAPI.KERNEL_COOP, CAPI.KERNEL_COOP_SPLIT, CAPI.KERNEL_COOP_SPLIT)):
        s.set_kernel(k); s.set_warm_start(step > 0); t.set_warm_start(step > 0)
        s.solve_ok(*B.arrays()); t.solve_ok(*B.arrays())
        assert s.kernel() == k
        (z1, d1), (z2, d2) = s.solution(), t.solution()
        assert np.array_equal(z1, z2), step
        assert [x.n_iter for x in d1] == [x.n_iter for x in d2]


@pytest.mark.parametrize("fallback", [True, False])
def test_instances_that_leave_for_the_cod_pass_between_the_launches(oracle, fallback):
    """A variable without cost or constraint: W = I - rho G has a vanishing pivot, conjugate gradients decline, the direct
    factorisation flags the instance and (fallback on) hands it to the COD pass -- the second launch must skip it."""
    B = random_batch(LARGE, 12, seed=3)
    Q, b, A, beq, mu, lb, ub = [np.array(a) for a in B.arrays()]
    dead = 5
    for i in (2, 7):
        Q[i, dead, :] = 0.0; Q[i, :, dead] = 0.0; A[i, :, dead] = 0.0; b[i, dead] = 0.0
    Bd = P.Batch(Q, b, A, beq, mu, lb, ub)

    def run(kernel):
        s = CAPI.Solver(*LARGE, batch=Bd.batch); s.set_options(**P.HARNESS_OPTIONS); s.set_kernel(kernel)
        s.set_presolve_fallback(fallback)
        s.solve_ok(*Bd.arrays())
        z, d = s.solution()
        return [(z, np.array([x.n_iter for x in d]), d, s.state())]
    one, two = run(CAPI.KERNEL_COOP), run(CAPI.KERNEL_COOP_SPLIT)
    assert_same_bits(one, two, f"dead variable, fallback={fallback}")
    flags = [x.flags for x in two[0][2]]
    for i in range(Bd.batch):
        assert bool(flags[i] & CAPI.FLAG_PRESOLVE_NUMERICAL) == (i in (2, 7)), (i, flags[i])
        if fallback:
            assert bool(flags[i] & CAPI.FLAG_PRESOLVE_COD) == (i in (2, 7))
    if fallback:
        z_ref, it_ref, _ = oracle_batch(oracle, LARGE, P.HARNESS_OPTIONS, Bd)
        assert_parity(two[0][0], two[0][1], z_ref, it_ref, what="coop_split with COD instances")


def test_several_rounds_and_what_auto_picks():
    import torch
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    nb = 2 * cus + 37                                   # AUTO's threshold: two rounds of the one-launch kernel
    B = P.workload("W-mix", nb, P.LARGE, distinct=24)
    one = gpu(LARGE, P.HARNESS_OPTIONS, B, CAPI.KERNEL_COOP)
    s = CAPI.Solver(*LARGE, batch=nb); s.set_options(**P.HARNESS_OPTIONS)
    s.solve_ok(*B.arrays())
    assert s.kernel() == CAPI.KERNEL_COOP_SPLIT, "AUTO: two launches from two rounds on"
    z, d = s.solution()
    assert np.array_equal(z, one[0][0]) and [x.n_iter for x in d] == one[0][1].tolist()
    # the tiled copies of an instance agree bit for bit
    for i in range(24, nb):
        assert np.array_equal(z[i], z[i % 24])
    small = CAPI.Solver(*LARGE, batch=16); small.set_options(**P.HARNESS_OPTIONS)
    Bs = P.workload("W-mix", 16, P.LARGE)
    small.solve_ok(*Bs.arrays())
    assert small.kernel() == CAPI.KERNEL_COOP, "AUTO: one launch for a small batch"


def test_shapes_without_a_two_launch_form_refuse_it():
    s = CAPI.Solver(*P.WALKING.ctor_args, batch=4); s.set_kernel(CAPI.KERNEL_COOP_SPLIT)
    B = P.workload("W-mix", 4, P.WALKING)
    rc = s.solve(*B.arrays())
    assert rc == CAPI.ERR_UNSUPPORTED and "two-launch" in s.last_error()
