"""The two-launch form of the fused kernel (FCCQP_KERNEL_FUSED_SPLIT: fused_kernel<.., SPLIT> up to the
presolve, then admm_kernel at several waves per SIMD) against the one-launch form and the oracle.

Both forms run the same instruction sequence per instance on the same values (rho G, x_c and x0 take
a round trip through device memory as doubles), so they must agree BIT FOR BIT -- z, n_iter, the
residual norms, the violation metrics, the warm-start state -- on every population, including the
instances that leave for the COD pass between the two launches.
Reference behaviour: FCCQP::Solve / DoADMM, /root/reference/src/fcc_qp.cpp:57-191."""
import numpy as np
import pytest

from fcc_qp_amd import problems as P
from tests import capi as CAPI
from tests.helpers import Z_TOL, Z_TOL_TIGHT, assert_parity, oracle_batch

pytestmark = pytest.mark.gpu


def run(shape, options, batch, kernel, warm_steps=0, fallback=True, f32=False):
    s = CAPI.Solver(*shape.ctor_args, batch=batch.batch)
    s.set_options(**options); s.set_kernel(kernel); s.set_presolve_fallback(fallback)
    outs = []
    for t in range(1 + warm_steps):
        s.set_warm_start(t > 0)
        (s.solve_f32_ok if f32 else s.solve_ok)(*batch.arrays())
        z, d = s.solution()
        outs.append((z, np.array([x.n_iter for x in d]), np.array([x.flags for x in d]),
                     np.array([[x.admm_residual_bounds, x.admm_residual_friction_cone, x.bounds_viol, x.friction_cone_viol] for x in d]),
                     s.state()))
    assert s.kernel() == kernel
    return outs


def assert_same(a, b, what):
    for t, (ra, rb) in enumerate(zip(a, b)):
        assert np.array_equal(ra[0], rb[0]), f"{what} step {t}: z differs (max {np.max(np.abs(ra[0] - rb[0])):.3e})"
        assert np.array_equal(ra[1], rb[1]), f"{what} step {t}: n_iter differs"
        assert np.array_equal(ra[2], rb[2]), f"{what} step {t}: flags differ"
        assert np.array_equal(ra[3], rb[3]), f"{what} step {t}: details differ"
        for u, v in zip(ra[4], rb[4]):
            assert np.array_equal(u, v), f"{what} step {t}: warm-start state differs"


@pytest.mark.parametrize("wl", ["W-sat", "W-mix", "W-psd"])
def test_split_equals_one_launch_and_oracle(oracle, wl):
    B = P.workload(wl, 200, P.WALKING, distinct=100)          # 200: not a multiple of 64 (ragged swizzle tail)
    one = run(P.WALKING, P.HARNESS_OPTIONS, B, CAPI.KERNEL_FUSED, warm_steps=2)
    two = run(P.WALKING, P.HARNESS_OPTIONS, B, CAPI.KERNEL_FUSED_SPLIT, warm_steps=2)
    assert_same(one, two, wl)
    z_ref, it_ref, _ = oracle_batch(oracle, P.WALKING.ctor_args, P.HARNESS_OPTIONS, B.slice(0, 100))
    assert_parity(two[0][0][:100], two[0][1][:100], z_ref, it_ref, tol=Z_TOL_TIGHT if wl != "W-psd" else Z_TOL, what=wl)


@pytest.mark.parametrize("options", [P.PAPER_OPTIONS, dict(max_iter=0, rho=5e-5, eps_fcone=1e-6, eps_bound=1e-6),
                                     dict(max_iter=1, rho=5e-5, eps_fcone=1e-6, eps_bound=1e-6),
                                     dict(max_iter=1000, rho=1e-2, eps_fcone=1e-6, eps_bound=1e-6)])
def test_split_option_sets(oracle, options):
    B = P.workload("W-mix", 48, P.WALKING)
    one = run(P.WALKING, options, B, CAPI.KERNEL_FUSED, warm_steps=1)
    two = run(P.WALKING, options, B, CAPI.KERNEL_FUSED_SPLIT, warm_steps=1)
    assert_same(one, two, str(options))
    z_ref, it_ref, _ = oracle_batch(oracle, P.WALKING.ctor_args, options, B)
    assert_parity(two[0][0], two[0][1], z_ref, it_ref, tol=Z_TOL, what=str(options))


@pytest.mark.parametrize("fallback", [True, False])
def test_split_with_instances_that_leave_for_the_cod_pass(oracle, fallback):
    base = P.workload("W-mix", 12, P.WALKING)
    qps = [base.instance(i) for i in range(12)]
    for i, j in {2: 25, 7: 30, 8: 23}.items():             # a dead variable: K exactly singular
        q = {k: np.array(v, copy=True) for k, v in qps[i].items()}
        q["Q"][j, :] = 0.0; q["Q"][:, j] = 0.0; q["A_eq"][:, j] = 0.0; q["b"][j] = 0.0
        qps[i] = q
    B = P.stack(qps)
    one = run(P.WALKING, P.HARNESS_OPTIONS, B, CAPI.KERNEL_FUSED, fallback=fallback)
    two = run(P.WALKING, P.HARNESS_OPTIONS, B, CAPI.KERNEL_FUSED_SPLIT, fallback=fallback)
    ok = [i for i in range(12) if i not in (2, 7, 8)]
    assert np.array_equal(one[0][0][ok], two[0][0][ok]) and np.array_equal(one[0][1][ok], two[0][1][ok])
    assert np.array_equal(one[0][2], two[0][2])
    if fallback:
        assert_same(one, two, "cod")
        z_ref, it_ref, _ = oracle_batch(oracle, P.WALKING.ctor_args, P.HARNESS_OPTIONS, B)
        assert_parity(two[0][0], two[0][1], z_ref, it_ref, tol=Z_TOL, what="split + cod")
        assert all(bool(f & CAPI.FLAG_PRESOLVE_COD) == (i in (2, 7, 8)) for i, f in enumerate(two[0][2]))


@pytest.mark.parametrize("dims", [(12, 5, 6, 3), (12, 8, 6, 3), (32, 14, 6, 20), (32, 18, 6, 20), (12, 7, 6, 3), (24, 6, 15, 4)])
def test_split_on_the_other_compiled_shapes(oracle, dims):
    rng = np.random.default_rng(sum(dims))
    n, m, nc, s = dims
    nb = 40
    Lq = rng.standard_normal((nb, n, n))
    Q = Lq @ np.transpose(Lq, (0, 2, 1)) + 0.5 * np.eye(n)
    A = rng.standard_normal((nb, m, n)); b = rng.standard_normal((nb, n)); beq = 0.3 * rng.standard_normal((nb, m))
    lb = np.full((nb, n), -np.inf); ub = np.full((nb, n), np.inf)
    lb[:, :2] = -0.2; ub[:, :2] = 0.3
    B = P.Batch(Q, b, A, beq, np.full((nb, nc // 3), 0.5), lb, ub)
    opts = dict(max_iter=60, rho=1e-2, eps_fcone=1e-7, eps_bound=1e-7)

    class _S:                      # the minimal shape object run() needs
        ctor_args = dims
    one = run(_S, opts, B, CAPI.KERNEL_FUSED, warm_steps=1)
    two = run(_S, opts, B, CAPI.KERNEL_FUSED_SPLIT, warm_steps=1)
    assert_same(one, two, str(dims))
    z_ref, it_ref, _ = oracle_batch(oracle, dims, opts, B)
    assert_parity(two[0][0], two[0][1], z_ref, it_ref, tol=Z_TOL, what=str(dims))


def test_split_fp32_storage():
    B = P.workload("W-mix", 64, P.WALKING)
    one = run(P.WALKING, P.HARNESS_OPTIONS, B, CAPI.KERNEL_FUSED, f32=True)
    two = run(P.WALKING, P.HARNESS_OPTIONS, B, CAPI.KERNEL_FUSED_SPLIT, f32=True)
    assert_same(one, two, "f32 storage")


def test_split_refused_without_a_fused_shape():
    s = CAPI.Solver(10, 4, 3, 2, batch=2)
    s.set_kernel(CAPI.KERNEL_FUSED_SPLIT)
    B = P.Batch(np.stack([np.eye(10)] * 2), np.zeros((2, 10)), np.ones((2, 4, 10)), np.zeros((2, 4)), np.full((2, 1), 0.5),
                np.full((2, 10), -np.inf), np.full((2, 10), np.inf))
    assert s.solve(*B.arrays()) == CAPI.ERR_UNSUPPORTED


# ---- the block -> instance order of a chain's next solve (fccqp_set_dispatch_order) --------------------------------------
def test_longest_first_order_changes_no_result():
    """FCCQP_ORDER_PREVIOUS_ITERATIONS: block b of a warm-started solve takes the instance with the b-th largest iteration
    count of the previous solve (a counting sort on the device in front of the launch).  Whatever the order, every
    instance is solved exactly once and by the same instruction sequence: z, n_iter, duals and flags must be bit-identical
    to the fixed permutation's, step after step of a chain whose inputs move."""
    import torch
    nb = 8 * torch.cuda.get_device_properties(0).multi_processor_count + 40        # (the order applies from 8 instances per CU on)
    seq = P.gait_sequence(260)
    offs = [(7 * c) % 250 for c in range(64)]

    def batch_at(t):
        base = P.stack([seq[o + t] for o in offs])
        reps = -(-nb // 64)
        return P.Batch(*(np.ascontiguousarray(np.tile(a, (reps,) + (1,) * (a.ndim - 1))[:nb]) for a in base.arrays()))
    batches = [batch_at(t) for t in range(4)]
    out = {}
    for order in (CAPI.ORDER_HASH, CAPI.ORDER_PREVIOUS_ITERATIONS):
        s = CAPI.Solver(*P.WALKING.ctor_args, batch=nb); s.set_options(**P.HARNESS_OPTIONS); s.set_dispatch_order(order)
        steps = []
        for t, B in enumerate(batches):
            s.set_warm_start(t > 0)
            s.solve_ok(*B.arrays())
            z, d = s.solution()
            steps.append((z, [x.n_iter for x in d], [x.flags for x in d], s.state()))
        assert s.kernel() == CAPI.KERNEL_FUSED
        out[order] = steps
        s.close()
    for t, (a, b) in enumerate(zip(out[CAPI.ORDER_HASH], out[CAPI.ORDER_PREVIOUS_ITERATIONS])):
        assert np.array_equal(a[0], b[0]) and a[1] == b[1] and a[2] == b[2], f"step {t}"
        for x, y in zip(a[3], b[3]):
            assert np.array_equal(x, y), f"state at step {t}"
    its = np.array(out[CAPI.ORDER_HASH][-1][1])
    assert its.min() < 50 and its.max() == P.HARNESS_OPTIONS["max_iter"], "the chain mixes short and long instances"
    s = CAPI.Solver(*P.WALKING.ctor_args, batch=4)
    assert s.L.fccqp_set_dispatch_order(s.h, 7) == CAPI.ERR_INVALID_ARG
