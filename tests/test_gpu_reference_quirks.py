"""GPU parity tests for the reference behaviours SURVEY.md 8(a) lists as "must reproduce" that the
first round only checked on the oracle: the two ADMM residual norms of GetSolution
(/root/reference/src/fcc_qp.cpp:97-98,196-197), finite bounds on the contact variables (the
lambda_c-segment overwrite, fcc_qp.cpp:81-82,95-109), shapes in the global-workspace range of the
generic kernel (198 < N <= 256), rank-deficient A_eq (fcc_qp.hpp:104-108 N.B.), semi-definite costs
(the reference's COD branch, fcc_qp.cpp:164-166).  All through the C ABI, against the oracle.
Run on the MI355X box:  pytest tests -m gpu"""
import numpy as np
import pytest

from fcc_qp_amd import problems as P
from tests import capi as CAPI
from tests.helpers import Z_TOL, Z_TOL_TIGHT, assert_parity, oracle_batch, rel_err

pytestmark = pytest.mark.gpu

KERNELS = [CAPI.KERNEL_GENERIC, CAPI.KERNEL_AUTO]
RES_TOL = 1e-6          # relative, on admm_residual_bounds / admm_residual_friction_cone


def gpu_solve(dims, options, batch, kernel=CAPI.KERNEL_AUTO, warm=False, solver=None):
    s = solver or CAPI.Solver(*dims, batch=batch.batch)
    s.set_options(**options)
    s.set_kernel(kernel)
    s.set_warm_start(warm)
    s.solve_ok(*batch.arrays())
    z, d = s.solution()
    return z, np.array([x.n_iter for x in d]), d, s


def assert_residual_fields(d, det_ref, it, it_ref, what):
    """eps_bounds / eps_friction_cone are the inf-norms of the LAST executed iteration's residuals
    (fcc_qp.cpp:97-98); comparable whenever both sides executed the same number of iterations."""
    for i in range(len(d)):
        if it[i] != it_ref[i]:
            continue
        for mine, ref, name in ((d[i].admm_residual_bounds, det_ref[i].admm_residual_bounds, "bounds"),
                                (d[i].admm_residual_friction_cone, det_ref[i].admm_residual_friction_cone, "cone")):
            assert abs(mine - ref) <= RES_TOL * (1 + abs(ref)) + 1e-12, f"{what}: instance {i} residual_{name} {mine} vs {ref}"


# ---- (a) the residual-norm fields ------------------------------------------------------------------
@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("wl", ["W-sat", "W-easy", "W-mix"])
def test_admm_residual_norms_match_oracle(oracle, wl, kernel):
    B = P.workload(wl, 64, P.WALKING)
    for opts in (P.HARNESS_OPTIONS, P.PAPER_OPTIONS, dict(max_iter=400, rho=1e-2, eps_fcone=1e-7, eps_bound=1e-7)):
        z_ref, it_ref, det_ref = oracle_batch(oracle, P.WALKING.ctor_args, opts, B)
        z, it, d, _ = gpu_solve(P.WALKING.ctor_args, opts, B, kernel)
        assert_parity(z, it, z_ref, it_ref, tol=Z_TOL, what=f"{wl}/k{kernel}")
        assert_residual_fields(d, det_ref, it, it_ref, f"{wl}/k{kernel}/{opts['max_iter']}")
        assert np.mean(it == it_ref) > 0.9          # the comparison above is not vacuous


def test_admm_residual_norms_large_shape(oracle):
    B = P.workload("W-mix", 16, P.LARGE)
    z_ref, it_ref, det_ref = oracle_batch(oracle, P.LARGE.ctor_args, P.HARNESS_OPTIONS, B)
    z, it, d, _ = gpu_solve(P.LARGE.ctor_args, P.HARNESS_OPTIONS, B)
    assert_parity(z, it, z_ref, it_ref, tol=Z_TOL, what="large")
    assert_residual_fields(d, det_ref, it, it_ref, "large")


# ---- (b) finite bounds on the contact variables ------------------------------------------------------
def _bound_contact_vars(qp, s, nc, rng, tight):
    """lb/ub on lambda_c: the reference clamps x_bar over them (fcc_qp.cpp:90) and their duals enter
    x_res and the exit test (:95-105), but q_rho's lambda_c segment is overwritten by the cone copy
    (:82), so those box duals never reach the primal update."""
    q = {k: np.array(v, copy=True) for k, v in qp.items()}
    width = rng.uniform(5.0, 40.0, nc) if tight else rng.uniform(200.0, 400.0, nc)
    q["lb"][s:s + nc] = -width
    q["ub"][s:s + nc] = width
    q["lb"][s + 2:s + nc:3] = 0.0          # normal forces: lambda_z >= 0
    return q


@pytest.mark.parametrize("kernel", KERNELS)
def test_finite_bounds_on_contact_variables_walking(oracle, kernel):
    rng = np.random.default_rng(11)
    base = P.workload("W-mix", 32, P.WALKING)
    s, nc = P.WALKING.lambda_c_start, P.WALKING.nc
    B = P.stack([_bound_contact_vars(base.instance(i), s, nc, rng, tight=(i % 2 == 0)) for i in range(32)])
    for opts in (P.HARNESS_OPTIONS, dict(max_iter=300, rho=1e-2, eps_fcone=1e-6, eps_bound=1e-6)):
        z_ref, it_ref, det_ref = oracle_batch(oracle, P.WALKING.ctor_args, opts, B)
        z, it, d, _ = gpu_solve(P.WALKING.ctor_args, opts, B, kernel)
        assert_parity(z, it, z_ref, it_ref, tol=Z_TOL_TIGHT, what=f"bounded lambda_c k{kernel}")
        assert_residual_fields(d, det_ref, it, it_ref, f"bounded lambda_c k{kernel}")
        for i in range(B.batch):
            assert abs(d[i].bounds_viol - det_ref[i].bounds_viol) <= 1e-6 * (1 + abs(det_ref[i].bounds_viol))
    # the quirk is live in this population: some tight boxes are violated by z (nothing pulls the
    # primal lambda_c into its box), i.e. the test would notice a kernel that "fixed" the overwrite
    assert max(det_ref[i].bounds_viol for i in range(B.batch)) > 1e-3


def test_finite_bounds_on_contact_variables_odd_shape(oracle):
    from tests.test_gpu_parity import _random_qp
    dims = (33, 9, 9, 24)
    n, m, nc, s = dims
    rng = np.random.default_rng(12)
    qps = []
    for _ in range(8):
        q = _random_qp(rng, *dims)
        q["lb"][s:s + nc] = -rng.uniform(0.01, 0.3, nc); q["ub"][s:s + nc] = rng.uniform(0.01, 0.3, nc)
        qps.append(q)
    B = P.stack(qps)
    opts = dict(max_iter=80, rho=0.1, eps_fcone=1e-7, eps_bound=1e-7)
    z_ref, it_ref, det_ref, _ = oracle.solve_batch(dims, opts, *B.arrays(), nthreads=2)
    z, it, d, _ = gpu_solve(dims, opts, B)
    assert_parity(z, it, z_ref, it_ref, tol=Z_TOL, what="bounded lambda_c odd shape")
    assert_residual_fields(d, det_ref, it, it_ref, "bounded lambda_c odd shape")


# ---- (c) shapes whose KKT does not fit in LDS: per-block global workspace (198 < N <= 512) ------------
# (Round 2 withdrew N > 256 after sporadically wrong iterates; round 3 found the cause -- an agent-scope fence
# lowered without its `s_waitcnt vmcnt(0)` on one control path, so that sweeps read the presolve's factor through
# stale L2 data -- wrote the waits out (agent_release / agent_acquire, fccqp_device.h) and restored the range.)
@pytest.mark.parametrize("dims", [(140, 60, 12, 100),     # N = 200: just past the LDS limit
                                  (160, 70, 30, 120),     # N = 230
                                  (196, 60, 12, 100),     # N = 256: all four slots of the 4-slot kernel full
                                  (130, 126, 9, 100),     # N = 256 with as many equality rows as the shape allows
                                  (180, 80, 12, 100),     # N = 260: first shape of the 8-slot kernel, five slots used
                                  (250, 134, 12, 100),    # N = 384
                                  (300, 212, 9, 100)])    # N = 512: the maximum, all eight slots full
def test_global_workspace_shapes_match_oracle(oracle, dims):
    from tests.test_gpu_parity import _random_qp
    n, m, nc, s = dims
    rng = np.random.default_rng(sum(dims))
    nb = 3
    B = P.stack([_random_qp(rng, n, m, nc, s) for _ in range(nb)])
    opts = dict(max_iter=25, rho=0.1, eps_fcone=1e-7, eps_bound=1e-7)
    z_ref, it_ref, det_ref, _ = oracle.solve_batch(dims, opts, *B.arrays(), nthreads=3)
    for rep in range(3):                      # fresh handles: recycled device memory, same answer
        z, it, d, solver = gpu_solve(dims, opts, B)
        assert_parity(z, it, z_ref, it_ref, tol=Z_TOL, what=f"{dims} rep {rep}")
        assert_residual_fields(d, det_ref, it, it_ref, str(dims))
        assert solver.kernel() == CAPI.KERNEL_GENERIC
        solver.close()


def test_shapes_beyond_the_supported_size_are_refused():
    """N = n + m > 512: FCCQP_ERR_UNSUPPORTED at construction (documented limit), not a wrong answer."""
    L = CAPI.lib()
    import ctypes
    h = ctypes.c_void_p()
    assert L.fccqp_create(400, 113, 12, 100, 1, 0, ctypes.byref(h)) == CAPI.ERR_UNSUPPORTED
    assert L.fccqp_create(600, 182, 9, 300, 1, 0, ctypes.byref(h)) == CAPI.ERR_UNSUPPORTED
    assert L.fccqp_create(330, 182, 9, 300, 1, 0, ctypes.byref(h)) == CAPI.OK       # N = 512
    L.fccqp_destroy(h)


def test_global_workspace_phases_do_not_see_each_other_stale(oracle):
    """The failure mode behind round 2's withdrawal of N > 256, pinned: a cold solve runs a presolve phase (assemble K,
    factorise, sweep) and then re-assembles K_rho over that factor in the same global workspace; with the fence
    between the phases lowered without its store wait, the ADMM sweeps read the presolve's factor and z was off by
    O(rho), differently from run to run.  Here: one iteration and two, cold, six fresh handles each, every result
    within rounding of the oracle AND bit-identical across the repetitions -- for the 4-slot and the 8-slot kernel."""
    from tests.test_gpu_parity import _random_qp
    for dims in ((160, 70, 30, 120), (180, 80, 12, 100), (300, 212, 9, 100)):
        n, m, nc, s = dims
        rng = np.random.default_rng(sum(dims))
        B = P.stack([_random_qp(rng, n, m, nc, s) for _ in range(2)])
        for max_iter in (1, 2):
            opts = dict(max_iter=max_iter, rho=0.1, eps_fcone=1e-7, eps_bound=1e-7)
            z_ref, it_ref, _, _ = oracle.solve_batch(dims, opts, *B.arrays(), nthreads=2)
            first = None
            for rep in range(6):
                z, it, d, solver = gpu_solve(dims, opts, B)
                solver.close()
                assert rel_err(z, z_ref).max() <= 1e-10, (dims, max_iter, rep, rel_err(z, z_ref))
                first = z if first is None else first
                assert np.array_equal(z, first), (dims, max_iter, rep)


@pytest.mark.parametrize("dims", [(160, 70, 30, 120), (250, 134, 30, 120)], ids=["N230", "N384"])
def test_global_workspace_warm_chain_and_rank_deficient_presolve(oracle, dims):
    """N = 230 / 384 (matrix in the global workspace, 4- and 8-slot kernel): a warm-started chain, and a cold batch in
    which one instance has a dead variable (zero row of K: the reference's COD branch)."""
    from tests.test_gpu_parity import _random_qp, _dead_variable
    n, m, nc, s = dims
    rng = np.random.default_rng(99)
    chains, steps = 2, 3
    opts = dict(max_iter=12, rho=0.1, eps_fcone=1e-7, eps_bound=1e-7)
    seqs = [[_random_qp(rng, n, m, nc, s) for _ in range(steps)] for _ in range(chains)]
    solver = CAPI.Solver(*dims, batch=chains); solver.set_options(**opts)
    orcs = []
    for c in range(chains):
        o = oracle.OracleFCCQP(*dims); op = oracle.FCCQPOptions(); op.__dict__.update(opts); o.set_options(op)
        orcs.append(o)
    for t in range(steps):
        B = P.stack([seqs[c][t] for c in range(chains)])
        solver.set_warm_start(t > 0)
        solver.solve_ok(*B.arrays())
        z, d = solver.solution()
        z_ref = np.zeros_like(z); it_ref = []
        for c, o in enumerate(orcs):
            o.set_warm_start(t > 0); o.Solve(**seqs[c][t]); sol = o.GetSolution()
            z_ref[c] = sol.z; it_ref.append(sol.details.n_iter)
        assert_parity(z, [x.n_iter for x in d], z_ref, it_ref, tol=Z_TOL, what=f"N={n + m} chain step {t}")
    qps = [_random_qp(rng, n, m, nc, s) for _ in range(3)]
    qps[1] = _dead_variable(qps[1], 7)
    B = P.stack(qps)
    z_ref, it_ref, det_ref, _ = oracle.solve_batch(dims, opts, *B.arrays(), nthreads=3)
    assert [det_ref[i].presolve_ldlt_info for i in range(3)] == [0, 1, 0]
    z, it, d, _ = gpu_solve(dims, opts, B)
    assert_parity(z, it, z_ref, it_ref, tol=Z_TOL, what=f"N={n + m} cod")
    assert [bool(x.flags & CAPI.FLAG_PRESOLVE_COD) for x in d] == [False, True, False]


# ---- (d) rank-deficient A_eq --------------------------------------------------------------------------
# The reference warns that it is untested for rank-deficient A_eq (fcc_qp.hpp:104-108), and its
# restatement shows why: with a duplicated equality row whose copy is NOT adjacent in the pivot
# order, Eigen's LDLT meets a pivot of rounding-error size instead of 0 and the oracle returns
# |A z - b_eq| ~ 1e3 (garbage; checked on the CPU).  Parity is only defined where the reference is
# well defined: an all-zero row (exact zero pivot over an exactly zero column -- Eigen leaves the
# column alone and LDLT::solve drops the pivot, the pseudo-inverse rule; the unpivoted LDL^T here
# does the same), and a duplicate of the LAST row placed right before it (nothing is eliminated
# between the two copies).
def _zero_row(qp, r):
    q = {k: np.array(v, copy=True) for k, v in qp.items()}
    q["A_eq"][r] = 0.0
    q["b_eq"][r] = 0.0
    return q


def _duplicate_last_row(qp):
    q = {k: np.array(v, copy=True) for k, v in qp.items()}
    q["A_eq"][-2] = q["A_eq"][-1]
    q["b_eq"][-2] = q["b_eq"][-1]
    return q


@pytest.mark.parametrize("kernel", KERNELS)
def test_rank_deficient_equality_rows(oracle, kernel):
    base = P.workload("W-mix", 8, P.WALKING)
    qps = [base.instance(i) for i in range(8)]
    zero = {1: 30, 4: 5}                 # instance -> all-zero row of A_eq
    dup = (2, 6)                         # instances whose last two rows are identical
    for i, r in zero.items():
        qps[i] = _zero_row(qps[i], r)
    for i in dup:
        qps[i] = _duplicate_last_row(qps[i])
    B = P.stack(qps)
    z_ref, it_ref, det_ref = oracle_batch(oracle, P.WALKING.ctor_args, P.HARNESS_OPTIONS, B)
    # the oracle is sane on this population (the premise of the comparison)
    r_ref = np.einsum("bij,bj->bi", B.A_eq, z_ref) - B.b_eq
    assert np.max(np.abs(r_ref)) <= 1e-6 * (1 + np.max(np.abs(B.b_eq)))
    z, it, d, _ = gpu_solve(P.WALKING.ctor_args, P.HARNESS_OPTIONS, B, kernel)
    assert np.all(np.isfinite(z))
    # stated tolerance for this population: the general one (1e-6 relative, |dn_iter| <= 1)
    assert_parity(z, it, z_ref, it_ref, tol=Z_TOL, what=f"rank-deficient A_eq k{kernel}")
    for i in range(8):
        if i in zero:      # exact zero pivot in the ADMM factorisation: reported, result still valid
            assert d[i].flags & CAPI.FLAG_ADMM_NUMERICAL, i
        elif i not in dup:
            assert not (d[i].flags & CAPI.FLAG_ADMM_NUMERICAL), i
    r = np.einsum("bij,bj->bi", B.A_eq, z) - B.b_eq
    assert np.max(np.abs(r)) <= 1e-6 * (1 + np.max(np.abs(B.b_eq)))


def test_duplicate_rows_inside_a_tile_equal_the_reduced_problem(oracle):
    """Two adjacent, identical equality rows in the middle of A_eq (KKT rows 70 and 71: one 16 x 16 diagonal
    tile of the fused kernel).  The reference's restatement is not well defined here (above), but the
    problem is: the copy adds nothing, so the solution is that of the QP without it.  The fused kernel's
    multipliers are exact quotients (fccqp_fused_impl.h: pivot_quotient), so the copy eliminates to an
    exactly zero column and pivot, which the pseudo-inverse rule drops -- compared with the oracle on
    the reduced shape FCCQP(60, 37, 12, 38)."""
    base = P.workload("W-mix", 6, P.WALKING)
    qps, reduced = [], []
    for i in range(6):
        q = {k: np.array(v, copy=True) for k, v in base.instance(i).items()}
        q["A_eq"][11] = q["A_eq"][10]; q["b_eq"][11] = q["b_eq"][10]
        qps.append(q)
        r = dict(q); r["A_eq"] = np.delete(q["A_eq"], 11, axis=0); r["b_eq"] = np.delete(q["b_eq"], 11)
        reduced.append(r)
    B = P.stack(qps)
    z, it, d, _ = gpu_solve(P.WALKING.ctor_args, P.HARNESS_OPTIONS, B, CAPI.KERNEL_FUSED)
    assert np.all(np.isfinite(z))
    for i, r in enumerate(reduced):
        o = oracle.OracleFCCQP(60, 37, 12, 38)
        op = oracle.FCCQPOptions(); op.__dict__.update(P.HARNESS_OPTIONS); o.set_options(op)
        o.Solve(**r)
        zr = o.GetSolution().z
        assert np.max(np.abs(z[i] - zr)) <= Z_TOL * (1 + np.max(np.abs(zr))), i
        assert d[i].flags & CAPI.FLAG_ADMM_NUMERICAL, i      # the dropped pivot is reported
    res = np.einsum("bij,bj->bi", B.A_eq, z) - B.b_eq
    assert np.max(np.abs(res)) <= 1e-6 * (1 + np.max(np.abs(B.b_eq)))


# ---- semi-definite costs: the reference's COD branch without the COD ------------------------------------
@pytest.mark.parametrize("kernel", KERNELS)
def test_w_psd_matches_oracle(oracle, kernel):
    """No cost on the constraint forces: the reference's LDLT fails on K and it takes the COD
    (fcc_qp.cpp:164-166).  K is non-singular, so x0 = K^-1 r either way."""
    B = P.workload("W-psd", 48, P.WALKING)
    for opts in (P.HARNESS_OPTIONS, dict(max_iter=0, rho=5e-5, eps_fcone=1e-6, eps_bound=1e-6), P.PAPER_OPTIONS):
        z_ref, it_ref, det_ref = oracle_batch(oracle, P.WALKING.ctor_args, opts, B)
        assert all(det_ref[i].presolve_ldlt_info == 1 for i in range(B.batch))
        z, it, d, _ = gpu_solve(P.WALKING.ctor_args, opts, B, kernel)
        assert_parity(z, it, z_ref, it_ref, tol=Z_TOL, what=f"W-psd k{kernel} max_iter={opts['max_iter']}")
        assert_residual_fields(d, det_ref, it, it_ref, f"W-psd k{kernel}")
        if kernel == CAPI.KERNEL_AUTO:
            # the fused kernel's presolve never factorises K: no breakdown, no fallback pass
            assert not any(x.flags & (CAPI.FLAG_PRESOLVE_NUMERICAL | CAPI.FLAG_PRESOLVE_COD) for x in d)


def _low_rank_cost_qp(rng, n, m, nc, s, rank):
    """Dense rank-deficient PSD cost Q = J^T J (J: rank x n) with a random full-rank A_eq whose
    null space the cost is definite on: K is non-singular and well conditioned, but the leading
    block of an unpivoted LDL^T of K is singular (a pivot of rounding-error size, not 0)."""
    J = rng.normal(size=(rank, n))
    lb = np.full(n, -np.inf); ub = np.full(n, np.inf)
    return dict(Q=J.T @ J, b=rng.normal(size=n), A_eq=rng.normal(size=(m, n)), b_eq=rng.normal(size=m),
                friction_coeffs=rng.uniform(0.3, 1.0, nc // 3), lb=lb, ub=ub)


def test_low_rank_dense_cost_equality_only(oracle):
    """nc = 0, infinite bounds: the presolve IS the answer (fcc_qp.cpp:132-133,159-178)."""
    rng = np.random.default_rng(21)
    n, m = 20, 18
    qps = [_low_rank_cost_qp(rng, n, m, 0, 0, rank=6) for _ in range(6)]
    B = P.stack(qps)
    s = CAPI.Solver(n, m, 0, 0, batch=6); s.set_options(100, 1e-3, 1e-6, 1e-6)
    s.solve_ok(*B.arrays())
    z, d = s.solution()
    for i, q in enumerate(qps):
        K = np.block([[q["Q"], q["A_eq"].T], [q["A_eq"], np.zeros((m, m))]])
        x = np.linalg.solve(K, np.r_[-q["b"], q["b_eq"]])[:n]
        assert np.max(np.abs(z[i] - x)) <= 1e-8 * (1 + np.max(np.abs(x))), i
        assert d[i].flags & CAPI.FLAG_EQUALITY_ONLY and d[i].n_iter == 0
        # the unpivoted presolve broke down on a tiny pivot and said so; the COD pass solved it
        assert d[i].flags & CAPI.FLAG_PRESOLVE_NUMERICAL and d[i].flags & CAPI.FLAG_PRESOLVE_COD
    o = oracle.OracleFCCQP(n, m, 0, 0)
    o.Solve(**qps[0])
    assert np.max(np.abs(o.GetSolution().z - z[0])) <= 1e-8 * (1 + np.max(np.abs(z[0])))


@pytest.mark.parametrize("dims", [(20, 14, 6, 10), (12, 5, 6, 3), (12, 8, 6, 3)], ids=["generic", "mini-fused", "padded-fused"])
def test_low_rank_dense_cost_with_admm(oracle, dims):
    """With contacts and bounds.  x0 (max_iter = 0: z = x0) is checked against K^-1 r from LAPACK.  The
    ADMM result is compared with the oracle on the instances where the oracle's OWN presolve was
    accurate: Eigen's LDLT reports success on some of these matrices after dividing by a pivot of
    rounding-error size, and the reference then starts ADMM from garbage (x0 off by O(1); seen in
    the restatement on the CPU) -- there is nothing to be in parity with on those."""
    n, m, nc, s = dims
    rng = np.random.default_rng(22)
    qps = []
    for _ in range(8):
        q = _low_rank_cost_qp(rng, n, m, nc, s, rank=max(2, (n - m) + 1))
        idx = [i for i in range(n) if not (s <= i < s + nc)][:3]
        q["lb"][idx] = -0.3; q["ub"][idx] = 0.3
        qps.append(q)
    B = P.stack(qps)
    x0 = np.zeros((B.batch, n))
    for i, q in enumerate(qps):
        K = np.block([[q["Q"], q["A_eq"].T], [q["A_eq"], np.zeros((m, m))]])
        x0[i] = np.linalg.solve(K, np.r_[-q["b"], q["b_eq"]])[:n]
    opts0 = dict(max_iter=0, rho=1e-2, eps_fcone=1e-7, eps_bound=1e-7)
    z0, _, d0, _ = gpu_solve(dims, opts0, B)
    assert rel_err(z0, x0).max() <= 1e-8, rel_err(z0, x0)
    z0_ref, _, _, _ = oracle.solve_batch(dims, opts0, *B.arrays(), nthreads=2)
    oracle_x0_err = rel_err(z0_ref, x0)
    sane = np.where(oracle_x0_err <= 1e-9)[0]
    dropped = np.where(oracle_x0_err > 1e-9)[0]
    opts = dict(max_iter=40, rho=1e-2, eps_fcone=1e-7, eps_bound=1e-7)
    z_ref, it_ref, det_ref, _ = oracle.solve_batch(dims, opts, *B.arrays(), nthreads=2)
    z, it, d, _ = gpu_solve(dims, opts, B)
    assert np.all(np.isfinite(z))
    assert_parity(z[sane], it[sane], z_ref[sane], it_ref[sane], tol=Z_TOL, what=f"low-rank cost {dims}")
    # The deviation, as a recorded number instead of a silent filter: how many of the 8 instances are NOT compared
    # with the oracle, how far the oracle's own presolve is from K^-1 r on them (the reason), and how far the HIP
    # result is from the oracle's there.  (The oracle is an unpinned restatement: this documents a property of the
    # restatement)  Pinned at the counts observed when the test was written.
    far = rel_err(z[dropped], z_ref[dropped]) if len(dropped) else np.zeros(0)
    print(f"low-rank cost {dims}: {len(dropped)} of {B.batch} instances not compared; oracle presolve error there "
          f"{oracle_x0_err[dropped].tolist()}, HIP-vs-oracle distance there {far.tolist()}")
    assert len(sane) >= 1, f"{len(dropped)} of {B.batch} instances dropped"
    # observed (the oracle and the generator are deterministic): 0, 6 and 2 of 8 dropped, the oracle's x0 off by
    # 0.07 .. 38 (relative) on them, and the HIP result within 6e-3 of the oracle's after 40 iterations all the same
    assert len(dropped) == {(20, 14, 6, 10): 0, (12, 5, 6, 3): 6, (12, 8, 6, 3): 2}[dims]
    assert far.size == 0 or far.max() <= 2e-2
    # on every dropped instance the HIP presolve IS accurate (checked above for all 8 against LAPACK), so the
    # distance to the oracle there is the oracle's own x0 error propagated through 40 iterations, not a HIP error
    for j, i in enumerate(dropped):
        assert rel_err(z0[i:i + 1], x0[i:i + 1])[0] <= 1e-8 and oracle_x0_err[i] > 1e-9


# ---- (e) the flag is visible from Python ------------------------------------------------------------
def test_presolve_cod_flag_exported_to_python():
    import fcc_qp_amd as F
    assert F.FLAG_PRESOLVE_COD == CAPI.FLAG_PRESOLVE_COD == 16
    from tests.test_gpu_parity import _dead_variable
    base = P.workload("W-mix", 4, P.WALKING)
    qps = [base.instance(i) for i in range(4)]
    qps[2] = _dead_variable(qps[2], 25)
    B = P.stack(qps)
    solver = F.FCCQPBatch(*P.WALKING.ctor_args, 4)
    o = F.FCCQPOptions(); o.max_iter, o.rho, o.eps_fcone, o.eps_bound = 100, 5e-5, 1e-6, 1e-6
    solver.set_options(o)
    solver.Solve(*B.arrays())
    fl = np.asarray(solver.GetSolution().flags)
    assert [bool(f & F.FLAG_PRESOLVE_COD) for f in fl] == [False, False, True, False]
