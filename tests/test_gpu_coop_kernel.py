"""The cooperative kernel (FCCQP_KERNEL_COOP, fccqp_coop_impl.h: one workgroup of several wavefronts per QP, the
register tiles shared out over the waves) against the oracle, through the C ABI.

Covered: every shape of coop_shapes.def (two / four waves, one / two row halves, n a multiple of 16, ragged tiles),
cold solves, warm-start chains, option sets with intermediate n_iter, the reported residual norms and violation
metrics, fp32 storage, instances that leave for the COD pass, batches larger than one round, and agreement with
the one-wave fused kernel and the generic kernel on the shapes they share.
Reference behaviour: FCCQP::Solve / DoADMM, /root/reference/src/fcc_qp.cpp:57-191."""
import numpy as np
import pytest

from fcc_qp_amd import problems as P
from tests import capi as CAPI
from tests.helpers import Z_TOL, Z_TOL_TIGHT, assert_parity, oracle_batch

pytestmark = pytest.mark.gpu

COOP_DIMS = [(114, 60, 24, 66), (60, 38, 12, 38), (70, 20, 9, 40), (32, 14, 6, 20), (20, 8, 6, 5)]


class Dims:
    def __init__(self, dims):
        self.ctor_args = dims


def random_batch(dims, nb, seed, bounded=3):
    """Dense well-conditioned QPs of an arbitrary shape: SPD cost, full-rank equalities, a few finite bounds (some on
    contact variables: the lambda_c overwrite quirk of fcc_qp.cpp:82), friction 0.3 .. 0.9."""
    n, m, nc, s = dims
    rng = np.random.default_rng(seed)
    Lq = rng.standard_normal((nb, n, n)) / np.sqrt(n)
    Q = Lq @ np.transpose(Lq, (0, 2, 1)) + 0.3 * np.eye(n)
    A = rng.standard_normal((nb, m, n))
    b = rng.standard_normal((nb, n))
    beq = 0.3 * rng.standard_normal((nb, m))
    lb = np.full((nb, n), -np.inf); ub = np.full((nb, n), np.inf)
    idx = rng.permutation(n)[:bounded + 2]
    lb[:, idx] = -0.15; ub[:, idx] = 0.2
    lb[:, s] = -0.05                                   # a finite bound on a contact variable
    mu = rng.uniform(0.3, 0.9, (nb, nc // 3))
    return P.Batch(Q, b, A, beq, mu, lb, ub)


def gpu(dims, options, batch, kernel, warm_steps=0, f32=False):
    s = CAPI.Solver(*dims, batch=batch.batch)
    s.set_options(**options); s.set_kernel(kernel)
    outs = []
    for t in range(1 + warm_steps):
        s.set_warm_start(t > 0)
        (s.solve_f32_ok if f32 else s.solve_ok)(*batch.arrays())
        z, d = s.solution()
        outs.append((z, np.array([x.n_iter for x in d]), d, s.state()))
    assert s.kernel() == kernel
    return outs


@pytest.mark.parametrize("dims", COOP_DIMS)
def test_coop_random_population_matches_oracle(oracle, dims):
    opts = dict(max_iter=80, rho=1e-2, eps_fcone=1e-7, eps_bound=1e-7)
    B = random_batch(dims, 24, seed=sum(dims))
    z_ref, it_ref, det_ref = oracle_batch(oracle, dims, opts, B)
    out = gpu(dims, opts, B, CAPI.KERNEL_COOP)
    z, it, d, _ = out[0]
    assert_parity(z, it, z_ref, it_ref, tol=Z_TOL_TIGHT, what=f"coop {dims}")
    for i in range(B.batch):
        assert d[i].solve_status == det_ref[i].solve_status
        for f in ("admm_residual_bounds", "admm_residual_friction_cone", "bounds_viol", "friction_cone_viol"):
            got, ref = getattr(d[i], f), getattr(det_ref[i], f)
            assert abs(got - ref) <= 1e-6 * (1 + abs(ref)), (dims, i, f, got, ref)
        assert d[i].flags & CAPI.FLAG_PRESOLVE_RAN and not (d[i].flags & (CAPI.FLAG_PRESOLVE_NUMERICAL | CAPI.FLAG_ADMM_NUMERICAL))


@pytest.mark.parametrize("dims", COOP_DIMS)
def test_coop_warm_start_chain(oracle, dims):
    opts = dict(max_iter=25, rho=1e-2, eps_fcone=1e-9, eps_bound=1e-9)
    nb, steps = 6, 4
    seq = [random_batch(dims, nb, seed=100 + t) for t in range(steps)]
    solver = CAPI.Solver(*dims, batch=nb); solver.set_options(**opts); solver.set_kernel(CAPI.KERNEL_COOP)
    orcs = []
    for c in range(nb):
        o = oracle.OracleFCCQP(*dims); op = oracle.FCCQPOptions(); op.__dict__.update(opts); o.set_options(op); orcs.append(o)
    for t in range(steps):
        solver.set_warm_start(t > 0)
        solver.solve_ok(*seq[t].arrays())
        z, d = solver.solution()
        z_ref = np.zeros_like(z); it_ref = []
        for c, o in enumerate(orcs):
            o.set_warm_start(t > 0)
            o.Solve(**seq[t].instance(c)); sol = o.GetSolution()
            z_ref[c] = sol.z; it_ref.append(sol.details.n_iter)
        assert_parity(z, [x.n_iter for x in d], z_ref, it_ref, tol=Z_TOL, what=f"coop chain {dims} step {t}")
        if t > 0:
            assert all(not (x.flags & CAPI.FLAG_PRESOLVE_RAN) for x in d)
    x, mux, mulc = solver.state()
    for c, o in enumerate(orcs):
        xo, mo, lo = o.get_state()
        assert np.max(np.abs(mux[c] - mo)) <= 1e-6 * (1 + np.max(np.abs(mo)))
        assert np.max(np.abs(mulc[c] - lo)) <= 1e-6 * (1 + np.max(np.abs(lo)))


@pytest.mark.parametrize("wl", ["W-sat", "W-mix", "W-psd"])
def test_coop_walking_workloads(oracle, wl):
    B = P.workload(wl, 96, P.WALKING)
    z_ref, it_ref, _ = oracle_batch(oracle, P.WALKING.ctor_args, P.HARNESS_OPTIONS, B)
    z, it, d, _ = gpu(P.WALKING.ctor_args, P.HARNESS_OPTIONS, B, CAPI.KERNEL_COOP)[0]
    assert_parity(z, it, z_ref, it_ref, tol=Z_TOL_TIGHT if wl != "W-psd" else Z_TOL, what=f"coop walking {wl}")
    # and the one-wave kernel agrees to rounding
    z1, it1, _, _ = gpu(P.WALKING.ctor_args, P.HARNESS_OPTIONS, B, CAPI.KERNEL_FUSED)[0]
    assert np.max(np.abs(z - z1)) <= 1e-8 * (1 + np.max(np.abs(z1))) and np.max(np.abs(it - it1)) <= 1


@pytest.mark.parametrize("options", [P.PAPER_OPTIONS, dict(max_iter=0, rho=5e-5, eps_fcone=1e-6, eps_bound=1e-6),
                                     dict(max_iter=1, rho=5e-5, eps_fcone=1e-6, eps_bound=1e-6),
                                     dict(max_iter=1000, rho=1e-2, eps_fcone=1e-6, eps_bound=1e-6),
                                     dict(max_iter=3, rho=10.0, eps_fcone=1e9, eps_bound=1e9)])
def test_coop_option_sets_walking_and_large(oracle, options):
    for shape, nb in ((P.WALKING, 32), (P.LARGE, 8)):
        B = P.workload("W-mix", nb, shape)
        z_ref, it_ref, _ = oracle_batch(oracle, shape.ctor_args, options, B)
        z, it, _, _ = gpu(shape.ctor_args, options, B, CAPI.KERNEL_COOP, warm_steps=0)[0]
        assert_parity(z, it, z_ref, it_ref, tol=Z_TOL, what=f"coop {shape.ctor_args} {options}")


def test_coop_large_shape_workloads_and_auto_choice(oracle):
    for wl in ("W-sat", "W-mix"):
        B = P.workload(wl, 16, P.LARGE)
        z_ref, it_ref, det_ref = oracle_batch(oracle, P.LARGE.ctor_args, P.HARNESS_OPTIONS, B)
        out = gpu(P.LARGE.ctor_args, P.HARNESS_OPTIONS, B, CAPI.KERNEL_COOP, warm_steps=1)
        assert_parity(out[0][0], out[0][1], z_ref, it_ref, tol=Z_TOL_TIGHT, what=f"coop large {wl}")
    # AUTO takes the cooperative kernel for the large shape (no one-wave kernel for n >= 64) ...
    s = CAPI.Solver(*P.LARGE.ctor_args, batch=4)
    assert s.kernel() == CAPI.KERNEL_COOP
    # ... and keeps the one-wave kernel for the walking shape at every batch size (measured: tools/coop_bench.py)
    assert CAPI.Solver(*P.WALKING.ctor_args, batch=1).kernel() == CAPI.KERNEL_FUSED
    assert CAPI.Solver(*P.WALKING.ctor_args, batch=4096).kernel() == CAPI.KERNEL_FUSED


def test_coop_rank_deficient_presolve_goes_to_the_cod_pass(oracle):
    for shape in (P.WALKING, P.LARGE):
        base = P.workload("W-mix", 6, shape)
        qps = [base.instance(i) for i in range(6)]
        dead = {1: shape.nv + 2, 4: shape.nv + 5}          # an actuator variable without cost or constraint
        for i, j in dead.items():
            q = {k: np.array(v, copy=True) for k, v in qps[i].items()}
            q["Q"][j, :] = 0.0; q["Q"][:, j] = 0.0; q["A_eq"][:, j] = 0.0; q["b"][j] = 0.0
            qps[i] = q
        B = P.stack(qps)
        z_ref, it_ref, det_ref = oracle_batch(oracle, shape.ctor_args, P.HARNESS_OPTIONS, B)
        z, it, d, _ = gpu(shape.ctor_args, P.HARNESS_OPTIONS, B, CAPI.KERNEL_COOP)[0]
        assert_parity(z, it, z_ref, it_ref, tol=Z_TOL, what=f"coop cod {shape.ctor_args}")
        for i in range(6):
            assert bool(d[i].flags & CAPI.FLAG_PRESOLVE_COD) == (i in dead)


def test_coop_fp32_storage_equals_fp64_path_on_widened_values():
    for dims in ((60, 38, 12, 38), (70, 20, 9, 40)):
        B = random_batch(dims, 12, seed=5)
        B32 = P.Batch(*(a.astype(np.float32).astype(np.float64) for a in B.arrays()))
        opts = dict(max_iter=40, rho=1e-2, eps_fcone=1e-7, eps_bound=1e-7)
        a = gpu(dims, opts, B32, CAPI.KERNEL_COOP)[0]
        b = gpu(dims, opts, B32, CAPI.KERNEL_COOP, f32=True)[0]
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_coop_batch_of_several_rounds_is_permutation_equivariant():
    nb = 700                                                # > 256 workgroups: several rounds, ragged permutation domain
    B = P.workload("W-mix", nb, P.WALKING, distinct=64)
    z, it, _, _ = gpu(P.WALKING.ctor_args, P.HARNESS_OPTIONS, B, CAPI.KERNEL_COOP)[0]
    perm = np.random.default_rng(3).permutation(nb)
    Bp = P.Batch(*(np.ascontiguousarray(a[perm]) for a in B.arrays()))
    zp, itp, _, _ = gpu(P.WALKING.ctor_args, P.HARNESS_OPTIONS, Bp, CAPI.KERNEL_COOP)[0]
    assert np.array_equal(zp, z[perm]) and np.array_equal(itp, it[perm])
    assert np.array_equal(z[:64], z[64:128])                # tiled copies are bit-identical


def test_coop_refused_for_an_unlisted_shape():
    s = CAPI.Solver(10, 4, 3, 2, batch=2)
    s.set_kernel(CAPI.KERNEL_COOP)
    B = random_batch((10, 4, 3, 2), 2, seed=1)
    assert s.solve(*B.arrays()) == CAPI.ERR_UNSUPPORTED


def test_a_refused_solve_leaves_no_stale_solution_and_the_handle_usable():
    """A solve that fails after an earlier good one: GetSolution must not hand the earlier z out as this solve's (it
    answers FCCQP_ERR_NOT_SOLVED), and the handle solves again once the request is one it can serve."""
    shape = (10, 4, 3, 2)
    s = CAPI.Solver(*shape, batch=2)
    s.set_options(**P.HARNESS_OPTIONS)
    B = random_batch(shape, 2, seed=1)
    s.solve_ok(*B.arrays())
    z0, d0 = s.solution()
    s.set_kernel(CAPI.KERNEL_COOP)
    assert s.solve(*B.arrays()) == CAPI.ERR_UNSUPPORTED
    z = np.zeros((2, 10))
    d = (CAPI.Details * 2)()
    assert s.L.fccqp_get_solution(s.h, z.ctypes.data_as(CAPI.C.POINTER(CAPI.C.c_double)), d) == CAPI.ERR_NOT_SOLVED
    s.set_kernel(CAPI.KERNEL_AUTO)
    s.solve_ok(*B.arrays())
    z1, d1 = s.solution()
    assert np.array_equal(z0, z1) and [x.n_iter for x in d0] == [x.n_iter for x in d1]

