"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle):
CPU: the oracle and the numpy restatement reproduce them; GPU: the HIP path reproduces them."""
import glob
import os

import numpy as np
import pytest

from oracle.numpy_ref import NumpyFCCQP
from tests.helpers import Z_TOL

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))


def load(path):
    g = np.load(path)
    dims = tuple(int(v) for v in g["dims"])
    o = g["options"]
    opts = dict(max_iter=int(o[0]), rho=float(o[1]), eps_fcone=float(o[2]), eps_bound=float(o[3]))
    qps = [dict(Q=g["Q"][i], b=g["b"][i], A_eq=g["A_eq"][i], b_eq=g["b_eq"][i],
                friction_coeffs=g["friction_coeffs"][i], lb=g["lb"][i], ub=g["ub"][i]) for i in range(g["Q"].shape[0])]
    return g, dims, opts, qps


def test_fixtures_exist():
    assert len(FIXTURES) >= 4


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_oracle_reproduces_golden(oracle, path):
    g, dims, opts, qps = load(path)
    warm = "warm_chain" in path
    o = oracle.OracleFCCQP(*dims)
    op = oracle.FCCQPOptions(); op.__dict__.update(opts); o.set_options(op)
    for i, qp in enumerate(qps):
        o.set_warm_start(warm and i > 0)
        o.enable_trace(6)
        o.Solve(**qp); s = o.GetSolution(); tr = o.trace()
        assert s.details.n_iter == g["out_n_iter"][i]
        assert s.details.solve_status == g["out_solve_status"][i]
        # same code, possibly another compiler/CPU: allow rounding-level drift only
        assert np.max(np.abs(s.z - g["out_z"][i])) <= 1e-10 * (1 + np.max(np.abs(g["out_z"][i])))
        k = int(g["out_trace_len"][i])
        assert len(tr["x"]) == k
        assert np.allclose(tr["x"], g["out_trace_x"][i][:k], rtol=1e-10, atol=1e-10)
        assert np.allclose(tr["mu_x"], g["out_trace_mu_x"][i][:k], rtol=1e-10, atol=1e-10)
        for key, mine in (("bounds_viol", s.details.bounds_viol), ("friction_cone_viol", s.details.friction_cone_viol),
                          ("eps_bounds", s.details.eps_bounds), ("eps_friction_cone", s.details.eps_friction_cone)):
            ref = float(g["out_" + key][i])
            assert abs(mine - ref) <= 1e-9 * (1 + abs(ref)), key


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_numpy_restatement_reproduces_golden(path):
    g, dims, opts, qps = load(path)
    warm = "warm_chain" in path
    r = NumpyFCCQP(*dims); r.set_options(**opts); r.keep_history = True
    for i, qp in enumerate(qps):
        r.set_warm_start(warm and i > 0)
        z = r.Solve(**qp)
        assert abs(r.n_iter - int(g["out_n_iter"][i])) <= 1
        assert np.max(np.abs(z - g["out_z"][i])) <= 1e-8 * (1 + np.max(np.abs(g["out_z"][i])))


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", [1, 0], ids=["generic", "auto"])
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_hip_path_reproduces_golden(path, kernel):
    from tests import capi as C
    g, dims, opts, qps = load(path)
    warm = "warm_chain" in path
    s = C.Solver(*dims, batch=1); s.set_options(**opts); s.set_kernel(kernel)
    for i, qp in enumerate(qps):
        s.set_warm_start(warm and i > 0)
        s.solve_ok(qp["Q"][None], qp["b"][None], qp["A_eq"][None], qp["b_eq"][None],
                   qp["friction_coeffs"][None], qp["lb"][None], qp["ub"][None])
        z, d = s.solution()
        zr = g["out_z"][i]
        assert np.max(np.abs(z[0] - zr)) <= Z_TOL * (1 + np.max(np.abs(zr)))
        assert abs(d[0].n_iter - int(g["out_n_iter"][i])) <= 1
        assert d[0].solve_status == int(g["out_solve_status"][i]) or abs(d[0].n_iter - int(g["out_n_iter"][i])) == 1
        for key, mine in (("bounds_viol", d[0].bounds_viol), ("friction_cone_viol", d[0].friction_cone_viol)):
            ref = float(g["out_" + key][i])
            assert abs(mine - ref) <= 1e-6 * (1 + abs(ref)), key
        # the reported ADMM residual norms (python: eps_bounds / eps_friction_cone, main.cpp:24-25) are those of
        # the last iteration executed: comparable when both ran the same number of iterations.  They are
        # differences of iterates, so the tolerance is relative to the iterate, like z's.
        if d[0].n_iter == int(g["out_n_iter"][i]):
            scale = 1 + np.max(np.abs(zr))
            for key, mine in (("eps_bounds", d[0].admm_residual_bounds), ("eps_friction_cone", d[0].admm_residual_friction_cone)):
                ref = float(g["out_" + key][i])
                assert abs(mine - ref) <= Z_TOL * scale, (key, mine, ref)
