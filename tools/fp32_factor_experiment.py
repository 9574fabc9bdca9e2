"""SURVEY.md section 7 step 5 / hard part 3: would an fp32-STORED factor do?  (LDS: 20 KB per walking QP
instead of 39 KB, i.e. 8 instead of 4 QPs per CU.)  Emulation in numpy of exactly what such a kernel
would compute: the LDL^T of K_rho accumulated in fp64, every entry of L and D rounded to fp32 when it
is stored, solves in fp64 arithmetic on the rounded factor, optionally followed by steps of iterative
refinement against the fp64 K_rho (which a kernel would have to re-read from HBM: the factorisation
overwrote it).  Reported: the error of one KKT solve (the x-part, for the right-hand side of the first
ADMM iteration) and of the final z after the full ADMM run, relative to the fp64 path, on W-sat and
W-mix at the harness options.  IEEE arithmetic is the same on the host and on the device; no GPU needed.

    python tools/fp32_factor_experiment.py
"""
import os, sys
import numpy as np
import scipy.linalg as sl
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcc_qp_amd import problems as P
from oracle.numpy_ref import project_bounds, project_cones, _inf_norm


def ldl(K, store32):
    N = K.shape[0]; A = K.copy()
    for k in range(N):
        d = A[k, k]
        if store32: d = np.float64(np.float32(d)); A[k, k] = d
        col = A[k + 1:, k] / d
        if store32: col = col.astype(np.float32).astype(np.float64)
        A[k + 1:, k] = col
        A[k + 1:, k + 1:] -= np.outer(col, col) * d
    return np.tril(A, -1) + np.eye(N), np.diag(A).copy()


def solve(L, D, r):
    y = sl.solve_triangular(L, r, lower=True, unit_diagonal=True)
    return sl.solve_triangular(L.T, y / D, lower=False, unit_diagonal=True)


def admm(qp, opts, dims, store32, refine):
    n, m, nc, s = dims
    Q, A, b, beq = qp["Q"], qp["A_eq"], qp["b"], qp["b_eq"]
    rho = opts["rho"]
    K = np.zeros((n + m, n + m)); K[:n, :n] = Q + rho * np.eye(n); K[n:, :n] = A; K[:n, n:] = A.T
    L, D = ldl(K, store32)
    def kkt(r):
        x = solve(L, D, r)
        for _ in range(refine):
            x = x + solve(L, D, r - K @ x)
        return x
    K0 = K.copy(); K0[:n, :n] -= rho * np.eye(n)
    x = np.linalg.solve(K0, np.r_[-b, beq])[:n]             # presolve in fp64: the question is the ADMM factor
    xbar = x.copy(); lcbar = x[s:s + nc].copy(); mux = np.zeros(n); mul = np.zeros(nc)
    first = None; nit = opts["max_iter"]
    for it in range(opts["max_iter"]):
        w = xbar - mux; w[s:s + nc] = lcbar - mul
        x = kkt(np.r_[-(b - rho * w), beq])[:n]
        if first is None: first = x.copy()
        xbar = project_bounds(x + mux, qp["lb"], qp["ub"]); lcbar = project_cones(x[s:s + nc] + mul, qp["friction_coeffs"])
        xr = x - xbar; lr = x[s:s + nc] - lcbar; mux += xr; mul += lr
        if _inf_norm(lr) < opts["eps_fcone"] and _inf_norm(xr) < opts["eps_bound"]:
            nit = it; break
    return first, x, nit


if __name__ == "__main__":
    dims = P.WALKING.ctor_args
    for wl in ("W-sat", "W-mix"):
        B = P.workload(wl, 8, P.WALKING)
        for refine in (0, 1, 2):
            e1 = []; ez = []; dn = []
            for i in range(B.batch):
                f64 = admm(B.instance(i), P.HARNESS_OPTIONS, dims, False, 0)
                f32 = admm(B.instance(i), P.HARNESS_OPTIONS, dims, True, refine)
                e1.append(np.max(np.abs(f32[0] - f64[0])) / (1 + np.max(np.abs(f64[0]))))
                ez.append(np.max(np.abs(f32[1] - f64[1])) / (1 + np.max(np.abs(f64[1]))))
                dn.append(abs(f32[2] - f64[2]))
            print(f"{wl}: fp32-stored factor, {refine} refinement step(s): first KKT solve rel err max {max(e1):.1e}, "
                  f"final z rel err max {max(ez):.1e}, |dn_iter| max {max(dn)}  (tolerance 1e-6, |dn_iter| <= 1)")
