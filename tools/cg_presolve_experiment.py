"""Would conjugate gradients on W = I - rho G be a cheaper presolve than factorising W?  Emulates the arithmetic on the
CPU (IEEE fp64 as on the device): iterations to reach a relative residual, and the error of x0 against the direct solve,
per workload.  G has rank n - m, so CG ends in at most n - m + 1 steps in exact arithmetic whatever cond(W) is.
    python tools/cg_presolve_experiment.py"""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from fcc_qp_amd import problems as P

def w_and_xc(qp, rho, n, m):
    Q = np.tril(qp['Q']); Q = Q + np.tril(Q, -1).T
    A = qp['A_eq']; N = n + m
    K = np.zeros((N, N)); K[:n, :n] = Q + rho * np.eye(n); K[n:, :n] = A; K[:n, n:] = A.T
    r = np.concatenate([-qp['b'], qp['b_eq']])
    Kinv = np.linalg.inv(K)
    G = Kinv[:n, :n]; xc = (Kinv @ r)[:n]
    return np.eye(n) - rho * G, xc

def cg(W, b, tol, cap):
    x = b.copy(); r = b - W @ x; p = r.copy(); rs = r @ r; bb = b @ b
    for it in range(1, cap + 1):
        Ap = W @ p
        alpha = rs / (p @ Ap)
        x += alpha * p; r -= alpha * Ap
        rs_new = r @ r
        if rs_new <= tol * tol * bb: return x, it
        p = r + (rs_new / rs) * p; rs = rs_new
    return x, cap + 1

for shape, name in ((P.WALKING, "walking"), (P.LARGE, "large")):
    n, m, nc, s = shape.ctor_args
    for wl in ("W-sat", "W-mix", "W-psd"):
        nb = 24 if name == "walking" else 4
        try:
            B = P.workload(wl, nb, shape)
        except Exception as e:
            print(name, wl, "n/a", e); continue
        its, errs, conds = [], [], []
        for i in range(nb):
            W, xc = w_and_xc(B.instance(i), P.HARNESS_OPTIONS["rho"], n, m)
            W = 0.5 * (W + W.T)
            xd = np.linalg.solve(W, xc)
            x, it = cg(W, xc, 1e-13, 80)
            its.append(it); errs.append(np.max(np.abs(x - xd)) / (1 + np.max(np.abs(xd))))
            ev = np.linalg.eigvalsh(W); conds.append(ev[-1] / ev[0])
        print(f"{name:8s} {wl:6s} n-m={n-m:3d}  CG iterations min/med/max {min(its)}/{int(np.median(its))}/{max(its)}   "
              f"max err vs direct {max(errs):.1e}   cond(W) {min(conds):.1e} .. {max(conds):.1e}")
