"""Independent numpy restatement of the reference ADMM (TEST INFRASTRUCTURE ONLY).

Second opinion for oracle/fccqp_oracle.c: same algorithm (/root/reference/src/fcc_qp.cpp:57-191,
/root/reference/src/constraint_utils.cpp:5-65) but written separately, vectorised, and with a
*different* linear solver (LAPACK LU via numpy.linalg.solve on the full symmetric KKT matrix, or
lstsq for the rank-deficient presolve) so that agreement between the two is evidence about the
algorithm and not about a shared linear-algebra routine.  PARITY UNPINNED (see fccqp_oracle.h).
"""
from __future__ import annotations

import numpy as np


def project_cone3(f, mu):
    """constraint_utils.cpp:5-25"""
    f = np.asarray(f, dtype=float)
    nxy = np.sqrt(f[0] * f[0] + f[1] * f[1])
    if mu * f[2] >= nxy:
        return f.copy()
    if f[2] < -mu * nxy:
        return np.zeros(3)
    k = mu * f[2] / nxy
    r = np.array([k * f[0], k * f[1], f[2]])
    z = r @ r
    if z > 0:
        r = r / np.sqrt(z)
    return (r @ f) * r


def project_cones(f, mus):
    """constraint_utils.cpp:27-35"""
    f = np.asarray(f, dtype=float)
    out = np.zeros_like(f)
    for i in range(f.size // 3):
        out[3 * i:3 * i + 3] = project_cone3(f[3 * i:3 * i + 3], mus[i])  # IndexError like .at()
    return out


def project_bounds(x, lb, ub):
    """constraint_utils.cpp:37-46"""
    return np.maximum(np.minimum(x, ub), lb)


def cone_violation(f, mus):
    """constraint_utils.cpp:48-59"""
    v = 0.0
    for i in range(len(f) // 3):
        v += max(0.0, np.sqrt(f[3 * i] ** 2 + f[3 * i + 1] ** 2) - mus[i] * f[3 * i + 2])
    return v


def bound_violation(x, lb, ub):
    """constraint_utils.cpp:61-65"""
    return float(np.linalg.norm(x - project_bounds(x, lb, ub)))


def _inf_norm(v):
    """fcc_qp.cpp:97-98  abs(max(maxCoeff, -minCoeff)); empty -> 0"""
    if v.size == 0:
        return 0.0
    return abs(max(v.max(), -v.min()))


class NumpyFCCQP:
    """fcc_qp.cpp:24-207 in numpy.  ``history`` collects per-iteration iterates when enabled."""

    def __init__(self, num_vars, num_equality_constraints, nc, lambda_c_start):
        assert nc % 3 == 0 and lambda_c_start + nc <= num_vars
        self.n, self.m, self.nc, self.s = num_vars, num_equality_constraints, nc, lambda_c_start
        self.max_iter, self.rho, self.eps_fcone, self.eps_bound = 1000, 1e-6, 1e-3, 1e-6
        self.warm_start = False
        self.x = np.zeros(self.n)
        self.mu_x = np.zeros(self.n)
        self.mu_lc = np.zeros(self.nc)
        self.keep_history = False
        self.history = []
        self.n_iter = 0
        self.x_res_norm = 0.0
        self.lc_res_norm = 0.0
        self.bounds_viol = 0.0
        self.friction_cone_viol = 0.0

    def set_options(self, max_iter, rho, eps_fcone, eps_bound):
        self.max_iter, self.rho, self.eps_fcone, self.eps_bound = max_iter, rho, eps_fcone, eps_bound

    def set_warm_start(self, w):
        self.warm_start = bool(w)

    def Solve(self, Q, b, A_eq, b_eq, friction_coeffs, lb, ub):
        n, m, nc, s = self.n, self.m, self.nc, self.s
        Q = np.asarray(Q, float); A = np.asarray(A_eq, float).reshape(m, n)
        b = np.asarray(b, float); b_eq = np.asarray(b_eq, float)
        lb = np.asarray(lb, float); ub = np.asarray(ub, float)
        mus = np.asarray(friction_coeffs, float)
        if mus.size < nc // 3:
            raise IndexError("friction_coeffs")
        Ql = np.tril(Q) + np.tril(Q, -1).T          # LDLT reads the lower triangle only
        eq_only = nc == 0 and np.all(np.isinf(lb)) and np.all(np.isinf(ub))   # :132-133
        if not self.warm_start:                      # :136-139
            self.mu_x[:] = 0; self.mu_lc[:] = 0
        K = np.zeros((n + m, n + m))
        K[:n, :n] = Ql; K[n:, :n] = A; K[:n, n:] = A.T
        rhs = np.concatenate([-b, b_eq])
        self.n_iter = 0; self.x_res_norm = 0.0; self.lc_res_norm = 0.0
        self.history = []
        if eq_only or not self.warm_start:           # :159-178
            try:
                if np.linalg.matrix_rank(K) < n + m:
                    raise np.linalg.LinAlgError
                sol = np.linalg.solve(K, rhs)
            except np.linalg.LinAlgError:
                sol = np.linalg.lstsq(K, rhs, rcond=None)[0]   # min-norm LS, as COD
            self.x = sol[:n].copy()
        if not eq_only:                              # :180-182 -> DoADMM :57-112
            Kr = K.copy()
            Kr[:n, :n] += self.rho * np.eye(n)
            lu = np.linalg.inv(Kr) if False else None
            x_bar = self.x.copy(); lc_bar = self.x[s:s + nc].copy()
            self.n_iter = self.max_iter
            for it in range(self.max_iter):
                q = -self.rho * (x_bar - self.mu_x)
                q[s:s + nc] = -self.rho * (lc_bar - self.mu_lc)
                r = rhs.copy(); r[:n] = -(b + q)
                self.x = np.linalg.solve(Kr, r)[:n]
                x_bar = project_bounds(self.x + self.mu_x, lb, ub)
                lc_bar = project_cones(self.x[s:s + nc] + self.mu_lc, mus)
                x_res = self.x - x_bar
                lc_res = self.x[s:s + nc] - lc_bar
                self.x_res_norm = _inf_norm(x_res); self.lc_res_norm = _inf_norm(lc_res)
                self.mu_x = self.mu_x + x_res; self.mu_lc = self.mu_lc + lc_res
                if self.keep_history:
                    self.history.append(dict(x=self.x.copy(), x_bar=x_bar.copy(), mu_x=self.mu_x.copy(),
                                             lambda_c_bar=lc_bar.copy(), mu_lambda_c=self.mu_lc.copy()))
                if self.lc_res_norm < self.eps_fcone and self.x_res_norm < self.eps_bound:
                    self.n_iter = it
                    break
        self.bounds_viol = bound_violation(self.x, lb, ub)
        self.friction_cone_viol = cone_violation(self.x[s:s + nc], mus)
        return self.x
