"""ctypes binding of the CPU oracle (oracle/fccqp_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package (fcc_qp_amd) never does.  PARITY UNPINNED: see fccqp_oracle.h.

The class below mirrors the Python surface of the reference binding
(/root/reference/src/main.cpp:19-56) so that tests read like the reference harness.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfccqp_oracle.so")


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (make -C oracle)."""
    src = os.path.join(_HERE, "fccqp_oracle.c")
    stale = (not os.path.exists(_LIB_PATH)) or (
        os.path.exists(src) and os.path.getmtime(_LIB_PATH) < os.path.getmtime(src))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


class Details(C.Structure):
    """fccqp_oracle_details (FCCQPDetails, /root/reference/src/fcc_qp.hpp:19-28, + diagnostics)."""

    _fields_ = [
        ("n_iter", C.c_int),
        ("admm_residual_bounds", C.c_double),
        ("admm_residual_friction_cone", C.c_double),
        ("solve_time", C.c_double),
        ("factorization_time", C.c_double),
        ("bounds_viol", C.c_double),
        ("friction_cone_viol", C.c_double),
        ("solve_status", C.c_int),
        ("presolve_ran", C.c_int),
        ("presolve_ldlt_info", C.c_int),
    ]

    # Python attribute names of the reference binding (main.cpp:24-25)
    @property
    def eps_bounds(self):
        return self.admm_residual_bounds

    @property
    def eps_friction_cone(self):
        return self.admm_residual_friction_cone


_lib = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.fccqp_oracle_create.restype = C.c_void_p
        L.fccqp_oracle_create.argtypes = [C.c_int] * 4
        L.fccqp_oracle_destroy.argtypes = [C.c_void_p]
        L.fccqp_oracle_set_options.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double]
        L.fccqp_oracle_get_options.argtypes = [C.c_void_p, _ip, _dp, _dp, _dp]
        L.fccqp_oracle_set_warm_start.argtypes = [C.c_void_p, C.c_int]
        L.fccqp_oracle_set_trace.argtypes = [C.c_void_p, _dp, C.c_int]
        L.fccqp_oracle_trace_count.argtypes = [C.c_void_p]
        L.fccqp_oracle_trace_count.restype = C.c_int
        L.fccqp_oracle_solve.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp, C.c_int, _dp, _dp]
        L.fccqp_oracle_solve.restype = C.c_int
        L.fccqp_oracle_get_solution.argtypes = [C.c_void_p, _dp, C.POINTER(Details)]
        L.fccqp_oracle_get_state.argtypes = [C.c_void_p, _dp, _dp, _dp]
        L.fccqp_oracle_set_state.argtypes = [C.c_void_p, _dp, _dp, _dp]
        L.fccqp_oracle_project_cone3.argtypes = [_dp, C.c_double, _dp]
        L.fccqp_oracle_project_cones.argtypes = [_dp, C.c_int, _dp, C.c_int, _dp]
        L.fccqp_oracle_project_cones.restype = C.c_int
        L.fccqp_oracle_project_bounds.argtypes = [_dp, _dp, _dp, C.c_int, _dp]
        L.fccqp_oracle_cone_violation.argtypes = [_dp, C.c_int, _dp, C.c_int, _dp]
        L.fccqp_oracle_cone_violation.restype = C.c_int
        L.fccqp_oracle_bound_violation.argtypes = [_dp, _dp, _dp, C.c_int]
        L.fccqp_oracle_bound_violation.restype = C.c_double
        L.fccqp_oracle_validate_bounds.argtypes = [_dp, _dp, C.c_int]
        L.fccqp_oracle_validate_bounds.restype = C.c_int
        L.fccqp_oracle_ldlt_compute.argtypes = [_dp, C.c_int, _ip, _dp]
        L.fccqp_oracle_ldlt_compute.restype = C.c_int
        L.fccqp_oracle_ldlt_solve.argtypes = [_dp, C.c_int, _ip, _dp]
        L.fccqp_oracle_cod_solve.argtypes = [_dp, C.c_int, _dp, _dp]
        L.fccqp_oracle_cod_solve.restype = C.c_int
        L.fccqp_oracle_solve_batch.argtypes = (
            [C.c_int] * 5 + [C.c_double] * 3 + [C.c_int] + [_dp] * 7 + [_dp, _ip, C.POINTER(Details), C.c_int])
        L.fccqp_oracle_solve_batch.restype = C.c_double
        _lib = L
    return _lib


def _d(a):
    """float64 C-contiguous view/copy + pointer (what pybind's Eigen casters do for the reference)."""
    arr = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    return arr, arr.ctypes.data_as(_dp)


class FCCQPOptions:
    """/root/reference/src/fcc_qp.hpp:30-35 ; python attrs main.cpp:31-36."""

    def __init__(self):
        self.max_iter = 1000
        self.rho = 1e-6
        self.eps_fcone = 1e-3
        self.eps_bound = 1e-6


class FCCQPSolution:
    """/root/reference/src/fcc_qp.hpp:37-40 ; python attrs main.cpp:38-40."""

    def __init__(self, details: Details, z: np.ndarray):
        self.details = details
        self.z = z


class OracleFCCQP:
    """The oracle behind the Python surface of the reference's ``FCCQP`` (main.cpp:42-54)."""

    def __init__(self, num_vars: int, num_equality_constraints: int, nc: int, lambda_c_start: int):
        self._L = lib()
        self._h = self._L.fccqp_oracle_create(num_vars, num_equality_constraints, nc, lambda_c_start)
        if not self._h:
            raise ValueError("invalid FCCQP sizes")
        self.n, self.m, self.nc, self.s = num_vars, num_equality_constraints, nc, lambda_c_start
        self._trace = None

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.fccqp_oracle_destroy(self._h)
            self._h = None

    def _opts(self):
        mi = C.c_int(); r = C.c_double(); ef = C.c_double(); eb = C.c_double()
        self._L.fccqp_oracle_get_options(self._h, C.byref(mi), C.byref(r), C.byref(ef), C.byref(eb))
        return mi.value, r.value, ef.value, eb.value

    def set_rho(self, rho: float):
        mi, _, ef, eb = self._opts()
        self._L.fccqp_oracle_set_options(self._h, mi, rho, ef, eb)

    def set_max_iter(self, n: int):
        _, r, ef, eb = self._opts()
        self._L.fccqp_oracle_set_options(self._h, int(n), r, ef, eb)

    def set_warm_start(self, warm_start: bool):
        self._L.fccqp_oracle_set_warm_start(self._h, int(bool(warm_start)))

    def set_options(self, opt):
        self._L.fccqp_oracle_set_options(self._h, int(opt.max_iter), float(opt.rho),
                                         float(opt.eps_fcone), float(opt.eps_bound))

    def enable_trace(self, max_records: int):
        rec = 3 * self.n + 2 * self.nc
        self._trace = np.zeros((max_records, rec))
        self._L.fccqp_oracle_set_trace(self._h, self._trace.ctypes.data_as(_dp), max_records)

    def trace(self):
        """dict of per-iteration iterates recorded since enable_trace()."""
        k = self._L.fccqp_oracle_trace_count(self._h)
        t = self._trace[:k]
        n, nc = self.n, self.nc
        return dict(x=t[:, :n], x_bar=t[:, n:2 * n], mu_x=t[:, 2 * n:3 * n],
                    lambda_c_bar=t[:, 3 * n:3 * n + nc], mu_lambda_c=t[:, 3 * n + nc:])

    def Solve(self, Q, b, A_eq, b_eq, friction_coeffs, lb, ub):
        Qa, Qp = _d(Q); ba, bp = _d(b); Aa, Ap = _d(A_eq); bea, bep = _d(b_eq)
        fa, fp = _d(friction_coeffs); la, lp = _d(lb); ua, up = _d(ub)
        if Qa.shape != (self.n, self.n) or ba.shape != (self.n,) or la.shape != (self.n,) \
                or ua.shape != (self.n,) or Aa.size != self.m * self.n \
                or bea.shape != (self.m,):
            raise ValueError("shape mismatch")
        rc = self._L.fccqp_oracle_solve(self._h, Qp, bp, Ap, bep, fp, int(fa.size), lp, up)
        if rc != 0:
            raise IndexError("friction_coeffs has fewer than nc/3 entries")

    def GetSolution(self) -> FCCQPSolution:
        z = np.zeros(self.n)
        d = Details()
        self._L.fccqp_oracle_get_solution(self._h, z.ctypes.data_as(_dp), C.byref(d))
        return FCCQPSolution(d, z)

    def get_state(self):
        x = np.zeros(self.n); mx = np.zeros(self.n); ml = np.zeros(max(self.nc, 1))
        self._L.fccqp_oracle_get_state(self._h, x.ctypes.data_as(_dp), mx.ctypes.data_as(_dp),
                                       ml.ctypes.data_as(_dp))
        return x, mx, ml[: self.nc]

    def set_state(self, x, mu_x, mu_lc):
        xa, xp = _d(x); ma, mp = _d(mu_x); la, lp = _d(np.concatenate([np.asarray(mu_lc, float), [0.0]]))
        self._L.fccqp_oracle_set_state(self._h, xp, mp, lp)


def solve_batch(ctor_args, options: dict, Q, b, A, b_eq, mu, lb, ub, nthreads: int = 1):
    """Cold-start solve of an instance-major batch; returns (z[B,n], n_iter[B], details[B], seconds)."""
    L = lib()
    n, m, nc, s = ctor_args
    B = Q.shape[0]
    arrs = [_d(a) for a in (Q, b, A, b_eq, mu, lb, ub)]
    z = np.zeros((B, n))
    it = np.zeros(B, dtype=np.int32)
    det = (Details * B)()
    secs = L.fccqp_oracle_solve_batch(
        n, m, nc, s, int(options["max_iter"]), float(options["rho"]), float(options["eps_fcone"]),
        float(options["eps_bound"]), B, *[p for _, p in arrs], z.ctypes.data_as(_dp),
        it.ctypes.data_as(_ip), det, int(nthreads))
    return z, it, det, secs


# thin functional wrappers over the restated constraint_utils.cpp ----------------------------
def project_cone3(f, mu):
    fa, fp = _d(f); out = np.zeros(3)
    lib().fccqp_oracle_project_cone3(fp, float(mu), out.ctypes.data_as(_dp))
    return out


def project_cones(f, mus):
    fa, fp = _d(f); ma, mp = _d(mus); out = np.zeros(fa.size)
    rc = lib().fccqp_oracle_project_cones(fp, int(fa.size), mp, int(ma.size), out.ctypes.data_as(_dp))
    if rc != 0:
        raise IndexError("friction_coeffs too short")
    return out


def project_bounds(x, lb, ub):
    xa, xp = _d(x); la, lp = _d(lb); ua, up = _d(ub); out = np.zeros(xa.size)
    lib().fccqp_oracle_project_bounds(xp, lp, up, int(xa.size), out.ctypes.data_as(_dp))
    return out


def cone_violation(f, mus):
    fa, fp = _d(f); ma, mp = _d(mus); v = C.c_double()
    rc = lib().fccqp_oracle_cone_violation(fp, int(fa.size), mp, int(ma.size), C.byref(v))
    if rc != 0:
        raise IndexError("friction_coeffs too short")
    return v.value


def bound_violation(x, lb, ub):
    xa, xp = _d(x); la, lp = _d(lb); ua, up = _d(ub)
    return lib().fccqp_oracle_bound_violation(xp, lp, up, int(xa.size))


def validate_bounds(lb, ub):
    la, lp = _d(lb); ua, up = _d(ub)
    return bool(lib().fccqp_oracle_validate_bounds(lp, up, int(la.size)))


def ldlt(a):
    """Eigen-style LDLT of a symmetric matrix (lower triangle read).  Returns (LD, transp, info)."""
    a = np.asarray(a, dtype=np.float64)
    N = a.shape[0]
    w = np.asfortranarray(a.copy())
    tr = np.zeros(N + 1, dtype=np.int32)
    tmp = np.zeros(N + 1)
    info = lib().fccqp_oracle_ldlt_compute(w.ctypes.data_as(_dp), N, tr.ctypes.data_as(_ip),
                                           tmp.ctypes.data_as(_dp))
    return w, tr[:N], info


def ldlt_solve(LD, transp, rhs):
    x = np.array(rhs, dtype=np.float64).copy()
    tr = np.ascontiguousarray(np.concatenate([transp, [0]]).astype(np.int32))
    w = np.asfortranarray(LD)
    lib().fccqp_oracle_ldlt_solve(w.ctypes.data_as(_dp), w.shape[0], tr.ctypes.data_as(_ip),
                                  x.ctypes.data_as(_dp))
    return x


def cod_solve(a, rhs):
    """Minimum-norm least-squares solve (Eigen COD semantics).  Returns (x, rank)."""
    w = np.asfortranarray(np.array(a, dtype=np.float64).copy())
    r, rp = _d(rhs)
    x = np.zeros(w.shape[0])
    rank = lib().fccqp_oracle_cod_solve(w.ctypes.data_as(_dp), w.shape[0], rp, x.ctypes.data_as(_dp))
    return x, rank
