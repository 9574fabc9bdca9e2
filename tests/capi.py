"""ctypes binding of the product C ABI (include/fccqp.h) used by the parity tests, so that the
`-m gpu` tests call the HIP path through exactly the entry points a reference maintainer would bind."""
from __future__ import annotations

import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("FCCQP_LIB_PATH") or os.path.join(ROOT, "fcc_qp_amd", "libfccqp_hip.so")   # override: debug builds
HEADER = os.path.join(ROOT, "include", "fccqp.h")

OK, ERR_INVALID_ARG, ERR_SHAPE, ERR_NO_DEVICE, ERR_HIP, ERR_ALLOC, ERR_FRICTION, ERR_NOT_SOLVED, ERR_UNSUPPORTED = range(9)
KERNEL_AUTO, KERNEL_GENERIC, KERNEL_FUSED, KERNEL_FUSED_SPLIT, KERNEL_COOP, KERNEL_COOP_SPLIT = 0, 1, 2, 3, 4, 5
FLAG_PRESOLVE_RAN, FLAG_PRESOLVE_NUMERICAL, FLAG_ADMM_NUMERICAL, FLAG_EQUALITY_ONLY, FLAG_PRESOLVE_COD, FLAG_EQ_RESIDUAL, FLAG_PRESOLVE_CG = 1, 2, 4, 8, 16, 32, 64
TIMING_NONE, TIMING_DEVICE_EVENTS, TIMING_HOST_CLOCK = 0, 1, 2
ORIGIN_BUILTIN, ORIGIN_JIT, ORIGIN_GENERIC_ONLY = 0, 1, 2
ORDER_HASH, ORDER_PREVIOUS_ITERATIONS = 0, 1


class Options(C.Structure):
    _fields_ = [("max_iter", C.c_int), ("rho", C.c_double), ("eps_fcone", C.c_double), ("eps_bound", C.c_double)]


class Details(C.Structure):
    _fields_ = [("n_iter", C.c_int), ("admm_residual_bounds", C.c_double),
                ("admm_residual_friction_cone", C.c_double), ("solve_time", C.c_double),
                ("factorization_time", C.c_double), ("bounds_viol", C.c_double),
                ("friction_cone_viol", C.c_double), ("solve_status", C.c_int), ("flags", C.c_int)]


def declared_functions():
    """Names of every function declared in include/fccqp.h."""
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fccqp_[a-z_0-9]+)\s*\(", text)))


_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_lib = None


def lib():
    global _lib
    if _lib is None:
        try:
            import torch  # noqa: F401  (load torch's bundled HIP runtime first: see fcc_qp_amd/__init__.py)
        except Exception:
            pass
        L = C.CDLL(LIB_PATH)
        L.fccqp_strerror.restype = C.c_char_p
        L.fccqp_last_error.restype = C.c_char_p
        L.fccqp_last_error.argtypes = [C.c_void_p]
        L.fccqp_create.argtypes = [C.c_int] * 6 + [C.POINTER(C.c_void_p)]
        L.fccqp_destroy.argtypes = [C.c_void_p]
        L.fccqp_set_options.argtypes = [C.c_void_p, C.POINTER(Options)]
        L.fccqp_get_options.argtypes = [C.c_void_p, C.POINTER(Options)]
        L.fccqp_default_options.argtypes = [C.POINTER(Options)]
        L.fccqp_set_warm_start.argtypes = [C.c_void_p, C.c_int]
        L.fccqp_set_kernel.argtypes = [C.c_void_p, C.c_int]
        L.fccqp_set_presolve_fallback.argtypes = [C.c_void_p, C.c_int]
        L.fccqp_set_dispatch_order.argtypes = [C.c_void_p, C.c_int]
        L.fccqp_set_kernel_timing.argtypes = [C.c_void_p, C.c_int]
        if hasattr(L, "fccqp_set_self_check"):          # (older builds under FCCQP_LIB_PATH, tools/ab_bench.py)
            L.fccqp_set_self_check.argtypes = [C.c_void_p, C.c_int]
            L.fccqp_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]
        L.fccqp_get_kernel.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        if hasattr(L, "fccqp_kernel_for_shape"):
            L.fccqp_kernel_for_shape.argtypes = [C.c_int] * 4
        L.fccqp_solve.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp, C.c_int, _dp, _dp]
        L.fccqp_solve_device.argtypes = [C.c_void_p] + [C.c_void_p] * 7 + [C.c_void_p]
        L.fccqp_solve_f32.argtypes = [C.c_void_p, _fp, _fp, _fp, _fp, _fp, C.c_int, _fp, _fp]
        L.fccqp_solve_device_f32.argtypes = [C.c_void_p] + [C.c_void_p] * 7 + [C.c_void_p]
        L.fccqp_synchronize.argtypes = [C.c_void_p]
        L.fccqp_get_solution.argtypes = [C.c_void_p, _dp, C.POINTER(Details)]
        L.fccqp_get_state.argtypes = [C.c_void_p, _dp, _dp, _dp]
        L.fccqp_set_state.argtypes = [C.c_void_p, _dp, _dp, _dp]
        L.fccqp_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.fccqp_algorithmic_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
        L.fccqp_kernel_origin_for_shape.argtypes = [C.c_int] * 4
        L.fccqp_jit_prepare.argtypes = [C.c_int] * 4
        L.fccqp_get_kernel_origin.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.fccqp_set_jit.argtypes = [C.c_int]
        L.fccqp_phase_seconds.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.fccqp_debug_phase_ticks.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.c_longlong]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(_dp)


class Solver:
    """Thin RAII wrapper; raises RuntimeError with the library's message on any non-zero status."""

    def __init__(self, n, m, nc, s, batch=1, device=0):
        self.L = lib()
        self.h = C.c_void_p()
        rc = self.L.fccqp_create(n, m, nc, s, batch, device, C.byref(self.h))
        if rc != OK:
            raise RuntimeError(f"fccqp_create: {rc} {self.L.fccqp_strerror(rc).decode()}")
        self.n, self.m, self.nc, self.s, self.batch = n, m, nc, s, batch

    def close(self):
        h, self.h = self.h, None
        if h:
            self.L.fccqp_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:      # interpreter shutdown: module globals may already be gone
            pass

    def _ck(self, rc, what):
        if rc != OK:
            raise RuntimeError(f"{what}: {rc} {self.L.fccqp_strerror(rc).decode()} "
                               f"[{self.L.fccqp_last_error(self.h).decode()}]")

    def set_options(self, max_iter, rho, eps_fcone, eps_bound):
        o = Options(int(max_iter), float(rho), float(eps_fcone), float(eps_bound))
        self._ck(self.L.fccqp_set_options(self.h, C.byref(o)), "set_options")

    def set_warm_start(self, w):
        self._ck(self.L.fccqp_set_warm_start(self.h, int(bool(w))), "set_warm_start")

    def set_kernel(self, k):
        self._ck(self.L.fccqp_set_kernel(self.h, int(k)), "set_kernel")

    def set_dispatch_order(self, order):
        self._ck(self.L.fccqp_set_dispatch_order(self.h, int(order)), "set_dispatch_order")

    def set_presolve_fallback(self, on):
        self._ck(self.L.fccqp_set_presolve_fallback(self.h, int(bool(on))), "set_presolve_fallback")

    def set_kernel_timing(self, on):
        self._ck(self.L.fccqp_set_kernel_timing(self.h, int(bool(on))), "set_kernel_timing")

    def kernel_origin(self):
        o = C.c_int()
        self._ck(self.L.fccqp_get_kernel_origin(self.h, C.byref(o)), "get_kernel_origin")
        return o.value

    def last_error(self):
        return self.L.fccqp_last_error(self.h).decode()

    def kernel(self):
        k = C.c_int()
        self._ck(self.L.fccqp_get_kernel(self.h, C.byref(k)), "get_kernel")
        return k.value

    def solve(self, Q, b, A, b_eq, mu, lb, ub):
        arrs = [np.ascontiguousarray(np.asarray(a, dtype=np.float64)) for a in (Q, b, A, b_eq, mu, lb, ub)]
        nf = arrs[4].shape[-1] if arrs[4].ndim >= 1 and arrs[4].size else 0
        rc = self.L.fccqp_solve(self.h, _p(arrs[0]), _p(arrs[1]), _p(arrs[2]), _p(arrs[3]), _p(arrs[4]),
                                int(nf), _p(arrs[5]), _p(arrs[6]))
        return rc

    def solve_ok(self, *a):
        self._ck(self.solve(*a), "solve")

    def solve_f32(self, Q, b, A, b_eq, mu, lb, ub):
        """fp32-storage entry point: every array is handed over as float32."""
        arrs = [np.ascontiguousarray(np.asarray(a, dtype=np.float32)) for a in (Q, b, A, b_eq, mu, lb, ub)]
        nf = arrs[4].shape[-1] if arrs[4].ndim >= 1 and arrs[4].size else 0
        return self.L.fccqp_solve_f32(self.h, *(a.ctypes.data_as(_fp) for a in arrs[:5]), int(nf),
                                      arrs[5].ctypes.data_as(_fp), arrs[6].ctypes.data_as(_fp))

    def solve_f32_ok(self, *a):
        self._ck(self.solve_f32(*a), "solve_f32")

    def solution(self):
        z = np.zeros((self.batch, self.n))
        d = (Details * self.batch)()
        self._ck(self.L.fccqp_get_solution(self.h, _p(z), d), "get_solution")
        return z, d

    def state(self):
        x = np.zeros((self.batch, self.n)); mx = np.zeros((self.batch, self.n))
        ml = np.zeros((self.batch, max(self.nc, 1)))
        self._ck(self.L.fccqp_get_state(self.h, _p(x), _p(mx), _p(ml)), "get_state")
        return x, mx, ml[:, : self.nc]

    def kernel_ms(self):
        ms = C.c_float()
        self._ck(self.L.fccqp_last_kernel_ms(self.h, C.byref(ms)), "last_kernel_ms")
        return ms.value

    def last_timing(self):
        ms, by = C.c_float(), C.c_int()
        self._ck(self.L.fccqp_last_timing(self.h, C.byref(ms), C.byref(by)), "last_timing")
        return ms.value, by.value

    def set_self_check(self, on):
        self._ck(self.L.fccqp_set_self_check(self.h, int(bool(on))), "set_self_check")

    def phase_seconds(self):
        out = (C.c_double * 8)()
        self._ck(self.L.fccqp_phase_seconds(self.h, out), "phase_seconds")
        return list(out)

    def phase_ticks(self):
        """[8, batch] raw 100 MHz stamps (see include/fccqp.h)"""
        import numpy as np
        out = np.zeros(8 * self.batch, dtype=np.int64)
        self._ck(self.L.fccqp_debug_phase_ticks(self.h, out.ctypes.data_as(C.POINTER(C.c_longlong)), out.size), "phase_ticks")
        return out.reshape(8, self.batch)

    def algorithmic_bytes(self):
        b = C.c_longlong()
        self._ck(self.L.fccqp_algorithmic_bytes(self.h, C.byref(b)), "algorithmic_bytes")
        return b.value
