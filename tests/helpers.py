"""Shared helpers for the parity tests: run a batch through the oracle and through the C ABI."""
from __future__ import annotations

import numpy as np

# Stated fp64 parity tolerance of the HIP path against the CPU oracle (BASELINE.md section 3,
# SURVEY.md 8c):   ||z_gpu - z_cpu||_inf <= Z_TOL * (1 + ||z_cpu||_inf)   and   |dn_iter| <= 1.
# The kernels factor the KKT matrix in natural order instead of Eigen's largest-diagonal order,
# so agreement is at rounding level (observed ~1e-12), not bitwise.
Z_TOL = 1e-6
# what we actually observe and additionally enforce on the synthetic workloads
Z_TOL_TIGHT = 1e-9


def oracle_batch(O, ctor_args, options, batch, nthreads=8):
    z, it, det, _ = O.solve_batch(ctor_args, options, *batch.arrays(), nthreads=nthreads)
    return z, it, det


def rel_err(z, z_ref):
    """per-instance  ||z - z_ref||_inf / (1 + ||z_ref||_inf)"""
    return np.max(np.abs(z - z_ref), axis=1) / (1.0 + np.max(np.abs(z_ref), axis=1))


def assert_parity(z, it, z_ref, it_ref, tol=Z_TOL, max_iter=None, what=""):
    err = rel_err(z, z_ref)
    worst = int(np.argmax(err))
    assert np.all(np.isfinite(z)), f"{what}: non-finite z"
    assert err[worst] <= tol, f"{what}: instance {worst} rel err {err[worst]:.3e} > {tol:g}"
    dn = np.abs(np.asarray(it, dtype=np.int64) - np.asarray(it_ref, dtype=np.int64))
    assert dn.max() <= 1, f"{what}: n_iter differs by {dn.max()} at instance {int(np.argmax(dn))}"
    return float(err.max()), int(dn.max())
