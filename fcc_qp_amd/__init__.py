"""fcc_qp_amd -- MI355X-native batched ADMM friction-cone QP solver (drop-in for fcc_qp).

Python surface of the reference (``/root/reference/src/main.cpp:19-56``, re-exported by
``/root/reference/fcc_qp/__init__.py:1``): ``FCCQP, FCCQPOptions, FCCQPSolution, FCCQPDetails``
with identical attribute, method and keyword names -- plus ``FCCQPBatch`` for solving thousands of
independent QPs in one kernel launch.  Everything numeric runs in hand-written HIP kernels for
gfx950 behind the C ABI of ``include/fccqp.h``; there is NO CPU fallback: if the extension is not
built, importing this package raises, and without a GPU constructing a solver raises.
"""
from __future__ import annotations

import os as _os

_HERE = _os.path.dirname(_os.path.abspath(__file__))

# PyTorch-ROCm wheels bundle their own libamdhip64.so.7 (same SONAME as /opt/rocm's).  Whichever
# copy is loaded first serves the whole process; torch refuses to see any GPU when the system copy
# got there first.  So if torch is installed, let it load its runtime before our library does --
# libfccqp_hip.so then binds to that same runtime and device pointers / streams are shared.
try:  # pragma: no cover - depends on the environment
    import torch as _torch  # noqa: F401
except Exception:  # torch is optional plumbing, not a dependency of the solver
    _torch = None

try:
    from .fcc_qp_solver import (  # noqa: F401
        FCCQP, FCCQPBatch, FCCQPBatchSolution, FCCQPDetails, FCCQPError, FCCQPOptions, FCCQPSharded, FCCQPSolution,
        device_count, kernel_for_shape, KERNEL_AUTO, KERNEL_COOP, KERNEL_FUSED, KERNEL_FUSED_SPLIT, KERNEL_COOP_SPLIT, KERNEL_GENERIC, FLAG_ADMM_NUMERICAL,
        FLAG_EQ_RESIDUAL, FLAG_EQUALITY_ONLY, FLAG_PRESOLVE_CG, FLAG_PRESOLVE_COD, FLAG_PRESOLVE_NUMERICAL, FLAG_PRESOLVE_RAN)
except ImportError as exc:  # fail loudly: the HIP extension IS the product
    raise ImportError(
        "fcc_qp_amd: the native extension (fcc_qp_amd/libfccqp_hip.so + fcc_qp_solver*.so) is not "
        "built or cannot be loaded. Run `python fcc_qp_amd/build.py` or __graft_entry__.build() (needs hipcc, gfx950). "
        f"There is no CPU fallback. Original error: {exc}") from exc

from . import problems  # noqa: E402,F401
from .batch import result_tensors, solve_tensors, solve_tensors_sharded  # noqa: E402,F401

LIB_PATH = _os.path.join(_HERE, "libfccqp_hip.so")

__all__ = ["FCCQP", "FCCQPBatch", "FCCQPBatchSolution", "FCCQPDetails", "FCCQPError", "FCCQPOptions",
           "FCCQPSharded", "FCCQPSolution", "device_count", "kernel_for_shape", "problems", "solve_tensors", "solve_tensors_sharded", "result_tensors", "LIB_PATH"]
