"""The C-ABI shared library (fcc_qp_amd/libfccqp_hip.so): loads, exports every function declared in
include/fccqp.h, and fails loudly -- never silently on a CPU path -- when there is no GPU."""
import ctypes
import os

import numpy as np
import pytest

from tests import capi as C


def test_library_is_built_in_tree():
    assert os.path.exists(C.LIB_PATH), "run `python fcc_qp_amd/build.py` (or __graft_entry__.build())"


def test_exports_every_declared_function():
    names = C.declared_functions()
    assert len(names) >= 20 and "fccqp_solve" in names and "fccqp_solve_device" in names
    lib = ctypes.CDLL(C.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/fccqp.h but not exported: {missing}"


def test_version_and_strerror():
    L = C.lib()
    assert L.fccqp_version() == 1
    assert L.fccqp_strerror(C.OK) == b"ok"
    assert b"no HIP device" in L.fccqp_strerror(C.ERR_NO_DEVICE)
    assert b"CPU" in L.fccqp_strerror(C.ERR_NO_DEVICE)


def test_default_options_match_reference():
    """/root/reference/src/fcc_qp.hpp:30-35"""
    o = C.Options()
    assert C.lib().fccqp_default_options(ctypes.byref(o)) == C.OK
    assert (o.max_iter, o.rho, o.eps_fcone, o.eps_bound) == (1000, 1e-6, 1e-3, 1e-6)


def test_create_validates_sizes_before_touching_the_device():
    """/root/reference/src/fcc_qp.cpp:29-33 asserts become FCCQP_ERR_SHAPE"""
    L = C.lib(); h = ctypes.c_void_p()
    assert L.fccqp_create(60, 38, 13, 38, 1, 0, ctypes.byref(h)) == C.ERR_SHAPE
    assert L.fccqp_create(60, 38, 12, 50, 1, 0, ctypes.byref(h)) == C.ERR_SHAPE
    assert L.fccqp_create(-1, 0, 0, 0, 1, 0, ctypes.byref(h)) == C.ERR_SHAPE
    assert L.fccqp_create(60, 38, 12, 38, 0, 0, ctypes.byref(h)) == C.ERR_SHAPE
    assert L.fccqp_create(60, 38, 12, 38, 1, 0, None) == C.ERR_INVALID_ARG


def test_no_device_is_a_loud_error_not_a_fallback():
    L = C.lib()
    if L.fccqp_device_count() > 0:
        pytest.skip("a GPU is present: covered by the -m gpu tests")
    h = ctypes.c_void_p()
    assert L.fccqp_create(60, 38, 12, 38, 4, 0, ctypes.byref(h)) == C.ERR_NO_DEVICE
    assert not h
    with pytest.raises(RuntimeError, match="no HIP device"):
        C.Solver(60, 38, 12, 38)


def test_null_handles_are_rejected():
    L = C.lib()
    o = C.Options(1, 1.0, 1.0, 1.0)
    assert L.fccqp_set_options(None, ctypes.byref(o)) == C.ERR_INVALID_ARG
    assert L.fccqp_set_warm_start(None, 1) == C.ERR_INVALID_ARG
    assert L.fccqp_get_solution(None, None, None) == C.ERR_INVALID_ARG
    assert L.fccqp_destroy(None) == C.OK


def test_python_surface_names_match_reference_binding():
    """/root/reference/src/main.cpp:22-54: class, attribute, method and keyword names."""
    import fcc_qp_amd as F
    import fcc_qp
    assert fcc_qp.FCCQP is F.FCCQP and fcc_qp.FCCQPOptions is F.FCCQPOptions and fcc_qp.FCCQPSolution is F.FCCQPSolution
    o = F.FCCQPOptions()
    assert (o.max_iter, o.rho, o.eps_fcone, o.eps_bound) == (1000, 1e-6, 1e-3, 1e-6)
    o.max_iter = 7; o.rho = 2.0
    assert (o.max_iter, o.rho) == (7, 2.0)
    for attr in ("n_iter", "eps_bounds", "eps_friction_cone", "bounds_viol", "friction_cone_viol",
                 "solve_time", "factorization_time"):
        assert hasattr(F.FCCQPDetails, attr)
    with pytest.raises(TypeError):
        F.FCCQPDetails()                      # no constructor in the reference binding either
    for meth in ("set_rho", "set_max_iter", "set_warm_start", "set_options", "Solve", "GetSolution"):
        assert hasattr(F.FCCQP, meth)
    doc = F.FCCQP.Solve.__doc__
    for kw in ("Q", "b", "A_eq", "b_eq", "friction_coeffs", "lb", "ub"):
        assert kw in doc
    init_doc = F.FCCQP.__init__.__doc__
    for kw in ("num_vars", "num_equality_constraints", "nc", "lambda_c_start"):
        assert kw in init_doc
    if F.device_count() == 0:
        with pytest.raises(F.FCCQPError, match="no HIP device"):
            F.FCCQP(60, 38, 12, 38)
        with pytest.raises(ValueError):
            F.FCCQP(60, 38, 13, 38)


def test_kernel_for_shape_needs_no_device():
    """fccqp_kernel_for_shape: what FCCQP_KERNEL_AUTO takes for a shape in this build -- the shapes of fused_shapes.def /
    coop_shapes.def get their kernels, any other valid shape the generic one, invalid sizes a negative status."""
    L = C.lib()
    assert L.fccqp_kernel_for_shape(60, 38, 12, 38) == C.KERNEL_FUSED          # reference harness shape
    assert L.fccqp_kernel_for_shape(114, 60, 24, 66) == C.KERNEL_COOP          # large shape
    assert L.fccqp_kernel_for_shape(61, 38, 12, 38) == C.KERNEL_GENERIC        # (run-time compilation is off in the suite:
    assert L.fccqp_kernel_for_shape(20, 5, 0, 0) == C.KERNEL_GENERIC           #  tests/conftest.py; tests/test_jit_cpu.py)
    assert L.fccqp_kernel_for_shape(10, 4, 4, 0) == -C.ERR_SHAPE               # nc % 3
    assert L.fccqp_kernel_for_shape(10, 4, 6, 6) == -C.ERR_SHAPE               # lambda_c_start + nc > num_vars
    assert L.fccqp_kernel_for_shape(400, 113, 12, 100) == -C.ERR_UNSUPPORTED   # n + m > 512
    import fcc_qp_amd as F
    assert F.kernel_for_shape(60, 38, 12, 38) == F.KERNEL_FUSED and F.kernel_for_shape(7, 2, 3, 4) == F.KERNEL_GENERIC
    with pytest.raises(ValueError):
        F.kernel_for_shape(10, 4, 4, 0)
