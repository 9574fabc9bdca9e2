"""Run-time shape specialisation without a GPU: hiprtc instantiates the one-wave / cooperative kernel template for a
shape that is not compiled into the library (fcc_qp_amd/csrc/fccqp_jit.cpp) -- the compile, the disk cache and the
answers of the shape queries need no device.  The reference takes its sizes as constructor arguments at run time
(/root/reference/src/fcc_qp.cpp:24-55); the GPU side of the same feature is tests/test_gpu_jit.py."""
import glob
import os
import time

import pytest

from tests import capi as C
from tests import codeobj


@pytest.fixture
def jit_on():
    L = C.lib()
    L.fccqp_set_jit(1)
    yield L
    L.fccqp_set_jit(0)


def test_shape_queries_with_and_without_the_jit(jit_on):
    L = jit_on
    assert L.fccqp_kernel_for_shape(60, 38, 12, 38) == C.KERNEL_FUSED and L.fccqp_kernel_origin_for_shape(60, 38, 12, 38) == C.ORIGIN_BUILTIN
    assert L.fccqp_kernel_for_shape(114, 60, 24, 66) == C.KERNEL_COOP and L.fccqp_kernel_origin_for_shape(114, 60, 24, 66) == C.ORIGIN_BUILTIN
    # unlisted shapes: one wave for num_vars <= 63, the cooperative kernel beyond, neither for nc == 0 or num_vars > 128
    assert L.fccqp_kernel_for_shape(61, 38, 12, 38) == C.KERNEL_FUSED and L.fccqp_kernel_origin_for_shape(61, 38, 12, 38) == C.ORIGIN_JIT
    assert L.fccqp_kernel_for_shape(100, 50, 18, 60) == C.KERNEL_COOP and L.fccqp_kernel_origin_for_shape(100, 50, 18, 60) == C.ORIGIN_JIT
    assert L.fccqp_kernel_for_shape(20, 5, 0, 0) == C.KERNEL_GENERIC and L.fccqp_kernel_origin_for_shape(20, 5, 0, 0) == C.ORIGIN_GENERIC_ONLY
    assert L.fccqp_kernel_for_shape(148, 74, 36, 82) == C.KERNEL_GENERIC and L.fccqp_kernel_origin_for_shape(148, 74, 36, 82) == C.ORIGIN_GENERIC_ONLY
    assert L.fccqp_kernel_for_shape(100, 50, 27, 60) == C.KERNEL_GENERIC          # nine contacts above 64 variables
    assert L.fccqp_kernel_origin_for_shape(10, 4, 4, 0) == -C.ERR_SHAPE
    L.fccqp_set_jit(0)
    assert L.fccqp_kernel_for_shape(61, 38, 12, 38) == C.KERNEL_GENERIC and L.fccqp_kernel_origin_for_shape(61, 38, 12, 38) == C.ORIGIN_GENERIC_ONLY
    assert L.fccqp_kernel_origin_for_shape(60, 38, 12, 38) == C.ORIGIN_BUILTIN


def test_prepare_compiles_an_unlisted_shape_into_the_disk_cache(jit_on):
    L = jit_on
    cache = os.environ["FCCQP_JIT_CACHE_DIR"]
    dims = (44, 30, 6, 30)
    for f in glob.glob(os.path.join(cache, "fused_44_30_6_30_*")):
        os.remove(f)
    L.fccqp_debug_jit_forget()
    t0 = time.perf_counter()
    assert L.fccqp_jit_prepare(*dims) == C.KERNEL_FUSED
    t_compile = time.perf_counter() - t0
    cos = glob.glob(os.path.join(cache, "fused_44_30_6_30_w1_f64_gfx950_*.co"))
    assert len(cos) == 1 and os.path.getsize(cos[0]) > 10000
    # the code object is a gfx950 kernel of the same template as the built-in shapes: no scratch, the MFMA tile code
    meta = codeobj.kernel_metadata(cos[0])
    (name, rec), = meta.items()
    assert open(cos[0][:-3] + ".sym").read() == name and "fused_kernel" in name and "Li44ELi30ELi6ELi30E" in name
    assert rec["private_segment_fixed_size"] == 0 and rec["group_segment_fixed_size"] <= 40 * 1024
    ops = [op for body in codeobj.disassemble(cos[0]).values() for op, _ in body]
    assert sum(1 for op in ops if op.startswith("v_mfma_f64_16x16x4")) > 100
    # a second request is answered from memory, a request after forgetting from the file: no compiler either way
    t0 = time.perf_counter(); assert L.fccqp_jit_prepare(*dims) == C.KERNEL_FUSED; t_mem = time.perf_counter() - t0
    L.fccqp_debug_jit_forget()
    t0 = time.perf_counter(); assert L.fccqp_jit_prepare(*dims) == C.KERNEL_FUSED; t_disk = time.perf_counter() - t0
    assert t_mem < 0.05 and t_disk < 0.1 < t_compile, (t_compile, t_mem, t_disk)
    # a listed shape needs nothing; a shape without a specialised kernel says so
    assert L.fccqp_jit_prepare(60, 38, 12, 38) == C.KERNEL_FUSED
    assert L.fccqp_jit_prepare(20, 5, 0, 0) == -C.ERR_UNSUPPORTED
    assert L.fccqp_jit_prepare(10, 4, 4, 0) == -C.ERR_SHAPE


def test_prepare_reports_a_shape_the_template_refuses(jit_on, capfd):
    """num_vars = 63 with m = 64 passes the host-side screen but the kernel's LDS carve does not fit a quarter of a CU's
    LDS... or it compiles: either way prepare answers with a kernel id or -FCCQP_ERR_UNSUPPORTED, never crashes."""
    L = jit_on
    rc = L.fccqp_jit_prepare(63, 64, 48, 15)
    assert rc in (C.KERNEL_FUSED, C.KERNEL_COOP, -C.ERR_UNSUPPORTED)
