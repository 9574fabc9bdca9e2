"""The block -> instance map of a batch launch (block_instance, fcc_qp_amd/csrc/fccqp_device.h) is a bijection of
[0, batch) for every batch size -- checked on the CPU through fccqp_debug_block_instance, which runs the very function
the kernels call (pure integer arithmetic) -- and has no low-order structure left: the instances that land on one XCD
(blocks b = x mod 8) or on one CU slot cover the residue classes of the instance index evenly.  On the GPU the same
property shows as bitwise permutation equivariance (tests/test_gpu_parity.py::test_batch_4096_properties)."""
import ctypes

import numpy as np
import pytest

from tests import capi as C


def perm(batch):
    L = C.lib()
    L.fccqp_debug_block_instance.argtypes = [ctypes.c_int, ctypes.c_int]
    L.fccqp_debug_block_instance.restype = ctypes.c_int
    return np.array([L.fccqp_debug_block_instance(batch, b) for b in range(batch)])


@pytest.mark.parametrize("batch", [1, 2, 63, 64, 127, 128, 129, 200, 700, 1000, 1024, 1025, 4096, 5000, 8192, 65536, 100003])
def test_block_instance_is_a_bijection(batch):
    p = perm(batch)
    assert np.array_equal(np.sort(p), np.arange(batch))
    if batch < 128:
        assert np.array_equal(p, np.arange(batch))          # small launches are resident at once: identity
    L = C.lib()
    assert L.fccqp_debug_block_instance(batch, batch) == -1 and L.fccqp_debug_block_instance(batch, -1) == -1


def test_no_residue_class_is_pinned_to_an_xcd_or_a_cu():
    p = perm(4096)
    # blocks b = x (mod 8) share an XCD, blocks b = x (mod 64) a CU-sized slice of its dealing order: the instances they
    # get must cover the residue classes of the instance index modulo 2 .. 16 like a random sample would (within four
    # standard deviations); with instance = blockIdx.x every such class is pinned to one value
    for stride in (8, 64):
        for period in (2, 4, 8, 16):
            n = 4096 // stride
            expect = n / period
            if expect < 16:
                continue
            for x in range(stride):
                counts = np.bincount(p[x::stride] % period, minlength=period)
                assert np.all(np.abs(counts - expect) <= 4.0 * np.sqrt(expect) + 1), (stride, period, x, counts)
    ident = np.arange(4096)
    assert len(set(ident[1::8] % 8)) == 1                   # (the identity map pins them: what round 2 shipped)
