"""Zero-copy batched solve from torch tensors already resident in HBM.

PyTorch is plumbing here (device memory + streams): the tensors' device pointers are handed to the
C ABI entry ``fccqp_solve_device`` (include/fccqp.h) and the kernel is enqueued on torch's current
stream, so ``torch.cuda.Event`` timing and stream ordering see it.
"""
from __future__ import annotations

from typing import Sequence


def solve_tensors(solver, Q, b, A_eq, b_eq, friction_coeffs, lb, ub, stream=None):
    """Enqueue ``solver`` (an ``FCCQPBatch``) on device tensors; returns immediately.

    All seven tensors must share one dtype -- float64, or float32 for the fp32-storage path
    (``fccqp_solve_device_f32``: widened on load, arithmetic in fp64) -- and be contiguous, on the
    solver's GPU and instance-major
    (``Q[B,n,n] b[B,n] A_eq[B,m,n] b_eq[B,m] friction_coeffs[B,nc/3] lb[B,n] ub[B,n]``).
    """
    import torch

    tensors: Sequence = (Q, b, A_eq, b_eq, friction_coeffs, lb, ub)
    dtype = Q.dtype
    if dtype not in (torch.float64, torch.float32):
        raise ValueError("solve_tensors needs float64 or float32 tensors")
    for t in tensors:
        if not (t.is_cuda and t.dtype == dtype and t.is_contiguous()):
            raise ValueError("solve_tensors needs contiguous tensors of one dtype on the GPU")
    if stream is None:
        stream = torch.cuda.current_stream(Q.device)
    solver.SolveDevice(*(int(t.data_ptr()) for t in tensors), int(stream.cuda_stream),
                       dtype == torch.float32)
