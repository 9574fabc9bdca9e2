"""Zero-copy batched solve on data already resident in HBM.

PyTorch is plumbing here (device memory + streams): the arrays' device pointers are handed to the
C ABI entry ``fccqp_solve_device`` (include/fccqp.h) and the kernel is enqueued on torch's current
stream, so ``torch.cuda.Event`` timing and stream ordering see it.  Inputs may be torch tensors or
any object that exports ``__dlpack__`` (CuPy, JAX, Numba device arrays, another torch) or
``__cuda_array_interface__``; results come back as torch tensors that ALIAS the solver's device
buffers (no copy, no host round trip) and can be handed on through ``__dlpack__`` again.

The reference has no counterpart: its ``Solve`` takes host Eigen references and copies them into
Eigen members (/root/reference/src/fcc_qp.cpp:146-150; binding /root/reference/src/main.cpp:50-54).
"""
from __future__ import annotations

from typing import Dict, Sequence

_NAMES = ("Q", "b", "A_eq", "b_eq", "friction_coeffs", "lb", "ub")


def as_device_tensor(x, device=None):
    """``x`` as a torch tensor on the GPU without copying: torch tensors pass through, ``__dlpack__``
    and ``__cuda_array_interface__`` producers are imported zero-copy."""
    import torch

    if isinstance(x, torch.Tensor):
        return x
    if hasattr(x, "__dlpack__"):
        return torch.from_dlpack(x)
    if hasattr(x, "__cuda_array_interface__"):
        return torch.as_tensor(x, device=device if device is not None else "cuda")
    raise TypeError(f"solve_tensors: {type(x).__name__} is neither a torch tensor nor a __dlpack__ / "
                    "__cuda_array_interface__ producer")


def _expected_shapes(solver):
    B, n, m, nc = solver.batch, solver.num_vars, solver.num_equality_constraints, solver.nc
    return {"Q": (B, n, n), "b": (B, n), "A_eq": (B, m, n), "b_eq": (B, m),
            "friction_coeffs": (B, nc // 3), "lb": (B, n), "ub": (B, n)}


def solve_tensors(solver, Q, b, A_eq, b_eq, friction_coeffs, lb, ub, stream=None):
    """Enqueue ``solver`` (an ``FCCQPBatch``) on device arrays; returns immediately.

    All seven arrays must share one dtype -- float64, or float32 for the fp32-storage path
    (``fccqp_solve_device_f32``: widened on load, arithmetic in fp64) -- and be contiguous, on the
    solver's GPU and instance-major
    (``Q[B,n,n] b[B,n] A_eq[B,m,n] b_eq[B,m] friction_coeffs[B,nc/3] lb[B,n] ub[B,n]``).
    Shapes, dtype, contiguity and the device are checked here: the kernels read raw pointers.
    """
    import torch

    tensors: Sequence = tuple(as_device_tensor(t) for t in (Q, b, A_eq, b_eq, friction_coeffs, lb, ub))
    dtype = tensors[0].dtype
    if dtype not in (torch.float64, torch.float32):
        raise ValueError("solve_tensors needs float64 or float32 arrays")
    want = _expected_shapes(solver)
    dev = solver.device
    for name, t in zip(_NAMES, tensors):
        if not t.is_cuda:
            raise ValueError(f"solve_tensors: {name} is not on a GPU")
        if t.device.index != dev:
            raise ValueError(f"solve_tensors: {name} is on cuda:{t.device.index}, the solver on cuda:{dev}")
        if t.dtype != dtype:
            raise ValueError(f"solve_tensors: {name} has dtype {t.dtype}, Q has {dtype}")
        if tuple(t.shape) != want[name] and not (t.numel() == 0 and 0 in want[name]):
            raise ValueError(f"solve_tensors: {name} has shape {tuple(t.shape)}, expected {want[name]}")
        if not t.is_contiguous():
            raise ValueError(f"solve_tensors: {name} is not contiguous")
    if stream is None:
        stream = torch.cuda.current_stream(tensors[0].device)
    solver.SolveDevice(*(int(t.data_ptr()) for t in tensors), int(stream.cuda_stream),
                       dtype == torch.float32)


class _DeviceView:
    """A solver-owned device buffer seen through ``__cuda_array_interface__`` (version 2)."""

    def __init__(self, ptr: int, shape, typestr: str, owner):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}
        self._owner = owner      # keeps the solver (and its buffers) alive as long as the view is


def result_tensors(solver) -> Dict[str, "object"]:
    """Results of the last solve as torch tensors aliasing the solver's device buffers:

    ``z [B, n] f64`` · ``n_iter [B] i32`` · ``flags [B] i32`` · ``eps_bounds``, ``eps_friction_cone``,
    ``bounds_viol``, ``friction_cone_viol`` ``[B] f64`` (the fields of ``FCCQPDetails``,
    /root/reference/src/fcc_qp.hpp:19-28, python names of /root/reference/src/main.cpp:24-25).

    Nothing is copied and nothing is synchronised: the tensors are valid once the solve has
    finished on its stream (work enqueued on the same stream afterwards is ordered behind it) and
    until the next solve on this solver overwrites them; they keep the solver alive.  Each tensor
    exports ``__dlpack__`` for other frameworks.
    """
    import torch

    p = solver.device_pointers()
    B, n = solver.batch, solver.num_vars
    dev = f"cuda:{solver.device}"

    def view(ptr, shape, typestr):
        return torch.as_tensor(_DeviceView(ptr, shape, typestr, solver), device=dev)

    metrics = view(p["metrics"], (4, B), "<f8")
    return {
        "z": view(p["z"], (B, n), "<f8"),
        "n_iter": view(p["n_iter"], (B,), "<i4"),
        "flags": view(p["flags"], (B,), "<i4"),
        "eps_bounds": metrics[0], "eps_friction_cone": metrics[1],
        "bounds_viol": metrics[2], "friction_cone_viol": metrics[3],
    }


def solve_tensors_sharded(sharded, shard_tensors, streams=None):
    """Enqueue an ``FCCQPSharded`` on device-resident data: ``shard_tensors[g]`` = the seven float64 arrays of shard
    ``g`` -- the instances ``[sharded.shard_begin(g), sharded.shard_begin(g + 1))`` -- on ``cuda:sharded.device_of(g)``
    (torch tensors, ``__dlpack__`` or ``__cuda_array_interface__`` producers).  Returns immediately; ``sharded.Synchronize()``
    waits, ``sharded.GetSolution()`` gathers.  ``streams``: one torch stream per shard (default: each shard's own stream).
    One process driving several GPUs at their device-resident rate (the host-pointer ``Solve`` is bound by the PCIe copies)."""
    import torch

    G = sharded.shards
    if len(shard_tensors) != G:
        raise ValueError(f"solve_tensors_sharded: {len(shard_tensors)} tensor sets for {G} shards")
    n, m, nc = sharded.num_vars, sharded.num_equality_constraints, sharded.nc
    ptrs = []
    for g, tens in enumerate(shard_tensors):
        tens = tuple(as_device_tensor(t) for t in tens)
        if len(tens) != 7:
            raise ValueError("solve_tensors_sharded: seven arrays per shard (Q, b, A_eq, b_eq, friction_coeffs, lb, ub)")
        nb = (sharded.shard_begin(g + 1) if g + 1 < G else sharded.batch) - sharded.shard_begin(g)
        want = {"Q": (nb, n, n), "b": (nb, n), "A_eq": (nb, m, n), "b_eq": (nb, m), "friction_coeffs": (nb, nc // 3),
                "lb": (nb, n), "ub": (nb, n)}
        for name, t in zip(_NAMES, tens):
            if not t.is_cuda or t.device.index != sharded.device_of(g):
                raise ValueError(f"solve_tensors_sharded: shard {g} {name} is not on cuda:{sharded.device_of(g)}")
            if t.dtype != torch.float64 or not t.is_contiguous():
                raise ValueError(f"solve_tensors_sharded: shard {g} {name} must be a contiguous float64 array")
            if tuple(t.shape) != want[name] and not (t.numel() == 0 and 0 in want[name]):
                raise ValueError(f"solve_tensors_sharded: shard {g} {name} has shape {tuple(t.shape)}, expected {want[name]}")
        ptrs.append([int(t.data_ptr()) for t in tens])
    sharded.SolveDevice(ptrs, [] if streams is None else [int(s.cuda_stream) for s in streams])
