"""What a host-pointer solve out of PAGEABLE memory costs since the GPU no longer addresses caller memory:
fccqp_solve + fccqp_get_solution of the headline batch (walking shape, 4096 instances, 202 MB in, 2 MB out) from numpy
arrays, against the same from page-locked arrays (fccqp_host_register).  Prints solves/s and GB/s.
    python tools/pageable_rate.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fcc_qp_amd import problems as P
from tests import capi as C

nb = 4096
B = P.workload("W-sat", nb, P.WALKING, distinct=256)
arrs = [np.ascontiguousarray(a) for a in B.arrays()]
nbytes = sum(a.nbytes for a in arrs)
s = C.Solver(*P.WALKING.ctor_args, batch=nb); s.set_options(**P.HARNESS_OPTIONS)
for name in ("pageable", "page-locked"):
    if name == "page-locked":
        for a in arrs:
            assert s.L.fccqp_host_register(a.ctypes.data_as(C.C.c_void_p), C.C.c_size_t(a.nbytes)) == 0
    for _ in range(2):
        s.solve_ok(*arrs)
    t = []
    for _ in range(5):
        t0 = time.perf_counter(); s.solve_ok(*arrs); t.append(time.perf_counter() - t0)
    z, d = s.solution()
    t0 = time.perf_counter(); z, d = s.solution(); tg = time.perf_counter() - t0
    m = min(t)
    print(f"{name}: fccqp_solve {1e3 * m:.2f} ms = {nb / m / 1e6:.3f} M solves/s, {nbytes / m / 1e9:.1f} GB/s in; fccqp_get_solution {1e3 * tg:.2f} ms", flush=True)
for a in arrs:
    s.L.fccqp_host_unregister(a.ctypes.data_as(C.C.c_void_p))
