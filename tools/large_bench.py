import sys, numpy as np
sys.path.insert(0, ".")
from fcc_qp_amd import problems as P
from tests import capi as C
for shape, name, nb in [(P.LARGE, "large", 4096), (P.WALKING, "walking-generic", 4096)]:
    B = P.workload("W-sat", nb, shape, distinct=64)
    s = C.Solver(*shape.ctor_args, batch=nb); s.set_options(**P.HARNESS_OPTIONS); s.set_kernel(C.KERNEL_GENERIC)
    for rep in range(2):
        s.solve_ok(*B.arrays()); ms = s.kernel_ms()
    z, d = s.solution()
    print(name, "kernel ms", ms, "->", nb / ms * 1e3, "solves/s; mean n_iter", np.mean([x.n_iter for x in d]), flush=True)
