"""Generic-kernel timing for the large (n=114) and walking shapes, split by max_iter so the
per-QP cost of presolve+factorizations (max_iter=0) and of one ADMM iteration can be read off."""
import sys, numpy as np
sys.path.insert(0, ".")
from fcc_qp_amd import problems as P
from tests import capi as C
for shape, name, nb in [(P.LARGE, "large", 4096), (P.WALKING, "walking-generic", 4096)]:
    B = P.workload("W-sat", nb, shape, distinct=64)
    s = C.Solver(*shape.ctor_args, batch=nb); s.set_kernel(C.KERNEL_GENERIC)
    s.phase_seconds()            # enables the device phase stamps
    for it in (0, 16, 17, 58, 100):
        opt = dict(P.HARNESS_OPTIONS); opt["max_iter"] = it
        s.set_options(**opt)
        for rep in range(2):
            s.solve_ok(*B.arrays()); ms = s.kernel_ms()
        z, d = s.solution()
        print(name, "max_iter", it, "kernel ms %.3f" % ms, "-> %.0f solves/s" % (nb / ms * 1e3),
              "; mean n_iter", np.mean([x.n_iter for x in d]),
              "; factorisations %.1f us/QP" % (1e6 * np.mean([x.factorization_time for x in d])),
              "; phases us [to end of assembly, factorisations, presolve in build, operator build, iterations, total]:",
              ["%.0f" % (1e6 * v) for v in s.phase_seconds()[:7]], flush=True)
