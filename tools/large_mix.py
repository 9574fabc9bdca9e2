"""Large-shape generic kernel on the mixed workload / paper options (effect of the G switch point)."""
import sys, numpy as np
sys.path.insert(0, ".")
from fcc_qp_amd import problems as P
from tests import capi as C
nb = 4096
for wl, opts in (("W-mix", P.HARNESS_OPTIONS), ("W-sat", P.PAPER_OPTIONS), ("W-mix", P.PAPER_OPTIONS)):
    B = P.workload(wl, nb, P.LARGE, distinct=64)
    s = C.Solver(*P.LARGE.ctor_args, batch=nb); s.set_options(**opts)
    for rep in range(2):
        s.solve_ok(*B.arrays()); ms = s.kernel_ms()
    z, d = s.solution()
    it = np.array([x.n_iter for x in d])
    print(f"large {wl} max_iter={opts['max_iter']}: kernel {ms:.3f} ms -> {nb/ms*1e3:.0f} solves/s; n_iter mean {it.mean():.1f} "
          f"hist {dict(zip(*np.unique(it, return_counts=True)))}", flush=True)
