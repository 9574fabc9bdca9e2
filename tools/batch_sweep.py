"""Kernel time of the fused kernel against the batch size, for a saturating and an early-exit
workload: separates per-instance compute (rounds of 1024 resident workgroups) from dispatch effects."""
import sys, numpy as np
sys.path.insert(0, ".")
from fcc_qp_amd import problems as P
from tests import capi as C
for wl in ("W-sat", "W-easy"):
    base = P.workload(wl, 256, P.WALKING)
    for nb in (256, 512, 1024, 1536, 2048, 4096, 8192, 16384):
        reps = nb // 256
        B = P.Batch(*(np.ascontiguousarray(np.tile(a, (reps,) + (1,) * (a.ndim - 1))) for a in base.arrays()))
        s = C.Solver(*P.WALKING.ctor_args, batch=nb); s.set_options(**P.HARNESS_OPTIONS)
        ms = []
        for rep in range(4):
            s.solve_ok(*B.arrays()); ms.append(s.kernel_ms())
        print(f"{wl} batch {nb:6d}: kernel {min(ms):.3f} ms  -> {nb / min(ms) * 1e3:,.0f} solves/s, {min(ms) * 1e3 / nb * 1024:.1f} us per 1024", flush=True)
        s.close()
