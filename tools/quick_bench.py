"""Quick GPU check used while developing: parity of both kernels against the oracle on a sample and
kernel time at batch 4096 (hipEvent time reported by the C ABI)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from fcc_qp_amd import problems as P
from tests import capi as C
from oracle import oracle as O

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
wl = sys.argv[2] if len(sys.argv) > 2 else "W-sat"
B = P.workload(wl, nb, P.WALKING, distinct=256)
zr, itr, _, _ = O.solve_batch(P.WALKING.ctor_args, P.HARNESS_OPTIONS, *B.slice(0, 256).arrays(), nthreads=8)
for kern, name in [(C.KERNEL_GENERIC, "generic"), (C.KERNEL_FUSED, "fused")]:
    s = C.Solver(*P.WALKING.ctor_args, batch=nb)
    s.set_options(**P.HARNESS_OPTIONS); s.set_kernel(kern)
    ms = []
    s.phase_seconds()
    for rep in range(4):
        s.solve_ok(*B.arrays()); ms.append(s.kernel_ms())
    z, d = s.solution()
    it = np.array([x.n_iter for x in d])
    err = np.max(np.abs(z[:256] - zr), axis=1) / (1 + np.max(np.abs(zr), axis=1))
    print(f"{name:8s} batch={nb} {wl}: kernel ms {['%.3f' % m for m in ms]} -> {nb / min(ms) * 1e3:,.0f} solves/s; "
          f"max rel err {err.max():.2e}; dn_iter max {np.abs(it[:256] - itr).max()}; mean n_iter {it.mean():.1f}; "
          f"factor_time mean {np.mean([x.factorization_time for x in d]) * 1e6:.1f} us", flush=True)
    print("   phases us [asm, factor, presolve-solve, P3, P4, total]:", ["%.1f" % (v * 1e6) for v in s.phase_seconds()[:8]], flush=True)
