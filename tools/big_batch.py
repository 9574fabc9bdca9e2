"""Scale check: 65536 walking-shape instances (BASELINE configs[4] total, here on one GPU): kernel time
and bit-identical results across the tiled copies."""
import sys, numpy as np
sys.path.insert(0, ".")
from fcc_qp_amd import problems as P
from tests import capi as C
nb = 65536
base = P.workload("W-mix", 256, P.WALKING)
B = P.Batch(*(np.ascontiguousarray(np.tile(a, (nb // 256,) + (1,) * (a.ndim - 1))) for a in base.arrays()))
s = C.Solver(*P.WALKING.ctor_args, batch=nb); s.set_options(**P.HARNESS_OPTIONS)
s.solve_ok(*B.arrays()); ms = s.kernel_ms()
z, d = s.solution()
it = np.array([x.n_iter for x in d])
ok = all(np.array_equal(z[:256], z[256*r:256*(r+1)]) for r in range(1, nb // 256))
print("batch", nb, "kernel ms %.2f" % ms, "-> %.0f solves/s" % (nb / ms * 1e3), "tiles identical:", ok, "n_iter hist", dict(zip(*np.unique(it, return_counts=True))))
