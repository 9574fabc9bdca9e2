"""The two-launch form of the cooperative kernel against the one-launch kernel at a batch well beyond the benchmarked 4096
(large shape, 20 000 instances, 2.4 GB of hand-over workspace): bit-identical z and n_iter, tiled copies identical.
    python tools/big_split_check.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fcc_qp_amd import problems as P
from tests import capi as C
nb = 20000
B = P.workload("W-mix", nb, P.LARGE, distinct=24)
out = {}
for k in (C.KERNEL_COOP, C.KERNEL_COOP_SPLIT):
    s = C.Solver(*P.LARGE.ctor_args, batch=nb); s.set_options(**P.HARNESS_OPTIONS); s.set_kernel(k)
    s.solve_ok(*B.arrays())
    z, d = s.solution()
    out[k] = (z, [x.n_iter for x in d])
    print("kernel", s.kernel(), "ms", s.kernel_ms(), flush=True)
    s.close()
assert np.array_equal(out[C.KERNEL_COOP][0], out[C.KERNEL_COOP_SPLIT][0]) and out[C.KERNEL_COOP][1] == out[C.KERNEL_COOP_SPLIT][1]
for i in range(24, nb, 997):
    assert np.array_equal(out[C.KERNEL_COOP_SPLIT][0][i], out[C.KERNEL_COOP_SPLIT][0][i % 24])
print("batch", nb, "two-launch == one-launch bit for bit; tiled copies identical")
