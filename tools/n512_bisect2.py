"""8-slot debug build: first solve WARM from the zero state (no presolve phase before the K_rho factorisation)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fcc_qp_amd import problems as P
from tests import capi as C
from oracle import oracle as O
from tests.test_gpu_parity import _random_qp

dims = (180, 80, 12, 100)
n, m, nc, s = dims
rng = np.random.default_rng(sum(dims))
qp = _random_qp(rng, n, m, nc, s)
B = P.stack([qp])
for it in (1, 2, 25):
    opts = dict(max_iter=it, rho=0.1, eps_fcone=1e-7, eps_bound=1e-7)
    o = O.OracleFCCQP(*dims); op = O.FCCQPOptions(); op.__dict__.update(opts); o.set_options(op)
    o.set_warm_start(True); o.Solve(**qp); r = o.GetSolution().z
    errs = []
    for rep in range(6):
        sv = C.Solver(*dims, batch=1); sv.set_options(**opts); sv.set_kernel(C.KERNEL_GENERIC); sv.set_warm_start(True)
        sv.solve_ok(*B.arrays()); z, d = sv.solution(); sv.close()
        errs.append("%.1e" % (np.max(np.abs(z[0] - r)) / (1 + np.max(np.abs(r)))))
    print("warm-from-zero max_iter", it, errs, flush=True)
