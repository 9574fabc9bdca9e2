"""Bisection experiments on the 8-slot debug build (see tools/n512_probe.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fcc_qp_amd import problems as P
from tests import capi as C
from oracle import oracle as O
from tests.test_gpu_parity import _random_qp
from tests.helpers import rel_err

dims = (180, 80, 12, 100)
n, m, nc, s = dims
rng = np.random.default_rng(sum(dims))
qps = [_random_qp(rng, n, m, nc, s) for _ in range(3)]
for nb in (1, 3):
    B = P.stack(qps[:nb])
    for it in (0, 1, 2, 5, 25):
        opts = dict(max_iter=it, rho=0.1, eps_fcone=1e-7, eps_bound=1e-7)
        r = O.solve_batch(dims, opts, *B.arrays(), nthreads=3)[0]
        errs = []
        zs = []
        for rep in range(4):
            sv = C.Solver(*dims, batch=nb); sv.set_options(**opts); sv.set_kernel(C.KERNEL_GENERIC)
            sv.solve_ok(*B.arrays()); z, d = sv.solution(); sv.close()
            errs.append("%.1e" % rel_err(z, r).max()); zs.append(z)
        same = all(np.array_equal(zs[0], z) for z in zs[1:])
        # which entries differ from the oracle most
        e = np.abs(zs[0] - r) / (1 + np.abs(r).max())
        print("batch", nb, "max_iter", it, "errs", errs, "bitwise-repeatable", same, "worst entries", np.argsort(e[0])[-4:].tolist(), flush=True)
