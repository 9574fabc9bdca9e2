"""An all-singular batch (every instance has a dead variable: K has a zero row, the reference's COD
branch, fcc_qp.cpp:164-166): throughput of the device COD pass, and parity of a sample."""
import sys
import numpy as np
sys.path.insert(0, ".")
from fcc_qp_amd import problems as P
from tests import capi as C
from oracle import oracle as O
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
base = P.workload("W-mix", 64, P.WALKING)
qps = []
for i in range(64):
    q = {k: np.array(v, copy=True) for k, v in base.instance(i).items()}
    j = 22 + (i % 10)
    q["Q"][j, :] = 0.0; q["Q"][:, j] = 0.0; q["A_eq"][:, j] = 0.0; q["b"][j] = 0.0
    qps.append(q)
B64 = P.stack(qps)
B = P.Batch(*(np.ascontiguousarray(np.tile(a, (nb // 64,) + (1,) * (a.ndim - 1))) for a in B64.arrays()))
zr, itr, _, _ = O.solve_batch(P.WALKING.ctor_args, P.HARNESS_OPTIONS, *B64.arrays(), nthreads=8)
for kern, name in [(C.KERNEL_AUTO, "auto (fused main kernel)"), (C.KERNEL_GENERIC, "generic")]:
    s = C.Solver(*P.WALKING.ctor_args, batch=nb); s.set_options(**P.HARNESS_OPTIONS); s.set_kernel(kern)
    ms = []
    for rep in range(3):
        s.solve_ok(*B.arrays()); ms.append(s.kernel_ms())
    z, d = s.solution()
    err = np.max(np.abs(z[:64] - zr), axis=1) / (1 + np.max(np.abs(zr), axis=1))
    cod = sum(1 for x in d if x.flags & C.FLAG_PRESOLVE_COD)
    print(f"{name}: all-COD batch {nb}: kernel ms {['%.2f' % m for m in ms]} -> {nb / min(ms) * 1e3:,.0f} solves/s; COD instances {cod}; "
          f"max rel err {err.max():.2e}; dn_iter {np.abs(np.array([x.n_iter for x in d[:64]]) - itr).max()}", flush=True)
