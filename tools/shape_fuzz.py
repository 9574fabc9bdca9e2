"""Random problem shapes through the C ABI against the oracle (any-shape generic kernel; the specialised kernels where a
drawn shape happens to be listed): N = num_vars + num_eq from 4 to 512, cold solve + one warm solve.
    python tools/shape_fuzz.py [count] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fcc_qp_amd import problems as P
from tests import capi as C
from oracle import oracle as O
from tests.test_gpu_parity import _random_qp
from tests.helpers import rel_err

count = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
bad = 0
for k in range(count):
    N = int(rng.choice([rng.integers(4, 64), rng.integers(64, 200), rng.integers(200, 513)]))
    n = int(rng.integers(max(3, N // 2 + 1), max(4, (3 * N) // 4) + 2))     # more variables than equality rows:
    m = max(1, N - n)                                                        # a random A_eq then has full row rank
    if m >= n: m = n - 1
    ncont = int(rng.integers(1, max(2, min(16, n // 3))))
    nc = 3 * ncont
    if nc > n: nc = 3 * (n // 3)
    s = int(rng.integers(0, n - nc + 1))
    dims = (n, m, nc, s)
    nb = 2
    B = P.stack([_random_qp(rng, n, m, nc, s) for _ in range(nb)])
    opts = dict(max_iter=int(rng.integers(1, 30)), rho=float(10.0 ** rng.uniform(-3, -0.5)), eps_fcone=1e-7, eps_bound=1e-7)
    zr, itr = O.solve_batch(dims, opts, *B.arrays(), nthreads=2)[:2]
    sv = C.Solver(*dims, batch=nb); sv.set_options(**opts)
    sv.solve_ok(*B.arrays()); z, d = sv.solution()
    e = rel_err(z, zr).max(); dn = max(abs(d[i].n_iter - int(itr[i])) for i in range(nb))
    ok = np.isfinite(e) and e <= 1e-6 and dn <= 1
    bad += int(not ok)
    print(f"{dims} N={n + m} kernel {sv.kernel()} max_iter {opts['max_iter']} rho {opts['rho']:.1e}: rel err {e:.1e} dn_iter {dn} {'ok' if ok else 'FAIL'}", flush=True)
    sv.close()
print("failures:", bad)
sys.exit(1 if bad else 0)
