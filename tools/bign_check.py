"""Global-workspace shapes (198 < N <= 256) of the generic kernel against the oracle, repeated; with
argument `dirty`, registers / LDS / scratch of the device are filled with NaN patterns before every solve
(tools/dirty_regs.hip)."""
import sys, ctypes, subprocess
import numpy as np
sys.path.insert(0, ".")
from fcc_qp_amd import problems as P
from tests import capi as C
from oracle import oracle as O
from tests.test_gpu_parity import _random_qp
from tests.helpers import rel_err
dirty = None
if "dirty" in sys.argv:
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O1", "-w", "-shared", "-fPIC", "tools/dirty_regs.hip", "-o", "/tmp/libdirty.so"])
    C.lib()
    dirty = ctypes.CDLL("/tmp/libdirty.so").dirty_device
tot = 0
for dims in [(140, 60, 12, 100), (160, 70, 30, 120), (196, 60, 12, 100), (130, 126, 9, 100), (130, 70, 24, 100)]:
    n, m, nc, s = dims
    rng = np.random.default_rng(sum(dims))
    B = P.stack([_random_qp(rng, n, m, nc, s) for _ in range(3)])
    opts = dict(max_iter=25, rho=0.1, eps_fcone=1e-7, eps_bound=1e-7)
    r = O.solve_batch(dims, opts, *B.arrays(), nthreads=3)[0]
    for kern in (C.KERNEL_GENERIC, C.KERNEL_AUTO):
        bad = 0; ms = 0.0; worst = 0.0
        for rep in range(6):
            sv = C.Solver(*dims, batch=3); sv.set_options(**opts); sv.set_kernel(kern)
            if dirty: assert dirty() == 0
            sv.solve_ok(*B.arrays()); z, d = sv.solution(); ms = sv.kernel_ms(); sv.close()
            e = rel_err(z, r).max(); worst = max(worst, e if np.isfinite(e) else 1e300)
            bad += int(not (e <= 1e-9))
        tot += bad
        print(dims, "kernel", kern, "ms %.1f" % ms, "bad %d/6" % bad, "worst %.1e" % worst, flush=True)
sys.exit(1 if tot else 0)
