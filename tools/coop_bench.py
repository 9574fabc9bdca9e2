"""Cooperative kernel (FCCQP_KERNEL_COOP) against the kernels it replaces:
  * large shape FCCQP(114, 60, 24, 66), batch 4096, W-sat: coop vs generic (kernel ms, device phase stamps);
  * walking shape at small batches (1 .. 1024): coop vs the one-wave fused kernel (kernel ms = one QP's latency
    when the batch fits the chip).
    python tools/coop_bench.py [large|small|all]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fcc_qp_amd import problems as P
from tests import capi as C

what = sys.argv[1] if len(sys.argv) > 1 else "all"
if what in ("large", "all"):
    nb = 4096
    B = P.workload("W-sat", nb, P.LARGE, distinct=32)
    for kname, kern in (("coop", C.KERNEL_COOP), ("generic", C.KERNEL_GENERIC)):
        s = C.Solver(*P.LARGE.ctor_args, batch=nb); s.set_kernel(kern); s.set_options(**P.HARNESS_OPTIONS)
        s.phase_seconds()
        ms = []
        for _ in range(5):
            s.solve_ok(*B.arrays()); ms.append(s.kernel_ms())
        z, d = s.solution()
        print(f"large {kname}: kernel ms {['%.3f' % v for v in ms]} -> {nb / min(ms[1:]) * 1e3:,.0f} solves/s; mean n_iter {np.mean([x.n_iter for x in d]):.1f}; "
              f"phases us [load, factor, presolve, gram, admm, total] {['%.1f' % (1e6 * v) for v in s.phase_seconds()[:6]]}", flush=True)
        s.close()
if what in ("small", "all"):
    for nb in (1, 16, 64, 256, 512, 1024):
        B = P.workload("W-sat", nb, P.WALKING, distinct=min(nb, 64))
        row = []
        for kname, kern in (("coop", C.KERNEL_COOP), ("fused", C.KERNEL_FUSED)):
            for oname, opts, warm in (("cold100", P.HARNESS_OPTIONS, False), ("warm15", P.PAPER_OPTIONS, True)):
                s = C.Solver(*P.WALKING.ctor_args, batch=nb); s.set_kernel(kern); s.set_options(**opts)
                s.solve_ok(*B.arrays())
                s.set_warm_start(warm)
                ms = []
                for _ in range(12):
                    s.solve_ok(*B.arrays()); ms.append(s.kernel_ms())
                row.append(f"{kname}/{oname} {1e3 * np.median(ms[2:]):.1f} us")
                s.close()
        print(f"walking batch {nb}: " + "; ".join(row), flush=True)
