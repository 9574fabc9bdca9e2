"""Runs the two-launch form (or the one-launch form) a few times for rocprofv3 --kernel-trace --stats and prints the
device phase stamps.   usage: python tools/split_trace.py <W-sat|W-mix> <fused|split> [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fcc_qp_amd import problems as P
from tests import capi as C

wl = sys.argv[1] if len(sys.argv) > 1 else "W-sat"
kern = {"fused": C.KERNEL_FUSED, "split": C.KERNEL_FUSED_SPLIT}[sys.argv[2] if len(sys.argv) > 2 else "split"]
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
B = P.workload(wl, nb, P.WALKING, distinct=256)
s = C.Solver(*P.WALKING.ctor_args, batch=nb)
s.set_options(**P.HARNESS_OPTIONS); s.set_kernel(kern)
s.phase_seconds()
ms = []
for _ in range(12):
    s.solve_ok(*B.arrays()); ms.append(s.kernel_ms())
print(wl, sys.argv[2:], "kernel ms", ["%.3f" % m for m in ms])
print("phases us [load, factor, presolve, gram, admm, total]:", ["%.1f" % (v * 1e6) for v in s.phase_seconds()[:6]])
t = s.phase_ticks()
print("admm phase per instance us: min %.1f median %.1f max %.1f" % (t[4].min() / 100, np.median(t[4]) / 100, t[4].max() / 100))
