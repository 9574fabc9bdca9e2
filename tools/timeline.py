"""Lays one launch of the fused kernel out on the device clock from the per-instance stamps
(fccqp_debug_phase_ticks): when each instance started and ended, how the instances queue up behind
each other, and how much of the launch (by hipEvents) is not covered by instance work."""
import sys
import numpy as np
sys.path.insert(0, ".")
from fcc_qp_amd import problems as P
from tests import capi as C

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
wl = sys.argv[2] if len(sys.argv) > 2 else "W-sat"
B = P.workload(wl, nb, P.WALKING, distinct=256)
s = C.Solver(*P.WALKING.ctor_args, batch=nb)
s.set_options(**P.HARNESS_OPTIONS); s.set_kernel(C.KERNEL_FUSED)
s.phase_seconds()
for rep in range(3):
    s.solve_ok(*B.arrays())
ms = s.kernel_ms()
t = s.phase_ticks()
start, dur = t[7] * 1e-2, t[5] * 1e-2          # microseconds
t0 = start.min()
end = start + dur - t0
start = start - t0
print(f"kernel by events {ms * 1e3:.1f} us; first start 0, last end {end.max():.1f} us; mean duration {dur.mean():.1f} us "
      f"(min {dur.min():.1f}, max {dur.max():.1f})")
order = np.argsort(start)
rounds = np.array_split(order, 4)
for k, r in enumerate(rounds):
    print(f"  quarter {k} of the instances by start time: start {start[r].min():7.1f} .. {start[r].max():7.1f} us, "
          f"duration mean {dur[r].mean():6.1f} us, end {end[r].min():7.1f} .. {end[r].max():7.1f}")
ph = ["asm", "factor", "presolve", "P3", "P4"]
for k, r in enumerate(rounds):
    print(f"  quarter {k} phases us:", {p: round(float(t[i][r].mean()) * 1e-2, 1) for i, p in enumerate(ph)})
