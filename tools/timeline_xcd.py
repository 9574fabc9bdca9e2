"""Per-XCD view of one launch of the fused kernel: end time of the last instance and mean instance duration by
blockIdx.x % 8 (workgroups are dealt round-robin over the 8 XCDs, so block b runs on the XCD of b % 8), and by
wave-round within an XCD.  Shows whether the launch ends on a slow XCD / a slow CU or on dispatch gaps."""
import sys, ctypes
import numpy as np
sys.path.insert(0, ".")
from fcc_qp_amd import problems as P
from tests import capi as C

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
B = P.workload("W-sat", nb, P.WALKING, distinct=256)
s = C.Solver(*P.WALKING.ctor_args, batch=nb)
s.set_options(**P.HARNESS_OPTIONS); s.set_kernel(C.KERNEL_FUSED)
s.phase_seconds()
for rep in range(3):
    s.solve_ok(*B.arrays())
t = s.phase_ticks()
start, dur = t[7] * 1e-2, t[5] * 1e-2
t0 = start.min(); start = start - t0; end = start + dur
L = C.lib()
L.fccqp_debug_block_instance.argtypes = [ctypes.c_int, ctypes.c_int]
inst_of_block = np.array([L.fccqp_debug_block_instance(nb, b) for b in range(nb)])
blk = np.empty(nb, dtype=np.int64); blk[inst_of_block] = np.arange(nb)
xcd = blk % 8
print(f"launch: last end {end.max():.1f} us, mean duration {dur.mean():.1f}")
for x in range(8):
    m = xcd == x
    o = np.argsort(start[m]); q = np.array_split(o, 4)
    gaps = []
    print(f"  xcd-slot {x}: instances {m.sum()}, duration mean {dur[m].mean():.1f} (p5 {np.percentile(dur[m],5):.1f} p95 {np.percentile(dur[m],95):.1f}), "
          f"first start {start[m].min():.1f}, last end {end[m].max():.1f}, ends of last quarter {np.percentile(end[m][q[3]],5):.1f}..{end[m][q[3]].max():.1f}")
# busy fraction: sum of durations / (1024 slots x last end)
print(f"  busy fraction of the 1024 wave slots: {dur.sum() / (1024 * end.max()):.3f}")
