"""PCIe-inclusive rate of the host-pointer entry point fccqp_solve (H2D of all inputs + kernel +
D2H of z and details), for DESIGN.md; never the benchmark's `value`."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from fcc_qp_amd import problems as P
from tests import capi as C

nb = 4096
B = P.workload("W-sat", nb, P.WALKING, distinct=512)
s = C.Solver(*P.WALKING.ctor_args, batch=nb); s.set_options(**P.HARNESS_OPTIONS)
for rep in range(5):
    t0 = time.perf_counter(); s.solve_ok(*B.arrays()); t1 = time.perf_counter()
    z, d = s.solution(); t2 = time.perf_counter()
    print(f"rep {rep}: solve (H2D + kernel) {1e3*(t1-t0):.2f} ms, get_solution (D2H) {1e3*(t2-t1):.2f} ms, "
          f"kernel {s.kernel_ms():.3f} ms -> end-to-end {nb/(t2-t0):,.0f} solves/s", flush=True)

# the fp32-storage entry point: half the bytes over PCIe, same arithmetic
a32 = [np.ascontiguousarray(x.astype(np.float32)) for x in B.arrays()]
for rep in range(4):
    t0 = time.perf_counter(); s.solve_f32_ok(*a32); t1 = time.perf_counter()
    z, d = s.solution(); t2 = time.perf_counter()
    print(f"f32 rep {rep}: solve (H2D + kernel) {1e3*(t1-t0):.2f} ms, get_solution (D2H) {1e3*(t2-t1):.2f} ms "
          f"-> end-to-end {nb/(t2-t0):,.0f} solves/s", flush=True)
