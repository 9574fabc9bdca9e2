"""Which path does the HIP runtime take for transfers between the CALLER's pageable arrays and the device?  Runs
host-pointer solves, result fetches and state transplants whose arrays are well above 1 MiB under AMD_LOG_LEVEL=4 (set by
the caller) -- the runtime's own log then says, per copy, "HSA Copy Using Pinned resource" (it page-locked the caller's
pages and a copy kernel reads them in place: the path of the two process aborts) or "HSA Copy Using
Staging resource" (its own page-locked buffers).  Since round 5 the library bounces pageable arrays through page-locked
buffers of its own and neither line may appear for a caller array; tests/test_gpu_transfers.py runs this file in a child
process and checks the log.
    AMD_LOG_LEVEL=4 AMD_LOG_LEVEL_FILE=/tmp/copy_probe.log python tools/pageable_copy_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fcc_qp_amd import problems as P
from tests import capi as C

for name, shape, nb in (("large x24", P.LARGE, 24), ("walking x4096", P.WALKING, 4096), ("walking x4", P.WALKING, 4)):
    B = P.workload("W-mix", nb, shape, distinct=min(nb, 16))
    s = C.Solver(*shape.ctor_args, batch=nb); s.set_options(**P.HARNESS_OPTIONS)
    print("MARK begin", name, "bytes per array", [a.nbytes for a in B.arrays()], flush=True)
    s.solve_ok(*B.arrays())
    z, d = s.solution()                                   # z: nb x n doubles out to a pageable array
    x, mx, ml = s.state()
    t = C.Solver(*shape.ctor_args, batch=nb); t.set_options(**P.HARNESS_OPTIONS)
    assert t.L.fccqp_set_state(t.h, C._p(x), C._p(mx), C._p(np.ascontiguousarray(ml))) == 0      # ... and back in
    t.set_warm_start(True); t.solve_ok(*B.arrays())
    s.set_warm_start(True); s.solve_ok(*B.arrays())
    z1, _ = s.solution(); z2, _ = t.solution()
    assert np.array_equal(z1, z2), "a transplanted state continues exactly as the original"
    s.set_warm_start(False); s.solve_f32_ok(*B.arrays())  # fp32-storage entry point
    zf, _ = s.solution()
    assert np.all(np.isfinite(zf))
    print("MARK end", name, flush=True)
    s.close(); t.close()
print("MARK done", flush=True)
