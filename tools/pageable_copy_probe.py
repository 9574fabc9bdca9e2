"""Which path does the HIP runtime take for the pageable host-to-device copies of fccqp_solve?  Runs a few host-pointer
solves of two shapes under AMD_LOG_LEVEL=4 (set by the caller) and greps the runtime's log for its own words: "HSA Copy
Using Pinned resource" (the runtime page-locks the CALLER's memory and a copy kernel reads it in place) against "HSA Copy
Using Staging resource" (the runtime copies through its own page-locked buffers).  Evidence for DESIGN.md section 7.3.
    AMD_LOG_LEVEL=4 AMD_LOG_LEVEL_FILE=gpurun_out/copy_probe.log python tools/pageable_copy_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fcc_qp_amd import problems as P
from tests import capi as C

for name, shape, nb in (("large x16", P.LARGE, 16), ("walking x64", P.WALKING, 64), ("walking x4", P.WALKING, 4)):
    B = P.workload("W-mix", nb, shape, distinct=min(nb, 16))
    s = C.Solver(*shape.ctor_args, batch=nb); s.set_options(**P.HARNESS_OPTIONS)
    print("MARK begin", name, "bytes per array", [a.nbytes for a in B.arrays()], flush=True)
    for _ in range(2):
        s.solve_ok(*B.arrays())
    print("MARK end", name, flush=True)
    s.close()
