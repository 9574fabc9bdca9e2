"""What the solver's own start-of-kernel hipEventRecord costs in stream time: 200 solves of the headline batch back to
back (no events of the caller in between), wall clock per solve, with fccqp_set_kernel_timing on (default) and off."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fcc_qp_amd as F
from fcc_qp_amd import problems as P
nb = 4096
dev = torch.device("cuda:0")
B = P.workload("W-sat", nb, P.WALKING, distinct=256)
solver = F.FCCQPBatch(*P.WALKING.ctor_args, nb, 0)
o = F.FCCQPOptions(); h = P.HARNESS_OPTIONS
o.max_iter, o.rho, o.eps_fcone, o.eps_bound = h["max_iter"], h["rho"], h["eps_fcone"], h["eps_bound"]
solver.set_options(o)
tens = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in B.arrays()]
sets = [tens, [t.clone() for t in tens]]
def run(k):
    for i in range(k): F.solve_tensors(solver, *sets[i % 2])
    torch.cuda.synchronize()
for on in (True, False, True, False):
    solver.set_kernel_timing(on)
    run(20)
    t0 = time.perf_counter(); run(200); t1 = time.perf_counter()
    print(f"kernel timing {'on ' if on else 'off'} {1e6 * (t1 - t0) / 200:8.2f} us per solve", flush=True)
