"""Device phase stamps of one kernel on one shape at one batch size:
    python tools/phase_probe.py <walking|large|mid> <coop|coop_split|fused|generic|split> <batch> [W-sat] [max_iter]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fcc_qp_amd import problems as P
from tests import capi as C
shape = {"walking": P.WALKING, "large": P.LARGE, "mid": P.MID}[sys.argv[1]]
kern = {"coop": C.KERNEL_COOP, "fused": C.KERNEL_FUSED, "generic": C.KERNEL_GENERIC, "split": C.KERNEL_FUSED_SPLIT, "coop_split": C.KERNEL_COOP_SPLIT}[sys.argv[2]]
nb = int(sys.argv[3])
wl = sys.argv[4] if len(sys.argv) > 4 else "W-sat"
opts = dict(P.HARNESS_OPTIONS)
if len(sys.argv) > 5:
    opts["max_iter"] = int(sys.argv[5])
B = P.workload(wl, nb, shape, distinct=min(nb, 32))
s = C.Solver(*shape.ctor_args, batch=nb); s.set_kernel(kern); s.set_options(**opts)
s.phase_seconds()
ms = []
if os.environ.get("PROBE_SPARSE"):
    # host-pointer solves: a 200 MB copy and an idle GPU between the launches -- the clocks of occasional use
    for _ in range(6):
        s.solve_ok(*B.arrays()); ms.append(s.kernel_ms())
else:
    # device-pointer solves back to back, as bench.py times them: after ~70 launches (20 ms) the card has ramped its
    # clocks and a launch is ~9 % shorter than the first ones after an idle phase (bench.py _settle)
    import ctypes, torch
    dev = [torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda() for a in B.arrays()]
    ptr = [ctypes.c_void_p(t.data_ptr()) for t in dev]
    for _ in range(int(os.environ.get("PROBE_REPS", "200"))):
        assert s.L.fccqp_solve_device(s.h, *ptr, ctypes.c_void_p(-1)) == 0
    assert s.L.fccqp_synchronize(s.h) == 0
    ms = [s.kernel_ms()] * 5
print(sys.argv[1:], "kernel us %.1f" % (1e3 * np.median(ms[-5:])), "phases us [load, factor, presolve, gram, admm, total]:",
      ["%.1f" % (1e6 * v) for v in s.phase_seconds()[:6]], "slot6 (coop: W factorisation) %.1f" % (1e6 * s.phase_seconds()[6]), flush=True)
