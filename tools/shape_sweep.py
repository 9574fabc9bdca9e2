"""Parity of fused kernels compiled for shapes beyond the shipped list (the claim of fused_shapes.def: "add your robot's
shape"): run with FCCQP_LIB_PATH pointing at a library built with the extra shapes in fused_shapes.def
(tools/variant_lib.sh), one-launch and two-launch form, cold and warm, f64 and f32 storage, against the oracle.
    FCCQP_LIB_PATH=tools/ab/libfccqp_hip_sweep.so python tools/shape_sweep.py "7 2 3 4" "16 0 3 0" ...
SWEEP_KERNEL=coop: the cooperative kernel instead (shapes added to coop_shapes.def)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcc_qp_amd import problems as P
from oracle import oracle as O
from tests import capi as CAPI
from tools.fuzz_parity import random_batch

bad = 0
for spec in sys.argv[1:]:
    dims = tuple(int(v) for v in spec.split())
    rng = np.random.default_rng(sum(dims))
    worst = 0.0
    for trial in range(12):
        opts = dict(max_iter=int(rng.choice([0, 1, 9, 40, 150])), rho=float(10.0 ** rng.uniform(-5, -1)),
                    eps_fcone=float(10.0 ** rng.uniform(-8, -3)), eps_bound=float(10.0 ** rng.uniform(-8, -3)))
        nb = int(rng.choice([1, 5, 70]))
        f32 = trial % 4 == 3
        B = random_batch(rng, dims, nb, singular=(trial == 5))
        if f32: B = P.Batch(*(np.asarray(np.asarray(a, dtype=np.float32), dtype=np.float64) for a in B.arrays()))
        z_ref, it_ref, _, _ = O.solve_batch(dims, opts, *B.arrays(), nthreads=8)
        for kernel in ((CAPI.KERNEL_COOP,) if os.environ.get("SWEEP_KERNEL") == "coop" else (CAPI.KERNEL_FUSED, CAPI.KERNEL_FUSED_SPLIT)):
            s = CAPI.Solver(*dims, batch=nb); s.set_options(**opts); s.set_kernel(kernel)
            (s.solve_f32_ok if f32 else s.solve_ok)(*B.arrays())
            assert s.kernel() == kernel
            z, d = s.solution()
            it = np.array([x.n_iter for x in d])
            err = np.max(np.abs(z - z_ref), axis=1) / (1.0 + np.max(np.abs(z_ref), axis=1))
            e, dn = float(err.max()) if np.all(np.isfinite(z)) else float("inf"), int(np.abs(it - it_ref).max())
            tol = 1e-4 if trial == 5 else 1e-6
            if not (e <= tol) or (dn > 1 and trial != 5):
                bad += 1
                print("BROKEN", dims, "kernel", kernel, "trial", trial, opts, "batch", nb, f"rel err {e:.2e} dn_iter {dn}", "f32" if f32 else "", flush=True)
            worst = max(worst, e if np.isfinite(e) and trial != 5 else 0.0)
            if nb <= 5 and opts["max_iter"] > 0 and trial != 5:      # a warm second solve
                s.set_warm_start(True); (s.solve_f32_ok if f32 else s.solve_ok)(*B.arrays()); z2, d2 = s.solution()
                for i in range(nb):
                    o = O.OracleFCCQP(*dims); oo = O.FCCQPOptions()
                    oo.max_iter, oo.rho, oo.eps_fcone, oo.eps_bound = opts["max_iter"], opts["rho"], opts["eps_fcone"], opts["eps_bound"]
                    o.set_options(oo); a = [x[i] for x in B.arrays()]
                    o.Solve(*a); o.set_warm_start(True); o.Solve(*a); sol = o.GetSolution()
                    e2 = float(np.max(np.abs(z2[i] - sol.z)) / (1.0 + np.max(np.abs(sol.z))))
                    if not (e2 <= 1e-6) or abs(int(d2[i].n_iter) - int(sol.details.n_iter)) > 1:
                        bad += 1; print("BROKEN warm", dims, "kernel", kernel, "trial", trial, f"rel err {e2:.2e}", flush=True)
                    worst = max(worst, e2)
            s.close()
    print(dims, "N", dims[0] + dims[1], "worst rel err %.2e" % worst, flush=True)
print("shape sweep:", bad, "broken")
sys.exit(1 if bad else 0)
