"""Latency of the reference-style single-instance class (FCCQP, batch = 1): wall time of Solve (host
buffers in, H2D + kernel + sync) and device kernel time, cold and warm start."""
import sys, time, numpy as np
sys.path.insert(0, ".")
import fcc_qp_amd as F
from fcc_qp_amd import problems as P

qps = P.gait_sequence(60, P.WALKING, seed0=5000)
for name, opts, warm in (("harness/cold", P.HARNESS_OPTIONS, False), ("paper/warm", P.PAPER_OPTIONS, True)):
    s = F.FCCQP(60, 38, 12, 38)
    o = F.FCCQPOptions(); o.max_iter, o.rho, o.eps_fcone, o.eps_bound = (opts["max_iter"], opts["rho"], opts["eps_fcone"], opts["eps_bound"])
    s.set_options(o)
    wall, wall_get, its = [], [], []
    for i, qp in enumerate(qps):
        s.set_warm_start(warm and i > 0)
        t0 = time.perf_counter()
        s.Solve(qp["Q"], qp["b"], qp["A_eq"], qp["b_eq"], qp["friction_coeffs"], qp["lb"], qp["ub"])
        t1 = time.perf_counter()
        r = s.GetSolution()
        t2 = time.perf_counter()
        wall.append(t1 - t0); wall_get.append(t2 - t1); its.append(r.details.n_iter)
    w = np.array(wall[5:]) * 1e6
    g = np.array(wall_get[5:]) * 1e6
    print(f"{name}: Solve() wall median {np.median(w):.0f} us  (p10 {np.percentile(w,10):.0f}, p90 {np.percentile(w,90):.0f}); "
          f"GetSolution() median {np.median(g):.0f} us; details.solve_time last {r.details.solve_time*1e6:.0f} us; "
          f"mean n_iter {np.mean(its):.1f}", flush=True)
