"""Non-interactive counterpart of the reference's demo harness (/root/reference/fcc_qp_test.py).

The reference script loads ``test_data/id_qp_log_walking.npz`` (:22-24), runs the warm-started
sequence ``solver.set_warm_start(i > 0); solver.Solve(...); solver.GetSolution()`` (:86-89) with
rho=5e-5, eps=1e-6/1e-6, max_iter=100 (:78-83) and shows eight matplotlib panels (:44-69).  Here the
same loop produces a JSON-serialisable report with the same eight series (no interactive backend
needed), and a second driver solves the whole log as ONE cold-start batch on the GPU -- the
``FCCQPBatch`` use case (the warm-start chain is sequential by construction and cannot be batched
along the sequence without changing results; it can be batched across independent logs: ``run_chains``).

    python -m fcc_qp_amd.harness [path/to/id_qp_log_walking.npz] [--out report.json] [--png plots.png]

With no path a synthetic gait sequence of the walking shape is used (the reference's data file is
not in its tree: .MISSING_LARGE_BLOBS).
"""
from __future__ import annotations

import argparse
import json
import time
from typing import Dict, List, Sequence

import numpy as np

from . import problems as P


def _options(F, opts: Dict):
    o = F.FCCQPOptions()
    o.max_iter, o.rho, o.eps_fcone, o.eps_bound = opts["max_iter"], opts["rho"], opts["eps_fcone"], opts["eps_bound"]
    return o


def _series(z: np.ndarray, details: Sequence, shape: P.Shape) -> Dict[str, List]:
    """The eight panels of make_plots() (fcc_qp_test.py:44-69), variable slices at :52-56."""
    nv, nu, nh = shape.nv, shape.nu, shape.nh
    s = shape.lambda_c_start
    return {
        "u": z[:, nv:nv + nu].tolist(),                                  # 'U Solution'
        "solve_time": [d.solve_time for d in details],                   # 'Solve Time'
        "n_iter": [int(d.n_iter) for d in details],                      # 'Number of Iterations'
        "friction_cone_viol": [d.friction_cone_viol for d in details],   # 'Fcone Viol'
        "vdot": z[:, :nv].tolist(),                                      # 'Vdot Solution'
        "lambda_h": z[:, nv + nu:nv + nu + nh].tolist(),                 # 'Lambda_h Solution'
        "lambda_c": z[:, s:s + shape.nc].tolist(),                       # 'Lambda_c Solution'
        "bounds_viol": [d.bounds_viol for d in details],                 # 'Bounds Viol'
    }


def run_sequence(qps: Sequence[Dict[str, np.ndarray]], shape: P.Shape = P.WALKING, options: Dict = None,
                 device: int = 0) -> Dict:
    """The reference loop, one instance at a time, warm-started from its predecessor."""
    import fcc_qp_amd as F

    options = dict(P.HARNESS_OPTIONS if options is None else options)
    solver = F.FCCQP(*shape.ctor_args, device)
    solver.set_options(_options(F, options))
    results = []
    t0 = time.perf_counter()
    for i, qp in enumerate(qps):
        solver.set_warm_start(i > 0)                                    # fcc_qp_test.py:87
        solver.Solve(qp["Q"], qp["b"], qp["A_eq"], qp["b_eq"], qp["friction_coeffs"], qp["lb"], qp["ub"])
        results.append(solver.GetSolution())
    wall = time.perf_counter() - t0
    z = np.vstack([r.z for r in results])
    rep = _series(z, [r.details for r in results], shape)
    rep.update(mode="sequential warm-start chain (reference semantics)", count=len(qps), wall_seconds=wall,
               options=options)
    return rep


def run_batch(qps: Sequence[Dict[str, np.ndarray]], shape: P.Shape = P.WALKING, options: Dict = None,
              device: int = 0) -> Dict:
    """The whole log as one cold-start batch: one kernel launch."""
    import fcc_qp_amd as F

    options = dict(P.HARNESS_OPTIONS if options is None else options)
    B = P.stack(qps)
    solver = F.FCCQPBatch(*shape.ctor_args, B.batch, device)
    solver.set_options(_options(F, options))
    t0 = time.perf_counter()
    solver.Solve(*B.arrays())
    sol = solver.GetSolution()
    wall = time.perf_counter() - t0
    rep = _series(sol.z, [sol[i].details for i in range(B.batch)], shape)
    rep.update(mode="one cold-start batch", count=B.batch, wall_seconds=wall, options=options,
               kernel_ms=solver.LastKernelMilliseconds())
    return rep


def run_chains(logs: Sequence[Sequence[Dict[str, np.ndarray]]], shape: P.Shape = P.WALKING, options: Dict = None,
               device: int = 0) -> List[Dict]:
    """K independent logs (robots, episodes) advanced in LOCK-STEP as one batch of K warm-started chains.

    Every chain follows the reference loop (/root/reference/fcc_qp_test.py:86-89) --
    ``set_warm_start(t > 0); Solve(qp_t); GetSolution()`` -- so chain k's results are those of ``run_sequence(logs[k])``;
    the K solves of a time step are one kernel launch of ``FCCQPBatch(batch=K)``, whose warm-start state (x, mu_x,
    mu_lambda_c per instance) stays resident on the device between steps.  This is the way to batch warm-started
    solves: along ONE sequence they are sequential by construction.  Logs may have different lengths: a chain that
    has ended keeps re-solving its last QP (its slot of the batch cannot be left out of a launch) and those results
    are dropped.  Returns one report per log, each with the eight series of ``run_sequence``.
    """
    import fcc_qp_amd as F

    options = dict(P.HARNESS_OPTIONS if options is None else options)
    logs = [list(l) for l in logs]
    if not logs or any(len(l) == 0 for l in logs):
        raise ValueError("run_chains: every log needs at least one QP")
    K, steps = len(logs), max(len(l) for l in logs)
    solver = F.FCCQPBatch(*shape.ctor_args, K, device)
    solver.set_options(_options(F, options))
    zs = [[] for _ in range(K)]
    details = [[] for _ in range(K)]
    kernel_ms = []
    t0 = time.perf_counter()
    for t in range(steps):
        B = P.stack([l[min(t, len(l) - 1)] for l in logs])
        solver.set_warm_start(t > 0)                                     # fcc_qp_test.py:87, for every chain
        solver.Solve(*B.arrays())
        sol = solver.GetSolution()
        kernel_ms.append(solver.LastTiming())        # (ms, 'device_events' | 'host_clock' | 'none')
        for k, l in enumerate(logs):
            if t < len(l):
                zs[k].append(np.array(sol.z[k], copy=True))
                details[k].append(sol[k].details)
    wall = time.perf_counter() - t0
    reports = []
    for k, l in enumerate(logs):
        rep = _series(np.vstack(zs[k]), details[k], shape)
        # one kind of measurement per report: small batches take the low-latency path, whose time is the host's
        # clock from launch to completion (results copied), larger ones a HIP event pair around the launches
        kinds = sorted({by for _, by in kernel_ms})
        by = kinds[0] if len(kinds) == 1 else "mixed"
        ms = [m for m, b in kernel_ms if b != "none"]
        rep.update(mode=f"warm-start chain {k} of {K} advanced in lock-step", count=len(l), wall_seconds=wall,
                   options=options, kernel_ms_per_step=(float(np.mean(ms)) if ms and by != "mixed" else None),
                   kernel_ms_measured_by=by)
        reports.append(rep)
    return reports


def save_png(report: Dict, path: str) -> bool:
    """The 2 x 4 figure of the reference (fcc_qp_test.py:58-68), written to a file; needs matplotlib."""
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except Exception:
        return False
    idx = list(range(report["count"]))
    panels = [("U Solution", "u", "u (nm)"), ("Solve Time", "solve_time", "Solve Time (seconds)"),
              ("Number of Iterations", "n_iter", "Iterations"), ("Fcone Viol", "friction_cone_viol", "Viol"),
              ("Vdot Solution", "vdot", "vdot"), ("Lambda_h Solution", "lambda_h", "Lambda_h (N or Nm)"),
              ("Lambda_c Solution", "lambda_c", "lambda_c (N)"), ("Bounds Viol", "bounds_viol", "viol")]
    fig, ax = plt.subplots(2, 4, figsize=(15, 9))
    for k, (title, key, ylabel) in enumerate(panels):
        a = ax[k // 4][k % 4]
        a.plot(idx, report[key]); a.set_title(title); a.set_xlabel("QP #"); a.set_ylabel(ylabel)
    fig.tight_layout(); fig.savefig(path); plt.close(fig)
    return True


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("log", nargs="?", help="id_qp_log_<name>.npz in the reference's format")
    ap.add_argument("--length", type=int, default=64, help="synthetic sequence length when no log is given")
    ap.add_argument("--mode", choices=["sequence", "batch", "both"], default="both")
    ap.add_argument("--out", default=None)
    ap.add_argument("--png", default=None)
    args = ap.parse_args(argv)
    qps = P.load_qp_log(args.log) if args.log else P.gait_sequence(args.length)
    report = {}
    if args.mode in ("sequence", "both"):
        report["sequence"] = run_sequence(qps)
    if args.mode in ("batch", "both"):
        report["batch"] = run_batch(qps)
    if args.png:
        save_png(report.get("sequence") or report["batch"], args.png)
    text = json.dumps(report)
    if args.out:
        open(args.out, "w").write(text)
    else:
        for k, r in report.items():
            print(f"{k}: {r['count']} QPs in {r['wall_seconds']:.3f} s, n_iter min/mean/max "
                  f"{min(r['n_iter'])}/{np.mean(r['n_iter']):.1f}/{max(r['n_iter'])}, "
                  f"max cone viol {max(r['friction_cone_viol']):.2e}, max bounds viol {max(r['bounds_viol']):.2e}")
    return report


if __name__ == "__main__":
    main()
