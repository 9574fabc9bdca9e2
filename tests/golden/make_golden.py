"""Generates tests/golden/*.npz -- inputs and expected outputs of the hot path.

The reference cannot be built or run in this image (its Eigen submodule is empty and Eigen is not
installed: SURVEY.md F2) and ships no golden vectors of its own (F3), so these vectors come from
the CPU oracle (oracle/fccqp_oracle.c, a restatement of the reference algorithm: PARITY UNPINNED).
They pin (a) the oracle against silent change across compilers/machines and (b) the HIP path on
the GPU box, where /root/reference and this generator's seeds are the only common ground.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fcc_qp_amd import problems as P          # noqa: E402
from oracle import oracle as O                # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def solve_cases(dims, options, qps, trace=6, warm_chain=False):
    o = O.OracleFCCQP(*dims)
    op = O.FCCQPOptions(); op.__dict__.update(options); o.set_options(op)
    out = []
    for i, qp in enumerate(qps):
        o.set_warm_start(bool(warm_chain and i > 0))
        o.enable_trace(trace)
        o.Solve(**qp)
        s = o.GetSolution(); tr = o.trace(); d = s.details
        def padded(a):                        # traces are shorter than `trace` after an early exit
            full = np.full((trace, a.shape[1]), np.nan); full[:a.shape[0]] = a
            return full
        out.append(dict(z=s.z, n_iter=d.n_iter, eps_bounds=d.eps_bounds, eps_friction_cone=d.eps_friction_cone,
                        bounds_viol=d.bounds_viol, friction_cone_viol=d.friction_cone_viol,
                        solve_status=d.solve_status, trace_len=len(tr["x"]), trace_x=padded(tr["x"]),
                        trace_mu_x=padded(tr["mu_x"]), trace_lambda_c_bar=padded(tr["lambda_c_bar"])))
    return out


def pack(name, dims, options, qps, results, note):
    arrs = dict(dims=np.array(dims), note=np.array(note),
                options=np.array([options["max_iter"], options["rho"], options["eps_fcone"], options["eps_bound"]]))
    for key in ("Q", "b", "A_eq", "b_eq", "friction_coeffs", "lb", "ub"):
        arrs[key] = np.stack([np.asarray(q[key], dtype=np.float64) for q in qps])
    for key in results[0]:
        vals = [r[key] for r in results]
        arrs["out_" + key] = np.stack([np.asarray(v) for v in vals]) if np.ndim(vals[0]) else np.array(vals)
    np.savez_compressed(os.path.join(HERE, name), **arrs)
    print(name, {k: v.shape for k, v in arrs.items() if k.startswith("out_")})


def main():
    # 1. walking shape, harness options: two saturating, one feasible-from-the-start instance
    qps = [P.osc_qp(1000, P.WALKING, 4.0), P.osc_qp(1001, P.WALKING, 4.0),
           P.osc_qp(1009, P.WALKING, 0.0, ulim=(400.0, 600.0), lambda_reg=1e-2)]
    pack("walking_harness.npz", P.WALKING.ctor_args, P.HARNESS_OPTIONS, qps,
         solve_cases(P.WALKING.ctor_args, P.HARNESS_OPTIONS, qps), "cold start, fcc_qp_test.py:78-83 options")
    # 2. walking shape, converging options (intermediate iteration counts)
    opts = dict(max_iter=1000, rho=1e-2, eps_fcone=1e-6, eps_bound=1e-6)
    qps = [P.osc_qp(1000 + i, P.WALKING, 4.0) for i in (3, 14)]
    pack("walking_converging.npz", P.WALKING.ctor_args, opts, qps, solve_cases(P.WALKING.ctor_args, opts, qps),
         "cold start, rho=1e-2: exits between iteration 17 and 1000")
    # 3. warm-started chain (fcc_qp_test.py:86-89 semantics), paper options
    qps = P.gait_sequence(4, P.WALKING, seed0=7000)
    pack("walking_warm_chain.npz", P.WALKING.ctor_args, P.PAPER_OPTIONS, qps,
         solve_cases(P.WALKING.ctor_args, P.PAPER_OPTIONS, qps, warm_chain=True),
         "set_warm_start(i > 0) over a 4-long sequence, fccqp.pdf Table 1 options")
    # 4. small shape (the second fused instantiation)
    rng = np.random.default_rng(42)
    dims = (12, 5, 6, 3)
    qps = []
    for _ in range(3):
        G = rng.normal(size=(12, 12)); lb = np.full(12, -np.inf); ub = np.full(12, np.inf)
        lb[:3] = -0.25; ub[:3] = 0.25
        qps.append(dict(Q=G @ G.T + 0.5 * np.eye(12), b=rng.normal(size=12), A_eq=rng.normal(size=(5, 12)),
                        b_eq=rng.normal(size=5), friction_coeffs=rng.uniform(0.3, 1.0, 2), lb=lb, ub=ub))
    opts = dict(max_iter=60, rho=0.1, eps_fcone=1e-7, eps_bound=1e-7)
    pack("mini_shape.npz", dims, opts, qps, solve_cases(dims, opts, qps), "FCCQP(12,5,6,3) random SPD cost")


if __name__ == "__main__":
    main()
