"""Block -> instance map of a batch launch: static (instance = blockIdx.x) against the ticketed dynamic map
(claim_instance, fccqp_device.h), on workloads whose cost does or does not depend on the instance index.

    python tools/dispatch_order.py            # runs itself twice (FCCQP_DISPATCH=static, then dynamic), one JSON line

W-mix of fcc_qp_amd/problems.py puts the early-exit instances on the even indices: with the static map and
workgroups dealt round-robin over the 8 XCDs, four XCDs get every hard instance.  `shuffled` is the same batch
in a random order (the host-side control experiment VERDICT r2 asked for: if the diagnosis is right, shuffling
alone recovers the time the dynamic map recovers).
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure():
    import numpy as np
    from fcc_qp_amd import problems as P
    from tests import capi as C

    nb = 4096
    out = {}
    for wl in ("W-mix", "W-sat"):
        base = P.workload(wl, nb, P.WALKING, distinct=256)
        perm = np.random.default_rng(7).permutation(nb)
        for order, batch in (("natural", base), ("shuffled", P.Batch(*(np.ascontiguousarray(a[perm]) for a in base.arrays())))):
            s = C.Solver(*P.WALKING.ctor_args, batch=nb)
            s.set_options(**P.HARNESS_OPTIONS)
            ms = []
            for _ in range(8):
                s.solve_ok(*batch.arrays())
                ms.append(s.kernel_ms())
            z, d = s.solution()
            it = np.array([x.n_iter for x in d])
            out[f"{wl}/{order}"] = {"kernel_ms_min": min(ms[2:]), "kernel_ms_mean": float(np.mean(ms[2:])),
                                    "n_iter_mean": float(it.mean()), "z_checksum": float(np.abs(z).sum())}
            s.close()
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        print("RESULT " + json.dumps(measure()), flush=True)
    else:
        res = {}
        for mode in ("static", "dynamic"):
            env = dict(os.environ, FCCQP_DISPATCH=mode)
            txt = subprocess.check_output([sys.executable, os.path.abspath(__file__), "--child"], env=env, text=True)
            res[mode] = json.loads([l for l in txt.splitlines() if l.startswith("RESULT ")][-1][7:])
        print(json.dumps(res))
