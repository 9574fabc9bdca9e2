"""One-launch fused kernel against its two-launch form (FCCQP_KERNEL_FUSED_SPLIT) at batch 4096, walking shape:
kernel time by HIP events around the whole enqueue (both launches + the fallback-list launch), per workload,
for 2 and 3 ADMM waves per SIMD (FCCQP_SPLIT_WAVES, read once per process -> one child process each).

    python tools/split_bench.py [batch]      # one JSON line
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure(nb):
    import numpy as np
    from fcc_qp_amd import problems as P
    from tests import capi as C

    out = {}
    cases = [("W-sat", P.HARNESS_OPTIONS, False), ("W-mix", P.HARNESS_OPTIONS, False), ("W-sat", P.PAPER_OPTIONS, True)]
    for wl, opts, warm in cases:
        batch = P.workload(wl, nb, P.WALKING, distinct=256)
        for kname, kern in (("fused", C.KERNEL_FUSED), ("split", C.KERNEL_FUSED_SPLIT)):
            s = C.Solver(*P.WALKING.ctor_args, batch=nb)
            s.set_options(**opts); s.set_kernel(kern)
            s.solve_ok(*batch.arrays())
            s.set_warm_start(warm)
            ms = []
            for _ in range(10):
                s.solve_ok(*batch.arrays())
                ms.append(s.kernel_ms())
            z, d = s.solution()
            it = np.array([x.n_iter for x in d])
            out[f"{wl}/{'paper-warm' if warm else 'harness-cold'}/{kname}"] = {
                "kernel_ms_min": min(ms[2:]), "kernel_ms_mean": float(np.mean(ms[2:])), "n_iter_mean": float(it.mean()),
                "solves_per_s": nb / (1e-3 * float(np.mean(ms[2:]))), "z_checksum": float(np.abs(z).sum())}
            s.close()
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        print("RESULT " + json.dumps(measure(int(sys.argv[2]))), flush=True)
    else:
        nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
        res = {}
        for waves in ("2", "3"):
            env = dict(os.environ, FCCQP_SPLIT_WAVES=waves)
            txt = subprocess.check_output([sys.executable, os.path.abspath(__file__), "--child", str(nb)], env=env, text=True)
            res[f"admm_waves_per_simd={waves}"] = json.loads([l for l in txt.splitlines() if l.startswith("RESULT ")][-1][7:])
        print(json.dumps(res))
