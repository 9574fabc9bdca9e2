"""Same-box A/B of two builds of libfccqp_hip.so on the headline workload (W-sat, walking shape, batch 4096):
    python tools/ab_bench.py tools/ab/libfccqp_hip_r2.so fcc_qp_amd/libfccqp_hip.so
Each library runs in its own child process (FCCQP_LIB_PATH), alternating A B A B; prints median kernel ms."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def child(wl):
    import numpy as np
    from fcc_qp_amd import problems as P
    from tests import capi as C
    nb = 4096
    B = P.workload(wl, nb, P.WALKING, distinct=256)
    s = C.Solver(*P.WALKING.ctor_args, batch=nb); s.set_options(**P.HARNESS_OPTIONS)
    ms = []
    for _ in range(30):
        s.solve_ok(*B.arrays()); ms.append(s.kernel_ms())
    print("RESULT " + json.dumps({"median": float(np.median(ms[5:])), "min": min(ms[5:])}))

if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        libs = sys.argv[1:]
        for wl in ("W-sat", "W-mix"):
            res = {l: [] for l in libs}
            for rep in range(3):
                for l in libs:
                    env = dict(os.environ, FCCQP_LIB_PATH=os.path.abspath(l))
                    txt = subprocess.check_output([sys.executable, os.path.abspath(__file__), "--child", wl], env=env, text=True)
                    res[l].append(json.loads([x for x in txt.splitlines() if x.startswith("RESULT ")][-1][7:])["median"])
            print(wl, {l: ["%.4f" % v for v in r] for l, r in res.items()}, flush=True)
