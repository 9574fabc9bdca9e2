"""Same-box A/B of builds of libfccqp_hip.so on what bench.py times: K back-to-back device-pointer solves (main kernel +
fallback-list launch each), host clock around the whole span, inputs resident (W-sat, walking shape, batch 4096):
    python tools/ab_span.py tools/ab/libfccqp_hip_a.so fcc_qp_amd/libfccqp_hip.so
Each library runs in its own child process (FCCQP_LIB_PATH), alternating; prints ms per step."""
import ctypes as C, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(wl, steps):
    import numpy as np, torch
    from fcc_qp_amd import problems as P
    from tests import capi as CAPI
    nb = 4096
    B = P.workload(wl, nb, P.WALKING, distinct=256)
    s = CAPI.Solver(*P.WALKING.ctor_args, batch=nb); s.set_options(**P.HARNESS_OPTIONS); s.set_kernel_timing(False)
    dev = [torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda() for a in B.arrays()]
    dp = C.POINTER(C.c_double)
    ptr = [C.cast(t.data_ptr(), dp) for t in dev]
    L = s.L
    L.fccqp_solve_device.argtypes = [C.c_void_p] + [dp] * 7 + [C.c_void_p]
    own = C.c_void_p(-1)                                  # FCCQP_STREAM_OWN = (void *)-1
    def run(k):
        for _ in range(k):
            rc = L.fccqp_solve_device(s.h, *ptr, own)
            assert rc == 0, rc
        assert L.fccqp_synchronize(s.h) == 0
    run(20)
    out = []
    for _ in range(5):
        t0 = time.perf_counter(); run(steps); out.append((time.perf_counter() - t0) / steps * 1e3)
    print("RESULT " + json.dumps({"median": float(np.median(out)), "min": min(out)}))


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]))
    else:
        libs = sys.argv[1:]
        for wl in ("W-sat",):
            res = {l: [] for l in libs}
            for rep in range(3):
                for l in libs:
                    env = dict(os.environ, FCCQP_LIB_PATH=os.path.abspath(l))
                    txt = subprocess.check_output([sys.executable, os.path.abspath(__file__), "--child", wl, "200"], env=env, text=True)
                    res[l].append(json.loads([x for x in txt.splitlines() if x.startswith("RESULT ")][-1][7:])["median"])
            print(wl, {l: ["%.4f" % v for v in r] for l, r in res.items()}, flush=True)
