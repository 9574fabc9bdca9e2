"""Same-box A/B of the cooperative kernel's one-launch and two-launch forms on the large shape (BASELINE configs[3]):
K back-to-back device-pointer solves, host clock around the span, inputs resident, batch 4096 (or argv), W-sat and W-mix.
    python tools/coop_split_ab.py [batch] [steps]
Variants (each in its own child process, alternating, three rounds): coop (one launch), coop_split with 2 and with 3
workgroups per CU in the second launch (FCCQP_COOP_SPLIT_WGS).  Prints ms per step and M solves/s; JSON on the last line."""
import ctypes as C, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VARIANTS = (("coop", "coop", None), ("coop_split_2", "coop_split", "2"), ("coop_split_3", "coop_split", "3"))


def child(kernel, wl, nb, steps):
    import numpy as np, torch
    from fcc_qp_amd import problems as P
    from tests import capi as CAPI
    B = P.workload(wl, nb, P.LARGE, distinct=min(nb, 32))
    s = CAPI.Solver(*P.LARGE.ctor_args, batch=nb); s.set_options(**P.HARNESS_OPTIONS); s.set_kernel_timing(False)
    s.set_kernel({"coop": CAPI.KERNEL_COOP, "coop_split": CAPI.KERNEL_COOP_SPLIT, "auto": CAPI.KERNEL_AUTO}[kernel])
    dev = [torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda() for a in B.arrays()]
    dp = C.POINTER(C.c_double)
    ptr = [C.cast(t.data_ptr(), dp) for t in dev]
    L = s.L
    L.fccqp_solve_device.argtypes = [C.c_void_p] + [dp] * 7 + [C.c_void_p]
    own = C.c_void_p(-1)                                  # FCCQP_STREAM_OWN

    def run(k):
        for _ in range(k):
            rc = L.fccqp_solve_device(s.h, *ptr, own)
            assert rc == 0, (rc, s.last_error())
        assert L.fccqp_synchronize(s.h) == 0
    run(30)
    out = []
    for _ in range(5):
        t0 = time.perf_counter(); run(steps); out.append((time.perf_counter() - t0) / steps * 1e3)
    z, d = s.solution()
    print("RESULT " + json.dumps({"median": float(np.median(out)), "min": min(out), "kernel": s.kernel(),
                                  "checksum": float(np.sum(z)), "n_iter_mean": float(np.mean([x.n_iter for x in d]))}))


if __name__ == "__main__":
    if sys.argv[1:2] == ["--child"]:
        child(sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]))
    else:
        nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
        steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
        summary = {"batch": nb, "steps": steps}
        for wl in ("W-sat", "W-mix"):
            res = {v[0]: [] for v in VARIANTS}
            sums = {}
            for rep in range(3):
                for name, kernel, wgs in VARIANTS:
                    env = dict(os.environ)
                    if wgs: env["FCCQP_COOP_SPLIT_WGS"] = wgs
                    txt = subprocess.check_output([sys.executable, os.path.abspath(__file__), "--child", kernel, wl, str(nb), str(steps)], env=env, text=True)
                    r = json.loads([x for x in txt.splitlines() if x.startswith("RESULT ")][-1][7:])
                    res[name].append(r["median"]); sums[name] = (r["checksum"], r["n_iter_mean"], r["kernel"])
            assert len({v[0] for v in sums.values()}) == 1, ("results differ between the forms", sums)
            summary[wl] = {k: {"ms_per_step": v, "Msolves_per_s": [nb / x / 1e3 for x in v]} for k, v in res.items()}
            print(wl, {k: ["%.4f" % x for x in v] for k, v in res.items()}, "M solves/s", {k: "%.3f" % (nb / min(v) / 1e3) for k, v in res.items()}, flush=True)
        print(json.dumps(summary))
