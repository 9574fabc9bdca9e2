"""Same-box A/B of the block -> instance order of the one-wave kernel (fccqp_set_dispatch_order) on chains of warm-started
solves: 4096 chains walk through a synthetic gait (problems.gait_sequence: pushes ramp 0 -> 4 -> 0 every 400 QPs) in lock
step, every chain at its own phase, so that a batch mixes instances that leave after a few iterations with instances that
run all of them -- and the previous step's iteration count says which is which.  Per variant: a cold solve of step 0, then
T warm steps, each between HIP events on the launch stream; R repetitions of the whole chain, variants alternating.
    python tools/order_ab.py [T] [R]
Also W-mix (cold, the same inputs every step: the previous solve's counts are a perfect forecast -- an upper bound, not a
use case).  Checks that the results of the two orders are bit-identical.  JSON on the last line."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fcc_qp_amd import problems as P
from tests import capi as CAPI

NB, PHASES = 4096, 256


def to_dev(batch):
    return [torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda() for a in batch.arrays()]


def run_chain(steps_dev, order, options, reps):
    """ms per warm step (mean over steps and repetitions), n_iter histogram of the last step, z of the last step"""
    dp = C.POINTER(C.c_double)
    s = CAPI.Solver(*P.WALKING.ctor_args, batch=NB); s.set_options(**options); s.set_kernel_timing(False)
    s.set_dispatch_order(order)
    L = s.L
    L.fccqp_solve_device.argtypes = [C.c_void_p] + [dp] * 7 + [C.c_void_p]
    st = torch.cuda.current_stream()
    sp = C.c_void_p(st.cuda_stream)
    ms = []
    for rep in range(reps + 1):
        for t, dev in enumerate(steps_dev):
            s.set_warm_start(t > 0)
            ptr = [C.cast(x.data_ptr(), dp) for x in dev]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            assert L.fccqp_solve_device(s.h, *ptr, sp) == 0, s.last_error()
            e1.record(st)
            torch.cuda.synchronize()
            if t > 0 and rep > 0:
                ms.append(e0.elapsed_time(e1))
    z, d = s.solution()
    it = np.array([x.n_iter for x in d])
    s.close()
    return float(np.mean(ms)), it, z


def main(T=4, R=3):
    seq = P.gait_sequence(400 + T + 1)
    offs = [(c * 400) // PHASES for c in range(PHASES)]
    out = {"batch": NB, "warm_steps": T, "repetitions": R}
    for name, options in (("harness options (max_iter 100, eps 1e-6)", P.HARNESS_OPTIONS), ("paper options (max_iter 15, eps 1e-4)", P.PAPER_OPTIONS)):
        steps = []
        for t in range(T + 1):
            base = P.stack([seq[o + t] for o in offs])
            reps = NB // PHASES
            steps.append(to_dev(P.Batch(*(np.ascontiguousarray(np.tile(a, (reps,) + (1,) * (a.ndim - 1))) for a in base.arrays()))))
        res = {}
        for rnd in range(2):
            for oname, order in (("hash", CAPI.ORDER_HASH), ("previous_iterations", CAPI.ORDER_PREVIOUS_ITERATIONS)):
                m, it, z = run_chain(steps, order, options, R)
                res.setdefault(oname, {"ms_per_step": []})["ms_per_step"].append(m)
                res[oname]["z"] = z; res[oname]["it"] = it
        assert np.array_equal(res["hash"]["z"], res["previous_iterations"]["z"]) and np.array_equal(res["hash"]["it"], res["previous_iterations"]["it"])
        it = res["hash"]["it"]
        hist = {int(k): int(v) for k, v in zip(*np.unique(np.minimum(it, 100) // 10 * 10, return_counts=True))}
        a, b = min(res["hash"]["ms_per_step"]), min(res["previous_iterations"]["ms_per_step"])
        out[name] = {"hash_ms": res["hash"]["ms_per_step"], "longest_first_ms": res["previous_iterations"]["ms_per_step"],
                     "gain": 1.0 - b / a, "n_iter_hist_by_tens_last_step": hist, "n_iter_mean": float(it.mean())}
        print(name, "hash", ["%.4f" % x for x in res["hash"]["ms_per_step"]], "longest first", ["%.4f" % x for x in res["previous_iterations"]["ms_per_step"]],
              "gain %.1f %%" % (100 * (1 - b / a)), "n_iter by tens", hist, flush=True)
        del steps
    # W-mix, cold, the same inputs every step
    dev = to_dev(P.workload("W-mix", NB, P.WALKING, distinct=256))
    res = {}
    for rnd in range(2):
        for oname, order in (("hash", CAPI.ORDER_HASH), ("previous_iterations", CAPI.ORDER_PREVIOUS_ITERATIONS)):
            dp = C.POINTER(C.c_double)
            s = CAPI.Solver(*P.WALKING.ctor_args, batch=NB); s.set_options(**P.HARNESS_OPTIONS); s.set_kernel_timing(False)
            s.set_dispatch_order(order)
            s.L.fccqp_solve_device.argtypes = [C.c_void_p] + [dp] * 7 + [C.c_void_p]
            st = torch.cuda.current_stream(); sp = C.c_void_p(st.cuda_stream)
            ptr = [C.cast(x.data_ptr(), dp) for x in dev]
            for _ in range(30):
                assert s.L.fccqp_solve_device(s.h, *ptr, sp) == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(40):
                assert s.L.fccqp_solve_device(s.h, *ptr, sp) == 0
            e1.record(st); torch.cuda.synchronize()
            res.setdefault(oname, []).append(e0.elapsed_time(e1) / 40)
            z, _ = s.solution(); res[oname + "_z"] = z
            s.close()
    assert np.array_equal(res["hash_z"], res["previous_iterations_z"])
    a, b = min(res["hash"]), min(res["previous_iterations"])
    out["W-mix cold, same inputs every step (perfect forecast)"] = {"hash_ms": res["hash"], "longest_first_ms": res["previous_iterations"], "gain": 1 - b / a}
    print("W-mix cold", "hash", ["%.4f" % x for x in res["hash"]], "longest first", ["%.4f" % x for x in res["previous_iterations"]], "gain %.1f %%" % (100 * (1 - b / a)), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 4, int(sys.argv[2]) if len(sys.argv) > 2 else 3)
