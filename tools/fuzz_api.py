"""Randomised run of the C ABI's call sequences (test infrastructure): one handle, a random chain of option changes,
warm / cold switches, kernel requests, state transplants and solves through every entry point -- host pointers
(synchronous and asynchronous), device pointers on the solver's stream and on other streams, f64 and f32 storage --
with results fetched by GetSolution, by the asynchronous copy, or not at all before the next solve; mirrored step by
step by one oracle object per instance (which carries the warm-start state the way the reference's class does).
    python tools/fuzz_api.py [seconds] [seed]
What this is after: ordering between solves on different streams, the page-locked mirrors and deferred fallback of the
small synchronous path, stale results after a state transplant, warm state surviving kernel changes.  Exit code 1 if a
solve's z or n_iter leaves the contract (rel err <= 1e-6, |dn_iter| <= 1)."""
import ctypes as C
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fcc_qp_amd import problems as P
from oracle import oracle as O
from tests import capi as CAPI
from tools.fuzz_parity import FUSED, COOP, random_shape, random_batch

_dp, _fp = C.POINTER(C.c_double), C.POINTER(C.c_float)


def oracle_options(o, opts):
    oo = O.FCCQPOptions()
    oo.max_iter, oo.rho, oo.eps_fcone, oo.eps_bound = opts["max_iter"], opts["rho"], opts["eps_fcone"], opts["eps_bound"]
    o.set_options(oo)


def main(budget=120.0, seed=1, max_chains=None):
    rng = np.random.default_rng(seed)
    L = CAPI.lib()
    L.fccqp_solve_async.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp, C.c_int, _dp, _dp]
    L.fccqp_get_z_async.argtypes = [C.c_void_p, _dp]
    t_end = time.time() + budget
    chains = solves = bad = 0
    worst = 0.0
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    t_print = time.time()
    while time.time() < t_end and (max_chains is None or chains < max_chains):
        chains += 1
        r = rng.random()
        dims = FUSED[int(rng.integers(len(FUSED)))] if r < 0.4 else (COOP[int(rng.integers(len(COOP)))] if r < 0.55 else random_shape(rng))
        if dims[0] + dims[1] > 200: continue
        n, m, nc, s0 = dims
        nb = int(rng.choice([1, 1, 2, 3, 5, 9]))
        s = CAPI.Solver(*dims, batch=nb)
        orc = [O.OracleFCCQP(*dims) for _ in range(nb)]
        opts = dict(max_iter=40, rho=1e-3, eps_fcone=1e-6, eps_bound=1e-6)
        s.set_options(**opts)
        for o in orc: oracle_options(o, opts)
        warm = False
        log = [f"chain {chains} dims {dims} batch {nb}"]
        keep = []                                             # device / pinned buffers of solves that may still be running
        pending_fetch = None
        for step in range(int(rng.integers(2, 9))):
            a = rng.random()
            if a < 0.2:
                opts = dict(max_iter=int(rng.choice([0, 1, 3, 15, 40, 120])), rho=float(10.0 ** rng.uniform(-5, -1)),
                            eps_fcone=float(10.0 ** rng.uniform(-8, -3)), eps_bound=float(10.0 ** rng.uniform(-8, -3)))
                s.set_options(**opts)
                for o in orc: oracle_options(o, opts)
                log.append(f"options {opts}")
            if rng.random() < 0.4:
                warm = bool(rng.random() < 0.6)
                s.set_warm_start(warm)
                for o in orc: o.set_warm_start(warm)
                log.append(f"warm {warm}")
            if rng.random() < 0.25:
                kernels = [CAPI.KERNEL_AUTO, CAPI.KERNEL_GENERIC]
                if dims in FUSED: kernels += [CAPI.KERNEL_FUSED, CAPI.KERNEL_FUSED_SPLIT]
                if dims in COOP: kernels += [CAPI.KERNEL_COOP]
                k = kernels[int(rng.integers(len(kernels)))]
                s.set_kernel(k); log.append(f"kernel {k}")
            if rng.random() < 0.12:
                # state transplant: read the state, scale the duals, write it back -- on both sides
                x, mx, ml = s.state()
                mx2, ml2 = 0.5 * mx, 0.5 * ml
                mlw = np.zeros((nb, max(nc, 1))); mlw[:, :nc] = ml2
                assert L.fccqp_set_state(s.h, CAPI._p(np.ascontiguousarray(x)), CAPI._p(np.ascontiguousarray(mx2)), CAPI._p(mlw)) == 0
                for i, o in enumerate(orc):
                    ox, omx, oml = o.get_state()
                    o.set_state(ox, 0.5 * omx, 0.5 * oml)
                log.append("state transplant")
            B = random_batch(rng, dims, nb, singular=False)
            f32 = rng.random() < 0.15
            if f32: B = P.Batch(*(np.asarray(np.asarray(x, dtype=np.float32), dtype=np.float64) for x in B.arrays()))
            arrs = [np.ascontiguousarray(x) for x in B.arrays()]
            how = int(rng.integers(0, 5))
            log.append(f"solve via {['host', 'host-async', 'device-own', 'device-stream0', 'device-stream1'][how]}" + (" f32" if f32 else ""))
            nf = nc // 3
            if how == 0 or (how == 1 and f32):
                rc = s.solve_f32(*arrs) if f32 else s.solve(*arrs)
            elif how == 1:
                pin = [torch.from_numpy(x).pin_memory() for x in arrs]
                keep.append(pin)
                ptr = [C.cast(t.data_ptr(), _dp) for t in pin]
                rc = L.fccqp_solve_async(s.h, ptr[0], ptr[1], ptr[2], ptr[3], ptr[4], nf, ptr[5], ptr[6])
            else:
                dt = torch.float32 if f32 else torch.float64
                st = None if how == 2 else streams[how - 3]
                with torch.cuda.stream(st) if st is not None else torch.cuda.stream(torch.cuda.current_stream()):
                    dev = [torch.from_numpy(x).to("cuda", dtype=dt, non_blocking=False) for x in arrs]
                if st is not None: st.wait_stream(torch.cuda.current_stream())
                keep.append(dev)
                ptr = [C.c_void_p(t.data_ptr()) for t in dev]
                sp = C.c_void_p(-1) if st is None else C.c_void_p(st.cuda_stream)
                rc = (L.fccqp_solve_device_f32 if f32 else L.fccqp_solve_device)(s.h, *ptr, sp)
            if rc != 0:
                print("ERROR", " | ".join(log), rc, L.fccqp_last_error(s.h).decode(), flush=True); bad += 1; break
            solves += 1
            for i, o in enumerate(orc): o.Solve(*[x[i] for x in arrs])
            fetch = rng.random()
            if fetch < 0.25 and step + 1 < 8:
                log.append("(not fetched)"); continue       # straight into the next solve
            if fetch < 0.5:
                zp = torch.empty((nb, n), dtype=torch.float64).pin_memory()
                assert L.fccqp_get_z_async(s.h, C.cast(zp.data_ptr(), _dp)) == 0
                assert L.fccqp_synchronize(s.h) == 0
                z = zp.numpy().copy(); it = None
                log.append("fetched by async copy")
            else:
                z, d = s.solution(); it = np.array([x.n_iter for x in d])
                log.append("fetched by GetSolution")
            for i, o in enumerate(orc):
                sol = o.GetSolution()
                e = float(np.max(np.abs(z[i] - sol.z)) / (1.0 + np.max(np.abs(sol.z)))) if np.all(np.isfinite(z[i])) else float("inf")
                dn = 0 if it is None else abs(int(it[i]) - int(sol.details.n_iter))
                worst = max(worst, e if np.isfinite(e) else 0.0)
                if not (e <= 1e-6) or dn > 1:
                    bad += 1
                    print("BROKEN", " | ".join(log), f"instance {i} rel err {e:.2e} dn_iter {dn}", flush=True)
                    break
        assert L.fccqp_synchronize(s.h) == 0
        torch.cuda.synchronize()
        s.close()
        if time.time() - t_print > 50:
            print(f"... {chains} chains, {solves} solves, {bad} broken, worst rel err {worst:.2e}", flush=True); t_print = time.time()
    print(f"fuzz_api: {chains} chains, {solves} solves, {bad} broken, worst rel err {worst:.2e}", flush=True)
    if max_chains is not None and chains < max_chains and not bad:
        # (a run bounded by a case count must reach it: a budget that cuts it short is a failure, not a shorter test)
        print(f"fuzz: the time budget ended after {chains} of {max_chains} cases", flush=True)
        return 2
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1))
