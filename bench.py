#!/usr/bin/env python3
"""bench.py -- QP solves/sec at batch = 4096, walking-log shape, on N MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch: every rank solves its own 4096 independent
cold-start QPs (workload W-sat of fcc_qp_amd/problems.py: FCCQP(60, 38, 12, 38) instances that run
the full max_iter = 100 ADMM iterations under the reference harness options, fcc_qp_test.py:78-83)
with one launch of the fused HIP kernel through the C ABI (fccqp_solve_device, include/fccqp.h).
Inputs are resident in HBM before the timed region starts; z and the details stay on the device.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

The batch shards embarrassingly: one process per GPU, no data-path collective (torch.distributed
over RCCL is used only for the barriers and the MAX over ranks of the elapsed time).  Scaling is
weak: every GPU gets its own 4096 instances, `value` counts all of them.

Rank 0 prints ONE JSON line with the driver's fields plus
  "roofline":     algorithmic HBM bytes per launch / mean kernel time (HIP events on the launch
                  stream) against the 8 TB/s HBM3E peak -- the contract figure; this path is bound by the
                  fp64 unit (matrix cores + DPP-fed vector FMAs) and instruction issue, not by bandwidth
                  (DESIGN.md section 4), so the fraction is small;
  "cpu_baseline": the CPU restatement of the reference algorithm (oracle/, kind "port": the
                  reference itself needs Eigen, which is not in the image) on the host cores.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is achievable
BATCH_PER_GPU = 4096
SETTLE = True           # untimed steps until the clocks are steady, in front of every timed region (--no-settle: off)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="instances per GPU (metric: 4096)")
    ap.add_argument("--workload", default="W-sat", choices=["W-sat", "W-mix", "W-easy", "W-psd"])
    ap.add_argument("--shape", default="walking", choices=["walking", "large"],
                    help="walking = FCCQP(60,38,12,38), the metric's shape; large = FCCQP(114,60,24,66), "
                         "BASELINE configs[3] (secondary: cooperative kernel, eight waves per QP)")
    ap.add_argument("--kernel", default="auto", choices=["auto", "generic", "fused", "fused_split", "coop", "coop_split"])
    ap.add_argument("--options", default="harness", choices=["harness", "paper"],
                    help="harness: fcc_qp_test.py:78-83 (headline); paper: fccqp.pdf Table 1 (max_iter=15, eps=1e-4)")
    ap.add_argument("--warm", action="store_true",
                    help="secondary mode: time warm-started solves (every step continues from the previous one)")
    ap.add_argument("--inputs", default="f64", choices=["f64", "f32"],
                    help="storage type of the resident inputs; f32 = BASELINE configs[2]'s fp32-storage "
                         "path (widened on load, arithmetic stays fp64; the batch is the float-rounded problem)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every GPU gets --batch instances (default); strong: --batch-global instances are "
                         "split over the GPUs, rank g taking [g B / G, (g + 1) B / G)  (SURVEY.md 8e)")
    ap.add_argument("--batch-global", type=int, default=BATCH_PER_GPU,
                    help="global batch for --scaling strong (4096 = the metric's batch; 65536 = BASELINE configs[4])")
    ap.add_argument("--single-input-set", action="store_true",
                    help="re-solve ONE resident batch every step (rounds 1-2 did; its inputs then partly sit in the "
                         "Infinity Cache); default: alternate between two resident copies")
    ap.add_argument("--no-settle", action="store_true",
                    help="skip the untimed clock-settling steps in front of the W warm-up steps (see _settle)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-gate", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary measurements appended to the JSON line (W-psd, W-mix, paper/warm, "
                         "fp32 storage, large shape, end-to-end from pinned host memory); they run on 1 GPU only")
    return ap.parse_args()


FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X vector fp64 (MI355X_MICROARCH.md)
KERNEL_NAMES = {1: "generic", 2: "fused", 3: "fused_split", 4: "coop", 5: "coop_split"}


def fused_mfma_count(n, m):
    """v_mfma_f64_16x16x4 instructions the one-wave fused kernel executes per QP: (K_rho factorisation + forward
    solve, Gram product, factorisation of W) -- the loop bounds of tile_factor_forward / gram_output /
    tile_factor_store in fcc_qp_amd/csrc/fccqp_fused_impl.h restated (walking shape: 456 + 196 + 64 = 716, the count
    in the shipped code object, not counting the second pass of a diagonal tile's column, which holds none)."""
    N = n + m
    border = N % 16 if (N % 16 != 0 and N - N % 16 >= n and (n % 16) + (N % 16) <= 15 and N % 16 <= 3) else 0
    N1 = N - border
    NRB, NCB, cbd, NW = (N1 + 15) // 16, 4, n // 16, (n + 15) // 16
    zero_tile = lambda K, cb: K < cb and cb != cbd
    kmin = lambda cb: 0 if cb == cbd else cb
    f = g = w = 0
    for P in range(NRB):
        ks = sum(1 for r in range(4) if 16 * P + 4 * r < N1)
        live = sum(1 for cb in range(NCB) if not zero_tile(P, cb))
        # (the tile of unit columns 16 P .. 16 P + 15 is still the identity when panel P solves it: L^-1 is read from LDS)
        solved = live - (1 if (P < NCB and P != cbd) else 0)
        f += ks * ((NRB - P - 1) + solved)
        for I in range(P + 1, NRB):
            f += ks * ((NRB - I) + live)
    for I in range(cbd + 1 if border else NW):
        for K in range(NRB):
            if K >= kmin(I):
                ks = sum(1 for r in range(4) if 4 * (4 * K + r) < N1)
                g += ks * sum(1 for J in range(I, cbd + 1) if K >= kmin(J))
    for P in range(NW):
        ks = sum(1 for r in range(4) if 16 * P + 4 * r < n)
        w += ks * (NW - P - 1) + sum(4 * (NW - I) for I in range(P + 1, NW))
    return f, g, w


def fp64_diag(shape, warm, n_iter_mean, at_max_frac, nb, kernel_ms, fused=True):
    """Two flop counts per launch against the 78.6 TFLOP/s vector-fp64 peak:
    `frac`           the REFERENCE algorithm's flops: f factorisations of N^3/3, p presolve
                     back-solves and K iterations of 2 N^2 each -- what a CPU running fcc_qp.cpp would execute;
    `frac_executed`  what the fused kernel executes: its MFMAs (1024 FMAs each; one N^3/3-class factorisation with the
                     forward solve of n + 1 right-hand sides fused in, the Gram product, and -- cold starts -- the
                     n x n factorisation of W instead of a second N^3/3) plus the K mat-vecs of n DPP-fed wave FMAs."""
    N = shape.n + shape.m
    f, p = (1, 0) if warm else (2, 1)
    k_exec = n_iter_mean + (1.0 - at_max_frac)        # a converged instance executed n_iter + 1 iterations
    flops = nb * (f * N ** 3 / 3.0 + (p + k_exec) * 2.0 * N * N)
    ach = flops / (kernel_ms * 1e-3) / 1e12
    out = {"flops_per_launch": flops, "achieved": ach, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
           "frac": ach / FP64_VECTOR_PEAK_TFLOPS, "counts": "reference algorithm (two N^3/3 on a cold start)"}
    if fused and shape.n < 64:
        # (since round 4 a cold start solves the presolve's n x n system by ~12 conjugate-gradient mat-vecs on rho G
        # instead of factorising W: the `mw` MFMAs of that factorisation only run where CG declines)
        mf, mg, mw = fused_mfma_count(shape.n, shape.m)
        mfma = mf + mg
        cg_matvecs = 0.0 if warm else 12.0
        ex = nb * (mfma * 2048.0 + (k_exec + cg_matvecs) * shape.n * 64 * 2.0)
        out.update(flops_executed_per_launch=ex, mfma_per_solve=mfma, cg_matvecs_per_cold_solve_estimate=cg_matvecs,
                   achieved_executed=ex / (kernel_ms * 1e-3) / 1e12,
                   frac_executed=ex / (kernel_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS)
    return out


def cpu_baseline(shape, options, batch_obj, log):
    """Oracle (CPU restatement of the reference algorithm) on the host cores: a bounded sample of the
    same workload.  Returns the cpu_baseline JSON object.  This is the checker being TIMED as the
    baseline, never part of the product path."""
    from oracle import oracle as O

    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # a one-GPU box's CPU share is 16 cores; FCCQP_BENCH_CPU_THREADS overrides
    cores = max(1, min(avail, int(os.environ.get("FCCQP_BENCH_CPU_THREADS", "16"))))
    # -O3 -march=native build of the same C file (the parity build keeps -ffp-contract=off)
    fast = os.path.join(ROOT, "oracle", "libfccqp_oracle_native.so")
    lib = None
    try:
        src = os.path.join(ROOT, "oracle", "fccqp_oracle.c")
        if (not os.path.exists(fast)) or os.path.getmtime(fast) < os.path.getmtime(src):
            subprocess.check_call(["gcc", "-O3", "-march=native", "-DNDEBUG", "-fPIC", "-std=c99",
                                   "-D_POSIX_C_SOURCE=200809L", "-shared", "-o", fast, src, "-lm", "-lpthread"])
        lib = ctypes.CDLL(fast)
        lib.fccqp_oracle_solve_batch.restype = ctypes.c_double
        lib.fccqp_oracle_solve_batch.argtypes = O.lib().fccqp_oracle_solve_batch.argtypes
    except Exception as exc:  # fall back to the parity build of the oracle
        log(f"[bench] native oracle build failed ({exc}); timing the -O3 parity build")
        lib = O.lib()

    import numpy as np

    n, m, nc, s = shape.ctor_args
    sample = min(batch_obj.batch, 4096)
    sl = batch_obj.slice(0, sample)
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in sl.arrays()]
    ptr = [a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)) for a in arrs]
    z = np.zeros((sample, n))
    it = np.zeros(sample, dtype=np.int32)

    def run(nthreads, count):
        return lib.fccqp_oracle_solve_batch(
            n, m, nc, s, int(options["max_iter"]), float(options["rho"]), float(options["eps_fcone"]),
            float(options["eps_bound"]), count, *ptr, z.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
            it.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), None, nthreads)

    # one thread: comparable with the reference's single-threaded use (fccqp.pdf Fig. 1)
    one = min(sample, 256)
    t1 = min(run(1, one) for _ in range(2))
    # all cores: one solver instance per thread over a static partition of the sample
    tn = min(run(cores, sample) for _ in range(3))
    return {
        "value": sample / tn, "unit": "solves/s", "cores": cores, "kind": "port",
        "sample": f"first {sample} instances of the same batch, best of 3 passes, {cores} threads "
                  f"(one oracle solver per thread, gcc -O3 -march=native)",
        "single_thread_value": one / t1, "single_thread_us_per_solve": 1e6 * t1 / one,
    }


def _settle(torch, step, enabled=True, block=10, tol=0.005, max_blocks=40, host_clock=False):
    """Untimed steps until the GPU has reached its steady clocks: a card that has idled for the seconds the CPU parts of
    this script take (problem generation, the oracle) starts ~6 % slow and needs ~25 launches (7 ms) to ramp up -- measured:
    steps 6-25 after an idle phase 0.282 ms, steps 51-70 0.264 ms, same kernel, same inputs.  Blocks of `block` steps,
    each between one event pair; done when two successive blocks agree with their predecessor to `tol`, or after
    `max_blocks`.  Returns the number of steps run (reported as `settle_steps`).  Not part of the contract's W warm-up
    steps, which follow; nothing here is timed into `value`.  `host_clock`: time a block on the host (perf_counter
    around the block and a device synchronize) -- for steps that launch on streams of their own, where an event pair
    recorded on torch's current stream brackets nothing."""
    _settle.trace = []                   # ms per step of every block of the last call (reported for the headline)
    if not enabled:
        return 0
    prev, stable, n = None, 0, 0
    for _ in range(max_blocks):
        if host_clock:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(block):
                step()
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0)
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(block):
                step()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
        n += block
        _settle.trace.append(ms / block)
        stable = stable + 1 if (prev is not None and abs(ms - prev) <= tol * ms) else 0
        prev = ms
        if stable >= 2:
            break
    return n


def _timed_steps(torch, step, steps, warmup):
    """W untimed + K timed steps of `step` on torch's current stream, back to back, ONE HIP event pair around the
    K steps (an event between two steps makes the second wait for the stream to drain: ~5 us of stream time each,
    DESIGN.md section 4.3); returns (wall seconds, ms per launch = event span / K as a one-element list)."""
    for _ in range(warmup):
        step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, [e0.elapsed_time(e1) / steps]


def _isolated_launch_ms(torch, step, steps):
    """K more steps, each between its own HIP event pair (the stream drains in front of every step): the duration of
    ONE launch with nothing in front of it -- what a rocprofv3 kernel trace of such a run reports per dispatch."""
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    torch.cuda.synchronize()
    for e0, e1 in ev:
        e0.record(); step(); e1.record()
    torch.cuda.synchronize()
    return [e0.elapsed_time(e1) for e0, e1 in ev]


def _parity(O, shape, options, batch, z, n_iter, k):
    import numpy as np
    z_ref, it_ref, _, _ = O.solve_batch(shape.ctor_args, options, *batch.slice(0, k).arrays(), nthreads=8)
    err = np.max(np.abs(z[:k] - z_ref), axis=1) / (1.0 + np.max(np.abs(z_ref), axis=1))
    dn = np.abs(np.asarray(n_iter[:k]).astype(np.int64) - it_ref.astype(np.int64))
    return {"sample": int(k), "max_rel_err": float(err.max()), "max_dn_iter": int(dn.max())}


def secondary_measurements(F, P, O, torch, dev, dev_index, log, steps=20, warmup=3):
    """The numbers beside the headline that DESIGN.md quotes, measured in the same process so that
    the driver's BENCH json carries them (`extra_keys`): each is {value [solves/s], ms_per_step,
    kernel_ms_mean, roofline_frac, parity vs the oracle on a sample, ...}.  1 GPU, batch 4096."""
    import numpy as np
    out = {}
    nb = BATCH_PER_GPU

    def run(name, shape, wl, options, kernel=None, warm=False, f32=False, distinct=256, parity_k=64, seed0=1000,
            steps=steps, warmup=warmup):
        t0 = time.time()
        batch = P.workload(wl, nb, shape, seed0=seed0, distinct=distinct)
        if f32:
            batch = P.Batch(*(a.astype(np.float32).astype(np.float64) for a in batch.arrays()))
        solver = F.FCCQPBatch(*shape.ctor_args, nb, dev_index)
        solver.set_kernel_timing(False)       # timed here, with events of our own (see main())
        o = F.FCCQPOptions()
        o.max_iter, o.rho, o.eps_fcone, o.eps_bound = (options["max_iter"], options["rho"], options["eps_fcone"], options["eps_bound"])
        solver.set_options(o)
        if kernel is not None:
            solver.set_kernel(kernel)
        tens = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in batch.arrays()]
        if f32:
            tens = [t.float() for t in tens]
        step = lambda: F.solve_tensors(solver, *tens)
        step(); torch.cuda.synchronize()
        cold = solver.GetSolution()
        par = _parity(O, shape, options, batch, cold.z, cold.n_iter, parity_k)
        _settle(torch, step, SETTLE)          # (cold solves: a warm chain would converge further with every step)
        if warm:
            solver.set_warm_start(True)
        wall, kms = _timed_steps(torch, step, steps, warmup)
        sol = solver.GetSolution()
        it = np.asarray(sol.n_iter)
        ab = solver.AlgorithmicBytes()
        mean_ms = float(np.mean(kms))
        rec = {"value": nb * steps / wall, "unit": "solves/s", "ms_per_step": 1e3 * wall / steps, "kernel_ms_mean": mean_ms,
               "roofline_frac": ab / (mean_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "algorithmic_bytes_per_launch": ab,
               "kernel": KERNEL_NAMES[solver.kernel_in_use()], "parity": par,
               "n_iter_mean": float(it.mean()), "n_iter_at_max_frac": float(np.mean(it == options["max_iter"])),
               "workload": f"{wl} FCCQP{tuple(shape.ctor_args)} {'warm' if warm else 'cold'} max_iter={options['max_iter']}"
                           f"{' fp32 storage' if f32 else ''}, {distinct} distinct instances tiled to {nb}"}
        if par["max_rel_err"] > 1e-6 or par["max_dn_iter"] > 1:
            rec["parity_failed"] = True
        out[name] = rec
        log(f"[bench] {name}: {rec['value'] / 1e6:.3f} M solves/s, kernel {mean_ms:.3f} ms, parity {par} ({time.time() - t0:.1f} s)")
        return solver, batch, sol

    # semi-definite cost (no force regularisation): the reference's own presolve takes its COD branch here
    run("w_psd", P.WALKING, "W-psd", P.HARNESS_OPTIONS)
    # half the instances exit early: the n_iter histogram SURVEY.md 8d asks for
    _, _, sol = run("w_mix", P.WALKING, "W-mix", P.HARNESS_OPTIONS)
    it = np.asarray(sol.n_iter)
    edges = [0, 1, 2, 5, 10, 20, 50, 100, 101]
    out["w_mix"]["n_iter_hist"] = {f"[{edges[i]},{edges[i + 1]})": int(np.sum((it >= edges[i]) & (it < edges[i + 1]))) for i in range(len(edges) - 1)}
    # the paper's deployment: max_iter 15, eps 1e-4, warm start (fccqp.pdf Table 1)
    run("paper_warm", P.WALKING, "W-sat", P.PAPER_OPTIONS, warm=True)
    # BASELINE configs[2]: fp32 storage of the inputs, fp64 arithmetic
    run("f32_storage", P.WALKING, "W-sat", P.HARNESS_OPTIONS, f32=True)
    # the walking shape on the any-shape kernel
    run("generic_kernel", P.WALKING, "W-sat", P.HARNESS_OPTIONS, kernel=F.KERNEL_GENERIC)
    # the two-launch form of the fused kernel (ADMM loop at two waves per SIMD): the measured A/B of DESIGN.md section 7
    run("fused_split", P.WALKING, "W-sat", P.HARNESS_OPTIONS, kernel=F.KERNEL_FUSED_SPLIT)
    run("fused_split_w_mix", P.WALKING, "W-mix", P.HARNESS_OPTIONS, kernel=F.KERNEL_FUSED_SPLIT)
    # BASELINE configs[3]: 36-DoF / 8-contact shape -- cooperative kernel (eight waves per QP share the register
    # tiles; at this batch size AUTO takes its two-launch form: factorisation + Gram product at one QP per CU, conjugate
    # gradients + ADMM iterations at three), the same in one launch, and the generic kernel it replaced for this shape
    run("large_shape", P.LARGE, "W-sat", P.HARNESS_OPTIONS, distinct=32, parity_k=16, steps=8, warmup=2)
    run("large_shape_one_launch", P.LARGE, "W-sat", P.HARNESS_OPTIONS, kernel=F.KERNEL_COOP, distinct=32, parity_k=16, steps=8, warmup=2)
    run("large_shape_generic", P.LARGE, "W-sat", P.HARNESS_OPTIONS, kernel=F.KERNEL_GENERIC, distinct=32, parity_k=16, steps=4, warmup=1)
    run("large_shape_w_mix", P.LARGE, "W-mix", P.HARNESS_OPTIONS, distinct=32, parity_k=16, steps=8, warmup=2)
    # a humanoid-sized shape no specialised kernel is compiled for: 44 DoF, 38 inputs, 12 contacts -> FCCQP(148, 74, 36, 82),
    # N = 222: the generic kernel with the KKT matrix in its per-block global workspace (198 < N <= 512)
    run("mid_shape", P.MID, "W-sat", P.HARNESS_OPTIONS, distinct=16, parity_k=8, steps=4, warmup=1)
    # a shape the library was NOT built for, FCCQP(47, 30, 9, 30): fccqp_create compiles the one-wave kernel for it at run
    # time (hiprtc in process, disk cache: fcc_qp_amd/csrc/fccqp_jit.cpp); beside it the generic kernel such a shape ran on
    # through round 4.  create_s: wall time of the handle's creation (first in this process: compile or disk cache).
    t_c = time.time()
    sv, _, _ = run("jit_shape", P.SMALL, "W-sat", P.HARNESS_OPTIONS, parity_k=32)
    out["jit_shape"]["create_plus_run_s"] = time.time() - t_c
    out["jit_shape"]["kernel_origin"] = {0: "builtin", 1: "jit", 2: "generic_only"}.get(sv.kernel_origin(), "?")
    del sv
    run("jit_shape_generic", P.SMALL, "W-sat", P.HARNESS_OPTIONS, kernel=F.KERNEL_GENERIC, parity_k=32, steps=8, warmup=2)
    out["two_streams"] = two_streams(F, P, torch, dev, dev_index, log)
    out["batch_sweep"] = batch_sweep(F, P, torch, dev, dev_index, log)
    out["single_instance"] = single_instance(F, P, log)
    out["end_to_end"] = end_to_end(P, O, dev_index, log)
    out["end_to_end_f32"] = end_to_end(P, O, dev_index, log, f32=True)
    return out


def two_streams(F, P, torch, dev, dev_index, log, steps=20, warmup=4):
    """The headline workload as a continuous feed: two solver handles on two HIP streams, steps alternating between
    them, so that the tail of one launch (the last of its four wave-rounds draining) overlaps the head of the next.
    Each handle owns its outputs and state, so the two launches share nothing.  NOT the headline: `value` times the
    steps of ONE handle in stream order."""
    import numpy as np
    nb = BATCH_PER_GPU
    batch = P.workload("W-sat", nb, P.WALKING, distinct=256)
    sets = [[torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in batch.arrays()] for _ in range(2)]
    solvers, streams = [], [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    for _ in range(2):
        sv = F.FCCQPBatch(*P.WALKING.ctor_args, nb, dev_index)
        sv.set_kernel_timing(False)
        o = F.FCCQPOptions()
        o.max_iter, o.rho, o.eps_fcone, o.eps_bound = (P.HARNESS_OPTIONS[k] for k in ("max_iter", "rho", "eps_fcone", "eps_bound"))
        sv.set_options(o)
        solvers.append(sv)

    def run(k):
        for i in range(k):
            F.solve_tensors(solvers[i & 1], *sets[i & 1], stream=streams[i & 1])

    _settle(torch, lambda: run(2), SETTLE, block=5, host_clock=True)     # (the launches are on the two side streams)
    run(warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    z0, z1 = (np.asarray(sv.GetSolution().z) for sv in solvers)
    rec = {"value": nb * steps / wall, "unit": "solves/s", "ms_per_step": 1e3 * wall / steps, "handles": 2, "streams": 2,
           "results_bit_identical_across_handles": bool(np.array_equal(z0, z1)),
           "workload": "W-sat FCCQP(60, 38, 12, 38) cold max_iter=100, 256 distinct instances tiled to 4096, steps "
                       "alternating between two handles on two streams"}
    log(f"[bench] two_streams: {rec['value'] / 1e6:.3f} M solves/s, {rec['ms_per_step']:.3f} ms per step")
    return rec


def batch_sweep(F, P, torch, dev, dev_index, log):
    """solves/s and kernel time of the headline workload (W-sat, walking shape, harness options, cold) at the batch
    sizes a reader needs to predict the multi-GPU curves from ONE GPU: 512 / 1024 / 2048 = the per-GPU share of the
    metric's batch 4096 split over 8 / 4 / 2 GPUs (--scaling strong), 8192 = one GPU's shard of BASELINE
    configs[4] (65536 over 8 GPUs) and 65536 = all of it on one GPU.  Inputs resident, 256 distinct instances tiled."""
    import numpy as np
    out = {}
    base = P.workload("W-sat", 256, P.WALKING)
    for nb in (512, 1024, 2048, 4096, 8192, 65536):
        reps = -(-nb // 256)
        tens = [torch.from_numpy(np.ascontiguousarray(a)).to(dev).repeat((reps,) + (1,) * (a.ndim - 1))[:nb].contiguous()
                for a in base.arrays()]
        solver = F.FCCQPBatch(*P.WALKING.ctor_args, nb, dev_index)
        solver.set_kernel_timing(False)
        o = F.FCCQPOptions()
        o.max_iter, o.rho, o.eps_fcone, o.eps_bound = (P.HARNESS_OPTIONS[k] for k in ("max_iter", "rho", "eps_fcone", "eps_bound"))
        solver.set_options(o)
        steps = 20 if nb <= 8192 else 5
        _settle(torch, lambda: F.solve_tensors(solver, *tens), SETTLE, block=(10 if nb <= 8192 else 2))
        wall, kms = _timed_steps(torch, lambda: F.solve_tensors(solver, *tens), steps, 2)
        sol = solver.GetSolution()
        z = np.asarray(sol.z)
        tiled_ok = bool(np.array_equal(z[:256], z[256:512])) if nb >= 512 else None
        out[str(nb)] = {"solves_per_s": nb * steps / wall, "kernel_ms_mean": float(np.mean(kms)), "ms_per_step": 1e3 * wall / steps,
                        "us_per_wave_round": 1e3 * float(np.mean(kms)) / max(1.0, nb / 1024.0), "tiled_copies_bit_identical": tiled_ok}
        log(f"[bench] batch_sweep {nb}: {out[str(nb)]['solves_per_s'] / 1e6:.3f} M solves/s, kernel {np.mean(kms):.3f} ms")
        del solver, tens
        torch.cuda.empty_cache()
    # what the strong-scaling split of the metric's batch can reach: 4096 / G instances per GPU, G GPUs side by side
    t1 = out["4096"]["kernel_ms_mean"]
    out["predicted_strong_scaling_efficiency_batch4096"] = {
        str(g): t1 / (g * out[str(4096 // g)]["kernel_ms_mean"]) for g in (2, 4, 8)}
    return out


def single_instance(F, P, log, length=60):
    """One QP at a time through the reference's own class surface (FCCQP.Solve + GetSolution, host arrays in and
    out: /root/reference/src/main.cpp:42-54, loop of /root/reference/fcc_qp_test.py:86-89): median wall microseconds
    per call over a synthetic gait sequence, cold with the harness options and warm-started with the paper's."""
    import numpy as np
    out = {}
    qps = P.gait_sequence(length, P.WALKING, seed0=5000)
    for name, opts, warm in (("cold_harness", P.HARNESS_OPTIONS, False), ("warm_paper", P.PAPER_OPTIONS, True)):
        s = F.FCCQP(*P.WALKING.ctor_args)
        o = F.FCCQPOptions()
        o.max_iter, o.rho, o.eps_fcone, o.eps_bound = (opts[k] for k in ("max_iter", "rho", "eps_fcone", "eps_bound"))
        s.set_options(o)
        w_solve, w_get, its, ktime = [], [], [], []
        for i, qp in enumerate(qps):
            s.set_warm_start(warm and i > 0)
            t0 = time.perf_counter()
            s.Solve(qp["Q"], qp["b"], qp["A_eq"], qp["b_eq"], qp["friction_coeffs"], qp["lb"], qp["ub"])
            t1 = time.perf_counter()
            r = s.GetSolution()
            t2 = time.perf_counter()
            w_solve.append(t1 - t0); w_get.append(t2 - t1); its.append(r.details.n_iter)
        ws, wg = 1e6 * np.array(w_solve[5:]), 1e6 * np.array(w_get[5:])
        out[name] = {"solve_us_median": float(np.median(ws)), "solve_us_p10": float(np.percentile(ws, 10)),
                     "solve_us_p90": float(np.percentile(ws, 90)), "get_solution_us_median": float(np.median(wg)),
                     "n_iter_mean": float(np.mean(its)), "calls": len(ws)}
        log(f"[bench] single_instance {name}: Solve {np.median(ws):.0f} us + GetSolution {np.median(wg):.0f} us, mean n_iter {np.mean(its):.1f}")
    return out


def end_to_end(P, O, dev_index, log, handles=3, rounds=12, f32=False):
    """Host memory in, host memory out (SURVEY.md 8d timing protocol): page-locked input and result
    buffers, `handles` solver handles round-robin so that the H2D copy of one batch, the kernel of
    another and the D2H copy of z of a third overlap (fccqp_solve_async / fccqp_get_z_async).  Per
    4096-batch 202 MB cross PCIe: this rate is bounded by the link, not by the kernel, and is never
    the headline `value`.  f32: the inputs are float32 arrays (BASELINE configs[2]: fp32 is a storage
    format -- widened on load, arithmetic and z in fp64) through fccqp_solve_async_f32: half the bytes."""
    import numpy as np
    from tests import capi as C
    L = C.lib()
    dp = ctypes.POINTER(ctypes.c_double)
    fp = ctypes.POINTER(ctypes.c_float)
    L.fccqp_solve_async.argtypes = [ctypes.c_void_p, dp, dp, dp, dp, dp, ctypes.c_int, dp, dp]
    L.fccqp_solve_async_f32.argtypes = [ctypes.c_void_p, fp, fp, fp, fp, fp, ctypes.c_int, fp, fp]
    L.fccqp_get_z_async.argtypes = [ctypes.c_void_p, dp]
    L.fccqp_host_register.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    L.fccqp_host_unregister.argtypes = [ctypes.c_void_p]
    nb, shape, options = BATCH_PER_GPU, P.WALKING, P.HARNESS_OPTIONS
    batch = P.workload("W-sat", nb, shape, distinct=256)
    if f32:
        ins = [[np.ascontiguousarray(a.astype(np.float32)) for a in batch.arrays()] for _ in range(handles)]
        batch = P.Batch(*(a.astype(np.float64) for a in ins[0]))       # the oracle sees the rounded values
    else:
        ins = [[np.ascontiguousarray(a.copy()) for a in batch.arrays()] for _ in range(handles)]
    zs = [np.zeros((nb, shape.n)) for _ in range(handles)]
    solvers = [C.Solver(*shape.ctor_args, batch=nb, device=dev_index) for _ in range(handles)]
    pinned = []
    for k in range(handles):
        solvers[k].set_options(**options)
        for a in ins[k] + [zs[k]]:
            if L.fccqp_host_register(a.ctypes.data, a.nbytes) != C.OK:
                raise RuntimeError("fccqp_host_register failed")
            pinned.append(a)
    nf = shape.nc // 3

    def submit(k):
        a = ins[k]
        if f32:
            q = [x.ctypes.data_as(fp) for x in a]
            rc = L.fccqp_solve_async_f32(solvers[k].h, q[0], q[1], q[2], q[3], q[4], nf, q[5], q[6])
        else:
            rc = L.fccqp_solve_async(solvers[k].h, C._p(a[0]), C._p(a[1]), C._p(a[2]), C._p(a[3]), C._p(a[4]), nf, C._p(a[5]), C._p(a[6]))
        rc = rc or L.fccqp_get_z_async(solvers[k].h, C._p(zs[k]))
        if rc != C.OK:
            raise RuntimeError(f"async solve failed: {rc}")

    for k in range(handles):                       # fill the pipeline (also the warm-up)
        submit(k)
    for k in range(handles):
        L.fccqp_synchronize(solvers[k].h)
    t0 = time.perf_counter()
    for k in range(handles):
        submit(k)
    for r in range(rounds):
        k = r % handles
        L.fccqp_synchronize(solvers[k].h)          # batch r is back in host memory ...
        if r + handles < rounds + handles:
            submit(k)                              # ... and its handle takes the next one
    for k in range(handles):
        L.fccqp_synchronize(solvers[k].h)
    wall = time.perf_counter() - t0
    done = rounds + handles
    z_ref, it_ref, _, _ = O.solve_batch(shape.ctor_args, options, *batch.slice(0, 64).arrays(), nthreads=8)
    err = max(float(np.max(np.abs(z[:64] - z_ref) / (1.0 + np.max(np.abs(z_ref), axis=1, keepdims=True)))) for z in zs)
    for a in pinned:
        L.fccqp_host_unregister(a.ctypes.data)
    bytes_per_batch = sum(a.nbytes for a in ins[0]) + zs[0].nbytes
    rec = {"value": nb * done / wall, "unit": "solves/s", "ms_per_batch": 1e3 * wall / done, "handles": handles,
           "pcie_gb_per_s": bytes_per_batch * done / wall / 1e9, "host_bytes_per_batch": bytes_per_batch,
           "parity": {"sample": 64, "max_rel_err": err},
           "what": "page-locked host buffers -> H2D + kernel + D2H of z, batches pipelined over the handles"}
    if f32:
        rec["what"] = "float32 inputs, " + rec["what"]
    log(f"[bench] end_to_end{'_f32' if f32 else ''}: {rec['value'] / 1e6:.3f} M solves/s, {rec['pcie_gb_per_s']:.1f} GB/s over PCIe, max rel err {err:.1e}")
    return rec


def sharded_global_batch(F, P, S, torch, dev, dev_index, rank, world, global_batch, options, args, timed_region, red_dev, log):
    """ONE global batch of W-sat walking-shape QPs (instance i: seed 1000 + i) split contiguously over the ranks,
    rank g solving [g B / G, (g + 1) B / G) on its GPU with one launch per step; no data-path collective.  Timed like
    the headline (settling steps, W warm-up steps, K steps between barriers, MAX over ranks); every rank checks the first
    instances of its own shard against the oracle and the worst rank's figures are reported.  Returns the record
    (identical on every rank)."""
    import numpy as np
    from oracle import oracle as O
    lo, hi, seed0, distinct = S.shard_plan(global_batch, world, rank)
    nbs = hi - lo
    # (generation costs ~1 ms per instance: shards above 2048 instances are tiled from their first 256)
    batch = P.workload("W-sat", nbs, P.WALKING, seed0=seed0, distinct=distinct)
    solver = F.FCCQPBatch(*P.WALKING.ctor_args, nbs, dev_index)
    solver.set_kernel_timing(False)
    o = F.FCCQPOptions()
    o.max_iter, o.rho, o.eps_fcone, o.eps_bound = (options["max_iter"], options["rho"], options["eps_fcone"], options["eps_bound"])
    solver.set_options(o)
    tens = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in batch.arrays()]
    step = lambda: F.solve_tensors(solver, *tens)
    step(); torch.cuda.synchronize()
    sol = solver.GetSolution()
    k = min(16, nbs)
    par = _parity(O, P.WALKING, options, batch, sol.z, sol.n_iter, k)
    worst_err = S.max_over_ranks(par["max_rel_err"], red_dev)
    worst_dn = int(S.max_over_ranks(float(par["max_dn_iter"]), red_dev))
    settle_steps = _settle(torch, step, SETTLE)
    elapsed, region_ms = timed_region(step, args.steps, args.warmup)
    rec = {"value": global_batch * args.steps / elapsed, "unit": "solves/s", "scaling": "strong", "global_batch": global_batch,
           "n_gpus": world, "shard_of_rank0": [lo, hi] if rank == 0 else None, "batch_per_gpu": global_batch // world,
           "ms_per_step": 1e3 * elapsed / args.steps, "kernel_ms_mean_rank0": region_ms / args.steps,
           "steps": args.steps, "warmup": args.warmup, "settle_steps_rank0": settle_steps,
           "parity": {"sample_per_rank": k, "max_rel_err_over_ranks": worst_err, "max_dn_iter_over_ranks": worst_dn},
           "workload": f"W-sat FCCQP(60, 38, 12, 38) cold max_iter={options['max_iter']}, instance i = seed 1000 + i, rank g "
                       f"owns [g B / G, (g + 1) B / G), no collective on the data path"}
    if worst_err > 1e-6 or worst_dn > 1:
        rec["parity_failed"] = True
    log(f"[bench] {global_batch} instances over {world} ranks: {rec['value'] / 1e6:.3f} M solves/s, {rec['ms_per_step']:.3f} ms per step, "
        f"parity over ranks {worst_err:.1e}")
    del solver, tens
    torch.cuda.empty_cache()
    return rec


def main():
    args = parse_args()
    global SETTLE
    SETTLE = not args.no_settle
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    log = (lambda *a: print(*a, file=sys.stderr, flush=True)) if rank == 0 else (lambda *a: None)

    import numpy as np
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # rehearsal aid: FCCQP_BENCH_ALL_ON_DEVICE0=1 lets several ranks share one GPU (1-GPU dev box)
    dev_index = 0 if os.environ.get("FCCQP_BENCH_ALL_ON_DEVICE0") == "1" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "gloo" if os.environ.get("FCCQP_BENCH_ALL_ON_DEVICE0") == "1" else "nccl"
        dist.init_process_group(backend=backend, **({"device_id": dev} if backend == "nccl" else {}))

    import fcc_qp_amd as F
    from fcc_qp_amd import problems as P
    from fcc_qp_amd import sharding as S

    shape = P.WALKING if args.shape == "walking" else P.LARGE
    options = dict(P.HARNESS_OPTIONS if args.options == "harness" else P.PAPER_OPTIONS)
    if args.scaling == "strong":
        # one global batch, instance i = seed 1000 + i, rank g owning [g B / G, (g + 1) B / G)
        lo, hi = S.shard_bounds(args.batch_global, world, rank)
        nb, seed0 = hi - lo, 1000 + lo
    else:
        nb, seed0 = args.batch, S.rank_seed0(rank, args.batch)
    t0 = time.time()
    # instance i: seed 1000 + i (the large shape's generator is slow: 64 distinct instances, tiled)
    batch = P.workload(args.workload, nb, shape, seed0=seed0,
                       **({"distinct": 64} if args.shape == "large" else {}))
    log(f"[bench] generated {nb} '{args.workload}' instances per rank in {time.time() - t0:.1f} s")

    solver = F.FCCQPBatch(*shape.ctor_args, nb, dev_index)
    # The library's own start-of-kernel event (fccqp_last_kernel_ms) is off: a hipEventRecord in front of the kernel
    # costs ~5 us of stream time per solve, and the kernel time of this run is measured below with HIP events of
    # the bench's own on the launch stream.
    solver.set_kernel_timing(False)
    o = F.FCCQPOptions()
    o.max_iter, o.rho, o.eps_fcone, o.eps_bound = (options["max_iter"], options["rho"],
                                                   options["eps_fcone"], options["eps_bound"])
    solver.set_options(o)
    solver.set_warm_start(False)
    solver.set_kernel({"auto": F.KERNEL_AUTO, "generic": F.KERNEL_GENERIC, "fused": F.KERNEL_FUSED,
                       "fused_split": F.KERNEL_FUSED_SPLIT, "coop": F.KERNEL_COOP, "coop_split": F.KERNEL_COOP_SPLIT}[args.kernel])
    if args.inputs == "f32":   # the problem IS the float-rounded one: oracle and CPU baseline see the same values
        batch = P.Batch(*(a.astype(np.float32).astype(np.float64) for a in batch.arrays()))
    tens = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in batch.arrays()]   # resident in HBM
    if args.inputs == "f32":
        tens = [t.float() for t in tens]
    # A second resident copy of the batch: the timed steps alternate between the two, so that a step's inputs are
    # not what the previous step left in the 256 MiB Infinity Cache (one walking-shape batch of 4096 is 202 MB; two
    # do not fit).  With a single resident set, PMC traffic read below the algorithmic bytes.
    sets = [tens, [t.clone() for t in tens]] if not args.single_input_set else [tens]
    torch.cuda.synchronize()
    step_no = [0]

    def step():
        F.solve_tensors(solver, *sets[step_no[0] % len(sets)])          # one launch on torch's current stream
        step_no[0] += 1

    # ---- parity gate: a sample against the CPU oracle --------------------
    step()
    torch.cuda.synchronize()
    sol = solver.GetSolution()
    n_iter = np.asarray(sol.n_iter)
    parity = None
    if rank == 0 and not args.no_parity_gate:
        from oracle import oracle as O
        k = min(256, nb)
        z_ref, it_ref, _, _ = O.solve_batch(shape.ctor_args, options, *batch.slice(0, k).arrays(), nthreads=8)
        err = np.max(np.abs(sol.z[:k] - z_ref), axis=1) / (1.0 + np.max(np.abs(z_ref), axis=1))
        dn = np.abs(n_iter[:k].astype(np.int64) - it_ref.astype(np.int64))
        parity = {"sample": k, "max_rel_err": float(err.max()), "max_dn_iter": int(dn.max())}
        if not (err.max() <= 1e-6 and dn.max() <= 1):
            raise SystemExit(f"parity gate failed: {parity}")
        log(f"[bench] parity vs oracle on {k} instances: {parity}")

    # ---- timed regions -----------------------------------------------------------------------------
    red_dev = dev if (world > 1 and dist.get_backend() == "nccl") else None

    def timed_region(step_fn, steps, warmup):
        """W untimed + exactly K timed steps of `step_fn`, barrier + synchronize on both sides; ONE HIP event pair
        around the K steps (on the launch stream): the steps run back to back, an event between two steps would make
        the second wait for the stream to drain.  Returns (wall seconds: MAX over ranks, HIP-event ms of the region)."""
        for _ in range(warmup):
            step_fn()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        S.barrier()
        torch.cuda.synchronize()
        t_start = time.perf_counter()
        ev0.record()
        for _ in range(steps):
            step_fn()
        ev1.record()
        torch.cuda.synchronize()
        S.barrier()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t_start
        return S.max_over_ranks(wall, red_dev), ev0.elapsed_time(ev1)

    # (a) cold clocks: the contract's protocol and nothing else -- W warm-up steps, K timed steps -- right behind the
    # CPU phases above (problem generation, the oracle's parity sample), i.e. on a card whose clocks have fallen.
    # Reported as `value_cold_clock`; rounds 1-3's `value` was this measurement.  (Not in the secondary --warm mode: a
    # warm chain converges further with every step, two regions would not time the same work.)
    elapsed_cold, region_ms_cold = (None, None) if args.warm else timed_region(step, args.steps, args.warmup)
    # (b) steady clocks: untimed settling steps until the step time is steady, then W warm-up + K timed steps = `value`
    settle_steps = _settle(torch, step, SETTLE)     # (cold solves)
    settle_trace = [round(v, 4) for v in _settle.trace]
    if settle_trace:
        log(f"[bench] settle: {settle_steps} untimed steps, ms per step by block of 10: {settle_trace}")
    if args.warm:
        solver.set_warm_start(True)     # state (x, mu) stays resident; the presolve is skipped (fcc_qp.cpp:159)
    elapsed, region_ms = timed_region(step, args.steps, args.warmup)
    kernel_ms = [region_ms / args.steps]                       # average launch duration inside the timed region
    # beside it (outside the timed region): every launch between its own event pair, the stream drained in front
    isolated_ms = _isolated_launch_ms(torch, step, min(args.steps, 20))

    solved = S.sum_over_ranks(float(nb * args.steps), red_dev)      # (elapsed: the slowest rank defines the step time)

    # ---- N > 1: the metric read the other way ----------------------------------------
    # `value` above is weak scaling (every GPU its own 4096).  The same invocation also times ONE global batch split
    # contiguously over the ranks -- 4096 (the metric's batch: 4096 / G per GPU, launch-bound at G = 8) and 65536
    # (BASELINE configs[4]: 8192 per GPU at G = 8) -- each in its own barrier-bracketed region, every rank checking a
    # sample of ITS shard against the oracle.
    sharded = {}
    if world > 1 and args.shape == "walking" and not args.warm and args.inputs == "f64" and os.environ.get("FCCQP_BENCH_NO_SHARDED") != "1":
        for label, gb in S.SHARDED_RECORDS:
            sharded[label] = sharded_global_batch(F, P, S, torch, dev, dev_index, rank, world, gb, options, args, timed_region,
                                                  red_dev, log)

    if args.warm:
        # iteration statistics of the TIMED (warm) solves, not of the cold parity solve
        n_iter = np.asarray(solver.GetSolution().n_iter)
    if rank == 0:
        value = solved / elapsed
        mean_ms = float(np.mean(kernel_ms))
        alg_bytes = solver.AlgorithmicBytes()                  # 8 * batch * (n^2 + mn + 4n + m + nc/3)
        achieved = alg_bytes / (mean_ms * 1e-3) / 1e9
        traffic = None
        tr_path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tr_path):
            try:
                tr = json.load(open(tr_path)).get(args.shape, {})     # rocprofv3 PMC passes, tools/parse_profile.py
                if tr.get("batch") == nb and tr.get("workload") == args.workload and tr.get("kernel") == solver.kernel_in_use() \
                        and args.inputs == "f64" and not args.warm and args.options == "harness":
                    traffic = tr.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        fp64 = fp64_diag(shape, args.warm, float(n_iter.mean()), float(np.mean(n_iter == options["max_iter"])),
                         nb, mean_ms, fused=solver.kernel_in_use() in (2, 3))
        out = {
            "metric": ("QP solves/sec at batch=4096, walking-log shape" if args.shape == "walking" else
                       "QP solves/sec, 36-DoF / 8-contact shape (BASELINE configs[3], secondary)"),
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            # the same K steps behind the same W warm-up steps WITHOUT the settling steps, timed first (cold clocks:
            # the card has idled through the CPU phases of this script); `value` is the steady-clock rate
            "value_cold_clock": (solved / elapsed_cold if elapsed_cold else None),
            "ms_per_step_cold_clock": (1e3 * elapsed_cold / args.steps if elapsed_cold else None),
            "config": {
                "workload": f"{args.workload}: FCCQP{tuple(shape.ctor_args)} {args.shape}-shape OSC QPs, "
                            f"{'warm' if args.warm else 'cold'} start, rho={options['rho']:g} "
                            f"eps={options['eps_fcone']:g}/{options['eps_bound']:g} max_iter={options['max_iter']} "
                            f"({'fcc_qp_test.py:78-83' if args.options == 'harness' else 'fccqp.pdf Table 1'})",
                "batch_per_gpu": nb, "global_batch": (args.batch_global if args.scaling == "strong" else nb * world),
                "parallelism": f"batch-sharded x{world}, no collective",
                # RCCL carries the two timing barriers and two scalar all-reduces (MAX of elapsed, SUM of solved)
                "collective_bytes_per_run": 16 if world > 1 else 0,
                "kernel": KERNEL_NAMES[solver.kernel_in_use()], "input_storage": args.inputs,
                "n_iter_mean": float(n_iter.mean()), "n_iter_at_max_frac": float(np.mean(n_iter == options["max_iter"])),
                "resident_input_sets": len(sets), "settle_steps": settle_steps,
                "settle_ms_per_step_first_last": ([settle_trace[0], settle_trace[-1]] if settle_trace else None),
            },
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel_ms_mean": mean_ms, "kernel_ms_min": float(np.min(isolated_ms)),
                         "kernel_ms_mean_cold_clock": (region_ms_cold / args.steps if region_ms_cold else None),
                         "kernel_ms_isolated_mean": float(np.mean(isolated_ms)),
                         "kernel_ms_isolated_min": float(np.min(isolated_ms)),
                         # the bound that actually binds (the path is not bandwidth-bound): vector fp64
                         "compute": {"bound": "fp64", "achieved": fp64["achieved"], "peak": FP64_VECTOR_PEAK_TFLOPS,
                                     "unit": "TFLOP/s", "frac": fp64["frac"], "frac_executed": fp64.get("frac_executed"),
                                     "counts": "frac: flops of the reference algorithm (fcc_qp.cpp: two N^3/3 + (1 + K) 2 N^2 per cold "
                                               "solve); frac_executed: what the kernel executes (its MFMAs + mat-vecs)"},
                         "timing": "kernel_ms_mean = HIP-event span of the K back-to-back timed steps / K (includes the "
                                   "near-empty fallback-list launch of every step); kernel_ms_isolated_* (= kernel_ms_min): K more "
                                   "steps, each between its own event pair with the stream drained in front; "
                                   "kernel_ms_mean_cold_clock: the span / K of the region timed before the settling steps",
                         "algorithmic_bytes_per_launch": alg_bytes},
            # diagnostic beside the contractual HBM roofline: algorithmic flops
            # F = f N^3/3 + (p + K) 2 N^2 per solve against the 78.6 TFLOP/s vector-fp64 peak
            "fp64": fp64,
            "parity": parity,
        }
        out.update(sharded)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(shape, options, batch, log)
        if world == 1 and not args.no_extras and args.shape == "walking" and args.workload == "W-sat" \
                and not args.warm and args.inputs == "f64" and args.options == "harness":
            from oracle import oracle as O
            del tens, sets
            torch.cuda.empty_cache()
            try:
                out.update(secondary_measurements(F, P, O, torch, dev, dev_index, log))
            except Exception as exc:          # the headline line must survive a failing extra
                out["extras_error"] = repr(exc)
                log(f"[bench] secondary measurements failed: {exc!r}")
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
