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
                  stream) against the 8 TB/s HBM3E peak -- the contract figure; this path is
                  latency/LDS-bound, not bandwidth-bound, so the fraction is small;
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


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="instances per GPU (metric: 4096)")
    ap.add_argument("--workload", default="W-sat", choices=["W-sat", "W-mix", "W-easy", "W-psd"])
    ap.add_argument("--shape", default="walking", choices=["walking", "large"],
                    help="walking = FCCQP(60,38,12,38), the metric's shape; large = FCCQP(114,60,24,66), "
                         "BASELINE configs[3] (secondary: generic kernel, four waves per QP)")
    ap.add_argument("--kernel", default="auto", choices=["auto", "generic", "fused"])
    ap.add_argument("--options", default="harness", choices=["harness", "paper"],
                    help="harness: fcc_qp_test.py:78-83 (headline); paper: fccqp.pdf Table 1 (max_iter=15, eps=1e-4)")
    ap.add_argument("--warm", action="store_true",
                    help="secondary mode: time warm-started solves (every step continues from the previous one)")
    ap.add_argument("--inputs", default="f64", choices=["f64", "f32"],
                    help="storage type of the resident inputs; f32 = BASELINE configs[2]'s fp32-storage "
                         "path (widened on load, arithmetic stays fp64; the batch is the float-rounded problem)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-gate", action="store_true")
    return ap.parse_args()


FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X vector fp64 (MI355X_MICROARCH.md)


def fp64_diag(shape, warm, n_iter_mean, at_max_frac, nb, kernel_ms):
    """Algorithmic flops of the reference algorithm per launch: f
    factorisations of N^3/3, p presolve back-solves and K iterations of 2 N^2 each."""
    N = shape.n + shape.m
    f, p = (1, 0) if warm else (2, 1)
    k_exec = n_iter_mean + (1.0 - at_max_frac)        # a converged instance executed n_iter + 1 iterations
    flops = nb * (f * N ** 3 / 3.0 + (p + k_exec) * 2.0 * N * N)
    ach = flops / (kernel_ms * 1e-3) / 1e12
    return {"flops_per_launch": flops, "achieved": ach, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": ach / FP64_VECTOR_PEAK_TFLOPS}


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


def main():
    args = parse_args()
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
    nb = args.batch
    t0 = time.time()
    # instance i: seed 1000 + i (the large shape's generator is slow: 64 distinct instances, tiled)
    batch = P.workload(args.workload, nb, shape, seed0=S.rank_seed0(rank, nb),
                       **({"distinct": 64} if args.shape == "large" else {}))
    log(f"[bench] generated {nb} '{args.workload}' instances per rank in {time.time() - t0:.1f} s")

    solver = F.FCCQPBatch(*shape.ctor_args, nb, dev_index)
    o = F.FCCQPOptions()
    o.max_iter, o.rho, o.eps_fcone, o.eps_bound = (options["max_iter"], options["rho"],
                                                   options["eps_fcone"], options["eps_bound"])
    solver.set_options(o)
    solver.set_warm_start(False)
    solver.set_kernel({"auto": F.KERNEL_AUTO, "generic": F.KERNEL_GENERIC, "fused": F.KERNEL_FUSED}[args.kernel])
    if args.inputs == "f32":   # the problem IS the float-rounded one: oracle and CPU baseline see the same values
        batch = P.Batch(*(a.astype(np.float32).astype(np.float64) for a in batch.arrays()))
    tens = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in batch.arrays()]   # resident in HBM
    if args.inputs == "f32":
        tens = [t.float() for t in tens]
    torch.cuda.synchronize()

    def step():
        F.solve_tensors(solver, *tens)          # one launch on torch's current stream

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

    # ---- timed region ------------------------------------------------------------------------------
    if args.warm:
        solver.set_warm_start(True)     # state (x, mu) stays resident; the presolve is skipped (fcc_qp.cpp:159)
    for _ in range(args.warmup):
        step()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    S.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    for e0, e1 in ev:
        e0.record()
        step()
        e1.record()
    torch.cuda.synchronize()
    S.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    kernel_ms = [e0.elapsed_time(e1) for e0, e1 in ev]        # HIP events on the launch stream

    red_dev = dev if (world > 1 and dist.get_backend() == "nccl") else None
    elapsed = S.max_over_ranks(elapsed, red_dev)      # the slowest rank defines the step time
    solved = S.sum_over_ranks(float(nb * args.steps), red_dev)

    if rank == 0:
        value = solved / elapsed
        mean_ms = float(np.mean(kernel_ms))
        alg_bytes = solver.AlgorithmicBytes()                  # 8 * batch * (n^2 + mn + 4n + m + nc/3)
        achieved = alg_bytes / (mean_ms * 1e-3) / 1e9
        traffic = None
        tr_path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tr_path):
            try:
                tr = json.load(open(tr_path))
                if tr.get("batch") == nb and tr.get("workload") == args.workload and tr.get("kernel") == solver.kernel_in_use() \
                        and args.shape == "walking" and args.inputs == "f64" and not args.warm:
                    traffic = tr.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": ("QP solves/sec at batch=4096, walking-log shape" if args.shape == "walking" else
                       "QP solves/sec, 36-DoF / 8-contact shape (BASELINE configs[3], secondary)"),
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: FCCQP{tuple(shape.ctor_args)} {args.shape}-shape OSC QPs, "
                            f"{'warm' if args.warm else 'cold'} start, rho={options['rho']:g} "
                            f"eps={options['eps_fcone']:g}/{options['eps_bound']:g} max_iter={options['max_iter']} "
                            f"({'fcc_qp_test.py:78-83' if args.options == 'harness' else 'fccqp.pdf Table 1'})",
                "batch_per_gpu": nb, "global_batch": nb * world, "parallelism": f"batch-sharded x{world}, no collective",
                "kernel": {1: "generic", 2: "fused"}[solver.kernel_in_use()], "input_storage": args.inputs,
                "n_iter_mean": float(n_iter.mean()), "n_iter_at_max_frac": float(np.mean(n_iter == options["max_iter"])),
            },
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel_ms_mean": mean_ms, "kernel_ms_min": float(np.min(kernel_ms)),
                         "algorithmic_bytes_per_launch": alg_bytes},
            # diagnostic beside the contractual HBM roofline: algorithmic flops
            # F = f N^3/3 + (p + K) 2 N^2 per solve against the 78.6 TFLOP/s vector-fp64 peak
            "fp64": fp64_diag(shape, args.warm, float(n_iter.mean()), float(np.mean(n_iter == options["max_iter"])),
                              nb, mean_ms),
            "parity": parity,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(shape, options, batch, log)
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
