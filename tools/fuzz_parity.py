"""Randomised parity run: the HIP path (through the C ABI) against the CPU oracle on random shapes, options, kernels and
inputs, cold and warm.  Test infrastructure like tests/: the oracle is the checker.
    python tools/fuzz_parity.py [seconds] [seed]
Prints one line per case that is notable (rel err > 1e-9, n_iter different, flags raised) and a summary; exit code 1
if a case breaks the stated contract (tests/helpers.py: rel err <= 1e-6, |dn_iter| <= 1, on non-degenerate inputs).

Cases: a listed (fused / cooperative) shape or a random one (generic kernel); any kernel that serves the shape; options
rho 1e-6 .. 1, eps 1e-8 .. 1e-2, max_iter 0 .. 300; dense random QPs of varying conditioning with a random set of
finite bounds (also on contact variables, also lb == ub), friction 0 .. 1.2, now and then a zero right-hand side, a
contact with f_z pinned to zero, a variable without cost or constraint (singular K: the COD pass, compared loosely); batch
1 .. 200; a second, warm-started solve of the same inputs (small batches, through the oracle's class)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fcc_qp_amd import problems as P
from oracle import oracle as O
from tests import capi as CAPI

FUSED = [(60, 38, 12, 38), (12, 5, 6, 3), (12, 8, 6, 3), (32, 14, 6, 20), (32, 18, 6, 20), (12, 7, 6, 3), (24, 6, 15, 4)]
COOP = [(114, 60, 24, 66), (60, 38, 12, 38), (70, 20, 9, 40), (32, 14, 6, 20), (20, 8, 6, 5)]


def random_shape(rng):
    if rng.random() < (0.9 if os.environ.get("FUZZ_BIG") else 0.2):    # three / four / eight register slots, N up to 512
        n = int(rng.integers(70, 260))
        m = int(rng.integers(0, min(n, 512 - n, 140)))
    else:
        n = int(rng.integers(3, 90))
        m = int(rng.integers(0, min(n, 50)))
    ncmax = min(n // 3, 8)
    k = int(rng.integers(0, ncmax + 1))
    nc = 3 * k
    s = int(rng.integers(0, n - nc + 1))
    return (n, m, nc, s)


def random_batch(rng, dims, nb, singular):
    n, m, nc, s = dims
    Lq = rng.standard_normal((nb, n, n)) / np.sqrt(n)
    delta = 10.0 ** rng.uniform(-3, 0.5)
    Q = Lq @ np.transpose(Lq, (0, 2, 1)) + delta * np.eye(n)
    if rng.random() < 0.3:                                   # badly scaled variables
        sc = 10.0 ** rng.uniform(-1.5, 1.5, n)
        Q = Q * sc[None, :, None] * sc[None, None, :]
    A = rng.standard_normal((nb, m, n))
    b = rng.standard_normal((nb, n)) * 10.0 ** rng.uniform(-1, 1)
    beq = rng.standard_normal((nb, m)) * (0.0 if rng.random() < 0.1 else 0.3)
    if singular:
        # a variable without cost or constraint: K singular, K_rho regular (rho fills the zero on the diagonal) -- the
        # reference's COD branch (fcc_qp.cpp:164-177), minimum-norm x0.  (Rank-deficient A_eq is NOT a parity case: K_rho
        # is singular too, Eigen's LDLT keeps a rounding-level pivot and the reference's iterates are rounding noise.)
        free = [j for j in range(n) if not (s <= j < s + nc)]
        j = free[int(rng.integers(len(free)))] if free else 0
        # (some or all instances of the batch: the fallback list of a launch is a mixture)
        who = np.flatnonzero(rng.random(nb) < rng.choice([0.05, 0.5, 1.0]))
        if who.size == 0: who = np.array([int(rng.integers(nb))])
        for i in who:
            Q[i, j, :] = 0.0; Q[i, :, j] = 0.0; A[i, :, j] = 0.0; b[i, j] = 0.0
    lb = np.full((nb, n), -np.inf); ub = np.full((nb, n), np.inf)
    k = int(rng.integers(0, n + 1)) if rng.random() < 0.7 else 0
    idx = rng.permutation(n)[:k]
    w = 10.0 ** rng.uniform(-2, 0.5)
    lb[:, idx] = -w * rng.uniform(0.1, 1.0, (nb, len(idx))); ub[:, idx] = w * rng.uniform(0.1, 1.0, (nb, len(idx)))
    if k and rng.random() < 0.2:
        ub[:, idx[0]] = lb[:, idx[0]]                        # lb == ub
    if nc and rng.random() < 0.2:
        lb[:, s + 2] = 0.0; ub[:, s + 2] = 0.0               # f_z pinned to zero: the zero-ray quirk's neighbourhood
    mu = rng.uniform(0.0 if rng.random() < 0.2 else 0.2, 1.2, (nb, nc // 3))
    return P.Batch(Q, b, A, beq, mu, lb, ub)


def main(budget=120.0, seed=1, replay=(), max_cases=None):
    # python tools/fuzz_parity.py <seconds> <seed> <case> [<case> ...]: only those cases of the seed's sequence are solved
    # (the others are generated and dropped), and each is saved as gpurun_out/fuzz_case_<seed>_<case>.npz
    replay = sorted(int(a) for a in replay)
    t_end = time.time() + budget
    cases = bad = notable = 0
    worst = 0.0
    t_print = time.time()
    while time.time() < t_end and (max_cases is None or cases < max_cases):
        cases += 1
        if replay:
            if cases > replay[-1]: break
            if cases not in replay: continue
        rng = np.random.default_rng([seed, cases])              # every case has a generator of its own: replayable alone
        r = rng.random()
        if os.environ.get("FUZZ_BIG"): r = 0.5 + 0.5 * r       # (FUZZ_BIG=1: random shapes only, nine in ten of them large)
        if r < 0.35: dims = FUSED[int(rng.integers(len(FUSED)))]
        elif r < 0.5: dims = COOP[int(rng.integers(len(COOP)))]
        else: dims = random_shape(rng)
        kernels = [CAPI.KERNEL_AUTO, CAPI.KERNEL_GENERIC]
        if dims in FUSED: kernels += [CAPI.KERNEL_FUSED, CAPI.KERNEL_FUSED_SPLIT]
        if dims in COOP: kernels += [CAPI.KERNEL_COOP]
        if dims in COOP and dims[0] > 64: kernels += [CAPI.KERNEL_COOP_SPLIT]      # (the two-launch form: row-split shapes)
        kernel = kernels[int(rng.integers(len(kernels)))]
        nb = int(rng.choice([1, 2, 3, 7, 17, 64, 65, 130, 200, 1100, 2500]))
        if dims[0] > 64: nb = min(nb, 200)
        if dims[0] > 100: nb = min(nb, 17)
        if dims[0] + dims[1] > 256: nb = min(nb, 3)
        f32 = rng.random() < 0.1                             # fp32-storage entry point: the oracle gets the rounded inputs
        singular = rng.random() < 0.08
        if dims[1] >= dims[0] - 1: singular = False         # (a dead variable under n - 1 or n equality rows makes A_eq rank-deficient)
        opts = dict(max_iter=int(rng.choice([0, 1, 2, 5, 15, 40, 100, 300])), rho=float(10.0 ** rng.uniform(-6, 0)),
                    eps_fcone=float(10.0 ** rng.uniform(-8, -2)), eps_bound=float(10.0 ** rng.uniform(-8, -2)))
        B = random_batch(rng, dims, nb, singular)
        if f32:
            B = P.Batch(*(np.asarray(np.asarray(a, dtype=np.float32), dtype=np.float64) for a in B.arrays()))
        tag = f"case {cases} dims {dims} kernel {kernel} batch {nb} opts {opts} singular {singular}" + (" f32" if f32 else "")
        try:
            s = CAPI.Solver(*dims, batch=nb); s.set_options(**opts); s.set_kernel(kernel)
            if os.environ.get("FUZZ_SELF_CHECK"): s.set_self_check(True)     # the equality-residual check behind every solve
            (s.solve_f32_ok if f32 else s.solve_ok)(*B.arrays())
            z, d = s.solution()
        except Exception as e:                               # noqa: BLE001
            print("ERROR", tag, repr(e), flush=True); bad += 1; continue
        it = np.array([x.n_iter for x in d]); fl = np.array([x.flags for x in d])
        z_ref, it_ref, _, _ = O.solve_batch(dims, opts, *B.arrays(), nthreads=8)
        if replay:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            np.savez(os.path.join(ROOT, "gpurun_out", f"fuzz_case_{seed}_{cases}.npz"), dims=np.array(dims), kernel=kernel,
                     max_iter=opts["max_iter"], rho=opts["rho"], eps_fcone=opts["eps_fcone"], eps_bound=opts["eps_bound"],
                     z_gpu=z, it_gpu=it, flags=fl, z_oracle=z_ref, it_oracle=it_ref,
                     **{k: v for k, v in zip(("Q", "b", "A", "b_eq", "mu", "lb", "ub"), B.arrays())})
        err = np.max(np.abs(z - z_ref), axis=1) / (1.0 + np.max(np.abs(z_ref), axis=1))
        dn = np.abs(it.astype(np.int64) - it_ref.astype(np.int64))
        e, dm = float(np.nanmax(err)) if np.all(np.isfinite(z)) else float("inf"), int(dn.max())
        tol = 1e-4 if singular else 1e-6
        broken = (not np.isfinite(e)) or e > tol or (dm > 1 and not singular)
        if broken and np.isfinite(e) and not singular:
            # who is right?  The same algorithm in 80-bit arithmetic on the worst instance (tools/referee_longdouble.py):
            # an ill-conditioned K_rho shows as BOTH sides being off by about as much -- not a defect of either
            from tools import referee_longdouble as R
            i = int(np.argmax(err))
            zr, _ = R.solve(*dims, opts, *[x[i] for x in B.arrays()])
            sc = 1 + float(np.max(np.abs(zr)))
            eg = float(np.max(np.abs(z[i] - zr))) / sc; eo = float(np.max(np.abs(z_ref[i] - zr))) / sc
            n_, m_ = dims[0], dims[1]
            Ai = B.arrays()[2][i].reshape(m_, n_); Qi = B.arrays()[0][i]
            Kr = np.block([[np.tril(Qi) + np.tril(Qi, -1).T + opts["rho"] * np.eye(n_), Ai.T], [Ai, np.zeros((m_, m_))]])
            cond = float(np.linalg.cond(Kr))
            tag += f" | referee: gpu {eg:.1e} oracle {eo:.1e} cond(K_rho) {cond:.1e}"
            # benign: both sides off alike, or off by no more than the conditioning explains (~50 eps cond)
            if dm <= 1 and (eg <= 10 * max(eo, 1e-12) or eg <= 1e-14 * cond):
                broken = False; tag += " (ill-conditioned input)"
        if os.environ.get("FUZZ_SELF_CHECK") and not singular and np.any(fl & 32):
            broken = True; tag += " | FLAG_EQ_RESIDUAL raised on a sound solve"
        if broken: bad += 1
        if broken or e > 1e-9 or dm > 0:
            notable += 1
            print("BROKEN" if broken else "note", tag, f"rel err {e:.2e} dn_iter {dm} flags {sorted(set(fl.tolist()))} "
                  f"n_iter ref {sorted(set(it_ref.tolist()))[:6]}", flush=True)
        worst = max(worst, e if np.isfinite(e) and not singular else 0.0)
        # warm-started second solve of the same inputs, against the oracle's class (small batches)
        if nb <= 3 and not singular and opts["max_iter"] > 0:
            arr2 = list(B.arrays())
            arr2[1] = arr2[1] + 0.05 * rng.standard_normal(arr2[1].shape)      # the next QP of a sequence: b moves a little
            if f32: arr2[1] = np.asarray(np.asarray(arr2[1], dtype=np.float32), dtype=np.float64)
            s.set_warm_start(True); (s.solve_f32_ok if f32 else s.solve_ok)(*arr2); z2, d2 = s.solution()
            for i in range(nb):
                o = O.OracleFCCQP(*dims); oo = O.FCCQPOptions()
                oo.max_iter, oo.rho, oo.eps_fcone, oo.eps_bound = opts["max_iter"], opts["rho"], opts["eps_fcone"], opts["eps_bound"]
                o.set_options(oo)
                a = [x[i] for x in B.arrays()]
                o.Solve(*a); o.set_warm_start(True); o.Solve(*[x[i] for x in arr2])
                sol = o.GetSolution()
                e2 = float(np.max(np.abs(z2[i] - sol.z)) / (1.0 + np.max(np.abs(sol.z))))
                d2n = abs(int(d2[i].n_iter) - int(sol.details.n_iter))
                if e2 > 1e-6 or d2n > 1:
                    bad += 1
                    print("BROKEN warm", tag, f"instance {i} rel err {e2:.2e} dn_iter {d2n}", flush=True)
                worst = max(worst, e2)
        s.close()
        if time.time() - t_print > 50:
            print(f"... {cases} cases, {bad} broken, worst rel err {worst:.2e}", flush=True); t_print = time.time()
    print(f"fuzz: {cases} cases, {notable} notable, {bad} broken, worst rel err (non-singular) {worst:.2e}", flush=True)
    if max_cases is not None and cases < max_cases and not bad and not replay:
        # (a run bounded by a case count must reach it: a budget that cuts it short is a failure, not a shorter test)
        print(f"fuzz: the time budget ended after {cases} of {max_cases} cases", flush=True)
        return 2
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1, sys.argv[3:]))
