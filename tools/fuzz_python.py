"""Randomised run of the Python surface (test infrastructure): fcc_qp_amd.FCCQP -- the reference's own class,
src/main.cpp:42-54 -- fed with numpy arrays in every layout its Eigen::Ref casters accept (C / Fortran order, strided
and reversed views, float32 / integer dtypes, (n,) (n,1) (1,n) vectors, lists and tuples, keyword arguments), and
FCCQPBatch with stacked arrays; results against the CPU oracle on the converted values.
    python tools/fuzz_python.py [seconds] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fcc_qp_amd as F
from oracle import oracle as O
from tools.fuzz_parity import random_shape, random_batch


def dress(rng, a, kind):
    """the same values in another memory layout / container"""
    a = np.asarray(a, dtype=np.float64)
    k = int(rng.integers(0, 7))
    if k == 0: return np.ascontiguousarray(a)
    if k == 1: return np.asfortranarray(a)
    if k == 2:                                             # every second element of a larger array
        big = np.zeros(tuple(2 * d for d in a.shape)); sl = tuple(slice(None, None, 2) for _ in a.shape)
        big[sl] = a; return big[sl]
    if k == 3:                                             # reversed view (negative strides)
        rev = tuple(slice(None, None, -1) for _ in a.shape)
        return np.ascontiguousarray(a[rev])[rev]
    if k == 4 and kind == "vec": return a.reshape(-1, 1) if rng.random() < 0.5 else a.reshape(1, -1)
    if k == 5 and kind == "vec" and np.all(np.isfinite(a)): return a.tolist()
    if k == 6 and a.ndim == 2: return np.asfortranarray(a.T).T   # C-contiguous data seen through a transposed F array
    return a


def main(budget=60.0, seed=1, max_cases=None):
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    cases = bad = 0
    worst = 0.0
    while time.time() < t_end and (max_cases is None or cases < max_cases):
        cases += 1
        dims = random_shape(rng)
        while dims[0] + dims[1] > 150: dims = random_shape(rng)
        n, m, nc, s = dims
        B = random_batch(rng, dims, 3, singular=False)
        f32 = rng.random() < 0.3
        arrs = [np.asarray(np.asarray(x, dtype=np.float32), dtype=np.float64) if f32 else x for x in B.arrays()]
        opts = dict(max_iter=int(rng.choice([0, 1, 15, 60])), rho=float(10.0 ** rng.uniform(-5, -1)),
                    eps_fcone=float(10.0 ** rng.uniform(-8, -3)), eps_bound=float(10.0 ** rng.uniform(-8, -3)))
        fo = F.FCCQPOptions(); fo.max_iter, fo.rho, fo.eps_fcone, fo.eps_bound = opts["max_iter"], opts["rho"], opts["eps_fcone"], opts["eps_bound"]
        z_ref, it_ref, _, _ = O.solve_batch(dims, opts, *arrs, nthreads=3)
        # the reference's class, one instance at a time
        q = F.FCCQP(n, m, nc, s); q.set_options(fo)
        for i in range(3):
            Qi, bi, Ai, bei, mui, lbi, ubi = (x[i] for x in arrs)
            cast = (lambda v: np.asarray(v, dtype=np.float32)) if f32 else (lambda v: v)
            args = dict(Q=dress(rng, cast(Qi), "mat"), b=dress(rng, cast(bi), "vec"), A_eq=dress(rng, cast(Ai), "mat"),
                        b_eq=dress(rng, cast(bei), "vec"), friction_coeffs=(list(mui) if rng.random() < 0.5 else cast(mui)),
                        lb=dress(rng, cast(lbi), "vec"), ub=dress(rng, cast(ubi), "vec"))
            if f32: args = {k: (np.asarray(v, dtype=np.float32) if isinstance(v, np.ndarray) else v) for k, v in args.items()}
            try:
                if rng.random() < 0.5: q.Solve(**args)
                else: q.Solve(*[args[k] for k in ("Q", "b", "A_eq", "b_eq", "friction_coeffs", "lb", "ub")])
                sol = q.GetSolution()
            except Exception as e:                          # noqa: BLE001
                bad += 1; print("ERROR", dims, i, {k: (type(v).__name__, getattr(v, "shape", None), getattr(v, "dtype", None), getattr(v, "strides", None)) for k, v in args.items()}, repr(e), flush=True); continue
            z = np.asarray(sol.z)
            e = float(np.max(np.abs(z - z_ref[i])) / (1 + np.max(np.abs(z_ref[i]))))
            worst = max(worst, e)
            if not (e <= 1e-6) or abs(int(sol.details.n_iter) - int(it_ref[i])) > 1:
                bad += 1
                print("BROKEN FCCQP", dims, opts, "instance", i, f"rel err {e:.2e}", "n_iter", sol.details.n_iter, int(it_ref[i]),
                      {k: (type(v).__name__, getattr(v, "shape", None), getattr(v, "dtype", None), getattr(v, "strides", None)) for k, v in args.items()}, flush=True)
        # the batch class on the stacked arrays (C order required? any layout is converted)
        qb = F.FCCQPBatch(n, m, nc, s, 3); qb.set_options(fo)
        stacked = [dress(rng, x, "mat3") for x in arrs]
        if f32: stacked = [np.asarray(x, dtype=np.float32) for x in stacked]
        try:
            qb.Solve(*stacked); sb = qb.GetSolution()
            zb = np.asarray(sb.z).reshape(3, n)
            eb = float(np.max(np.abs(zb - z_ref) / (1 + np.max(np.abs(z_ref), axis=1, keepdims=True))))
            worst = max(worst, eb)
            if not (eb <= 1e-6):
                bad += 1; print("BROKEN FCCQPBatch", dims, opts, f"rel err {eb:.2e}", [(x.shape, x.dtype, x.strides) for x in stacked], flush=True)
        except Exception as e:                              # noqa: BLE001
            bad += 1; print("ERROR batch", dims, repr(e), flush=True)
    print(f"fuzz_python: {cases} cases, {bad} broken, worst rel err {worst:.2e}", flush=True)
    if max_cases is not None and cases < max_cases and not bad:
        # (a run bounded by a case count must reach it: a budget that cuts it short is a failure, not a shorter test)
        print(f"fuzz: the time budget ended after {cases} of {max_cases} cases", flush=True)
        return 2
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1))
