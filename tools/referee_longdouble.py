"""Extended-precision referee for a parity disagreement (test infrastructure): the reference's algorithm
(/root/reference/src/fcc_qp.cpp:57-191, constraint_utils.cpp:5-46) in numpy.longdouble (80-bit x87: 64-bit mantissa) with
an LU of its own, run on the cases tools/fuzz_parity.py saved:  python tools/referee_longdouble.py gpurun_out/fuzz_case_*.npz
Prints, per case, how far the HIP result and the CPU oracle's are from it."""
import sys
import numpy as np
LD = np.longdouble


def lu_inverse(K):
    n = K.shape[0]
    M = np.concatenate([K.astype(LD), np.eye(n, dtype=LD)], axis=1)
    for k in range(n):
        p = k + int(np.argmax(np.abs(M[k:, k])))
        if p != k: M[[k, p]] = M[[p, k]]
        M[k] = M[k] / M[k, k]
        f = M[:, k].copy(); f[k] = 0
        M -= np.outer(f, M[k])
    return M[:, n:]


def cone3(f, mu):
    nxy = np.sqrt(f[0] * f[0] + f[1] * f[1])
    if mu * f[2] >= nxy: return f.copy()
    if f[2] < -mu * nxy: return np.zeros(3, dtype=LD)
    k = mu * f[2] / nxy
    r = np.array([k * f[0], k * f[1], f[2]], dtype=LD)
    z = r @ r
    if z > 0: r = r / np.sqrt(z)
    return (r @ f) * r


def solve(n, m, nc, s, opts, Q, b, A, beq, mu, lb, ub):
    Q = Q.astype(LD); A = A.astype(LD).reshape(m, n); b = b.astype(LD); beq = beq.astype(LD)
    lb = lb.astype(LD); ub = ub.astype(LD); mu = mu.astype(LD)
    Ql = np.tril(Q) + np.tril(Q, -1).T
    K = np.zeros((n + m, n + m), dtype=LD); K[:n, :n] = Ql; K[n:, :n] = A; K[:n, n:] = A.T
    rhs = np.concatenate([-b, beq])
    x = (lu_inverse(K) @ rhs)[:n]
    eq_only = nc == 0 and np.all(np.isinf(lb)) and np.all(np.isinf(ub))
    if eq_only: return x, 0
    Kr = K.copy(); Kr[:n, :n] += LD(opts["rho"]) * np.eye(n, dtype=LD)
    Ki = lu_inverse(Kr)
    mux = np.zeros(n, dtype=LD); mulc = np.zeros(nc, dtype=LD)
    xb = x.copy(); lcb = x[s:s + nc].copy()
    rho = LD(opts["rho"]); n_iter = opts["max_iter"]
    for it in range(opts["max_iter"]):
        q = -rho * (xb - mux); q[s:s + nc] = -rho * (lcb - mulc)
        r = rhs.copy(); r[:n] = -(b + q)
        x = (Ki @ r)[:n]
        xb = np.maximum(np.minimum(x + mux, ub), lb)
        f = x[s:s + nc] + mulc
        lcb = np.concatenate([cone3(f[3 * i:3 * i + 3], mu[i]) for i in range(nc // 3)]) if nc else np.zeros(0, dtype=LD)
        xr = x - xb; lr = x[s:s + nc] - lcb
        mux = mux + xr; mulc = mulc + lr
        xn = np.max(np.abs(xr)) if n else 0; ln = np.max(np.abs(lr)) if nc else 0
        if ln < opts["eps_fcone"] and xn < opts["eps_bound"]:
            n_iter = it; break
    return x, n_iter


for path in (sys.argv[1:] if __name__ == "__main__" else []):
    d = np.load(path)
    n, m, nc, s = (int(v) for v in d["dims"])
    opts = dict(max_iter=int(d["max_iter"]), rho=float(d["rho"]), eps_fcone=float(d["eps_fcone"]), eps_bound=float(d["eps_bound"]))
    B = d["Q"].shape[0]
    err = np.max(np.abs(d["z_gpu"] - d["z_oracle"]), axis=1) / (1 + np.max(np.abs(d["z_oracle"]), axis=1))
    worst = np.argsort(-err)[:3]
    print(path, "dims", (n, m, nc, s), opts, "kernel", int(d["kernel"]))
    for i in worst:
        zr, itr = solve(n, m, nc, s, opts, d["Q"][i], d["b"][i], d["A"][i], d["b_eq"][i], d["mu"][i], d["lb"][i], d["ub"][i])
        sc = 1 + float(np.max(np.abs(zr)))
        eg = float(np.max(np.abs(d["z_gpu"][i] - zr))) / sc; eo = float(np.max(np.abs(d["z_oracle"][i] - zr))) / sc
        Kc = np.linalg.cond(np.block([[np.tril(d["Q"][i]) + np.tril(d["Q"][i], -1).T + opts["rho"] * np.eye(n), d["A"][i].reshape(m, n).T],
                                      [d["A"][i].reshape(m, n), np.zeros((m, m))]]))
        print(f"  instance {i}: gpu-oracle {err[i]:.2e} | gpu-referee {eg:.2e} oracle-referee {eo:.2e} | n_iter gpu {int(d['it_gpu'][i])} "
              f"oracle {int(d['it_oracle'][i])} referee {itr} | cond(K_rho) {Kc:.1e}")
