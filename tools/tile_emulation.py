"""numpy emulation of what the fused kernel computes between the load of K_rho and the ADMM loop
(fcc_qp_amd/csrc/fccqp_fused_impl.h: tile_factor_forward, tile_ldl, gram_output, tile_factor_store),
written tile by tile in the kernel's own order so that the algebra can be checked on the CPU:

  * right-looking U^T D U by 16 x 16 tiles with the forward solve of [E | r | border] fused in;
  * the diagonal tile in the column-per-lane form of tile_ldl (elimination + explicit L^-1);
  * bordering: the last N % 16 rows ride along as right-hand sides, their pivots are the LDL^T of the
    Schur complement read off the Gram product, (K_rho^-1)_11 and x_c get rank-kBorder fix-ups;
  * W = I - rho G and the presolve x0 = W^-1 x_c (Woodbury: x0 = (K^-1 r)[0:n]).

`python tools/tile_emulation.py` prints the errors against numpy.linalg on a random walking-shape KKT;
tests/test_tile_algorithm.py runs it on the shapes compiled into the fused kernel.
"""
import numpy as np

T = 16


def border_rows(n, m):
    """Shape::kBorder of the kernel."""
    N = n + m
    b = N % T
    return b if (b != 0 and N - b >= n and (n % T) + b <= 15 and b <= 3) else 0


def tile_ldl(S):
    """Diagonal tile: column c of S per 'lane'.  Returns (L^-1, d, 1/d) with the kernel's pivot rules
    (1/d = 0 for |d| <= DBL_MIN; multipliers are exact quotients)."""
    Tm = S.copy()
    d = np.zeros(T)
    rinv = np.zeros(T)
    lanes = np.arange(T)
    for k in range(T):
        d[k] = Tm[k, k]
        rinv[k] = 1.0 / d[k] if abs(d[k]) > np.finfo(float).tiny else 0.0
        mult = np.where(lanes > k, Tm[k, :] / d[k] if rinv[k] != 0.0 else 0.0, 0.0)   # S_kc / d_k, lanes c > k
        for i in range(k + 1, T):
            Tm[i, :] = Tm[i, :] - Tm[i, k] * mult            # S_ic -= S_ik (from lane k) * m_c
    Ls = -Tm * rinv[None, :]                                  # column c scaled by its own 1/d_c: -L
    Y = np.eye(T)
    for k in range(T - 1):
        for i in range(k + 1, T):
            Y[i, :] = Y[i, :] + Ls[i, k] * Y[k, :]
    return np.tril(Y), d, rinv


def factor_forward_gram(Q, A, b, b_eq, rho):
    """Returns (G, x_c, pivots) with G = (K_rho^-1)[0:n,0:n], x_c = (K_rho^-1 [-b; b_eq])[0:n], computed the
    kernel's way.  Only the lower triangle of Q is read."""
    n, m = Q.shape[0], A.shape[0]
    N = n + m
    Ql = np.tril(Q) + np.tril(Q, -1).T
    K = np.zeros((N, N))
    K[:n, :n] = Ql + rho * np.eye(n)
    K[n:, :n] = A
    K[:n, n:] = A.T
    r = np.concatenate([-b, b_eq])
    NB = border_rows(n, m)
    N1 = N - NB
    NRB = (N1 + T - 1) // T
    NP = NRB * T
    C = np.eye(NP)
    C[:N1, :N1] = K[:N1, :N1]
    ncols = n + 1 + NB
    X = np.zeros((NP, ncols))
    X[:n, :n] = np.eye(n)
    X[:N1, n] = r[:N1]
    for bb in range(NB):
        X[:N1, n + 1 + bb] = K[:N1, N1 + bb]
    rinv_all = np.zeros(NP)
    d_all = np.zeros(NP)
    Y = np.zeros_like(X)
    for p in range(NRB):
        s = slice(T * p, T * p + T)
        Linv, d, rv = tile_ldl(C[s, s])
        Lstrict = np.tril(Linv, -1)
        rinv_all[s], d_all[s] = rv, d
        V = {}
        for j in range(p + 1, NRB):
            sj = slice(T * j, T * j + T)
            V[j] = C[s, sj] + Lstrict @ C[s, sj]                  # V_pj = L^-1 C_pj
        Yp = X[s, :] + Lstrict @ X[s, :]                        # Y_p = L^-1 X_p
        Y[s, :] = Yp
        for i in range(p + 1, NRB):
            si = slice(T * i, T * i + T)
            Aop = V[i] * (-rv[:, None])                          # -(V_pi / d): the A operand, transposed by the MFMA
            for j in range(i, NRB):
                sj = slice(T * j, T * j + T)
                C[si, sj] = C[si, sj] + Aop.T @ V[j]
            X[si, :] = X[si, :] + Aop.T @ Yp
    P = (Y.T * rinv_all[None, :]) @ Y                           # Gram product: K1^-1 on the right-hand sides
    pivots = list(d_all[:N1])
    G = P[:n, :n].copy()
    xc = P[:n, n].copy()
    if NB:
        Z = P[:n, n + 1:]
        S = -P[n + 1:, n + 1:]                                  # D2 = 0: the bordered rows lie in the zero block
        t = r[N1:] - P[n, n + 1:]
        # unpivoted LDL^T of S, exact quotients, pseudo-inverse rule
        Sm = S.copy()
        dinv = np.zeros(NB)
        for k in range(NB):
            dk = Sm[k, k]
            pivots.append(dk)
            dinv[k] = 1.0 / dk if abs(dk) > np.finfo(float).tiny else 0.0
            lk = np.array([Sm[i, k] / dk if dinv[k] != 0.0 else 0.0 for i in range(NB)])
            for i in range(k + 1, NB):
                for j in range(k + 1, i + 1):
                    Sm[i, j] -= lk[i] * Sm[j, k]
            for i in range(k + 1, NB):
                Sm[i, k] = lk[i]

        def ssolve(y):
            y = y.copy()
            for i in range(NB):
                for k in range(i):
                    y[i] -= Sm[i, k] * y[k]
            y *= dinv
            for i in range(NB - 1, -1, -1):
                for k in range(i + 1, NB):
                    y[i] -= Sm[k, i] * y[k]
            return y
        Sinv = np.column_stack([ssolve(e) for e in np.eye(NB)])
        x2 = ssolve(t)
        G = G + Z @ Sinv @ Z.T
        xc = xc - Z @ x2
    return G, xc, np.array(pivots)


def presolve(G, xc, rho):
    """x0 = (I - rho G)^-1 x_c through the same tile factorisation (tile_factor_store + two sweeps)."""
    n = G.shape[0]
    NW = (n + T - 1) // T
    C = np.eye(NW * T)
    C[:n, :n] = np.eye(n) - rho * G
    L = np.eye(NW * T)
    dinv = np.zeros(NW * T)
    for p in range(NW):
        s = slice(T * p, T * p + T)
        Linv, d, rv = tile_ldl(C[s, s])
        L[s, s] = np.linalg.inv(Linv)
        dinv[s] = rv
        Lstrict = np.tril(Linv, -1)
        V = {}
        for j in range(p + 1, NW):
            sj = slice(T * j, T * j + T)
            V[j] = C[s, sj] + Lstrict @ C[s, sj]
            L[sj, s] = (V[j] * rv[:, None]).T
        for i in range(p + 1, NW):
            si = slice(T * i, T * i + T)
            for j in range(i, NW):
                sj = slice(T * j, T * j + T)
                C[si, sj] = C[si, sj] + (V[i] * (-rv[:, None])).T @ V[j]
    y = np.zeros(NW * T)
    y[:n] = xc
    y = np.linalg.solve(L, y) * dinv
    return np.linalg.solve(L.T, y)[:n]


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    n, m, rho = 60, 38, 5e-5
    Gm = rng.standard_normal((n, n))
    Q = Gm @ Gm.T + 0.1 * np.eye(n)
    A = rng.standard_normal((m, n))
    b, b_eq = rng.standard_normal(n), rng.standard_normal(m)
    G, xc, piv = factor_forward_gram(Q, A, b, b_eq, rho)
    K = np.block([[Q + rho * np.eye(n), A.T], [A, np.zeros((m, m))]])
    Kinv = np.linalg.inv(K)
    r = np.r_[-b, b_eq]
    print("border rows", border_rows(n, m))
    print("G   rel err", np.abs(G - Kinv[:n, :n]).max() / np.abs(Kinv[:n, :n]).max())
    print("x_c rel err", np.abs(xc - (Kinv @ r)[:n]).max() / np.abs((Kinv @ r)[:n]).max())
    K0 = np.block([[Q, A.T], [A, np.zeros((m, m))]])
    x0 = np.linalg.solve(K0, r)[:n]
    print("x0  rel err", np.abs(presolve(G, xc, rho) - x0).max() / np.abs(x0).max())
