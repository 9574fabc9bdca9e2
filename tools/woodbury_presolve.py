"""Accuracy of the presolve the fused kernel uses: x0 = (I - rho G)^-1 x_c from the K_rho factorisation
alone (Woodbury on K = K_rho - rho E E^T), against x0 = (K^-1 [-b; b_eq]).head(n) computed with
long-double iterative refinement, and against a direct unpivoted LDL^T of K (what round 1 did, and
what the generic kernel still does first).  CPU only; prints one line per (shape, workload, rho).

    python tools/woodbury_presolve.py

Reference behaviour replaced: the presolve of FCCQP::Solve, /root/reference/src/fcc_qp.cpp:159-178.
"""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from fcc_qp_amd import problems as P
import scipy.linalg as sla

def ldlt_nat(K):
    # unpivoted LDL^T natural order (emulates kernel)
    N = K.shape[0]
    L = np.eye(N); d = np.zeros(N)
    A = K.copy()
    for k in range(N):
        d[k] = A[k,k]
        L[k+1:,k] = A[k+1:,k]/d[k]
        A[k+1:,k+1:] -= np.outer(L[k+1:,k], A[k+1:,k])   # = l * d * l^T
    return L, d

def solve_ldl(L,d,r):
    y = sla.solve_triangular(L, r, lower=True, unit_diagonal=True)
    y = y/d[:,None] if y.ndim==2 else y/d
    return sla.solve_triangular(L.T, y, lower=False, unit_diagonal=True)

def exact_x0(K, r):
    # high precision reference via longdouble refinement
    Kl = K.astype(np.longdouble); rl = r.astype(np.longdouble)
    x = np.linalg.solve(K, r)
    for _ in range(5):
        res = (rl - Kl @ x.astype(np.longdouble)).astype(np.float64)
        x = (x.astype(np.longdouble) + np.linalg.solve(K, res).astype(np.longdouble)).astype(np.float64)
    return x

def run(qp, rho, n, m):
    Q = np.tril(qp['Q']); Q = Q + np.tril(Q,-1).T
    A = qp['A_eq']; N = n+m
    K = np.zeros((N,N)); K[:n,:n]=Q; K[n:,:n]=A; K[:n,n:]=A.T
    r = np.concatenate([-qp['b'], qp['b_eq']])
    x0 = exact_x0(K, r)[:n]
    Kr = K.copy(); Kr[:n,:n] += rho*np.eye(n)
    L,d = ldlt_nat(Kr)
    rhs = np.zeros((N, n+1)); rhs[:n,:n] = np.eye(n); rhs[:,n] = r
    X = solve_ldl(L,d,rhs)
    G = X[:n,:n]; xc = X[:n,n]
    Mw = np.eye(n) - rho*G
    Mw = 0.5*(Mw+Mw.T)
    try:
        c = np.linalg.cholesky(Mw)
        xw = sla.cho_solve((c,True), xc)
        minpiv = np.min(np.diag(c))**2
    except np.linalg.LinAlgError:
        xw = np.full(n, np.nan); minpiv = np.nan
    # direct natural unpivoted on K
    with np.errstate(all='ignore'):
        L0,d0 = ldlt_nat(K)
        xd = solve_ldl(L0,d0,r)[:n] if np.all(np.isfinite(L0)) and np.all(np.isfinite(d0)) and np.all(d0!=0) else np.full(n,np.nan)
    sc = 1+np.max(np.abs(x0))
    return np.max(np.abs(xw-x0))/sc, np.max(np.abs(xd-x0))/sc, minpiv, np.linalg.cond(Mw)

if __name__=="__main__":
  for shape_name in ['walking','large']:
      sh = P.SHAPES[shape_name]
      n,m = sh.n, sh.m
      for wl, kw in [('sat',dict(activity=4.0)), ('easy',dict(activity=0.0, ulim=(400.,600.), lambda_reg=1e-2)), ('psd',dict(activity=4.0, lambda_reg=0.0))]:
          for rho in [1e-6, 5e-5, 1e-3, 1e-1]:
              ew=[];ed=[];mp=[];cd=[]
              for seed in range(1000, 1000+(12 if shape_name=='walking' else 4)):
                  qp = P.osc_qp(seed, sh, **kw)
                  a,b,c,d_ = run(qp, rho, n, m)
                  ew.append(a); ed.append(b); mp.append(c); cd.append(d_)
              print(f"{shape_name:8s} {wl:5s} rho={rho:7.0e}  woodbury err max {np.nanmax(ew):.2e}  direct-nat err max {np.nanmax(ed):.2e}  min chol piv {np.nanmin(mp):.2e} cond(I-rhoG) {np.max(cd):.2e}")
