"""Synthetic operational-space-control QPs in the shape of the reference's walking log.

The reference's data set ``test_data/id_qp_log_walking.npz`` is missing from its tree
(/root/reference/.MISSING_LARGE_BLOBS:1; loader at /root/reference/fcc_qp_test.py:22-24), so the
benchmark and the parity tests use problems synthesised here.  The structure follows eq. (10) of
the FCCQP paper (fccqp.pdf) and SURVEY.md section 8(d):

    decision vector  x = [ vdot (nv) | u (nu) | lambda_h (nh) | lambda_c (nc) | eps (ne) ]
    dynamics         [ M, -B, -J_h^T, -J_c^T, 0 ] x = -C                      (nv rows)
    holonomic        [ J_h, 0, 0, 0, 0 ] x          = -Jdot_h v               (nh rows)
    soft contact     [ J_c,act, 0, 0, 0, I ] x      = -Jdot_c v               (ne rows)
    cost             1/2 x^T Q x + b^T x  (task-space tracking + regularisation)
    bounds           torque limits on u, +-inf elsewhere; friction cones on lambda_c

Walking shape (fcc_qp_test.py:52-56,77):  FCCQP(60, 38, 12, 38):
    nv=22 nu=10 nh=6 nc=12 ne=10 -> n=60, m=22+6+10=38, lambda_c_start=38.
Large shape: nv=36 nu=30 nh=0 nc=24 ne=24 -> n=114, m=60, start=66.

Every instance is generated from ``numpy.random.default_rng(seed)`` alone, so fixtures and the
GPU box regenerate identical inputs.  Instance i of a batch uses seed ``1000 + i``.
"""
from __future__ import annotations

import dataclasses
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np


@dataclasses.dataclass(frozen=True)
class Shape:
    """Sizes of one QP family (the four FCCQP constructor arguments plus the block split)."""

    nv: int
    nu: int
    nh: int
    nc: int
    ne: int

    @property
    def n(self) -> int:  # num_vars
        return self.nv + self.nu + self.nh + self.nc + self.ne

    @property
    def m(self) -> int:  # num_equality_constraints
        return self.nv + self.nh + self.ne

    @property
    def lambda_c_start(self) -> int:
        return self.nv + self.nu + self.nh

    @property
    def ctor_args(self):
        """(num_vars, num_equality_constraints, nc, lambda_c_start) as FCCQP's constructor takes them."""
        return (self.n, self.m, self.nc, self.lambda_c_start)


WALKING = Shape(nv=22, nu=10, nh=6, nc=12, ne=10)   # FCCQP(60, 38, 12, 38)
LARGE = Shape(nv=36, nu=30, nh=0, nc=24, ne=24)     # FCCQP(114, 60, 24, 66)
MID = Shape(nv=44, nu=38, nh=0, nc=36, ne=30)       # FCCQP(148, 74, 36, 82): humanoid-sized, N = 222 (no specialised kernel)
SMALL = Shape(nv=18, nu=8, nh=4, nc=9, ne=8)        # FCCQP(47, 30, 9, 30): a smaller biped, NOT in fused_shapes.def (run-time compiled kernel)
SHAPES = {"walking": WALKING, "large": LARGE, "mid": MID, "small": SMALL}

#: options of the reference harness (fcc_qp_test.py:78-83) -- canonical for all timed runs
HARNESS_OPTIONS = dict(max_iter=100, rho=5e-5, eps_fcone=1e-6, eps_bound=1e-6)
#: options of the paper's hardware deployment (fccqp.pdf Table 1)
PAPER_OPTIONS = dict(max_iter=15, rho=5e-5, eps_fcone=1e-4, eps_bound=1e-4)


def _skew(p):
    return np.array([[0.0, -p[2], p[1]], [p[2], 0.0, -p[0]], [-p[1], p[0], 0.0]])


def osc_qp(seed: int, shape: Shape = WALKING, activity: float = 4.0,
           ulim: Sequence[float] = (60.0, 150.0), lambda_reg: float = 1e-6) -> Dict[str, np.ndarray]:
    """One synthetic OSC QP as a dict with the keys of the reference log
    (``Q, b, A_eq, b_eq, friction_coeffs, lb, ub``; fcc_qp_test.py:28-30).

    ``activity`` is a lateral push acceleration demanded of the floating base (m/s^2): 0 leaves
    the contact forces well inside their cones, >= 2 drives them onto the cone surface.
    ``lambda_reg`` is the cost weight on the constraint forces: 1e-6 leaves the distribution of
    load over the contact points almost free (some land on the cone boundary even unpushed),
    1e-2 spreads the load evenly so that an unpushed instance is strictly feasible.
    """
    rng = np.random.default_rng(seed)
    nv, nu, nh, nc, ne = shape.nv, shape.nu, shape.nh, shape.nc, shape.ne
    n, m = shape.n, shape.m
    nj = nv - 6                      # actuated + passive joints
    ncontact = nc // 3

    # mass matrix: floating base [rot(3) | trans(3)] + joints, SPD
    L = np.tril(rng.normal(0.0, 0.05, size=(nv, nv)), -1)
    d = np.concatenate([rng.uniform(1.0, 3.0, 3), np.full(3, 30.0), rng.uniform(0.1, 1.0, nj)])
    Lfull = L + np.diag(np.sqrt(d))
    M = Lfull @ Lfull.T

    # actuation: the first nu joints are actuated
    B = np.zeros((nv, nu))
    for k in range(nu):
        B[6 + (k % nj), k] = 1.0

    # holonomic (loop-closure) Jacobian acting on the joints
    J_h = np.zeros((nh, nv))
    if nh:
        J_h[:, 6:] = rng.normal(0.0, 0.5, size=(nh, nj))

    # contact points: foot corners 0.9 m below the base
    J_c = np.zeros((nc, nv))
    for i in range(ncontact):
        sx = 1.0 if (i % 2 == 0) else -1.0
        sy = 1.0 if ((i // 2) % 2 == 0) else -1.0
        p = np.array([0.10 * sx + 0.02 * (i // 4), 0.15 * sy, -0.9]) + rng.normal(0.0, 0.01, 3)
        J_c[3 * i:3 * i + 3, 0:3] = -_skew(p)
        J_c[3 * i:3 * i + 3, 3:6] = np.eye(3)
        J_c[3 * i:3 * i + 3, 6:] = 0.1 * rng.normal(0.0, 1.0, size=(3, nj))
    # rows of J_c that carry a soft constraint (a line contact loses one row per point pair)
    act_rows = list(range(nc))
    drop = nc - ne
    for k in range(drop):
        act_rows.remove(6 * k + 5)
    J_c_act = J_c[act_rows, :]

    # bias: gravity on the base z translation (index 5) plus small Coriolis terms
    C = rng.normal(0.0, 0.5, nv)
    C[5] += 30.0 * 9.81

    o_u, o_h, o_c, o_e = nv, nv + nu, nv + nu + nh, nv + nu + nh + nc
    A = np.zeros((m, n))
    A[0:nv, 0:nv] = M
    A[0:nv, o_u:o_u + nu] = -B
    if nh:
        A[0:nv, o_h:o_h + nh] = -J_h.T
    A[0:nv, o_c:o_c + nc] = -J_c.T
    b_eq = np.zeros(m)
    b_eq[0:nv] = -C
    if nh:
        A[nv:nv + nh, 0:nv] = J_h
        b_eq[nv:nv + nh] = rng.normal(0.0, 0.1, nh)
    A[nv + nh:, 0:nv] = J_c_act
    A[nv + nh:, o_e:o_e + ne] = np.eye(ne)
    b_eq[nv + nh:] = rng.normal(0.0, 0.1, ne)

    # cost: base-pose task on vdot[0:6] + regularisation
    W = np.diag([20.0, 20.0, 5.0, 10.0, 10.0, 50.0])
    Jy = np.zeros((6, nv))
    Jy[:, 0:6] = np.eye(6)
    ydd = rng.normal(0.0, 0.2, 6)
    ydd[4] += activity                           # lateral push
    Q = np.zeros((n, n))
    Q[0:nv, 0:nv] = 2.0 * Jy.T @ W @ Jy + 2e-2 * np.eye(nv)
    Q[o_u:o_u + nu, o_u:o_u + nu] = 2e-5 * np.eye(nu)
    Q[o_h:o_e, o_h:o_e] = lambda_reg * np.eye(nh + nc)
    Q[o_e:, o_e:] = 2e4 * np.eye(ne)
    b = np.zeros(n)
    b[0:nv] = -2.0 * Jy.T @ W @ ydd

    lb = np.full(n, -np.inf)
    ub = np.full(n, np.inf)
    lim = rng.uniform(ulim[0], ulim[1], nu)
    lb[o_u:o_u + nu] = -lim
    ub[o_u:o_u + nu] = lim
    friction = np.full(ncontact, 0.6)
    return dict(Q=Q, b=b, A_eq=A, b_eq=b_eq, friction_coeffs=friction, lb=lb, ub=ub)


@dataclasses.dataclass
class Batch:
    """Instance-major batch tensors, the layout of the C ABI (include/fccqp.h):
    ``Q[B,n,n]  b[B,n]  A_eq[B,m,n]  b_eq[B,m]  friction_coeffs[B,nc/3]  lb[B,n]  ub[B,n]``."""

    Q: np.ndarray
    b: np.ndarray
    A_eq: np.ndarray
    b_eq: np.ndarray
    friction_coeffs: np.ndarray
    lb: np.ndarray
    ub: np.ndarray

    @property
    def batch(self) -> int:
        return self.Q.shape[0]

    def instance(self, i: int) -> Dict[str, np.ndarray]:
        return dict(Q=self.Q[i], b=self.b[i], A_eq=self.A_eq[i], b_eq=self.b_eq[i],
                    friction_coeffs=self.friction_coeffs[i], lb=self.lb[i], ub=self.ub[i])

    def arrays(self):
        return (self.Q, self.b, self.A_eq, self.b_eq, self.friction_coeffs, self.lb, self.ub)

    def slice(self, lo: int, hi: int) -> "Batch":
        return Batch(*(a[lo:hi] for a in self.arrays()))


def stack(qps: Iterable[Dict[str, np.ndarray]]) -> Batch:
    """Stack log-format dicts into instance-major contiguous float64 tensors."""
    qps = list(qps)
    def cat(key):
        return np.ascontiguousarray(np.stack([np.asarray(q[key], dtype=np.float64) for q in qps]))
    return Batch(cat("Q"), cat("b"), cat("A_eq"), cat("b_eq"), cat("friction_coeffs"),
                 cat("lb"), cat("ub"))


def workload(name: str, batch: int, shape: Shape = WALKING, seed0: int = 1000,
             distinct: Optional[int] = None) -> Batch:
    """Pinned synthetic workloads.

    ``W-sat``  every instance pushed at 4 m/s^2: all run exactly ``max_iter`` ADMM iterations under
               the harness options, so the work per solve is deterministic (headline workload).
    ``W-mix``  even instances as W-easy, odd ones as W-sat.
    ``W-easy`` all instances unpushed, wide torque limits, force regularisation 1e-2: most exit
               at iteration 0 (ADMM at rho = 5e-5 is bimodal: an instance either starts feasible or
               runs to max_iter -- the floor-plus-spikes shape of fccqp.pdf Fig. 1).

    ``W-psd``  W-sat with NO cost on the constraint forces (``lambda_reg = 0``): Q is only positive
               semi-definite, with exact zeros on its diagonal.  K stays non-singular, but the
               reference's diagonally pivoted LDLT fails on it and its presolve takes the COD branch
               (fcc_qp.cpp:164-166) -- the likely situation for a logged OSC QP whose cost has no
               force regularisation (hard part 6).

    ``distinct`` caps the number of distinct generated instances; the batch is then tiled from
    them into *separate* buffers (no dedup on the device) -- generation costs ~1 ms/instance.
    """
    k = batch if distinct is None else min(batch, distinct)
    qps: List[Dict[str, np.ndarray]] = []
    for i in range(k):
        seed = seed0 + i
        if name == "W-sat":
            qps.append(osc_qp(seed, shape, activity=4.0))
        elif name == "W-easy":
            qps.append(osc_qp(seed, shape, activity=0.0, ulim=(400.0, 600.0), lambda_reg=1e-2))
        elif name == "W-psd":
            qps.append(osc_qp(seed, shape, activity=4.0, lambda_reg=0.0))
        elif name == "W-mix":
            if i % 2 == 0:
                qps.append(osc_qp(seed, shape, activity=0.0, ulim=(400.0, 600.0), lambda_reg=1e-2))
            else:
                qps.append(osc_qp(seed, shape, activity=4.0))
        else:
            raise ValueError(f"unknown workload {name!r}")
    base = stack(qps)
    if k == batch:
        return base
    reps = -(-batch // k)
    return Batch(*(np.ascontiguousarray(np.tile(a, (reps,) + (1,) * (a.ndim - 1))[:batch])
                   for a in base.arrays()))


# --------------------------------------------------------------------------------------------
# Log files in the reference's format (fcc_qp_test.py:22-24): an .npz whose entry ``qps`` is an
# object array of dicts.
# --------------------------------------------------------------------------------------------
def save_qp_log(path: str, qps: Sequence[Dict[str, np.ndarray]]) -> None:
    arr = np.empty(len(qps), dtype=object)
    for i, q in enumerate(qps):
        arr[i] = dict(q)
    np.savez(path, qps=arr)


def load_qp_log(path: str) -> List[Dict[str, np.ndarray]]:
    """Read ``id_qp_log_<name>.npz`` exactly as the reference harness does."""
    data = np.load(path, allow_pickle=True)
    return list(data["qps"])


def gait_sequence(length: int, shape: Shape = WALKING, seed0: int = 5000) -> List[Dict[str, np.ndarray]]:
    """A synthetic stand-in for the walking log: pushes ramp 0 -> 4 -> 0 every 400 QPs."""
    out = []
    for i in range(length):
        phase = (i % 400) / 400.0
        act = 4.0 * (1.0 - abs(2.0 * phase - 1.0))
        out.append(osc_qp(seed0 + i, shape, activity=act))
    return out
