"""The algebra of the fused kernel's register-tile factorisation (fccqp_fused_impl.h: tile_factor_forward,
tile_ldl, gram_output with bordering, tile_factor_store), emulated tile by tile in numpy
(tools/tile_emulation.py) and checked against numpy.linalg on the three shapes compiled into the kernel:
walking (two bordered rows), mini (one), padded (identity-padded last tile row, no border).  CPU only;
the GPU tests check the kernel itself against the oracle."""
import importlib.util
import os

import numpy as np
import pytest

_spec = importlib.util.spec_from_file_location(
    "tile_emulation", os.path.join(os.path.dirname(__file__), "..", "tools", "tile_emulation.py"))
E = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(E)

SHAPES = {"walking": (60, 38), "mini": (12, 5), "border3": (12, 7), "padded": (12, 8), "n32-border": (32, 18),
          "n32-padded": (32, 14)}


def _qp(rng, n, m):
    g = rng.standard_normal((n, n))
    return g @ g.T + 0.1 * np.eye(n), rng.standard_normal((m, n)), rng.standard_normal(n), rng.standard_normal(m)


@pytest.mark.parametrize("name", SHAPES)
def test_border_rule(name):
    n, m = SHAPES[name]
    assert E.border_rows(n, m) == {"walking": 2, "mini": 1, "border3": 3, "padded": 0, "n32-border": 2, "n32-padded": 0}[name]


@pytest.mark.parametrize("name", SHAPES)
@pytest.mark.parametrize("rho", [5e-5, 0.1])
def test_operator_and_presolve_match_lapack(name, rho):
    n, m = SHAPES[name]
    rng = np.random.default_rng(n + m)
    for _ in range(3):
        Q, A, b, b_eq = _qp(rng, n, m)
        G, xc, piv = E.factor_forward_gram(Q, A, b, b_eq, rho)
        K = np.block([[Q + rho * np.eye(n), A.T], [A, np.zeros((m, m))]])
        Kinv = np.linalg.inv(K)
        r = np.r_[-b, b_eq]
        assert np.abs(G - Kinv[:n, :n]).max() <= 1e-11 * np.abs(Kinv[:n, :n]).max()
        assert np.abs(xc - (Kinv @ r)[:n]).max() <= 1e-11 * (1 + np.abs(Kinv @ r).max())
        # the pivots are those of the natural-order unpivoted LDL^T of K_rho (same elimination order)
        L = np.eye(n + m); D = K.copy()
        d = np.zeros(n + m)
        for k in range(n + m):
            d[k] = D[k, k]
            l = D[k + 1:, k] / d[k]
            D[k + 1:, k + 1:] -= np.outer(l, D[k, k + 1:])
        assert np.allclose(piv, d, rtol=1e-8, atol=1e-10)
        x0 = np.linalg.solve(np.block([[Q, A.T], [A, np.zeros((m, m))]]), r)[:n]
        assert np.abs(E.presolve(G, xc, rho) - x0).max() <= 1e-9 * (1 + np.abs(x0).max())


def test_only_lower_triangle_of_q_is_read():
    n, m = SHAPES["walking"]
    rng = np.random.default_rng(3)
    Q, A, b, b_eq = _qp(rng, n, m)
    junk = Q + np.triu(rng.standard_normal((n, n)), 1)      # garbage above the diagonal
    G1, x1, _ = E.factor_forward_gram(Q, A, b, b_eq, 1e-3)
    G2, x2, _ = E.factor_forward_gram(junk, A, b, b_eq, 1e-3)
    assert np.array_equal(G1, G2) and np.array_equal(x1, x2)


def test_duplicated_border_rows_give_an_exactly_zero_pivot():
    """The reference's true division makes a duplicated equality row eliminate to an exactly zero pivot
    (dropped by the pseudo-inverse rule); the kernel's multipliers are exact quotients for the same reason."""
    n, m = SHAPES["walking"]
    rng = np.random.default_rng(5)
    Q, A, b, b_eq = _qp(rng, n, m)
    A[-2] = A[-1]; b_eq[-2] = b_eq[-1]
    G, xc, piv = E.factor_forward_gram(Q, A, b, b_eq, 5e-5)
    assert piv[-1] == 0.0 and piv[-2] != 0.0
    assert np.all(np.isfinite(G)) and np.all(np.isfinite(xc))
    # consistent duplicate: the solution satisfies the (duplicated) constraints
    Ared = np.delete(A, -2, axis=0); bred = np.delete(b_eq, -2)
    K = np.block([[Q + 5e-5 * np.eye(n), Ared.T], [Ared, np.zeros((m - 1, m - 1))]])
    ref = np.linalg.solve(K, np.r_[-b, bred])[:n]
    assert np.abs(xc - ref).max() <= 1e-9 * (1 + np.abs(ref).max())
