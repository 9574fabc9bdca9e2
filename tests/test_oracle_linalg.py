"""The oracle's restatement of the Eigen routines the reference calls
(LDLT compute/solve/info at /root/reference/src/fcc_qp.cpp:68,86,161,164,174; COD at :165,176)."""
import numpy as np
import pytest


def unpack(LD, transp):
    N = LD.shape[0]
    L = np.tril(LD, -1) + np.eye(N)
    D = np.diag(LD).copy()
    P = np.eye(N)
    for k in range(N):                       # P = T_{N-1} ... T_0
        if transp[k] != k:
            P[[k, transp[k]]] = P[[transp[k], k]]
    return L, D, P


@pytest.mark.parametrize("N", [1, 2, 5, 17, 98])
def test_ldlt_reconstructs_and_solves(oracle, N):
    rng = np.random.default_rng(N)
    G = rng.normal(size=(N, N))
    A = G @ G.T + 0.1 * np.eye(N)
    LD, tr, info = oracle.ldlt(A)
    assert info == 0
    L, D, P = unpack(LD, tr)
    assert np.allclose(P.T @ L @ np.diag(D) @ L.T @ P, A, rtol=1e-11, atol=1e-11)
    b = rng.normal(size=N)
    assert np.allclose(oracle.ldlt_solve(LD, tr, b), np.linalg.solve(A, b), rtol=1e-9, atol=1e-9)


def test_ldlt_reads_lower_triangle_only(oracle):
    rng = np.random.default_rng(1)
    G = rng.normal(size=(6, 6)); A = G @ G.T + np.eye(6)
    junk = A.copy(); junk[np.triu_indices(6, 1)] = 1e9        # Eigen LDLT<.,Lower> never looks there
    assert np.array_equal(oracle.ldlt(A)[0][np.tril_indices(6)], oracle.ldlt(junk)[0][np.tril_indices(6)])


def test_ldlt_pivot_order_is_descending_original_diagonal(oracle):
    """The pivot search is left-looking: it sees the not-yet-updated diagonal, so the first pivots
    are the original diagonal entries in descending |.| order."""
    A = np.diag([1.0, 5.0, 3.0, 4.0]) + 0.01
    _, tr, info = oracle.ldlt(A)
    assert info == 0 and list(tr) == [1, 3, 2, 3]


def test_ldlt_on_kkt_matrix_with_zero_block(oracle):
    rng = np.random.default_rng(2)
    n, m = 8, 3
    G = rng.normal(size=(n, n)); Q = G @ G.T + np.eye(n); A = rng.normal(size=(m, n))
    K = np.block([[Q, A.T], [A, np.zeros((m, m))]])
    LD, tr, info = oracle.ldlt(K)
    assert info == 0
    L, D, P = unpack(LD, tr)
    assert (D[:n] > 0).all() and (D[n:] < 0).all()            # quasi-definite: n positive, m negative pivots
    r = rng.normal(size=n + m)
    assert np.allclose(oracle.ldlt_solve(LD, tr, r), np.linalg.solve(K, r), rtol=1e-9, atol=1e-9)


def test_ldlt_info_rules(oracle):
    assert oracle.ldlt(np.zeros((4, 4)))[2] == 0               # zero matrix: Success
    Z = np.zeros((3, 3)); Z[2, 0] = Z[0, 2] = 1.0              # zero diagonal, non-zero off-diagonal
    assert oracle.ldlt(Z)[2] == 1                              # NumericalIssue
    S = np.diag([2.0, 0.0, 0.0])                               # rank 1, consistent zeros: Success
    assert oracle.ldlt(S)[2] == 0
    # a non-zero pivot after a zero pivot -> NumericalIssue: PSD Q with a free direction + constraints
    K = np.array([[1.0, 0, 0, 1], [0, 0, 0, 0], [0, 0, 0, 1.0], [1, 0, 1.0, 0]])
    assert oracle.ldlt(K)[2] == 1


def test_ldlt_solve_pseudo_inverse_rule_on_zero_pivots(oracle):
    A = np.diag([2.0, 0.0, 4.0])
    LD, tr, info = oracle.ldlt(A)
    x = oracle.ldlt_solve(LD, tr, np.array([2.0, 5.0, 8.0]))
    assert np.array_equal(x, np.array([1.0, 0.0, 2.0]))        # |d| <= DBL_MIN -> component set to 0


@pytest.mark.parametrize("rank_def", [0, 1, 3])
def test_cod_is_minimum_norm_least_squares(oracle, rank_def):
    rng = np.random.default_rng(10 + rank_def)
    N = 12
    B = rng.normal(size=(N, N - rank_def))
    A = B @ B.T                                                # symmetric, rank N - rank_def
    b = A @ rng.normal(size=N)                                 # consistent
    x, rank = oracle.cod_solve(A, b)
    assert rank == N - rank_def
    assert np.allclose(x, np.linalg.pinv(A) @ b, rtol=1e-8, atol=1e-8)
    b2 = rng.normal(size=N)                                    # inconsistent: least squares + min norm
    x2, _ = oracle.cod_solve(A, b2)
    assert np.allclose(x2, np.linalg.lstsq(A, b2, rcond=None)[0], rtol=1e-7, atol=1e-7)
