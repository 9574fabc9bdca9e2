"""Synthetic problem generator and the reference's log format (fcc_qp_amd/problems.py)."""
import numpy as np

from fcc_qp_amd import problems as P


def test_walking_shape_is_the_reference_harness_shape():
    """/root/reference/fcc_qp_test.py:77  FCCQP(60, 38, 12, 38); slices at :52-56"""
    assert P.WALKING.ctor_args == (60, 38, 12, 38)
    assert P.LARGE.ctor_args == (114, 60, 24, 66)
    assert P.HARNESS_OPTIONS == dict(max_iter=100, rho=5e-5, eps_fcone=1e-6, eps_bound=1e-6)   # :78-83


def test_instances_are_reproducible_and_well_formed():
    a = P.osc_qp(1234); b = P.osc_qp(1234)
    for k in a:
        assert np.array_equal(a[k], b[k])
    n, m, nc, s = P.WALKING.ctor_args
    assert a["Q"].shape == (n, n) and a["A_eq"].shape == (m, n) and a["friction_coeffs"].shape == (nc // 3,)
    assert np.allclose(a["Q"], a["Q"].T) and np.linalg.eigvalsh(a["Q"]).min() > 0
    assert np.linalg.matrix_rank(a["A_eq"]) == m                       # full row rank (fcc_qp.hpp:107-110)
    assert np.all(np.isinf(a["lb"][s:s + nc])) and np.all(np.isinf(a["ub"][s:s + nc]))   # fcc_qp.hpp:99-105
    assert np.all(a["lb"] <= a["ub"])


def test_batch_layout_is_instance_major_contiguous():
    B = P.workload("W-mix", 5, P.WALKING)
    n, m, nc, s = P.WALKING.ctor_args
    assert B.Q.shape == (5, n, n) and B.A_eq.shape == (5, m, n) and B.friction_coeffs.shape == (5, nc // 3)
    for a in B.arrays():
        assert a.flags["C_CONTIGUOUS"] and a.dtype == np.float64
    assert np.array_equal(B.instance(3)["Q"], P.osc_qp(1003, P.WALKING, 4.0)["Q"])    # seed = 1000 + i
    T = P.workload("W-sat", 10, P.WALKING, distinct=4)
    assert np.array_equal(T.Q[0], T.Q[4]) and np.array_equal(T.Q[1], T.Q[9])


def test_log_round_trip_in_reference_format(tmp_path):
    """np.load(path, allow_pickle=True)['qps'] -> object array of dicts (fcc_qp_test.py:22-30)"""
    qps = P.gait_sequence(3)
    path = str(tmp_path / "id_qp_log_walking.npz")
    P.save_qp_log(path, qps)
    raw = np.load(path, allow_pickle=True)["qps"]
    assert raw.dtype == object and set(raw[0].keys()) == {"Q", "b", "A_eq", "b_eq", "friction_coeffs", "lb", "ub"}
    back = P.load_qp_log(path)
    assert len(back) == 3 and np.array_equal(back[2]["A_eq"], qps[2]["A_eq"])
    B = P.stack(back)
    assert B.batch == 3
