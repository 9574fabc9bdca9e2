"""Known-answer tests for the oracle's restatement of /root/reference/src/constraint_utils.cpp.
Every branch of project_to_friction_cone (:5-25) plus the edge cases listed in SURVEY.md 8a/P1."""
import numpy as np
import pytest

from oracle import numpy_ref as R


def both(oracle, f, mu):
    a = oracle.project_cone3(f, mu)
    b = R.project_cone3(f, mu)
    # two independent restatements agree to the last bits (numpy's dot may fuse/reorder the sums)
    assert np.max(np.abs(a - b)) <= 2e-14 * max(np.max(np.abs(a)), 1e-300), (a, b)
    return a


def test_inside_cone_is_identity(oracle):                       # :10-12
    f = np.array([0.3, -0.4, 1.0])                              # |f_xy| = 0.5 <= 0.6 * 1
    assert np.array_equal(both(oracle, f, 0.6), f)


def test_boundary_counts_as_inside(oracle):                     # '>=' at :10
    f = np.array([0.6, 0.0, 1.0])
    assert np.array_equal(both(oracle, f, 0.6), f)


def test_polar_cone_maps_to_origin(oracle):                     # :15-17
    f = np.array([0.1, 0.0, -1.0])                              # f_z < -mu |f_xy|
    assert np.array_equal(both(oracle, f, 0.6), np.zeros(3))


def test_side_projection_matches_closed_form(oracle):           # :20-23
    mu = 0.5
    f = np.array([3.0, 4.0, 1.0])                               # |f_xy| = 5 > mu f_z
    p = both(oracle, f, mu)
    # Euclidean projection onto the second-order cone {|p_xy| <= mu p_z}
    r = 5.0
    t = (mu * r + f[2]) / (1 + mu * mu)
    expect = np.array([mu * t * f[0] / r, mu * t * f[1] / r, t])
    assert np.allclose(p, expect, rtol=1e-14, atol=0)
    assert abs(np.hypot(p[0], p[1]) - mu * p[2]) < 1e-14       # lands on the cone surface


def test_projection_is_idempotent_and_feasible(oracle):
    rng = np.random.default_rng(0)
    for _ in range(200):
        f = rng.normal(size=3) * rng.choice([0.1, 1.0, 100.0])
        mu = rng.uniform(0.05, 1.5)
        p = both(oracle, f, mu)
        assert np.hypot(p[0], p[1]) <= mu * p[2] + 1e-12 * (1 + np.abs(p).max())
        assert np.allclose(both(oracle, p, mu), p, rtol=1e-12, atol=1e-12)


def test_mu_zero_projects_onto_z_axis(oracle):
    p = both(oracle, np.array([1.0, 2.0, 3.0]), 0.0)            # ratio = 0 -> ray (0,0,f_z)
    assert np.array_equal(p, np.array([0.0, 0.0, 3.0]))
    assert np.array_equal(both(oracle, np.array([1.0, 2.0, -3.0]), 0.0), np.zeros(3))


def test_zero_tangential_force(oracle):
    assert np.array_equal(both(oracle, np.array([0.0, 0.0, 2.0]), 0.6), np.array([0.0, 0.0, 2.0]))
    assert np.array_equal(both(oracle, np.array([0.0, 0.0, -2.0]), 0.6), np.zeros(3))


def test_fz_exactly_zero_quirk(oracle):
    """f_z == 0 with |f_xy| > 0: neither early branch fires, the ray is the zero vector, Eigen's
    normalize() leaves it untouched and the result is 0 -- NOT the true projection
    (mu^2/(1+mu^2) f_xy, ...).  Reproduced on purpose."""
    p = both(oracle, np.array([1.0, 0.0, 0.0]), 0.6)
    assert np.array_equal(p, np.zeros(3))


def test_stacked_projection_and_too_few_coefficients(oracle):   # :27-35
    f = np.array([3.0, 4.0, 1.0, 0.1, 0.0, 1.0])
    out = oracle.project_cones(f, [0.5, 0.6])
    assert np.array_equal(out[:3], oracle.project_cone3(f[:3], 0.5))
    assert np.array_equal(out[3:], f[3:])
    assert np.array_equal(out, R.project_cones(f, [0.5, 0.6]))
    with pytest.raises(IndexError):                             # .at(i) throws std::out_of_range (:32)
        oracle.project_cones(f, [0.5])
    with pytest.raises(IndexError):
        R.project_cones(f, [0.5])


def test_project_to_bounds_order_of_min_max(oracle):            # :37-46  max(min(x, ub), lb)
    x = np.array([-2.0, 0.5, 3.0, 7.0])
    lb = np.array([-1.0, -np.inf, 0.0, 8.0])
    ub = np.array([1.0, np.inf, 2.0, 6.0])                      # last entry: lb > ub -> lb wins
    out = oracle.project_bounds(x, lb, ub)
    assert np.array_equal(out, np.array([-1.0, 0.5, 2.0, 8.0]))
    assert np.array_equal(out, R.project_bounds(x, lb, ub))


def test_violation_metrics(oracle):                             # :48-65
    f = np.array([3.0, 4.0, 1.0, 0.1, 0.0, 1.0, 0.0, 1.0, -1.0])
    mus = [0.5, 0.6, 0.7]
    # sum over contacts of max(0, |f_xy| - mu f_z): 4.5 + 0 + 1.7
    assert oracle.cone_violation(f, mus) == pytest.approx(4.5 + 1.7, rel=1e-15)
    assert oracle.cone_violation(f, mus) == pytest.approx(R.cone_violation(f, mus), rel=1e-15)
    with pytest.raises(IndexError):
        oracle.cone_violation(f, mus[:2])                       # .at(i) at :55
    x = np.array([2.0, -3.0, 0.0]); lb = np.array([-1.0, -1.0, -1.0]); ub = np.array([1.0, 1.0, 1.0])
    assert oracle.bound_violation(x, lb, ub) == pytest.approx(np.sqrt(1 + 4), rel=1e-15)   # 2-norm
    assert oracle.bound_violation(x, lb, ub) == pytest.approx(R.bound_violation(x, lb, ub), rel=1e-15)


def test_validate_bounds(oracle):                               # :67-75
    assert oracle.validate_bounds([0, -np.inf], [1, np.inf])
    assert oracle.validate_bounds([1.0], [1.0])
    assert not oracle.validate_bounds([0, 2.0], [1, 1.0])
