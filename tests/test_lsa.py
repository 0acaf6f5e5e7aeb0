"""csrc/lsa.h (host build through sa_lsa_host) against scipy.optimize.linear_sum_assignment,
the solver the reference calls (sleap/nn/utils.py:79-98)."""
import numpy as np
import pytest
from scipy.optimize import linear_sum_assignment

from sleap_amd.ops import lsa_host


@pytest.mark.parametrize("seed", range(5))
def test_random_rectangular(seed):
    rng = np.random.default_rng(seed)
    for _ in range(200):
        nr, nc = rng.integers(1, 12, size=2)
        cost = rng.normal(size=(nr, nc))
        r, c = linear_sum_assignment(cost)
        r2, c2 = lsa_host(cost)
        np.testing.assert_array_equal(r, r2)
        np.testing.assert_array_equal(c, c2)


def test_ties_small_integers():
    rng = np.random.default_rng(1)
    for _ in range(500):
        nr, nc = rng.integers(1, 8, size=2)
        cost = rng.integers(0, 3, size=(nr, nc)).astype(np.float64)
        r, c = linear_sum_assignment(cost)
        r2, c2 = lsa_host(cost)
        np.testing.assert_array_equal(r, r2)
        np.testing.assert_array_equal(c, c2)


def test_constant_matrix_identity():
    r, c = lsa_host(np.ones((4, 4)))
    np.testing.assert_array_equal(c, [0, 1, 2, 3])


def test_inf_entries_and_infeasible():
    cost = np.array([[np.inf, 1.0], [2.0, np.inf]])
    r, c = lsa_host(cost)
    np.testing.assert_array_equal(c, [1, 0])
    with pytest.raises(ValueError):
        lsa_host(np.array([[np.inf, np.inf], [1.0, 2.0]]))
    with pytest.raises(ValueError):
        linear_sum_assignment(np.array([[np.inf, np.inf], [1.0, 2.0]]))


def test_reference_case():  # tests/nn/test_paf_grouping.py:132-160 (scores [-0.5, 1.0] as 2x1)
    r, c = lsa_host(-np.array([[-0.5], [1.0]]))
    np.testing.assert_array_equal(r, [1])
    np.testing.assert_array_equal(c, [0])


# ---- the wave-cooperative form of the solver (csrc/lsa.h: lsa_solve_wave -- what the matching kernel runs, one wavefront per
# (frame, edge)), with its 64 lanes emulated on the host: same answers as SciPy, ties included, also beyond 64 columns
@pytest.mark.parametrize("seed", range(4))
def test_wave_solver_random(seed):
    rng = np.random.default_rng(100 + seed)
    for _ in range(150):
        nr, nc = rng.integers(1, 14, size=2)
        cost = rng.normal(size=(nr, nc))
        r, c = linear_sum_assignment(cost)
        r2, c2 = lsa_host(cost, wave=True)
        np.testing.assert_array_equal(r, r2)
        np.testing.assert_array_equal(c, c2)


def test_wave_solver_ties():
    rng = np.random.default_rng(7)
    for _ in range(1500):
        nr, nc = rng.integers(1, 9, size=2)
        cost = rng.integers(0, 3, size=(nr, nc)).astype(np.float64)
        r, c = linear_sum_assignment(cost)
        r2, c2 = lsa_host(cost, wave=True)
        np.testing.assert_array_equal(r, r2)
        np.testing.assert_array_equal(c, c2)


def test_wave_solver_wider_than_a_wave_and_inf():
    rng = np.random.default_rng(8)
    for shape in ((70, 130), (130, 70), (64, 64), (65, 3), (1, 200)):
        for ties in (False, True):
            cost = rng.integers(0, 4, size=shape).astype(np.float64) if ties else rng.normal(size=shape)
            r, c = linear_sum_assignment(cost)
            r2, c2 = lsa_host(cost, wave=True)
            np.testing.assert_array_equal(r, r2)
            np.testing.assert_array_equal(c, c2)
    cost = rng.normal(size=(6, 9))
    cost[rng.random(cost.shape) < 0.3] = np.inf
    cost[2, :] = np.inf
    with pytest.raises(ValueError):
        lsa_host(cost, wave=True)
    cost[2, 4] = 0.5
    r, c = linear_sum_assignment(cost)
    r2, c2 = lsa_host(cost, wave=True)
    np.testing.assert_array_equal(c, c2)
