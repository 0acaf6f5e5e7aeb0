"""Pins oracle/peak_finding.py to the reference's known-answer tests
(reference: tests/nn/test_peak_finding.py; line numbers cited per test)."""
import numpy as np
from numpy.testing import assert_allclose, assert_array_equal

from oracle import peak_finding as pf
from oracle.synth import make_confmaps, make_grid_vectors, make_multi_confmaps


def test_find_local_offsets():  # ref :28-46
    off = pf.find_offsets_local_direction(
        np.array([[0.0, 1.0, 0.0], [1.0, 3.0, 2.0], [0.0, 1.0, 0.0]]).reshape(1, 3, 3, 1), 0.25
    )
    assert off.shape == (1, 2)
    assert off[0][0] == 0.25 and off[0][1] == 0.0
    off = pf.find_offsets_local_direction(
        np.array([[0.0, 1.0, 0.0], [1.0, 3.0, 1.0], [0.0, 1.0, 0.0]]).reshape(1, 3, 3, 1), 0.25
    )
    assert off[0][0] == 0.0 and off[0][1] == 0.0


def test_find_global_peaks_rough():  # ref :49-73
    xv, yv = make_grid_vectors(8, 8, 1)
    points = np.array([[1, 2], [3, 4], [5, 6]], np.float32)
    cm = make_confmaps(points, xv, yv, sigma=1.0)
    points2 = points + 1
    cms = np.stack([cm, make_confmaps(points2, xv, yv, sigma=1.0)])
    peaks, vals = pf.find_global_peaks(cms, threshold=0.1, refinement=None)
    assert peaks.shape == (2, 3, 2) and vals.shape == (2, 3)
    assert_array_equal(peaks[0], points)
    assert_array_equal(vals[0], [1, 1, 1])
    assert_array_equal(peaks[1], points2)
    assert_array_equal(vals[1], [1, 1, 1])
    peaks, vals = pf.find_global_peaks_rough(np.zeros((1, 8, 8, 3), np.float32), threshold=0.1)
    assert peaks.shape == (1, 3, 2) and vals.shape == (1, 3)
    assert np.isnan(peaks).all()
    assert_array_equal(vals, [[0, 0, 0]])


def test_find_global_peaks_integral():  # ref :76-121
    xv, yv = make_grid_vectors(12, 12, 1)
    points = np.array([[1.5, 2.5], [3.5, 4.5], [5.5, 6.5]], np.float32)
    cm = make_confmaps(points, xv, yv, sigma=1.0)
    peaks, vals = pf.find_global_peaks(cm[None], threshold=0.1, refinement="integral", integral_patch_size=5)
    assert peaks.shape == (1, 3, 2)
    assert_allclose(peaks[0], points, atol=0.1)
    assert_allclose(vals[0], [1, 1, 1], atol=0.3)
    peaks, vals = pf.find_global_peaks(
        np.zeros((1, 8, 8, 3), np.float32), threshold=0.1, refinement="integral", integral_patch_size=5
    )
    assert np.isnan(peaks).all()
    assert_array_equal(vals, [[0, 0, 0]])
    peaks, vals = pf.find_global_peaks(
        np.stack([np.zeros((12, 12, 3), np.float32), cm]), threshold=0.1, refinement="integral"
    )
    assert peaks.shape == (2, 3, 2)
    assert np.isnan(peaks[0]).all()
    assert_allclose(peaks[1], points, atol=0.1)


def test_find_global_peaks_local():  # ref :124-138
    xv, yv = make_grid_vectors(12, 12, 1)
    points = np.array([[1.6, 2.6], [3.6, 4.6], [5.6, 6.6]], np.float32)
    cm = make_confmaps(points, xv, yv, sigma=1.0)
    peaks, vals = pf.find_global_peaks(cm[None], threshold=0.1, refinement="local")
    assert_allclose(peaks[0], np.array([[1.75, 2.75], [3.75, 4.75], [5.75, 6.75]]))
    assert_allclose(vals[0], [1, 1, 1], atol=0.3)


def _two_sample_cms(size, scale=1.0, shift=0.0):
    xv, yv = make_grid_vectors(size, size, 1)
    inst = np.array([[[1, 2], [3, 4]], [[5, 6], [7, 8]], [[np.nan, np.nan], [11, 12]]], np.float32) * scale + shift
    inst2 = np.array([[[2, 3], [4, 5]], [[6, 7], [8, 9]]], np.float32) * scale + shift
    return np.stack(
        [make_multi_confmaps(inst, xv, yv, 1.0), make_multi_confmaps(inst2, xv, yv, 1.0)], axis=0
    )


EXPECTED_PTS = np.array([[1, 2], [3, 4], [5, 6], [7, 8], [11, 12], [2, 3], [4, 5], [6, 7], [8, 9]], np.float32)


def test_find_local_peaks_rough():  # ref :141-198
    cms = _two_sample_cms(16)
    pts, vals, si, ci = pf.find_local_peaks(cms, threshold=0.1, refinement=None)
    assert pts.shape == (9, 2)
    assert_array_equal(pts, EXPECTED_PTS)
    assert_array_equal(vals, np.ones(9))
    assert_array_equal(si, [0, 0, 0, 0, 0, 1, 1, 1, 1])
    assert_array_equal(ci, [0, 1, 0, 1, 1, 0, 1, 0, 1])
    pts, vals, si, ci = pf.find_local_peaks(np.zeros([1, 4, 4, 3], np.float32), threshold=0.1)
    assert pts.shape == (0, 2) and vals.shape == (0,) and si.shape == (0,) and ci.shape == (0,)


def test_find_local_peaks_integral():  # ref :201-280
    cms = _two_sample_cms(32, 2.0, 0.3)
    pts, vals, si, ci = pf.find_local_peaks(cms, threshold=0.1, refinement="integral", integral_patch_size=5)
    assert pts.shape == (9, 2)
    assert_allclose(pts, EXPECTED_PTS * 2 + 0.3, atol=0.2)
    assert_allclose(vals, np.ones(9), atol=0.1)
    assert_array_equal(si, [0, 0, 0, 0, 0, 1, 1, 1, 1])
    assert_array_equal(ci, [0, 1, 0, 1, 1, 0, 1, 0, 1])
    pts, *_ = pf.find_local_peaks(np.zeros([1, 4, 4, 3], np.float32), refinement="integral")
    assert pts.shape == (0, 2)


def test_find_local_peaks_local():  # ref :283-337
    cms = _two_sample_cms(32, 2.0, 0.25)
    pts, vals, si, ci = pf.find_local_peaks(cms, threshold=0.1, refinement="local")
    assert_allclose(pts, EXPECTED_PTS * 2 + 0.25)
    assert_allclose(vals, np.ones(9), atol=0.1)
    assert_array_equal(ci, [0, 1, 0, 1, 1, 0, 1, 0, 1])


def test_offsets_variants():  # restates the intent of ref :340-391 with analytic offsets
    rng = np.random.default_rng(0)
    xv, yv = make_grid_vectors(32, 32, 1)
    inst = np.array([[[5.3, 6.2], [20.6, 9.4]], [[14.1, 25.7], [27.2, 22.9]]], np.float32)
    cms = make_multi_confmaps(inst, xv, yv, 1.5)[None]
    # learned offsets: (point - grid) at every pixel for the nearest point (grid units, [dx, dy])
    offs = np.zeros((1, 32, 32, 2, 2), np.float32)
    for c in range(2):
        for a in range(2):
            p = inst[a, c]
            yy, xx = np.mgrid[0:32, 0:32]
            near = (np.abs(xx - p[0]) < 3) & (np.abs(yy - p[1]) < 3)
            offs[0, :, :, c, 0][near] = (p[0] - xx)[near]
            offs[0, :, :, c, 1][near] = (p[1] - yy)[near]
    offs = offs.reshape(1, 32, 32, 4)
    pts, vals, si, ci = pf.find_local_peaks_with_offsets(cms, offs, threshold=0.2)
    got = {(int(c), tuple(np.round(p, 3))) for p, c in zip(pts, ci)}
    want = {(c, tuple(np.round(inst[a, c], 3))) for a in range(2) for c in range(2)}
    assert got == want
    # global variant on a single-instance map
    cms1 = make_confmaps(inst[0], xv, yv, 1.5)[None]
    g, gv = pf.find_global_peaks_with_offsets(cms1, offs, threshold=0.2)
    assert_allclose(g[0], inst[0], atol=1e-3)


def test_border_peaks_zero_extrapolation():
    # a peak in the corner: the 5x5 patch reads zeros outside the image (crop_and_resize extrapolation)
    cms = np.zeros((1, 8, 8, 1), np.float32)
    cms[0, 0, 0, 0] = 1.0
    cms[0, 0, 1, 0] = 0.5
    pts, vals, _, _ = pf.find_local_peaks(cms, threshold=0.2, refinement="integral", integral_patch_size=5)
    assert_allclose(pts, [[0.5 / 1.5, 0.0]], atol=1e-6)
    assert_array_equal(vals, [1.0])
