"""oracle/inference.py against the reference's identity-model layer tests (tests/nn/test_inference.py:214-254
CentroidCrop, :257-379 FindInstancePeaks, :542-589 SingleInstanceInferenceLayer): with an identity network the
layer output is a pure function of the confidence maps, which is what the oracle restates."""
import numpy as np
from numpy.testing import assert_allclose, assert_array_equal

from oracle import inference as oinf
from oracle.synth import make_confmaps, make_grid_vectors


def _cms(points, size=12, stride=1):
    xv, yv = make_grid_vectors(size, size, stride)
    return make_confmaps(np.asarray(points, np.float32), xv, yv, sigma=1.0)


def test_single_instance_layer():  # ref :542-589: peaks == points for on-grid maxima, NaN below threshold
    pts = np.array([[1, 2], [3, 4], [5, 6]], np.float32)
    cms = np.stack([_cms(pts), _cms(pts + 1)])
    peaks, vals = oinf.single_instance_peaks(cms, None, 0.2, None, 5, 1, 1.0)
    assert peaks.shape == (2, 1, 3, 2) and vals.shape == (2, 1, 3)
    assert_array_equal(peaks[0, 0], pts)
    assert_array_equal(peaks[1, 0], pts + 1)
    peaks, _ = oinf.single_instance_peaks(cms, None, 0.2, None, 5, 2, 0.5)  # stride 2, input scale 0.5
    assert_array_equal(peaks[0, 0], pts * 2 / 0.5 + 0.5)
    peaks, vals = oinf.single_instance_peaks(np.zeros((1, 8, 8, 3), np.float32), None, 0.2, "integral", 5, 1, 1.0)
    assert np.isnan(peaks).all() and (vals == 0).all()


def test_centroid_crop_local_refinement():  # ref :214-254: local refinement gives [1.75, 2.75] ...
    pts = np.array([[1.6, 2.6], [3.6, 4.6], [5.6, 6.6]], np.float32)
    cms = np.stack([_cms(pts[i : i + 1]) for i in range(3)])  # one centroid channel per sample
    imgs = (np.arange(3 * 12 * 12, dtype=np.float32).reshape(3, 12, 12, 1) % 251).astype(np.uint8)
    out = oinf.centroid_crop(imgs, cms, None, 0.2, "local", 5, 1, 1.0, crop_size=4)
    assert_allclose(out["centroids"], [[1.75, 2.75], [3.75, 4.75], [5.75, 6.75]])
    assert_array_equal(out["crop_sample_inds"], [0, 1, 2])
    assert_allclose(out["crop_offsets"], out["centroids"] - 2.0)
    assert out["crops"].shape == (3, 4, 4, 1) and out["crops"].dtype == np.uint8
    # a crop centred on an integer+0.5 position with even size reproduces the pixels exactly
    out2 = oinf.centroid_crop(imgs, np.stack([_cms([[3.5, 4.5]])] * 1), None, 0.2, None, 5, 1, 1.0, crop_size=4)
    # rough peak of a point at (3.5, 4.5) is one of the 4 neighbours; just check shapes / dtype here
    assert out2["crops"].shape[1:] == (4, 4, 1)


def test_centroid_crop_max_instances_keeps_top_values():
    a = _cms([[2, 2]], 16)[..., 0] * 0.9
    b = _cms([[8, 8]], 16)[..., 0] * 0.5
    c = _cms([[12, 4]], 16)[..., 0] * 0.7
    cms = np.maximum(np.maximum(a, b), c)[None, ..., None].astype(np.float32)
    imgs = np.zeros((1, 16, 16, 1), np.uint8)
    out = oinf.centroid_crop(imgs, cms, None, 0.2, None, 5, 1, 1.0, crop_size=4, max_instances=2)
    assert_array_equal(out["centroids"], [[2, 2], [12, 4]])  # ordered by value: 0.9, 0.7
    assert_allclose(out["centroid_vals"], [0.9, 0.7], rtol=1e-6)


def test_find_instance_peaks_offsets_and_scale():  # ref :257-379
    pts = np.array([[1, 2], [3, 4]], np.float32)
    cms = np.stack([_cms(pts), _cms(pts + 2)])
    off = np.array([[10, 20], [30, 40]], np.float32)
    peaks, vals = oinf.find_instance_peaks(cms, None, off, 0.2, None, 5, 1, 1.0)
    assert_array_equal(peaks[0], pts + off[0])
    assert_array_equal(peaks[1], pts + 2 + off[1])
    peaks, _ = oinf.find_instance_peaks(cms, None, off, 0.2, None, 5, 2, 0.5)
    assert_array_equal(peaks[0], pts * 2 / 0.5 + 0.5 + off[0] / 0.5)
