"""The oracle's restatements of the reference's data-side functions -- preprocessing (normalization.py, resizing.py), box
cropping (instance_cropping.py, peak_finding.py:135-190) and the synthetic confidence-map / PAF generators the parity tests
feed on (confidence_maps.py, edge_maps.py, data/utils.py) -- pinned to the reference's own known-answer tests
(tests/nn/data/test_{normalization,resizing,instance_cropping,confidence_maps,edge_maps,utils}.py; line numbers below)."""
import numpy as np
from numpy.testing import assert_allclose, assert_array_equal

from oracle import keras_graph as kg
from oracle import peak_finding as pf
from oracle import synth


# ------------------------------------------------------------------------------------------ test_normalization.py
def test_ensure_float():  # ref :17-19, and the 1/255 of convert_image_dtype
    assert kg.ensure_float(np.zeros((1, 2, 2, 1), np.uint8)).dtype == np.float32
    assert kg.ensure_float(np.zeros((1, 2, 2, 1), np.float32)).dtype == np.float32
    assert_array_equal(kg.ensure_float(np.full((1, 1, 1, 1), 255, np.uint8)), np.float32(255) * np.float32(1 / 255))


def test_ensure_grayscale():  # ref :37-50
    assert_array_equal(kg.ensure_grayscale(np.full((1, 2, 2, 3), 255, np.uint8)), np.full((1, 2, 2, 1), 255, np.uint8))
    assert_array_equal(kg.ensure_grayscale(np.full((1, 2, 2, 1), 255, np.uint8)), np.full((1, 2, 2, 1), 255, np.uint8))
    assert_allclose(kg.ensure_grayscale(np.ones((1, 2, 2, 3), np.float32)), np.ones((1, 2, 2, 1), np.float32), atol=1e-4)


def test_ensure_rgb():  # ref :53-61
    assert_array_equal(kg.ensure_rgb(np.full((1, 2, 2, 3), 255, np.uint8)), np.full((1, 2, 2, 3), 255, np.uint8))
    assert_array_equal(kg.ensure_rgb(np.full((1, 2, 2, 1), 255, np.uint8)), np.full((1, 2, 2, 3), 255, np.uint8))


# ------------------------------------------------------------------------------------------ test_resizing.py
def test_pad_to_stride():  # ref :13-49 (find_padding_for_stride: (127, 129) @ 32 -> (1, 31); (128, 128) -> (0, 0))
    assert kg.pad_to_stride(np.ones((1, 127, 129, 1), np.uint8), 32).shape == (1, 128, 160, 1)
    assert kg.pad_to_stride(np.ones((1, 128, 128, 1), np.uint8), 32).shape == (1, 128, 128, 1)
    want = np.array([[1, 1, 1, 1, 1, 0], [1, 1, 1, 1, 1, 0], [1, 1, 1, 1, 1, 0], [0, 0, 0, 0, 0, 0]])[None, ..., None]
    for dt in (np.float32, np.uint8):
        y = kg.pad_to_stride(np.ones((1, 3, 5, 1), dt), 2)
        assert y.dtype == dt
        assert_array_equal(y, want)
    assert kg.pad_to_stride(np.ones((1, 4, 4, 1), np.float32), 2).shape == (1, 4, 4, 1)


def test_resize_image():  # ref :52-66 (scalar scales; the [0.25, 3] list form is not on the inference path)
    for dt in (np.uint8, np.float32):
        y = kg.resize_image(np.ones((1, 4, 8, 1), dt), 0.5)
        assert y.shape == (1, 2, 4, 1) and y.dtype == dt
    # int(W * scale) truncation (resizing.py:93-97)
    assert kg.resize_image(np.ones((1, 5, 7, 1), np.float32), 0.5).shape == (1, 2, 3, 1)


# ------------------------------------------------------------------------------------------ test_instance_cropping.py
def test_normalize_bboxes():  # ref :14-20
    assert_array_equal(pf.normalize_bboxes(np.array([[0, 0, 3, 3]], np.float32), 9, 9), [[0, 0, 0.375, 0.375]])


def test_make_centered_bboxes():  # ref :23-32
    assert_array_equal(pf.make_centered_bboxes(np.array([[1, 1]], np.float32), 3, 3), [[0, 0, 2, 2]])
    assert_array_equal(pf.make_centered_bboxes(np.array([[2, 2]], np.float32), 4, 4), [[0.5, 0.5, 3.5, 3.5]])


def test_crop_bboxes():  # ref :35-52 (the reference crops one image; the restatement takes a batch + sample indices)
    XX, YY = np.meshgrid(np.arange(4, dtype=np.uint8), np.arange(5, dtype=np.uint8))
    img = np.stack([XX, YY], axis=-1)
    bboxes = pf.make_centered_bboxes(np.array([[1, 1]], np.float32), 3, 3)
    crops = pf.crop_bboxes(img[None].astype(np.float32), bboxes, np.zeros(1, np.int32))
    assert_array_equal(crops, img[:3, :3, :][None])


def test_crop_bboxes_rounding():  # ref :55-63: box size = round(y2 - y1 + 1), not truncation
    bboxes = pf.make_centered_bboxes(np.array([[464.42838, 550.14276]], np.float32), 100, 100)
    crops = pf.crop_bboxes(np.zeros((1, 16, 16, 1), np.float32), bboxes, np.zeros(1, np.int32))
    assert crops.shape == (1, 100, 100, 1)


# ------------------------------------------------------------------------------------------ test_utils.py
def test_make_grid_vectors():  # ref :38-51
    xv, yv = synth.make_grid_vectors(4, 3, 1)
    assert xv.dtype == np.float32 and yv.dtype == np.float32
    assert_allclose(xv, [0, 1, 2])
    assert_allclose(yv, [0, 1, 2, 3])
    xv, yv = synth.make_grid_vectors(4, 3, 2)
    assert_allclose(xv, [0, 2])
    assert_allclose(yv, [0, 2])


def test_gaussian_pdf():  # ref :54-57 (float32 arithmetic as in TF)
    assert synth.gaussian_pdf(0, 1) == 1.0
    assert_allclose(synth.gaussian_pdf(1, 1), 0.6065306597126334, rtol=1e-7)
    assert_allclose(synth.gaussian_pdf(1, 2), 0.8824969025845955, rtol=1e-7)


# ------------------------------------------------------------------------------------------ test_confidence_maps.py
def test_make_confmaps():  # ref :21-85
    xv, yv = synth.make_grid_vectors(4, 5, 1)
    cm = synth.make_confmaps(np.array([[0.5, 1.0], [3, 3.5], [2.0, 2.0]], np.float32), xv, yv, 1.0)
    assert cm.dtype == np.float32 and cm.shape == (4, 5, 3)
    assert_allclose(cm, [
        [[0.535, 0.0, 0.018], [0.535, 0.0, 0.082], [0.197, 0.001, 0.135], [0.027, 0.002, 0.082], [0.001, 0.001, 0.018]],
        [[0.882, 0.0, 0.082], [0.882, 0.006, 0.368], [0.325, 0.027, 0.607], [0.044, 0.044, 0.368], [0.002, 0.027, 0.082]],
        [[0.535, 0.004, 0.135], [0.535, 0.044, 0.607], [0.197, 0.197, 1.0], [0.027, 0.325, 0.607], [0.001, 0.197, 0.135]],
        [[0.119, 0.01, 0.082], [0.119, 0.119, 0.368], [0.044, 0.535, 0.607], [0.006, 0.882, 0.368], [0.0, 0.535, 0.082]],
    ], atol=1e-3)
    cm = synth.make_confmaps(np.array([[2, 3]], np.float32), xv, yv, 1.0)  # grid aligned peak
    assert cm.shape == (4, 5, 1) and cm[3, 2] == 1.0
    xv, yv = synth.make_grid_vectors(8, 8, 2)  # output stride
    cm = synth.make_confmaps(np.array([[2, 4]], np.float32), xv, yv, 1.0)
    assert cm.shape == (4, 4, 1) and cm[2, 1] == 1.0
    cmn = synth.make_confmaps(np.array([[2, 4], [np.nan, np.nan]], np.float32), xv, yv, 1.0)  # missing points
    assert cmn.shape == (4, 4, 2) and cmn.dtype == np.float32
    assert_array_equal(cmn[:, :, 0], cm[:, :, 0])
    assert (cmn[:, :, 1] == 0).all()


def test_make_multi_confmaps():  # ref :88-108
    xv, yv = synth.make_grid_vectors(4, 5, 1)
    inst = np.array([[[0.5, 1.0], [2.0, 2.0]], [[1.5, 1.0], [2.0, 3.0]], [[np.nan, np.nan], [-1.0, 5.0]]], np.float32)
    cms = synth.make_multi_confmaps(inst, xv, yv, 1.0)
    assert cms.shape == (4, 5, 2) and cms.dtype == np.float32
    each = [synth.make_confmaps(i, xv, yv, 1.0) for i in inst]
    assert_array_equal(cms, np.max(np.stack(each, axis=-1), axis=-1))


# ------------------------------------------------------------------------------------------ test_edge_maps.py
SRC = np.array([[1, 0.5], [0, 0]], np.float32)
DST = np.array([[1, 1.5], [2, 2]], np.float32)


def test_distance_to_edge():  # ref :12-31 (squared distances)
    xv, yv = synth.make_grid_vectors(3, 3, 1)
    grid = np.stack(np.meshgrid(xv, yv), axis=-1)
    assert_allclose(synth.distance_to_edge(grid, SRC, DST), [
        [[1.25, 0.0], [0.25, 0.5], [1.25, 2.0]],
        [[1.0, 0.5], [0.0, 0.0], [1.0, 0.5]],
        [[1.25, 2.0], [0.25, 0.5], [1.25, 0.0]]], atol=1e-3)


def test_edge_confidence_map():  # ref :34-56: make_edge_maps = gaussian_pdf(distance_to_edge)
    xv, yv = synth.make_grid_vectors(3, 3, 1)
    grid = np.stack(np.meshgrid(xv, yv), axis=-1)
    assert_allclose(synth.gaussian_pdf(synth.distance_to_edge(grid, SRC, DST), 1.0), [
        [[0.458, 1.000], [0.969, 0.882], [0.458, 0.135]],
        [[0.607, 0.882], [1.000, 1.000], [0.607, 0.882]],
        [[0.458, 0.135], [0.969, 0.882], [0.458, 1.000]]], atol=1e-3)


PAFS = [
    [[[0.0, 0.458], [0.707, 0.707]], [[0.0, 0.969], [0.624, 0.624]], [[0.0, 0.458], [0.096, 0.096]]],
    [[[0.0, 0.607], [0.624, 0.624]], [[0.0, 1.0], [0.707, 0.707]], [[0.0, 0.607], [0.624, 0.624]]],
    [[[0.0, 0.458], [0.096, 0.096]], [[0.0, 0.969], [0.624, 0.624]], [[0.0, 0.458], [0.707, 0.707]]]]


def test_make_pafs():  # ref :59-93
    xv, yv = synth.make_grid_vectors(3, 3, 1)
    assert_allclose(synth.make_pafs(xv, yv, SRC, DST, 1.0), PAFS, atol=1e-3)


def test_make_multi_pafs():  # ref :96-142: two identical instances sum to twice the single-instance field
    xv, yv = synth.make_grid_vectors(3, 3, 1)
    pafs = synth.make_multi_pafs(xv, yv, np.stack([SRC, SRC]), np.stack([DST, DST]), 1.0)
    assert_allclose(pafs, 2 * np.asarray(PAFS), atol=2e-3)
    assert_allclose(pafs[1, 1], [[0.0, 2.0], [1.414, 1.414]], atol=1e-3)
