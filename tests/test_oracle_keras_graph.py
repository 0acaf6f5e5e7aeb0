"""Structure checks for the Keras-graph oracle on the reference's fixture models.

The reference pins only structure for the network (tests/nn/architectures/test_unet.py:12-158:
layer counts / parameter counts / output shapes), not numerics, so this is what can be pinned.
Parameter counts are those of the fixture models themselves.
"""
import json
import os

import numpy as np
import pytest

from oracle.keras_graph import KerasGraph, load_npz_model, preprocess, _same_pads

MODELS = os.path.join(os.path.dirname(__file__), "golden", "models")

CASES = [
    ("minimal_instance.UNet.bottomup", 150748, (384, 384, 1), [(192, 192, 2), (96, 96, 2), (192, 192, 4)]),
    ("minimal_instance.UNet.centroid", 127235, (384, 384, 1), None),
    ("minimal_instance.UNet.centered_instance", 150674, (96, 96, 1), None),
    ("minimal_robot.UNet.single_instance", 18250, (160, 280, 3), [(40, 70, 2)]),
    ("min_tracks_2node.UNet.bottomup_multiclass", 80683, (512, 512, 1), None),
]


@pytest.mark.parametrize("name,n_params,in_shape,out_shapes", CASES)
def test_fixture_structure(name, n_params, in_shape, out_shapes):
    cfg, w = load_npz_model(os.path.join(MODELS, name, "best_model.npz"))
    g = KerasGraph(cfg, w)
    assert g.n_params() == n_params
    rng = np.random.default_rng(0)
    h, w_, c = in_shape
    h, w_ = min(h, 64), min(w_, 64)  # fully convolutional: run small
    x = rng.random((1, h, w_, c), dtype=np.float32)
    outs = g(x)
    assert all(np.isfinite(o).all() for o in outs)
    if out_shapes is not None:
        for o, s in zip(outs, out_shapes):
            stride = in_shape[0] // s[0]
            assert o.shape == (1, h // stride, w_ // stride, s[2])


def test_same_padding_rules():
    assert _same_pads(384, 3, 1) == (1, 1)
    assert _same_pads(384, 7, 2) == (2, 3)  # asymmetric (hourglass stem), SURVEY §8a
    assert _same_pads(384, 2, 2) == (0, 0)
    assert _same_pads(5, 2, 2) == (0, 1)


def test_conv2d_transpose_same_k3s2():
    """out[j] = sum_{2i+k=j} x[i] w[k], cropped at the END to 2n."""
    cfg = {"config": {"layers": [
        {"class_name": "InputLayer", "name": "input", "config": {"batch_input_shape": [None, 1, 3, 1]}, "inbound_nodes": []},
        {"class_name": "Conv2DTranspose", "name": "t", "config": {"filters": 1, "kernel_size": [1, 3], "strides": [1, 2],
         "padding": "same", "activation": "linear", "use_bias": False}, "inbound_nodes": [[["input", 0, 0, {}]]]},
    ], "input_layers": [["input", 0, 0]], "output_layers": [["t", 0, 0]]}}
    w = {"t/kernel": np.array([1.0, 10.0, 100.0], np.float32).reshape(1, 3, 1, 1)}
    x = np.array([1.0, 2.0, 3.0], np.float32).reshape(1, 1, 3, 1)
    (y,) = KerasGraph(cfg, w)(x)
    # full: [1,10,100+2,20,200+3,30,300] -> first 6
    np.testing.assert_allclose(y.reshape(-1), [1, 10, 102, 20, 203, 30])


def test_preprocess():
    x = (np.arange(2 * 5 * 6 * 1) % 256).astype(np.uint8).reshape(2, 5, 6, 1)
    y = preprocess(x, input_scale=1.0, pad_stride=4)
    assert y.shape == (2, 8, 8, 1) and y.dtype == np.float32
    np.testing.assert_allclose(y[:, :5, :6], x.astype(np.float32) * np.float32(1 / 255))
    assert (y[:, 5:] == 0).all() and (y[:, :, 6:] == 0).all()
    y = preprocess(x, input_scale=0.5, pad_stride=1)
    assert y.shape == (2, 2, 3, 1)
