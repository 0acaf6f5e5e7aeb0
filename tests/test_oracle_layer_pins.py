"""oracle/keras_graph.py against HAND-DERIVED vectors for the decoder / normalisation layers that no reference-held golden
exercises offline (tests/layer_pin_vectors.py states the TensorFlow rules and writes every expected array out). The `-m gpu`
twin, tests/test_gpu_layer_pins.py, holds the device kernels to the same arrays."""
import numpy as np
import pytest

from oracle.keras_graph import KerasGraph

import layer_pin_vectors as V

F = np.float32


def _graph(layers, out):
    g = [{"class_name": "InputLayer", "name": "input", "config": {"batch_input_shape": [None, None, None, layers[0][3]]},
          "inbound_nodes": []}]
    prev = "input"
    for cn, name, cfg, _cin, *src in layers:
        cfg = dict(cfg, name=name)
        g.append({"class_name": cn, "name": name, "config": cfg, "inbound_nodes": [[[s, 0, 0, {}] for s in (src[0] if src else [prev])]]})
        prev = name
    return {"class_name": "Functional", "config": {"name": "m", "layers": g, "input_layers": [["input", 0, 0]],
                                                  "output_layers": [[out, 0, 0]]}}


def _convt(k, cin=1, cout=1):
    return ("Conv2DTranspose", "ct", {"filters": cout, "kernel_size": [k, k], "strides": [2, 2], "padding": "same",
                                      "activation": "linear", "use_bias": True, "dilation_rate": [1, 1]}, cin)


def _delta(n, at, value=1.0, c=1, ch=0):
    x = np.zeros((1, n, n, c), F)
    x[0, at[0], at[1], ch] = value
    return x


@pytest.mark.parametrize("at,want", [((1, 1), V.CONVT3_DELTA_11), ((0, 0), V.CONVT3_DELTA_00), ((2, 2), V.CONVT3_DELTA_22)])
def test_conv2d_transpose_k3_s2_same_is_cropped_at_the_end(at, want):
    g = KerasGraph(_graph([_convt(3)], "ct"), {"ct/kernel": V.W3[:, :, None, None], "ct/bias": np.zeros(1, F)})
    np.testing.assert_array_equal(g(_delta(3, at))[0][0, :, :, 0], want)
    np.testing.assert_array_equal(V.stamp(3, 2, 3, 0, V.W3, at), want)  # the placement helper states the same rule


def test_conv2d_transpose_k3_overlapping_stamps_and_bias():
    g = KerasGraph(_graph([_convt(3)], "ct"), {"ct/kernel": V.W3[:, :, None, None], "ct/bias": np.array([0.5], F)})
    x = _delta(3, (0, 0)) + _delta(3, (0, 1), 10.0)
    np.testing.assert_array_equal(g(x)[0][0, :, :, 0], V.CONVT3_TWO + F(0.5))


@pytest.mark.parametrize("at,want", [((0, 0), V.CONVT4_DELTA_00), ((1, 1), V.CONVT4_DELTA_11), ((2, 2), V.CONVT4_DELTA_22)])
def test_conv2d_transpose_k4_s2_same_crops_one_on_each_side(at, want):
    g = KerasGraph(_graph([_convt(4)], "ct"), {"ct/kernel": V.W4[:, :, None, None], "ct/bias": np.zeros(1, F)})
    np.testing.assert_array_equal(g(_delta(3, at))[0][0, :, :, 0], want)
    np.testing.assert_array_equal(V.stamp(3, 2, 4, 1, V.W4, at), want)


def test_conv2d_transpose_kernel_layout_is_kh_kw_cout_cin():
    k = V.W3[:, :, None, None] * V.CONVT_MIX[None, None, :, :]  # [kh, kw, co, ci] = M[co][ci] * W3
    g = KerasGraph(_graph([_convt(3, 2, 2)], "ct"), {"ct/kernel": k, "ct/bias": np.zeros(2, F)})
    y = g(_delta(3, (1, 1), V.CONVT_IN_VALUES[0], 2, 0) + _delta(3, (1, 1), V.CONVT_IN_VALUES[1], 2, 1))[0][0]
    for co, s in enumerate(V.CONVT3_2CH_SCALE):
        np.testing.assert_array_equal(y[:, :, co], F(s) * V.CONVT3_DELTA_11)


def test_conv2d_k7_s2_same_pads_two_before_three_after():
    conv = ("Conv2D", "c", {"filters": 1, "kernel_size": [7, 7], "strides": [2, 2], "padding": "same", "activation": "linear",
                            "use_bias": True, "dilation_rate": [1, 1]}, 1)
    g = KerasGraph(_graph([conv], "c"), {"c/kernel": V.W7[:, :, None, None], "c/bias": np.zeros(1, F)})
    np.testing.assert_array_equal(g(_delta(8, (3, 4)))[0][0, :, :, 0], V.CONV7S2_DELTA_34)


def test_maxpool_same_on_an_odd_size_ignores_the_padding():
    pool = ("MaxPooling2D", "p", {"pool_size": [2, 2], "strides": [2, 2], "padding": "same"}, 1)
    g = KerasGraph(_graph([pool], "p"), {})
    np.testing.assert_array_equal(g(V.POOL_IN[None, :, :, None])[0][0, :, :, 0], V.POOL_OUT)
    np.testing.assert_array_equal(g(-V.POOL_IN[None, :, :, None] - 1)[0][0, :, :, 0],
                                  -np.array([[0, 2, 4], [10, 12, 14], [20, 22, 24]], F) - 1)  # (zero padding would win here)


@pytest.mark.parametrize("x,want,mode", [(V.UP_IN, V.UP_BILINEAR, "bilinear"), (V.UP_IN3, V.UP_BILINEAR3, "bilinear"),
                                         (V.UP_IN, V.UP_NEAREST, "nearest")])
def test_upsampling2d_half_pixel_centres(x, want, mode):
    up = ("UpSampling2D", "u", {"size": [2, 2], "interpolation": mode}, 1)
    g = KerasGraph(_graph([up], "u"), {})
    np.testing.assert_array_equal(g(x[None, :, :, None])[0][0, :, :, 0], want)


def _bn_weights(name="bn"):
    return {f"{name}/gamma": V.BN_GAMMA, f"{name}/beta": V.BN_BETA, f"{name}/moving_mean": V.BN_MEAN,
            f"{name}/moving_variance": V.BN_VAR}


def test_batch_normalization_inference_formula():
    bn = ("BatchNormalization", "bn", {"axis": [3], "epsilon": V.BN_EPS}, 3)
    g = KerasGraph(_graph([bn], "bn"), _bn_weights())
    np.testing.assert_allclose(g(V.BN_X[None, None, None, :])[0][0, 0, 0], V.BN_Y, rtol=0, atol=2e-6)


def test_hourglass_conv_is_relu_then_batchnorm():
    conv = ("Conv2D", "c", {"filters": 3, "kernel_size": [1, 1], "strides": [1, 1], "padding": "same", "activation": "relu",
                            "use_bias": True, "dilation_rate": [1, 1]}, 3)
    bn = ("BatchNormalization", "bn", {"axis": [3], "epsilon": V.BN_EPS}, 3)
    w = dict(_bn_weights(), **{"c/kernel": np.eye(3, dtype=F)[None, None], "c/bias": np.zeros(3, F)})
    g = KerasGraph(_graph([conv, bn], "bn"), w)
    np.testing.assert_allclose(g(V.BN_AFTER_RELU_X[None, None, None, :])[0][0, 0, 0], V.BN_AFTER_RELU_Y, rtol=0, atol=2e-6)


def test_concatenate_puts_the_skip_first_and_add_adds():
    a = ("Conv2D", "a", {"filters": 1, "kernel_size": [1, 1], "strides": [1, 1], "padding": "same", "activation": "linear",
                         "use_bias": True, "dilation_rate": [1, 1]}, 1, ["input"])
    b = ("Conv2D", "b", dict(a[2]), 1, ["input"])
    cat = ("Concatenate", "cat", {"axis": -1}, 2, ["a", "b"])  # [skip, x]
    add = ("Add", "add", {}, 1, ["a", "b"])
    w = {"a/kernel": np.full((1, 1, 1, 1), 2, F), "a/bias": np.zeros(1, F), "b/kernel": np.full((1, 1, 1, 1), 3, F),
         "b/bias": np.zeros(1, F)}
    x = np.ones((1, 1, 1, 1), F)
    assert KerasGraph(_graph([a, b, cat], "cat"), w)(x)[0][0, 0, 0].tolist() == [2.0, 3.0]
    assert KerasGraph(_graph([a, b, add], "add"), w)(x)[0][0, 0, 0].tolist() == [5.0]
