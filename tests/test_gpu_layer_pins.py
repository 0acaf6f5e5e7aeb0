"""The device kernels against the HAND-DERIVED layer vectors of tests/layer_pin_vectors.py (the CPU twin,
tests/test_oracle_layer_pins.py, holds the oracle to the same arrays): Conv2DTranspose k3 / k4 s2 'same' (crop side, kernel
layout), bilinear / nearest UpSampling2D (half-pixel centres, clamped borders), Conv2D k7 s2 'same' (2 / 3 padding),
ReLU-then-BatchNormalization, Concatenate([skip, x]) order. Each case is a tiny Keras-style graph compiled by the engine:

    input (float32 or uint8) -> 3x3 'copy' conv (centre tap) -> LAYER UNDER TEST -> 1x1 linear head (identity) -> float32

All values are small integers or dyadic fractions, exact in fp16 and bf16 storage, so the comparison is (near) equality."""
import numpy as np
import pytest
import torch

import layer_pin_vectors as V

pytestmark = pytest.mark.gpu
F = np.float32


def _layer(cn, name, cfg, src):
    return {"class_name": cn, "name": name, "config": dict(cfg, name=name), "inbound_nodes": [[[s, 0, 0, {}] for s in src]]}


def _conv(name, src, filters, k=3, stride=1, act="linear"):
    return _layer("Conv2D", name, {"filters": filters, "kernel_size": [k, k], "strides": [stride, stride], "padding": "same",
                                   "activation": act, "use_bias": True, "dilation_rate": [1, 1]}, [src])


def _model(layers, out, cin=1):
    inp = {"class_name": "InputLayer", "name": "input", "config": {"batch_input_shape": [None, None, None, cin], "name": "input"},
           "inbound_nodes": []}
    return {"class_name": "Functional", "config": {"name": "m", "layers": [inp] + layers, "input_layers": [["input", 0, 0]],
                                                  "output_layers": [[o, 0, 0] for o in out]}}


def _copy_kernel(gains):
    """3x3 kernel (kh, kw, 1, C) whose centre tap multiplies the image by gains[c]."""
    k = np.zeros((3, 3, 1, len(gains)), F)
    k[1, 1, 0, :] = gains
    return k


def _eye_head(c):
    return np.eye(c, dtype=F)[None, None]


def _run(mc, w, x, dtype):
    from sleap_amd.nn.engine import DeviceNetwork

    net = DeviceNetwork(mc, w, dtype=dtype)
    outs = net.forward(torch.from_numpy(np.ascontiguousarray(x)).cuda())
    return [o.cpu().numpy() for o in outs]


from parity_helpers import STORAGE_DTYPES as DTYPES  # noqa: E402  (fp16; + bf16 under SLEAP_AMD_TEST_BF16=1)


def _convt_graph(k, cin, cout):
    ct = _layer("Conv2DTranspose", "ct", {"filters": cout, "kernel_size": [k, k], "strides": [2, 2], "padding": "same",
                                          "activation": "linear", "use_bias": True, "dilation_rate": [1, 1]}, ["copy"])
    return _model([_conv("copy", "input", cin), ct, _conv("SingleInstanceConfmapsHead", "ct", cout, k=1)],
                  ["SingleInstanceConfmapsHead"])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("k,wk,pad", [(3, V.W3, 0), (4, V.W4, 1)])
def test_conv2d_transpose_s2_same_crop_side_on_device(k, wk, pad, dtype):
    """k3: the full transposed conv cropped at the END; k4: one row / column cropped on each side. 8 x 8 frames with single
    pixels at the corners and inside, each compared with the written-out stamps (literal 3 x 3 cases) and the placement rule."""
    mc = _convt_graph(k, 1, 1)
    w = {"copy/kernel": _copy_kernel([1.0]), "copy/bias": np.zeros(1, F), "ct/kernel": wk[:, :, None, None], "ct/bias": np.zeros(1, F),
         "SingleInstanceConfmapsHead/kernel": _eye_head(1), "SingleInstanceConfmapsHead/bias": np.zeros(1, F)}
    lit = {3: {(1, 1): V.CONVT3_DELTA_11, (0, 0): V.CONVT3_DELTA_00, (2, 2): V.CONVT3_DELTA_22},
           4: {(1, 1): V.CONVT4_DELTA_11, (0, 0): V.CONVT4_DELTA_00, (2, 2): V.CONVT4_DELTA_22}}[k]
    for at, want in lit.items():  # the placement helper reproduces the literal arrays (also asserted on the CPU)
        np.testing.assert_array_equal(V.stamp(3, 2, k, pad, wk, at), want)
    n = 8
    ats = [(0, 0), (3, 4), (7, 7), (0, 7), (7, 0)]
    x = np.zeros((len(ats), n, n, 1), F)
    for b, at in enumerate(ats):
        x[b, at[0], at[1], 0] = 1.0
    got = _run(mc, w, x, dtype)[0]
    assert got.shape == (len(ats), 2 * n, 2 * n, 1)
    for b, at in enumerate(ats):
        np.testing.assert_array_equal(got[b, :, :, 0], V.stamp(n, 2, k, pad, wk, at), err_msg=f"pixel at {at}")
    # overlapping stamps + bias (literal case embedded in the top-left corner)
    w2 = dict(w, **{"ct/bias": np.array([0.5], F)})
    second = 10.0 if k == 3 else 2.0  # (every sum must stay exact in bf16's 8 significant bits: k4 reaches 16 * 2 + 13 + 0.5)
    x = np.zeros((1, n, n, 1), F)
    x[0, 0, 0, 0], x[0, 0, 1, 0] = 1.0, second
    got = _run(mc, w2, x, dtype)[0][0, :, :, 0]
    want = V.stamp(n, 2, k, pad, wk, (0, 0)) + V.stamp(n, 2, k, pad, wk, (0, 1), second) + F(0.5)
    np.testing.assert_array_equal(got, want)
    if k == 3:
        np.testing.assert_array_equal(got[:6, :6], (V.CONVT3_TWO + F(0.5)))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("k", [3, 4])
def test_conv2d_transpose_kernel_layout_on_device(k, dtype):
    wk, pad = (V.W3, 0) if k == 3 else (V.W4, 1)
    mc = _convt_graph(k, 2, 2)
    w = {"copy/kernel": _copy_kernel(V.CONVT_IN_VALUES), "copy/bias": np.zeros(2, F),
         "ct/kernel": wk[:, :, None, None] * V.CONVT_MIX[None, None, :, :], "ct/bias": np.zeros(2, F),
         "SingleInstanceConfmapsHead/kernel": _eye_head(2), "SingleInstanceConfmapsHead/bias": np.zeros(2, F)}
    n = 8
    x = np.zeros((1, n, n, 1), F)
    x[0, 1, 1, 0] = 1.0
    got = _run(mc, w, x, dtype)[0][0]
    for co, s in enumerate(V.CONVT3_2CH_SCALE):
        np.testing.assert_array_equal(got[:, :, co], F(s) * V.stamp(n, 2, k, pad, wk, (1, 1)))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mode", ["bilinear", "nearest"])
def test_upsampling2d_on_device(mode, dtype):
    """UpSampling2D between two convs (the UNet decoder's position) on the ramp x[r][c] = 8 r + 4 c, whose top-left 2 x 2 block
    is the literal case [[0, 4], [8, 12]]: bilinear interpolation is exact on a ramp, so the half-pixel rule gives
    out[i][j] = 8 src(i) + 4 src(j) with src(j) = clip((j + 0.5) / 2 - 0.5, 0, n - 1) -- clamped at BOTH ends."""
    up = _layer("UpSampling2D", "up", {"size": [2, 2], "interpolation": mode}, ["copy"])
    mc = _model([_conv("copy", "input", 1), up, _conv("after", "up", 1), _conv("SingleInstanceConfmapsHead", "after", 1, k=1)],
                ["SingleInstanceConfmapsHead"])
    w = {"copy/kernel": _copy_kernel([1.0]), "copy/bias": np.zeros(1, F), "after/kernel": _copy_kernel([1.0]), "after/bias": np.zeros(1, F),
         "SingleInstanceConfmapsHead/kernel": _eye_head(1), "SingleInstanceConfmapsHead/bias": np.zeros(1, F)}
    n = 8
    x = np.zeros((1, n, n, 1), F)
    x[0, :, :, 0] = 4.0 * np.arange(n)[None, :] + 8.0 * np.arange(n)[:, None]  # x[r][c] = 8 r + 4 c: the 2x2 block is V.UP_IN
    got = _run(mc, w, x, dtype)[0][0, :, :, 0]
    np.testing.assert_array_equal(x[0, :2, :2, 0], V.UP_IN)
    if mode == "nearest":
        np.testing.assert_array_equal(got[:4, :4], V.UP_NEAREST)
        np.testing.assert_array_equal(got, np.repeat(np.repeat(x[0, :, :, 0], 2, 0), 2, 1))
        return
    # on a linear ramp the half-pixel rule gives out[j] = ramp((j + 0.5) / 2 - 0.5), clamped at both ends
    np.testing.assert_array_equal(got[:3, :3], V.UP_BILINEAR[:3, :3])  # (row / column 3 of the literal is the 2x2 case's clamp)
    src = np.clip((np.arange(2 * n) + 0.5) / 2 - 0.5, 0, n - 1).astype(F)
    np.testing.assert_array_equal(got, 8.0 * src[:, None] + 4.0 * src[None, :])
    np.testing.assert_array_equal(got[0, :6], np.array([0, 1, 3, 5, 7, 9], F))  # a, .75a+.25b, .25a+.75b, ... (UP_BILINEAR3's pattern)
    # the clamped END: source columns [.., 20, 24, 28] -> [.., 23, 25, 27, 28]
    np.testing.assert_array_equal(got[0, -4:], np.array([23, 25, 27, 28], F))


@pytest.mark.parametrize("dtype", DTYPES)
def test_relu_then_batchnorm_on_device(dtype):
    """hourglass conv(): Conv2D(activation=relu) THEN BatchNormalization -- the conv epilogue's post-affine."""
    bn = _layer("BatchNormalization", "bn", {"axis": [3], "epsilon": V.BN_EPS, "center": True, "scale": True}, ["c"])
    mc = _model([_conv("copy", "input", 3), _conv("c", "copy", 3, act="relu"), bn, _conv("SingleInstanceConfmapsHead", "bn", 3, k=1)],
                ["SingleInstanceConfmapsHead"])
    kc = np.zeros((3, 3, 3, 3), F)
    kc[1, 1] = np.eye(3, dtype=F)
    w = {"copy/kernel": _copy_kernel(V.BN_AFTER_RELU_X), "copy/bias": np.zeros(3, F), "c/kernel": kc, "c/bias": np.zeros(3, F),
         "bn/gamma": V.BN_GAMMA, "bn/beta": V.BN_BETA, "bn/moving_mean": V.BN_MEAN, "bn/moving_variance": V.BN_VAR,
         "SingleInstanceConfmapsHead/kernel": _eye_head(3), "SingleInstanceConfmapsHead/bias": np.zeros(3, F)}
    got = _run(mc, w, np.ones((1, 8, 8, 1), F), dtype)[0][0]
    np.testing.assert_allclose(got, np.broadcast_to(V.BN_AFTER_RELU_Y, got.shape), rtol=0, atol=2e-3)
    # and plain BN of a positive input (no relu effect): BN_X -> BN_Y
    w["copy/kernel"] = _copy_kernel(V.BN_X)
    got = _run(mc, w, np.ones((1, 8, 8, 1), F), dtype)[0][0]
    np.testing.assert_allclose(got, np.broadcast_to(V.BN_Y, got.shape), rtol=0, atol=4e-3 if dtype == "fp16" else 4e-2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("u8", [False, True])
def test_conv2d_k7_s2_same_padding_on_device(u8, dtype):
    """The k7 s2 'same' stem conv on the image (hourglass.py:75-85; UNet stem blocks): 2 rows / columns of padding before,
    3 after. A single pixel at (3, 4) of an 8 x 8 corner (frame 16 x 16), literal expected block."""
    bn = _layer("BatchNormalization", "bn", {"axis": [3], "epsilon": 1e-3, "center": True, "scale": True}, ["stem"])
    mc = _model([_conv("stem", "input", 1, k=7, stride=2, act="relu"), bn, _conv("SingleInstanceConfmapsHead", "bn", 1, k=1)],
                ["SingleInstanceConfmapsHead"])
    w = {"stem/kernel": V.W7[:, :, None, None], "stem/bias": np.zeros(1, F), "bn/gamma": np.ones(1, F), "bn/beta": np.zeros(1, F),
         "bn/moving_mean": np.zeros(1, F), "bn/moving_variance": np.ones(1, F) - F(1e-3),
         "SingleInstanceConfmapsHead/kernel": _eye_head(1), "SingleInstanceConfmapsHead/bias": np.zeros(1, F)}
    n = 16
    if u8:
        x = np.zeros((1, n, n, 1), np.uint8)
        x[0, 3, 4, 0] = 255  # ensure_float: 255 * (1 / 255) = 1
    else:
        x = np.zeros((1, n, n, 1), F)
        x[0, 3, 4, 0] = 1.0
    got = _run(mc, w, x, dtype)[0][0, :, :, 0]
    assert got.shape == (8, 8)
    np.testing.assert_allclose(got[:4, :4], V.CONV7S2_DELTA_34, rtol=2e-3, atol=1e-3)
    assert np.abs(got[4:, :]).max() <= 1e-3 and np.abs(got[:, 4:]).max() <= 1e-3


@pytest.mark.parametrize("dtype", DTYPES)
def test_concatenate_order_skip_first_on_device(dtype):
    """Concatenate([skip, upsampled]) as the two-source K loop of the consuming conv: channel 0 of the concatenation is the
    SKIP tensor's channel 0, channel 16 the upsampled tensor's channel 0 (encoder_decoder.py:360-362)."""
    pool = _layer("MaxPooling2D", "pool", {"pool_size": [2, 2], "strides": [2, 2], "padding": "same"}, ["a_relu"])
    up = _layer("UpSampling2D", "up", {"size": [2, 2], "interpolation": "bilinear"}, ["b_relu"])
    cat = _layer("Concatenate", "cat", {"axis": -1}, ["a_relu", "up"])
    mc = _model([_conv("a", "input", 16), _layer("Activation", "a_relu", {"activation": "relu"}, ["a"]), pool,
                 _conv("b", "pool", 16), _layer("Activation", "b_relu", {"activation": "relu"}, ["b"]), up, cat,
                 _conv("c", "cat", 16), _layer("Activation", "c_relu", {"activation": "relu"}, ["c"]),
                 _conv("SingleInstanceConfmapsHead", "c_relu", 2, k=1)], ["SingleInstanceConfmapsHead"])
    ka = np.zeros((3, 3, 1, 16), F)
    ka[1, 1, 0, 0] = 1.0  # skip channel 0 = image
    kb = np.zeros((3, 3, 16, 16), F)
    kb[1, 1, 0, 0] = 3.0  # low-resolution channel 0 = 3 x pooled image
    kc = np.zeros((3, 3, 32, 16), F)
    kc[1, 1, 0, 0] = 1.0   # out 0 <- concat channel 0  (the skip)
    kc[1, 1, 16, 1] = 1.0  # out 1 <- concat channel 16 (the upsampled tensor)
    kh = np.zeros((1, 1, 16, 2), F)
    kh[0, 0, 0, 0] = kh[0, 0, 1, 1] = 1.0
    w = {"a/kernel": ka, "a/bias": np.zeros(16, F), "b/kernel": kb, "b/bias": np.zeros(16, F), "c/kernel": kc, "c/bias": np.zeros(16, F),
         "SingleInstanceConfmapsHead/kernel": kh, "SingleInstanceConfmapsHead/bias": np.zeros(2, F)}
    got = _run(mc, w, np.ones((1, 16, 16, 1), F), dtype)[0][0]
    np.testing.assert_array_equal(got[:, :, 0], np.ones((16, 16), F))
    np.testing.assert_array_equal(got[:, :, 1], np.full((16, 16), 3.0, F))
