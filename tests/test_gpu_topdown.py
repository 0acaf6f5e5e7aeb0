"""Single-instance and top-down layers (BASELINE configs[0-2]) on the GPU against the NumPy
restatement in oracle/inference.py, on the reference's fixture models (reference: tests/nn/test_inference.py:214-379,
542-589 layer tests; :592-766 predictor tests)."""
import os

import numpy as np
import pytest
import torch
from numpy.testing import assert_allclose, assert_array_equal

from oracle import inference as oinf
from oracle import peak_finding as opf
from oracle.keras_graph import resize_image as oracle_resize

pytestmark = pytest.mark.gpu
MODELS = os.path.join(os.path.dirname(__file__), "golden", "models")


def _n(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_crop_and_resize_vs_oracle(dtype):
    from sleap_amd import ops

    rng = np.random.default_rng(0)
    imgs = (rng.random((3, 40, 52, 2)) * (255 if dtype == np.uint8 else 1)).astype(dtype)
    centres = np.concatenate([rng.uniform(-4, 56, (20, 1)), rng.uniform(-4, 44, (20, 1))], axis=1).astype(np.float32)
    centres[:4] = [[0, 0], [51, 39], [25.5, 20.25], [7, 3]]
    sinds = rng.integers(0, 3, 20).astype(np.int32)
    for crop in (8, 5):
        want = opf.crop_bboxes(imgs.astype(np.float32), opf.make_centered_bboxes(centres, crop, crop), sinds)
        if dtype == np.uint8:
            want = want.astype(np.uint8)
        got = _n(ops.crop_and_resize(torch.from_numpy(imgs).cuda(), torch.from_numpy(centres).cuda(),
                                     torch.from_numpy(sinds).cuda(), crop))
        if dtype == np.uint8:
            assert_array_equal(got, want)
        else:
            assert_allclose(got, want, atol=1e-6)


def test_resize_kernel_vs_oracle():
    from sleap_amd.nn.inference import _resize_image

    rng = np.random.default_rng(1)
    for shape, scale in (((2, 31, 45, 1), 0.5), ((1, 20, 28, 3), 0.75), ((1, 16, 16, 1), 2.0)):
        x = rng.random(shape).astype(np.float32)
        got = _n(_resize_image(torch.from_numpy(x).cuda(), scale))
        assert_allclose(got, oracle_resize(x, scale), atol=2e-6)


def test_fused_uint8_resize_is_bitwise_convert_then_resize():
    """sa_resize_bilinear_u8_f32 (ensure_float folded into the resize, one launch) == float32(x) * float32(1/255) followed by
    sa_resize_bilinear_f32, bit for bit; without the scale it is the plain uint8 resize."""
    from sleap_amd.nn.inference import _resize_image

    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.integers(0, 256, (3, 37, 52, 1), dtype=np.uint8)).cuda()
    for scale in (0.5, 0.75, 1.5):
        two_step = _resize_image(x.to(torch.float32) * np.float32(1.0 / 255.0), scale)
        assert torch.equal(_resize_image(x, scale, to_float=True), two_step)
        assert torch.equal(_resize_image(x, scale), _resize_image(x.to(torch.float32), scale).to(torch.uint8))


def test_single_instance_predictor_robot_fixture():
    """RGB model with input_scaling 0.5 (resize kernel, +0.5 un-scaling) -- BASELINE configs[0]/[1] layer path."""
    from sleap_amd.nn.inference import SingleInstancePredictor, load_model

    p = load_model(os.path.join(MODELS, "minimal_robot.UNet.single_instance"), batch_size=2)
    assert isinstance(p, SingleInstancePredictor) and not p.is_grayscale
    layer = p.inference_model.single_instance_layer
    assert layer.input_scale == 0.5 and layer.output_stride == 4 and layer.offsets_ind is None
    rng = np.random.default_rng(2)
    frames = rng.integers(0, 256, (3, 320, 560, 3), dtype=np.uint8)
    # choose a threshold that keeps some peaks and drops others on these out-of-distribution frames
    cms = _n(layer.keras_model.forward(layer.preprocess(frames))[layer.confmaps_ind])
    layer.peak_threshold = float(np.median(cms.max(axis=(1, 2))))
    outs = p.predict(frames, make_labels=False)
    got = np.concatenate([o["instance_peaks"] for o in outs])
    gotv = np.concatenate([o["instance_peak_vals"] for o in outs])
    want, wantv = oinf.single_instance_peaks(cms, None, layer.peak_threshold, "integral", 5, 4, 0.5)
    assert got.shape == (3, 1, 2, 2)
    assert np.isnan(want).any() and not np.isnan(want).all()
    assert_allclose(got, want, atol=1e-4, equal_nan=True)
    assert_array_equal(gotv, wantv)
    assert_array_equal(np.concatenate([o["frame_ind"] for o in outs]), [0, 1, 2])


@pytest.fixture(scope="module")
def topdown():
    from sleap_amd.nn.inference import load_model
    from sleap_amd.synth import render_frames

    p = load_model([os.path.join(MODELS, "minimal_instance.UNet.centroid"),
                    os.path.join(MODELS, "minimal_instance.UNet.centered_instance")], batch_size=2)
    frames = render_frames(3, 384, 384, n_animals=2, seed=21)[0]
    cc = p.inference_model.centroid_crop
    cms = _n(cc.keras_model.forward(cc.preprocess(frames))[cc.confmaps_ind])
    vals = np.sort(cms[opf.nms_mask(cms, -np.inf)])[::-1]
    cc.peak_threshold = float(vals[min(len(vals) - 1, 4 * cms.shape[0])])  # ~4 centroids per frame
    return p, frames


def test_topdown_wiring(topdown):
    from sleap_amd.nn.inference import TopDownPredictor

    p, _ = topdown
    assert isinstance(p, TopDownPredictor)
    cc, ip = p.inference_model.centroid_crop, p.inference_model.instance_peaks
    assert cc.crop_size == 96 and cc.output_stride == 4 and cc.input_scale == 1.0 and cc.precrop_resize == 1.0
    assert ip.output_stride == 2 and ip.resize_input_image is False and ip.input_scale == 1.0
    assert cc.offsets_ind is not None and ip.offsets_ind is not None  # both fixtures have learned-offset heads


@pytest.mark.parametrize("max_instances", [None, 2])
def test_topdown_layers_vs_oracle(topdown, max_instances):
    p, frames = topdown
    cc, ip = p.inference_model.centroid_crop, p.inference_model.instance_peaks
    cc.max_instances = max_instances
    try:
        out = cc(frames)
        outs = cc.keras_model.forward(cc.preprocess(frames))
        cms, offs = _n(outs[cc.confmaps_ind]), _n(outs[cc.offsets_ind])
        want = oinf.centroid_crop(frames, cms, offs, cc.peak_threshold, "integral", 5, 4, 1.0, 96, max_instances)
        assert len(want["centroids"]) >= 3
        if max_instances is not None:
            assert np.bincount(want["crop_sample_inds"], minlength=3).max() <= max_instances
        assert_array_equal(_n(out["crop_sample_inds"]), want["crop_sample_inds"])
        assert_allclose(_n(out["centroids"]), want["centroids"], atol=1e-4)
        assert_array_equal(_n(out["centroid_vals"]), want["centroid_vals"])
        assert_allclose(_n(out["crop_offsets"]), want["crop_offsets"], atol=1e-4)
        # crops: bilinear on uint8 with truncation; identical centroids -> identical crops except where the
        # centroid differs in the last float bit (then a value may move by one level)
        d = np.abs(_n(out["crops"]).astype(np.int32) - want["crops"].astype(np.int32))
        assert d.max() <= 1 and (d > 0).mean() < 1e-3
        # second stage on the device crops
        res = ip(out)
        couts = ip.keras_model.forward(ip.preprocess(out["crops"], resize_img=False))
        ccms, coffs = _n(couts[ip.confmaps_ind]), _n(couts[ip.offsets_ind])
        wp, wv = oinf.find_instance_peaks(ccms, coffs, _n(out["crop_offsets"]), ip.peak_threshold, "integral", 5, 2, 1.0)
        n_valid = _n(res["n_valid"])
        assert_array_equal(n_valid, np.bincount(want["crop_sample_inds"], minlength=3))
        k = 0
        for b in range(3):
            for i in range(n_valid[b]):
                assert_allclose(_n(res["instance_peaks"])[b, i], wp[k], atol=1e-4, equal_nan=True)
                assert_array_equal(_n(res["instance_peak_vals"])[b, i], wv[k])
                k += 1
            assert np.isnan(_n(res["instance_peaks"])[b, n_valid[b]:]).all()
    finally:
        cc.max_instances = None


def test_topdown_predict_api(topdown):
    p, frames = topdown
    outs = p.predict(frames, make_labels=False)
    assert len(outs) == 2
    for k in ("instance_peaks", "instance_peak_vals", "centroids", "centroid_vals", "n_valid", "frame_ind"):
        assert k in outs[0]
    assert outs[0]["instance_peaks"].shape[2:] == (2, 2)
    a = p.inference_model.predict(frames, numpy=True, batch_size=2)
    assert a["instance_peaks"].shape[0] == 3 and a["n_valid"].shape == (3,)


def test_topdown_predict_with_flow_tracker_uses_the_uploaded_frames(topdown):
    """The generic predict loop hands a flow tracker the batch it uploaded (no second host-to-device copy): the same tracks as
    tracking the finished predictions with the frames re-read from the array."""
    from sleap_amd.nn.tracking import Tracker, run_tracker

    p, frames = topdown
    frames = np.concatenate([frames, frames[::-1]])  # 6 frames, 3 batches of 2
    try:
        p.tracker = Tracker.make_tracker_by_name(tracker="flow", track_window=3)
        outs = p.predict(frames, make_labels=False)
    finally:
        p.tracker = None
    assert all("image_dev" not in ex and "image_ready" not in ex for ex in outs)
    plain = p.predict(frames, make_labels=False)
    again = run_tracker([dict(ex) for ex in plain], Tracker.make_tracker_by_name(tracker="flow", track_window=3), data=frames)
    for a, b in zip(outs, again):
        assert np.array_equal(a["track_inds"], b["track_inds"])
    assert max(int(ex["track_inds"].max()) for ex in outs) >= 0


@pytest.mark.parametrize("max_instances", [None, 2])
def test_topdown_device_path_equals_ragged_path_without_host_round_trips(topdown, max_instances):
    """`TopDownInferenceModel.call` runs on fixed crop slots per frame (sa_select_centroids / sa_finish_instance_peaks: the
    centroid model's counts never visit the host). Same instances, values and order as the reference-shaped ragged layers,
    NaN in the empty slots; more centroids than slots raise the overflow flag and the checked call grows the slots."""
    p, frames = topdown
    m = p.inference_model
    cc, ip = m.centroid_crop, m.instance_peaks
    cc.max_instances = max_instances
    old = cc.max_crops
    try:
        flat = cc(frames)
        want = ip(flat)
        n_valid = _n(want["n_valid"])
        assert n_valid.max() >= 3 or max_instances is not None
        cc.max_crops = 8
        got = m.call(torch.from_numpy(frames).cuda())
        assert got["instance_peaks"].shape[:2] == (3, 8 if max_instances is None else max_instances)
        assert_array_equal(_n(got["n_valid"]), n_valid)
        for b in range(3):
            k = n_valid[b]
            assert_allclose(_n(got["instance_peaks"])[b, :k], _n(want["instance_peaks"])[b, :k], atol=1e-4, equal_nan=True)
            assert_array_equal(_n(got["instance_peak_vals"])[b, :k], _n(want["instance_peak_vals"])[b, :k])
            assert_allclose(_n(got["centroids"])[b, :k], _n(want["centroids"])[b, :k], atol=1e-5)
            assert_array_equal(_n(got["centroid_vals"])[b, :k], _n(want["centroid_vals"])[b, :k])
            assert np.isnan(_n(got["instance_peaks"])[b, k:]).all() and np.isnan(_n(got["centroid_vals"])[b, k:]).all()
        assert int(_n(got["status"]).max()) == 0
        if max_instances is None:
            # too few slots: flagged, not silently truncated; the checked call doubles the slots and re-runs
            cc.max_crops = 1
            over = m.call(torch.from_numpy(frames).cuda())
            assert int(_n(over["status"]).max()) & 4 and int(_n(over["n_valid"]).max()) == 1
            res = m.outputs_to_numpy(over)
            assert cc.max_crops >= n_valid.max()
            assert_array_equal(res["n_valid"], n_valid)
            assert res["instance_peaks"].shape[1] == n_valid.max()  # unragged to the batch's bounding shape
            assert cc.max_crops == n_valid.max()  # and the slot count now follows what was seen
    finally:
        cc.max_instances, cc.max_crops = None, old


def test_select_centroids_top_k_order():
    """tf.math.top_k semantics on the device (inference.py:1884-1896): value descending, ties by lower index; un-scaling
    (p / input_scale) + 0.5 and precrop_resize as separate fp32 steps."""
    from sleap_amd import _lib, ops

    xy = torch.tensor([[[10., 20.], [30., 40.], [50., 60.], [70., 80.], [0., 0.]]], device="cuda")
    val = torch.tensor([[0.5, 0.9, 0.5, 0.7, 0.0]], device="cuda")
    cnt = torch.tensor([4], dtype=torch.int32, device="cuda")
    K = 3
    out = [torch.empty((1, K, 2), device="cuda"), torch.empty((1, K), device="cuda"), torch.empty((1, K, 2), device="cuda"),
           torch.empty((1, K, 2), device="cuda"), torch.empty((1,), dtype=torch.int32, device="cuda"),
           torch.zeros((1,), dtype=torch.int32, device="cuda")]
    _lib.check(_lib.lib().sa_select_centroids(ops._ptr(xy), ops._ptr(val), ops._ptr(cnt), 1, 5, K, 3, 0.5, 2.0, 8,
                                              *[ops._ptr(t) for t in out], ops._stream()))
    cent, cval, centre, off, nv, st = [_n(t) for t in out]
    assert nv[0] == 3 and st[0] == 0
    assert_array_equal(cval[0], np.array([0.9, 0.7, 0.5], np.float32))  # the first of the two 0.5s
    want = (np.array([[30., 40.], [70., 80.], [10., 20.]], np.float32) / np.float32(0.5) + np.float32(0.5)) * np.float32(2.0)
    assert_array_equal(cent[0], want)
    assert_array_equal(off[0], want - np.float32(4.0))
    assert_array_equal(centre[0], want)
