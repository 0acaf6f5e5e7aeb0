"""The reference's `test_bottomup_predictor` (tests/nn/test_inference.py:769-786) in its own words, on the device path: load the
trained bottom-up model folder, predict frame 0 of centered_pair_low_quality.mp4 READ FROM THE MP4 (sleap_amd.io.video.MediaVideo,
the package's own key-frame decoder), and compare with (a) the predictions TensorFlow wrote for that frame
(labels_pr.val.slp -- the fp32 oracle reproduces them to 2e-5 px, tests/test_frame0_golden.py) and (b) the user labels, at the
reference's own 1.75 px. Storage types: fp16 (the default) within 0.05 px of TensorFlow, bf16 within 0.5 px."""
import os

import numpy as np
import pytest

from test_frame0_golden import FROZEN, MODEL, MP4, _gt_crops, golden_predictions, gt_centroids, match_instances

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,tol_px,tol_score", [("fp16", 0.05, 5e-3), ("bf16", 0.5, 5e-2)])
def test_bottomup_predictor_on_the_real_frame_matches_tensorflow(dtype, tol_px, tol_score):
    from sleap_amd.io.video import Video, VideoReader
    from sleap_amd.nn.inference import BottomUpPredictor, load_model

    predictor = load_model(MODEL, batch_size=4, progress_reporting="none", dtype=dtype)
    assert isinstance(predictor, BottomUpPredictor) and predictor.is_grayscale is True
    video = Video.from_filename(MP4)
    outs = predictor.predict(VideoReader(video, example_indices=[0]), make_labels=False)
    assert len(outs) == 1 and outs[0]["frame_ind"].tolist() == [0]
    n = int(outs[0]["n_valid"][0])
    assert n == 2  # len(labels_pr[0].instances) == 2
    pts = outs[0]["instance_peaks"][0, :n]
    want_pts, want_sc, want_inst = golden_predictions()
    order = match_instances(pts, want_pts)
    assert sorted(order) == [0, 1]
    d = np.linalg.norm(pts[order] - want_pts, axis=-1)
    dsc = np.abs(outs[0]["instance_peak_vals"][0, :n][order] - want_sc).max()
    dinst = np.abs(outs[0]["instance_scores"][0, :n][order] - want_inst).max()
    print(f"{dtype} device path vs labels_pr.val.slp (TensorFlow): max point distance {d.max():.5f} px, point score delta {dsc:.5f}, "
          f"instance score delta {dinst:.5f}")
    assert d.max() <= tol_px and dsc <= tol_score and dinst <= tol_score
    gt = np.load(FROZEN)["gt_points"]
    np.testing.assert_allclose(pts[match_instances(pts, gt)], gt, atol=1.75)  # the reference's assertion against the user labels
    # labels through the default path (make_labels=True), as the reference test reads them
    labels = predictor.predict(VideoReader(video, example_indices=[0]))
    assert len(labels) == 1 and len(labels[0].instances) == 2


def test_top_down_models_on_the_real_frame_match_tensorflow():
    """The top-down pair of the same fixture set against ITS TensorFlow-produced files (tests/test_frame0_golden.py explains what
    they hold): the centroid model's confidences at the two animals, and the centered-instance model's peaks on crops around the
    ground-truth centroids (sa_crop_and_resize at fractional centres, uint8 crops) -- then the pair end to end through
    `load_model([centroid, centered_instance])`, which must land within the reference's 1.75 px of the user labels."""
    import torch

    from sleap_amd import ops
    from sleap_amd.nn.inference import TopDownPredictor, load_model

    z = np.load(FROZEN)
    models = os.path.dirname(MODEL)
    p = load_model([os.path.join(models, "minimal_instance.UNet.centroid"), os.path.join(models, "minimal_instance.UNet.centered_instance")],
                   batch_size=1, progress_reporting="none")
    assert isinstance(p, TopDownPredictor)
    frame = torch.from_numpy(z["gray"][None, :, :, None]).cuda()
    cen = gt_centroids(z["gt_points"])
    # (a) the centroid model: confidences
    crop_layer = p.inference_model.centroid_crop
    o = crop_layer.call(frame)
    cpts, cvals = o["centroids"].cpu().numpy().reshape(-1, 2), o["centroid_vals"].cpu().numpy().reshape(-1)
    assert len(cpts) == 2
    order = [int(np.argmin(np.linalg.norm(cpts - c, axis=-1))) for c in cen]
    _, _, want_conf = golden_predictions("centroid")
    print(f"device centroid confidences {cvals[order]} vs TensorFlow's {want_conf}")
    assert sorted(order) == [0, 1] and np.abs(cvals[order] - want_conf).max() <= 1e-2  # (8e-3 of it is the file's other decode)
    # (b) the centered-instance model on ground-truth crops
    peaks_layer = p.inference_model.instance_peaks
    crops = ops.crop_and_resize(frame, torch.from_numpy(cen).cuda(), torch.zeros(2, dtype=torch.int32, device="cuda"), 96)
    _, want_crops, crop_offsets = _gt_crops(z["gray"], z["gt_points"])
    np.testing.assert_array_equal(crops.cpu().numpy(), want_crops)
    out = peaks_layer.call({"crops": crops, "crop_offsets": torch.from_numpy(crop_offsets).cuda()})
    pts = out["instance_peaks"].cpu().numpy().reshape(-1, 2, 2)
    vals = out["instance_peak_vals"].cpu().numpy().reshape(-1, 2)
    want_pts, want_sc, _ = golden_predictions("centered_instance")
    order = match_instances(pts, want_pts)
    d = np.linalg.norm(pts[order] - want_pts, axis=-1)
    print(f"device centered-instance peaks vs TensorFlow: max distance {d.max():.5f} px, score delta {np.abs(vals[order] - want_sc).max():.5f}")
    assert sorted(order) == [0, 1] and d.max() <= 0.05 and np.abs(vals[order] - want_sc).max() <= 5e-3
    # (c) the pair end to end (predicted centroids -> crops -> peaks), the reference's test_topdown_predictor assertions
    outs = p.predict(z["gray"][None, :, :, None], make_labels=False)
    n = int(outs[0]["n_valid"][0])
    assert n == 2
    got = outs[0]["instance_peaks"][0, :n]
    np.testing.assert_allclose(got[match_instances(got, z["gt_points"])], z["gt_points"], atol=1.75)


def test_bottomup_predictor_on_inter_coded_frames_of_the_mp4_matches_the_oracle():
    """Frames 1-4 of the same file are B and P pictures (display order: samples 3, 2, 4, 1): read through `MediaVideo` (the
    package's P / B decoder, tests/test_h264_inter.py), predicted by the device path, compared with the fp32 oracle on the same
    decoded frames -- the whole chain mp4 -> frames -> network -> peaks -> grouping from the reference's default input format."""
    from sleap_amd.io.video import Video, VideoReader
    from sleap_amd.nn.inference import load_model
    from test_frame0_golden import _oracle

    predictor = load_model(MODEL, batch_size=4, progress_reporting="none")
    video = Video.from_filename(MP4)
    idx = [0, 1, 2, 3, 4]
    outs = predictor.predict(VideoReader(video, example_indices=idx), make_labels=False)
    got = {int(f): (o["instance_peaks"][k, :int(o["n_valid"][k])], o["instance_scores"][k, :int(o["n_valid"][k])])
           for o in outs for k, f in enumerate(o["frame_ind"].tolist())}
    assert sorted(got) == idx
    worst = 0.0
    for f in idx:
        gray = video.get_frame(f)[..., 0]
        want_pts, _, want_inst = _oracle(gray)
        pts, inst = got[f]
        assert len(pts) == len(want_pts) == 2, (f, len(pts), len(want_pts))  # two flies in every frame
        order = match_instances(pts, want_pts)
        assert sorted(order) == [0, 1]
        d = np.linalg.norm(pts[order] - want_pts, axis=-1)
        worst = max(worst, float(np.nanmax(d)))
        assert np.nanmax(d) <= 0.1 and np.abs(inst[order] - want_inst).max() <= 2e-2, (f, d, inst, want_inst)
    print(f"device vs fp32 oracle on display frames 0-4 of the mp4 (I, B, B, B, P pictures): max point distance {worst:.4f} px")
