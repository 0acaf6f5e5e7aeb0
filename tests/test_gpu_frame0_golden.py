"""The reference's `test_bottomup_predictor` (tests/nn/test_inference.py:769-786) in its own words, on the device path: load the
trained bottom-up model folder, predict frame 0 of centered_pair_low_quality.mp4 READ FROM THE MP4 (sleap_amd.io.video.MediaVideo,
the package's own key-frame decoder), and compare with (a) the predictions TensorFlow wrote for that frame
(labels_pr.val.slp -- the fp32 oracle reproduces them to 2e-5 px, tests/test_frame0_golden.py) and (b) the user labels, at the
reference's own 1.75 px. Storage types: fp16 (the default) within 0.05 px of TensorFlow, bf16 within 0.5 px."""
import os

import numpy as np
import pytest

from test_frame0_golden import FROZEN, MODEL, MP4, golden_predictions, match_instances

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
