"""The reference's tracking integration tests (tests/nn/test_tracking_integration.py) on its own prediction file
`centered_pair_predictions.slp` (tests/golden/slp/centered_pair_predictions.arrays.npz, tools/make_golden_tracks.py): the number
of tracks the re-tracked file must hold. Host code only (csrc/tracker.hip is host C++, nn/kalman.py Python)."""
import os

import numpy as np
import pytest

from sleap_amd.nn.tracking import Tracker, run_tracker

GOLD = os.path.join(os.path.dirname(__file__), "golden", "slp", "centered_pair_predictions.arrays.npz")
KEYS = ("instance_peaks", "instance_peak_vals", "instance_scores", "n_valid", "frame_ind")


def _n_tracks(**kw):
    """`sleap-track --tracking.* ... centered_pair_predictions.slp` -> len(labels.tracks): run_tracker over the file's frames
    (inference.py:5712-5733), distinct tracks on the instances that come out"""
    d = np.load(GOLD)
    tracker = Tracker.make_tracker_by_name(**kw)
    out = run_tracker([{k: d[k].copy() for k in KEYS}], tracker, img_hw=(384, 384))[0]
    assert out["track_inds"].shape == d["instance_scores"].shape
    tr = out["track_inds"]
    return len(np.unique(tr[tr >= 0])), tracker


def test_simple_tracker():
    """test_tracking_integration.py:199-209: `--tracking.tracker simple` -> 27 tracks"""
    n, tracker = _n_tracks(tracker="simple")
    assert n == 27 and len(tracker.spawned_tracks) == 27


def test_simplemax_tracker():
    """:212-223: `--tracking.tracker simplemaxtracks --tracking.max_tracking 1 --tracking.max_tracks 2` -> 2 tracks"""
    n, _ = _n_tracks(tracker="simplemaxtracks", max_tracking=True, max_tracks=2)
    assert n == 2


@pytest.mark.parametrize("tracker_name", ["simple", "simplemaxtracks"])
@pytest.mark.parametrize("similarity", ["instance", "object_keypoint", "centroid", "iou"])
@pytest.mark.parametrize("match", ["hungarian", "greedy"])
def test_kalman_tracker(tracker_name, similarity, match):
    """:27-196: the Kalman tracker over a simple init tracker keeps the pair of animals on two tracks for the whole file --
    with target_instance_count alone, with max_tracks as well, with the IoU pre-cull, with pre_cull_to_target."""
    base = dict(tracker=tracker_name, similarity=similarity, match=match, track_window=5, kf_init_frame_count=10,
                kf_node_indices=[0, 1], target_instance_count=2)
    assert _n_tracks(**base)[0] == 2
    mt = dict(base, max_tracking=True, max_tracks=2)
    assert _n_tracks(**mt)[0] == 2
    if match == "greedy":  # (the remaining variants of the reference's test once per similarity)
        assert _n_tracks(**dict(mt, pre_cull_iou_threshold=0.8))[0] == 2
        assert _n_tracks(**dict(mt, pre_cull_to_target=True))[0] == 2
        assert _n_tracks(**dict(mt, post_connect_single_breaks=False))[0] == 2


@pytest.mark.parametrize("tracker_name", ["flow", "flowmaxtracks"])
def test_kalman_tracker_refuses_flow_and_normalized_instance(tracker_name):
    kw = dict(tracker=tracker_name, max_tracking=True, max_tracks=2, track_window=5, kf_init_frame_count=10, kf_node_indices=[0, 1])
    with pytest.raises(ValueError, match="Kalman filter requires simple tracker for initial tracking."):
        Tracker.make_tracker_by_name(**kw)
    with pytest.raises(ValueError, match="Kalman filter does not support normalized_instance_similarity."):
        Tracker.make_tracker_by_name(**dict(kw, tracker="simple", similarity="normalized_instance"))
