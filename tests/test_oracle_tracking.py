"""oracle/tracking.py against the reference's own known-answer tests
(tests/nn/test_tracker_components.py: test_nms, test_nms_with_target, test_nms_instances_to_remove,
test_frame_match_object, test_max_tracking_large_gap_single_track, test_max_tracking_small_gap_on_both_tracks,
test_max_tracking_extra_detections) re-expressed on plain arrays."""
import numpy as np
import pytest

from oracle import tracking as T


def test_nms():
    boxes = np.array([[10, 10, 20, 20], [10, 10, 15, 15], [30, 30, 40, 40], [32, 32, 42, 42]])
    scores = np.array([1, 0.3, 1, 0.5])
    assert sorted(T.nms_fast(boxes, scores, iou_threshold=0.5)) == [0, 2]


def test_nms_with_target():
    boxes = np.array([[10, 10, 20, 20], [10, 10, 15, 15], [30, 30, 40, 40], [32, 32, 42, 42]])
    assert sorted(T.nms_fast(boxes, np.array([1, 0.3, 1, 0.5]), iou_threshold=0.5, target_count=3)) == [0, 2, 3]
    assert sorted(T.nms_fast(boxes, np.array([1, 0.5, 1, 0.3]), iou_threshold=0.5, target_count=3)) == [0, 1, 2]


def test_nms_instances_to_remove():
    pts = [((10, 10), (20, 20), 1), ((10, 10), (15, 15), 0.3), ((30, 30), (40, 40), 1), ((32, 32), (42, 42), 0.5)]
    insts = [T.Inst(np.array([a, b], float), score=s) for a, b, s in pts]
    keep, remove = T.nms_instances(insts, iou_threshold=0.5, target_count=3)
    assert len(remove) == 1 and remove[0] is insts[1]


def test_frame_match_object():
    instances, tracks = ["instance a", "instance b"], ["track a", "track b"]
    fm = T.FrameMatches.from_cost_matrix(np.array([[10, 200], [75, 150]]), instances, tracks, T.greedy_matching)
    assert not fm.has_only_first_choice_matches
    m = fm.matches
    assert len(m) == 2
    assert (m[0].track, m[0].instance, m[0].score) == ("track a", "instance a", -10)
    assert (m[1].track, m[1].instance, m[1].score) == ("track b", "instance b", -150)
    fm = T.FrameMatches.from_cost_matrix(np.array([[10, 200], [150, 75]]), instances, tracks, T.greedy_matching)
    assert fm.has_only_first_choice_matches


def make_insts(trx):
    def make_inst(x, y):
        return T.Inst(np.array([[-0.1, -0.1], [0.0, 0.0], [0.1, 0.1]]) + np.array([[x, y]]), [1, 1, 1], 1)

    return [[make_inst(x, y) for x, y in frame] for frame in trx]


def _n_tracks(preds, **kw):
    tr = T.Tracker(**kw)
    tracked = [tr.track(list(insts), img_hw=(1, 1)) for insts in preds]
    return len({i.track for f in tracked for i in f}), tracked


GAP_SINGLE = [[(0, 0), (0, 1)], [(0.1, 0), (0.1, 1)], [(0.2, 0), (0.2, 1)], [(0.3, 0)], [(0.4, 0)], [(0.5, 0), (0.5, 1)],
              [(0.6, 0), (0.6, 1)]]
GAP_BOTH = [[(0, 0), (0, 1)], [(0.1, 0), (0.1, 1)], [(0.2, 0), (0.2, 1)], [], [], [(0.5, 0), (0.5, 1)], [(0.6, 0), (0.6, 1)]]
EXTRA = [[(0, 0), (0, 1)], [(0.1, 0), (0.1, 1)], [(0.2, 0), (0.2, 1)], [(0.3, 0)], [(0.4, 0)], [(0.5, 0), (0.5, 1)],
         [(0.6, 0), (0.6, 1), (0.6, 0.5)]]


def test_max_tracking_large_gap_single_track():
    assert _n_tracks(make_insts(GAP_SINGLE), tracker="simple", match="hungarian", track_window=2)[0] == 3
    assert _n_tracks(make_insts(GAP_SINGLE), tracker="simplemaxtracks", match="hungarian", track_window=2, max_tracks=2,
                     max_tracking=True)[0] == 2


def test_max_tracking_small_gap_on_both_tracks():
    assert _n_tracks(make_insts(GAP_BOTH), tracker="simple", match="hungarian", track_window=2)[0] == 4
    assert _n_tracks(make_insts(GAP_BOTH), tracker="simplemaxtracks", match="hungarian", track_window=2, max_tracks=2,
                     max_tracking=True)[0] == 2


def test_max_tracking_extra_detections():
    assert _n_tracks(make_insts(EXTRA), tracker="simple", match="hungarian", track_window=2)[0] == 4
    assert _n_tracks(make_insts(EXTRA), tracker="simplemaxtracks", match="hungarian", track_window=2, max_tracks=2,
                     max_tracking=True)[0] == 2


def test_similarity_known_values():
    """Hand-computed values of the similarity definitions (components.py:33-196)."""
    a = T.Inst([[0, 0], [1, 1], [np.nan, np.nan]])
    b = T.Inst([[0, 1], [1, 1], [5, 5]])
    assert T.instance_similarity(a, b) == pytest.approx((np.exp(-1) + 1) / 2)  # normalised by the REFERENCE's visible nodes
    assert T.instance_similarity(b, a) == pytest.approx((np.exp(-1) + 1) / 3)
    assert T.normalized_instance_similarity(a, b, img_hw=(2, 4)) == pytest.approx((np.exp(-0.25) + 1) / 2)
    assert T.centroid_distance(a, b) == pytest.approx(-np.linalg.norm(np.array([0.5, 0.5]) - np.array([1, 1])))
    # boxes [y1,x1,y2,x2] = [0,0,1,1] and [1,0,5,5] with the +1 pixel convention: inter 1x2, areas 4 and 30
    assert T.instance_iou(a, b) == pytest.approx(2 / (4 + 30 - 2))
    oks = T.factory_object_keypoint_similarity(keypoint_errors=[1, 2], score_weighting=False, normalization_keypoints="union")
    assert oks(a, b) == pytest.approx((np.exp(-1 / 2) + 1) / 2)  # errors padded by repeating the last value
    assert T.factory_object_keypoint_similarity()(T.Inst(np.full((0, 2), np.nan)), T.Inst(np.full((0, 2), np.nan))) == 0


def test_connect_single_track_breaks():
    def f(*tracks):
        return [T.Inst([[0, 0]], track=t) for t in tracks]

    frames = [f(0, 1), f(0, 1), f(0, 2), f(0, 2), f(0, 2, 3)]
    T.connect_single_track_breaks(frames, 2)
    assert [[i.track for i in fr] for fr in frames] == [[0, 1], [0, 1], [0, 1], [0, 1], [0, 1, 3]]
