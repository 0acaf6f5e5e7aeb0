"""The native tracker (csrc/tracker.hip through the C ABI) against the NumPy restatement of the reference's tracker
(oracle/tracking.py, itself pinned to the reference's tests in test_oracle_tracking.py): identical track assignment, order
of returned instances and spawned-track numbering on random walks with drop-outs, missing nodes and extra detections;
tracking scores to 1e-12 (exp() implementations differ in the last bit). Host code only: runs without a GPU."""
import itertools

import numpy as np
import pytest

from oracle import tracking as T


def _sequence(seed, n_frames=40, n_animals=4, n_nodes=6, p_drop=0.15, p_miss=0.15, p_extra=0.1, step=3.0):
    rng = np.random.default_rng(seed)
    base = rng.uniform(20, 200, (n_animals, 1, 2)) + rng.normal(0, 6, (n_animals, n_nodes, 2))
    frames = []
    for _ in range(n_frames):
        base = base + rng.normal(0, step, (n_animals, 1, 2))
        insts = []
        for a in rng.permutation(n_animals):
            if rng.random() < p_drop:
                continue
            pts = (base[a] + rng.normal(0, 0.5, (n_nodes, 2))).astype(np.float32)
            pts[rng.random(n_nodes) < p_miss] = np.nan
            insts.append((pts, rng.uniform(0.2, 1, n_nodes).astype(np.float32), np.float32(rng.uniform(0.3, 1))))
        if rng.random() < p_extra:
            pts = rng.uniform(20, 200, (n_nodes, 2)).astype(np.float32)
            insts.append((pts, rng.uniform(0.2, 1, n_nodes).astype(np.float32), np.float32(rng.uniform(0.1, 0.5))))
        frames.append(insts)
    frames[3] = []  # an empty frame
    return frames


def _run_oracle(frames, img_hw, **kw):
    tr = T.Tracker(**kw)
    out = []
    for insts in frames:
        lst = [T.Inst(p, s, sc, uid=i) for i, (p, s, sc) in enumerate(insts)]
        res = tr.track(lst, img_hw=img_hw)
        out.append([(r.uid, r.track, r.tracking_score) for r in res])
    return out, len(tr.spawned_tracks)


def _run_native(frames, img_hw, **kw):
    from sleap_amd.nn.tracking import Tracker

    tr = Tracker.make_tracker_by_name(**kw)
    out = []
    for insts in frames:
        if insts:
            pts = np.stack([p for p, _, _ in insts])
            ps = np.stack([s for _, s, _ in insts])
            sc = np.array([c for _, _, c in insts], np.float32)
            r = tr.track(pts, ps, sc, img_hw=img_hw)
        else:
            r = tr.track(np.zeros((0, 6, 2), np.float32), img_hw=img_hw)
        out.append(list(zip(r["index"].tolist(), r["track"].tolist(), r["tracking_score"].tolist())))
    return out, len(tr.spawned_tracks)


def _compare(a, b):
    assert len(a) == len(b)
    for f, (fa, fb) in enumerate(zip(a, b)):
        assert [(u, t) for u, t, _ in fa] == [(u, t) for u, t, _ in fb], f"frame {f}: {fa} vs {fb}"
        for (_, _, sa_), (_, _, sb) in zip(fa, fb):
            assert sa_ == pytest.approx(sb, rel=1e-12, abs=1e-300), f"frame {f}"


CASES = [dict(tracker=t, similarity=s, match=m) for t, s, m in itertools.product(
    ["simple", "simplemaxtracks"], ["instance", "normalized_instance", "centroid", "iou", "object_keypoint"],
    ["greedy", "hungarian"])]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c['tracker']}-{c['similarity']}-{c['match']}")
def test_native_tracker_matches_oracle(case):
    kw = dict(case, track_window=3)
    if case["tracker"] == "simplemaxtracks":
        kw.update(max_tracks=4, max_tracking=True)
    frames = _sequence(hash(str(sorted(case.items()))) % 1000, step=0.7 if case["similarity"] == "instance" else 3.0)
    a, na = _run_oracle(frames, (256, 320), **kw)
    b, nb = _run_native(frames, (256, 320), **kw)
    _compare(a, b)
    assert na == nb and na >= 4


@pytest.mark.parametrize("kw", [
    dict(tracker="simple", similarity="instance", match="greedy", robust=0.8, track_window=5),
    dict(tracker="simple", similarity="iou", match="hungarian", robust=0.95, track_window=4),
    dict(tracker="simple", similarity="centroid", match="greedy", min_new_track_points=5, min_match_points=4),
    dict(tracker="simple", similarity="centroid", match="hungarian", target_instance_count=3, pre_cull_to_target=True),
    dict(tracker="simple", similarity="iou", match="greedy", target_instance_count=3, pre_cull_to_target=True,
         pre_cull_iou_threshold=0.3),
    dict(tracker="simple", similarity="object_keypoint", match="greedy", oks_errors=[1.5, 2.5, 4], oks_score_weighting=True,
         oks_normalization="union"),
    dict(tracker="simple", similarity="object_keypoint", match="hungarian", oks_errors=[3.0], oks_normalization="ref"),
    dict(tracker="simple", similarity="centroid", match="greedy", max_tracks=3, max_tracking=True, track_window=2),
    dict(tracker="simplemaxtracks", similarity="iou", match="greedy", max_tracks=2, max_tracking=False),
], ids=lambda k: "-".join(f"{a}={b}" for a, b in k.items() if a not in ("tracker", "similarity", "match"))[:60])
def test_native_tracker_options(kw):
    frames = _sequence(11, n_frames=50, p_drop=0.25, p_extra=0.3)
    a, na = _run_oracle(frames, (1, 1), **kw)
    b, nb = _run_native(frames, (1, 1), **kw)
    _compare(a, b)
    assert na == nb


def test_reference_known_answers_native():
    """The reference's max-tracking scenarios (tests/nn/test_tracker_components.py) through the native tracker."""
    from test_oracle_tracking import EXTRA, GAP_BOTH, GAP_SINGLE, make_insts

    def n_tracks(trx, **kw):
        frames = [[(i.points.astype(np.float32), np.ones(3, np.float32), np.float32(1)) for i in f] for f in make_insts(trx)]
        out, _ = _run_native([[(p, s, c) for p, s, c in f] for f in frames], (1, 1), **kw)
        return len({t for f in out for _, t, _ in f})

    simple = dict(tracker="simple", match="hungarian", track_window=2)
    maxtr = dict(tracker="simplemaxtracks", match="hungarian", track_window=2, max_tracks=2, max_tracking=True)
    assert n_tracks(GAP_SINGLE, **simple) == 3 and n_tracks(GAP_SINGLE, **maxtr) == 2
    assert n_tracks(GAP_BOTH, **simple) == 4 and n_tracks(GAP_BOTH, **maxtr) == 2
    assert n_tracks(EXTRA, **simple) == 4 and n_tracks(EXTRA, **maxtr) == 2


def test_track_frames_equals_per_frame_calls_and_connect_breaks():
    from sleap_amd.nn.tracking import Tracker, connect_single_track_breaks

    frames = _sequence(5, n_frames=30, n_nodes=6)
    I = max(len(f) for f in frames)
    pts = np.full((len(frames), I, 6, 2), np.nan, np.float32)
    ps = np.zeros((len(frames), I, 6), np.float32)
    sc = np.zeros((len(frames), I), np.float32)
    nv = np.zeros((len(frames),), np.int32)
    for f, insts in enumerate(frames):
        nv[f] = len(insts)
        for i, (p, s, c) in enumerate(insts):
            pts[f, i], ps[f, i], sc[f, i] = p, s, c
    kw = dict(tracker="simple", similarity="iou", match="hungarian", track_window=3, target_instance_count=4,
              post_connect_single_breaks=True)
    res = Tracker.make_tracker_by_name(**kw).track_frames(pts, ps, sc, nv, img_hw=(1, 1))
    per, _ = _run_native(frames, (1, 1), **kw)
    for f, lst in enumerate(per):
        assert sorted((u, t) for u, t, _ in lst) == sorted((i, int(res["track"][f, i])) for i in range(I) if res["track"][f, i] >= 0)
        for k, (u, _, s) in enumerate(lst):
            assert res["order"][f, u] == k and res["tracking_score"][f, u] == s
    # connect_single_track_breaks vs the oracle on the tracked table
    ofr = [[T.Inst([[0, 0]], track=int(res["track"][f, i]), uid=i) for i in np.argsort(res["order"][f], kind="stable")
            if res["track"][f, i] >= 0] for f in range(len(frames))]
    T.connect_single_track_breaks(ofr, 4)
    tbl = res["track"].copy()
    connect_single_track_breaks(tbl, 4, res["order"])
    for f, lst in enumerate(ofr):
        for inst in lst:
            assert tbl[f, inst.uid] == inst.track
    assert (tbl != res["track"]).any()  # the scenario actually reconnects something


def test_errors_like_reference():
    from sleap_amd.nn.tracking import Tracker

    with pytest.raises(ValueError, match="is not a valid tracker"):
        Tracker.make_tracker_by_name(tracker="nope")
    with pytest.raises(ValueError, match="not a valid tracker similarity"):
        Tracker.make_tracker_by_name(tracker="simple", similarity="cosine")
    with pytest.raises(ValueError, match="not a valid tracker matching"):
        Tracker.make_tracker_by_name(tracker="simple", match="optimal")
    with pytest.raises(ValueError, match="Kalman filter requires max tracks or target instance count"):
        Tracker.make_tracker_by_name(tracker="simple", kf_init_frame_count=10)
    d = Tracker.make_tracker_by_name()  # the reference's default is the optical-flow tracker
    assert d.uses_image and d.get_name() == "FlowCandidateMaker.instance_similarity.greedy_matching"
    assert Tracker.make_tracker_by_name(tracker="flow", img_scale=0.5).img_scale == 0.5
    with pytest.raises(ValueError, match="img_scale"):
        Tracker.make_tracker_by_name(tracker="flow", img_scale=0.0)
    assert Tracker.make_tracker_by_name(tracker="flow", save_shifted_instances=True).uses_image
    # tracking.py:914-919: only "flow" (not "flowmaxtracks") takes the of_* / img_scale arguments
    m = Tracker.make_tracker_by_name(tracker="flowmaxtracks", img_scale=0.5, of_window_size=9, max_tracks=2, max_tracking=True)
    assert m.of_window_size == 21 and m.has_max_tracking and m.uses_image
    t = Tracker.make_tracker_by_name(tracker="simple", similarity="iou", match="hungarian")
    assert t.get_name() == "SimpleCandidateMaker.instance_iou.hungarian_matching"


def test_retrack_prediction_file(tmp_path):
    """read .slp -> run_tracker (time step inferred, as tracking.py:1542-1581) -> write .slp, against the oracle tracker."""
    import json

    from sleap_amd.io import slp
    from sleap_amd.nn.tracking import retrack

    frames = _sequence(21, n_frames=25, n_nodes=6, p_miss=0.1)
    I = max(len(f) for f in frames)
    ex = {"instance_peaks": np.full((len(frames), I, 6, 2), np.nan, np.float32),
          "instance_peak_vals": np.zeros((len(frames), I, 6), np.float32), "instance_scores": np.zeros((len(frames), I), np.float32),
          "n_valid": np.array([len(f) for f in frames], np.int32), "frame_ind": np.arange(100, 100 + len(frames)),
          "video_ind": np.zeros(len(frames), np.int64)}
    for f, insts in enumerate(frames):
        for i, (p, s, c) in enumerate(insts):
            ex["instance_peaks"][f, i], ex["instance_peak_vals"][f, i], ex["instance_scores"][f, i] = p, s, c
    names = list("abcdef")
    src, dst = str(tmp_path / "pred.slp"), str(tmp_path / "pred.tracked.slp")
    slp.write_slp(src, [ex], names, [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5)], video={"filename": "v.mp4", "grayscale": True,
                                                                                      "bgr": True, "dataset": "", "input_format": ""})
    kw = dict(tracker="simple", similarity="centroid", match="hungarian", track_window=3)
    t = retrack(src, dst, **kw)
    r = slp.read_slp(dst)
    # oracle on the instances as stored in the file (frames with all-NaN instances already dropped by the writer)
    ot = T.Tracker(**kw)
    back = slp.tables_to_arrays(slp.read_slp(src), 6)[0]
    want = []
    for f in range(len(frames)):
        nv = int(back["n_valid"][f])
        lst = [T.Inst(back["instance_peaks"][f, i], back["instance_peak_vals"][f, i], back["instance_scores"][f, i], uid=i)
               for i in range(nv)]
        want.append([(x.uid, x.track) for x in ot.track(lst, img_hw=(1, 1))])
    seen = []
    for f, lst in enumerate(want):
        a, b = int(r["frames"]["instance_id_start"][f]), int(r["frames"]["instance_id_end"][f])
        assert b - a == len(lst)
        for k, (uid, tr) in enumerate(lst):
            if tr not in seen:
                seen.append(tr)
            assert int(r["instances"]["track"][a + k]) == seen.index(tr)
            p0 = int(r["instances"]["point_id_start"][a + k])
            np.testing.assert_array_equal(r["pred_points"]["x"][p0:p0 + 6].astype(np.float32), back["instance_peaks"][f, uid, :, 0])
    assert [json.loads(s)[1] for s in r["tracks_json"].tolist()] == [f"track_{i}" for i in seen]
    assert json.loads(str(r["json"]))["nodes"][2]["name"] == "c" and np.array_equal(r["frames"]["frame_idx"], ex["frame_ind"])
    assert len(t["tracks"]) == len(seen) >= 4


@pytest.mark.parametrize("max_tracks", [2, 3, 5])
def test_max_tracks_matching_queue(max_tracks):
    """tests/nn/test_inference.py::test_max_tracks_matching_queue: with max tracking on, the matching queue never holds more
    than `max_tracks` tracks nor more than `track_window` instances per track. (The number of SPAWNED tracks can exceed
    max_tracks: spawn_for_untracked_instances checks the queue, which is only updated after the whole frame, so a first frame
    with 5 detections spawns 5 tracks of which 2 enter the queue -- the reference behaves like that and so do both
    implementations here.) Checked on the oracle's state; the native tracker must return identical assignments."""
    frames = _sequence(100 + max_tracks, n_frames=60, n_animals=5, p_drop=0.2, p_extra=0.4)
    kw = dict(tracker="simple", similarity="instance", match="greedy", track_window=5, max_tracks=max_tracks, max_tracking=True)
    tr = T.Tracker(**kw)
    a = []
    for insts in frames:
        res = tr.track([T.Inst(p, s, sc, uid=i) for i, (p, s, sc) in enumerate(insts)], img_hw=(1, 1))
        a.append([(r.uid, r.track, r.tracking_score) for r in res])
        assert len(tr.track_matching_queue_dict) <= max_tracks
        assert all(len(q) <= 5 for q in tr.track_matching_queue_dict.values())
    b, nb = _run_native(frames, (1, 1), **kw)
    _compare(a, b)
    assert nb == len(tr.spawned_tracks)


# ---------------------------------------------------------------------------------------------------------------------------
# What reaches the tracker from a prediction dict (inference.py:2641-2668, 3283-3313): all-NaN instances never do, bottom-up
# `max_instances` is applied (by score, stable) BEFORE tracking, `t` follows frame_ind.
# ---------------------------------------------------------------------------------------------------------------------------
def _example_from(frames, n_nodes=6, nan_slots=()):
    """frames (list of instance lists) -> prediction dict with NaN-padded arrays; `nan_slots` = {(frame, slot)} of all-NaN
    instances inserted in the MIDDLE of the valid range (what the top-down model emits for a crop with no node above the
    threshold)."""
    rows = []
    for f, insts in enumerate(frames):
        lst = list(insts)
        for (ff, slot) in sorted(nan_slots):
            if ff == f:
                lst.insert(min(slot, len(lst)), (np.full((n_nodes, 2), np.nan, np.float32),
                                                 np.full((n_nodes,), np.nan, np.float32), np.float32(0.9)))
        rows.append(lst)
    I = max(len(r) for r in rows)
    F = len(rows)
    ex = {"instance_peaks": np.full((F, I, n_nodes, 2), np.nan, np.float32),
          "instance_peak_vals": np.full((F, I, n_nodes), np.nan, np.float32),
          "instance_scores": np.full((F, I), np.nan, np.float32), "n_valid": np.zeros((F,), np.int64),
          "frame_ind": np.arange(F, dtype=np.int64), "video_ind": np.zeros((F,), np.int64)}
    for f, lst in enumerate(rows):
        ex["n_valid"][f] = len(lst)
        for i, (p, s, c) in enumerate(lst):
            ex["instance_peaks"][f, i], ex["instance_peak_vals"][f, i], ex["instance_scores"][f, i] = p, s, c
    return ex, rows


def _oracle_over_rows(rows, img_hw, max_instances=None, t_of=None, **kw):
    """The reference's loop: drop all-NaN, optional sorted-by-score truncation, tracker.track(t=frame_ind)."""
    tr = T.Tracker(**kw)
    out = []
    for f, lst in enumerate(rows):
        insts = [T.Inst(p, s, c, uid=i) for i, (p, s, c) in enumerate(lst) if not np.isnan(p).all()]
        if max_instances is not None:
            insts = sorted(insts, key=lambda a: a.score, reverse=True)[: min(max_instances, len(insts))]
        res = tr.track(insts, img_hw=img_hw, t=None if t_of is None else int(t_of[f]))
        out.append([(r.uid, r.track) for r in res])
    return out, tr


@pytest.mark.parametrize("kw", [dict(tracker="simple", similarity="instance", match="greedy"),
                                dict(tracker="simple", similarity="normalized_instance", match="hungarian", robust=0.9),
                                dict(tracker="simplemaxtracks", similarity="iou", match="hungarian", max_tracks=4,
                                     max_tracking=True)])
def test_all_nan_instances_never_reach_the_tracker(kw):
    from sleap_amd.nn.tracking import Tracker, track_example

    frames = _sequence(11, n_frames=25)
    nan_slots = {(2, 1), (5, 0), (6, 2), (7, 1), (12, 1), (20, 3)}
    ex, rows = _example_from(frames, nan_slots=nan_slots)
    want, otr = _oracle_over_rows(rows, (200, 220), t_of=ex["frame_ind"], **kw)
    tr = Tracker.make_tracker_by_name(**kw)
    track_example(tr, ex, img_hw=(200, 220))
    for f, lst in enumerate(want):
        got = sorted((int(ex["track_order"][f, i]), i, int(ex["track_inds"][f, i])) for i in range(ex["track_inds"].shape[1])
                     if ex["track_order"][f, i] >= 0)
        assert [(i, t) for _, i, t in got] == lst, f"frame {f}"
    for (f, slot) in nan_slots:
        assert ex["track_inds"][f, slot] == -1 and ex["track_order"][f, slot] == -1
    assert len(tr.spawned_tracks) == len(otr.spawned_tracks)
    # and a control: feeding the NaN instances unfiltered changes the numbering (or makes the Hungarian cost matrix infeasible)
    from sleap_amd._lib import SleapAmdError

    ex2, _ = _example_from(frames, nan_slots=nan_slots)
    tr2 = Tracker.make_tracker_by_name(**kw)
    try:
        r = tr2.track_frames(ex2["instance_peaks"], ex2["instance_peak_vals"], ex2["instance_scores"], ex2["n_valid"],
                             img_hw=(200, 220), t0=0)
        assert len(tr2.spawned_tracks) != len(tr.spawned_tracks) or (r["track"] != ex["track_inds"]).any()
    except SleapAmdError as e:
        assert "infeasible" in str(e)


def test_max_instances_is_applied_before_tracking_and_survives_to_the_tables():
    from sleap_amd.io import slp
    from sleap_amd.nn.tracking import Tracker, finish_tracks, track_example

    kw = dict(tracker="simple", similarity="instance", match="greedy")
    frames = _sequence(12, n_frames=20, p_extra=0.5)
    ex, rows = _example_from(frames)
    want, _ = _oracle_over_rows(rows, (1, 1), max_instances=2, t_of=ex["frame_ind"], **kw)
    tr = Tracker.make_tracker_by_name(**kw)
    track_example(tr, ex, img_hw=(1, 1), max_instances=2)
    finish_tracks([ex], tr)
    t = slp.build_tables([ex], max_instances=2)
    for f, lst in enumerate(want):
        a, b = int(t["frames"]["instance_id_start"][f]), int(t["frames"]["instance_id_end"][f])
        assert b - a == len(lst) <= 2
        got_tracks = [int(t["tracks"][k]) for k in t["instances"]["track"][a:b]]
        assert got_tracks == [trk for _, trk in lst], f"frame {f}"
        # the rows written are the instances the oracle tracked, in its order
        for k, (uid, _) in enumerate(lst):
            np.testing.assert_array_equal(t["instances"]["score"][a + k], rows[f][uid][2])


def test_tracker_time_follows_frame_ind():
    """A reader over a subset of frames (VideoReader(example_indices=...)): `t` is the frame index, not position."""
    from sleap_amd.nn.tracking import Tracker, track_example

    frames = _sequence(13, n_frames=12)
    ex, rows = _example_from(frames)
    ex["frame_ind"] = np.array([0, 1, 2, 10, 11, 12, 30, 31, 40, 50, 51, 52], np.int64)
    kw = dict(tracker="simple", similarity="centroid", match="hungarian")
    want, otr = _oracle_over_rows(rows, (1, 1), t_of=ex["frame_ind"], **kw)
    tr = Tracker.make_tracker_by_name(**kw)
    track_example(tr, ex)
    for f, lst in enumerate(want):
        got = sorted((int(ex["track_order"][f, i]), i, int(ex["track_inds"][f, i])) for i in range(ex["track_inds"].shape[1])
                     if ex["track_order"][f, i] >= 0)
        assert [(i, t) for _, i, t in got] == lst


def test_instance_scores_for_topdown_and_single_instance_outputs():
    """inference.py:2639-2660 (top-down: instance score = centroid confidence) and :1578 (single instance: nansum of the
    point confidences) -- what Labels / .slp files store as `score` and what the tracker's culling ranks by."""
    from sleap_amd.io import slp
    from sleap_amd.nn.inference import Predictor

    rng = np.random.default_rng(0)
    p = Predictor.__new__(Predictor)
    p.tracker = None
    pk = rng.random((3, 2, 4, 2)).astype(np.float32)
    pv = rng.random((3, 2, 4)).astype(np.float32)
    pv[1, 0, 2] = np.nan
    cv = rng.random((3, 2)).astype(np.float32)
    base = {"instance_peaks": pk, "instance_peak_vals": pv, "n_valid": np.array([2, 2, 1]), "frame_ind": np.arange(3),
            "video_ind": np.zeros(3, np.int64)}
    td = p._apply_tracker([dict(base, centroids=pk[:, :, 0], centroid_vals=cv)])
    np.testing.assert_array_equal(td[0]["instance_scores"], cv)
    t = slp.build_tables(td)
    np.testing.assert_array_equal(t["instances"]["score"], [cv[0, 0], cv[0, 1], cv[1, 0], cv[1, 1], cv[2, 0]])
    si = p._apply_tracker([dict(base)])
    np.testing.assert_allclose(si[0]["instance_scores"], np.nansum(pv, axis=-1))
    assert np.isfinite(si[0]["instance_scores"]).all()
    bu = p._apply_tracker([dict(base, instance_scores=cv * 2)])
    np.testing.assert_array_equal(bu[0]["instance_scores"], cv * 2)  # a model that emits scores keeps them
