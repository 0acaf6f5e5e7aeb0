"""Array-backed result containers (sleap_amd/io/labels.py): views equal the arrays they were built from, `numpy()` follows
sleap/io/dataset.py:2442-2561 (checked against a literal loop over frame / instance objects), save -> load round trip."""
import numpy as np
import pytest

from sleap_amd.io.labels import Labels
from sleap_amd.io.video import Video
from test_slp_writer import _random_outs


def _ref_numpy(labels, n_video_frames, all_frames, untracked, return_confidence):
    """dataset.py:2493-2561 on the LabeledFrame / PredictedInstance views."""
    lfs = labels.labeled_frames
    idxs = sorted(lf.frame_idx for lf in lfs)
    first = 0 if all_frames else idxs[0]
    last = n_video_frames - 1 if all_frames else idxs[-1]
    n_insts = max(lf.n_predicted_instances for lf in lfs)
    untracked = untracked or n_insts == 1
    n_tracks = n_insts if untracked else len(labels.tracks)
    out = np.full((last - first + 1, n_tracks, len(labels.skeleton.nodes), 3 if return_confidence else 2), np.nan, np.float32)
    for lf in lfs:
        i = lf.frame_idx - first
        for j, inst in enumerate(lf.predicted_instances):
            if not untracked:
                if inst.track is None:
                    continue
                j = labels.tracks.index(inst.track)
            out[i, j] = inst.points_and_scores_array if return_confidence else inst.numpy()
    return out


@pytest.mark.parametrize("tracked", [False, True])
def test_views_and_numpy(tracked):
    outs = _random_outs(7, tracked)
    video = Video.from_numpy(np.zeros((40, 8, 8, 1), np.uint8))
    labels = Labels.from_predictions(outs, ["a", "b", "c", "d"], [(0, 1), (1, 2), (1, 3)], video=video,
                                     track_names=[f"track_{i}" for i in range(6)] if tracked else None)
    frame_ind = np.concatenate([ex["frame_ind"] for ex in outs])
    assert len(labels) == len(frame_ind) and [lf.frame_idx for lf in labels] == frame_ind.tolist()
    assert labels.skeleton.node_names == ["a", "b", "c", "d"] and labels.skeleton.edge_names[2] == ("b", "d")
    assert labels.videos == [video] and labels[-1].frame_idx == int(frame_ind[-1])
    # every instance view equals the arrays it came from (all-NaN instances and slots beyond n_valid are dropped)
    k = 0
    for ex in outs:
        for b in range(len(ex["frame_ind"])):
            lf = labels[k]
            k += 1
            if tracked:
                keep = [i for i in np.argsort(ex["track_order"][b], kind="stable") if ex["track_order"][b, i] >= 0
                        and i < ex["n_valid"][b] and not np.isnan(ex["instance_peaks"][b, i]).all()]
            else:
                keep = [i for i in range(int(ex["n_valid"][b])) if not np.isnan(ex["instance_peaks"][b, i]).all()]
            assert len(lf) == len(keep)
            for inst, i in zip(lf.instances, keep):
                pts = ex["instance_peaks"][b, i].copy()
                pts[np.isnan(pts).any(axis=1)] = np.nan
                np.testing.assert_array_equal(inst.numpy(), pts)
                assert inst.score == pytest.approx(float(ex["instance_scores"][b, i]))
                if tracked:
                    assert inst.track.name == f"track_{int(ex['track_inds'][b, i])}"
                    assert inst.tracking_score == pytest.approx(float(ex["tracking_scores"][b, i]), rel=1e-6)
                else:
                    assert inst.track is None
    for all_frames in (True, False):
        for untracked in (False, True):
            for conf in (False, True):
                if not tracked and not untracked:
                    continue  # no tracks: the tracked view has zero slots in the reference as well
                got = labels.numpy(all_frames=all_frames, untracked=untracked, return_confidence=conf)
                want = _ref_numpy(labels, 40, all_frames, untracked, conf)
                np.testing.assert_array_equal(got, want)
    assert labels.numpy().shape[0] == 40


def test_save_load_round_trip(tmp_path):
    outs = _random_outs(9, True)
    labels = Labels.from_predictions(outs, ["a", "b", "c", "d"], [(0, 1), (1, 2), (1, 3)],
                                     video={"filename": "clip.mp4", "grayscale": True, "bgr": True, "dataset": "", "input_format": ""},
                                     track_names=[f"trk{i}" for i in range(6)], provenance={"model": "x"})
    path = str(tmp_path / "p.slp")
    labels.save(path)
    back = Labels.load_file(path)
    assert len(back) == len(labels) and back.skeleton.node_names == labels.skeleton.node_names
    assert back.skeleton.edge_inds == labels.skeleton.edge_inds and back.provenance == {"model": "x"}
    assert [t.name for t in back.tracks] == [t.name for t in labels.tracks] and back.video["filename"] == "clip.mp4"
    for a, b in zip(labels, back):
        assert a.frame_idx == b.frame_idx and len(a) == len(b)
        for x, y in zip(a.instances, b.instances):
            np.testing.assert_array_equal(x.numpy(), y.numpy())
            np.testing.assert_array_equal(x.scores, y.scores)
            assert x.score == y.score and (x.track.name if x.track else None) == (y.track.name if y.track else None)
    np.testing.assert_array_equal(labels.numpy(untracked=True), back.numpy(untracked=True))
    back.save(str(tmp_path / "p2.slp"))  # a loaded file can be written again
    assert len(Labels.load_file(str(tmp_path / "p2.slp")).predicted_instances) == len(labels.predicted_instances)


def test_predict_to_labels_with_tracker_without_gpu(tmp_path):
    """`Predictor.predict(frames)` end to end on the host side -- pipelined generator, native tracker, Labels views,
    `.numpy()` -- around a stand-in model (two animals walking; the "network" reads them out of the frame)."""
    from types import SimpleNamespace

    import torch

    from sleap_amd.nn.inference import BottomUpInferenceModel, BottomUpPredictor
    from sleap_amd.nn.tracking import Tracker

    T, N = 10, 3
    rng = np.random.default_rng(0)
    walk = np.cumsum(rng.normal(0, 1.0, (T, 2, 1, 2)), axis=0) + np.array([[[20.0, 20.0]], [[80.0, 60.0]]])[None]
    pts = (walk + np.arange(N)[None, None, :, None] * 3.0).astype(np.float32)  # (T, 2 animals, N nodes, 2)
    frames = np.zeros((T, 8, 8, 1), np.uint8)
    frames[:, 0, 0, 0] = np.arange(T)
    layer = SimpleNamespace(paf_scorer=SimpleNamespace(max_instances=4, n_nodes=N, max_node_peaks=8, part_names=["a", "b", "c"],
                                                       edge_inds=[(0, 1), (1, 2)]),
                            max_peaks=64, keras_model=SimpleNamespace(device=torch.device("cpu")), last_upload_done=None)

    class Model(BottomUpInferenceModel):
        def __init__(self):
            self.bottomup_layer = layer

        def call(self, batch):
            t = batch[:, 0, 0, 0].to(torch.int64).numpy()
            b = len(t)
            peaks = np.full((b, 4, N, 2), np.nan, np.float32)
            vals = np.full((b, 4, N), np.nan, np.float32)
            scores = np.full((b, 4), np.nan, np.float32)
            for f in range(b):
                order = [1, 0] if t[f] % 2 else [0, 1]  # the detector does not keep the animals in order
                for i, a in enumerate(order):
                    peaks[f, i], vals[f, i], scores[f, i] = pts[t[f], a], 0.9, 2.0
            return {"instance_peaks": torch.from_numpy(peaks), "instance_peak_vals": torch.from_numpy(vals),
                    "instance_scores": torch.from_numpy(scores), "n_valid": torch.full((b,), 2, dtype=torch.int32),
                    "status": torch.zeros((b,), dtype=torch.int32)}

    pred = BottomUpPredictor.__new__(BottomUpPredictor)
    pred.inference_model, pred.batch_size, pred.verbosity, pred.report_rate = Model(), 4, "none", 2.0
    pred.max_instances = None
    pred.tracker = Tracker.make_tracker_by_name(tracker="simple", similarity="centroid", match="hungarian", track_window=3)
    labels = pred.predict(frames)  # make_labels=True, the reference's default
    assert len(labels) == T and len(labels.tracks) == 2
    arr = labels.numpy()  # (frames, tracks, nodes, 2): each track follows ONE animal although the detection order alternates
    assert arr.shape == (T, 2, N, 2)
    first = [int(np.argmin([np.abs(arr[0, k] - pts[0, a]).max() for a in range(2)])) for k in range(2)]
    assert sorted(first) == [0, 1]
    for k in range(2):
        np.testing.assert_allclose(arr[:, k], pts[:, first[k]], atol=1e-5)
    lf = labels[3]
    assert lf.frame_idx == 3 and len(lf.instances) == 2 and all(i.track is not None for i in lf.instances)
