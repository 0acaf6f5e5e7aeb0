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
