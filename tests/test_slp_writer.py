"""`.slp` prediction writer (sleap_amd/io/slp.py): the vectorised table builder against (1) the tables
stored in the reference's own files (tests/golden/slp, extracted by tools/make_golden_slp.py), (2) an object-by-object
restatement of the reference's loops (inference.py:3273-3322 + hdf5.py:462-522), and an HDF5 write / read round trip."""
import json
import os

import numpy as np
import pytest

from sleap_amd.io import slp

GOLD = os.path.join(os.path.dirname(__file__), "golden", "slp")


def _load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize("name", ["bottomup.labels_pr.val", "bottomup.labels_pr.train"])
def test_reproduces_reference_prediction_files(name):
    """Feed the predictions stored in the reference's file back through the builder: identical tables (the files predate
    the tracking_score column, every other column must match bit for bit)."""
    g = _load(name)
    n_nodes = int(g["instances"]["point_id_end"][0] - g["instances"]["point_id_start"][0])
    arrays = slp.tables_to_arrays(g, n_nodes)
    for ex in arrays:
        del ex["track_inds"], ex["tracking_scores"]
    t = slp.build_tables(arrays)
    assert t["frames"].dtype == g["frames"].dtype and np.array_equal(t["frames"], g["frames"])
    for col in g["instances"].dtype.names:
        assert np.array_equal(t["instances"][col], g["instances"][col]), col
    assert t["pred_points"].dtype == g["pred_points"].dtype
    for col in ("x", "y", "visible", "complete", "score"):
        assert np.array_equal(t["pred_points"][col], g["pred_points"][col], equal_nan=True), col
    assert len(t["points"]) == 0 and t["points"].dtype == g["points"].dtype


def test_schema_matches_format_1_2_file():
    """dance.mp4.labels.slp is a format-1.2 file: same dataset dtypes, and its predicted instances round-trip."""
    g = _load("dance.labels")
    assert float(g["format_id"]) == slp.FORMAT_ID
    assert g["instances"].dtype == slp.INSTANCE_DTYPE and g["frames"].dtype == slp.FRAME_DTYPE
    assert g["pred_points"].dtype == slp.PRED_POINT_DTYPE and g["points"].dtype == slp.POINT_DTYPE
    meta = json.loads(str(g["json"]))
    ours = slp.skeleton_json([n["name"] for n in meta["nodes"]], [(l["source"], l["target"]) for l in
                                                                  sorted(meta["skeletons"][0]["links"], key=lambda l: l["edge_insert_idx"])])
    assert set(ours) == set(meta) and set(ours["skeletons"][0]) == set(meta["skeletons"][0])
    assert ours["nodes"] == meta["nodes"] and ours["version"] == meta["version"]
    assert {json.dumps(l["type"]) for l in ours["skeletons"][0]["links"]} == {json.dumps(l["type"]) for l in meta["skeletons"][0]["links"]}


def _object_oracle(outs, max_instances=None):
    """The reference's loops on plain Python objects: inference.py:3273-3322 then hdf5.py:462-522."""
    frames, instances, points = [], [], []
    seen_tracks = []
    for ex in outs:
        for b in range(len(ex["frame_ind"])):
            nv = int(ex["n_valid"][b])
            insts = []
            for i in range(nv):
                pts = ex["instance_peaks"][b, i]
                if np.isnan(pts).all():
                    continue
                insts.append({"i": i, "pts": pts, "conf": ex["instance_peak_vals"][b, i], "score": ex["instance_scores"][b, i],
                              "track": None, "tscore": 0.0})
            if "track_inds" in ex:
                kept = [a for a in insts if ex["track_order"][b, a["i"]] >= 0]
                kept.sort(key=lambda a: ex["track_order"][b, a["i"]])
                for a in kept:
                    a["track"] = int(ex["track_inds"][b, a["i"]])
                    a["tscore"] = float(ex["tracking_scores"][b, a["i"]])
                insts = kept
            elif max_instances is not None:
                insts = sorted(insts, key=lambda a: a["score"], reverse=True)[: min(max_instances, len(insts))]
            start = len(instances)
            for a in insts:
                if a["track"] is not None and a["track"] not in seen_tracks:
                    seen_tracks.append(a["track"])
                pid = len(points)
                for p, c in zip(a["pts"], a["conf"]):
                    if np.isnan(p).any():
                        points.append((np.nan, np.nan, True, False, 0.0))
                    else:
                        points.append((float(p[0]), float(p[1]), True, False, float(c)))
                instances.append((len(instances), 1, len(frames), 0, a["track"], -1, a["score"], pid, len(points), a["tscore"]))
            frames.append((len(frames), int(ex["video_ind"][b]), int(ex["frame_ind"][b]), start, len(instances)))
    inst = np.zeros(len(instances), slp.INSTANCE_DTYPE)
    for k, row in enumerate(instances):
        row = list(row)
        row[4] = -1 if row[4] is None else seen_tracks.index(row[4])
        inst[k] = tuple(row)
    return (np.array(frames, slp.FRAME_DTYPE) if frames else np.zeros(0, slp.FRAME_DTYPE), inst,
            np.array(points, slp.PRED_POINT_DTYPE) if points else np.zeros(0, slp.PRED_POINT_DTYPE), seen_tracks)


def _random_outs(seed, tracked):
    rng = np.random.default_rng(seed)
    outs, f0 = [], 10
    for b, imax in zip((3, 4, 2), (5, 3, 6)):
        peaks = rng.uniform(0, 300, (b, imax, 4, 2)).astype(np.float32)
        peaks[rng.random((b, imax, 4)) < 0.25] = np.nan
        peaks[rng.random((b, imax)) < 0.15] = np.nan  # whole instance missing
        nv = rng.integers(0, imax + 1, (b,)).astype(np.int32)
        ex = {"instance_peaks": peaks, "instance_peak_vals": rng.random((b, imax, 4)).astype(np.float32),
              "instance_scores": np.round(rng.random((b, imax)), 1).astype(np.float32), "n_valid": nv,
              "frame_ind": np.arange(f0, f0 + b), "video_ind": np.zeros(b, np.int64)}
        if tracked:
            ex["track_inds"] = np.full((b, imax), -1, np.int32)
            ex["tracking_scores"] = np.full((b, imax), np.nan)
            ex["track_order"] = np.full((b, imax), -1, np.int32)
            for f in range(b):
                live = [i for i in range(nv[f]) if not np.isnan(peaks[f, i]).all() and rng.random() < 0.85]
                for k, i in enumerate(rng.permutation(live)):
                    ex["track_order"][f, i] = k
                    ex["track_inds"][f, i] = rng.integers(0, 6)
                    ex["tracking_scores"][f, i] = rng.random()
        outs.append(ex)
        f0 += b
    return outs


@pytest.mark.parametrize("tracked,max_instances", [(False, None), (False, 2), (True, None)])
def test_build_tables_equals_object_oracle(tracked, max_instances):
    for seed in range(5):
        outs = _random_outs(seed, tracked)
        t = slp.build_tables(outs, max_instances=max_instances)
        fr, inst, pp, tracks = _object_oracle(outs, max_instances)
        assert np.array_equal(t["frames"], fr)
        for col in slp.INSTANCE_DTYPE.names:
            assert np.array_equal(t["instances"][col], inst[col], equal_nan=True), col
        for col in slp.PRED_POINT_DTYPE.names:
            assert np.array_equal(t["pred_points"][col], pp[col], equal_nan=True), col
        assert t["tracks"].tolist() == tracks


def test_hdf5_round_trip(tmp_path):
    outs = _random_outs(3, tracked=True)
    path = str(tmp_path / "pred.slp")
    t = slp.write_slp(path, outs, ["a", "b", "c", "d"], [(0, 1), (1, 2), (1, 3)],
                      video={"filename": "clip.mp4", "grayscale": True, "bgr": True, "dataset": "", "input_format": ""},
                      track_names=[f"track_{i}" for i in range(6)], provenance={"sleap_amd": "test"})
    r = slp.read_slp(path)
    assert float(r["format_id"]) == 1.2
    for k in ("frames", "instances", "pred_points", "points"):
        assert r[k].dtype == t[k].dtype
        for col in t[k].dtype.names:
            assert np.array_equal(r[k][col], t[k][col], equal_nan=True), (k, col)
    assert json.loads(str(r["videos_json"][0]))["backend"]["filename"] == "clip.mp4"
    tr = [json.loads(s) for s in r["tracks_json"].tolist()]
    assert [x[1] for x in tr] == [f"track_{i}" for i in t["tracks"].tolist()] and all(isinstance(x[0], int) for x in tr)
    meta = json.loads(str(r["json"]))
    assert [n["name"] for n in meta["nodes"]] == ["a", "b", "c", "d"] and meta["provenance"] == {"sleap_amd": "test"}
    assert len(r["suggestions_json"]) == 0
    back = slp.tables_to_arrays(r, 4)[0]
    assert np.array_equal(back["frame_ind"], np.concatenate([ex["frame_ind"] for ex in outs]))
