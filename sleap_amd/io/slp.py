"""Columnar fast path from the predictor's gathered arrays to the `.slp` (HDF5) prediction file.

The reference builds one `PredictedInstance` / `LabeledFrame` Python object per result
(sleap/nn/inference.py:3230-3348) and serialises them back into flat tables in `LabelsV1Adaptor.write`
(sleap/io/format/hdf5.py:265-575). Here the tables are assembled directly, vectorised over all frames:

    frames       (frame_id u8, video u4, frame_idx u8, instance_id_start u8, instance_id_end u8)
    instances    (instance_id i8, instance_type u1 = 1, frame_id u8, skeleton u4 = 0, track i4, from_predicted i8 = -1,
                  score f4, point_id_start u8, point_id_end u8, tracking_score f4)
    pred_points  (x f8, y f8, visible ? = True, complete ? = False, score f8)      -- one row per skeleton node
    points       empty (user-labelled instances only)

with the reference's object semantics: instances whose points are all NaN are skipped (inference.py:3285-3286), a node
with any NaN coordinate is stored as the default point (NaN, NaN, visible, score 0: `from_arrays` skips it,
instance.py:1108-1113), `max_instances` keeps the highest scoring instances (stable, :3297-3304), a tracker's returned
order (matches first, then new tracks) is the order of the frame's instances, every frame gets a row even when empty,
`track` indexes the list of tracks in order of first appearance, metadata JSON in the layout of
`Labels.to_dict(skip_labels=True)`.

HDF5 itself is written by h5py: in-process when importable, else by `sleap_amd/io/_slp_io.py` (a self-contained file of the package) under the interpreter named by
SLEAP_AMD_H5_PYTHON (default /opt/conda/bin/python3.9), exactly like `model_io` reads `best_model.h5`.
"""
import json
import os
import subprocess
import tempfile
from typing import Dict, List, Optional, Sequence

import numpy as np

FORMAT_ID = 1.2  # hdf5.py:31

INSTANCE_DTYPE = np.dtype([("instance_id", "i8"), ("instance_type", "u1"), ("frame_id", "u8"), ("skeleton", "u4"),
                           ("track", "i4"), ("from_predicted", "i8"), ("score", "f4"), ("point_id_start", "u8"),
                           ("point_id_end", "u8"), ("tracking_score", "f4")])
FRAME_DTYPE = np.dtype([("frame_id", "u8"), ("video", "u4"), ("frame_idx", "u8"), ("instance_id_start", "u8"),
                        ("instance_id_end", "u8")])
POINT_DTYPE = np.dtype([("x", "f8"), ("y", "f8"), ("visible", "?"), ("complete", "?")])
PRED_POINT_DTYPE = np.dtype([("x", "f8"), ("y", "f8"), ("visible", "?"), ("complete", "?"), ("score", "f8")])


def _cat(outs: Sequence[Dict[str, np.ndarray]], key: str, fill, imax: int):
    parts = []
    for ex in outs:
        v = np.asarray(ex[key])
        if v.ndim >= 2 and v.shape[1] < imax:
            pad = [(0, 0), (0, imax - v.shape[1])] + [(0, 0)] * (v.ndim - 2)
            v = np.pad(v, pad, constant_values=fill)
        parts.append(v)
    return np.concatenate(parts, axis=0)


def build_tables(outs: Sequence[Dict[str, np.ndarray]], max_instances: Optional[int] = None) -> Dict[str, np.ndarray]:
    """Per-batch prediction dicts (Predictor.predict(make_labels=False)) -> the four `.slp` tables + the list of track ids
    in order of first appearance (`tracks`, values are the tracker's integer ids)."""
    outs = list(outs)
    if not outs:
        return {"frames": np.zeros(0, FRAME_DTYPE), "instances": np.zeros(0, INSTANCE_DTYPE),
                "pred_points": np.zeros(0, PRED_POINT_DTYPE), "points": np.zeros(0, POINT_DTYPE),
                "tracks": np.zeros(0, np.int64), "track_spawned_on": np.zeros(0, np.int64)}
    imax = max(np.asarray(ex["instance_peaks"]).shape[1] for ex in outs)
    peaks = _cat(outs, "instance_peaks", np.nan, imax).astype(np.float32, copy=False)
    F, I, N = peaks.shape[0], peaks.shape[1], peaks.shape[2]
    vals = _cat(outs, "instance_peak_vals", np.nan, imax) if "instance_peak_vals" in outs[0] else np.zeros((F, I, N), np.float32)
    scores = _cat(outs, "instance_scores", np.nan, imax) if "instance_scores" in outs[0] else np.zeros((F, I), np.float32)
    n_valid = np.concatenate([np.asarray(ex["n_valid"]) for ex in outs]) if "n_valid" in outs[0] else np.full((F,), I)
    frame_ind = np.concatenate([np.asarray(ex["frame_ind"]) for ex in outs]).astype(np.int64)
    video_ind = (np.concatenate([np.asarray(ex["video_ind"]) for ex in outs]).astype(np.int64)
                 if "video_ind" in outs[0] else np.zeros((F,), np.int64))
    tracked = "track_inds" in outs[0]
    keep = (np.arange(I)[None, :] < n_valid[:, None]) & ~np.isnan(peaks).all(axis=(2, 3))
    # position of every instance inside its frame's list
    pos = np.where(keep, np.cumsum(keep, axis=1) - 1, I + 1).astype(np.int64)
    if max_instances is not None and not tracked:
        # sorted(key=score, reverse=True)[:max_instances]: descending, ties keep their original order
        key = np.where(keep, -scores.astype(np.float64), np.inf)
        rank = np.argsort(np.argsort(key, axis=1, kind="stable"), axis=1, kind="stable")
        keep = keep & (rank < max_instances)
        pos = np.where(keep, rank, I + 1)
    if tracked:
        trk = _cat(outs, "track_inds", -1, imax).astype(np.int64)
        tsc = _cat(outs, "tracking_scores", np.nan, imax).astype(np.float64)
        order = _cat(outs, "track_order", -1, imax).astype(np.int64)
        keep = keep & (order >= 0)
        pos = np.where(keep, order, I + 1)
    f_idx, i_idx = np.nonzero(keep)
    srt = np.lexsort((pos[f_idx, i_idx], f_idx))
    f_idx, i_idx = f_idx[srt], i_idx[srt]
    n_inst = f_idx.size
    counts = np.bincount(f_idx, minlength=F)
    ends = np.cumsum(counts)
    frames = np.zeros(F, FRAME_DTYPE)
    frames["frame_id"] = np.arange(F)
    frames["video"] = video_ind
    frames["frame_idx"] = frame_ind
    frames["instance_id_start"] = ends - counts
    frames["instance_id_end"] = ends
    inst = np.zeros(n_inst, INSTANCE_DTYPE)
    inst["instance_id"] = np.arange(n_inst)
    inst["instance_type"] = 1
    inst["frame_id"] = f_idx
    inst["skeleton"] = 0
    inst["from_predicted"] = -1
    inst["score"] = scores[f_idx, i_idx]
    inst["point_id_start"] = np.arange(n_inst) * N
    inst["point_id_end"] = np.arange(n_inst) * N + N
    tracks = np.zeros(0, np.int64)
    spawned = np.zeros(0, np.int64)
    if tracked:
        tid = trk[f_idx, i_idx]
        uniq, first = np.unique(tid[tid >= 0], return_index=True)
        by_appearance = np.argsort(first, kind="stable")
        tracks = uniq[by_appearance]
        lut = {int(t): k for k, t in enumerate(tracks)}
        inst["track"] = np.array([lut.get(int(t), -1) for t in tid], dtype=np.int32) if n_inst else np.zeros(0, np.int32)
        spawned = frame_ind[f_idx[tid >= 0][first[by_appearance]]] if tracks.size else spawned
        ts = tsc[f_idx, i_idx]
        inst["tracking_score"] = np.where(np.isnan(ts), 0.0, ts)
    else:
        inst["track"] = -1
        inst["tracking_score"] = 0.0
    p = peaks[f_idx, i_idx].astype(np.float64)  # (n_inst, N, 2)
    missing = np.isnan(p).any(axis=2)
    pp = np.zeros(n_inst * N, PRED_POINT_DTYPE)
    pp["x"] = np.where(missing, np.nan, p[..., 0]).reshape(-1)
    pp["y"] = np.where(missing, np.nan, p[..., 1]).reshape(-1)
    pp["visible"] = True
    pp["complete"] = False
    pp["score"] = np.where(missing, 0.0, vals[f_idx, i_idx].astype(np.float64)).reshape(-1)
    return {"frames": frames, "instances": inst, "pred_points": pp, "points": np.zeros(0, POINT_DTYPE), "tracks": tracks,
            "track_spawned_on": spawned}


def skeleton_json(part_names: Sequence[str], edges: Sequence[Sequence[int]], name: str = "Skeleton-0") -> dict:
    """`Labels.to_dict(skip_labels=True)` for one skeleton (node-link graph with jsonpickle'd EdgeType.BODY, as stored in
    the reference's files): links index the global node list, the first link carries the enum, later ones refer to it."""
    links = []
    for k, (s, d) in enumerate(edges):
        typ = {"py/reduce": [{"py/type": "sleap.skeleton.EdgeType"}, {"py/tuple": [1]}]} if k == 0 else {"py/id": 1}
        links.append({"edge_insert_idx": k, "key": 0, "source": int(s), "target": int(d), "type": typ})
    return {"version": "2.0.0",
            "skeletons": [{"directed": True, "graph": {"name": name, "num_edges_inserted": len(edges)}, "links": links,
                           "multigraph": True, "nodes": [{"id": i} for i in range(len(part_names))]}],
            "nodes": [{"name": n, "weight": 1.0} for n in part_names],
            "videos": [], "tracks": [], "suggestions": [], "negative_anchors": {}, "provenance": {}}


def _dumps(o) -> str:
    return json.dumps(o, separators=(",", ":"))


def _h5_python() -> str:
    return os.environ.get("SLEAP_AMD_H5_PYTHON", "/opt/conda/bin/python3.9")


def _tool() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "_slp_io.py")


def write_tables(filename: str, t: Dict[str, np.ndarray], part_names: Sequence[str], edges: Sequence[Sequence[int]],
                 video: Optional[dict] = None, track_names: Optional[Sequence[str]] = None,
                 provenance: Optional[dict] = None) -> None:
    """Write already built tables (build_tables) in the reference's `.slp` layout."""
    meta = skeleton_json(part_names, edges)
    if provenance:
        meta["provenance"] = provenance
    videos = [_dumps({"backend": video or {"filename": "", "grayscale": True, "bgr": True, "dataset": "", "input_format": ""}})]
    tracks = []
    for tid, spawned in zip(np.asarray(t["tracks"]).tolist(), np.asarray(t["track_spawned_on"]).tolist()):
        nm = track_names[tid] if track_names is not None and tid < len(track_names) else f"track_{tid}"
        tracks.append(_dumps([int(spawned), nm]))
    payload = {"frames": t["frames"], "instances": t["instances"], "pred_points": t["pred_points"], "points": t["points"],
               "videos_json": np.array(videos, dtype=str), "tracks_json": np.array(tracks, dtype=str),
               "suggestions_json": np.array([], dtype=str), "format_id": np.float64(FORMAT_ID),
               "json": np.array(_dumps(meta), dtype=str)}
    with tempfile.TemporaryDirectory() as td:
        npz = os.path.join(td, "t.npz")
        np.savez(npz, **payload)
        try:
            import h5py  # noqa: F401

            from . import _slp_io

            _slp_io.write(npz, filename)
        except ImportError:
            subprocess.run([_h5_python(), _tool(), "write", npz, filename], check=True)


def write_slp(filename: str, outs: Sequence[Dict[str, np.ndarray]], part_names: Sequence[str], edges: Sequence[Sequence[int]],
              video: Optional[dict] = None, track_names: Optional[Sequence[str]] = None,
              max_instances: Optional[int] = None, provenance: Optional[dict] = None) -> Dict[str, np.ndarray]:
    """Write predictions to `filename` in the reference's `.slp` layout; returns the tables that were written.

    video: the `backend` dictionary of the source video, e.g. {"filename": "clip.mp4", "grayscale": True, "bgr": True,
    "dataset": "", "input_format": ""}; track_names[i] names tracker track i (default "track_<i>")."""
    t = build_tables(outs, max_instances=max_instances)
    write_tables(filename, t, part_names, edges, video=video, track_names=track_names, provenance=provenance)
    return t


def read_slp(filename: str) -> Dict[str, np.ndarray]:
    """-> {"frames", "instances", "points", "pred_points", "videos_json", "tracks_json", "suggestions_json", "format_id",
    "json"} (tables as structured arrays)."""
    with tempfile.TemporaryDirectory() as td:
        npz = os.path.join(td, "t.npz")
        try:
            import h5py  # noqa: F401

            from . import _slp_io

            _slp_io.read(filename, npz)
        except ImportError:
            subprocess.run([_h5_python(), _tool(), "read", filename, npz], check=True)
        z = np.load(npz, allow_pickle=False)
        return {k: z[k] for k in z.files}


def tables_to_arrays(tables: Dict[str, np.ndarray], n_nodes: int) -> List[Dict[str, np.ndarray]]:
    """Inverse of `build_tables` (one batch): structured tables -> NaN-padded arrays, for reading prediction files."""
    fr, inst, pp = tables["frames"], tables["instances"], tables["pred_points"]
    F = len(fr)
    counts = (fr["instance_id_end"] - fr["instance_id_start"]).astype(np.int64)
    I = int(counts.max()) if F else 0
    peaks = np.full((F, I, n_nodes, 2), np.nan, np.float32)
    vals = np.full((F, I, n_nodes), np.nan, np.float32)
    scores = np.full((F, I), np.nan, np.float32)
    trk = np.full((F, I), -1, np.int32)
    tsc = np.full((F, I), np.nan, np.float64)
    for f in range(F):
        for k, j in enumerate(range(int(fr["instance_id_start"][f]), int(fr["instance_id_end"][f]))):
            a, b = int(inst["point_id_start"][j]), int(inst["point_id_end"][j])
            peaks[f, k, : b - a, 0] = pp["x"][a:b]
            peaks[f, k, : b - a, 1] = pp["y"][a:b]
            vals[f, k, : b - a] = pp["score"][a:b]
            scores[f, k] = inst["score"][j]
            trk[f, k] = inst["track"][j]
            if "tracking_score" in inst.dtype.names:
                tsc[f, k] = inst["tracking_score"][j]
    return [{"instance_peaks": peaks, "instance_peak_vals": vals, "instance_scores": scores, "n_valid": counts.astype(np.int32),
             "frame_ind": fr["frame_idx"].astype(np.int64), "video_ind": fr["video"].astype(np.int64), "track_inds": trk,
             "tracking_scores": tsc}]
