"""Array-backed stand-ins for the reference's result containers, so that `Predictor.predict(data)` (make_labels=True, the
reference's default) works without the `sleap` package being installed.

The reference returns `sleap.Labels` holding one `LabeledFrame` per frame and one `PredictedInstance` per animal
(sleap/io/dataset.py, sleap/instance.py) -- Python objects built one by one (inference.py:3230-3348). Here `Labels` keeps the
`.slp` tables built by `sleap_amd.io.slp.build_tables` (same filtering / ordering rules as the reference's object builder)
and materialises `LabeledFrame` / `PredictedInstance` views lazily. Covered surface: `len`, indexing / iteration,
`labeled_frames`, `videos`, `skeleton(s)`, `tracks`, `numpy(video, all_frames, untracked, return_confidence)` with the
reference's semantics (dataset.py:2442-2561), `save(filename)` -> a `.slp` file `sleap.load_file` opens, `load_file`.
These are NOT the reference classes (no GUI state, suggestions, user instances, merging); `Labels.to_sleap()` converts to the
real ones when `sleap` is importable.
"""
import json
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import slp


class Skeleton:
    def __init__(self, node_names: Sequence[str], edge_inds: Sequence[Sequence[int]], name: str = "Skeleton-0"):
        self.node_names = list(node_names)
        self.edge_inds = [(int(a), int(b)) for a, b in edge_inds]
        self.name = name

    @property
    def nodes(self) -> List[str]:
        return self.node_names

    @property
    def edge_names(self):
        return [(self.node_names[a], self.node_names[b]) for a, b in self.edge_inds]

    def __len__(self):
        return len(self.node_names)


class Track:
    def __init__(self, spawned_on: int = 0, name: str = ""):
        self.spawned_on, self.name = int(spawned_on), name

    def __repr__(self):
        return f"Track(spawned_on={self.spawned_on}, name={self.name!r})"


class PredictedInstance:
    """points (N, 2) float32 with NaN rows for missing nodes, scores (N,), instance score, track, tracking score."""

    def __init__(self, points, scores, score, skeleton, track: Optional[Track] = None, tracking_score: float = 0.0):
        self.points_array = np.asarray(points, dtype=np.float32)
        self.scores = np.asarray(scores, dtype=np.float32)
        self.score = float(score)
        self.skeleton = skeleton
        self.track = track
        self.tracking_score = float(tracking_score)

    def numpy(self) -> np.ndarray:
        return self.points_array

    @property
    def points_and_scores_array(self) -> np.ndarray:
        return np.concatenate([self.points_array, self.scores[:, None]], axis=1)

    @property
    def n_visible_points(self) -> int:
        return int((~np.isnan(self.points_array).any(axis=1)).sum())


class LabeledFrame:
    def __init__(self, video, frame_idx: int, instances: List[PredictedInstance]):
        self.video, self.frame_idx, self.instances = video, int(frame_idx), instances

    predicted_instances = property(lambda self: self.instances)
    user_instances = property(lambda self: [])
    n_predicted_instances = property(lambda self: len(self.instances))
    n_user_instances = property(lambda self: 0)

    def __len__(self):
        return len(self.instances)

    def __getitem__(self, index):  # sleap/instance.py:1390-1392: `labeled_frame[i]` is its i-th instance
        return self.instances[index]

    def __iter__(self):
        return iter(self.instances)

    def numpy(self) -> np.ndarray:
        """(n_instances, n_nodes, 2) with NaN for missing nodes (instance.py `LabeledFrame.numpy`)."""
        if not self.instances:
            return np.zeros((0, 0, 2), np.float32)
        return np.stack([inst.numpy() for inst in self.instances])


class Labels:
    def __init__(self, tables: Dict[str, np.ndarray], skeleton: Skeleton, video=None, track_names: Optional[Sequence[str]] = None,
                 provenance: Optional[dict] = None):
        self._t = tables
        self.skeleton = skeleton
        self.video = video
        self.provenance = provenance or {}
        tids = np.asarray(tables.get("tracks", np.zeros(0, np.int64))).tolist()
        spawned = np.asarray(tables.get("track_spawned_on", np.zeros(0, np.int64))).tolist()
        self.tracks = [Track(s, track_names[t] if track_names is not None and t < len(track_names) else f"track_{t}")
                       for t, s in zip(tids, spawned)]
        self._track_names = None if track_names is None else list(track_names)
        self._n = len(skeleton)

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_predictions(cls, outs: Sequence[Dict[str, np.ndarray]], part_names: Sequence[str], edges: Sequence[Sequence[int]],
                         video=None, track_names: Optional[Sequence[str]] = None, max_instances: Optional[int] = None,
                         provenance: Optional[dict] = None) -> "Labels":
        return cls(slp.build_tables(outs, max_instances=max_instances), Skeleton(part_names, edges), video, track_names, provenance)

    @classmethod
    def load_file(cls, filename: str) -> "Labels":
        r = slp.read_slp(filename)
        meta = json.loads(str(r["json"]))
        names = [n["name"] for n in meta["nodes"]]
        sk = meta["skeletons"][0]
        order = [n["id"] for n in sk["nodes"]]
        pos = {g: k for k, g in enumerate(order)}
        edges = [(pos[l["source"]], pos[l["target"]]) for l in sorted(sk["links"], key=lambda l: l["edge_insert_idx"])]
        tr = [json.loads(s) for s in r["tracks_json"].tolist()]
        t = {k: r[k] for k in ("frames", "instances", "pred_points", "points")}
        t["tracks"] = np.arange(len(tr), dtype=np.int64)
        t["track_spawned_on"] = np.array([x[0] for x in tr], dtype=np.int64)
        video = json.loads(str(r["videos_json"][0]))["backend"] if len(r["videos_json"]) else None
        return cls(t, Skeleton([names[i] for i in order], edges, sk["graph"].get("name", "Skeleton-0")), video,
                   [x[1] for x in tr], meta.get("provenance"))

    # ------------------------------------------------------------------ container surface
    skeletons = property(lambda self: [self.skeleton])
    videos = property(lambda self: [self.video])

    def __len__(self) -> int:
        return len(self._t["frames"])

    def _frame(self, i: int) -> LabeledFrame:
        fr, inst, pp = self._t["frames"][i], self._t["instances"], self._t["pred_points"]
        out = []
        for j in range(int(fr["instance_id_start"]), int(fr["instance_id_end"])):
            a, b = int(inst["point_id_start"][j]), int(inst["point_id_end"][j])
            pts = np.stack([pp["x"][a:b], pp["y"][a:b]], axis=1).astype(np.float32)
            tr = int(inst["track"][j])
            ts = float(inst["tracking_score"][j]) if "tracking_score" in inst.dtype.names else 0.0
            out.append(PredictedInstance(pts, pp["score"][a:b], inst["score"][j], self.skeleton,
                                         self.tracks[tr] if tr >= 0 else None, ts))
        return LabeledFrame(self.video, int(fr["frame_idx"]), out)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._frame(k) for k in range(*i.indices(len(self)))]
        return self._frame(int(i) % len(self) if i < 0 else int(i))

    def __iter__(self):
        return (self._frame(i) for i in range(len(self)))

    @property
    def labeled_frames(self) -> List[LabeledFrame]:
        return list(self)

    @property
    def predicted_instances(self) -> List[PredictedInstance]:
        return [inst for lf in self for inst in lf.instances]

    # ------------------------------------------------------------------ dataset.py:2442-2561
    def numpy(self, video=None, all_frames: bool = True, untracked: bool = False, return_confidence: bool = False) -> np.ndarray:
        """-> (n_frames, n_tracks, n_nodes, 2 | 3) float32, NaN where there is no data; tracked instances go to their
        track's slot (instances without a track are skipped) unless `untracked` or every frame has at most one instance.
        `all_frames` needs the video length (a `Video` object or anything with `len`); without it the array spans 0 ..
        the last predicted frame."""
        fr, inst, pp = self._t["frames"], self._t["instances"], self._t["pred_points"]
        if len(fr) == 0:
            raise IndexError("There are no labeled frames. No points matrix to return.")
        frame_idx = fr["frame_idx"].astype(np.int64)
        counts = (fr["instance_id_end"] - fr["instance_id_start"]).astype(np.int64)
        first = 0 if all_frames else int(frame_idx.min())
        if all_frames and self.video is not None and hasattr(self.video, "__len__"):
            last = len(self.video) - 1
        else:
            last = int(frame_idx.max())
        n_insts = int(counts.max())
        untracked = untracked or n_insts == 1
        n_tracks = n_insts if untracked else len(self.tracks)
        c = 3 if return_confidence else 2
        out = np.full((last - first + 1, n_tracks, self._n, c), np.nan, dtype=np.float32)
        for f in range(len(fr)):
            i = int(frame_idx[f]) - first
            if i < 0 or i >= out.shape[0]:
                continue
            for k, j in enumerate(range(int(fr["instance_id_start"][f]), int(fr["instance_id_end"][f]))):
                slot = k if untracked else int(inst["track"][j])
                if slot < 0:
                    continue
                a, b = int(inst["point_id_start"][j]), int(inst["point_id_end"][j])
                out[i, slot, : b - a, 0] = pp["x"][a:b]
                out[i, slot, : b - a, 1] = pp["y"][a:b]
                if return_confidence:
                    out[i, slot, : b - a, 2] = pp["score"][a:b]
        return out

    def save(self, filename: str) -> None:
        v = self.video.backend_dict() if hasattr(self.video, "backend_dict") else (self.video if isinstance(self.video, dict) else None)
        names = None
        if len(self.tracks):  # tables index tracker ids; names are stored per id
            names = {int(t): tr.name for t, tr in zip(np.asarray(self._t["tracks"]).tolist(), self.tracks)}
            names = [names.get(i, f"track_{i}") for i in range(max(names) + 1)]
        slp.write_tables(filename, self._t, self.skeleton.node_names, self.skeleton.edge_inds, video=v, track_names=names,
                         provenance=self.provenance or None)

    def to_sleap(self):  # pragma: no cover (needs the sleap package)
        """The same predictions as the reference's own `sleap.Labels`."""
        import sleap

        sk = sleap.Skeleton.from_names_and_edge_inds(self.skeleton.node_names, self.skeleton.edge_inds)
        tracks = [sleap.Track(spawned_on=t.spawned_on, name=t.name) for t in self.tracks]
        video = self.video if isinstance(self.video, sleap.Video) else None
        lfs = []
        for lf in self:
            insts = [sleap.PredictedInstance.from_numpy(points=i.points_array, point_confidences=i.scores, instance_score=i.score,
                                                        skeleton=sk, track=tracks[self.tracks.index(i.track)] if i.track else None,
                                                        tracking_score=i.tracking_score) for i in lf.instances]
            lfs.append(sleap.LabeledFrame(video=video, frame_idx=lf.frame_idx, instances=insts))
        return sleap.Labels(lfs)
