"""Cross-frame identity tracking on the predictor's array outputs.

Mirrors `sleap.nn.tracking.Tracker` (sleap/nn/tracking.py:542-992) for the `simple` / `simplemaxtracks` candidate makers and
the optical-flow ones, `flow` (the reference's default) / `flowmaxtracks` (:108-440, 1194-1240): same `make_tracker_by_name`
keyword arguments, defaults and error messages, same per-frame `track` semantics
(candidate pool from the last `track_window` frames, similarity matrix against the best / robust-quantile candidate of
every track, greedy or Hungarian assignment, new tracks for what is left, optional pre-cull and single-break
connection). The work happens in `libsleap_amd.so` (`csrc/tracker.hip`, host C++ -- no Python objects per instance, no
GIL while a batch of frames is tracked); this module only marshals arrays.

Instances are arrays, not `PredictedInstance`s: points (n, N, 2) with NaN for missing nodes, point scores (n, N),
instance scores (n,). Tracks are integers, `spawned_tracks[i]` is the reference's name for track i.
Flow trackers take the frame with every step (`img=` / `images=`: uint8 arrays or CUDA tensors); the Lucas-Kanade flow that
the reference gets from `cv2.calcOpticalFlowPyrLK` runs on the device (csrc/flow.hip; `img_scale != 1`: the frames pass through the device restatement of `cv2.resize`, INTER_LINEAR on uint8). `kf_init_frame_count` / `kf_node_indices` wrap the tracker in the Kalman one (`kalman.py`).
"""
import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from .. import _lib
from .._lib import check

SIMILARITY = {"instance": 0, "centroid": 1, "iou": 2, "normalized_instance": 3, "object_keypoint": 4}
MATCH = {"greedy": 0, "hungarian": 1}
OKS_NORM = {"all": 0, "ref": 1, "union": 2}
_FLOW_STREAMS = {}  # device -> the flow trackers' stream


class _Config(C.Structure):
    _fields_ = [("max_tracks_mode", C.c_int), ("similarity", C.c_int), ("match", C.c_int), ("track_window", C.c_int),
                ("robust", C.c_double), ("min_new_track_points", C.c_int), ("min_match_points", C.c_int),
                ("target_instance_count", C.c_int), ("pre_cull_to_target", C.c_int), ("pre_cull_iou_threshold", C.c_double),
                ("max_tracks", C.c_int), ("max_tracking", C.c_int), ("oks_n_errors", C.c_int),
                ("oks_errors", C.POINTER(C.c_double)), ("oks_score_weighting", C.c_int), ("oks_normalization", C.c_int),
                ("flow", C.c_int), ("of_window_size", C.c_int), ("of_max_levels", C.c_int),
                ("save_shifted_instances", C.c_int), ("img_scale", C.c_double)]


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _device_frames(img, single: bool = False):
    """uint8 frames for the flow tracker as a contiguous CUDA tensor: (F, H, W, C), or (H, W, C) with `single`. Float frames
    are converted as `ensure_int` does (normalization.py:52-77)."""
    import torch

    if not torch.is_tensor(img):
        img = np.asarray(img)
        if img.dtype.kind == "f":
            img = (np.clip(np.floor(img.astype(np.float32) * np.float32(255.5)), 0, 255) if img.size and img.max() <= 1.0
                   else img).astype(np.uint8)
        img = torch.from_numpy(np.ascontiguousarray(img))
    if img.dtype != torch.uint8:
        raise TypeError("flow tracker: frames must be uint8 (or float arrays in [0, 1])")
    want = 3 if single else 4
    if img.dim() == want - 1:
        img = img[..., None]
    if img.dim() != want or img.shape[-1] not in (1, 3):
        raise ValueError(f"flow tracker: unsupported frame shape {tuple(img.shape)}")
    return img.contiguous().cuda()


class Tracker:
    """`Tracker.make_tracker_by_name(tracker="simple", similarity="instance", match="greedy", track_window=5, ...)`."""

    def __init__(self, **kwargs):
        self._h = None
        self._flow_stream = None
        self._configure(**kwargs)

    def _stream_for(self, frames, ready=None):
        """The HIP stream of the flow tracker's device work (pyramids, Lucas-Kanade), as a ctypes handle. It is a stream of the
        tracker's own: on the caller's current stream that work would queue behind whatever is already there -- inside
        `Predictor.predict` the NEXT batch's post-processing, which waits for the next batch's network -- and the tracker's
        stream synchronisations would serialise tracking with inference. `ready`: a CUDA event after which `frames` is
        complete (default: everything queued on the current stream so far)."""
        import torch

        if self._flow_stream is None:
            self._flow_stream = _FLOW_STREAMS.get(frames.device)
        if self._flow_stream is None:
            # one per device for all trackers (creating a hardware queue costs milliseconds)
            # (high priority = a hardware queue of its own on ROCm: normal-priority streams share a few queues, and behind a
            # stream that is parked on an event the tracker's launches would be parked too -- see InferenceLayer.run_network)
            self._flow_stream = _FLOW_STREAMS[frames.device] = torch.cuda.Stream(device=frames.device, priority=-1)
        st = self._flow_stream
        if ready is not None:
            st.wait_event(ready)
        else:
            st.wait_stream(torch.cuda.current_stream(frames.device))
        frames.record_stream(st)
        return C.c_void_p(st.cuda_stream)

    @classmethod
    def make_tracker_by_name(cls, tracker: str = "flow", similarity: str = "instance", match: str = "greedy",
                             track_window: int = 5, robust: float = 1.0, min_new_track_points: int = 0,
                             min_match_points: int = 0, img_scale: float = 1.0, of_window_size: int = 21,
                             of_max_levels: int = 3, save_shifted_instances: bool = False, target_instance_count: int = 0,
                             pre_cull_to_target: bool = False, pre_cull_iou_threshold: Optional[float] = None,
                             post_connect_single_breaks: bool = False, clean_instance_count: int = 0,
                             clean_iou_threshold: Optional[float] = None, kf_init_frame_count: int = 0,
                             kf_node_indices: Optional[list] = None, max_tracks: Optional[int] = None,
                             max_tracking: bool = False, oks_errors: Optional[list] = None,
                             oks_score_weighting: bool = False, oks_normalization: str = "all", **kwargs):
        obj = cls(tracker=tracker, similarity=similarity, match=match, track_window=track_window, robust=robust,
                  min_new_track_points=min_new_track_points, min_match_points=min_match_points, img_scale=img_scale,
                  of_window_size=of_window_size, of_max_levels=of_max_levels, save_shifted_instances=save_shifted_instances,
                  target_instance_count=target_instance_count, pre_cull_to_target=pre_cull_to_target,
                  pre_cull_iou_threshold=pre_cull_iou_threshold, post_connect_single_breaks=post_connect_single_breaks,
                  clean_instance_count=clean_instance_count, max_tracks=max_tracks, max_tracking=max_tracking,
                  oks_errors=oks_errors, oks_score_weighting=oks_score_weighting, oks_normalization=oks_normalization)
        # tracking.py:955-993: the Kalman tracker wraps the regular one
        if (max_tracks or target_instance_count) and kf_init_frame_count:
            if not kf_node_indices:
                raise ValueError("Kalman filter requires node indices for instance tracking.")
            if obj.uses_flow:
                raise ValueError("Kalman filter requires simple tracker for initial tracking.")
            if similarity == "normalized_instance":
                raise ValueError("Kalman filter does not support normalized_instance_similarity.")
            from .kalman import KalmanTracker

            return KalmanTracker.make_tracker(init_tracker=obj, init_frame_count=kf_init_frame_count,
                                              node_indices=list(kf_node_indices),
                                              instance_count=target_instance_count or max_tracks,
                                              instance_iou_threshold=pre_cull_iou_threshold)
        if kf_init_frame_count:
            raise ValueError("Kalman filter requires max tracks or target instance count.")
        return obj

    def _configure(self, tracker="simple", similarity="instance", match="greedy", track_window=5, robust=1.0,
                   min_new_track_points=0, min_match_points=0, img_scale=1.0, of_window_size=21, of_max_levels=3,
                   save_shifted_instances=False, target_instance_count=0, pre_cull_to_target=False,
                   pre_cull_iou_threshold=None, post_connect_single_breaks=False, clean_instance_count=0,
                   max_tracks=None, max_tracking=False, oks_errors=None,
                   oks_score_weighting=False, oks_normalization="all"):
        max_tracking = max_tracking if max_tracks else False  # tracking.py:879
        if max_tracking and tracker in ("simple", "flow"):
            tracker += "maxtracks"
        if tracker not in ("simple", "flow", "simplemaxtracks", "flowmaxtracks"):
            raise ValueError(f"{tracker} is not a valid tracker.")
        if similarity not in SIMILARITY:
            raise ValueError(f"{similarity} is not a valid tracker similarity function.")
        if match not in MATCH:
            raise ValueError(f"{match} is not a valid tracker matching function.")
        self.uses_flow = tracker.startswith("flow")
        if tracker != "flow":  # tracking.py:914-919 configures the candidate maker for "flow" only: "flowmaxtracks" keeps
            img_scale, of_window_size, of_max_levels, save_shifted_instances = 1.0, 21, 3, False  # the class defaults
        if self.uses_flow and not (img_scale and float(img_scale) > 0):
            raise ValueError("flow tracker: img_scale must be positive")
        if clean_instance_count:
            raise NotImplementedError("clean_instance_count (deprecated TrackCleaner) is not implemented; use "
                                      "target_instance_count with pre_cull_to_target")
        if oks_normalization not in OKS_NORM:
            raise ValueError(f"{oks_normalization} is not a valid object keypoint normalization.")
        self.tracker_name, self.similarity, self.match = tracker, similarity, match
        self.track_window = int(track_window)
        self.robust_best_instance = float(robust)
        self.min_new_track_points, self.min_match_points = int(min_new_track_points), int(min_match_points)
        self.target_instance_count = int(target_instance_count)
        self.post_connect_single_breaks = bool(post_connect_single_breaks)
        self.max_tracks, self.max_tracking = max_tracks, bool(max_tracking)
        errs = None
        if oks_errors is not None and np.size(oks_errors) > 0:
            errs = np.ascontiguousarray(np.atleast_1d(oks_errors), dtype=np.float64)
        self._oks_errors = errs  # keep alive
        self.of_window_size, self.of_max_levels = int(of_window_size), int(of_max_levels)
        cfg = _Config(1 if tracker.endswith("maxtracks") else 0, SIMILARITY[similarity], MATCH[match], self.track_window,
                      self.robust_best_instance, self.min_new_track_points, self.min_match_points,
                      self.target_instance_count, int(bool(pre_cull_to_target)),
                      float(pre_cull_iou_threshold) if pre_cull_iou_threshold else 0.0, int(max_tracks or 0),
                      int(self.max_tracking), 0 if errs is None else errs.size,
                      None if errs is None else errs.ctypes.data_as(C.POINTER(C.c_double)), int(bool(oks_score_weighting)),
                      OKS_NORM[oks_normalization], int(self.uses_flow), self.of_window_size, self.of_max_levels,
                      int(bool(save_shifted_instances) and tracker == "flow"), float(img_scale))
        self.img_scale = float(img_scale)
        h = _lib.lib()
        self._h = h.sa_tracker_create(C.byref(cfg))
        if not self._h:
            raise ValueError(h.sa_last_error().decode())

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                _lib.lib().sa_tracker_destroy(C.c_void_p(self._h))
            except Exception:
                pass
            self._h = None

    # ------------------------------------------------------------------ reference-shaped surface
    @property
    def is_valid(self) -> bool:
        return True

    @property
    def uses_image(self) -> bool:
        return self.uses_flow

    @property
    def has_max_tracking(self) -> bool:
        return self.tracker_name.endswith("maxtracks")

    @property
    def spawned_tracks(self) -> List[str]:
        return [f"track_{i}" for i in range(_lib.lib().sa_tracker_n_tracks(C.c_void_p(self._h)))]

    @property
    def last_first_choice(self) -> Optional[bool]:
        """`FrameMatches.has_only_first_choice_matches` (components.py:560-573) of the last tracked frame: every match was its
        instance's best column. None before the first frame."""
        v = _lib.lib().sa_tracker_last_first_choice(C.c_void_p(self._h))
        return None if v < 0 else bool(v)

    def reset_candidates(self):
        check(_lib.lib().sa_tracker_reset(C.c_void_p(self._h)), "sa_tracker_reset")

    def get_name(self) -> str:
        maker = ("Flow" if self.uses_flow else "Simple") + ("MaxTracksCandidateMaker" if self.has_max_tracking else "CandidateMaker")
        sim = {"instance": "instance_similarity", "centroid": "centroid_distance", "iou": "instance_iou",
               "normalized_instance": "normalized_instance_similarity",
               "object_keypoint": "object_keypoint_similarity"}[self.similarity]
        return f"{maker}.{sim}.{self.match}_matching"

    def track(self, points, point_scores=None, instance_scores=None, img_hw=(1, 1), img=None, t: Optional[int] = None):
        """One frame (tracking.py:642-773). -> dict(index (m,), track (m,), tracking_score (m,)): the tracked instances
        in the reference's order (matches first, then newly spawned tracks); `index` points back into the input."""
        pts = _f32(points)
        n = 0 if pts is None else pts.shape[0]
        n_nodes = pts.shape[1] if n else 1
        ps, sc = _f32(point_scores), _f32(instance_scores)
        if self.uses_flow and img is not None:
            frame = _device_frames(img, single=True)
            check(_lib.lib().sa_tracker_set_image(C.c_void_p(self._h), C.c_void_p(frame.data_ptr()), frame.shape[0], frame.shape[1],
                                                  frame.shape[2], self._stream_for(frame)), "sa_tracker_set_image")
        elif self.uses_flow and n:
            raise ValueError("flow tracker: track() needs the frame (img=...)")
        idx, trk = np.full((max(n, 1),), -1, np.int32), np.full((max(n, 1),), -1, np.int32)
        tsc = np.zeros((max(n, 1),), np.float64)
        n_out = C.c_int(0)
        check(_lib.lib().sa_tracker_track(C.c_void_p(self._h), n, n_nodes, _ptr(pts) if n else None, _ptr(ps) if n else None,
                                          _ptr(sc) if n else None, int(img_hw[0]), int(img_hw[1]), -1 if t is None else int(t),
                                          _ptr(idx), _ptr(trk), _ptr(tsc), C.byref(n_out)), "sa_tracker_track")
        m = n_out.value
        return {"index": idx[:m].copy(), "track": trk[:m].copy(), "tracking_score": tsc[:m].copy()}

    def track_frames(self, instance_peaks, instance_peak_vals=None, instance_scores=None, n_valid=None, img_hw=(1, 1),
                     t0: Optional[int] = None, images=None, frame_t=None, images_ready=None):
        """A run of consecutive frames in the predictor's output layout (F, I, N, 2) NaN padded. ->
        dict(track (F, I) int32 with -1 for empty / dropped slots, tracking_score (F, I), order (F, I)).
        Flow trackers need `images` (F, H, W, C) uint8 (numpy or CUDA tensor; `images_ready`: a CUDA event after which a device
        tensor is complete, default = the current stream's work so far); `frame_t` (F,) gives every frame its own time
        step (default: t0 + f, or inferred)."""
        pts = _f32(instance_peaks)
        F, I, N = pts.shape[0], pts.shape[1], pts.shape[2]
        if n_valid is None:
            n_valid = (~np.isnan(pts).all(axis=(2, 3))).sum(axis=1)
        nv = np.ascontiguousarray(n_valid, dtype=np.int32)
        ps, sc = _f32(instance_peak_vals), _f32(instance_scores)
        trk = np.full((F, I), -1, np.int32)
        tsc = np.full((F, I), np.nan, np.float64)
        order = np.full((F, I), -1, np.int32)
        if self.uses_flow or frame_t is not None:
            if self.uses_flow and images is None:
                raise ValueError("flow tracker: track_frames() needs the frames (images=...)")
            ft = None if frame_t is None else np.ascontiguousarray(frame_t, dtype=np.int32)
            dev, fh, fw, ch, st = None, 1, 1, 1, None
            if self.uses_flow:
                dev = _device_frames(images)
                assert dev.shape[0] == F
                fh, fw, ch = (int(v) for v in dev.shape[1:])
                st = self._stream_for(dev, images_ready)
            check(_lib.lib().sa_tracker_track_frames_images(
                C.c_void_p(self._h), F, I, N, _ptr(pts), _ptr(ps), _ptr(sc), _ptr(nv), int(img_hw[0]), int(img_hw[1]),
                -1 if t0 is None else int(t0), _ptr(ft), None if dev is None else C.c_void_p(dev.data_ptr()), fh, fw, ch, st,
                _ptr(trk), _ptr(tsc), _ptr(order)), "sa_tracker_track_frames_images")
            return {"track": trk, "tracking_score": tsc, "order": order}
        check(_lib.lib().sa_tracker_track_frames(C.c_void_p(self._h), F, I, N, _ptr(pts), _ptr(ps), _ptr(sc), _ptr(nv),
                                                 int(img_hw[0]), int(img_hw[1]), -1 if t0 is None else int(t0), _ptr(trk),
                                                 _ptr(tsc), _ptr(order)), "sa_tracker_track_frames")
        return {"track": trk, "tracking_score": tsc, "order": order}

    def final_pass(self, track: np.ndarray, order: Optional[np.ndarray] = None) -> np.ndarray:
        """tracking.py:816-835 on the (F, I) track table of a whole video (modified in place and returned)."""
        if (self.target_instance_count or self.max_tracks) and self.post_connect_single_breaks:
            if not self.target_instance_count:
                self.target_instance_count = int(self.max_tracks)
            connect_single_track_breaks(track, self.target_instance_count, order)
        return track


def connect_single_track_breaks(track: np.ndarray, instance_count: int, order: Optional[np.ndarray] = None) -> np.ndarray:
    """components.py:419-466 on a (F, I) int32 track table (-1 = no instance), in place."""
    assert track.dtype == np.int32 and track.flags.c_contiguous and track.ndim == 2
    od = None if order is None else np.ascontiguousarray(order, dtype=np.int32)
    check(_lib.lib().sa_connect_single_track_breaks(track.shape[0], track.shape[1], _ptr(od), _ptr(track), int(instance_count)),
          "sa_connect_single_track_breaks")
    return track


def select_instances(ex, max_instances: Optional[int] = None):
    """Which slots of a prediction dict the reference hands to `tracker.track`, and in which order -> list (one per frame) of
    int arrays of slot indices. Reference: inference.py:2641-2660 (top-down) / 3283-3304 (bottom-up): the first `n_valid`
    slots, minus instances whose points are ALL NaN (`if np.isnan(pts).all(): continue` -- they must never reach the
    tracker: they would spawn or steal a track and poison the similarity quantiles with NaN), and -- bottom-up with
    `max_instances` -- `sorted(key=score, reverse=True)[:max_instances]` (stable), which also becomes the tracker's input
    order."""
    pts = np.asarray(ex["instance_peaks"])
    F, I = pts.shape[0], pts.shape[1]
    n_valid = np.asarray(ex["n_valid"]) if ex.get("n_valid") is not None else np.full((F,), I)
    ok = (np.arange(I)[None, :] < n_valid[:, None]) & ~np.isnan(pts).all(axis=(2, 3))
    scores = ex.get("instance_scores")
    sel = []
    for f in range(F):
        idx = np.nonzero(ok[f])[0]
        if max_instances is not None and scores is not None:
            idx = idx[np.argsort(-np.asarray(scores[f], np.float64)[idx], kind="stable")][:max_instances]
        sel.append(idx)
    return sel


def frames_of(ex, data=None):
    """The frames of one prediction dict for a flow tracker: the carried `image` (single-GPU bottom-up runs), else re-read from
    the source the predictor was given (array, `Video`, `VideoReader`) by `frame_ind`."""
    if ex.get("image_dev") is not None:  # the batch as the predictor uploaded it (single-GPU runs): nothing to copy
        return ex["image_dev"]
    if ex.get("image") is not None:
        return ex["image"]
    idx = [int(i) for i in np.asarray(ex["frame_ind"])]
    video = getattr(data, "video", data)  # VideoReader -> its Video
    if isinstance(video, np.ndarray):
        return video[idx]
    if hasattr(video, "get_frames"):
        return video.get_frames(idx)
    raise ValueError("flow tracker: the frames of this batch are not available (pass the video / array the predictions came from)")


def track_example(tracker: Tracker, ex, img_hw=(1, 1), max_instances: Optional[int] = None, use_frame_ind: bool = True,
                  images=None, images_ready=None):
    """Run `tracker` over the frames of one prediction dict and add `track_inds`, `tracking_scores`, `track_order` (all
    (F, I), -1 / NaN / -1 for slots that were not tracked). `images` (F, H, W, C): the frames, for flow trackers.
    Only the instances `select_instances` picks are tracked (compacted,
    with a source-index map back to the slots); `t` follows `frame_ind` frame by frame (non-contiguous readers), as
    inference.py:2662-2668 passes `t=frame_ind`."""
    pts = _f32(ex["instance_peaks"])
    F, I, N = pts.shape[0], pts.shape[1], pts.shape[2]
    sel = select_instances(ex, max_instances)
    m = max([len(s) for s in sel] + [1])
    cp = np.full((F, m, N, 2), np.nan, np.float32)
    vals, scores = ex.get("instance_peak_vals"), ex.get("instance_scores")
    cv = None if vals is None else np.full((F, m, N), np.nan, np.float32)
    cs = None if scores is None else np.full((F, m), np.nan, np.float32)
    nv = np.zeros((F,), np.int32)
    for f, idx in enumerate(sel):
        k = len(idx)
        nv[f] = k
        cp[f, :k] = pts[f, idx]
        if cv is not None:
            cv[f, :k] = np.asarray(vals)[f, idx]
        if cs is not None:
            cs[f, :k] = np.asarray(scores)[f, idx]
    trk = np.full((F, I), -1, np.int32)
    tsc = np.full((F, I), np.nan, np.float64)
    order = np.full((F, I), -1, np.int32)
    fi = np.asarray(ex["frame_ind"], np.int64) if (use_frame_ind and "frame_ind" in ex) else None
    # runs of consecutive frame indices share one native call (t = t0 + f inside it)
    a = 0
    while a < F:
        b = a + 1
        while fi is not None and b < F and fi[b] == fi[b - 1] + 1:
            b += 1
        if fi is None:
            b = F
        r = tracker.track_frames(cp[a:b], None if cv is None else cv[a:b], None if cs is None else cs[a:b], nv[a:b],
                                 img_hw=img_hw, t0=None if fi is None else int(fi[a]),
                                 images=None if images is None else images[a:b],
                                 **({} if images_ready is None else {"images_ready": images_ready}))
        for f in range(a, b):
            k = nv[f]
            trk[f, sel[f]] = r["track"][f - a, :k]
            tsc[f, sel[f]] = r["tracking_score"][f - a, :k]
            order[f, sel[f]] = r["order"][f - a, :k]
        a = b
    ex["track_inds"], ex["tracking_scores"], ex["track_order"] = trk, tsc, order
    return ex


def image_hw_of(ex, default=(1, 1)):
    """Frame height / width for the size-normalised similarities: the carried raw size (`image_hw`, present in every
    predictor output) or the image itself."""
    if "image_hw" in ex:
        return int(ex["image_hw"][0]), int(ex["image_hw"][1])
    if "image" in ex:
        return tuple(ex["image"].shape[1:3])
    return default


def finish_tracks(outs, tracker: Tracker):
    """`Tracker.final_pass` (tracking.py:816-835) over the concatenated (F, I) track table of all batches."""
    imax = max(ex["track_inds"].shape[1] for ex in outs)

    def cat(key):
        return np.ascontiguousarray(np.concatenate(
            [np.pad(ex[key], ((0, 0), (0, imax - ex[key].shape[1])), constant_values=-1) for ex in outs]).astype(np.int32))

    table = tracker.final_pass(cat("track_inds"), cat("track_order"))
    o = 0
    for ex in outs:
        b, i = ex["track_inds"].shape
        ex["track_inds"] = table[o:o + b, :i].copy()
        o += b
    return outs


def run_tracker(outs, tracker: Tracker, img_hw=(1, 1), data=None):
    """tracking.py:1542-1581 on per-batch prediction dicts: every frame goes through `tracker.track` in order with the time
    step inferred (the reference does not pass `t` here), existing tracks are discarded; then `final_pass`.
    Adds / replaces `track_inds`, `tracking_scores`, `track_order` in every dict and returns the list."""
    outs = list(outs)
    if not outs:
        return outs
    for ex in outs:
        track_example(tracker, ex, img_hw=image_hw_of(ex, img_hw), use_frame_ind=False,
                      images=frames_of(ex, data) if tracker.uses_image else None)
    return finish_tracks(outs, tracker)


def retrack(slp_in: str, slp_out: str, **tracker_kwargs):
    """`sleap-track --tracking.tracker ... predictions.slp` (inference.py:5712-5733): read a prediction file, run the tracker
    over its predicted instances, write the result. Returns the tables written."""
    import json

    from ..io import slp

    src = slp.read_slp(slp_in)
    meta = json.loads(str(src["json"]))
    names = [n["name"] for n in meta["nodes"]]
    sk = meta["skeletons"][0]
    order = [n["id"] for n in sk["nodes"]]  # skeleton node order -> index into the global node list
    part_names = [names[i] for i in order]
    pos = {g: k for k, g in enumerate(order)}
    edges = [(pos[l["source"]], pos[l["target"]]) for l in sorted(sk["links"], key=lambda l: l["edge_insert_idx"])]
    pred = dict(src)
    keep = src["instances"]["instance_type"] == 1  # predicted instances only (user labels are not re-tracked here)
    if not keep.all():
        raise NotImplementedError("retrack(): files that mix user-labelled and predicted instances")
    outs = slp.tables_to_arrays(pred, len(part_names))
    for ex in outs:
        del ex["track_inds"], ex["tracking_scores"]
    tracker = Tracker.make_tracker_by_name(**tracker_kwargs)
    outs = run_tracker(outs, tracker)
    video = json.loads(str(src["videos_json"][0]))["backend"] if len(src["videos_json"]) else None
    return slp.write_slp(slp_out, outs, part_names, edges, video=video, track_names=tracker.spawned_tracks,
                         provenance=meta.get("provenance") or None)
