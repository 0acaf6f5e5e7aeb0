"""CPU restatement of SLEAP's cross-frame identity tracker (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

Follows, function by function:
    sleap/nn/tracker/components.py:33-196   instance / normalized / object-keypoint similarity, centroid distance, IoU
    sleap/nn/tracker/components.py:199-226  hungarian_matching, greedy_matching
    sleap/nn/tracker/components.py:229-311  nms_instances, nms_fast
    sleap/nn/tracker/components.py:366-466  cull_frame_instances, connect_single_track_breaks
    sleap/nn/tracker/components.py:469-640  Match, FrameMatches
    sleap/nn/tracking.py:442-507            SimpleCandidateMaker, SimpleMaxTracksCandidateMaker
    sleap/nn/tracking.py:108-440, 1194-1240 FlowCandidateMaker, FlowMaxTracksCandidateMaker; the optical flow itself is oracle/optical_flow.py
    sleap/nn/tracking.py:542-841            Tracker.track / spawn_for_untracked_instances / final_pass
    sleap/nn/utils.py:45-76                 compute_iou
    sleap/instance.py:866-901               Instance.centroid / bounding_box / n_visible_points

Instances are plain records of arrays (the reference's `PredictedInstance` is an attrs class over the same numbers;
coordinates are float64 there: `Point` stores x and y as f8). Tracks are integers: the index into `spawned_tracks`
(the reference names them f"track_{index}"). Out of scope: the Kalman tracker (pykalman). Pinned by the reference's known-answer tests in tests/test_oracle_tracking.py. One declared divergence:
`greedy_matching` sorts with a STABLE sort (the reference's `np.argsort` default is an unstable introsort, so the order of
exactly tied costs is unspecified there -- "ties parity unpinned").
"""
from collections import defaultdict, deque
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
from scipy.optimize import linear_sum_assignment


class Inst:
    """points (N, 2) float64 with NaN rows for missing nodes, point scores (N,), instance score, track id or None."""

    __slots__ = ("points", "scores", "score", "track", "tracking_score", "uid")

    def __init__(self, points, scores=None, score=0.0, track=None, tracking_score=0.0, uid=None):
        self.points = np.asarray(points, dtype=np.float64)
        self.scores = np.ones(len(self.points)) if scores is None else np.asarray(scores, dtype=np.float64)
        self.score = float(score)
        self.track = track
        self.tracking_score = tracking_score
        self.uid = uid

    @property
    def points_array(self):
        return self.points

    @property
    def n_visible_points(self):  # from_arrays skips nodes with any NaN coordinate (instance.py:1108-1113)
        return int((~np.isnan(self.points).any(axis=1)).sum())

    @property
    def centroid(self):  # instance.py:866-875
        with np.errstate(all="ignore"):
            import warnings

            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                return np.nanmedian(self.points, axis=0)

    @property
    def bounding_box(self):  # instance.py:877-886, [y1, x1, y2, x2]
        if np.isnan(self.points).all():
            return np.array([np.nan] * 4)
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return np.concatenate([np.nanmin(self.points, axis=0)[::-1], np.nanmax(self.points, axis=0)[::-1]])

    def evolve(self, **kw):
        o = Inst(self.points, self.scores, self.score, self.track, self.tracking_score, self.uid)
        for k, v in kw.items():
            setattr(o, k, v)
        return o


# ------------------------------------------------------------------------------------------------ similarities
def instance_similarity(ref, query) -> float:
    ref_visible = ~(np.isnan(ref.points_array).any(axis=1))
    dists = np.sum((query.points_array - ref.points_array) ** 2, axis=1)
    with np.errstate(all="ignore"):
        return np.nansum(np.exp(-dists)) / np.sum(ref_visible)


def normalized_instance_similarity(ref, query, img_hw) -> float:
    nf = np.array((img_hw[1], img_hw[0]))
    ref_visible = ~(np.isnan(ref.points_array).any(axis=1))
    dists = np.sum((query.points_array / nf - ref.points_array / nf) ** 2, axis=1)
    with np.errstate(all="ignore"):
        return np.nansum(np.exp(-dists)) / np.sum(ref_visible)


def factory_object_keypoint_similarity(keypoint_errors=None, score_weighting=False, normalization_keypoints="all"):
    keypoint_errors = 1 if keypoint_errors is None else keypoint_errors
    with np.errstate(divide="ignore"):
        kp_precision = 1 / (2 * np.array(keypoint_errors, dtype=np.float64) ** 2)

    def object_keypoint_similarity(ref, query) -> float:
        nonlocal kp_precision
        ref_points, query_points = ref.points_array, query.points_array
        ref_scores, query_scores = (ref.scores, query.scores) if score_weighting else (1, 1)
        if normalization_keypoints in ("ref", "union"):
            ref_visible = ~(np.isnan(ref_points).any(axis=1))
            if normalization_keypoints == "ref":
                max_n = np.sum(ref_visible)
            else:
                max_n = np.sum(np.logical_and(ref_visible, ~(np.isnan(query_points).any(axis=1))))
        else:
            max_n = len(ref_points)
        if max_n == 0:
            return 0
        if kp_precision.size > 1 and 2 * kp_precision.size != ref_points.size:
            n_points = ref_points.size // 2
            if kp_precision.size > n_points:
                kp_precision = kp_precision[:n_points]
            else:
                kp_precision = np.pad(kp_precision, (0, n_points - kp_precision.size), "edge")
        dists = np.sum((query_points - ref_points) ** 2, axis=1) * kp_precision
        return np.nansum(ref_scores * query_scores * np.exp(-dists)) / max_n

    return object_keypoint_similarity


def centroid_distance(ref, query) -> float:
    return -np.linalg.norm(ref.centroid - query.centroid)


def compute_iou(b1, b2) -> float:  # utils.py:45-76
    y1, x1, y2, x2 = b1
    v1, u1, v2, u2 = b2
    iy1, ix1, iy2, ix2 = max(y1, v1), max(x1, u1), min(y2, v2), min(x2, u2)
    inter = max(ix2 - ix1 + 1, 0) * max(iy2 - iy1 + 1, 0)
    a1 = (x2 - x1 + 1) * (y2 - y1 + 1)
    a2 = (u2 - u1 + 1) * (v2 - v1 + 1)
    return inter / (a1 + a2 - inter)


def instance_iou(ref, query) -> float:
    return compute_iou(ref.bounding_box, query.bounding_box)


# ------------------------------------------------------------------------------------------------ matching
def hungarian_matching(cost_matrix) -> List[Tuple[int, int]]:
    r, c = linear_sum_assignment(cost_matrix)
    return list(zip(r, c))


def greedy_matching(cost_matrix) -> List[Tuple[int, int]]:
    rows, cols = np.unravel_index(np.argsort(cost_matrix, axis=None, kind="stable"), cost_matrix.shape)
    unassigned = list(zip(rows, cols))
    out = []
    while unassigned:
        r, c = unassigned.pop(0)
        out.append((r, c))
        unassigned = [e for e in unassigned if e[0] != r and e[1] != c]
    return out


# ------------------------------------------------------------------------------------------------ culling
def nms_fast(boxes, scores, iou_threshold, target_count=None) -> List[int]:
    if len(boxes) == 0:
        return []
    if target_count and len(boxes) < target_count:
        return list(range(len(boxes)))
    boxes = np.asarray(boxes)
    if boxes.dtype.kind == "i":
        boxes = boxes.astype("float")
    picked, nms_idxs = [], []
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    area = (x2 - x1 + 1) * (y2 - y1 + 1)
    idxs = np.argsort(scores)
    while len(idxs) > 0:
        p = idxs[-1]
        picked.append(p)
        xx1 = np.maximum(x1[p], x1[idxs[:-1]])
        yy1 = np.maximum(y1[p], y1[idxs[:-1]])
        xx2 = np.minimum(x2[p], x2[idxs[:-1]])
        yy2 = np.minimum(y2[p], y2[idxs[:-1]])
        w = np.maximum(0, xx2 - xx1 + 1)
        h = np.maximum(0, yy2 - yy1 + 1)
        overlap = (w * h) / area[idxs[:-1]]
        sup = np.where(overlap > iou_threshold)[0]
        nms_idxs.extend(list(idxs[sup]))
        idxs = np.delete(idxs, sup)[:-1]
    if target_count and nms_idxs and len(picked) < target_count:
        nms_idxs.sort(key=lambda i: -scores[i])
        add_back = min(len(nms_idxs), len(picked) - target_count)  # NB negative: python slice [:negative] semantics
        picked.extend(nms_idxs[:add_back])
    return [int(i) for i in picked]


def nms_instances(instances, iou_threshold, target_count=None):
    boxes = np.array([i.bounding_box for i in instances])
    scores = np.array([i.score for i in instances])
    picks = nms_fast(boxes, scores, iou_threshold, target_count)
    keep = [inst for i, inst in enumerate(instances) if i in picks]
    remove = [inst for i, inst in enumerate(instances) if i not in picks]
    return keep, remove


def cull_frame_instances(instances_list, instance_count, iou_threshold=None):
    if not instances_list:
        return
    if len(instances_list) > instance_count:
        keep = instances_list
        if iou_threshold:
            keep, extra = nms_instances(keep, iou_threshold=iou_threshold, target_count=instance_count)
            for inst in extra:
                instances_list.remove(inst)
        if len(keep) > instance_count:
            extra = sorted(keep, key=lambda i: i.score)[:-instance_count]
            for inst in extra:
                instances_list.remove(inst)
    return instances_list


def connect_single_track_breaks(frames: List[List[Inst]], instance_count: int):
    """frames: per-frame lists of tracked instances (modified in place)."""
    if not frames:
        return frames
    fix = dict()
    last_good = {i.track for i in frames[0]}
    for insts in frames:
        frame_tracks = {i.track for i in insts}
        if frame_tracks.intersection(set(fix.keys())):
            for inst in insts:
                if inst.track in fix and fix[inst.track] not in frame_tracks:
                    inst.track = fix[inst.track]
                    frame_tracks = {i.track for i in insts}
        extra = frame_tracks - last_good
        missing = last_good - frame_tracks
        if len(extra) == 1 and len(missing) == 1:
            for inst in insts:
                if inst.track in extra:
                    old, new = inst.track, missing.pop()
                    fix[old] = new
                    inst.track = new
                    break
        else:
            if len(frame_tracks) == instance_count:
                last_good = frame_tracks
    return frames


# ------------------------------------------------------------------------------------------------ frame matching
class Match:
    def __init__(self, track, instance, score=None, is_first_choice=False):
        self.track, self.instance, self.score, self.is_first_choice = track, instance, score, is_first_choice


class FrameMatches:
    def __init__(self, matches, cost_matrix, unmatched_instances):
        self.matches, self.cost_matrix, self.unmatched_instances = matches, cost_matrix, unmatched_instances

    @property
    def has_only_first_choice_matches(self):
        return all(m.is_first_choice for m in self.matches)

    @classmethod
    def from_candidate_instances(cls, untracked_instances, candidate_instances, similarity_function, matching_function,
                                 robust_best_instance=1.0):
        cost = np.ndarray((0,))
        candidate_tracks = []
        if candidate_instances:
            by_track = defaultdict(list)
            for inst in candidate_instances:
                by_track[inst.track].append(inst)
            candidate_tracks = list(by_track.keys())
            sims = np.full((len(untracked_instances), len(candidate_tracks)), np.nan)
            for i, u in enumerate(untracked_instances):
                for j, tr in enumerate(candidate_tracks):
                    vals = [similarity_function(u, c) for c in by_track[tr]]
                    if 0 < robust_best_instance < 1:
                        best = np.quantile(vals, robust_best_instance)
                    else:
                        best = np.max(vals)
                    sims[i, j] = best
            cost = -sims
            cost[np.isnan(cost)] = np.inf
        return cls.from_cost_matrix(cost, untracked_instances, candidate_tracks, matching_function)

    @classmethod
    def from_cost_matrix(cls, cost_matrix, instances, tracks, matching_function):
        matches, matched_inds = [], []
        if instances and tracks:
            match_inds = matching_function(cost_matrix)
            best = cost_matrix.argmin(axis=1)
            for i, j in match_inds:
                matched_inds.append(i)
                matches.append(Match(instance=instances[i], track=tracks[j], score=-cost_matrix[i, j],
                                     is_first_choice=bool(best[i] == j)))
        unmatched = [u for i, u in enumerate(instances) if i not in matched_inds]
        return cls(matches, cost_matrix, unmatched)


# ------------------------------------------------------------------------------------------------ tracker
SIMILARITIES = {"instance": instance_similarity, "centroid": centroid_distance, "iou": instance_iou,
                "normalized_instance": normalized_instance_similarity, "object_keypoint": factory_object_keypoint_similarity}
MATCHERS = {"hungarian": hungarian_matching, "greedy": greedy_matching}


class Tracker:
    """tracking.py:542-841 with the `simple` / `simplemaxtracks` / `flow` / `flowmaxtracks` candidate makers
    (make_tracker_by_name, :844-992)."""

    def __init__(self, tracker="simple", similarity="instance", match="greedy", track_window=5, robust=1.0,
                 min_new_track_points=0, min_match_points=0, of_window_size=21, of_max_levels=3, save_shifted_instances=False,
                 img_scale=1.0, target_instance_count=0, pre_cull_to_target=False,
                 pre_cull_iou_threshold=None, post_connect_single_breaks=False, max_tracks=None, max_tracking=False,
                 oks_errors=None, oks_score_weighting=False, oks_normalization="all"):
        max_tracking = max_tracking if max_tracks else False
        if max_tracking and tracker in ("simple", "flow"):
            tracker += "maxtracks"
        if tracker not in ("simple", "simplemaxtracks", "flow", "flowmaxtracks"):
            raise ValueError(f"{tracker} is not a valid tracker.")
        self.uses_flow = tracker.startswith("flow")
        if tracker != "flow":  # :914-919: only "flow" is configured; "flowmaxtracks" keeps the class defaults
            of_window_size, of_max_levels, save_shifted_instances, img_scale = 21, 3, False, 1.0
        self.of_window_size, self.of_max_levels, self.img_scale = of_window_size, of_max_levels, img_scale
        self.save_shifted_instances = bool(save_shifted_instances)
        self._shifted = {}  # (ref_t, t) -> (shifted instances, frame t)   (FlowCandidateMaker.shifted_instances, :136-138)
        self._images = {}  # t -> frame (MatchedFrameInstance(s).img_t)
        if similarity not in SIMILARITIES:
            raise ValueError(f"{similarity} is not a valid tracker similarity function.")
        if match not in MATCHERS:
            raise ValueError(f"{match} is not a valid tracker matching function.")
        self.has_max_tracking = tracker.endswith("maxtracks")
        self.min_match_points = min_match_points
        if similarity == "object_keypoint":
            self.similarity_function = factory_object_keypoint_similarity(oks_errors, oks_score_weighting, oks_normalization)
        else:
            self.similarity_function = SIMILARITIES[similarity]
        self.normalized = similarity == "normalized_instance"
        self.matching_function = MATCHERS[match]
        self.track_window = track_window
        self.robust_best_instance = robust
        self.min_new_track_points = min_new_track_points
        self.max_tracks, self.max_tracking = max_tracks, max_tracking
        self.target_instance_count = target_instance_count
        self.post_connect_single_breaks = post_connect_single_breaks
        self.pre_cull = None
        if target_instance_count and pre_cull_to_target:
            self.pre_cull = lambda lst: cull_frame_instances(lst, target_instance_count, pre_cull_iou_threshold)
        self.track_matching_queue = deque(maxlen=track_window)
        self.track_matching_queue_dict: Dict[int, deque] = dict()
        self.spawned_tracks: List[int] = []
        self.last_matches = None

    def _shift(self, ref_t, ref_instances, img, ref_img=None):
        """FlowCandidateMaker.get_shifted_instances -> flow_shift_instances (:180-208, 258-356)"""
        from .optical_flow import flow_shift_points

        out = []
        for i, pts, _score in flow_shift_points([r.points for r in ref_instances], self._images[ref_t] if ref_img is None else ref_img, img,
                                                min_shifted_points=self.min_match_points, scale=self.img_scale,
                                                window_size=self.of_window_size, max_levels=self.of_max_levels):
            out.append(ref_instances[i].evolve(points=pts.astype(np.float64)))  # ShiftedInstance.from_instance: the reference's track
        return out

    def _flow_candidates(self, img, t=None):
        out = []
        if self.has_max_tracking:  # FlowMaxTracksCandidateMaker.get_candidates (:1194-1240)
            tracks = []
            for track, matched in self.track_matching_queue_dict.items():
                if not self.max_tracking or len(tracks) < self.max_tracks:
                    tracks.append(track)
                    for (ref_t, _inst) in matched:
                        refs = [inst for items in self.track_matching_queue_dict.values() for (tt, inst) in items if tt == ref_t]
                        if refs:
                            out.extend(self._shift(ref_t, refs, img))
        else:  # FlowCandidateMaker.get_candidates (:210-237)
            if self.save_shifted_instances:  # prune_shifted_instances (:239-256)
                self._shifted = {k: v for k, v in self._shifted.items() if not (t - k[0] > self.track_window)}
            for (ref_t, insts) in self.track_matching_queue:
                ref_img = None
                if self.save_shifted_instances:  # get_shifted_instances_from_earlier_time (:146-166)
                    for ti in reversed(range(ref_t, t)):
                        if (ref_t, ti) in self._shifted and len(self._shifted[(ref_t, ti)][0]) > 0:
                            insts, ref_img = self._shifted[(ref_t, ti)]
                            break
                if len(insts) > 0:
                    shifted = self._shift(ref_t, insts, img, ref_img)
                    if self.save_shifted_instances:
                        self._shifted[(ref_t, t)] = (shifted, img)
                    out.extend(shifted)
        return out

    def _candidates(self):
        out = []
        if self.has_max_tracking:
            tracks = []
            for track, matched in self.track_matching_queue_dict.items():
                if not self.max_tracking or len(tracks) < self.max_tracks:
                    tracks.append(track)
                    for (_t, inst) in matched:
                        if inst.n_visible_points >= self.min_match_points:
                            out.append(inst)
        else:
            for (_t, insts) in self.track_matching_queue:
                for inst in insts:
                    if inst.n_visible_points >= self.min_match_points:
                        out.append(inst)
        return out

    def track(self, untracked_instances: List[Inst], img_hw=(1, 1), t: Optional[int] = None, img=None) -> List[Inst]:
        sim = self.similarity_function
        if self.normalized:
            sim = lambda a, b: normalized_instance_similarity(a, b, img_hw=img_hw)  # noqa: E731
        if t is None:
            if self.has_max_tracking:
                if len(self.track_matching_queue_dict) > 0:
                    tr = max(self.track_matching_queue_dict, key=lambda k: len(self.track_matching_queue_dict[k]))
                    t = self.track_matching_queue_dict[tr][-1][0] + 1
                else:
                    t = 0
            else:
                t = self.track_matching_queue[-1][0] + 1 if len(self.track_matching_queue) > 0 else 0
        tracked = []
        if untracked_instances:
            if self.pre_cull:
                self.pre_cull(untracked_instances)
            cands = self._flow_candidates(img, t) if self.uses_flow else self._candidates()
            fm = FrameMatches.from_candidate_instances(untracked_instances, cands, sim, self.matching_function,
                                                       self.robust_best_instance)
            self.last_matches = fm
            tracked.extend(m.instance.evolve(track=m.track, tracking_score=m.score) for m in fm.matches)
            tracked.extend(self._spawn(fm.unmatched_instances, t))
        if self.has_max_tracking:
            for inst in tracked:
                if inst.track in self.track_matching_queue_dict:
                    self.track_matching_queue_dict[inst.track].append((t, inst))
                elif not self.max_tracking or len(self.track_matching_queue_dict) < self.max_tracks:
                    self.track_matching_queue_dict[inst.track] = deque(maxlen=self.track_window)
                    self.track_matching_queue_dict[inst.track].append((t, inst))
        else:
            self.track_matching_queue.append((t, tracked))
        if self.uses_flow:
            self._images[t] = img
            live = ({tt for items in self.track_matching_queue_dict.values() for (tt, _i) in items} if self.has_max_tracking
                    else {tt for (tt, _l) in self.track_matching_queue})
            self._images = {k: v for k, v in self._images.items() if k in live}
        return tracked

    def _spawn(self, unmatched, t):
        out = []
        for inst in unmatched:
            if inst.n_visible_points < self.min_new_track_points:
                continue
            if self.has_max_tracking and self.max_tracking and len(self.track_matching_queue_dict) >= self.max_tracks:
                break
            new_track = len(self.spawned_tracks)
            self.spawned_tracks.append(new_track)
            out.append(inst.evolve(track=new_track))
        return out

    def final_pass(self, frames: List[List[Inst]]):
        if (self.target_instance_count or self.max_tracks) and self.post_connect_single_breaks:
            if not self.target_instance_count:
                self.target_instance_count = self.max_tracks
            connect_single_track_breaks(frames, self.target_instance_count)
