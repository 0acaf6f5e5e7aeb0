"""NumPy float32 restatement of `sleap/nn/paf_grouping.py` (TEST INFRASTRUCTURE ONLY).

Per-sample structure of the reference is kept: candidates -> PAF line sampling -> line
scores -> Hungarian matching per edge (SciPy, as the reference does through
`sleap/nn/utils.py:79-98`) -> greedy instance assembly.
"""
from collections import namedtuple

import numpy as np
from scipy.optimize import linear_sum_assignment

F32 = np.float32

PeakID = namedtuple("PeakID", ["node_ind", "peak_ind"])  # paf_grouping.py:33-46
EdgeType = namedtuple("EdgeType", ["src_node_ind", "dst_node_ind"])  # :49-63


class EdgeConnection:  # paf_grouping.py:66-79
    __slots__ = ("src_peak_ind", "dst_peak_ind", "score")

    def __init__(self, src_peak_ind, dst_peak_ind, score):
        self.src_peak_ind = src_peak_ind
        self.dst_peak_ind = dst_peak_ind
        self.score = score

    def __repr__(self):
        return f"EdgeConnection({self.src_peak_ind}, {self.dst_peak_ind}, {self.score})"


def get_connection_candidates(peak_channel_inds_sample, skeleton_edges, n_nodes):
    """paf_grouping.py:82-142 -- stable argsort by channel, then src-major (ij meshgrid)."""
    ch = np.asarray(peak_channel_inds_sample, np.int32).reshape(-1)
    skeleton_edges = np.asarray(skeleton_edges, np.int32).reshape(-1, 2)
    peak_inds = np.argsort(ch, kind="stable")
    node_inds = ch[peak_inds]
    grouped = [peak_inds[node_inds == n] for n in range(n_nodes)]
    edge_inds, edge_peak_inds = [], []
    for k, (s, d) in enumerate(skeleton_edges):
        src, dst = grouped[s], grouped[d]
        ss, dd = np.meshgrid(src, dst, indexing="ij")
        sd = np.stack([ss, dd], axis=2).reshape(-1, 2)
        edge_inds.append(np.full((sd.shape[0],), k, np.int32))
        edge_peak_inds.append(sd.astype(np.int32))
    if len(edge_inds) == 0:
        return np.zeros((0,), np.int32), np.zeros((0, 2), np.int32)
    return np.concatenate(edge_inds), np.concatenate(edge_peak_inds).reshape(-1, 2)


def linspace_tf(start, stop, num):
    """`tf.linspace(start, stop, num, axis=-1)` in float32.

    TensorFlow computes `delta = (stop - start) / (num - 1)` and returns
    `[start, start + delta * i (i = 1..num-2), stop]` with exact end points.
    """
    start = np.asarray(start, F32)[..., None]
    stop = np.asarray(stop, F32)[..., None]
    n_steps = max(num - 1, 1)
    delta = (stop - start) / F32(n_steps)
    rng = np.arange(1, n_steps, dtype=np.int64).astype(F32)
    res = start + delta * rng
    return np.concatenate([start, res, stop], axis=-1)[..., :num].astype(F32)


def make_line_subs(peaks_sample, edge_peak_inds, edge_inds, n_line_points, pafs_stride):
    """paf_grouping.py:145-222 -- nearest-pixel subscripts, `tf.round` = half-to-even, NO clipping."""
    peaks_sample = np.asarray(peaks_sample, F32).reshape(-1, 2)
    edge_peak_inds = np.asarray(edge_peak_inds, np.int32).reshape(-1, 2)
    edge_inds = np.asarray(edge_inds, np.int32).reshape(-1)
    src = peaks_sample[edge_peak_inds[:, 0]]
    dst = peaks_sample[edge_peak_inds[:, 1]]
    n = src.shape[0]
    XY = linspace_tf(src, dst, n_line_points)  # (n, 2, n_points), dim 1 = [x, y]
    with np.errstate(invalid="ignore"):
        XY = np.rint(XY / F32(pafs_stride))
        XY = np.where(np.isfinite(XY), XY, -(2 ** 31)).astype(np.int64).astype(np.int32)
    XY = XY[:, [1, 0], :]  # [row, col]
    e = np.broadcast_to(edge_inds.reshape(-1, 1, 1), (n, 1, n_line_points))
    line_subs = np.concatenate([XY, e], axis=1).transpose(0, 2, 1)  # (n, n_points, 3)
    mul = np.array([1, 1, 2], np.int32).reshape(1, 1, 3)
    add = np.array([0, 0, 1], np.int32).reshape(1, 1, 3)
    return np.stack([line_subs * mul, line_subs * mul + add], axis=2).astype(np.int32)


def get_paf_lines(
    pafs_sample, peaks_sample, edge_peak_inds, edge_inds, n_line_points, pafs_stride, oob="raise"
):
    """paf_grouping.py:225-275 -- `tf.gather_nd(pafs_sample, line_subs)`.

    Out-of-bounds subscripts: TF-CPU raises InvalidArgumentError, TF-GPU returns 0
    (SURVEY.md §8a row a8). `oob="raise"` follows the CPU reference; `oob="zero"` is the
    GPU behaviour, which is what the HIP kernel implements.
    """
    pafs_sample = np.asarray(pafs_sample, F32)
    subs = make_line_subs(peaks_sample, edge_peak_inds, edge_inds, n_line_points, pafs_stride)
    H, W, C = pafs_sample.shape
    r, c, ch = subs[..., 0], subs[..., 1], subs[..., 2]
    ok = (r >= 0) & (r < H) & (c >= 0) & (c < W) & (ch >= 0) & (ch < C)
    if not ok.all():
        if oob == "raise":
            raise IndexError("PAF line subscripts out of bounds (TF-CPU gather_nd would raise)")
        rr, cc, hh = np.clip(r, 0, H - 1), np.clip(c, 0, W - 1), np.clip(ch, 0, C - 1)
        return np.where(ok, pafs_sample[rr, cc, hh], F32(0)).astype(F32)
    return pafs_sample[r, c, ch]


def compute_distance_penalty(spatial_vec_lengths, max_edge_length, dist_penalty_weight=1.0):
    """paf_grouping.py:278-322 -- `min(max_len / d - 1, 0) * weight`."""
    d = np.asarray(spatial_vec_lengths, F32)
    with np.errstate(divide="ignore", invalid="ignore"):
        return (
            np.minimum((F32(max_edge_length) / d) - F32(1), F32(0)) * F32(dist_penalty_weight)
        ).astype(F32)


def score_paf_lines(
    paf_lines_sample, peaks_sample, edge_peak_inds_sample, max_edge_length, dist_penalty_weight=1.0
):
    """paf_grouping.py:325-403 -- mean over line points of `paf . unit(dst - src)` + penalty."""
    paf_lines_sample = np.asarray(paf_lines_sample, F32)
    peaks_sample = np.asarray(peaks_sample, F32).reshape(-1, 2)
    epi = np.asarray(edge_peak_inds_sample, np.int32).reshape(-1, 2)
    src = peaks_sample[epi[:, 0]]
    dst = peaks_sample[epi[:, 1]]
    vec = dst - src
    with np.errstate(divide="ignore", invalid="ignore"):
        # tf.norm = sqrt(sum(x*x))
        length = np.sqrt((vec * vec).sum(axis=1, keepdims=True, dtype=F32)).astype(F32)
        vec = vec / length
        # (n, P, 2) @ (n, 2, 1): x*vx + y*vy
        line_scores = paf_lines_sample[:, :, 0] * vec[:, 0:1] + paf_lines_sample[:, :, 1] * vec[:, 1:2]
        pen = compute_distance_penalty(length, max_edge_length, dist_penalty_weight)[:, 0]
        if line_scores.shape[1] > 0:
            mean = line_scores.sum(axis=1, dtype=F32) / F32(line_scores.shape[1])
        else:
            mean = np.full((line_scores.shape[0],), np.nan, F32)
    return (mean + pen).astype(F32)


def score_paf_lines_batch(
    pafs,
    peaks,
    peak_channel_inds,
    skeleton_edges,
    n_line_points,
    pafs_stride,
    max_edge_length_ratio,
    dist_penalty_weight,
    n_nodes,
    oob="raise",
):
    """paf_grouping.py:406-550. `peaks` / `peak_channel_inds` are per-sample lists (ragged).

    `max_edge_length = ratio * max(pafs.shape[1:]) * stride` -- the max runs over
    (H, W, 2*E) exactly as `tf.reduce_max(tf.shape(pafs[0]))` does (:469-473).
    """
    pafs = np.asarray(pafs, F32)
    max_edge_length = F32(max_edge_length_ratio) * F32(max(pafs.shape[1:])) * F32(pafs_stride)
    edge_inds, edge_peak_inds, line_scores = [], [], []
    for s in range(pafs.shape[0]):
        ei, epi = get_connection_candidates(peak_channel_inds[s], skeleton_edges, n_nodes)
        lines = get_paf_lines(pafs[s], peaks[s], epi, ei, n_line_points, pafs_stride, oob=oob)
        sc = score_paf_lines(lines, peaks[s], epi, max_edge_length, dist_penalty_weight)
        edge_inds.append(ei)
        edge_peak_inds.append(epi)
        line_scores.append(sc)
    return edge_inds, edge_peak_inds, line_scores


def match_candidates_sample(edge_inds_sample, edge_peak_inds_sample, line_scores_sample, n_edges):
    """paf_grouping.py:553-670 -- per edge: reshape to (n_src, n_dst), cost = -score (NaN -> +inf),
    SciPy `linear_sum_assignment`; returned indices are WITHIN the node's peak list."""
    ei = np.asarray(edge_inds_sample, np.int32).reshape(-1)
    epi = np.asarray(edge_peak_inds_sample, np.int32).reshape(-1, 2)
    ls = np.asarray(line_scores_sample, F32).reshape(-1)
    me, ms, md, msc = [], [], [], []
    for k in range(n_edges):
        sel = np.nonzero(ei == k)[0]
        epk = epi[sel]
        lsk = ls[sel]
        n_src = len(dict.fromkeys(epk[:, 0].tolist()))
        n_dst = len(dict.fromkeys(epk[:, 1].tolist()))
        scores = lsk.reshape(n_src, n_dst)
        cost = np.where(np.isnan(scores), F32(np.inf), -scores)
        r, c = linear_sum_assignment(cost)
        me.append(np.full((len(r),), k, np.int32))
        ms.append(r.astype(np.int32))
        md.append(c.astype(np.int32))
        msc.append(scores[r, c].astype(F32))
    if n_edges == 0:
        z = np.zeros((0,), np.int32)
        return z, z, z, np.zeros((0,), F32)
    return np.concatenate(me), np.concatenate(ms), np.concatenate(md), np.concatenate(msc)


def match_candidates_batch(edge_inds, edge_peak_inds, line_scores, n_edges):
    """paf_grouping.py:673-796 (per-sample lists in, per-sample lists out)."""
    out = [
        match_candidates_sample(edge_inds[s], edge_peak_inds[s], line_scores[s], n_edges)
        for s in range(len(edge_inds))
    ]
    return tuple([o[i] for o in out] for i in range(4))


def assign_connections_to_instances(connections, min_instance_peaks=0, n_nodes=None):
    """paf_grouping.py:799-914 -- the 3-case greedy assembly, order-dependent."""
    instance_assignments = dict()
    for edge_type, edge_connections in connections.items():
        for connection in edge_connections:
            src_id = PeakID(edge_type.src_node_ind, connection.src_peak_ind)
            dst_id = PeakID(edge_type.dst_node_ind, connection.dst_peak_ind)
            src_instance = instance_assignments.get(src_id, None)
            dst_instance = instance_assignments.get(dst_id, None)
            if src_instance is None and dst_instance is None:
                new_instance = max(instance_assignments.values(), default=-1) + 1
                instance_assignments[src_id] = new_instance
                instance_assignments[dst_id] = new_instance
            elif src_instance is not None and dst_instance is None:
                instance_assignments[dst_id] = src_instance
            elif src_instance is not None and dst_instance is not None:
                instance_assignments[dst_id] = src_instance
                src_nodes = set(
                    p.node_ind for p, inst in instance_assignments.items() if inst == src_instance
                )
                dst_nodes = set(
                    p.node_ind for p, inst in instance_assignments.items() if inst == dst_instance
                )
                if len(src_nodes.intersection(dst_nodes)) == 0:
                    for p in instance_assignments:
                        if instance_assignments[p] == dst_instance:
                            instance_assignments[p] = src_instance
    if min_instance_peaks > 0:
        if isinstance(min_instance_peaks, float):
            if n_nodes is None:
                all_nodes = set()
                for et in connections:
                    all_nodes.add(et.src_node_ind)
                    all_nodes.add(et.dst_node_ind)
                n_nodes = len(all_nodes)
            min_instance_peaks = int(min_instance_peaks * n_nodes)
        ids, counts = np.unique(list(instance_assignments.values()), return_counts=True)
        counts = {i: c for i, c in zip(ids, counts)}
        instance_assignments = {
            p: i for p, i in instance_assignments.items() if counts[i] >= min_instance_peaks
        }
    return instance_assignments


def make_predicted_instances(peaks, peak_scores, connections, instance_assignments):
    """paf_grouping.py:917-981 -- contiguous re-indexing via np.unique, score = sum of edge scores."""
    instance_assignments = dict(instance_assignments)
    vals = list(instance_assignments.values())
    if len(vals) > 0:
        instance_ids, instance_inds = np.unique(vals, return_inverse=True)
    else:
        instance_ids, instance_inds = np.zeros((0,)), np.zeros((0,), np.int64)
    for p, ind in zip(list(instance_assignments.keys()), instance_inds):
        instance_assignments[p] = int(ind)
    n_instances = len(instance_ids)
    scores = np.full((n_instances,), 0.0, dtype=F32)
    for edge_type, edge_connections in connections.items():
        for ec in edge_connections:
            src = PeakID(edge_type.src_node_ind, ec.src_peak_ind)
            if src in instance_assignments:
                scores[instance_assignments[src]] += ec.score
    n_nodes = len(peaks)
    inst = np.full((n_instances, n_nodes, 2), np.nan, dtype=F32)
    pscores = np.full((n_instances, n_nodes), np.nan, dtype=F32)
    for p, ind in instance_assignments.items():
        inst[ind, p.node_ind, :] = peaks[p.node_ind][p.peak_ind]
        pscores[ind, p.node_ind] = peak_scores[p.node_ind][p.peak_ind]
    return inst, pscores, scores


def group_instances_sample(
    peaks_sample,
    peak_scores_sample,
    peak_channel_inds_sample,
    match_edge_inds_sample,
    match_src_peak_inds_sample,
    match_dst_peak_inds_sample,
    match_line_scores_sample,
    n_nodes,
    sorted_edge_inds,
    edge_types,
    min_instance_peaks,
    min_line_scores=0.25,
):
    """paf_grouping.py:984-1112."""
    peaks_sample = np.asarray(peaks_sample, F32).reshape(-1, 2)
    peak_scores_sample = np.asarray(peak_scores_sample, F32).reshape(-1)
    ch = np.asarray(peak_channel_inds_sample, np.int32).reshape(-1)
    me = np.asarray(match_edge_inds_sample, np.int32).reshape(-1)
    ms = np.asarray(match_src_peak_inds_sample, np.int32).reshape(-1)
    md = np.asarray(match_dst_peak_inds_sample, np.int32).reshape(-1)
    msc = np.asarray(match_line_scores_sample, F32).reshape(-1)
    with np.errstate(invalid="ignore"):
        valid = msc >= F32(min_line_scores)
    me, ms, md, msc = me[valid], ms[valid], md[valid], msc[valid]
    peaks = [peaks_sample[ch == i] for i in range(n_nodes)]
    pscores = [peak_scores_sample[ch == i] for i in range(n_nodes)]
    connections = {}
    for edge_ind in sorted_edge_inds:
        in_edge = me == edge_ind
        et = edge_types[edge_ind]
        if not isinstance(et, EdgeType):
            et = EdgeType(int(et[0]), int(et[1]))
        connections[et] = [
            EdgeConnection(int(s), int(d), sc) for s, d, sc in zip(ms[in_edge], md[in_edge], msc[in_edge])
        ]
    assignments = assign_connections_to_instances(
        connections, min_instance_peaks=min_instance_peaks, n_nodes=n_nodes
    )
    return make_predicted_instances(peaks, pscores, connections, assignments)


def group_instances_batch(
    peaks,
    peak_vals,
    peak_channel_inds,
    match_edge_inds,
    match_src_peak_inds,
    match_dst_peak_inds,
    match_line_scores,
    n_nodes,
    sorted_edge_inds,
    edge_types,
    min_instance_peaks,
    min_line_scores=0.25,
):
    """paf_grouping.py:1115-1290 (per-sample lists)."""
    out = [
        group_instances_sample(
            peaks[s],
            peak_vals[s],
            peak_channel_inds[s],
            match_edge_inds[s],
            match_src_peak_inds[s],
            match_dst_peak_inds[s],
            match_line_scores[s],
            n_nodes,
            sorted_edge_inds,
            edge_types,
            min_instance_peaks,
            min_line_scores,
        )
        for s in range(len(peaks))
    ]
    return tuple([o[i] for o in out] for i in range(3))


def toposort_edges(edge_types):
    """paf_grouping.py:1293-1315 -- networkx topological_sort root, then bfs_edges from it."""
    import networkx as nx

    edges = [(int(e[0]), int(e[1])) for e in edge_types]
    dg = nx.DiGraph(edges)
    root = next(nx.topological_sort(dg))
    return tuple(edges.index(e) for e in nx.bfs_edges(dg, root))


class PAFScorer:
    """paf_grouping.py:1318-1705 (scoring/matching/grouping orchestration)."""

    def __init__(
        self,
        part_names,
        edges,
        pafs_stride,
        max_edge_length_ratio=0.25,
        dist_penalty_weight=1.0,
        n_points=10,
        min_instance_peaks=0,
        min_line_scores=0.25,
        oob="raise",
    ):
        self.part_names = list(part_names)
        self.edges = [tuple(e) for e in edges]
        self.pafs_stride = pafs_stride
        self.max_edge_length_ratio = max_edge_length_ratio
        self.dist_penalty_weight = dist_penalty_weight
        self.n_points = n_points
        self.min_instance_peaks = min_instance_peaks
        self.min_line_scores = min_line_scores
        self.oob = oob
        self.edge_inds = [
            (self.part_names.index(s), self.part_names.index(d)) for s, d in self.edges
        ]
        self.edge_types = [EdgeType(s, d) for s, d in self.edge_inds]
        self.n_nodes = len(self.part_names)
        self.n_edges = len(self.edges)
        self.sorted_edge_inds = toposort_edges(self.edge_types)

    def predict(self, pafs, peaks, peak_vals, peak_channel_inds):
        """paf_grouping.py:1629-1705. Ragged inputs/outputs are per-sample lists."""
        ei, epi, ls = score_paf_lines_batch(
            pafs,
            peaks,
            peak_channel_inds,
            self.edge_inds,
            self.n_points,
            self.pafs_stride,
            self.max_edge_length_ratio,
            self.dist_penalty_weight,
            self.n_nodes,
            oob=self.oob,
        )
        me, ms, md, msc = match_candidates_batch(ei, epi, ls, self.n_edges)
        inst, pscores, iscores = group_instances_batch(
            peaks,
            peak_vals,
            peak_channel_inds,
            me,
            ms,
            md,
            msc,
            self.n_nodes,
            self.sorted_edge_inds,
            self.edge_types,
            self.min_instance_peaks,
            self.min_line_scores,
        )
        return inst, pscores, iscores, ei, epi, ls
