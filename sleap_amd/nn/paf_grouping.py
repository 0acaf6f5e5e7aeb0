"""PAF-based instance grouping with the reference's surface (`sleap/nn/paf_grouping.py`), executed by
the HIP kernels in csrc/postproc.hip.

Ragged tensors of the reference (`tf.RaggedTensor`) are represented as Python lists with one array
per sample; the hot path (`PAFScorer.predict_padded`) never leaves the device and works on
fixed-shape, count-prefixed buffers instead.
"""
from collections import deque, namedtuple
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import _lib, ops

PeakID = namedtuple("PeakID", ["node_ind", "peak_ind"])  # paf_grouping.py:33-46
EdgeType = namedtuple("EdgeType", ["src_node_ind", "dst_node_ind"])  # paf_grouping.py:49-63
EdgeConnection = namedtuple("EdgeConnection", ["src_peak_ind", "dst_peak_ind", "score"])  # :66-79


class GroupingOverflowError(RuntimeError):
    """A fixed-shape buffer (peaks per node / instances per frame) was too small for a frame."""


def toposort_edges(edge_types: Sequence[Tuple[int, int]]) -> Tuple[int, ...]:
    """paf_grouping.py:1293-1315 without networkx.

    `nx.DiGraph(edges)` inserts nodes in first-appearance order; `next(nx.topological_sort(dg))` is
    the first such node with in-degree 0; `nx.bfs_edges(dg, root)` walks successors in edge-insertion
    order. Edges not reachable from that root are dropped, as in the reference.
    """
    edges = [(int(e[0]), int(e[1])) for e in edge_types]
    nodes, succ, indeg = [], {}, {}
    for u, v in edges:
        for n in (u, v):
            if n not in succ:
                succ[n] = []
                indeg[n] = 0
                nodes.append(n)
        if v not in succ[u]:
            succ[u].append(v)
            indeg[v] += 1
    if not nodes:
        return tuple()
    roots = [n for n in nodes if indeg[n] == 0]
    if not roots:
        raise ValueError("Graph contains a cycle or graph changed during iteration")
    root = roots[0]
    out, seen, q = [], {root}, deque([root])
    while q:
        u = q.popleft()
        for v in succ[u]:
            if v not in seen:
                seen.add(v)
                out.append(edges.index((u, v)))
                q.append(v)
    return tuple(out)


def _status_check(status, what):
    st = 0
    for v in status.tolist():  # one small D2H copy; only the reference-shaped (non hot path) API calls this
        st |= int(v)
    if st & _lib.STATUS_LSA_INFEASIBLE:
        # scipy.optimize.linear_sum_assignment raises ValueError("cost matrix is infeasible")
        raise ValueError("cost matrix is infeasible")
    if st & (_lib.STATUS_NODE_PEAK_OVERFLOW | _lib.STATUS_INSTANCE_OVERFLOW | _lib.STATUS_PEAK_OVERFLOW):
        raise GroupingOverflowError(f"{what}: fixed-shape buffer overflow (status bits {st}); raise the caps")
    return st


class PAFScorer:
    """Scoring pipeline based on part affinity fields (paf_grouping.py:1318-1705).

    Attributes (same names, defaults and meaning as the reference's attrs class):
        part_names, edges, pafs_stride, max_edge_length_ratio=0.25, dist_penalty_weight=1.0,
        n_points=10, min_instance_peaks=0, min_line_scores=0.25
    Derived: edge_inds, edge_types, n_nodes, n_edges, sorted_edge_inds.
    Device buffer caps (new): max_node_peaks (peaks of one node type per frame), max_instances.
    """

    def __init__(self, part_names: List[str], edges: List[Tuple[str, str]], pafs_stride: int,
                 max_edge_length_ratio: float = 0.25, dist_penalty_weight: float = 1.0, n_points: int = 10,
                 min_instance_peaks=0, min_line_scores: float = 0.25, max_node_peaks: int = 32,
                 max_instances: int = 32, strict_oob: bool = False):
        self.part_names = list(part_names)
        self.edges = [tuple(e) for e in edges]
        self.pafs_stride = pafs_stride
        self.max_edge_length_ratio = max_edge_length_ratio
        self.dist_penalty_weight = dist_penalty_weight
        self.n_points = n_points
        self.min_instance_peaks = min_instance_peaks
        self.min_line_scores = min_line_scores
        self.max_node_peaks = max_node_peaks
        self.max_instances = max_instances
        self.strict_oob = strict_oob
        # __attrs_post_init__ (paf_grouping.py:1392-1404)
        self.edge_inds = [(self.part_names.index(s), self.part_names.index(d)) for s, d in self.edges]
        self.edge_types = [EdgeType(s, d) for s, d in self.edge_inds]
        self.n_nodes = len(self.part_names)
        self.n_edges = len(self.edges)
        self.sorted_edge_inds = toposort_edges(self.edge_types)
        self._dev_cache = {}

    @classmethod
    def from_config(cls, config, max_edge_length_ratio: float = 0.25, dist_penalty_weight: float = 1.0,
                    n_points: int = 10, min_instance_peaks=0, min_line_scores: float = 0.25, **kw):
        """paf_grouping.py:1406-1440. `config` is the `model.heads.multi_instance` dict (or an object
        with `.confmaps.part_names`, `.pafs.edges`, `.pafs.output_stride`)."""
        get = (lambda o, k: o[k]) if isinstance(config, dict) else getattr
        cm, pf = get(config, "confmaps"), get(config, "pafs")
        return cls(part_names=get(cm, "part_names"), edges=[tuple(e) for e in get(pf, "edges")],
                   pafs_stride=get(pf, "output_stride"), max_edge_length_ratio=max_edge_length_ratio,
                   dist_penalty_weight=dist_penalty_weight, n_points=n_points,
                   min_instance_peaks=min_instance_peaks, min_line_scores=min_line_scores, **kw)

    # ------------------------------------------------------------------ device constants
    def _consts(self, device):
        key = str(device)
        if key not in self._dev_cache:
            edges = torch.tensor(self.edge_inds, dtype=torch.int32, device=device).reshape(-1, 2).contiguous()
            sorted_e = torch.tensor(list(self.sorted_edge_inds), dtype=torch.int32, device=device)
            self._dev_cache[key] = (edges, sorted_e)
        return self._dev_cache[key]

    def _min_instance_peaks_int(self):
        m = self.min_instance_peaks
        if isinstance(m, float):
            return int(m * self.n_nodes) if m > 0 else 0  # paf_grouping.py:887-897
        return int(m)

    def max_edge_length(self, pafs_shape):
        """paf_grouping.py:469-473 -- max over (H, W, 2E) of the PAF tensor, float32 products."""
        return float(np.float32(self.max_edge_length_ratio) * np.float32(max(pafs_shape[1:]))
                     * np.float32(self.pafs_stride))

    # ------------------------------------------------------------------ hot path (padded, on device)
    def predict_padded(self, pafs, peak_xy, peak_val, peak_chan, peak_count, status=None, return_graph=False):
        """Device-resident equivalent of `predict`: all tensors fixed-shape, nothing synchronises.

        Returns (instance_peaks [B,I,N,2], instance_peak_vals [B,I,N], instance_scores [B,I], n_instances [B],
        status [B]) (+ (node_count, node_peaks, line_scores, match_dst, match_score) if return_graph).
        """
        dev = pafs.device
        B = pafs.shape[0]
        edges, sorted_e = self._consts(dev)
        if status is None:
            status = torch.zeros((B,), dtype=torch.int32, device=dev)
        node_count, node_peaks, line_scores = ops.paf_score(
            pafs, peak_xy, peak_chan, peak_count, edges, self.n_nodes, self.n_points, float(self.pafs_stride),
            self.max_edge_length(pafs.shape), self.dist_penalty_weight, self.max_node_peaks, status)
        match_dst, match_score = ops.paf_match(line_scores, node_count, edges, status)
        inst, vals, scores, n_inst = ops.paf_group(
            peak_xy, peak_val, node_count, node_peaks, match_dst, match_score, edges, sorted_e,
            self.min_line_scores, self._min_instance_peaks_int(), self.max_instances, status)
        out = (inst, vals, scores, n_inst, status)
        if return_graph:
            out = out + ((node_count, node_peaks, line_scores, match_dst, match_score),)
        return out

    def predict_from_maps(self, cms, offsets, pafs, peak_threshold, refinement, integral_patch_size, cm_output_stride, max_peaks,
                          status=None):
        """find_peaks + predict on the network's maps in two launches (sa_bottomup_postproc; the hot path of
        BottomUpInferenceLayer.call). -> dict with every stage's fixed-shape device tensors (ops.bottomup_postproc)."""
        edges, sorted_e = self._consts(pafs.device)
        return ops.bottomup_postproc(cms, offsets, pafs, peak_threshold, refinement, integral_patch_size, float(cm_output_stride),
                                     max_peaks, edges, sorted_e, self.n_nodes, self.n_points, float(self.pafs_stride),
                                     self.max_edge_length(pafs.shape), self.dist_penalty_weight, self.max_node_peaks,
                                     self.min_line_scores, self._min_instance_peaks_int(), self.max_instances, status)

    # ------------------------------------------------------------------ ragged helpers
    @staticmethod
    def _pad_peaks(peaks, peak_vals, peak_channel_inds, device):
        B = len(peaks)
        P = max([len(p) for p in peaks] + [1])
        xy = torch.zeros((B, P, 2), dtype=torch.float32)
        val = torch.zeros((B, P), dtype=torch.float32)
        ch = torch.zeros((B, P), dtype=torch.int32)
        cnt = torch.zeros((B,), dtype=torch.int32)
        for b in range(B):
            n = len(peaks[b])
            cnt[b] = n
            if n:
                xy[b, :n] = torch.as_tensor(np.asarray(_np(peaks[b]), np.float32).reshape(n, 2))
                if peak_vals is not None:
                    val[b, :n] = torch.as_tensor(np.asarray(_np(peak_vals[b]), np.float32).reshape(n))
                ch[b, :n] = torch.as_tensor(np.asarray(_np(peak_channel_inds[b]), np.int32).reshape(n))
        return xy.to(device), val.to(device), ch.to(device), cnt.to(device)

    def _graph_to_ragged(self, graph, B):
        """Dense (edge, src, dst) tables -> the reference's ragged (edge_inds, edge_peak_inds, line_scores)."""
        node_count, node_peaks, line_scores = (t.cpu().numpy() for t in graph[:3])
        e_out, p_out, s_out = [], [], []
        for b in range(B):
            ei, epi, ls = [], [], []
            for k, (sn, dn) in enumerate(self.edge_inds):
                ns, nd = node_count[b, sn], node_count[b, dn]
                for s in range(ns):
                    for d in range(nd):
                        ei.append(k)
                        epi.append((node_peaks[b, sn, s], node_peaks[b, dn, d]))
                        ls.append(line_scores[b, k, s, d])
            e_out.append(np.asarray(ei, np.int32))
            p_out.append(np.asarray(epi, np.int32).reshape(-1, 2))
            s_out.append(np.asarray(ls, np.float32))
        return e_out, p_out, s_out

    def _matches_to_ragged(self, graph, B):
        node_count = graph[0].cpu().numpy()
        match_dst, match_score = graph[3].cpu().numpy(), graph[4].cpu().numpy()
        me, ms, md, msc = [], [], [], []
        for b in range(B):
            a, s_, d_, sc = [], [], [], []
            for k, (sn, dn) in enumerate(self.edge_inds):
                for s in range(node_count[b, sn]):
                    if match_dst[b, k, s] >= 0:
                        a.append(k)
                        s_.append(s)
                        d_.append(match_dst[b, k, s])
                        sc.append(match_score[b, k, s])
            me.append(np.asarray(a, np.int32))
            ms.append(np.asarray(s_, np.int32))
            md.append(np.asarray(d_, np.int32))
            msc.append(np.asarray(sc, np.float32))
        return me, ms, md, msc

    # ------------------------------------------------------------------ reference-shaped API
    def score_paf_lines(self, pafs, peaks, peak_channel_inds):
        """paf_grouping.py:1453-1500 -> ragged (edge_inds, edge_peak_inds, line_scores) as lists."""
        pafs = ops.to_cuda_f32(pafs)
        xy, val, ch, cnt = self._pad_peaks(peaks, None, peak_channel_inds, pafs.device)
        edges, _ = self._consts(pafs.device)
        status = torch.zeros((pafs.shape[0],), dtype=torch.int32, device=pafs.device)
        g = ops.paf_score(pafs, xy, ch, cnt, edges, self.n_nodes, self.n_points, float(self.pafs_stride),
                          self.max_edge_length(pafs.shape), self.dist_penalty_weight, self.max_node_peaks, status)
        self._check(status)
        return self._graph_to_ragged(g, pafs.shape[0])

    def _check(self, status):
        st = _status_check(status, "PAFScorer")
        if self.strict_oob and (st & _lib.STATUS_PAF_OOB):
            raise IndexError("PAF line subscripts out of bounds (TensorFlow-CPU gather_nd raises here)")
        return st

    def match_candidates(self, edge_inds, edge_peak_inds, line_scores):
        """paf_grouping.py:1498-1550: `match_candidates_batch(edge_inds, edge_peak_inds, line_scores, self.n_edges)`."""
        return match_candidates_batch(edge_inds, edge_peak_inds, line_scores, self.n_edges)

    def group_instances(self, peaks, peak_vals, peak_channel_inds, match_edge_inds, match_src_peak_inds,
                        match_dst_peak_inds, match_line_scores):
        """paf_grouping.py:1552-1627: `group_instances_batch(...)` with this scorer's skeleton tables and thresholds."""
        return group_instances_batch(peaks, peak_vals, peak_channel_inds, match_edge_inds, match_src_peak_inds,
                                     match_dst_peak_inds, match_line_scores, self.n_nodes, self.sorted_edge_inds,
                                     self.edge_types, self.min_instance_peaks, min_line_scores=self.min_line_scores,
                                     max_instances=max(self.max_instances, 64))

    def predict(self, pafs, peaks, peak_vals, peak_channel_inds):
        """paf_grouping.py:1629-1705.

        Args: pafs (n_samples, H, W, 2*n_edges); peaks / peak_vals / peak_channel_inds: per-sample lists of
        (n_peaks, 2) / (n_peaks,) / (n_peaks,) arrays.
        Returns the reference's 6-tuple, each a per-sample list:
        (predicted_instances (n_inst, n_nodes, 2), predicted_peak_scores (n_inst, n_nodes),
         predicted_instance_scores (n_inst,), edge_inds, edge_peak_inds, line_scores).
        """
        pafs = ops.to_cuda_f32(pafs)
        B = pafs.shape[0]
        xy, val, ch, cnt = self._pad_peaks(peaks, peak_vals, peak_channel_inds, pafs.device)
        inst, vals, scores, n_inst, status, graph = self.predict_padded(pafs, xy, val, ch, cnt, return_graph=True)
        self._check(status)
        n = n_inst.cpu().numpy()
        inst, vals, scores = inst.cpu().numpy(), vals.cpu().numpy(), scores.cpu().numpy()
        ei, epi, ls = self._graph_to_ragged(graph, B)
        return ([inst[b, : n[b]] for b in range(B)], [vals[b, : n[b]] for b in range(B)],
                [scores[b, : n[b]] for b in range(B)], ei, epi, ls)


def _np(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)


# ----------------------------------------------------------------------------------------------------
# module-level functions with the reference's names (thin views over the same kernels)
# ----------------------------------------------------------------------------------------------------
def _scorer_for(skeleton_edges, n_nodes, n_line_points, pafs_stride, max_edge_length_ratio, dist_penalty_weight,
                **kw):
    names = [str(i) for i in range(n_nodes)]
    edges = [(str(int(s)), str(int(d))) for s, d in np.asarray(skeleton_edges).reshape(-1, 2)]
    return PAFScorer(names, edges, pafs_stride, max_edge_length_ratio, dist_penalty_weight, n_line_points, **kw)


def _dev_f32(x):
    return torch.as_tensor(np.ascontiguousarray(_np(x), dtype=np.float32)).cuda()


def _dev_i32(x):
    return torch.as_tensor(np.ascontiguousarray(_np(x), dtype=np.int32)).cuda()


def get_connection_candidates(peak_channel_inds_sample, skeleton_edges, n_nodes):
    """paf_grouping.py:82-142 -> (edge_inds (K,) int32, edge_peak_inds (K, 2) int32): for every skeleton edge, in order, all
    (source peak, destination peak) pairs, source-major; peaks of a node in their input order (stable argsort).

    The tables come from the scoring kernel's own bucketing pass (sa_paf_score phase 1: node_count / node_peaks), i.e. from
    the code the hot path runs; only their expansion into the reference's flat lists happens here."""
    ops.require_cuda()
    ch = np.ascontiguousarray(_np(peak_channel_inds_sample), dtype=np.int32).reshape(-1)
    edges_np = np.ascontiguousarray(_np(skeleton_edges), dtype=np.int32).reshape(-1, 2)
    n, E = len(ch), len(edges_np)
    if n == 0 or E == 0:
        return np.zeros((0,), np.int32), np.zeros((0, 2), np.int32)
    NP = max(int(np.bincount(ch, minlength=n_nodes).max()), 1)
    dev = torch.device("cuda")
    status = torch.zeros((1,), dtype=torch.int32, device=dev)
    node_count, node_peaks, _ = ops.paf_score(
        torch.zeros((1, 1, 1, 2 * E), dtype=torch.float32, device=dev), torch.zeros((1, n, 2), dtype=torch.float32, device=dev),
        torch.from_numpy(ch)[None].to(dev), torch.tensor([n], dtype=torch.int32, device=dev), torch.from_numpy(edges_np).to(dev),
        int(n_nodes), 1, 1.0, 1.0, 0.0, NP, status)
    node_count, node_peaks = node_count.cpu().numpy()[0], node_peaks.cpu().numpy()[0]
    ei, epi = [], []
    for k, (sn, dn) in enumerate(edges_np):
        src, dst = node_peaks[sn, : node_count[sn]], node_peaks[dn, : node_count[dn]]
        if len(src) and len(dst):
            s, d = np.meshgrid(src, dst, indexing="ij")
            epi.append(np.stack([s, d], axis=2).reshape(-1, 2))
            ei.append(np.full((len(src) * len(dst),), k, np.int32))
    if not ei:
        return np.zeros((0,), np.int32), np.zeros((0, 2), np.int32)
    return np.concatenate(ei).astype(np.int32), np.concatenate(epi).astype(np.int32)


def make_line_subs(peaks_sample, edge_peak_inds, edge_inds, n_line_points, pafs_stride):
    """paf_grouping.py:145-222 -> (n_candidates, n_line_points, 2, 3) int32 `[row, col, channel]` subscripts into the PAFs."""
    ops.require_cuda()
    epi = _dev_i32(edge_peak_inds).reshape(-1, 2)
    K = epi.shape[0]
    subs = torch.empty((K, int(n_line_points), 2, 3), dtype=torch.int32, device="cuda")
    if K:
        # (named locals: a temporary's device memory goes back to the allocator as soon as its pointer has been taken)
        peaks, einds = _dev_f32(peaks_sample).reshape(-1, 2), _dev_i32(edge_inds).reshape(-1)
        _lib.check(_lib.lib().sa_paf_line_subs(ops._ptr(peaks), ops._ptr(epi), ops._ptr(einds), K, int(n_line_points),
                                               float(pafs_stride), ops._ptr(subs), ops._stream()), "sa_paf_line_subs")
    return subs.cpu().numpy()


def get_paf_lines(pafs_sample, peaks_sample, edge_peak_inds, edge_inds, n_line_points, pafs_stride):
    """paf_grouping.py:225-275 -> (n_candidates, n_line_points, 2) float32 PAF vectors along every candidate line.
    TensorFlow-CPU raises for a subscript outside the tensor (TF-GPU yields 0): here it raises IndexError."""
    ops.require_cuda()
    pafs = _dev_f32(pafs_sample)
    H, W, Cc = pafs.shape
    subs = torch.as_tensor(make_line_subs(peaks_sample, edge_peak_inds, edge_inds, n_line_points, pafs_stride)).cuda()
    K = subs.shape[0]
    out = torch.zeros((K, int(n_line_points), 2), dtype=torch.float32, device="cuda")
    status = torch.zeros((1,), dtype=torch.int32, device="cuda")
    if K:
        _lib.check(_lib.lib().sa_gather_nd3(ops._ptr(pafs), H, W, Cc, ops._ptr(subs), K * int(n_line_points) * 2, ops._ptr(out),
                                            ops._ptr(status), ops._stream()), "sa_gather_nd3")
    if int(status.item()) & _lib.STATUS_PAF_OOB:
        raise IndexError("PAF line subscripts out of bounds (TensorFlow-CPU gather_nd raises here)")
    return out.cpu().numpy()


def compute_distance_penalty(spatial_vec_lengths, max_edge_length, dist_penalty_weight=1.0):
    """paf_grouping.py:278-322: `min(max_edge_length / length - 1, 0) * dist_penalty_weight`, same shape as the input."""
    ops.require_cuda()
    x = _dev_f32(spatial_vec_lengths)
    out = torch.empty_like(x)
    _lib.check(_lib.lib().sa_distance_penalty(ops._ptr(x), x.numel(), float(max_edge_length), float(dist_penalty_weight),
                                              ops._ptr(out), ops._stream()), "sa_distance_penalty")
    return out.cpu().numpy()


def score_paf_lines(paf_lines_sample, peaks_sample, edge_peak_inds_sample, max_edge_length, dist_penalty_weight=1.0):
    """paf_grouping.py:325-403 -> (n_candidates,) float32: mean projection of the line's PAF vectors on the unit vector of
    the connection + distance penalty."""
    ops.require_cuda()
    lines = _dev_f32(paf_lines_sample)
    K, n_pts = lines.shape[0], lines.shape[1]
    out = torch.empty((K,), dtype=torch.float32, device="cuda")
    if K:
        peaks, epi = _dev_f32(peaks_sample).reshape(-1, 2), _dev_i32(edge_peak_inds_sample).reshape(-1, 2)
        _lib.check(_lib.lib().sa_paf_line_scores(ops._ptr(lines), ops._ptr(peaks), ops._ptr(epi), K, n_pts,
                                                 float(max_edge_length), float(dist_penalty_weight), ops._ptr(out),
                                                 ops._stream()), "sa_paf_line_scores")
    return out.cpu().numpy()


def score_paf_lines_batch(pafs, peaks, peak_channel_inds, skeleton_edges, n_line_points, pafs_stride,
                          max_edge_length_ratio, dist_penalty_weight, n_nodes):
    """paf_grouping.py:406-550 (ragged in/out as per-sample lists)."""
    sc = _scorer_for(skeleton_edges, n_nodes, n_line_points, pafs_stride, max_edge_length_ratio, dist_penalty_weight)
    return sc.score_paf_lines(pafs, peaks, peak_channel_inds)


def _group_connections(connections, n_nodes, peaks=None, peak_scores=None, min_instance_peaks=0, assignments=None,
                       max_instances=None):
    """One frame through sa_paf_group_connections: `connections` {EdgeType: [EdgeConnection]} in dictionary order. ->
    (assign table (N, NP) int32, instances (I, N, 2), peak scores (I, N), instance scores (I,))."""
    ops.require_cuda()
    edge_types = list(connections.keys())
    E = len(edge_types)
    used_nodes = [n for et in edge_types for n in (int(et[0]), int(et[1]))]
    N = int(n_nodes) if n_nodes is not None else (max(used_nodes) + 1 if used_nodes else 1)
    if used_nodes and max(used_nodes) >= N:
        raise IndexError(f"edge type refers to node {max(used_nodes)} of {N}")
    conns = [(k, int(c[0]), int(c[1]), float(c[2])) for k, et in enumerate(edge_types) for c in connections[et]]
    NP = 1 + max([0] + [max(c[1], c[2]) for c in conns])
    if peaks is not None:
        NP = max([NP] + [len(p) for p in peaks])
    if assignments is not None:
        NP = max([NP] + [int(pid[1]) + 1 for pid in assignments])
    if NP > 512:
        raise GroupingOverflowError(f"{NP} peaks of one node type exceed the device tables (512)")
    I = int(max_instances) if max_instances is not None else max(N * NP, 1)  # the reference has no cap
    dev = torch.device("cuda")
    i32, f32 = torch.int32, torch.float32
    K = max(len(conns), 1)
    ce = torch.zeros((1, K), dtype=i32)
    cs, cd, csc = ce.clone(), ce.clone(), torch.zeros((1, K), dtype=f32)
    for q, (k, a, b_, sc) in enumerate(conns):
        ce[0, q], cs[0, q], cd[0, q], csc[0, q] = k, a, b_, sc
    edges = torch.tensor([[int(et[0]), int(et[1])] for et in edge_types] or [[0, 0]], dtype=i32).reshape(-1, 2)
    xy = torch.zeros((1, N * NP, 2), dtype=f32)
    val = torch.zeros((1, N * NP), dtype=f32)
    node_count = torch.full((1, N), NP, dtype=i32)
    if peaks is not None:
        for nd in range(N):
            n = len(peaks[nd]) if nd < len(peaks) else 0
            node_count[0, nd] = n
            if n:
                xy[0, nd * NP: nd * NP + n] = torch.as_tensor(np.asarray(_np(peaks[nd]), np.float32).reshape(n, 2))
                val[0, nd * NP: nd * NP + n] = torch.as_tensor(np.asarray(_np(peak_scores[nd]), np.float32).reshape(n))
        if assignments is None:
            node_count[:] = NP  # the walk may name any slot; unnamed slots stay unassigned
    node_peaks = torch.arange(N * NP, dtype=i32).reshape(1, N, NP)
    a_in = o_in = o_cnt = None
    if assignments is not None:
        a_in = torch.full((1, N * NP), -1, dtype=i32)
        o_in = torch.zeros((1, N * NP), dtype=i32)
        for j, (pid, inst) in enumerate(assignments.items()):
            a_in[0, int(pid[0]) * NP + int(pid[1])] = int(inst)
            o_in[0, j] = int(pid[0]) * NP + int(pid[1])
        o_cnt = torch.tensor([len(assignments)], dtype=i32)
        a_in, o_in, o_cnt = a_in.to(dev), o_in.to(dev), o_cnt.to(dev)
    mip = int(min_instance_peaks * N) if isinstance(min_instance_peaks, float) else int(min_instance_peaks)
    inst = torch.empty((1, I, N, 2), dtype=f32, device=dev)
    vals = torch.empty((1, I, N), dtype=f32, device=dev)
    scores = torch.empty((1, I), dtype=f32, device=dev)
    n_inst = torch.zeros((1,), dtype=i32, device=dev)
    assign = torch.empty((1, N * NP), dtype=i32, device=dev)
    status = torch.zeros((1,), dtype=i32, device=dev)
    h = _lib.lib()
    ws = torch.empty((h.sa_paf_workspace(1, max(E, 1), N, NP),), dtype=torch.uint8, device=dev)
    P = ops._ptr
    t = [x.to(dev) for x in (xy, val, node_count, node_peaks, ce, cs, cd, csc, edges)]
    cnt = torch.tensor([len(conns)], dtype=i32, device=dev)
    _lib.check(h.sa_paf_group_connections(P(t[0]), P(t[1]), P(t[2]), P(t[3]), N * NP, P(t[4]), P(t[5]), P(t[6]), P(t[7]), P(cnt),
                                          K, P(t[8]), 1, E, N, NP, float("-inf"), mip, I, P(inst), P(vals), P(scores), P(n_inst),
                                          P(assign), P(a_in), P(o_in), P(o_cnt), P(status), P(ws), ws.numel(), ops._stream()),
               "sa_paf_group_connections")
    _status_check(status, "group connections")
    n = int(n_inst.item())
    return (assign.cpu().numpy().reshape(N, NP), inst.cpu().numpy()[0, :n], vals.cpu().numpy()[0, :n], scores.cpu().numpy()[0, :n])


def assign_connections_to_instances(connections, min_instance_peaks=0, n_nodes=None):
    """paf_grouping.py:799-914: {EdgeType: [EdgeConnection]} (walked in dictionary order) -> {PeakID: instance id}. The greedy
    walk runs in the grouping kernel (csrc/postproc.hip: frame_group_wave); ids are the reference's raw ids (not re-indexed)."""
    if isinstance(min_instance_peaks, float) and n_nodes is None:  # :887-896: infer from the edge types
        n_nodes_thr = len({n for et in connections for n in (et[0], et[1])})
        min_instance_peaks = int(min_instance_peaks * n_nodes_thr)
    assign, _, _, _ = _group_connections(connections, n_nodes, min_instance_peaks=min_instance_peaks)
    out = {}
    for nd, p in zip(*np.nonzero(assign >= 0)):
        out[PeakID(int(nd), int(p))] = int(assign[nd, p])
    return out


def make_predicted_instances(peaks, peak_scores, connections, instance_assignments):
    """paf_grouping.py:917-981: node-grouped peaks / scores + connections + a {PeakID: instance id} dictionary ->
    (predicted_instances (n, n_nodes, 2), predicted_peak_scores (n, n_nodes), predicted_instance_scores (n,)). Like the
    reference it re-indexes the dictionary's ids in place (contiguous, ascending)."""
    _, inst, vals, scores = _group_connections(connections, len(peaks), peaks, peak_scores, assignments=instance_assignments)
    ids = sorted(set(instance_assignments.values()))
    lut = {v: i for i, v in enumerate(ids)}
    for k in instance_assignments:
        instance_assignments[k] = lut[instance_assignments[k]]
    return inst, vals, scores


def match_candidates_batch(edge_inds, edge_peak_inds, line_scores, n_edges):
    """paf_grouping.py:673-796. Candidates must be in the order `score_paf_lines_batch` emits them
    (edge-major, src-major), which is how the reference reshapes them to (n_src, n_dst) (:621-622)."""
    require = ops.require_cuda
    require()
    B = len(edge_inds)
    NPmax = 1
    per = []
    for b in range(B):
        ei = _np(edge_inds[b]).reshape(-1)
        epi = _np(edge_peak_inds[b]).reshape(-1, 2)
        ls = _np(line_scores[b]).reshape(-1)
        rows = []
        for k in range(n_edges):
            sel = np.nonzero(ei == k)[0]
            ns = len(dict.fromkeys(epi[sel, 0].tolist()))
            nd = len(dict.fromkeys(epi[sel, 1].tolist()))
            rows.append((ns, nd, ls[sel].reshape(ns, nd)))
            NPmax = max(NPmax, ns, nd)
        per.append(rows)
    # synthetic skeleton: edge k joins private node types 2k -> 2k+1
    N = 2 * n_edges
    scores = torch.full((B, n_edges, NPmax, NPmax), float("nan"), dtype=torch.float32)
    node_count = torch.zeros((B, max(N, 1)), dtype=torch.int32)
    for b in range(B):
        for k, (ns, nd, m) in enumerate(per[b]):
            node_count[b, 2 * k], node_count[b, 2 * k + 1] = ns, nd
            if ns and nd:
                scores[b, k, :ns, :nd] = torch.as_tensor(np.asarray(m, np.float32))
    edges = torch.tensor([[2 * k, 2 * k + 1] for k in range(n_edges)], dtype=torch.int32).reshape(-1, 2)
    status = torch.zeros((B,), dtype=torch.int32, device="cuda")
    md, msc = ops.paf_match(scores.cuda(), node_count.cuda(), edges.cuda(), status)
    _status_check(status, "match_candidates_batch")
    md, msc = md.cpu().numpy(), msc.cpu().numpy()
    out = ([], [], [], [])
    for b in range(B):
        a, s_, d_, sc = [], [], [], []
        for k, (ns, nd, _) in enumerate(per[b]):
            for s in range(ns):
                if md[b, k, s] >= 0:
                    a.append(k)
                    s_.append(s)
                    d_.append(md[b, k, s])
                    sc.append(msc[b, k, s])
        for lst, v, dt in zip(out, (a, s_, d_, sc), (np.int32, np.int32, np.int32, np.float32)):
            lst.append(np.asarray(v, dt))
    return out


def match_candidates_sample(edge_inds_sample, edge_peak_inds_sample, line_scores_sample, n_edges):
    """paf_grouping.py:553-670."""
    out = match_candidates_batch([edge_inds_sample], [edge_peak_inds_sample], [line_scores_sample], n_edges)
    return tuple(o[0] for o in out)


def group_instances_batch(peaks, peak_vals, peak_channel_inds, match_edge_inds, match_src_peak_inds,
                          match_dst_peak_inds, match_line_scores, n_nodes, sorted_edge_inds, edge_types,
                          min_instance_peaks, min_line_scores: float = 0.25, max_instances: int = 64):
    """paf_grouping.py:1115-1290 (ragged in/out as per-sample lists)."""
    ops.require_cuda()
    dev = torch.device("cuda")
    B = len(peaks)
    xy, val, ch, cnt = PAFScorer._pad_peaks(peaks, peak_vals, peak_channel_inds, dev)
    E = len(edge_types)
    chn = ch.cpu().numpy()
    cn = cnt.cpu().numpy()
    NP = 1
    for b in range(B):
        if cn[b]:
            NP = max(NP, int(np.bincount(chn[b, : cn[b]], minlength=n_nodes).max()))
    node_count = torch.zeros((B, n_nodes), dtype=torch.int32)
    node_peaks = torch.full((B, n_nodes, NP), -1, dtype=torch.int32)
    match_dst = torch.full((B, max(E, 1), NP), -1, dtype=torch.int32)
    match_score = torch.full((B, max(E, 1), NP), float("nan"), dtype=torch.float32)
    for b in range(B):
        for nd in range(n_nodes):
            idx = np.nonzero(chn[b, : cn[b]] == nd)[0]
            node_count[b, nd] = len(idx)
            node_peaks[b, nd, : len(idx)] = torch.as_tensor(idx.astype(np.int32))
        me, ms, md, msc = (_np(x[b]).reshape(-1) for x in
                           (match_edge_inds, match_src_peak_inds, match_dst_peak_inds, match_line_scores))
        for k, s, d, sc in zip(me, ms, md, msc):
            match_dst[b, int(k), int(s)] = int(d)
            match_score[b, int(k), int(s)] = float(sc)
    edges = torch.tensor([[int(e[0]), int(e[1])] for e in edge_types], dtype=torch.int32).reshape(-1, 2)
    sorted_e = torch.tensor([int(i) for i in _np(sorted_edge_inds).reshape(-1)], dtype=torch.int32)
    status = torch.zeros((B,), dtype=torch.int32, device=dev)
    mip = int(min_instance_peaks * n_nodes) if isinstance(min_instance_peaks, float) else int(min_instance_peaks)
    inst, vals, scores, n_inst = ops.paf_group(xy, val, node_count.to(dev), node_peaks.to(dev), match_dst.to(dev),
                                               match_score.to(dev), edges.to(dev), sorted_e.to(dev),
                                               min_line_scores, mip, max_instances, status)
    _status_check(status, "group_instances_batch")
    n = n_inst.cpu().numpy()
    inst, vals, scores = inst.cpu().numpy(), vals.cpu().numpy(), scores.cpu().numpy()
    return ([inst[b, : n[b]] for b in range(B)], [vals[b, : n[b]] for b in range(B)],
            [scores[b, : n[b]] for b in range(B)])


def group_instances_sample(peaks_sample, peak_scores_sample, peak_channel_inds_sample, match_edge_inds_sample,
                           match_src_peak_inds_sample, match_dst_peak_inds_sample, match_line_scores_sample,
                           n_nodes, sorted_edge_inds, edge_types, min_instance_peaks, min_line_scores: float = 0.25):
    """paf_grouping.py:984-1112."""
    out = group_instances_batch([peaks_sample], [peak_scores_sample], [peak_channel_inds_sample],
                                [match_edge_inds_sample], [match_src_peak_inds_sample],
                                [match_dst_peak_inds_sample], [match_line_scores_sample], n_nodes,
                                sorted_edge_inds, edge_types, min_instance_peaks, min_line_scores)
    return tuple(o[0] for o in out)
