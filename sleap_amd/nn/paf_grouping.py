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


def score_paf_lines_batch(pafs, peaks, peak_channel_inds, skeleton_edges, n_line_points, pafs_stride,
                          max_edge_length_ratio, dist_penalty_weight, n_nodes):
    """paf_grouping.py:406-550 (ragged in/out as per-sample lists)."""
    sc = _scorer_for(skeleton_edges, n_nodes, n_line_points, pafs_stride, max_edge_length_ratio, dist_penalty_weight)
    return sc.score_paf_lines(pafs, peaks, peak_channel_inds)


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
