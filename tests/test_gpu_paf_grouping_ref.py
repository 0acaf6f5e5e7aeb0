"""The reference's own tests/nn/test_paf_grouping.py, pointed at `sleap_amd.nn.paf_grouping` one for one (same inputs, same
expected arrays; tf.Tensor -> NumPy, tf.RaggedTensor -> per-sample lists). Every function runs on the HIP kernels. Then the
fused per-frame post-processing kernel (sa_bottomup_postproc) against the separate stages and the wave-cooperative matcher
against SciPy on the device."""
import numpy as np
import pytest
import torch
from numpy.testing import assert_allclose, assert_array_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pg():
    from sleap_amd.nn import paf_grouping

    return paf_grouping


def test_get_connection_candidates(pg):  # ref :28-42
    edge_inds, edge_peak_inds = pg.get_connection_candidates([0, 0, 0, 1, 1, 2], [[0, 1], [1, 2], [2, 3]], 4)
    assert_array_equal(edge_inds, [0, 0, 0, 0, 0, 0, 1, 1])
    assert_array_equal(edge_peak_inds, [[0, 3], [0, 4], [1, 3], [1, 4], [2, 3], [2, 4], [3, 5], [4, 5]])
    # unsorted channel indices: peaks of a node keep their input order (stable argsort, :105)
    ei, epi = pg.get_connection_candidates([1, 0, 1, 0], [[0, 1]], 2)
    assert_array_equal(epi, [[1, 0], [1, 2], [3, 0], [3, 2]])


def test_make_line_subs(pg):  # ref :45-57
    subs = pg.make_line_subs(np.array([[0, 0], [4, 8]], np.float32), np.array([[0, 1]], np.int32), np.array([0], np.int32),
                             n_line_points=3, pafs_stride=2)
    assert_array_equal(subs, [[[[0, 0, 0], [0, 0, 1]], [[2, 1, 0], [2, 1, 1]], [[4, 2, 0], [4, 2, 1]]]])
    # tf.round is round-half-to-even: x = 1, 3, 5 at stride 2 -> 0.5, 1.5, 2.5 -> 0, 2, 2
    subs = pg.make_line_subs(np.array([[1, 0], [5, 0]], np.float32), np.array([[0, 1]], np.int32), np.array([2], np.int32), 3, 2)
    assert_array_equal(subs[0, :, 0], [[0, 0, 4], [0, 2, 4], [0, 2, 4]])


def _lines(pg):
    pafs_sample = np.arange(6 * 4 * 2, dtype=np.float32).reshape(6, 4, 2)
    peaks_sample = np.array([[0, 0], [4, 8]], np.float32)
    epi, ei = np.array([[0, 1]], np.int32), np.array([0], np.int32)
    return pg.get_paf_lines(pafs_sample, peaks_sample, epi, ei, n_line_points=3, pafs_stride=2), peaks_sample, epi


def test_paf_lines(pg):  # ref :60-74
    paf_lines, _, _ = _lines(pg)
    assert_array_equal(paf_lines, [[[0, 1], [18, 19], [36, 37]]])


def test_paf_lines_out_of_bounds_raises_like_tf_cpu(pg):
    with pytest.raises(IndexError):
        pg.get_paf_lines(np.zeros((3, 3, 2), np.float32), np.array([[0, 0], [40, 8]], np.float32), np.array([[0, 1]], np.int32),
                         np.array([0], np.int32), 3, 2)


def test_score_paf_lines(pg):  # ref :77-92
    paf_lines, peaks_sample, epi = _lines(pg)
    scores = pg.score_paf_lines(paf_lines, peaks_sample, epi, max_edge_length=2)
    assert_allclose(scores, [24.27], atol=1e-2)


def test_compute_distance_penalty(pg):  # ref :95-104
    assert_allclose(pg.compute_distance_penalty(np.array([1, 2, 3, 4], np.float32), max_edge_length=2), [0, 0, 2 / 3 - 1, 2 / 4 - 1],
                    atol=1e-6)
    assert_allclose(pg.compute_distance_penalty(np.array([1, 2, 3, 4], np.float32), max_edge_length=2, dist_penalty_weight=2),
                    [0, 0, -0.6666666, -1], atol=1e-6)
    assert pg.compute_distance_penalty(np.ones((3, 1), np.float32), 2).shape == (3, 1)


def test_score_paf_lines_batch(pg):  # ref :107-129
    pafs = np.arange(6 * 4 * 2, dtype=np.float32).reshape(1, 6, 4, 2)
    ei, epi, ls = pg.score_paf_lines_batch(pafs, [np.array([[0, 0], [4, 8]], np.float32)], [np.array([0, 1], np.int32)],
                                           np.array([[0, 1], [1, 2], [2, 3]], np.int32), 3, 2, 2 / 12, 1.0, 4)
    assert_array_equal([x.tolist() for x in ei], [[0]])
    assert_array_equal([x.tolist() for x in epi], [[[0, 1]]])
    assert_allclose(ls[0], [24.27], atol=1e-2)


def test_match_candidates_sample(pg):  # ref :132-155
    me, ms, md, msc = pg.match_candidates_sample(np.array([0, 0]), np.array([[0, 1], [2, 1]]), np.array([-0.5, 1.0], np.float32), 1)
    assert_array_equal(me, [0])
    assert_array_equal(ms, [1])
    assert_array_equal(md, [0])
    assert_array_equal(msc, [1.0])
    src_k = np.array([0, 2])  # tf.unique(edge_peak_inds[:, 0])
    assert src_k[ms][0] == 2


def test_match_candidates_batch_and_scorer_method(pg):  # ref :158-185, paf_grouping.py:1498
    args = ([np.array([0, 0], np.int32)], [np.array([[0, 1], [2, 1]], np.int32)], [np.array([-0.5, 1.0], np.float32)])
    for out in (pg.match_candidates_batch(*args, 1), pg.PAFScorer(["a", "b"], [("a", "b")], 2).match_candidates(*args)):
        assert_array_equal(out[0][0], [0])
        assert_array_equal(out[1][0], [1])
        assert_array_equal(out[2][0], [0])
        assert_array_equal(out[3][0], [1.0])


GROUP_ARGS = (np.arange(10, dtype=np.float32).reshape(5, 2), np.arange(5, dtype=np.float32), np.array([0, 1, 2, 0, 1], np.int32),
              np.array([0, 1, 0], np.int32), np.array([0, 0, 1], np.int32), np.array([0, 0, 1], np.int32), np.ones(3, np.float32))
WANT_INST = [[[0.0, 1.0], [2.0, 3.0], [4.0, 5.0]], [[6.0, 7.0], [8.0, 9.0], [np.nan, np.nan]]]


def test_group_instances_sample(pg):  # ref :188-231
    inst, ps, sc = pg.group_instances_sample(*GROUP_ARGS, 3, (0, 1), [pg.EdgeType(0, 1), pg.EdgeType(1, 2)], 0)
    assert_array_equal(inst, WANT_INST)
    assert_array_equal(ps, [[0.0, 1.0, 2.0], [3.0, 4.0, np.nan]])
    assert_array_equal(sc, [2.0, 1.0])


def test_group_instances_batch_and_scorer_method(pg):  # ref :234-299, paf_grouping.py:1552
    ragged = tuple([a] for a in GROUP_ARGS)
    scorer = pg.PAFScorer(["a", "b", "c"], [("a", "b"), ("b", "c")], 2)
    assert scorer.sorted_edge_inds == (0, 1)
    for out in (pg.group_instances_batch(*ragged, 3, (0, 1), [pg.EdgeType(0, 1), pg.EdgeType(1, 2)], 0),
                scorer.group_instances(*ragged)):
        assert_array_equal(out[0][0], WANT_INST)
        assert_array_equal(out[1][0], [[0.0, 1.0, 2.0], [3.0, 4.0, np.nan]])
        assert_array_equal(out[2][0], [2.0, 1.0])


def _connections_15(pg):
    E, C = pg.EdgeType, pg.EdgeConnection
    return {E(5, 7): [C(0, 0, 1.0465653)], E(5, 8): [C(0, 0, 1.0607507)], E(5, 9): [C(0, 0, 0.9563284)], E(5, 6): [C(0, 1, 0.5797864)],
            E(5, 11): [C(0, 0, 0.9892818)], E(5, 12): [C(0, 0, 0.7557168)], E(1, 0): [], E(1, 3): [], E(1, 2): [], E(1, 10): [],
            E(1, 13): [], E(1, 14): [], E(4, 5): [C(0, 0, 0.9735552)], E(4, 1): [C(0, 0, 0.31536198)]}


def test_assign_connections_to_instances(pg):  # ref :342-403
    P = pg.PeakID
    connections = _connections_15(pg)
    got = pg.assign_connections_to_instances(connections, min_instance_peaks=0, n_nodes=15)
    assert got == {P(5, 0): 0, P(7, 0): 0, P(8, 0): 0, P(9, 0): 0, P(6, 1): 0, P(11, 0): 0, P(12, 0): 0, P(4, 0): 1, P(1, 0): 1}
    edge_types = list(connections.keys())
    order = pg.toposort_edges(edge_types)
    got = pg.assign_connections_to_instances({edge_types[i]: connections[edge_types[i]] for i in order}, min_instance_peaks=0,
                                             n_nodes=15)
    assert len(got) == 9 and all(x == 0 for x in got.values())
    # min_instance_peaks (:887-913): the 2-peak instance goes, as an int and as a fraction of the node count
    for mip in (3, 0.2):
        got = pg.assign_connections_to_instances(connections, min_instance_peaks=mip, n_nodes=15)
        assert set(got.values()) == {0} and len(got) == 7


def test_make_predicted_instances(pg):
    """paf_grouping.py:917-981 on the reference's 15-node example: ids re-indexed contiguously, instance score = sum of the
    matched edge scores whose source peak is assigned (float32, dictionary order), NaN for missing nodes."""
    P = pg.PeakID
    connections = _connections_15(pg)
    rng = np.random.default_rng(0)
    peaks = [rng.random((2, 2)).astype(np.float32) for _ in range(15)]
    vals = [rng.random((2,)).astype(np.float32) for _ in range(15)]
    assign = {P(5, 0): 3, P(7, 0): 3, P(8, 0): 3, P(9, 0): 3, P(6, 1): 3, P(11, 0): 3, P(12, 0): 3, P(4, 0): 7, P(1, 0): 7}
    inst, pv, sc = pg.make_predicted_instances(peaks, vals, connections, assign)
    assert set(assign.values()) == {0, 1} and assign[P(4, 0)] == 1  # re-indexed in place like the reference
    assert inst.shape == (2, 15, 2) and pv.shape == (2, 15) and sc.shape == (2,)
    want0 = np.float32(0)
    for s_ in (1.0465653, 1.0607507, 0.9563284, 0.5797864, 0.9892818, 0.7557168):
        want0 = np.float32(want0 + np.float32(s_))
    want1 = np.float32(np.float32(0.9735552) + np.float32(0.31536198))
    assert_array_equal(sc, [want0, want1])
    assert_array_equal(inst[0, 6], peaks[6][1])
    assert_array_equal(inst[1, 4], peaks[4][0])
    assert_array_equal(pv[0, 5], vals[5][0])
    assert np.isnan(inst[0, 4]).all() and np.isnan(inst[1, 5]).all() and np.isnan(pv[1, 7])


def test_toposort_edges(pg):  # ref :302-339
    e1 = [(5, 7), (5, 8), (5, 9), (5, 6), (5, 11), (5, 12), (1, 0), (1, 3), (1, 2), (1, 10), (1, 13), (1, 14), (4, 5), (4, 1)]
    assert pg.toposort_edges([pg.EdgeType(*e) for e in e1]) == (12, 13, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11)
    e2 = [(1, 4), (1, 5), (6, 8), (6, 7), (6, 9), (9, 10), (1, 0), (1, 3), (1, 2), (6, 1)]
    assert pg.toposort_edges([pg.EdgeType(*e) for e in e2]) == (2, 3, 4, 9, 5, 0, 1, 6, 7, 8)


# ------------------------------------------------------------------ device matcher / fused kernel
def test_wave_matcher_equals_scipy_on_device(pg):
    """One wavefront per (frame, edge): random and tie-heavy score blocks, rectangular both ways, NaN scores (-> +inf cost)."""
    from scipy.optimize import linear_sum_assignment

    from sleap_amd import ops

    rng = np.random.default_rng(3)
    B, E, NP = 6, 5, 12
    scores = np.full((B, E, NP, NP), np.nan, np.float32)
    node_count = np.zeros((B, 2 * E), np.int32)
    for b in range(B):
        for k in range(E):
            ns, nd = rng.integers(0, NP + 1, 2)
            node_count[b, 2 * k], node_count[b, 2 * k + 1] = ns, nd
            m = rng.integers(0, 3, (ns, nd)).astype(np.float32) if (b + k) % 2 else rng.normal(size=(ns, nd)).astype(np.float32)
            if ns and nd and k == 2:
                m[rng.random(m.shape) < 0.2] = np.nan
                m[0, 0] = 0.5  # keep row 0 feasible
                m[:, 0] = np.where(np.isnan(m[:, 0]), 0.25, m[:, 0])
            scores[b, k, :ns, :nd] = m
    edges = torch.tensor([[2 * k, 2 * k + 1] for k in range(E)], dtype=torch.int32).cuda()
    status = torch.zeros((B,), dtype=torch.int32, device="cuda")
    md, msc = ops.paf_match(torch.from_numpy(scores).cuda(), torch.from_numpy(node_count).cuda(), edges, status)
    md, msc = md.cpu().numpy(), msc.cpu().numpy()
    assert int(status.cpu().numpy().max()) == 0
    for b in range(B):
        for k in range(E):
            ns, nd = node_count[b, 2 * k], node_count[b, 2 * k + 1]
            want = np.full((NP,), -1)
            if ns and nd:
                cost = np.where(np.isnan(scores[b, k, :ns, :nd]), np.inf, -scores[b, k, :ns, :nd].astype(np.float64))
                r, c = linear_sum_assignment(cost)
                want[r] = c
            assert_array_equal(md[b, k], want, err_msg=f"frame {b} edge {k} ({ns}x{nd})")
            sel = want >= 0
            assert_array_equal(msc[b, k][sel], scores[b, k][np.nonzero(sel)[0], want[sel]])
            assert np.isnan(msc[b, k][~sel]).all()


@pytest.mark.parametrize("seed,noise,refinement", [(0, 0.0, "integral"), (5, 0.01, "local"), (6, 0.05, "integral")])
def test_fused_postproc_equals_separate_stages(pg, seed, noise, refinement):
    """sa_bottomup_postproc (NMS scan + one workgroup per frame) == find_local_peaks + paf_score + paf_match + paf_group, bit for
    bit, on every table (analytic 13-node maps with 4 animals + noise)."""
    from oracle.synth import FLIES13_EDGES, FLIES13_NODES

    from sleap_amd import ops
    from test_gpu_postproc import _synth_batch

    cms, pafs, _ = _synth_batch(seed, B=3, size=512, animals=4, noise=noise)
    cms_t, pafs_t = torch.from_numpy(cms).cuda(), torch.from_numpy(pafs).cuda()
    sc = pg.PAFScorer(FLIES13_NODES, FLIES13_EDGES, 8, max_node_peaks=16, max_instances=16)
    fused = sc.predict_from_maps(cms_t, None, pafs_t, 0.2, refinement, 5, 4, 256)
    xy, val, ch, cnt, st = ops.find_local_peaks(cms_t, None, 0.2, refinement, 5, 4.0, 256)
    inst, vals, scores, n_inst, status, graph = sc.predict_padded(pafs_t, xy, val, ch, cnt, st, return_graph=True)
    torch.cuda.synchronize()
    n = cnt.cpu().numpy()
    assert_array_equal(fused["peak_count"].cpu().numpy(), n)
    assert n.min() >= 40
    for b in range(len(n)):
        assert_array_equal(fused["peak_xy"][b, : n[b]].cpu().numpy(), xy[b, : n[b]].cpu().numpy())
        assert_array_equal(fused["peak_chan"][b, : n[b]].cpu().numpy(), ch[b, : n[b]].cpu().numpy())
    assert_array_equal(fused["node_count"].cpu().numpy(), graph[0].cpu().numpy())
    assert_array_equal(fused["node_peaks"].cpu().numpy(), graph[1].cpu().numpy())
    assert_array_equal(fused["match_dst"].cpu().numpy(), graph[3].cpu().numpy())
    assert_array_equal(fused["match_score"].cpu().numpy(), graph[4].cpu().numpy())
    assert_array_equal(fused["n_instances"].cpu().numpy(), n_inst.cpu().numpy())
    assert n_inst.cpu().numpy().min() >= 3
    assert_array_equal(fused["instance_peaks"].cpu().numpy(), inst.cpu().numpy())
    assert_array_equal(fused["instance_peak_vals"].cpu().numpy(), vals.cpu().numpy())
    assert_array_equal(fused["instance_scores"].cpu().numpy(), scores.cpu().numpy())
    assert_array_equal(fused["status"].cpu().numpy(), status.cpu().numpy())
