"""Pins oracle/paf_grouping.py to the reference's known-answer tests
(reference: tests/nn/test_paf_grouping.py; line numbers cited per test)."""
import numpy as np
from numpy.testing import assert_allclose, assert_array_equal

from oracle import paf_grouping as pg
from oracle.paf_grouping import EdgeConnection, EdgeType, PeakID


def test_get_connection_candidates():  # ref :28-41
    ei, epi = pg.get_connection_candidates([0, 0, 0, 1, 1, 2], [[0, 1], [1, 2], [2, 3]], 4)
    assert_array_equal(ei, [0, 0, 0, 0, 0, 0, 1, 1])
    assert_array_equal(epi, [[0, 3], [0, 4], [1, 3], [1, 4], [2, 3], [2, 4], [3, 5], [4, 5]])


def test_make_line_subs():  # ref :44-55 (includes half-to-even rounding: 1/2 -> 0... 2/2 -> 1)
    subs = pg.make_line_subs(np.array([[0, 0], [4, 8]], np.float32), [[0, 1]], [0], 3, 2)
    assert_array_equal(subs, [[[[0, 0, 0], [0, 0, 1]], [[2, 1, 0], [2, 1, 1]], [[4, 2, 0], [4, 2, 1]]]])


def test_round_half_to_even():
    # x = 1, 3, 5 with stride 2 -> 0.5, 1.5, 2.5 -> 0, 2, 2 (tf.round)
    subs = pg.make_line_subs(np.array([[1, 0], [5, 0]], np.float32), [[0, 1]], [0], 3, 2)
    assert_array_equal(subs[0, :, 0, 1], [0, 2, 2])


def _pafs():
    return np.arange(6 * 4 * 2, dtype=np.float32).reshape(6, 4, 2)


def test_paf_lines():  # ref :58-72
    lines = pg.get_paf_lines(_pafs(), np.array([[0, 0], [4, 8]], np.float32), [[0, 1]], [0], 3, 2)
    assert_array_equal(lines, [[[0, 1], [18, 19], [36, 37]]])


def test_score_paf_lines():  # ref :75-90
    peaks = np.array([[0, 0], [4, 8]], np.float32)
    lines = pg.get_paf_lines(_pafs(), peaks, [[0, 1]], [0], 3, 2)
    scores = pg.score_paf_lines(lines, peaks, [[0, 1]], max_edge_length=2)
    assert_allclose(scores, [24.27], atol=1e-2)


def test_compute_distance_penalty():  # ref :93-102
    p = pg.compute_distance_penalty(np.array([1, 2, 3, 4], np.float32), max_edge_length=2)
    assert_allclose(p, [0, 0, 2 / 3 - 1, 2 / 4 - 1], atol=1e-6)
    p = pg.compute_distance_penalty(np.array([1, 2, 3, 4], np.float32), 2, dist_penalty_weight=2)
    assert_allclose(p, [0, 0, -0.6666666, -1], atol=1e-6)


def test_score_paf_lines_batch():  # ref :105-129
    ei, epi, ls = pg.score_paf_lines_batch(
        _pafs()[None], [np.array([[0, 0], [4, 8]], np.float32)], [np.array([0, 1])],
        [[0, 1], [1, 2], [2, 3]], 3, 2, 2 / 12, 1.0, 4,
    )
    assert_array_equal(ei[0], [0])
    assert_array_equal(epi[0], [[0, 1]])
    assert_allclose(ls[0], [24.27], atol=1e-2)


def test_match_candidates_sample():  # ref :132-160
    me, ms, md, msc = pg.match_candidates_sample([0, 0], [[0, 1], [2, 1]], [-0.5, 1.0], 1)
    assert_array_equal(me, [0])
    assert_array_equal(ms, [1])
    assert_array_equal(md, [0])
    assert_array_equal(msc, [1.0])


def test_match_candidates_batch():  # ref :163-185
    me, ms, md, msc = pg.match_candidates_batch([[0, 0]], [[[0, 1], [2, 1]]], [[-0.5, 1.0]], 1)
    assert_array_equal(np.concatenate(me), [0])
    assert_array_equal(np.concatenate(ms), [1])
    assert_array_equal(np.concatenate(md), [0])
    assert_array_equal(np.concatenate(msc), [1.0])


def _group_args():
    return dict(
        peaks_sample=np.arange(10, dtype=np.float32).reshape(5, 2),
        peak_scores_sample=np.arange(5, dtype=np.float32),
        peak_channel_inds_sample=np.array([0, 1, 2, 0, 1], np.int32),
        match_edge_inds_sample=np.array([0, 1, 0], np.int32),
        match_src_peak_inds_sample=np.array([0, 0, 1], np.int32),
        match_dst_peak_inds_sample=np.array([0, 0, 1], np.int32),
        match_line_scores_sample=np.ones(3, np.float32),
        n_nodes=3,
        sorted_edge_inds=(0, 1),
        edge_types=[EdgeType(0, 1), EdgeType(1, 2)],
        min_instance_peaks=0,
    )


def test_group_instances_sample():  # ref :188-231
    inst, ps, sc = pg.group_instances_sample(**_group_args())
    assert_array_equal(
        inst, [[[0.0, 1.0], [2.0, 3.0], [4.0, 5.0]], [[6.0, 7.0], [8.0, 9.0], [np.nan, np.nan]]]
    )
    assert_array_equal(ps, [[0.0, 1.0, 2.0], [3.0, 4.0, np.nan]])
    assert_array_equal(sc, [2.0, 1.0])


def test_group_instances_batch():  # ref :234-299
    a = _group_args()
    lists = {k: [v] for k, v in a.items() if k.endswith("_sample")}
    inst, ps, sc = pg.group_instances_batch(
        lists["peaks_sample"], lists["peak_scores_sample"], lists["peak_channel_inds_sample"],
        lists["match_edge_inds_sample"], lists["match_src_peak_inds_sample"],
        lists["match_dst_peak_inds_sample"], lists["match_line_scores_sample"],
        3, (0, 1), a["edge_types"], 0,
    )
    assert_array_equal(sc[0], [2.0, 1.0])
    assert_array_equal(ps[0], [[0.0, 1.0, 2.0], [3.0, 4.0, np.nan]])


EDGES_15 = [(5, 7), (5, 8), (5, 9), (5, 6), (5, 11), (5, 12), (1, 0), (1, 3), (1, 2), (1, 10),
            (1, 13), (1, 14), (4, 5), (4, 1)]


def test_toposort_edges():  # ref :302-339
    assert pg.toposort_edges([EdgeType(*e) for e in EDGES_15]) == (12, 13, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11)
    e2 = [(1, 4), (1, 5), (6, 8), (6, 7), (6, 9), (9, 10), (1, 0), (1, 3), (1, 2), (6, 1)]
    assert pg.toposort_edges([EdgeType(*e) for e in e2]) == (2, 3, 4, 9, 5, 0, 1, 6, 7, 8)


def test_assign_connections_to_instances():  # ref :342-403
    scores = {(5, 7): (0, 0, 1.0465653), (5, 8): (0, 0, 1.0607507), (5, 9): (0, 0, 0.9563284),
              (5, 6): (0, 1, 0.5797864), (5, 11): (0, 0, 0.9892818), (5, 12): (0, 0, 0.7557168),
              (4, 5): (0, 0, 0.9735552), (4, 1): (0, 0, 0.31536198)}
    connections = {}
    for e in EDGES_15:
        connections[EdgeType(*e)] = [EdgeConnection(*scores[e])] if e in scores else []
    ia = pg.assign_connections_to_instances(connections, min_instance_peaks=0, n_nodes=15)
    assert ia == {
        PeakID(5, 0): 0, PeakID(7, 0): 0, PeakID(8, 0): 0, PeakID(9, 0): 0, PeakID(6, 1): 0,
        PeakID(11, 0): 0, PeakID(12, 0): 0, PeakID(4, 0): 1, PeakID(1, 0): 1,
    }
    ets = list(connections.keys())
    order = pg.toposort_edges(ets)
    ia = pg.assign_connections_to_instances({ets[i]: connections[ets[i]] for i in order}, 0, 15)
    assert all(x == 0 for x in ia.values())


def test_paf_scorer_end_to_end_synthetic():
    """Full oracle chain on analytic maps recovers the planted instances (mirrors the
    intent of ref tests/nn/test_inference.py:769-806 without the undecodable mp4)."""
    from oracle import peak_finding as pf
    from oracle.synth import FLIES13_EDGES, FLIES13_NODES, random_fly_instances, synth_bottomup_maps

    rng = np.random.default_rng(3)
    inst = random_fly_instances(rng, 3, 512, 512, margin=96)
    cms, pafs, _ = synth_bottomup_maps(inst, 512, 512, noise=0.0)
    pts, vals, si, ci = pf.find_local_peaks(cms[None], 0.2, "integral", 5)
    pts = pts * 4
    scorer = pg.PAFScorer(FLIES13_NODES, FLIES13_EDGES, pafs_stride=8)
    out = scorer.predict(pafs[None], [pts], [vals], [ci])
    got = out[0][0]
    assert got.shape == (3, 13, 2)
    # match each predicted instance to a planted one by thorax distance
    for g in got:
        d = np.linalg.norm(inst[:, 1] - g[1], axis=-1)
        j = int(np.argmin(d))
        assert np.nanmax(np.linalg.norm(inst[j] - g, axis=-1)) < 1.0
