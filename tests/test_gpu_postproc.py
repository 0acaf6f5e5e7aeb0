"""Parity of the HIP post-processing kernels (through the C ABI) against the CPU oracle and the
reference's known-answer vectors. Peak order / indices / assignments: exact. Coordinates and
scores: |delta| <= 1e-5 (float32 summation order is the only degree of freedom)."""
import numpy as np
import pytest
import torch
from numpy.testing import assert_allclose, assert_array_equal

from oracle import paf_grouping as opg
from oracle import peak_finding as opf
from oracle.synth import (FLIES13_EDGES, FLIES13_NODES, make_confmaps, make_grid_vectors, make_multi_confmaps,
                          random_fly_instances, synth_bottomup_maps)

pytestmark = pytest.mark.gpu

TOL = 1e-5


def _n(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def pf():
    from sleap_amd.nn import peak_finding

    return peak_finding


@pytest.fixture(scope="module")
def pg():
    from sleap_amd.nn import paf_grouping

    return paf_grouping


def _two_sample_cms(size, scale=1.0, shift=0.0):
    xv, yv = make_grid_vectors(size, size, 1)
    inst = np.array([[[1, 2], [3, 4]], [[5, 6], [7, 8]], [[np.nan, np.nan], [11, 12]]], np.float32) * scale + shift
    inst2 = np.array([[[2, 3], [4, 5]], [[6, 7], [8, 9]]], np.float32) * scale + shift
    return np.stack([make_multi_confmaps(inst, xv, yv, 1.0), make_multi_confmaps(inst2, xv, yv, 1.0)], axis=0)


EXPECTED_PTS = np.array([[1, 2], [3, 4], [5, 6], [7, 8], [11, 12], [2, 3], [4, 5], [6, 7], [8, 9]], np.float32)


# ------------------------------------------------------------------ reference known answers
def test_local_peaks_rough_known_answer(pf):  # ref tests/nn/test_peak_finding.py:141-198
    pts, vals, si, ci = pf.find_local_peaks(_two_sample_cms(16), threshold=0.1, refinement=None)
    assert_array_equal(_n(pts), EXPECTED_PTS)
    assert_array_equal(_n(vals), np.ones(9))
    assert_array_equal(_n(si), [0, 0, 0, 0, 0, 1, 1, 1, 1])
    assert_array_equal(_n(ci), [0, 1, 0, 1, 1, 0, 1, 0, 1])
    pts, vals, si, ci = pf.find_local_peaks(np.zeros([1, 4, 4, 3], np.float32), threshold=0.1)
    assert tuple(pts.shape) == (0, 2) and tuple(vals.shape) == (0,)


def test_local_peaks_integral_and_local_known_answer(pf):  # ref :201-337
    pts, vals, si, ci = pf.find_local_peaks(_two_sample_cms(32, 2.0, 0.3), 0.1, "integral", 5)
    assert_allclose(_n(pts), EXPECTED_PTS * 2 + 0.3, atol=0.2)
    assert_array_equal(_n(ci), [0, 1, 0, 1, 1, 0, 1, 0, 1])
    pts, vals, si, ci = pf.find_local_peaks(_two_sample_cms(32, 2.0, 0.25), 0.1, "local")
    assert_allclose(_n(pts), EXPECTED_PTS * 2 + 0.25)


def test_global_peaks_known_answer(pf):  # ref :49-138
    xv, yv = make_grid_vectors(8, 8, 1)
    points = np.array([[1, 2], [3, 4], [5, 6]], np.float32)
    cms = np.stack([make_confmaps(points, xv, yv, 1.0), make_confmaps(points + 1, xv, yv, 1.0)])
    peaks, vals = pf.find_global_peaks(cms, threshold=0.1, refinement=None)
    assert_array_equal(_n(peaks)[0], points)
    assert_array_equal(_n(peaks)[1], points + 1)
    assert_array_equal(_n(vals), np.ones((2, 3)))
    peaks, vals = pf.find_global_peaks_rough(np.zeros((1, 8, 8, 3), np.float32), threshold=0.1)
    assert np.isnan(_n(peaks)).all()
    assert_array_equal(_n(vals), [[0, 0, 0]])
    xv, yv = make_grid_vectors(12, 12, 1)
    p2 = np.array([[1.6, 2.6], [3.6, 4.6], [5.6, 6.6]], np.float32)
    peaks, vals = pf.find_global_peaks(make_confmaps(p2, xv, yv, 1.0)[None], 0.1, "local")
    assert_allclose(_n(peaks)[0], [[1.75, 2.75], [3.75, 4.75], [5.75, 6.75]])


# ------------------------------------------------------------------ oracle parity on seeded inputs
def _synth_batch(seed, B=4, size=512, animals=4, noise=0.01):
    rng = np.random.default_rng(seed)
    cms, pafs, insts = [], [], []
    for _ in range(B):
        inst = random_fly_instances(rng, animals, size, size, margin=96)
        c, p, _ = synth_bottomup_maps(inst, size, size, noise=noise, rng=rng)
        cms.append(c)
        pafs.append(p)
        insts.append(inst)
    return np.stack(cms), np.stack(pafs), insts


@pytest.mark.parametrize("refinement", [None, "integral", "local"])
def test_local_peaks_vs_oracle(pf, refinement):
    cms, _, _ = _synth_batch(0)
    o = opf.find_local_peaks(cms, 0.2, refinement, 5)
    g = pf.find_local_peaks(cms, 0.2, refinement, 5)
    assert len(o[0]) > 100
    assert_array_equal(_n(g[2]), o[2])
    assert_array_equal(_n(g[3]), o[3])
    assert_array_equal(_n(g[1]), o[1])
    assert_allclose(_n(g[0]), o[0], atol=TOL)


def test_rough_scan_entry_point_vs_oracle():
    """sa_find_local_peaks_rough (round 6: the NMS scan alone -- the kernel bench.py's roofline_postproc times): the SET of
    linear (y, x, c) keys per frame == the oracle's find_local_peaks_rough (peak_finding.py:249-308); arrival order is free."""
    from sleap_amd import ops

    cms, _, _ = _synth_batch(0)
    B, H, W, C = cms.shape
    keys, cnt, st = ops.find_local_peaks_rough(torch.from_numpy(cms).cuda(), 0.2, max_peaks=1024)
    assert int(st.max()) == 0
    pts, _, si, ci = opf.find_local_peaks_rough(cms, 0.2)
    want = ((pts[:, 1].astype(np.int64) * W + pts[:, 0].astype(np.int64)) * C + ci)
    keys, cnt = _n(keys).view(np.uint32), _n(cnt)
    assert int(cnt.sum()) == len(want) > 100
    for b in range(B):
        assert_array_equal(np.sort(keys[b, :cnt[b]].astype(np.int64)), np.sort(want[si == b]))


def test_local_peaks_random_noise_many_peaks(pf):
    rng = np.random.default_rng(1)
    cms = rng.random((2, 37, 53, 5)).astype(np.float32)  # odd sizes: scalar (non-float4) path, borders everywhere
    for ref in ("integral", "local"):
        o = opf.find_local_peaks(cms, 0.5, ref, 5)
        g = pf.find_local_peaks(cms, 0.5, ref, 5, max_peaks=4096)
        assert_array_equal(_n(g[3]), o[3])
        assert_array_equal(_n(g[1]), o[1])
        assert_allclose(_n(g[0]), o[0], atol=TOL, equal_nan=True)
    o = opf.find_local_peaks(cms, 0.5, "integral", 3)
    g = pf.find_local_peaks(cms, 0.5, "integral", 3, max_peaks=4096)
    assert_allclose(_n(g[0]), o[0], atol=TOL)


def test_local_peaks_with_offsets_vs_oracle(pf):
    rng = np.random.default_rng(2)
    cms, _, _ = _synth_batch(2, B=2, size=256, animals=2)
    offs = rng.normal(0, 0.3, cms.shape[:3] + (2 * cms.shape[3],)).astype(np.float32)
    o = opf.find_local_peaks_with_offsets(cms, offs, 0.2)
    g = pf.find_local_peaks_with_offsets(cms, offs, 0.2)
    assert_array_equal(_n(g[3]), o[3])
    assert_allclose(_n(g[0]), o[0], atol=1e-6)


def test_local_peaks_with_offsets_on_the_sleap_trained_fixtures_own_maps(pf):
    """Row a5 on REAL weights: the confidence maps and the learned offset maps of the one bottom-up model
    SLEAP itself trained (`minimal_instance.UNet.bottomup`: OffsetRefinementHead, peak_finding.py:646-707), computed by the fp32
    oracle on six synthetic frames and handed to BOTH peak finders -- same fp32 maps in, so there is no storage error and no
    excuse: identical peak sets (sample, channel, value, order) at three thresholds, coordinates within 1e-5 px. The maps are
    noisy far outside the training data (211 maxima per frame above 0.2, margins over a neighbour down to 4e-5), i.e. a harder
    input for the strict-`>` scan and the ordering than the smooth synthetic maps above."""
    import os

    from oracle.keras_graph import KerasGraph, load_npz_model, preprocess
    from sleap_amd.synth import render_frames

    model = os.path.join(os.path.dirname(__file__), "golden", "models", "minimal_instance.UNet.bottomup")
    frames = render_frames(6, 384, 384, n_animals=2, seed=11)[0]
    cms, _, offs = KerasGraph(*load_npz_model(os.path.join(model, "best_model.npz")))(preprocess(frames))
    assert cms.shape == (6, 192, 192, 2) and offs.shape == (6, 192, 192, 4)
    for thr, at_least in ((0.2, 1000), (0.5, 400), (0.9, 20)):
        o = opf.find_local_peaks_with_offsets(cms, offs, thr)
        g = pf.find_local_peaks_with_offsets(cms, offs, thr, max_peaks=2048)
        assert len(o[0]) >= at_least
        assert_array_equal(_n(g[2]), o[2])
        assert_array_equal(_n(g[3]), o[3])
        assert_array_equal(_n(g[1]), o[1])
        assert_allclose(_n(g[0]), o[0], atol=TOL)


def test_peak_overflow_is_flagged(pf):
    rng = np.random.default_rng(3)
    cms = rng.random((1, 64, 64, 4)).astype(np.float32)
    with pytest.raises(pf.PeakOverflowError):
        pf.find_local_peaks(cms, 0.2, None, max_peaks=16)


@pytest.mark.parametrize("C", [4, 3])  # float4 scan and scalar scan
@pytest.mark.parametrize("bad", [float("inf"), float("-inf"), float("nan")])
def test_nonfinite_confidence_maps_are_flagged(C, bad):
    """SA_STATUS_NONFINITE: an fp16-storage network that overflowed hands inf / NaN maps to peak finding; the frame is
    flagged (BottomUpInferenceModel.call_checked raises) instead of silently producing fewer peaks."""
    import torch

    from sleap_amd import _lib, ops

    cms = torch.rand((3, 17, 19, C), generator=torch.Generator().manual_seed(1)).mul(0.1).cuda()
    st = ops.find_local_peaks(cms, None, 0.2, None, 5, 1.0, 64)[4].cpu().numpy()
    assert (st & _lib.STATUS_NONFINITE == 0).all()
    cms[1, 16, 18, C - 1] = bad
    st = ops.find_local_peaks(cms, None, 0.2, None, 5, 1.0, 64)[4].cpu().numpy()
    assert [int(v) & _lib.STATUS_NONFINITE for v in st] == [0, _lib.STATUS_NONFINITE, 0]


@pytest.mark.parametrize("refinement", [None, "integral", "local"])
def test_global_peaks_vs_oracle(pf, refinement):
    rng = np.random.default_rng(4)
    xv, yv = make_grid_vectors(40, 48, 1)
    cms = np.stack([make_confmaps(rng.uniform(1, 38, (6, 2)).astype(np.float32), xv, yv, 1.5) for _ in range(3)])
    cms[1, :, :, 2] = 0.0  # below threshold -> NaN
    cms += rng.normal(0, 0.01, cms.shape).astype(np.float32)
    o = opf.find_global_peaks(cms, 0.2, refinement, 5)
    g = pf.find_global_peaks(cms, 0.2, refinement, 5)
    assert_allclose(_n(g[0]), o[0], atol=TOL, equal_nan=True)
    assert_array_equal(_n(g[1]), o[1])


# ------------------------------------------------------------------ PAF grouping
def test_paf_known_answers(pg):  # ref tests/nn/test_paf_grouping.py:105-129, 132-185, 188-231
    pafs = np.arange(6 * 4 * 2, dtype=np.float32).reshape(1, 6, 4, 2)
    ei, epi, ls = pg.score_paf_lines_batch(pafs, [np.array([[0, 0], [4, 8]], np.float32)], [np.array([0, 1])],
                                           [[0, 1], [1, 2], [2, 3]], 3, 2, 2 / 12, 1.0, 4)
    assert_array_equal(ei[0], [0])
    assert_array_equal(epi[0], [[0, 1]])
    assert_allclose(ls[0], [24.27], atol=1e-2)
    me, ms, md, msc = pg.match_candidates_sample([0, 0], [[0, 1], [2, 1]], [-0.5, 1.0], 1)
    assert_array_equal(me, [0])
    assert_array_equal(ms, [1])
    assert_array_equal(md, [0])
    assert_array_equal(msc, [1.0])
    inst, ps, sc = pg.group_instances_sample(
        np.arange(10, dtype=np.float32).reshape(5, 2), np.arange(5, dtype=np.float32),
        np.array([0, 1, 2, 0, 1], np.int32), np.array([0, 1, 0], np.int32), np.array([0, 0, 1], np.int32),
        np.array([0, 0, 1], np.int32), np.ones(3, np.float32), 3, (0, 1), [(0, 1), (1, 2)], 0)
    assert_array_equal(inst, [[[0.0, 1.0], [2.0, 3.0], [4.0, 5.0]], [[6.0, 7.0], [8.0, 9.0], [np.nan, np.nan]]])
    assert_array_equal(ps, [[0.0, 1.0, 2.0], [3.0, 4.0, np.nan]])
    assert_array_equal(sc, [2.0, 1.0])


def test_assign_connections_known_answer(pg):  # ref :342-403 via the grouping kernel
    EDGES_15 = [(5, 7), (5, 8), (5, 9), (5, 6), (5, 11), (5, 12), (1, 0), (1, 3), (1, 2), (1, 10), (1, 13),
                (1, 14), (4, 5), (4, 1)]
    conn = {(5, 7): (0, 0, 1.0465653), (5, 8): (0, 0, 1.0607507), (5, 9): (0, 0, 0.9563284),
            (5, 6): (0, 1, 0.5797864), (5, 11): (0, 0, 0.9892818), (5, 12): (0, 0, 0.7557168),
            (4, 5): (0, 0, 0.9735552), (4, 1): (0, 0, 0.31536198)}
    # two peaks per node type so that (6, peak 1) exists
    ch = np.repeat(np.arange(15), 2).astype(np.int32)
    peaks = np.stack([np.arange(30), ch], axis=1).astype(np.float32)
    me = np.array([EDGES_15.index(e) for e in conn], np.int32)
    ms = np.array([v[0] for v in conn.values()], np.int32)
    md = np.array([v[1] for v in conn.values()], np.int32)
    sc = np.array([v[2] for v in conn.values()], np.float32)
    for order, n_expected in ((tuple(range(14)), 2), (pg.toposort_edges(EDGES_15), 1)):
        inst, ps, isc = pg.group_instances_sample(peaks, np.ones(30, np.float32), ch, me, ms, md, sc, 15, order,
                                                  EDGES_15, 0)
        o = opg.group_instances_sample(peaks, np.ones(30, np.float32), ch, me, ms, md, sc, 15, order,
                                       [opg.EdgeType(*e) for e in EDGES_15], 0)
        assert inst.shape[0] == n_expected == o[0].shape[0]
        assert_array_equal(inst, o[0])
        assert_allclose(isc, o[2], rtol=1e-6)


@pytest.mark.parametrize("seed,noise", [(0, 0.0), (5, 0.01), (6, 0.05)])
def test_full_postproc_vs_oracle(pf, pg, seed, noise):
    cms, pafs, insts = _synth_batch(seed, B=4, size=512, animals=4, noise=noise)
    B = cms.shape[0]
    # oracle
    o_pts, o_vals, o_si, o_ci = opf.find_local_peaks(cms, 0.2, "integral", 5)
    o_pts = o_pts * np.float32(4)
    peaks = [o_pts[o_si == b] for b in range(B)]
    vals = [o_vals[o_si == b] for b in range(B)]
    chans = [o_ci[o_si == b] for b in range(B)]
    osc = opg.PAFScorer(FLIES13_NODES, FLIES13_EDGES, 8, oob="zero")
    o = osc.predict(pafs, peaks, vals, chans)
    # device, hot-path form
    from sleap_amd import ops

    dcms, dpafs = ops.to_cuda_f32(cms), ops.to_cuda_f32(pafs)
    pxy, pval, pch, pcnt, st = ops.find_local_peaks(dcms, None, 0.2, "integral", 5, 4.0, 512)
    sc = pg.PAFScorer(FLIES13_NODES, FLIES13_EDGES, 8)
    inst, ivals, iscores, n_inst, st, graph = sc.predict_padded(dpafs, pxy, pval, pch, pcnt, st, return_graph=True)
    assert int(st.max().item()) & ~16 == 0
    n = _n(n_inst)
    ei, epi, ls = sc._graph_to_ragged(graph, B)
    for b in range(B):
        assert n[b] == o[0][b].shape[0]
        assert_allclose(_n(inst)[b, : n[b]], o[0][b], atol=TOL * 4, equal_nan=True)
        assert_array_equal(np.isnan(_n(inst)[b, : n[b]]), np.isnan(o[0][b]))
        assert_array_equal(_n(ivals)[b, : n[b]], o[1][b])
        assert_allclose(_n(iscores)[b, : n[b]], o[2][b], atol=1e-5)
        assert_array_equal(ei[b], o[3][b])
        assert_array_equal(epi[b], o[4][b])
        assert_allclose(ls[b], o[5][b], atol=1e-5)
        assert np.isnan(_n(inst)[b, n[b]:]).all()
    # reference-shaped API gives the same thing
    r = sc.predict(pafs, peaks, vals, chans)
    for b in range(B):
        assert_allclose(r[0][b], o[0][b], atol=TOL * 4, equal_nan=True)


def test_empty_frames(pf, pg):
    from sleap_amd import ops

    cms = torch.zeros((2, 64, 64, 13), device="cuda")
    pafs = torch.zeros((2, 32, 32, 24), device="cuda")
    pxy, pval, pch, pcnt, st = ops.find_local_peaks(cms, None, 0.2, "integral", 5, 4.0, 64)
    sc = pg.PAFScorer(FLIES13_NODES, FLIES13_EDGES, 8)
    inst, ivals, iscores, n_inst, st = sc.predict_padded(pafs, pxy, pval, pch, pcnt, st)
    assert _n(n_inst).tolist() == [0, 0] and _n(st).tolist() == [0, 0]
    assert np.isnan(_n(inst)).all()


def test_min_line_scores_filters_everything(pg):  # ref tests/nn/test_inference.py:795-800 (min_line_scores=1.1)
    cms, pafs, _ = _synth_batch(7, B=1, size=256, animals=2, noise=0.0)
    o = opf.find_local_peaks(cms, 0.2, "integral", 5)
    sc = pg.PAFScorer(FLIES13_NODES, FLIES13_EDGES, 8, min_line_scores=1.1)
    r = sc.predict(pafs, [o[0] * 4], [o[1]], [o[3]])
    assert r[0][0].shape[0] == 0


def test_matching_random_vs_scipy(pg):
    rng = np.random.default_rng(8)
    ei, epi, ls = [], [], []
    for _ in range(16):
        ns, nd = rng.integers(1, 9, size=2)
        s, d = np.meshgrid(np.arange(ns), np.arange(nd) + 100, indexing="ij")
        epi.append(np.stack([s, d], -1).reshape(-1, 2))
        ei.append(np.zeros(ns * nd, np.int32))
        v = rng.normal(size=ns * nd).astype(np.float32)
        v[rng.random(ns * nd) < 0.1] = np.nan
        if np.isnan(v.reshape(ns, nd)).all(axis=1).any() or np.isnan(v.reshape(ns, nd)).all(axis=0).any():
            v = np.nan_to_num(v, nan=0.0)
        ls.append(v)
    try:
        o = opg.match_candidates_batch(ei, epi, ls, 1)
    except ValueError:
        pytest.skip("random draw infeasible")
    g = pg.match_candidates_batch(ei, epi, ls, 1)
    for b in range(16):
        assert_array_equal(g[1][b], o[1][b])
        assert_array_equal(g[2][b], o[2][b])
        assert_array_equal(g[3][b], o[3][b])


def test_hundreds_of_peaks_per_node_type(pf, pg):
    """No fixed small cap: ~200 peaks per node type (tables in the global workspace instead of LDS/registers)."""
    from sleap_amd import ops

    rng = np.random.default_rng(9)
    cms = rng.random((1, 64, 64, 3)).astype(np.float32)
    pafs = rng.normal(0, 1, (1, 32, 32, 4)).astype(np.float32)
    nodes, edges = ["a", "b", "c"], [("a", "b"), ("b", "c")]
    o_pts, o_vals, o_si, o_ci = opf.find_local_peaks(cms, 0.93, "integral", 5)
    assert 150 < np.bincount(o_ci).max() <= 512
    o_pts = o_pts * np.float32(2)
    osc = opg.PAFScorer(nodes, edges, 2, oob="zero")
    o = osc.predict(pafs, [o_pts], [o_vals], [o_ci])
    sc = pg.PAFScorer(nodes, edges, 2, max_node_peaks=512, max_instances=1024)
    pxy, pval, pch, pcnt, st = ops.find_local_peaks(ops.to_cuda_f32(cms), None, 0.93, "integral", 5, 2.0, 2048)
    inst, ivals, iscores, n_inst, st = sc.predict_padded(ops.to_cuda_f32(pafs), pxy, pval, pch, pcnt, st)
    assert int(st.max().item()) & ~16 == 0
    n = int(n_inst[0].item())
    assert n == o[0][0].shape[0] and n > 50
    assert_allclose(_n(inst)[0, :n], o[0][0], atol=1e-4, equal_nan=True)
    assert_allclose(_n(iscores)[0, :n], o[2][0], atol=1e-4)
