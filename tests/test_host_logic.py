"""CPU-only checks of host logic that does not touch the GPU."""
import numpy as np
import pytest

from sleap_amd.nn.paf_grouping import PAFScorer, toposort_edges
from oracle import paf_grouping as opg
from oracle.synth import FLIES13_EDGES, FLIES13_NODES

EDGES_15 = [(5, 7), (5, 8), (5, 9), (5, 6), (5, 11), (5, 12), (1, 0), (1, 3), (1, 2), (1, 10),
            (1, 13), (1, 14), (4, 5), (4, 1)]


def test_toposort_matches_reference_vectors():  # reference tests/nn/test_paf_grouping.py:302-339
    assert toposort_edges(EDGES_15) == (12, 13, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11)
    e2 = [(1, 4), (1, 5), (6, 8), (6, 7), (6, 9), (9, 10), (1, 0), (1, 3), (1, 2), (6, 1)]
    assert toposort_edges(e2) == (2, 3, 4, 9, 5, 0, 1, 6, 7, 8)


def test_toposort_matches_networkx_random_trees():
    rng = np.random.default_rng(0)
    for _ in range(200):
        n = int(rng.integers(2, 20))
        edges = []
        for v in range(1, n):
            edges.append((int(rng.integers(0, v)), v))
        perm = rng.permutation(n)
        edges = [(int(perm[a]), int(perm[b])) for a, b in edges]
        rng.shuffle(edges)
        assert toposort_edges(edges) == opg.toposort_edges(edges)


def test_paf_scorer_attrs():
    s = PAFScorer(FLIES13_NODES, FLIES13_EDGES, pafs_stride=8)
    assert s.n_nodes == 13 and s.n_edges == 12
    assert s.edge_inds[0] == (1, 0) and s.edge_inds[10] == (0, 11)
    assert s.sorted_edge_inds == opg.PAFScorer(FLIES13_NODES, FLIES13_EDGES, 8).sorted_edge_inds
    # max over (H, W, 2E): 0.25 * 128 * 8
    assert s.max_edge_length((1, 128, 128, 24)) == 256.0
    assert s.max_edge_length((1, 8, 8, 24)) == 0.25 * 24 * 8
    cfg = {"confmaps": {"part_names": ["a", "b"]}, "pafs": {"edges": [["a", "b"]], "output_stride": 4}}
    s2 = PAFScorer.from_config(cfg, min_line_scores=0.1)
    assert s2.pafs_stride == 4 and s2.edge_inds == [(0, 1)] and s2.min_line_scores == 0.1


def test_top_level_exports_are_lazy():
    """`import sleap_amd` must stay cheap (no torch, no HIP library); sleap.load_model / load_file / Video / Labels names."""
    import subprocess
    import sys

    code = ("import sys, sleap_amd; assert 'torch' not in sys.modules; "
            "assert sleap_amd.Video.__name__ == 'Video' and sleap_amd.Labels.__name__ == 'Labels'; "
            "assert callable(sleap_amd.load_file); assert 'torch' not in sys.modules; print('ok')")
    assert subprocess.run([sys.executable, "-c", code], capture_output=True, text=True).stdout.strip() == "ok"


def test_find_head_and_output_stride():  # ref tests/nn/test_inference.py:500-539 (on stand-ins for the Keras models)
    from types import SimpleNamespace

    from sleap_amd.nn.inference import find_head, get_model_output_stride

    m = SimpleNamespace(output_names=["A_0"], output_strides=lambda: [1])
    assert find_head(m, "A") == 0 and find_head(m, "B") is None
    assert get_model_output_stride(m) == 1
    m = SimpleNamespace(output_names=["MultiInstanceConfmapsHead_0", "PartAffinityFieldsHead_0"], output_strides=lambda: [2, 4])
    assert get_model_output_stride(m) == 4  # output_ind=-1: the last output
    assert get_model_output_stride(m, output_ind=0) == 2 and get_model_output_stride(m, output_ind=1) == 4
    assert find_head(m, "PartAffinityFieldsHead") == 1


# ---------------------------------------------------------------------------------------------------------------------------
# Predictor surface: verbosity / progress reporting (inference.py:158-173, 421-491) and make_pipeline (:329-371)
# ---------------------------------------------------------------------------------------------------------------------------
def test_verbosity_default_and_validation():
    from sleap_amd.nn.inference import BottomUpPredictor, Predictor, ProgressReporter, SingleInstancePredictor, TopDownPredictor

    import inspect

    assert Predictor.verbosity == "rich" and Predictor.report_rate == 2.0  # the reference's attrs defaults
    for cls in (BottomUpPredictor, SingleInstancePredictor, TopDownPredictor):
        assert inspect.signature(cls.__init__).parameters["verbosity"].default == "rich"
    p = Predictor()
    assert p.report_period == 0.5
    import pytest

    with pytest.raises(ValueError):
        ProgressReporter("loud", 2.0, 10)


def test_json_progress_lines_have_the_reference_fields(capsys):
    import json
    import time

    from sleap_amd.nn.inference import ProgressReporter

    with ProgressReporter("json", report_rate=1e6, n_total=12) as r:  # report every batch
        for _ in range(3):
            time.sleep(0.002)
            r.update(4)
    lines = [json.loads(l) for l in capsys.readouterr().out.strip().splitlines()]
    assert len(lines) == 3
    assert set(lines[0]) == {"n_processed", "n_total", "elapsed", "rate", "eta"}
    assert [l["n_processed"] for l in lines] == [4, 8, 12] and lines[-1]["n_total"] == 12
    assert lines[-1]["eta"] == 0 and lines[0]["rate"] > 0
    # rank != 0 (enabled=False) and "none" stay silent; "rich" renders a bar without raising
    for r in (ProgressReporter("json", 1e6, 4, enabled=False), ProgressReporter("none", 2.0, 4), ProgressReporter("rich", 2.0, 4)):
        with r:
            r.update(4)
    assert "n_processed" not in capsys.readouterr().out


def test_make_pipeline_describes_the_reference_chain():
    import numpy as np

    from sleap_amd.nn.inference import Pipeline, Predictor

    class P(Predictor):
        batch_size = 3
        is_grayscale = True
        data_config = {"preprocessing": {"resize_and_pad_to_target": True, "target_height": 16, "target_width": 20}}

    frames = np.arange(7 * 4 * 5, dtype=np.uint8).reshape(7, 4, 5, 1)
    p = P()
    pipe = p.make_pipeline(frames)
    assert isinstance(pipe, Pipeline) and p.pipeline is pipe and len(pipe) == 7
    assert [t[0] for t in pipe.transformers] == ["SizeMatcher", "Normalizer", "Batcher", "Prefetcher"]
    assert pipe.transformers[1][1] == {"ensure_float": False, "ensure_grayscale": True, "ensure_rgb": False}
    assert pipe.transformers[2][1] == {"batch_size": 3, "drop_remainder": False, "unrag": False}
    batches = list(pipe.make_dataset())
    assert [len(b["frame_ind"]) for b in batches] == [3, 3, 1]  # drop_remainder=False
    np.testing.assert_array_equal(np.concatenate([b["image"] for b in batches]), frames)
    np.testing.assert_array_equal(np.concatenate([b["frame_ind"] for b in batches]), np.arange(7))
    assert P().make_pipeline().providers == []


def test_imgconv_tiled_packer_collapses_the_three_identical_channels():
    """sa_imgconv_pack_tiled (ResNet's `tile_channels` input, resnet.py:326-362): a grayscale frame repeated three times under a
    3-channel k7 kernel == ONE K slot per tap with the channel-summed weight. Host function, no GPU: the image k-steps equal the
    CinW = 1 packing of the (float64-)summed kernel, the indicator k-steps and the bias carry sum_c w_c * mean_c."""
    import ctypes as C

    import numpy as np

    from sleap_amd import _lib

    rng = np.random.default_rng(5)
    k, cout, coutp = 7, 24, 32
    w3 = rng.normal(0, 0.1, (k, k, 3, cout)).astype(np.float32)
    scale3 = np.full(3, 1.0, np.float32)
    mean3 = np.array([123.68, 116.779, 103.939], np.float32)
    bias = rng.normal(0, 1, coutp).astype(np.float32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    for dt in _lib.DTYPES:
        h = _lib.lib(dt)
        n1 = h.sa_imgconv_packed_elems(k, 1, coutp)
        assert n1 < h.sa_imgconv_packed_elems(k, 3, coutp)
        pt, bt = np.zeros(n1, np.uint16), bias.copy()
        _lib.check(h.sa_imgconv_pack_tiled(vp(w3), k, cout, coutp, vp(scale3), vp(mean3), vp(pt), vp(bt)), "sa_imgconv_pack_tiled")
        w1 = w3.astype(np.float64).sum(axis=2, keepdims=True).astype(np.float32)
        p1, b1 = np.zeros(n1, np.uint16), bias.copy()
        _lib.check(h.sa_imgconv_pack(vp(np.ascontiguousarray(w1)), k, 1, cout, coutp, vp(scale3[:1]), None, vp(p1), vp(b1)), "sa_imgconv_pack")
        nk16, nki16 = (k * k + 15) // 16, (k * k + 15) // 16
        a = pt.reshape(coutp // 32, nk16 + nki16, 2, 64, 8)
        b = p1.reshape(coutp // 32, nk16 + nki16, 2, 64, 8)
        assert np.array_equal(a[:, :nk16], b[:, :nk16])           # image part: the summed kernel
        assert not a[:, nk16:].any() == False and not b[:, nk16:].any()  # indicator part only with means
        want = bias[:cout].astype(np.float64) - (w3.astype(np.float64) * mean3[None, None, :, None]).sum(axis=(0, 1, 2))
        assert np.allclose(bt[:cout], want, rtol=1e-6, atol=1e-4) and np.array_equal(bt[cout:], bias[cout:])
