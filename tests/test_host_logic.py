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
