"""BASELINE.json's full frame size (1024x1024, benchmark model of bench.py): the oracle still finishes two frames in seconds,
so they are compared directly; larger batches are covered through size-independent properties of the path -- every frame
is processed independently (batch-composition invariance, permutation equivariance, bitwise) and the post-processing on
the device's own network outputs equals the oracle's exactly."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

H = W = 1024


@pytest.fixture(scope="module")
def setup():
    from sleap_amd.benchmark_model import build_benchmark_predictor
    from sleap_amd.synth import render_flies

    pred, mc, weights = build_benchmark_predictor(H, W, batch_size=8, seed=0)  # the fitted benchmark model
    pred.verbosity = "none"
    frames, _ = render_flies(16, H, W, n_animals=4, seed=100)
    return pred, mc, weights, frames


def _flat(outs):
    keys = ("instance_peaks", "instance_peak_vals", "instance_scores")
    imax = max(ex["instance_scores"].shape[1] for ex in outs)
    res = {}
    for k in keys:
        res[k] = np.concatenate([np.pad(ex[k], [(0, 0), (0, imax - ex[k].shape[1])] + [(0, 0)] * (ex[k].ndim - 2),
                                        constant_values=np.nan) for ex in outs])
    res["n_valid"] = np.concatenate([ex["n_valid"] for ex in outs])
    return res


def test_full_size_postprocessing_equals_oracle_on_device_maps(setup):
    """Peaks, PAF scoring, matching and grouping at 256x256x13 / 128x128x24 maps: instance membership identical,
    coordinates within 1e-4 px of the oracle run on the same (device-produced) maps."""
    from oracle import paf_grouping as opg
    from oracle import peak_finding as opf
    from sleap_amd.synth import FLIES13_EDGES, FLIES13_NODES

    pred, _, _, frames = setup
    n = 4
    out = pred.predict(frames[:n], make_labels=False)[0]
    layer = pred.inference_model.bottomup_layer
    cms, pafs, _ = layer.forward_pass(frames[:n])
    cms, pafs = cms.cpu().numpy(), pafs.cpu().numpy()
    assert cms.shape == (n, 256, 256, 13) and pafs.shape == (n, 128, 128, 24)
    pts, vals, si, ci = opf.find_local_peaks(cms, 0.2, "integral", 5)
    pts = pts * np.float32(4)
    sc = opg.PAFScorer(FLIES13_NODES, FLIES13_EDGES, 8, oob="zero")
    o = sc.predict(pafs, [pts[si == b] for b in range(n)], [vals[si == b] for b in range(n)], [ci[si == b] for b in range(n)])
    total = 0
    for b in range(n):
        k = int(out["n_valid"][b])
        assert k == o[0][b].shape[0]
        np.testing.assert_allclose(out["instance_peaks"][b, :k], o[0][b], atol=1e-4, equal_nan=True)
        np.testing.assert_allclose(out["instance_scores"][b, :k], o[2][b], rtol=1e-4, atol=1e-5)
        total += int(np.isfinite(o[0][b][..., 0]).sum())
    assert total == n * 4 * 13 and len(pts) >= 52 * n  # 4 complete animals per frame


def test_full_size_network_vs_fp32_oracle(setup):
    """16-bit-storage MFMA network vs the fp32 CPU oracle at 1024x1024: heads within 3 % of their range (the bf16 bound;
    the default fp16 build is held to 4e-3 in tests/test_gpu_benchmark_parity.py)."""
    from oracle.keras_graph import KerasGraph, preprocess

    pred, mc, weights, frames = setup
    layer = pred.inference_model.bottomup_layer
    cms, pafs, _ = layer.forward_pass(frames[:2])
    ref = KerasGraph(mc, weights)(preprocess(frames[:2]))
    for got, want in zip((cms.cpu().numpy(), pafs.cpu().numpy()), ref[:2]):
        assert got.shape == want.shape
        assert float(np.abs(got - want).max() / np.abs(want).max()) < 3e-2


def test_frames_are_independent_bitwise(setup):
    """Size-independent property of the sharded path: a frame's result does not depend on its batch. 16 frames in batches of
    8 == batches of 4 == any permutation of the frames (bitwise), which is what makes frame sharding across GPUs exact."""
    pred, _, _, frames = setup
    base = _flat(pred.predict(frames, make_labels=False))
    pred.batch_size = 4
    four = _flat(pred.predict(frames, make_labels=False))
    perm = np.random.default_rng(0).permutation(len(frames))
    shuf = _flat(pred.predict(frames[perm], make_labels=False))
    pred.batch_size = 8
    for k in base:
        np.testing.assert_array_equal(base[k], four[k])
        np.testing.assert_array_equal(base[k][perm], shuf[k])
    assert base["n_valid"].tolist() == [4] * 16
