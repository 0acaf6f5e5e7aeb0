"""Post-processing parity at the SHAPE of BASELINE configs[4] on IDENTICAL maps: 24-node / 23-edge mouse skeleton, 8 animals per
1024 x 1024 frame -> confidence maps 256 x 256 x 24 (stride 4), part-affinity fields 128 x 128 x 46 (stride 8). This separates
"is the matcher / grouper right at E = 23 with 8 peaks per node type" (here: every table exact) from "is the ResNet close
enough" (tests/test_gpu_config_parity.py).

Reference: sleap/nn/paf_grouping.py:553-670 (match_candidates_*), 799-1112 (group_instances_*), 1629-1705 (PAFScorer.predict),
peak_finding.py:451-532. Two sources of maps:

  * analytic (SURVEY.md 8(d) mode 1): `make_multi_confmaps` sigma 2.5 / `make_multi_pafs` sigma 75 of seeded mouse instances
    + N(0, 0.01) noise -- known ground truth: 8 complete 24-node instances per frame;
  * the device's own ResNet-50 outputs (the configs[4] task model), handed to the oracle's post-processing.
"""
import numpy as np
import pytest
import torch
from numpy.testing import assert_allclose, assert_array_equal

from oracle import paf_grouping as opg
from oracle import peak_finding as opf
from oracle.synth import synth_bottomup_maps

N_NODES, N_EDGES, N_ANIMALS, SIZE = 24, 23, 8, 1024


def _mouse():
    from sleap_amd.synth import MOUSE24

    return MOUSE24


def analytic_batch(seed, B=2, noise=0.01, body=(60.0, 90.0), min_sep=200.0):
    """-> cms (B, 256, 256, 24), pafs (B, 128, 128, 46), instances list of (8, 24, 2)."""
    from sleap_amd.synth import random_instances

    sk = _mouse()
    rng = np.random.default_rng(seed)
    cms, pafs, insts = [], [], []
    for _ in range(B):
        inst = random_instances(rng, N_ANIMALS, SIZE, SIZE, body=body, margin=128.0, min_sep=min_sep, template=sk.template)
        assert inst.shape == (N_ANIMALS, N_NODES, 2)
        c, p, _ = synth_bottomup_maps(inst, SIZE, SIZE, node_names=sk.nodes, edges=sk.edges, noise=noise, rng=rng)
        cms.append(c)
        pafs.append(p)
        insts.append(inst)
    return np.stack(cms), np.stack(pafs), insts


def oracle_postproc(cms, pafs, nodes, edges, threshold=0.2):
    pts, vals, si, ci = opf.find_local_peaks(cms, threshold, "integral", 5)
    pts = pts * np.float32(4)
    B = cms.shape[0]
    sc = opg.PAFScorer(nodes, edges, 8, oob="zero")
    peaks = [pts[si == b] for b in range(B)]
    pv = [vals[si == b] for b in range(B)]
    pc = [ci[si == b] for b in range(B)]
    return sc.predict(pafs, peaks, pv, pc), (peaks, pv, pc)


def _assert_truth_recovered(instances, truth, tol=1.5):
    """every rendered animal comes back as ONE complete instance, each node within `tol` px (sigma 2.5 maps + noise)."""
    inst = np.asarray(instances).reshape(-1, N_NODES, 2)
    assert inst.shape[0] == len(truth) and not np.isnan(inst).any()
    for gt in truth:
        d = np.linalg.norm(inst - gt[None], axis=-1).max(axis=1)
        assert d.min() <= tol, d.min()


def test_oracle_groups_eight_complete_mice_at_the_configs4_shape():
    """(CPU) the oracle itself at E = 23 / 8 animals: the analytic maps give back the 8 animals, complete."""
    cms, pafs, insts = analytic_batch(41, B=1)
    assert cms.shape == (1, 256, 256, N_NODES) and pafs.shape == (1, 128, 128, 2 * N_EDGES)
    sk = _mouse()
    o, (peaks, _, pc) = oracle_postproc(cms, pafs, sk.nodes, sk.edges)
    assert len(peaks[0]) == N_ANIMALS * N_NODES and np.bincount(pc[0]).tolist() == [N_ANIMALS] * N_NODES
    _assert_truth_recovered(o[0][0], insts[0])


@pytest.mark.gpu
@pytest.mark.parametrize("seed,noise", [(42, 0.01), (43, 0.01), (44, 0.03)])
def test_mouse24_postproc_on_analytic_maps_every_table_exact(seed, noise):
    """Device post-processing (peaks -> scoring -> matching -> grouping, the separate-stage form with the connection graph
    returned) vs the oracle on identical 256 x 256 x 24 / 128 x 128 x 46 maps: candidate order, edge / peak indices, match
    results, instance membership and NaN masks EXACT; coordinates and scores to float32 summation order."""
    from sleap_amd import ops
    from sleap_amd.nn import paf_grouping as pg

    sk = _mouse()
    cms, pafs, insts = analytic_batch(seed, B=2, noise=noise)
    B = cms.shape[0]
    o, (peaks, pv, pc) = oracle_postproc(cms, pafs, sk.nodes, sk.edges)
    dcms, dpafs = ops.to_cuda_f32(cms), ops.to_cuda_f32(pafs)
    pxy, pval, pch, pcnt, st = ops.find_local_peaks(dcms, None, 0.2, "integral", 5, 4.0, 1024)
    sc = pg.PAFScorer(sk.nodes, sk.edges, 8)
    inst, ivals, iscores, n_inst, st, graph = sc.predict_padded(dpafs, pxy, pval, pch, pcnt, st, return_graph=True)
    assert int(st.max().item()) & ~16 == 0
    n = n_inst.cpu().numpy()
    ei, epi, ls = sc._graph_to_ragged(graph, B)
    inst, ivals, iscores = inst.cpu().numpy(), ivals.cpu().numpy(), iscores.cpu().numpy()
    for b in range(B):
        assert int(pcnt[b]) == len(peaks[b])
        assert_array_equal(pch[b, : len(peaks[b])].cpu().numpy(), pc[b])
        assert_allclose(pxy[b, : len(peaks[b])].cpu().numpy(), peaks[b], atol=4e-5)
        assert n[b] == o[0][b].shape[0]
        assert_array_equal(np.isnan(inst[b, : n[b]]), np.isnan(o[0][b]))
        assert_allclose(inst[b, : n[b]], o[0][b], atol=4e-5, equal_nan=True)
        assert_array_equal(ivals[b, : n[b]], o[1][b])
        assert_allclose(iscores[b, : n[b]], o[2][b], atol=2e-5)
        assert_array_equal(ei[b], o[3][b])
        assert_array_equal(epi[b], o[4][b])
        assert_allclose(ls[b], o[5][b], atol=1e-5)
        assert np.isnan(inst[b, n[b]:]).all()
        if noise <= 0.01:
            _assert_truth_recovered(inst[b, : n[b]], insts[b])
    # the hot-path form (one fused launch after the scan) gives the same bits as the separate stages
    f = sc.predict_from_maps(dcms, None, dpafs, 0.2, "integral", 5, 4, 1024)
    assert_array_equal(f["n_instances"].cpu().numpy(), n)
    for b in range(B):  # (rows past n_instances are unspecified in the fused form's score buffer)
        assert_array_equal(f["instance_peaks"][b, : n[b]].cpu().numpy().view(np.uint32), inst[b, : n[b]].view(np.uint32))
        assert_array_equal(f["instance_scores"][b, : n[b]].cpu().numpy().view(np.uint32), iscores[b, : n[b]].view(np.uint32))


@pytest.mark.gpu
def test_mouse24_postproc_on_crowded_maps_exact():
    """Animals closer together than their own length (min_sep 90 px for 100-155 px long mice): PAFs of neighbours overlap,
    several candidates per edge score above the cut -- the Hungarian step decides. Still every table equal to the oracle's."""
    from sleap_amd import ops
    from sleap_amd.nn import paf_grouping as pg

    sk = _mouse()
    cms, pafs, _ = analytic_batch(45, B=2, noise=0.02, min_sep=90.0)
    o, _ = oracle_postproc(cms, pafs, sk.nodes, sk.edges)
    sc = pg.PAFScorer(sk.nodes, sk.edges, 8)
    f = sc.predict_from_maps(ops.to_cuda_f32(cms), None, ops.to_cuda_f32(pafs), 0.2, "integral", 5, 4, 1024)
    assert int(f["status"].max().item()) & ~16 == 0
    n = f["n_instances"].cpu().numpy()
    for b in range(2):
        assert n[b] == o[0][b].shape[0]
        got = f["instance_peaks"][b, : n[b]].cpu().numpy()
        assert_array_equal(np.isnan(got), np.isnan(o[0][b]))
        assert_allclose(got, o[0][b], atol=4e-5, equal_nan=True)
        assert_allclose(f["instance_scores"][b, : n[b]].cpu().numpy(), o[2][b], atol=2e-5)


@pytest.mark.gpu
def test_mouse24_postproc_on_the_devices_own_resnet_maps_equals_oracle():
    """The configs[4] network's OWN outputs (ResNet-50 + UpsamplingStack task model, fp16 storage) handed to the oracle's
    post-processing: same instance count, same node assignment (NaN mask), coordinates to float32 summation order -- the
    device's peak finder / scorer / matcher / grouper at 24 nodes / 23 edges / 8 animals on real network maps."""
    import config_models as C
    from sleap_amd.nn.engine import DeviceNetwork
    from sleap_amd.nn.inference import BottomUpPredictor

    task, n = "c4_resnet", 3
    sk = C.skeleton(task)
    frames, _ = C.render(task, n, seed=304)
    mc, w = C.load_task_weights(task, SIZE, SIZE)
    net = DeviceNetwork(mc, w, dtype="fp16")
    pred = BottomUpPredictor(bottomup_config=C.training_config(task), bottomup_model=net, batch_size=n, verbosity="none")
    layer = pred.inference_model.bottomup_layer
    out = {k: v.cpu().numpy() for k, v in pred.inference_model.call_checked(torch.from_numpy(frames).cuda()).items()
           if isinstance(v, torch.Tensor)}
    cms, pafs, _ = layer.forward_pass(torch.from_numpy(frames).cuda())
    cms, pafs = cms.cpu().numpy(), pafs.cpu().numpy()
    assert cms.shape == (n, 256, 256, N_NODES) and pafs.shape == (n, 128, 128, 2 * N_EDGES)
    o, (peaks, _, _) = oracle_postproc(cms, pafs, sk.nodes, sk.edges)
    total = 0
    for b in range(n):
        k = int(out["n_valid"][b])
        assert k == o[0][b].shape[0], (b, k, o[0][b].shape)
        got = out["instance_peaks"][b, :k]
        assert_array_equal(np.isnan(got), np.isnan(o[0][b]))
        assert_allclose(got, o[0][b], atol=2e-4, equal_nan=True)
        assert_allclose(out["instance_scores"][b, :k], o[2][b], rtol=1e-4, atol=2e-5)
        total += int(np.isfinite(o[0][b][..., 0]).sum())
    assert total >= n * N_ANIMALS * 20  # the task model detects (nearly) every node of every animal
