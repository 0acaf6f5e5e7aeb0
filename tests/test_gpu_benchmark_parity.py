"""BASELINE configs[3] end to end, on a network that DETECTS what is in the frames: bottom-up UNet `baseline_medium_rf` +
PAFs, 1024 x 1024 x 1 uint8, flies13 (13 nodes / 12 edges), 4 animals per frame, weights fitted to the synthetic fly video
(sleap_amd/data/benchmark_unet_flies13.npz, tools/train_benchmark_model.py). Device path (16-bit storage MFMA network +
fused post-processing) vs the fp32 CPU oracle (torch-CPU Keras graph + restated peak finding / PAF grouping) on the SAME
uint8 frames with the SAME weights, compared POSITIONALLY as SURVEY.md 8(d) prescribes:

    same number of instances per frame, same NaN mask, max ||delta(x, y)|| <= 0.5 px over EVERY non-NaN peak

-- north_star's tolerance, asserted at 100 %, not at a percentage. The frames are rendered with seeds the model was not
fitted to. >= 16 frames -> >= 800 peaks."""
import numpy as np
import pytest
import torch

from oracle import paf_grouping as opg
from oracle import peak_finding as opf
from oracle.keras_graph import KerasGraph, preprocess

pytestmark = pytest.mark.gpu

TOL_PX = 0.5
N_FRAMES = 16


@pytest.fixture(scope="module")
def workload():
    from sleap_amd.benchmark_model import build_benchmark_graph, load_trained_weights
    from sleap_amd.synth import FLIES13_EDGES, FLIES13_NODES, render_flies

    frames, insts = render_flies(N_FRAMES, 1024, 1024, n_animals=4, seed=200)
    cfg, mc, _ = build_benchmark_graph(1024, 1024)
    w = load_trained_weights()
    cms, pafs = KerasGraph(mc, w)(preprocess(frames))[:2]
    pts, vals, si, ci = opf.find_local_peaks(cms, 0.2, "integral", 5)
    pts = pts * np.float32(4)
    sc = opg.PAFScorer(FLIES13_NODES, FLIES13_EDGES, 8, oob="zero")
    B = len(frames)
    ref = sc.predict(pafs, [pts[si == b] for b in range(B)], [vals[si == b] for b in range(B)], [ci[si == b] for b in range(B)])
    return dict(frames=frames, insts=insts, cfg=cfg, mc=mc, w=w, ref=ref, cms=cms, pafs=pafs, n_peaks=[int((si == b).sum()) for b in range(B)])


def test_oracle_sees_four_complete_instances_per_frame(workload):
    """The workload is what configs[3] names: 4 animals x 13 nodes in every frame, for the ORACLE (so the comparison below is
    about 832 real peaks, not about an empty set), close to the rendered ground truth."""
    ref, insts = workload["ref"], workload["insts"]
    assert [len(x) for x in ref[0]] == [4] * N_FRAMES
    assert all(n >= 52 for n in workload["n_peaks"]) and sum(workload["n_peaks"]) <= 52 * N_FRAMES + 2, workload["n_peaks"]
    err = []
    for b in range(N_FRAMES):
        pred = np.asarray(ref[0][b]).reshape(-1, 13, 2)
        assert not np.isnan(pred).any()
        for gt in insts[b]:
            d = np.linalg.norm(pred - gt[None], axis=-1).mean(axis=1)
            err.append(np.linalg.norm(pred[int(d.argmin())] - gt, axis=-1))
    err = np.concatenate(err)
    assert err.mean() < 1.0 and err.max() < 6.0, (err.mean(), err.max())


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_configs3_end_to_end_every_peak_within_half_a_pixel(workload, dtype):
    from sleap_amd.nn.engine import DeviceNetwork
    from sleap_amd.nn.inference import BottomUpPredictor

    net = DeviceNetwork(workload["mc"], workload["w"], dtype=dtype)
    pred = BottomUpPredictor(bottomup_config=workload["cfg"], bottomup_model=net, batch_size=8, verbosity="none")
    outs = pred.predict(workload["frames"], make_labels=False)
    got_n = np.concatenate([o["n_valid"] for o in outs])
    ref = workload["ref"]
    assert got_n.tolist() == [len(x) for x in ref[0]]
    n_pk, worst, worst_val, worst_score = 0, 0.0, 0.0, 0.0
    f = 0
    for o in outs:
        for b in range(len(o["n_valid"])):
            want = np.asarray(ref[0][f]).reshape(-1, 13, 2)
            got = o["instance_peaks"][b, : len(want)]
            assert np.array_equal(np.isnan(got), np.isnan(want)), f"frame {f}: different node assignment"
            d = np.linalg.norm(got - want, axis=-1)
            n_pk += int(np.isfinite(d).sum())
            worst = max(worst, float(np.nanmax(d)))
            worst_val = max(worst_val, float(np.nanmax(np.abs(o["instance_peak_vals"][b, : len(want)] - np.asarray(ref[1][f])))))
            worst_score = max(worst_score, float(np.abs(o["instance_scores"][b, : len(want)] - np.asarray(ref[2][f])).max()))
            f += 1
    print(f"{dtype}: {n_pk} peaks, max delta {worst:.4f} px, max |peak value delta| {worst_val:.5f}, max |instance score delta| {worst_score:.5f}")
    assert n_pk == N_FRAMES * 52
    if dtype == "fp16":
        # the default storage type, the one the bench line is quoted on: north_star's tolerance on EVERY peak, with margin
        # (measured: 0.009 px over 832 peaks, profiles/r02_parity.md)
        assert worst <= TOL_PX and worst <= 0.1, (dtype, worst)
    else:
        # bf16 storage (8 mantissa bits, the fallback for networks that overflow fp16): identical instances and assignments
        # (asserted above); coordinates within 1 px -- measured 0.64 px on the worst of 832 peaks, i.e. it does NOT meet the
        # 0.5 px tolerance everywhere, which is why it is not the default
        assert worst <= 1.0, (dtype, worst)


def test_configs3_network_maps_vs_fp32_oracle(workload):
    """The network half in isolation at full size: confidence maps and PAFs of the fp16-storage device path within 4e-3 of the
    fp32 oracle's range (the 1e-3 level that leaves the decisions above untouched)."""
    from sleap_amd.nn.engine import DeviceNetwork

    net = DeviceNetwork(workload["mc"], workload["w"], dtype="fp16")
    outs = net.forward(torch.from_numpy(workload["frames"][:4]).cuda())
    for o, r in zip(outs, (workload["cms"][:4], workload["pafs"][:4])):
        o = o.cpu().numpy()
        assert np.isfinite(o).all()
        assert float(np.abs(o - r).max()) <= 4e-3 * float(np.abs(r).max())
