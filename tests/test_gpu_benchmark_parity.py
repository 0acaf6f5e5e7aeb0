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
from parity_helpers import compare_with_threshold_decisions

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


# ---------------------------------------------------------------------------------------------------------------------------
# Round 3 (VERDICT r2 "next" 1b): the same comparison WITHOUT the two things that made it easy -- weights that are exactly
# fp16-representable, and frames whose peaks are all far from the 0.2 threshold.
# ---------------------------------------------------------------------------------------------------------------------------
def _float32_master(w16, seed=7):
    """A float32 master of the stored weights: every kernel / bias value moved by a seeded uniform amount inside (-0.24, 0.24)
    fp16 ulp, so that (a) none of them is fp16-representable any more and (b) rounding them to fp16 gives the stored values
    back -- exactly the relation between real (float32) SLEAP weights and what the device path makes of them. (The fit's own
    float32 master was not kept: tools/train_benchmark_model.py stored the rounded values.) The ORACLE computes with these
    float32 values; the device rounds them itself."""
    rng = np.random.default_rng(seed)
    out = {}
    for k, v in w16.items():
        v = np.asarray(v, np.float32)
        ulp = np.spacing(np.abs(v).astype(np.float16)).astype(np.float32)  # fp16 ulp at each value
        m = v + rng.uniform(-0.24, 0.24, v.shape).astype(np.float32) * ulp
        assert np.array_equal(m.astype(np.float16), v.astype(np.float16))
        out[k] = m
    n_rep = sum(int((m.astype(np.float16).astype(np.float32) == m).sum()) for m in out.values())
    assert n_rep < 1e-3 * sum(m.size for m in out.values())
    return out


def _oracle_bottomup(mc, w, frames):
    from sleap_amd.synth import FLIES13_EDGES, FLIES13_NODES

    cms, pafs = KerasGraph(mc, w)(preprocess(frames))[:2]
    pts, vals, si, ci = opf.find_local_peaks(cms, 0.2, "integral", 5)
    pts = pts * np.float32(4)
    sc = opg.PAFScorer(FLIES13_NODES, FLIES13_EDGES, 8, oob="zero")
    B = len(frames)
    ref = sc.predict(pafs, [pts[si == b] for b in range(B)], [vals[si == b] for b in range(B)], [ci[si == b] for b in range(B)])
    return ref, (pts, vals, si, ci)


def test_configs3_float32_master_weights_oracle_unrounded_device_rounds(workload):
    """configs[3] with weights that are NOT fp16-representable: the fp32 oracle computes with the float32 master, the device
    path rounds it to its storage type -- every peak within 0.5 px, same counts, same assignments, on 8 frames / 416 peaks."""
    from sleap_amd.nn.engine import DeviceNetwork
    from sleap_amd.nn.inference import BottomUpPredictor

    n = 8
    frames = workload["frames"][:n]
    w32 = _float32_master(workload["w"])
    ref, _ = _oracle_bottomup(workload["mc"], w32, frames)
    net = DeviceNetwork(workload["mc"], w32, dtype="fp16")
    pred = BottomUpPredictor(bottomup_config=workload["cfg"], bottomup_model=net, batch_size=8, verbosity="none")
    o = pred.predict(frames, make_labels=False)[0]
    assert o["n_valid"].tolist() == [len(x) for x in ref[0]] == [4] * n
    n_pk, worst = 0, 0.0
    for b in range(n):
        want = np.asarray(ref[0][b]).reshape(-1, 13, 2)
        got = o["instance_peaks"][b, : len(want)]
        assert np.array_equal(np.isnan(got), np.isnan(want)), f"frame {b}: different node assignment"
        d = np.linalg.norm(got - want, axis=-1)
        n_pk += int(np.isfinite(d).sum())
        worst = max(worst, float(np.nanmax(d)))
    print(f"float32 master weights: {n_pk} peaks, max delta {worst:.4f} px")
    assert n_pk == n * 52 and worst <= TOL_PX and worst <= 0.1, worst


# the model was fitted at noise 4, contrast 1, min_sep 170. "hard": ~4 borderline maxima per frame; "very hard": the network is
# far outside what it was fitted to and reports ~200 maxima per frame, ~90 of them within 0.05 of the threshold
HARD = {"hard": (8, dict(noise=7.0, contrast=0.75, min_sep=100.0)), "very_hard": (4, dict(noise=8.0, contrast=0.6, min_sep=110.0))}
MAP_EPS = 5e-3  # what fp16 storage moves a confidence-map value by (test_configs3_network_maps_vs_fp32_oracle: 4e-3 of ~1)


@pytest.mark.parametrize("variant", list(HARD))
def test_configs3_hard_frames_differences_are_threshold_decisions(workload, variant):
    """The "hard" variant: noisier, lower-contrast frames with animals closer than the fitted distribution, float32 master
    weights. The maps now hold borderline local maxima (a dozen per frame within 0.05 of the 0.2 threshold), i.e. the detected
    SET is decided by comparisons of nearly equal numbers and can legitimately differ between an fp32 and a 16-bit-storage
    network. What must hold, and is asserted:

      * every peak the two paths BOTH detect (same channel, nearest neighbour) agrees within 0.5 px;
      * every peak only ONE of them detects has a confidence within MAP_EPS of the threshold (a threshold decision on a map
        value that differs by the storage precision) -- nothing else may differ;
      * frames whose peak sets agree give the same instances: count, node assignment, every coordinate within 0.5 px.

    The count of frames that differ is printed (the honest number for this variant), not hidden."""
    from sleap_amd.nn.engine import DeviceNetwork
    from sleap_amd.nn.inference import BottomUpPredictor
    from sleap_amd.synth import render_animals

    n, how = HARD[variant]
    frames, _ = render_animals(n, 1024, 1024, 4, seed=400, **how)
    w32 = _float32_master(workload["w"])
    ref, (pts, vals, si, ci) = _oracle_bottomup(workload["mc"], w32, frames)
    near = int((np.abs(vals - 0.2) < 0.05).sum())
    assert near >= 8, "the variant is not hard: no oracle peak near the threshold"
    net = DeviceNetwork(workload["mc"], w32, dtype="fp16")
    pred = BottomUpPredictor(bottomup_config=workload["cfg"], bottomup_model=net, batch_size=n, verbosity="none")
    layer = pred.inference_model.bottomup_layer
    layer.return_paf_graph = True
    o = {k: v.cpu().numpy() for k, v in pred.inference_model.call_checked(torch.from_numpy(frames).cuda()).items()
         if isinstance(v, torch.Tensor)}
    assert not int(np.bitwise_or.reduce(o["status"])), "capacity overflow / non-finite status"
    g_xy, g_val, g_ch, g_n = (o[k] for k in ("peaks", "peak_vals", "peak_channel_inds", "peak_count"))
    differing, n_common, worst, n_only, n_tie = compare_with_threshold_decisions(
        (pts, vals, si, ci), (g_xy, g_val, g_ch, g_n), ref, o, n_nodes=13, map_eps=MAP_EPS, tol_px=TOL_PX)
    assert n_tie == 0  # (no maps handed over: near ties are not excused here)
    print(f"{variant} variant ({n} frames): {n_common} common peaks (max delta {worst:.4f} px), {near} oracle peaks within 0.05 of the threshold, "
          f"{n_only} peaks detected by one path only (all within {MAP_EPS} of the threshold), frames that differ: {differing}")
    assert worst <= TOL_PX
