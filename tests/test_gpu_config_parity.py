"""End-to-end parity at the NAMED SHAPES of BASELINE.json configs[0], [1], [2] and [4] (configs[3] is
tests/test_gpu_benchmark_parity.py): the device path (16-bit-storage MFMA network + device post-processing, through the
reference-shaped predictors) against the fp32 CPU oracle running ITS OWN networks (torch-CPU Keras graph + restated peak
finding / cropping / PAF grouping) on the SAME uint8 frames with the SAME float32 weights, compared POSITIONALLY as SURVEY.md
8(d) prescribes: same number of instances per frame, same NaN mask, every peak within north_star's 0.5 px.

The networks are the architectures of the reference's shipped training profiles, fitted to the synthetic videos of
`sleap_amd.synth.render_animals` (tests/config_models.py, tools/train_config_models.py) and stored as float32 masters: the
oracle computes with the fp32 values, the device rounds them to fp16 itself -- as it would real SLEAP weights. Frames are
rendered with seeds the models were not fitted to.

    configs[0]  256 x 256, 5 nodes: UNet (baseline.centroid profile, input x0.5) through SingleInstanceInferenceLayer /
                find_global_peaks (inference.py:1319-1380), and the centroid model through CentroidCrop / find_local_peaks
    configs[1]  512 x 512, 13 nodes, batch 32: single-instance UNet (baseline_medium_rf.single)
    configs[2]  1024 x 1024 top-down, 2 animals: centroid UNet (x0.5) -> crops -> centered-instance UNet (f24)
                (inference.py:1747-1966, 2059-2200)
    configs[4]  1024 x 1024 bottom-up ResNet-50 + UpsamplingStack + PAFs, 24 nodes / 23 edges, 8 animals (resnet.py:467-541)
"""
import os

import numpy as np
import pytest
import torch

from oracle import inference as oinf
from oracle import paf_grouping as opg
from oracle import peak_finding as opf
from oracle.keras_graph import KerasGraph, preprocess

pytestmark = pytest.mark.gpu

TOL_PX = 0.5


def _net(task, h, w, dtype="fp16"):
    import config_models as C
    from sleap_amd.nn.engine import DeviceNetwork

    mc, wts = C.load_task_weights(task, h, w)
    return DeviceNetwork(mc, wts, dtype=dtype), mc, wts


def _compare(got, want, what):
    """positional comparison of two (.., 2) point arrays -> (n finite, worst distance)."""
    got, want = np.asarray(got, np.float32), np.asarray(want, np.float32)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert np.array_equal(np.isnan(got), np.isnan(want)), f"{what}: different NaN mask (a different set of detected nodes)"
    d = np.linalg.norm(got - want, axis=-1)
    ok = np.isfinite(d)
    return int(ok.sum()), (float(d[ok].max()) if ok.any() else 0.0)


def _single_instance_case(task, n_frames, batch, seed, margin_px=0.1):
    import config_models as C
    from sleap_amd.nn.inference import SingleInstancePredictor

    t = C.TASKS[task]
    frames, insts = C.render(task, n_frames, seed)
    x = preprocess(frames, input_scale=t["input_scale"], pad_stride=t["unet"][2] if "unet" in t else t["hourglass"]["max_stride"])
    net, mc, wts = _net(task, x.shape[1], x.shape[2])
    cms = KerasGraph(mc, wts)(x)[0]
    want, want_vals = oinf.single_instance_peaks(cms, None, 0.2, "integral", 5, t["heads"][0][2], t["input_scale"])
    # the oracle detects every node of the one animal, close to the rendered truth (the comparison is about real peaks)
    assert not np.isnan(want).any() and float(want_vals.min()) > 0.4
    truth = np.stack([a[0] for a in insts])
    assert float(np.linalg.norm(want[:, 0] - truth, axis=-1).mean()) < 2.0
    pred = SingleInstancePredictor(confmap_config=C.training_config(task), confmap_model=net, batch_size=batch, verbosity="none")
    outs = pred.predict(frames, make_labels=False)
    assert len(outs) == -(-n_frames // batch)
    got = np.concatenate([o["instance_peaks"] for o in outs])
    got_vals = np.concatenate([o["instance_peak_vals"] for o in outs])
    n, worst = _compare(got, want, task)
    print(f"{task}: {n} peaks, max delta {worst:.4f} px, max |peak value delta| {np.abs(got_vals - want_vals).max():.5f}")
    assert n == n_frames * len(C.skeleton(task).nodes)
    assert worst <= TOL_PX and worst <= margin_px, worst  # measured ~0.01 px (UNets): the tolerance with a wide margin
    assert float(np.abs(got_vals - want_vals).max()) <= 5e-3


def test_configs0_single_instance_unet_256_5_nodes():
    """configs[0] read as SURVEY.md 8(d) reads it: the centroid-profile UNet (f16 r2 s16 -> 2, input_scaling 0.5) on 256 x 256
    frames, 5 nodes, through SingleInstanceInferenceLayer (global peak per node, integral refinement, un-scaling + 0.5)."""
    _single_instance_case("c0_single5", 16, 4, seed=300)


def _two_stack_hourglass(h, w):
    """A TWO-stack hourglass with per-stack heads named `<head>_<s>` (older SLEAP naming, the reference's own fixtures): stem,
    stack 0 and `SingleInstanceConfmapsHead_0` carry the FITTED one-stack weights (stack 0 sees exactly what the one-stack model
    sees), stack 1 and `..._1` seeded He-normal values (BatchNormalization neutral)."""
    import config_models as C
    from sleap_amd.nn import architectures as A

    t = C.TASKS["hg_single13"]
    hg = dict(t["hourglass"], stacks=2)
    mc, shapes = A.build_hourglass_model_config((h, w, 1), heads=t["heads"], legacy_head_suffix=True, **hg)
    _, fitted = C.load_task_weights("hg_single13", h, w)
    wts = A.he_normal_weights(shapes, seed=11)
    for k in wts:
        if k.endswith(("/gamma", "/moving_variance")):
            wts[k][:] = 1.0
        elif k.endswith(("/beta", "/moving_mean")):
            wts[k][:] = 0.0
    head = t["heads"][0][0]
    for k, v in fitted.items():
        k2 = k.replace(head + "/", head + "_0/")
        assert k2 in wts and wts[k2].shape == v.shape, k
        wts[k2] = np.asarray(v, np.float32)
    return mc, wts


def test_two_stack_hourglass_end_to_end_uses_stack_0_head():
    """SURVEY.md 8(a) a2' caveat (stacks > 1 were features-only): a two-stack hourglass with per-stack heads through
    SingleInstancePredictor. `find_head` returns the first output whose name contains the head type (inference.py:1223-1226,
    2885-2888) -- stack 0's head -- on both sides; the device computes BOTH stacks and both heads. Peaks vs the fp32 oracle
    positionally within 0.5 px; the second stack's maps (a seeded stack on fitted features) within 1e-2 of their range."""
    import config_models as C
    from sleap_amd.nn.engine import DeviceNetwork
    from sleap_amd.nn.inference import SingleInstancePredictor

    n = 8
    frames, insts = C.render("hg_single13", n, seed=306)
    x = preprocess(frames, pad_stride=32)
    mc, wts = _two_stack_hourglass(x.shape[1], x.shape[2])
    ref = KerasGraph(mc, wts)(x)
    want, want_vals = oinf.single_instance_peaks(ref[0], None, 0.2, "integral", 5, 4, 1.0)
    assert not np.isnan(want).any() and float(want_vals.min()) > 0.4
    net = DeviceNetwork(mc, wts, dtype="fp16")
    assert net.output_names == ["SingleInstanceConfmapsHead_0", "SingleInstanceConfmapsHead_1"]
    pred = SingleInstancePredictor(confmap_config=C.training_config("hg_single13"), confmap_model=net, batch_size=4, verbosity="none")
    assert pred.inference_model.single_instance_layer.confmaps_ind == 0
    outs = pred.predict(frames, make_labels=False)
    got = np.concatenate([o["instance_peaks"] for o in outs])
    npk, worst = _compare(got, want, "two-stack hourglass")
    print(f"two-stack hourglass: {npk} peaks, max delta {worst:.4f} px")
    assert npk == n * 13 and worst <= TOL_PX and worst <= 0.25
    dev = [o.cpu().numpy() for o in net.forward(torch.from_numpy(frames[:2]).cuda())]
    for d, r in zip(dev, ref):
        assert np.isfinite(d).all()
        assert float(np.abs(d - r[:2]).max()) <= 1e-2 * float(np.abs(r[:2]).max())


def test_configs1_single_instance_unet_512_13_nodes_batch_32():
    """configs[1]: baseline_medium_rf.single (UNet f16 r2 s16 -> 2) on 512 x 512 frames, 13-node fly, ONE batch of 32."""
    _single_instance_case("c1_single13", 32, 32, seed=301)


def test_hourglass_single_instance_end_to_end_512_13_nodes():
    """SURVEY.md 8(a) row a2' end to end (not a BASELINE config): a one-stack hourglass of the reference's structure
    (hourglass.py:17-316: k7 s2 stem + pooling, Conv -> ReLU -> BatchNormalization, nearest-neighbour upsampling with additive
    skips; a quarter of the default width), fitted to the fly video, through SingleInstanceInferenceLayer on 512 x 512 frames:
    the fp32 oracle runs its own network on the float32 master weights, the device path its fp16 one -- same NaN mask, every
    peak within 0.5 px."""
    _single_instance_case("hg_single13", 16, 8, seed=305, margin_px=0.25)  # measured 0.125 px on the worst of 208 peaks


def _topdown_oracle(frames, crop_size):
    import config_models as C

    x = preprocess(frames, input_scale=0.5, pad_stride=16)
    mc, w = C.load_task_weights("c2_centroid", x.shape[1], x.shape[2])
    cms = KerasGraph(mc, w)(x)[0]
    cc = oinf.centroid_crop(frames, cms, None, 0.2, "integral", 5, 2, 0.5, crop_size)
    mc2, w2 = C.load_task_weights("c2_centered", crop_size, crop_size)
    cm2 = KerasGraph(mc2, w2)(preprocess(cc["crops"]))[0]
    pk, vals = oinf.find_instance_peaks(cm2, None, cc["crop_offsets"], 0.2, "integral", 5, 4, 1.0)
    return cc, pk, vals


def test_configs0_centroid_model_through_centroid_crop_256():
    """configs[0]'s other reading ("centroid model"): the centroid UNet (input x0.5) through CentroidCrop / find_local_peaks on
    256 x 256 frames with one animal: same centroids (count, order, <= 0.5 px) and the same crops up to the resampling of a
    sub-0.01-px centroid difference."""
    import config_models as C
    from sleap_amd.nn.inference import CentroidCrop

    frames, insts = C.render("c0_single5", 8, seed=302, skeleton=C.skeleton("c2_centroid"))  # a whole 13-node fly per 256 x 256 frame
    x = preprocess(frames, input_scale=0.5, pad_stride=16)
    net, mc, w = _net("c2_centroid", x.shape[1], x.shape[2])
    cms = KerasGraph(mc, w)(x)[0]
    want = oinf.centroid_crop(frames, cms, None, 0.2, "integral", 5, 2, 0.5, 160)
    assert np.bincount(want["crop_sample_inds"], minlength=8).tolist() == [1] * 8
    layer = CentroidCrop(net, crop_size=160, input_scale=0.5, pad_to_stride=16, output_stride=2, peak_threshold=0.2,
                         refinement="integral", integral_patch_size=5)
    got = layer(frames)
    g = {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in got.items()}
    assert g["crop_sample_inds"].tolist() == want["crop_sample_inds"].tolist()
    n, worst = _compare(g["centroids"], want["centroids"], "centroids")
    print(f"configs[0] centroid path: {n} centroids, max delta {worst:.4f} px")
    assert worst <= TOL_PX and worst <= 0.1
    assert float(np.abs(g["centroid_vals"] - want["centroid_vals"]).max()) <= 5e-3
    assert g["crops"].shape == want["crops"].shape and g["crops"].dtype == np.uint8
    # uint8 crops resampled at centroids that differ by < 0.1 px: grey levels within a few counts, most of them identical
    diff = np.abs(g["crops"].astype(np.int16) - want["crops"].astype(np.int16))
    assert diff.max() <= 24 and float((diff <= 1).mean()) > 0.9, (diff.max(), float((diff <= 1).mean()))


@pytest.mark.parametrize("seed", [303, 313, 323])
def test_configs2_topdown_1024_two_animals_oracle_runs_its_own_networks(seed):
    """(three seeds since round 4: the 0.42 px worst case of seed 303 is one flat node, not the rule -- VERDICT r3 weak #3)
    configs[2]: centroid UNet (baseline.centroid, x0.5) + centered-instance UNet (baseline_medium_rf.topdown, f24) on
    160 x 160 crops, 1024 x 1024 frames with 2 animals. The ORACLE runs both of its own networks (centroids from its fp32
    centroid maps, crops at its own centroids, peaks from its fp32 crop maps) -- nothing of the device path enters it."""
    import config_models as C
    from sleap_amd.nn.engine import DeviceNetwork
    from sleap_amd.nn.inference import TopDownPredictor

    n_frames, crop = 8, C.TASKS["c2_centered"]["crop"]
    frames, insts = C.render("c2_centroid", n_frames, seed=seed)
    cc, want, want_vals = _topdown_oracle(frames, crop)
    counts = np.bincount(cc["crop_sample_inds"], minlength=n_frames)
    assert counts.tolist() == [2] * n_frames, counts  # the oracle finds both animals in every frame ...
    assert not np.isnan(want).any() and float(want_vals.min()) > 0.25  # ... and all 13 nodes of each (threshold 0.2)
    truth = np.concatenate(insts)
    j = np.linalg.norm(cc["centroids"][:, None] - truth[None, :, C.ANCHOR], axis=-1).argmin(axis=1)
    assert float(np.linalg.norm(want - truth[j], axis=-1).mean()) < 2.0
    mc_c, w_c = C.load_task_weights("c2_centroid", 512, 512)
    mc_i, w_i = C.load_task_weights("c2_centered", crop, crop)
    pred = TopDownPredictor(centroid_config=C.training_config("c2_centroid"), centroid_model=DeviceNetwork(mc_c, w_c),
                            confmap_config=C.training_config("c2_centered"), confmap_model=DeviceNetwork(mc_i, w_i),
                            batch_size=4, verbosity="none")
    outs = pred.predict(frames, make_labels=False)
    nv = np.concatenate([o["n_valid"] for o in outs])
    assert nv.tolist() == counts.tolist()
    got = np.concatenate([o["instance_peaks"][:, :2] for o in outs]).reshape(-1, 13, 2)
    got_vals = np.concatenate([o["instance_peak_vals"][:, :2] for o in outs]).reshape(-1, 13)
    got_c = np.concatenate([o["centroids"][:, :2] for o in outs]).reshape(-1, 2)
    n, worst = _compare(got, want, "instance peaks")
    _, worst_c = _compare(got_c, cc["centroids"], "centroids")
    dist = np.linalg.norm(got - want, axis=-1).ravel()
    print(f"configs[2] seed {seed}: {n} peaks, max delta {worst:.4f} px (95th percentile {np.percentile(dist, 95):.4f}); centroids max delta "
          f"{worst_c:.4f} px; max |peak value delta| {np.abs(got_vals - want_vals).max():.5f}")
    assert n == n_frames * 2 * 13
    # every peak inside north_star's tolerance; all but a few far inside it. (The two paths do not see the same crop: the uint8
    # crop is resampled at centroids that differ by ~0.002 px, which changes a few grey levels by one count -- a flat,
    # low-confidence node of one crop moved by 0.42 px on that, the measured worst case.)
    assert worst <= TOL_PX and worst_c <= TOL_PX and float(np.percentile(dist, 95)) <= 0.05, (worst, worst_c)


# (three frames each that the model was not fitted to: the fit draws seeds >= 10000; SA_C4_SEEDS="a,b,..." widens the sweep --
#  tools/r04_p.sh ran eight seeds once, profiles/r04_ab_session.md section 8)
C4_SEEDS = tuple(int(v) for v in os.environ["SA_C4_SEEDS"].split(",")) if os.environ.get("SA_C4_SEEDS") else (304, 306, 308, 310)


@pytest.fixture(scope="module", params=C4_SEEDS)
def resnet_workload(request):
    import config_models as C

    task, n_frames = "c4_resnet", 3
    sk = C.skeleton(task)
    frames, insts = C.render(task, n_frames, seed=request.param)
    mc, w = C.load_task_weights(task, 1024, 1024)
    cms, pafs = KerasGraph(mc, w)(preprocess(frames))[:2]
    pts, vals, si, ci = opf.find_local_peaks(cms, 0.2, "integral", 5)
    rough = opf.find_local_peaks_rough(cms, 0.2)[0]  # the grid maxima, in the same order
    pts = pts * np.float32(4)
    sc = opg.PAFScorer(sk.nodes, sk.edges, 8, oob="zero")
    B = n_frames
    ref = sc.predict(pafs, [pts[si == b] for b in range(B)], [vals[si == b] for b in range(B)], [ci[si == b] for b in range(B)])
    return dict(task=task, frames=frames, insts=insts, mc=mc, w=w, ref=ref, cms=cms, pafs=pafs, peaks=(pts, vals, si, ci), rough=rough,
                n_peaks=[int((si == b).sum()) for b in range(B)])


def test_configs4_oracle_detects_the_animals(resnet_workload):
    """The workload is what configs[4] names for the ORACLE: 8 animals per 1024 x 1024 frame, 24 nodes each. Round 4 refitted the
    ResNet task model with a hard-negative term (tools/train_config_models.py --hard-neg: the strongest 0.05 % of the background
    cells pushed below 0.05): the maps no longer hold hundreds of cross-type maxima at 0.2-0.4 next to the real peaks (round 3:
    ~380 maxima per frame for 192 nodes, which needed a six-threshold "well-conditioned" filter to compare anything). Asserted:
    every rendered animal comes back as one instance with >= 22 of its 24 nodes within 3 px (mean), and at most 2 % of the
    maxima are not a rendered node."""
    ref, insts = resnet_workload["ref"], resnet_workload["insts"]
    total = 0
    for b, inst in enumerate(ref[0]):
        inst = np.asarray(inst).reshape(-1, 24, 2)
        nn = (~np.isnan(inst[..., 0])).sum(axis=1)
        for gt in insts[b]:
            d = np.nanmean(np.linalg.norm(inst - gt[None], axis=-1), axis=1)
            j = int(np.nanargmin(d))
            total += int(nn[j] >= 22 and d[j] < 3.0)
    n_true = 3 * 8 * 24
    print(f"configs[4] oracle: {total} of 24 animals resolved, peaks per frame {resnet_workload['n_peaks']} (rendered nodes: 192)")
    assert total == 24, total
    assert all(192 - 4 <= n <= 192 + 4 for n in resnet_workload["n_peaks"]), resnet_workload["n_peaks"]
    assert abs(sum(resnet_workload["n_peaks"]) - n_true) <= 0.02 * n_true


def test_configs4_resnet50_bottomup_identical_instance_assignments(resnet_workload):
    """configs[4] at north_star's criterion: ResNet-50 (imagenet preprocessing Lambdas folded into the stem) + transposed-conv
    UpsamplingStack with concatenated skips + PAF head, 24 nodes / 23 edges, 8 animals, fp16 storage, device path vs the fp32
    oracle running its own network on float32 master weights. **Every** oracle peak is compared (no conditioning filter):

      * a device peak of the same channel within 2 px must exist and lie within **0.5 px** -- or the difference is a decision on
        nearly equal numbers, counted and bounded: a confidence within 5e-3 of the 0.2 threshold (a threshold decision) or two
        neighbouring cells whose oracle values differ by <= 5e-3 (which of them is "the" maximum);
      * frames without such a decision -- at least 2 of the 3 -- give IDENTICAL instances: same count, same node assignment
        (NaN mask), every coordinate within 0.5 px (`strict_instances=True`)."""
    from parity_helpers import compare_with_threshold_decisions
    import config_models as C
    from sleap_amd.nn.engine import DeviceNetwork
    from sleap_amd.nn.inference import BottomUpPredictor

    wl = resnet_workload
    net = DeviceNetwork(wl["mc"], wl["w"], dtype="fp16")
    pred = BottomUpPredictor(bottomup_config=C.training_config(wl["task"]), bottomup_model=net, batch_size=len(wl["frames"]),
                             verbosity="none")
    layer = pred.inference_model.bottomup_layer
    layer.return_paf_graph = True
    o = {k: v.cpu().numpy() for k, v in pred.inference_model.call_checked(torch.from_numpy(wl["frames"]).cuda()).items()
         if isinstance(v, torch.Tensor)}
    assert not int(np.bitwise_or.reduce(o["status"])), "capacity overflow / non-finite / out-of-bounds status"
    dev = tuple(o[k] for k in ("peaks", "peak_vals", "peak_channel_inds", "peak_count"))
    stats = {}
    differing, n_common, worst, n_only, n_tie = compare_with_threshold_decisions(
        wl["peaks"], dev, wl["ref"], o, n_nodes=24, map_eps=5e-3, tol_px=TOL_PX, cms=wl["cms"], stride=4,
        strict_instances=True, stats=stats)
    n_oracle = len(wl["peaks"][0])
    same_sets = stats.get("frames_with_equal_peak_sets", [])
    print(f"configs[4]: {n_common} of {n_oracle} oracle peaks matched, max delta {worst:.4f} px; {n_only} threshold decisions, {n_tie} "
          f"neighbouring-cell ties; frames with equal peak sets {same_sets}, of those with identical instances "
          f"{stats.get('frames_with_equal_instances', [])}")
    assert worst <= TOL_PX, worst
    assert n_common >= 0.99 * n_oracle, (n_common, n_oracle)  # (a bound on what is NOT compared positionally)
    assert len(same_sets) >= 2, differing                       # >= 2 of 3 frames reach the instance-level comparison ...
    assert stats.get("frames_with_equal_instances", []) == same_sets  # ... and are identical there (asserted inside as well)
    for b in same_sets:  # stated once more, in the words of north_star: same count, same assignment, <= 0.5 px
        want = np.asarray(wl["ref"][0][b]).reshape(-1, 24, 2)
        got = o["instance_peaks"][b, : int(o["n_valid"][b])]
        assert got.shape == want.shape and np.array_equal(np.isnan(got), np.isnan(want))
        assert float(np.nanmax(np.linalg.norm(got - want, axis=-1))) <= TOL_PX


def test_configs4_network_maps_vs_fp32_oracle(resnet_workload):
    from sleap_amd.nn.engine import DeviceNetwork

    wl = resnet_workload
    net = DeviceNetwork(wl["mc"], wl["w"], dtype="fp16")
    outs = net.forward(torch.from_numpy(wl["frames"][:2]).cuda())
    for o, r in zip(outs, (wl["cms"][:2], wl["pafs"][:2])):
        o = o.cpu().numpy()
        assert np.isfinite(o).all()
        assert float(np.abs(o - r).max()) <= 1e-2 * float(np.abs(r).max())


def test_configs4_model_directory_through_load_model(tmp_path, resnet_workload):
    """The drop-in boundary for a ResNet backbone: a SLEAP model FOLDER (`training_config.json` + the Keras graph / weights as the
    extracted `best_model.npz`, inference.py:132-144, 3204-3209) through `load_model` -> `BottomUpPredictor.predict` gives what the
    directly constructed predictor gives (same networks, same frames), and the backbone block of the config is read as the
    reference reads it (`max_stride` 32 -> pad_to_stride)."""
    import json

    import config_models as C
    from sleap_amd.nn.engine import DeviceNetwork
    from sleap_amd.nn.inference import BottomUpPredictor, load_model

    wl = resnet_workload
    d = tmp_path / "mouse24.ResNet50.bottomup"
    d.mkdir()
    (d / "training_config.json").write_text(json.dumps(C.training_config(wl["task"])))
    np.savez(d / "best_model.npz", __model_config__=np.frombuffer(json.dumps(wl["mc"]).encode(), dtype=np.uint8), **wl["w"])
    pred = load_model(str(d), batch_size=3, peak_threshold=0.2, refinement="integral", progress_reporting="none")
    assert isinstance(pred, BottomUpPredictor)
    layer = pred.inference_model.bottomup_layer
    assert layer.pad_to_stride == 32 and layer.cm_output_stride == 4 and layer.paf_output_stride == 8
    assert layer.paf_scorer.n_nodes == 24 and len(layer.paf_scorer.edges) == 23
    got = pred.predict(wl["frames"], make_labels=False)[0]
    ref = BottomUpPredictor(bottomup_config=C.training_config(wl["task"]), bottomup_model=DeviceNetwork(wl["mc"], wl["w"]),
                            batch_size=3, verbosity="none").predict(wl["frames"], make_labels=False)[0]
    # the fp32 oracle's instance counts: 8 per frame on the default seeds (test_configs4_oracle_detects_the_animals asserts it); in
    # the eight-seed sweep (SA_C4_SEEDS) one frame of seed 309 has 9 on BOTH sides -- model and oracle split the same mouse
    n_oracle = [int(np.asarray(x).reshape(-1, 24, 2).shape[0]) for x in wl["ref"][0]]
    assert np.array_equal(got["n_valid"], ref["n_valid"]) and got["n_valid"].tolist() == n_oracle
    np.testing.assert_array_equal(got["instance_peaks"], ref["instance_peaks"])
    labels = pred.predict(wl["frames"])  # make_labels=True, the reference's default: 3 labeled frames with 8 instances each
    assert len(labels) == 3 and [len(lf.instances) for lf in labels] == n_oracle
