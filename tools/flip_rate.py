"""How often does 16-bit storage change an instance ASSIGNMENT?  (a flip rate with its denominator)

The device path (fp16-storage MFMA network + fused post-processing) against the fp32 CPU oracle (torch-CPU Keras graph on the
float32 weights + restated peak finding / PAF grouping) on the SAME uint8 frames, frame by frame, for BASELINE configs[3] (the
benchmark UNet, 13 nodes, 4 animals) or configs[4] (ResNet-50, 24 nodes, 8 animals; tests/config_models.py). Per frame:

    identical   same instance count, same node assignment (NaN mask), every coordinate within 0.5 px   (north_star's criterion)
    else        classified: the PEAK SETS differ (a threshold / neighbouring-cell decision on nearly equal map values, counted) or
                the peak sets are equal and the MATCHING / GROUPING differs (a "flip": the Hungarian assignment or the grouping
                decided differently on nearly equal line scores)

    python tools/flip_rate.py [--config 3|4] [--frames 1024] [--chunk 16] [--seed 5000]      (GPU + host cores; ~5 frames/s)
"""
import argparse
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from oracle import paf_grouping as opg  # noqa: E402
from oracle import peak_finding as opf  # noqa: E402
from oracle.keras_graph import KerasGraph, preprocess  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=3)
ap.add_argument("--frames", type=int, default=1024)
ap.add_argument("--chunk", type=int, default=16)
ap.add_argument("--seed", type=int, default=5000)
ap.add_argument("--threshold", type=float, default=0.2)
args = ap.parse_args()

from sleap_amd.nn.engine import DeviceNetwork  # noqa: E402
from sleap_amd.nn.inference import BottomUpPredictor  # noqa: E402

if args.config == 3:
    from sleap_amd.benchmark_model import build_benchmark_graph, load_trained_weights
    from sleap_amd.synth import FLIES13_EDGES, FLIES13_NODES, render_flies

    cfg, mc, _ = build_benchmark_graph(1024, 1024)
    w = load_trained_weights()
    nodes, edges = FLIES13_NODES, FLIES13_EDGES

    def render(n, seed):
        return render_flies(n, 1024, 1024, n_animals=4, seed=seed)[0]
else:
    import config_models as C

    task = "c4_resnet"
    sk = C.skeleton(task)
    mc, w = C.load_task_weights(task, 1024, 1024)
    cfg = C.training_config(task)
    nodes, edges = sk.nodes, sk.edges

    def render(n, seed):
        return C.render(task, n, seed=seed)[0]

N = len(nodes)
graph = KerasGraph(mc, w)
scorer = opg.PAFScorer(nodes, edges, 8, oob="zero")
pred = BottomUpPredictor(bottomup_config=cfg, bottomup_model=DeviceNetwork(mc, w, dtype="fp16"), batch_size=args.chunk, verbosity="none",
                         peak_threshold=args.threshold)
layer = pred.inference_model.bottomup_layer
layer.return_paf_graph = True

tot = dict(frames=0, identical=0, peak_set_differs=0, flip_equal_peak_sets=0, peaks=0, worst_px=0.0, count_differs=0)
flips = []
t0 = time.time()
for c0 in range(0, args.frames, args.chunk):
    n = min(args.chunk, args.frames - c0)
    frames = render(n, args.seed + c0)
    cms, pafs = graph(preprocess(frames))[:2]
    pts, vals, si, ci = opf.find_local_peaks(cms, args.threshold, "integral", 5)
    rough = opf.find_local_peaks_rough(cms, args.threshold)[0]
    pts = pts * np.float32(4)
    ref = scorer.predict(pafs, [pts[si == b] for b in range(n)], [vals[si == b] for b in range(n)], [ci[si == b] for b in range(n)])
    o = {k: v.cpu().numpy() for k, v in pred.inference_model.call_checked(torch.from_numpy(frames).cuda()).items() if isinstance(v, torch.Tensor)}
    assert not int(np.bitwise_or.reduce(o["status"])), "status bits set"
    for b in range(n):
        want = np.asarray(ref[0][b]).reshape(-1, N, 2)
        nv = int(o["n_valid"][b])
        got = o["instance_peaks"][b, :nv]
        tot["frames"] += 1
        same = got.shape == want.shape and np.array_equal(np.isnan(got), np.isnan(want))
        d = float(np.nanmax(np.linalg.norm(got - want, axis=-1))) if same and want.size and np.isfinite(want).any() else 0.0
        if same and d <= 0.5:
            tot["identical"] += 1
            tot["peaks"] += int(np.isfinite(want[..., 0]).sum())
            tot["worst_px"] = max(tot["worst_px"], d)
            continue
        # the detected peak SETS: (channel, rounded grid cell) of the oracle vs the device
        k = int(o["peak_count"][b])
        dev_set = {(int(c), int(round(float(x) / 4)), int(round(float(y) / 4))) for (x, y), c in zip(o["peaks"][b, :k], o["peak_channel_inds"][b, :k])}
        m = si == b
        ora_set = {(int(c), int(x), int(y)) for (x, y), c in zip(rough[m], ci[m])}
        if dev_set != ora_set:
            tot["peak_set_differs"] += 1
        else:
            tot["flip_equal_peak_sets"] += 1
            flips.append((args.seed + c0, b, nv, len(want), round(d, 3)))
        if nv != len(want):
            tot["count_differs"] += 1
    if (c0 // args.chunk) % 8 == 7:
        print(f"  ... {tot['frames']} frames, {time.time() - t0:.0f} s: {tot}", file=sys.stderr, flush=True)

f = tot["frames"]
print(f"# flip rate, BASELINE configs[{args.config}], fp16 storage vs the fp32 oracle, threshold {args.threshold}, {f} frames (render seeds {args.seed}..)")
print(f"identical frames (count, assignment, every peak <= 0.5 px): {tot['identical']} of {f} = {100.0 * tot['identical'] / f:.2f} %; "
      f"{tot['peaks']} peaks compared, worst {tot['worst_px']:.4f} px")
print(f"frames whose PEAK SETS differ (threshold / neighbouring-cell decisions): {tot['peak_set_differs']} = {1000.0 * tot['peak_set_differs'] / f:.2f} per 1000 frames")
print(f"frames with equal peak sets and a different matching / grouping (flips): {tot['flip_equal_peak_sets']} = "
      f"{1000.0 * tot['flip_equal_peak_sets'] / f:.2f} per 1000 frames   {flips[:20]}")
print(f"frames with a different instance COUNT: {tot['count_differs']}")
