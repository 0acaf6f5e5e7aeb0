"""End-to-end rate of the top-down predictor (BASELINE configs[2]: centroid UNet on x0.5 frames + centered-instance UNet on
256x256 crops), HOST frames in -> result dictionaries out, random-init calibrated weights as tools/bench_configs.py.

    python tools/predict_e2e_topdown.py [T] [batch]
"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
sys.argv, argv = sys.argv[:1], sys.argv[1:]
from sleap_amd.nn import architectures as A
from sleap_amd.nn.engine import DeviceNetwork
from sleap_amd.nn.inference import CentroidCrop, FindInstancePeaks, TopDownInferenceModel, TopDownPredictor
from sleap_amd.synth import render_frames

T = int(argv[0]) if argv else 512
B = int(argv[1]) if len(argv) > 1 else 16


def unet(shape, filters, max_stride, out_stride, heads, seed=0):
    cfg, sh = A.build_unet_model_config(shape, filters, 2, max_stride, out_stride, True, True, heads=heads)
    return DeviceNetwork(cfg, A.he_normal_weights(sh, seed))


def calibrate_local(net, x, idx, k_per_channel):
    cms = net.forward(x)[idx].clone().permute(0, 3, 1, 2)
    is_max = torch.nn.functional.max_pool2d(cms, 3, 1, 1) == cms
    scale, shift = [], []
    for c in range(cms.shape[1]):
        v = cms[:, c][is_max[:, c]]
        k = max(min(k_per_channel * cms.shape[0], v.numel() // 2), 1)
        top = torch.topk(v, 2 * k).values
        a = 0.2 / max(float(top[k - 1] - top[2 * k - 1]), 1e-6)
        scale.append(a)
        shift.append(0.3 - a * float(top[k - 1]))
    net.rescale_head(idx, scale, shift)


def calibrate_global(net, x, idx):
    cms = net.forward(x)[idx].clone()
    mx = cms.amax(dim=(0, 1, 2))
    med = cms.flatten(0, 2).median(dim=0).values
    a = 0.8 / (mx - med).clamp_min(1e-6)
    net.rescale_head(idx, a.tolist(), (-a * med).tolist())


base = render_frames(16, 1024, 1024, n_animals=2, seed=3)[0]
fr = torch.from_numpy(base).cuda()
cnet = unet((512, 512, 1), 16, 16, 2, [("CentroidConfmapsHead", 1, 2)])
crop = CentroidCrop(cnet, crop_size=256, input_scale=0.5, pad_to_stride=16, peak_threshold=0.2, refinement="integral", max_instances=None)
calibrate_local(cnet, crop.preprocess(fr), 0, 2)
inet = unet((256, 256, 1), 24, 16, 4, [("CenteredInstanceConfmapsHead", 13, 4)])
peaks = FindInstancePeaks(inet, peak_threshold=0.2, refinement="integral")
calibrate_global(inet, fr[:, :256, :256].contiguous(), 0)
pred = TopDownPredictor.__new__(TopDownPredictor)
pred.inference_model = TopDownInferenceModel(crop, peaks)
pred.batch_size, pred.verbosity, pred.report_rate, pred.tracker, pred.model_paths, pred.max_instances = B, "none", 2.0, None, [], None
frames = np.ascontiguousarray(np.tile(base, (T // 16, 1, 1, 1)))
pred.predict(frames[: 4 * B], make_labels=False)
for rep in range(3):
    t0 = time.perf_counter()
    out = pred.predict(frames, make_labels=False)
    dt = time.perf_counter() - t0
    n = sum(len(o["frame_ind"]) for o in out)
    inst = float(np.mean([np.mean(o["n_valid"]) for o in out]))
    print(f"top-down predict(make_labels=False), batch {B}: {T} host frames in {dt * 1e3:.1f} ms = {T / dt:.0f} frames/s ({n} results, {inst:.2f} instances per frame)", flush=True)
# the same model on frames resident in HBM, no result conversion
dev = torch.from_numpy(frames[:B]).cuda()
for _ in range(3):
    pred.inference_model.call(dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    pred.inference_model.call(dev)
torch.cuda.synchronize()
print(f"model.call on resident frames, batch {B}: {10 * B / (time.perf_counter() - t0):.0f} frames/s")
if len(argv) > 2 and argv[2] == "profile":
    import cProfile
    import pstats

    pr = cProfile.Profile()
    pr.enable()
    pred.predict(frames, make_labels=False)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
if len(argv) > 2 and argv[2] == "resident":
    alld = torch.from_numpy(frames).cuda()
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = pred.predict(alld, make_labels=False)
        dt = time.perf_counter() - t0
        print(f"predict() on a CUDA tensor of all frames: {T / dt:.0f} frames/s")
