"""Timings of the BASELINE.json configurations that are NOT the bench line (configs[0], [1], [2], [4]) on one MI355X, with
random-init weights of the named architectures and synthetic frames resident in HBM. These are reported numbers for
DESIGN.md section 5, not `bench.py` output; the heads are affinely calibrated so that peak finding / cropping / grouping run
on realistic counts (as sleap_amd/benchmark_model.py does for the bench line).

    python tools/bench_configs.py [steps]
"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")  # config_models.py: test infrastructure
from sleap_amd.nn import architectures as A
from sleap_amd.nn.engine import DeviceNetwork
from sleap_amd.nn.inference import (BottomUpInferenceLayer, BottomUpInferenceModel, CentroidCrop, FindInstancePeaks,
                                    SingleInstanceInferenceLayer, SingleInstanceInferenceModel, TopDownInferenceModel)
from sleap_amd.nn.paf_grouping import PAFScorer
from sleap_amd.synth import FLIES13_EDGES, FLIES13_NODES, render_frames

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def unet(shape, filters, max_stride, out_stride, heads, seed=0):
    cfg, sh = A.build_unet_model_config(shape, filters, 2, max_stride, out_stride, True, True, heads=heads)
    return DeviceNetwork(cfg, A.he_normal_weights(sh, seed))


def calibrate_local(net, x, idx, k_per_channel):
    """k-th largest local maximum of every channel -> 0.3, 2k-th -> 0.1 (threshold 0.2 keeps ~k peaks per channel)."""
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
    """per-channel maximum -> 0.8, median -> 0: every node has one global peak above the threshold."""
    cms = net.forward(x)[idx].clone()
    mx = cms.amax(dim=(0, 1, 2))
    med = cms.flatten(0, 2).median(dim=0).values
    a = 0.8 / (mx - med).clamp_min(1e-6)
    net.rescale_head(idx, a.tolist(), (-a * med).tolist())


def timed(fn, frames, label, n_frames, extra=""):
    for _ in range(3):
        out = fn(frames)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        out = fn(frames)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / STEPS
    print(f"| {label} | {n_frames} | {dt * 1e3:.2f} | {n_frames / dt:.0f} | {extra} |", flush=True)
    return out


def dev(frames):
    return torch.from_numpy(frames).cuda()


print("| configuration | frames / step | ms / step | frames / s | notes |\n|---|---|---|---|---|")

# configs[0]: centroid UNet (baseline.centroid: filters 16, rate 2, stride 16 -> 2, input_scaling 0.5), 256x256, 5 nodes,
# through the single-instance layer (global peak per node)
fr = dev(render_frames(64, 256, 256, n_animals=1, seed=1)[0])
net = unet((128, 128, 1), 16, 16, 2, [("SingleInstanceConfmapsHead", 5, 2)])
layer = SingleInstanceInferenceLayer(net, input_scale=0.5, pad_to_stride=16, peak_threshold=0.2, refinement="integral")
calibrate_global(net, layer.preprocess(fr), 0)
model = SingleInstanceInferenceModel(layer)
o = timed(model.call, fr, "configs[0] single-instance UNet (centroid profile f16 s16->2, input x0.5), 256x256, 5 nodes", 64)
o0 = f"{int(torch.isfinite(o['instance_peaks'][..., 0]).sum())} peaks"

# configs[1]: single-instance UNet (baseline_medium_rf.single: f16 r2 s16->2... output stride 4 here), 512x512, 13 nodes, batch 32
fr = dev(render_frames(32, 512, 512, n_animals=1, seed=2)[0])
net = unet((512, 512, 1), 16, 16, 4, [("SingleInstanceConfmapsHead", 13, 4)])
layer = SingleInstanceInferenceLayer(net, pad_to_stride=16, peak_threshold=0.2, refinement="integral")
calibrate_global(net, fr, 0)
model = SingleInstanceInferenceModel(layer)
timed(model.call, fr, "configs[1] single-instance UNet f16 r2 s16->4, 512x512, 13 nodes", 32)

# configs[2]: top-down, centroid UNet (f16 s16->2, input x0.5) + centered-instance UNet (f24 r2 s16->4) on 256x256 crops,
# 1024x1024 frames, ~2 animals
fr = dev(render_frames(16, 1024, 1024, n_animals=2, seed=3)[0])
cnet = unet((512, 512, 1), 16, 16, 2, [("CentroidConfmapsHead", 1, 2)])
crop = CentroidCrop(cnet, crop_size=256, input_scale=0.5, pad_to_stride=16, peak_threshold=0.2, refinement="integral",
                    max_instances=None)
calibrate_local(cnet, crop.preprocess(fr), 0, 2)
inet = unet((256, 256, 1), 24, 16, 4, [("CenteredInstanceConfmapsHead", 13, 4)])
peaks = FindInstancePeaks(inet, peak_threshold=0.2, refinement="integral")
calibrate_global(inet, fr[:, :256, :256].contiguous(), 0)
td = TopDownInferenceModel(crop, peaks)
o = timed(td.call, fr, "configs[2] top-down: centroid UNet (x0.5) + centered-instance UNet f24 on 256x256 crops, 1024x1024", 16)
print(f"|   (configs[2] instances per frame: {float(o['n_valid'].float().mean()):.2f}) | | | | |")

# configs[4]: bottom-up ResNet-50 (+ imagenet preprocessing Lambdas) with a transposed-conv upsampling stack, 24 nodes
nodes = [f"n{i}" for i in range(24)]
edges = [(nodes[(i - 1) // 2], nodes[i]) for i in range(1, 24)]  # a tree with 23 edges
fr = dev(render_frames(16, 1024, 1024, n_animals=8, seed=4)[0])
cfg, sh = A.build_resnet_model_config((1024, 1024, 1), "ResNet50", 32, pretrained=True,
                                      upsampling=dict(output_stride=4, method="transposed_conv", skip_connections="concatenate",
                                                      filters=64, refine_convs=2),
                                      heads=[("MultiInstanceConfmapsHead", 24, 4), ("PartAffinityFieldsHead", 46, 8)])
# residual_scale 0.25: plain He init lets the residual stream of ResNet-50 double its variance per block and leave fp16's
# range (the fp16 build reports it: FloatingPointError from the engine's range check); a trained network's BN keeps it O(1-10)
rnet = DeviceNetwork(cfg, A.he_normal_weights(sh, 0, residual_scale=0.25))
calibrate_local(rnet, fr[:2], 0, 8)
pafs = rnet.forward(fr[:2])[1]
rnet.rescale_head(1, [0.5 / max(float(pafs.std()), 1e-6)] * 46, [0.0] * 46)
scorer = PAFScorer(part_names=nodes, edges=edges, pafs_stride=8, max_instances=64)
bl = BottomUpInferenceLayer(rnet, scorer, pad_to_stride=32, cm_output_stride=4, paf_output_stride=8, peak_threshold=0.2,
                            refinement="integral", max_peaks=1024)
bl.assume_inputs_ready = True
bm = BottomUpInferenceModel(bl)
o = timed(bm.call_checked, fr, "configs[4] bottom-up ResNet-50 + transposed-conv upsampling stack + PAFs, 1024x1024, 24 nodes / 23 edges", 16)
# the ResNet forward alone against BOTH rooflines (round 6, VERDICT r5 item 8: 25 of its 58 launches are 1x1 convs that move their
# bytes at 3-4.5 TB/s -- the HBM roofline, not the MFMA one, is the yardstick of this network): algorithmic bytes = every launch's
# inputs read once and outputs written once (DeviceNetwork.op_bytes), FLOPs = DeviceNetwork.op_descriptions
for _ in range(3):
    rnet.forward(fr)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(STEPS):
    rnet.forward(fr)
torch.cuda.synchronize()
net_ms = (time.perf_counter() - t0) / STEPS * 1e3
nb4 = float(sum(rnet.op_bytes(1024, 1024))) * 16
fl4 = float(sum(f for _, _, f in rnet.op_descriptions(1024, 1024))) * 16
print(f"|   (configs[4] network forward alone: {net_ms:.2f} ms per 16 frames; algorithmic {nb4 / 1e9:.2f} GB -> {nb4 / net_ms / 1e9:.2f} TB/s = "
      f"**{nb4 / net_ms / 1e9 / 8.0:.3f} of the 8 TB/s HBM roofline**; {fl4 / 1e12:.2f} TFLOP -> {fl4 / net_ms / 1e9:.0f} TFLOP/s = "
      f"{fl4 / net_ms / 1e9 / 2500.0:.3f} of the MFMA roofline) | | | | |")
print(f"|   (configs[4] status bits {int(np.bitwise_or.reduce(o['status'].cpu().numpy().astype(np.int64)))}, "
      f"instances per frame {float(o['n_valid'].float().mean()):.2f}; configs[0] {o0}) | | | | |")

# configs[4] with the FITTED task model of the parity test (tests/data/config_c4_resnet.npz: 8 mice per frame found as 8 instances,
# 192 peaks) instead of random weights: the same architecture, post-processing on realistic counts
try:
    import config_models as C
    from sleap_amd.nn.inference import BottomUpPredictor

    frames, _ = C.render("c4_resnet", 16, seed=400)
    mc, w = C.load_task_weights("c4_resnet", 1024, 1024)
    pred = BottomUpPredictor(bottomup_config=C.training_config("c4_resnet"), bottomup_model=DeviceNetwork(mc, w), batch_size=16,
                             verbosity="none")
    pred.inference_model.bottomup_layer.assume_inputs_ready = True
    fr = dev(frames)
    o = timed(pred.inference_model.call, fr, "configs[4] the fitted task model (24-node mouse, 8 animals; tests/data/config_c4_resnet.npz)", 16)
    print(f"|   (fitted configs[4]: instances per frame {float(o['n_valid'].float().mean()):.2f}, status bits "
          f"{int(np.bitwise_or.reduce(o['status'].cpu().numpy().astype(np.int64)))}) | | | | |")
except FileNotFoundError as e:  # the fixture is a test artefact; the tool works without it
    print(f"|   (fitted configs[4] model not available: {e}) | | | | |")
