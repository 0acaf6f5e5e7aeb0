"""Benchmark models of the reference's shipped training profiles.

Architecture of `sleap/training_profiles/baseline_medium_rf.bottomup.json` (UNet filters 16, rate 2, max_stride 32,
output_stride 4, bilinear upsampling; confmaps @ stride 4, PAFs @ stride 8) with the 13-node / 12-edge `flies13` skeleton.

`build_benchmark_predictor()` (the default, what bench.py and the configs[3] parity tests use) loads weights FITTED to the
synthetic fly video of `sleap_amd.synth.render_flies` (`sleap_amd/data/benchmark_unet_flies13.npz`, produced by
tools/train_benchmark_model.py; fp16-representable values stored as float16): the network detects the 4 rendered animals x 13
nodes with peaks far above the 0.2 threshold and PAF scores far above the 0.25 cut, so the post-processing runs on real
instance counts and the end-to-end comparison with the fp32 oracle is well conditioned.

`trained=False` gives the round-1 stand-in: seeded He-normal weights with the two 1x1 heads affinely calibrated so that every
confidence-map channel has about `n_animals` local maxima above the threshold (noise-like maps: fine for timing the network,
ill-conditioned for parity).
"""
import os

import numpy as np
import torch

from .nn.architectures import build_unet_model_config, he_normal_weights
from .nn.engine import DeviceNetwork
from .nn.inference import BottomUpPredictor
from .synth import FLIES13_EDGES, FLIES13_NODES, render_frames


def medium_rf_bottomup_config(nodes=FLIES13_NODES, edges=FLIES13_EDGES, filters=16, filters_rate=2.0, max_stride=32,
                              output_stride=4):
    """The fields of training_config.json the inference path reads, baseline_medium_rf values."""
    return {
        "data": {"preprocessing": {"ensure_rgb": False, "ensure_grayscale": False, "input_scaling": 1.0,
                                   "pad_to_stride": None, "resize_and_pad_to_target": True,
                                   "target_height": None, "target_width": None},
                 "labels": {"skeletons": []}},
        "model": {"backbone": {"leap": None, "unet": {"stem_stride": None, "max_stride": max_stride,
                                                      "output_stride": output_stride, "filters": filters,
                                                      "filters_rate": filters_rate, "middle_block": True,
                                                      "up_interpolate": True, "stacks": 1},
                               "hourglass": None, "resnet": None, "pretrained_encoder": None},
                  "heads": {"single_instance": None, "centroid": None, "centered_instance": None,
                            "multi_instance": {"confmaps": {"part_names": list(nodes), "sigma": 2.5, "output_stride": 4,
                                                            "loss_weight": 1.0, "offset_refinement": False},
                                               "pafs": {"edges": [list(e) for e in edges], "sigma": 75.0,
                                                        "output_stride": 8, "loss_weight": 1.0}},
                            "multi_class_bottomup": None, "multi_class_topdown": None}},
    }


def build_benchmark_graph(height, width, cfg=None, seed=0):
    cfg = cfg or medium_rf_bottomup_config()
    u = cfg["model"]["backbone"]["unet"]
    mi = cfg["model"]["heads"]["multi_instance"]
    mc, shapes = build_unet_model_config(
        (height, width, 1), u["filters"], u["filters_rate"], u["max_stride"], u["output_stride"], u["middle_block"],
        u["up_interpolate"], None,
        heads=[("MultiInstanceConfmapsHead", len(mi["confmaps"]["part_names"]), mi["confmaps"]["output_stride"]),
               ("PartAffinityFieldsHead", 2 * len(mi["pafs"]["edges"]), mi["pafs"]["output_stride"])])
    return cfg, mc, he_normal_weights(shapes, seed)


def calibrate_heads(net: DeviceNetwork, frames_u8: torch.Tensor, n_animals=4, cm_index=0, paf_index=1, weights=None):
    """Affine-rescale the random heads (see module docstring). Calibration only; not part of any timed region.
    If `weights` is given, the calibrated head kernels/biases are written back so that the same model can be
    evaluated elsewhere (e.g. by the CPU oracle)."""
    outs = net.forward(frames_u8)
    cms = outs[cm_index].clone()
    pafs = outs[paf_index].clone()
    B = cms.shape[0]
    x = cms.permute(0, 3, 1, 2)
    is_max = (torch.nn.functional.max_pool2d(x, 3, 1, 1) == x)
    scale, shift = [], []
    for c in range(cms.shape[3]):
        v = x[:, c][is_max[:, c]]
        k = max(min(n_animals * B, v.numel() // 2), 1)
        top = torch.topk(v, 2 * k).values
        vk, v2k = float(top[k - 1]), float(top[2 * k - 1])  # k-th largest local maximum -> 0.3, 2k-th -> 0.1
        a = 0.2 / max(vk - v2k, 1e-6)
        scale.append(a)
        shift.append(0.3 - a * vk)
    net.rescale_head(cm_index, scale, shift)
    ps = float(pafs.std())
    net.rescale_head(paf_index, [0.5 / max(ps, 1e-6)] * pafs.shape[3], [0.0] * pafs.shape[3])
    if weights is not None:
        for idx in (cm_index, paf_index):
            k, b = net.export_head(idx)
            name = net.output_names[idx]
            weights[f"{name}/kernel"], weights[f"{name}/bias"] = k, b


TRAINED_WEIGHTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "benchmark_unet_flies13.npz")


def load_trained_weights(path=TRAINED_WEIGHTS):
    """-> {"<layer>/kernel|bias": float32 array} of the fitted benchmark model (values are exactly fp16-representable)."""
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} is missing: run `python tools/train_benchmark_model.py` (plain torch, CPU is enough) or "
                                "pass trained=False for the random-init stand-in")
    z = np.load(path)
    return {k: z[k].astype(np.float32) for k in z.files if k != "__model_config__"}


def build_benchmark_predictor(height=1024, width=1024, batch_size=64, n_animals=4, seed=0, calib_frames=2, dtype=None,
                              trained=True):
    """-> (BottomUpPredictor on the current CUDA device, keras-style model_config, weights dict)."""
    cfg, mc, weights = build_benchmark_graph(height, width, seed=seed)
    if trained:
        fitted = load_trained_weights()
        assert set(fitted) == set(weights) and all(fitted[k].shape == weights[k].shape for k in weights)
        weights = fitted
    net = DeviceNetwork(mc, weights, dtype=dtype)
    if not trained:
        frames, _ = render_frames(calib_frames, height, width, n_animals, seed=1234)
        calibrate_heads(net, torch.from_numpy(frames).cuda(), n_animals, weights=weights)
    pred = BottomUpPredictor(bottomup_config=cfg, bottomup_model=net, batch_size=batch_size)
    return pred, mc, weights
