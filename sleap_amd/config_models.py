"""Small FITTED networks of the architectures BASELINE.json configs[0], [1], [2] and [4] name, for the end-to-end parity tests
of those configurations (tests/test_gpu_config_parity.py) -- the counterparts of the configs[3] benchmark model
(sleap_amd/benchmark_model.py). Architectures follow the reference's shipped training profiles (sleap/training_profiles/):

    task          architecture (training profile)                                  input                       head(s)
    c0_single5    UNet f16 r2 s16->2 bilinear (baseline.centroid.json)             256^2 x0.5, 1 animal        SingleInstanceConfmapsHead 5 @2
    c1_single13   UNet f16 r2 s16->2 bilinear (baseline_medium_rf.single.json)     512^2, 1 animal             SingleInstanceConfmapsHead 13 @2
    c2_centroid   UNet f16 r2 s16->2 bilinear (baseline.centroid.json)             1024^2 x0.5, 2 animals      CentroidConfmapsHead 1 @2
    c2_centered   UNet f24 r2 s16->4 bilinear (baseline_medium_rf.topdown.json)    160^2 crops                 CenteredInstanceConfmapsHead 13 @4
    c4_resnet     ResNet-50 (resnet.py:544-595, imagenet preprocessing Lambdas) +  1024^2, 8 animals           MultiInstanceConfmapsHead 24 @4,
                  UpsamplingStack (transposed conv k4 s2 + BN, concatenate skips)                              PartAffinityFieldsHead 46 @8

Weights: `sleap_amd/data/config_<task>.npz`, fitted by tools/train_config_models.py to the synthetic videos of
`sleap_amd.synth.render_animals` and stored as **float32 masters** (real SLEAP weights are fp32: the device path rounds them to
its 16-bit storage type itself, the fp32 oracle does not). For `c4_resnet` only the stem, conv2, conv3, the upsampling stack
and the heads are fitted and stored (~5 M parameters); conv4 / conv5 (22 M parameters) keep their seeded He-normal values,
which `he_normal_weights(shapes, seed=0)` regenerates bit for bit (a checksum of them is stored and verified at load), with
BatchNormalization statistics calibrated on the video and stored.
"""
import os

import numpy as np

from . import synth
from .nn import architectures as A

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
ANCHOR = 1  # centroid = the thorax node of the fly (sleap/nn/data/instance_centroids.py: anchor part)

TASKS = {
    "c0_single5": dict(kind="single", skeleton="FLIES5", frame=256, n_animals=1, input_scale=0.5, crop=None, render_margin=96.0,
                       unet=(16, 2.0, 16, 2), heads=[("SingleInstanceConfmapsHead", 5, 2)], steps=1200, batch=16, pool=512),
    "c1_single13": dict(kind="single", skeleton="FLIES13", frame=512, n_animals=1, input_scale=1.0, crop=256, render_margin=128.0,
                        unet=(16, 2.0, 16, 2), heads=[("SingleInstanceConfmapsHead", 13, 2)], steps=1500, batch=8, pool=256),
    "c2_centroid": dict(kind="centroid", skeleton="FLIES13", frame=1024, n_animals=2, input_scale=0.5, crop=256, render_margin=128.0,
                        unet=(16, 2.0, 16, 2), heads=[("CentroidConfmapsHead", 1, 2)], steps=1000, batch=8, pool=96),
    "c2_centered": dict(kind="centered", skeleton="FLIES13", frame=1024, n_animals=2, input_scale=1.0, crop=160, render_margin=128.0,
                        unet=(24, 2.0, 16, 4), heads=[("CenteredInstanceConfmapsHead", 13, 4)], steps=1500, batch=12, pool=96),
    # not a BASELINE config: the hourglass row of SURVEY.md 8(a) (a2') end to end -- a ONE-stack hourglass of the reference's
    # structure (hourglass.py:17-316: stem k7 s2 + pooling, Conv -> ReLU -> BatchNormalization everywhere, nearest-neighbour
    # upsampling with additive skips) at a quarter of its default width, single-instance head at stride 4
    "hg_single13": dict(kind="single", skeleton="FLIES13", frame=512, n_animals=1, input_scale=1.0, crop=256, render_margin=128.0,
                        hourglass=dict(stem_stride=4, max_stride=32, output_stride=4, stem_filters=32, filters=64, filter_increase=32,
                                       stacks=1), heads=[("SingleInstanceConfmapsHead", 13, 4)], steps=1500, batch=8, pool=256, loss_margin=0),
    "c4_resnet": dict(kind="multi", skeleton="MOUSE24", frame=1024, n_animals=8, input_scale=1.0, crop=256, render_margin=128.0,
                      body=(60.0, 90.0), min_sep=200.0,
                      resnet=dict(version="ResNet50", features_output_stride=32, pretrained=True,
                                  upsampling=dict(output_stride=4, method="transposed_conv", skip_connections="concatenate",
                                                  filters=64, refine_convs=2)),
                      heads=[("MultiInstanceConfmapsHead", 24, 4), ("PartAffinityFieldsHead", 46, 8)],
                      freeze=r"^conv[45]_", steps=3000, batch=8, pool=64, loss_margin=0, nonneg=50.0),
}


def skeleton(task):
    return getattr(synth, TASKS[task]["skeleton"])


def render(task, n_frames, seed, **kw):
    """Frames of a task's video (seeds < 10000 are never fitted to) -> (uint8 (T, H, W, 1), list of (A, N, 2) instances)."""
    t = TASKS[task]
    args = dict(skeleton=skeleton(task), margin=t["render_margin"], body=t.get("body", (80.0, 120.0)), min_sep=t.get("min_sep", 170.0))
    args.update(kw)
    return synth.render_animals(n_frames, t["frame"], t["frame"], t["n_animals"], seed=seed, **args)


def task_graph(task, height, width):
    """-> (model_config, weight shapes) of a task's network at the given input size (after input scaling / cropping)."""
    t = TASKS[task] if isinstance(task, str) else task
    if "unet" in t:
        f, r, ms, os_ = t["unet"]
        return A.build_unet_model_config((height, width, 1), f, r, ms, os_, True, True, None, heads=t["heads"])
    if "hourglass" in t:
        return A.build_hourglass_model_config((height, width, 1), heads=t["heads"], **t["hourglass"])
    r = t["resnet"]
    return A.build_resnet_model_config((height, width, 1), r["version"], r["features_output_stride"], r["pretrained"],
                                       upsampling=r["upsampling"], heads=t["heads"])


def weights_path(task):
    return os.path.join(DATA_DIR, f"config_{task}.npz")


def load_task_weights(task, height, width, path=None, seed=0):
    """-> (model_config, {"<layer>/<weight>": float32 array}) of a fitted task network at the given input size."""
    mc, shapes = task_graph(task, height, width)
    path = path or weights_path(task)
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} is missing: run `python tools/train_config_models.py {task}` (plain torch, CPU)")
    z = np.load(path)
    stored = {k: z[k] for k in z.files if not k.startswith("__")}
    if "__frozen_checksum__" in z.files:
        w = A.he_normal_weights(shapes, seed=seed)
        frozen = [k for k in sorted(w) if k not in stored]
        for k in frozen:  # the fit starts every BatchNormalization neutral (tools/train_config_models.py: gamma 1, beta 0)
            if k.endswith("/gamma"):
                w[k][:] = 1.0
            elif k.endswith("/beta"):
                w[k][:] = 0.0
        chk = float(sum(np.abs(w[k].astype(np.float64)).sum() for k in frozen))
        if abs(chk - float(z["__frozen_checksum__"])) > 1e-6 * abs(chk):
            raise RuntimeError("the seeded (not stored) weights of this model do not reproduce on this NumPy build")
        w.update(stored)
    else:
        w = stored
    assert set(w) == set(shapes) and all(tuple(w[k].shape) == tuple(shapes[k]) for k in shapes)
    return mc, {k: np.asarray(v, np.float32) for k, v in w.items()}


def training_config(task, nodes=None, edges=None):
    """The fields of training_config.json the predictors read for a task's model."""
    t = TASKS[task]
    sk = skeleton(task)
    nodes = list(nodes or sk.nodes)
    heads = {k: None for k in ("single_instance", "centroid", "centered_instance", "multi_instance", "multi_class_bottomup",
                               "multi_class_topdown")}
    hs = t["heads"][0][2]
    if t["kind"] == "single":
        heads["single_instance"] = {"part_names": nodes, "sigma": 2.5, "output_stride": hs, "offset_refinement": False}
    elif t["kind"] == "centroid":
        heads["centroid"] = {"anchor_part": nodes[ANCHOR], "sigma": 2.5, "output_stride": hs, "offset_refinement": False}
    elif t["kind"] == "centered":
        heads["centered_instance"] = {"anchor_part": nodes[ANCHOR], "part_names": nodes, "sigma": 2.5, "output_stride": hs,
                                      "offset_refinement": False}
    else:
        heads["multi_instance"] = {"confmaps": {"part_names": nodes, "sigma": 2.5, "output_stride": hs, "offset_refinement": False},
                                   "pafs": {"edges": [list(e) for e in (edges or sk.edges)], "sigma": 75.0,
                                            "output_stride": t["heads"][1][2]}}
    backbone = {k: None for k in ("leap", "unet", "hourglass", "resnet", "pretrained_encoder")}
    if "unet" in t:
        f, r, ms, os_ = t["unet"]
        backbone["unet"] = {"stem_stride": None, "max_stride": ms, "output_stride": os_, "filters": f, "filters_rate": r,
                            "middle_block": True, "up_interpolate": True, "stacks": 1}
    elif "hourglass" in t:
        backbone["hourglass"] = dict(t["hourglass"])
    else:
        backbone["resnet"] = {"version": t["resnet"]["version"], "weights": "frozen", "upsampling": dict(t["resnet"]["upsampling"]),
                              "max_stride": t["resnet"]["features_output_stride"], "output_stride": t["resnet"]["upsampling"]["output_stride"]}
    return {"data": {"preprocessing": {"ensure_rgb": False, "ensure_grayscale": False, "input_scaling": t["input_scale"],
                                       "pad_to_stride": None, "resize_and_pad_to_target": True, "target_height": None,
                                       "target_width": None},
                     "instance_cropping": {"center_on_part": nodes[ANCHOR] if t["kind"] == "centered" else None,
                                           "crop_size": t["crop"] if t["kind"] == "centered" else None,
                                           "crop_size_detection_padding": 16},
                     "labels": {"skeletons": []}},
            "model": {"backbone": backbone, "heads": heads}}
