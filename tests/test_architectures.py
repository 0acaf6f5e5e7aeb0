"""The UNet graph builder reproduces the reference's layer names / wiring / parameter counts
(reference: tests/nn/architectures/test_unet.py pins names, shapes and parameter counts; the fixture
model's own stored graph is the strongest available check)."""
import json
import os

import pytest
import numpy as np

from sleap_amd.nn.architectures import build_unet_model_config, he_normal_weights
from oracle.keras_graph import KerasGraph, load_npz_model

MODELS = os.path.join(os.path.dirname(__file__), "golden", "models")


def _sig(cfg):
    out = []
    for l in cfg["config"]["layers"]:
        inb = [n[0] for n in l["inbound_nodes"][0]] if l["inbound_nodes"] else []
        c = l["config"]
        keys = ("filters", "kernel_size", "strides", "padding", "activation", "pool_size", "size", "interpolation")
        out.append((l["class_name"], l["name"], tuple(inb), tuple((k, json.dumps(c[k])) for k in keys if k in c)))
    return out


def test_builder_reproduces_bottomup_fixture_graph():
    ref_cfg, ref_w = load_npz_model(os.path.join(MODELS, "minimal_instance.UNet.bottomup", "best_model.npz"))
    cfg, shapes = build_unet_model_config(
        (384, 384, 1), filters=16, filters_rate=1.5, max_stride=8, output_stride=2, middle_block=True,
        up_interpolate=False,
        heads=[("MultiInstanceConfmapsHead_0", 2, 2), ("PartAffinityFieldsHead_0", 2, 4), ("OffsetRefinementHead_0", 4, 2)])
    assert _sig(cfg) == _sig(ref_cfg)
    assert cfg["config"]["output_layers"] == ref_cfg["config"]["output_layers"]
    assert {k: tuple(v.shape) for k, v in ref_w.items()} == shapes


def test_builder_reproduces_bilinear_fixture_backbone():
    ref_cfg, ref_w = load_npz_model(os.path.join(MODELS, "min_tracks_2node.UNet.bottomup_multiclass", "best_model.npz"))
    cfg, shapes = build_unet_model_config((512, 512, 1), filters=8, filters_rate=1.5, max_stride=16, output_stride=2,
                                          up_interpolate=True, heads=[])
    ref_backbone = [s for s in _sig(ref_cfg) if s[1].startswith(("stack0", "input"))]
    assert _sig(cfg) == ref_backbone


def test_benchmark_model_params_and_flops():
    # baseline_medium_rf.bottomup + flies13: 7.82 M params, 99.56 GFLOP/frame
    cfg, shapes = build_unet_model_config((1024, 1024, 1), 16, 2, 32, 4, True, True,
                                          heads=[("MultiInstanceConfmapsHead", 13, 4), ("PartAffinityFieldsHead", 24, 8)])
    n_params = sum(int(np.prod(s)) for s in shapes.values())
    assert abs(n_params - 7.82e6) < 0.01e6
    # flops from shapes: 2*H*W*Cin*Cout*k*k
    stride = {}
    g = KerasGraph(cfg, he_normal_weights(shapes))
    outs, allt = g(np.zeros((1, 64, 64, 1), np.float32), return_all=True)
    flops = 0
    for l in cfg["config"]["layers"]:
        if l["class_name"] == "Conv2D":
            k = shapes[l["name"] + "/kernel"]
            h = allt[l["name"]].shape[1] * 16  # scale 64 -> 1024
            flops += 2 * h * h * k[0] * k[1] * k[2] * k[3]
    assert abs(flops / 1e9 - 99.56) < 0.05
    assert outs[0].shape == (1, 16, 16, 13) and outs[1].shape == (1, 8, 8, 24)


def test_hourglass_reference_structure():
    """tests/nn/architectures/test_hourglass.py: 3-stack AE hourglass on (256,256,1): 116 layers, 156 trainable
    weight tensors, 65,969,408 trainable / 66,002,944 total parameters, three (64,64,256) outputs."""
    from oracle.keras_graph import KerasGraph
    from sleap_amd.nn.architectures import build_hourglass_model_config, he_normal_weights

    cfg, shapes = build_hourglass_model_config((256, 256, 1), stem_stride=4, max_stride=64, output_stride=4,
                                               stem_filters=128, filters=256, filter_increase=128, stacks=3)
    layers = cfg["config"]["layers"]
    assert len(layers) == 116
    trainable = {k: v for k, v in shapes.items() if not k.endswith(("/moving_mean", "/moving_variance"))}
    assert len(trainable) == 156
    assert sum(int(np.prod(v)) for v in trainable.values()) == 65969408
    assert sum(int(np.prod(v)) for v in shapes.values()) == 66002944
    assert [o[0] for o in cfg["config"]["output_layers"]] == [f"stack{i}_dec3_skip_add" for i in range(3)]
    names = {l["name"] for l in layers}
    for n in ("stem0_conv7x7_conv", "stem0_conv7x7_bn", "stem0_conv3x3_conv", "stem0_pool", "stem0_conv3x3_out_bn",
              "stack0_enc0_pool", "stack0_enc3_conv_conv", "stack2_dec0_conv_bn", "stack1_dec2_nearest",
              "stack1_dec2_skip_conv", "stack2_dec3_skip_add"):
        assert n in names, n
    by = {l["name"]: l for l in layers}
    assert by["stem0_conv7x7_conv"]["config"]["strides"] == [2, 2] and by["stem0_conv7x7_conv"]["config"]["kernel_size"] == [7, 7]
    # last decoder block of EVERY stack takes its skip from the stem output (encoder_decoder.py:655-664)
    for i in range(3):
        assert by[f"stack{i}_dec3_skip_conv"]["inbound_nodes"][0][0][0] == "stem0_conv3x3_out_bn"
    # small instance executes in the oracle with the right output shapes
    cfg, shapes = build_hourglass_model_config((64, 64, 1), 4, 16, 4, 8, 16, 8, stacks=2)
    outs = KerasGraph(cfg, he_normal_weights(shapes, 0))(np.zeros((1, 64, 64, 1), np.float32))
    assert [o.shape for o in outs] == [(1, 16, 16, 16)] * 2


def test_hourglass_heads_and_errors():
    from sleap_amd.nn.architectures import build_hourglass_model_config

    cfg, shapes = build_hourglass_model_config((128, 128, 1), 4, 32, 4, 16, 32, 16, stacks=1,
                                               heads=[("MultiInstanceConfmapsHead", 13, 4), ("PartAffinityFieldsHead", 24, 8)])
    assert shapes["MultiInstanceConfmapsHead/kernel"] == (1, 1, 32, 13)
    by = {l["name"]: l for l in cfg["config"]["layers"]}
    # stride-8 head reads the input of the last decoder block (make_decoder's intermediate features)
    assert by["PartAffinityFieldsHead"]["inbound_nodes"][0][0][0] == "stack0_dec1_skip_add"
    assert shapes["PartAffinityFieldsHead/kernel"] == (1, 1, 48, 24)
    with pytest.raises(ValueError, match="used 2 times"):
        build_hourglass_model_config((128, 128, 1), stacks=2, heads=[("MultiInstanceConfmapsHead", 13, 4)])
    with pytest.raises(ValueError, match="symmetric"):
        build_hourglass_model_config((128, 128, 1), stem_stride=4, max_stride=64, output_stride=8, stacks=2)
