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


def test_stacked_unet_with_stem_reference_counts():
    """tests/nn/architectures/test_unet.py:159-196 (`test_stacked_unet_with_stem`): UNet(stem_blocks=2, stacks=3, filters=16,
    filters_rate=2, kernel_size=3, down_blocks=3, up_blocks=3, up_interpolate=True) on a 160 x 160 x 1 input -> 122 layers,
    92 trainable weights, 23 396 592 parameters, three stack outputs of 64 channels at stride 4 (:187-196)."""
    cfg, shapes = build_unet_model_config((160, 160, 1), filters=16, filters_rate=2, middle_block=True, up_interpolate=True,
                                          stacks=3, stem_blocks=2, down_blocks=3, up_blocks=3, stem_kernel_size=3)
    assert len(cfg["config"]["layers"]) == 122
    assert len(shapes) == 92
    assert sum(int(np.prod(s)) for s in shapes.values()) == 23396592
    assert len(cfg["config"]["output_layers"]) == 3
    names = [l["name"] for l in cfg["config"]["layers"]]
    # the stem is built once (not per stack), ends with a pooling-only block, and pools BEFORE the convs from block 1 on
    assert [n for n in names if n.startswith("stem")] == [
        "stem0_conv0", "stem0_act0_relu", "stem0_conv1", "stem0_act1_relu", "stem1_pool", "stem1_conv0", "stem1_act0_relu",
        "stem1_conv1", "stem1_act1_relu", "stem2_last_pool"]
    g = KerasGraph(cfg, he_normal_weights(shapes))
    outs = g(np.zeros((1, 160, 160, 1), np.float32))
    assert [o.shape for o in outs] == [(1, 40, 40, 64)] * 3  # stride 4 (the stem's), filters * rate^(stem_blocks + 0)
    # the decoder block that returns to the stem's stride concatenates the STEM output, not encoder block 0's
    # (make_decoder: the first skip source with a matching stride, encoder_decoder.py:589-593)
    by = {l["name"]: l for l in cfg["config"]["layers"]}
    cat = by["stack0_dec2_s8_to_s4_skip_concat"]
    assert [n[0] for n in cat["inbound_nodes"][0]] == ["stem2_last_pool", "stack0_dec2_s8_to_s4_interp_bilinear"]
    with pytest.raises(ValueError, match="symmetric"):
        build_unet_model_config((160, 160, 1), stacks=2, stem_blocks=0, down_blocks=3, up_blocks=2)


def test_unet_from_config_with_stem_stride():
    """UNet.from_config (unet.py:250-278) with stem_stride=4, max_stride=32, output_stride=4: 2 stem blocks with 7 x 7 kernels,
    3 down blocks whose filters continue the progression (filters * rate^(block + stem_blocks)), 3 up blocks."""
    cfg, shapes = build_unet_model_config((128, 128, 1), filters=8, filters_rate=2, max_stride=32, output_stride=4,
                                          stem_stride=4, heads=[("SingleInstanceConfmapsHead", 3, 4)])
    assert shapes["stem0_conv0/kernel"] == (7, 7, 1, 8) and shapes["stem1_conv1/kernel"] == (7, 7, 16, 16)
    assert shapes["stack0_enc0_conv0/kernel"] == (3, 3, 16, 32) and shapes["stack0_enc2_conv1/kernel"] == (3, 3, 128, 128)
    assert shapes["stack0_enc4_middle_expand_conv0/kernel"] == (3, 3, 128, 256)
    assert shapes["SingleInstanceConfmapsHead/kernel"] == (1, 1, 32, 3)
    out = KerasGraph(cfg, he_normal_weights(shapes))(np.zeros((1, 128, 128, 1), np.float32))[0]
    assert out.shape == (1, 32, 32, 3)


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


def test_stacked_unet_heads_and_errors():
    """A stacked UNet with heads: Keras' duplicate-layer-name ValueError (model.py:336-359 names every stack's head alike);
    `legacy_head_suffix=True` gives one `<head>_<s>` output per stack (this used to be a NameError)."""
    from sleap_amd.nn.architectures import build_unet_model_config

    kw = dict(filters=8, stacks=2, down_blocks=2, up_blocks=2)
    with pytest.raises(ValueError, match="used 2 times"):
        build_unet_model_config((64, 64, 1), heads=[("H", 3, 1)], **kw)
    cfg, shapes = build_unet_model_config((64, 64, 1), heads=[("H", 3, 1), ("P", 4, 2)], legacy_head_suffix=True, **kw)
    assert [o[0] for o in cfg["config"]["output_layers"]] == ["H_0", "H_1", "P_0", "P_1"]
    by = {l["name"]: l for l in cfg["config"]["layers"]}
    assert by["H_0"]["inbound_nodes"][0][0][0].startswith("stack0_dec1")
    assert by["H_1"]["inbound_nodes"][0][0][0].startswith("stack1_dec1")
    assert by["P_1"]["inbound_nodes"][0][0][0].startswith("stack1_dec0")
    assert shapes["H_1/kernel"] == (1, 1, 8, 3) and shapes["P_0/kernel"] == (1, 1, 16, 4)
    # one stack: unchanged, unsuffixed
    cfg1, _ = build_unet_model_config((64, 64, 1), filters=8, down_blocks=2, up_blocks=2, heads=[("H", 3, 1)])
    assert [o[0] for o in cfg1["config"]["output_layers"]] == ["H"]


def _counts(cfg, shapes):
    tr = {k: v for k, v in shapes.items() if not k.endswith(("/moving_mean", "/moving_variance"))}
    return (len(cfg["config"]["layers"]), len(tr), sum(int(np.prod(v)) for v in tr.values()),
            sum(int(np.prod(v)) for v in shapes.values()))


def test_resnet_reference_structure():
    """tests/nn/architectures/test_resnet.py: layer / weight / parameter counts of ResNet50/101/152 on (160,160,1)."""
    from sleap_amd.nn.architectures import build_resnet_model_config

    assert _counts(*build_resnet_model_config((160, 160, 1), "ResNet50", 32)) == (175, 212, 23528320, 23581440)
    assert _counts(*build_resnet_model_config((160, 160, 1), "ResNet50", 16)) == (175, 212, 23528320, 23581440)
    # pretrained adds the two preprocessing Lambdas and a 3-channel stem kernel
    assert _counts(*build_resnet_model_config((160, 160, 1), "ResNet50", 16, pretrained=True)) == (177, 212, 23534592, 23587712)
    assert _counts(*build_resnet_model_config((160, 160, 1), "ResNet101", 16)) == (345, 416, 42546560, 42651904)
    assert _counts(*build_resnet_model_config((160, 160, 1), "ResNet152", 16)) == (515, 620, 58213248, 58364672)
    cfg, _ = build_resnet_model_config((160, 160, 1), "ResNet50", 16)
    by = {l["name"]: l for l in cfg["config"]["layers"]}
    assert by["conv5_block1_1_conv"]["config"]["strides"] == [1, 1]
    assert by["conv5_block1_1_conv"]["config"]["dilation_rate"] == [2, 2]
    assert by["conv4_block1_1_conv"]["config"]["strides"] == [2, 2]
    with pytest.raises(ValueError, match="Invalid ResNet version"):
        build_resnet_model_config((160, 160, 1), "ResNet18")


def test_resnet_output_shapes_in_oracle():
    """Output shapes asserted by the reference tests: stride-32 -> (5,5,2048), stride-16 -> (10,10,2048), upsampling
    stack to stride 4 with 64 refine filters -> (40,40,64) (test_resnet.py:30-31, 50-51, 72-73)."""
    from oracle.keras_graph import KerasGraph
    from sleap_amd.nn.architectures import build_resnet_model_config, he_normal_weights

    x = np.zeros((1, 160, 160, 1), np.float32)
    for kw, want in [(dict(features_output_stride=32), (1, 5, 5, 2048)), (dict(features_output_stride=16), (1, 10, 10, 2048)),
                     (dict(features_output_stride=32, upsampling=dict(output_stride=4, method="transposed_conv", filters=64)),
                      (1, 40, 40, 64))]:
        cfg, shapes = build_resnet_model_config((160, 160, 1), "ResNet50", **kw)
        assert KerasGraph(cfg, he_normal_weights(shapes, 0))(x)[0].shape == want


def test_upsampling_stack_structure():
    """tests/nn/architectures/test_upsampling.py: filter-rate progressions of the transposed-conv / refine layers and
    layer naming `upsample_s{a}_to_s{b}_...`."""
    from sleap_amd.nn.architectures import build_resnet_model_config

    cfg, shapes = build_resnet_model_config((64, 64, 1), "ResNet50", 16,
                                            upsampling=dict(output_stride=2, method="transposed_conv", filters=16,
                                                            filters_rate=2, refine_convs=0, batch_norm=False))
    assert shapes["upsample_s16_to_s8_trans_conv/kernel"] == (4, 4, 16, 2048)
    assert shapes["upsample_s8_to_s4_trans_conv/kernel"] == (4, 4, 32, 16)
    assert shapes["upsample_s4_to_s2_trans_conv/kernel"] == (4, 4, 64, 32)
    cfg, shapes = build_resnet_model_config((64, 64, 1), "ResNet50", 16,
                                            upsampling=dict(output_stride=4, method="interpolation", filters=16,
                                                            filters_rate=2, refine_convs=2, skip_connections="concatenate"))
    names = [l["name"] for l in cfg["config"]["layers"]]
    assert "upsample_s16_to_s8_interp" in names and "upsample_s16_to_s8_skip_concat" in names
    assert shapes["upsample_s16_to_s8_refine0_conv/kernel"] == (3, 3, 2048 + 512, 16)  # conv3 output (stride 8) is the skip
    assert shapes["upsample_s8_to_s4_refine1_conv/kernel"] == (3, 3, 32, 32)
    assert "upsample_s8_to_s4_refine1_bn/gamma" in shapes


def test_two_stack_hourglass_oracle_side_is_the_fitted_one_stack_model():
    """(no GPU) output 0 of the two-stack graph == the fitted one-stack model's head, and find_head picks it."""
    import numpy as np

    from oracle.keras_graph import KerasGraph, preprocess
    import config_models as C
    from sleap_amd.nn.inference import find_head
    from test_gpu_config_parity import _two_stack_hourglass

    frames, _ = C.render("hg_single13", 1, seed=306)
    x = preprocess(frames[:, :256, :256], pad_stride=32)
    mc, wts = _two_stack_hourglass(256, 256)
    names = [o[0] for o in mc["config"]["output_layers"]]
    assert names == ["SingleInstanceConfmapsHead_0", "SingleInstanceConfmapsHead_1"]

    class _M:
        output_names = names

    assert find_head(_M, "SingleInstanceConfmapsHead") == 0  # inference.py:1223-1226: the FIRST match = stack 0
    mc1, w1 = C.load_task_weights("hg_single13", 256, 256)
    a, b = KerasGraph(mc, wts)(x)[0], KerasGraph(mc1, w1)(x)[0]
    assert np.array_equal(a, b)
