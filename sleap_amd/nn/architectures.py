"""Builds the Keras functional-graph description of SLEAP's UNet and (stacked) hourglass backbones + heads.

Mirrors, as plain data (layer list with the reference's layer names and wiring), what the reference
builds with Keras objects:
    sleap/nn/architectures/unet.py:46-278             UNet (stem / encoder / decoder stacks, from_config)
    sleap/nn/architectures/encoder_decoder.py:57-676  SimpleConvBlock, SimpleUpsamplingBlock, make_backbone
    sleap/nn/architectures/hourglass.py:17-316        conv (Conv+ReLU+BN), StemBlock, Downsampling/UpsamplingBlock
    sleap/nn/architectures/resnet.py:46-541           make_resnet_model, block_v1, stack_v1, make_backbone_fn, ResNetv1
    sleap/nn/architectures/upsampling.py:118-259      UpsamplingStack.make_stack
    sleap/nn/heads.py:42-62                           Head.make_head (1x1 linear Conv2D named after the head class)
    sleap/nn/model.py:312-364                         Model.make_model (heads attach at matching stride)
The result has the same schema as `json.loads(h5.attrs["model_config"])` so the engine, the
oracle and a real `best_model.h5` all go through one code path. Used for randomly initialised
benchmark models (there are no trained 1024x1024 checkpoints offline).
"""
import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


class _G:
    def __init__(self):
        self.layers = []

    def add(self, class_name, name, config, inputs):
        cfg = dict(config)
        cfg["name"] = name
        self.layers.append({
            "class_name": class_name, "name": name, "config": cfg,
            "inbound_nodes": [[[i, 0, 0, {}] for i in inputs]] if inputs else [],
        })
        return name


def _conv(g, x, name, filters, k=3, activation="linear"):
    return g.add("Conv2D", name, {"filters": int(filters), "kernel_size": [k, k], "strides": [1, 1], "padding": "same",
                                  "activation": activation, "use_bias": True, "dilation_rate": [1, 1]}, [x])


def _relu(g, x, name):
    return g.add("Activation", name, {"activation": "relu"}, [x])


def build_unet_model_config(input_shape: Tuple[int, int, int], filters: int = 16, filters_rate: float = 2.0,
                            max_stride: int = 32, output_stride: int = 4, middle_block: bool = True,
                            up_interpolate: bool = True, stem_stride: Optional[int] = None,
                            heads: Sequence[Tuple[str, int, int]] = (), stacks: int = 1, stem_kernel_size: int = 7,
                            stem_blocks: Optional[int] = None, down_blocks: Optional[int] = None,
                            up_blocks: Optional[int] = None,
                            legacy_head_suffix: bool = False) -> Tuple[dict, Dict[str, tuple]]:
    """`heads` = [(head_class_name, channels, output_stride), ...] in model-output order.

    Heads on a stacked backbone: Model.make_model (model.py:336-359) makes one head per stack output with the SAME layer name,
    which Keras rejects for stacks > 1 -- the same ValueError is raised here. `legacy_head_suffix=True` names the head of stack
    s `<head>_<s>` instead (one output per head and stack, stack-major inside a head), as build_hourglass_model_config does.

    `stem_stride` / `max_stride` / `output_stride` / `stacks` are UNetConfig's fields (UNet.from_config, unet.py:250-278:
    stem_blocks = log2(stem_stride), down_blocks = log2(max_stride) - stem_blocks, up_blocks = log2(max_stride / output_stride),
    stem_kernel_size 7); `stem_blocks` / `down_blocks` / `up_blocks` / `stem_kernel_size` override them for graphs built from
    the UNet class directly (what tests/nn/architectures/test_unet.py does).

    Stem (unet.py:105-127, encoder_decoder.py:484-505): `stem_blocks` SimpleConvBlocks of two k x k convs ("stem{i}_conv{j}",
    pooling BEFORE the convs from the second block on) + a pooling-only block "stem{n}_last_pool"; built once, not repeated
    per stack; its output is the FIRST skip source at its stride (make_decoder takes the first source with a matching stride,
    encoder_decoder.py:589-593, so the same-stride output of encoder block 0 is not used as a skip).

    Returns (model_config, weight_shapes) where weight_shapes maps "<layer>/kernel|bias" to shapes.
    """
    convs_per_block, kernel = 2, 3  # UNet.from_config fixes these (unet.py:266-278)
    if stem_blocks is None:
        stem_blocks = 0 if stem_stride is None else int(math.log2(stem_stride))
    if down_blocks is None:
        down_blocks = int(math.log2(max_stride)) - stem_blocks
    if up_blocks is None:
        up_blocks = int(math.log2(max_stride / output_stride))
    if stacks > 1 and down_blocks != up_blocks:  # encoder_decoder.py:632-639
        raise ValueError("If using a stacked configuration, the backbone must define symmetric encoder and decoder. "
                         "Create a stem for initial downsampling if an output stride > 1 is desired.")
    if stacks > 1 and heads and not legacy_head_suffix:
        raise ValueError(f'The name "{heads[0][0]}" is used {stacks} times in the model. All layer names should be unique.')
    g = _G()
    shapes = {}
    x = g.add("InputLayer", "input", {"batch_input_shape": [None, input_shape[0], input_shape[1], input_shape[2]]}, [])
    cur_c = input_shape[2]

    def conv(x, name, f, cin, k=3):
        shapes[f"{name}/kernel"] = (k, k, cin, int(f))
        shapes[f"{name}/bias"] = (int(f),)
        return _conv(g, x, name, f, k)

    def pool(x, name):
        return g.add("MaxPooling2D", name, {"pool_size": [2, 2], "strides": [2, 2], "padding": "same"}, [x])

    # ---- stem
    stride = 1
    stem_skip = None
    if stem_blocks > 0:
        for block in range(stem_blocks):
            f = int(filters * (filters_rate ** block))
            if block > 0:
                x = pool(x, f"stem{block}_pool")
                stride *= 2
            for i in range(convs_per_block):
                x = conv(x, f"stem{block}_conv{i}", f, cur_c, k=stem_kernel_size)
                cur_c = f
                x = _relu(g, x, f"stem{block}_act{i}_relu")
        x = pool(x, f"stem{stem_blocks}_last_pool")
        stride *= 2
        stem_skip = (stride, x, cur_c)
    stem_out_stride = stride
    stack_outs = []
    mids_per_stack = []
    for st in range(stacks):
        # ---- encoder (unet.py:136-205; SimpleConvBlock.make_block encoder_decoder.py:92-144)
        prefix = f"stack{st}_enc"
        stride = stem_out_stride
        skips = {}  # stride -> (tensor name, channels); the FIRST source of a stride wins
        if stem_skip is not None:
            skips[stem_skip[0]] = (stem_skip[1], stem_skip[2])
        enc_feats = []
        for block in range(down_blocks):
            f = int(filters * (filters_rate ** (block + stem_blocks)))
            if block > 0:
                x = pool(x, f"{prefix}{block}_pool")
                stride *= 2
            for i in range(convs_per_block):
                x = conv(x, f"{prefix}{block}_conv{i}", f, cur_c)
                cur_c = f
                x = _relu(g, x, f"{prefix}{block}_act{i}_relu")
            if stride not in [q[0] for q in enc_feats]:
                enc_feats.append((stride, x, cur_c))
        x = pool(x, f"{prefix}{down_blocks}_last_pool")
        stride *= 2
        for s_, t_, c_ in enc_feats:  # (make_encoder drops the LAST feature: the pooled / middle output is the decoder input)
            skips.setdefault(s_, (t_, c_))
        bi = down_blocks + 1
        if middle_block:
            f = int(filters * (filters_rate ** (down_blocks + stem_blocks)))
            if convs_per_block > 1:
                name = f"{prefix}{bi}_middle_expand"
                x = conv(x, f"{name}_conv0", f, cur_c)
                cur_c = f
                x = _relu(g, x, f"{name}_act0_relu")
                bi += 1
            name = f"{prefix}{bi}_middle_contract"
            x = conv(x, f"{name}_conv0", f, cur_c)
            cur_c = f
            x = _relu(g, x, f"{name}_act0_relu")
        # ---- decoder (unet.py:207-247; SimpleUpsamplingBlock.make_block encoder_decoder.py:275-399)
        mids = {stride: (x, cur_c)}
        mids_per_stack.append(mids)
        for block in range(up_blocks):
            f = int(filters * (filters_rate ** (down_blocks + stem_blocks - 1 - block)))
            nxt = stride // 2
            name = f"stack{st}_dec{block}_s{stride}_to_s{nxt}"
            if up_interpolate:
                x = g.add("UpSampling2D", f"{name}_interp_bilinear", {"size": [2, 2], "interpolation": "bilinear"}, [x])
            else:
                shapes[f"{name}_trans_conv/kernel"] = (kernel, kernel, int(f), cur_c)
                shapes[f"{name}_trans_conv/bias"] = (int(f),)
                x = g.add("Conv2DTranspose", f"{name}_trans_conv",
                          {"filters": int(f), "kernel_size": [3, 3], "strides": [2, 2], "padding": "same",
                           "activation": "linear", "use_bias": True, "dilation_rate": [1, 1], "output_padding": None}, [x])
                cur_c = f
                x = _relu(g, x, f"{name}_trans_conv_act_relu")
            if nxt in skips:
                sk, sc = skips[nxt]
                x = g.add("Concatenate", f"{name}_skip_concat", {"axis": -1}, [sk, x])
                cur_c += sc
            for i in range(convs_per_block):
                x = conv(x, f"{name}_refine_conv{i}", f, cur_c)
                cur_c = f
                x = _relu(g, x, f"{name}_refine_conv{i}_act_relu")
            stride = nxt
            mids[stride] = (x, cur_c)
        stack_outs.append(x)
    # ---- heads (model.py:336-359): main output if strides match, else the decoder feature of that stride
    outs = []
    for head_name, channels, hs in heads:
        for si in (range(stacks) if legacy_head_suffix else [stacks - 1]):
            if hs not in mids_per_stack[si]:
                raise ValueError(f"Could not find a feature activation for output at stride {hs}.")
            src, sc = mids_per_stack[si][hs]
            name = f"{head_name}_{si}" if legacy_head_suffix else head_name
            shapes[f"{name}/kernel"] = (1, 1, sc, int(channels))
            shapes[f"{name}/bias"] = (int(channels),)
            outs.append(_conv(g, src, name, channels, k=1))
    if not heads:  # a backbone graph (what the architecture tests build): the stack outputs
        outs = stack_outs if stacks > 1 else stack_outs[:1]
    cfg = {"class_name": "Functional",
           "config": {"name": "model", "layers": g.layers, "input_layers": [["input", 0, 0]],
                      "output_layers": [[o, 0, 0] for o in outs]}}
    return cfg, shapes


def build_hourglass_model_config(input_shape: Tuple[int, int, int], stem_stride: int = 4, max_stride: int = 64,
                                 output_stride: int = 4, stem_filters: int = 128, filters: int = 256,
                                 filter_increase: int = 128, stacks: int = 3, interp_method: str = "nearest",
                                 heads: Sequence[Tuple[str, int, int]] = (),
                                 legacy_head_suffix: bool = False) -> Tuple[dict, Dict[str, tuple]]:
    """Hourglass.from_config (hourglass.py:299-316) + make_backbone (encoder_decoder.py:606-676) + Model.make_model.

    Without `heads` the model outputs are the stack outputs (what tests/nn/architectures/test_hourglass.py builds).
    With `heads`, every head attaches to every stack output under the same layer name (model.py:336-359), which
    Keras rejects for stacks > 1; the same ValueError is raised here. `legacy_head_suffix=True` names the head of stack s
    `<head>_<s>` instead -- the naming of models saved by older SLEAP releases (the reference's own fixture models carry
    `MultiInstanceConfmapsHead_0`), the only way a stacked model with heads reaches inference: outputs in
    Model.make_model's order (per head: stack 0, stack 1, ...), and `find_head` (inference.py:1204-1226) then picks the FIRST
    match, i.e. stack 0's head.
    """
    if stem_stride not in (2, 4):
        raise NotImplementedError("hourglass stem_stride must be 2 or 4 (stride-1 'same' max pooling is not implemented)")
    stem_blocks = int(math.log2(stem_stride))
    down_blocks = int(math.log2(max_stride)) - stem_blocks
    up_blocks = int(math.log2(max_stride / output_stride))
    if stacks > 1 and down_blocks != up_blocks:
        raise ValueError("If using a stacked configuration, the backbone must define symmetric encoder and decoder. "
                         "Create a stem for initial downsampling if an output stride > 1 is desired.")
    if stacks > 1 and heads and not legacy_head_suffix:
        raise ValueError(f'The name "{heads[0][0]}" is used {stacks} times in the model. All layer names should be unique.')
    g = _G()
    shapes: Dict[str, tuple] = {}
    x = g.add("InputLayer", "input", {"batch_input_shape": [None, input_shape[0], input_shape[1], input_shape[2]]}, [])

    def conv(x, cin, f, prefix, k=3, stride=1):  # hourglass.py:17-45
        shapes[f"{prefix}_conv/kernel"] = (k, k, int(cin), int(f))
        shapes[f"{prefix}_conv/bias"] = (int(f),)
        for v in ("gamma", "beta", "moving_mean", "moving_variance"):
            shapes[f"{prefix}_bn/{v}"] = (int(f),)
        x = g.add("Conv2D", prefix + "_conv", {"filters": int(f), "kernel_size": [k, k], "strides": [stride, stride],
                                                "padding": "same", "activation": "relu", "use_bias": True,
                                                "dilation_rate": [1, 1]}, [x])
        return g.add("BatchNormalization", prefix + "_bn", {"axis": [3], "momentum": 0.99, "epsilon": 0.001,
                                                             "center": True, "scale": True}, [x])

    def pool(x, name, stride=2):
        return g.add("MaxPooling2D", name, {"pool_size": [2, 2], "strides": [stride, stride], "padding": "same"}, [x])

    # ---- stem (StemBlock.make_block, hourglass.py:73-103)
    x = conv(x, input_shape[2], stem_filters, "stem0_conv7x7", k=7, stride=2 if stem_stride == 4 else 1)
    x = conv(x, stem_filters, 2 * stem_filters, "stem0_conv3x3")
    x = pool(x, "stem0_pool")
    x = conv(x, 2 * stem_filters, filters, "stem0_conv3x3_out")
    cur_c = filters
    stem_out = (x, cur_c)
    outs, mids_per_stack = [], []
    for s in range(stacks):
        stride = stem_stride
        feats = []  # encoder IntermediateFeatures (stride, tensor, channels)
        for i in range(down_blocks):  # DownsamplingBlock (hourglass.py:123-139)
            f = filters + i * filter_increase
            x = pool(x, f"stack{s}_enc{i}_pool")
            x = conv(x, cur_c, f, f"stack{s}_enc{i}_conv")
            cur_c, stride = f, stride * 2
            feats.append((stride, x, cur_c))
        sources = [(stem_stride,) + stem_out] + feats[:-1]
        mids = {}
        for i in range(up_blocks):  # UpsamplingBlock (hourglass.py:162-191)
            mids.setdefault(stride, (x, cur_c))
            f = filters + (down_blocks - i - 1) * filter_increase
            nxt = stride // 2
            skip = next(((t, c) for (st, t, c) in sources if st == nxt), None)
            if skip is None:
                raise ValueError(f"hourglass decoder block {i} has no skip source at stride {nxt}")
            pre = f"stack{s}_dec{i}"
            x = conv(x, cur_c, f, pre + "_conv")
            x = g.add("UpSampling2D", f"{pre}_{interp_method}", {"size": [2, 2], "interpolation": interp_method}, [x])
            xs = conv(skip[0], skip[1], f, pre + "_skip")
            x = g.add("Add", pre + "_skip_add", {}, [x, xs])
            cur_c, stride = f, nxt
        outs.append((x, cur_c, stride))
        mids_per_stack.append(mids)
    out_layers = [o[0] for o in outs]
    if heads:
        out_layers = []
        for head_name, channels, hs in heads:
            for si in range(stacks if legacy_head_suffix else 1):
                x, c, stride = outs[si]
                if hs == stride:
                    src, sc = x, c
                elif hs in mids_per_stack[si]:
                    src, sc = mids_per_stack[si][hs]
                else:
                    raise ValueError(f"Could not find a feature activation for output at stride {hs}.")
                name = f"{head_name}_{si}" if legacy_head_suffix else head_name
                shapes[f"{name}/kernel"] = (1, 1, int(sc), int(channels))
                shapes[f"{name}/bias"] = (int(channels),)
                out_layers.append(_conv(g, src, name, channels, k=1))
    cfg = {"class_name": "Functional",
           "config": {"name": "model", "layers": g.layers, "input_layers": [["input", 0, 0]],
                      "output_layers": [[o, 0, 0] for o in out_layers]}}
    return cfg, shapes


RESNET_STACKS = {  # resnet.py:585-588, 639-642, 693-696
    "ResNet50": [(64, 3, 1, "conv2"), (128, 4, 2, "conv3"), (256, 6, 2, "conv4"), (512, 3, 2, "conv5")],
    "ResNet101": [(64, 3, 1, "conv2"), (128, 4, 2, "conv3"), (256, 23, 2, "conv4"), (512, 3, 2, "conv5")],
    "ResNet152": [(64, 3, 1, "conv2"), (128, 8, 2, "conv3"), (256, 36, 2, "conv4"), (512, 3, 2, "conv5")],
}


def build_resnet_model_config(input_shape: Tuple[int, int, int], version: str = "ResNet50",
                              features_output_stride: int = 32, pretrained: bool = False,
                              upsampling: Optional[dict] = None,
                              heads: Sequence[Tuple[str, int, int]] = ()) -> Tuple[dict, Dict[str, tuple]]:
    """ResNetv1.make_backbone (resnet.py:467-541) [+ UpsamplingStack.make_stack] + Model.make_model.

    `pretrained=True` adds the `tile_channels` (grayscale input) and `imagenet_preproc_v1` Lambda layers (weights are
    still caller supplied; there is no download). `upsampling` = UpsamplingConfig fields as a dict: `output_stride`,
    `method` ("transposed_conv" | "interpolation"), `skip_connections` (None | "add" | "concatenate"), `block_stride`,
    `filters`, `filters_rate`, `refine_convs`, `batch_norm`, `transposed_conv_kernel_size` (config/model.py:515-559).
    Without `heads` the output is the backbone / upsampling-stack output.
    """
    if version not in RESNET_STACKS:
        raise ValueError(f"Invalid ResNet version in the configuration: {version}")
    g = _G()
    shapes: Dict[str, tuple] = {}
    x = g.add("InputLayer", "input", {"batch_input_shape": [None, input_shape[0], input_shape[1], input_shape[2]]}, [])
    cur_c = input_shape[2]
    if pretrained:
        if cur_c == 1:
            x = g.add("Lambda", "tile_channels", {"function": "tile_channels"}, [x])
            cur_c = 3
        x = g.add("Lambda", "imagenet_preproc_v1", {"function": "imagenet_preproc_v1"}, [x])

    def conv(x, name, cin, f, k, stride=1, padding="valid", dilation=1):
        shapes[f"{name}/kernel"] = (k, k, int(cin), int(f))
        shapes[f"{name}/bias"] = (int(f),)
        return g.add("Conv2D", name, {"filters": int(f), "kernel_size": [k, k], "strides": [stride, stride],
                                      "padding": padding, "activation": "linear", "use_bias": True,
                                      "dilation_rate": [dilation, dilation]}, [x])

    def bn(x, name, c, eps):
        for v in ("gamma", "beta", "moving_mean", "moving_variance"):
            shapes[f"{name}/{v}"] = (int(c),)
        return g.add("BatchNormalization", name, {"axis": [3], "momentum": 0.99, "epsilon": eps, "center": True,
                                                   "scale": True}, [x])

    EPS = 1.001e-5
    stem_stride1 = 1 if features_output_stride == 1 else 2
    stem_stride2 = 1 if features_output_stride <= 2 else 2
    # ---- stem (make_resnet_model, resnet.py:109-129)
    x = g.add("ZeroPadding2D", "conv1_pad", {"padding": [[3, 3], [3, 3]]}, [x])
    x = conv(x, "conv1_conv", cur_c, 64, 7, stem_stride1)
    x = bn(x, "conv1_bn", 64, EPS)
    x = _relu(g, x, "conv1_relu")
    cur_c = 64
    feats = [(stem_stride1, x, cur_c)]
    x = g.add("ZeroPadding2D", "pool1_pad", {"padding": [[1, 1], [1, 1]]}, [x])
    x = g.add("MaxPooling2D", "pool1_pool", {"pool_size": [3, 3], "strides": [stem_stride2, stem_stride2],
                                             "padding": "valid"}, [x])
    stride = stem_stride1 * stem_stride2
    feats.append((stride, x, cur_c))

    def block(x, cin, f, name, s=1, dilation=1, conv_shortcut=True):  # block_v1, resnet.py:168-229
        if conv_shortcut:
            sc = conv(x, name + "_0_conv", cin, 4 * f, 1, s, dilation=dilation)
            sc = bn(sc, name + "_0_bn", 4 * f, EPS)
        else:
            sc = x
        y = conv(x, name + "_1_conv", cin, f, 1, s, dilation=dilation)
        y = bn(y, name + "_1_bn", f, EPS)
        y = _relu(g, y, name + "_1_relu")
        y = conv(y, name + "_2_conv", f, f, 3, padding="same")
        y = bn(y, name + "_2_bn", f, EPS)
        y = _relu(g, y, name + "_2_relu")
        y = conv(y, name + "_3_conv", f, 4 * f, 1)
        y = bn(y, name + "_3_bn", 4 * f, EPS)
        y = g.add("Add", name + "_add", {}, [sc, y])
        return _relu(g, y, name + "_out")

    # ---- residual stacks with stride -> dilation conversion (make_backbone_fn, resnet.py:289-321)
    dilation = 1
    for f, blocks, stride1, name in RESNET_STACKS[version]:
        if stride < features_output_stride:
            stride *= stride1
            s1 = stride1
        elif stride == features_output_stride:
            s1 = 1
            if stride1 > 1:
                dilation *= 2
        else:
            raise ValueError(f"Could not adjust output stride. Current: {stride}, desired: {features_output_stride}")
        x = block(x, cur_c, f, name + "_block1", s=s1, dilation=dilation)
        cur_c = 4 * f
        for i in range(2, blocks + 1):
            x = block(x, cur_c, f, f"{name}_block{i}", conv_shortcut=False)
        feats.append((stride, x, cur_c))
    mids = {}
    out_stride = features_output_stride
    if upsampling is not None:  # UpsamplingStack.make_stack, upsampling.py:118-259
        u = dict(method="interpolation", skip_connections=None, block_stride=2, filters=64, filters_rate=1, refine_convs=2,
                 batch_norm=True, transposed_conv_kernel_size=4)
        u.update(upsampling)
        us = u["block_stride"]
        skip_sources = feats[2:] if u["skip_connections"] is not None else None
        num_blocks = int((math.log(stride) - math.log(u["output_stride"])) / math.log(us))
        mids[stride] = (x, cur_c)
        for blk in range(num_blocks):
            new_stride = stride // us
            pre = f"upsample_s{stride}_to_s{new_stride}"
            if u["method"] == "transposed_conv":
                f = int(u["filters"] * u["filters_rate"] ** blk)
                k = u["transposed_conv_kernel_size"]
                shapes[f"{pre}_trans_conv/kernel"] = (k, k, f, int(cur_c))
                shapes[f"{pre}_trans_conv/bias"] = (f,)
                x = g.add("Conv2DTranspose", pre + "_trans_conv",
                          {"filters": f, "kernel_size": [k, k], "strides": [us, us], "padding": "same",
                           "activation": "linear", "use_bias": True, "dilation_rate": [1, 1], "output_padding": None}, [x])
                cur_c = f
                if u["batch_norm"]:
                    x = bn(x, pre + "_bn", cur_c, 1e-3)
                x = _relu(g, x, pre + "_relu")
            else:
                x = g.add("UpSampling2D", pre + "_interp", {"size": [us, us], "interpolation": "bilinear"}, [x])
            stride = new_stride
            if skip_sources is not None:
                src = next(((t, c) for (st, t, c) in skip_sources if st == stride), None)
                if src is not None:
                    if u["skip_connections"] == "add":
                        sx, sc = src
                        if sc != cur_c:
                            sx = conv(sx, pre + "_skip_conv1x1", sc, cur_c, 1, padding="same")
                        x = g.add("Add", pre + "_skip_add", {}, [sx, x])
                    else:
                        x = g.add("Concatenate", pre + "_skip_concat", {"axis": -1}, [src[0], x])
                        cur_c += src[1]
            f = int(u["filters"] * u["filters_rate"] ** blk)
            for i in range(u["refine_convs"]):
                x = conv(x, f"{pre}_refine{i}_conv", cur_c, f, 3, padding="same")
                cur_c = f
                if u["batch_norm"]:
                    x = bn(x, f"{pre}_refine{i}_bn", cur_c, 1e-3)
                x = _relu(g, x, f"{pre}_refine{i}_relu")
            mids[stride] = (x, cur_c)
        out_stride = stride
    else:
        for st, t, c in feats:
            mids.setdefault(st, (t, c))
    out_layers = [x]
    if heads:
        out_layers = []
        for head_name, channels, hs in heads:
            if hs == out_stride:
                src, sc = x, cur_c
            elif hs in mids:
                src, sc = mids[hs]
            else:
                raise ValueError(f"Could not find a feature activation for output at stride {hs}.")
            shapes[f"{head_name}/kernel"] = (1, 1, int(sc), int(channels))
            shapes[f"{head_name}/bias"] = (int(channels),)
            out_layers.append(_conv(g, src, head_name, channels, k=1))
    cfg = {"class_name": "Functional",
           "config": {"name": "model", "layers": g.layers, "input_layers": [["input", 0, 0]],
                      "output_layers": [[o, 0, 0] for o in out_layers]}}
    return cfg, shapes


def he_normal_weights(shapes: Dict[str, tuple], seed: int = 0, residual_scale: float = 1.0) -> Dict[str, np.ndarray]:
    """Deterministic He-normal kernels / small biases (random-init weights for benchmarking).

    `residual_scale` multiplies the last conv of every ResNet bottleneck block (`*_block<i>_3_conv`). With 1.0 and these
    near-identity BatchNormalization statistics the residual stream doubles its variance per block (ResNet-50 on x255
    ImageNet-range inputs: activations 5e2 after the stem, > 6.5e4 = fp16's range from conv4_block3 on); ~0.25 keeps a
    random-init ResNet in the range a trained one lives in (the usual "zero-init the last BN of each block" practice)."""
    import re

    rng = np.random.default_rng(seed)
    w = {}
    for k in sorted(shapes):
        s = shapes[k]
        if k.endswith("/kernel"):
            fan_in = s[0] * s[1] * (s[3] if "trans_conv" in k else s[2])
            w[k] = (rng.standard_normal(s) * math.sqrt(2.0 / fan_in)).astype(np.float32)
            if residual_scale != 1.0 and re.search(r"_block\d+_3_conv/kernel$", k):
                w[k] *= np.float32(residual_scale)
        elif k.endswith("/gamma"):
            w[k] = (1.0 + 0.1 * rng.standard_normal(s)).astype(np.float32)
        elif k.endswith("/moving_variance"):
            w[k] = rng.uniform(0.5, 1.5, s).astype(np.float32)
        elif k.endswith(("/beta", "/moving_mean")):
            w[k] = (rng.standard_normal(s) * 0.1).astype(np.float32)
        else:
            w[k] = (rng.standard_normal(s) * 0.01).astype(np.float32)
    return w


def unet_from_training_config(cfg: dict, input_shape, n_nodes=None, n_edges=None):
    """Model.from_config + make_model for a `multi_instance` (bottom-up) training config dict."""
    u = cfg["model"]["backbone"]["unet"]
    mi = cfg["model"]["heads"]["multi_instance"]
    nodes = mi["confmaps"].get("part_names") or [None] * n_nodes
    edges = mi["pafs"].get("edges") or [None] * n_edges
    heads = [("MultiInstanceConfmapsHead", len(nodes), mi["confmaps"]["output_stride"]),
             ("PartAffinityFieldsHead", 2 * len(edges), mi["pafs"]["output_stride"])]
    if mi["confmaps"].get("offset_refinement"):
        heads.append(("OffsetRefinementHead", 2 * len(nodes), mi["confmaps"]["output_stride"]))
    return build_unet_model_config(input_shape, u["filters"], u["filters_rate"], u["max_stride"], u["output_stride"],
                                   u.get("middle_block", True), u.get("up_interpolate", True), u.get("stem_stride"),
                                   heads)
