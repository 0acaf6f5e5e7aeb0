"""torch-CPU float32 interpreter for the Keras functional graphs SLEAP saves in `best_model.h5`
(TEST INFRASTRUCTURE ONLY).

Stands in for `tf.keras.models.load_model(...)(imgs)` (sleap/nn/inference.py:3207, 2864-2890).
Pinned to TensorFlow since round 6 for Conv2D / MaxPooling2D / Conv2DTranspose(k3 s2 same) / Concatenate / 1x1 heads: the
TensorFlow-produced predictions of the reference's bottom-up and centered-instance fixtures are reproduced to 3e-5 px
(tests/test_frame0_golden.py); UpSampling2D(bilinear) and BatchNormalization rest on hand-derived vectors
(tests/layer_pin_vectors.py). Layer semantics follow the
documented TF/Keras rules collected in SURVEY.md §8(a):

  SAME padding: pad_total = max((ceil(n/s) - 1) * s + k - n, 0), pad_before = pad_total // 2
  Conv2D kernel (kh, kw, Cin, Cout), cross-correlation
  Conv2DTranspose kernel (kh, kw, Cout, Cin); k3 s2 same -> full transposed conv cropped at
    the END to 2n; general: crop pad_before = (k - s) // 2 ... (k + s) rule below
  MaxPooling2D same: -inf padding (bottom/right first when odd)
  UpSampling2D bilinear = half-pixel centres, align_corners=False
  BatchNormalization (inference): gamma * (x - mean) / sqrt(var + eps) + beta
"""
import json
import math

import os

import numpy as np
import torch
import torch.nn.functional as F


def load_npz_model(path):
    """Read the `.npz` written by sleap_amd/nn/_h5_extract.py -> (model_config dict, {layer/weight: array})."""
    z = np.load(path)
    cfg = json.loads(bytes(z["__model_config__"]).decode("utf-8"))
    weights = {k: z[k] for k in z.files if k != "__model_config__"}
    return cfg, weights


def _same_pads(n, k, s, d=1):
    keff = (k - 1) * d + 1
    out = math.ceil(n / s)
    total = max((out - 1) * s + keff - n, 0)
    return total // 2, total - total // 2


def _act(x, name):
    if name in (None, "linear"):
        return x
    if name == "relu":
        return torch.relu(x)
    if name == "sigmoid":
        return torch.sigmoid(x)
    if name == "tanh":
        return torch.tanh(x)
    raise NotImplementedError(f"activation {name}")


class KerasGraph:
    """Executes the layer list of a Keras `Functional` model config in NCHW torch tensors."""

    def __init__(self, model_config, weights, emulate_bf16=False, fp32_input_heads=(), round_layers=None, emulate_dtype=None):
        """`emulate_bf16=True` rounds weights/activations to the engine's 16-bit storage type (`emulate_dtype`) at exactly the
        points where the HIP engine stores it (conv / transposed-conv / upsample outputs, 3x3 conv weights except the fp32 stem; heads
        keep fp32 weights and outputs), with fp32 accumulation in between. It separates "the kernels compute what
        they claim" (tight tolerance against this mode) from "bf16 storage is accurate enough" (loose tolerance
        against the fp32 mode)."""
        self.emulate_bf16 = emulate_bf16
        # storage type emulated at the rounding points: torch.float16 / torch.bfloat16; None follows the same switch as the
        # engine under test (SLEAP_AMD_DTYPE, default fp16) so that one test suite checks either build of the library
        if emulate_dtype is None:
            emulate_dtype = {"fp16": torch.float16, "bf16": torch.bfloat16}[os.environ.get("SLEAP_AMD_DTYPE", "fp16")]
        self.emulate_dtype = emulate_dtype
        # with emulate_bf16: layer names after which a value is rounded to bf16. None = after every Conv2D /
        # Conv2DTranspose / bilinear UpSampling2D (the UNet engine's storage points); a set = exactly those layers
        # (an engine that fuses BatchNormalization / Add into the conv epilogue stores only the fused result)
        self.round_layers = None if round_layers is None else set(round_layers)
        # heads the engine computes from the un-rounded fp32 accumulators of the producing conv (fused epilogue)
        self.fp32_input_heads = set(fp32_input_heads)
        cfg = model_config["config"]
        self.layers = cfg["layers"]
        self.input_names = [l[0] for l in cfg["input_layers"]]
        self.output_names = [l[0] for l in cfg["output_layers"]]
        self.w = {k: torch.from_numpy(np.ascontiguousarray(v)).float() for k, v in weights.items()}

    def n_params(self):
        return int(sum(v.numel() for v in self.w.values()))

    def input_channels(self):
        for l in self.layers:
            if l["class_name"] == "InputLayer":
                return l["config"]["batch_input_shape"][-1]

    @torch.no_grad()
    def __call__(self, x_nhwc, return_all=False):
        """x_nhwc: (B, H, W, C) float32 array/tensor -> list of NHWC float32 numpy outputs."""
        x = torch.as_tensor(np.asarray(x_nhwc), dtype=torch.float32).permute(0, 3, 1, 2).contiguous()
        t = {}
        u = {}  # un-rounded twins of conv (+activation) outputs in emulate_bf16 mode
        self._x_in = x
        for l in self.layers:
            cn, name, c = l["class_name"], l["name"], l["config"]
            if cn == "InputLayer":
                t[name] = x
                continue
            src = [n[0] for n in l["inbound_nodes"][0]]
            ins = [t[n] for n in src]
            if self.emulate_bf16 and name in self.fp32_input_heads and src[0] in u:
                ins = [u[src[0]]]
            t[name] = self._layer(cn, name, c, ins)
            if self.emulate_bf16:
                if cn == "Conv2D":
                    u[name] = self._layer_fp32(cn, name, c, ins)
                elif cn == "Activation" and src[0] in u:
                    u[name] = _act(u[src[0]], c["activation"])
        outs = [t[n].permute(0, 2, 3, 1).contiguous().numpy() for n in self.output_names]
        if return_all:
            return outs, {k: v.permute(0, 2, 3, 1).contiguous().numpy() for k, v in t.items()}
        return outs

    def _r(self, t):
        return t.to(self.emulate_dtype).to(torch.float32) if self.emulate_bf16 else t

    def _layer(self, cn, name, c, ins):
        y = self._layer_fp32(cn, name, c, ins)
        if self.emulate_bf16:
            is_head = cn == "Conv2D" and name in self.output_names
            if self.round_layers is not None:
                if name in self.round_layers and not is_head:
                    y = self._r(y)
            elif cn in ("Conv2D", "Conv2DTranspose", "UpSampling2D") and not is_head:
                y = self._r(y)
        return y

    def _kernel(self, name, x_is_input):
        k = self.w[f"{name}/kernel"]
        if self.emulate_bf16 and not x_is_input and name not in self.output_names:
            k = self._r(k)
        return k

    def _layer_fp32(self, cn, name, c, ins):
        w = self.w
        if cn == "Conv2D":
            x = ins[0]
            k = self._kernel(name, x is self._x_in).permute(3, 2, 0, 1).contiguous()  # (Cout, Cin, kh, kw)
            b = w.get(f"{name}/bias") if c.get("use_bias", True) else None
            sh, sw = c["strides"]
            dh, dw = c.get("dilation_rate", [1, 1])
            if c["padding"] == "same":
                pt, pb = _same_pads(x.shape[2], k.shape[2], sh, dh)
                pl, pr = _same_pads(x.shape[3], k.shape[3], sw, dw)
                x = F.pad(x, (pl, pr, pt, pb))
            y = F.conv2d(x, k, b, stride=(sh, sw), dilation=(dh, dw))
            return _act(y, c.get("activation"))
        if cn == "Conv2DTranspose":
            x = ins[0]
            k = self._kernel(name, False).permute(3, 2, 0, 1).contiguous()  # (Cin, Cout, kh, kw)
            b = w.get(f"{name}/bias") if c.get("use_bias", True) else None
            sh, sw = c["strides"]
            y = F.conv_transpose2d(x, k, None, stride=(sh, sw))  # full: (n-1)*s + k
            if c["padding"] == "same":
                # TF deconv output length n*s; the forward conv it is the gradient of has
                # pad_before = max(k - s, 0) // 2, so the crop starts there.
                oh, ow = x.shape[2] * sh, x.shape[3] * sw
                ch = max(k.shape[2] - sh, 0) // 2
                cw = max(k.shape[3] - sw, 0) // 2
                y = y[:, :, ch : ch + oh, cw : cw + ow]
            if b is not None:
                y = y + b.view(1, -1, 1, 1)
            return _act(y, c.get("activation"))
        if cn == "Activation":
            return _act(ins[0], c["activation"])
        if cn == "ReLU":
            return torch.relu(ins[0])
        if cn == "MaxPooling2D":
            x = ins[0]
            kh, kw = c["pool_size"]
            sh, sw = c["strides"] or c["pool_size"]
            if c["padding"] == "same":
                pt, pb = _same_pads(x.shape[2], kh, sh)
                pl, pr = _same_pads(x.shape[3], kw, sw)
                x = F.pad(x, (pl, pr, pt, pb), value=float("-inf"))
            return F.max_pool2d(x, (kh, kw), (sh, sw))
        if cn == "UpSampling2D":
            x = ins[0]
            size = c["size"]
            if c.get("interpolation", "nearest") == "bilinear":
                return F.interpolate(x, scale_factor=tuple(float(s) for s in size), mode="bilinear",
                                     align_corners=False)
            return F.interpolate(x, scale_factor=tuple(float(s) for s in size), mode="nearest")
        if cn == "Concatenate":
            return torch.cat(ins, dim=1)
        if cn == "Add":
            y = ins[0]
            for z in ins[1:]:
                y = y + z
            return y
        if cn == "BatchNormalization":
            x = ins[0]
            eps = c.get("epsilon", 1e-3)
            g = w.get(f"{name}/gamma", torch.ones(x.shape[1]))
            be = w.get(f"{name}/beta", torch.zeros(x.shape[1]))
            mu = w[f"{name}/moving_mean"]
            var = w[f"{name}/moving_variance"]
            scale = g / torch.sqrt(var + eps)
            return x * scale.view(1, -1, 1, 1) + (be - mu * scale).view(1, -1, 1, 1)
        if cn == "Lambda":
            # the only Lambda layers the reference creates (resnet.py:326-362, 476-483), identified by layer name
            x = ins[0]
            if name == "tile_channels":
                return x.repeat(1, 3, 1, 1)
            if name == "imagenet_preproc_v1":
                mean = torch.tensor([103.939, 116.779, 123.68], dtype=torch.float32).view(1, 3, 1, 1)
                return (x * 255.0).flip(1) - mean
            raise NotImplementedError(f"Lambda layer {name}")
        if cn == "ZeroPadding2D":
            (pt, pb), (pl, pr) = c["padding"]
            return F.pad(ins[0], (pl, pr, pt, pb))
        raise NotImplementedError(f"Keras layer {cn} ({name})")


# ----------------------------------------------------------------------------------------------
# preprocessing (sleap/nn/inference.py:940-967 InferenceLayer.preprocess)
# ----------------------------------------------------------------------------------------------
def ensure_float(imgs):
    """normalization.py:34-49 -- uint8 -> float32 * (1/255); floats pass through."""
    imgs = np.asarray(imgs)
    if imgs.dtype == np.uint8:
        return imgs.astype(np.float32) * np.float32(1.0 / 255.0)
    return imgs.astype(np.float32)


def ensure_grayscale(imgs):
    """normalization.py:81-96 -- tf.image.rgb_to_grayscale weights [0.2989, 0.5870, 0.1140];
    for integer images the result is cast back (truncation) as convert_image_dtype does."""
    imgs = np.asarray(imgs)
    if imgs.shape[-1] == 1:
        return imgs
    w = np.array([0.2989, 0.5870, 0.1140], np.float32)
    if imgs.dtype == np.uint8:
        f = imgs.astype(np.float32) * np.float32(1.0 / 255.0)
        g = (f * w).sum(axis=-1, keepdims=True, dtype=np.float32)
        # convert_image_dtype float->uint8 (saturate=False): x * (255 + 0.5) truncated
        return (g * np.float32(255.5)).astype(np.uint8)
    return (imgs.astype(np.float32) * w).sum(axis=-1, keepdims=True, dtype=np.float32)


def ensure_rgb(imgs):
    """normalization.py:99-114 -- tile single channel to 3."""
    imgs = np.asarray(imgs)
    if imgs.shape[-1] == 1:
        return np.tile(imgs, (1,) * (imgs.ndim - 1) + (3,))
    return imgs


def resize_image(imgs, scale):
    """resizing.py:71-105 -- bilinear, half-pixel centres, no antialias; size = int(dim * scale);
    result cast back to the input dtype (tf.cast truncates for uint8)."""
    imgs = np.asarray(imgs)
    H, W = imgs.shape[1], imgs.shape[2]
    nh, nw = int(H * scale), int(W * scale)
    x = torch.from_numpy(imgs.astype(np.float32)).permute(0, 3, 1, 2)
    y = F.interpolate(x, size=(nh, nw), mode="bilinear", align_corners=False, antialias=False)
    y = y.permute(0, 2, 3, 1).numpy()
    if imgs.dtype == np.uint8:
        return y.astype(np.uint8)
    return y.astype(imgs.dtype)


def pad_to_stride(imgs, max_stride):
    """resizing.py:34-68 -- zero-pad bottom/right to a multiple of max_stride."""
    imgs = np.asarray(imgs)
    H, W = imgs.shape[1], imgs.shape[2]
    ph = (max_stride - H % max_stride) % max_stride
    pw = (max_stride - W % max_stride) % max_stride
    if ph or pw:
        imgs = np.pad(imgs, ((0, 0), (0, ph), (0, pw), (0, 0)))
    return imgs


def preprocess(imgs, input_scale=1.0, pad_stride=1, ensure_gray=False, ensure_color=False):
    """inference.py:940-967."""
    if ensure_gray:
        imgs = ensure_grayscale(imgs)
    elif ensure_color:
        imgs = ensure_rgb(imgs)
    imgs = ensure_float(imgs)
    if input_scale != 1.0:
        imgs = resize_image(imgs, input_scale)
    if pad_stride > 1:
        imgs = pad_to_stride(imgs, pad_stride)
    return imgs
