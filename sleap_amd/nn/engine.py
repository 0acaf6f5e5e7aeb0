"""Executes a SLEAP Keras functional graph (the content of `best_model.h5`) on MI355X.

Stands in for `tf.keras.models.load_model(...)` + `keras_model(imgs)` of the reference
(sleap/nn/inference.py:3207, 2864-2890). The graph is compiled once into a flat plan of fused HIP
launches (include/sleap_amd.h):

    Conv2D(k3)+bias+Activation(relu)                      -> sa_conv3x3_bf16 (MFMA implicit GEMM)
    MaxPooling2D(2) feeding a Conv2D                       -> folded into that conv's tile load
    Concatenate([skip, UpSampling2D(2,bilinear)(x)])+Conv  -> two-source K loop, upsample-on-load
    InputLayer(+ensure_float) + first Conv2D               -> sa_stem_conv3x3 (u8 in, bf16 out)
    1x1 linear head convs                                  -> sa_conv1x1_head (f32 out)
    Conv2DTranspose(k3,s2)+Activation                      -> sa_convt3x3s2_bf16
    Conv2D(relu)+BatchNormalization [+Add(skip, UpSampling2D(nearest)(x))]  (hourglass.py:17-45, 186-191)
                                                           -> sa_conv3x3_ex_bf16 (post-affine + residual epilogue)
    first Conv2D of any kernel size / stride (hourglass stem k7 s2) -> sa_image_conv_bf16

Activations are 16-bit (fp16 by default, bf16 optional) with channels padded to a multiple of 16, stored as 16-channel planes
[B, CP/16, H, W, 16] when every launch of the plan supports them (the UNet family) and NHWC otherwise (`DeviceNetwork.layout`);
accumulation is fp32. torch is used for device memory only.
"""
import ctypes as C
import json
import os
from typing import Dict, List, Optional

import numpy as np
import torch

from .. import _lib
from .._lib import check
from ..ops import _ptr, _stream, require_cuda


def _pad16(c):
    return (c + 15) // 16 * 16


class _T:
    """A (possibly virtual) activation tensor in the plan."""

    def __init__(self, kind, c, scale_num=1, scale_den=1, buf=None, parts=None, src=None, interp=None):
        self.kind = kind  # "input" | "real" | "pool" | "up" | "concat" | "f32out"
        self.c = c  # logical channels
        self.cp = _pad16(c)
        self.num, self.den = scale_num, scale_den  # spatial size = in_size * num / den
        self.buf = buf  # plan buffer id for real tensors
        self.parts = parts  # concat: list of _T
        self.src = src  # pool/up/zpad: source _T
        self.interp = interp
        self.tile = False  # input: tile_channels Lambda applied (resnet.py:326-339)
        self.preproc = False  # input: imagenet_preproc_v1 Lambda applied (resnet.py:342-362)
        self.pads = None  # zpad: ((top, bottom), (left, right))


class ConvOp(list):
    """One 3x3 (`kind` "conv") or 1x1 (`kind` "conv1x1") convolution launch of the plan. It is still a list -- the executor
    unpacks it positionally -- but the passes that inspect or rewrite plan entries use the slot names below.
      src0, src1   input tensors (src1: second Concatenate operand, None otherwise)
      mode         SRC1_* / SRC0_POOL2X flag of the 3x3 kernel; the STRIDE for a 1x1 conv
      w, bias      packed weights (device), bias (device, padded)
      out          output tensor; out_pool: its fused MaxPool2D(2) copy or None; need_full: `out` itself is stored
      heads        1x1 head ops computed in this conv's epilogue
      ext          None or {"ps", "pt", "res", "res_mode", "relu_last"}: post-affine (BatchNormalization), residual, last ReLU
    """
    FIELDS = ("kind", "src0", "src1", "mode", "w", "bias", "out", "relu", "out_pool", "need_full", "heads", "name", "ext")


def _slot(i):
    return property(lambda self: self[i], lambda self, v: self.__setitem__(i, v))


for _i, _f in enumerate(ConvOp.FIELDS):
    setattr(ConvOp, _f, _slot(_i))
# the same slot read as what it means for a "conv1x1" op: bits 0-7 the stride of a 1x1 conv, bits 8+ the window size k of a
# stride-1 "same" k x k conv run through the same tap GEMM (sa_convk_bf16: the 7x7 convs of the UNet stem blocks)
ConvOp.stride_word = ConvOp.mode
ConvOp.stride = property(lambda self: self[3] & 255)
ConvOp.ksize = property(lambda self: max(self[3] >> 8, 1))


DEFAULT_DTYPE = None  # set (temporarily) by load_model(dtype=...); otherwise _lib.DEFAULT_DTYPE (SLEAP_AMD_DTYPE or "fp16")


class DeviceNetwork:
    def __init__(self, model_config: dict, weights: Dict[str, np.ndarray], device=None, fuse_upsample: Optional[bool] = None,
                 fuse_heads: bool = True, fuse_stem: bool = True, use_stem16: bool = True, mfma_convt: bool = True,
                 mfma_stem: bool = True, fuse_pairs: bool = True, dtype: Optional[str] = None, layout: Optional[str] = None,
                 range_safe: Optional[bool] = None, range_log2_scale: Optional[Dict[str, int]] = None,
                 fuse_bneck: bool = True):
        """`fuse_upsample`: UpSampling2D(bilinear) folded into the consuming conv. On 16-channel planes the DMA kernel copies the
        half-resolution tile of source chunk c+1 and expands it in LDS into the idle stage while chunk c is multiplied (the
        upsampled tensor never exists in HBM); on NHWC tensors the register-staged first-generation kernel does it on load
        (slower than materialising the tensor for the DMA kernel). True = every such conv, False = none, None / "auto" = the
        environment's SA_FUSE_UPSAMPLE ("1" / "0") or, by default, PER LAYER: only where the expansion is cheaper than the
        upsampling launch it replaces. Measured (round 3, profiles/r03_ab_session.md): the expansion costs ~0.11 ms per layer
        of the benchmark decoder whatever its shape (its work grows with tiles x source chunks x output-channel tiles), the
        stand-alone upsampling 0.06 / 0.12 / 0.24 ms at 64^2 / 128^2 / 256^2 -- so it pays for convs with <= 64 output
        channels (the full-resolution end of the decoder) and only with fp16 storage on planes (packed-fp16 interpolation)."""
        require_cuda()
        # 16-bit storage type of activations and conv weights: "fp16" (default; 11-bit mantissa: heads within 0.1-0.5 % of an
        # fp32 network, finite range 65504) or "bf16" (fp32 range, 8-bit mantissa: 1-4 %). SLEAP_AMD_DTYPE sets the default.
        # Each type is its own build of the kernel library (csrc/bf16.h, _lib.lib(dtype)); accumulation is fp32 in both.
        self.dtype = dtype or DEFAULT_DTYPE or _lib.DEFAULT_DTYPE
        self._h = _lib.lib(self.dtype)
        self._tdtype = torch.float16 if self.dtype == "fp16" else torch.bfloat16
        self._range_checked = False
        # fp16 storage only: a network whose activations leave fp16's range on the first batch of an input shape is
        # re-compiled with per-tensor power-of-two scales folded into its weights (nn/range_scaling.py) instead of raising;
        # SA_RANGE_SAFE=0 / range_safe=False restore the FloatingPointError. `range_log2_scale`: None = unscaled.
        self.range_safe = (os.environ.get("SA_RANGE_SAFE", "1") != "0") if range_safe is None else bool(range_safe)
        self.range_log2_scale = None
        self._range_calibrated = False  # exponents came from calibrate_range() / range_log2_scale= (never changed implicitly)
        self._range_measured: Dict[str, float] = {}
        self._pending_scan = None   # (largest finite value, inf / NaN seen) of forwards under torch.distributed, until the ranks agree
        self._pending_imgs = None
        self._dist_agreed = False
        self._dist_defer = False    # defer_range_agreement(): the caller promised a dist_agree_range() on every rank
        self._preset_scales = dict(range_log2_scale) if range_log2_scale else None
        self.model_config = model_config
        # layout of the 16-bit activation tensors: None = 16-channel planes when every launch of the compiled plan supports
        # them (the UNet family), NHWC otherwise; "nhwc" / "planes16" force one (SA_LAYOUT in the environment likewise)
        self._layout_request = layout or os.environ.get("SA_LAYOUT") or None
        auto_up = fuse_upsample is None or fuse_upsample == "auto"
        env_up = os.environ.get("SA_FUSE_UPSAMPLE", "auto")
        # True / False = all / none; "auto" = per layer (self._fuse_up)
        self.fuse_upsample = ({"1": True, "0": False}.get(env_up, "auto")) if auto_up else bool(fuse_upsample)
        self.fuse_upsample_max_cout = int(os.environ.get("SA_FUSE_UPSAMPLE_MAX_COUT", "64"))
        self.fuse_heads = fuse_heads
        self.fuse_ext_heads = os.environ.get("SA_FUSE_EXT_HEADS", "1") != "0"  # (A/B: heads behind a Conv + BN + ReLU stay launches)
        self.fuse_stem = fuse_stem
        self.use_stem16 = use_stem16
        # 16->32->32 encoder block in one launch (csrc/convpair.hip); SA_FUSE_PAIRS=0 turns it off for A/B measurements
        self.fuse_pairs = fuse_pairs and os.environ.get("SA_FUSE_PAIRS", "1") != "0"
        # ResNet bottleneck tails in one launch (round 4, sa_conv3x3_bneck_bf16): 3x3 (64 maps) -> 1x1 expand + BN + shortcut Add +
        # ReLU -> the next block's 1x1 reduce; SA_FUSE_BNECK=0 turns it off for A/B measurements
        self.fuse_bneck = fuse_bneck and os.environ.get("SA_FUSE_BNECK", "1") != "0"
        self.mfma_stem = mfma_stem  # k7 first-layer convs of uint8 frames on the matrix cores (csrc/imgconv.hip)
        self.mfma_convt = mfma_convt  # Conv2DTranspose on the matrix cores (tap GEMM per output phase) vs the VALU kernel
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        cfg = model_config["config"]
        self.layers = cfg["layers"]
        self.input_name = cfg["input_layers"][0][0]
        self.output_names = [l[0] for l in cfg["output_layers"]]
        self.master_weights = weights  # the model's own float32 weights, never modified
        self.weights = weights         # what the plan is compiled from: master_weights, or them with range scales folded in
        self.in_channels = None
        for l in self.layers:
            if l["class_name"] == "InputLayer":
                self.in_channels = l["config"]["batch_input_shape"][-1]
        if self._preset_scales and self.dtype == "fp16" and any(self._preset_scales.values()):
            # persisted exponents of an earlier calibration (calibrate_range): the same network on every rank and in every
            # run, no first-batch measurement, no twin
            from . import range_scaling as RS

            self.weights = RS.fold_scales(model_config, weights, self._preset_scales)
            self.range_log2_scale = dict(self._preset_scales)
            self._range_calibrated = True
        self._compile()
        if auto_up and self.fuse_upsample and not self.planar and any(
                op[0] == "conv" and op.mode == _lib.SRC1_UPSAMPLE2X for op in self.plan):  # (only the plane kernels gain from it)
            self.fuse_upsample = False
            self._compile()
        self._buffers = {}

    # ------------------------------------------------------------------ compile
    def _fuse_up(self, cout: int) -> bool:
        """Does the conv with `cout` output channels read its bilinear-upsampled source at half resolution (see __init__)?"""
        if self.fuse_upsample == "auto":
            return self.dtype == "fp16" and self._layout_request != "nhwc" and _pad16(cout) <= self.fuse_upsample_max_cout
        return bool(self.fuse_upsample)

    def _consumers(self):
        cons = {}
        for l in self.layers:
            for node in l["inbound_nodes"][:1]:
                for inp in node:
                    cons.setdefault(inp[0], []).append(l["name"])
        return cons

    def _compile(self):
        dev = self.device
        cons = self._consumers()
        by_name = {l["name"]: l for l in self.layers}
        t: Dict[str, _T] = {}
        plan = []
        n_buf = [0]
        self.buf_meta = {}  # id -> (cp, num, den, dtype)
        skip = set()  # layers fused into a predecessor
        self.round_points = set()  # layer names whose output is the value stored (rounded) as bf16
        pooled_of: Dict[str, _T] = {}  # MaxPooling2D layer name -> pooled tensor written by the producing conv
        t_alias: List[str] = []  # layer names fused into the conv being compiled (all map to its output tensor)
        conv_of: Dict[int, list] = {}  # id(output _T) -> the plan op that writes it (for Add fusion)
        round_of: Dict[int, str] = {}  # id(output _T) -> its entry in round_points

        def new_buf(c_alloc, num, den, dtype):
            i = n_buf[0]
            n_buf[0] += 1
            self.buf_meta[i] = (c_alloc, num, den, dtype)
            return i

        def materialize(v: _T) -> _T:
            """Return a real bf16 tensor for v, emitting standalone kernels if needed."""
            if v.kind == "real":
                return v
            if v.kind == "pool":
                s = materialize(v.src)
                o = _T("real", v.c, v.num, v.den, buf=new_buf(v.cp, v.num, v.den, "bf16"))
                plan.append(("pool", s, o))
                return o
            if v.kind == "up":
                s = materialize(v.src)
                o = _T("real", v.c, v.num, v.den, buf=new_buf(v.cp, v.num, v.den, "bf16"))
                plan.append(("up", s, o, 1 if v.interp == "bilinear" else 0))
                return o
            raise NotImplementedError(f"cannot materialize tensor kind {v.kind}")

        def upload_f32(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)

        def padded_bias(name, cout, coutp, use_bias=True):
            b = np.zeros((coutp,), np.float32)
            if use_bias and f"{name}/bias" in self.weights:
                b[:cout] = self.weights[f"{name}/bias"]
            return upload_f32(b)

        def conv_activation(l):
            """Fuse a following Activation(relu) layer if it is the conv's only consumer."""
            name, act = l["name"], l["config"].get("activation", "linear")
            out_name = name
            c = cons.get(name, [])
            if act == "linear" and len(c) == 1 and by_name[c[0]]["class_name"] == "Activation" \
                    and name not in self.output_names:
                a = by_name[c[0]]["config"]["activation"]
                if a in ("relu", "linear"):
                    act = a
                    out_name = c[0]
                    skip.add(c[0])
            if act not in ("relu", "linear"):
                raise NotImplementedError(f"activation {act} on {name}")
            return out_name, 1 if act == "relu" else 0

        def sole_consumer(name, class_name):
            c = cons.get(name, [])
            if len(c) == 1 and by_name[c[0]]["class_name"] == class_name and name not in self.output_names:
                return by_name[c[0]]
            return None

        def bn_affine(l, coutp):
            """Inference-mode BatchNormalization as y = x * scale + shift per channel (padded channels -> 0)."""
            n, c = l["name"], l["config"]
            ax = c.get("axis", -1)
            ax = ax[0] if isinstance(ax, (list, tuple)) else ax
            if ax not in (-1, 3):
                raise NotImplementedError(f"BatchNormalization {n}: axis {ax}")
            var = np.asarray(self.weights[f"{n}/moving_variance"], np.float64)
            mean = np.asarray(self.weights[f"{n}/moving_mean"], np.float64)
            gamma = np.asarray(self.weights[f"{n}/gamma"], np.float64) if c.get("scale", True) else np.ones_like(var)
            beta = np.asarray(self.weights[f"{n}/beta"], np.float64) if c.get("center", True) else np.zeros_like(var)
            sc = gamma / np.sqrt(var + c.get("epsilon", 1e-3))
            scale, shift = np.zeros((coutp,), np.float32), np.zeros((coutp,), np.float32)
            scale[: sc.shape[0]] = sc
            shift[: sc.shape[0]] = beta - mean * sc
            return upload_f32(scale), upload_f32(shift)

        def conv_epilogue(l, coutp):
            """Conv2D[+Activation][+BatchNormalization[+Activation(relu)]] -> (out_name, relu, ext dict or None)."""
            out_name, relu = conv_activation(l)
            ext = None
            bn = sole_consumer(out_name, "BatchNormalization")
            if bn is not None:
                ps, pt = bn_affine(bn, coutp)
                ext = {"ps": ps, "pt": pt, "res": None, "res_mode": 0, "relu_last": 0}
                skip.add(bn["name"])
                t_alias.append(bn["name"])
                out_name = bn["name"]
                a = sole_consumer(out_name, "Activation")
                if a is not None and a["config"]["activation"] == "relu":
                    ext["relu_last"] = 1
                    skip.add(a["name"])
                    t_alias.append(a["name"])
                    out_name = a["name"]
            return out_name, relu, ext

        for l in self.layers:
            cn, name, c = l["class_name"], l["name"], l["config"]
            if name in skip:
                continue
            if cn == "InputLayer":
                t[name] = _T("input", c["batch_input_shape"][-1])
                continue
            ins = [t[n[0]] for n in l["inbound_nodes"][0]]
            if cn == "Conv2D":
                k = tuple(c["kernel_size"])
                strides = tuple(c["strides"])
                x = ins[0]
                img = x.src if x.kind == "zpad" else x  # ZeroPadding2D + valid conv on the image (resnet.py:109-114)
                on_image = img.kind == "input"
                if x.kind == "zpad" and not (on_image and c["padding"] == "valid"):
                    raise NotImplementedError(f"Conv2D {name}: ZeroPadding2D is only folded into a valid conv on the input image")
                if tuple(c.get("dilation_rate", (1, 1))) != (1, 1) and k != (1, 1):  # dilating a 1x1 kernel is a no-op
                    raise NotImplementedError(f"Conv2D {name}: dilation is not implemented")
                if c["padding"] != "same" and not (k == (1, 1) or x.kind == "zpad"):
                    raise NotImplementedError(f"Conv2D {name}: padding='valid' is only implemented for 1x1 kernels")
                if strides[0] != strides[1] or strides[0] not in (1, 2):
                    raise NotImplementedError(f"Conv2D {name}: stride {strides}")
                window = not on_image and k not in ((1, 1), (3, 3)) and k[0] == k[1] and k[0] <= 9 and strides == (1, 1) \
                    and c["padding"] == "same"  # k x k on a feature tensor (UNet stem blocks, unet.py:105-127): tap GEMM
                if not on_image and not (k == (3, 3) and strides == (1, 1)) and k != (1, 1) and not window:
                    raise NotImplementedError(f"Conv2D {name}: kernel {k} stride {strides} is only implemented on the input image")
                kern = np.asarray(self.weights[f"{name}/kernel"], np.float32)
                cin, cout = kern.shape[2], kern.shape[3]
                t_alias.clear()
                if k == (1, 1) and name in self.output_names and strides == (1, 1) and not on_image:
                    act = {"linear": 0, "sigmoid": 1}.get(c.get("activation", "linear"))
                    if act is None:
                        raise NotImplementedError(f"head {name}: activation {c.get('activation')}")
                    s = materialize(x)
                    w = np.zeros((cout, s.cp), np.float32)
                    w[:, :cin] = kern[0, 0].T
                    bias = np.zeros((cout,), np.float32)
                    if c.get("use_bias", True):
                        bias[:] = self.weights[f"{name}/bias"]
                    o = _T("f32out", cout, s.num, s.den, buf=new_buf(cout, s.num, s.den, "f32"))
                    plan.append(("head", s, o, upload_f32(w), upload_f32(bias), act))
                    t[name] = o
                    continue
                coutp = _pad16(cout)
                out_name, relu, ext = conv_epilogue(l, coutp)
                bias = padded_bias(name, cout, coutp, c.get("use_bias", True))
                # consumers: MaxPooling2D(2) consumers are served by a pooled copy written by this conv's epilogue
                consumers = [by_name[n] for n in cons.get(out_name, [])]
                pools = [q for q in consumers if q["class_name"] == "MaxPooling2D"
                         and tuple(q["config"]["pool_size"]) == (2, 2) and tuple(q["config"]["strides"]) == (2, 2)]
                need_full = (len(pools) != len(consumers)) or out_name in self.output_names or not consumers
                o = _T("real", cout, x.num, x.den)  # its buffer is allocated below, only if someone reads it
                o_pool = None
                if on_image and (k != (3, 3) or strides != (1, 1) or ext is not None or cin not in (1, 3)
                                 or x.kind == "zpad" or img.tile or img.preproc):
                    # general first-layer conv on the raw image (hourglass stem: k7 s2 + ReLU + BN; ResNet stem:
                    # [tile_channels, imagenet_preproc_v1,] ZeroPadding2D(3), k7 s2 valid, BN, ReLU)
                    st = strides[0]
                    o = _T("real", cout, x.num, x.den * st, buf=new_buf(coutp, x.num, x.den * st, "bf16"))
                    w = np.zeros((k[0], k[1], cin, coutp), np.float32)
                    w[..., :cout] = kern
                    b_np = np.zeros((coutp,), np.float32)
                    if c.get("use_bias", True):
                        b_np[:cout] = self.weights[f"{name}/bias"]
                    if 4 * w.size > 64 * 1024:
                        raise NotImplementedError(f"Conv2D {name}: first-layer weights exceed the 64 KiB LDS budget")
                    ps = pt = None
                    if ext is not None and not relu:
                        # BatchNormalization directly after the linear conv: fold into the fp32 weights and bias
                        sc, sh = ext["ps"].cpu().numpy(), ext["pt"].cpu().numpy()
                        w = w * sc
                        b_np = b_np * sc + sh
                        relu = ext["relu_last"]
                    elif ext is not None:
                        if ext["relu_last"]:
                            raise NotImplementedError(f"Conv2D {name}: ReLU-BN-ReLU after the first-layer conv")
                        ps, pt = ext["ps"], ext["pt"]
                    in_affine = None
                    if img.preproc:
                        # X*255, RGB->BGR, minus the caffe channel means: the flip becomes a permutation of the weight
                        # channels, scale and shift are applied per in-bounds tap inside the kernel
                        if cin != 3:
                            raise NotImplementedError("imagenet_preproc_v1 expects 3 channels")
                        w = np.ascontiguousarray(w[:, :, ::-1, :])
                        in_affine = upload_f32(np.array([255.0] * 3 + [-123.68, -116.779, -103.939], np.float32))
                    src_c = 1 if img.tile else cin
                    pads = (x.pads[0][0], x.pads[1][0]) if x.kind == "zpad" else None
                    mf = None
                    if k == (7, 7) and cin in (1, 3) and self.mfma_stem:
                        # uint8 frames: the same conv on the matrix cores (csrc/imgconv.hip); weights x input scale as
                        # hi + lo bf16 fragments, ImageNet means folded into the bias + an exact border indicator term
                        h = self._h
                        wk = np.ascontiguousarray(w[..., :cout], dtype=np.float32)
                        # a tiled grayscale frame (tile_channels) under a 3-channel kernel: one K slot per tap with the channel-
                        # summed weight (sa_imgconv_pack_tiled) -- the CinW = 1 kernel, 4 instead of 10 k-steps
                        tiled = bool(img.tile) and cin == 3
                        cin_k = 1 if tiled else cin
                        packed = np.zeros((h.sa_imgconv_packed_elems(7, cin_k, coutp),), np.uint16)
                        bias_io = np.ascontiguousarray(b_np, dtype=np.float32).copy()
                        scale = np.full((cin,), 1.0 if img.preproc else 1.0 / 255.0, np.float32)
                        mean = np.array([123.68, 116.779, 103.939], np.float32) if img.preproc else None
                        vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None  # noqa: E731
                        if tiled:
                            check(h.sa_imgconv_pack_tiled(vp(wk), 7, cout, coutp, vp(scale), vp(mean), vp(packed), vp(bias_io)),
                                  "sa_imgconv_pack_tiled")
                        else:
                            check(h.sa_imgconv_pack(vp(wk), 7, cin, cout, coutp, vp(scale), vp(mean), vp(packed), vp(bias_io)),
                                  "sa_imgconv_pack")
                        mf = {"w": torch.from_numpy(packed.view(np.int16)).to(dev), "bias": upload_f32(bias_io),
                              "has_mean": int(img.preproc), "cin_w": cin_k}
                    plan.append(["imgconv", o, upload_f32(np.ascontiguousarray(w)), upload_f32(b_np), src_c, relu, name, k, st,
                                 ps, pt, cin, in_affine, pads, mf])
                elif k == (1, 1) or window:
                    # Conv2D(k1, stride 1|2) on the matrix cores (ResNet bottleneck convs, skip projections); Conv2D(k x k,
                    # stride 1, same) through the same tap GEMM with k*k taps
                    s0 = materialize(x)
                    st = strides[0]
                    o = _T("real", cout, x.num, x.den * st, buf=new_buf(coutp, x.num, x.den * st, "bf16"))
                    h = self._h
                    nt = k[0] * k[1]
                    packed = np.zeros((h.sa_tapconv_packed_elems(nt, s0.cp, coutp),), np.uint16)
                    kc = np.ascontiguousarray(kern.reshape(nt, cin, cout))
                    check(h.sa_pack_tapconv_weights(kc.ctypes.data_as(C.c_void_p), nt, cin, s0.cp, cout, coutp,
                                                    packed.ctypes.data_as(C.c_void_p)), "sa_pack_tapconv_weights")
                    wdev = torch.from_numpy(packed.view(np.int16)).to(dev)
                    plan.append(ConvOp(["conv1x1", s0, None, st | ((k[0] << 8) if window else 0), wdev, bias, o, relu, None, True,
                                        [], name, ext]))
                    conv_of[id(o)] = plan[-1]
                elif x.kind == "input":
                    w = np.zeros((3, 3, cin, coutp), np.float32)
                    w[..., :cout] = kern
                    o.buf = new_buf(coutp, x.num, x.den, "bf16")
                    plan.append(("stem", o, upload_f32(w), bias, cin, relu, name))
                else:
                    mode, s0, s1 = _lib.SRC1_NONE, None, None
                    if x.kind == "concat":
                        if len(x.parts) != 2:
                            raise NotImplementedError("Concatenate with != 2 inputs")
                        s0 = materialize(x.parts[0])
                        p1 = x.parts[1]
                        if p1.kind == "up" and p1.interp == "bilinear" and self._fuse_up(cout):
                            s1, mode = materialize(p1.src), _lib.SRC1_UPSAMPLE2X
                        else:
                            s1, mode = materialize(p1), _lib.SRC1_DIRECT
                    elif x.kind == "pool" and ext is None:
                        s0, mode = materialize(x.src), _lib.SRC0_POOL2X
                    else:
                        s0 = materialize(x)
                    if pools and mode in (_lib.SRC1_NONE, _lib.SRC1_DIRECT):
                        o_pool = _T("real", cout, x.num, x.den * 2, buf=new_buf(coutp, x.num, x.den * 2, "bf16"))
                        for q in pools:
                            pooled_of[q["name"]] = o_pool
                    else:
                        need_full = True
                    if need_full:
                        o.buf = new_buf(coutp, x.num, x.den, "bf16")
                    c0, c1 = s0.c, (s1.c if s1 is not None else 0)
                    assert c0 + c1 == cin, (name, c0, c1, cin)
                    c0p, c1p = s0.cp, (s1.cp if s1 is not None else 0)
                    h = self._h
                    n = h.sa_conv3x3_packed_elems(c0p, c1p, coutp)
                    packed = np.zeros((n,), np.uint16)
                    kc = np.ascontiguousarray(kern)
                    check(h.sa_pack_conv3x3_weights(kc.ctypes.data_as(C.c_void_p), c0, c0p, c1, c1p, cout, coutp,
                                                    packed.ctypes.data_as(C.c_void_p)), "sa_pack_conv3x3_weights")
                    wdev = torch.from_numpy(packed.view(np.int16)).to(dev)
                    plan.append(ConvOp(["conv", s0, s1, mode, wdev, bias, o, relu, o_pool, need_full, [], name, ext]))
                    conv_of[id(o)] = plan[-1]
                t[out_name] = o
                t[name] = o
                for a in t_alias:
                    t[a] = o
                self.round_points.add(out_name)
                round_of[id(o)] = out_name
            elif cn == "Conv2DTranspose":
                ksz = tuple(c["kernel_size"])
                if ksz not in ((3, 3), (4, 4)) or tuple(c["strides"]) != (2, 2) or c["padding"] != "same":
                    raise NotImplementedError(f"Conv2DTranspose {name}: only k3/k4 s2 same is implemented")
                kern = np.asarray(self.weights[f"{name}/kernel"], np.float32)  # (kh, kw, Cout, Cin)
                cout, cin = kern.shape[2], kern.shape[3]
                s = materialize(ins[0])
                coutp = _pad16(cout)
                t_alias.clear()
                out_name, relu, ext = conv_epilogue(l, coutp)
                bias = padded_bias(name, cout, coutp, c.get("use_bias", True))
                o = _T("real", cout, s.num * 2, s.den, buf=new_buf(coutp, s.num * 2, s.den, "bf16"))
                if ksz == (3, 3) and ext is None and not self.mfma_convt:
                    w = np.zeros((3, 3, coutp, s.cp), np.float32)
                    w[:, :, :cout, :cin] = kern
                    wb = torch.from_numpy(w).to(dev).to(self._tdtype).contiguous()
                    plan.append(("convt", s, wb, bias, o, relu))
                else:
                    # one tap-GEMM launch per output phase; phase weights = the kernel taps of that phase, transposed
                    h = self._h
                    phases = []
                    for ph in range(4):
                        ky, kx = (C.c_int * 4)(), (C.c_int * 4)()
                        n = h.sa_convt_s2_phase_taps(ksz[0], ph, ky, kx)
                        taps = np.ascontiguousarray(np.stack([kern[ky[i], kx[i]].T for i in range(n)]))  # (n, Cin, Cout)
                        packed = np.zeros((h.sa_tapconv_packed_elems(n, s.cp, coutp),), np.uint16)
                        check(h.sa_pack_tapconv_weights(taps.ctypes.data_as(C.c_void_p), n, cin, s.cp, cout, coutp,
                                                        packed.ctypes.data_as(C.c_void_p)), "sa_pack_tapconv_weights")
                        phases.append(torch.from_numpy(packed.view(np.int16)).to(dev))
                    plan.append(("convt2", s, phases, bias, o, relu, ksz[0], ext, cin))
                t[out_name] = o
                t[name] = o
                for a in t_alias:
                    t[a] = o
                self.round_points.add(out_name)
            elif cn == "Activation":
                raise NotImplementedError(f"standalone Activation {name} (not fused into a conv)")
            elif cn == "Lambda":
                x = ins[0]
                if x.kind != "input" or name not in ("tile_channels", "imagenet_preproc_v1"):
                    raise NotImplementedError(f"Lambda layer {name}")
                y = _T("input", 3 if name == "tile_channels" else x.c)
                y.tile, y.preproc = x.tile or name == "tile_channels", x.preproc or name == "imagenet_preproc_v1"
                t[name] = y
            elif cn == "ZeroPadding2D":
                x = ins[0]
                pd = c["padding"]
                y = _T("zpad", x.c, x.num, x.den, src=x)
                y.pads = ((pd[0][0], pd[0][1]), (pd[1][0], pd[1][1]))
                t[name] = y
            elif cn == "MaxPooling2D":
                x = ins[0]
                ksz, pst = tuple(c["pool_size"]), tuple(c["strides"] or c["pool_size"])
                if x.kind == "zpad" or ksz != (2, 2) or pst != (2, 2):
                    # general window (ResNet: ZeroPadding2D(1) + MaxPooling2D(3, s2, valid)); the output must be exactly
                    # in/stride so that the engine's stride bookkeeping (sizes = input * num / den) holds
                    if ksz[0] != ksz[1] or pst[0] != pst[1]:
                        raise NotImplementedError(f"MaxPooling2D {name}: anisotropic window")
                    kk, st = ksz[0], pst[0]
                    if x.kind == "zpad":
                        if c["padding"] != "valid":
                            raise NotImplementedError(f"MaxPooling2D {name}: ZeroPadding2D + 'same' pooling")
                        (ptop, pbot), (pleft, pright) = x.pads
                        if ptop != pleft or pbot != pright or not (kk - st <= ptop + pbot <= kk - 1) or st not in (1, 2):
                            raise NotImplementedError(f"MaxPooling2D {name}: padding {x.pads} for window {kk} stride {st}")
                        srcT, pad, pad_zero = x.src, ptop, 1
                    else:
                        if c["padding"] != "same" or st not in (1, 2):
                            raise NotImplementedError(f"MaxPooling2D {name}: window {kk} stride {st} padding {c['padding']}")
                        srcT, pad, pad_zero = x, None, 0  # TF SAME: pad_before = pad_total // 2, resolved at run time
                    s0 = materialize(srcT)
                    o = _T("real", s0.c, s0.num, s0.den * st, buf=new_buf(s0.cp, s0.num, s0.den * st, "bf16"))
                    plan.append(("poolg", s0, o, kk, st, pad, pad_zero))
                    t[name] = o
                    continue
                t[name] = pooled_of[name] if name in pooled_of else _T("pool", x.c, x.num, x.den * 2, src=x)
            elif cn == "UpSampling2D":
                if tuple(c["size"]) != (2, 2):
                    raise NotImplementedError(f"UpSampling2D {name}: only x2 is implemented")
                x = ins[0]
                t[name] = _T("up", x.c, x.num * 2, x.den, src=x, interp=c.get("interpolation", "nearest"))
                if c.get("interpolation", "nearest") == "bilinear":
                    self.round_points.add(name)
            elif cn == "Concatenate":
                t[name] = _T("concat", sum(i.c for i in ins), ins[0].num, ins[0].den, parts=ins)
            elif cn == "Add":
                if len(ins) != 2 or ins[0].c != ins[1].c:
                    raise NotImplementedError(f"Add {name}: needs two inputs with equal channels")
                in_names = [n[0] for n in l["inbound_nodes"][0]]
                # fold the addition into the epilogue of the conv that produced the later operand when this Add is
                # that tensor's only reader; the other operand (possibly UpSampling2D(nearest) of a half-resolution
                # tensor, read with (y>>1, x>>1)) becomes the residual
                fused = False
                for i in (1, 0):
                    a, b = ins[i], ins[1 - i]
                    op = conv_of.get(id(a)) if a.kind == "real" else None
                    if op is None or cons.get(in_names[i], []) != [name] or op.heads or op.out_pool is not None:
                        continue
                    if op.kind == "conv" and op.mode not in (_lib.SRC1_NONE, _lib.SRC1_DIRECT):
                        continue
                    if op.kind == "conv1x1" and b.kind == "up" and b.interp == "nearest":
                        continue  # the GEMM epilogue only reads a same-resolution residual
                    if op.ext is not None and (op.ext["relu_last"] or op.ext["res"] is not None):
                        continue
                    # the residual must exist before the conv runs: kernels emitted to materialise it are moved in
                    # front of the conv; an operand written by a later op can not be folded
                    idx, n0 = next(i for i, q in enumerate(plan) if q is op), len(plan)
                    if b.kind == "up" and b.interp == "nearest":
                        res, res_mode = materialize(b.src), 1
                    else:
                        res, res_mode = materialize(b), 0
                    moved = plan[n0:]
                    del plan[n0:]
                    plan[idx:idx] = moved
                    idx += len(moved)
                    if any(self._writes(q, res) for q in plan[idx:]):
                        continue
                    if op.ext is None:
                        one = np.ones((a.cp,), np.float32)
                        op.ext = {"ps": upload_f32(one), "pt": upload_f32(0 * one), "res": None, "res_mode": 0, "relu_last": 0}
                    op.ext["res"], op.ext["res_mode"] = res, res_mode
                    self.round_points.discard(round_of[id(a)])
                    act = sole_consumer(name, "Activation")
                    if act is not None and act["config"]["activation"] == "relu":
                        op.ext["relu_last"] = 1
                        skip.add(act["name"])
                        t[act["name"]] = a
                    round_of[id(a)] = act["name"] if op.ext["relu_last"] else name
                    self.round_points.add(round_of[id(a)])
                    t[name] = a
                    fused = True
                    break
                if not fused:
                    a, b, half = ins[0], ins[1], 0
                    if a.kind == "up" and a.interp == "nearest" and b.kind != "up":
                        a, b = b, a
                    a = materialize(a)
                    if b.kind == "up" and b.interp == "nearest":
                        b, half = materialize(b.src), 1
                    else:
                        b = materialize(b)
                    relu = 0
                    act = sole_consumer(name, "Activation")
                    o = _T("real", a.c, a.num, a.den, buf=new_buf(a.cp, a.num, a.den, "bf16"))
                    if act is not None and act["config"]["activation"] == "relu":
                        relu = 1
                        skip.add(act["name"])
                        t[act["name"]] = o
                    plan.append(("add", a, b, half, relu, o))
                    t[name] = o
                    self.round_points.add(act["name"] if relu else name)
            elif cn == "BatchNormalization":
                raise NotImplementedError(f"BatchNormalization {name} does not directly follow a Conv2D")
            else:
                raise NotImplementedError(f"Keras layer {cn} ({name}) is not implemented in the HIP engine")
        plan = self._fuse_heads(plan) if self.fuse_heads else plan
        plan = self._fuse_bottlenecks(plan, [t[n] for n in self.output_names]) if self.fuse_bneck else plan
        self.plan = self._fuse_stem(plan) if self.fuse_stem else plan
        self.outputs = []
        for n in self.output_names:
            o = t[n]
            if o.kind == "real" and o.buf is not None:
                pass  # backbone feature output (bf16 in HBM); forward() hands back an fp32 copy
            elif o.kind != "f32out":
                raise NotImplementedError(f"model output {n} is neither a 1x1 head nor a stored feature tensor")
            self.outputs.append(o)
        if self.fuse_pairs:
            self.plan = self._fuse_pairs(self.plan)
        self._tensor_of = t
        self.n_buf = n_buf[0]
        self.layout = self._pick_layout()
        # reduce fractions for stride bookkeeping
        self.max_stride = max(den // max(num, 1) for (_, num, den, _) in self.buf_meta.values())

    def _pick_layout(self) -> int:
        """SA_LAYOUT_PLANES16 when every launch of the plan can read and write 16-channel planes (include/sleap_amd.h): the
        fused stem, the fused encoder block, 3x3 convs on the DMA path (plain / concatenated sources, fused heads, pooled
        copies, BatchNormalization / residual epilogues), the materialised upsampling, first-layer convs on the image and
        un-fused 1x1 heads of <= 64 channels (the last three since round 3: the whole hourglass family); 16-bit model outputs
        only with 16 padded channels (the same bytes in both layouts). Round 4: the tap GEMM (1x1 / k x k / transposed convs) has a
        plane variant and stand-alone pools / adds run per plane, so the ResNet family is on planes too. What is left for NHWC:
        the first-generation conv kernels (pooled / upsampled source modes folded into the tile load), the fixture-only
        VALU transposed conv, the VALU first-layer conv -- one layout per plan, no conversion launches."""
        req = self._layout_request
        if req not in (None, "nhwc", "planes16"):
            raise ValueError(f"layout must be 'nhwc' or 'planes16', got {req!r}")

        def fits(op):
            k = op[0]
            if k in ("stem2", "pair", "up", "bneck"):
                return True
            if k == "imgconv":  # first-layer conv on the raw image (hourglass / ResNet stems): writes planes (round 3)
                return op[1].cp % 16 == 0
            if k == "head":  # un-fused 1x1 head on the matrix cores: reads planes (round 3); <= 64 output channels
                s_, o_ = op[1], op[2]
                return o_.c <= 64 and s_.cp % 16 == 0 and s_.cp // 16 * 2048 * (1 if o_.c <= 32 else 2) <= 64 * 1024
            if k in ("conv1x1", "convt2"):  # tap GEMM on planes (round 4: tapconv_kernel<.., PL>): 1x1 (stride 1 / 2), k x k, transposed
                return all(v is None or v.cp % 16 == 0 for v in (op[1], op.out if k == "conv1x1" else op[4]))
            if k in ("poolg", "pool", "add"):  # per-plane launches: a plane of a frame is a frame of 16 channels
                return True
            if k != "conv":
                return False
            if op.ext is not None:  # BatchNormalization / residual epilogue: plane-capable since round 3 (plain sources only)
                return op.mode in (_lib.SRC1_NONE, _lib.SRC1_DIRECT)
            return op.mode in (_lib.SRC1_NONE, _lib.SRC1_DIRECT) or (op.mode == _lib.SRC1_UPSAMPLE2X and not op.heads)

        ok = all(fits(op) for op in self.plan) and all(o.kind == "f32out" or o.cp == 16 for o in self.outputs)
        if req == "planes16" and not ok:
            raise NotImplementedError("layout='planes16': this plan holds launches that only exist for NHWC tensors")
        return _lib.LAYOUT_PLANES16 if (ok and req != "nhwc") else _lib.LAYOUT_NHWC

    @property
    def planar(self) -> bool:
        return self.layout == _lib.LAYOUT_PLANES16

    def stored_tensor(self, buf_id: int, shape_key=None) -> torch.Tensor:
        """The plan tensor `buf_id` of the resident workspace as a [B,H,W,CP] tensor (a COPY when the plan is in planes:
        tests / diagnostics; the kernels read the workspace itself)."""
        bufs = self._buffers[shape_key] if shape_key is not None else next(iter(self._buffers.values()))
        t = bufs[buf_id]
        if not self.planar or t.dtype == torch.float32 or t.shape[3] == 16:
            return t
        b, hh, ww, cp = t.shape
        return t.reshape(b, cp // 16, hh, ww, 16).permute(0, 2, 3, 1, 4).reshape(b, hh, ww, cp)

    @staticmethod
    def _writes(op, tensor):
        k = op[0]
        outs = {"conv": (6, 8), "conv1x1": (6,), "stem": (1,), "imgconv": (1,), "pool": (2,), "poolg": (2,), "up": (2,),
                "convt": (4,), "convt2": (4,), "add": (5,), "head": (2,)}.get(k)
        if k in ("stem2", "pair"):
            return DeviceNetwork._writes(op[2], tensor)
        if k == "bneck":  # ["bneck", C, X, Y | None, xw, yw]: X's (and Y's) outputs; C's activation stays on chip
            return op[2].out is tensor or (op[3] is not None and op[3].out is tensor)
        return any(op[i] is tensor for i in outs)

    @staticmethod
    def _reads(op):
        k = op[0]
        if k == "pair":
            return [op[1][1]]
        if k == "bneck":
            r = [op[1].src0]
            if op[2].ext is not None and op[2].ext["res"] is not None:
                r.append(op[2].ext["res"])
            return r
        if k in ("conv", "conv1x1"):
            r = [t for t in (op.src0, op.src1) if t is not None]
            if op.ext is not None and op.ext["res"] is not None:
                r.append(op.ext["res"])
            return r
        if k in ("head", "pool", "poolg", "up", "convt", "convt2"):
            return [op[1]]
        if k == "add":
            return [op[1], op[2]]
        return []

    def op_bytes(self, H, W):
        """Algorithmic HBM bytes of every launch of the plan for ONE frame, in `op_descriptions` order: each input tensor read
        once + each output tensor written once (16-bit activations with padded channels, float32 model outputs, the uint8
        image; weights not counted: they stay in L2). What a launch's GB/s is quoted on in tools/net_profile.py."""
        def tb(t):
            if t is None or getattr(t, "buf", None) is None and t.kind != "input":
                return 0
            if t.kind == "input":
                return H * W * t.c
            h, w = H * t.num // t.den, W * t.num // t.den
            return h * w * (t.c * 4 if t.kind == "f32out" else t.cp * 2)

        out = []
        for op in self.plan:
            k = op[0]
            rd = list(self._reads(op))
            if k == "bneck":
                n = sum(tb(t) for t in rd) + tb(op[2].out) + (tb(op[3].out) if op[3] is not None else 0)
            elif k in ("stem2", "pair"):
                wr = ([op[2].out] if op[2].need_full else []) + ([op[2].out_pool] if op[2].out_pool is not None else [])
                n = sum(tb(t) for t in rd) + sum(tb(t) for t in wr) + (H * W * self.in_channels if k == "stem2" else 0)
            else:
                idx = {"conv": (6, 8), "conv1x1": (6,), "stem": (1,), "imgconv": (1,), "pool": (2,), "poolg": (2,), "up": (2,),
                       "convt": (4,), "convt2": (4,), "add": (5,), "head": (2,)}[k]
                wr = [op[i] for i in idx if op[i] is not None]
                if k == "conv":
                    if not op.need_full:
                        wr = [t for t in wr if t is not op.out]
                    wr += [hd[2] for hd in op.heads]
                n = sum(tb(t) for t in rd) + sum(tb(t) for t in wr) + (H * W * self.in_channels if k in ("stem", "imgconv") else 0)
            out.append(int(n))
        return out

    def _fuse_bottlenecks(self, plan, out_tensors):
        """ResNet bottleneck tails (resnet.py:168-253) as ONE launch each (`sa_conv3x3_bneck_bf16`):

            C: Conv2D(k3, 64 maps) + BN + ReLU       -- its activation never reaches HBM (268 MB per launch at 256 x 256, 16 frames)
            X: Conv2D(k1, 64 -> 4 x 64) + BN + Add(shortcut) + ReLU   -- stored: it is the next block's shortcut
            Y: the NEXT block's Conv2D(k1, 4 x 64 -> 64) + BN + ReLU  -- computed from X's stored (rounded) values while they are
                                                                         still in registers: X's output is not read back

        Conditions: C is a plain 3 x 3 conv with the extended epilogue and exactly 64 padded output channels whose only reader is
        X; X and Y are stride-1 1 x 1 convs, X with <= 256 padded output channels, Y with 64 and no residual. The fused op takes
        X's place in the plan (its shortcut operand is ready there); C and Y disappear."""
        readers: Dict[int, list] = {}
        for op in plan:
            for tns in self._reads(op):
                readers.setdefault(id(tns), []).append(op)
        drop, repl = set(), {}
        for c in plan:
            if not (c[0] == "conv" and c.src1 is None and c.mode == _lib.SRC1_NONE and c.ext is not None and c.ext["res"] is None
                    and not c.heads and c.out_pool is None and c.out.cp == 64 and id(c) not in drop):
                continue
            rs = readers.get(id(c.out), [])
            if len(rs) != 1:
                continue
            x = rs[0]
            if not (x[0] == "conv1x1" and x.stride == 1 and x.ksize == 1 and x.src0 is c.out and x.out.cp % 32 == 0
                    and 32 <= x.out.cp <= 256 and id(x) not in repl):
                continue
            if x.ext is not None and x.ext["res"] is not None and x.ext["res"].cp != x.out.cp:
                continue
            y = None
            for q in readers.get(id(x.out), []):
                if (q[0] == "conv1x1" and q.stride == 1 and q.ksize == 1 and q.src0 is x.out and q.out.cp == 64
                        and (q.ext is None or q.ext["res"] is None) and id(q) not in drop and id(q) not in repl):
                    y = q
                    break
            # a model output must stay a stored tensor: C's activation is not stored any more
            if any(o is c.out or (y is not None and o is x.out and False) for o in out_tensors):
                continue
            drop.add(id(c))
            # C's activation never reaches HBM: no workspace tensor (it was still allocated, zeroed and range-scanned -- ~134 MB
            # per fused block at 1024 x 1024, 16 frames). The plan encoder gives it a shape-only entry (`bid`).
            self.buf_meta.pop(c.out.buf, None)
            c.out.buf = None
            if y is not None:
                drop.add(id(y))
            repl[id(x)] = ["bneck", c, x, y, self._pointwise_from_tap(x.w, x.src0.cp, x.out.cp),
                           self._pointwise_from_tap(y.w, y.src0.cp, y.out.cp) if y is not None else None]
        out = []
        for op in plan:
            if id(op) in drop:
                continue
            out.append(repl.get(id(op), op))
        # Y may sit BEFORE X's position only if it read something else; it reads X.out, so it follows X: order is preserved
        return out

    @staticmethod
    def _pointwise_from_tap(w_tap: torch.Tensor, cinp: int, coutp: int) -> torch.Tensor:
        """The tap GEMM's packed 1x1 weights [co32][1 tap][k16][64 lanes][8] (element (half, j) of a lane = input channel
        16 k + 8 half + j) re-ordered for the fused stages, whose B operands come straight from accumulator registers: element
        (half', j') = input channel 16 k + 8 (j' >> 2) + 4 half' + (j' & 3) (sa_pack_pointwise_weights' layout). A permutation of
        stored values on the device -- no arithmetic."""
        k16, co32 = cinp // 16, (coutp + 31) // 32
        w = w_tap.reshape(co32, k16, 2, 32, 8)  # [.., half, n, j]
        hp = torch.arange(2).view(2, 1)  # half'
        jp = torch.arange(8).view(1, 8)  # j'
        src_half = (jp >> 2).expand(2, 8)            # half  = j' >> 2
        src_j = (4 * hp + (jp & 3)).expand(2, 8)     # j     = 4 half' + (j' & 3)
        dev = w.device
        o = w[:, :, src_half.to(dev), :, src_j.to(dev)]        # -> [2, 8, co32, k16, 32] (advanced indices first)
        return o.permute(2, 3, 0, 4, 1).contiguous().reshape(-1)  # [co32, k16, half', n, j']

    @staticmethod
    def _bneck_words(op, bid, dp):
        """[src, w, bias, relu, ps, pt, relu_last | xw, xbias, xps, xpt, xres, xrelu, xrelu_last, x_out | has_y, yw, ybias, yps, ypt,
        yrelu, yrelu_last, y_out] (K_BNECK of csrc/network.hip; the arguments of sa_conv3x3_bneck_bf16)"""
        _, c, x, y, xw, yw = op
        ce, xe = c.ext, x.ext
        a = [bid(c.src0), dp(c.w), dp(c.bias), c.relu, dp(ce["ps"]), dp(ce["pt"]), ce["relu_last"],
             dp(xw), dp(x.bias), x.relu, dp(xe["ps"]) if xe else 0, dp(xe["pt"]) if xe else 0, bid(xe["res"]) if xe else -1,
             xe["relu_last"] if xe else 0, bid(x.out)]
        if y is not None:
            ye = y.ext
            a += [1, dp(yw), dp(y.bias), y.relu, dp(ye["ps"]) if ye else 0, dp(ye["pt"]) if ye else 0, ye["relu_last"] if ye else 0,
                  bid(y.out)]
        else:
            a += [0, 0, 0, 0, 0, 0, 0, -1]
        return a

    def _fuse_heads(self, plan):
        """Move 1x1 heads into the epilogue of the conv that produces their input (<= 2 heads, <= 64 channels each,
        producer with <= 128 padded output channels on the DMA path). If the heads were the only readers the bf16
        feature tensor is never written."""
        convs = {id(op.out): op for op in plan if op[0] == "conv"}
        out = []
        for op in plan:
            if op[0] == "head":  # ("head", src, out, w, bias, activation)
                prod = convs.get(id(op[1]))
                # (a producer with the extended epilogue -- Conv + BN + ReLU, the ResNet decoder -- takes heads when it has no
                #  residual / pooled output and 33..64 padded channels: sa_conv3x3_ex_heads_bf16)
                ext_ok = prod is not None and (prod.ext is None or (
                    prod.ext["res"] is None and prod.out_pool is None and prod.out.cp in (48, 64) and self.fuse_ext_heads))
                if (prod is not None and prod.mode in (_lib.SRC1_NONE, _lib.SRC1_DIRECT) and prod.out.cp <= 128
                        and len(prod.heads) < 2 and op[2].c <= 64 and ext_ok):
                    prod.heads.append(op)
                    continue
            out.append(op)
        for op in out:
            if op[0] == "conv" and op.heads:
                o = op.out
                readers = sum(1 for q in out for t in self._reads(q) if t is o)
                if readers == 0:
                    op.need_full = False
                    self.buf_meta.pop(o.buf, None)
                    o.buf = None
        return out

    def _n_cu(self) -> int:
        if getattr(self, "_n_cu_cached", None) is None:
            from .. import ops as _ops

            self._n_cu_cached = int(_ops.device_info(self.device.index or 0)["n_cu"])
        return self._n_cu_cached

    def _fuse_pairs(self, plan):
        """conv(16 -> 32) whose only reader is a conv(32 -> 32), or (round 6) conv(32 -> 64) whose only reader is a conv(64 -> 64)
        -> one launch (sa_conv3x3_pair_bf16); the intermediate tensor is never allocated. Plain convs only (no concat /
        pooled-source / heads / extended epilogue on either). SA_FUSE_PAIRS64=0 keeps the second form as two launches (A/B)."""
        out = list(plan)
        shapes = {(16, 32, 32)}
        if os.environ.get("SA_FUSE_PAIRS64", "1") != "0":
            shapes.add((32, 64, 64))

        def plain(op):
            return (op[0] == "conv" and op.src1 is None and op.mode == _lib.SRC1_NONE and not op.heads and op.ext is None)

        for x in list(out):
            if not plain(x) or x.out_pool is not None or x.out.buf is None:
                continue
            readers = [q for q in out if any(t is x.out for t in self._reads(q))]
            if len(readers) != 1 or not plain(readers[0]):
                continue
            y = readers[0]
            if y.src0 is not x.out or (x.src0.cp, x.out.cp, y.out.cp) not in shapes or any(x.out is o for o in self.outputs):
                continue
            i = next(k for k, q in enumerate(out) if q is y)
            out[i] = ["pair", x, y]
            del out[next(k for k, q in enumerate(out) if q is x)]
            if x.src0.cp == 32:
                # the 32 -> 64 -> 64 block keeps its intermediate BUFFER (never touched at the sizes the fused launch is for): one
                # persistent workgroup per CU needs >= ~6 tiles each to beat two launches (measured: 8 frames of 1024^2 = 4 tiles
                # per CU 0.790 vs 0.780 ms per step, 16 frames 1.37 vs 1.40, 32 frames 2.66 vs 2.70), so small launches run as
                # the two convolutions through it (decided per launch by the executor: csrc/network.hip K_PAIR, _launch_pair)
                continue
            self.buf_meta.pop(x.out.buf, None)
            x.out.buf = None
        return out

    def _fuse_stem(self, plan):
        """stem conv + the conv that follows it -> one launch (sa_stem_conv3x3x2_bf16) when the stem output has no
        other reader, 16/32 padded channels, and the second conv has <= 64 output channels and no fused heads."""
        out = list(plan)
        for i, op in enumerate(out):
            if op[0] != "stem":
                continue
            so = op[1]
            readers = [q for q in out if any(t is so for t in self._reads(q))]
            if len(readers) != 1 or readers[0][0] != "conv":
                continue
            cv = readers[0]
            if cv.src0 is not so or cv.src1 is not None or cv.mode != _lib.SRC1_NONE or cv.heads or so.cp not in (16, 32) \
                    or cv.out.cp > 64 or cv.ext is not None:
                continue
            w1_16 = None
            stem_name = op[6]  # ("stem", out, w, bias, cin, relu, layer name)
            if so.cp == 16 and cv.out.cp == 16:  # register-resident 16x16x32 MFMA specialisation (uint8 input)
                h = self._h
                k0 = np.ascontiguousarray(self.weights[stem_name + "/kernel"], dtype=np.float32)
                k1 = np.ascontiguousarray(self.weights[cv.name + "/kernel"], dtype=np.float32)
                b0 = np.ascontiguousarray(self.weights.get(stem_name + "/bias", np.zeros(k0.shape[3])), dtype=np.float32)
                b1 = np.ascontiguousarray(self.weights.get(cv.name + "/bias", np.zeros(k1.shape[3])), dtype=np.float32)
                blob = np.zeros((h.sa_stem16_blob_bytes(),), np.uint8)
                vp = lambda a: a.ctypes.data_as(C.c_void_p)
                check(h.sa_stem16_pack(vp(k0), vp(b0), k0.shape[2], k0.shape[3], vp(k1), vp(b1), k1.shape[3], vp(blob)),
                      "sa_stem16_pack")
                w1_16 = torch.from_numpy(blob).to(self.device)
            out[next(i for i, q in enumerate(out) if q is cv)] = ["stem2", op, cv, w1_16]
            del out[next(i for i, q in enumerate(out) if q is op)]
            self.buf_meta.pop(so.buf, None)
            so.buf = None
        return out

    def output_strides(self):
        return [o.den // o.num for o in self.outputs]

    # ------------------------------------------------------------------ C executor (include/sleap_amd.h: sa_network_*)
    PLAN_MAGIC = 0x53414E4554303032  # "SANET002"
    _K = {"bneck": 14, "stem2": 1, "stem": 2, "conv": 3, "pair": 4, "conv1x1": 5, "convt2": 6, "convt": 7, "poolg": 8, "imgconv": 9,
          "add": 10, "head": 11, "pool": 12, "up": 13}

    def plan_words(self) -> np.ndarray:
        """The compiled plan as the int64 word stream `sa_network_create` reads (csrc/network.hip): header, buffer table
        (cp, num, den, kind: 0 = 16-bit, 1 = f32, 2 = virtual / shape only), output table, one record per launch. Weight operands
        are the device addresses of tensors this object keeps alive -- the words are valid inside this process only."""
        bufs = {i: [c, num, den, 1 if dt == "f32" else 0] for i, (c, num, den, dt) in self.buf_meta.items()}
        n_ids = [max(list(bufs) + [-1]) + 1]
        virtual = {}

        def bid(t):
            """buffer id of a plan tensor; tensors the fusion passes keep on chip get a virtual (shape-only) entry"""
            if t is None:
                return -1
            if t.buf is not None:
                return t.buf
            if id(t) not in virtual:
                virtual[id(t)] = n_ids[0]
                bufs[n_ids[0]] = [t.cp, t.num, t.den, 2]
                n_ids[0] += 1
            return virtual[id(t)]

        def dp(x):
            return 0 if x is None else int(x.data_ptr())

        ops = []
        for op in self.plan:
            k = op[0]
            if k == "stem2":
                _, (_k, so, w0, b0, cin, relu0, _n0), cv, w1_16 = op
                a = [cin, dp(w0), dp(b0), so.cp, relu0, dp(cv.w), dp(cv.bias), cv.out.cp, cv.relu,
                     bid(cv.out) if cv.need_full else -1, bid(cv.out_pool), dp(w1_16) if (w1_16 is not None and self.use_stem16) else 0]
            elif k == "stem":
                _, o, w, bias, cin, relu, _n0 = op
                a = [bid(o), dp(w), dp(bias), cin, relu]
            elif k == "conv":
                a = [bid(op.src0), bid(op.src1), op.mode, dp(op.w), dp(op.bias), bid(op.out), op.relu, bid(op.out_pool),
                     1 if op.need_full else 0, len(op.heads)]
                for hd in op.heads:  # ("head", src, out, w, bias, activation)
                    assert hd[3].shape[1] == op.out.cp
                    a += [dp(hd[3]), dp(hd[4]), hd[2].c, hd[5], bid(hd[2])]
                e = op.ext
                a += [1, dp(e["ps"]), dp(e["pt"]), bid(e["res"]), e["res_mode"], e["relu_last"]] if e is not None else [0] * 6
            elif k == "pair":
                _, xa, yb = op
                a = [bid(xa.src0), dp(xa.w), dp(xa.bias), xa.relu, xa.out.cp, dp(yb.w), dp(yb.bias), yb.relu, bid(yb.out),
                     1 if yb.need_full else 0, bid(yb.out_pool), bid(xa.out) if xa.out.buf is not None else -1]
            elif k == "bneck":
                a = self._bneck_words(op, bid, dp)
            elif k == "conv1x1":
                e = op.ext
                a = [bid(op.src0), dp(op.w), dp(op.bias), op.relu, op.stride_word, 1 if e else 0, dp(e["ps"]) if e else 0,
                     dp(e["pt"]) if e else 0, bid(e["res"]) if e else -1, e["relu_last"] if e else 0, bid(op.out)]
            elif k == "convt2":
                _, s_, phases, bias, o, relu, ksz, ext, _cin = op
                a = [bid(s_)] + [dp(q) for q in phases] + [ksz, dp(bias), relu, 1 if ext else 0, dp(ext["ps"]) if ext else 0,
                                                            dp(ext["pt"]) if ext else 0, ext["relu_last"] if ext else 0, bid(o)]
            elif k == "convt":
                _, s_, w, bias, o, relu = op
                a = [bid(s_), dp(w), dp(bias), relu, bid(o)]
            elif k == "poolg":
                _, s_, o, kk, stride, pad, pad_zero = op
                a = [bid(s_), bid(o), kk, stride, -1 if pad is None else pad, pad_zero]
            elif k == "imgconv":
                _, o, w, bias, cin, relu, _nm, kk, stride, ps, pt, cin_w, in_affine, pads, mf = op
                a = [bid(o), dp(w), dp(bias), cin, relu, kk[0], kk[1], stride, dp(ps), dp(pt), cin_w, dp(in_affine),
                     1 if pads is not None else 0, pads[0] if pads is not None else 0, pads[1] if pads is not None else 0,
                     dp(mf["w"]) if mf else 0, dp(mf["bias"]) if mf else 0, mf["has_mean"] if mf else 0,
                     mf["cin_w"] if mf else 0]  # (weight channels of the PACKED operand: 1 for a tiled grayscale frame)
            elif k == "add":
                _, ta, tb, half, relu, o = op
                a = [bid(ta), bid(tb), half, relu, bid(o)]
            elif k == "head":
                _, s_, o, w, bias, act = op
                a = [bid(s_), dp(w), dp(bias), o.c, act, bid(o)]
            elif k == "pool":
                a = [bid(op[1]), bid(op[2])]
            elif k == "up":
                a = [bid(op[1]), bid(op[2]), op[3]]
            else:
                raise AssertionError(k)
            ops.append([self._K[k], len(a)] + [int(v) for v in a])
        words = [self.PLAN_MAGIC, n_ids[0], len(self.outputs), len(ops), int(self.in_channels), int(self.max_stride), int(self.layout)]
        for i in range(n_ids[0]):
            words += bufs.get(i, [16, 1, 1, 2])  # ids the fusion passes retired: virtual placeholders
        for o in self.outputs:
            words += [o.buf, o.c, 1 if o.kind == "f32out" else 0]
        for r in ops:
            words += r
        return np.asarray(words, dtype=np.int64)

    def _handle(self):
        """The `sa_network_t*` of this plan (created on first use; the plan is immutable after `_compile`)."""
        if getattr(self, "_net", None) is None:
            words = self.plan_words()
            out = C.c_void_p()
            check(self._h.sa_network_create(words.ctypes.data_as(C.c_void_p), words.size, C.byref(out)), "sa_network_create")
            self._net = out
        return self._net

    def __del__(self):
        net, self._net = getattr(self, "_net", None), None
        if net:
            try:
                self._h.sa_network_destroy(net)
            except Exception:  # noqa: BLE001 -- interpreter shutdown
                pass

    # ------------------------------------------------------------------ run
    def _get_buffers(self, B, H, W):
        """One device workspace per input shape (sa_network_workspace_bytes), zeroed once, kept resident; `bufs[i]` are typed
        views of the plan's tensors inside it at the offsets the C executor uses (sa_network_buffer)."""
        key = (B, H, W)
        if key not in self._buffers:
            net, h = self._handle(), self._h
            n = int(h.sa_network_workspace_bytes(net, B, H, W))
            ws = torch.zeros((max(n, 256),), dtype=torch.uint8, device=self.device)
            bufs = {}
            for i, (c_alloc, num, den, dt) in self.buf_meta.items():
                hh, ww, cp, f32 = C.c_int(), C.c_int(), C.c_int(), C.c_int()
                addr = h.sa_network_buffer(net, i, B, H, W, _ptr(ws), C.byref(hh), C.byref(ww), C.byref(cp), C.byref(f32))
                assert addr and (hh.value, ww.value, cp.value) == (H * num // den, W * num // den, c_alloc)
                off = addr - ws.data_ptr()
                dtype = self._tdtype if dt == "bf16" else torch.float32
                nbytes = B * hh.value * ww.value * c_alloc * (2 if dt == "bf16" else 4)
                bufs[i] = ws[off:off + nbytes].view(dtype).view(B, hh.value, ww.value, c_alloc)
            self._buffers = {key: bufs}  # keep one shape resident
            self._workspace = ws
            self._slot1 = {}
            self._range_checked = False
        return self._buffers[key]

    def _slot_buffers(self, bufs, slot):
        """Output buffers of slot 1 (slot 0 = the base set): a second copy of every model-output tensor so that the
        consumer of call i can still read its outputs while call i+1 runs on another stream."""
        if slot == 0:
            return bufs
        view = dict(bufs)
        for o in self.outputs:
            if o.buf not in self._slot1:
                self._slot1[o.buf] = torch.zeros_like(bufs[o.buf])
            view[o.buf] = self._slot1[o.buf]
        return view

    def rescale_head(self, output_index: int, scale, shift):
        """out' = scale[c] * out + shift[c] folded into the 1x1 head's weights (used to calibrate random heads)."""
        target = self.outputs[output_index]
        for op in self._all_heads():
            if op[0] == "head" and op[2] is target:
                w, b = op[3], op[4]
                sc = torch.as_tensor(scale, dtype=torch.float32, device=w.device).reshape(-1)
                sh = torch.as_tensor(shift, dtype=torch.float32, device=w.device).reshape(-1)
                w.mul_(sc[:, None])
                b.mul_(sc).add_(sh)
                return
        raise KeyError(output_index)

    def fused_head_names(self):
        """Output layer names whose 1x1 head is computed inside the producing conv's epilogue (on the matrix cores, from
        the same bf16-rounded features the un-fused path would store)."""
        names = []
        for op in self.plan:
            if op[0] == "conv":
                for hd in op.heads:
                    names.append(self.output_names[[id(o) for o in self.outputs].index(id(hd[2]))])
        return names

    def _all_heads(self):
        for op in self.plan:
            if op[0] == "head":
                yield op
            elif op[0] == "conv":
                for hd in op.heads:
                    yield hd

    def export_head(self, output_index: int):
        """-> (kernel (1,1,Cin,Cout) float32, bias (Cout,)) of the 1x1 head as currently held on the device."""
        target = self.outputs[output_index]
        for op in self._all_heads():
            if op[0] == "head" and op[2] is target:
                s, w, b = op[1], op[3].cpu().numpy(), op[4].cpu().numpy()
                return np.ascontiguousarray(w[:, : s.c].T)[None, None], b.copy()
        raise KeyError(output_index)

    def op_descriptions(self, H, W):
        """[(kind, name-ish, algorithmic FLOPs per frame)] in plan order."""
        out = []
        for op in self.plan:
            k = op[0]
            if k == "stem2":
                so, cin, o = op[1][1], op[1][4], op[2].out  # ["stem2", stem op, ConvOp, stem16 blob]
                hh = H * o.num // o.den
                f = 2 * hh * (W * o.num // o.den) * (cin * so.c + so.c * o.c) * 9
                out.append(("conv", f"stem+conv3x3 {cin}->{so.c}->{o.c} @{hh}", f))
            elif k == "stem":
                o, cin = op[1], op[4]
                f = 2 * (H * o.num // o.den) * (W * o.num // o.den) * cin * o.c * 9
                out.append((k, f"stem {cin}->{o.c} @{H * o.num // o.den}", f))
            elif k == "pair":
                s0, mid, o = op[1].src0, op[1].out, op[2].out  # ["pair", ConvOp a, ConvOp b]
                hh = H * o.num // o.den
                f = 2 * hh * (W * o.num // o.den) * (s0.c * mid.c + mid.c * o.c) * 9
                out.append(("conv", f"conv3x3 pair {s0.c}->{mid.c}->{o.c} @{hh}", f))
            elif k == "bneck":
                c, x, y = op[1], op[2], op[3]
                hh, ww = H * c.out.num // c.out.den, W * c.out.num // c.out.den
                f = 2 * hh * ww * (c.src0.c * c.out.c * 9 + c.out.c * x.out.c + (x.out.c * y.out.c if y is not None else 0))
                nm = f"bneck 3x3 {c.src0.c}->{c.out.c} | 1x1 ->{x.out.c}" + (" +res" if x.ext and x.ext["res"] is not None else "")
                out.append(("conv", nm + (f" | 1x1 ->{y.out.c}" if y is not None else "") + f" @{hh}", f))
            elif k == "conv1x1":
                s0, o = op.src0, op.out
                f = 2 * (H * o.num // o.den) * (W * o.num // o.den) * s0.c * o.c * op.ksize ** 2
                nm = f"conv{op.ksize}x{op.ksize}s{op.stride} {s0.c}->{o.c} @{H * o.num // o.den}"
                if op.ext is not None:
                    nm += " +affine" + (" +res" if op.ext["res"] is not None else "")
                out.append(("conv", nm, f))
            elif k == "convt2":
                s, o, ksz = op[1], op[4], op[6]
                f = 2 * (H * s.num // s.den) * (W * s.num // s.den) * s.c * o.c * ksz * ksz
                out.append(("convt", f"convT{ksz} {s.c}->{o.c} @{H * s.num // s.den}", f))
            elif k == "imgconv":
                o, cin, kk = op[1], op[11], op[7]
                f = 2 * (H * o.num // o.den) * (W * o.num // o.den) * cin * o.c * kk[0] * kk[1]
                out.append(("conv", f"imgconv{kk[0]}x{kk[1]}s{op[8]} {cin}->{o.c} @{H * o.num // o.den}", f))
            elif k == "conv":
                s0, s1, o = op.src0, op.src1, op.out
                cin = s0.c + (s1.c if s1 is not None else 0)
                f = 2 * (H * o.num // o.den) * (W * o.num // o.den) * cin * o.c * 9
                nm = f"conv3x3 {cin}->{o.c} @{H * o.num // o.den} mode{op.mode}"
                for hd in op.heads:
                    nm += f" +head{hd[2].c}"
                if op.ext is not None:
                    nm += " +affine" + (" +res" if op.ext["res"] is not None else "")
                out.append((k, nm, f))
            elif k == "head":
                s, o = op[1], op[2]
                out.append((k, f"head {s.c}->{o.c} @{H * s.num // s.den}", 2 * (H * s.num // s.den) * (W * s.num // s.den) * s.c * o.c))
            elif k == "convt":
                s, o = op[1], op[4]
                out.append((k, f"convT {s.c}->{o.c} @{H * s.num // s.den}", 2 * (H * s.num // s.den) * (W * s.num // s.den) * s.c * o.c * 9))
            else:
                out.append((k, k, 0))
        return out

    def forward(self, imgs: torch.Tensor, profile: Optional[list] = None, slot: int = 0) -> List[torch.Tensor]:
        """imgs: (B, H, W, C) uint8 or float32 CUDA tensor, H and W multiples of the max stride.
        Returns the model outputs (float32, NHWC) in `output_names` order. The returned tensors are
        views of cached buffers that the next call with the same `slot` (0 or 1) overwrites."""
        assert imgs.is_cuda and imgs.is_contiguous()
        B, H, W, Cin = imgs.shape
        if H % self.max_stride or W % self.max_stride:
            raise ValueError(f"input size {(H, W)} must be a multiple of the model stride {self.max_stride}")
        bufs = self._slot_buffers(self._get_buffers(B, H, W), slot)
        h = self._h
        st = _stream()
        if profile is None and os.environ.get("SA_ENGINE_PYTHON_LOOP", "0") != "1":
            # product path: the whole launch sequence runs inside the library (csrc/network.hip); the Python loop below is
            # the same sequence with a HIP event pair around every launch (bench.py --layers, tools/net_profile.py)
            if imgs.dtype not in (torch.uint8, torch.float32):
                raise ValueError("images must be uint8 or float32")
            if Cin != self.in_channels:
                raise ValueError(f"model expects {self.in_channels} input channels, got {Cin}")
            outs = [bufs[o.buf] if o.kind == "f32out" else
                    torch.empty((B, H * o.num // o.den, W * o.num // o.den, o.c), dtype=torch.float32, device=self.device)
                    for o in self.outputs]
            arr = (C.c_void_p * len(outs))(*[t.data_ptr() for t in outs])
            ws = self._workspace
            check(h.sa_network_forward(self._handle(), _ptr(imgs), 1 if imgs.dtype == torch.uint8 else 0, B, H, W, Cin, arr,
                                       _ptr(ws), ws.numel(), st), "sa_network_forward")
            if self.dtype == "fp16" and not self._range_checked and self._range_gate(bufs, imgs):
                return self.forward(imgs, profile, slot)
            return outs

        def hw(tt):
            return H * tt.num // tt.den, W * tt.num // tt.den

        for op in self.plan:
            kind = op[0]
            if profile is not None:
                ev0 = torch.cuda.Event(enable_timing=True)
                ev1 = torch.cuda.Event(enable_timing=True)
                ev0.record()
                profile.append((ev0, ev1))
            if kind == "stem2":
                _, (_k, so, w0, b0, cin, relu0, _n0), (_c, _s0, _s1, _m, w1, b1, o, relu1, o_pool, need_full, _hd, _nm, _ext), w1_16 = op
                if cin != Cin:
                    raise ValueError(f"model expects {cin} input channels, got {Cin}")
                is_u8 = 1 if imgs.dtype == torch.uint8 else 0
                if not is_u8 and imgs.dtype != torch.float32:
                    raise ValueError("images must be uint8 or float32")
                if is_u8 and w1_16 is not None and self.use_stem16:
                    check(h.sa_stem16_u8_bf16(_ptr(imgs), B, H, W, cin, _ptr(w1_16), relu0, relu1,
                                              _ptr(bufs[o.buf]) if need_full else None,
                                              _ptr(bufs[o_pool.buf]) if o_pool is not None else None, st),
                          "sa_stem16_u8_bf16")
                else:
                    check(h.sa_stem_conv3x3x2_bf16(_ptr(imgs), is_u8, B, H, W, cin, _ptr(w0), _ptr(b0), so.cp, relu0,
                                                   _ptr(w1), _ptr(b1), o.cp, relu1,
                                                   _ptr(bufs[o.buf]) if need_full else None,
                                                   _ptr(bufs[o_pool.buf]) if o_pool is not None else None, self.layout, st),
                          "sa_stem_conv3x3x2_bf16")
            elif kind == "stem":
                _, o, w, bias, cin, relu, _n0 = op
                if cin != Cin:
                    raise ValueError(f"model expects {cin} input channels, got {Cin}")
                is_u8 = 1 if imgs.dtype == torch.uint8 else 0
                if not is_u8 and imgs.dtype != torch.float32:
                    raise ValueError("images must be uint8 or float32")
                check(h.sa_stem_conv3x3(_ptr(imgs), is_u8, B, H, W, cin, _ptr(w), _ptr(bias), o.cp, relu,
                                        _ptr(bufs[o.buf]), st), "sa_stem_conv3x3")
            elif kind == "conv" and op.heads:
                _, s0, s1, mode, w, bias, o, relu, o_pool, need_full, heads, _nm, ext = op
                oh, ow = hw(o)
                n = len(heads)
                arr = (C.c_void_p * n)
                hw_ = arr(*[hd[3].data_ptr() for hd in heads])
                hb_ = arr(*[hd[4].data_ptr() for hd in heads])
                hd_ = arr(*[bufs[hd[2].buf].data_ptr() for hd in heads])
                hc_ = (C.c_int * n)(*[hd[2].c for hd in heads])
                ha_ = (C.c_int * n)(*[hd[5] for hd in heads])
                for hd in heads:  # the fused kernel indexes head weights with the producer's padded channel count
                    assert hd[3].shape[1] == o.cp
                if ext is not None:  # Conv + BatchNormalization + ReLU in front of the heads (_fuse_heads: no residual)
                    assert ext["res"] is None and o_pool is None
                    check(h.sa_conv3x3_ex_heads_bf16(_ptr(bufs[s0.buf]), s0.cp, _ptr(bufs[s1.buf]) if s1 is not None else None,
                                                     s1.cp if s1 is not None else 0, mode | self.layout, _ptr(w), _ptr(bias), o.cp, relu,
                                                     B, oh, ow, _ptr(bufs[o.buf]) if need_full else None, _ptr(ext["ps"]),
                                                     _ptr(ext["pt"]), ext["relu_last"], n, hw_, hb_, hc_, ha_, hd_, st),
                          "sa_conv3x3_ex_heads_bf16")
                else:
                    check(h.sa_conv3x3_heads_bf16(_ptr(bufs[s0.buf]), s0.cp, _ptr(bufs[s1.buf]) if s1 is not None else None,
                                                  s1.cp if s1 is not None else 0, mode | self.layout, _ptr(w), _ptr(bias), o.cp, relu, B,
                                                  oh, ow, _ptr(bufs[o.buf]) if need_full else None, n, hw_, hb_, hc_, ha_, hd_, st),
                          "sa_conv3x3_heads_bf16")
            elif kind == "conv" and op.ext is not None:
                _, s0, s1, mode, w, bias, o, relu, o_pool, need_full, _heads, _nm, ext = op
                oh, ow = hw(o)
                res = ext["res"]
                check(h.sa_conv3x3_ex_bf16(_ptr(bufs[s0.buf]), s0.cp, _ptr(bufs[s1.buf]) if s1 is not None else None,
                                           s1.cp if s1 is not None else 0, mode | self.layout, _ptr(w), _ptr(bias), o.cp, relu, B, oh, ow,
                                           _ptr(bufs[o.buf]) if need_full else None,
                                           _ptr(bufs[o_pool.buf]) if o_pool is not None else None,
                                           _ptr(ext["ps"]), _ptr(ext["pt"]),
                                           _ptr(bufs[res.buf]) if res is not None else None, ext["res_mode"],
                                           ext["relu_last"], st), "sa_conv3x3_ex_bf16")
            elif kind == "pair":
                _, xa, yb = op
                s0, o, o_pool, need_full = xa[1], yb[6], yb[8], yb[9]
                oh, ow = hw(o)
                mid = xa[6]
                if mid.buf is not None and B * ((oh + 15) // 16) * ((ow + 31) // 32) < 6 * self._n_cu():
                    # (a small launch of the 32 -> 64 -> 64 block: two convolutions through the kept intermediate buffer, see _fuse_pairs)
                    check(h.sa_conv3x3_bf16(_ptr(bufs[s0.buf]), s0.cp, None, 0, self.layout, _ptr(xa[4]), _ptr(xa[5]), mid.cp, xa[7], B, oh, ow,
                                            _ptr(bufs[mid.buf]), None, st), "sa_conv3x3_bf16")
                    check(h.sa_conv3x3_bf16(_ptr(bufs[mid.buf]), mid.cp, None, 0, self.layout, _ptr(yb[4]), _ptr(yb[5]), o.cp, yb[7], B, oh, ow,
                                            _ptr(bufs[o.buf]) if need_full else None,
                                            _ptr(bufs[o_pool.buf]) if o_pool is not None else None, st), "sa_conv3x3_bf16")
                else:
                    check(h.sa_conv3x3_pair_bf16(_ptr(bufs[s0.buf]), s0.cp, _ptr(xa[4]), _ptr(xa[5]), xa[7], xa[6].cp, _ptr(yb[4]),
                                                 _ptr(yb[5]), yb[7], o.cp, B, oh, ow, _ptr(bufs[o.buf]) if need_full else None,
                                                 _ptr(bufs[o_pool.buf]) if o_pool is not None else None, self.layout, st),
                          "sa_conv3x3_pair_bf16")
            elif kind == "bneck":
                _, c, x, y, xw, yw = op
                oh, ow = hw(c.out)
                ce, xe, ye = c.ext, x.ext, (y.ext if y is not None else None)
                xres = xe["res"] if xe else None
                check(h.sa_conv3x3_bneck_bf16(
                    _ptr(bufs[c.src0.buf]), c.src0.cp, self.layout, _ptr(c.w), _ptr(c.bias), c.relu, _ptr(ce["ps"]), _ptr(ce["pt"]),
                    ce["relu_last"], B, oh, ow, _ptr(xw), _ptr(x.bias), _ptr(xe["ps"]) if xe else None, _ptr(xe["pt"]) if xe else None,
                    _ptr(bufs[xres.buf]) if xres is not None else None, x.relu, xe["relu_last"] if xe else 0, x.out.cp,
                    _ptr(bufs[x.out.buf]), _ptr(yw) if y is not None else None, _ptr(y.bias) if y is not None else None,
                    _ptr(ye["ps"]) if ye else None, _ptr(ye["pt"]) if ye else None, y.relu if y is not None else 0,
                    ye["relu_last"] if ye else 0, y.out.cp if y is not None else 0, _ptr(bufs[y.out.buf]) if y is not None else None, st),
                    "sa_conv3x3_bneck_bf16")
            elif kind == "conv1x1":
                _, s0, _s1, _sw, w, bias, o, relu, _op, _nf, _heads, _nm, ext = op
                sh, sw = hw(s0)
                res = ext["res"] if ext else None
                stride = op.stride
                if op.ksize > 1:
                    check(h.sa_convk_bf16(_ptr(bufs[s0.buf]), s0.cp, _ptr(w), op.ksize, _ptr(bias), o.cp, relu | self.layout, B, sh, sw,
                                          _ptr(ext["ps"]) if ext else None, _ptr(ext["pt"]) if ext else None,
                                          _ptr(bufs[res.buf]) if res is not None else None, ext["relu_last"] if ext else 0,
                                          _ptr(bufs[o.buf]), st), "sa_convk_bf16")
                else:
                    check(h.sa_conv1x1_bf16(_ptr(bufs[s0.buf]), s0.cp, _ptr(w), _ptr(bias), o.cp, relu | self.layout, B, sh, sw, stride,
                                            _ptr(ext["ps"]) if ext else None, _ptr(ext["pt"]) if ext else None,
                                            _ptr(bufs[res.buf]) if res is not None else None, ext["relu_last"] if ext else 0,
                                            _ptr(bufs[o.buf]), st), "sa_conv1x1_bf16")
            elif kind == "convt2":
                _, s, phases, bias, o, relu, ksz, ext, _cin = op
                sh, sw = hw(s)
                arr = (C.c_void_p * 4)(*[q.data_ptr() for q in phases])
                check(h.sa_convt_s2_bf16(_ptr(bufs[s.buf]), s.cp, arr, ksz, _ptr(bias), o.cp, relu | self.layout, B, sh, sw,
                                         _ptr(ext["ps"]) if ext else None, _ptr(ext["pt"]) if ext else None,
                                         ext["relu_last"] if ext else 0, _ptr(bufs[o.buf]), st), "sa_convt_s2_bf16")
            elif kind == "poolg":
                _, s, o, kk, stride, pad, pad_zero = op
                sh, sw = hw(s)
                oh, ow = hw(o)
                if pad is None:
                    pad_t = max((oh - 1) * stride + kk - sh, 0) // 2
                    pad_l = max((ow - 1) * stride + kk - sw, 0) // 2
                else:
                    pad_t = pad_l = pad
                nb, cpp = (B * (s.cp // 16), 16) if self.planar else (B, s.cp)  # planes: a plane of a frame = a 16-channel frame
                check(h.sa_maxpool_bf16(_ptr(bufs[s.buf]), nb, sh, sw, cpp, kk, stride, pad_t, pad_l, pad_zero, oh, ow,
                                        _ptr(bufs[o.buf]), st), "sa_maxpool_bf16")
            elif kind == "imgconv":
                _, o, w, bias, cin, relu, _nm, k, stride, ps, pt, cin_w, in_affine, pads, mf = op
                if cin != Cin:
                    raise ValueError(f"model expects {cin} input channels, got {Cin}")
                is_u8 = 1 if imgs.dtype == torch.uint8 else 0
                if not is_u8 and imgs.dtype != torch.float32:
                    raise ValueError("images must be uint8 or float32")
                oh, ow = hw(o)
                # TF "SAME": out = ceil(in / s), pad_total = max((out - 1) * s + k - in, 0), pad_before = pad_total // 2
                pt_ = max((oh - 1) * stride + k[0] - H, 0) // 2
                pl_ = max((ow - 1) * stride + k[1] - W, 0) // 2
                if pads is not None:  # explicit ZeroPadding2D + valid conv
                    pt_, pl_ = pads
                if is_u8 and mf is not None:
                    check(h.sa_imgconv_u8_bf16(_ptr(imgs), B, H, W, cin, mf["cin_w"], k[0], stride, pt_, pl_, oh, ow, _ptr(mf["w"]),
                                               _ptr(mf["bias"]), o.cp, relu | self.layout, mf["has_mean"],
                                               _ptr(ps) if ps is not None else None, _ptr(pt) if pt is not None else None,
                                               _ptr(bufs[o.buf]), st), "sa_imgconv_u8_bf16")
                    if profile is not None:
                        ev1.record()
                    continue
                check(h.sa_image_conv_bf16(_ptr(imgs), is_u8, B, H, W, cin, cin_w,
                                           _ptr(in_affine) if in_affine is not None else None, k[0], k[1], stride, pt_, pl_,
                                           oh, ow, _ptr(w),
                                           _ptr(bias), o.cp, relu | self.layout, _ptr(ps) if ps is not None else None,
                                           _ptr(pt) if pt is not None else None, _ptr(bufs[o.buf]), st), "sa_image_conv_bf16")
            elif kind == "add":
                _, a, b, half, relu, o = op
                oh, ow = hw(o)
                nb, cpp = (B * (o.cp // 16), 16) if self.planar else (B, o.cp)
                check(h.sa_add_bf16(_ptr(bufs[a.buf]), _ptr(bufs[b.buf]), nb, oh, ow, cpp, half, relu, _ptr(bufs[o.buf]), st),
                      "sa_add_bf16")
            elif kind == "conv":
                _, s0, s1, mode, w, bias, o, relu, o_pool, need_full, _heads, _nm, _ext = op
                oh, ow = hw(o)
                check(h.sa_conv3x3_bf16(_ptr(bufs[s0.buf]), s0.cp, _ptr(bufs[s1.buf]) if s1 is not None else None,
                                        s1.cp if s1 is not None else 0, mode | self.layout, _ptr(w), _ptr(bias), o.cp, relu, B, oh, ow,
                                        _ptr(bufs[o.buf]) if need_full else None,
                                        _ptr(bufs[o_pool.buf]) if o_pool is not None else None, st), "sa_conv3x3_bf16")
            elif kind == "head":
                _, s, o, w, bias, act = op
                sh, sw = hw(s)
                check(h.sa_conv1x1_head(_ptr(bufs[s.buf]), s.cp, _ptr(w), _ptr(bias), o.c, act | self.layout, B, sh, sw,
                                        _ptr(bufs[o.buf]), st), "sa_conv1x1_head")
            elif kind == "pool":
                _, s, o = op
                sh, sw = hw(s)
                nb, cpp = (B * (s.cp // 16), 16) if self.planar else (B, s.cp)
                check(h.sa_maxpool2x2_bf16(_ptr(bufs[s.buf]), nb, sh, sw, cpp, _ptr(bufs[o.buf]), st), "sa_maxpool2x2_bf16")
            elif kind == "up":
                _, s, o, bil = op
                sh, sw = hw(s)
                if self.planar:  # every 16-channel plane of every frame is a frame of 16 channels
                    check(h.sa_upsample2x_bf16(_ptr(bufs[s.buf]), B * (s.cp // 16), sh, sw, 16, bil, _ptr(bufs[o.buf]), st), "sa_upsample2x_bf16")
                else:
                    check(h.sa_upsample2x_bf16(_ptr(bufs[s.buf]), B, sh, sw, s.cp, bil, _ptr(bufs[o.buf]), st), "sa_upsample2x_bf16")
            elif kind == "convt":
                _, s, w, bias, o, relu = op
                sh, sw = hw(s)
                check(h.sa_convt3x3s2_bf16(_ptr(bufs[s.buf]), s.cp, _ptr(w), _ptr(bias), o.cp, relu, B, sh, sw,
                                           _ptr(bufs[o.buf]), st), "sa_convt3x3s2_bf16")
            else:
                raise AssertionError(kind)
            if profile is not None:
                ev1.record()
        if self.dtype == "fp16" and not self._range_checked and profile is None and self._range_gate(bufs, imgs):
            return self.forward(imgs, profile, slot)
        from ..ops import from_bf16
        return [bufs[o.buf] if o.kind == "f32out" else from_bf16(bufs[o.buf], o.c) for o in self.outputs]

    # ------------------------------------------------------------------ range-safe fp16 (nn/range_scaling.py)
    def layer_aliases(self) -> List[List[str]]:
        """Layer names whose outputs live in ONE stored tensor of this plan (a conv with its fused epilogue layers)."""
        groups: Dict[int, List[str]] = {}
        for name, v in self._tensor_of.items():
            if v.kind in ("real", "f32out"):
                groups.setdefault(id(v), []).append(name)
        return [g for g in groups.values() if len(g) > 1]

    def _scan_tensors(self, tensors: Dict) -> Dict:
        """key -> (largest finite |x|, an inf was seen, a NaN was seen) for 16-bit storage / float32 device tensors: one
        `sa_tensor_absmax` launch per tensor (wave-reduced integer max over the magnitude bits, csrc/layers.hip), ONE copy to
        the host for all of them -- no torch arithmetic."""
        keys = list(tensors)
        out = torch.zeros((max(len(keys), 1), 2), dtype=torch.float32, device=self.device)
        st = _stream()
        for j, k in enumerate(keys):
            t = tensors[k]
            assert t.is_contiguous() and t.dtype in (self._tdtype, torch.float32), (k, t.dtype)
            check(self._h.sa_tensor_absmax(_ptr(t), t.numel(), 1 if t.dtype == torch.float32 else 0, _ptr(out[j]), st),
                  "sa_tensor_absmax")
        host = out.cpu().numpy()
        flags = host[:, 1].copy().view(np.uint32)
        return {k: (float(host[j, 0]), bool(flags[j] & 1), bool(flags[j] & 2)) for j, k in enumerate(keys)}

    def layer_ranges(self, imgs: torch.Tensor) -> Dict[str, float]:
        """max |activation| of every layer whose output this plan stores in HBM, on the given batch (one forward + one
        reduction launch per stored tensor; calibration, not a hot path). Layers sharing a tensor (`layer_aliases`) report the
        same value; outputs that only ever live in LDS / registers (fused stem, fused encoder block) are absent. A tensor that
        holds an infinity reports `inf`."""
        self.forward(imgs)
        bufs = next(iter(self._buffers.values()))
        stored = {v.buf: bufs[v.buf] for v in self._tensor_of.values()
                  if v.kind in ("real", "f32out") and v.buf is not None and v.buf in bufs}
        scan = self._scan_tensors(stored)
        out = {}
        for name, v in self._tensor_of.items():
            if v.kind in ("real", "f32out") and v.buf in scan:
                m, has_inf, has_nan = scan[v.buf]
                out[name] = float("inf") if (has_inf or has_nan) else m
        return out

    # ---- what the range gate does, in one place ------------------------------------------------------------------------------
    #   first batch of an input shape (fp16 storage): scan every stored tensor.
    #     fits (finite, <= range / 4)                      -> nothing happens, the plan stays bit for bit what it was
    #     does not fit, range_safe, not yet scaled         -> calibrate on THIS batch (all of its frames), fold, re-compile, re-run
    #     does not fit and (scaling off | already scaled)  -> inf / NaN seen: FloatingPointError; finite: a warning
    #   under torch.distributed (world > 1) NOTHING is rescaled in here (a collective inside forward() would deadlock: shard sizes
    #   differ, empty shards skip the network). Without a promise an overflow raises at once; after `defer_range_agreement()` (the
    #   predictors, bench.py) the scan is recorded and the caller runs `dist_agree_range` on EVERY rank once, after the first
    #   global batch -- scan results and, if needed, per-layer ranges are MAX-reduced there, so all ranks fold the
    #   same exponents (ranks must run numerically identical networks) and nothing is rescaled implicitly afterwards.
    #   Deterministic alternative: `calibrate_range(frames)` right after loading, or `range_log2_scale=` (persisted exponents).
    def _range_gate(self, bufs, imgs) -> bool:
        """First batch of an input shape under fp16 storage: scan the stored tensors. -> True when the plan was re-compiled
        with range scales (the caller runs the batch again), False when the network is left as it is. Raises
        FloatingPointError only when the scan actually saw inf / NaN and scaling is off, was already applied, or cannot help."""
        worst, nonfinite, where = self._check_fp16_range(bufs)
        from .range_scaling import dist_world

        if dist_world() > 1:
            # frame-sharded run: NO collective in here (ranks see different shapes, empty shards skip the network: "first batch
            # of a shape" is not a common event and a collective would deadlock), and nothing is rescaled implicitly: ranks
            # must run numerically identical networks.
            if self._dist_agreed:
                if nonfinite:
                    self._range_checked = False
                    raise FloatingPointError(
                        f"activations left the range of fp16 storage (65504) in plan tensor {where} after the ranks agreed on their "
                        "range scales: call calibrate_range() on representative frames (or load the model with dtype='bf16')")
                return False
            if not self._dist_defer:
                # nobody promised an agreement (a custom loop, a tool, DeviceNetwork.forward called directly inside an
                # initialised process group): an overflow is an error HERE, as without range scaling
                if nonfinite:
                    self._range_checked = False
                    raise FloatingPointError(
                        f"activations left the range of fp16 storage (65504) in plan tensor {where} inside an initialised process "
                        "group, where range scales are never chosen implicitly: call calibrate_range() on the same frames on every "
                        "rank, or defer_range_agreement() before the first forward and dist_agree_range() on EVERY rank after it "
                        "(the predictors do), or load the model with dtype='bf16'")
                if worst > 65504.0 / 4:
                    import warnings

                    warnings.warn(f"fp16 storage: the largest activation of the first batch is {worst:.0f}, within 4x of the "
                                  "format's range (65504); consider calibrate_range() or dtype='bf16' for this model")
                return False
            # deferred mode (the predictors, bench.py): the scan is recorded and all ranks agree once, at one program point
            # (`dist_agree_range`). A SECOND scan that arrives while an overflow is still waiting for that agreement means the
            # promise was not kept: raise instead of handing out overflowed results batch after batch.
            pw, pn = self._pending_scan or (0.0, False)
            if pn and self._pending_scan is not None:
                self._range_checked = False
                raise FloatingPointError(
                    "activations left the range of fp16 storage (65504) on an earlier batch and dist_agree_range() was not called "
                    "since (defer_range_agreement() promises one call on every rank after the first global batch)")
            # ONE batch is kept until the ranks have agreed -- what this rank would calibrate on: the batch of the WORST scan so far
            # (round 5 kept the first one: an overflow that came from a later shape was then calibrated on a batch that had not
            # overflowed, and the next batch raised "after the ranks agreed")
            if self._pending_imgs is None or worst > pw or (nonfinite and not pn):
                self._pending_imgs = imgs
            self._pending_scan = (max(worst, pw), nonfinite or pn)
            return False
        if not nonfinite and worst <= 65504.0 / 4:
            return False
        # not yet scaled: calibrate on this batch. Already scaled (from a first batch, not from explicit / persisted exponents
        # -- those are the user's decision) and this batch still overflowed: calibrate again with the LARGER of the old and the
        # new ranges (a ratchet: scales only ever get more conservative)
        may = self.range_log2_scale is None or (nonfinite and not self._range_calibrated)
        if self.range_safe and may and self._apply_range_scaling(imgs, must=nonfinite):
            return True
        if nonfinite:
            self._range_checked = False
            raise FloatingPointError(
                f"activations left the range of fp16 storage (65504) in plan tensor {where}"
                + (" although range scales are applied (a batch far outside the calibration batch: call calibrate_range() on "
                   "representative frames)" if self.range_log2_scale is not None else "")
                + ": load the model with dtype='bf16' (or SLEAP_AMD_DTYPE=bf16), which has fp32's range")
        import warnings

        warnings.warn(f"fp16 storage: the largest activation of the first batch is {worst:.0f}, within 4x of the format's "
                      "range (65504); consider dtype='bf16' for this model")
        return False

    def measure_ranges(self, imgs: torch.Tensor, chunk: int = 8):
        """-> ({layer: max |activation|} over ALL frames of `imgs`, the twin's layer aliases): measured with a bf16-storage twin
        of this network (fp32's range, every layer output stored, built from the un-scaled master weights), `chunk` frames at
        a time. This rank's frames only: ranks combine their ranges in `dist_agree_range`."""
        twin = DeviceNetwork(self.model_config, self.master_weights, device=self.device, fuse_heads=False, fuse_stem=False,
                             fuse_pairs=False, fuse_upsample=False, mfma_convt=self.mfma_convt, mfma_stem=self.mfma_stem,
                             dtype="bf16", fuse_bneck=False)
        ranges: Dict[str, float] = {}
        n = int(imgs.shape[0])
        chunk = max(1, min(chunk, n))
        for i in range(0, n, chunk):
            part = imgs[i:i + chunk]
            if part.shape[0] < chunk and i:  # keep ONE buffer shape in the twin: the last chunk overlaps the one before
                part = imgs[n - chunk:n]
            for k, v in twin.layer_ranges(part.contiguous()).items():
                ranges[k] = max(ranges.get(k, 0.0), v)
        aliases = twin.layer_aliases()
        del twin
        return ranges, aliases

    def defer_range_agreement(self, on: bool = True) -> None:
        """Frame-sharded runs: promise that EVERY rank calls `dist_agree_range()` once after the first global batch. Until then
        the range gate only records its scan (no exception, no rescaling); without this promise an overflow inside an
        initialised process group raises at once. No effect without a process group."""
        self._dist_defer = bool(on)

    def reset_pending_range(self) -> None:
        """Forget the scan / batch recorded for the agreement and scan the next forward again (a network fed by another
        network that was just re-compiled: what it saw so far came from overflowed inputs)."""
        self._pending_scan = None
        self._pending_imgs = None
        self._range_checked = False

    def dist_agree_range(self, imgs: Optional[torch.Tensor] = None) -> bool:
        """Frame-sharded runs (torch.distributed, world > 1): EVERY rank calls this once, at the same program point -- the
        predictors do after the first global batch. `imgs` = the network input this rank would calibrate on (default: the batch
        of its first forward, kept by the range gate; a rank whose shard was empty has none and takes part with neutral values).
        One small MAX all-reduce of the first-forward scans; only if some rank left fp16's comfortable range a second one of the
        per-layer ranges (each rank measures its own frames), after which all ranks fold the SAME exponents. -> True when the
        plan was re-compiled (the caller runs its batch again). No-op without a process group or for bf16 storage."""
        from . import range_scaling as RS

        if RS.dist_world() <= 1 or self.dtype != "fp16" or self._dist_agreed:
            return False
        aliases_box = {}

        def twin():
            return DeviceNetwork(self.model_config, self.master_weights, device=self.device, fuse_heads=False, fuse_stem=False,
                                 fuse_pairs=False, fuse_upsample=False, mfma_convt=self.mfma_convt, mfma_stem=self.mfma_stem,
                                 dtype="bf16", fuse_bneck=False)

        def keys():
            t = twin()
            aliases_box["a"] = t.layer_aliases()
            names = sorted(n for n, v in t._tensor_of.items() if v.kind in ("real", "f32out") and v.buf is not None)
            del t
            return names

        def measure():
            r, aliases_box["a"] = self.measure_ranges(imgs.contiguous())
            return r

        imgs = imgs if imgs is not None else self._pending_imgs
        self._pending_imgs = None
        has = imgs is not None and int(imgs.shape[0]) > 0
        ks = RS.dist_agree(self._pending_scan if has else None, self.range_log2_scale is not None, self.range_safe, keys,
                           measure if has else None, lambda r: RS.plan_scales(self.model_config, r, aliases_box.get("a")), self.device)
        self._dist_agreed = True
        self._dist_defer = False
        self._pending_scan = None
        self._range_calibrated = True  # from here on the exponents are every rank's: never changed implicitly
        if ks:
            self._install_scales(ks)
            return True
        return False

    def calibrate_range(self, imgs: torch.Tensor) -> Dict[str, int]:
        """Explicit, deterministic range calibration (fp16 storage): measure on `imgs` (uint8 / float32 frames on the device,
        the shape the model will see), choose the power-of-two scales, fold them into a COPY of the master weights and
        re-compile. -> the exponents ({} when every tensor fits: the plan is untouched). Persist them
        (`json.dump(net.range_log2_scale)`) and pass `range_log2_scale=` at construction to skip the measurement."""
        if self.dtype != "fp16":
            return {}
        self._apply_range_scaling(imgs.contiguous(), must=False, quiet=True)
        self._range_calibrated = True
        return dict(self.range_log2_scale or {})

    def _apply_range_scaling(self, imgs, must, quiet=False) -> bool:
        """-> True when scales were folded and the plan re-compiled."""
        from . import range_scaling as RS

        ranges, aliases = self.measure_ranges(imgs)
        for k, v in self._range_measured.items():
            ranges[k] = max(ranges.get(k, 0.0), v)
        self._range_measured = dict(ranges)
        ks = RS.plan_scales(self.model_config, ranges, aliases)
        if ks == (self.range_log2_scale or None):
            return False
        if not any(ks.values()):
            # nothing can be rescaled: every tensor fits, or the offender is a model input / output (pinned to scale 1)
            if must:
                return False
            if not quiet:
                import warnings

                warnings.warn("fp16 storage: range calibration found no tensor to rescale (the large values sit in a model "
                              "input / output or in a tensor pinned to scale 1); the plan is unchanged")
            return False
        self._install_scales(ks)
        return True

    def _install_scales(self, ks: Dict[str, int]):
        from . import range_scaling as RS

        self.weights = RS.fold_scales(self.model_config, self.master_weights, ks)  # master_weights stay the model's own
        self.range_log2_scale = dict(ks)
        net, self._net = getattr(self, "_net", None), None
        if net:
            torch.cuda.synchronize(self.device)
            self._h.sa_network_destroy(net)
        self._buffers = {}
        self._compile()

    def _check_fp16_range(self, bufs):
        """fp16 storage has a finite range. Once per input shape (after the first forward; one synchronisation) every stored
        activation tensor is scanned (`sa_tensor_absmax`). -> (largest finite value, inf / NaN seen, first offending tensor).
        Why here and not only at the outputs: ReLU is a v_max, which returns the non-NaN operand, so the NaNs that an
        overflowed (+inf) activation produces downstream (inf - inf) are scrubbed to 0 again and the heads of a badly
        overflowed network can come out FINITE; the first tensor that overflowed, however, holds +inf in HBM.
        (Intermediates that live only in LDS -- fused stem / encoder block -- are not seen by this scan; peak finding's
        SA_STATUS_NONFINITE catches what reaches the maps.)"""
        self._range_checked = True
        scan = self._scan_tensors({i: t for i, t in bufs.items() if t.dtype == torch.float16})
        worst, nonfinite, where = 0.0, False, None
        for i, (m, has_inf, has_nan) in scan.items():
            worst = max(worst, m)
            if (has_inf or has_nan) and not nonfinite:
                nonfinite, where = True, i
        return worst, nonfinite, where

    def conv_flops(self, H, W):
        """2*H*W*Cin*Cout*k*k over all convs for ONE frame (logical channels)."""
        total = 0
        for op in self.plan:
            if op[0] == "stem2":
                so, cin, o = op[1][1], op[1][4], op[2].out
                total += 2 * (H * o.num // o.den) * (W * o.num // o.den) * (cin * so.c + so.c * o.c) * 9
            elif op[0] == "stem":
                o, cin = op[1], op[4]
                total += 2 * (H * o.num // o.den) * (W * o.num // o.den) * cin * o.c * 9
            elif op[0] == "imgconv":
                o, cin, kk = op[1], op[11], op[7]
                total += 2 * (H * o.num // o.den) * (W * o.num // o.den) * cin * o.c * kk[0] * kk[1]
            elif op[0] == "pair":
                s0, mid, o = op[1].src0, op[1].out, op[2].out
                total += 2 * (H * o.num // o.den) * (W * o.num // o.den) * (s0.c * mid.c + mid.c * o.c) * 9
            elif op[0] == "bneck":
                c, x, y = op[1], op[2], op[3]
                total += 2 * (H * c.out.num // c.out.den) * (W * c.out.num // c.out.den) * (
                    c.src0.c * c.out.c * 9 + c.out.c * x.out.c + (x.out.c * y.out.c if y is not None else 0))
            elif op[0] == "conv1x1":
                s0, o = op.src0, op.out
                total += 2 * (H * o.num // o.den) * (W * o.num // o.den) * s0.c * o.c * op.ksize ** 2
            elif op[0] == "convt2":
                s, o, ksz = op[1], op[4], op[6]
                total += 2 * (H * s.num // s.den) * (W * s.num // s.den) * s.c * o.c * ksz * ksz
            elif op[0] == "conv":
                s0, s1, o = op.src0, op.src1, op.out
                cin = s0.c + (s1.c if s1 is not None else 0)
                total += 2 * (H * o.num // o.den) * (W * o.num // o.den) * cin * o.c * 9
            elif op[0] == "head":
                s, o = op[1], op[2]
                total += 2 * (H * s.num // s.den) * (W * s.num // s.den) * s.c * o.c
            elif op[0] == "convt":
                s, o = op[1], op[4]
                total += 2 * (H * s.num // s.den) * (W * s.num // s.den) * s.c * o.c * 9
        return total


def load_keras_npz(path):
    """Read a model extracted by sleap_amd/nn/_h5_extract.py -> (model_config, weights)."""
    z = np.load(path)
    cfg = json.loads(bytes(z["__model_config__"]).decode("utf-8"))
    return cfg, {k: z[k] for k in z.files if k != "__model_config__"}
