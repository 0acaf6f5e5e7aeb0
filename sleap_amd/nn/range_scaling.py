"""Range-safe fp16 storage: per-tensor power-of-two activation scales folded into the weights.

The reference computes in float32; the device path stores activations as fp16 (11 mantissa bits, finite range
65504). A network whose activations leave that range -- a ResNet-50 without trained BatchNormalization statistics grows by a
factor of ~1.5 per residual block -- used to have bf16 storage as its only remedy, which does not meet north_star's 0.5 px
(8 mantissa bits). This module keeps fp16's mantissa AND the range:

    every stored activation tensor T gets a scale s_T = 2^k and the device stores s_T * T.

All layers of the graphs the engine runs (Conv2D / Conv2DTranspose + bias, ReLU, BatchNormalization at inference,
MaxPooling2D, UpSampling2D, Add, Concatenate) are positively homogeneous or affine, so the scales fold into the float32 weights
(exactly: multiplying a float by a power of two changes the exponent only) and the stored values are the true ones times a
power of two (rounding to fp16 commutes with such a scale as long as nothing becomes subnormal):

    Conv2D / Conv2DTranspose   kernel[.., cin-segment, ..] *= s_out / s_in(segment)      bias *= s_out
    BatchNormalization         gamma *= s_out / s_in      beta *= s_out      moving_mean *= s_in   (variance and epsilon stay:
                               gamma' (s x - s mean) / sqrt(var + eps) + beta' = s_out BN(x))
    ReLU / pool / upsample     s_out = s_in            Add: both operands and the sum share one scale
    Concatenate                the output keeps one scale per input segment (the consuming conv scales its kernel rows)
    model inputs and outputs   scale 1 (the heads un-scale: their outputs are the reference's float32 maps)

`plan_scales` chooses the exponents from measured ranges (max |activation| per layer of a calibration batch, measured on the
device with bf16 storage -- fp32's range -- by `DeviceNetwork.layer_ranges`); tensors that stay below `trigger` keep scale 1, so
a network that fits fp16 anyway (every UNet seen so far) is left bit for bit as it is. `fold_scales` returns the new weight
dictionary; the model_config is unchanged.
"""
import math
from typing import Dict, List, Optional, Tuple

import numpy as np

# groups whose largest calibration value exceeds TRIGGER are brought to <= LIMIT (64x headroom below 65504 for frames with larger
# activations than the calibration batch); the others keep scale 1 (>= 16x headroom)
LIMIT, TRIGGER = 1024.0, 4096.0

PRODUCERS = ("Conv2D", "Conv2DTranspose", "BatchNormalization")
PASS_THROUGH = ("Activation", "MaxPooling2D", "UpSampling2D", "ZeroPadding2D", "Lambda")


class _Groups:
    """union-find over layer names; a group can be pinned to scale 1."""

    def __init__(self):
        self.parent: Dict[str, str] = {}
        self.fixed = set()

    def find(self, a):
        self.parent.setdefault(a, a)
        while self.parent[a] != a:
            self.parent[a] = self.parent[self.parent[a]]
            a = self.parent[a]
        return a

    def tie(self, a, b):
        ra, rb = self.find(a), self.find(b)
        if ra != rb:
            self.parent[ra] = rb
            if ra in self.fixed:
                self.fixed.discard(ra)
                self.fixed.add(rb)

    def pin(self, a):
        self.fixed.add(self.find(a))

    def is_fixed(self, a):
        return self.find(a) in self.fixed


def _layers(model_config):
    cfg = model_config["config"]
    return cfg["layers"], [l[0] for l in cfg["output_layers"]]


def _segments(layers) -> Dict[str, List[Tuple[int, str]]]:
    """layer name -> [(channels, name of the layer whose scale the segment carries)] -- only Concatenate outputs have more
    than one segment. Channel counts come from the kernels' shapes where they are needed (fold_scales), so only the order and
    the carrier are recorded here; a channel count of -1 means "all"."""
    seg = {}
    for l in layers:
        name = l["name"]
        if l["class_name"] == "Concatenate":
            parts = []
            for n in l["inbound_nodes"][0]:
                parts.extend(seg[n[0]])
            seg[name] = parts
        else:
            seg[name] = [(-1, name)]
    return seg


def scale_groups(model_config, aliases: Optional[List[List[str]]] = None) -> _Groups:
    """Which layer outputs must share a scale. `aliases`: lists of layer names whose outputs the engine keeps in ONE stored
    tensor (a conv with its fused BatchNormalization / Add / ReLU epilogue): only the last value is ever stored, so they are
    tied (and measured) together."""
    layers, outputs = _layers(model_config)
    by_name = {l["name"]: l for l in layers}
    g = _Groups()
    seg = _segments(layers)
    for l in layers:
        cn, name = l["class_name"], l["name"]
        g.find(name)
        if cn == "InputLayer":
            g.pin(name)
            continue
        ins = [n[0] for n in l["inbound_nodes"][0]]
        if cn in PASS_THROUGH or cn == "Add":
            for i in ins:
                for _, carrier in seg[i]:  # a multi-segment input of a non-conv consumer: one scale for all of it
                    g.tie(carrier, name)
        elif cn == "BatchNormalization":
            if not l["config"].get("scale", True):  # no gamma to carry s_out / s_in
                for _, carrier in seg[ins[0]]:
                    g.tie(carrier, name)
            elif len(seg[ins[0]]) > 1:
                first = seg[ins[0]][0][1]
                for _, carrier in seg[ins[0]][1:]:
                    g.tie(carrier, first)
        elif cn == "Concatenate":
            pass
        elif cn not in PRODUCERS:
            raise NotImplementedError(f"range scaling: Keras layer {cn} ({name})")
    for a in aliases or []:
        a = [n for n in a if n in by_name]
        for n in a[1:]:
            g.tie(a[0], n)
    for o in outputs:
        for _, carrier in seg[o]:
            g.pin(carrier)
    return g


def plan_scales(model_config, ranges: Dict[str, float], aliases: Optional[List[List[str]]] = None, limit: float = LIMIT,
                trigger: float = TRIGGER) -> Dict[str, int]:
    """-> {layer name: k} with s = 2^k. `ranges`: max |activation| of the layers that were measured (unmeasured layers follow
    their group). Groups whose largest member stays below `trigger` keep k = 0; the others are brought to <= `limit`
    (LIMIT / TRIGGER above)."""
    layers, _ = _layers(model_config)
    g = scale_groups(model_config, aliases)
    top: Dict[str, float] = {}
    for name, m in ranges.items():
        if name not in g.parent:
            continue
        if not math.isfinite(m):
            raise FloatingPointError(f"range scaling: the measured range of {name} is not finite")
        r = g.find(name)
        top[r] = max(top.get(r, 0.0), float(m))
    k_of = {}
    for r, m in top.items():
        if r in g.fixed or m <= trigger:
            k_of[r] = 0
        else:
            k_of[r] = int(math.floor(math.log2(limit / m)))
    return {l["name"]: k_of.get(g.find(l["name"]), 0) for l in layers}


def fold_scales(model_config, weights: Dict[str, np.ndarray], log2_scale: Dict[str, int]) -> Dict[str, np.ndarray]:
    """-> the weights of the network that computes 2^k(layer) * (the original layer output) for every layer."""
    layers, _ = _layers(model_config)
    seg = _segments(layers)
    by_name = {l["name"]: l for l in layers}
    out = {k: np.array(v, dtype=np.float32, copy=True) for k, v in weights.items()}

    def chan_count(name):
        """channels of a single-segment layer output"""
        l = by_name[name]
        cn = l["class_name"]
        if cn == "InputLayer":
            return l["config"]["batch_input_shape"][-1]
        if cn == "Conv2D":
            return weights[f"{name}/kernel"].shape[3]
        if cn == "Conv2DTranspose":
            return weights[f"{name}/kernel"].shape[2]
        if cn == "BatchNormalization":
            return weights[f"{name}/moving_mean"].shape[0]
        if cn == "Lambda" and name == "tile_channels":
            return 3
        return chan_count(l["inbound_nodes"][0][0][0])

    def in_exponents(src_name, cin):
        """per input channel exponent of a consumer reading `src_name`"""
        parts = seg[src_name]
        if len(parts) == 1:
            return np.full((cin,), log2_scale.get(parts[0][1], 0), np.int64)
        e = np.concatenate([np.full((chan_count(c),), log2_scale.get(c, 0), np.int64) for _, c in parts])
        assert e.shape[0] == cin, (src_name, e.shape, cin)
        return e

    for l in layers:
        cn, name = l["class_name"], l["name"]
        if cn not in PRODUCERS:
            continue
        src = l["inbound_nodes"][0][0][0]
        ko = log2_scale.get(name, 0)
        if cn == "Conv2D":
            k = out[f"{name}/kernel"]
            e = ko - in_exponents(src, k.shape[2])
            if e.any():
                k *= np.exp2(e.astype(np.float32))[None, None, :, None]
            if ko and f"{name}/bias" in out:
                out[f"{name}/bias"] *= np.float32(2.0 ** ko)
        elif cn == "Conv2DTranspose":
            k = out[f"{name}/kernel"]  # (kh, kw, Cout, Cin)
            e = ko - in_exponents(src, k.shape[3])
            if e.any():
                k *= np.exp2(e.astype(np.float32))[None, None, None, :]
            if ko and f"{name}/bias" in out:
                out[f"{name}/bias"] *= np.float32(2.0 ** ko)
        else:
            n = out[f"{name}/moving_mean"].shape[0]
            ei = in_exponents(src, n)
            assert (ei == ei[0]).all(), f"{name}: BatchNormalization input with mixed scales"
            ki = int(ei[0])
            if ki:
                out[f"{name}/moving_mean"] *= np.float32(2.0 ** ki)
            if ko != ki:
                if f"{name}/gamma" not in out:
                    raise AssertionError(f"{name}: no gamma to carry the scale change")
                out[f"{name}/gamma"] *= np.float32(2.0 ** (ko - ki))
            if ko and f"{name}/beta" in out:
                out[f"{name}/beta"] *= np.float32(2.0 ** ko)
    return out


def dist_max(values: List[float], device=None) -> List[float]:
    """Element-wise MAX of a list of floats over the ranks of the default torch.distributed process group (the identity
    without one): the scan result of the range gate and the calibration ranges go through it, so that every rank of a
    frame-sharded run takes the same decision and folds the same exponents -- ranks must run numerically identical networks
    whatever frames their shards hold. inf and NaN (as inf) survive the reduction."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return list(values)
    on_gpu = dist.get_backend() == "nccl"
    t = torch.tensor(list(values), dtype=torch.float64, device=device if on_gpu else "cpu")
    t = torch.nan_to_num(t, nan=float("inf"), posinf=float("inf"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(x) for x in t.cpu()]


def dist_world() -> int:
    import torch.distributed as dist

    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def dist_agree(local_scan, already_scaled: bool, range_safe: bool, keys, measure, plan, device=None):
    """The range decision of a frame-sharded run, taken ONCE by ALL ranks at one program point (the predictor calls it after the
    first global batch; a rank whose shard was empty takes part with neutral values). Collectives inside a network's `forward`
    would deadlock: ranks see different buffer shapes (shard sizes differ, empty shards skip the network), so "first batch of a
    shape" is not a common event.

    local_scan  (largest finite value, inf / NaN seen) of this rank's first forward, or None (no frames)
    keys()      sorted layer names of the stored tensors (the same on every rank: a property of the plan, not of the data)
    measure()   {layer: max |activation|} on this rank's frames, or None (no frames)
    plan(r)     ranges -> exponents (plan_scales)

    -> the exponents every rank must install ({} / None: leave the network as it is). Raises FloatingPointError on EVERY rank when
    some rank overflowed and scaling is off / already applied / cannot help."""
    worst, nonfinite = local_scan if local_scan is not None else (0.0, False)
    red = dist_max([worst, 1.0 if nonfinite else 0.0, 1.0 if already_scaled else 0.0], device)
    need = red[1] > 0 or red[0] > 65504.0 / 4
    if not need:
        return None
    if range_safe and red[2] == 0:
        names = list(keys())
        mine = measure() if measure is not None else None
        vec = [float((mine or {}).get(k, 0.0)) for k in names]
        ranges = dict(zip(names, dist_max(vec, device)))
        ks = plan(ranges)
        if any(ks.values()):
            return ks
    if red[1] > 0:
        raise FloatingPointError(
            "activations left the range of fp16 storage (65504) on at least one rank and could not be rescaled: load the model "
            "with dtype='bf16' (or SLEAP_AMD_DTYPE=bf16), which has fp32's range, or calibrate_range() on representative frames")
    return None
