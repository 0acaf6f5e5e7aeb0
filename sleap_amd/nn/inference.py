"""Bottom-up inference with the reference's `sleap.nn.inference` surface, running on MI355X.

Mirrors (names, arguments, defaults, output dictionary keys, error behaviour):
    InferenceLayer            sleap/nn/inference.py:897-978
    InferenceModel            :981-1171   (predict / predict_on_batch; NaN-padded "unragged" outputs + n_valid)
    get_model_output_stride   :1174-1201
    find_head                 :1204-1226
    BottomUpInferenceLayer    :2737-3003
    BottomUpInferenceModel    :3006-3052
    Predictor                 :158-591    (from_model_paths, predict, _predict_generator)
    BottomUpPredictor         :3055-3348
    load_model                :4865-5004

The network runs through `engine.DeviceNetwork`, peaks and grouping through the HIP kernels behind
`ops`; a batch never leaves the device until the final, single D2H copy of the fixed-shape result.
With `torch.distributed` initialised, `BottomUpPredictor.predict` shards every global batch by
contiguous frame ranges and gathers the packed results with ONE all-gather (RCCL) per batch.
"""
import json
from collections import deque
import os
import time
import zipfile
import tempfile
from typing import Dict, Iterator, List, Optional, Union

import numpy as np
import torch

from .. import _lib, ops, parallel
from . import model_io
from .engine import DeviceNetwork
from .paf_grouping import GroupingOverflowError, PAFScorer


def find_head(model, name: str) -> Optional[int]:
    """inference.py:1204-1226 -- index of the first output whose name CONTAINS `name`."""
    for i, head_name in enumerate(model.output_names):
        if name in head_name:
            return i
    return None


def get_model_output_stride(model, input_ind: int = 0, output_ind: int = -1) -> int:
    """inference.py:1174-1201 -- input size / output size of the given output."""
    return int(model.output_strides()[output_ind])


def _as_device_images(data) -> torch.Tensor:
    if isinstance(data, dict):
        data = data["image"]
    if isinstance(data, torch.Tensor):
        t = data
    else:
        t = torch.from_numpy(np.ascontiguousarray(data))
    if t.ndim == 3:
        t = t[None]
    if t.ndim != 4:
        raise ValueError(f"images must have shape (samples, height, width, channels), got {tuple(t.shape)}")
    ops.require_cuda()
    if t.dtype not in (torch.uint8, torch.float32):
        t = t.to(torch.float32)
    return t.cuda(non_blocking=True).contiguous()


def start_download(outs):
    """Queue asynchronous copies of the CUDA tensors of a result dict (all but the carried input frames) into page-locked memory
    on the CURRENT stream -> (host tensors, event recorded behind them). With `finish_download` this is how a pipelined loop
    brings batch k to the host: the copies are queued right behind batch k's kernels, BEFORE batch k+1 is queued, and the host
    later waits for the event only. (`.cpu()` / a stream synchronisation at conversion time waits for everything queued on the
    stream by then -- batch k+1 included -- which turned the two-deep loop of the top-down predictor into a one-deep one:
    uploads and compute strictly in series, tools/timeline_gaps.py, tools/copy_issue_vs_run.py.)"""
    host = {k: torch.empty(v.shape, dtype=v.dtype, pin_memory=True) for k, v in outs.items()
            if isinstance(v, torch.Tensor) and v.is_cuda and k != "_input"}
    if not host:  # nothing on a device (CPU stand-ins in the multi-process tests)
        return host, None
    for k, h in host.items():
        h.copy_(outs[k], non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return host, ev


def finish_download(host, ev):
    """-> NumPy copies of the tensors `start_download` queued (waits for its event only)."""
    if ev is not None:
        ev.synchronize()
    return {k: h.numpy().copy() for k, h in host.items()}


def results_to_numpy(outs, host=None):
    """A result dict as NumPy: `host` (from `finish_download`) where given, a download now for what is still on a device."""
    got = host if host is not None else finish_download(*start_download(outs))
    return {k: (got[k] if k in got else (v.cpu().numpy() if isinstance(v, torch.Tensor) else v)) for k, v in outs.items()}


class InferenceLayer:
    """Wraps the network with the reference's preprocessing (inference.py:897-978).

    Attributes: keras_model (a `DeviceNetwork`), input_scale, pad_to_stride, ensure_grayscale
    (None = infer from the model's input channels), ensure_float.
    """

    def __init__(self, keras_model: DeviceNetwork, input_scale: float = 1.0, pad_to_stride: int = 1,
                 ensure_grayscale: Optional[bool] = None, ensure_float: bool = True, **kwargs):
        self.keras_model = keras_model
        self.input_scale = input_scale
        self.pad_to_stride = pad_to_stride
        if ensure_grayscale is None:
            ensure_grayscale = keras_model.in_channels == 1
        self.ensure_grayscale = ensure_grayscale
        self.ensure_float = ensure_float
        # Network / post-processing overlap across consecutive calls (new; the reference runs everything in one TF
        # graph): the network of call i+1 runs on a side HIP stream while the caller's stream still executes the
        # post-processing of call i. Model outputs are double buffered. `assume_inputs_ready`: a DEVICE input tensor
        # is normally ordered after everything already queued on the caller's stream (safe, but that includes the
        # previous call's post-processing); set it when the frames were produced long before (resident video).
        self.overlap_postproc = False
        self.assume_inputs_ready = False
        self._net_stream = None
        self._copy_stream = None
        self.last_upload_done = None  # event: the most recent host batch has been copied to the device
        self.last_upload = None       # that batch on the device
        self._slot = 0
        self._slot_free = [None, None]
        self._pending_slot = None

    def run_network(self, data):
        """preprocess + network forward -> list of model outputs, valid on the CURRENT stream."""
        if not self.overlap_postproc:
            return self.keras_model.forward(self.preprocess(data))
        cur = torch.cuda.current_stream()
        if self._net_stream is None:
            # High priority: the network is the critical path, and on ROCm it also decides which HARDWARE queue the stream
            # lives in. Normal-priority streams of a process share a few hardware queues round-robin; when the RCCL stream
            # of the result gather landed in the network stream's queue, the gather of step k (waiting for post-processing
            # k) sat in front of network k+1 and serialised the two: +0.45 ms per step, measured with tools/dist_probe.py
            # (7.45 -> 6.94 ms). High-priority streams get queues of their own.
            self._net_stream = torch.cuda.Stream(priority=-1)
        ns = self._net_stream
        self.release_outputs()  # outputs of a previous call that nobody released: everything queued so far may read them
        slot = self._slot
        self._slot ^= 1
        src = data["image"] if isinstance(data, dict) else data
        on_device = isinstance(src, torch.Tensor) and src.is_cuda
        if on_device and not self.assume_inputs_ready:
            ready = torch.cuda.Event()
            ready.record(cur)
            ns.wait_event(ready)
        if isinstance(src, torch.Tensor) and not src.is_cuda and src.is_pinned() and src.ndim == 4:
            # page-locked host batch (the FramePrefetcher's): upload on a copy stream of its own so that the DMA of batch
            # k+1 runs under the network of batch k instead of queueing behind it on the network stream
            if self._copy_stream is None:
                # high priority for the same reason as the network stream: a queue of its own -- in a shared hardware queue the
                # upload of batch k+2 can sit behind the post-processing of batch k+1, which waits for network k+1
                self._copy_stream = torch.cuda.Stream(priority=int(os.environ.get("SLEAP_AMD_COPY_STREAM_PRIORITY", "-1")))
            with torch.cuda.stream(self._copy_stream):
                dev = src.cuda(non_blocking=True)
                up = torch.cuda.Event()
                up.record(self._copy_stream)
            dev.record_stream(ns)
            ns.wait_event(up)
            data = dict(data, image=dev) if isinstance(data, dict) else dev
            self.last_upload_done = up
            self.last_upload = dev  # the frames on the device (a flow tracker reads them again; the caller drops the reference)
        if self._slot_free[slot] is not None:
            ns.wait_event(self._slot_free[slot])  # the consumer of this slot's previous outputs has finished
        with torch.cuda.stream(ns):
            imgs = self.preprocess(data)
            preds = self.keras_model.forward(imgs, slot=slot)
            done = torch.cuda.Event()
            done.record(ns)
        if on_device:
            src.record_stream(ns)
        cur.wait_event(done)
        self._pending_slot = slot
        return preds

    def release_outputs(self):
        """Call once the post-processing that reads the last `run_network` outputs has been queued."""
        if self._pending_slot is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self._slot_free[self._pending_slot] = ev
            self._pending_slot = None

    def preprocess(self, imgs, resize_img: bool = True) -> torch.Tensor:
        """inference.py:940-967: grayscale|rgb -> float -> resize -> pad (bottom/right zeros).

        uint8 batches that need no resize stay uint8: the `* 1/255` of ensure_float
        (normalization.py:49) is fused into the first convolution's load (sa_stem_conv3x3).
        """
        x = _as_device_images(imgs)
        if self.ensure_grayscale:
            x = _ensure_grayscale(x)
        else:
            x = _ensure_rgb(x)
        if resize_img and self.input_scale != 1.0:
            # (uint8 frames: ensure_float's `* 1/255` happens inside the resize kernel, tap by tap -- one launch, no torch arithmetic)
            x = _resize_image(x, self.input_scale, to_float=self.ensure_float)
        if self.pad_to_stride > 1:
            x = _pad_to_stride(x, self.pad_to_stride)
        return x

    def call(self, data):
        return self.keras_model.forward(self.preprocess(data))

    __call__ = call


def _ensure_grayscale(x):
    """normalization.py:81-96 (tf.image.rgb_to_grayscale; integer images round-trip through float)."""
    if x.shape[-1] == 1:
        return x
    w = torch.tensor([0.2989, 0.5870, 0.1140], dtype=torch.float32, device=x.device)
    if x.dtype == torch.uint8:
        g = ((x.to(torch.float32) * np.float32(1.0 / 255.0)) * w).sum(dim=-1, keepdim=True)
        return (g * np.float32(255.5)).to(torch.uint8).contiguous()
    return (x * w).sum(dim=-1, keepdim=True).contiguous()


def _ensure_rgb(x):
    """normalization.py:99-114."""
    if x.shape[-1] == 1:
        return x.expand(-1, -1, -1, 3).contiguous()
    return x


def _resize_image(x, scale, to_float: bool = False):
    """resizing.py:71-105: bilinear, half-pixel centres, no antialias; size = int(dim * scale). `to_float` with uint8 frames:
    ensure_float (normalization.py:49) applied to every tap inside the kernel -> float32 result, as converting first gives."""
    B, H, W, Cc = x.shape
    nh, nw = int(H * scale), int(W * scale)
    y = torch.empty((B, nh, nw, Cc), dtype=torch.float32, device=x.device)
    if x.dtype == torch.uint8:
        xc = x.contiguous()
        _lib.check(_lib.lib().sa_resize_bilinear_u8_f32(ops._ptr(xc), B, H, W, Cc, nh, nw,
                                                        float(np.float32(1.0 / 255.0)) if to_float else 1.0, ops._ptr(y),
                                                        ops._stream()), "sa_resize_bilinear_u8_f32")
        return y if to_float else y.to(torch.uint8)  # (tf.cast back to the input dtype truncates)
    xf = x.to(torch.float32).contiguous()
    _lib.check(_lib.lib().sa_resize_bilinear_f32(ops._ptr(xf), B, H, W, Cc, nh, nw, ops._ptr(y), ops._stream()),
               "sa_resize_bilinear_f32")
    return y


def _pad_to_stride(x, max_stride):
    """resizing.py:34-68."""
    B, H, W, Cc = x.shape
    ph = (max_stride - H % max_stride) % max_stride
    pw = (max_stride - W % max_stride) % max_stride
    if ph == 0 and pw == 0:
        return x
    out = torch.zeros((B, H + ph, W + pw, Cc), dtype=x.dtype, device=x.device)
    out[:, :H, :W] = x
    return out


class InferenceModel:
    """inference.py:981-1171 -- input handling and "unragging" shared by all inference models."""

    def call(self, example):  # overridden
        raise NotImplementedError

    def export_model(self, save_path: str, signatures: str = "serving_default", save_traces: bool = True,
                     model_name: Optional[str] = None, tensors: Optional[Dict[str, str]] = None, unrag_outputs: bool = True):
        """inference.py:1092-1171 (`InferenceModel.export_model`): see `Predictor.export_model` -- out of scope here."""
        raise NotImplementedError("out of scope: TF SavedModel / frozen-graph export (InferenceModel.export_model, "
                                  "sleap/nn/inference.py:1092); the model folder is the deployable artefact of sleap_amd")

    def __call__(self, example):
        return self.call(example)

    @staticmethod
    def _unrag(outs: Dict[str, torch.Tensor], numpy: bool):
        """unrag_example (data/utils.py:118-146): crop the instance axis to the batch's bounding shape,
        add `n_valid` (row lengths of the first ragged output, inference.py:1083-1088)."""
        if not numpy:
            return outs
        n_valid = outs["n_valid"].cpu().numpy()
        bound = int(n_valid.max()) if n_valid.size else 0
        res = {}
        for k, v in outs.items():
            if k in ("instance_peaks", "instance_peak_vals", "instance_scores"):
                res[k] = v[:, :bound].cpu().numpy()
            elif isinstance(v, torch.Tensor):
                res[k] = v.cpu().numpy()
            else:
                res[k] = v
        res["n_valid"] = n_valid.astype(np.int64)
        return res

    def _device_networks(self):
        """the DeviceNetwork(s) behind this model's layers (attribute `keras_model` of every layer attribute, as the reference)."""
        nets = []
        for v in vars(self).values():
            km = getattr(v, "keras_model", None)
            if km is not None and hasattr(km, "_range_gate"):
                nets.append(km)
        return nets

    # hard limits of the device kernels (csrc/postproc.hip: MAXNP; sa_find_local_peaks: max_peaks <= 16384)
    _HARD_MAX_PEAKS = 16384
    _HARD_MAX_NODE_PEAKS = 512
    _HARD_MAX_INSTANCES = 1024

    def call_checked(self, data):
        """`call` + inspection of the per-frame status words (one tiny D2H copy). The reference's ragged tensors
        have no capacity limits, so on an overflow of the fixed-shape device buffers the caps are doubled (they
        stay grown for later batches) and the batch is re-run; scipy's "infeasible" error is re-raised."""
        rescanned = False
        while True:
            outs = self.call(data)
            bits = 0
            for v in outs["status"].tolist():
                bits |= int(v)
            if bits & _lib.STATUS_NONFINITE:
                # a batch far outside the one the fp16 range scales were chosen on (or the first overflow of a shape whose
                # first batch fitted): let the networks scan this batch again -- the range gate re-calibrates with the larger
                # of the old and new ranges (DeviceNetwork._range_gate) -- and run it once more before giving up
                # (not while a network's range agreement is deferred to dist_agree_range(): its second scan of a pending overflow
                #  would raise "dist_agree_range() was not called" -- the wrong message for this caller)
                nets = [n for n in self._device_networks() if getattr(n, "dtype", None) == "fp16" and n.range_safe]
                if nets and not rescanned and not any(getattr(n, "_dist_defer", False) for n in nets):
                    rescanned = True
                    for n in nets:
                        n._range_checked = False
                    continue
                raise FloatingPointError(NONFINITE_MESSAGE)
            if bits & _lib.STATUS_LSA_INFEASIBLE:
                raise ValueError("cost matrix is infeasible")  # what scipy raises inside the reference
            over = bits & (_lib.STATUS_PEAK_OVERFLOW | _lib.STATUS_NODE_PEAK_OVERFLOW | _lib.STATUS_INSTANCE_OVERFLOW)
            if not over:
                return outs
            if not self._grow_caps(over):
                raise GroupingOverflowError(
                    f"a frame exceeds the hard capacity of the device buffers (status bits {bits}): "
                    f"> {self._HARD_MAX_PEAKS} peaks, > {self._HARD_MAX_NODE_PEAKS} peaks of one node type or "
                    f"> {self._HARD_MAX_INSTANCES} instances")

    def _grow_caps(self, over) -> bool:
        layer = getattr(self, "bottomup_layer", None)
        if layer is None:
            return False
        sc = layer.paf_scorer
        grown = False
        if over & _lib.STATUS_PEAK_OVERFLOW and layer.max_peaks < self._HARD_MAX_PEAKS:
            layer.max_peaks = min(layer.max_peaks * 2, self._HARD_MAX_PEAKS)
            grown = True
        if over & _lib.STATUS_NODE_PEAK_OVERFLOW and sc.max_node_peaks < self._HARD_MAX_NODE_PEAKS:
            sc.max_node_peaks = min(sc.max_node_peaks * 2, self._HARD_MAX_NODE_PEAKS)
            grown = True
        if over & _lib.STATUS_INSTANCE_OVERFLOW and sc.max_instances < self._HARD_MAX_INSTANCES:
            sc.max_instances = min(sc.max_instances * 2, self._HARD_MAX_INSTANCES)
            grown = True
        return grown

    def predict_on_batch(self, data, numpy: bool = False, **kwargs):
        """inference.py:1047-1090. numpy=False returns device tensors without synchronising (check
        `status` yourself or use `call_checked`)."""
        outs = self.call_checked(data) if numpy else self.call(data)
        return self._unrag(outs, numpy)

    def predict(self, data, numpy: bool = True, batch_size: int = 4, **kwargs):
        """inference.py:989-1045 -- `data`: array/tensor (samples, H, W, C) or dict with key "image"."""
        imgs = data["image"] if isinstance(data, dict) else data
        n = len(imgs)
        parts = []
        for i in range(0, n, batch_size):
            o = self.call_checked(imgs[i : i + batch_size])
            parts.append({k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in o.items()})
        imax = max(p["instance_scores"].shape[1] for p in parts)  # caps may have grown between batches
        for p in parts:
            for k in ("instance_peaks", "instance_peak_vals", "instance_scores"):
                v = p[k]
                if v.shape[1] < imax:
                    pad = torch.full((v.shape[0], imax - v.shape[1]) + tuple(v.shape[2:]), float("nan"),
                                     dtype=v.dtype, device=v.device)
                    p[k] = torch.cat([v, pad], dim=1)
        keys = ("instance_peaks", "instance_peak_vals", "instance_scores", "n_valid", "status")
        outs = {k: torch.cat([p[k] for p in parts], dim=0) for k in keys}
        return self._unrag(outs, numpy)


NONFINITE_MESSAGE = ("the network's confidence maps contain inf / NaN: activations left the range of the 16-bit storage type "
                     "(fp16: 65504). Load the model with dtype='bf16' (or SLEAP_AMD_DTYPE=bf16), which has fp32's range.")


class BottomUpInferenceLayer(InferenceLayer):
    """inference.py:2737-3003. Same constructor arguments and mutable attributes as the reference.

    New, device-buffer related: `max_peaks` (local peaks per frame held on the device).
    """

    def __init__(self, keras_model: DeviceNetwork, paf_scorer: PAFScorer, input_scale: float = 1.0,
                 pad_to_stride: int = 1, cm_output_stride: Optional[int] = None,
                 paf_output_stride: Optional[int] = None, peak_threshold: float = 0.2,
                 refinement: Optional[str] = "local", integral_patch_size: int = 5, return_confmaps: bool = False,
                 return_pafs: bool = False, return_paf_graph: bool = False, confmaps_ind: Optional[int] = None,
                 pafs_ind: Optional[int] = None, offsets_ind: Optional[int] = None, max_peaks: int = 512, **kwargs):
        super().__init__(keras_model=keras_model, input_scale=input_scale, pad_to_stride=pad_to_stride, **kwargs)
        self.paf_scorer = paf_scorer
        self.confmaps_ind = confmaps_ind
        self.pafs_ind = pafs_ind
        self.offsets_ind = offsets_ind
        if self.confmaps_ind is None:
            self.confmaps_ind = find_head(self.keras_model, "MultiInstanceConfmapsHead")
        if self.confmaps_ind is None:
            raise ValueError("Index of the confidence maps output tensor must be specified if not "
                             "named 'MultiInstanceConfmapsHead'.")
        if self.pafs_ind is None:
            self.pafs_ind = find_head(self.keras_model, "PartAffinityFieldsHead")
        if self.pafs_ind is None:
            raise ValueError("Index of the part affinity fields output tensor must be specified if "
                             "not named 'PartAffinityFieldsHead'.")
        if self.offsets_ind is None:
            self.offsets_ind = find_head(self.keras_model, "OffsetRefinementHead")
        if cm_output_stride is None:
            cm_output_stride = get_model_output_stride(self.keras_model, output_ind=self.confmaps_ind)
        self.cm_output_stride = cm_output_stride
        if paf_output_stride is None:
            paf_output_stride = get_model_output_stride(self.keras_model, output_ind=self.pafs_ind)
        self.paf_output_stride = paf_output_stride
        self.peak_threshold = peak_threshold
        self.refinement = refinement
        self.integral_patch_size = integral_patch_size
        self.return_confmaps = return_confmaps
        self.return_pafs = return_pafs
        self.return_paf_graph = return_paf_graph
        self.max_peaks = max_peaks
        self.overlap_postproc = True

    def forward_pass(self, data):
        """inference.py:2864-2890 -> (cms, pafs, offsets|None), float32 NHWC device tensors."""
        preds = self.run_network(data)
        offsets = preds[self.offsets_ind] if self.offsets_ind is not None else None
        return preds[self.confmaps_ind], preds[self.pafs_ind], offsets

    def find_peaks(self, cms, offsets, status=None):
        """inference.py:2892-2936 in padded form: (peak_xy [B,P,2] image px, peak_val, peak_chan, peak_count, status).

        `peaks * cm_output_stride` (:2921) is applied inside the kernel as a separate fp32 multiply."""
        refinement = self.refinement if self.refinement in ("integral", "local") else None
        return ops.find_local_peaks(cms, offsets, self.peak_threshold, refinement, self.integral_patch_size,
                                    float(self.cm_output_stride), self.max_peaks, status)

    def call(self, data):
        """inference.py:2938-3003. Returns fixed-shape device tensors:
        instance_peaks (B, I, N, 2), instance_peak_vals (B, I, N), instance_scores (B, I) NaN-padded along I,
        n_valid (B,), status (B,) [+ confmaps, part_affinity_fields, paf graph tables if requested]."""
        cms, pafs, offsets = self.forward_pass(data)
        refinement = self.refinement if self.refinement in ("integral", "local") else None
        r = self.paf_scorer.predict_from_maps(cms, offsets, pafs, self.peak_threshold, refinement, self.integral_patch_size,
                                              self.cm_output_stride, self.max_peaks)
        inst, vals, scores, n_inst, status = (r["instance_peaks"], r["instance_peak_vals"], r["instance_scores"],
                                              r["n_instances"], r["status"])
        if self.input_scale != 1.0:
            inst = (inst / np.float32(self.input_scale)) + np.float32(0.5)  # :2980-2984
        out = {"instance_peaks": inst, "instance_peak_vals": vals, "instance_scores": scores, "n_valid": n_inst,
               "status": status}
        keep = (lambda t: t.clone()) if self.overlap_postproc else (lambda t: t)  # slot buffers get recycled
        if self.return_confmaps:
            out["confmaps"] = keep(cms)
        if self.return_pafs:
            out["part_affinity_fields"] = keep(pafs)
        if self.return_paf_graph:
            out["peaks"], out["peak_vals"], out["peak_channel_inds"], out["peak_count"] = (r["peak_xy"], r["peak_val"],
                                                                                            r["peak_chan"], r["peak_count"])
            out["paf_node_count"], out["paf_node_peaks"], out["line_scores"] = r["node_count"], r["node_peaks"], r["line_scores"]
        self.release_outputs()
        return out

    __call__ = call


class BottomUpInferenceModel(InferenceModel):
    """inference.py:3006-3052."""

    def __init__(self, bottomup_layer: BottomUpInferenceLayer, **kwargs):
        self.bottomup_layer = bottomup_layer

    def call(self, example):
        if isinstance(example, dict):
            example = example["image"]
        return self.bottomup_layer(example)


# ----------------------------------------------------------------------------------------------------
# Single-instance and top-down layers
# ----------------------------------------------------------------------------------------------------
def _heads(net, confmaps_name, confmaps_ind, offsets_ind):
    if confmaps_ind is None:
        confmaps_ind = find_head(net, confmaps_name)
    if confmaps_ind is None:
        raise ValueError(f"Index of the confidence maps output tensor must be specified if not named '{confmaps_name}'.")
    if offsets_ind is None:
        offsets_ind = find_head(net, "OffsetRefinementHead")
    return confmaps_ind, offsets_ind


def _pad_ragged(flat, sample_inds, n_samples, fill=float("nan")):
    """flat (n, ...) + sample index per row (sample-major order) -> (B, Imax, ...) padded tensor, counts (B,)."""
    counts = torch.bincount(sample_inds.to(torch.int64), minlength=n_samples)
    imax = int(counts.max().item()) if flat.shape[0] else 0
    out = torch.full((n_samples, imax) + tuple(flat.shape[1:]), fill, dtype=flat.dtype, device=flat.device)
    if flat.shape[0]:
        start = torch.cumsum(counts, 0) - counts
        pos = torch.arange(flat.shape[0], device=flat.device) - start[sample_inds.to(torch.int64)]
        out[sample_inds.to(torch.int64), pos] = flat
    return out, counts.to(torch.int32)


class SingleInstanceInferenceLayer(InferenceLayer):
    """inference.py:1229-1380: network -> global peak per channel -> (B, 1, N, 2)."""

    def __init__(self, keras_model, input_scale: float = 1.0, pad_to_stride: int = 1, output_stride: Optional[int] = None,
                 peak_threshold: float = 0.2, refinement: Optional[str] = "local", integral_patch_size: int = 5,
                 return_confmaps: bool = False, confmaps_ind: Optional[int] = None, offsets_ind: Optional[int] = None,
                 **kwargs):
        super().__init__(keras_model=keras_model, input_scale=input_scale, pad_to_stride=pad_to_stride, **kwargs)
        self.confmaps_ind, self.offsets_ind = _heads(keras_model, "SingleInstanceConfmapsHead", confmaps_ind, offsets_ind)
        if output_stride is None:
            output_stride = get_model_output_stride(keras_model, output_ind=self.confmaps_ind)
        self.output_stride = output_stride
        self.peak_threshold = peak_threshold
        self.refinement = refinement
        self.integral_patch_size = integral_patch_size
        self.return_confmaps = return_confmaps

    def call(self, data):
        imgs = self.preprocess(data)
        preds = self.keras_model.forward(imgs)
        cms = preds[self.confmaps_ind]
        offsets = preds[self.offsets_ind] if self.offsets_ind is not None else None
        refinement = self.refinement if self.refinement in ("integral", "local") else None
        peaks, vals = ops.find_global_peaks(cms, offsets, self.peak_threshold, refinement, self.integral_patch_size,
                                            float(self.output_stride))
        if self.input_scale != 1.0:
            peaks = (peaks / np.float32(self.input_scale)) + np.float32(0.5)
        out = {"instance_peaks": peaks[:, None], "instance_peak_vals": vals[:, None]}
        if self.return_confmaps:
            out["confmaps"] = cms
        return out

    __call__ = call


class SingleInstanceInferenceModel(InferenceModel):
    """inference.py:1383-1412."""

    def __init__(self, single_instance_layer, **kwargs):
        self.single_instance_layer = single_instance_layer

    def call(self, example):
        return self.single_instance_layer(example)

    def call_checked(self, data):
        return self.call(data)

    def predict(self, data, numpy: bool = True, batch_size: int = 4, **kwargs):
        imgs = data["image"] if isinstance(data, dict) else data
        parts = [self.call(imgs[i : i + batch_size]) for i in range(0, len(imgs), batch_size)]
        outs = {k: torch.cat([p[k].clone() for p in parts], dim=0) for k in parts[0]}
        return {k: v.cpu().numpy() for k, v in outs.items()} if numpy else outs

    @staticmethod
    def outputs_to_numpy(outs, host=None):
        """`host`: what `finish_download` returned for these outputs (a pipelined loop started the copies earlier)."""
        return results_to_numpy(outs, host)

    def predict_on_batch(self, data, numpy: bool = False, **kwargs):
        outs = self.call(data)
        return self.outputs_to_numpy(outs) if numpy else outs


class CentroidCrop(InferenceLayer):
    """inference.py:1638-1966: centroid network -> local peaks -> crops of the UN-preprocessed full image."""

    def __init__(self, keras_model, crop_size: int, input_scale: float = 1.0, pad_to_stride: int = 1,
                 output_stride: Optional[int] = None, peak_threshold: float = 0.2, refinement: Optional[str] = "local",
                 integral_patch_size: int = 5, return_confmaps: bool = False, confmaps_ind: Optional[int] = None,
                 offsets_ind: Optional[int] = None, return_crops: bool = True, max_instances: Optional[int] = None,
                 precrop_resize: float = 1.0, max_peaks: int = 512, **kwargs):
        super().__init__(keras_model=keras_model, input_scale=input_scale, pad_to_stride=pad_to_stride, **kwargs)
        self.crop_size = crop_size
        self.confmaps_ind, self.offsets_ind = _heads(keras_model, "CentroidConfmapsHead", confmaps_ind, offsets_ind)
        if output_stride is None:
            output_stride = get_model_output_stride(keras_model, output_ind=self.confmaps_ind)
        self.output_stride = output_stride
        self.peak_threshold = peak_threshold
        self.refinement = refinement
        self.integral_patch_size = integral_patch_size
        self.return_confmaps = return_confmaps
        self.return_crops = return_crops
        self.max_instances = max_instances
        self.precrop_resize = precrop_resize
        self.max_peaks = max_peaks

    # crop slots per frame of the device path when max_instances is None: grows on overflow (status bit, like the bottom-up
    # capacities) and follows the largest count seen so far (`observe`)
    max_crops = 8

    def _slots(self) -> int:
        return max(1, int(self.max_instances)) if self.max_instances is not None else max(1, int(self.max_crops))

    def observe(self, n_valid_max: int):
        """Host feedback once a batch's results are on the host: with max_instances None the slot count follows the largest
        number of centroids seen in a frame so far (a constant number of animals is the common case), so later batches do
        not run the instance network on empty slots."""
        if self.max_instances is None and n_valid_max > 0:
            self._seen = max(getattr(self, "_seen", 0), int(n_valid_max))
            self.max_crops = max(self._seen, 1)

    def call_padded(self, inputs):
        """The device form of `call` (inference.py:1747-1966): K fixed crop slots per frame, nothing is brought to the host.
        -> centroids (B, K, 2) NaN padded, centroid_vals (B, K), n_valid (B,), crops (B*K, crop, crop, C) (zeros in empty
        slots), crop_offsets (B*K, 2), status (B,) [SA_STATUS_INSTANCE_OVERFLOW: more centroids than slots], samples, K."""
        full_imgs = _as_device_images(inputs)
        imgs = self.preprocess(full_imgs)
        out = self.keras_model.forward(imgs)
        cms = out[self.confmaps_ind]
        offsets = out[self.offsets_ind] if self.offsets_ind is not None else None
        B = cms.shape[0]
        refinement = self.refinement if self.refinement in ("integral", "local") else None
        pxy, pval, pch, pcnt, status = ops.find_local_peaks(cms, offsets, self.peak_threshold, refinement,
                                                            self.integral_patch_size, float(self.output_stride), self.max_peaks)
        K = self._slots()
        dev = cms.device
        cent = torch.empty((B, K, 2), dtype=torch.float32, device=dev)
        cval = torch.empty((B, K), dtype=torch.float32, device=dev)
        centres = torch.empty((B, K, 2), dtype=torch.float32, device=dev)
        coff = torch.empty((B, K, 2), dtype=torch.float32, device=dev)
        n_valid = torch.empty((B,), dtype=torch.int32, device=dev)
        _lib.check(_lib.lib().sa_select_centroids(
            ops._ptr(pxy), ops._ptr(pval), ops._ptr(pcnt), B, pxy.shape[1], K, -1 if self.max_instances is None else int(self.max_instances),
            float(self.input_scale), float(self.precrop_resize), int(self.crop_size), ops._ptr(cent), ops._ptr(cval), ops._ptr(centres),
            ops._ptr(coff), ops._ptr(n_valid), ops._ptr(status), ops._stream()), "sa_select_centroids")
        if self.precrop_resize != 1.0:
            full_imgs = _resize_image(full_imgs, self.precrop_resize)
        sample_inds = torch.arange(B, device=dev, dtype=torch.int32).repeat_interleave(K)
        crops = ops.crop_and_resize(full_imgs, centres.reshape(B * K, 2), sample_inds, self.crop_size)
        outputs = dict(centroids=cent, centroid_vals=cval, n_valid=n_valid, status=status, samples=B, K=K, padded=True,
                       crops=crops, crop_offsets=coff.reshape(B * K, 2))
        if self.return_confmaps:
            outputs["centroid_confmaps"] = cms
        return outputs

    def call(self, inputs):
        """inference.py:1747-1966 with the reference's FLAT (ragged) outputs: centroids (n, 2), centroid_vals (n,),
        crop_sample_inds (n,), crops (n, crop, crop, C), crop_offsets (n, 2). Built from `call_padded` -- sizing a ragged
        result needs the counts on the host (one small copy); `TopDownInferenceModel` uses the padded form directly."""
        while True:
            o = self.call_padded(inputs)
            bits = int(np.bitwise_or.reduce(o["status"].cpu().numpy())) if o["samples"] else 0
            if bits & _lib.STATUS_NONFINITE:
                raise FloatingPointError(NONFINITE_MESSAGE)
            if bits & _lib.STATUS_PEAK_OVERFLOW and self.max_peaks < 16384:
                self.max_peaks *= 2
                continue
            if bits & _lib.STATUS_INSTANCE_OVERFLOW and self.max_instances is None and self.max_crops < 1024:
                self.max_crops *= 2
                continue
            break
        B, K = o["samples"], o["K"]
        nv = o["n_valid"].cpu().numpy()
        keep = torch.from_numpy((np.arange(K)[None, :] < nv[:, None]).reshape(-1)).to(o["centroids"].device)
        sample_inds = torch.arange(B, device=keep.device, dtype=torch.int32).repeat_interleave(K)[keep]
        outputs = dict(centroids=o["centroids"].reshape(B * K, 2)[keep], centroid_vals=o["centroid_vals"].reshape(B * K)[keep],
                       crop_sample_inds=sample_inds, samples=B)
        if self.return_confmaps:
            outputs["centroid_confmaps"] = o["centroid_confmaps"]
        if self.return_crops:
            outputs["crops"] = o["crops"][keep]
            outputs["crop_offsets"] = o["crop_offsets"][keep]
        return outputs

    __call__ = call


class FindInstancePeaks(InferenceLayer):
    """inference.py:1969-2200: centered-instance network on crops -> global peak per channel, shifted by the crop
    offsets back into full-image coordinates."""

    def __init__(self, keras_model, input_scale: float = 1.0, output_stride: Optional[int] = None,
                 peak_threshold: float = 0.2, refinement: Optional[str] = "local", integral_patch_size: int = 5,
                 return_confmaps: bool = False, confmaps_ind: Optional[int] = None, offsets_ind: Optional[int] = None,
                 resize_input_image: bool = True, **kwargs):
        super().__init__(keras_model=keras_model, input_scale=input_scale, pad_to_stride=1, **kwargs)
        self.confmaps_ind, self.offsets_ind = _heads(keras_model, "CenteredInstanceConfmapsHead", confmaps_ind, offsets_ind)
        if output_stride is None:
            output_stride = get_model_output_stride(keras_model, output_ind=self.confmaps_ind)
        self.output_stride = output_stride
        self.peak_threshold = peak_threshold
        self.refinement = refinement
        self.integral_patch_size = integral_patch_size
        self.return_confmaps = return_confmaps
        self.resize_input_image = resize_input_image

    def call_padded(self, inputs):
        """`call` on the fixed-slot output of `CentroidCrop.call_padded`: the instance network runs on all B*K crop slots,
        empty slots come out as NaN rows; nothing is brought to the host. -> instance_peaks (B, K, N, 2), instance_peak_vals
        (B, K, N), centroids (B, K, 2), centroid_vals (B, K), n_valid (B,), status (B,)."""
        crops, B, K = inputs["crops"], inputs["samples"], inputs["K"]
        x = self.preprocess(crops, resize_img=self.resize_input_image)
        out = self.keras_model.forward(x)
        cms = out[self.confmaps_ind]
        offsets = out[self.offsets_ind] if self.offsets_ind is not None else None
        refinement = self.refinement if self.refinement in ("integral", "local") else None
        peak_points, peak_vals = ops.find_global_peaks(cms, offsets, self.peak_threshold, refinement, self.integral_patch_size,
                                                       float(self.output_stride))
        N = peak_vals.shape[1]
        _lib.check(_lib.lib().sa_finish_instance_peaks(ops._ptr(peak_points), ops._ptr(peak_vals), ops._ptr(inputs["crop_offsets"]),
                                                       ops._ptr(inputs["n_valid"]), B, K, N, float(self.input_scale),
                                                       ops._stream()), "sa_finish_instance_peaks")
        outputs = {"instance_peaks": peak_points.reshape(B, K, N, 2), "instance_peak_vals": peak_vals.reshape(B, K, N),
                   "n_valid": inputs["n_valid"], "centroids": inputs["centroids"], "centroid_vals": inputs["centroid_vals"],
                   "status": inputs["status"]}
        if "centroid_confmaps" in inputs:
            outputs["centroid_confmaps"] = inputs["centroid_confmaps"]
        if self.return_confmaps:
            outputs["instance_confmaps"] = cms
        return outputs

    def call(self, inputs):
        if isinstance(inputs, dict) and inputs.get("padded"):
            return self.call_padded(inputs)
        if isinstance(inputs, dict):
            crops = inputs["crops"]
        else:
            crops, inputs = inputs, {}
        crops = _as_device_images(crops) if not isinstance(crops, torch.Tensor) or not crops.is_cuda else crops
        n = crops.shape[0]
        if "crop_sample_inds" in inputs:
            samples, crop_sample_inds = inputs["samples"], inputs["crop_sample_inds"]
        else:
            samples, crop_sample_inds = n, torch.arange(n, dtype=torch.int32, device=crops.device)
        n_nodes = self.keras_model.outputs[self.confmaps_ind].c
        if n == 0:
            peak_points = torch.zeros((0, n_nodes, 2), dtype=torch.float32, device=crops.device)
            peak_vals = torch.zeros((0, n_nodes), dtype=torch.float32, device=crops.device)
        else:
            x = self.preprocess(crops, resize_img=self.resize_input_image)
            out = self.keras_model.forward(x)
            cms = out[self.confmaps_ind]
            offsets = out[self.offsets_ind] if self.offsets_ind is not None else None
            refinement = self.refinement if self.refinement in ("integral", "local") else None
            peak_points, peak_vals = ops.find_global_peaks(cms, offsets, self.peak_threshold, refinement,
                                                           self.integral_patch_size, float(self.output_stride))
            if self.input_scale != 1.0:
                peak_points = (peak_points / np.float32(self.input_scale)) + np.float32(0.5)
            if "crop_offsets" in inputs:
                peak_points = peak_points + (inputs["crop_offsets"][:, None, :] / np.float32(self.input_scale))
        peaks, counts = _pad_ragged(peak_points, crop_sample_inds, samples)
        vals, _ = _pad_ragged(peak_vals, crop_sample_inds, samples)
        outputs = {"instance_peaks": peaks, "instance_peak_vals": vals, "n_valid": counts}
        if "centroids" in inputs:
            outputs["centroids"], _ = _pad_ragged(inputs["centroids"], crop_sample_inds, samples)
            outputs["centroid_vals"], _ = _pad_ragged(inputs["centroid_vals"], crop_sample_inds, samples)
        if "centroid_confmaps" in inputs:
            outputs["centroid_confmaps"] = inputs["centroid_confmaps"]
        return outputs

    __call__ = call


class TopDownInferenceModel(InferenceModel):
    """inference.py:2246-2311."""

    def __init__(self, centroid_crop: CentroidCrop, instance_peaks: FindInstancePeaks, **kwargs):
        self.centroid_crop = centroid_crop
        self.instance_peaks = instance_peaks

    def call(self, example):
        """inference.py:2279-2311. Runs entirely on the device (fixed crop slots per frame, `CentroidCrop.call_padded`):
        nothing synchronises; `status` carries the overflow / non-finite flags, which `outputs_to_numpy` (or `call_checked`)
        acts on once the results are on the host anyway."""
        if isinstance(example, dict):
            example = example["image"]
        outs = self.instance_peaks.call_padded(self.centroid_crop.call_padded(example))
        outs["_input"] = example  # what to re-run if the deferred status words report an overflow
        return outs

    def call_checked(self, data):
        """`call` + inspection of the status words: capacities are doubled and the batch re-run on overflow."""
        while True:
            outs = self.call(data)
            if not self._grow_on_status(int(np.bitwise_or.reduce(outs["status"].cpu().numpy())) if len(outs["status"]) else 0):
                return outs

    def _grow_on_status(self, bits) -> bool:
        cc = self.centroid_crop
        if bits & _lib.STATUS_NONFINITE:
            raise FloatingPointError(NONFINITE_MESSAGE)
        grown = False
        if bits & _lib.STATUS_PEAK_OVERFLOW and cc.max_peaks < 16384:
            cc.max_peaks *= 2
            grown = True
        if bits & _lib.STATUS_INSTANCE_OVERFLOW and cc.max_instances is None and cc.max_crops < 1024:
            cc.max_crops *= 2
            grown = True
        return grown

    @staticmethod
    def _to_numpy(outs, host=None):
        """Device tensors -> NumPy (asynchronous copies into page-locked memory, one event wait); `host`: already downloaded."""
        return results_to_numpy(outs, host)

    def outputs_to_numpy(self, outs, host=None):
        """Device results -> NumPy in the reference's unragged form (instance axis cropped to the batch's bounding shape,
        data/utils.py:118-146). This is where the deferred status words are looked at: an overflowed batch is re-run (with
        grown capacities) before anything is returned. `host`: `finish_download` of these outputs (pipelined loops)."""
        while True:
            src = outs.pop("_input", None)
            res = self._to_numpy(outs, host)
            host = None  # (a re-run's outputs are downloaded here)
            status = res.pop("status", None)
            bits = int(np.bitwise_or.reduce(status)) if status is not None and len(status) else 0
            if not self._grow_on_status(bits) or src is None:
                break
            outs = self.call(src)
        res["n_valid"] = res["n_valid"].astype(np.int64)
        bound = int(res["n_valid"].max()) if len(res["n_valid"]) else 0
        self.centroid_crop.observe(bound)
        for k in ("instance_peaks", "instance_peak_vals", "centroids", "centroid_vals"):
            if k in res:
                res[k] = np.ascontiguousarray(res[k][:, :bound])
        return res

    def predict_on_batch(self, data, numpy: bool = False, **kwargs):
        outs = self.call(data)
        return self.outputs_to_numpy(outs) if numpy else outs

    def predict(self, data, numpy: bool = True, batch_size: int = 4, **kwargs):
        """inference.py:989-1045: all batches, concatenated (instance axis padded to the widest batch)."""
        imgs = data["image"] if isinstance(data, dict) else data
        parts = [self.call_checked(imgs[i : i + batch_size]) for i in range(0, len(imgs), batch_size)]
        parts = [{k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in p.items() if k != "_input"} for p in parts]
        imax = max(p["instance_peaks"].shape[1] for p in parts)
        keys = [k for k in ("instance_peaks", "instance_peak_vals", "centroids", "centroid_vals") if k in parts[0]]
        outs = {}
        for k in keys:
            padded = []
            for p in parts:
                v = p[k]
                if v.shape[1] < imax:
                    pad = torch.full((v.shape[0], imax - v.shape[1]) + tuple(v.shape[2:]), float("nan"), dtype=v.dtype,
                                     device=v.device)
                    v = torch.cat([v, pad], dim=1)
                padded.append(v)
            outs[k] = torch.cat(padded, dim=0)
        outs["n_valid"] = torch.cat([p["n_valid"] for p in parts], dim=0)
        if not numpy:
            return outs
        res = self._to_numpy(outs)
        res["n_valid"] = res["n_valid"].astype(np.int64)
        bound = int(res["n_valid"].max()) if len(res["n_valid"]) else 0
        for k in keys:
            res[k] = np.ascontiguousarray(res[k][:, :bound])
        return res


# ----------------------------------------------------------------------------------------------------
# Data pipeline description and progress reporting of the predictors
# ----------------------------------------------------------------------------------------------------
class Pipeline:
    """What `Predictor.make_pipeline` returns (inference.py:329-371; sleap/nn/data/pipelines.py `Pipeline`): the provider and
    the transformer chain of the reference's inference pipeline -- [SizeMatcher,] Normalizer(ensure_float=False,
    ensure_grayscale / ensure_rgb), Batcher(batch_size, drop_remainder=False, unrag=False), Prefetcher -- as a description
    plus `make_dataset()`, a generator of batched examples (`image (b,H,W,C)`, `raw_image_size`, `video_ind`, `frame_ind`,
    `scale`). The predict loop does not pull frames through this generator (it reads this rank's shard of every batch ahead
    into page-locked buffers, io/video.py: FramePrefetcher); channel conversion happens on the device in
    `InferenceLayer.preprocess`."""

    def __init__(self, providers=None, transformers=None, batch_size: int = 4):
        self.providers = list(providers or [])
        self.transformers = list(transformers or [])
        self.batch_size = int(batch_size)

    def __iadd__(self, other):
        self.transformers.append(other)
        return self

    def __len__(self):
        return len(self.providers[0]) if self.providers else 0

    @property
    def output_keys(self) -> List[str]:
        return ["image", "raw_image_size", "video_ind", "frame_ind", "scale"]

    def make_dataset(self) -> Iterator[Dict[str, np.ndarray]]:
        if not self.providers:
            raise ValueError("Pipeline has no provider")
        batch = []
        for ex in self.providers[0].make_dataset():
            batch.append(ex)
            if len(batch) == self.batch_size:
                yield {k: np.stack([np.asarray(e[k]) for e in batch]) for k in batch[0]}
                batch = []
        if batch:  # drop_remainder=False
            yield {k: np.stack([np.asarray(e[k]) for e in batch]) for k in batch[0]}


class ProgressReporter:
    """Progress reporting of `Predictor._predict_generator` (inference.py:421-491): `verbosity` "rich" (progress bar with
    percentage, ETA and FPS columns, refreshed manually every `report_period` seconds so that notebooks work), "json" (one
    line `{"n_processed", "n_total", "elapsed", "rate", "eta"}` per report period, rate over the last 30 batches) or "none".
    Rank 0 reports; use as a context manager and call `update(n_frames_in_batch)` per batch."""

    def __init__(self, verbosity: str, report_rate: float, n_total: int, enabled: bool = True):
        if verbosity not in ("none", "rich", "json"):
            raise ValueError(f"'verbosity' must be in ['none', 'rich', 'json'] (got {verbosity!r})")
        self.verbosity = verbosity if enabled else "none"
        self.report_period = 1.0 / max(float(report_rate), 1e-6)
        self.n_total = int(n_total)
        self._progress = self._task = None

    def __enter__(self):
        now = time.time()
        self._last_report = self._t0_all = self._t0_batch = now
        self.n_processed = 0
        self._n_recent, self._elapsed_recent = deque(maxlen=30), deque(maxlen=30)
        if self.verbosity == "rich":
            try:
                import rich.progress as rp
            except ImportError:  # the reference hard-depends on rich; without it the bar degrades to silence
                self.verbosity = "none"
                return self

            class RateColumn(rp.ProgressColumn):  # inference.py:145-153
                def render(self, task):
                    if task.speed is None:
                        return rp.Text("?", style="progress.data.speed")
                    return rp.Text(f"{task.speed:.1f} FPS", style="progress.data.speed")

            self._progress = rp.Progress("{task.description}", rp.BarColumn(), "[progress.percentage]{task.percentage:>3.0f}%",
                                         "ETA:", rp.TimeRemainingColumn(), RateColumn(), auto_refresh=False,
                                         refresh_per_second=1.0 / self.report_period, speed_estimate_period=5)
            self._progress.__enter__()
            self._task = self._progress.add_task("Predicting...", total=self.n_total)
        return self

    def update(self, n_batch: int):
        now = time.time()
        self.n_processed += int(n_batch)
        if self.verbosity == "rich":
            self._progress.update(self._task, advance=int(n_batch))
            if now - self._last_report > self.report_period:
                self._progress.refresh()
                self._last_report = now
        elif self.verbosity == "json":
            self._n_recent.append(int(n_batch))
            self._elapsed_recent.append(now - self._t0_batch)
            self._t0_batch = now
            rate = sum(self._n_recent) / max(sum(self._elapsed_recent), 1e-9)
            if now - self._last_report > self.report_period:
                print(json.dumps({"n_processed": self.n_processed, "n_total": self.n_total, "elapsed": now - self._t0_all,
                                  "rate": rate, "eta": (self.n_total - self.n_processed) / max(rate, 1e-9)}), flush=True)
                self._last_report = now

    def __exit__(self, *exc):
        if self._progress is not None:
            self._progress.refresh()
            self._progress.__exit__(*exc)
            self._progress = None
        return False


# ----------------------------------------------------------------------------------------------------
# Predictors
# ----------------------------------------------------------------------------------------------------
class Predictor:
    """inference.py:158-591 (the parts on the bottom-up path)."""

    verbosity = "rich"  # inference.py:162-166: one of "none", "rich", "json"
    report_rate = 2.0
    model_paths: List[str] = []
    pipeline = None

    @property
    def report_period(self) -> float:
        """Time between progress reports in seconds (inference.py:170-173)."""
        return 1.0 / self.report_rate

    def make_pipeline(self, data_provider=None) -> Pipeline:
        """inference.py:329-371: the data pipeline for `data_provider` (a `VideoReader`, `Video` or array); also stored as
        `self.pipeline`, and rebuilt automatically when predicting from a new source."""
        from ..io.video import Video, VideoReader

        if isinstance(data_provider, np.ndarray):
            data_provider = Video.from_numpy(data_provider)
        if isinstance(data_provider, Video):
            data_provider = VideoReader(data_provider)
        pipeline = Pipeline(providers=[data_provider] if data_provider is not None else [], batch_size=self.batch_size)
        try:
            pre = (self.data_config or {}).get("preprocessing", {})
        except AttributeError:
            pre = {}
        if pre.get("resize_and_pad_to_target"):
            pipeline += ("SizeMatcher", {"max_image_height": pre.get("target_height"), "max_image_width": pre.get("target_width")})
        try:
            gray = bool(self.is_grayscale)
        except AttributeError:  # a model object that does not describe its input (stand-ins): follow the source
            gray = data_provider is None or data_provider.video.channels == 1
        pipeline += ("Normalizer", {"ensure_float": False, "ensure_grayscale": gray, "ensure_rgb": not gray})
        pipeline += ("Batcher", {"batch_size": self.batch_size, "drop_remainder": False, "unrag": False})
        pipeline += ("Prefetcher", {})
        self.pipeline = pipeline
        return pipeline

    @classmethod
    def from_model_paths(cls, model_paths, peak_threshold: float = 0.2, integral_refinement: bool = True,
                         integral_patch_size: int = 5, batch_size: int = 4, resize_input_layer: bool = True,
                         max_instances: Optional[int] = None) -> "Predictor":
        """inference.py:176-311: dispatch on `model.heads` (only `multi_instance` is implemented)."""
        if isinstance(model_paths, str):
            model_paths = [model_paths]
        model_configs = [model_io.load_training_config(p) for p in model_paths]
        model_paths = [model_io.model_dir(p) for p in model_paths]
        model_types = [model_io.head_type(c) for c in model_configs]
        if "single_instance" in model_types:
            predictor = SingleInstancePredictor.from_trained_models(
                model_paths[model_types.index("single_instance")], peak_threshold=peak_threshold,
                integral_refinement=integral_refinement, integral_patch_size=integral_patch_size,
                batch_size=batch_size, resize_input_layer=resize_input_layer)
        elif "centroid" in model_types and "centered_instance" in model_types:
            predictor = TopDownPredictor.from_trained_models(
                centroid_model_path=model_paths[model_types.index("centroid")],
                confmap_model_path=model_paths[model_types.index("centered_instance")],
                peak_threshold=peak_threshold, integral_refinement=integral_refinement,
                integral_patch_size=integral_patch_size, batch_size=batch_size,
                resize_input_layer=resize_input_layer, max_instances=max_instances)
        elif "multi_instance" in model_types:
            i = model_types.index("multi_instance")
            predictor = BottomUpPredictor.from_trained_models(
                model_paths[i], peak_threshold=peak_threshold, integral_refinement=integral_refinement,
                integral_patch_size=integral_patch_size, batch_size=batch_size,
                resize_input_layer=resize_input_layer, max_instances=max_instances)
        else:
            raise ValueError("Could not create predictor from model paths:" + "\n".join(model_paths)
                             + f"\n(model types {model_types}: implemented are single_instance, centroid + "
                               "centered_instance (top-down) and multi_instance (bottom-up); ground-truth-centroid "
                               "variants, multi-class and MoveNet models are out of scope)")
        predictor.model_paths = model_paths
        return predictor

    @property
    def is_grayscale(self) -> bool:
        return self._input_network().in_channels == 1

    def _input_network(self):
        return self.inference_model.bottomup_layer.keras_model

    # -- generic batch loop (single-instance / top-down); BottomUpPredictor overrides it with the packed all-gather
    def _frames_of(self, data):
        if isinstance(data, (np.ndarray, torch.Tensor)):
            return data[None] if data.ndim == 3 else data
        if hasattr(data, "__len__") and hasattr(data, "__getitem__"):
            return data
        raise TypeError(f"unsupported data type for predict(): {type(data)}")

    def _predict_generator(self, data):
        """inference.py:377-494. Under torch.distributed each rank predicts its contiguous slice of the batch and
        the (small, ragged) NumPy results are exchanged with one all_gather_object."""
        import torch.distributed as dist

        frames = self._frames_of(data)
        n = len(frames)
        rank, world = parallel.rank_world()
        if world > 1 and hasattr(self.inference_model, "_device_networks"):
            for net in self.inference_model._device_networks():
                net.defer_range_agreement()  # kept in submit(): dist_agree_range() after the first global batch, on every rank
        small = {"instance_peaks", "instance_peak_vals", "instance_scores", "n_valid", "centroids", "centroid_vals", "status", "_input"}
        from ..io.video import Video, VideoReader

        self.make_pipeline(data if isinstance(data, (np.ndarray, Video, VideoReader)) else None)
        reporter = ProgressReporter(self.verbosity, self.report_rate, n, enabled=rank == 0)
        reporter.__enter__()
        try:

            # Frame sources the prefetcher can read (arrays, Video, VideoReader): this rank's batches are staged into page-locked
            # buffers by a producer thread and uploaded on a copy stream of their own, so the host copy and the DMA of batch k+1
            # run under the device work of batch k (from a pageable array, `frames[lo:hi].cuda()` is a synchronous staged copy on
            # the compute stream: top-down at 1024 x 1024 ran at 82 % of its HBM-resident rate, tools/predict_e2e_topdown.py)
            feed = feeder = copy_stream = None
            if isinstance(data, (np.ndarray, Video, VideoReader)) and torch.cuda.is_available() and n > 0 \
                    and not (isinstance(data, np.ndarray) and data.ndim == 3):
                from ..io.video import FramePrefetcher

                ranges = [parallel.shard_range(i0, min(i0 + self.batch_size, n), rank, world) for i0 in range(0, n, self.batch_size)]
                feeder = FramePrefetcher(data, [r for r in ranges if r[1] > r[0]], depth=4)
                feed = iter(feeder)
                copy_stream = getattr(self, "_upload_stream", None)  # (kept: creating a hardware queue costs milliseconds)
                if copy_stream is None:
                    copy_stream = self._upload_stream = torch.cuda.Stream(priority=int(os.environ.get("SLEAP_AMD_COPY_STREAM_PRIORITY", "-1")))

            # With prefetched sources the device work runs on a stream of the predictor's own instead of the default stream (as the
            # bottom-up layer does): on the default stream the uploads of the copy stream did not overlap it -- the top-down rate
            # from host frames was that of upload + compute in series (7.2 k frames/s against 8.6 k from a CUDA tensor).
            compute_stream = None
            if feed is not None:
                compute_stream = getattr(self, "_compute_stream", None)
                if compute_stream is None:
                    compute_stream = self._compute_stream = torch.cuda.Stream(priority=-1)

            keep_frames = world == 1 and bool(getattr(getattr(self, "tracker", None), "uses_image", False))
            import inspect

            # (an inference model whose outputs_to_numpy does not take the early-downloaded arrays converts at hand-out time)
            early_download = "host" in inspect.signature(self.inference_model.outputs_to_numpy).parameters

            def on_compute():
                import contextlib

                return torch.cuda.stream(compute_stream) if compute_stream is not None else contextlib.nullcontext()

            def batch_of(lo, hi):
                """This rank's frames of one batch on the device (prefetched sources) or as the source hands them out."""
                if feed is None:
                    return frames[lo:hi], None
                try:
                    _lo, _hi, _inds, pinned = next(feed)
                except StopIteration:  # the producer stops at a frame the source cannot deliver (inference.py:3333-3339)
                    raise KeyError(f"Unable to load frame: the source ended before frames {lo}..{hi - 1}") from None
                key = feeder.hold()
                with torch.cuda.stream(copy_stream):
                    dev_frames = pinned.cuda(non_blocking=True)
                    up = torch.cuda.Event()
                    up.record(copy_stream)
                cur = torch.cuda.current_stream()
                cur.wait_event(up)
                dev_frames.record_stream(cur)
                feeder.release_key(key, up)
                return dev_frames, up

            def submit(i0):
                """Queue one batch (everything asynchronous that the model leaves asynchronous) -> a ticket."""
                i1 = min(i0 + self.batch_size, n)
                lo, hi = parallel.shard_range(i0, i1, rank, world)
                dev, kept = None, None
                with on_compute():
                    if hi > lo:
                        batch, _up = batch_of(lo, hi)
                        dev = self.inference_model.predict_on_batch(batch, numpy=False)
                        if keep_frames and _up is not None:
                            kept = (batch, _up)  # a flow tracker reads the frames where they already are
                    if world > 1 and i0 == 0:
                        # fp16 range scales of a frame-sharded run: every rank (empty shards included) passes here exactly once,
                        # after the first global batch -- the networks agree on their exponents (DeviceNetwork.dist_agree_range)
                        # Network by network: when an earlier network (the centroid model) is re-compiled, what the
                        # later ones (the centred-instance model) saw so far came from its overflowed outputs -- they forget it,
                        # the batch runs again, and only then are they asked to agree.
                        nets = self.inference_model._device_networks() if hasattr(self.inference_model, "_device_networks") else []
                        for k, net in enumerate(nets):
                            if net.dist_agree_range():
                                for later in nets[k + 1:]:
                                    later.reset_pending_range()
                                if hi > lo:  # re-compiled plan: run the batch again
                                    dev = self.inference_model.predict_on_batch(batch, numpy=False)
                    pre = None
                    if dev is not None and not set(dev) <= small:
                        # maps / crops may alias network buffers that the next batch overwrites: convert before anything else runs
                        dev = self.inference_model.outputs_to_numpy(dev)
                    elif dev is not None and early_download:
                        pre = start_download(dev)  # queued behind this batch's kernels, before the next batch's
                return i0, i1, dev, pre, kept

            # two-deep pipeline: batch k+1 is queued before batch k is brought to the host (the conversion synchronises), so the
            # GPU works on k+1 while the host waits for, converts and hands out k
            if n == 0:
                return
            f0 = frames[0]
            image_hw = np.asarray(f0.shape[:2] if hasattr(f0, "shape") else (1, 1), np.int64)
            tickets = [submit(0)]
            for i_next in list(range(self.batch_size, n, self.batch_size)) + [None]:
                nxt = submit(i_next) if i_next is not None else None
                i0, i1, dev, pre, kept = tickets.pop(0)
                if nxt is not None:
                    tickets.append(nxt)
                ex = None
                if dev is not None:
                    with on_compute():
                        if isinstance(next(iter(dev.values())), np.ndarray):
                            ex = dev
                        elif pre is None:
                            ex = self.inference_model.outputs_to_numpy(dev)
                        else:
                            ex = self.inference_model.outputs_to_numpy(dev, host=finish_download(*pre))
                if world > 1:
                    parts = [None] * world
                    dist.all_gather_object(parts, ex)
                    parts = [p for p in parts if p is not None]
                    imax = max(p["instance_peaks"].shape[1] for p in parts)
                    ex = {}
                    for k in parts[0]:
                        vs = []
                        for p in parts:
                            v = p[k]
                            if v.ndim >= 2 and k != "n_valid" and v.shape[1] < imax:
                                pad = np.full((v.shape[0], imax - v.shape[1]) + v.shape[2:], np.nan, v.dtype)
                                v = np.concatenate([v, pad], axis=1)
                            vs.append(v)
                        ex[k] = np.concatenate(vs, axis=0)
                ex["video_ind"] = np.zeros((i1 - i0,), np.int64)
                ex["frame_ind"] = np.arange(i0, i1, dtype=np.int64)
                ex["scale"] = np.ones((i1 - i0, 2), np.float32)
                ex["image_hw"] = image_hw
                if kept is not None:
                    ex["image_dev"], ex["image_ready"] = kept
                reporter.update(i1 - i0)
                yield ex
        finally:  # also when the generator raises or is abandoned before its last batch (rich keeps the cursor hidden otherwise)
            reporter.__exit__(None, None, None)

    def predict(self, data, make_labels: bool = True):
        """inference.py:496-531 (see BottomUpPredictor.predict for the output contract)."""
        try:
            outs = self._apply_tracker(self._predict_generator(data), data)
        finally:
            self._clear_range_deferral()
        return self._make_labels(outs, data) if make_labels else outs

    def _clear_range_deferral(self):
        """The generators switch the networks' fp16 range gate to its deferred mode under torch.distributed (one agreement of all
        ranks after the first global batch). No batch (n == 0) or an exception before that point would leave the promise
        standing: later direct forward() calls would record an overflow silently instead of raising."""
        nets = getattr(self.inference_model, "_device_networks", None)
        for net in (nets() if nets is not None else []):
            if getattr(net, "_dist_defer", False):
                net.defer_range_agreement(False)

    def _skeleton_info(self):
        """(part_names, edge_inds) for the result containers; edges are only known to the bottom-up predictor (PAFScorer)."""
        scorer = getattr(getattr(self.inference_model, "bottomup_layer", None), "paf_scorer", None)
        if scorer is not None:
            return scorer.part_names, scorer.edge_inds
        for attr in ("confmap_config", "centroid_config"):
            cfg = getattr(self, attr, None)
            if cfg:
                for head in (cfg.get("model", {}).get("heads", {}) or {}).values():
                    if isinstance(head, dict):
                        names = head.get("part_names") or (head.get("confmaps") or {}).get("part_names")
                        if names:
                            return list(names), []
        return None, []

    def _make_labels(self, outs, data):
        """`predict(data)` -> array-backed `sleap_amd.io.labels.Labels` (same filtering / ordering as the reference's object
        builder, inference.py:3230-3348; `.save()` writes a `.slp`, `.to_sleap()` gives the reference's own classes)."""
        from ..io.labels import Labels
        from ..io.video import Video, VideoReader

        names, edges = self._skeleton_info()
        if names is None:
            n = outs[0]["instance_peaks"].shape[2] if outs else 0
            names = [f"node_{i}" for i in range(n)]
        video = data.video if isinstance(data, VideoReader) else data if isinstance(data, Video) else (
            Video.from_numpy(data) if isinstance(data, np.ndarray) else None)
        tn = self.tracker.spawned_tracks if self.tracker and hasattr(self.tracker, "spawned_tracks") else None
        return Labels.from_predictions(outs, names, edges, video=video, track_names=tn,
                                       max_instances=getattr(self, "max_instances", None))

    def export_model(self, save_path: str, signatures: str = "serving_default", save_traces: bool = True,
                     model_name: Optional[str] = None, tensors: Optional[Dict[str, str]] = None, unrag_outputs: bool = True,
                     max_instances: Optional[int] = None):
        """inference.py:533-591 (and the per-predictor overrides, :1615, :2707, :4196): exports the Keras model as a frozen
        TensorFlow graph for TF-Serving-style deployments. Present so that the attribute path of the reference exists; there is no
        TensorFlow graph on this path to freeze (SURVEY.md section 2 marks export out of scope)."""
        raise NotImplementedError(
            "out of scope: TF SavedModel / frozen-graph export (Predictor.export_model, sleap/nn/inference.py:533). The model folder "
            "(training_config.json + best_model.h5) IS the deployable artefact of sleap_amd: load it with sleap_amd.nn.inference."
            "load_model(path); use the reference's own `sleap-export` for a TensorFlow graph.")

    def save_predictions(self, filename: str, outs: List[Dict[str, np.ndarray]], video: Optional[dict] = None,
                         part_names: Optional[List[str]] = None, edges=None):
        """Write `predict(make_labels=False)` results as a `.slp` file the reference can open (`sleap.load_file`):
        the columnar writer of `sleap_amd.io.slp` instead of per-instance `PredictedInstance` objects + `Labels.save`."""
        from ..io import slp

        if part_names is None:
            scorer = getattr(getattr(self.inference_model, "bottomup_layer", None), "paf_scorer", None)
            if scorer is None:
                raise ValueError("part_names / edges are required for predictors without a PAFScorer")
            part_names, edges = scorer.part_names, scorer.edge_inds
        names = self.tracker.spawned_tracks if self.tracker and hasattr(self.tracker, "spawned_tracks") else None
        return slp.write_slp(filename, outs, part_names, edges or [], video=video, track_names=names,
                             max_instances=self.max_instances if hasattr(self, "max_instances") else None)

    def _instance_scores(self, outs):
        """The score the reference gives each PredictedInstance, as `instance_scores` where the model does not emit one:
        top-down -> the centroid confidence (inference.py:2639-2660 `instance_score=score` over `centroid_vals`);
        single-instance -> `np.nansum(confidences)` (inference.py:1578). Bottom-up models emit it themselves."""
        for ex in outs:
            if "instance_scores" in ex:
                continue
            if "centroid_vals" in ex:
                ex["instance_scores"] = np.asarray(ex["centroid_vals"], np.float32)
            elif "instance_peak_vals" in ex:
                ex["instance_scores"] = np.nansum(np.asarray(ex["instance_peak_vals"], np.float32), axis=-1)
        return outs

    def _apply_tracker(self, outs, data=None) -> List[Dict[str, np.ndarray]]:
        """Identity tracking over the gathered per-batch arrays, strictly in frame order, where the reference runs it
        (inference.py:3306-3313, 3345-3346). `outs` may be the predict generator itself: every batch is tracked as soon as it
        arrives, i.e. while the device already runs the next batch's network (the generator keeps two batches in flight), so
        the tracker -- host matching, and for flow trackers the Lucas-Kanade launches -- overlaps inference instead of
        following it. Adds `track_inds (b, I)` (-1 = no track), `tracking_scores (b, I)` and `track_order (b, I)` (position in
        the tracker's returned list) to every batch (`data`: the source, for flow trackers); `predictor.tracker.spawned_tracks`
        names the tracks. What reaches the tracker is what the reference hands it (`tracking.select_instances`): all-NaN
        instances dropped, and -- bottom-up only (inference.py:3297-3304) -- the `max_instances` best by score, in that
        order. Requires the array tracker of `sleap_amd.nn.tracking`."""
        trk = self.tracker
        tracking = bool(trk) and hasattr(trk, "track_frames")
        if tracking:
            from .tracking import finish_tracks, frames_of, image_hw_of, track_example

            cap = getattr(self, "max_instances", None) if isinstance(self, BottomUpPredictor) else None
        done = []
        for ex in outs:
            self._instance_scores([ex])
            if tracking:
                # flow trackers look at the frames (tracker.track(..., img=...), inference.py:2662-2668, 3306-3313): the
                # carried `image`, or re-read from the source on the rank that tracks
                track_example(trk, ex, img_hw=image_hw_of(ex), max_instances=cap,
                              images=frames_of(ex, data) if getattr(trk, "uses_image", False) else None,
                              images_ready=ex.get("image_ready") if ex.get("image_dev") is not None else None)
            ex.pop("image_dev", None)  # the device copy of the frames was only kept for the tracker
            ex.pop("image_ready", None)
            done.append(ex)
        return finish_tracks(done, trk) if (tracking and done) else done


class SingleInstancePredictor(Predictor):
    """inference.py:1415-1635."""

    def __init__(self, confmap_config: dict, confmap_model: DeviceNetwork, inference_model=None, peak_threshold: float = 0.2,
                 integral_refinement: bool = True, integral_patch_size: int = 5, batch_size: int = 4,
                 verbosity: str = "rich", report_rate: float = 2.0, model_paths=None):
        self.confmap_config = confmap_config
        self.confmap_model = confmap_model
        self.peak_threshold = peak_threshold
        self.integral_refinement = integral_refinement
        self.integral_patch_size = integral_patch_size
        self.batch_size = batch_size
        self.verbosity, self.report_rate, self.model_paths = verbosity, report_rate, model_paths or []
        self.tracker = None
        self._initialize_inference_model()

    def _initialize_inference_model(self):
        """inference.py:1456-1468."""
        cfg = self.confmap_config
        self.inference_model = SingleInstanceInferenceModel(
            SingleInstanceInferenceLayer(
                keras_model=self.confmap_model,
                input_scale=cfg["data"]["preprocessing"].get("input_scaling", 1.0),
                pad_to_stride=model_io.maximum_stride(cfg),
                peak_threshold=self.peak_threshold,
                refinement="integral" if self.integral_refinement else "local",
                integral_patch_size=self.integral_patch_size,
                output_stride=cfg["model"]["heads"]["single_instance"]["output_stride"]))

    def _input_network(self):
        return self.confmap_model

    @property
    def data_config(self):
        return self.confmap_config["data"]

    @classmethod
    def from_trained_models(cls, model_path: str, peak_threshold: float = 0.2, integral_refinement: bool = True,
                            integral_patch_size: int = 5, batch_size: int = 4, resize_input_layer: bool = True):
        """inference.py:1479-1529."""
        cfg = model_io.load_training_config(model_path)
        mc, weights = model_io.load_keras_model(model_io.model_dir(model_path))
        obj = cls(confmap_config=cfg, confmap_model=DeviceNetwork(mc, weights), peak_threshold=peak_threshold,
                  integral_refinement=integral_refinement, integral_patch_size=integral_patch_size, batch_size=batch_size)
        obj.model_paths = [model_io.model_dir(model_path)]
        return obj


class TopDownPredictor(Predictor):
    """inference.py:2314-2734 (both models given; ground-truth centroid / peak variants are out of scope)."""

    def __init__(self, centroid_config: dict, centroid_model: DeviceNetwork, confmap_config: dict,
                 confmap_model: DeviceNetwork, inference_model=None, peak_threshold: float = 0.2, batch_size: int = 4,
                 integral_refinement: bool = True, integral_patch_size: int = 5, max_instances: Optional[int] = None,
                 verbosity: str = "rich", report_rate: float = 2.0, model_paths=None):
        self.centroid_config, self.centroid_model = centroid_config, centroid_model
        self.confmap_config, self.confmap_model = confmap_config, confmap_model
        self.peak_threshold = peak_threshold
        self.batch_size = batch_size
        self.integral_refinement = integral_refinement
        self.integral_patch_size = integral_patch_size
        self.max_instances = max_instances
        self.verbosity, self.report_rate, self.model_paths = verbosity, report_rate, model_paths or []
        self.tracker = None
        self._initialize_inference_model()

    def _initialize_inference_model(self):
        """inference.py:2373-2424."""
        ccfg, icfg = self.centroid_config, self.confmap_config
        pad = ccfg["data"]["preprocessing"].get("pad_to_stride") or model_io.maximum_stride(ccfg)
        centroid_crop_layer = CentroidCrop(
            keras_model=self.centroid_model,
            crop_size=icfg["data"]["instance_cropping"]["crop_size"],
            input_scale=ccfg["data"]["preprocessing"].get("input_scaling", 1.0),
            precrop_resize=1.0,
            pad_to_stride=pad,
            output_stride=ccfg["model"]["heads"]["centroid"]["output_stride"],
            peak_threshold=self.peak_threshold,
            refinement="integral" if self.integral_refinement else "local",
            integral_patch_size=self.integral_patch_size,
            return_confmaps=False,
            max_instances=self.max_instances)
        instance_peaks_layer = FindInstancePeaks(
            keras_model=self.confmap_model,
            input_scale=icfg["data"]["preprocessing"].get("input_scaling", 1.0),
            peak_threshold=self.peak_threshold,
            output_stride=icfg["model"]["heads"]["centered_instance"]["output_stride"],
            refinement="integral" if self.integral_refinement else "local",
            integral_patch_size=self.integral_patch_size,
            return_confmaps=False,
            resize_input_image=False)
        centroid_crop_layer.precrop_resize = icfg["data"]["preprocessing"].get("input_scaling", 1.0)
        self.inference_model = TopDownInferenceModel(centroid_crop=centroid_crop_layer, instance_peaks=instance_peaks_layer)

    def _input_network(self):
        return self.centroid_model

    @property
    def data_config(self):
        return self.centroid_config["data"]

    @classmethod
    def from_trained_models(cls, centroid_model_path: Optional[str] = None, confmap_model_path: Optional[str] = None,
                            batch_size: int = 4, peak_threshold: float = 0.2, integral_refinement: bool = True,
                            integral_patch_size: int = 5, resize_input_layer: bool = True,
                            max_instances: Optional[int] = None):
        """inference.py:2447-2540."""
        if centroid_model_path is None or confmap_model_path is None:
            raise ValueError("Both a centroid and a centered-instance model are required: predicting from ground truth "
                             "centroids / peaks (CentroidCropGroundTruth, FindInstancePeaksGroundTruth) is out of scope.")
        nets, cfgs = [], []
        for mp in (centroid_model_path, confmap_model_path):
            cfgs.append(model_io.load_training_config(mp))
            mc, weights = model_io.load_keras_model(model_io.model_dir(mp))
            nets.append(DeviceNetwork(mc, weights))
        obj = cls(centroid_config=cfgs[0], centroid_model=nets[0], confmap_config=cfgs[1], confmap_model=nets[1],
                  peak_threshold=peak_threshold, batch_size=batch_size, integral_refinement=integral_refinement,
                  integral_patch_size=integral_patch_size, max_instances=max_instances)
        obj.model_paths = [model_io.model_dir(centroid_model_path), model_io.model_dir(confmap_model_path)]
        return obj


class BottomUpPredictor(Predictor):
    """inference.py:3055-3348. Same fields and defaults as the reference's attrs class."""

    def __init__(self, bottomup_config: dict, bottomup_model: DeviceNetwork, inference_model=None, pipeline=None,
                 tracker=None, peak_threshold: float = 0.2, batch_size: int = 4, integral_refinement: bool = True,
                 integral_patch_size: int = 5, max_edge_length_ratio: float = 0.25, dist_penalty_weight: float = 1.0,
                 paf_line_points: int = 10, min_line_scores: float = 0.25, max_instances: Optional[int] = None,
                 verbosity: str = "rich", report_rate: float = 2.0, model_paths=None):
        self.bottomup_config = bottomup_config
        self.bottomup_model = bottomup_model
        self.inference_model = inference_model
        self.pipeline = pipeline
        self.tracker = tracker
        self.peak_threshold = peak_threshold
        self.batch_size = batch_size
        self.integral_refinement = integral_refinement
        self.integral_patch_size = integral_patch_size
        self.max_edge_length_ratio = max_edge_length_ratio
        self.dist_penalty_weight = dist_penalty_weight
        self.paf_line_points = paf_line_points
        self.min_line_scores = min_line_scores
        self.max_instances = max_instances
        self.verbosity = verbosity
        self.report_rate = report_rate
        self.model_paths = model_paths or []
        self._initialize_inference_model()

    def _initialize_inference_model(self):
        """inference.py:3119-3140."""
        cfg = self.bottomup_config
        mi = cfg["model"]["heads"]["multi_instance"]
        pre = cfg["data"]["preprocessing"]
        self.inference_model = BottomUpInferenceModel(
            BottomUpInferenceLayer(
                keras_model=self.bottomup_model,
                paf_scorer=PAFScorer.from_config(
                    mi, max_edge_length_ratio=self.max_edge_length_ratio,
                    dist_penalty_weight=self.dist_penalty_weight, n_points=self.paf_line_points,
                    min_instance_peaks=0, min_line_scores=self.min_line_scores),
                input_scale=pre.get("input_scaling", 1.0),
                pad_to_stride=model_io.maximum_stride(cfg),
                peak_threshold=self.peak_threshold,
                refinement="integral" if self.integral_refinement else "local",
                integral_patch_size=self.integral_patch_size,
                cm_output_stride=mi["confmaps"]["output_stride"],
                paf_output_stride=mi["pafs"]["output_stride"],
            ))

    @property
    def data_config(self):
        return self.bottomup_config["data"]

    @classmethod
    def from_trained_models(cls, model_path: str, batch_size: int = 4, peak_threshold: float = 0.2,
                            integral_refinement: bool = True, integral_patch_size: int = 5,
                            max_edge_length_ratio: float = 0.25, dist_penalty_weight: float = 1.0,
                            paf_line_points: int = 10, min_line_scores: float = 0.25,
                            resize_input_layer: bool = True, max_instances: Optional[int] = None) -> "BottomUpPredictor":
        """inference.py:3152-3228: reads `<dir>/training_config.json` + `<dir>/best_model.h5`.
        (`resize_input_layer` is accepted for compatibility: the engine is fully convolutional.)"""
        cfg = model_io.load_training_config(model_path)
        mc, weights = model_io.load_keras_model(model_io.model_dir(model_path))
        net = DeviceNetwork(mc, weights)
        obj = cls(bottomup_config=cfg, bottomup_model=net, peak_threshold=peak_threshold, batch_size=batch_size,
                  integral_refinement=integral_refinement, integral_patch_size=integral_patch_size,
                  max_edge_length_ratio=max_edge_length_ratio, dist_penalty_weight=dist_penalty_weight,
                  paf_line_points=paf_line_points, min_line_scores=min_line_scores, max_instances=max_instances)
        obj.model_paths = [model_io.model_dir(model_path)]
        return obj

    # ------------------------------------------------------------------ prediction
    def _frames_of(self, data):
        if isinstance(data, np.ndarray) or isinstance(data, torch.Tensor):
            if data.ndim == 3:
                data = data[None]
            return data
        if hasattr(data, "__len__") and hasattr(data, "__getitem__"):
            return data  # video-like: data[i:j] -> (n, H, W, C)
        raise TypeError(f"unsupported data type for predict(): {type(data)}")

    def _predict_generator(self, data) -> Iterator[Dict[str, np.ndarray]]:
        """inference.py:377-494: one dict per (global) batch, NumPy values, coordinates in image pixels.

        Under torch.distributed each rank runs the network on its contiguous slice of the batch and one
        all-gather assembles the batch on every rank (parallel.gather_batch_results)."""
        from ..io.video import FramePrefetcher, Video, VideoReader

        if isinstance(data, torch.Tensor):
            data = data.cpu().numpy()
        if isinstance(data, np.ndarray):
            data = Video.from_numpy(data)
        if isinstance(data, Video):
            data = VideoReader(data)
        reader = data if isinstance(data, VideoReader) else VideoReader(Video.from_numpy(np.stack(list(self._frames_of(data)))))
        n = len(reader)
        index_of = np.asarray(reader.indices(), dtype=np.int64)
        image_hw = np.asarray(reader.video.shape[1:3], np.int64)  # raw frame size (tracker similarities; present on every rank)
        layer = self.inference_model.bottomup_layer
        rank, world = parallel.rank_world()
        if world > 1 and hasattr(layer.keras_model, "defer_range_agreement"):
            layer.keras_model.defer_range_agreement()  # kept below: dist_agree_range() after the first global batch, on every rank
        batches = [(i0, min(i0 + self.batch_size, n)) for i0 in range(0, n, self.batch_size)]
        mine = [parallel.shard_range(i0, i1, rank, world) for i0, i1 in batches]
        # This rank's frames are read ahead by a producer thread into page-locked buffers (sleap_amd/io/video.py). The loop is a
        # two-deep software pipeline: batch k+1 is SUBMITTED (upload on the copy stream, network, post-processing, result
        # gather and the copy of the packed results to a page-locked host buffer -- all asynchronous) before batch k is
        # FINALISED (wait for its results, inspect the status words, build the example). A per-batch synchronisation would
        # leave the GPU idle between batches and undo the network / post-processing overlap.
        feeder = FramePrefetcher(reader, [r for r in mine if r[1] > r[0]], depth=4)
        feed = iter(feeder)
        src = getattr(reader.video.backend, "_data", None)  # in-memory / memory-mapped source: "image" is a view of it
        n_nodes = layer.paf_scorer.n_nodes
        dev = layer.keras_model.device
        over_mask = _lib.STATUS_PEAK_OVERFLOW | _lib.STATUS_NODE_PEAK_OVERFLOW | _lib.STATUS_INSTANCE_OVERFLOW
        host_pool = {}

        agreed = [False]

        def caps():
            return (layer.max_peaks, layer.paf_scorer.max_node_peaks, layer.paf_scorer.max_instances)

        def run_shard(batch):
            """-> (packed results of the whole batch on this rank's device or None, instance capacity)"""
            packed, ig = None, layer.paf_scorer.max_instances
            if batch is not None:
                outs = self.inference_model.call(batch)
            if world > 1 and not agreed[0]:
                # fp16 range scales of a frame-sharded run: all ranks (empty shards included) agree ONCE, here, after the first
                # global batch -- the only program point every rank passes (DeviceNetwork.dist_agree_range)
                agreed[0] = True
                agree = getattr(layer.keras_model, "dist_agree_range", None)
                if agree is not None and agree() and batch is not None:
                    outs = self.inference_model.call(batch)  # the plan was re-compiled with range scales: run the batch again
            if batch is not None:
                ig = outs["instance_scores"].shape[1]
                packed = parallel.pack_results(outs)
            return packed, ig

        def to_host(packed, n_batch, ig):
            g = parallel.gather_batch_results(packed, n_batch, ig, n_nodes, world, device=dev)
            if g.is_cuda:
                key = tuple(g.shape)
                ring = host_pool.setdefault(key, [])
                if len(ring) < 3:  # at most two batches are in flight; page-locked allocations are expensive
                    ring.append(torch.empty(g.shape, dtype=g.dtype).pin_memory())
                host_pool["n"] = host_pool.get("n", 0) + 1
                host = ring[host_pool["n"] % len(ring)] if len(ring) == 3 else ring[-1]
                host.copy_(g, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                return host, ev
            return g, None

        def submit(i0, i1, lo, hi):
            t = {"i0": i0, "i1": i1, "lo": lo, "hi": hi, "batch": None, "key": None, "image": None, "up": None}
            if hi > lo:
                _lo, _hi, _inds, batch = next(feed)
                t["batch"], t["key"] = batch, feeder.hold()
            t["caps"] = caps()
            packed, t["ig"] = run_shard(t["batch"])
            if t["batch"] is not None:
                t["up"] = layer.last_upload_done
                if world == 1 and getattr(self.tracker, "uses_image", False):
                    t["image_dev"] = layer.last_upload  # a flow tracker reads the frames where they already are
                layer.last_upload = None
                if world == 1:
                    if src is None:
                        t["image"] = t["batch"].numpy().copy()  # the page-locked buffer is recycled
                    else:
                        t["image"] = src[lo:hi] if reader.example_indices is None else src[index_of[lo:hi]]
            t["host"], t["ev"] = to_host(packed, i1 - i0, t["ig"])
            return t

        def finalise(t):
            while True:
                if t["ev"] is not None:
                    t["ev"].synchronize()
                # host side in NumPy only (see parallel.unpack_results_np); the page-locked buffer is recycled, hence the copy
                res = parallel.unpack_results_np(t["host"].numpy().copy(), t["ig"], n_nodes)
                bits = int(np.bitwise_or.reduce(res["status"].astype(np.int64))) if len(res["status"]) else 0
                if bits & _lib.STATUS_NONFINITE:
                    raise FloatingPointError(NONFINITE_MESSAGE)
                if bits & _lib.STATUS_LSA_INFEASIBLE:
                    raise ValueError("cost matrix is infeasible")  # what scipy raises inside the reference
                over = bits & over_mask
                if not over:
                    break
                # a fixed-capacity device buffer overflowed somewhere in the batch. Every rank sees the same status words
                # (they travel in the gathered rows), so every rank takes the same decision: if the caps have not changed
                # since this result was computed, double the ones that overflowed; then re-run the shard. (A batch that was
                # in flight while an earlier one grew the caps is first re-run with the caps as they are now.)
                if t["caps"] == caps() and not self.inference_model._grow_caps(over):
                    raise GroupingOverflowError(
                        f"a frame exceeds the hard capacity of the device buffers (status bits {bits})")
                t["caps"] = caps()
                packed, t["ig"] = run_shard(t["batch"])
                t["host"], t["ev"] = to_host(packed, t["i1"] - t["i0"], t["ig"])
            if t["key"] is not None:
                # the results of this batch are on the host, so its upload finished long ago: no event for the producer
                # thread to wait on (and no HIP call from that thread at all)
                feeder.release_key(t["key"], None)
            bound = int(res["n_valid"].max()) if len(res["n_valid"]) else 0  # unrag_example: the batch's bounding shape
            ex = {k: np.ascontiguousarray(res[k][:, :bound]) for k in ("instance_peaks", "instance_peak_vals", "instance_scores")}
            ex["n_valid"] = res["n_valid"].astype(np.int64)
            i0, i1 = t["i0"], t["i1"]
            ex["video_ind"] = np.zeros((i1 - i0,), np.int64)
            ex["frame_ind"] = index_of[i0:i1].copy()
            ex["scale"] = np.ones((i1 - i0, 2), np.float32)
            ex["image_hw"] = image_hw
            if world == 1:
                ex["image"] = t["image"]
                if t.get("image_dev") is not None:
                    ex["image_dev"], ex["image_ready"] = t["image_dev"], t.get("up")
            return ex

        self.make_pipeline(reader)
        reporter = ProgressReporter(self.verbosity, self.report_rate, n, enabled=rank == 0)

        pending = []
        with reporter:
            for (i0, i1), (lo, hi) in zip(batches, mine):
                try:
                    pending.append(submit(i0, i1, lo, hi))
                except StopIteration:  # the source stopped early ("Unable to load frame"): end like the reference does
                    break
                if len(pending) > 1:
                    ex = finalise(pending.pop(0))
                    reporter.update(len(ex["frame_ind"]))
                    yield ex
            while pending:
                ex = finalise(pending.pop(0))
                reporter.update(len(ex["frame_ind"]))
                yield ex

    def predict(self, data, make_labels: bool = True):
        """inference.py:496-531. `make_labels=False` -> list of per-batch dicts of NumPy arrays
        (`instance_peaks (b, Imax, N, 2)` NaN-padded, `instance_peak_vals`, `instance_scores`, `n_valid`,
        `video_ind`, `frame_ind`, ...). `make_labels=True` (the reference's default) returns the array-backed
        `sleap_amd.io.labels.Labels` (`len`, indexing, `.numpy()`, `.save("x.slp")`, `.to_sleap()`)."""
        try:
            outs = self._apply_tracker(self._predict_generator(data), data)
        finally:
            self._clear_range_deferral()
        return self._make_labels(outs, data) if make_labels else outs


def load_model(model_path: Union[str, List[str]], batch_size: int = 4, peak_threshold: float = 0.2,
               refinement: str = "integral", tracker: Optional[str] = None, tracker_window: int = 5,
               tracker_max_instances: Optional[int] = None, disable_gpu_preallocation: bool = True,
               progress_reporting: str = "rich", resize_input_layer: bool = True,
               max_instances: Optional[int] = None, dtype: Optional[str] = None) -> Predictor:
    """inference.py:4865-5004: accepts model folders, `training_config.json` paths or `.zip` archives.

    New: `dtype` = 16-bit storage type of the network's activations / conv weights: "fp16" (default, SLEAP_AMD_DTYPE; the one
    that tracks the reference's fp32 numerics, finite range 65504) or "bf16" (fp32's range; see DESIGN.md sections 2 and 4)."""
    if isinstance(model_path, str):
        model_paths = [model_path]
    else:
        model_paths = list(model_path)
    resolved = []
    for mp in model_paths:
        if mp.endswith(".zip"):
            tmp = tempfile.mkdtemp()
            with zipfile.ZipFile(mp) as z:
                z.extractall(tmp)
            found = [r for r, _, f in os.walk(tmp) if "training_config.json" in f]
            resolved.extend(found)
        else:
            resolved.append(mp)
    tracker_obj = None
    if tracker is not None:  # inference.py:4985-4998
        from .tracking import Tracker

        use_max_tracker = tracker_max_instances is not None
        if use_max_tracker and not tracker.endswith("maxtracks"):
            tracker += "maxtracks"
        tracker_obj = Tracker.make_tracker_by_name(tracker=tracker, track_window=tracker_window,
                                                   post_connect_single_breaks=True, max_tracking=use_max_tracker,
                                                   max_tracks=tracker_max_instances)
    from . import engine as _engine

    prev, _engine.DEFAULT_DTYPE = _engine.DEFAULT_DTYPE, dtype or _engine.DEFAULT_DTYPE
    try:
        predictor = Predictor.from_model_paths(resolved, peak_threshold=peak_threshold,
                                               integral_refinement=refinement == "integral", batch_size=batch_size,
                                               resize_input_layer=resize_input_layer, max_instances=max_instances)
    finally:
        _engine.DEFAULT_DTYPE = prev
    predictor.verbosity = progress_reporting  # inference.py:4981: "rich" | "json" | "none"
    predictor.tracker = tracker_obj
    return predictor
