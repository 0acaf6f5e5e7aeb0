"""Thin Python wrappers over the C ABI: torch is used ONLY for device memory and streams.

Every function takes/returns torch CUDA tensors (contiguous) and enqueues work on the current
torch stream. Shapes and meanings follow include/sleap_amd.h.
"""
import collections
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import REFINE, check


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t_or_dev=None):
    if isinstance(t_or_dev, torch.Tensor):
        return t_or_dev.device
    return torch.device("cuda", torch.cuda.current_device()) if t_or_dev is None else torch.device(t_or_dev)


def require_cuda():
    if not torch.cuda.is_available():
        raise RuntimeError("sleap_amd needs a ROCm GPU (gfx950); no CPU fallback exists")


def to_cuda_f32(x):
    require_cuda()
    if isinstance(x, torch.Tensor):
        return x.to(device="cuda", dtype=torch.float32).contiguous()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float32))).cuda()


def device_info(device=0):
    n_cu, lds, wave = C.c_int(), C.c_int(), C.c_int()
    arch = C.create_string_buffer(64)
    check(_lib.lib().sa_device_info(device, C.byref(n_cu), C.byref(lds), C.byref(wave), arch, 64), "sa_device_info")
    return {"n_cu": n_cu.value, "lds_bytes": lds.value, "wave_size": wave.value, "arch": arch.value.decode()}


# --------------------------------------------------------------------------------------------------
# peak finding
# --------------------------------------------------------------------------------------------------
def find_local_peaks(cms, offsets=None, threshold=0.2, refinement=None, patch_size=5, xy_scale=1.0,
                     max_peaks=512, status=None):
    """-> peak_xy [B,P,2], peak_val [B,P], peak_chan [B,P] i32, peak_count [B] i32, status [B] i32."""
    B, H, W, Cc = cms.shape
    dev = cms.device
    peak_xy = torch.empty((B, max_peaks, 2), dtype=torch.float32, device=dev)
    peak_val = torch.empty((B, max_peaks), dtype=torch.float32, device=dev)
    peak_chan = torch.empty((B, max_peaks), dtype=torch.int32, device=dev)
    peak_count = torch.empty((B,), dtype=torch.int32, device=dev)
    if status is None:
        status = torch.zeros((B,), dtype=torch.int32, device=dev)
    h = _lib.lib()
    ws_bytes = h.sa_find_local_peaks_workspace(B, max_peaks)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    mode = REFINE["offsets"] if offsets is not None else REFINE[refinement]
    check(
        h.sa_find_local_peaks(_ptr(cms), _ptr(offsets), B, H, W, Cc, float(threshold), mode, int(patch_size),
                              float(xy_scale), int(max_peaks), _ptr(peak_xy), _ptr(peak_val), _ptr(peak_chan),
                              _ptr(peak_count), _ptr(status), _ptr(ws), ws_bytes, _stream()),
        "sa_find_local_peaks",
    )
    return peak_xy, peak_val, peak_chan, peak_count, status


def find_local_peaks_rough(cms, threshold=0.2, max_peaks=512, status=None):
    """sa_find_local_peaks_rough: the NMS scan alone -> keys [B,max_peaks] (linear (y, x, c) indices as int32 bit patterns, arrival
    order), peak_count [B] i32, status [B] i32."""
    B, H, W, Cc = cms.shape
    keys = torch.empty((B, max_peaks), dtype=torch.int32, device=cms.device)
    peak_count = torch.empty((B,), dtype=torch.int32, device=cms.device)
    if status is None:
        status = torch.zeros((B,), dtype=torch.int32, device=cms.device)
    check(_lib.lib().sa_find_local_peaks_rough(_ptr(cms), B, H, W, Cc, float(threshold), int(max_peaks), _ptr(keys), _ptr(peak_count),
                                               _ptr(status), _stream()), "sa_find_local_peaks_rough")
    return keys, peak_count, status


def find_global_peaks(cms, offsets=None, threshold=0.2, refinement=None, patch_size=5, xy_scale=1.0):
    """-> peak_xy [B,C,2] (NaN below threshold), peak_val [B,C]."""
    B, H, W, Cc = cms.shape
    peak_xy = torch.empty((B, Cc, 2), dtype=torch.float32, device=cms.device)
    peak_val = torch.empty((B, Cc), dtype=torch.float32, device=cms.device)
    mode = REFINE["offsets"] if offsets is not None else REFINE[refinement]
    check(
        _lib.lib().sa_find_global_peaks(_ptr(cms), _ptr(offsets), B, H, W, Cc, float(threshold), mode,
                                        int(patch_size), float(xy_scale), _ptr(peak_xy), _ptr(peak_val), _stream()),
        "sa_find_global_peaks",
    )
    return peak_xy, peak_val


def crop_and_resize(images, centres_xy, sample_inds, crop_size):
    """-> crops (n, crop, crop, C) of the images' dtype (uint8 or float32) centred on fractional (x, y) centres."""
    B, H, W, Cc = images.shape
    n = centres_xy.shape[0]
    out = torch.empty((n, crop_size, crop_size, Cc), dtype=images.dtype, device=images.device)
    if n:
        is_u8 = 1 if images.dtype == torch.uint8 else 0
        if not is_u8 and images.dtype != torch.float32:
            raise ValueError("crop_and_resize: images must be uint8 or float32")
        centres_xy = centres_xy.to(torch.float32).contiguous()
        sample_inds = sample_inds.to(torch.int32).contiguous()
        check(_lib.lib().sa_crop_and_resize(_ptr(images), is_u8, H, W, Cc, _ptr(centres_xy), _ptr(sample_inds), n,
                                            int(crop_size), _ptr(out), _stream()), "sa_crop_and_resize")
    return out


# --------------------------------------------------------------------------------------------------
# PAF grouping
# --------------------------------------------------------------------------------------------------
def paf_score(pafs, peak_xy, peak_chan, peak_count, edges, n_nodes, n_points, pafs_stride, max_edge_length,
              dist_penalty_weight, max_node_peaks, status):
    B, Hp, Wp, C2 = pafs.shape
    E = C2 // 2
    dev = pafs.device
    P = peak_xy.shape[1]
    node_count = torch.empty((B, n_nodes), dtype=torch.int32, device=dev)
    node_peaks = torch.empty((B, n_nodes, max_node_peaks), dtype=torch.int32, device=dev)
    line_scores = torch.full((B, E, max_node_peaks, max_node_peaks), float("nan"), dtype=torch.float32, device=dev)
    check(
        _lib.lib().sa_paf_score(_ptr(pafs), B, Hp, Wp, E, _ptr(peak_xy), _ptr(peak_chan), _ptr(peak_count), P,
                                _ptr(edges), n_nodes, int(n_points), float(pafs_stride), float(max_edge_length),
                                float(dist_penalty_weight), int(max_node_peaks), _ptr(node_count),
                                _ptr(node_peaks), _ptr(line_scores), _ptr(status), _stream()),
        "sa_paf_score",
    )
    return node_count, node_peaks, line_scores


def paf_match(line_scores, node_count, edges, status):
    B, E, NP, _ = line_scores.shape
    N = node_count.shape[1]
    dev = line_scores.device
    match_dst = torch.empty((B, E, NP), dtype=torch.int32, device=dev)
    match_score = torch.empty((B, E, NP), dtype=torch.float32, device=dev)
    ws = _paf_workspace(B, E, N, NP, dev)
    check(
        _lib.lib().sa_paf_match(_ptr(line_scores), _ptr(node_count), _ptr(edges), B, E, N, NP, _ptr(match_dst),
                                _ptr(match_score), _ptr(status), _ptr(ws), ws.numel(), _stream()),
        "sa_paf_match",
    )
    return match_dst, match_score


_WS_CACHE = {}


def _paf_workspace(B, E, N, NP, dev):
    """One cached workspace per (shape, device): match and group run back to back on the same stream."""
    key = (B, E, N, NP, str(dev))
    if key not in _WS_CACHE:
        if len(_WS_CACHE) > 8:
            _WS_CACHE.clear()
        _WS_CACHE[key] = torch.empty((_lib.lib().sa_paf_workspace(B, E, N, NP),), dtype=torch.uint8, device=dev)
    return _WS_CACHE[key]


def paf_group(peak_xy, peak_val, node_count, node_peaks, match_dst, match_score, edges, sorted_edge_inds,
              min_line_scores, min_instance_peaks, max_instances, status):
    B, N, NP = node_peaks.shape
    E = match_dst.shape[1]
    P = peak_xy.shape[1]
    dev = peak_xy.device
    inst = torch.empty((B, max_instances, N, 2), dtype=torch.float32, device=dev)
    vals = torch.empty((B, max_instances, N), dtype=torch.float32, device=dev)
    scores = torch.empty((B, max_instances), dtype=torch.float32, device=dev)
    n_inst = torch.empty((B,), dtype=torch.int32, device=dev)
    ws = _paf_workspace(B, E, N, NP, dev)
    check(
        _lib.lib().sa_paf_group(_ptr(peak_xy), _ptr(peak_val), _ptr(node_count), _ptr(node_peaks), P,
                                _ptr(match_dst), _ptr(match_score), _ptr(edges), _ptr(sorted_edge_inds),
                                int(sorted_edge_inds.numel()), B, E, N, NP, float(min_line_scores),
                                int(min_instance_peaks), int(max_instances), _ptr(inst), _ptr(vals), _ptr(scores),
                                _ptr(n_inst), _ptr(status), _ptr(ws), ws.numel(), _stream()),
        "sa_paf_group",
    )
    return inst, vals, scores, n_inst


# Scratch of the fused post-processing, one per (shape, device, STREAM): two layers (or threads) with equal shapes on different
# streams must not share scan counters / keys / PAF scratch, and a buffer is only ever used -- and therefore only ever handed
# back to torch's caching allocator -- on the stream it was allocated under, so evicting one entry cannot free memory under a
# kernel queued elsewhere. The scan counters at its head are zeroed at allocation and handed back zeroed by every SUCCESSFUL
# call (sleap_amd.h); a failed call drops its entry, so the next call starts from a freshly zeroed one.
_PP_WS = collections.OrderedDict()
_PP_WS_MAX = 8


def bottomup_postproc(cms, offsets, pafs, threshold, refinement, patch_size, xy_scale, max_peaks, edges, sorted_edge_inds, n_nodes,
                      n_points, pafs_stride, max_edge_length, dist_penalty_weight, max_node_peaks, min_line_scores,
                      min_instance_peaks, max_instances, status=None):
    """find_peaks + PAFScorer.predict in two launches (sa_bottomup_postproc: NMS scan, then one workgroup per frame for sort /
    refine / score / match / group). -> dict of the fixed-shape device tensors of every stage."""
    B, H, W, Cc = cms.shape
    _, Hp, Wp, C2 = pafs.shape
    E = C2 // 2
    dev = cms.device
    f32, i32 = torch.float32, torch.int32
    NP, I, N = int(max_node_peaks), int(max_instances), int(n_nodes)
    o = {"peak_xy": torch.empty((B, max_peaks, 2), dtype=f32, device=dev), "peak_val": torch.empty((B, max_peaks), dtype=f32, device=dev),
         "peak_chan": torch.empty((B, max_peaks), dtype=i32, device=dev), "peak_count": torch.empty((B,), dtype=i32, device=dev),
         "node_count": torch.empty((B, N), dtype=i32, device=dev), "node_peaks": torch.empty((B, N, NP), dtype=i32, device=dev),
         "line_scores": torch.full((B, E, NP, NP), float("nan"), dtype=f32, device=dev),
         "match_dst": torch.empty((B, E, NP), dtype=i32, device=dev), "match_score": torch.empty((B, E, NP), dtype=f32, device=dev),
         "instance_peaks": torch.empty((B, I, N, 2), dtype=f32, device=dev), "instance_peak_vals": torch.empty((B, I, N), dtype=f32, device=dev),
         "instance_scores": torch.empty((B, I), dtype=f32, device=dev), "n_instances": torch.empty((B,), dtype=i32, device=dev)}
    if status is None:
        status = torch.zeros((B,), dtype=i32, device=dev)
    o["status"] = status
    h = _lib.lib()
    key = (B, max_peaks, E, N, NP, str(dev), int(torch.cuda.current_stream().cuda_stream))
    ws = _PP_WS.get(key)
    if ws is None:
        while len(_PP_WS) >= _PP_WS_MAX:
            _PP_WS.popitem(last=False)  # the least recently used entry only
        ws = _PP_WS[key] = torch.zeros((h.sa_bottomup_postproc_workspace(B, max_peaks, E, N, NP),), dtype=torch.uint8, device=dev)
    else:
        _PP_WS.move_to_end(key)
    mode = REFINE["offsets"] if offsets is not None else REFINE[refinement]
    try:
        _bottomup_postproc_call(h, cms, offsets, B, H, W, Cc, threshold, mode, patch_size, xy_scale, max_peaks, pafs, Hp, Wp, E,
                                edges, sorted_edge_inds, N, n_points, pafs_stride, max_edge_length, dist_penalty_weight, NP,
                                min_line_scores, min_instance_peaks, I, o, status, ws)
    except Exception:
        _PP_WS.pop(key, None)  # its counters may be left non-zero: never reuse it
        raise
    return o


def _bottomup_postproc_call(h, cms, offsets, B, H, W, Cc, threshold, mode, patch_size, xy_scale, max_peaks, pafs, Hp, Wp, E, edges,
                            sorted_edge_inds, N, n_points, pafs_stride, max_edge_length, dist_penalty_weight, NP, min_line_scores,
                            min_instance_peaks, I, o, status, ws):
    check(h.sa_bottomup_postproc(
        _ptr(cms), _ptr(offsets), B, H, W, Cc, float(threshold), mode, int(patch_size), float(xy_scale), int(max_peaks), _ptr(pafs),
        Hp, Wp, E, _ptr(edges), _ptr(sorted_edge_inds), int(sorted_edge_inds.numel()), N, int(n_points), float(pafs_stride),
        float(max_edge_length), float(dist_penalty_weight), NP, float(min_line_scores), int(min_instance_peaks), I,
        _ptr(o["peak_xy"]), _ptr(o["peak_val"]), _ptr(o["peak_chan"]), _ptr(o["peak_count"]), _ptr(o["node_count"]),
        _ptr(o["node_peaks"]), _ptr(o["line_scores"]), _ptr(o["match_dst"]), _ptr(o["match_score"]), _ptr(o["instance_peaks"]),
        _ptr(o["instance_peak_vals"]), _ptr(o["instance_scores"]), _ptr(o["n_instances"]), _ptr(status), _ptr(ws), ws.numel(),
        _stream()), "sa_bottomup_postproc")


def lsa_host(cost, wave: bool = False):
    """Host Hungarian solve with the library's code (scipy.optimize.linear_sum_assignment semantics). `wave=True` runs the
    wave-cooperative form the matching kernel uses, its 64 lanes emulated on the host."""
    cost = np.ascontiguousarray(np.asarray(cost, dtype=np.float64))
    nr, nc = cost.shape
    n = min(nr, nc)
    rows = np.zeros((max(n, 1),), np.int64)
    cols = np.zeros((max(n, 1),), np.int64)
    fn = _lib.lib().sa_lsa_host_wave if wave else _lib.lib().sa_lsa_host
    rc = fn(cost.ctypes.data_as(C.c_void_p), nr, nc, rows.ctypes.data_as(C.c_void_p), cols.ctypes.data_as(C.c_void_p))
    if rc < 0:
        raise ValueError("cost matrix is infeasible")
    return rows[:rc], cols[:rc]


# --------------------------------------------------------------------------------------------------
# network layers (bf16 NHWC, channels padded to 16)
# --------------------------------------------------------------------------------------------------
def pad16(c):
    return (c + 15) // 16 * 16


# 16-bit storage type of the network kernels <-> the torch dtype of the buffers that hold it (sleap_amd/_lib.py: lib(dtype))
TORCH_DTYPE = {"bf16": torch.bfloat16, "fp16": torch.float16}


def _dtype_of(t) -> str:
    return "fp16" if t.dtype == torch.float16 else "bf16"


def to_bf16_padded(x_f32, dtype: str = None):
    """(B,H,W,C) f32 CUDA -> (B,H,W,pad16(C)) 16-bit CUDA tensor of the storage type `dtype` ("bf16" / "fp16"; None: the
    default, _lib.DEFAULT_DTYPE), zero padded. (The entry points keep their `_bf16` names in both library builds.)"""
    dtype = dtype or _lib.DEFAULT_DTYPE
    B, H, W, Cc = x_f32.shape
    out = torch.empty((B, H, W, pad16(Cc)), dtype=TORCH_DTYPE[dtype], device=x_f32.device)
    check(_lib.lib(dtype).sa_f32_to_bf16_padded(_ptr(x_f32), B * H * W, Cc, pad16(Cc), _ptr(out), _stream()), "sa_f32_to_bf16_padded")
    return out


def from_bf16(x_bf16, c):
    B, H, W, CP = x_bf16.shape
    out = torch.empty((B, H, W, c), dtype=torch.float32, device=x_bf16.device)
    check(_lib.lib(_dtype_of(x_bf16)).sa_bf16_to_f32(_ptr(x_bf16), B * H * W, CP, c, _ptr(out), _stream()), "sa_bf16_to_f32")
    return out


def to_planes16(x):
    """[B,H,W,CP] 16-bit tensor -> the same values as 16-channel planes [B,CP/16,H,W,16], returned in the NHWC SHAPE (the
    kernels take raw pointers; only the byte order differs)."""
    b, h, w, c = x.shape
    return x.reshape(b, h, w, c // 16, 16).permute(0, 3, 1, 2, 4).contiguous().view(b, h, w, c)


def from_planes16(x):
    """inverse of to_planes16"""
    b, h, w, c = x.shape
    return x.reshape(b, c // 16, h, w, 16).permute(0, 2, 3, 1, 4).reshape(b, h, w, c)


def pack_conv3x3_weights(kernel, c0, c1=0, dtype: str = None):
    """Keras (3,3,c0+c1,cout) f32 numpy -> packed 16-bit MFMA fragments (as int16 CUDA tensor) for the `dtype` library."""
    kernel = np.ascontiguousarray(kernel, dtype=np.float32)
    cout = kernel.shape[3]
    c0p, c1p, coutp = pad16(c0), (pad16(c1) if c1 else 0), pad16(cout)
    h = _lib.lib(dtype)
    n = h.sa_conv3x3_packed_elems(c0p, c1p, coutp)
    packed = np.zeros((n,), np.uint16)
    check(h.sa_pack_conv3x3_weights(kernel.ctypes.data_as(C.c_void_p), c0, c0p, c1, c1p, cout, coutp,
                                    packed.ctypes.data_as(C.c_void_p)), "sa_pack_conv3x3_weights")
    return torch.from_numpy(packed.view(np.int16)).cuda()


def conv3x3(src0, src1, mode, packed_w, bias_padded, coutp, relu, out_hw, full=True, pooled=False, out=None, out_pool=None):
    """sa_conv3x3_bf16. `mode` may carry _lib.LAYOUT_PLANES16: sources and outputs are then 16-channel planes
    ([B,CP/16,H,W,16] bytes held in tensors of the NHWC shape; to_planes16 / from_planes16 convert). `out` / `out_pool`:
    caller-owned output tensors (tests poison them first: a tile the launch skipped must not look computed)."""
    B = src0.shape[0]
    H, W = out_hw
    if full and out is None:
        out = torch.empty((B, H, W, coutp), dtype=src0.dtype, device=src0.device)
    outp = out_pool
    if pooled and outp is None:
        outp = torch.empty((B, H // 2, W // 2, coutp), dtype=src0.dtype, device=src0.device)
    if not full:
        out = None
    if not pooled:
        outp = None
    check(_lib.lib(_dtype_of(src0)).sa_conv3x3_bf16(_ptr(src0), src0.shape[3], _ptr(src1), src1.shape[3] if src1 is not None else 0,
                                     mode, _ptr(packed_w), _ptr(bias_padded), coutp, int(relu), B, H, W, _ptr(out),
                                     _ptr(outp), _stream()), "sa_conv3x3_bf16")
    if full and pooled:
        return out, outp
    return outp if pooled else out


# --------------------------------------------------------------------------------------------------
# sparse pyramidal Lucas-Kanade optical flow (the flow tracker's cv2.calcOpticalFlowPyrLK; csrc/flow.hip)
# --------------------------------------------------------------------------------------------------
class FlowPyramid:
    """One frame's image pyramid + Scharr derivatives on the device (sa_flow_pyramid_build). `image`: (H, W), (H, W, 1) or
    (H, W, 3) uint8, numpy or CUDA tensor."""

    def __init__(self, image, win: int = 21, max_level: int = 3, img_scale: float = 1.0):
        """`img_scale != 1`: the (gray) frame passes through the device restatement of cv2.resize(img, None, None, s, s) first
        (FlowCandidateMaker.img_scale, tracking.py:311-314); H, W are then the scaled size."""
        if not torch.is_tensor(image):
            image = torch.from_numpy(np.ascontiguousarray(image))
        if image.dtype != torch.uint8:
            raise TypeError("FlowPyramid: uint8 frames only")
        if image.dim() == 2:
            image = image[..., None]
        image = image.contiguous().to(_dev())
        src_h, src_w, self.C = (int(v) for v in image.shape)
        self.win, self.max_level, self.img_scale = int(win), int(max_level), float(img_scale)
        h = _lib.lib()
        hs, ws = C.c_int(src_h), C.c_int(src_w)
        if self.img_scale != 1.0:
            check(h.sa_flow_scaled_size(src_h, src_w, self.img_scale, C.byref(hs), C.byref(ws)), "sa_flow_scaled_size")
        self.H, self.W = hs.value, ws.value
        self.n_levels = h.sa_flow_pyramid_levels(self.H, self.W, self.win, self.max_level)
        self.buf = torch.empty((max(h.sa_flow_pyramid_bytes(self.H, self.W, self.win, self.max_level), 256),), dtype=torch.uint8,
                               device=image.device)
        if self.img_scale != 1.0:
            check(h.sa_flow_pyramid_build_scaled(_ptr(image), 1, src_h, src_w, self.C, self.img_scale, self.win, self.max_level,
                                                 _ptr(self.buf), None, _stream()), "sa_flow_pyramid_build_scaled")
        else:
            check(h.sa_flow_pyramid_build(_ptr(image), self.H, self.W, self.C, self.win, self.max_level, _ptr(self.buf), _stream()),
                  "sa_flow_pyramid_build")


def optical_flow_pyr_lk(prev: "FlowPyramid", nxt: "FlowPyramid", points, max_count: int = 30, epsilon: float = 0.01):
    """cv2.calcOpticalFlowPyrLK(prev, next, points) with the window / level count of the pyramids -> (next_points (n, 2) f32,
    status (n,) uint8, err (n,) f32) as CUDA tensors. `prev` may also be a list of pyramids, one per point."""
    pts = torch.as_tensor(np.asarray(points, np.float32) if not torch.is_tensor(points) else points, dtype=torch.float32,
                          device=nxt.buf.device).reshape(-1, 2).contiguous()
    n = pts.shape[0]
    prevs = prev if isinstance(prev, (list, tuple)) else [prev] * n
    assert len(prevs) == n and all((q.H, q.W, q.win, q.max_level) == (nxt.H, nxt.W, nxt.win, nxt.max_level) for q in prevs)
    ptrs = torch.tensor([q.buf.data_ptr() for q in prevs], dtype=torch.int64, device=pts.device)
    out = torch.empty_like(pts)
    status = torch.empty((n,), dtype=torch.uint8, device=pts.device)
    err = torch.empty((n,), dtype=torch.float32, device=pts.device)
    check(_lib.lib().sa_flow_lk(_ptr(ptrs), _ptr(nxt.buf), nxt.H, nxt.W, nxt.win, nxt.max_level, n, _ptr(pts), _ptr(out),
                                _ptr(status), _ptr(err), int(max_count), float(epsilon), _stream()), "sa_flow_lk")
    return out, status, err
