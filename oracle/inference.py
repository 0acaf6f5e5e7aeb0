"""NumPy restatement of the post-network logic of the single-instance and top-down inference layers
(TEST INFRASTRUCTURE ONLY). Network outputs (confidence maps / offsets) are inputs here.

  single_instance_peaks   SingleInstanceInferenceLayer.call   sleap/nn/inference.py:1319-1380
  centroid_crop           CentroidCrop.call                   sleap/nn/inference.py:1747-1966
  find_instance_peaks     FindInstancePeaks.call              sleap/nn/inference.py:2059-2200
"""
import numpy as np

from . import peak_finding as pf

F32 = np.float32


def single_instance_peaks(cms, offsets, threshold, refinement, integral_patch_size, output_stride, input_scale):
    """-> instance_peaks (B, 1, N, 2), instance_peak_vals (B, 1, N)."""
    if offsets is None:
        peaks, vals = pf.find_global_peaks(cms, threshold, refinement, integral_patch_size)
    else:
        peaks, vals = pf.find_global_peaks_with_offsets(cms, offsets, threshold)
    peaks = peaks * F32(output_stride)
    if input_scale != 1.0:
        peaks = (peaks / F32(input_scale)) + F32(0.5)
    return peaks[:, None].astype(F32), vals[:, None].astype(F32)


def centroid_crop(full_imgs, cms, offsets, threshold, refinement, integral_patch_size, output_stride, input_scale,
                  crop_size, max_instances=None, precrop_resize=1.0, resize_fn=None):
    """-> dict(centroids, centroid_vals, crop_sample_inds, crops, crop_offsets) (flat, sample-major)."""
    if offsets is None:
        pts, vals, sinds, _ = pf.find_local_peaks(cms, threshold, refinement, integral_patch_size)
    else:
        pts, vals, sinds, _ = pf.find_local_peaks_with_offsets(cms, offsets, threshold)
    pts = pts * F32(output_stride)
    if input_scale != 1.0:
        pts = (pts / F32(input_scale)) + F32(0.5)
    if precrop_resize != 1.0:
        full_imgs = resize_fn(full_imgs, precrop_resize)
        pts = pts * F32(precrop_resize)
    B = cms.shape[0]
    if len(pts) > 0 and max_instances is not None:
        keep_p, keep_v, keep_s = [], [], []
        for b in range(B):
            sel = np.nonzero(sinds == b)[0]
            if max_instances < len(sel):
                # tf.math.top_k: descending by value, ties -> lower index first
                order = np.argsort(-vals[sel], kind="stable")[:max_instances]
                sel = sel[order]
            keep_p.append(pts[sel])
            keep_v.append(vals[sel])
            keep_s.append(np.full((len(sel),), b, np.int32))
        pts, vals, sinds = np.concatenate(keep_p), np.concatenate(keep_v), np.concatenate(keep_s)
    crop_offsets = pts - F32(crop_size / 2)
    if len(pts) > 0:
        bboxes = pf.make_centered_bboxes(pts, crop_size, crop_size)
        crops = pf.crop_bboxes(full_imgs.astype(F32), bboxes, sinds)
        if full_imgs.dtype == np.uint8:
            crops = crops.astype(np.uint8)  # tf.cast back to the image dtype (truncation)
    else:
        crops = np.zeros((0, crop_size, crop_size, full_imgs.shape[3]), full_imgs.dtype)
    return dict(centroids=pts.astype(F32), centroid_vals=vals.astype(F32), crop_sample_inds=sinds.astype(np.int32),
                crops=crops, crop_offsets=crop_offsets.astype(F32))


def find_instance_peaks(cms, offsets, crop_offsets, threshold, refinement, integral_patch_size, output_stride,
                        input_scale):
    """cms of the crops (n, h, w, N) -> peak_points (n, N, 2) in full-image coordinates, peak_vals (n, N)."""
    if offsets is None:
        peaks, vals = pf.find_global_peaks(cms, threshold, refinement, integral_patch_size)
    else:
        peaks, vals = pf.find_global_peaks_with_offsets(cms, offsets, threshold)
    peaks = peaks * F32(output_stride)
    if input_scale != 1.0:
        peaks = (peaks / F32(input_scale)) + F32(0.5)
    if crop_offsets is not None:
        peaks = peaks + (np.asarray(crop_offsets, F32)[:, None, :] / F32(input_scale))
    return peaks.astype(F32), vals.astype(F32)
