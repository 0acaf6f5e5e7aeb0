"""NumPy float32 restatement of `sleap/nn/peak_finding.py` (TEST INFRASTRUCTURE ONLY).

All arithmetic is done in float32 with one rounding per operation (no fused
multiply-add), mirroring the op-by-op evaluation of the TensorFlow CPU kernels the
reference lowers to. Coordinates are (x, y); confidence maps are (samples, height, width,
channels) float32.
"""
import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------------------
# helpers restating sleap/nn/data/instance_cropping.py
# --------------------------------------------------------------------------------------
def make_centered_bboxes(centroids, box_height, box_width):
    """instance_cropping.py:124-166 -- `[y - (h-1)/2, x - (w-1)/2, y + (h-1)/2, x + (w-1)/2]`."""
    centroids = np.asarray(centroids, dtype=F32).reshape(-1, 2)
    delta = (
        np.array([[-box_height + 1, -box_width + 1, box_height - 1, box_width - 1]], F32)
        * F32(0.5)
    )
    return centroids[:, [1, 0, 1, 0]] + delta


def normalize_bboxes(bboxes, image_height, image_width):
    """instance_cropping.py:58-87 -- divide by (H-1, W-1, H-1, W-1) in float32."""
    factor = np.array([[image_height, image_width, image_height, image_width]], F32) - F32(1)
    return np.asarray(bboxes, F32) / factor


def crop_and_resize_bilinear(images, boxes, box_indices, crop_size):
    """`tf.image.crop_and_resize(method="bilinear", extrapolation_value=0)` CPU semantics.

    Call site: peak_finding.py:180-186. Restates the documented kernel arithmetic:
    `scale = (y2 - y1) * (H - 1) / (crop_h - 1)`, `in_y = y1 * (H - 1) + y * scale`, rows /
    columns with `in < 0` or `in > size - 1` are filled with 0, otherwise
    `top + (bottom - top) * y_lerp` with `top = tl + (tr - tl) * x_lerp`.
    """
    images = np.asarray(images, F32)
    boxes = np.asarray(boxes, F32).reshape(-1, 4)
    box_indices = np.asarray(box_indices, np.int64).reshape(-1)
    n = boxes.shape[0]
    ch, cw = int(crop_size[0]), int(crop_size[1])
    _, H, W, D = images.shape
    out = np.zeros((n, ch, cw, D), F32)
    if n == 0:
        return out
    y1, x1, y2, x2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    Hm1, Wm1 = F32(H - 1), F32(W - 1)
    if ch > 1:
        hscale = ((y2 - y1) * Hm1) / F32(ch - 1)
    else:
        hscale = np.zeros(n, F32)
    if cw > 1:
        wscale = ((x2 - x1) * Wm1) / F32(cw - 1)
    else:
        wscale = np.zeros(n, F32)
    for yy in range(ch):
        if ch > 1:
            in_y = y1 * Hm1 + F32(yy) * hscale
        else:
            in_y = F32(0.5) * (y1 + y2) * Hm1
        y_ok = ~((in_y < 0) | (in_y > Hm1))
        top = np.floor(in_y)
        bot = np.ceil(in_y)
        y_lerp = (in_y - top).astype(F32)
        ti = np.clip(top, 0, H - 1).astype(np.int64)
        bi = np.clip(bot, 0, H - 1).astype(np.int64)
        for xx in range(cw):
            if cw > 1:
                in_x = x1 * Wm1 + F32(xx) * wscale
            else:
                in_x = F32(0.5) * (x1 + x2) * Wm1
            x_ok = ~((in_x < 0) | (in_x > Wm1))
            left = np.floor(in_x)
            right = np.ceil(in_x)
            x_lerp = (in_x - left).astype(F32)[:, None]
            li = np.clip(left, 0, W - 1).astype(np.int64)
            ri = np.clip(right, 0, W - 1).astype(np.int64)
            tl = images[box_indices, ti, li]
            tr = images[box_indices, ti, ri]
            bl = images[box_indices, bi, li]
            br = images[box_indices, bi, ri]
            t = tl + (tr - tl) * x_lerp
            b = bl + (br - bl) * x_lerp
            v = t + (b - t) * y_lerp[:, None]
            ok = (y_ok & x_ok)[:, None]
            out[:, yy, xx, :] = np.where(ok, v, F32(0))
    return out


def crop_bboxes(images, bboxes, sample_inds):
    """peak_finding.py:135-190."""
    bboxes = np.asarray(bboxes, F32).reshape(-1, 4)
    if bboxes.shape[0] == 0:
        return np.zeros((0, 0, 0, images.shape[-1]), F32)
    y1x1 = bboxes[0, 0:2]
    y2x2 = bboxes[0, 2:4]
    # tf.math.round == round-half-to-even == np.rint
    box_size = np.rint((y2x2 - y1x1) + F32(1)).astype(np.int32)
    H, W = images.shape[1], images.shape[2]
    nb = normalize_bboxes(bboxes, H, W)
    return crop_and_resize_bilinear(images, nb, sample_inds, box_size)


# --------------------------------------------------------------------------------------
# peak_finding.py
# --------------------------------------------------------------------------------------
def find_offsets_local_direction(centered_patches, delta=0.25):
    """peak_finding.py:78-132 -- `sign(right-left), sign(bottom-top)` times delta."""
    p = np.asarray(centered_patches)
    dx = p[:, 1, 2, :] - p[:, 1, 0, :]
    dy = p[:, 2, 1, :] - p[:, 0, 1, :]
    return (np.sign(np.stack([dx, dy], axis=1)[..., 0]) * delta).astype(p.dtype)


def integral_regression(cms, xv, yv):
    """peak_finding.py:311-334 -- `sum(x*p)/sum(p)`; zero mass gives NaN."""
    cms = np.asarray(cms, F32)
    xv = np.asarray(xv, F32)
    yv = np.asarray(yv, F32)
    with np.errstate(invalid="ignore", divide="ignore"):
        z = cms.sum(axis=(1, 2), dtype=F32)
        x_hat = (xv.reshape(1, 1, -1, 1) * cms).sum(axis=(1, 2), dtype=F32) / z
        y_hat = (yv.reshape(1, -1, 1, 1) * cms).sum(axis=(1, 2), dtype=F32) / z
    return x_hat.astype(F32), y_hat.astype(F32)


def find_global_peaks_rough(cms, threshold=0.1):
    """peak_finding.py:193-246 -- row argmax and column argmax are taken INDEPENDENTLY."""
    cms = np.asarray(cms, F32)
    B, H, W, C = cms.shape
    max_img_rows = cms.max(axis=2)  # (B, H, C)
    argmax_rows = max_img_rows.argmax(axis=1).reshape(-1)  # (B*C,)
    max_img_cols = cms.max(axis=1)  # (B, W, C)
    argmax_cols = max_img_cols.argmax(axis=1).reshape(-1)
    total = argmax_cols.shape[0]
    sample_subs = np.arange(total) // C
    channel_subs = np.arange(total) % C
    peak_vals = cms[sample_subs, argmax_rows, argmax_cols, channel_subs]
    peak_points = np.stack([argmax_cols, argmax_rows], axis=-1).astype(F32).reshape(-1, C, 2)
    peak_vals = peak_vals.reshape(-1, C)
    peak_points = np.where(peak_vals[..., None] < F32(threshold), F32(np.nan), peak_points)
    return peak_points.astype(F32), peak_vals.astype(F32)


def nms_mask(cms, threshold):
    """The mask of peak_finding.py:274-290: `(cms > max(8 nbrs, centre - 1)) & (cms > thr)`.

    `tf.nn.dilation2d` with kernel [[0,0,0],[0,-1,0],[0,0,0]] and SAME padding ignores
    out-of-bounds taps.
    """
    cms = np.asarray(cms, F32)
    B, H, W, C = cms.shape
    pad = np.full((B, H + 2, W + 2, C), -np.inf, F32)
    pad[:, 1:-1, 1:-1] = cms
    max_img = cms - F32(1)
    for dy in range(3):
        for dx in range(3):
            if dy == 1 and dx == 1:
                continue
            max_img = np.maximum(max_img, pad[:, dy : dy + H, dx : dx + W])
    return (cms > max_img) & (cms > F32(threshold))


def find_local_peaks_rough(cms, threshold=0.2):
    """peak_finding.py:249-308 -- peaks ordered row-major over (sample, y, x, channel)."""
    cms = np.asarray(cms, F32)
    subs = np.argwhere(nms_mask(cms, threshold))  # row-major == tf.where
    peak_vals = cms[subs[:, 0], subs[:, 1], subs[:, 2], subs[:, 3]]
    peak_points = subs[:, [2, 1]].astype(F32)
    return (
        peak_points,
        peak_vals.astype(F32),
        subs[:, 0].astype(np.int32),
        subs[:, 3].astype(np.int32),
    )


def _refine(cms_flat, rough, box_sample_inds, refinement, integral_patch_size):
    crop_size = integral_patch_size if refinement == "integral" else 3
    bboxes = make_centered_bboxes(rough, crop_size, crop_size)
    crops = crop_bboxes(cms_flat, bboxes, box_sample_inds)
    if refinement == "integral":
        gv = np.arange(crop_size, dtype=F32) - F32((crop_size - 1) / 2)
        dx, dy = integral_regression(crops, gv, gv)
        return np.concatenate([dx, dy], axis=1).astype(F32)
    return find_offsets_local_direction(crops, 0.25).astype(F32)


def find_global_peaks(cms, threshold=0.2, refinement=None, integral_patch_size=5):
    """peak_finding.py:337-420."""
    cms = np.asarray(cms, F32)
    rough, vals = find_global_peaks_rough(cms, threshold)
    if refinement not in ("integral", "local") or np.all(np.isnan(rough)):
        return rough, vals
    B, H, W, C = cms.shape
    flat = rough.reshape(B * C, 2).copy()
    valid_idx = np.nonzero(~np.isnan(flat[:, 0]))[0]
    cms_flat = cms.transpose(0, 3, 1, 2).reshape(B * C, H, W, 1)
    offsets = _refine(cms_flat, flat[valid_idx], valid_idx, refinement, integral_patch_size)
    flat[valid_idx] = flat[valid_idx] + offsets
    return flat.reshape(B, C, 2), vals


def find_local_peaks(cms, threshold=0.2, refinement=None, integral_patch_size=5):
    """peak_finding.py:451-532."""
    cms = np.asarray(cms, F32)
    rough, vals, sample_inds, channel_inds = find_local_peaks_rough(cms, threshold)
    if rough.shape[0] == 0 or refinement not in ("integral", "local"):
        return rough, vals, sample_inds, channel_inds
    B, H, W, C = cms.shape
    cms_flat = cms.transpose(0, 3, 1, 2).reshape(B * C, H, W, 1)
    box_sample_inds = sample_inds.astype(np.int64) * C + channel_inds
    offsets = _refine(cms_flat, rough, box_sample_inds, refinement, integral_patch_size)
    return (rough + offsets).astype(F32), vals, sample_inds, channel_inds


def find_global_peaks_with_offsets(cms, offsets, threshold=0.2):
    """peak_finding.py:566-643 -- learned offsets (grid units) gathered at the rough peak."""
    cms = np.asarray(cms, F32)
    offsets = np.asarray(offsets, F32)
    rough, vals = find_global_peaks_rough(cms, threshold)
    if np.all(np.isnan(rough)):
        return rough, vals
    B, H, W, C = cms.shape
    flat = rough.reshape(B * C, 2).copy()
    valid_idx = np.nonzero(~np.isnan(flat[:, 0]))[0]
    off5 = offsets.reshape(B, H, W, -1, 2)
    b = valid_idx // C
    c = valid_idx % C
    x = flat[valid_idx, 0].astype(np.int32)
    y = flat[valid_idx, 1].astype(np.int32)
    flat[valid_idx] = flat[valid_idx] + off5[b, y, x, c]
    return flat.reshape(B, C, 2), vals


def find_local_peaks_with_offsets(cms, offsets, threshold=0.2):
    """peak_finding.py:646-707."""
    cms = np.asarray(cms, F32)
    offsets = np.asarray(offsets, F32)
    rough, vals, sample_inds, channel_inds = find_local_peaks_rough(cms, threshold)
    if rough.shape[0] == 0:
        return rough, vals, sample_inds, channel_inds
    B, H, W, C = cms.shape
    off5 = offsets.reshape(B, H, W, -1, 2)
    x = rough[:, 0].astype(np.int32)
    y = rough[:, 1].astype(np.int32)
    refined = rough + off5[sample_inds, y, x, channel_inds]
    return refined.astype(F32), vals, sample_inds, channel_inds
