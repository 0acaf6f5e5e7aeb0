"""CPU restatement of the sparse pyramidal Lucas-Kanade optical flow that SLEAP's flow tracker calls (TEST INFRASTRUCTURE
ONLY -- see oracle/__init__.py).

The reference's call site is `FlowCandidateMaker.flow_shift_instances` (sleap/nn/tracking.py:258-356):

    cv2.calcOpticalFlowPyrLK(ref_img, new_img, pts, None, winSize=(w, w), maxLevel=L,
                             criteria=(cv2.TERM_CRITERIA_EPS | cv2.TERM_CRITERIA_COUNT, 30, 0.01))

The algorithm lives in a third-party dependency that is ABSENT here: OpenCV (`opencv-python>=4.2.0,<=4.7.0`,
pypi_requirements.txt:16; conda `opencv <4.9.0`, environment.yml:20). This module restates what that function is published
to do (J.-Y. Bouguet, "Pyramidal implementation of the Lucas Kanade feature tracker", 2000; OpenCV 4.x
modules/video/src/lkpyramid.cpp, scalar code path):

  * `buildOpticalFlowPyramid`: level 0 = the uint8 image, level l+1 = `pyrDown` of level l (5x5 binomial [1 4 6 4 1]^2 / 256
    in integer arithmetic with rounding, BORDER_REFLECT_101, size (n+1)//2); levels stop before an image would be <= the
    window. Pixels outside a level are BORDER_REFLECT_101 (the pyramid is built with a window-wide border).
  * `calcSharrDeriv`: Ix = [3 10 3]^T (x) [-1 0 1], Iy = [-1 0 1]^T (x) [3 10 3] on the uint8 level (int16, un-normalised,
    BORDER_REFLECT_101 at the image edge); OUTSIDE the level the derivative is 0 (BORDER_CONSTANT).
  * per point, from the coarsest level down: the window patch of I and its derivatives at the sub-pixel position by integer
    bilinear weights (14 bits; patch values x 32, int16), the 2x2 structure matrix A (float32 sums x 2^-20), rejection by
    minEig < 1e-4 or det < FLT_EPSILON, then <= 30 Newton steps delta = A^-1 b with b from the bilinear patch of J, stopping at
    |delta|^2 <= 0.01^2 or when two consecutive steps cancel (|delta + prev| < 0.01 per axis: the point moves back half a step);
    `err` = mean |J - I| over the window / 32 at level 0 (0 for points that leave the image at level 0).
  * status = 0 for points that leave the image or fail the eigenvalue test AT LEVEL 0; a failure at a coarser level only skips
    that level.

**Parity unpinned**: no OpenCV build can be run here and the reference's own tests only smoke-test the flow trackers
(tests/nn/test_tracker_components.py:37-60, on an H.264 video that cannot be decoded offline), so there is no reference-held
vector for this function. What pins it instead (tests/test_oracle_optical_flow.py): analytic cases (pure translations of a
smooth texture are recovered to a few hundredths of a pixel at every pyramid depth, out-of-image and untextured points report
status 0, NaN points report status 0 as `cvFloor(NaN)` = INT_MIN does). OpenCV's SIMD builds sum the window in a different
order than its scalar code (float32 lane sums), so even two OpenCV builds agree only to ~1e-4 px; the scalar order is used here.
Also restated: `cv2.cvtColor(img, cv2.COLOR_BGR2GRAY)` for 3-channel frames (fixed point: (B 1868 + G 9617 + R 4899 + 8192) >> 14,
applied to whatever channel order the caller has, as the reference does to its RGB frames) and `ensure_int`
(sleap/nn/data/normalization.py:52-77), and `cv2.resize(img, None, None, scale, scale)` for img_scale != 1 (tracking.py:311-314:
INTER_LINEAR on uint8 -- `cv_resize_linear_u8` below, OpenCV's 11-bit fixed-point form; parity unpinned like the rest).
"""
import numpy as np

W_BITS = 14
FLT_SCALE = np.float32(1.0 / (1 << 20))
FLT_EPSILON = np.float32(1.1920929e-07)
MIN_EIG_THRESHOLD = np.float32(1e-4)


def ensure_int(img):
    """normalization.py:52-77: float images in [0, 1] -> uint8 via tf.image.convert_image_dtype (x 255.5 truncation: TF scales
    by (max + 0.5) and casts, saturating), other floats -> truncating cast; integers unchanged."""
    img = np.asarray(img)
    if img.dtype.kind == "f":
        if img.size and img.max() <= 1.0:
            return np.clip(np.floor(img.astype(np.float32) * np.float32(255.5)), 0, 255).astype(np.uint8)
        return img.astype(np.uint8)
    return img


def to_gray(img):
    """The frame as a rank-2 uint8 array, as flow_shift_instances prepares it (tracking.py:286-301)."""
    img = ensure_int(img)
    if img.ndim > 3:
        img = np.squeeze(img)
    if img.ndim == 3 and img.shape[-1] == 1:
        img = img[..., 0]
    if img.ndim == 3 and img.shape[-1] == 3:
        c = img.astype(np.int64)
        img = ((c[..., 0] * 1868 + c[..., 1] * 9617 + c[..., 2] * 4899 + 8192) >> 14).astype(np.uint8)
    if img.ndim != 2:
        raise ValueError(f"unsupported image shape {img.shape}")
    return np.ascontiguousarray(img)


def _reflect101(i, n):
    """BORDER_REFLECT_101 index (gfedcb|abcdefgh|gfedcba), for any offset."""
    if n == 1:
        return np.zeros_like(i)
    p = 2 * (n - 1)
    i = np.mod(i, p)
    return np.where(i >= n, p - i, i)


def pyr_down(img):
    """cv2.pyrDown on uint8: out(x, y) = (sum_ij k_i k_j in(2x + i - 2, 2y + j - 2) + 128) >> 8, k = [1 4 6 4 1]."""
    h, w = img.shape
    oh, ow = (h + 1) // 2, (w + 1) // 2
    k = np.array([1, 4, 6, 4, 1], np.int64)
    ys = _reflect101(2 * np.arange(oh)[:, None] + np.arange(-2, 3)[None, :], h)  # (oh, 5)
    xs = _reflect101(2 * np.arange(ow)[:, None] + np.arange(-2, 3)[None, :], w)  # (ow, 5)
    src = img.astype(np.int64)
    rows = (src[ys] * k[None, :, None]).sum(axis=1)          # (oh, w): vertical pass
    out = (rows[:, xs] * k[None, None, :]).sum(axis=2)       # (oh, ow)
    return ((out + 128) >> 8).astype(np.uint8)


def scharr_deriv(img):
    """calcSharrDeriv -> (Ix, Iy) int16, BORDER_REFLECT_101 at the edges of the image."""
    h, w = img.shape
    s = img.astype(np.int32)
    up, dn = s[_reflect101(np.arange(h) - 1, h)], s[_reflect101(np.arange(h) + 1, h)]
    t0 = (up + dn) * 3 + s * 10   # vertical smoothing
    t1 = dn - up                  # vertical difference
    xl, xr = _reflect101(np.arange(w) - 1, w), _reflect101(np.arange(w) + 1, w)
    ix = t0[:, xr] - t0[:, xl]
    iy = (t1[:, xr] + t1[:, xl]) * 3 + t1 * 10
    return ix.astype(np.int16), iy.astype(np.int16)


def build_pyramid(img, win, max_level):
    """buildOpticalFlowPyramid: list of uint8 levels; fewer than max_level + 1 when the next level would be <= the window."""
    levels = [img]
    h, w = img.shape
    for _ in range(max_level):
        h, w = (h + 1) // 2, (w + 1) // 2
        if w <= win or h <= win:
            break
        levels.append(pyr_down(levels[-1]))
    return levels


def _cv_round(x):
    return int(np.rint(x))  # cvRound: round half to even


def _weights(a, b):
    one = np.float32(1.0)
    s = np.float32(1 << W_BITS)
    iw00 = _cv_round((one - a) * (one - b) * s)
    iw01 = _cv_round(a * (one - b) * s)
    iw10 = _cv_round((one - a) * b * s)
    return iw00, iw01, iw10, (1 << W_BITS) - iw00 - iw01 - iw10


def _gather_img(img, x0, y0, win):
    """(win + 1, win + 1) int64 block of `img` starting at (x0, y0), BORDER_REFLECT_101 outside."""
    h, w = img.shape
    ys = _reflect101(y0 + np.arange(win + 1), h)
    xs = _reflect101(x0 + np.arange(win + 1), w)
    return img[np.ix_(ys, xs)].astype(np.int64)


def _gather_deriv(d, x0, y0, win):
    """the same for a derivative plane, 0 outside the level"""
    h, w = d.shape
    ys, xs = y0 + np.arange(win + 1), x0 + np.arange(win + 1)
    oky, okx = (ys >= 0) & (ys < h), (xs >= 0) & (xs < w)
    blk = d[np.ix_(np.clip(ys, 0, h - 1), np.clip(xs, 0, w - 1))].astype(np.int64)
    return blk * (oky[:, None] & okx[None, :])


def _bilinear(blk, iw, shift):
    iw00, iw01, iw10, iw11 = iw
    v = blk[:-1, :-1] * iw00 + blk[:-1, 1:] * iw01 + blk[1:, :-1] * iw10 + blk[1:, 1:] * iw11
    return (v + (1 << (shift - 1))) >> shift  # CV_DESCALE


def _f32_sum(v):
    """float32 accumulation in row-major order (the scalar code path: `iA11 += (float)(ix * ix)`)."""
    acc = np.float32(0.0)
    for x in v.reshape(-1).astype(np.float32):
        acc = np.float32(acc + x)
    return acc


def _floor(v):
    """cvFloor; NaN / inf -> INT_MIN like the cvtsd2si / cast it compiles to"""
    return int(np.floor(v)) if np.isfinite(v) else -(1 << 31)


def calc_optical_flow_pyr_lk(prev_img, next_img, prev_pts, win=21, max_level=3, max_count=30, epsilon=0.01):
    """-> (next_pts (n, 2) float32, status (n,) uint8, err (n,) float32)."""
    prev_img, next_img = np.ascontiguousarray(prev_img), np.ascontiguousarray(next_img)
    assert prev_img.dtype == np.uint8 and prev_img.ndim == 2 and prev_img.shape == next_img.shape
    pts = np.asarray(prev_pts, np.float32).reshape(-1, 2)
    n = len(pts)
    max_count = min(max(int(max_count), 0), 100)
    eps2 = np.float32(min(max(float(epsilon), 0.0), 10.0)) ** 2
    pyr_i = build_pyramid(prev_img, win, max_level)
    pyr_j = build_pyramid(next_img, win, max_level)
    top = len(pyr_i) - 1
    derivs = [scharr_deriv(p) for p in pyr_i]
    nxt = np.zeros((n, 2), np.float32)
    status = np.ones((n,), np.uint8)
    err = np.zeros((n,), np.float32)
    half = np.float32((win - 1) * 0.5)
    for level in range(top, -1, -1):
        I, J = pyr_i[level], pyr_j[level]
        dx, dy = derivs[level]
        h, w = I.shape
        for k in range(n):
            prev = pts[k] * np.float32(1.0 / (1 << level))
            nextp = prev.copy() if level == top else nxt[k] * np.float32(2.0)
            nxt[k] = nextp
            prev = prev - half
            ipx, ipy = _floor(prev[0]), _floor(prev[1])
            if ipx < -win or ipx >= w or ipy < -win or ipy >= h:
                if level == 0:
                    status[k], err[k] = 0, 0.0
                continue
            a, b = np.float32(prev[0] - np.float32(ipx)), np.float32(prev[1] - np.float32(ipy))
            iw = _weights(a, b)
            ipatch = _bilinear(_gather_img(I, ipx, ipy, win), iw, W_BITS - 5)
            ixp = _bilinear(_gather_deriv(dx, ipx, ipy, win), iw, W_BITS)
            iyp = _bilinear(_gather_deriv(dy, ipx, ipy, win), iw, W_BITS)
            a11 = _f32_sum(ixp * ixp) * FLT_SCALE
            a12 = _f32_sum(ixp * iyp) * FLT_SCALE
            a22 = _f32_sum(iyp * iyp) * FLT_SCALE
            det = np.float32(a11 * a22 - a12 * a12)
            min_eig = np.float32((a22 + a11 - np.sqrt(np.float32((a11 - a22) * (a11 - a22) + np.float32(4.0) * a12 * a12))) /
                                 np.float32(2 * win * win))
            if min_eig < MIN_EIG_THRESHOLD or det < FLT_EPSILON:
                if level == 0:
                    status[k] = 0
                continue
            det = np.float32(1.0) / det
            nextp = nextp - half
            prev_delta = np.zeros((2,), np.float32)
            for j in range(max_count):
                inx, iny = _floor(nextp[0]), _floor(nextp[1])
                if inx < -win or inx >= w or iny < -win or iny >= h:
                    if level == 0:
                        status[k] = 0
                    break
                a, b = np.float32(nextp[0] - np.float32(inx)), np.float32(nextp[1] - np.float32(iny))
                diff = _bilinear(_gather_img(J, inx, iny, win), _weights(a, b), W_BITS - 5) - ipatch
                b1 = _f32_sum(diff * ixp) * FLT_SCALE
                b2 = _f32_sum(diff * iyp) * FLT_SCALE
                delta = np.array([np.float32((a12 * b2 - a22 * b1) * det), np.float32((a12 * b1 - a11 * b2) * det)], np.float32)
                nextp = nextp + delta
                nxt[k] = nextp + half
                if np.float32(delta[0] * delta[0] + delta[1] * delta[1]) <= eps2:
                    break
                if j > 0 and abs(delta[0] + prev_delta[0]) < 0.01 and abs(delta[1] + prev_delta[1]) < 0.01:
                    nxt[k] = nxt[k] - delta * np.float32(0.5)
                    break
                prev_delta = delta
            if status[k] and level == 0:
                p = nxt[k] - half
                inx, iny = _floor(p[0]), _floor(p[1])
                if inx < -win or inx >= w or iny < -win or iny >= h:
                    status[k] = 0
                    continue
                a, b = np.float32(p[0] - np.float32(inx)), np.float32(p[1] - np.float32(iny))
                diff = _bilinear(_gather_img(J, inx, iny, win), _weights(a, b), W_BITS - 5) - ipatch
                err[k] = _f32_sum(np.abs(diff)) * np.float32(1.0 / (32 * win * win))
    return nxt, status, err


def cv_round(x):
    """cvRound / saturate_cast<int>(double): round half to even (lrint)."""
    return int(np.rint(x))


def cv_resize_linear_u8(img, fx, fy):
    """cv2.resize(img, None, None, fx, fy) (interpolation = INTER_LINEAR, the default) of a rank-2 uint8 image, as OpenCV's
    imgproc/resize.cpp computes it for 8-bit data (published algorithm, restated; OpenCV is absent: parity unpinned):

      dsize = (cvRound(W fx), cvRound(H fy)); scale = 1 / f (double)
      per destination column dx:  x = float((dx + 0.5) scale_x - 0.5); sx = floor(x); a = x - sx  (float32);
                                  sx < 0 -> (0, a = 0);  sx >= W - 1 -> (W - 1, a = 0)
                                  coefficients cvRound((1 - a) 2048), cvRound(a 2048)  (INTER_RESIZE_COEF_BITS = 11)
      per destination row dy:     the same for (sy, b) without the clipping of b; the two source rows are clip(sy), clip(sy + 1)
      horizontal pass (int32):    D_r[dx] = S_r[sx] a0 + S_r[sx + 1] a1          (S_r[sx] 2048 where sx + 1 is outside)
      vertical pass:              dst = (((b0 (D0 >> 4)) >> 16) + ((b1 (D1 >> 4)) >> 16) + 2) >> 2     (VResizeLinear, uchar)
    """
    img = np.asarray(img)
    assert img.ndim == 2 and img.dtype == np.uint8
    H, W = img.shape
    dw, dh = cv_round(W * fx), cv_round(H * fy)
    sx_scale, sy_scale = 1.0 / fx, 1.0 / fy

    def axis(n_dst, n_src, scale, clip_coef):
        d = np.arange(n_dst, dtype=np.float64)
        x = ((d + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(x).astype(np.int64)
        a = (x - s.astype(np.float32)).astype(np.float32)
        if clip_coef:
            lo, hi = s < 0, s >= n_src - 1
            a = np.where(lo | hi, np.float32(0), a)
            s = np.where(lo, 0, np.where(hi, n_src - 1, s))
        c0 = np.rint((np.float32(1) - a) * np.float32(2048)).astype(np.int64)
        c1 = np.rint(a * np.float32(2048)).astype(np.int64)
        return s, c0, c1

    sx, a0, a1 = axis(dw, W, sx_scale, True)
    sy, b0, b1 = axis(dh, H, sy_scale, False)
    src = img.astype(np.int64)
    sx1 = np.minimum(sx + 1, W - 1)
    a0 = np.where(sx + 1 >= W, 2048, a0)
    a1 = np.where(sx + 1 >= W, 0, a1)
    r0, r1 = np.clip(sy, 0, H - 1), np.clip(sy + 1, 0, H - 1)
    D0 = src[r0][:, sx] * a0[None, :] + src[r0][:, sx1] * a1[None, :]
    D1 = src[r1][:, sx] * a0[None, :] + src[r1][:, sx1] * a1[None, :]
    out = (((b0[:, None] * (D0 >> 4)) >> 16) + ((b1[:, None] * (D1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def flow_shift_points(ref_points, ref_img, new_img, min_shifted_points=0, scale=1.0, window_size=21, max_levels=3):
    """FlowCandidateMaker.flow_shift_instances (tracking.py:258-356) on point arrays: `ref_points` = list of (N, 2) arrays (one
    per reference instance). -> list of (index of the reference instance, shifted points with NaN where the flow was lost,
    shift_score = -mean error of the points found), only for instances with MORE than `min_shifted_points` points found."""
    ref_img, new_img = to_gray(ref_img), to_gray(new_img)
    if scale != 1:  # tracking.py:311-314
        ref_img, new_img = cv_resize_linear_u8(ref_img, scale, scale), cv_resize_linear_u8(new_img, scale, scale)
    if not ref_points:
        return []
    allp = np.concatenate([np.asarray(p, np.float64) for p in ref_points], axis=0).astype(np.float32)
    if scale != 1:
        allp = (allp * np.float32(scale)).astype(np.float32)  # `.astype("float32") * scale` (:321)
    shifted, status, errs = calc_optical_flow_pyr_lk(ref_img, new_img, allp, win=window_size, max_level=max_levels)
    if scale != 1:
        shifted = (shifted / np.float32(scale)).astype(np.float32)  # `shifted_pts /= scale` (:333)
    out, o = [], 0
    for i, p in enumerate(ref_points):
        m = len(p)
        pts, found, err = shifted[o:o + m].astype(np.float32).copy(), status[o:o + m].astype(bool), errs[o:o + m]
        o += m
        if found.sum() > min_shifted_points:
            pts[~found] = np.nan
            out.append((i, pts, -float(np.mean(err[found]))))
    return out
