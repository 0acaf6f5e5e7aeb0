"""oracle/optical_flow.py (the restated cv2.calcOpticalFlowPyrLK of the flow tracker) against what can be known without OpenCV:
analytic motions, the integer conventions of its building blocks, the status rules. Parity with OpenCV itself is unpinned (the
module's header says why)."""
import numpy as np
import pytest

from oracle import optical_flow as of


def texture(h, w, dx=0.0, dy=0.0, seed=1, n=12):
    """a smooth synthetic texture sampled at (x - dx, y - dy): sub-pixel shifts are exact"""
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    x, y = x - dx, y - dy
    r = np.random.default_rng(seed)
    v = np.zeros((h, w))
    for _ in range(n):
        fx, fy, ph = r.uniform(-0.25, 0.25), r.uniform(-0.25, 0.25), r.uniform(0, 6.28)
        v += np.sin(fx * x + fy * y + ph)
    return np.round((v + n) / (2 * n) * 255).astype(np.uint8)


def test_building_blocks_follow_opencv_conventions():
    img = np.arange(35, dtype=np.uint8).reshape(5, 7) * 3
    # reflect-101: gfedcb|abcdefgh|gfedcba
    assert list(of._reflect101(np.array([-2, -1, 0, 6, 7, 8]), 7)) == [2, 1, 0, 6, 5, 4]
    d = of.pyr_down(img)
    assert d.shape == (3, 4) and d.dtype == np.uint8
    k = np.array([1, 4, 6, 4, 1])
    ys, xs = of._reflect101(np.arange(-2, 3) + 2, 5), of._reflect101(np.arange(-2, 3) + 4, 7)  # output (1, 2)
    want = (np.outer(k, k) * img[np.ix_(ys, xs)].astype(int)).sum()
    assert d[1, 2] == (want + 128) >> 8
    const = np.full((9, 11), 77, np.uint8)
    assert (of.pyr_down(const) == 77).all()
    ix, iy = of.scharr_deriv(const)
    assert not ix.any() and not iy.any()
    ramp = np.tile(np.arange(11, dtype=np.uint8) * 5, (9, 1))  # dI/dx = 5: Scharr x response 2 * 16 * 5 inside, 0 at the reflecting edges
    ix, iy = of.scharr_deriv(ramp)
    assert (ix[:, 1:-1] == 160).all() and (ix[:, [0, -1]] == 0).all() and not iy.any()
    # pyramid stops before a level would be <= the window
    assert [p.shape for p in of.build_pyramid(np.zeros((100, 90), np.uint8), 21, 3)] == [(100, 90), (50, 45), (25, 23)]
    assert len(of.build_pyramid(np.zeros((40, 40), np.uint8), 21, 3)) == 1
    # BGR2GRAY fixed point on the given channel order; ensure_int's 255.5 scaling of [0, 1] floats
    rgb = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 20, 30]]], np.uint8)
    assert list(of.to_gray(rgb)[0]) == [29, 150, 76, (10 * 1868 + 20 * 9617 + 30 * 4899 + 8192) >> 14]
    assert list(of.ensure_int(np.array([0.0, 0.5, 1.0], np.float32))) == [0, 127, 255]


@pytest.mark.parametrize("shift,levels", [((0.0, 0.0), 3), ((1.5, -0.75), 3), ((5.25, 3.5), 3), ((-9.0, 6.0), 3), ((2.25, 1.0), 0)])
def test_translations_are_recovered(shift, levels):
    h, w = 160, 200
    i0, i1 = texture(h, w), texture(h, w, *shift)
    pts = np.random.default_rng(0).uniform(35, 125, (10, 2)).astype(np.float32)
    nxt, status, err = of.calc_optical_flow_pyr_lk(i0, i1, pts, max_level=levels)
    assert status.all()
    assert np.abs(nxt - pts - np.array(shift, np.float32)).max() < 0.05
    assert (err >= 0).all() and err.max() < 1.0
    if shift == (0.0, 0.0):
        assert np.array_equal(nxt, pts) and not err.any()


def test_status_rules():
    h, w = 120, 150
    i0, i1 = texture(h, w), texture(h, w, 1.0, 1.0)
    pts = np.array([[-60.0, -60.0], [np.nan, 10.0], [5.0, 5.0], [w - 1.0, h - 1.0], [400.0, 50.0]], np.float32)
    nxt, status, err = of.calc_optical_flow_pyr_lk(i0, i1, pts)
    assert list(status) == [0, 0, 1, 1, 0] and err[0] == 0 and err[1] == 0 and err[4] == 0
    assert np.abs(nxt[2] - pts[2] - 1.0).max() < 0.1  # windows that hang over the border still track (reflected pixels)
    flat = np.full((h, w), 100, np.uint8)
    _, status, _ = of.calc_optical_flow_pyr_lk(flat, flat, np.array([[50, 50]], np.float32))
    assert list(status) == [0]  # no texture: smallest eigenvalue below the threshold


def test_flow_shift_points_follows_the_reference_call_site():
    """tracking.py:336-354: an instance becomes a candidate when MORE than min_shifted_points of its points were found; lost
    points are NaN; the shift score is minus the mean error of the found points."""
    h, w = 120, 150
    i0, i1 = texture(h, w), texture(h, w, 2.0, -1.0)
    a = np.array([[40.0, 40.0], [60.0, 50.0], [np.nan, np.nan]])
    b = np.array([[-80.0, -80.0], [np.nan, np.nan], [np.nan, np.nan]])  # nothing can be found
    out = of.flow_shift_points([a, b], i0[..., None], i1[..., None], min_shifted_points=0)
    assert [i for i, _, _ in out] == [0]
    _, pts, score = out[0]
    assert np.isnan(pts[2]).all() and np.abs(pts[:2] - a[:2] - [2.0, -1.0]).max() < 0.05 and score <= 0
    assert of.flow_shift_points([a, b], i0, i1, min_shifted_points=2) == []
    # img_scale (tracking.py:311-314, 321, 333): frames through cv2.resize, points x scale before and / scale after the flow
    out = of.flow_shift_points([a], i0, i1, scale=0.5)
    assert [i for i, _, _ in out] == [0]
    assert np.abs(out[0][1][:2] - a[:2] - [2.0, -1.0]).max() < 0.15 and np.isnan(out[0][1][2]).all()


def test_cv_resize_linear_u8_conventions():
    """cv2.resize(img, None, None, f, f) with INTER_LINEAR on uint8, OpenCV's documented conventions (imgproc/resize.cpp;
    OpenCV is absent, so these pin the restatement to the published algorithm, not to a build):
      * dsize = cvRound(size * f), round half to even (37 x 0.5 -> 18, 53 x 0.5 -> 26);
      * pixel centres: source x = (dx + 0.5) / f - 0.5 -- at f = 0.5 every output is the mean of a 2 x 2 block, rounded half up
        ((sum + 2) >> 2, which the two fixed-point passes reproduce exactly);
      * f = 1 is the identity; constant images stay constant at any f; the borders clamp (f = 2 keeps the corner pixels);
      * 11-bit coefficients: the weights of an output sum to 2048 and the result is within one grey level of the
        real-valued bilinear sample."""
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    half = of.cv_resize_linear_u8(img, 0.5, 0.5)
    assert half.shape == (18, 26)
    i = img.astype(int)
    ref = (i[:36:2, :52:2] + i[:36:2, 1:53:2] + i[1:37:2, :52:2] + i[1:37:2, 1:53:2] + 2) // 4
    assert np.array_equal(half, ref)
    assert np.array_equal(of.cv_resize_linear_u8(img, 1.0, 1.0), img)
    assert np.unique(of.cv_resize_linear_u8(np.full((20, 30), 137, np.uint8), 0.7, 0.7)).tolist() == [137]
    up = of.cv_resize_linear_u8(img, 2.0, 2.0)
    assert up.shape == (74, 106) and up[0, 0] == img[0, 0] and up[-1, -1] == img[-1, -1]
    # real-valued bilinear sample (half-pixel centres, clamped) at an arbitrary scale
    f = 0.6
    out = of.cv_resize_linear_u8(img, f, f)
    assert out.shape == (int(np.rint(37 * f)), int(np.rint(53 * f)))
    ys = np.clip((np.arange(out.shape[0]) + 0.5) / f - 0.5, 0, 36)
    xs = np.clip((np.arange(out.shape[1]) + 0.5) / f - 0.5, 0, 52)
    y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
    y1, x1 = np.minimum(y0 + 1, 36), np.minimum(x0 + 1, 52)
    wy, wx = (ys - y0)[:, None], (xs - x0)[None, :]
    real = (i[y0][:, x0] * (1 - wx) + i[y0][:, x1] * wx) * (1 - wy) + (i[y1][:, x0] * (1 - wx) + i[y1][:, x1] * wx) * wy
    assert np.abs(out.astype(float) - real).max() <= 1.0


def test_flow_tracker_saved_shifted_instances_are_pruned_to_the_track_window():
    """The reference's test_flow_tracker (tests/nn/test_inference.py:1963-1997) on the oracle tracker: with
    save_shifted_instances the keys (reference time, shifted-to time) never reach back further than track_window, and the
    identities of an easy scene are still carried."""
    from oracle import tracking as T
    from tests.test_gpu_flow import _moving_scene

    frames, insts = _moving_scene()
    h, w = frames.shape[1:3]
    track_window = 3
    trk = T.Tracker(tracker="flow", track_window=track_window, save_shifted_instances=True)
    seen = 0
    for t, lst in enumerate(insts[:9]):
        trk.track([T.Inst(p, s, sc, uid=i) for i, (p, s, sc) in enumerate(lst)], img_hw=(h, w), img=frames[t], t=t)
        for (ref_t, to_t) in trk._shifted:
            assert t - ref_t <= track_window and abs(ref_t - to_t) <= track_window
        seen += len(trk._shifted)
    assert seen > 0 and len(trk.spawned_tracks) == 3
    # flowmaxtracks keeps the class default (no saving), tracking.py:914-919
    assert not T.Tracker(tracker="flowmaxtracks", max_tracks=2, max_tracking=True, save_shifted_instances=True).save_shifted_instances
