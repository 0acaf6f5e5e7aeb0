"""csrc/flow.hip (sparse pyramidal Lucas-Kanade flow on the device) against oracle/optical_flow.py."""
import numpy as np
import pytest
import torch

from tests.test_oracle_optical_flow import texture

pytestmark = pytest.mark.gpu


def _device_pyramid_levels(p):
    """(level image uint8, Ix, Iy) per level, read back from the device buffer with the layout of csrc/flow.hip"""
    out, off, h, w = [], 0, p.H, p.W
    buf = p.buf.cpu().numpy()
    al = lambda v: (v + 255) & ~255  # noqa: E731
    for l in range(p.n_levels):
        if l:
            h, w = (h + 1) // 2, (w + 1) // 2
        img = buf[off:off + h * w].reshape(h, w)
        off += al(h * w)
        d = buf[off:off + 4 * h * w].view(np.int16).reshape(h, w, 2)
        off += al(4 * h * w)
        out.append((img, d[..., 0], d[..., 1]))
    return out


@pytest.mark.parametrize("shape,C", [((160, 200), 1), ((97, 131), 3), ((64, 50), 1), ((128, 256), 1), ((200, 192), 3), ((70, 72), 1)])
def test_pyramid_and_derivatives_are_exact(shape, C):
    from oracle import optical_flow as of
    from sleap_amd import ops

    rng = np.random.default_rng(sum(shape))
    img = rng.integers(0, 256, shape + (C,), dtype=np.uint8)
    p = ops.FlowPyramid(img, win=21, max_level=3)
    ref = of.build_pyramid(of.to_gray(img), 21, 3)
    assert p.n_levels == len(ref)
    for (g, ix, iy), r in zip(_device_pyramid_levels(p), ref):
        rx, ry = of.scharr_deriv(r)
        assert np.array_equal(g, r) and np.array_equal(ix, rx) and np.array_equal(iy, ry)


@pytest.mark.parametrize("shape,C,scale", [((160, 200), 1, 0.5), ((97, 131), 3, 0.5), ((128, 256), 1, 0.75), ((111, 143), 1, 0.6),
                                           ((64, 80), 3, 2.0), ((97, 131), 1, 0.25)])
def test_scaled_pyramid_is_the_oracles_resize_then_pyramid(shape, C, scale):
    """FlowCandidateMaker.img_scale != 1 (tracking.py:311-314): gray, then cv2.resize(INTER_LINEAR, uint8), then the pyramid --
    the device's level 0 equals the oracle's restatement of cv2.resize bit for bit, and so do all levels and derivatives."""
    from oracle import optical_flow as of
    from sleap_amd import ops

    rng = np.random.default_rng(sum(shape) + C)
    img = rng.integers(0, 256, shape + (C,), dtype=np.uint8)
    small = of.cv_resize_linear_u8(of.to_gray(img), scale, scale)
    p = ops.FlowPyramid(img, win=21, max_level=3, img_scale=scale)
    assert (p.H, p.W) == small.shape
    ref = of.build_pyramid(small, 21, 3)
    assert p.n_levels == len(ref)
    for (g, ix, iy), r in zip(_device_pyramid_levels(p), ref):
        rx, ry = of.scharr_deriv(r)
        assert np.array_equal(g, r) and np.array_equal(ix, rx) and np.array_equal(iy, ry)


@pytest.mark.parametrize("shape,C", [((128, 256), 1), ((97, 131), 3), ((160, 200), 1)])
def test_batched_pyramid_build_equals_single_builds(shape, C):
    """sa_flow_pyramid_build_batch (blockIdx.z = frame, buffers listed in a device table) == sa_flow_pyramid_build per frame"""
    import ctypes as ct

    from sleap_amd import _lib, ops

    rng = np.random.default_rng(7)
    frames = torch.from_numpy(rng.integers(0, 256, (5,) + shape + (C,), dtype=np.uint8)).cuda()
    single = [ops.FlowPyramid(frames[f], win=21, max_level=3) for f in range(5)]
    bufs = [torch.zeros_like(single[0].buf) for _ in range(5)]
    table = torch.tensor([b.data_ptr() for b in bufs], dtype=torch.int64, device="cuda")
    rc = _lib.lib().sa_flow_pyramid_build_batch(ct.c_void_p(frames.data_ptr()), 5, shape[0], shape[1], C, 21, 3,
                                                ct.c_void_p(table.data_ptr()), ops._stream())
    assert rc == 0
    torch.cuda.synchronize()
    for f in range(5):
        p, q = single[f], bufs[f]
        # (alignment gaps between the levels are never written: compare the level contents)
        ref = _device_pyramid_levels(p)
        p.buf = q
        for a, b in zip(ref, _device_pyramid_levels(p)):
            assert all(np.array_equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("shift,win,levels", [((0.0, 0.0), 21, 3), ((1.5, -0.75), 21, 3), ((5.25, 3.5), 21, 3), ((-9.0, 6.0), 15, 2),
                                               ((2.25, 1.0), 21, 0), ((3.0, -2.0), 31, 1)])
def test_lk_matches_oracle(shift, win, levels):
    """Same status everywhere, positions within 2e-3 px and errors within 1e-2 of the CPU restatement (the float32 window sums
    are ordered differently), including windows that hang over the border, lost, untextured and NaN points."""
    from oracle import optical_flow as of
    from sleap_amd import ops

    h, w = 160, 200
    i0, i1 = texture(h, w), texture(h, w, *shift)
    i0[100:, 150:] = 90  # an untextured corner
    i1[100:, 150:] = 90
    rng = np.random.default_rng(3)
    pts = np.concatenate([rng.uniform(-5, [w + 5, h + 5], (40, 2)), [[np.nan, 4.0], [-70.0, 10.0], [175.0, 130.0], [0.0, 0.0],
                                                                    [w - 1.0, h - 1.0]]]).astype(np.float32)
    want, wstat, werr = of.calc_optical_flow_pyr_lk(i0, i1, pts, win=win, max_level=levels)
    p0, p1 = ops.FlowPyramid(i0, win, levels), ops.FlowPyramid(i1, win, levels)
    got, gstat, gerr = (t.cpu().numpy() for t in ops.optical_flow_pyr_lk(p0, p1, pts))
    assert np.array_equal(gstat, wstat)
    ok = wstat.astype(bool)
    assert ok.sum() >= 20 and (~ok).sum() >= 3
    assert np.abs(got[ok] - want[ok]).max() < 2e-3
    assert np.abs(gerr - werr).max() < 1e-2  # quantised: |J - I| is an integer patch, a 1e-3 px shift flips a few counts
    if shift != (0.0, 0.0):
        inside = ok & (pts[:, 0] > 25) & (pts[:, 0] < 140) & (pts[:, 1] > 25) & (pts[:, 1] < 95)
        assert np.abs(got[inside] - pts[inside] - np.array(shift, np.float32)).max() < 0.08


def test_points_of_several_reference_frames_in_one_launch():
    from sleap_amd import ops

    h, w = 128, 160
    frames = [texture(h, w, 1.5 * k, -0.5 * k) for k in range(4)]
    pyr = [ops.FlowPyramid(f) for f in frames]
    pts = np.random.default_rng(0).uniform(40, 90, (6, 2)).astype(np.float32)
    allp = np.concatenate([pts + [1.5 * k, -0.5 * k] for k in range(3)]).astype(np.float32)  # where the points are in frame k
    prev = [pyr[k] for k in range(3) for _ in range(6)]
    got, st, _ = (t.cpu().numpy() for t in ops.optical_flow_pyr_lk(prev, pyr[3], allp))
    assert st.all()
    assert np.abs(got - np.tile(pts + [4.5, -1.5], (3, 1))).max() < 0.1
    one, _, _ = ops.optical_flow_pyr_lk(pyr[1], pyr[3], allp[6:12])
    assert np.array_equal(one.cpu().numpy(), got[6:12])


def _moving_scene(n_frames=14, h=192, w=224, n_animals=3, n_nodes=5, seed=0):
    """A textured background with textured patches gliding over it, and the node positions riding on the patches."""
    rng = np.random.default_rng(seed)
    bg = texture(h, w, seed=7).astype(np.float64) * 0.5
    pos = rng.uniform([50, 50], [w - 50, h - 50], (n_animals, 2))
    vel = rng.uniform(-4, 4, (n_animals, 2))
    rel = rng.uniform(-14, 14, (n_animals, n_nodes, 2))
    patches = [texture(41, 41, seed=20 + a, n=6).astype(np.float64) for a in range(n_animals)]
    yy, xx = np.mgrid[0:41, 0:41] - 20.0
    mask = np.exp(-(xx ** 2 + yy ** 2) / (2 * 11.0 ** 2))
    frames, insts = [], []
    for t in range(n_frames):
        img = bg.copy()
        lst = []
        for a in range(n_animals):
            c = pos[a] + vel[a] * t
            x0, y0 = int(round(c[0])) - 20, int(round(c[1])) - 20
            if 0 <= x0 and x0 + 41 <= w and 0 <= y0 and y0 + 41 <= h:
                img[y0:y0 + 41, x0:x0 + 41] = img[y0:y0 + 41, x0:x0 + 41] * (1 - mask) + patches[a] * mask
            pts = (np.round(c) + rel[a]).astype(np.float32)
            if t % 5 == 3 and a == 1:
                pts[2] = np.nan  # a missing node
            if not (t == 6 and a == 2):  # one dropped detection
                lst.append((pts, rng.uniform(0.3, 1, n_nodes).astype(np.float32), np.float32(rng.uniform(0.4, 1))))
        order = rng.permutation(len(lst))
        insts.append([lst[i] for i in order])
        frames.append(np.clip(np.round(img), 0, 255).astype(np.uint8)[..., None])
    insts[9] = []  # an empty frame
    return np.stack(frames), insts


@pytest.mark.parametrize("kw", [dict(tracker="flow"), dict(tracker="flow", similarity="iou", match="hungarian", track_window=3),
                                dict(tracker="flow", similarity="centroid", min_match_points=2, of_window_size=15, of_max_levels=2),
                                dict(tracker="flowmaxtracks", max_tracks=3, max_tracking=True),
                                dict(tracker="flowmaxtracks", max_tracks=2, max_tracking=True, similarity="object_keypoint", robust=0.8),
                                dict(tracker="flow", save_shifted_instances=True),
                                dict(tracker="flow", save_shifted_instances=True, track_window=3, min_match_points=1, similarity="iou"),
                                dict(tracker="flow", img_scale=0.5), dict(tracker="flow", img_scale=0.75, of_window_size=15)])
def test_flow_tracker_equals_oracle(kw):
    """The native tracker with device Lucas-Kanade candidates against the oracle tracker with the CPU restatement: the same
    tracks for every instance of every frame, tracking scores to 1e-3 (the shifted points differ by < 2e-3 px)."""
    from oracle import tracking as T
    from sleap_amd.nn.tracking import Tracker

    frames, insts = _moving_scene()
    h, w = frames.shape[1:3]
    okw = {k: v for k, v in kw.items()}
    ref = T.Tracker(**okw)
    want = []
    for t, lst in enumerate(insts):
        res = ref.track([T.Inst(p, s, sc, uid=i) for i, (p, s, sc) in enumerate(lst)], img_hw=(h, w), img=frames[t])
        want.append([(r.uid, r.track, r.tracking_score) for r in res])
    nat = Tracker.make_tracker_by_name(**kw)
    assert nat.uses_image
    dev = torch.from_numpy(frames).cuda()
    n_spawn_checked = 0
    for t, lst in enumerate(insts):
        pts = np.stack([p for p, _, _ in lst]) if lst else np.zeros((0, 5, 2), np.float32)
        ps = np.stack([s for _, s, _ in lst]) if lst else None
        sc = np.array([c for _, _, c in lst], np.float32) if lst else None
        r = nat.track(pts, ps, sc, img_hw=(h, w), img=dev[t])
        assert list(r["index"]) == [u for u, _, _ in want[t]], (t, r, want[t])
        assert list(r["track"]) == [tr for _, tr, _ in want[t]], (t, r, want[t])
        assert np.allclose(r["tracking_score"], [s for _, _, s in want[t]], atol=1e-3), (t, r, want[t])
        n_spawn_checked += len(lst)
    assert len(nat.spawned_tracks) == len(ref.spawned_tracks) and n_spawn_checked > 30
    if kw["tracker"] == "flow" and len(kw) <= 2 and "similarity" not in kw:
        # the scene is easy: identities must actually be carried (3 animals -> 3 tracks, despite the shuffled detection order)
        assert len(nat.spawned_tracks) == 3


@pytest.mark.parametrize("kw", [dict(tracker="flow"), dict(tracker="flow", track_window=2, min_match_points=1),
                                dict(tracker="flow", save_shifted_instances=True), dict(tracker="flow", img_scale=0.5),
                                dict(tracker="flowmaxtracks", max_tracks=3, max_tracking=True),
                                dict(tracker="flowmaxtracks", max_tracks=2, max_tracking=True, track_window=3)])
def test_flow_tracker_batched_frames_equal_single_steps(kw):
    """track_frames(images=...) shifts the queued instances of a whole run of frames in ONE Lucas-Kanade launch before the
    frame-by-frame matching (max-tracks mode: with an on-demand launch for what was not computed ahead); per-frame track()
    launches per frame. Same tracks, same scores, also when the run continues a queue filled by earlier calls."""
    from sleap_amd.nn.tracking import Tracker

    frames, insts = _moving_scene(seed=3)
    h, w = frames.shape[1:3]
    F, I, N = len(insts), 3, 5
    pts = np.full((F, I, N, 2), np.nan, np.float32)
    vals = np.full((F, I, N), np.nan, np.float32)
    sc = np.full((F, I), np.nan, np.float32)
    nv = np.zeros((F,), np.int32)
    for f, lst in enumerate(insts):
        nv[f] = len(lst)
        for i, (p, s, c) in enumerate(lst):
            pts[f, i], vals[f, i], sc[f, i] = p, s, c
    a = Tracker.make_tracker_by_name(**kw)
    one = [a.track(pts[f, :nv[f]], vals[f, :nv[f]], sc[f, :nv[f]], img_hw=(h, w), img=frames[f]) for f in range(F)]
    b = Tracker.make_tracker_by_name(**kw)
    cut = 5  # two runs: the second one starts from the queue the first one left
    r1 = b.track_frames(pts[:cut], vals[:cut], sc[:cut], nv[:cut], img_hw=(h, w), images=frames[:cut])
    r2 = b.track_frames(pts[cut:], vals[cut:], sc[cut:], nv[cut:], img_hw=(h, w), images=frames[cut:])
    r = {k: np.concatenate([r1[k], r2[k]]) for k in r1}
    for f in range(F):
        assert list(r["track"][f, one[f]["index"]]) == list(one[f]["track"]), f
        assert np.array_equal(r["tracking_score"][f, one[f]["index"]], one[f]["tracking_score"]), f
        assert list(r["order"][f, one[f]["index"]]) == list(range(len(one[f]["index"]))), f
        assert (r["track"][f, nv[f]:] == -1).all()
    assert a.spawned_tracks == b.spawned_tracks
    with pytest.raises(ValueError):
        Tracker.make_tracker_by_name(tracker="flow").track_frames(pts, vals, sc, nv, img_hw=(h, w))
