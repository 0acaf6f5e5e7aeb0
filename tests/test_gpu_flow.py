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


@pytest.mark.parametrize("shape,C", [((160, 200), 1), ((97, 131), 3), ((64, 50), 1)])
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


@pytest.mark.parametrize("shift,win,levels", [((0.0, 0.0), 21, 3), ((1.5, -0.75), 21, 3), ((5.25, 3.5), 21, 3), ((-9.0, 6.0), 15, 2),
                                               ((2.25, 1.0), 21, 0), ((3.0, -2.0), 31, 1)])
def test_lk_matches_oracle(shift, win, levels):
    """Same status everywhere, positions within 2e-3 px and errors within 1e-3 of the CPU restatement (the float32 window sums
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
    assert np.abs(gerr - werr).max() < 1e-3
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
