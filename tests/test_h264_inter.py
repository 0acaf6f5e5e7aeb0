"""The inter-picture half of the package's H.264 decoder (sleap_amd/io/_h264.py: P and B pictures of progressive Main-profile
CABAC streams) -- SURVEY 8(f) row 2, the reference's `MediaVideo` (sleap/io/video.py:340-504) without cv2 / FFmpeg.

No second decoder exists in this image, so pixel parity with FFmpeg is pinned indirectly, by everything that IS checkable:
  * parsing: a wrong context table entry, binarisation or neighbour rule desynchronises the arithmetic decoder; every picture
    must end with end_of_slice_flag exactly at the last macroblock and the slice data used up (asserted inside the decoder);
  * key frames: the general decoder and the intra-only module (pinned to TensorFlow's golden through frame 0,
    tests/test_frame0_golden.py) must produce the same planes;
  * inter prediction arithmetic (interpolation, weights, edge filter; invisible to the parser): the reference holds the SAME
    video encoded twice (tests/data/json_format_v1/centered_pair_low_quality.mp4 at QP ~21, tests/data/videos/
    centered_pair_small.mp4 at QP ~10-19: different motion vectors, residuals, weights). The two decodes must agree on P and B
    pictures as closely as on the key frame, where both are known to be right (measured: 49.0 dB at the key frame, 47.2-48.5 on
    the inter pictures, flat over a 150-picture GOP -- tools/h264_cross_check.py, profiles/r06_h264_cross_check.txt);
  * the interpolation routines against a sample-by-sample restatement of 8.4.2.2.1 / 8.4.2.2.2 in this file;
  * frozen digests of the first pictures (a regression pin of THIS decoder, not a golden)."""
import hashlib
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOW = os.path.join(ROOT, "tests", "golden", "video", "centered_pair_low_quality.mp4")
SMALL = os.path.join(ROOT, "tests", "golden", "video", "centered_pair_small.mp4")


def psnr(a, b):
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    return 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)


@pytest.fixture(scope="module")
def first_pictures():
    """the first nine samples of both files in decoding order (I P B B B P B B B): Pic objects"""
    from sleap_amd.io import _h264 as D
    from sleap_amd.io import _h264_intra as H

    out = {}
    for name, path in (("low", LOW), ("small", SMALL)):
        tr = H.Mp4H264(path)
        dec = D.H264Decoder(tr.sps, tr.pps)
        out[name] = (tr, [dec.decode_sample(tr.nal_units(i), i) for i in range(9)])
    return out


def test_p_and_b_pictures_parse_to_the_last_macroblock(first_pictures):
    for name in ("low", "small"):
        tr, pics = first_pictures[name]
        assert [p.stats["type"] for p in pics] == list("IPBBBPBBB")
        assert [p.poc for p in pics] == [0, 8, 4, 2, 6, 16, 12, 10, 14]  # display order 0 3 2 4 1 7 6 8 5 (the MP4's ctts says the same)
        assert tr.display_order[:9] == [0, 3, 2, 4, 1, 7, 6, 8, 5]
        for p in pics:
            st = p.stats
            assert st["I4"] + st["I16"] + st["skip"] + st["inter"] == 576 and -16 <= st["bits_left"] <= 16
            assert int(p.C[0].min()) == int(p.C[0].max()) == 128  # a grey stream stays grey through prediction and weighting
            assert 15 <= float(p.Y.mean()) <= 30
        assert all(p.stats["skip"] > 100 and p.stats["inter"] > 50 for p in pics[1:])


def test_key_frames_equal_the_intra_module(first_pictures):
    from sleap_amd.io import _h264 as D
    from sleap_amd.io import _h264_intra as H

    tr, pics = first_pictures["low"]
    y, cb, cr, _ = H.decode_intra(tr, 0)
    np.testing.assert_array_equal(pics[0].Y, y)
    np.testing.assert_array_equal(pics[0].C[0], cb)
    dec = D.H264Decoder(tr.sps, tr.pps)
    p = dec.decode_sample(tr.nal_units(150), 150)  # the second key frame, from a fresh decoder
    np.testing.assert_array_equal(p.Y, H.decode_intra(tr, 150)[0])


def test_two_encodings_of_the_same_video_agree_on_inter_pictures(first_pictures):
    (_, low), (_, small) = first_pictures["low"], first_pictures["small"]
    vals = [psnr(a.Y, b.Y) for a, b in zip(low, small)]
    print("cross PSNR of the two encodings, samples 0..8 (I P B B B P B B B):", [round(v, 2) for v in vals])
    assert vals[0] >= 48.0                       # the key frame: both decodes are the intra module's
    assert min(vals[1:]) >= vals[0] - 2.5, vals  # P and B pictures: within the coarser quantisation of inter pictures (QP + 0..2)
    # and the inter pictures are not copies of the key frame (the flies move: 28-37 dB between neighbours)
    assert psnr(low[0].Y, low[1].Y) < 40.0


def test_reader_serves_display_order_and_random_access(first_pictures):
    from sleap_amd.io import _h264 as D

    tr, pics = first_pictures["low"]
    rd = D.H264Reader(LOW)
    got = [rd.frame(k)[0] for k in range(5)]  # display frames 0..4 = samples 0, 3, 2, 4, 1
    for k, s in enumerate([0, 3, 2, 4, 1]):
        np.testing.assert_array_equal(got[k], pics[s].Y.astype(np.uint8))
    rd2 = D.H264Reader(LOW)
    np.testing.assert_array_equal(rd2.frame(3)[0], got[3])  # a jump: decodes samples 0..4 behind the scenes
    np.testing.assert_array_equal(rd2.frame(1)[0], got[1])  # backwards within the cache
    with pytest.raises(IndexError):
        rd.frame(1100)


def test_frozen_digests_of_the_first_pictures(first_pictures):
    """regression pin of this decoder's output (sha1 of the luma planes, decoding order)"""
    want = {"low": FROZEN_LOW, "small": FROZEN_SMALL}
    for name in ("low", "small"):
        got = [hashlib.sha1(np.ascontiguousarray(p.Y.astype(np.uint8)).tobytes()).hexdigest()[:12] for p in first_pictures[name][1]]
        assert got == want[name], got


FROZEN_LOW = ["417cac09ee64", "7781ed8966b1", "4dea7507286d", "3e60911c4abb", "46f6abc96f6c", "32372a2fc310", "20365100874a", "ae584c9e4dce", "000517e34704"]
FROZEN_SMALL = ["fc81cf6a064b", "6a1b589048c6", "e2185991d4d4", "cd41d95ca1f5", "5165731116c8", "04ece549160d", "7481070054b8", "657d815657d0", "f75411a51688"]


def test_unsupported_streams_are_refused_by_name():
    from sleap_amd.io import _h264 as D

    sps = {"profile": 77, "mb_w": 2, "mb_h": 2, "log2_max_frame_num": 4, "poc_type": 0, "log2_max_poc_lsb": 4, "num_ref_frames": 1,
           "direct_8x8_inference": 1, "crop": (0, 0, 0, 0)}
    pps = {"cabac": 1, "constrained_intra": 0, "num_ref_idx_default": (1, 1), "weighted_pred": 0, "weighted_bipred_idc": 0}
    with pytest.raises(NotImplementedError, match="profile_idc 110"):  # High 10 (High itself is read since the 8x8 transform exists)
        D.H264Decoder(dict(sps, profile=110), pps)
    with pytest.raises(NotImplementedError, match="constrained_intra_pred"):
        D.H264Decoder(sps, dict(pps, constrained_intra=1))


def test_baseline_cavlc_stream_against_frames_a_real_decoder_extracted():
    """tests/data/videos/small_robot.mp4 (Baseline: CAVLC, one key frame + 165 P pictures, colour) is the file the reference's own
    MediaVideo tests read (tests/io/test_video.py:84-126), and tests/data/videos/robot0.jpg .. robot2.jpg are frames 56, 86, 116
    of it as FFmpeg decoded them (then JPEG-compressed: ~38.5 dB is that compression's own loss; the neighbouring frames are
    ~30 dB away). Frame 56 sits behind a chain of 56 P pictures: motion compensation incl. chroma, CAVLC residuals, intra
    macroblocks inside P pictures, the edge filter and the colour conversion all have to be right to land on it. (Measured over the
    whole file: frames 56 / 86 / 116 at 38.44 / 38.49 / 38.29 dB -- no drift along the chain.)"""
    from sleap_amd.io.video import MediaVideo, Video

    v = Video.from_filename(os.path.join(ROOT, "tests", "golden", "video", "small_robot.mp4"))
    assert isinstance(v.backend, MediaVideo) and v.shape == (166, 320, 560, 3) and v.backend.fps == 30.0  # (test_video.py:84-93)
    assert v.backend.keyframes == [0] and v.backend.grayscale is False
    want = np.load(os.path.join(ROOT, "tests", "golden", "robot.npz"))["frames"][0]  # robot0.jpg as RGB
    got = {k: v.get_frame(k) for k in (55, 56, 57)}  # (sequential: the decoder runs once through frames 0..57)
    ps = {k: psnr(got[k], want) for k in got}
    print("small_robot.mp4 frames 55, 56, 57 against robot0.jpg:", {k: round(x, 2) for k, x in ps.items()})
    assert ps[56] >= 37.5 and ps[56] - max(ps[55], ps[57]) >= 3.0, ps
    st = v.backend._reader._dec.stats if hasattr(v.backend._reader._dec, "stats") else None  # noqa: F841


# ---- interpolation against a sample-by-sample restatement of the standard's formulas
def _ref_luma(ref, xq, yq):
    """8.4.2.2.1 for ONE sample at quarter position (xq, yq); names as in Figure 8-4"""
    H, W = ref.shape

    def s(x, y):
        return int(ref[min(max(y, 0), H - 1), min(max(x, 0), W - 1)])

    xi, yi, fx, fy = xq >> 2, yq >> 2, xq & 3, yq & 3

    def b1(x, y):  # horizontal intermediate between (x, y) and (x + 1, y)
        return s(x - 2, y) - 5 * s(x - 1, y) + 20 * s(x, y) + 20 * s(x + 1, y) - 5 * s(x + 2, y) + s(x + 3, y)

    def h1(x, y):
        return s(x, y - 2) - 5 * s(x, y - 1) + 20 * s(x, y) + 20 * s(x, y + 1) - 5 * s(x, y + 2) + s(x, y + 3)

    def c(v):
        return min(max(v, 0), 255)

    G, Hs, M = s(xi, yi), s(xi + 1, yi), s(xi, yi + 1)
    b, h = c((b1(xi, yi) + 16) >> 5), c((h1(xi, yi) + 16) >> 5)
    sm, m = c((b1(xi, yi + 1) + 16) >> 5), c((h1(xi + 1, yi) + 16) >> 5)
    j1 = b1(xi, yi - 2) - 5 * b1(xi, yi - 1) + 20 * b1(xi, yi) + 20 * b1(xi, yi + 1) - 5 * b1(xi, yi + 2) + b1(xi, yi + 3)
    j = c((j1 + 512) >> 10)
    table = {(0, 0): G, (1, 0): (G + b + 1) >> 1, (2, 0): b, (3, 0): (b + Hs + 1) >> 1,
             (0, 1): (G + h + 1) >> 1, (1, 1): (b + h + 1) >> 1, (2, 1): (b + j + 1) >> 1, (3, 1): (b + m + 1) >> 1,
             (0, 2): h, (1, 2): (h + j + 1) >> 1, (2, 2): j, (3, 2): (j + m + 1) >> 1,
             (0, 3): (h + M + 1) >> 1, (1, 3): (h + sm + 1) >> 1, (2, 3): (j + sm + 1) >> 1, (3, 3): (sm + m + 1) >> 1}
    return table[(fx, fy)]


def test_luma_interpolation_all_sixteen_positions_incl_clamped_edges():
    from sleap_amd.io._h264 import mc_luma

    rng = np.random.default_rng(0)
    ref = rng.integers(0, 256, (24, 28)).astype(np.int32)
    for fx in range(4):
        for fy in range(4):
            for (x0, y0, w, h) in ((8, 8, 8, 8), (-3, -2, 4, 4), (22, 19, 8, 4)):  # interior, top-left and bottom-right overhang
                got = mc_luma(ref, x0 * 4 + fx, y0 * 4 + fy, w, h)
                want = np.array([[_ref_luma(ref, (x0 + i) * 4 + fx, (y0 + j) * 4 + fy) for i in range(w)] for j in range(h)])
                np.testing.assert_array_equal(got, want, err_msg=f"fraction ({fx}, {fy}) at ({x0}, {y0})")


def test_chroma_interpolation():
    from sleap_amd.io._h264 import mc_chroma

    rng = np.random.default_rng(1)
    ref = rng.integers(0, 256, (12, 14)).astype(np.int32)
    H, W = ref.shape

    def s(x, y):
        return int(ref[min(max(y, 0), H - 1), min(max(x, 0), W - 1)])

    for fx in range(8):
        for fy in range(8):
            for (x0, y0) in ((4, 4), (-2, -1), (11, 9)):
                got = mc_chroma(ref, x0 * 8 + fx, y0 * 8 + fy, 4, 4)
                want = np.array([[((8 - fx) * (8 - fy) * s(x0 + i, y0 + j) + fx * (8 - fy) * s(x0 + i + 1, y0 + j) +
                                   (8 - fx) * fy * s(x0 + i, y0 + j + 1) + fx * fy * s(x0 + i + 1, y0 + j + 1) + 32) >> 6
                                  for i in range(4)] for j in range(4)])
                np.testing.assert_array_equal(got, want)
