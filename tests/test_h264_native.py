"""The native slice decoder (sleap_amd/csrc/h264dec.hip, `sa_h264_decode_slice`: what `MediaVideo` runs) against the Python
decoder it restates (sleap_amd/io/_h264.py; its own pins: tests/test_h264_inter.py, tests/test_frame0_golden.py): the decoded
planes AND the per-4x4 motion data (vectors, reference indices, intra map) must be equal picture by picture. Here the first
pictures of the four reference files -- CABAC I / P / B with weighted prediction and a B pyramid, spatial and temporal direct
(dance.mp4, samples 4 and 7), CAVLC I / P; the whole files (1100 + 1100 + 450 + 166 pictures): tools/h264_native_vs_python.py,
profiles/r06_h264_native_vs_python.txt. Host code only: runs without a GPU."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VID = os.path.join(ROOT, "tests", "golden", "video")


@pytest.mark.parametrize("name,n", [("centered_pair_low_quality.mp4", 9), ("centered_pair_small.mp4", 6), ("dance.mp4", 8), ("small_robot.mp4", 6),
                                    ("small_robot_3_frame.mp4", 3), ("clip.mp4", 3)])
def test_native_decoder_equals_the_python_decoder(name, n):
    from sleap_amd.io import _h264 as D
    from sleap_amd.io import _h264_intra as H

    tr = H.Mp4H264(os.path.join(VID, name))
    a, b = D.H264Decoder(tr.sps, tr.pps, "native"), D.H264Decoder(tr.sps, tr.pps, "python")
    kinds = []
    for i in range(n):
        pa, pb = a.decode_sample(tr.nal_units(i), i), b.decode_sample(tr.nal_units(i), i)
        assert pa.stats == pb.stats, (i, pa.stats, pb.stats)
        kinds.append(pa.stats["type"])
        assert pa.Y.dtype == np.uint8
        np.testing.assert_array_equal(pa.Y, pb.Y, err_msg=f"{name} sample {i} luma")
        np.testing.assert_array_equal(pa.C[0], pb.C[0])
        np.testing.assert_array_equal(pa.C[1], pb.C[1])
        np.testing.assert_array_equal(pa.mv, pb.mv)
        np.testing.assert_array_equal(pa.ref, pb.ref)
        np.testing.assert_array_equal(pa.intra4.astype(bool), pb.intra4)
        assert (pa.poc, pa.frame_num) == (pb.poc, pb.frame_num)
    assert kinds[0] == "I" and "P" in kinds and (name == "small_robot.mp4" or "B" in kinds)
    if name in ("small_robot_3_frame.mp4", "clip.mp4"):  # High profile: Intra 8x8 macroblocks and the 8x8 transform are in play
        assert tr.sps["profile"] == 100 and tr.pps["transform8x8"] == 1 and a.dpb[0].stats["I8"] > 100


def test_native_decoder_reports_a_corrupted_slice_instead_of_decoding_garbage():
    from sleap_amd.io import _h264 as D
    from sleap_amd.io import _h264_intra as H

    tr = H.Mp4H264(os.path.join(VID, "centered_pair_low_quality.mp4"))
    dec = D.H264Decoder(tr.sps, tr.pps, "native")
    nals = [bytes(n) for n in tr.nal_units(0)]
    k = max(range(len(nals)), key=lambda i: len(nals[i]))
    broken = bytearray(nals[k])
    for j in range(200, 260):  # flip bits in the middle of the slice data
        broken[j] ^= 0x5A
    nals[k] = bytes(broken)
    # (garbage usually ends in "end_of_slice_flag is not at the last macroblock"; it may also name a tool such as I_PCM first)
    with pytest.raises((AssertionError, NotImplementedError), match="sa_h264_decode_slice"):
        dec.decode_sample(nals, 0)


def test_media_video_runs_on_the_native_engine():
    from sleap_amd.io.video import MediaVideo

    v = MediaVideo(os.path.join(VID, "small_robot.mp4"))
    assert v._reader.engine == "native"
    f = v.get_frames(0, 12)
    assert f.shape == (12, 320, 560, 3) and f.dtype == np.uint8 and 90 < float(f.mean()) < 130


def test_gop_parallel_reads_equal_sequential_reads_across_a_key_frame():
    from sleap_amd.io.video import MediaVideo

    path = os.path.join(VID, "centered_pair_low_quality.mp4")
    seq, par = MediaVideo(path, workers=1), MediaVideo(path, workers=4)
    assert seq._gops is None and par._gops is not None and par._gops.closed
    a, b = seq.get_frames(140, 160), par.get_frames(140, 160)  # frames 140..149 end the first GOP, 150 is a key frame
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(par.get_frame(3), seq.get_frame(3))  # backwards
    with pytest.raises(KeyError, match="Unable to load frame 1100"):
        par.get_frame(1100)


def test_high_profile_file_against_the_baseline_encoding_of_the_same_frames():
    """tests/data/videos/small_robot_3_frame.mp4 (High profile: Intra 8x8, 8x8 transform, CABAC, I + P + B) holds frames 56, 86, 116 of
    small_robot.mp4 re-encoded from full-range images: its planes are the Baseline file's through the limited -> full range map
    (gains 255 / 219 and 255 / 224), so after an affine fit per plane the two decodes -- different profiles, entropy coders,
    transforms and picture types -- must agree to the encoders' quantisation (measured: luma 39.9 / 39.0 / 38.5 dB, chroma 41-43)."""
    from sleap_amd.io import _h264 as D

    hi = D.H264Reader(os.path.join(VID, "small_robot_3_frame.mp4"))
    base = D.H264Reader(os.path.join(VID, "small_robot.mp4"))
    for k, j in enumerate((56, 86, 116)):
        for comp, want_gain in ((0, 255 / 219), (1, 255 / 224), (2, 255 / 224)):
            a, b = hi.frame(k)[comp].astype(np.float64), base.frame(j)[comp].astype(np.float64)
            A = np.vstack([b.ravel(), np.ones(b.size)]).T
            gain, off = np.linalg.lstsq(A, a.ravel(), rcond=None)[0]
            ps = 10 * np.log10(255.0 ** 2 / np.mean((a - (gain * b + off)) ** 2))
            assert abs(gain - want_gain) < 0.03 and ps >= (37.5 if comp == 0 else 40.0), (k, comp, gain, off, ps)


def test_media_video_reads_the_1024_high_profile_clip():
    from sleap_amd.io.video import Video

    v = Video.from_filename(os.path.join(VID, "clip.mp4"))  # tests/data/tracks/clip.mp4: 1500 frames of 1024 x 1024, grey
    assert v.shape == (1500, 1024, 1024, 1) and v.backend.keyframes[:3] == [0, 250, 500]
    f = v.get_frames([0, 1, 2, 3])
    assert f.shape == (4, 1024, 1024, 1) and 5 < float(f.mean()) < 60


def test_native_colour_conversion_equals_the_numpy_one():
    import ctypes as C

    from sleap_amd import _lib
    from sleap_amd.io._h264_intra import swscale_blue, swscale_bgr

    lib = _lib.lib()
    rng = np.random.default_rng(0)
    for (h, w) in ((320, 560), (31, 45), (384, 384)):
        y = rng.integers(0, 256, (h, w)).astype(np.uint8)
        cb = rng.integers(0, 256, ((h + 1) // 2, (w + 1) // 2)).astype(np.uint8)
        cr = rng.integers(0, 256, cb.shape).astype(np.uint8)
        for ch, want in ((3, swscale_bgr(y, cb, cr)), (1, swscale_blue(y, cb)[..., None])):
            out = np.empty((h, w, ch), np.uint8)
            rc = lib.sa_yuv420_to_bgr(C.c_void_p(y.ctypes.data), C.c_void_p(cb.ctypes.data), C.c_void_p(cr.ctypes.data), w, h, w, cb.shape[1],
                                      C.c_void_p(out.ctypes.data), ch)
            assert rc == 0
            np.testing.assert_array_equal(out, want)
