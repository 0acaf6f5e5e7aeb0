"""H.264 decoder for progressive 8-bit 4:2:0 Baseline / Main / High-profile streams: I, P and B pictures, CABAC or CAVLC, frame
macroblocks, one slice per picture, 4x4 and 8x8 transform with flat scaling matrices -- what x264 writes for `-profile baseline`
/ `main` / `high`, and what ALL the reference's test videos are: tests/data/videos/small_robot_3_frame.mp4 and
tests/data/tracks/clip.mp4 (High: Intra 8x8, 8x8 transform; the latter 1500 frames of 1024 x 1024), tests/data/videos/small_robot.mp4 (Baseline: CAVLC, P pictures; the file its MediaVideo tests
read), tests/data/json_format_v1/centered_pair_low_quality.mp4, tests/data/videos/centered_pair_small.mp4, dance.mp4 (Main:
CABAC, B pyramids, weighted prediction). Two engines for the macroblock layer behind the same Python front end (MP4 tables,
parameter sets, slice headers, picture order, reference lists and marking): `engine="native"` -- `sa_h264_decode_slice`, host C++
in the package's library (csrc/h264dec.hip), 325 (1024 x 1024) ... 545 (384 x 384) pictures/s per core, what `MediaVideo` runs -- and `engine="python"` -- the classes of
this file on top of io/_h264_intra.py (bit reader, CABAC engine, intra prediction, transforms, edge filter), ~0.2-0.8 s per
384 x 384 picture, the restatement the native engine is checked against (tests/test_h264_native.py: equal planes and motion data).
`sleap_amd.io.video.MediaVideo` reads every frame of such a file through `H264Reader` (display order = the MP4's composition times, as cv2.VideoCapture numbers frames:
sleap/io/video.py:340-504).

Implemented beyond the intra module (clause numbers of ITU-T H.264): slice headers of P / B slices incl. reference picture list
modification, the prediction weight table and memory management control operation 1 (7.3.3); picture order count type 0
(8.2.1.1); reference list initialisation and modification (8.2.4), sliding-window and adaptive marking (8.2.5); the CABAC
syntax of P and B macroblocks for cabac_init_idc 0 (9.3: mb_skip_flag, mb_type, sub_mb_type, ref_idx, mvd, intra macroblocks in
inter slices) and the CAVLC syntax of I and P slices (9.2: coeff_token, levels, total_zeros, run_before; mb_skip_run, me(v) / te(v)
/ se(v) elements); motion vector prediction incl. P_Skip, spatial and temporal direct with direct_8x8_inference (8.4.1); quarter-
sample luma and eighth-sample chroma interpolation, default / explicit / implicit weighted prediction (8.4.2); the edge
filter's boundary strengths for inter pictures (8.7.2.1); the High-profile tools for CABAC streams: transform_size_8x8_flag, 8x8
residual blocks (ctxBlockCat 5), Intra 8x8 prediction with its reference sample filter (8.3.2), the 8x8 inverse transform and
scaling (8.5.13), edge filtering by transform size, second_chroma_qp_index_offset. NOT implemented, each refused with a
message: scaling matrices, 4:2:2 / 4:4:4 and more than 8 bits, the 8x8 transform with CAVLC, cabac_init_idc 1 and 2 (their context tables are not
held: no stream here uses them and nothing could validate them), B slices with CAVLC, I_PCM, long-term references, fields /
MBAFF, several slices per picture, constrained intra prediction.

Checks. (1) A decode is self-checking like the intra module's: a wrong table entry, binarisation or neighbour rule
desynchronises the entropy decoder, and the slice then does not end exactly at the last macroblock with its data used up --
asserted for every picture (all 1100 + 1100 + 450 + 166 + 3 + 1500 pictures of the six files decode); the VLC tables are checked to be
prefix-free codes at import. (2) Key frames are decoded by BOTH modules and must agree bit for bit. (3) An external decoder's
pixels: tests/data/videos/robot0..2.jpg are frames 56, 86, 116 of small_robot.mp4 as FFmpeg decoded them -- this decoder lands on
them at 38.3-38.5 dB (the JPEGs' own compression loss; neighbouring frames: ~30 dB), equally at the end of a 116-picture P chain.
(4) The same video encoded twice (centered_pair_low_quality / centered_pair_small): the two decodes agree on P and B pictures
as on key frames (49 dB at key frames, 47-48 after them, following the encoders' quantisers down to 45 at the end of a GOP;
tools/h264_cross_check.py, profiles/r06_h264_cross_check.txt). What remains unpinned: bit-exactness of inter pictures against
FFmpeg (no second decoder in this image), and the tools no fixture exercises (explicit bi-prediction weights, temporal direct is
exercised by dance.mp4's parser checks only). tests/test_h264_inter.py.
"""
import numpy as np

from ._h264_intra import (ALPHA, BETA, BLK_XY, CAT_ABS, CAT_CBF, CAT_SIG, CTX_I, QPC, TC0, XY_BLK, ZIGZAG, Bits, Cabac, Mp4H264,
                          _filter_line, idct4, level_scale, pred4, pred16, pred_chroma, rbsp)


class Unsupported(NotImplementedError):
    """The stream uses a coding tool this decoder does not have."""


# ----------------------------------------------------------------------------------------------------------------- CABAC contexts
# (m, n) of Tables 9-12 .. 9-23, cabac_init_idc 0, ctxIdx 11 .. 275 (P / B slices; 0 .. 10 as in I slices)
CTX_PB0 = {k: CTX_I[k] for k in range(11)}


def _fill(start, pairs, end):
    assert start + len(pairs) == end + 1, (start, len(pairs), end)
    for k, mn in enumerate(pairs):
        assert start + k not in CTX_PB0
        CTX_PB0[start + k] = mn


_fill(11, [(23, 33), (23, 2), (21, 0), (1, 9), (0, 49), (-37, 118), (5, 57), (-13, 78), (-11, 65), (1, 62), (12, 49), (-4, 73), (17, 50)], 23)
_fill(24, [(18, 64), (9, 43), (29, 0), (26, 67), (16, 90), (9, 104), (-46, 127), (-20, 104), (1, 67), (-13, 78), (-11, 65), (1, 62), (-6, 86),
           (-17, 95), (-6, 61), (9, 45)], 39)
_fill(40, [(-3, 69), (-6, 81), (-11, 96), (6, 55), (7, 67), (-5, 86), (2, 88), (0, 58), (-3, 76), (-10, 94), (5, 54), (4, 69), (-3, 81),
           (0, 88)], 53)
_fill(54, [(-7, 67), (-5, 74), (-4, 74), (-5, 80), (-7, 72), (1, 58)], 59)
_fill(60, [(0, 41), (0, 63), (0, 63), (0, 63), (-9, 83), (4, 86), (0, 97), (-7, 72), (13, 41), (3, 62)], 69)
_fill(70, [(0, 45), (-4, 78), (-3, 96), (-27, 126), (-28, 98), (-25, 101), (-23, 67), (-28, 82), (-20, 94), (-16, 83), (-22, 110), (-21, 91),
           (-18, 102), (-13, 93), (-29, 127), (-7, 92), (-5, 89), (-7, 96), (-13, 108), (-3, 46), (-1, 65), (-1, 57), (-9, 93), (-3, 74),
           (-9, 92), (-8, 87), (-23, 126), (5, 54), (6, 60), (6, 59), (6, 69), (-1, 48), (0, 68), (-4, 69), (-8, 88)], 104)
_fill(105, [(-2, 85), (-6, 78), (-1, 75), (-7, 77), (2, 54), (5, 50), (-3, 68), (1, 50), (6, 42), (-4, 81), (1, 63), (-4, 70), (0, 67),
            (2, 57), (-2, 76), (11, 35), (4, 64), (1, 61), (11, 35), (18, 25), (12, 24), (13, 29), (13, 36), (-10, 93), (-7, 73), (-2, 73),
            (13, 46), (9, 49), (-7, 100), (9, 53), (2, 53), (5, 53), (-2, 61), (0, 56), (0, 56), (-13, 63), (-5, 60), (-1, 62), (4, 57),
            (-6, 69), (4, 57), (14, 39), (4, 51), (13, 68), (3, 64), (1, 61), (9, 63), (7, 50), (16, 39), (5, 44), (4, 52), (11, 48),
            (-5, 60), (-1, 59), (0, 59), (22, 33), (5, 44), (14, 43), (-1, 78), (0, 60), (9, 69)], 165)
_fill(166, [(11, 28), (2, 40), (3, 44), (0, 49), (0, 46), (2, 44), (2, 51), (0, 47), (4, 39), (2, 62), (6, 46), (0, 54), (3, 54), (2, 58),
            (4, 63), (6, 51), (6, 57), (7, 53), (6, 52), (6, 55), (11, 45), (14, 36), (8, 53), (-1, 82), (7, 55), (-3, 78), (15, 46),
            (22, 31), (-1, 84), (25, 7), (30, -7), (28, 3), (28, 4), (32, 0), (34, -1), (30, 6), (30, 6), (32, 9), (31, 19), (26, 27),
            (26, 30), (37, 20), (28, 34), (17, 70), (1, 67), (5, 59), (9, 67), (16, 30), (18, 32), (18, 35), (22, 29), (24, 31), (23, 38),
            (18, 43), (20, 41), (11, 63), (9, 59), (9, 64), (-1, 94), (-2, 89), (-9, 108)], 226)
_fill(227, [(-6, 76), (-2, 44), (0, 45), (0, 52), (-3, 64), (-2, 59), (-4, 70), (-4, 75), (-8, 82), (-17, 102), (-9, 77), (3, 24), (0, 42),
            (0, 48), (0, 55), (-6, 59), (-7, 71), (-12, 83), (-11, 87), (-30, 119), (1, 58), (-3, 29), (-1, 36), (1, 38), (2, 43), (-6, 55),
            (0, 58), (0, 64), (-3, 74), (-10, 90), (0, 70), (-4, 29), (5, 31), (7, 42), (1, 59), (-2, 58), (-3, 72), (-3, 81), (-11, 97),
            (0, 58), (8, 5), (10, 14), (14, 18), (13, 27), (2, 40), (0, 58), (-3, 70), (-6, 79), (-8, 85)], 275)
assert sorted(CTX_PB0) == list(range(276))
# High profile: transform_size_8x8_flag (399..401) and the 8x8 luma blocks of frame macroblocks (ctxBlockCat 5: significant_coeff_flag
# 402..416, last_significant_coeff_flag 417..425, coeff_abs_level_minus1 426..435), I slices / cabac_init_idc 0
_CTX8_I = [(31, 21), (31, 31), (25, 50), (-17, 120), (-20, 112), (-18, 114), (-11, 85), (-15, 92), (-14, 89), (-26, 71), (-15, 81), (-14, 80),
           (0, 68), (-14, 70), (-24, 56), (-23, 68), (-24, 50), (-11, 74), (23, -13), (26, -13), (40, -15), (49, -14), (44, 3), (45, 6),
           (44, 34), (33, 54), (19, 82), (-3, 75), (-1, 23), (1, 34), (1, 43), (0, 54), (-2, 55), (0, 61), (1, 64), (0, 68), (-9, 92)]
_CTX8_PB0 = [(12, 40), (11, 51), (14, 59), (-4, 79), (-7, 71), (-5, 69), (-9, 70), (-8, 66), (-10, 68), (-19, 73), (-12, 69), (-16, 70),
             (-15, 67), (-20, 62), (-19, 70), (-16, 66), (-22, 65), (-20, 63), (9, -2), (26, -9), (33, -9), (39, -7), (41, -2), (45, 3),
             (49, 9), (45, 27), (36, 59), (-6, 66), (-7, 35), (-7, 42), (-8, 45), (-5, 48), (-12, 56), (-6, 60), (-5, 62), (-8, 66), (-8, 76)]
assert len(_CTX8_I) == len(_CTX8_PB0) == 37
CTX_I_HIGH = {**CTX_I, **{399 + k: v for k, v in enumerate(_CTX8_I)}}
CTX_PB0_HIGH = {**CTX_PB0, **{399 + k: v for k, v in enumerate(_CTX8_PB0)}}
# ctxIdxInc of significant_coeff_flag / last_significant_coeff_flag for the 63 scan positions of an 8x8 block (Table 9-43, frame)
SIG8 = [0, 1, 2, 3, 4, 5, 5, 4, 4, 3, 3, 4, 4, 4, 5, 5, 4, 4, 4, 4, 3, 3, 6, 7, 7, 7, 8, 9, 10, 9, 8, 7, 7, 6, 11, 12, 13, 11, 6, 7, 8, 9, 14, 10, 9, 8,
        6, 11, 12, 13, 11, 6, 9, 14, 10, 9, 11, 12, 13, 11, 14, 10, 12]
LAST8 = [0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3, 4, 4, 4, 4, 4, 4, 4, 4,
         5, 5, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 8, 8, 8]
assert len(SIG8) == len(LAST8) == 63
# 8x8 zig-zag scan (Figure 6-? / Table 8-?: the classic one): (x, y) per scan position
ZIGZAG8 = []
for _d in range(15):
    _xs = range(min(_d, 7), max(_d - 7, 0) - 1, -1) if _d % 2 else range(max(_d - 7, 0), min(_d, 7) + 1)
    ZIGZAG8 += [(_x, _d - _x) for _x in _xs]
assert len(ZIGZAG8) == 64 and ZIGZAG8[:6] == [(0, 0), (1, 0), (0, 1), (0, 2), (1, 1), (2, 0)] and ZIGZAG8[-1] == (7, 7)
NORM_ADJUST8 = [(20, 18, 32, 19, 25, 24), (22, 19, 35, 21, 28, 26), (26, 23, 42, 24, 33, 31), (28, 25, 45, 26, 35, 33), (32, 28, 51, 30, 40, 38),
                (36, 32, 58, 34, 46, 43)]


def level_scale8(qp, i, j):
    """16 x normAdjust8x8 (8.5.9, flat scaling matrix) for row i, column j"""
    v = NORM_ADJUST8[qp % 6]
    if i % 4 == 0 and j % 4 == 0:
        k = 0
    elif i % 2 == 1 and j % 2 == 1:
        k = 1
    elif i % 4 == 2 and j % 4 == 2:
        k = 2
    elif (i % 4 == 0 and j % 2 == 1) or (i % 2 == 1 and j % 4 == 0):
        k = 3
    elif (i % 4 == 0 and j % 4 == 2) or (i % 4 == 2 and j % 4 == 0):
        k = 4
    else:
        k = 5
    return 16 * v[k]


def idct8(d):
    """8.5.13: d[y][x] scaled coefficients of an 8x8 block -> residual ((x + 32) >> 6 applied)"""
    def one(v):
        d0, d1, d2, d3, d4, d5, d6, d7 = v
        a0, a4, a2, a6 = d0 + d4, d0 - d4, (d2 >> 1) - d6, d2 + (d6 >> 1)
        b0, b2, b4, b6 = a0 + a6, a4 + a2, a4 - a2, a0 - a6
        a1 = -d3 + d5 - d7 - (d7 >> 1)
        a3 = d1 + d7 - d3 - (d3 >> 1)
        a5 = -d1 + d7 + d5 + (d5 >> 1)
        a7 = d3 + d5 + d1 + (d1 >> 1)
        b1, b7, b3, b5 = a1 + (a7 >> 2), a7 - (a1 >> 2), a3 + (a5 >> 2), (a3 >> 2) - a5
        return [b0 + b7, b2 + b5, b4 + b3, b6 + b1, b6 - b1, b4 - b3, b2 - b5, b0 - b7]

    rows = [one(r) for r in d]
    cols = [one([rows[y][x] for y in range(8)]) for x in range(8)]
    return [[(cols[x][y] + 32) >> 6 for x in range(8)] for y in range(8)]


def pred8(pic, m, mx, my, b8):
    """Intra 8x8 prediction of luma 8x8 block b8 (8.3.2): reference samples low-pass filtered, nine modes"""
    bx, by = b8 & 1, b8 >> 1
    x0, y0 = mx * 16 + bx * 8, my * 16 + by * 8
    Y = pic.Y
    left = bx > 0 or pic.mb(mx - 1, my) is not None
    top = by > 0 or pic.mb(mx, my - 1) is not None
    if b8 == 0:
        tr, tl = top, pic.mb(mx - 1, my - 1) is not None
    elif b8 == 1:
        tr, tl = pic.mb(mx + 1, my - 1) is not None, top
    elif b8 == 2:
        tr, tl = True, left
    else:
        tr, tl = False, True
    p = {}
    if top:
        for i in range(8):
            p[(i, -1)] = int(Y[y0 - 1, x0 + i])
        for i in range(8, 16):
            p[(i, -1)] = int(Y[y0 - 1, x0 + i]) if tr else p[(7, -1)]
    if left:
        for j in range(8):
            p[(-1, j)] = int(Y[y0 + j, x0 - 1])
    if tl:
        p[(-1, -1)] = int(Y[y0 - 1, x0 - 1])
    q = {}  # filtered samples (8.3.2.2.1)
    if top:
        q[(0, -1)] = (p[(-1, -1)] + 2 * p[(0, -1)] + p[(1, -1)] + 2) >> 2 if tl else (3 * p[(0, -1)] + p[(1, -1)] + 2) >> 2
        for x in range(1, 15):
            q[(x, -1)] = (p[(x - 1, -1)] + 2 * p[(x, -1)] + p[(x + 1, -1)] + 2) >> 2
        q[(15, -1)] = (p[(14, -1)] + 3 * p[(15, -1)] + 2) >> 2
    if tl:
        if top and left:
            q[(-1, -1)] = (p[(0, -1)] + 2 * p[(-1, -1)] + p[(-1, 0)] + 2) >> 2
        elif top:
            q[(-1, -1)] = (3 * p[(-1, -1)] + p[(0, -1)] + 2) >> 2
        elif left:
            q[(-1, -1)] = (3 * p[(-1, -1)] + p[(-1, 0)] + 2) >> 2
        else:
            q[(-1, -1)] = p[(-1, -1)]
    if left:
        q[(-1, 0)] = (p[(-1, -1)] + 2 * p[(-1, 0)] + p[(-1, 1)] + 2) >> 2 if tl else (3 * p[(-1, 0)] + p[(-1, 1)] + 2) >> 2
        for y in range(1, 7):
            q[(-1, y)] = (p[(-1, y - 1)] + 2 * p[(-1, y)] + p[(-1, y + 1)] + 2) >> 2
        q[(-1, 7)] = (p[(-1, 6)] + 3 * p[(-1, 7)] + 2) >> 2
    mode = m.modes[BLK8_FIRST[b8]]
    out = [[0] * 8 for _ in range(8)]
    for y in range(8):
        for x in range(8):
            if mode == 0:
                v = q[(x, -1)]
            elif mode == 1:
                v = q[(-1, y)]
            elif mode == 2:
                if top and left:
                    v = (sum(q[(i, -1)] for i in range(8)) + sum(q[(-1, j)] for j in range(8)) + 8) >> 4
                elif left:
                    v = (sum(q[(-1, j)] for j in range(8)) + 4) >> 3
                elif top:
                    v = (sum(q[(i, -1)] for i in range(8)) + 4) >> 3
                else:
                    v = 128
            elif mode == 3:
                v = (q[(14, -1)] + 3 * q[(15, -1)] + 2) >> 2 if (x == 7 and y == 7) else (q[(x + y, -1)] + 2 * q[(x + y + 1, -1)] + q[(x + y + 2, -1)] + 2) >> 2
            elif mode == 4:
                if x > y:
                    v = (q[(x - y - 2, -1)] + 2 * q[(x - y - 1, -1)] + q[(x - y, -1)] + 2) >> 2
                elif x < y:
                    v = (q[(-1, y - x - 2)] + 2 * q[(-1, y - x - 1)] + q[(-1, y - x)] + 2) >> 2
                else:
                    v = (q[(0, -1)] + 2 * q[(-1, -1)] + q[(-1, 0)] + 2) >> 2
            elif mode == 5:
                z = 2 * x - y
                if z >= 0 and z % 2 == 0:
                    v = (q[(x - (y >> 1) - 1, -1)] + q[(x - (y >> 1), -1)] + 1) >> 1
                elif z >= 0:
                    v = (q[(x - (y >> 1) - 2, -1)] + 2 * q[(x - (y >> 1) - 1, -1)] + q[(x - (y >> 1), -1)] + 2) >> 2
                elif z == -1:
                    v = (q[(-1, 0)] + 2 * q[(-1, -1)] + q[(0, -1)] + 2) >> 2
                else:
                    v = (q[(-1, y - 2 * x - 1)] + 2 * q[(-1, y - 2 * x - 2)] + q[(-1, y - 2 * x - 3)] + 2) >> 2
            elif mode == 6:
                z = 2 * y - x
                if z >= 0 and z % 2 == 0:
                    v = (q[(-1, y - (x >> 1) - 1)] + q[(-1, y - (x >> 1))] + 1) >> 1
                elif z >= 0:
                    v = (q[(-1, y - (x >> 1) - 2)] + 2 * q[(-1, y - (x >> 1) - 1)] + q[(-1, y - (x >> 1))] + 2) >> 2
                elif z == -1:
                    v = (q[(-1, 0)] + 2 * q[(-1, -1)] + q[(0, -1)] + 2) >> 2
                else:
                    v = (q[(x - 2 * y - 1, -1)] + 2 * q[(x - 2 * y - 2, -1)] + q[(x - 2 * y - 3, -1)] + 2) >> 2
            elif mode == 7:
                if y % 2 == 0:
                    v = (q[(x + (y >> 1), -1)] + q[(x + (y >> 1) + 1, -1)] + 1) >> 1
                else:
                    v = (q[(x + (y >> 1), -1)] + 2 * q[(x + (y >> 1) + 1, -1)] + q[(x + (y >> 1) + 2, -1)] + 2) >> 2
            else:
                z = x + 2 * y
                if z > 13:
                    v = q[(-1, 7)]
                elif z == 13:
                    v = (q[(-1, 6)] + 3 * q[(-1, 7)] + 2) >> 2
                elif z % 2 == 0:
                    v = (q[(-1, y + (x >> 1))] + q[(-1, y + (x >> 1) + 1)] + 1) >> 1
                else:
                    v = (q[(-1, y + (x >> 1))] + 2 * q[(-1, y + (x >> 1) + 1)] + q[(-1, y + (x >> 1) + 2)] + 2) >> 2
            out[y][x] = v
    Y[y0:y0 + 8, x0:x0 + 8] = np.array(out, np.int32)


BLK8_FIRST = [0, 4, 8, 12]  # luma4x4BlkIdx of the first 4x4 block of 8x8 block b8

# B slice mb_type (Table 7-14): (partition shape, prediction of partition 0, of partition 1); predictions: 0 = L0, 1 = L1, 2 = Bi
B_MB = {1: ("16x16", 0, None), 2: ("16x16", 1, None), 3: ("16x16", 2, None), 4: ("16x8", 0, 0), 5: ("8x16", 0, 0), 6: ("16x8", 1, 1),
        7: ("8x16", 1, 1), 8: ("16x8", 0, 1), 9: ("8x16", 0, 1), 10: ("16x8", 1, 0), 11: ("8x16", 1, 0), 12: ("16x8", 0, 2),
        13: ("8x16", 0, 2), 14: ("16x8", 1, 2), 15: ("8x16", 1, 2), 16: ("16x8", 2, 0), 17: ("8x16", 2, 0), 18: ("16x8", 2, 1),
        19: ("8x16", 2, 1), 20: ("16x8", 2, 2), 21: ("8x16", 2, 2)}
# B sub_mb_type (Table 7-18): (sub-partition shape, prediction); 0 = direct
B_SUB = {1: ("8x8", 0), 2: ("8x8", 1), 3: ("8x8", 2), 4: ("8x4", 0), 5: ("4x8", 0), 6: ("8x4", 1), 7: ("4x8", 1), 8: ("8x4", 2),
         9: ("4x8", 2), 10: ("4x4", 0), 11: ("4x4", 1), 12: ("4x4", 2)}
P_SUB = {0: "8x8", 1: "8x4", 2: "4x8", 3: "4x4"}
SHAPE_PARTS = {"16x16": [(0, 0, 4, 4)], "16x8": [(0, 0, 4, 2), (0, 2, 4, 2)], "8x16": [(0, 0, 2, 4), (2, 0, 2, 4)],
               "8x8": [(0, 0, 2, 2)], "8x4": [(0, 0, 2, 1), (0, 1, 2, 1)], "4x8": [(0, 0, 1, 2), (1, 0, 1, 2)],
               "4x4": [(0, 0, 1, 1), (1, 0, 1, 1), (0, 1, 1, 1), (1, 1, 1, 1)]}


class MBInfo:
    __slots__ = ("typ", "i16", "qp", "cbp_luma", "cbp_chroma", "chroma_mode", "modes", "cbf_dc", "cbf_luma", "cbf_cdc", "cbf_cac",
                 "qp_delta_nz", "skip", "direct16", "intra", "ref0", "t8")

    def __init__(self):
        self.typ = None        # "I4", "I8" (both mb_type I_NxN), "I16", "P" (any inter macroblock)
        self.modes = [2] * 16
        self.cbf_dc = 0
        self.cbf_luma = [0] * 16
        self.cbf_cdc = [0, 0]
        self.cbf_cac = [[0] * 4, [0] * 4]
        self.cbp_luma = 0
        self.cbp_chroma = 0
        self.chroma_mode = 0
        self.qp_delta_nz = 0
        self.skip = False
        self.direct16 = False  # B_Skip or B_Direct_16x16 (mb_type's context increment)
        self.intra = False
        self.ref0 = False      # P_8x8ref0 (CAVLC): no ref_idx is sent, all zero
        self.t8 = False        # transform_size_8x8_flag
        self.i16 = 0
        self.qp = 0


class Pic:
    """A decoded (or being decoded) frame with the per-4x4-block motion data later pictures and the edge filter need."""
    _next_id = 0

    def __init__(self, sps, pps):
        self.sps, self.pps = sps, pps
        self.W, self.Hh = sps["mb_w"], sps["mb_h"]
        self.Y = np.zeros((self.Hh * 16, self.W * 16), np.int32)
        self.C = [np.zeros((self.Hh * 8, self.W * 8), np.int32) for _ in range(2)]
        self.mbs = [None] * (self.W * self.Hh)
        h4, w4 = self.Hh * 4, self.W * 4
        self.mv = np.zeros((2, h4, w4, 2), np.int32)
        self.ref = np.full((2, h4, w4), -1, np.int32)      # reference index into the slice's list, -1 = list not used
        self.refid = np.full((2, h4, w4), -1, np.int64)    # identity of the referenced picture, -1 = list not used
        self.mvd = np.zeros((2, h4, w4, 2), np.int32)      # |mvd| (context selection)
        self.direct = np.zeros((h4, w4), bool)
        self.done = np.zeros((h4, w4), bool)               # motion data of this list already derived (neighbour availability)
        self.nz = np.zeros((h4, w4), bool)                 # 4x4 luma block holds non-zero coefficients
        self.intra4 = np.zeros((h4, w4), bool)
        self.poc = 0
        self.frame_num = 0
        self.frame_num_wrap = 0
        self.is_ref = False
        self.id = Pic._next_id
        Pic._next_id += 1
        self.sample = -1

    def mb(self, mx, my):
        if mx < 0 or my < 0 or mx >= self.W or my >= self.Hh:
            return None
        return self.mbs[my * self.W + mx]


def _clip3(lo, hi, v):
    return lo if v < lo else (hi if v > hi else v)


# ----------------------------------------------------------------------------------------------------------------- interpolation (8.4.2.2)


def _tap(a, b, c, d, e, f):
    return a - 5 * b + 20 * c + 20 * d - 5 * e + f


def mc_luma(ref, x, y, w, h):
    """w x h luma samples predicted from `ref` (int32 plane) at the quarter-sample position (x, y) of the block's top-left."""
    xi, yi, fx, fy = x >> 2, y >> 2, x & 3, y & 3
    H, W = ref.shape
    if fx == 0 and fy == 0:
        xs = np.clip(np.arange(xi, xi + w), 0, W - 1)
        ys = np.clip(np.arange(yi, yi + h), 0, H - 1)
        return ref[np.ix_(ys, xs)]
    xs = np.clip(np.arange(xi - 2, xi + w + 4), 0, W - 1)
    ys = np.clip(np.arange(yi - 2, yi + h + 4), 0, H - 1)
    R = ref[np.ix_(ys, xs)]  # rows yi-2 .. yi+h+3, columns xi-2 .. xi+w+3
    G = R[2:2 + h, 2:2 + w]
    b1 = _tap(R[:, 0:w + 1], R[:, 1:w + 2], R[:, 2:w + 3], R[:, 3:w + 4], R[:, 4:w + 5], R[:, 5:w + 6])  # [row, c]: between xi+c, xi+c+1
    if fy == 0:
        b = np.clip((b1[2:2 + h, 0:w] + 16) >> 5, 0, 255)
        if fx == 2:
            return b
        return (b + (G if fx == 1 else R[2:2 + h, 3:3 + w]) + 1) >> 1
    h1 = _tap(R[0:h + 1], R[1:h + 2], R[2:h + 3], R[3:h + 4], R[4:h + 5], R[5:h + 6])  # [r, col]: between yi+r, yi+r+1
    if fx == 0:
        hh = np.clip((h1[0:h, 2:2 + w] + 16) >> 5, 0, 255)
        if fy == 2:
            return hh
        return (hh + (G if fy == 1 else R[3:3 + h, 2:2 + w]) + 1) >> 1
    if fx == 2 or fy == 2:
        j1 = _tap(b1[0:h + 1], b1[1:h + 2], b1[2:h + 3], b1[3:h + 4], b1[4:h + 5], b1[5:h + 6])
        j = np.clip((j1[0:h, 0:w] + 512) >> 10, 0, 255)
        if fx == 2 and fy == 2:
            return j
        if fx == 2:  # f (fy = 1): b + j; q (fy = 3): j + s
            o = np.clip((b1[(2 if fy == 1 else 3):(2 if fy == 1 else 3) + h, 0:w] + 16) >> 5, 0, 255)
        else:        # i (fx = 1): h + j; k (fx = 3): j + m
            c0 = 2 if fx == 1 else 3
            o = np.clip((h1[0:h, c0:c0 + w] + 16) >> 5, 0, 255)
        return (j + o + 1) >> 1
    # diagonal quarter positions e, g, p, r: a horizontal half sample (row of fy) and a vertical one (column of fx)
    r0 = 2 if fy == 1 else 3
    c0 = 2 if fx == 1 else 3
    bb = np.clip((b1[r0:r0 + h, 0:w] + 16) >> 5, 0, 255)
    hh = np.clip((h1[0:h, c0:c0 + w] + 16) >> 5, 0, 255)
    return (bb + hh + 1) >> 1


def mc_chroma(ref, x8, y8, w, h):
    """w x h chroma samples at the eighth-sample position (x8, y8)."""
    xi, yi, fx, fy = x8 >> 3, y8 >> 3, x8 & 7, y8 & 7
    H, W = ref.shape
    xs = np.clip(np.arange(xi, xi + w + 1), 0, W - 1)
    ys = np.clip(np.arange(yi, yi + h + 1), 0, H - 1)
    R = ref[np.ix_(ys, xs)]
    A, B, C, D = R[0:h, 0:w], R[0:h, 1:w + 1], R[1:h + 1, 0:w], R[1:h + 1, 1:w + 1]
    return ((8 - fx) * (8 - fy) * A + fx * (8 - fy) * B + (8 - fx) * fy * C + fx * fy * D + 32) >> 6


# ----------------------------------------------------------------------------------------------------------------- native engine
# The macroblock layer as host C++ in the package's library (csrc/h264dec.hip, `sa_h264_decode_slice`): the same state machine,
# several hundred pictures per second instead of 1-5. The Python classes below stay as its checker (tests/test_h264_native.py:
# planes AND motion data equal, picture by picture) and as the engine of last resort (`engine="python"`).
import ctypes as _C


class _CPic(_C.Structure):
    _fields_ = [("y", _C.c_void_p), ("cb", _C.c_void_p), ("cr", _C.c_void_p), ("mv", _C.c_void_p), ("ref", _C.c_void_p),
                ("refid", _C.c_void_p), ("intra4", _C.c_void_p), ("poc", _C.c_int32), ("id", _C.c_int32)]


class _CSlice(_C.Structure):
    _fields_ = [(n, _C.c_int32) for n in ("mb_w", "mb_h", "slice_type", "cabac", "qp", "chroma_qp_offset", "disable_deblock",
                                          "filter_offset_a", "filter_offset_b", "direct_spatial", "direct_8x8_inference")] + \
               [("nref", _C.c_int32 * 2), ("weighted_mode", _C.c_int32), ("luma_log2_denom", _C.c_int32), ("chroma_log2_denom", _C.c_int32),
                ("weights", _C.c_int32 * (2 * 32 * 3 * 2)), ("data_bit_offset", _C.c_int32), ("transform_8x8_mode", _C.c_int32),
                ("chroma_qp_offset_cr", _C.c_int32)]


class NativePic:
    """A picture decoded by the native engine: uint8 planes and the per-4x4 motion data in the layout of `sa_h264_pic`."""

    def __init__(self, sps, pps):
        self.sps, self.pps = sps, pps
        self.W, self.Hh = sps["mb_w"], sps["mb_h"]
        h4, w4 = self.Hh * 4, self.W * 4
        self.Y = np.zeros((self.Hh * 16, self.W * 16), np.uint8)
        self.C = [np.zeros((self.Hh * 8, self.W * 8), np.uint8) for _ in range(2)]
        self.mv = np.zeros((2, h4, w4, 2), np.int16)
        self.ref = np.full((2, h4, w4), -1, np.int8)
        self.refid = np.full((2, h4, w4), -1, np.int32)
        self.intra4 = np.zeros((h4, w4), np.uint8)
        self.poc = self.frame_num = self.frame_num_wrap = 0
        self.is_ref = False
        self.id = Pic._next_id
        Pic._next_id = (Pic._next_id + 1) & 0x3FFFFFFF
        self.sample = -1
        self.stats = {}

    def c_struct(self):
        return _CPic(self.Y.ctypes.data, self.C[0].ctypes.data, self.C[1].ctypes.data, self.mv.ctypes.data, self.ref.ctypes.data,
                     self.refid.ctypes.data, self.intra4.ctypes.data, int(self.poc), int(self.id))


def _decode_native(dec, h, r, cur, lists, payload):
    from .. import _lib

    lib = _lib.lib()
    sps, pps = dec.sps, dec.pps
    cs = _CSlice()
    cs.mb_w, cs.mb_h, cs.slice_type, cs.cabac = sps["mb_w"], sps["mb_h"], h["type"], pps["cabac"]
    cs.qp, cs.chroma_qp_offset = h["qp"], pps["chroma_qp_offset"]
    cs.disable_deblock, cs.filter_offset_a, cs.filter_offset_b = h["dbf"], h["off"][0], h["off"][1]
    cs.direct_spatial, cs.direct_8x8_inference = h["direct_spatial"], sps["direct_8x8_inference"]
    n0 = h["nref"][0] if h["type"] != 2 else 0
    n1 = h["nref"][1] if h["type"] == 1 else 0
    cs.nref[0], cs.nref[1] = n0, n1
    cs.weighted_mode = 1 if h["pwt"] is not None else (2 if (h["type"] == 1 and pps["weighted_bipred_idc"] == 2) else 0)
    if h["pwt"] is not None:
        ld, cd, tabs = h["pwt"]
        cs.luma_log2_denom, cs.chroma_log2_denom = ld, cd
        for lst, t in enumerate(tabs):
            for i, (lw, lo, cw, co) in enumerate(t):
                base = ((lst * 32 + i) * 3) * 2
                cs.weights[base:base + 6] = [lw, lo, cw[0], co[0], cw[1], co[1]]
    cs.data_bit_offset = r.p
    cs.transform_8x8_mode, cs.chroma_qp_offset_cr = int(pps.get("transform8x8", 0)), pps.get("chroma_qp_offset2", pps["chroma_qp_offset"])
    arrs = []
    for lst, n in ((0, n0), (1, n1)):
        a = (_CPic * max(n, 1))()
        for i in range(n):
            p = lists[lst][i]
            if p is not None:
                a[i] = p.c_struct()
        arrs.append(a)
    cc = cur.c_struct()
    stats = (_C.c_int32 * 8)()
    buf = (_C.c_uint8 * len(payload)).from_buffer_copy(payload)
    rc = lib.sa_h264_decode_slice(_C.byref(cs), buf, len(payload), arrs[0], arrs[1], _C.byref(cc), stats)
    if rc != 0:
        msg = lib.sa_last_error()
        msg = msg.decode() if msg else ""
        if "not implemented" in msg:
            raise Unsupported(msg)
        raise AssertionError(f"{msg} ({'PBI'[h['type']]} picture, sample {cur.sample})")
    cur.stats = {"I4": stats[0], "I16": stats[1], "skip": stats[2], "inter": stats[3], "type": "PBI"[h["type"]], "slice_qp": h["qp"]}
    if pps.get("transform8x8"):
        cur.stats["I8"], cur.stats["t8"] = stats[5], stats[6]
    cur.stats["bits_left"] = stats[4]
    cur.plain_copies = stats[7]



# ----------------------------------------------------------------------------------------------------------------- the decoder


class H264Decoder:
    """Feed the NAL units of consecutive samples (decoding order); `decode_sample` returns the decoded Pic of that sample."""

    def __init__(self, sps, pps, engine="native"):
        if engine not in ("native", "python"):
            raise ValueError(f"engine {engine!r}: 'native' (csrc/h264dec.hip) or 'python'")
        self.engine = engine
        if sps["profile"] not in (66, 77, 100):
            raise Unsupported(f"profile_idc {sps['profile']}: Baseline / Main / High (8-bit 4:2:0) only")
        if pps["constrained_intra"]:
            raise Unsupported("constrained_intra_pred_flag = 1 is not implemented")
        self.sps, self.pps = sps, pps
        self.dpb = []              # short-term reference pictures
        self.prev_poc_msb = 0
        self.prev_poc_lsb = 0
        self.prev_frame_num = 0
        self.stats = {}

    # ---- 7.3.3 slice header
    def _header(self, nal):
        sps, pps = self.sps, self.pps
        payload = rbsp(nal)
        r = Bits(payload + b"\x00" * 8)
        h = {"n_bits": len(payload) * 8}
        h["first_mb"] = r.ue()
        st = r.ue()
        h["type"] = st % 5  # 0 P, 1 B, 2 I
        if h["type"] > 2:
            raise Unsupported("SP / SI slices")
        r.ue()
        h["frame_num"] = r.u(sps["log2_max_frame_num"])
        h["idr"] = (nal[0] & 31) == 5
        h["ref_idc"] = (nal[0] >> 5) & 3
        if h["idr"]:
            r.ue()
        if sps["poc_type"] != 0:
            raise Unsupported("pic_order_cnt_type != 0")
        h["poc_lsb"] = r.u(sps["log2_max_poc_lsb"])
        if pps["pic_order_present"]:
            r.se()
        if pps["redundant_pic_cnt"]:
            r.ue()
        h["direct_spatial"] = r.u(1) if h["type"] == 1 else 1
        n0, n1 = pps["num_ref_idx_default"]
        if h["type"] in (0, 1) and r.u(1):
            n0 = r.ue() + 1
            if h["type"] == 1:
                n1 = r.ue() + 1
        h["nref"] = (n0, n1)
        h["rplm"] = [[], []]
        for lst in range(2 if h["type"] == 1 else (1 if h["type"] == 0 else 0)):
            if r.u(1):
                while True:
                    op = r.ue()
                    if op == 3:
                        break
                    if op == 2:
                        raise Unsupported("long-term reference pictures")
                    h["rplm"][lst].append((op, r.ue()))
        h["pwt"] = None
        if (pps["weighted_pred"] and h["type"] == 0) or (pps["weighted_bipred_idc"] == 1 and h["type"] == 1):
            ld, cd = r.ue(), r.ue()
            tabs = []
            for lst in range(2 if h["type"] == 1 else 1):
                t = []
                for _ in range(h["nref"][lst]):
                    lw, lo, cw, co = 1 << ld, 0, [1 << cd, 1 << cd], [0, 0]
                    if r.u(1):
                        lw, lo = r.se(), r.se()
                    if r.u(1):
                        for c in range(2):
                            cw[c], co[c] = r.se(), r.se()
                    t.append((lw, lo, cw, co))
                tabs.append(t)
            h["pwt"] = (ld, cd, tabs)
        h["mmco"] = []
        if h["ref_idc"]:
            if h["idr"]:
                r.u(1)
                if r.u(1):
                    raise Unsupported("long-term reference pictures")
            elif r.u(1):
                while True:
                    op = r.ue()
                    if op == 0:
                        break
                    if op != 1:
                        raise Unsupported(f"memory_management_control_operation {op}")
                    h["mmco"].append((op, r.ue()))
        h["cabac_init_idc"] = r.ue() if (h["type"] != 2 and pps["cabac"]) else None
        if h["cabac_init_idc"] not in (None, 0):
            raise Unsupported(f"cabac_init_idc {h['cabac_init_idc']}: only the tables of cabac_init_idc 0 are held")
        h["qp"] = pps["pic_init_qp"] + r.se()
        h["dbf"], h["off"] = 0, (0, 0)
        if pps["deblocking_control"]:
            h["dbf"] = r.ue()
            if h["dbf"] != 1:
                h["off"] = (2 * r.se(), 2 * r.se())
        if h["dbf"] == 2:
            raise Unsupported("disable_deblocking_filter_idc 2")
        if pps["cabac"]:
            while r.p & 7:
                assert r.u(1) == 1, "cabac_alignment_one_bit"
        if h["first_mb"] != 0:
            raise Unsupported("several slices per picture")
        return h, r

    # ---- 8.2.4 reference picture lists
    def _ref_lists(self, h, cur):
        sps = self.sps
        max_fn = 1 << sps["log2_max_frame_num"]
        for p in self.dpb:
            p.frame_num_wrap = p.frame_num - max_fn if p.frame_num > cur.frame_num else p.frame_num
        lists = [[], []]
        if h["type"] == 0:
            lists[0] = sorted(self.dpb, key=lambda p: -p.frame_num_wrap)
        elif h["type"] == 1:
            before = sorted([p for p in self.dpb if p.poc < cur.poc], key=lambda p: -p.poc)
            after = sorted([p for p in self.dpb if p.poc > cur.poc], key=lambda p: p.poc)
            lists[0], lists[1] = before + after, after + before
            if len(lists[1]) > 1 and [p.id for p in lists[0]] == [p.id for p in lists[1]]:
                lists[1][0], lists[1][1] = lists[1][1], lists[1][0]
        out = []
        for lst in range(2):
            n = h["nref"][lst]
            L = lists[lst][:n] + [None] * max(0, n - len(lists[lst]))
            if h["rplm"][lst]:
                pred = cur.frame_num
                idx = 0
                for op, v in h["rplm"][lst]:
                    d = v + 1
                    if op == 0:
                        pred = pred - d + (max_fn if pred - d < 0 else 0)
                    else:
                        pred = pred + d - (max_fn if pred + d >= max_fn else 0)
                    pn = pred - max_fn if pred > cur.frame_num else pred
                    target = next((p for p in self.dpb if p.frame_num_wrap == pn), None)
                    assert target is not None, f"reference list modification names picture number {pn}, not in the buffer"
                    L = L[:idx] + [target] + [p for p in L[idx:] if p is None or p.id != target.id]
                    L = (L + [None] * n)[:n]
                    idx += 1
            out.append(L)
        return out

    # ---- 8.2.5 marking
    def _mark(self, h, cur):
        if not h["ref_idc"]:
            return
        if h["idr"]:
            self.dpb = []
        elif h["mmco"]:
            for _, v in h["mmco"]:
                pn = cur.frame_num - (v + 1)
                for p in self.dpb:
                    if p.frame_num_wrap == pn:
                        self.dpb.remove(p)
                        break
        elif len(self.dpb) >= max(self.sps["num_ref_frames"], 1):
            self.dpb.remove(min(self.dpb, key=lambda p: p.frame_num_wrap))
        cur.is_ref = True
        self.dpb.append(cur)
        assert len(self.dpb) <= max(self.sps["num_ref_frames"], 1), "more reference pictures than num_ref_frames"

    # ---- one sample
    def decode_sample(self, nals, sample=-1):
        slices = [n for n in nals if n and (n[0] & 31) in (1, 5)]
        if len(slices) != 1:
            raise Unsupported(f"{len(slices)} slices in one sample")
        nal = slices[0]
        h, r = self._header(nal)
        sps, pps = self.sps, self.pps
        cur = (NativePic if self.engine == "native" else Pic)(sps, pps)
        cur.sample = sample
        cur.frame_num = h["frame_num"]
        # picture order count, type 0 (8.2.1.1)
        max_lsb = 1 << sps["log2_max_poc_lsb"]
        if h["idr"]:
            self.prev_poc_msb = self.prev_poc_lsb = 0
            self.dpb = []
        lsb = h["poc_lsb"]
        if lsb < self.prev_poc_lsb and self.prev_poc_lsb - lsb >= max_lsb // 2:
            msb = self.prev_poc_msb + max_lsb
        elif lsb > self.prev_poc_lsb and lsb - self.prev_poc_lsb > max_lsb // 2:
            msb = self.prev_poc_msb - max_lsb
        else:
            msb = self.prev_poc_msb
        cur.poc = msb + lsb
        if h["ref_idc"]:
            self.prev_poc_msb, self.prev_poc_lsb = msb, lsb
        for p in self.dpb:
            p.frame_num_wrap = p.frame_num - (1 << sps["log2_max_frame_num"]) if p.frame_num > cur.frame_num else p.frame_num
        lists = self._ref_lists(h, cur) if h["type"] != 2 else [[], []]
        if self.engine == "native":
            _decode_native(self, h, r, cur, lists, rbsp(nal))
        else:
            (_SliceDecoder if pps["cabac"] else _CavlcSliceDecoder)(self, h, r, cur, lists, payload_bits=h["n_bits"], payload=rbsp(nal)).run()
        self._mark(h, cur)
        return cur


class _SliceDecoder:
    def __init__(self, dec, h, r, cur, lists, payload_bits=0, payload=b""):
        self.dec, self.h, self.r, self.pic, self.lists = dec, h, r, cur, lists
        self.sps, self.pps = dec.sps, dec.pps
        self.stype = h["type"]
        self.cab = Cabac(r, h["qp"], CTX_I_HIGH if self.stype == 2 else CTX_PB0_HIGH) if self.pps["cabac"] else None
        self._payload = payload
        self.qp = h["qp"]
        self.prev_qp_delta_nz = 0
        self.stats = {"I4": 0, "I16": 0, "skip": 0, "inter": 0, "type": "PBI"[self.stype], "slice_qp": h["qp"]}
        self.t8mode = bool(self.pps.get("transform8x8"))
        if self.t8mode:
            self.stats["I8"] = self.stats["t8"] = 0
        # implicit bi-prediction weights per (ref0, ref1) index pair (8.4.2.3.1)
        self.implicit = {}
        if self.stype == 1 and self.pps["weighted_bipred_idc"] == 2:
            for i, p0 in enumerate(lists[0]):
                for j, p1 in enumerate(lists[1]):
                    if p0 is None or p1 is None:
                        continue
                    tb = _clip3(-128, 127, cur.poc - p0.poc)
                    td = _clip3(-128, 127, p1.poc - p0.poc)
                    w0 = w1 = 32
                    if td != 0:
                        tx = int((16384 + abs(td // 2 if td > 0 else -((-td) // 2))) / td)
                        dsf = _clip3(-1024, 1023, (tb * tx + 32) >> 6)
                        if -64 <= (dsf >> 2) <= 128:
                            w0, w1 = 64 - (dsf >> 2), dsf >> 2
                    self.implicit[(i, j)] = (w0, w1)

    # ------------------------------------------------------------------ neighbour motion data
    def _nb(self, lst, x4, y4):
        """(available, refIdx, mv) of the 4x4 block at (x4, y4) for list `lst`; refIdx -1 = intra / list not used."""
        pic = self.pic
        if x4 < 0 or y4 < 0 or x4 >= pic.W * 4 or y4 >= pic.Hh * 4 or not pic.done[y4, x4]:
            return False, -1, (0, 0)
        rf = int(pic.ref[lst, y4, x4])
        if rf < 0:
            return True, -1, (0, 0)
        return True, rf, (int(pic.mv[lst, y4, x4, 0]), int(pic.mv[lst, y4, x4, 1]))

    def _mvp(self, lst, x4, y4, w4, h4, ref, shape=None, part=0):
        """8.4.1.3: motion vector prediction for the partition at (x4, y4) of w4 x h4 blocks with reference index `ref`."""
        a = self._nb(lst, x4 - 1, y4)
        b = self._nb(lst, x4, y4 - 1)
        c = self._nb(lst, x4 + w4, y4 - 1)
        if not c[0]:
            c = self._nb(lst, x4 - 1, y4 - 1)
        if shape == "16x8":
            if part == 0 and b[1] == ref:
                return b[2]
            if part == 1 and a[1] == ref:
                return a[2]
        elif shape == "8x16":
            if part == 0 and a[1] == ref:
                return a[2]
            if part == 1 and c[1] == ref:
                return c[2]
        if not b[0] and not c[0] and a[0]:
            b = c = a
        same = [n for n in (a, b, c) if n[1] == ref]
        if len(same) == 1:
            return same[0][2]
        return (sorted((a[2][0], b[2][0], c[2][0]))[1], sorted((a[2][1], b[2][1], c[2][1]))[1])

    def _set_motion(self, lst, x4, y4, w4, h4, ref, mv, mvd=(0, 0)):
        pic = self.pic
        pic.ref[lst, y4:y4 + h4, x4:x4 + w4] = ref
        pic.refid[lst, y4:y4 + h4, x4:x4 + w4] = self.lists[lst][ref].id if ref >= 0 else -1
        pic.mv[lst, y4:y4 + h4, x4:x4 + w4] = mv
        pic.mvd[lst, y4:y4 + h4, x4:x4 + w4] = (abs(mvd[0]), abs(mvd[1]))

    # ------------------------------------------------------------------ direct prediction (8.4.1.2)
    def _col(self, x4, y4):
        """co-located block in RefPicList1[0]: (refid of its reference or None when intra, mv, the list it came from)"""
        col = self.lists[1][0]
        if col.intra4[y4, x4]:
            return None, (0, 0), 0, -1
        for lst in (0, 1):
            if col.ref[lst, y4, x4] >= 0:
                return int(col.refid[lst, y4, x4]), (int(col.mv[lst, y4, x4, 0]), int(col.mv[lst, y4, x4, 1])), lst, int(col.ref[lst, y4, x4])
        return None, (0, 0), 0, -1

    def _direct(self, mx, my, quads):
        """Derive the motion data of the direct-predicted 8x8 quadrants `quads` of macroblock (mx, my) and write it to the picture
        (with `done` still unset for the whole macroblock: the spatial neighbours are the macroblock's)."""
        pic = self.pic
        X4, Y4 = mx * 4, my * 4
        inf8 = self.sps["direct_8x8_inference"]
        if self.h["direct_spatial"]:
            refs, mvs = [-1, -1], [(0, 0), (0, 0)]
            for lst in (0, 1):
                a = self._nb(lst, X4 - 1, Y4)
                b = self._nb(lst, X4, Y4 - 1)
                c = self._nb(lst, X4 + 4, Y4 - 1)
                if not c[0]:
                    c = self._nb(lst, X4 - 1, Y4 - 1)
                rr = -1
                for n in (a, b, c):
                    if n[1] >= 0 and (rr < 0 or n[1] < rr):
                        rr = n[1]
                refs[lst] = rr
            zero_pred = refs[0] < 0 and refs[1] < 0
            if zero_pred:
                refs = [0, 0]
            else:
                for lst in (0, 1):
                    if refs[lst] >= 0:
                        mvs[lst] = self._mvp(lst, X4, Y4, 4, 4, refs[lst])
            for q in quads:
                qx, qy = (q & 1) * 2, (q >> 1) * 2
                blocks = [(qx, qy, 2, 2, (qx * 3) // 2, (qy * 3) // 2)] if inf8 else [(qx + i, qy + j, 1, 1, qx + i, qy + j) for j in range(2) for i in range(2)]
                for bx, by, w, hh, cx, cy in blocks:
                    col_ref, col_mv, _, col_idx = self._col(X4 + cx, Y4 + cy)
                    col_zero = col_ref is not None and col_idx == 0 and abs(col_mv[0]) <= 1 and abs(col_mv[1]) <= 1
                    for lst in (0, 1):
                        mv = mvs[lst]
                        if zero_pred or refs[lst] < 0 or (refs[lst] == 0 and col_zero):
                            mv = (0, 0)
                        self._set_motion(lst, X4 + bx, Y4 + by, w, hh, refs[lst], mv)
                    pic.direct[Y4 + by:Y4 + by + hh, X4 + bx:X4 + bx + w] = True
        else:
            l0, l1 = self.lists
            for q in quads:
                qx, qy = (q & 1) * 2, (q >> 1) * 2
                blocks = [(qx, qy, 2, 2, (qx * 3) // 2, (qy * 3) // 2)] if inf8 else [(qx + i, qy + j, 1, 1, qx + i, qy + j) for j in range(2) for i in range(2)]
                for bx, by, w, hh, cx, cy in blocks:
                    col_ref, col_mv, _, _ = self._col(X4 + cx, Y4 + cy)
                    r0 = 0
                    if col_ref is not None:
                        r0 = next((i for i, p in enumerate(l0) if p is not None and p.id == col_ref), None)
                        assert r0 is not None, "temporal direct: the co-located block's reference is not in RefPicList0"
                    p0, p1 = l0[r0], l1[0]
                    tb = _clip3(-128, 127, pic.poc - p0.poc)
                    td = _clip3(-128, 127, p1.poc - p0.poc)
                    if td == 0:
                        mv0, mv1 = col_mv, (0, 0)
                    else:
                        tx = int((16384 + abs(int(td / 2))) / td)
                        dsf = _clip3(-1024, 1023, (tb * tx + 32) >> 6)
                        mv0 = ((dsf * col_mv[0] + 128) >> 8, (dsf * col_mv[1] + 128) >> 8)
                        mv1 = (mv0[0] - col_mv[0], mv0[1] - col_mv[1])
                    self._set_motion(0, X4 + bx, Y4 + by, w, hh, r0, mv0)
                    self._set_motion(1, X4 + bx, Y4 + by, w, hh, 0, mv1)
                    pic.direct[Y4 + by:Y4 + by + hh, X4 + bx:X4 + bx + w] = True

    # ------------------------------------------------------------------ CABAC syntax elements of inter macroblocks
    def _intra_mb_type(self, base):
        """mb_type suffix of an intra macroblock inside a P (base 17) or B (base 32) slice -> ("I4" | "I16", i16 fields)"""
        cab = self.cab
        if cab.decision(base) == 0:
            return "I4", None
        if cab.terminate():
            raise Unsupported("I_PCM macroblocks")
        ac = cab.decision(base + 1)
        chroma = 0
        if cab.decision(base + 2):
            chroma = 1 + cab.decision(base + 2)
        pm = 2 * cab.decision(base + 3)
        pm += cab.decision(base + 3)
        return "I16", (pm, 15 if ac else 0, chroma)

    def _sub_mb_types(self):
        """the four sub_mb_type elements of P_8x8 / B_8x8 -> [(sub-partition shape | "direct", prediction)]"""
        cab, stype = self.cab, self.stype
        subs = []
        for q in range(4):
            if stype == 0:
                if cab.decision(21):
                    st = 0
                elif cab.decision(22) == 0:
                    st = 1
                else:
                    st = 3 if cab.decision(23) == 0 else 2
                subs.append((P_SUB[st], 0))
            else:
                if cab.decision(36) == 0:
                    subs.append(("direct", None))
                    continue
                if cab.decision(37) == 0:
                    st = 1 + cab.decision(39)
                else:
                    st = 3
                    if cab.decision(38):
                        if cab.decision(39):
                            st = 11 + cab.decision(39)
                            subs.append(B_SUB[st])
                            continue
                        st += 4
                    st += 2 * cab.decision(39)
                    st += cab.decision(39)
                subs.append(B_SUB[st])
        return subs

    def _ref_idx(self, lst, x4, y4):
        pic, cab = self.pic, self.cab
        ctx = 0
        if x4 > 0 and pic.ref[lst, y4, x4 - 1] > 0 and not pic.direct[y4, x4 - 1]:
            ctx += 1
        if y4 > 0 and pic.ref[lst, y4 - 1, x4] > 0 and not pic.direct[y4 - 1, x4]:
            ctx += 2
        v = 0
        while cab.decision(54 + ctx):
            v += 1
            ctx = (ctx >> 2) + 4
            assert v < 32, "ref_idx runaway: the CABAC decode lost synchronisation"
        return v

    def _mvd(self, lst, comp, x4, y4):
        pic, cab = self.pic, self.cab
        s = 0
        if x4 > 0:
            s += int(pic.mvd[lst, y4, x4 - 1, comp])
        if y4 > 0:
            s += int(pic.mvd[lst, y4 - 1, x4, comp])
        base = 40 if comp == 0 else 47
        if not cab.decision(base + (0 if s < 3 else (2 if s > 32 else 1))):
            return 0
        v, c = 1, base + 3
        while v < 9 and cab.decision(c):
            if v < 4:
                c += 1
            v += 1
        if v >= 9:
            k = 3
            while cab.bypass():
                v += 1 << k
                k += 1
                assert k < 24, "mvd runaway: the CABAC decode lost synchronisation"
            while k:
                k -= 1
                v += cab.bypass() << k
        return -v if cab.bypass() else v

    # ------------------------------------------------------------------ residual (as the intra module's, with the inter rules)
    def _residual_block(self, m, A, Bn, cat, n_coef, bx=0, by=0, comp=0):
        cab = self.cab

        def cbf_of(nb, blk):
            if cat == 0:
                return nb.cbf_dc if nb.typ == "I16" else None
            if cat in (1, 2):
                x, y = BLK_XY[blk]
                return nb.cbf_luma[blk] if (nb.cbp_luma >> ((y >> 1) * 2 + (x >> 1))) & 1 else None
            if cat == 3:
                return nb.cbf_cdc[comp] if nb.cbp_chroma else None
            return nb.cbf_cac[comp][blk] if nb.cbp_chroma == 2 else None

        def flag(dx, dy, size):
            x, y = bx + dx, by + dy
            if 0 <= x < size and 0 <= y < size:
                nb, xx, yy = m, x, y
            else:
                nb = A if dx else Bn
                if nb is None:
                    return 1 if m.intra else 0
                xx, yy = x % size, y % size
            blk = XY_BLK[(xx, yy)] if cat in (1, 2) else (yy * 2 + xx if cat == 4 else 0)
            v = cbf_of(nb, blk)
            return 0 if v is None else v

        size = 4 if cat in (1, 2) else (2 if cat == 4 else 1)
        if cat in (0, 3):
            bx = by = 0
        fa, fb = flag(-1, 0, size), flag(0, -1, size)
        coef = [0] * n_coef
        if not cab.decision(85 + CAT_CBF[cat] + fa + 2 * fb):
            return coef, 0
        sig = []
        last = n_coef - 1
        for i in range(n_coef - 1):
            inc_ = min(i, 2) if cat == 3 else i
            if cab.decision(105 + CAT_SIG[cat] + inc_):
                sig.append(i)
                if cab.decision(166 + CAT_SIG[cat] + inc_):
                    last = None
                    break
        if last is not None:
            sig.append(n_coef - 1)
        eq1, gt1 = 0, 0
        for i in reversed(sig):
            ctx0 = 227 + CAT_ABS[cat]
            inc_ = 0 if gt1 else min(4, 1 + eq1)
            v = 0
            if cab.decision(ctx0 + inc_):
                inc2 = 5 + min(4 - (1 if cat == 3 else 0), gt1)
                v = 1
                while v < 14 and cab.decision(ctx0 + inc2):
                    v += 1
                if v == 14:
                    k = 0
                    while cab.bypass():
                        v += 1 << k
                        k += 1
                        assert k < 24, "coefficient runaway: the CABAC decode lost synchronisation"
                    while k:
                        k -= 1
                        v += cab.bypass() << k
            if v == 0:
                eq1 += 1
            else:
                gt1 += 1
            coef[i] = -(v + 1) if cab.bypass() else v + 1
        return coef, 1

    # ------------------------------------------------------------------ inter prediction of one macroblock (8.4.2)
    def _predict_inter(self, mx, my, parts):
        pic, h = self.pic, self.h
        X4, Y4 = mx * 4, my * 4
        explicit = h["pwt"] is not None
        for bx, by, w4, h4 in parts:
            x4, y4 = X4 + bx, Y4 + by
            w, hh = w4 * 4, h4 * 4
            preds = []
            for lst in (0, 1):
                rf = int(pic.ref[lst, y4, x4])
                if rf < 0:
                    continue
                rp = self.lists[lst][rf]
                assert rp is not None, "prediction from an empty reference list entry"
                mvx, mvy = int(pic.mv[lst, y4, x4, 0]), int(pic.mv[lst, y4, x4, 1])
                py = mc_luma(rp.Y, x4 * 16 + mvx, y4 * 16 + mvy, w, hh)
                pc = [mc_chroma(rp.C[c], x4 * 16 + mvx, y4 * 16 + mvy, w // 2, hh // 2) for c in range(2)]
                preds.append((lst, rf, py, pc))
            assert preds, "inter partition without a reference"
            ys, xs = slice(y4 * 4, y4 * 4 + hh), slice(x4 * 4, x4 * 4 + w)
            yc, xc = slice(y4 * 2, y4 * 2 + hh // 2), slice(x4 * 2, x4 * 2 + w // 2)
            if len(preds) == 1:
                lst, rf, py, pc = preds[0]
                if explicit:
                    ld, cd, tabs = h["pwt"]
                    lw, lo, cw, co = tabs[lst][rf]
                    py = np.clip((((py * lw + (1 << (ld - 1))) >> ld) if ld >= 1 else py * lw) + lo, 0, 255)
                    pc = [np.clip((((pc[c] * cw[c] + (1 << (cd - 1))) >> cd) if cd >= 1 else pc[c] * cw[c]) + co[c], 0, 255) for c in range(2)]
                pic.Y[ys, xs] = py
                for c in range(2):
                    pic.C[c][yc, xc] = pc[c]
            else:
                (_, r0, y0, c0), (_, r1, y1, c1) = preds
                if explicit:
                    ld, cd, tabs = h["pwt"]
                    lw0, lo0, cw0, co0 = tabs[0][r0]
                    lw1, lo1, cw1, co1 = tabs[1][r1]
                    pic.Y[ys, xs] = np.clip(((y0 * lw0 + y1 * lw1 + (1 << ld)) >> (ld + 1)) + ((lo0 + lo1 + 1) >> 1), 0, 255)
                    for c in range(2):
                        pic.C[c][yc, xc] = np.clip(((c0[c] * cw0[c] + c1[c] * cw1[c] + (1 << cd)) >> (cd + 1)) + ((co0[c] + co1[c] + 1) >> 1), 0, 255)
                elif self.implicit:
                    w0, w1 = self.implicit[(r0, r1)]
                    pic.Y[ys, xs] = np.clip((y0 * w0 + y1 * w1 + 32) >> 6, 0, 255)
                    for c in range(2):
                        pic.C[c][yc, xc] = np.clip((c0[c] * w0 + c1[c] * w1 + 32) >> 6, 0, 255)
                else:
                    pic.Y[ys, xs] = (y0 + y1 + 1) >> 1
                    for c in range(2):
                        pic.C[c][yc, xc] = (c0[c] + c1[c] + 1) >> 1

    # ------------------------------------------------------------------ the macroblock loop
    def run(self):
        pic, cab, h = self.pic, self.cab, self.h
        n_mb = pic.W * pic.Hh
        for addr in range(n_mb):
            mx, my = addr % pic.W, addr // pic.W
            self._macroblock(addr, mx, my)
            end = cab.terminate()
            assert end == (1 if addr == n_mb - 1 else 0), (f"end_of_slice_flag = {end} at macroblock {addr} of {n_mb} "
                                                           f"({self.stats['type']} picture): the CABAC decode lost synchronisation")
        self.stats["bits_left"] = h["n_bits"] - self.r.p
        assert -16 <= self.stats["bits_left"] <= 16, self.stats
        if h["dbf"] != 1:
            deblock_inter(pic, h["off"][0], h["off"][1])
        pic.stats = self.stats

    def _skip_mb(self, addr, mx, my, m):
        """P_Skip / B_Skip: inferred motion, no residual"""
        pic, stype = self.pic, self.stype
        X4, Y4 = mx * 4, my * 4
        m.skip, m.typ, m.qp = True, "P", self.qp
        self.prev_qp_delta_nz = 0
        self.stats["skip"] += 1
        if stype == 0:
            a, b = self._nb(0, X4 - 1, Y4), self._nb(0, X4, Y4 - 1)
            if not a[0] or not b[0] or (a[1] == 0 and a[2] == (0, 0)) or (b[1] == 0 and b[2] == (0, 0)):
                mv = (0, 0)
            else:
                mv = self._mvp(0, X4, Y4, 4, 4, 0)
            self._set_motion(0, X4, Y4, 4, 4, 0, mv)
            parts = [(0, 0, 4, 4)]
        else:
            m.direct16 = True
            self._direct(mx, my, (0, 1, 2, 3))
            parts = [(0, 0, 2, 2), (2, 0, 2, 2), (0, 2, 2, 2), (2, 2, 2, 2)] if self.sps["direct_8x8_inference"] else \
                [(i, j, 1, 1) for j in range(4) for i in range(4)]
        pic.done[Y4:Y4 + 4, X4:X4 + 4] = True
        pic.mbs[addr] = m
        self._predict_inter(mx, my, parts)

    def _macroblock(self, addr, mx, my):
        pic, cab, stype = self.pic, self.cab, self.stype
        X4, Y4 = mx * 4, my * 4
        m = MBInfo()
        A, Bn = pic.mb(mx - 1, my), pic.mb(mx, my - 1)
        mb_done = pic.done[Y4:Y4 + 4, X4:X4 + 4]
        parts = None
        if stype != 2:
            ctx = (11 if stype == 0 else 24) + (1 if (A is not None and not A.skip) else 0) + (1 if (Bn is not None and not Bn.skip) else 0)
            if cab.decision(ctx):
                self._skip_mb(addr, mx, my, m)
                return
        # ---- mb_type
        inter = None  # (shape, [prediction of partition 0, 1]) or "8x8"
        if stype == 2:
            inc = (1 if (A is not None and A.typ not in ("I4", "I8")) else 0) + (1 if (Bn is not None and Bn.typ not in ("I4", "I8")) else 0)
            if cab.decision(3 + inc) == 0:
                m.typ = "I4"
            else:
                if cab.terminate():
                    raise Unsupported("I_PCM macroblocks")
                m.typ = "I16"
                ac = cab.decision(3 + 3)
                chroma = 0
                if cab.decision(3 + 4):
                    chroma = 1 + cab.decision(3 + 5)
                pm = 2 * cab.decision(3 + 6)
                pm += cab.decision(3 + 7)
                m.i16, m.cbp_luma, m.cbp_chroma = pm, (15 if ac else 0), chroma
        elif stype == 0:
            if cab.decision(14) == 0:
                if cab.decision(15) == 0:
                    inter = "8x8" if cab.decision(16) else ("16x16", [0, None])
                else:
                    inter = ("16x8", [0, 0]) if cab.decision(17) else ("8x16", [0, 0])
            else:
                m.typ, f = self._intra_mb_type(17)
                if f:
                    m.i16, m.cbp_luma, m.cbp_chroma = f
        else:
            inc = (1 if (A is not None and not A.direct16) else 0) + (1 if (Bn is not None and not Bn.direct16) else 0)
            if cab.decision(27 + inc) == 0:
                t = 0
            elif cab.decision(27 + 3) == 0:
                t = 1 + cab.decision(27 + 5)
            else:
                bits = cab.decision(27 + 4) << 3
                bits |= cab.decision(27 + 5) << 2
                bits |= cab.decision(27 + 5) << 1
                bits |= cab.decision(27 + 5)
                if bits < 8:
                    t = bits + 3
                elif bits == 13:
                    t = 23
                elif bits == 14:
                    t = 11
                elif bits == 15:
                    t = 22
                else:
                    t = ((bits << 1) | cab.decision(27 + 5)) - 4
            if t == 23:
                m.typ, f = self._intra_mb_type(32)
                if f:
                    m.i16, m.cbp_luma, m.cbp_chroma = f
            elif t == 22:
                inter = "8x8"
            elif t == 0:
                inter = "direct"
                m.direct16 = True
            else:
                shape, p0, p1 = B_MB[t]
                inter = (shape, [p0, p1])
        if inter is None:
            self._intra_tail(addr, mx, my, m, A, Bn)
        else:
            self._inter_tail(addr, mx, my, m, A, Bn, inter)

    def _inter_tail(self, addr, mx, my, m, A, Bn, inter):
        """motion data, prediction, coded_block_pattern, mb_qp_delta and residual of an inter macroblock; `inter` = "direct",
        "8x8" or (partition shape, [prediction of partition 0, 1])"""
        pic, stype = self.pic, self.stype
        X4, Y4 = mx * 4, my * 4
        mb_done = pic.done[Y4:Y4 + 4, X4:X4 + 4]
        m.typ = "P"
        self.stats["inter"] += 1
        # ---- motion data
        inf8 = bool(self.sps["direct_8x8_inference"])
        no_sub8 = inf8 if inter == "direct" else True  # noSubMbPartSizeLessThan8x8Flag (7.3.5), refined for 8x8 below
        if inter == "direct":
            self._direct(mx, my, (0, 1, 2, 3))
            parts = [(0, 0, 2, 2), (2, 0, 2, 2), (0, 2, 2, 2), (2, 2, 2, 2)] if self.sps["direct_8x8_inference"] else \
                [(i, j, 1, 1) for j in range(4) for i in range(4)]
            mb_done[:] = True
        else:
            if inter == "8x8":
                subs = self._sub_mb_types()
                no_sub8 = all((sh == "8x8") or (sh == "direct" and inf8) for sh, _ in subs)
                # partitions: (x, y, w, h, prediction, shape for the directional rule, index for it, reference group)
                plist = []
                dq = [q for q in range(4) if subs[q][0] == "direct"]
                if dq:
                    self._direct(mx, my, dq)
                for q in range(4):
                    if subs[q][0] == "direct":
                        continue
                    qx, qy = (q & 1) * 2, (q >> 1) * 2
                    for (sx, sy, w, hh) in SHAPE_PARTS[subs[q][0]]:
                        plist.append((qx + sx, qy + sy, w, hh, subs[q][1], None, 0, q))
                refgroups = [(q, (q & 1) * 2, (q >> 1) * 2, 2, 2, subs[q][1]) for q in range(4) if subs[q][0] != "direct"]
            else:
                shape, preds = inter
                plist, refgroups = [], []
                for i, (sx, sy, w, hh) in enumerate(SHAPE_PARTS[shape]):
                    plist.append((sx, sy, w, hh, preds[i], shape if shape != "16x16" else None, i, i))
                    refgroups.append((i, sx, sy, w, hh, preds[i]))
            # ref_idx_l0 of every partition, then ref_idx_l1 (the 8x8 quadrant is the unit in P_8x8 / B_8x8)
            refs = {}
            for lst in (0, 1):
                for g, gx, gy, gw, gh, pr in refgroups:
                    if pr in (lst, 2):
                        rf = self._ref_idx(lst, X4 + gx, Y4 + gy) if (self.h["nref"][lst] > 1 and not getattr(m, "ref0", False)) else 0
                        assert rf < self.h["nref"][lst], "ref_idx beyond the list: the CABAC decode lost synchronisation"
                        refs[(lst, g)] = rf
                        self.pic.ref[lst, Y4 + gy:Y4 + gy + gh, X4 + gx:X4 + gx + gw] = rf
            # mvd_l0 of every partition, then mvd_l1; the prediction sees the partitions derived before it in THIS list's pass
            for lst in (0, 1):
                mb_done[:] = False
                # (8x8: direct quadrants count as derived -- their motion data exists before any mvd is parsed; availability inside
                #  the macroblock follows the quadrant order, so a direct quadrant q only serves quadrants > q)
                qdone = -1
                for (sx, sy, w, hh, pr, shape, pi, g) in plist:
                    if inter == "8x8":
                        # quadrants before this one are complete (direct ones included)
                        for q in range(g):
                            if q > qdone:
                                qx, qy = (q & 1) * 2, (q >> 1) * 2
                                mb_done[qy:qy + 2, qx:qx + 2] = True
                                qdone = q
                    if pr in (lst, 2):
                        rf = refs[(lst, g)]
                        mvdx = self._mvd(lst, 0, X4 + sx, Y4 + sy)
                        mvdy = self._mvd(lst, 1, X4 + sx, Y4 + sy)
                        px, py = self._mvp(lst, X4 + sx, Y4 + sy, w, hh, rf, shape, pi)
                        self._set_motion(lst, X4 + sx, Y4 + sy, w, hh, rf, (px + mvdx, py + mvdy), (mvdx, mvdy))
                    mb_done[sy:sy + hh, sx:sx + w] = True
                mb_done[:] = True
            parts = [(sx, sy, w, hh) for (sx, sy, w, hh, *_rest) in plist]
            if inter == "8x8":
                for q in range(4):
                    if subs[q][0] == "direct":
                        qx, qy = (q & 1) * 2, (q >> 1) * 2
                        parts += [(qx, qy, 2, 2)] if self.sps["direct_8x8_inference"] else [(qx + i, qy + j, 1, 1) for j in range(2) for i in range(2)]
        pic.mbs[addr] = m
        self._predict_inter(mx, my, parts)
        # ---- coded_block_pattern, mb_qp_delta, residual
        self._cbp(m, A, Bn)
        if self.t8mode and m.cbp_luma and no_sub8 and self._t8_flag(A, Bn):
            m.t8 = True
            self.stats["t8"] += 1
        self._qp_delta(m, coded=bool(m.cbp_luma or m.cbp_chroma))
        self._residual(mx, my, m, A, Bn)

    def _cbp(self, m, A, Bn):
        cab = self.cab
        cbp = 0
        for b8 in range(4):
            x8, y8 = b8 & 1, b8 >> 1

            def cond(dx, dy):
                x, y = x8 + dx, y8 + dy
                if 0 <= x < 2 and 0 <= y < 2:
                    return 0 if (cbp >> (y * 2 + x)) & 1 else 1
                nb = A if dx else Bn
                if nb is None:
                    return 0
                return 0 if (nb.cbp_luma >> ((y % 2) * 2 + (x % 2))) & 1 else 1

            if cab.decision(73 + cond(-1, 0) + 2 * cond(0, -1)):
                cbp |= 1 << b8
        m.cbp_luma = cbp
        ca = 1 if (A is not None and A.cbp_chroma != 0) else 0
        cb = 1 if (Bn is not None and Bn.cbp_chroma != 0) else 0
        if cab.decision(77 + ca + 2 * cb):
            ca = 1 if (A is not None and A.cbp_chroma == 2) else 0
            cb = 1 if (Bn is not None and Bn.cbp_chroma == 2) else 0
            m.cbp_chroma = 1 + cab.decision(77 + 4 + ca + 2 * cb)

    def _qp_delta(self, m, coded):
        cab = self.cab
        if coded:
            k = 0
            if cab.decision(60 + self.prev_qp_delta_nz):
                k = 1
                if cab.decision(60 + 2):
                    k = 2
                    while cab.decision(60 + 3):
                        k += 1
                        assert k < 120, "mb_qp_delta runaway: the CABAC decode lost synchronisation"
            dqp = (k + 1) // 2 if k & 1 else -(k // 2)
            self.qp = (self.qp + dqp + 52) % 52
            m.qp_delta_nz = 1 if dqp else 0
        self.prev_qp_delta_nz = m.qp_delta_nz
        m.qp = self.qp

    def _t8_flag(self, A, Bn):
        """transform_size_8x8_flag (ctxIdx 399 + the neighbours' flags)"""
        return self.cab.decision(399 + (1 if (A is not None and A.t8) else 0) + (1 if (Bn is not None and Bn.t8) else 0))

    def _residual_block8(self):
        """the 64 levels of an 8x8 luma block in scan order (ctxBlockCat 5; its coded_block_flag is not sent in 4:2:0: inferred 1)"""
        cab = self.cab
        coef = [0] * 64
        sig = []
        last = 63
        for i in range(63):
            if cab.decision(402 + SIG8[i]):
                sig.append(i)
                if cab.decision(417 + LAST8[i]):
                    last = None
                    break
        if last is not None:
            sig.append(63)
        eq1, gt1 = 0, 0
        for i in reversed(sig):
            inc_ = 0 if gt1 else min(4, 1 + eq1)
            v = 0
            if cab.decision(426 + inc_):
                inc2 = 5 + min(4, gt1)
                v = 1
                while v < 14 and cab.decision(426 + inc2):
                    v += 1
                if v == 14:
                    k = 0
                    while cab.bypass():
                        v += 1 << k
                        k += 1
                        assert k < 24, "coefficient runaway: the CABAC decode lost synchronisation"
                    while k:
                        k -= 1
                        v += cab.bypass() << k
            if v == 0:
                eq1 += 1
            else:
                gt1 += 1
            coef[i] = -(v + 1) if cab.bypass() else v + 1
        return coef

    def _i4_mode(self):
        """prev_intra4x4_pred_mode_flag / rem_intra4x4_pred_mode -> None (use the predicted mode) or rem"""
        cab = self.cab
        if cab.decision(68):
            return None
        return cab.decision(69) | (cab.decision(69) << 1) | (cab.decision(69) << 2)

    def _chroma_mode(self, A, Bn):
        cab = self.cab
        inc = (1 if (A is not None and A.chroma_mode != 0) else 0) + (1 if (Bn is not None and Bn.chroma_mode != 0) else 0)
        cm = 0
        if cab.decision(64 + inc):
            cm = 1
            if cab.decision(64 + 3):
                cm = 2
                if cab.decision(64 + 3):
                    cm = 3
        return cm

    def _intra_tail(self, addr, mx, my, m, A, Bn):
        """prediction modes, chroma mode, coded_block_pattern, mb_qp_delta and residual of an intra macroblock"""
        pic, cab = self.pic, self.cab
        X4, Y4 = mx * 4, my * 4
        m.intra = True
        pic.intra4[Y4:Y4 + 4, X4:X4 + 4] = True
        if m.typ == "I4" and self.t8mode and self._t8_flag(A, Bn):
            m.typ, m.t8 = "I8", True  # (mb_type I_NxN with transform_size_8x8_flag: Intra 8x8)
        self.stats[m.typ] += 1
        pic.done[Y4:Y4 + 4, X4:X4 + 4] = True  # (refIdx -1 in both lists: "available, not inter")
        if m.typ in ("I4", "I8"):
            # Intra 4x4: sixteen blocks; Intra 8x8: four, whose neighbours are the 4x4 blocks left of / above their first 4x4 block
            # (8.3.2.1: Intra4x4PredMode[8x8 index * 4 + 1] of the left, [... + 2] of the upper neighbour); a mode is kept per 4x4 slot
            for blk in (range(16) if m.typ == "I4" else BLK8_FIRST):
                bx, by = BLK_XY[blk]

                def nmode(dx, dy):
                    x, y = bx + dx, by + dy
                    nb = m if (0 <= x < 4 and 0 <= y < 4) else pic.mb(mx + (x // 4 if x < 0 or x > 3 else 0), my + (y // 4 if y < 0 or y > 3 else 0))
                    if nb is None:
                        return None
                    if nb.typ not in ("I4", "I8"):
                        return 2
                    return nb.modes[XY_BLK[(x % 4, y % 4)]]

                ma, mb_ = nmode(-1, 0), nmode(0, -1)
                pred = 2 if (ma is None or mb_ is None) else min(ma, mb_)
                rem = self._i4_mode()
                mode = pred if rem is None else (rem if rem < pred else rem + 1)
                if m.typ == "I4":
                    m.modes[blk] = mode
                else:
                    for k in range(4):
                        m.modes[blk + k] = mode
        m.chroma_mode = self._chroma_mode(A, Bn)
        if m.typ in ("I4", "I8"):
            self._cbp(m, A, Bn)
        self._qp_delta(m, coded=(m.typ == "I16" or bool(m.cbp_luma or m.cbp_chroma)))
        pic.mbs[addr] = m
        self._residual(mx, my, m, A, Bn)

    def _residual(self, mx, my, m, A, Bn):
        """residual data + reconstruction (intra prediction interleaved block by block, inter prediction already in the planes)"""
        pic, pps = self.pic, self.pps
        qpy = m.qp
        px, py = mx * 16, my * 16
        X4, Y4 = mx * 4, my * 4
        dc16 = None
        if m.typ == "I16":
            lv, m.cbf_dc = self._residual_block(m, A, Bn, 0, 16)
            c = [[0] * 4 for _ in range(4)]
            for k, (x, y) in enumerate(ZIGZAG):
                c[y][x] = lv[k]
            Am = [[1, 1, 1, 1], [1, 1, -1, -1], [1, -1, -1, 1], [1, -1, 1, -1]]
            t = [[sum(Am[i][k] * c[k][j] for k in range(4)) for j in range(4)] for i in range(4)]
            f = [[sum(t[i][k] * Am[k][j] for k in range(4)) for j in range(4)] for i in range(4)]
            ls = level_scale(qpy, 0, 0)
            if qpy >= 36:
                dc16 = [[(f[i][j] * ls) << (qpy // 6 - 6) for j in range(4)] for i in range(4)]
            else:
                dc16 = [[(f[i][j] * ls + (1 << (5 - qpy // 6))) >> (6 - qpy // 6) for j in range(4)] for i in range(4)]
            pred16(pic, m, mx, my)
        if m.t8:
            # 8x8 transform: four luma blocks of 64 levels each (their coded_block_flag is inferred), Intra 8x8 prediction interleaved
            for b8 in range(4):
                bx8, by8 = b8 & 1, b8 >> 1
                if m.typ == "I8":
                    pred8(pic, m, mx, my, b8)
                if (m.cbp_luma >> b8) & 1:
                    lv = self._residual_block8()
                    d = [[0] * 8 for _ in range(8)]
                    for k, (x, y) in enumerate(ZIGZAG8):
                        if lv[k]:
                            ls8 = level_scale8(qpy, y, x)
                            d[y][x] = (lv[k] * ls8) << (qpy // 6 - 6) if qpy >= 36 else (lv[k] * ls8 + (1 << (5 - qpy // 6))) >> (6 - qpy // 6)
                    rr = np.array(idct8(d), np.int32)
                    ys, xs = slice(py + by8 * 8, py + by8 * 8 + 8), slice(px + bx8 * 8, px + bx8 * 8 + 8)
                    pic.Y[ys, xs] = np.clip(pic.Y[ys, xs] + rr, 0, 255)
                    for k in range(4):
                        m.cbf_luma[BLK8_FIRST[b8] + k] = 1
                    pic.nz[Y4 + by8 * 2:Y4 + by8 * 2 + 2, X4 + bx8 * 2:X4 + bx8 * 2 + 2] = True
        for blk in (range(16) if not m.t8 else ()):
            bx, by = BLK_XY[blk]
            d = None
            coded = False
            if (m.cbp_luma >> ((by >> 1) * 2 + (bx >> 1))) & 1:
                if m.typ == "I16":
                    lv, m.cbf_luma[blk] = self._residual_block(m, A, Bn, 1, 15, bx, by)
                    lv = [0] + lv
                else:
                    lv, m.cbf_luma[blk] = self._residual_block(m, A, Bn, 2, 16, bx, by)
                coded = m.cbf_luma[blk] == 1
                if coded:
                    d = [[0] * 4 for _ in range(4)]
                    for k, (x, y) in enumerate(ZIGZAG):
                        if lv[k]:
                            lsx = level_scale(qpy, x, y)
                            d[y][x] = (lv[k] * lsx) << (qpy // 6 - 4) if qpy >= 24 else (lv[k] * lsx + (1 << (3 - qpy // 6))) >> (4 - qpy // 6)
                    pic.nz[Y4 + by, X4 + bx] = True
            if m.typ == "I4":
                pred4(pic, m, mx, my, blk)
            if dc16 is not None and dc16[by][bx] != 0:
                if d is None:
                    d = [[0] * 4 for _ in range(4)]
                d[0][0] = dc16[by][bx]
                coded = True
            if coded:
                rr = np.array(idct4(d), np.int32)
                ys, xs = slice(py + by * 4, py + by * 4 + 4), slice(px + bx * 4, px + bx * 4 + 4)
                pic.Y[ys, xs] = np.clip(pic.Y[ys, xs] + rr, 0, 255)
        if m.intra:
            pred_chroma(pic, m, mx, my)
        qpcs = [QPC[min(max(qpy + pps[key], 0), 51)] for key in ("chroma_qp_offset", "chroma_qp_offset2")]
        dcs = [[0] * 4, [0] * 4]
        if m.cbp_chroma:
            for comp in range(2):
                qpc = qpcs[comp]
                lv, m.cbf_cdc[comp] = self._residual_block(m, A, Bn, 3, 4, comp=comp)
                c = [[lv[0], lv[1]], [lv[2], lv[3]]]
                f = [[c[0][0] + c[0][1] + c[1][0] + c[1][1], c[0][0] - c[0][1] + c[1][0] - c[1][1]],
                     [c[0][0] + c[0][1] - c[1][0] - c[1][1], c[0][0] - c[0][1] - c[1][0] + c[1][1]]]
                ls = level_scale(qpc, 0, 0)
                dcs[comp] = [((f[i][j] * ls) << (qpc // 6)) >> 5 for i in range(2) for j in range(2)]
        acs = [[None] * 4, [None] * 4]
        if m.cbp_chroma == 2:
            for comp in range(2):
                for blk in range(4):
                    lv, m.cbf_cac[comp][blk] = self._residual_block(m, A, Bn, 4, 15, blk & 1, blk >> 1, comp)
                    acs[comp][blk] = [0] + lv
        for comp in range(2):
            qpc = qpcs[comp]
            for blk in range(4):
                bx, by = blk & 1, blk >> 1
                d = [[0] * 4 for _ in range(4)]
                any_ = False
                if acs[comp][blk] is not None:
                    for k, (x, y) in enumerate(ZIGZAG):
                        v = acs[comp][blk][k]
                        if v:
                            lsx = level_scale(qpc, x, y)
                            d[y][x] = (v * lsx) << (qpc // 6 - 4) if qpc >= 24 else (v * lsx + (1 << (3 - qpc // 6))) >> (4 - qpc // 6)
                            any_ = True
                d[0][0] = dcs[comp][blk]
                if any_ or d[0][0]:
                    rr = np.array(idct4(d), np.int32)
                    P = pic.C[comp]
                    ys, xs = slice(my * 8 + by * 4, my * 8 + by * 4 + 4), slice(mx * 8 + bx * 4, mx * 8 + bx * 4 + 4)
                    P[ys, xs] = np.clip(P[ys, xs] + rr, 0, 255)


# ----------------------------------------------------------------------------------------------------------------- CAVLC (9.2)
# coeff_token: [table by nC][4 * TotalCoeff + TrailingOnes] -> (length, code); Tables 9-5 (a prefix-free code each: checked at import)
_CT_LEN = [
    [1, 0, 0, 0, 6, 2, 0, 0, 8, 6, 3, 0, 9, 8, 7, 5, 10, 9, 8, 6, 11, 10, 9, 7, 13, 11, 10, 8, 13, 13, 11, 9, 13, 13, 13, 10, 14, 14, 13, 11,
     14, 14, 14, 13, 15, 15, 14, 14, 15, 15, 15, 14, 16, 15, 15, 15, 16, 16, 16, 15, 16, 16, 16, 16, 16, 16, 16, 16],
    [2, 0, 0, 0, 6, 2, 0, 0, 6, 5, 3, 0, 7, 6, 6, 4, 8, 6, 6, 4, 8, 7, 7, 5, 9, 8, 8, 6, 11, 9, 9, 6, 11, 11, 11, 7, 12, 11, 11, 9,
     12, 12, 12, 11, 12, 12, 12, 11, 13, 13, 13, 12, 13, 13, 13, 13, 13, 14, 13, 13, 14, 14, 14, 13, 14, 14, 14, 14],
    [4, 0, 0, 0, 6, 4, 0, 0, 6, 5, 4, 0, 6, 5, 5, 4, 7, 5, 5, 4, 7, 5, 5, 4, 7, 6, 6, 4, 7, 6, 6, 4, 8, 7, 7, 5, 8, 8, 7, 6,
     9, 8, 8, 7, 9, 9, 8, 8, 9, 9, 9, 8, 10, 9, 9, 9, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10],
    [6, 0, 0, 0, 6, 6, 0, 0, 6, 6, 6, 0] + [6] * 56]
_CT_BITS = [
    [1, 0, 0, 0, 5, 1, 0, 0, 7, 4, 1, 0, 7, 6, 5, 3, 7, 6, 5, 3, 7, 6, 5, 4, 15, 6, 5, 4, 11, 14, 5, 4, 8, 10, 13, 4, 15, 14, 9, 4,
     11, 10, 13, 12, 15, 14, 9, 12, 11, 10, 13, 8, 15, 1, 9, 12, 11, 14, 13, 8, 7, 10, 9, 12, 4, 6, 5, 8],
    [3, 0, 0, 0, 11, 2, 0, 0, 7, 7, 3, 0, 7, 10, 9, 5, 7, 6, 5, 4, 4, 6, 5, 6, 7, 6, 5, 8, 15, 6, 5, 4, 11, 14, 13, 4, 15, 10, 9, 4,
     11, 14, 13, 12, 8, 10, 9, 8, 15, 14, 13, 12, 11, 10, 9, 12, 7, 11, 6, 8, 9, 8, 10, 1, 7, 6, 5, 4],
    [15, 0, 0, 0, 15, 14, 0, 0, 11, 15, 13, 0, 8, 12, 14, 12, 15, 10, 11, 11, 11, 8, 9, 10, 9, 14, 13, 9, 8, 10, 9, 8, 15, 14, 13, 13,
     11, 14, 10, 12, 15, 10, 13, 12, 11, 14, 9, 12, 8, 10, 13, 8, 13, 7, 9, 12, 9, 12, 11, 10, 5, 8, 7, 6, 1, 4, 3, 2],
    [3, 0, 0, 0, 0, 1, 0, 0, 4, 5, 6, 0] + list(range(8, 64))]
_CDC_LEN = [2, 0, 0, 0, 6, 1, 0, 0, 6, 6, 3, 0, 6, 7, 7, 6, 6, 8, 8, 7]
_CDC_BITS = [1, 0, 0, 0, 7, 1, 0, 0, 4, 6, 1, 0, 3, 3, 2, 5, 2, 3, 2, 0]
# total_zeros: [TotalCoeff - 1][total_zeros] (Tables 9-7, 9-8), chroma DC (Table 9-9a); run_before: [min(zerosLeft, 7) - 1][run] (9-10)
_TZ_LEN = [[1, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 9], [3, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 6, 6, 6, 6], [4, 3, 3, 3, 4, 4, 3, 3, 4, 5, 5, 6, 5, 6],
           [5, 3, 4, 4, 3, 3, 3, 4, 3, 4, 5, 5, 5], [4, 4, 4, 3, 3, 3, 3, 3, 4, 5, 4, 5], [6, 5, 3, 3, 3, 3, 3, 3, 4, 3, 6],
           [6, 5, 3, 3, 3, 2, 3, 4, 3, 6], [6, 4, 5, 3, 2, 2, 3, 3, 6], [6, 6, 4, 2, 2, 3, 2, 5], [5, 5, 3, 2, 2, 2, 4], [4, 4, 3, 3, 1, 3],
           [4, 4, 2, 1, 3], [3, 3, 1, 2], [2, 2, 1], [1, 1]]
_TZ_BITS = [[1, 3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 1], [7, 6, 5, 4, 3, 5, 4, 3, 2, 3, 2, 3, 2, 1, 0], [5, 7, 6, 5, 4, 3, 4, 3, 2, 3, 2, 1, 1, 0],
            [3, 7, 5, 4, 6, 5, 4, 3, 3, 2, 2, 1, 0], [5, 4, 3, 7, 6, 5, 4, 3, 2, 1, 1, 0], [1, 1, 7, 6, 5, 4, 3, 2, 1, 1, 0],
            [1, 1, 5, 4, 3, 3, 2, 1, 1, 0], [1, 1, 1, 3, 3, 2, 2, 1, 0], [1, 0, 1, 3, 2, 1, 1, 1], [1, 0, 1, 3, 2, 1, 1], [0, 1, 1, 2, 1, 3],
            [0, 1, 1, 1, 1], [0, 1, 1, 1], [0, 1, 1], [0, 1]]
_CTZ_LEN, _CTZ_BITS = [[1, 2, 3, 3], [1, 2, 2], [1, 1]], [[1, 1, 1, 0], [1, 1, 0], [1, 0]]
_RUN_LEN = [[1, 1], [1, 2, 2], [2, 2, 2, 2], [2, 2, 2, 3, 3], [2, 2, 3, 3, 3, 3], [2, 3, 3, 3, 3, 3, 3], [3, 3, 3, 3, 3, 3, 3, 4, 5, 6, 7, 8, 9, 10, 11]]
_RUN_BITS = [[1, 0], [1, 1, 0], [3, 2, 1, 0], [3, 2, 1, 1, 0], [3, 2, 3, 2, 1, 0], [3, 0, 1, 3, 2, 5, 4], [7, 6, 5, 4, 3, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1]]
# coded_block_pattern me(v), Table 9-4 (chroma formats 1, 2): codeNum -> cbp for Intra_4x4 and for inter macroblocks
_CBP_INTRA = [47, 31, 15, 0, 23, 27, 29, 30, 7, 11, 13, 14, 39, 43, 45, 46, 16, 3, 5, 10, 12, 19, 21, 26, 28, 35, 37, 42, 44, 1, 2, 4, 8, 17, 18, 20, 24,
              6, 9, 22, 25, 32, 33, 34, 36, 40, 38, 41]
_CBP_INTER = [0, 16, 1, 2, 4, 8, 32, 3, 5, 10, 12, 15, 47, 7, 11, 13, 14, 6, 9, 31, 35, 37, 42, 44, 33, 34, 36, 40, 39, 43, 45, 46, 17, 18, 20, 24, 19,
              21, 26, 28, 23, 27, 29, 30, 22, 25, 38, 41]
assert sorted(_CBP_INTRA) == list(range(48)) and sorted(_CBP_INTER) == list(range(48))


def _vlc(lens, bits):
    """{(length, code): symbol} of one table, checked to be a prefix-free code (a mistyped entry cannot hide)"""
    t = {(ln, b): k for k, (ln, b) in enumerate(zip(lens, bits)) if ln}
    strs = [format(b, "0%db" % ln) for ln, b in t]
    assert len(set(strs)) == len(strs) and not any(x != y and y.startswith(x) for x in strs for y in strs), "VLC table is not prefix-free"
    assert sum(2.0 ** -ln for ln, _ in t) <= 1.0 + 1e-12
    return t, max(ln for ln, _ in t)


_VLC_CT = [_vlc(a, b) for a, b in zip(_CT_LEN, _CT_BITS)]
_VLC_CDC = _vlc(_CDC_LEN, _CDC_BITS)
_VLC_TZ = [_vlc(a, b) for a, b in zip(_TZ_LEN, _TZ_BITS)]
_VLC_CTZ = [_vlc(a, b) for a, b in zip(_CTZ_LEN, _CTZ_BITS)]
_VLC_RUN = [_vlc(a, b) for a, b in zip(_RUN_LEN, _RUN_BITS)]


class _CavlcSliceDecoder(_SliceDecoder):
    """The same macroblock semantics with the syntax of 7.3.4 / 7.3.5 read as Exp-Golomb and CAVLC codes (Baseline streams)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        if self.stype == 1:
            raise Unsupported("B slices with CAVLC entropy coding")
        if self.t8mode:
            raise Unsupported("the 8x8 transform with CAVLC entropy coding")
        pic = self.pic
        pic.tc = np.zeros((pic.Hh * 4, pic.W * 4), np.int32)                              # TotalCoeff of luma 4x4 blocks
        pic.tcc = [np.zeros((pic.Hh * 2, pic.W * 2), np.int32) for _ in range(2)]         # ... of chroma AC blocks
        # position of the rbsp stop bit: more_rbsp_data() is "the read position is in front of it"
        pl = self._payload
        n = len(pl)
        while n and pl[n - 1] == 0:
            n -= 1
        assert n, "empty slice payload"
        last = pl[n - 1]
        self._stop = (n - 1) * 8 + 7 - ((last & -last).bit_length() - 1)
        self._mb_xy = (0, 0)

    def _more(self):
        return self.r.p < self._stop

    def _read_vlc(self, table):
        t, maxlen = table
        r = self.r
        code = 0
        for ln in range(1, maxlen + 1):
            code = (code << 1) | r.u(1)
            k = t.get((ln, code))
            if k is not None:
                return k
        raise AssertionError("no CAVLC codeword matches: the decode lost synchronisation")

    # ---- slice data (7.3.4)
    def run(self):
        pic, r, h = self.pic, self.r, self.h
        n_mb = pic.W * pic.Hh
        addr = 0
        more = True
        while more:
            if self.stype != 2:
                run = r.ue()
                assert addr + run <= n_mb, "mb_skip_run beyond the picture: the CAVLC decode lost synchronisation"
                for _ in range(run):
                    self._skip_mb(addr, addr % pic.W, addr // pic.W, MBInfo())
                    addr += 1
                if run:
                    more = self._more()
            if more:
                assert addr < n_mb, "macroblock data beyond the picture: the CAVLC decode lost synchronisation"
                self._macroblock(addr, addr % pic.W, addr // pic.W)
                addr += 1
                more = self._more()
        assert addr == n_mb, f"slice data ended at macroblock {addr} of {n_mb}: the CAVLC decode lost synchronisation"
        self.stats["bits_left"] = self._stop - r.p
        assert self.stats["bits_left"] == 0, self.stats
        if h["dbf"] != 1:
            deblock_inter(pic, h["off"][0], h["off"][1])
        pic.stats = self.stats

    def _macroblock(self, addr, mx, my):
        pic, r = self.pic, self.r
        self._mb_xy = (mx, my)
        m = MBInfo()
        A, Bn = pic.mb(mx - 1, my), pic.mb(mx, my - 1)
        t = r.ue()
        inter = None
        if self.stype == 0:
            if t < 5:
                inter = [("16x16", [0, None]), ("16x8", [0, 0]), ("8x16", [0, 0]), "8x8", "8x8"][t]
                m.ref0 = t == 4
            else:
                t -= 5
        if inter is None:
            if t == 0:
                m.typ = "I4"
            elif t == 25:
                raise Unsupported("I_PCM macroblocks")
            else:
                assert t < 25, "mb_type out of range: the CAVLC decode lost synchronisation"
                m.typ = "I16"
                m.i16, m.cbp_chroma, m.cbp_luma = (t - 1) % 4, ((t - 1) // 4) % 3, (15 if t >= 13 else 0)
            self._intra_tail(addr, mx, my, m, A, Bn)
        else:
            self._inter_tail(addr, mx, my, m, A, Bn, inter)

    # ---- syntax elements
    def _sub_mb_types(self):
        out = []
        for _ in range(4):
            st = self.r.ue()
            assert st < 4, "sub_mb_type out of range: the CAVLC decode lost synchronisation"
            out.append((P_SUB[st], 0))
        return out

    def _ref_idx(self, lst, x4, y4):
        n = self.h["nref"][lst]
        return (1 - self.r.u(1)) if n == 2 else self.r.ue()

    def _mvd(self, lst, comp, x4, y4):
        return self.r.se()

    def _i4_mode(self):
        return None if self.r.u(1) else self.r.u(3)

    def _chroma_mode(self, A, Bn):
        v = self.r.ue()
        assert v < 4, "intra_chroma_pred_mode out of range: the CAVLC decode lost synchronisation"
        return v

    def _cbp(self, m, A, Bn):
        v = self.r.ue()
        assert v < 48, "coded_block_pattern out of range: the CAVLC decode lost synchronisation"
        cbp = (_CBP_INTRA if m.intra else _CBP_INTER)[v]
        m.cbp_luma, m.cbp_chroma = cbp & 15, cbp >> 4

    def _qp_delta(self, m, coded):
        if coded:
            dqp = self.r.se()
            assert -26 <= dqp <= 25, "mb_qp_delta out of range: the CAVLC decode lost synchronisation"
            self.qp = (self.qp + dqp + 52) % 52
        m.qp = self.qp

    # ---- residual_block_cavlc (7.3.5.3.2, 9.2)
    def _residual_block(self, m, A, Bn, cat, n_coef, bx=0, by=0, comp=0):
        pic, r = self.pic, self.r
        mx, my = self._mb_xy
        if cat == 3:
            table = _VLC_CDC
        else:
            if cat == 4:
                arr, gx, gy = pic.tcc[comp], mx * 2 + bx, my * 2 + by
            else:
                arr, gx, gy = pic.tc, mx * 4 + (0 if cat == 0 else bx), my * 4 + (0 if cat == 0 else by)
            na = int(arr[gy, gx - 1]) if gx > 0 else None
            nb = int(arr[gy - 1, gx]) if gy > 0 else None
            nc = (na + nb + 1) >> 1 if (na is not None and nb is not None) else (na if na is not None else (nb if nb is not None else 0))
            table = _VLC_CT[0 if nc < 2 else (1 if nc < 4 else (2 if nc < 8 else 3))]
        k = self._read_vlc(table)
        total, t1 = k >> 2, k & 3
        if cat in (1, 2):
            pic.tc[my * 4 + by, mx * 4 + bx] = total
        elif cat == 4:
            pic.tcc[comp][my * 2 + by, mx * 2 + bx] = total
        coef = [0] * n_coef
        if total == 0:
            return coef, 0
        assert total <= n_coef and t1 <= min(total, 3), "coeff_token out of range: the CAVLC decode lost synchronisation"
        levels = []
        suffix_len = 1 if (total > 10 and t1 < 3) else 0
        for i in range(total):
            if i < t1:
                levels.append(1 - 2 * r.u(1))
                continue
            prefix = 0
            while r.u(1) == 0:
                prefix += 1
                assert prefix < 32, "level_prefix runaway: the CAVLC decode lost synchronisation"
            code = min(15, prefix) << suffix_len
            if suffix_len > 0 or prefix >= 14:
                size = 4 if (prefix == 14 and suffix_len == 0) else (prefix - 3 if prefix >= 15 else suffix_len)
                if size:
                    code += r.u(size)
            if prefix >= 15 and suffix_len == 0:
                code += 15
            if prefix >= 16:
                code += (1 << (prefix - 3)) - 4096
            if i == t1 and t1 < 3:
                code += 2
            lv = (code + 2) >> 1 if code % 2 == 0 else (-code - 1) >> 1
            levels.append(lv)
            if suffix_len == 0:
                suffix_len = 1
            if abs(lv) > (3 << (suffix_len - 1)) and suffix_len < 6:
                suffix_len += 1
        zeros_left = 0
        if total < n_coef:
            zeros_left = self._read_vlc(_VLC_CTZ[total - 1] if cat == 3 else _VLC_TZ[total - 1])
            assert total + zeros_left <= n_coef, "total_zeros out of range: the CAVLC decode lost synchronisation"
        pos = total + zeros_left - 1  # scan position of the highest-frequency coefficient (levels[0])
        for i in range(total):
            coef[pos] = levels[i]
            if i < total - 1:
                run = self._read_vlc(_VLC_RUN[min(zeros_left, 7) - 1]) if zeros_left > 0 else 0
                assert run <= zeros_left, "run_before out of range: the CAVLC decode lost synchronisation"
                zeros_left -= run
                pos -= 1 + run
        return coef, 1


# ----------------------------------------------------------------------------------------------------------------- edge filter (8.7)


def _bs(pic, py4, px4, qy4, qx4, mb_edge):
    """boundary strength between the 4x4 blocks p and q (8.7.2.1, frame pictures)"""
    if pic.intra4[py4, px4] or pic.intra4[qy4, qx4]:
        return 4 if mb_edge else 3
    if pic.nz[py4, px4] or pic.nz[qy4, qx4]:
        return 2
    pr = sorted(int(pic.refid[lst, py4, px4]) for lst in (0, 1) if pic.ref[lst, py4, px4] >= 0)
    qr = sorted(int(pic.refid[lst, qy4, qx4]) for lst in (0, 1) if pic.ref[lst, qy4, qx4] >= 0)
    if pr != qr:
        return 1
    pm = [(int(pic.refid[lst, py4, px4]), int(pic.mv[lst, py4, px4, 0]), int(pic.mv[lst, py4, px4, 1])) for lst in (0, 1) if pic.ref[lst, py4, px4] >= 0]
    qm = [(int(pic.refid[lst, qy4, qx4]), int(pic.mv[lst, qy4, qx4, 0]), int(pic.mv[lst, qy4, qx4, 1])) for lst in (0, 1) if pic.ref[lst, qy4, qx4] >= 0]

    def far(a, b):
        return abs(a[1] - b[1]) >= 4 or abs(a[2] - b[2]) >= 4

    if len(pm) == 1:
        return 1 if far(pm[0], qm[0]) else 0
    if pm[0][0] != pm[1][0]:  # two different reference pictures: compare the vectors that point into the same picture
        q0 = qm[0] if qm[0][0] == pm[0][0] else qm[1]
        q1 = qm[1] if qm[0][0] == pm[0][0] else qm[0]
        return 1 if (far(pm[0], q0) or far(pm[1], q1)) else 0
    # both vectors of both blocks point into the same picture: either pairing may match
    return 1 if ((far(pm[0], qm[0]) or far(pm[1], qm[1])) and (far(pm[0], qm[1]) or far(pm[1], qm[0]))) else 0


def deblock_inter(pic, off_a, off_b):
    cqos = (pic.pps["chroma_qp_offset"], pic.pps.get("chroma_qp_offset2", pic.pps["chroma_qp_offset"]))

    def qpc(q, comp):
        return QPC[min(max(q + cqos[comp], 0), 51)]

    all_intra = bool(pic.intra4.all())
    for my in range(pic.Hh):
        for mx in range(pic.W):
            m = pic.mb(mx, my)
            for vertical in (True, False):
                nb = pic.mb(mx - 1, my) if vertical else pic.mb(mx, my - 1)
                for e in range(4):
                    if e == 0 and nb is None:
                        continue
                    luma_edge = not (m.t8 and e % 2)  # (8x8 transform: the luma edges inside the 8x8 blocks are not filtered)
                    if not luma_edge and e % 2:
                        continue                      # (odd edges never carry chroma)
                    # boundary strengths of the edge's four 4-sample segments
                    bss = []
                    for k in range(4):
                        if vertical:
                            qy4, qx4 = my * 4 + k, mx * 4 + e
                            py4, px4 = qy4, qx4 - 1
                        else:
                            qy4, qx4 = my * 4 + e, mx * 4 + k
                            py4, px4 = qy4 - 1, qx4
                        bss.append((4 if e == 0 else 3) if all_intra else _bs(pic, py4, px4, qy4, qx4, e == 0))
                    if not any(bss):
                        continue
                    qp_p = nb.qp if e == 0 else m.qp
                    planes = [(pic.Y, 16, True, (qp_p + m.qp + 1) >> 1)] if luma_edge else []
                    if e % 2 == 0:
                        planes += [(P, 8, False, (qpc(qp_p, c) + qpc(m.qp, c) + 1) >> 1) for c, P in enumerate(pic.C)]
                    for P, size, luma, qpav in planes:
                        idx_a, idx_b = min(max(qpav + off_a, 0), 51), min(max(qpav + off_b, 0), 51)
                        alpha, beta = ALPHA[idx_a], BETA[idx_b]
                        if alpha == 0:
                            continue
                        pos = e * 4 if luma else e * 2
                        for k in range(size):
                            bs = bss[k >> 2] if luma else bss[k >> 1]
                            if bs == 0:
                                continue
                            if vertical:
                                y, x = my * size + k, mx * size + pos
                                px = [int(P[y, x + d]) if 0 <= x + d < P.shape[1] else 0 for d in (-4, -3, -2, -1, 0, 1, 2, 3)]
                            else:
                                y, x = my * size + pos, mx * size + k
                                px = [int(P[y + d, x]) if 0 <= y + d < P.shape[0] else 0 for d in (-4, -3, -2, -1, 0, 1, 2, 3)]
                            o = _filter_line(px, bs, alpha, beta, idx_a, luma)
                            if o is px:
                                continue
                            for d, v in zip((-3, -2, -1, 0, 1, 2), o[1:7]):
                                if vertical:
                                    P[y, x + d] = v
                                else:
                                    P[y + d, x] = v


# ----------------------------------------------------------------------------------------------------------------- frames of a file


class H264Reader:
    """Frames of an MP4's H.264 track in DISPLAY order. Sequential access decodes every picture once; a jump restarts at the key
    frame in front of the target (in decoding order) and decodes up to it."""

    def __init__(self, path, cache=8, engine="native"):
        self.engine = engine
        self.track = path if isinstance(path, Mp4H264) else Mp4H264(path)
        self._dec = None
        self._next = 0      # next sample (decoding order) the decoder expects
        self._cache = {}    # sample -> (Y, Cb, Cr)
        self._cache_n = cache
        self._order = []

    def __len__(self):
        return len(self.track)

    def _planes(self, pic):
        cl, cr, ct, cb = self.track.sps["crop"]
        Y = pic.Y[2 * ct:pic.Hh * 16 - 2 * cb, 2 * cl:pic.W * 16 - 2 * cr].astype(np.uint8)
        Cb, Cr = (p[ct:pic.Hh * 8 - cb, cl:pic.W * 8 - cr].astype(np.uint8) for p in pic.C)
        return Y, Cb, Cr

    def frame(self, k):
        """-> (Y, Cb, Cr) uint8 planes of the k-th frame in display order"""
        tr = self.track
        if not 0 <= k < len(tr):
            raise IndexError(k)
        s = tr.display_order[k]
        if s in self._cache:
            return self._cache[s]
        if self._dec is None or s < self._next:
            start = max(i for i in tr.sync if i <= s)
            self._dec, self._next = H264Decoder(tr.sps, tr.pps, self.engine), start
        elif any(self._next <= i <= s for i in tr.sync):
            start = max(i for i in tr.sync if i <= s)  # a key frame lies between: skip ahead to it
            self._dec, self._next = H264Decoder(tr.sps, tr.pps, self.engine), start
        while self._next <= s:
            i = self._next
            pic = self._dec.decode_sample(tr.nal_units(i), i)
            self._cache[i] = self._planes(pic)
            self._order.append(i)
            self._next += 1
            # keep what is ahead of the display cursor (B pictures are shown before the P picture decoded in front of them)
            while len(self._order) > max(self._cache_n, 1) + 8:
                old = self._order.pop(0)
                if old != s:
                    self._cache.pop(old, None)
        return self._cache[s]


class GopPool:
    """Whole groups of pictures decoded ahead on worker threads: the native engine releases the GIL for the macroblock layer, every
    closed GOP starts at a key frame and is independent of the others, so a sequential reader gets `workers` decoders' worth of
    throughput (`MediaVideo` with `workers` > 1). Falls back to None (-> the sequential `H264Reader`) for files whose GOPs are not
    closed in display order."""

    def __init__(self, track, workers=4, engine="native", keep=None, budget_bytes=6 << 30, convert=None, convert_pic=None):
        from concurrent.futures import ThreadPoolExecutor

        self.track, self.engine = track, engine
        n = len(track)
        self.starts = sorted(track.sync)
        self.ends = self.starts[1:] + [n]
        # decoded GOPs are held whole: no more of them in flight than `budget_bytes` of planes (long GOPs of large pictures)
        gop_bytes = max((b - a for a, b in zip(self.starts, self.ends)), default=1) * track.sps["mb_w"] * track.sps["mb_h"] * 384
        self.workers = max(1, min(int(workers), budget_bytes // max(gop_bytes * 2, 1)))
        rank = {s: k for k, s in enumerate(track.display_order)}
        self.closed = bool(self.starts) and self.starts[0] == 0 and all(
            sorted(rank[i] for i in range(a, b)) == list(range(a, b)) for a, b in zip(self.starts, self.ends))
        self._pool = ThreadPoolExecutor(max_workers=self.workers) if self.closed else None
        self._futures = {}          # GOP index -> future of {display index: (Y, Cb, Cr)}
        if engine == "native":
            from .. import _lib

            _lib.lib()  # (loaded once here, not by eight worker threads at the same time)
        self._keep = keep or self.workers + 1
        self._cl = track.sps["crop"]
        self.convert = convert  # (Y, Cb, Cr) -> what `frame` returns (run on the worker threads); None: the planes
        self.convert_pic = convert_pic  # or: picture -> what `frame` returns (native conversion straight from the picture's buffers)

    def _decode(self, g):
        tr = self.track
        dec = H264Decoder(tr.sps, tr.pps, self.engine)
        rank = {s: k for k, s in enumerate(tr.display_order)}
        cl, cr, ct, cb = self._cl
        out = {}
        for i in range(self.starts[g], self.ends[g]):
            pic = dec.decode_sample(tr.nal_units(i), i)
            if self.convert_pic is not None:
                out[rank[i]] = self.convert_pic(pic)
                continue
            Y = pic.Y[2 * ct:pic.Hh * 16 - 2 * cb, 2 * cl:pic.W * 16 - 2 * cr].astype(np.uint8)
            Cb, Cr = (p[ct:pic.Hh * 8 - cb, cl:pic.W * 8 - cr].astype(np.uint8) for p in pic.C)
            out[rank[i]] = (Y, Cb, Cr) if self.convert is None else self.convert(Y, Cb, Cr)
        return out

    def gop_of(self, k):
        import bisect

        return bisect.bisect_right(self.starts, k) - 1

    def frame(self, k):
        if not 0 <= k < len(self.track):
            raise IndexError(k)
        g = self.gop_of(k)
        for j in range(g, min(g + self.workers, len(self.starts))):  # this GOP and the ones a sequential reader needs next
            if j not in self._futures:
                self._futures[j] = self._pool.submit(self._decode, j)
        for j in [j for j in self._futures if j < g - 1 or j >= g + self._keep]:
            self._futures.pop(j)  # (a running decode finishes and is dropped)
        return self._futures[g].result()[k]


if __name__ == "__main__":
    import sys
    import time

    rd = H264Reader(sys.argv[1])
    n = int(sys.argv[2]) if len(sys.argv) > 2 else len(rd)
    dec = H264Decoder(rd.track.sps, rd.track.pps, sys.argv[3] if len(sys.argv) > 3 else "native")
    t0 = time.time()
    for i in range(n):
        p = dec.decode_sample(rd.track.nal_units(i), i)
        print(i, p.stats, "poc", p.poc, "mean", round(float(p.Y.mean()), 2), f"{time.time() - t0:.1f}s", flush=True)
