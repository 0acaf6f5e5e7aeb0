"""A small H.264 decoder for INTRA-coded pictures: the IDR frames (key frames) of an MP4 (ISO/IEC 14496-10 Baseline / Main
profile, CABAC, 4:2:0, frame macroblocks, one slice per picture). Pure Python + NumPy, ~2 s per 384 x 384 frame.

Why it exists: neither the build container nor the GPU box has ANY H.264 decoder (profiles/r06_decoder_probe.txt: no cv2, av,
ffmpeg, GStreamer, VA-API, rocDecode), the reference's default input format is H.264 in MP4 (`MediaVideo`,
sleap/io/video.py:340-504), and the only TensorFlow-produced end-to-end golden the reference holds --
tests/data/models/minimal_instance.UNet.bottomup/labels_pr.val.slp -- was predicted on frame 0 of
tests/data/json_format_v1/centered_pair_low_quality.mp4. This module decodes the key frames of such a file (frame 0 and every
sync sample), which is what the golden test needs (tests/test_frame0_golden.py: the fp32 oracle reproduces the TensorFlow
result on the decoded frame to 1e-4 px); `decode_intra` refuses everything else. P and B pictures, CAVLC and the native
engine that `sleap_amd.io.video.MediaVideo` runs came later in the round and live in io/_h264.py / csrc/h264dec.hip, which
reuse this module's bit reader, MP4 tables, CABAC engine, prediction and transform routines and must reproduce its planes on
key frames bit for bit.

What it implements of the standard (clause numbers of ITU-T H.264): the MP4 sample tables, NAL unit extraction, SPS / PPS /
slice header parsing (7.3), the CABAC engine and the syntax elements of I slices (9.3: mb_type, prev/rem_intra4x4_pred_mode,
intra_chroma_pred_mode, coded_block_pattern, mb_qp_delta, coded_block_flag, significance maps, levels, end_of_slice_flag), Intra
4x4 / 16x16 / chroma prediction (8.3), the 4x4 inverse transforms and scaling with flat matrices (8.5), and the deblocking
filter for intra pictures (8.7). NOT implemented: P / B slices, CAVLC, I_PCM, the 8x8 transform and scaling matrices (High
profile), fields / MBAFF, several slices per picture, FMO -- each refused with a message.

A decode is self-checking: CABAC desynchronises on any wrong table entry or context rule, and then end_of_slice_flag does not
arrive exactly at the last macroblock with the slice data exhausted -- `decode_intra` asserts both. H.264 decoding is specified
bit-exactly: the luma / chroma planes are what any conforming decoder produces.
"""
import struct

import numpy as np

# ----------------------------------------------------------------------------------------------------------------- bit reader


class Bits:
    def __init__(self, data: bytes):
        self.d = data
        self.p = 0  # bit position

    def u(self, n):
        v = 0
        for _ in range(n):
            v = (v << 1) | ((self.d[self.p >> 3] >> (7 - (self.p & 7))) & 1)
            self.p += 1
        return v

    def ue(self):
        z = 0
        while self.u(1) == 0:
            z += 1
        return (1 << z) - 1 + (self.u(z) if z else 0)

    def se(self):
        k = self.ue()
        return (k + 1) // 2 if k & 1 else -(k // 2)


def rbsp(nal: bytes) -> bytes:
    """NAL payload without emulation-prevention bytes (00 00 03 -> 00 00), header byte removed."""
    out, z = bytearray(), 0
    for x in nal[1:]:
        if z >= 2 and x == 3:
            z = 0
            continue
        out.append(x)
        z = z + 1 if x == 0 else 0
    return bytes(out)


# ----------------------------------------------------------------------------------------------------------------- MP4


def _boxes(b, off, end):
    while off + 8 <= end:
        sz, typ = struct.unpack(">I4s", b[off:off + 8])
        hdr = 8
        if sz == 1:
            sz = struct.unpack(">Q", b[off + 8:off + 16])[0]
            hdr = 16
        if sz == 0:
            sz = end - off
        yield typ, off + hdr, off + sz
        off += sz


def _find(b, path, off=0, end=None):
    end = len(b) if end is None else end
    for typ, s, e in _boxes(b, off, end):
        if typ == path[0]:
            return (s, e) if len(path) == 1 else _find(b, path[1:], s, e)
    raise KeyError(path)


class Mp4H264:
    """The first video track of an MP4 with an `avc1` sample entry: parameter sets, the byte range of every sample, sync samples."""

    def __init__(self, path):
        self.path = path
        b = open(path, "rb").read()
        self._b = b
        s, e = _find(b, [b"moov", b"trak", b"mdia", b"minf", b"stbl"])
        sd = _find(b, [b"stsd"], s, e)
        i = b.find(b"avcC", sd[0], sd[1])
        if i < 0:
            raise ValueError(f"{path}: no H.264 (avcC) sample description in the first track")
        a = b[i + 4:sd[1]]
        self.nal_length_size = (a[4] & 3) + 1
        p = 6
        sps = []
        for _ in range(a[5] & 31):
            ln = struct.unpack(">H", a[p:p + 2])[0]
            sps.append(a[p + 2:p + 2 + ln])
            p += 2 + ln
        n_pps = a[p]
        p += 1
        pps = []
        for _ in range(n_pps):
            ln = struct.unpack(">H", a[p:p + 2])[0]
            pps.append(a[p + 2:p + 2 + ln])
            p += 2 + ln
        self.sps, self.pps = parse_sps(sps[0]), parse_pps(pps[0])
        sz = _find(b, [b"stsz"], s, e)
        sample_size, count = struct.unpack(">II", b[sz[0] + 4:sz[0] + 12])
        sizes = [sample_size] * count if sample_size else list(struct.unpack(f">{count}I", b[sz[0] + 12:sz[0] + 12 + 4 * count]))
        try:
            co = _find(b, [b"stco"], s, e)
            n = struct.unpack(">I", b[co[0] + 4:co[0] + 8])[0]
            chunk_off = struct.unpack(f">{n}I", b[co[0] + 8:co[0] + 8 + 4 * n])
        except KeyError:
            co = _find(b, [b"co64"], s, e)
            n = struct.unpack(">I", b[co[0] + 4:co[0] + 8])[0]
            chunk_off = struct.unpack(f">{n}Q", b[co[0] + 8:co[0] + 8 + 8 * n])
        sc = _find(b, [b"stsc"], s, e)
        n = struct.unpack(">I", b[sc[0] + 4:sc[0] + 8])[0]
        stsc = [struct.unpack(">III", b[sc[0] + 8 + 12 * k:sc[0] + 20 + 12 * k]) for k in range(n)]
        self.offsets = []
        k = 0
        for ci, off in enumerate(chunk_off, start=1):
            while k + 1 < len(stsc) and stsc[k + 1][0] <= ci:
                k += 1
            for _ in range(stsc[k][1]):
                if len(self.offsets) == count:
                    break
                self.offsets.append(off)
                off += sizes[len(self.offsets) - 1]
        assert len(self.offsets) == count, (len(self.offsets), count)
        self.sizes = sizes
        try:
            ss = _find(b, [b"stss"], s, e)
            n = struct.unpack(">I", b[ss[0] + 4:ss[0] + 8])[0]
            self.sync = [v - 1 for v in struct.unpack(f">{n}I", b[ss[0] + 8:ss[0] + 8 + 4 * n])]
        except KeyError:
            self.sync = list(range(count))  # no sync table: every sample is a sync sample
        # composition time offsets (ctts): sample i (decoding order) is shown at dts(i) + cts(i); `display_order[k]` = the sample
        # shown k-th. Without a ctts box (no B pictures) the two orders are the same.
        self.cts = [0] * count
        try:
            ct_ = _find(b, [b"ctts"], s, e)
            ver = b[ct_[0]]
            n = struct.unpack(">I", b[ct_[0] + 4:ct_[0] + 8])[0]
            k = 0
            for j in range(n):
                cnt, off = struct.unpack(">Ii" if ver else ">II", b[ct_[0] + 8 + 8 * j:ct_[0] + 16 + 8 * j])
                for _ in range(cnt):
                    if k < count:
                        self.cts[k] = off
                        k += 1
        except KeyError:
            pass
        dts, tt = [], 0
        try:
            ts_ = _find(b, [b"stts"], s, e)
            n = struct.unpack(">I", b[ts_[0] + 4:ts_[0] + 8])[0]
            for j in range(n):
                cnt, delta = struct.unpack(">II", b[ts_[0] + 8 + 8 * j:ts_[0] + 16 + 8 * j])
                for _ in range(cnt):
                    dts.append(tt)
                    tt += delta
        except KeyError:
            pass
        if len(dts) != count:
            dts = list(range(count))
        self.display_order = sorted(range(count), key=lambda i: (dts[i] + self.cts[i], i))
        cl, cr, ct, cb = self.sps["crop"]
        self.width, self.height = self.sps["mb_w"] * 16 - 2 * (cl + cr), self.sps["mb_h"] * 16 - 2 * (ct + cb)
        # frames per second from the media header (timescale) and the first stts entry
        try:
            md = _find(b, [b"moov", b"trak", b"mdia", b"mdhd"])
            ver = b[md[0]]
            timescale = struct.unpack(">I", b[md[0] + (20 if ver else 12):md[0] + (24 if ver else 16)])[0]
            ts = _find(b, [b"stts"], s, e)
            delta = struct.unpack(">I", b[ts[0] + 12:ts[0] + 16])[0]
            self.fps = timescale / delta if delta else float("nan")
        except (KeyError, struct.error):
            self.fps = float("nan")

    def __len__(self):
        return len(self.sizes)

    def nal_units(self, i):
        sample = self._b[self.offsets[i]:self.offsets[i] + self.sizes[i]]
        nals, q, nls = [], 0, self.nal_length_size
        while q + nls <= len(sample):
            ln = int.from_bytes(sample[q:q + nls], "big")
            nals.append(sample[q + nls:q + nls + ln])
            q += nls + ln
        return nals


# ----------------------------------------------------------------------------------------------------------------- parameter sets


def parse_sps(nal):
    r = Bits(rbsp(nal))
    s = {"profile": r.u(8)}
    r.u(8)
    s["level"] = r.u(8)
    r.ue()
    if s["profile"] not in (66, 77, 100):
        raise NotImplementedError(f"H.264 profile_idc {s['profile']}: only Baseline / Main / High (8-bit 4:2:0) are implemented")
    if s["profile"] == 100:
        cf = r.ue()
        bd_l, bd_c, qpz, ssm = r.ue(), r.ue(), r.u(1), r.u(1)
        if cf != 1 or bd_l or bd_c or qpz or ssm:
            raise NotImplementedError(f"High profile with chroma_format_idc {cf}, bit depths 8 + {bd_l} / 8 + {bd_c}, "
                                      f"qpprime_y_zero_transform_bypass {qpz}, scaling matrices {ssm}: only 8-bit 4:2:0 with flat matrices")
    s["log2_max_frame_num"] = r.ue() + 4
    s["poc_type"] = r.ue()
    if s["poc_type"] == 0:
        s["log2_max_poc_lsb"] = r.ue() + 4
    elif s["poc_type"] == 1:
        r.u(1), r.se(), r.se()
        for _ in range(r.ue()):
            r.se()
    s["num_ref_frames"] = r.ue()
    r.u(1)  # gaps_in_frame_num_value_allowed_flag
    s["mb_w"] = r.ue() + 1
    s["mb_h"] = r.ue() + 1
    s["frame_mbs_only"] = r.u(1)
    if not s["frame_mbs_only"]:
        raise NotImplementedError("interlaced (field / MBAFF) streams are not implemented")
    s["direct_8x8_inference"] = r.u(1)
    s["crop"] = (0, 0, 0, 0)
    if r.u(1):
        s["crop"] = tuple(r.ue() for _ in range(4))  # left, right, top, bottom in 2-pixel units (4:2:0)
    return s


def parse_pps(nal):
    r = Bits(rbsp(nal))
    p = {}
    r.ue(), r.ue()
    p["cabac"] = r.u(1)
    p["pic_order_present"] = r.u(1)
    if r.ue() != 0:
        raise NotImplementedError("slice groups (FMO) are not implemented")
    p["num_ref_idx_default"] = (r.ue() + 1, r.ue() + 1)
    p["weighted_pred"], p["weighted_bipred_idc"] = r.u(1), r.u(2)
    p["pic_init_qp"] = 26 + r.se()
    r.se()
    p["chroma_qp_offset"] = r.se()
    p["deblocking_control"] = r.u(1)
    p["constrained_intra"] = r.u(1)
    p["redundant_pic_cnt"] = r.u(1)
    # High profile tail, present if more_rbsp_data(): transform_8x8_mode_flag, pic_scaling_matrix_present_flag,
    # second_chroma_qp_index_offset
    p["transform8x8"], p["chroma_qp_offset2"] = 0, p["chroma_qp_offset"]
    d = r.d
    n = len(d)
    while n and d[n - 1] == 0:
        n -= 1
    stop = (n - 1) * 8 + 7 - ((d[n - 1] & -d[n - 1]).bit_length() - 1) if n else 0
    if r.p < stop:
        p["transform8x8"] = r.u(1)
        if r.u(1):
            raise NotImplementedError("pic_scaling_matrix_present_flag = 1 (scaling matrices) is not implemented")
        p["chroma_qp_offset2"] = r.se()
    return p


# ----------------------------------------------------------------------------------------------------------------- CABAC tables
# Table 9-44 (rangeTabLPS), Table 9-45 (transIdxLPS; transIdxMPS = min(state + 1, 62))
RANGE_LPS = [
    (128, 176, 208, 240), (128, 167, 197, 227), (128, 158, 187, 216), (123, 150, 178, 205), (116, 142, 169, 195), (111, 135, 160, 185),
    (105, 128, 152, 175), (100, 122, 144, 166), (95, 116, 137, 158), (90, 110, 130, 150), (85, 104, 123, 142), (81, 99, 117, 135),
    (77, 94, 111, 128), (73, 89, 105, 122), (69, 85, 100, 116), (66, 80, 95, 110), (62, 76, 90, 104), (59, 72, 86, 99), (56, 69, 81, 94),
    (53, 65, 77, 89), (51, 62, 73, 85), (48, 59, 69, 80), (46, 56, 66, 76), (43, 53, 63, 72), (41, 50, 59, 69), (39, 48, 56, 65),
    (37, 45, 54, 62), (35, 43, 51, 59), (33, 41, 48, 56), (32, 39, 46, 53), (30, 37, 43, 50), (29, 35, 41, 48), (27, 33, 39, 45),
    (26, 31, 37, 43), (24, 30, 35, 41), (23, 28, 33, 39), (22, 27, 32, 37), (21, 26, 30, 35), (20, 24, 29, 33), (19, 23, 27, 31),
    (18, 22, 26, 30), (17, 21, 25, 28), (16, 20, 23, 27), (15, 19, 22, 25), (14, 18, 21, 24), (14, 17, 20, 23), (13, 16, 19, 22),
    (12, 15, 18, 21), (12, 14, 17, 20), (11, 14, 16, 19), (11, 13, 15, 18), (10, 12, 15, 17), (10, 12, 14, 16), (9, 11, 13, 15),
    (9, 11, 12, 14), (8, 10, 12, 14), (8, 9, 11, 13), (7, 9, 11, 12), (7, 9, 10, 12), (7, 8, 10, 11), (6, 8, 9, 11), (6, 7, 9, 10),
    (6, 7, 8, 9), (2, 2, 2, 2)]
TRANS_LPS = [0, 0, 1, 2, 2, 4, 4, 5, 6, 7, 8, 9, 9, 11, 11, 12, 13, 13, 15, 15, 16, 16, 18, 18, 19, 19, 21, 21, 22, 22, 23, 24, 24, 25, 26, 26,
             27, 27, 28, 29, 29, 30, 30, 30, 31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63]

# (m, n) of Tables 9-12 .. 9-23 for I slices: ctxIdx 0..10, 60..275 (frame-coded blocks, categories 0..4)
CTX_I = {}


def _fill(start, pairs, end=None):
    assert end is None or start + len(pairs) == end + 1, (start, len(pairs), end)
    for k, mn in enumerate(pairs):
        assert start + k not in CTX_I
        CTX_I[start + k] = mn


_fill(0, [(20, -15), (2, 54), (3, 74), (20, -15), (2, 54), (3, 74), (-28, 127), (-23, 104), (-6, 53), (-1, 54), (7, 51)], 10)
_fill(60, [(0, 41), (0, 63), (0, 63), (0, 63), (-9, 83), (4, 86), (0, 97), (-7, 72), (13, 41), (3, 62)], 69)
_fill(70, [(0, 11), (1, 55), (0, 69), (-17, 127), (-13, 102), (0, 82), (-7, 74), (-21, 107), (-27, 127), (-31, 127), (-24, 127), (-18, 95),
           (-27, 127), (-21, 114), (-30, 127), (-17, 123), (-12, 115), (-16, 122)], 87)
_fill(88, [(-11, 115), (-12, 63), (-2, 68), (-15, 84), (-13, 104), (-3, 70), (-8, 93), (-10, 90), (-30, 127), (-1, 74), (-6, 97), (-7, 91),
           (-20, 127), (-4, 56), (-5, 82), (-7, 76), (-22, 125)], 104)
_fill(105, [(-7, 93), (-11, 87), (-3, 77), (-5, 71), (-4, 63), (-4, 68), (-12, 84), (-7, 62), (-7, 65), (8, 61), (5, 56), (-2, 66), (1, 64),
            (0, 61), (-2, 78), (1, 50), (7, 52), (10, 35), (0, 44), (11, 38), (1, 45), (0, 46), (5, 44), (31, 17), (1, 51), (7, 50), (28, 19),
            (16, 33), (14, 62), (-13, 108), (-15, 100)], 135)
_fill(136, [(-13, 101), (-13, 91), (-12, 94), (-10, 88), (-16, 84), (-10, 86), (-7, 83), (-13, 87), (-19, 94), (1, 70), (0, 72), (-5, 74),
            (18, 59), (-8, 102), (-15, 100), (0, 95), (-4, 75), (2, 72), (-11, 75), (-3, 71), (15, 46), (-13, 69), (0, 62), (0, 65), (21, 37),
            (-15, 72), (9, 57), (16, 54), (0, 62), (12, 72)], 165)
_fill(166, [(24, 0), (15, 9), (8, 25), (13, 18), (15, 9), (13, 19), (10, 37), (12, 18), (6, 29), (20, 33), (15, 30), (4, 45), (1, 58), (0, 62),
            (7, 61), (12, 38), (11, 45), (15, 39), (11, 42), (13, 44), (16, 45), (12, 41), (10, 49), (30, 34), (18, 42), (10, 55), (17, 51),
            (17, 46), (0, 89), (26, -19), (22, -17)], 196)
_fill(197, [(26, -17), (30, -25), (28, -20), (33, -23), (37, -27), (33, -23), (40, -28), (38, -17), (33, -11), (40, -15), (41, -6), (38, 1),
            (41, 17), (30, -6), (27, 3), (26, 22), (37, -16), (35, -4), (38, -8), (38, -3), (37, 3), (38, 5), (42, 0), (35, 16), (39, 22),
            (14, 48), (27, 37), (21, 60), (12, 68), (2, 97)], 226)
_fill(227, [(-3, 71), (-6, 42), (-5, 50), (-3, 54), (-2, 62), (0, 58), (1, 63), (-2, 72), (-1, 74), (-9, 91), (-5, 67), (-5, 27), (-3, 39),
            (-2, 44), (0, 46), (-16, 64), (-8, 68), (-10, 78), (-6, 77), (-10, 86), (-12, 92), (-15, 55), (-10, 60), (-6, 62), (-4, 65),
            (-12, 73), (-8, 76), (-7, 80), (-9, 88), (-17, 110), (-11, 97), (-20, 84), (-11, 79), (-6, 73), (-4, 74), (-13, 86), (-13, 96),
            (-11, 97), (-19, 117), (-8, 78), (-5, 33), (-4, 48), (-2, 53), (-3, 62), (-13, 71), (-10, 79), (-12, 86), (-13, 90), (-14, 97)], 275)
assert sorted(CTX_I) == list(range(0, 11)) + list(range(60, 276)), "context table has holes"


class Cabac:
    def __init__(self, bits: Bits, qp: int, table=None):
        self.b = bits
        self.range = 510
        self.offset = bits.u(9)
        self.state = {}
        self.mps = {}
        q = min(max(qp, 0), 51)
        for k, (m, n) in (CTX_I if table is None else table).items():
            pre = min(max(((m * q) >> 4) + n, 1), 126)
            if pre <= 63:
                self.state[k], self.mps[k] = 63 - pre, 0
            else:
                self.state[k], self.mps[k] = pre - 64, 1

    def _renorm(self):
        while self.range < 256:
            self.range <<= 1
            self.offset = (self.offset << 1) | self.b.u(1)

    def decision(self, ctx):
        s = self.state[ctx]
        lps = RANGE_LPS[s][(self.range >> 6) & 3]
        self.range -= lps
        if self.offset >= self.range:
            v = 1 - self.mps[ctx]
            self.offset -= self.range
            self.range = lps
            if s == 0:
                self.mps[ctx] = 1 - self.mps[ctx]
            self.state[ctx] = TRANS_LPS[s]
        else:
            v = self.mps[ctx]
            self.state[ctx] = min(s + 1, 62)
        self._renorm()
        return v

    def bypass(self):
        self.offset = (self.offset << 1) | self.b.u(1)
        if self.offset >= self.range:
            self.offset -= self.range
            return 1
        return 0

    def terminate(self):
        self.range -= 2
        if self.offset >= self.range:
            return 1
        self._renorm()
        return 0


# ----------------------------------------------------------------------------------------------------------------- constants
BLK_XY = [(0, 0), (1, 0), (0, 1), (1, 1), (2, 0), (3, 0), (2, 1), (3, 1), (0, 2), (1, 2), (0, 3), (1, 3), (2, 2), (3, 2), (2, 3), (3, 3)]  # luma4x4BlkIdx -> (x, y)
XY_BLK = {xy: i for i, xy in enumerate(BLK_XY)}
ZIGZAG = [(0, 0), (1, 0), (0, 1), (0, 2), (1, 1), (2, 0), (3, 0), (2, 1), (1, 2), (0, 3), (1, 3), (2, 2), (3, 1), (3, 2), (2, 3), (3, 3)]  # (x, y)
NORM_ADJUST = [(10, 16, 13), (11, 18, 14), (13, 20, 16), (14, 23, 18), (16, 25, 20), (18, 29, 23)]
QPC = list(range(30)) + [29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39]
CAT_CBF = [0, 4, 8, 12, 16]
CAT_SIG = [0, 15, 29, 44, 47]
CAT_ABS = [0, 10, 20, 30, 39]
ALPHA = [0] * 16 + [4, 4, 5, 6, 7, 8, 9, 10, 12, 13, 15, 17, 20, 22, 25, 28, 32, 36, 40, 45, 50, 56, 63, 71, 80, 90, 101, 113, 127, 144, 162, 182,
                    203, 226, 255, 255]
BETA = [0] * 16 + [2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18]
TC0 = [(0, 0, 0)] * 17 + [(0, 0, 1)] * 4 + [(0, 1, 1)] * 2 + [(1, 1, 1)] * 4 + [(1, 1, 2)] * 4 + [(1, 2, 3)] * 2 + [(2, 2, 3), (2, 2, 4), (2, 3, 4), (2, 3, 4),
       (3, 3, 5), (3, 4, 6), (3, 4, 6), (4, 5, 7), (4, 5, 8), (4, 6, 9), (5, 7, 10), (6, 8, 11), (6, 8, 13), (7, 10, 14), (8, 11, 16), (9, 12, 18),
       (10, 13, 20), (11, 15, 23), (13, 17, 25)]
assert len(ALPHA) == 52 and len(BETA) == 52 and len(TC0) == 52 and len(QPC) == 52


def level_scale(qp, x, y):
    v = NORM_ADJUST[qp % 6]
    return 16 * (v[0] if (x % 2 == 0 and y % 2 == 0) else (v[1] if (x % 2 == 1 and y % 2 == 1) else v[2]))


def idct4(d):
    """8.5.12.2: d[y][x] scaled coefficients -> residual r[y][x] (before the (x + 32) >> 6 is applied: it is applied here)."""
    f = [[0] * 4 for _ in range(4)]
    for i in range(4):  # rows
        d0, d1, d2, d3 = d[i]
        e0, e1, e2, e3 = d0 + d2, d0 - d2, (d1 >> 1) - d3, d1 + (d3 >> 1)
        f[i] = [e0 + e3, e1 + e2, e1 - e2, e0 - e3]
    r = [[0] * 4 for _ in range(4)]
    for j in range(4):  # columns
        f0, f1, f2, f3 = f[0][j], f[1][j], f[2][j], f[3][j]
        g0, g1, g2, g3 = f0 + f2, f0 - f2, (f1 >> 1) - f3, f1 + (f3 >> 1)
        col = [g0 + g3, g1 + g2, g1 - g2, g0 - g3]
        for i in range(4):
            r[i][j] = (col[i] + 32) >> 6
    return r


def clip1(v):
    return 0 if v < 0 else (255 if v > 255 else v)


# ----------------------------------------------------------------------------------------------------------------- the picture


class MB:
    __slots__ = ("typ", "i16", "qp", "cbp_luma", "cbp_chroma", "chroma_mode", "modes", "cbf_dc", "cbf_luma", "cbf_cdc", "cbf_cac", "qp_delta_nz")

    def __init__(self):
        self.typ = None       # "I4", "I16"
        self.modes = [2] * 16  # Intra4x4PredMode per luma4x4BlkIdx
        self.cbf_dc = 0
        self.cbf_luma = [0] * 16
        self.cbf_cdc = [0, 0]
        self.cbf_cac = [[0] * 4, [0] * 4]
        self.cbp_luma = 0
        self.cbp_chroma = 0
        self.chroma_mode = 0
        self.qp_delta_nz = 0


class Picture:
    def __init__(self, sps, pps):
        self.sps, self.pps = sps, pps
        self.W, self.Hh = sps["mb_w"], sps["mb_h"]
        self.Y = np.zeros((self.Hh * 16, self.W * 16), np.int32)
        self.C = [np.zeros((self.Hh * 8, self.W * 8), np.int32) for _ in range(2)]
        self.mbs = [None] * (self.W * self.Hh)

    def mb(self, mx, my):
        if mx < 0 or my < 0 or mx >= self.W or my >= self.Hh:
            return None
        return self.mbs[my * self.W + mx]  # None until decoded: "not available"


class NotIntraCoded(NotImplementedError):
    """The requested sample is not an IDR picture made of one I slice (or uses a coding tool this decoder does not have)."""


def decode_intra(track, index=0):
    """-> (Y, Cb, Cr) uint8 planes of sample `index` of `track` (an Mp4H264 or a path), and a dict of facts about the decode."""
    if not isinstance(track, Mp4H264):
        track = Mp4H264(track)
    sps, pps = track.sps, track.pps
    if not pps["cabac"]:
        raise NotIntraCoded("CAVLC entropy coding is not implemented (CABAC streams only)")
    if pps.get("transform8x8"):
        raise NotIntraCoded("the 8x8 transform (High profile) is not implemented in the intra-only module: use io/_h264.py")
    slices = [n for n in track.nal_units(index) if n and (n[0] & 31) in (1, 5)]
    if len(slices) != 1 or (slices[0][0] & 31) != 5:
        raise NotIntraCoded(f"sample {index} of {track.path} is not a single-slice IDR picture: only key frames can be decoded "
                            f"(sync samples of this file: {track.sync[:8]}{'...' if len(track.sync) > 8 else ''})")
    nal = slices[0]
    r = Bits(rbsp(nal) + b"\x00" * 8)  # (the arithmetic decoder reads a few bits ahead of the last symbol)
    n_payload_bits = len(rbsp(nal)) * 8
    first_mb = r.ue()
    slice_type = r.ue()
    if first_mb != 0 or slice_type % 5 != 2:
        raise NotIntraCoded(f"sample {index}: an I slice starting at macroblock 0 was expected (slice_type {slice_type}, first_mb {first_mb})")
    r.ue()
    r.u(sps["log2_max_frame_num"])
    r.ue()  # idr_pic_id
    if sps["poc_type"] == 0:
        r.u(sps["log2_max_poc_lsb"])
        if pps["pic_order_present"]:
            r.se()
    if pps["redundant_pic_cnt"]:
        r.ue()
    if (nal[0] >> 5) & 3:
        r.u(1), r.u(1)  # IDR: no_output_of_prior_pics_flag, long_term_reference_flag
    qp = pps["pic_init_qp"] + r.se()
    disable_deblock, off_a, off_b = 0, 0, 0
    if pps["deblocking_control"]:
        disable_deblock = r.ue()
        if disable_deblock != 1:
            off_a, off_b = 2 * r.se(), 2 * r.se()
    while r.p & 7:  # cabac_alignment_one_bit
        assert r.u(1) == 1
    cab = Cabac(r, qp)
    pic = Picture(sps, pps)
    n_mb = pic.W * pic.Hh
    prev_qp_delta_nz = 0
    stats = {"I4": 0, "I16": 0, "slice_qp": qp, "deblock": disable_deblock != 1, "filter_offsets": (off_a, off_b)}
    for addr in range(n_mb):
        mx, my = addr % pic.W, addr // pic.W
        m = MB()
        A, Bn = pic.mb(mx - 1, my), pic.mb(mx, my - 1)
        # ---- mb_type (9.3.3.1.1.3 / Table 9-36)
        inc = (1 if (A is not None and A.typ != "I4") else 0) + (1 if (Bn is not None and Bn.typ != "I4") else 0)
        if cab.decision(3 + inc) == 0:
            m.typ = "I4"
        else:
            if cab.terminate():
                raise NotIntraCoded("I_PCM macroblocks are not implemented")
            m.typ = "I16"
            ac = cab.decision(3 + 3)
            chroma = 0
            if cab.decision(3 + 4):
                chroma = 1 + cab.decision(3 + 5)
            pm = 2 * cab.decision(3 + 6)
            pm += cab.decision(3 + 7)
            m.i16, m.cbp_luma, m.cbp_chroma = pm, (15 if ac else 0), chroma
        stats[m.typ] += 1
        # ---- prediction modes
        if m.typ == "I4":
            for blk in range(16):
                bx, by = BLK_XY[blk]

                def nmode(dx, dy):
                    x, y = bx + dx, by + dy
                    nb = m if (0 <= x < 4 and 0 <= y < 4) else pic.mb(mx + (x // 4 if x < 0 or x > 3 else 0), my + (y // 4 if y < 0 or y > 3 else 0))
                    if nb is None:
                        return None
                    if nb.typ != "I4":
                        return 2
                    return nb.modes[XY_BLK[(x % 4, y % 4)]]

                ma, mb_ = nmode(-1, 0), nmode(0, -1)
                pred = 2 if (ma is None or mb_ is None) else min(ma, mb_)
                if cab.decision(68):
                    mode = pred
                else:
                    rem = cab.decision(69) | (cab.decision(69) << 1) | (cab.decision(69) << 2)
                    mode = rem if rem < pred else rem + 1
                m.modes[blk] = mode
        # ---- intra_chroma_pred_mode
        inc = (1 if (A is not None and A.chroma_mode != 0) else 0) + (1 if (Bn is not None and Bn.chroma_mode != 0) else 0)
        cm = 0
        if cab.decision(64 + inc):
            cm = 1
            if cab.decision(64 + 3):
                cm = 2
                if cab.decision(64 + 3):
                    cm = 3
        m.chroma_mode = cm
        # ---- coded_block_pattern (I_NxN)
        if m.typ == "I4":
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
        # ---- mb_qp_delta
        if m.typ == "I16" or m.cbp_luma or m.cbp_chroma:
            k = 0
            if cab.decision(60 + prev_qp_delta_nz):
                k = 1
                if cab.decision(60 + 2):
                    k = 2
                    while cab.decision(60 + 3):
                        k += 1
            dqp = (k + 1) // 2 if k & 1 else -(k // 2)
            qp = (qp + dqp + 52) % 52
            m.qp_delta_nz = 1 if dqp else 0
        prev_qp_delta_nz = m.qp_delta_nz
        m.qp = qp
        pic.mbs[addr] = m  # (registered now: blocks of this macroblock are their own neighbours below)

        # ---- residual_block_cabac
        def cbf_of(nb, cat, blk, comp):
            """coded_block_flag of the neighbouring block or None when it is "not available" inside an available macroblock"""
            if cat == 0:
                return nb.cbf_dc if nb.typ == "I16" else None
            if cat in (1, 2):
                x, y = BLK_XY[blk]
                return nb.cbf_luma[blk] if (nb.cbp_luma >> ((y >> 1) * 2 + (x >> 1))) & 1 else None
            if cat == 3:
                return nb.cbf_cdc[comp] if nb.cbp_chroma else None
            return nb.cbf_cac[comp][blk] if nb.cbp_chroma == 2 else None

        def neighbour_flag(cat, bx, by, dx, dy, comp, size):
            x, y = bx + dx, by + dy
            if 0 <= x < size and 0 <= y < size:
                nb, xx, yy = m, x, y
            else:
                nb = A if dx else Bn
                if nb is None:
                    return 1  # not available, current macroblock is intra
                xx, yy = x % size, y % size
            blk = XY_BLK[(xx, yy)] if cat in (1, 2) else (yy * 2 + xx if cat == 4 else 0)
            v = cbf_of(nb, cat, blk, comp)
            return 0 if v is None else v

        def residual(cat, n_coef, bx=0, by=0, comp=0):
            """-> list of n_coef levels in scan order (all zero when coded_block_flag is 0), and the flag"""
            size = 4 if cat in (1, 2) else (2 if cat == 4 else 1)
            if cat in (0, 3):
                fa = neighbour_flag(cat, 0, 0, -1, 0, comp, 1)
                fb = neighbour_flag(cat, 0, 0, 0, -1, comp, 1)
            else:
                fa = neighbour_flag(cat, bx, by, -1, 0, comp, size)
                fb = neighbour_flag(cat, bx, by, 0, -1, comp, size)
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
                    if v == 14:  # Exp-Golomb k = 0 suffix, bypass
                        k = 0
                        while cab.bypass():
                            v += 1 << k
                            k += 1
                        while k:
                            k -= 1
                            v += cab.bypass() << k
                if v == 0:
                    eq1 += 1
                else:
                    gt1 += 1
                coef[i] = -(v + 1) if cab.bypass() else v + 1
            return coef, 1

        qpy = m.qp
        px, py = mx * 16, my * 16
        # luma
        dc16 = None
        if m.typ == "I16":
            lv, m.cbf_dc = residual(0, 16)
            c = [[0] * 4 for _ in range(4)]
            for k, (x, y) in enumerate(ZIGZAG):
                c[y][x] = lv[k]
            # 8.5.10: f = A c A, A = [[1,1,1,1],[1,1,-1,-1],[1,-1,-1,1],[1,-1,1,-1]]
            Am = [[1, 1, 1, 1], [1, 1, -1, -1], [1, -1, -1, 1], [1, -1, 1, -1]]
            t = [[sum(Am[i][k] * c[k][j] for k in range(4)) for j in range(4)] for i in range(4)]
            f = [[sum(t[i][k] * Am[k][j] for k in range(4)) for j in range(4)] for i in range(4)]
            ls = level_scale(qpy, 0, 0)
            if qpy >= 36:
                dc16 = [[(f[i][j] * ls) << (qpy // 6 - 6) for j in range(4)] for i in range(4)]
            else:
                dc16 = [[(f[i][j] * ls + (1 << (5 - qpy // 6))) >> (6 - qpy // 6) for j in range(4)] for i in range(4)]
            pred16(pic, m, mx, my)
        for blk in range(16):
            bx, by = BLK_XY[blk]
            d = [[0] * 4 for _ in range(4)]
            coded = False
            if (m.cbp_luma >> ((by >> 1) * 2 + (bx >> 1))) & 1:
                if m.typ == "I16":
                    lv, m.cbf_luma[blk] = residual(1, 15, bx, by)
                    lv = [0] + lv
                else:
                    lv, m.cbf_luma[blk] = residual(2, 16, bx, by)
                coded = m.cbf_luma[blk] == 1
                for k, (x, y) in enumerate(ZIGZAG):
                    if lv[k]:
                        lsx = level_scale(qpy, x, y)
                        d[y][x] = (lv[k] * lsx) << (qpy // 6 - 4) if qpy >= 24 else (lv[k] * lsx + (1 << (3 - qpy // 6))) >> (4 - qpy // 6)
            if m.typ == "I4":
                pred4(pic, m, mx, my, blk)
            if dc16 is not None:
                d[0][0] = dc16[by][bx]
                coded = coded or d[0][0] != 0
            if coded:
                rr = idct4(d)
                for yy in range(4):
                    for xx in range(4):
                        pic.Y[py + by * 4 + yy, px + bx * 4 + xx] = clip1(int(pic.Y[py + by * 4 + yy, px + bx * 4 + xx]) + rr[yy][xx])
        # chroma
        pred_chroma(pic, m, mx, my)
        qpc = QPC[min(max(qpy + pps["chroma_qp_offset"], 0), 51)]
        dcs = [[0] * 4, [0] * 4]
        if m.cbp_chroma:
            for comp in range(2):
                lv, m.cbf_cdc[comp] = residual(3, 4, comp=comp)
                c = [[lv[0], lv[1]], [lv[2], lv[3]]]
                f = [[c[0][0] + c[0][1] + c[1][0] + c[1][1], c[0][0] - c[0][1] + c[1][0] - c[1][1]],
                     [c[0][0] + c[0][1] - c[1][0] - c[1][1], c[0][0] - c[0][1] - c[1][0] + c[1][1]]]
                ls = level_scale(qpc, 0, 0)
                dcs[comp] = [((f[i][j] * ls) << (qpc // 6)) >> 5 for i in range(2) for j in range(2)]
        acs = [[None] * 4, [None] * 4]
        if m.cbp_chroma == 2:
            for comp in range(2):
                for blk in range(4):
                    lv, m.cbf_cac[comp][blk] = residual(4, 15, blk & 1, blk >> 1, comp)
                    acs[comp][blk] = [0] + lv
        for comp in range(2):
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
                    rr = idct4(d)
                    P = pic.C[comp]
                    for yy in range(4):
                        for xx in range(4):
                            P[my * 8 + by * 4 + yy, mx * 8 + bx * 4 + xx] = clip1(int(P[my * 8 + by * 4 + yy, mx * 8 + bx * 4 + xx]) + rr[yy][xx])
        end = cab.terminate()
        assert end == (1 if addr == n_mb - 1 else 0), f"end_of_slice_flag = {end} at macroblock {addr} of {n_mb}: the CABAC decode lost synchronisation"
    # the arithmetic decoder has read 9 + renormalisation bits ahead: everything but the trailing bits must be consumed
    stats["bits_left"] = n_payload_bits - r.p  # (what remains is rbsp_slice_trailing_bits; the engine is <= 9 + 7 bits ahead)
    assert -16 <= stats["bits_left"] <= 16, stats
    if disable_deblock != 1:
        deblock(pic, off_a, off_b)
    cl, cr, ct, cbm = sps["crop"]
    Y = pic.Y[2 * ct:pic.Hh * 16 - 2 * cbm, 2 * cl:pic.W * 16 - 2 * cr].astype(np.uint8)
    Cb, Cr = (p[ct:pic.Hh * 8 - cbm, cl:pic.W * 8 - cr].astype(np.uint8) for p in pic.C)
    return Y, Cb, Cr, stats


# ----------------------------------------------------------------------------------------------------------------- intra prediction (8.3)


def pred4(pic, m, mx, my, blk):
    bx, by = BLK_XY[blk]
    x0, y0 = mx * 16 + bx * 4, my * 16 + by * 4
    Y = pic.Y
    left = bx > 0 or pic.mb(mx - 1, my) is not None
    top = by > 0 or pic.mb(mx, my - 1) is not None
    # top-right 4x4 block: decoded before this one?
    if by == 0:
        tr = pic.mb(mx + (1 if bx == 3 else 0), my - 1) is not None if bx == 3 else top
    else:
        tr = bx < 3 and XY_BLK[(bx + 1, by - 1)] < blk
    tl = (bx > 0 and by > 0) or (bx > 0 and by == 0 and top) or (bx == 0 and by > 0 and left) or (bx == 0 and by == 0 and pic.mb(mx - 1, my - 1) is not None)
    p = {}
    if top:
        for i in range(4):
            p[(i, -1)] = int(Y[y0 - 1, x0 + i])
        for i in range(4, 8):
            p[(i, -1)] = int(Y[y0 - 1, x0 + i]) if tr else p[(3, -1)]
    if left:
        for j in range(4):
            p[(-1, j)] = int(Y[y0 + j, x0 - 1])
    if tl:
        p[(-1, -1)] = int(Y[y0 - 1, x0 - 1])
    mode = m.modes[blk]
    out = [[0] * 4 for _ in range(4)]
    if mode == 0:
        for y in range(4):
            for x in range(4):
                out[y][x] = p[(x, -1)]
    elif mode == 1:
        for y in range(4):
            for x in range(4):
                out[y][x] = p[(-1, y)]
    elif mode == 2:
        if top and left:
            v = (sum(p[(i, -1)] for i in range(4)) + sum(p[(-1, j)] for j in range(4)) + 4) >> 3
        elif left:
            v = (sum(p[(-1, j)] for j in range(4)) + 2) >> 2
        elif top:
            v = (sum(p[(i, -1)] for i in range(4)) + 2) >> 2
        else:
            v = 128
        out = [[v] * 4 for _ in range(4)]
    elif mode == 3:  # diagonal down-left
        for y in range(4):
            for x in range(4):
                if x == 3 and y == 3:
                    out[y][x] = (p[(6, -1)] + 3 * p[(7, -1)] + 2) >> 2
                else:
                    out[y][x] = (p[(x + y, -1)] + 2 * p[(x + y + 1, -1)] + p[(x + y + 2, -1)] + 2) >> 2
    elif mode == 4:  # diagonal down-right
        for y in range(4):
            for x in range(4):
                if x > y:
                    out[y][x] = (p[(x - y - 2, -1)] + 2 * p[(x - y - 1, -1)] + p[(x - y, -1)] + 2) >> 2
                elif x < y:
                    out[y][x] = (p[(-1, y - x - 2)] + 2 * p[(-1, y - x - 1)] + p[(-1, y - x)] + 2) >> 2
                else:
                    out[y][x] = (p[(0, -1)] + 2 * p[(-1, -1)] + p[(-1, 0)] + 2) >> 2
    elif mode == 5:  # vertical-right
        for y in range(4):
            for x in range(4):
                z = 2 * x - y
                if z >= 0 and z % 2 == 0:
                    out[y][x] = (p[(x - (y >> 1) - 1, -1)] + p[(x - (y >> 1), -1)] + 1) >> 1
                elif z >= 0:
                    out[y][x] = (p[(x - (y >> 1) - 2, -1)] + 2 * p[(x - (y >> 1) - 1, -1)] + p[(x - (y >> 1), -1)] + 2) >> 2
                elif z == -1:
                    out[y][x] = (p[(-1, 0)] + 2 * p[(-1, -1)] + p[(0, -1)] + 2) >> 2
                else:
                    out[y][x] = (p[(-1, y - 1)] + 2 * p[(-1, y - 2)] + p[(-1, y - 3)] + 2) >> 2
    elif mode == 6:  # horizontal-down
        for y in range(4):
            for x in range(4):
                z = 2 * y - x
                if z >= 0 and z % 2 == 0:
                    out[y][x] = (p[(-1, y - (x >> 1) - 1)] + p[(-1, y - (x >> 1))] + 1) >> 1
                elif z >= 0:
                    out[y][x] = (p[(-1, y - (x >> 1) - 2)] + 2 * p[(-1, y - (x >> 1) - 1)] + p[(-1, y - (x >> 1))] + 2) >> 2
                elif z == -1:
                    out[y][x] = (p[(-1, 0)] + 2 * p[(-1, -1)] + p[(0, -1)] + 2) >> 2
                else:
                    out[y][x] = (p[(x - 1, -1)] + 2 * p[(x - 2, -1)] + p[(x - 3, -1)] + 2) >> 2
    elif mode == 7:  # vertical-left
        for y in range(4):
            for x in range(4):
                if y % 2 == 0:
                    out[y][x] = (p[(x + (y >> 1), -1)] + p[(x + (y >> 1) + 1, -1)] + 1) >> 1
                else:
                    out[y][x] = (p[(x + (y >> 1), -1)] + 2 * p[(x + (y >> 1) + 1, -1)] + p[(x + (y >> 1) + 2, -1)] + 2) >> 2
    else:  # 8: horizontal-up
        for y in range(4):
            for x in range(4):
                z = x + 2 * y
                if z > 5:
                    out[y][x] = p[(-1, 3)]
                elif z == 5:
                    out[y][x] = (p[(-1, 2)] + 3 * p[(-1, 3)] + 2) >> 2
                elif z % 2 == 0:
                    out[y][x] = (p[(-1, y + (x >> 1))] + p[(-1, y + (x >> 1) + 1)] + 1) >> 1
                else:
                    out[y][x] = (p[(-1, y + (x >> 1))] + 2 * p[(-1, y + (x >> 1) + 1)] + p[(-1, y + (x >> 1) + 2)] + 2) >> 2
    for y in range(4):
        for x in range(4):
            Y[y0 + y, x0 + x] = out[y][x]


def pred16(pic, m, mx, my):
    Y = pic.Y
    x0, y0 = mx * 16, my * 16
    left, top = pic.mb(mx - 1, my) is not None, pic.mb(mx, my - 1) is not None
    T = [int(Y[y0 - 1, x0 + i]) for i in range(16)] if top else None
    L = [int(Y[y0 + j, x0 - 1]) for j in range(16)] if left else None
    if m.i16 == 0:
        for y in range(16):
            Y[y0 + y, x0:x0 + 16] = T
    elif m.i16 == 1:
        for y in range(16):
            Y[y0 + y, x0:x0 + 16] = L[y]
    elif m.i16 == 2:
        if top and left:
            v = (sum(T) + sum(L) + 16) >> 5
        elif left:
            v = (sum(L) + 8) >> 4
        elif top:
            v = (sum(T) + 8) >> 4
        else:
            v = 128
        Y[y0:y0 + 16, x0:x0 + 16] = v
    else:
        tl = int(Y[y0 - 1, x0 - 1])
        Hh = sum((i + 1) * (T[8 + i] - (T[6 - i] if 6 - i >= 0 else tl)) for i in range(8))
        Vv = sum((j + 1) * (L[8 + j] - (L[6 - j] if 6 - j >= 0 else tl)) for j in range(8))
        a = 16 * (L[15] + T[15])
        b = (5 * Hh + 32) >> 6
        c = (5 * Vv + 32) >> 6
        for y in range(16):
            for x in range(16):
                Y[y0 + y, x0 + x] = clip1((a + b * (x - 7) + c * (y - 7) + 16) >> 5)


def pred_chroma(pic, m, mx, my):
    left, top = pic.mb(mx - 1, my) is not None, pic.mb(mx, my - 1) is not None
    x0, y0 = mx * 8, my * 8
    for P in pic.C:
        T = [int(P[y0 - 1, x0 + i]) for i in range(8)] if top else None
        L = [int(P[y0 + j, x0 - 1]) for j in range(8)] if left else None
        mode = m.chroma_mode
        if mode == 0:  # DC, per 4x4 block (8.3.4.1-3)
            for by in range(2):
                for bx in range(2):
                    st = sum(T[bx * 4:bx * 4 + 4]) if top else None
                    sl = sum(L[by * 4:by * 4 + 4]) if left else None
                    if (bx, by) in ((0, 0), (1, 1)):
                        if top and left:
                            v = (st + sl + 4) >> 3
                        elif top:
                            v = (st + 2) >> 2
                        elif left:
                            v = (sl + 2) >> 2
                        else:
                            v = 128
                    elif (bx, by) == (1, 0):
                        v = (st + 2) >> 2 if top else ((sl + 2) >> 2 if left else 128)
                    else:
                        v = (sl + 2) >> 2 if left else ((st + 2) >> 2 if top else 128)
                    P[y0 + by * 4:y0 + by * 4 + 4, x0 + bx * 4:x0 + bx * 4 + 4] = v
        elif mode == 1:  # horizontal
            for y in range(8):
                P[y0 + y, x0:x0 + 8] = L[y]
        elif mode == 2:  # vertical
            for y in range(8):
                P[y0 + y, x0:x0 + 8] = T
        else:  # plane
            tl = int(P[y0 - 1, x0 - 1])
            Hh = sum((i + 1) * (T[4 + i] - (T[2 - i] if 2 - i >= 0 else tl)) for i in range(4))
            Vv = sum((j + 1) * (L[4 + j] - (L[2 - j] if 2 - j >= 0 else tl)) for j in range(4))
            a = 16 * (L[7] + T[7])
            b = (34 * Hh + 32) >> 6
            c = (34 * Vv + 32) >> 6
            for y in range(8):
                for x in range(8):
                    P[y0 + y, x0 + x] = clip1((a + b * (x - 3) + c * (y - 3) + 16) >> 5)


# ----------------------------------------------------------------------------------------------------------------- deblocking (8.7), intra pictures


def _filter_line(px, bs, alpha, beta, idx_a, luma):
    """px = [p3, p2, p1, p0, q0, q1, q2, q3] (chroma: p3, p2, q2, q3 unused) -> filtered copy"""
    p3, p2, p1, p0, q0, q1, q2, q3 = px
    if not (abs(p0 - q0) < alpha and abs(p1 - p0) < beta and abs(q1 - q0) < beta):
        return px
    o = list(px)
    if bs < 4:
        tc0 = TC0[idx_a][bs - 1]
        if luma:
            ap, aq = abs(p2 - p0), abs(q2 - q0)
            tc = tc0 + (1 if ap < beta else 0) + (1 if aq < beta else 0)
        else:
            tc = tc0 + 1
        delta = min(max((((q0 - p0) << 2) + (p1 - q1) + 4) >> 3, -tc), tc)
        o[3], o[4] = clip1(p0 + delta), clip1(q0 - delta)
        if luma:
            if ap < beta:
                o[2] = p1 + min(max((p2 + ((p0 + q0 + 1) >> 1) - (p1 << 1)) >> 1, -tc0), tc0)
            if aq < beta:
                o[5] = q1 + min(max((q2 + ((p0 + q0 + 1) >> 1) - (q1 << 1)) >> 1, -tc0), tc0)
    else:
        if luma:
            ap, aq = abs(p2 - p0), abs(q2 - q0)
            small = abs(p0 - q0) < ((alpha >> 2) + 2)
            if ap < beta and small:
                o[3] = (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3
                o[2] = (p2 + p1 + p0 + q0 + 2) >> 2
                o[1] = (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3
            else:
                o[3] = (2 * p1 + p0 + q1 + 2) >> 2
            if aq < beta and small:
                o[4] = (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3
                o[5] = (p0 + q0 + q1 + q2 + 2) >> 2
                o[6] = (2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3
            else:
                o[4] = (2 * q1 + q0 + p1 + 2) >> 2
        else:
            o[3] = (2 * p1 + p0 + q1 + 2) >> 2
            o[4] = (2 * q1 + q0 + p1 + 2) >> 2
    return o


def deblock(pic, off_a, off_b):
    cqo = pic.pps["chroma_qp_offset"]

    def qpc(q):
        return QPC[min(max(q + cqo, 0), 51)]

    for my in range(pic.Hh):
        for mx in range(pic.W):
            m = pic.mb(mx, my)
            for vertical in (True, False):  # vertical edges first (filtering across x), then horizontal
                nb = pic.mb(mx - 1, my) if vertical else pic.mb(mx, my - 1)
                for e in range(4):
                    if e == 0 and nb is None:
                        continue
                    bs = 4 if e == 0 else 3
                    qp_p = nb.qp if e == 0 else m.qp
                    planes = [(pic.Y, 16, True, (qp_p + m.qp + 1) >> 1)]
                    if e % 2 == 0:
                        planes += [(P, 8, False, (qpc(qp_p) + qpc(m.qp) + 1) >> 1) for P in pic.C]
                    for P, size, luma, qpav in planes:
                        idx_a, idx_b = min(max(qpav + off_a, 0), 51), min(max(qpav + off_b, 0), 51)
                        alpha, beta = ALPHA[idx_a], BETA[idx_b]
                        if alpha == 0:
                            continue
                        pos = e * 4 if luma else e * 2
                        for k in range(size):
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


def swscale_bgr(Y, Cb, Cr):
    """yuv420p (limited range, BT.601) -> BGR as libswscale's x86 SIMD path produces it -- what cv2.VideoCapture hands to
    sleap.io.video.MediaVideo: 13-bit coefficients, `pmulhw` (truncating) products of the samples shifted left by 3, chroma
    replicated 2 x 2, saturating pack. For a grey stream (Cb = Cr = 128) all three channels are `(8 (Y - 16) * 9539) >> 16`,
    and with THAT formula the fp32 oracle reproduces the reference's TensorFlow-produced golden of frame 0 of
    centered_pair_low_quality.mp4 to the last printed digit (the rounding C-table form of swscale, `(76309 (Y - 16) + 32768)
    >> 16`, is 0.07 px off; tests/test_frame0_golden.py). The chroma coefficients (13-bit forms of 1.596 / 0.813 / 0.391 / 2.018)
    are not pinned by any reference golden."""
    yy = _SWS_Y[Y]                                    # ((Y - 16) << 3) * 9539 >> 16 through a 256-entry table
    if not (Cb != 128).any() and not (Cr != 128).any():  # grey stream: the three channels are the luma term
        g8 = np.clip(yy, 0, 255).astype(np.uint8)
        return np.stack([g8, g8, g8], axis=-1)
    H, W = Y.shape
    up = lambda t: np.repeat(np.repeat(t, 2, 0), 2, 1)[:H, :W]  # noqa: E731  (chroma replicated 2 x 2)
    bu, gu, gv, rv = up(_SWS_BU[Cb]), up(_SWS_GU[Cb]), up(_SWS_GV[Cr]), up(_SWS_RV[Cr])
    return np.stack([np.clip(yy + bu, 0, 255), np.clip(yy - gu - gv, 0, 255), np.clip(yy + rv, 0, 255)], axis=-1).astype(np.uint8)


def swscale_blue(Y, Cb):
    """channel 0 (blue) of `swscale_bgr` alone -- what the reference keeps of a video it flagged grayscale (video.py:482-485);
    a 256-entry uint8 table when the chroma plane is neutral"""
    if not (Cb != 128).any():
        return _SWS_GRAY[Y]
    H, W = Y.shape
    bu = np.repeat(np.repeat(_SWS_BU[Cb], 2, 0), 2, 1)[:H, :W]
    return np.clip(_SWS_Y[Y] + bu, 0, 255).astype(np.uint8)


_SWS_Y = (((np.arange(256, dtype=np.int32) - 16) << 3) * 9539) >> 16
_SWS_GRAY = np.clip(_SWS_Y, 0, 255).astype(np.uint8)
_SWS_C = (np.arange(256, dtype=np.int32) - 128) << 3
_SWS_BU, _SWS_GU, _SWS_GV, _SWS_RV = (_SWS_C * 16531) >> 16, (_SWS_C * 3203) >> 16, (_SWS_C * 6660) >> 16, (_SWS_C * 13075) >> 16


if __name__ == "__main__":
    import sys

    tr = Mp4H264(sys.argv[1])
    print(f"{len(tr)} samples, {tr.width} x {tr.height}, {tr.fps:.2f} fps, sync samples {tr.sync}")
    for i in tr.sync if len(sys.argv) < 3 else [int(sys.argv[2])]:
        Y, Cb, Cr, st = decode_intra(tr, i)
        print(i, Y.shape, st, "luma mean", round(float(Y.mean()), 3), "chroma range", int(Cb.min()), int(Cb.max()), int(Cr.min()), int(Cr.max()))
