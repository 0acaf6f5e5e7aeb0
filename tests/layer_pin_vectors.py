"""Hand-derived known-answer vectors for the Keras layers NO reference-held golden reaches offline (VERDICT r02 missing #4: the
only tight golden of the reference that runs a decoder, `minimal_instance.UNet.bottomup/labels_pr.val.slp`, needs an H.264
frame). Every expected array below is WRITTEN OUT, derived by hand from TensorFlow's documented rules -- not computed by the
oracle or by torch -- so that a wrong crop / padding side / half-pixel convention in `oracle/keras_graph.py` (and, through the
`-m gpu` twin of the test, in the device kernels) fails against something that does not share its reading of those rules.

Rules used (TensorFlow `nn.convolution` / `nn.conv2d_transpose` / `image.resize` documentation):

  SAME padding of a strided window op:  out = ceil(n / s),  pad_total = max((out - 1) * s + k - n, 0),
                                        pad_before = pad_total // 2,  pad_after = pad_total - pad_before   (the EXTRA pixel goes after)
  Conv2D (cross-correlation):           y[i] = sum_k x[s*i + k - pad_before] * w[k]
  Conv2DTranspose = the gradient of that conv w.r.t. its input, for the conv whose INPUT has length s*n:
                                        out[j] = sum over (i, k) with s*i + k - pad_before == j of x[i] * w[k],   0 <= j < s*n
      k3 s2: pad_total = 1 -> pad_before 0: out[j] = sum_{2i+k=j}: the full (2n+1) transposed conv CROPPED AT THE END
      k4 s2: pad_total = 2 -> pad_before 1: out[j] = sum_{2i+k-1=j}: one row/column cropped on each side
      Keras kernel layout (kh, kw, Cout, Cin)
  MaxPooling2D same:                    the padded cells do not take part in the max
  UpSampling2D(bilinear) = tf.image.resize, half_pixel_centers: src = (j + 0.5) / 2 - 0.5, lower = max(floor(src), 0),
                                        upper = min(ceil(src), n - 1), lerp = src - floor(src)
      -> per axis [a, b, c] becomes [a, .75a+.25b, .25a+.75b, .75b+.25c, .25b+.75c, c]
  BatchNormalization (inference):       gamma * (x - mean) / sqrt(var + eps) + beta
  hourglass conv():                     Conv2D(activation=relu) THEN BatchNormalization (hourglass.py:17-45)
  Concatenate([skip, x]):               the skip tensor's channels come FIRST (encoder_decoder.py:360-362)

All numbers are small integers / dyadic fractions: exact in fp16 and bf16 storage, so the device twin asserts equality too.
"""
import numpy as np

F = np.float32

# ---- Conv2DTranspose k3 s2 same, one channel ------------------------------------------------------------------------------
W3 = np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9]], F)  # w[kh][kw]
# 3x3 input, a single 1 at (row 1, col 1): out[2*1 + kh][2*1 + kw] = w[kh][kw]
CONVT3_DELTA_11 = np.array([[0, 0, 0, 0, 0, 0],
                            [0, 0, 0, 0, 0, 0],
                            [0, 0, 1, 2, 3, 0],
                            [0, 0, 4, 5, 6, 0],
                            [0, 0, 7, 8, 9, 0],
                            [0, 0, 0, 0, 0, 0]], F)
# a single 1 at (0, 0): out[kh][kw] = w[kh][kw] -- nothing is cropped at the START
CONVT3_DELTA_00 = np.array([[1, 2, 3, 0, 0, 0],
                            [4, 5, 6, 0, 0, 0],
                            [7, 8, 9, 0, 0, 0],
                            [0, 0, 0, 0, 0, 0],
                            [0, 0, 0, 0, 0, 0],
                            [0, 0, 0, 0, 0, 0]], F)
# a single 1 at (2, 2): out[4 + kh][4 + kw], rows / cols 6 are cropped (the END): only kh, kw in {0, 1} survive
CONVT3_DELTA_22 = np.array([[0, 0, 0, 0, 0, 0],
                            [0, 0, 0, 0, 0, 0],
                            [0, 0, 0, 0, 0, 0],
                            [0, 0, 0, 0, 0, 0],
                            [0, 0, 0, 0, 1, 2],
                            [0, 0, 0, 0, 4, 5]], F)
# x[0][0] = 1 and x[0][1] = 10: the two stamps overlap in column 2 (kw = 2 of the first, kw = 0 of the second)
CONVT3_TWO = np.array([[1, 2, 13, 20, 30, 0],
                       [4, 5, 46, 50, 60, 0],
                       [7, 8, 79, 80, 90, 0],
                       [0, 0, 0, 0, 0, 0],
                       [0, 0, 0, 0, 0, 0],
                       [0, 0, 0, 0, 0, 0]], F)

# ---- Conv2DTranspose k4 s2 same, one channel ------------------------------------------------------------------------------
W4 = np.array([[1, 2, 3, 4], [5, 6, 7, 8], [9, 10, 11, 12], [13, 14, 15, 16]], F)
# a single 1 at (0, 0): out[kh - 1][kw - 1]: kh = kw = 0 falls off the START
CONVT4_DELTA_00 = np.array([[6, 7, 8, 0, 0, 0],
                            [10, 11, 12, 0, 0, 0],
                            [14, 15, 16, 0, 0, 0],
                            [0, 0, 0, 0, 0, 0],
                            [0, 0, 0, 0, 0, 0],
                            [0, 0, 0, 0, 0, 0]], F)
# a single 1 at (1, 1): out[1 + kh][1 + kw], everything inside
CONVT4_DELTA_11 = np.array([[0, 0, 0, 0, 0, 0],
                            [0, 1, 2, 3, 4, 0],
                            [0, 5, 6, 7, 8, 0],
                            [0, 9, 10, 11, 12, 0],
                            [0, 13, 14, 15, 16, 0],
                            [0, 0, 0, 0, 0, 0]], F)
# a single 1 at (2, 2): out[3 + kh][3 + kw]: kh = kw = 3 falls off the END
CONVT4_DELTA_22 = np.array([[0, 0, 0, 0, 0, 0],
                            [0, 0, 0, 0, 0, 0],
                            [0, 0, 0, 0, 0, 0],
                            [0, 0, 0, 1, 2, 3],
                            [0, 0, 0, 5, 6, 7],
                            [0, 0, 0, 9, 10, 11]], F)

# ---- Conv2DTranspose k3 s2 same, 2 -> 2 channels: the Keras layout is (kh, kw, COUT, CIN) ---------------------------------
# w[:, :, co, ci] = M[co][ci] * W3 with M = [[1, 2], [3, 4]]; input = a 1 at (1, 1) of channel 0 and a 2 at (1, 1) of channel 1:
# out channel 0 = (1 * 1 + 2 * 2) * stamp = 5 * stamp, out channel 1 = (3 * 1 + 4 * 2) * stamp = 11 * stamp.
# (Reading the layout as (kh, kw, Cin, Cout) would give 7 and 10.)
CONVT_MIX = np.array([[1, 2], [3, 4]], F)
CONVT_IN_VALUES = (1.0, 2.0)
CONVT3_2CH_SCALE = (5.0, 11.0)

# ---- Conv2D k7 s2 same on 8 x 8 (hourglass / UNet stem): pad 2 before, 3 after ----------------------------------------------
W7 = (10 * np.arange(7)[:, None] + np.arange(7)[None, :]).astype(F)  # w[kh][kw] = 10 kh + kw
# a single 1 at (3, 4): y[i][j] = w[3 - 2i + 2][4 - 2j + 2] = w[5 - 2i][6 - 2j] where that index exists
CONV7S2_DELTA_34 = np.array([[56, 54, 52, 50],
                             [36, 34, 32, 30],
                             [16, 14, 12, 10],
                             [0, 0, 0, 0]], F)

# ---- MaxPooling2D 2x2 s2 same on 5 x 5 (odd): one padded row / column at the END, ignored by the max ------------------------
POOL_IN = np.arange(25, dtype=F).reshape(5, 5)
POOL_OUT = np.array([[6, 8, 9], [16, 18, 19], [21, 23, 24]], F)

# ---- UpSampling2D(2, bilinear) ----------------------------------------------------------------------------------------------
UP_IN = np.array([[0, 4], [8, 12]], F)
UP_BILINEAR = np.array([[0, 1, 3, 4],
                        [2, 3, 5, 6],
                        [6, 7, 9, 10],
                        [8, 9, 11, 12]], F)
UP_IN3 = np.array([[0, 4, 8]], F)  # one row: [a, .75a+.25b, .25a+.75b, .75b+.25c, .25b+.75c, c], rows repeated
UP_BILINEAR3 = np.array([[0, 1, 3, 5, 7, 8], [0, 1, 3, 5, 7, 8]], F)
UP_NEAREST = np.array([[0, 0, 4, 4], [0, 0, 4, 4], [8, 8, 12, 12], [8, 8, 12, 12]], F)

# ---- BatchNormalization ---------------------------------------------------------------------------------------------------
BN_EPS = 1e-3
BN_X = np.array([1.0, 2.0, 3.0], F)
BN_GAMMA = np.array([2.0, 0.5, -1.0], F)
BN_BETA = np.array([0.5, 0.0, 1.0], F)
BN_MEAN = np.array([1.0, 0.0, -1.0], F)
BN_VAR = np.array([4.0, 1.0, 0.25], F) - F(BN_EPS)  # sqrt(var + eps) = 2, 1, 0.5
BN_Y = np.array([0.5, 1.0, -7.0], F)  # 2*(1-1)/2+.5 ; .5*(2-0)/1 ; -1*(3+1)/.5+1
# hourglass conv(): relu first, then BN. A pre-activation of -3 (channel 2 below) becomes 0 and then -1*(0+1)/.5+1 = -1;
# BN first would give -1*(-3+1)/.5+1 = 5 and relu would keep it.
BN_AFTER_RELU_X = np.array([1.0, 2.0, -3.0], F)
BN_AFTER_RELU_Y = np.array([0.5, 1.0, -1.0], F)


def stamp(n, s, k, pad_before, w, at, value=1.0):
    """The placement rule itself, for frames larger than the literal cases: a `value` at input (r, c) adds value * w[kh][kw] at
    out[s*r + kh - pad_before][s*c + kw - pad_before] wherever that lies inside the (s*n) x (s*n) output."""
    out = np.zeros((s * n, s * n), F)
    r, c = at
    for kh in range(k):
        for kw in range(k):
            y, x = s * r + kh - pad_before, s * c + kw - pad_before
            if 0 <= y < s * n and 0 <= x < s * n:
                out[y, x] += F(value) * w[kh][kw]
    return out
