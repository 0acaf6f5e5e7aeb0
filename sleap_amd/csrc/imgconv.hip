// First-layer convolution of a uint8 image on the matrix cores: Conv2D(k7, stride 1|2) on 1 or 3 channels, the stem of
// the hourglass (hourglass.py:75-85: Conv2D(k7, s2, same, relu) + BatchNormalization) and of ResNet (resnet.py:109-121:
// [tile_channels, imagenet_preproc_v1,] ZeroPadding2D(3) + Conv2D(k7, s2, valid) + BN + ReLU).
//
// GEMM view per 32-pixel output row segment: A = weights (32 couts x 16 k), B = im2col of the raw tile (16 k x 32 pixels),
// k = (dy*KW + dx)*CINW + c, K = 49*CINW padded to a multiple of 16. The raw tile lives in LDS as bf16 -- uint8 values are
// EXACT in bf16 -- and each lane gathers its 8 k-values per step with 16-bit LDS reads (compile-time tap offsets).
// The fp32 kernel times the input scale (1/255 from ensure_float, x255 from imagenet_preproc_v1) enters as hi + lo bf16
// fragments (two MFMAs, 16 mantissa bits), pre-packed on the host. The ImageNet channel means can not ride in the B operand
// (103.939 is not a bf16 number): for interior pixels they are a constant folded into the bias on the host; where a tap
// falls outside the image the reference pads the PREPROCESSED tensor with zeros, i.e. that tap must NOT receive the mean,
// which is added back through extra K steps whose B operand is the exact 0/1 out-of-image indicator of every tap and whose
// A operand is v[tap][cout] = sum_c w[tap][c][cout] * mean_c (only tiles that touch the border execute them).
#include <cstdint>
#include <cstring>
#include <type_traits>
#include <utility>
#include <vector>

#include "bf16.h"
#include "sa_common.h"

namespace {

using sa::h16x8_t;
using sa::mfma_h8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct ImgConvParams {
  const uint8_t* src;    // [B,H,W,CIN] u8 (CIN = 1 feeding CINW = 3 weight channels = tile_channels)
  const uint16_t* wfrag;  // packed [co32][NK16 + NKI16][2 terms][64][8] bf16 (sa_imgconv_pack)
  const float* bias;     // [CoutP], mean term already folded in
  const float* post_scale;
  const float* post_shift;
  uint16_t* dst;         // [B,Ho,Wo,CoutP] bf16
  int B, H, W, Ho, Wo, CoutP, pad_t, pad_l, relu, has_mean, src_c;
  int planar;            // dst in 16-channel planes [B, CoutP/16, Ho, Wo, 16] instead of NHWC
  unsigned pix_elems, blk_elems;  // elements between neighbouring pixels / 16-channel blocks of dst
  int tiles_x, tiles_y;
};

template <int KH, int CINW, int STRIDE>
__global__ void __launch_bounds__(256)
imgconv_mfma_kernel(const ImgConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int KW = KH, TH = 8, TW = 32, R = 2;
  constexpr int RH = (TH - 1) * STRIDE + KH, RW = (TW - 1) * STRIDE + KW;
  constexpr int KT = KH * KW * CINW, NK16 = (KT + 15) / 16, NKI16 = (KH * KW + 15) / 16, NKALL = NK16 + NKI16;
  __shared__ __attribute__((aligned(16))) uint16_t raw[RH * RW * CINW + 8];
  __shared__ __attribute__((aligned(16))) uint16_t oob[RH * RW + 8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, lx = lane & 31;
  int bid = blockIdx.x;
  const int tx_i = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int ty_i = bid % p.tiles_y;
  const int b = bid / p.tiles_y;
  const int ox0 = tx_i * TW, oy0 = ty_i * TH;
  const int iy0 = oy0 * STRIDE - p.pad_t, ix0 = ox0 * STRIDE - p.pad_l;
  const int H = p.H, W = p.W;
  // does any tap of this tile fall outside the image? (wave-uniform)
  const bool border = p.has_mean && (iy0 < 0 || ix0 < 0 || iy0 + RH > H || ix0 + RW > W);
  for (int i = tid; i < RH * RW; i += 256) {
    const int ty = i / RW, tx = i - ty * RW;
    const int gy = iy0 + ty, gx = ix0 + tx;
    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
    const uint8_t* s = p.src + (((size_t)b * H + (in ? gy : 0)) * W + (in ? gx : 0)) * p.src_c;
#pragma unroll
    for (int c = 0; c < CINW; ++c) {
      const float v = in ? (float)s[p.src_c == 1 ? 0 : c] : 0.0f;
      raw[i * CINW + c] = sa::f2h(v * sa::U8_ACT_SCALE);  // exact in bf16 and in fp16
    }
    oob[i] = in ? (uint16_t)0 : sa::f2h(1.0f);  // 1.0 where the tap is outside the image
  }
  __syncthreads();

  const int co32_n = (p.CoutP + 31) / 32;
  const uint4* wf = reinterpret_cast<const uint4*>(p.wfrag);
  for (int cp = 0; cp < co32_n; cp += 2) {  // two 32-cout tiles per pass
    f32x16 acc[2][R];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][r][i] = 0.0f;
    const bool two = cp + 1 < co32_n;
    const uint16_t* lane_raw = raw + ((wave * R * STRIDE) * RW + lx * STRIDE) * CINW;
    const uint16_t* lane_oob = oob + (wave * R * STRIDE) * RW + lx * STRIDE;
    // one K step: the 8 k-values of a lane are k = ks*16 + half*8 + j; after full unrolling the tap offsets of both
    // halves are compile-time constants, the lane picks its half with one v_cndmask per element
    auto step = [&](auto ks_c, auto ind_c) {
      constexpr int ks = decltype(ks_c)::value;
      constexpr bool ind = decltype(ind_c)::value;
      constexpr int kstep = ind ? NK16 + ks : ks;
      mfma_h8 a[2][2];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int c32 = (m == 1 && !two) ? cp : cp + m;
          a[m][t] = __builtin_bit_cast(mfma_h8, wf[(((size_t)c32 * NKALL + kstep) * 2 + t) * 64 + lane]);
        }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        h16x8_t bq;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          constexpr int CW = ind ? 1 : CINW, KLIM = ind ? KH * KW : KT;
          const int k0 = ks * 16 + j, k1 = ks * 16 + 8 + j;
          const int o0 = ((k0 / CW / KW) * RW + (k0 / CW) % KW) * CW + k0 % CW + r * STRIDE * RW * CW;
          const int o1 = ((k1 / CW / KW) * RW + (k1 / CW) % KW) * CW + k1 % CW + r * STRIDE * RW * CW;
          const bool v0 = k0 < KLIM, v1 = k1 < KLIM;
          if (!v0 && !v1) {
            bq[j] = 0;
          } else {
            const uint16_t* bp = ind ? lane_oob : lane_raw;
            const uint16_t x = bp[half ? (v1 ? o1 : 0) : (v0 ? o0 : 0)];
            bq[j] = (half ? v1 : v0) ? x : (uint16_t)0;
          }
        }
        const mfma_h8 bf = __builtin_bit_cast(mfma_h8, bq);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int t = 0; t < 2; ++t) acc[m][r] = SA_MFMA_32x32x16(a[m][t], bf, acc[m][r], 0, 0, 0);
      }
    };
    auto run = [&](auto ind_c, auto... ks) { (step(ks, ind_c), ...); };
    auto seq = [&](auto ind_c, auto n_c) {
      constexpr int n = decltype(n_c)::value;
      [&]<int... I>(std::integer_sequence<int, I...>) { run(ind_c, std::integral_constant<int, I>{}...); }
      (std::make_integer_sequence<int, n>{});
    };
    seq(std::false_type{}, std::integral_constant<int, NK16>{});
    if (border) seq(std::true_type{}, std::integral_constant<int, NKI16>{});
    // ---- epilogue: lane holds channels (i&3) + 8*(i>>2) + 4*half of output pixel (oy0 + wave*R + r, ox0 + lx)
    const float lowv = p.relu ? 0.0f : -INFINITY;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      if (m == 1 && !two) continue;
      const int cobase = (cp + m) * 32;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int gy = oy0 + wave * R + r, gx = ox0 + lx;
        const bool ok = gy < p.Ho && gx < p.Wo;
        uint2 pk[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int co = cobase + 8 * g + 4 * half;
          float4 bq = make_float4(0.f, 0.f, 0.f, 0.f), sq = make_float4(1.f, 1.f, 1.f, 1.f), tq = bq;
          if (co < p.CoutP) {
            bq = *reinterpret_cast<const float4*>(p.bias + co);
            if (p.post_scale) {
              sq = *reinterpret_cast<const float4*>(p.post_scale + co);
              tq = *reinterpret_cast<const float4*>(p.post_shift + co);
            }
          }
          const float bb[4] = {bq.x, bq.y, bq.z, bq.w}, ss[4] = {sq.x, sq.y, sq.z, sq.w}, tt[4] = {tq.x, tq.y, tq.z, tq.w};
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaf(fmaxf(acc[m][r][4 * g + j] + bb[j], lowv), ss[j], tt[j]);
          pk[g].x = sa::f2h2(v[0], v[1]);
          pk[g].y = sa::f2h2(v[2], v[3]);
        }
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          uint2 x = pk[2 * pr], y = pk[2 * pr + 1];
          sa::swap32(x.x, y.x);
          sa::swap32(x.y, y.y);
          const int co = cobase + 16 * pr + 8 * half;
          if (ok && co < p.CoutP) {
            // one formula for both layouts (element strides from the host): NHWC pix = CoutP, blk = 16; planes pix = 16, blk = Ho Wo 16
            uint16_t* o = p.dst + (size_t)b * p.Ho * p.Wo * p.CoutP + (size_t)(co >> 4) * p.blk_elems +
                          ((size_t)gy * p.Wo + gx) * p.pix_elems + (co & 15);
            *reinterpret_cast<uint4*>(o) = make_uint4(x.x, x.y, y.x, y.y);
          }
        }
      }
    }
  }
#endif
}

template <int KH, int CINW, int STRIDE>
int launch_imgconv(ImgConvParams p, hipStream_t st) {
  p.tiles_x = (p.Wo + 31) / 32;
  p.tiles_y = (p.Ho + 7) / 8;
  const size_t nblk = (size_t)p.tiles_x * p.tiles_y * p.B;
  if (nblk == 0 || nblk > 0x7fffffffull) return sa::fail(SA_ERR_INVALID_ARG, "sa_imgconv_u8_bf16: bad grid");
  hipLaunchKernelGGL((imgconv_mfma_kernel<KH, CINW, STRIDE>), dim3((unsigned)nblk), dim3(256), 0, st, p);
  SA_LAUNCH_CHECK();
  return SA_OK;
}

}  // namespace

extern "C" {

size_t sa_imgconv_packed_elems(int ksize, int CinW, int CoutP) {
  const size_t nk = (size_t)(ksize * ksize * CinW + 15) / 16 + (size_t)(ksize * ksize + 15) / 16;
  return (size_t)((CoutP + 31) / 32) * nk * 2 * 64 * 8;
}

// shared packer. `tiled`: w has CinS = 3 weight channels but the IMAGE has one (the `tile_channels` Lambda of ResNet's
// pretrained-encoder input, resnet.py:326-362: a grayscale frame repeated three times): the three products of a tap share their
// pixel, so sum_c w[tap][c] * s_c * x collapses to ONE K slot per tap with the summed weight (float64 sum, then the hi + lo
// split) -- K = 49 instead of 147: 4 instead of 10 k-steps, 64 instead of 112 MFMAs per tile and the CINW = 1 kernel (218
// registers, two waves per SIMD) instead of the CINW = 3 one (256 + spills, one wave). The mean / indicator term is already
// per tap. Packed layout = the CinW = 1 layout.
static int imgconv_pack_impl(const float* w, int ksize, int CinS, bool tiled, int Cout, int CoutP, const float* in_scale,
                             const float* mean, uint16_t* packed, float* bias_io) {
  const int CinW = tiled ? 1 : CinS;
  const int KT = ksize * ksize * CinW, NK16 = (KT + 15) / 16, NKI16 = (ksize * ksize + 15) / 16, NKALL = NK16 + NKI16;
  const int co32_n = (CoutP + 31) / 32;
  auto split = [&](float v, uint16_t* hi, uint16_t* lo) {
    *hi = sa::f2h(v);
    *lo = sa::f2h(v - sa::h2f(*hi));
  };
  std::vector<double> vsum((size_t)CoutP, 0.0);
  for (int c32 = 0; c32 < co32_n; ++c32)
    for (int ks = 0; ks < NKALL; ++ks)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
          const int co = c32 * 32 + (lane & 31);
          float v = 0.0f;
          if (co < Cout) {
            if (ks < NK16) {
              const int k = ks * 16 + (lane >> 5) * 8 + j;
              if (k < KT) {
                const int tap = k / CinW, c = k % CinW;
                if (tiled) {
                  double acc = 0.0;
                  for (int cc = 0; cc < CinS; ++cc)
                    acc += (double)w[((size_t)tap * CinS + cc) * Cout + co] * (in_scale ? (double)in_scale[cc] : 1.0);
                  v = (float)(acc * (double)(1.0f / sa::U8_ACT_SCALE));
                } else {
                  v = w[((size_t)tap * CinW + c) * Cout + co] * (in_scale ? in_scale[c] : 1.0f) * (1.0f / sa::U8_ACT_SCALE);
                }
              }
            } else if (mean) {
              const int tap = (ks - NK16) * 16 + (lane >> 5) * 8 + j;
              if (tap < ksize * ksize) {
                double acc = 0.0;
                for (int c = 0; c < CinS; ++c) acc += (double)w[((size_t)tap * CinS + c) * Cout + co] * mean[c];
                v = (float)acc;
                vsum[co] += acc;  // every (cout, tap) pair is visited exactly once
              }
            }
          }
          uint16_t hi, lo;
          split(v, &hi, &lo);
          const size_t base = ((((size_t)c32 * NKALL + ks) * 2) * 64 + lane) * 8 + j;
          packed[base] = hi;
          packed[base + 64 * 8] = lo;
        }
  if (mean)
    for (int co = 0; co < Cout; ++co) bias_io[co] = (float)((double)bias_io[co] - vsum[co]);
  return SA_OK;
}

int sa_imgconv_pack(const float* w, int ksize, int CinW, int Cout, int CoutP, const float* in_scale, const float* mean,
                    uint16_t* packed, float* bias_io) {
  // w [k][k][CinW][Cout] f32 (Keras layout); in_scale[c] multiplies the raw 0..255 pixel value of weight channel c
  // (1/255 for ensure_float alone, 1 with imagenet_preproc_v1); mean[c] (or NULL) is subtracted after scaling.
  // bias_io [CoutP]: in = layer bias, out = bias - sum over ALL taps of v (the interior-pixel constant).
  SA_REQUIRE(w && packed && bias_io && ksize > 0 && (CinW == 1 || CinW == 3) && Cout <= CoutP, "sa_imgconv_pack: bad arguments");
  return imgconv_pack_impl(w, ksize, CinW, false, Cout, CoutP, in_scale, mean, packed, bias_io);
}

int sa_imgconv_pack_tiled(const float* w3, int ksize, int Cout, int CoutP, const float* in_scale3, const float* mean3,
                          uint16_t* packed, float* bias_io) {
  SA_REQUIRE(w3 && packed && bias_io && ksize > 0 && Cout <= CoutP, "sa_imgconv_pack_tiled: bad arguments");
  return imgconv_pack_impl(w3, ksize, 3, true, Cout, CoutP, in_scale3, mean3, packed, bias_io);
}

int sa_imgconv_u8_bf16(const void* src, int B, int H, int W, int Cin, int CinW, int ksize, int stride, int pad_top,
                       int pad_left, int Ho, int Wo, const void* wfrag, const float* bias, int CoutP, int relu,
                       int has_mean, const float* post_scale, const float* post_shift, void* dst, sa_stream_t stream) {
  SA_REQUIRE(src && wfrag && bias && dst, "sa_imgconv_u8_bf16: NULL pointer");
  SA_REQUIRE(B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && CoutP % 16 == 0, "sa_imgconv_u8_bf16: bad shape");
  SA_REQUIRE((Cin == CinW || Cin == 1) && (CinW == 1 || CinW == 3), "sa_imgconv_u8_bf16: channels must be 1 or 3");
  SA_REQUIRE(!post_scale == !post_shift, "sa_imgconv_u8_bf16: post_scale and post_shift come together");
  ImgConvParams p = {};
  p.src = (const uint8_t*)src;
  p.wfrag = (const uint16_t*)wfrag;
  p.bias = bias;
  p.post_scale = post_scale;
  p.post_shift = post_shift;
  p.dst = (uint16_t*)dst;
  p.B = B;
  p.H = H;
  p.W = W;
  p.Ho = Ho;
  p.Wo = Wo;
  p.CoutP = CoutP;
  p.pad_t = pad_top;
  p.pad_l = pad_left;
  p.relu = relu & 1;  // (`relu`: bit 0 = ReLU, SA_LAYOUT_PLANES16 = write 16-channel planes)
  p.planar = (relu & SA_LAYOUT_PLANES16) ? 1 : 0;
  SA_REQUIRE((size_t)Ho * Wo * 16 < 0xFFFFFFFFull, "sa_imgconv_u8_bf16: output plane too large");
  p.pix_elems = p.planar ? 16u : (unsigned)CoutP;
  p.blk_elems = p.planar ? (unsigned)((size_t)Ho * Wo * 16) : 16u;
  p.has_mean = has_mean;
  p.src_c = Cin;
  hipStream_t st = (hipStream_t)stream;
  if (ksize == 7 && stride == 2) return CinW == 1 ? launch_imgconv<7, 1, 2>(p, st) : launch_imgconv<7, 3, 2>(p, st);
  if (ksize == 7 && stride == 1) return CinW == 1 ? launch_imgconv<7, 1, 1>(p, st) : launch_imgconv<7, 3, 1>(p, st);
  return sa::fail(SA_ERR_UNSUPPORTED, "sa_imgconv_u8_bf16: only 7x7 kernels with stride 1 or 2 (use sa_image_conv_bf16)");
}

}  // extern "C"
