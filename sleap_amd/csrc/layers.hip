// Bandwidth-bound network layers on bf16 NHWC activations for gfx950: the u8->f32 normalising stem
// convolution, 2x2 max-pool, x2 upsampling, the 1x1 linear heads, the (fixture-only) 3x3 stride-2
// transposed convolution and dtype plumbing. The MFMA 3x3 convolution lives in conv3x3.hip.
//
// All kernels move 16 B per lane (8 bf16 channels) so that a wave touches whole 1 KiB lines.
#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "bf16.h"
#include "sa_common.h"

namespace {

using sa::h16x8_t;

// ------------------------------------------------------------------------------------------------
// stem: ensure_float (x * 1/255, normalization.py:34-49) + Conv2D(k3,same) + bias + ReLU, Cin in {1,3}
// thread = (pixel, group of 8 output channels); fp32 arithmetic, bf16 store.
// ------------------------------------------------------------------------------------------------
template <int CIN, bool U8>
__global__ void __launch_bounds__(256)
stem_conv3x3_kernel(const void* __restrict__ src_, int B, int H, int W, const float* __restrict__ w,
                    const float* __restrict__ bias, int CoutP, int relu, uint16_t* __restrict__ dst) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* sw = reinterpret_cast<float*>(smem_raw);  // [9*CIN][CoutP] + bias[CoutP]
  const int nw = 9 * CIN * CoutP;
  for (int i = threadIdx.x; i < nw + CoutP; i += blockDim.x) sw[i] = (i < nw) ? w[i] : bias[i - nw];
  __syncthreads();
  const int groups = CoutP / 8;
  const size_t total = (size_t)B * H * W * groups;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(t % groups);
    const size_t p = t / groups;
    const int x = (int)(p % W);
    const int y = (int)((p / W) % H);
    const size_t b = p / ((size_t)W * H);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = sw[nw + g * 8 + j];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int yy = y + dy - 1;
      if (yy < 0 || yy >= H) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int xx = x + dx - 1;
        if (xx < 0 || xx >= W) continue;
        const size_t si = ((b * H + yy) * W + xx) * CIN;
#pragma unroll
        for (int c = 0; c < CIN; ++c) {
          float v;
          if (U8)
            v = (float)reinterpret_cast<const uint8_t*>(src_)[si + c] * (1.0f / 255.0f);
          else
            v = reinterpret_cast<const float*>(src_)[si + c];
          const float* wr = sw + ((dy * 3 + dx) * CIN + c) * CoutP + g * 8;
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = fmaf(v, wr[j], acc[j]);
        }
      }
    }
    h16x8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = sa::f2h(relu ? fmaxf(acc[j], 0.0f) : acc[j]);
    *reinterpret_cast<h16x8_t*>(dst + p * CoutP + g * 8) = o;
  }
}

// ------------------------------------------------------------------------------------------------
// General first-layer convolution on the raw image (any kernel size / stride / explicit TF SAME pads), e.g. the
// hourglass stem Conv2D(k7, s2, same)+ReLU+BatchNormalization (hourglass.py:75-85) or ResNet's
// ZeroPadding2D(3)+Conv2D(k7, s2, valid). ensure_float fused for uint8 input. fp32 arithmetic, bf16 store.
// thread = (output pixel, group of 8 output channels); weights [kh][kw][Cin][CoutP] f32 from LDS.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
image_conv_kernel(const void* __restrict__ src_, int is_u8, int B, int H, int W, int Cin, int CinW,
                  const float* __restrict__ in_affine, int kh, int kw, int stride,
                  int pad_t, int pad_l, int Ho, int Wo, const float* __restrict__ w, const float* __restrict__ bias,
                  int CoutP, int relu, const float* __restrict__ post_scale, const float* __restrict__ post_shift,
                  uint16_t* __restrict__ dst, int planar) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* sw = reinterpret_cast<float*>(smem_raw);
  const int nw = kh * kw * CinW * CoutP;
  for (int i = threadIdx.x; i < nw; i += blockDim.x) sw[i] = w[i];
  __syncthreads();
  const int groups = CoutP / 8;
  const size_t total = (size_t)B * Ho * Wo * groups;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(t % groups);
    const size_t p = t / groups;
    const int x = (int)(p % Wo);
    const int y = (int)((p / Wo) % Ho);
    const size_t b = p / ((size_t)Wo * Ho);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = bias[g * 8 + j];
    for (int dy = 0; dy < kh; ++dy) {
      const int yy = y * stride + dy - pad_t;
      if (yy < 0 || yy >= H) continue;
      for (int dx = 0; dx < kw; ++dx) {
        const int xx = x * stride + dx - pad_l;
        if (xx < 0 || xx >= W) continue;
        const size_t si = ((b * H + yy) * W + xx) * Cin;
        for (int c = 0; c < CinW; ++c) {
          const int cs = Cin == 1 ? 0 : c;  // tile_channels (resnet.py:326-339): a grayscale image feeds every channel
          float v = is_u8 ? (float)reinterpret_cast<const uint8_t*>(src_)[si + cs] * (1.0f / 255.0f)
                          : reinterpret_cast<const float*>(src_)[si + cs];
          if (in_affine) v = __fadd_rn(__fmul_rn(v, in_affine[c]), in_affine[CinW + c]);  // un-fused mul/add as the reference's X*255 - mean
          const float* wr = sw + ((dy * kw + dx) * CinW + c) * CoutP + g * 8;
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = fmaf(v, wr[j], acc[j]);
        }
      }
    }
    h16x8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = relu ? fmaxf(acc[j], 0.0f) : acc[j];
      if (post_scale) v = fmaf(v, post_scale[g * 8 + j], post_shift[g * 8 + j]);
      o[j] = sa::f2h(v);
    }
    if (planar)  // 16-channel planes [B, CoutP/16, Ho, Wo, 16]
      *reinterpret_cast<h16x8_t*>(dst + ((b * (CoutP / 16) + (g >> 1)) * (size_t)Ho * Wo + (size_t)y * Wo + x) * 16 + (g & 1) * 8) = o;
    else
      *reinterpret_cast<h16x8_t*>(dst + p * CoutP + g * 8) = o;
  }
}

// general MaxPooling2D (window k, stride, explicit pads); thread = (output pixel, 8 channels)
__global__ void __launch_bounds__(256)
maxpool_kernel(const uint16_t* __restrict__ src, int B, int H, int W, int CP, int k, int stride, int pad_t, int pad_l,
               int pad_zero, int Ho, int Wo, uint16_t* __restrict__ dst) {
  const int groups = CP / 8;
  const size_t total = (size_t)B * Ho * Wo * groups;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(t % groups);
    const size_t p = t / groups;
    const int x = (int)(p % Wo);
    const int y = (int)((p / Wo) % Ho);
    const size_t b = p / ((size_t)Wo * Ho);
    float m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
    for (int dy = 0; dy < k; ++dy)
      for (int dx = 0; dx < k; ++dx) {
        const int yy = y * stride + dy - pad_t, xx = x * stride + dx - pad_l;
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) {
          if (pad_zero) {
#pragma unroll
            for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], 0.0f);
          }
          continue;
        }
        const h16x8_t v = *reinterpret_cast<const h16x8_t*>(src + ((b * H + yy) * W + xx) * CP + g * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], sa::h2f(v[j]));
      }
    h16x8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = sa::f2h(m[j]);
    *reinterpret_cast<h16x8_t*>(dst + p * CP + g * 8) = o;
  }
}

// MaxPooling2D(3, s2) on ONE 16-channel plane per "frame" (ResNet's ZeroPadding2D(1) + MaxPool(3, s2) after the stem, resnet.py:
// 109-126, run per plane), round 5: the general kernel above reads nine 16-byte pieces per output piece straight from global
// memory -- 2.25 x the input through the vector cache, 0.207 ms for 16 frames of 512 x 512 x 64 (3.2 TB/s of algorithmic bytes).
// Here a workgroup copies the (2 TO + 1) x 65 input pixels under its TO x 32 outputs into LDS once (row-contiguous 16-byte
// loads: every fetched line is used completely; 1.14 x the input) and takes the nine taps from there; a wave = one output row,
// a store instruction = 1 KiB contiguous. Taps outside the image are 0 (pad_zero) or neutral (-inf): the general kernel's
// result, bit for bit (max of stored values is exact).
template <int TO>
__global__ void __launch_bounds__(64 * TO)
maxpool3x3s2_c16_kernel(const uint16_t* __restrict__ src, int H, int W, int pad_t, int pad_l, int pad_zero, int Ho, int Wo,
                        uint16_t* __restrict__ dst) {
  constexpr int IR = 2 * TO + 1, IC = 65;
  __shared__ __attribute__((aligned(16))) h16x8_t tile[IR * IC * 2];
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * 32, y0 = blockIdx.y * TO;
  const size_t b = blockIdx.z;
  const uint16_t* frame = src + b * (size_t)H * W * 16;
  const uint16_t fillv = sa::f2h(pad_zero ? 0.0f : -INFINITY);
  const h16x8_t fill = {fillv, fillv, fillv, fillv, fillv, fillv, fillv, fillv};
  for (int i = tid; i < IR * IC * 2; i += 64 * TO) {
    const int pix = i >> 1, hf = i & 1;
    const int r = pix / IC, c = pix - r * IC;
    const int gy = 2 * y0 - pad_t + r, gx = 2 * x0 - pad_l + c;
    h16x8_t v = fill;
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = *reinterpret_cast<const h16x8_t*>(frame + ((size_t)gy * W + gx) * 16 + hf * 8);
    tile[i] = v;
  }
  __syncthreads();
  const int hf = tid & 1, ox = (tid >> 1) & 31, oy = tid >> 6;
  const int gy = y0 + oy, gx = x0 + ox;
  if (gy >= Ho || gx >= Wo) return;
  float m[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const h16x8_t v = tile[((2 * oy + dy) * IC + 2 * ox + dx) * 2 + hf];
#pragma unroll
      for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], sa::h2f(v[j]));
    }
  h16x8_t o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = sa::f2h(m[j]);
  *reinterpret_cast<h16x8_t*>(dst + ((b * Ho + gy) * (size_t)Wo + gx) * 16 + hf * 8) = o;
}

// Add layer: dst = a + b (bf16, same shape); b may be half resolution read with nearest-neighbour x2 upsampling
__global__ void __launch_bounds__(256)
add_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ bsrc, int B, int H, int W, int CP, int b_half,
           int relu, uint16_t* __restrict__ dst) {
  const int groups = CP / 8;
  const size_t total = (size_t)B * H * W * groups;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(t % groups);
    const size_t p = t / groups;
    const int x = (int)(p % W);
    const int y = (int)((p / W) % H);
    const size_t b = p / ((size_t)W * H);
    const h16x8_t va = *reinterpret_cast<const h16x8_t*>(a + p * CP + g * 8);
    const size_t pb = b_half ? ((b * (H / 2) + (y >> 1)) * (W / 2) + (x >> 1)) : p;
    const h16x8_t vb = *reinterpret_cast<const h16x8_t*>(bsrc + pb * CP + g * 8);
    h16x8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = sa::h2f(va[j]) + sa::h2f(vb[j]);
      o[j] = sa::f2h(relu ? fmaxf(v, 0.0f) : v);
    }
    *reinterpret_cast<h16x8_t*>(dst + p * CP + g * 8) = o;
  }
}

// ------------------------------------------------------------------------------------------------
// MaxPooling2D(2, s2) on even sizes
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
maxpool2x2_kernel(const uint16_t* __restrict__ src, int B, int H, int W, int CP, uint16_t* __restrict__ dst) {
  const int Ho = H / 2, Wo = W / 2, groups = CP / 8;
  const size_t total = (size_t)B * Ho * Wo * groups;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(t % groups);
    const size_t p = t / groups;
    const int x = (int)(p % Wo);
    const int y = (int)((p / Wo) % Ho);
    const size_t b = p / ((size_t)Wo * Ho);
    const uint16_t* s = src + ((b * H + 2 * y) * W + 2 * x) * CP + g * 8;
    const h16x8_t a = *reinterpret_cast<const h16x8_t*>(s);
    const h16x8_t c = *reinterpret_cast<const h16x8_t*>(s + CP);
    const h16x8_t d = *reinterpret_cast<const h16x8_t*>(s + (size_t)W * CP);
    const h16x8_t e = *reinterpret_cast<const h16x8_t*>(s + (size_t)W * CP + CP);
    h16x8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      o[j] = sa::f2h(fmaxf(fmaxf(sa::h2f(a[j]), sa::h2f(c[j])), fmaxf(sa::h2f(d[j]), sa::h2f(e[j]))));
    *reinterpret_cast<h16x8_t*>(dst + p * CP + g * 8) = o;
  }
}

// ------------------------------------------------------------------------------------------------
// UpSampling2D(2): bilinear = half-pixel centres (tf.image.resize, align_corners=False): for output
// index o the source coordinate is (o + 0.5)/2 - 0.5, i.e. weights (0.25, 0.75) with edge clamping.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
upsample2x_kernel(const uint16_t* __restrict__ src, int B, int H, int W, int CP, int bilinear,
                  uint16_t* __restrict__ dst) {
  const int Ho = 2 * H, Wo = 2 * W, groups = CP / 8;
  const size_t total = (size_t)B * Ho * Wo * groups;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(t % groups);
    const size_t p = t / groups;
    const int x = (int)(p % Wo);
    const int y = (int)((p / Wo) % Ho);
    const size_t b = p / ((size_t)Wo * Ho);
    h16x8_t o;
    if (!bilinear) {
      o = *reinterpret_cast<const h16x8_t*>(src + ((b * H + y / 2) * W + x / 2) * CP + g * 8);
    } else {
      int y0, y1, x0, x1;
      float wy, wx;
      sa::up2_taps(y, H, y0, y1, wy);
      sa::up2_taps(x, W, x0, x1, wx);
      const uint16_t* base = src + b * H * W * (size_t)CP + g * 8;
      const h16x8_t a = *reinterpret_cast<const h16x8_t*>(base + ((size_t)y0 * W + x0) * CP);
      const h16x8_t c = *reinterpret_cast<const h16x8_t*>(base + ((size_t)y0 * W + x1) * CP);
      const h16x8_t d = *reinterpret_cast<const h16x8_t*>(base + ((size_t)y1 * W + x0) * CP);
      const h16x8_t e = *reinterpret_cast<const h16x8_t*>(base + ((size_t)y1 * W + x1) * CP);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = sa::f2h(sa::up2_lerp(sa::h2f(a[j]), sa::h2f(c[j]), sa::h2f(d[j]), sa::h2f(e[j]), wy, wx));
    }
    *reinterpret_cast<h16x8_t*>(dst + p * CP + g * 8) = o;
  }
}

// Bilinear x2, one thread per SOURCE pixel and 8-channel group: the 3x3 clamped neighbourhood is loaded once (9 loads
// for 4 outputs instead of 16) and the 2x2 output block is produced with the same operation order as up2_lerp
// (horizontal lerp of the two rows, then vertical), so results are bit-identical to upsample2x_kernel.
__global__ void __launch_bounds__(256)
upsample2x_bilinear_block_kernel(const uint16_t* __restrict__ src, int B, int H, int W, int CP, uint16_t* __restrict__ dst) {
  const int groups = CP / 8;
  const size_t total = (size_t)B * H * W * groups;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(t % groups);
    const size_t p = t / groups;
    const int j = (int)(p % W);
    const int i = (int)((p / W) % H);
    const size_t b = p / ((size_t)W * H);
    const int ys[3] = {max(i - 1, 0), i, min(i + 1, H - 1)};
    const int xs[3] = {max(j - 1, 0), j, min(j + 1, W - 1)};
    const uint16_t* base = src + b * H * W * (size_t)CP + g * 8;
    float hl[3][8], hr[3][8];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const uint16_t* row = base + (size_t)ys[r] * W * CP;
      const h16x8_t vm = *reinterpret_cast<const h16x8_t*>(row + (size_t)xs[0] * CP);
      const h16x8_t v0 = *reinterpret_cast<const h16x8_t*>(row + (size_t)xs[1] * CP);
      const h16x8_t vp = *reinterpret_cast<const h16x8_t*>(row + (size_t)xs[2] * CP);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const float m = sa::h2f(vm[c]), z = sa::h2f(v0[c]), q = sa::h2f(vp[c]);
        hl[r][c] = m + (z - m) * 0.75f;  // output column 2j:   taps (j-1, j), weight 0.75 on j
        hr[r][c] = z + (q - z) * 0.25f;  // output column 2j+1: taps (j, j+1), weight 0.25 on j+1
      }
    }
    h16x8_t o00, o01, o10, o11;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      o00[c] = sa::f2h(hl[0][c] + (hl[1][c] - hl[0][c]) * 0.75f);  // row 2i
      o01[c] = sa::f2h(hr[0][c] + (hr[1][c] - hr[0][c]) * 0.75f);
      o10[c] = sa::f2h(hl[1][c] + (hl[2][c] - hl[1][c]) * 0.25f);  // row 2i+1
      o11[c] = sa::f2h(hr[1][c] + (hr[2][c] - hr[1][c]) * 0.25f);
    }
    uint16_t* o = dst + (((b * 2 * H + 2 * i) * 2 * W) + 2 * j) * (size_t)CP + g * 8;
    *reinterpret_cast<h16x8_t*>(o) = o00;
    *reinterpret_cast<h16x8_t*>(o + CP) = o01;
    *reinterpret_cast<h16x8_t*>(o + (size_t)2 * W * CP) = o10;
    *reinterpret_cast<h16x8_t*>(o + (size_t)2 * W * CP + CP) = o11;
  }
}

// Bilinear x2 for 16-channel tensors (every tensor of an SA_LAYOUT_PLANES16 plan is a stack of those): one thread per SOURCE
// row, OUTPUT column and 8-channel group. With 32 bytes per pixel the block kernel above makes every store instruction write
// 16-byte pieces with gaps (output columns 2j and 2j+1 come from different instructions): measured 0.371 ms where the NHWC
// tensor of the same size took 0.257. Here consecutive lanes write consecutive 16 bytes -- one instruction = 1 KiB of one output
// row; a thread walks K source rows (K + 2 row loads of two columns for 2 K output rows); same operation order, same bits.
template <int K>  // source rows per thread (2 K output rows)
__global__ void __launch_bounds__(256)
upsample2x_bilinear_c16_kernel(const uint16_t* __restrict__ src, int B, int H, int W, uint16_t* __restrict__ dst) {
  const int Wo = 2 * W, HB = (H + K - 1) / K;
  const size_t total = (size_t)B * HB * Wo * 2;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(t & 1);
    const size_t q = t >> 1;
    const int x = (int)(q % Wo);
    const int i0 = (int)((q / Wo) % HB) * K;
    const size_t b = q / ((size_t)Wo * HB);
    const int j = x >> 1, odd = x & 1;
    // output column 2j: taps (j-1, j), weight 0.75 on j; column 2j+1: taps (j, j+1), weight 0.25 on j+1 (edge clamped)
    const int xa = odd ? j : max(j - 1, 0), xb = odd ? min(j + 1, W - 1) : j;
    const float wx = odd ? 0.25f : 0.75f;
    const uint16_t* base = src + b * H * W * (size_t)16 + g * 8;
    float hz[K + 2][8];  // horizontally interpolated source rows i0-1 .. i0+K (clamped)
#pragma unroll
    for (int r = 0; r < K + 2; ++r) {
      const int y = min(max(i0 + r - 1, 0), H - 1);
      const uint16_t* row = base + (size_t)y * W * 16;
      const h16x8_t va = *reinterpret_cast<const h16x8_t*>(row + (size_t)xa * 16);
      const h16x8_t vb = *reinterpret_cast<const h16x8_t*>(row + (size_t)xb * 16);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const float m = sa::h2f(va[c]), z = sa::h2f(vb[c]);
        hz[r][c] = m + (z - m) * wx;
      }
    }
    uint16_t* o = dst + (((b * 2 * H + 2 * i0) * Wo) + x) * (size_t)16 + g * 8;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (i0 + k >= H) break;
      h16x8_t o0, o1;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        o0[c] = sa::f2h(hz[k][c] + (hz[k + 1][c] - hz[k][c]) * 0.75f);      // output row 2 (i0 + k)
        o1[c] = sa::f2h(hz[k + 1][c] + (hz[k + 2][c] - hz[k + 1][c]) * 0.25f);  // output row 2 (i0 + k) + 1
      }
      *reinterpret_cast<h16x8_t*>(o + (size_t)(2 * k) * Wo * 16) = o0;
      *reinterpret_cast<h16x8_t*>(o + (size_t)(2 * k + 1) * Wo * 16) = o1;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// 1x1 linear head (heads.py:42-62): bf16 features x f32 weights -> f32 maps with exact channel count.
// thread = pixel; weights [Cout][CinP] staged in LDS (broadcast reads).
// ------------------------------------------------------------------------------------------------
template <int CO>  // couts handled per pass
__global__ void __launch_bounds__(256)
conv1x1_head_kernel(const uint16_t* __restrict__ src, int CinP, const float* __restrict__ w,
                    const float* __restrict__ bias, int Cout, int act, size_t n_pix, float* __restrict__ dst) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* sw = reinterpret_cast<float*>(smem_raw);  // [CoutR][CinP], CoutR = Cout rounded up to CO
  const int CoutR = (Cout + CO - 1) / CO * CO;
  for (int i = threadIdx.x; i < CoutR * CinP; i += blockDim.x) sw[i] = (i < Cout * CinP) ? w[i] : 0.0f;
  __syncthreads();
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_pix; p += (size_t)gridDim.x * blockDim.x) {
    const uint16_t* s = src + p * CinP;
    for (int c0 = 0; c0 < Cout; c0 += CO) {
      float acc[CO];
#pragma unroll
      for (int j = 0; j < CO; ++j) acc[j] = 0.0f;
      for (int k = 0; k < CinP; k += 8) {
        const h16x8_t v = *reinterpret_cast<const h16x8_t*>(s + k);
        float f[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = sa::h2f(v[i]);
#pragma unroll
        for (int j = 0; j < CO; ++j) {
          const float* wr = sw + (c0 + j) * CinP + k;
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[j] = fmaf(f[i], wr[i], acc[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < CO; ++j)
        if (c0 + j < Cout) {
          float v = acc[j] + bias[c0 + j];
          if (act == 1) v = 1.0f / (1.0f + __expf(-v));
          dst[p * Cout + c0 + j] = v;
        }
    }
  }
}

// The same head as a GEMM on the matrix cores (Cout <= 32, CinP % 16 == 0): A = head weights as hi + lo bf16 fragments
// (16 mantissa bits of the fp32 kernel), built once per workgroup into LDS; B = 32 pixels x 16 channels read straight
// from the NHWC feature tensor (one 16-byte load per lane and k-step, no staging: a pixel's 64-byte line serves two
// consecutive k-steps out of L1); fp32 accumulate; the lane that holds head channel n of pixel p stores it.
// Bandwidth-bound (reads the bf16 features once) instead of Cout x Cin VALU FMAs per pixel.
typedef sa::mfma_h8 head_bf16x8;
typedef __attribute__((ext_vector_type(16))) float head_f32x16;

// NT = 32-channel output tiles per pixel group (1: Cout <= 32; 2: Cout <= 64 -- round 3: the 46-channel PAF head of a 23-edge
// skeleton (BASELINE configs[4]) ran on the VALU kernel at 0.56 TB/s)
template <int NT>
__global__ void __launch_bounds__(256)
conv1x1_head_mfma_kernel(const uint16_t* __restrict__ src, int CinP, const float* __restrict__ w,
                         const float* __restrict__ bias, int Cout, int act, size_t n_pix, float* __restrict__ dst, int hw_planar) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  h16x8_t* frag = reinterpret_cast<h16x8_t*>(smem_raw);  // [K16][NT tiles][2 terms][64 lanes]
  const int K16 = CinP / 16;
  for (int i = threadIdx.x; i < K16 * NT * 64; i += blockDim.x) {
    const int l = i & 63, t = (i >> 6) % NT, k16 = (i >> 6) / NT;
    const int n = t * 32 + (l & 31), k0 = k16 * 16 + (l >> 5) * 8;
    h16x8_t hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = n < Cout ? w[(size_t)n * CinP + k0 + j] : 0.0f;
      hi[j] = sa::f2h(v);
      lo[j] = sa::f2h(v - sa::h2f(hi[j]));
    }
    frag[((k16 * NT + t) * 2 + 0) * 64 + l] = hi;
    frag[((k16 * NT + t) * 2 + 1) * 64 + l] = lo;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, half = lane >> 5, lx = lane & 31;
  const size_t n_groups = (n_pix + 31) / 32;
  const size_t wave0 = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (size_t)gridDim.x * 4;
  for (size_t g = wave0; g < n_groups; g += n_waves) {
    const size_t p = g * 32 + lx;
    const bool ok = p < n_pix;
    // source: NHWC rows of CinP channels, or (hw_planar = pixels per frame > 0) 16-channel planes [B, CinP/16, H*W, 16]
    const size_t pc = ok ? p : 0;
    const uint16_t* s;
    size_t kstep;
    if (hw_planar > 0) {
      const size_t fr = pc / (size_t)hw_planar, px = pc - fr * (size_t)hw_planar;
      s = src + (fr * K16 * (size_t)hw_planar + px) * 16 + half * 8;
      kstep = (size_t)hw_planar * 16;
    } else {
      s = src + pc * CinP + half * 8;
      kstep = 16;
    }
    head_f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
    for (int k16 = 0; k16 < K16; ++k16) {
      const head_bf16x8 b = *reinterpret_cast<const head_bf16x8*>(s + (size_t)k16 * kstep);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const head_bf16x8 a0 = __builtin_bit_cast(head_bf16x8, frag[((k16 * NT + t) * 2 + 0) * 64 + lane]);
        const head_bf16x8 a1 = __builtin_bit_cast(head_bf16x8, frag[((k16 * NT + t) * 2 + 1) * 64 + lane]);
        acc[t] = SA_MFMA_32x32x16(a0, b, acc[t], 0, 0, 0);
        acc[t] = SA_MFMA_32x32x16(a1, b, acc[t], 0, 0, 0);
      }
    }
    if (ok) {
      float* o = dst + p * Cout;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int n = t * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
          if (n < Cout) {
            float v = acc[t][i] + bias[n];
            if (act == 1) v = 1.0f / (1.0f + __expf(-v));
            o[n] = v;
          }
        }
    }
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// Conv2DTranspose(k3, s2, same): out[y][x] = sum_{2i+ky=y, 2j+kx=x} in[i][j] . w[ky][kx], output 2H x 2W
// (full transposed conv cropped at the end). Direct kernel: only the reference's small fixture
// models use transposed convolutions (training profiles ship with up_interpolate=true).
// thread = (output pixel, cout); w [3][3][CoutP][CinP] bf16.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
convt3x3s2_kernel(const uint16_t* __restrict__ src, int CinP, const uint16_t* __restrict__ w,
                  const float* __restrict__ bias, int CoutP, int relu, int B, int H, int W,
                  uint16_t* __restrict__ dst) {
  const int Ho = 2 * H, Wo = 2 * W;
  const size_t total = (size_t)B * Ho * Wo * CoutP;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const int co = (int)(t % CoutP);
    const size_t p = t / CoutP;
    const int x = (int)(p % Wo);
    const int y = (int)((p / Wo) % Ho);
    const size_t b = p / ((size_t)Wo * Ho);
    float acc = bias[co];
    for (int ky = (y & 1); ky < 3; ky += 2) {
      const int i = (y - ky) / 2;
      if (i < 0 || i >= H) continue;
      for (int kx = (x & 1); kx < 3; kx += 2) {
        const int j = (x - kx) / 2;
        if (j < 0 || j >= W) continue;
        const uint16_t* s = src + ((b * H + i) * W + j) * CinP;
        const uint16_t* wr = w + ((size_t)(ky * 3 + kx) * CoutP + co) * CinP;
        for (int k = 0; k < CinP; k += 8) {
          const h16x8_t a = *reinterpret_cast<const h16x8_t*>(s + k);
          const h16x8_t q = *reinterpret_cast<const h16x8_t*>(wr + k);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc = fmaf(sa::h2f(a[e]), sa::h2f(q[e]), acc);
        }
      }
    }
    dst[t] = sa::f2h(relu ? fmaxf(acc, 0.0f) : acc);
  }
}

// tf.image.resize(method="bilinear", antialias=False) with half-pixel centres (resizing.py:97-103):
// src = (dst + 0.5) * in/out - 0.5; taps clamp to the image; value = top + (bottom - top) * ylerp.
// SRC = float: the image as it is; SRC = uint8_t: every tap enters as float(v) * in_scale -- ensure_float's `x * 1/255`
// (normalization.py:49) folded into the resize that follows it (resizing.py:71-105), same float32 operations in the same order
template <typename SRC>
__global__ void __launch_bounds__(256)
resize_bilinear_f32_kernel(const SRC* __restrict__ src, int B, int H, int W, int C, int Ho, int Wo, float in_scale,
                           float* __restrict__ dst) {
  const float sy = (float)H / (float)Ho, sx = (float)W / (float)Wo;
  const size_t total = (size_t)B * Ho * Wo * C;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(t % C);
    size_t p = t / C;
    const int x = (int)(p % Wo);
    p /= Wo;
    const int y = (int)(p % Ho);
    const size_t b = p / Ho;
    const float fy = ((float)y + 0.5f) * sy - 0.5f, fx = ((float)x + 0.5f) * sx - 0.5f;
    const float fly = floorf(fy), flx = floorf(fx);
    const int y0 = max((int)fly, 0), y1 = min((int)ceilf(fy), H - 1);
    const int x0 = max((int)flx, 0), x1 = min((int)ceilf(fx), W - 1);
    const float ly = fy - fly, lx = fx - flx;
    const SRC* im = src + b * H * W * C + c;
    auto tap = [&](int yy, int xx) {
      const SRC v = im[((size_t)yy * W + xx) * C];
      if constexpr (sizeof(SRC) == 1)
        return __builtin_fmaf((float)v, in_scale, 0.0f);  // the product rounded on its own: as an FMA it cannot be contracted into
                                                           // the interpolation's FMAs (a plain `*`, also __fmul_rn, can)
      else
        return (float)v;
    };
    const float tl = tap(y0, x0), tr = tap(y0, x1);
    const float bl = tap(y1, x0), br = tap(y1, x1);
    const float top = tl + (tr - tl) * lx, bot = bl + (br - bl) * lx;
    dst[t] = top + (bot - top) * ly;
  }
}

__global__ void f32_to_bf16_padded_kernel(const float* __restrict__ src, size_t n_pix, int C, int CP,
                                          uint16_t* __restrict__ dst) {
  const size_t total = n_pix * CP;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(t % CP);
    const size_t p = t / CP;
    dst[t] = (c < C) ? sa::f2h(src[p * C + c]) : (uint16_t)0;
  }
}

__global__ void bf16_to_f32_kernel(const uint16_t* __restrict__ src, size_t n_pix, int CP, int C,
                                   float* __restrict__ dst) {
  const size_t total = n_pix * C;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(t % C);
    const size_t p = t / C;
    dst[t] = sa::h2f(src[p * CP + c]);
  }
}


// max |x| over a tensor of the 16-bit storage type (or float32) + non-finite flags: the range scan of fp16 storage
// (DeviceNetwork._check_fp16_range / layer_ranges). HBM-bound: 16-byte loads, the magnitude bits of a non-negative IEEE value
// order like the value, so the reduction runs on integers; one atomicMax / atomicOr per wavefront.
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
template <bool F32>
__global__ __launch_bounds__(256) void tensor_absmax_kernel(const u32x4_t* __restrict__ x, size_t n_vec, const void* __restrict__ tail,
                                                            int n_tail, uint32_t* __restrict__ out) {
#if defined(SA_HALF_FP16)
  constexpr uint32_t HINF = 0x7c00u;
#else
  constexpr uint32_t HINF = 0x7f80u;
#endif
  uint32_t m = 0, flags = 0;  // m: largest finite magnitude (bits), flags: 1 = inf seen, 2 = NaN seen
  auto take = [&](uint32_t a, uint32_t inf) {
    if (a < inf) m = max(m, a);
    else flags |= (a == inf) ? 1u : 2u;
  };
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * blockDim.x) {
    const u32x4_t v = __builtin_nontemporal_load(x + i);
    const uint32_t w[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (F32) {
        take(w[j] & 0x7fffffffu, 0x7f800000u);
      } else {
        take(w[j] & 0x7fffu, HINF);
        take((w[j] >> 16) & 0x7fffu, HINF);
      }
    }
  }
  if (blockIdx.x == 0 && (int)threadIdx.x < n_tail) {
    if (F32) take(((const uint32_t*)tail)[threadIdx.x] & 0x7fffffffu, 0x7f800000u);
    else take(((const uint16_t*)tail)[threadIdx.x] & 0x7fffu, HINF);
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) {
    m = max(m, (uint32_t)__shfl_xor((int)m, o));
    flags |= (uint32_t)__shfl_xor((int)flags, o);
  }
  if ((threadIdx.x & 63) == 0) {
    if (m) atomicMax(out, F32 ? m : __float_as_uint(sa::h2f((uint16_t)m)));  // float bits of non-negative values order like the values
    if (flags) atomicOr(out + 1, flags);
  }
}

inline int grid_for(size_t total, int block = 256, int cap = 256 * 16) {
  size_t g = (total + block - 1) / block;
  if (g > (size_t)cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" {

int sa_stem_conv3x3(const void* src, int src_is_u8, int B, int H, int W, int Cin, const float* w,
                    const float* bias, int CoutP, int relu, void* dst, sa_stream_t stream) {
  SA_REQUIRE(Cin == 1 || Cin == 3, "sa_stem_conv3x3: Cin must be 1 or 3, got %d", Cin);
  SA_REQUIRE(CoutP > 0 && CoutP % 8 == 0 && CoutP <= 256, "sa_stem_conv3x3: CoutP must be a multiple of 8 (<=256)");
  SA_REQUIRE(B > 0 && H > 0 && W > 0, "sa_stem_conv3x3: bad shape");
  const size_t total = (size_t)B * H * W * (CoutP / 8);
  const size_t lds = sizeof(float) * (size_t)(9 * Cin + 1) * CoutP;
  const dim3 g(grid_for(total)), blk(256);
  hipStream_t st = (hipStream_t)stream;
  uint16_t* d = (uint16_t*)dst;
#define SA_STEM(CIN, U8) \
  hipLaunchKernelGGL((stem_conv3x3_kernel<CIN, U8>), g, blk, lds, st, src, B, H, W, w, bias, CoutP, relu, d)
  if (Cin == 1 && src_is_u8) SA_STEM(1, true);
  else if (Cin == 1) SA_STEM(1, false);
  else if (src_is_u8) SA_STEM(3, true);
  else SA_STEM(3, false);
#undef SA_STEM
  SA_LAUNCH_CHECK();
  return SA_OK;
}

int sa_image_conv_bf16(const void* src, int src_is_u8, int B, int H, int W, int Cin, int CinW, const float* in_affine,
                       int kh, int kw, int stride, int pad_top, int pad_left, int Ho, int Wo, const float* w, const float* bias, int CoutP, int relu,
                       const float* post_scale, const float* post_shift, void* dst, sa_stream_t stream) {
  SA_REQUIRE(src && w && bias && dst, "sa_image_conv_bf16: NULL pointer");
  SA_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && kh > 0 && kw > 0 && stride > 0 && Ho > 0 && Wo > 0, "sa_image_conv_bf16: bad shape");
  SA_REQUIRE(CoutP % 8 == 0, "sa_image_conv_bf16: CoutP must be a multiple of 8");
  const int planar = (relu & SA_LAYOUT_PLANES16) ? 1 : 0;  // (`relu`: bit 0 = ReLU, SA_LAYOUT_PLANES16 = write 16-channel planes)
  relu &= 1;
  SA_REQUIRE(!planar || CoutP % 16 == 0, "sa_image_conv_bf16: SA_LAYOUT_PLANES16 needs CoutP % 16 == 0");
  SA_REQUIRE(CinW == Cin || Cin == 1, "sa_image_conv_bf16: CinW != Cin needs a single-channel image (tiled)");
  SA_REQUIRE(!post_scale == !post_shift, "sa_image_conv_bf16: post_scale and post_shift come together");
  const size_t lds = sizeof(float) * (size_t)kh * kw * CinW * CoutP;
  SA_REQUIRE(lds <= 64 * 1024, "sa_image_conv_bf16: weights (%zu B) exceed the LDS budget", lds);
  const size_t total = (size_t)B * Ho * Wo * (CoutP / 8);
  hipLaunchKernelGGL(image_conv_kernel, dim3(grid_for(total)), dim3(256), lds, (hipStream_t)stream, src, src_is_u8, B, H,
                     W, Cin, CinW, in_affine, kh, kw, stride, pad_top, pad_left, Ho, Wo, w, bias, CoutP, relu, post_scale, post_shift,
                     (uint16_t*)dst, planar);
  SA_LAUNCH_CHECK();
  return SA_OK;
}

int sa_maxpool_bf16(const void* src, int B, int H, int W, int CP, int k, int stride, int pad_top, int pad_left,
                    int pad_is_zero, int Ho, int Wo, void* dst, sa_stream_t stream) {
  SA_REQUIRE(src && dst && CP % 8 == 0 && B > 0 && k > 0 && stride > 0 && Ho > 0 && Wo > 0, "sa_maxpool_bf16: bad arguments");
  SA_REQUIRE((Ho - 1) * stride - pad_top < H && (Wo - 1) * stride - pad_left < W, "sa_maxpool_bf16: window outside the image");
  // SA_POOL_TILED=0: the general kernel for every shape (A/B)
  static const bool tiled = [] {
    const char* v = getenv("SA_POOL_TILED");
    return !v || atoi(v) != 0;
  }();
  if (tiled && CP == 16 && k == 3 && stride == 2 && B <= 65535 && (Ho + 3) / 4 <= 65535) {
    constexpr int TO = 4;
    hipLaunchKernelGGL(maxpool3x3s2_c16_kernel<TO>, dim3((unsigned)((Wo + 31) / 32), (unsigned)((Ho + TO - 1) / TO), (unsigned)B),
                       dim3(64 * TO), 0, (hipStream_t)stream, (const uint16_t*)src, H, W, pad_top, pad_left, pad_is_zero, Ho, Wo,
                       (uint16_t*)dst);
    SA_LAUNCH_CHECK();
    return SA_OK;
  }
  const size_t total = (size_t)B * Ho * Wo * (CP / 8);
  hipLaunchKernelGGL(maxpool_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)src, B, H, W,
                     CP, k, stride, pad_top, pad_left, pad_is_zero, Ho, Wo, (uint16_t*)dst);
  SA_LAUNCH_CHECK();
  return SA_OK;
}

int sa_add_bf16(const void* a, const void* b, int B, int H, int W, int CP, int b_half_res, int relu, void* dst,
                sa_stream_t stream) {
  SA_REQUIRE(a && b && dst && CP % 8 == 0, "sa_add_bf16: bad arguments");
  SA_REQUIRE(!b_half_res || (H % 2 == 0 && W % 2 == 0), "sa_add_bf16: half-resolution operand needs even H, W");
  const size_t total = (size_t)B * H * W * (CP / 8);
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)a,
                     (const uint16_t*)b, B, H, W, CP, b_half_res, relu, (uint16_t*)dst);
  SA_LAUNCH_CHECK();
  return SA_OK;
}

int sa_maxpool2x2_bf16(const void* src, int B, int H, int W, int CP, void* dst, sa_stream_t stream) {
  SA_REQUIRE(H % 2 == 0 && W % 2 == 0 && CP % 8 == 0, "sa_maxpool2x2_bf16: needs even H,W and CP%%8==0");
  const size_t total = (size_t)B * (H / 2) * (W / 2) * (CP / 8);
  hipLaunchKernelGGL(maxpool2x2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)src, B, H, W, CP, (uint16_t*)dst);
  SA_LAUNCH_CHECK();
  return SA_OK;
}

int sa_upsample2x_bf16(const void* src, int B, int H, int W, int CP, int bilinear, void* dst,
                       sa_stream_t stream) {
  SA_REQUIRE(CP % 8 == 0, "sa_upsample2x_bf16: CP%%8 != 0");
  if (bilinear && CP == 16) {
    // two source rows per thread: 0.494 ms (one row) -> 0.404 (two) / 0.43 (four) over the three upsamplings of a benchmark step
    // (64 frames, profiles/r02_ab_session.md); the NHWC block kernel below needs 0.465 for the same tensors
    constexpr int K = 2;
    const size_t total = (size_t)B * ((H + K - 1) / K) * W * 4;
    hipLaunchKernelGGL(upsample2x_bilinear_c16_kernel<K>, dim3(grid_for(total, 256, 1 << 20)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)src, B, H, W, (uint16_t*)dst);
  } else if (bilinear) {
    const size_t total = (size_t)B * H * W * (CP / 8);
    // one workgroup per 256 items up to 65536 workgroups: measured 4.0-4.3 TB/s vs 3.7 with an 8192-block grid-stride
    // loop (torch's copy kernel: 5.0 TB/s read+write on the same box, tools/bw_probe.py); nontemporal stores: no effect
    hipLaunchKernelGGL(upsample2x_bilinear_block_kernel, dim3(grid_for(total, 256, 65536)), dim3(256), 0,
                       (hipStream_t)stream, (const uint16_t*)src, B, H, W, CP, (uint16_t*)dst);
  } else {
    const size_t total = (size_t)B * 4 * H * W * (CP / 8);
    hipLaunchKernelGGL(upsample2x_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)src, B, H, W, CP, bilinear, (uint16_t*)dst);
  }
  SA_LAUNCH_CHECK();
  return SA_OK;
}

int sa_conv1x1_head(const void* src, int CinP, const float* w, const float* bias, int Cout, int act,
                    int B, int H, int W, float* dst, sa_stream_t stream) {
  SA_REQUIRE(CinP % 8 == 0 && Cout > 0, "sa_conv1x1_head: CinP%%8 != 0 or Cout <= 0");
  const int planar = (act & SA_LAYOUT_PLANES16) ? 1 : 0;  // (`act`: bit 0 = sigmoid, SA_LAYOUT_PLANES16 = src in 16-channel planes)
  act &= ~SA_LAYOUT_PLANES16;
  SA_REQUIRE(act == 0 || act == 1, "sa_conv1x1_head: act must be 0 (linear) or 1 (sigmoid)");
  const size_t n_pix_all = (size_t)B * H * W;
  const int hw_planar = planar ? H * W : 0;
  const int nt = Cout <= 32 ? 1 : 2;
  if (Cout <= 64 && CinP % 16 == 0 && (size_t)CinP / 16 * 2048 * nt <= 64 * 1024) {
    const size_t lds_m = (size_t)CinP / 16 * 2048 * nt;
    const size_t groups = (n_pix_all + 31) / 32;
    const int grid = (int)std::min<size_t>((groups + 3) / 4, 256 * 8);
    if (nt == 1)
      hipLaunchKernelGGL(conv1x1_head_mfma_kernel<1>, dim3(grid), dim3(256), lds_m, (hipStream_t)stream, (const uint16_t*)src,
                         CinP, w, bias, Cout, act, n_pix_all, dst, hw_planar);
    else
      hipLaunchKernelGGL(conv1x1_head_mfma_kernel<2>, dim3(grid), dim3(256), lds_m, (hipStream_t)stream, (const uint16_t*)src,
                         CinP, w, bias, Cout, act, n_pix_all, dst, hw_planar);
    SA_LAUNCH_CHECK();
    return SA_OK;
  }
  SA_REQUIRE(!planar, "sa_conv1x1_head: SA_LAYOUT_PLANES16 sources need the matrix-core kernel (Cout <= 64, CinP % 16 == 0)");
  constexpr int CO = 8;
  const int CoutR = (Cout + CO - 1) / CO * CO;
  const size_t lds = sizeof(float) * (size_t)CoutR * CinP;
  SA_REQUIRE(lds <= 64 * 1024, "sa_conv1x1_head: weights (%zu B) exceed the LDS budget", lds);
  const size_t n_pix = (size_t)B * H * W;
  hipLaunchKernelGGL((conv1x1_head_kernel<CO>), dim3(grid_for(n_pix)), dim3(256), lds, (hipStream_t)stream,
                     (const uint16_t*)src, CinP, w, bias, Cout, act, n_pix, dst);
  SA_LAUNCH_CHECK();
  return SA_OK;
}

int sa_convt3x3s2_bf16(const void* src, int CinP, const void* w, const float* bias, int CoutP,
                       int relu, int B, int H, int W, void* dst, sa_stream_t stream) {
  SA_REQUIRE(CinP % 8 == 0 && CoutP % 8 == 0, "sa_convt3x3s2_bf16: channel padding");
  const size_t total = (size_t)B * 4 * H * W * CoutP;
  hipLaunchKernelGGL(convt3x3s2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)src, CinP, (const uint16_t*)w, bias, CoutP, relu, B, H, W, (uint16_t*)dst);
  SA_LAUNCH_CHECK();
  return SA_OK;
}

int sa_resize_bilinear_f32(const float* src, int B, int H, int W, int C, int Ho, int Wo, float* dst,
                           sa_stream_t stream) {
  SA_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0, "sa_resize_bilinear_f32: bad shape");
  hipLaunchKernelGGL(resize_bilinear_f32_kernel<float>, dim3(grid_for((size_t)B * Ho * Wo * C)), dim3(256), 0,
                     (hipStream_t)stream, src, B, H, W, C, Ho, Wo, 1.0f, dst);
  SA_LAUNCH_CHECK();
  return SA_OK;
}

int sa_resize_bilinear_u8_f32(const void* src, int B, int H, int W, int C, int Ho, int Wo, float in_scale, float* dst,
                              sa_stream_t stream) {
  SA_REQUIRE(src && dst && B > 0 && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0, "sa_resize_bilinear_u8_f32: bad arguments");
  hipLaunchKernelGGL(resize_bilinear_f32_kernel<uint8_t>, dim3(grid_for((size_t)B * Ho * Wo * C)), dim3(256), 0,
                     (hipStream_t)stream, static_cast<const uint8_t*>(src), B, H, W, C, Ho, Wo, in_scale, dst);
  SA_LAUNCH_CHECK();
  return SA_OK;
}

int sa_f32_to_bf16_padded(const float* src, int n_pix, int C, int CP, void* dst, sa_stream_t stream) {
  SA_REQUIRE(CP >= C && C > 0, "sa_f32_to_bf16_padded: CP < C");
  hipLaunchKernelGGL(f32_to_bf16_padded_kernel, dim3(grid_for((size_t)n_pix * CP)), dim3(256), 0,
                     (hipStream_t)stream, src, (size_t)n_pix, C, CP, (uint16_t*)dst);
  SA_LAUNCH_CHECK();
  return SA_OK;
}

int sa_tensor_absmax(const void* x, size_t n, int is_f32, float* out2, sa_stream_t stream) {
  SA_REQUIRE(x && out2 && n > 0, "sa_tensor_absmax: bad arguments");
  SA_REQUIRE(((uintptr_t)x & 15) == 0, "sa_tensor_absmax: the tensor must be 16-byte aligned");
  SA_HIP_CHECK(hipMemsetAsync(out2, 0, 2 * sizeof(float), (hipStream_t)stream));
  const size_t per = is_f32 ? 4 : 8, n_vec = n / per;
  const int n_tail = (int)(n - n_vec * per);
  const void* tail = (const char*)x + n_vec * 16;
  const int grid = (int)std::min<size_t>(std::max<size_t>((n_vec + 255) / 256, 1), 256 * 8);
  if (is_f32)
    hipLaunchKernelGGL(tensor_absmax_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const u32x4_t*)x, n_vec, tail,
                       n_tail, (uint32_t*)out2);
  else
    hipLaunchKernelGGL(tensor_absmax_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const u32x4_t*)x, n_vec, tail,
                       n_tail, (uint32_t*)out2);
  SA_LAUNCH_CHECK();
  return SA_OK;
}

int sa_bf16_to_f32(const void* src, int n_pix, int CP, int C, float* dst, sa_stream_t stream) {
  SA_REQUIRE(CP >= C && C > 0, "sa_bf16_to_f32: CP < C");
  hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(grid_for((size_t)n_pix * C)), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)src, (size_t)n_pix, CP, C, dst);
  SA_LAUNCH_CHECK();
  return SA_OK;
}

}  // extern "C"
