// Sparse pyramidal Lucas-Kanade optical flow on the device -- the engine of SLEAP's flow tracker:
//   FlowCandidateMaker.flow_shift_instances (sleap/nn/tracking.py:258-356) calls
//   cv2.calcOpticalFlowPyrLK(ref_img, new_img, pts, None, winSize=(w, w), maxLevel=L, criteria=(EPS | COUNT, 30, 0.01)).
// OpenCV is a third-party dependency of the reference (pypi_requirements.txt:16) and is absent here; what is implemented is the
// published algorithm (Bouguet 2000) with OpenCV 4.x's integer conventions, as restated in oracle/optical_flow.py, which is
// the checker for this file (tests/test_gpu_flow.py):
//   pyramid   level 0 = the uint8 frame (3 channels -> gray by OpenCV's fixed-point BGR2GRAY weights), level l+1 = pyrDown
//             ([1 4 6 4 1]^2 / 256, integer, BORDER_REFLECT_101), levels end before an image <= the window;
//   Scharr    Ix = [3 10 3]^T (x) [-1 0 1], Iy = [-1 0 1]^T (x) [3 10 3], int16, reflect-101 inside, ZERO outside a level;
//   tracker   per point, coarse to fine: 14-bit integer bilinear patches of I, Ix, Iy (window w x w), 2x2 system in float32,
//             <= 30 Newton steps, OpenCV's stopping rules, status / error outputs.
// One WAVEFRONT per point: the w*w window pixels are spread over the 64 lanes (7 per lane for w = 21), a lane keeps its share of the
// I / Ix / Iy patch in registers, the part of J a level's iteration can touch is staged in LDS, every sum over the window is a
// butterfly reduction (all lanes end with the same bits, so the data-dependent control flow stays wave uniform). Points are independent; a frame pair of the tracker is a few hundred of them.
// Differences to a CPU OpenCV: the order of the float32 window sums (OpenCV's own SIMD and scalar builds differ in that too).
#include <cstdint>
#include <cstdlib>
#include <vector>

#include "sa_common.h"

namespace {

constexpr int MAX_LEVELS = 8;
constexpr int MAX_WIN = 31;

struct Level {
  int h, w;
  size_t img_off, deriv_off;  // bytes from the pyramid base: uint8 [h][w]; int16 [h][w][2]
};

struct PyrLayout {
  int n_levels;
  Level lv[MAX_LEVELS];
  size_t bytes;
};

size_t align256(size_t v) { return (v + 255) & ~size_t(255); }

PyrLayout pyr_layout(int H, int W, int win, int max_level) {
  PyrLayout p;
  p.n_levels = 0;
  size_t off = 0;
  int h = H, w = W;
  for (int l = 0; l <= max_level && l < MAX_LEVELS; ++l) {
    if (l > 0) {
      h = (h + 1) / 2;
      w = (w + 1) / 2;
      if (w <= win || h <= win) break;
    }
    Level& L = p.lv[p.n_levels++];
    L.h = h;
    L.w = w;
    L.img_off = off;
    off += align256((size_t)h * w);
    L.deriv_off = off;
    off += align256((size_t)h * w * 4);
  }
  p.bytes = off;
  return p;
}

// BORDER_REFLECT_101 index. The common cases -- inside, or one reflection -- cost a few compares; the general form (an integer
// modulo, ~40 instructions) is only reached by windows larger than the level.
__device__ __forceinline__ int reflect101(int i, int n) {
  if ((unsigned)i < (unsigned)n) return i;
  if (i < 0 && -i < n) return -i;
  if (i >= n && i <= 2 * n - 2) return 2 * n - 2 - i;
  if (n == 1) return 0;
  const int p = 2 * (n - 1);
  i %= p;
  if (i < 0) i += p;
  return i >= n ? p - i : i;
}

// Where a pyramid kernel reads and writes: one frame given by pointers (frames == nullptr), or frame blockIdx.z of a batch whose
// pyramid buffers are listed in `pyr` (offsets inside a buffer are the same for every frame).
struct PyrIO {
  const uint8_t* src;
  uint8_t* dst;
  uint8_t* const* pyr;
  size_t src_off, dst_off;
  __device__ __forceinline__ const uint8_t* in() const { return pyr ? pyr[blockIdx.z] + src_off : src; }
  __device__ __forceinline__ uint8_t* out() const { return pyr ? pyr[blockIdx.z] + dst_off : dst; }
};

// level 0: the frame as gray uint8. `frames` of a batch are contiguous ([F][n_pix][C]); 16 bytes per thread when C == 1
__global__ void __launch_bounds__(256)
flow_gray_kernel(const uint8_t* __restrict__ src, int n_pix, int C, PyrIO io) {
  const uint8_t* s0 = src + (size_t)blockIdx.z * n_pix * C;
  uint8_t* dst = io.out();
  const int t0 = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
  if (C == 1) {
    if ((n_pix & 15) == 0 && (((uintptr_t)s0 | (uintptr_t)dst) & 15) == 0) {
      const uint4* s4 = reinterpret_cast<const uint4*>(s0);
      uint4* d4 = reinterpret_cast<uint4*>(dst);
      for (int i = t0; i < (n_pix >> 4); i += nt) d4[i] = s4[i];
    } else {
      for (int i = t0; i < n_pix; i += nt) dst[i] = s0[i];
    }
  } else {  // cv2.COLOR_BGR2GRAY on the caller's channel order: (c0 * 1868 + c1 * 9617 + c2 * 4899 + 8192) >> 14
    for (int i = t0; i < n_pix; i += nt) {
      const uint8_t* s = s0 + (size_t)i * 3;
      dst[i] = (uint8_t)((s[0] * 1868 + s[1] * 9617 + s[2] * 4899 + 8192) >> 14);
    }
  }
}

// level 0 for img_scale != 1: gray conversion, then cv2.resize(img, None, None, fx, fy) with INTER_LINEAR on uint8 (tracking.py:
// 311-314) -- OpenCV's 11-bit fixed-point form (imgproc/resize.cpp, restated in oracle/optical_flow.py cv_resize_linear_u8):
// x = float((dx + 0.5) / fx - 0.5), sx = floor(x), a = x - sx, clamped with a = 0 at both ends; coefficients
// cvRound((1 - a) 2048), cvRound(a 2048); rows likewise (indices clipped, coefficient kept); horizontal pass in int32,
// vertical pass (((b0 (D0 >> 4)) >> 16) + ((b1 (D1 >> 4)) >> 16) + 2) >> 2. One thread per destination pixel.
__device__ __forceinline__ int gray_at(const uint8_t* s0, int C, size_t i) {
  if (C == 1) return s0[i];
  const uint8_t* s = s0 + i * 3;
  return (s[0] * 1868 + s[1] * 9617 + s[2] * 4899 + 8192) >> 14;
}

__global__ void __launch_bounds__(256)
flow_gray_resize_kernel(const uint8_t* __restrict__ src, int H, int W, int C, int Hs, int Ws, double scale_x, double scale_y,
                        PyrIO io) {
  const int dx = blockIdx.x * 64 + (threadIdx.x & 63), dy = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (dx >= Ws || dy >= Hs) return;
  const uint8_t* s0 = src + (size_t)blockIdx.z * H * W * C;
  const float x = (float)(((double)dx + 0.5) * scale_x - 0.5), y = (float)(((double)dy + 0.5) * scale_y - 0.5);
  int sx = (int)floorf(x), sy = (int)floorf(y);
  float a = x - (float)sx;
  const float b = y - (float)sy;
  if (sx < 0) sx = 0, a = 0.0f;
  if (sx >= W - 1) sx = W - 1, a = 0.0f;
  int a0 = (int)rintf((1.0f - a) * 2048.0f), a1 = (int)rintf(a * 2048.0f);
  const int b0 = (int)rintf((1.0f - b) * 2048.0f), b1 = (int)rintf(b * 2048.0f);
  const int sx1 = min(sx + 1, W - 1);
  if (sx + 1 >= W) a0 = 2048, a1 = 0;
  const int r0 = min(max(sy, 0), H - 1), r1 = min(max(sy + 1, 0), H - 1);
  const int D0 = gray_at(s0, C, (size_t)r0 * W + sx) * a0 + gray_at(s0, C, (size_t)r0 * W + sx1) * a1;
  const int D1 = gray_at(s0, C, (size_t)r1 * W + sx) * a0 + gray_at(s0, C, (size_t)r1 * W + sx1) * a1;
  const int v = (((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2;
  io.out()[(size_t)dy * Ws + dx] = (uint8_t)min(max(v, 0), 255);
}

// The two stencils run on a 2-D grid: 64 x 4 pixels per workgroup, blockIdx = (column block, row block, frame) -- no index
// division, and a thread's reflected row / column indices are computed once (first version: a flat index with a division and
// 10 modulo-based reflections per output; the pyramids of a 64-frame batch cost as much as its Lucas-Kanade launch).
__global__ void __launch_bounds__(256)
flow_pyrdown_kernel(PyrIO io, int h, int w, int oh, int ow) {
  const int ox = blockIdx.x * 64 + (threadIdx.x & 63), oy = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (ox >= ow || oy >= oh) return;
  const uint8_t* src = io.in();
  int xs[5];
#pragma unroll
  for (int dx = 0; dx < 5; ++dx) xs[dx] = reflect101(2 * ox + dx - 2, w);
  const int k[5] = {1, 4, 6, 4, 1};
  int acc = 0;
#pragma unroll
  for (int dy = 0; dy < 5; ++dy) {
    const uint8_t* row = src + (size_t)reflect101(2 * oy + dy - 2, h) * w;
    int r = 0;
#pragma unroll
    for (int dx = 0; dx < 5; ++dx) r += k[dx] * row[xs[dx]];
    acc += k[dy] * r;
  }
  io.out()[(size_t)oy * ow + ox] = (uint8_t)((acc + 128) >> 8);
}

__global__ void __launch_bounds__(256)
flow_scharr_kernel(PyrIO io, int h, int w) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  const uint8_t* src = io.in();
  int* dst = reinterpret_cast<int*>(io.out());  // (int16 Ix, int16 Iy) per pixel, one 4-byte store
  const uint8_t* r0 = src + (size_t)reflect101(y - 1, h) * w;
  const uint8_t* r1 = src + (size_t)y * w;
  const uint8_t* r2 = src + (size_t)reflect101(y + 1, h) * w;
  const int xl = reflect101(x - 1, w), xr = reflect101(x + 1, w);
  // vertical smoothing / difference at columns x-1, x, x+1
  const int t0l = (r0[xl] + r2[xl]) * 3 + r1[xl] * 10, t0r = (r0[xr] + r2[xr]) * 3 + r1[xr] * 10;
  const int t1l = r2[xl] - r0[xl], t1c = r2[x] - r0[x], t1r = r2[xr] - r0[xr];
  const int ix = t0r - t0l, iy = (t1r + t1l) * 3 + t1c * 10;  // both fit int16: |.| <= 16 * 255
  dst[(size_t)y * w + x] = (int)(((unsigned)ix & 0xFFFFu) | ((unsigned)iy << 16));
}

// ---- four pixels per thread (levels whose rows are dword aligned: every level of a 1024 x 1024 frame). The stencils are
// instruction bound with one pixel per thread (25 byte loads per pyrDown output); here a thread loads aligned dwords and
// writes its four results with one store. Same integer arithmetic, same results.
__device__ __forceinline__ int byte_of(unsigned v, int i) { return (int)((v >> (8 * i)) & 0xFFu); }

// w % 8 == 0 (so ow = w / 2 is a multiple of 4); thread -> outputs ox0 .. ox0+3, source columns 2 ox0 - 2 .. 2 ox0 + 8
__global__ void __launch_bounds__(256)
flow_pyrdown_x4_kernel(PyrIO io, int h, int w, int oh, int ow) {
  const int ox0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, oy = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (ox0 >= ow || oy >= oh) return;
  const uint8_t* src = io.in();
  const int sx = 2 * ox0;
  const bool left = sx == 0, right = sx + 8 >= w;  // column -2, -1 reflect to 2, 1; column w reflects to w - 2
  const int k[5] = {1, 4, 6, 4, 1};
  int acc[4] = {0, 0, 0, 0};
#pragma unroll
  for (int dy = 0; dy < 5; ++dy) {
    const unsigned* row = reinterpret_cast<const unsigned*>(src + (size_t)reflect101(2 * oy + dy - 2, h) * w + sx);
    const unsigned v0 = row[0], v1 = row[1];
    int c[11];
    if (left) {
      c[0] = byte_of(v0, 2);
      c[1] = byte_of(v0, 1);
    } else {
      const unsigned vm = row[-1];
      c[0] = byte_of(vm, 2);
      c[1] = byte_of(vm, 3);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      c[2 + i] = byte_of(v0, i);
      c[6 + i] = byte_of(v1, i);
    }
    c[10] = right ? byte_of(v1, 2) : byte_of(row[2], 0);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      acc[j] += k[dy] * (c[2 * j] + 4 * c[2 * j + 1] + 6 * c[2 * j + 2] + 4 * c[2 * j + 3] + c[2 * j + 4]);
  }
  unsigned out = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) out |= (unsigned)((acc[j] + 128) >> 8) << (8 * j);
  *reinterpret_cast<unsigned*>(io.out() + (size_t)oy * ow + ox0) = out;
}

// w % 4 == 0; thread -> pixels x0 .. x0+3 of row y
__global__ void __launch_bounds__(256)
flow_scharr_x4_kernel(PyrIO io, int h, int w) {
  const int x0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x0 >= w || y >= h) return;
  const uint8_t* src = io.in();
  const uint8_t* rows[3] = {src + (size_t)reflect101(y - 1, h) * w, src + (size_t)y * w, src + (size_t)reflect101(y + 1, h) * w};
  const int xl = x0 == 0 ? 1 : x0 - 1, xr = x0 + 4 >= w ? w - 2 : x0 + 4;  // reflect-101 of x0 - 1 and x0 + 4
  int c[3][6];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const unsigned v = *reinterpret_cast<const unsigned*>(rows[r] + x0);
    c[r][0] = rows[r][xl];
#pragma unroll
    for (int i = 0; i < 4; ++i) c[r][1 + i] = byte_of(v, i);
    c[r][5] = rows[r][xr];
  }
  int sm[6], df[6];  // vertical smoothing [3 10 3] and difference [-1 0 1] per column
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    sm[i] = (c[0][i] + c[2][i]) * 3 + c[1][i] * 10;
    df[i] = c[2][i] - c[0][i];
  }
  unsigned o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ix = sm[j + 2] - sm[j], iy = (df[j + 2] + df[j]) * 3 + df[j + 1] * 10;
    o[j] = ((unsigned)ix & 0xFFFFu) | ((unsigned)iy << 16);
  }
  *reinterpret_cast<uint4*>(reinterpret_cast<int*>(io.out()) + (size_t)y * w + x0) = make_uint4(o[0], o[1], o[2], o[3]);
}

struct LkParams {
  const uint8_t* const* pyr_prev;    // [n] device pointers: the reference frame's pyramid of every point
  const uint8_t* pyr_next;           // the target frame's pyramid (all points), or
  const uint8_t* const* pyr_next_v;  // [n] device pointers: one target pyramid per point (sa_flow_lk_pairs); NULL otherwise
  PyrLayout lay;
  int win, n, max_count;
  float eps2;
  const float* prev_pts;  // [n][2]
  float* next_pts;        // [n][2]
  uint8_t* status;        // [n]
  float* err;             // [n]
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s);
  return v;
}

__device__ __forceinline__ int cv_floor(float v) {
  // cvFloor of a non-finite value is INT_MIN on the reference's platform (cvtsd2si): such points are "outside the image"
  if (!(v == v) || v > 2.0e9f || v < -2.0e9f) return INT32_MIN;
  return (int)floorf(v);
}

__device__ __forceinline__ void lk_weights(float a, float b, int& w00, int& w01, int& w10, int& w11) {
  const float s = 16384.0f;  // 1 << W_BITS
  w00 = (int)rintf((1.0f - a) * (1.0f - b) * s);
  w01 = (int)rintf(a * (1.0f - b) * s);
  w10 = (int)rintf((1.0f - a) * b * s);
  w11 = 16384 - w00 - w01 - w10;
}

// one wave per point; blockDim = 256 (4 points). NPL = window pixels per lane (ceil(win^2 / 64)): the lane's share of the I / Ix /
// Iy patch lives in REGISTERS (a lane only ever reads its own pixels), and the part of J the iteration can touch -- the window
// box plus a margin of STAGE_MARGIN pixels around the level's starting guess -- is staged in LDS once per level (already
// reflected), so the <= 30 dependent Newton steps of a level read LDS instead of global memory; a step that leaves the staged box
// re-stages it around the current position. (First version: patches in LDS, J from global memory inside a rolled loop: 0.447 ms per
// launch however few the points -- 7 serialised global round trips per step; profiles/r02_flow_tracker.md.)
constexpr int STAGE_MARGIN = 4;

template <int NPL>
__global__ void __launch_bounds__(256)
flow_lk_kernel(const LkParams p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_u8[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = blockIdx.x * 4 + wave;
  if (k >= p.n) return;  // wave uniform
  const int win = p.win, area = win * win;
  const int S = win + 1 + 2 * STAGE_MARGIN;  // staged box edge
  uint8_t* stage = lds_u8 + (size_t)wave * S * S;
  const int lane_y = lane / S, lane_x = lane - lane_y * S, step_y = 64 / S, step_x = 64 - step_y * S;
  int wyk[NPL], wxk[NPL];
  bool act[NPL];
#pragma unroll
  for (int q = 0; q < NPL; ++q) {
    const int i = lane + 64 * q;
    act[q] = i < area;
    const int ic = act[q] ? i : 0;
    wyk[q] = ic / win;
    wxk[q] = ic - wyk[q] * win;
  }
  const uint8_t* base_i = p.pyr_prev[k];
  const uint8_t* base_j = p.pyr_next_v ? p.pyr_next_v[k] : p.pyr_next;
  const float half = (float)(win - 1) * 0.5f;
  const float px = p.prev_pts[2 * k], py = p.prev_pts[2 * k + 1];
  float nx = 0.0f, ny = 0.0f;  // nextPts[k]
  int st = 1;
  float er = 0.0f;
  const float FLT_SCALE = 1.0f / (float)(1 << 20);
  const int top = p.lay.n_levels - 1;
  for (int level = top; level >= 0; --level) {
    const Level L = p.lay.lv[level];
    const uint8_t* I = base_i + L.img_off;
    const int16_t* D = reinterpret_cast<const int16_t*>(base_i + L.deriv_off);
    const uint8_t* J = base_j + L.img_off;
    const int h = L.h, w = L.w;
    const float sc = 1.0f / (float)(1 << level);
    float prx = px * sc, pry = py * sc;
    float qx, qy;
    if (level == top) {
      qx = prx;
      qy = pry;
    } else {
      qx = nx * 2.0f;
      qy = ny * 2.0f;
    }
    nx = qx;
    ny = qy;
    prx -= half;
    pry -= half;
    const int ipx = cv_floor(prx), ipy = cv_floor(pry);
    if (ipx < -win || ipx >= w || ipy < -win || ipy >= h) {
      if (level == 0) {
        st = 0;
        er = 0.0f;
      }
      continue;
    }
    int w00, w01, w10, w11;
    lk_weights(prx - (float)ipx, pry - (float)ipy, w00, w01, w10, w11);
    // ---- the lane's pixels of the I / Ix / Iy patch (registers) and the structure matrix
    int ipatch[NPL], ixp[NPL], iyp[NPL];
    float a11 = 0.0f, a12 = 0.0f, a22 = 0.0f;
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
      const int y0 = ipy + wyk[q], x0 = ipx + wxk[q];
      const int ya = reflect101(y0, h), yb = reflect101(y0 + 1, h), xa = reflect101(x0, w), xb = reflect101(x0 + 1, w);
      const int iv = (I[(size_t)ya * w + xa] * w00 + I[(size_t)ya * w + xb] * w01 + I[(size_t)yb * w + xa] * w10 +
                      I[(size_t)yb * w + xb] * w11 + (1 << 8)) >> 9;  // CV_DESCALE(., W_BITS - 5)
      // derivatives are zero outside the level (BORDER_CONSTANT); clamped addresses keep the loads unconditional
      const bool oa = (unsigned)y0 < (unsigned)h, ob = (unsigned)(y0 + 1) < (unsigned)h;
      const bool pa = (unsigned)x0 < (unsigned)w, pb = (unsigned)(x0 + 1) < (unsigned)w;
      const int yc0 = min(max(y0, 0), h - 1), yc1 = min(max(y0 + 1, 0), h - 1);
      const int xc0 = min(max(x0, 0), w - 1), xc1 = min(max(x0 + 1, 0), w - 1);
      const int d00 = *reinterpret_cast<const int*>(D + 2 * ((size_t)yc0 * w + xc0));
      const int d01 = *reinterpret_cast<const int*>(D + 2 * ((size_t)yc0 * w + xc1));
      const int d10 = *reinterpret_cast<const int*>(D + 2 * ((size_t)yc1 * w + xc0));
      const int d11 = *reinterpret_cast<const int*>(D + 2 * ((size_t)yc1 * w + xc1));
      const int m00 = (oa && pa) ? w00 : 0, m01 = (oa && pb) ? w01 : 0, m10 = (ob && pa) ? w10 : 0, m11 = (ob && pb) ? w11 : 0;
      // (int16 x, int16 y) packed little endian: x = low half, y = high half (sign extended)
      const int ixv = ((int)(int16_t)d00 * m00 + (int)(int16_t)d01 * m01 + (int)(int16_t)d10 * m10 + (int)(int16_t)d11 * m11 + (1 << 13)) >> 14;
      const int iyv = ((d00 >> 16) * m00 + (d01 >> 16) * m01 + (d10 >> 16) * m10 + (d11 >> 16) * m11 + (1 << 13)) >> 14;
      ipatch[q] = iv;
      ixp[q] = act[q] ? ixv : 0;
      iyp[q] = act[q] ? iyv : 0;
      a11 += (float)(ixp[q] * ixp[q]);
      a12 += (float)(ixp[q] * iyp[q]);
      a22 += (float)(iyp[q] * iyp[q]);
    }
    a11 = wave_sum(a11) * FLT_SCALE;
    a12 = wave_sum(a12) * FLT_SCALE;
    a22 = wave_sum(a22) * FLT_SCALE;
    float det = a11 * a22 - a12 * a12;
    const float min_eig = (a22 + a11 - sqrtf((a11 - a22) * (a11 - a22) + 4.0f * a12 * a12)) / (float)(2 * win * win);
    if (min_eig < 1e-4f || det < 1.1920929e-07f) {
      if (level == 0) st = 0;
      continue;
    }
    det = 1.0f / det;
    qx -= half;
    qy -= half;
    // ---- J around the starting guess, reflected, in LDS: stage(y, x) = J(reflect(sy0 + y), reflect(sx0 + x))
    int sx0 = 0, sy0 = 0;
    bool staged = false;
    auto stage_box = [&](int cx, int cy) {
      sx0 = cx - STAGE_MARGIN;
      sy0 = cy - STAGE_MARGIN;
      // element i = lane + 64 j of the S x S box: (yy, xx) advances by (64 / S, 64 % S) per step -- no division in the loop
      for (int i = lane, yy = lane_y, xx = lane_x; i < S * S; i += 64) {
        stage[i] = J[(size_t)reflect101(sy0 + yy, h) * w + reflect101(sx0 + xx, w)];
        xx += step_x;
        yy += step_y;
        if (xx >= S) {
          xx -= S;
          ++yy;
        }
      }
      staged = true;  // the lanes of a wave run in lockstep: no barrier between these stores and the loads below is needed
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    };
    // sum over the window of f(J patch value - I patch value, lane pixel q) from the staged box
    auto ensure = [&](int inx, int iny) {
      if (!staged || inx < sx0 || iny < sy0 || inx + win + 1 > sx0 + S || iny + win + 1 > sy0 + S) stage_box(inx, iny);
    };
    float pdx = 0.0f, pdy = 0.0f;
    for (int j = 0; j < p.max_count; ++j) {
      const int inx = cv_floor(qx), iny = cv_floor(qy);
      if (inx < -win || inx >= w || iny < -win || iny >= h) {
        if (level == 0) st = 0;
        break;
      }
      ensure(inx, iny);
      lk_weights(qx - (float)inx, qy - (float)iny, w00, w01, w10, w11);
      const uint8_t* sb = stage + (iny - sy0) * S + (inx - sx0);
      float b1 = 0.0f, b2 = 0.0f;
#pragma unroll
      for (int q = 0; q < NPL; ++q) {
        const uint8_t* s0 = sb + wyk[q] * S + wxk[q];
        const int jv = (s0[0] * w00 + s0[1] * w01 + s0[S] * w10 + s0[S + 1] * w11 + (1 << 8)) >> 9;
        const int diff = jv - ipatch[q];
        b1 += (float)(diff * ixp[q]);
        b2 += (float)(diff * iyp[q]);
      }
      b1 = wave_sum(b1) * FLT_SCALE;
      b2 = wave_sum(b2) * FLT_SCALE;
      const float dx = (a12 * b2 - a22 * b1) * det, dy = (a12 * b1 - a11 * b2) * det;
      qx += dx;
      qy += dy;
      nx = qx + half;
      ny = qy + half;
      if (dx * dx + dy * dy <= p.eps2) break;
      if (j > 0 && fabsf(dx + pdx) < 0.01f && fabsf(dy + pdy) < 0.01f) {
        nx -= dx * 0.5f;
        ny -= dy * 0.5f;
        break;
      }
      pdx = dx;
      pdy = dy;
    }
    if (st && level == 0) {
      const float ex = nx - half, ey = ny - half;
      const int inx = cv_floor(ex), iny = cv_floor(ey);
      if (inx < -win || inx >= w || iny < -win || iny >= h) {
        st = 0;
        continue;
      }
      ensure(inx, iny);
      lk_weights(ex - (float)inx, ey - (float)iny, w00, w01, w10, w11);
      const uint8_t* sb = stage + (iny - sy0) * S + (inx - sx0);
      float e = 0.0f;
#pragma unroll
      for (int q = 0; q < NPL; ++q) {
        const uint8_t* s0 = sb + wyk[q] * S + wxk[q];
        const int jv = (s0[0] * w00 + s0[1] * w01 + s0[S] * w10 + s0[S + 1] * w11 + (1 << 8)) >> 9;
        e += act[q] ? fabsf((float)(jv - ipatch[q])) : 0.0f;
      }
      er = wave_sum(e) * (1.0f / (float)(32 * win * win));
    }
  }
  if (lane == 0) {
    p.next_pts[2 * k] = nx;
    p.next_pts[2 * k + 1] = ny;
    p.status[k] = (uint8_t)st;
    p.err[k] = er;
  }
}

int grid_for(size_t total, int cap = 4096) {
  size_t g = (total + 255) / 256;
  if (g > (size_t)cap) g = cap;
  return g < 1 ? 1 : (int)g;
}

// all levels of F frames (F == 1: `base` is the one pyramid buffer; else `pyr` lists F of them): 1 + 2 n_levels - 1 launches
// (H, W) = the size of the pyramid's level 0; (Hsrc, Wsrc) = the size of the frames: different when the flow tracker's img_scale
// is not 1 (the frames are resized on the way into level 0)
int pyramid_launch(const uint8_t* images, int F, int H, int W, int C, const PyrLayout& lay, uint8_t* base, uint8_t* const* pyr,
                   hipStream_t st, int Hsrc = 0, int Wsrc = 0, double fscale = 1.0) {
  auto io = [&](size_t src_off, size_t dst_off) {
    PyrIO v;
    v.src = base ? base + src_off : nullptr;
    v.dst = base ? base + dst_off : nullptr;
    v.pyr = pyr;
    v.src_off = src_off;
    v.dst_off = dst_off;
    return v;
  };
  auto tiles = [&](int h, int w) { return dim3((unsigned)((w + 63) / 64), (unsigned)((h + 3) / 4), (unsigned)F); };
  auto tiles4 = [&](int h, int w) { return dim3((unsigned)((w + 255) / 256), (unsigned)((h + 3) / 4), (unsigned)F); };
  SA_REQUIRE((H + 3) / 4 <= 65535 && F <= 65535, "sa_flow_pyramid_build: frame too tall for one launch");
  if (Hsrc > 0 && (Hsrc != H || Wsrc != W || fscale != 1.0))
    hipLaunchKernelGGL(flow_gray_resize_kernel, tiles(H, W), dim3(256), 0, st, images, Hsrc, Wsrc, C, H, W, 1.0 / fscale, 1.0 / fscale,
                       io(0, lay.lv[0].img_off));
  else
    hipLaunchKernelGGL(flow_gray_kernel, dim3((unsigned)grid_for((size_t)H * W / (C == 1 ? 16 : 1), 1024), 1, (unsigned)F), dim3(256),
                       0, st, images, H * W, C, io(0, lay.lv[0].img_off));
  for (int l = 0; l < lay.n_levels; ++l) {
    const Level& L = lay.lv[l];
    if (l > 0) {
      const Level& P = lay.lv[l - 1];
      if (P.w % 8 == 0 && P.w >= 16 && P.h >= 3)
        hipLaunchKernelGGL(flow_pyrdown_x4_kernel, tiles4(L.h, L.w), dim3(256), 0, st, io(P.img_off, L.img_off), P.h, P.w, L.h, L.w);
      else
        hipLaunchKernelGGL(flow_pyrdown_kernel, tiles(L.h, L.w), dim3(256), 0, st, io(P.img_off, L.img_off), P.h, P.w, L.h, L.w);
    }
    if (L.w % 4 == 0 && L.w >= 8)
      hipLaunchKernelGGL(flow_scharr_x4_kernel, tiles4(L.h, L.w), dim3(256), 0, st, io(L.img_off, L.deriv_off), L.h, L.w);
    else
      hipLaunchKernelGGL(flow_scharr_kernel, tiles(L.h, L.w), dim3(256), 0, st, io(L.img_off, L.deriv_off), L.h, L.w);
  }
  return SA_OK;
}

}  // namespace

extern "C" {

int sa_flow_pyramid_levels(int H, int W, int win, int max_level) {
  if (H <= 0 || W <= 0 || win < 3 || max_level < 0) return 0;
  return pyr_layout(H, W, win, max_level).n_levels;
}

size_t sa_flow_pyramid_bytes(int H, int W, int win, int max_level) {
  if (H <= 0 || W <= 0 || win < 3 || max_level < 0) return 0;
  return pyr_layout(H, W, win, max_level).bytes;
}

int sa_flow_pyramid_build(const void* image, int H, int W, int C, int win, int max_level, void* pyramid, sa_stream_t stream) {
  SA_REQUIRE(image && pyramid, "sa_flow_pyramid_build: NULL pointer");
  SA_REQUIRE(H > 0 && W > 0 && (C == 1 || C == 3), "sa_flow_pyramid_build: frames are [H,W,1] or [H,W,3] uint8");
  SA_REQUIRE(win >= 3 && win <= MAX_WIN && max_level >= 0, "sa_flow_pyramid_build: window must be in 3..%d", MAX_WIN);
  const PyrLayout lay = pyr_layout(H, W, win, max_level);
  uint8_t* base = static_cast<uint8_t*>(pyramid);
  hipStream_t st = (hipStream_t)stream;
  const int rc = pyramid_launch(static_cast<const uint8_t*>(image), 1, H, W, C, lay, base, nullptr, st);
  if (rc != SA_OK) return rc;
  SA_LAUNCH_CHECK();
  return SA_OK;
}

int sa_flow_scaled_size(int H, int W, double img_scale, int* Hs, int* Ws) {
  SA_REQUIRE(Hs && Ws && H > 0 && W > 0 && img_scale > 0.0, "sa_flow_scaled_size: bad arguments");
  *Hs = (int)rint((double)H * img_scale);  // cv::resize: dsize = saturate_cast<int>(size * f) = cvRound (half to even)
  *Ws = (int)rint((double)W * img_scale);
  SA_REQUIRE(*Hs > 0 && *Ws > 0, "sa_flow_scaled_size: the scaled frame is empty");
  return SA_OK;
}

int sa_flow_pyramid_build_scaled(const void* images, int F, int H, int W, int C, double img_scale, int win, int max_level,
                                 void* pyramid, void* const* pyramids, sa_stream_t stream) {
  if (F == 0) return SA_OK;
  SA_REQUIRE(images && (pyramid || pyramids) && F > 0 && F <= 65535 && (F == 1 || pyramids), "sa_flow_pyramid_build_scaled: bad arguments");
  SA_REQUIRE(H > 0 && W > 0 && (C == 1 || C == 3), "sa_flow_pyramid_build_scaled: frames are [H,W,1] or [H,W,3] uint8");
  SA_REQUIRE(win >= 3 && win <= MAX_WIN && max_level >= 0, "sa_flow_pyramid_build_scaled: window must be in 3..%d", MAX_WIN);
  int Hs = 0, Ws = 0;
  const int rs = sa_flow_scaled_size(H, W, img_scale, &Hs, &Ws);
  if (rs != SA_OK) return rs;
  const PyrLayout lay = pyr_layout(Hs, Ws, win, max_level);
  const int rc = pyramid_launch(static_cast<const uint8_t*>(images), F, Hs, Ws, C, lay, pyramids ? nullptr : static_cast<uint8_t*>(pyramid),
                                reinterpret_cast<uint8_t* const*>(pyramids), (hipStream_t)stream, H, W, img_scale);
  if (rc != SA_OK) return rc;
  SA_LAUNCH_CHECK();
  return SA_OK;
}

int sa_flow_pyramid_build_batch(const void* images, int F, int H, int W, int C, int win, int max_level, void* const* pyramids,
                                sa_stream_t stream) {
  if (F == 0) return SA_OK;
  SA_REQUIRE(images && pyramids && F > 0 && F <= 65535, "sa_flow_pyramid_build_batch: bad arguments");
  SA_REQUIRE(H > 0 && W > 0 && (C == 1 || C == 3), "sa_flow_pyramid_build_batch: frames are [H,W,1] or [H,W,3] uint8");
  SA_REQUIRE(win >= 3 && win <= MAX_WIN && max_level >= 0, "sa_flow_pyramid_build_batch: window must be in 3..%d", MAX_WIN);
  const PyrLayout lay = pyr_layout(H, W, win, max_level);
  uint8_t* const* pyr = reinterpret_cast<uint8_t* const*>(pyramids);
  hipStream_t st = (hipStream_t)stream;
  const int rc = pyramid_launch(static_cast<const uint8_t*>(images), F, H, W, C, lay, nullptr, pyr, st);
  if (rc != SA_OK) return rc;
  SA_LAUNCH_CHECK();
  return SA_OK;
}

static int flow_lk_launch(const void* const* pyr_prev, const void* pyr_next, const void* const* pyr_next_v, int H, int W, int win,
                          int max_level, int n, const float* prev_pts, float* next_pts, uint8_t* status, float* err, int max_count,
                          float epsilon, sa_stream_t stream) {
  if (n == 0) return SA_OK;
  SA_REQUIRE(pyr_prev && (pyr_next || pyr_next_v) && prev_pts && next_pts && status && err, "sa_flow_lk: NULL pointer");
  SA_REQUIRE(n > 0 && H > 0 && W > 0 && win >= 3 && win <= MAX_WIN && max_level >= 0, "sa_flow_lk: bad arguments");
  LkParams p;
  p.pyr_prev = reinterpret_cast<const uint8_t* const*>(pyr_prev);
  p.pyr_next = static_cast<const uint8_t*>(pyr_next);
  p.pyr_next_v = reinterpret_cast<const uint8_t* const*>(pyr_next_v);
  p.lay = pyr_layout(H, W, win, max_level);
  p.win = win;
  p.n = n;
  p.max_count = max_count < 0 ? 0 : (max_count > 100 ? 100 : max_count);  // TermCriteria clamps, lkpyramid.cpp
  const float e = epsilon < 0.0f ? 0.0f : (epsilon > 10.0f ? 10.0f : epsilon);
  p.eps2 = e * e;
  p.prev_pts = prev_pts;
  p.next_pts = next_pts;
  p.status = status;
  p.err = err;
  const int S = win + 1 + 2 * STAGE_MARGIN;
  const size_t lds = (size_t)4 * S * S;
  const dim3 grid((unsigned)((n + 3) / 4)), block(256);
  const int npl = (win * win + 63) / 64;
  if (npl <= 4)
    hipLaunchKernelGGL(flow_lk_kernel<4>, grid, block, lds, (hipStream_t)stream, p);
  else if (npl <= 7)
    hipLaunchKernelGGL(flow_lk_kernel<7>, grid, block, lds, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(flow_lk_kernel<16>, grid, block, lds, (hipStream_t)stream, p);
  SA_LAUNCH_CHECK();
  return SA_OK;
}

int sa_flow_lk(const void* const* pyr_prev, const void* pyr_next, int H, int W, int win, int max_level, int n,
               const float* prev_pts, float* next_pts, uint8_t* status, float* err, int max_count, float epsilon,
               sa_stream_t stream) {
  return flow_lk_launch(pyr_prev, pyr_next, nullptr, H, W, win, max_level, n, prev_pts, next_pts, status, err, max_count, epsilon,
                        stream);
}

int sa_flow_lk_pairs(const void* const* pyr_prev, const void* const* pyr_next, int H, int W, int win, int max_level, int n,
                     const float* prev_pts, float* next_pts, uint8_t* status, float* err, int max_count, float epsilon,
                     sa_stream_t stream) {
  return flow_lk_launch(pyr_prev, nullptr, pyr_next, H, W, win, max_level, n, prev_pts, next_pts, status, err, max_count, epsilon,
                        stream);
}

}  // extern "C"
