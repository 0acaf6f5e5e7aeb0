// Post-processing kernels of the bottom-up hot path for gfx950: local/global peak finding with
// sub-pixel refinement, PAF line scoring, Hungarian matching and greedy instance assembly.
//
// Everything here is HBM/latency-bound integer & fp32 work (no MFMA). Results stay on the device
// in fixed-shape, NaN-padded buffers so that a whole batch needs a single D2H copy (or a single
// RCCL all-gather) -- the reference bounces host<->device 13 times per frame
// (paf_grouping.py:615 tf.numpy_function, :1244 tf.py_function).
//
// fp32 arithmetic that feeds a discontinuous decision (floor/ceil/round/compare) is written with
// explicitly un-fused intrinsics (__fmul_rn/__fadd_rn/...) so it evaluates op-by-op like the
// TensorFlow CPU kernels the reference lowers to; this file is also compiled with
// -ffp-contract=off.
#include <cstdint>
#include <vector>

#include "bf16.h"
#include "lsa.h"
#include "sa_common.h"

namespace {

constexpr int MAXNP = 512;  // cap on max_node_peaks (peaks of one node type per frame); bounds table sizes only
constexpr int MAXNODES = 64;

// ------------------------------------------------------------------------------------------------
// Peak finding
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ bool nms_is_peak(const float* __restrict__ img, int H, int W, int C,
                                            int y, int x, int c, float v) {
  // tf.nn.dilation2d with kernel [[0,0,0],[0,-1,0],[0,0,0]], SAME: max(8 nbrs, centre - 1),
  // out-of-bounds taps ignored (peak_finding.py:274-290); strict compare.
  float m = __fsub_rn(v, 1.0f);
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy) {
    const int yy = y + dy;
    if (yy < 0 || yy >= H) continue;
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int xx = x + dx;
      if ((dy == 0 && dx == 0) || xx < 0 || xx >= W) continue;
      const float n = img[((size_t)yy * W + xx) * C + c];
      if (n > m) m = n;
    }
  }
  return v > m;
}

__device__ __forceinline__ void nms_emit(uint32_t e, int b, int max_peaks, uint32_t* keys,
                                         int32_t* counts, int32_t* status) {
  const int slot = atomicAdd(&counts[b], 1);
  if (slot < max_peaks)
    keys[(size_t)b * max_peaks + slot] = e;
  else
    atomicOr(&status[b], SA_STATUS_PEAK_OVERFLOW);
}

// One pass over the confidence maps: threshold test on coalesced float4 loads, the (rare)
// survivors do the 8-neighbour test and append their flat (y,x,c) key to the frame's list.
__global__ void __launch_bounds__(256)
nms_scan_kernel(const float* __restrict__ cms, int H, int W, int C, float thr, int max_peaks,
                uint32_t* __restrict__ keys, int32_t* __restrict__ counts,
                int32_t* __restrict__ status, int vec4) {
  const int b = blockIdx.y;
  const size_t plane = (size_t)H * W * C;
  const float* img = cms + (size_t)b * plane;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  if (vec4) {
    const size_t n4 = plane / 4;
    const float4* img4 = reinterpret_cast<const float4*>(img);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
      const float4 q = img4[i];
      // inf - inf and NaN - NaN are NaN: one test for "any of the four is not finite"
      const float qs = __fadd_rn(__fadd_rn(q.x, q.y), __fadd_rn(q.z, q.w));
      if (__fsub_rn(qs, qs) != 0.0f) atomicOr(&status[b], SA_STATUS_NONFINITE);
      if (!(q.x > thr || q.y > thr || q.z > thr || q.w > thr)) continue;
      const float vals[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v = vals[j];
        if (!(v > thr)) continue;
        const size_t e = i * 4 + j;
        const int c = (int)(e % C);
        const size_t p = e / C;
        const int x = (int)(p % W), y = (int)(p / W);
        if (nms_is_peak(img, H, W, C, y, x, c, v)) nms_emit((uint32_t)e, b, max_peaks, keys, counts, status);
      }
    }
  } else {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < plane; e += stride) {
      const float v = img[e];
      if (__fsub_rn(v, v) != 0.0f) atomicOr(&status[b], SA_STATUS_NONFINITE);
      if (!(v > thr)) continue;
      const int c = (int)(e % C);
      const size_t p = e / C;
      const int x = (int)(p % W), y = (int)(p / W);
      if (nms_is_peak(img, H, W, C, y, x, c, v)) nms_emit((uint32_t)e, b, max_peaks, keys, counts, status);
    }
  }
}

// tf.image.crop_and_resize(bilinear, extrapolation 0) of one k x k box centred on the integer
// pixel (cx, cy), evaluated with the CPU kernel's float32 op sequence (see oracle/peak_finding.py
// crop_and_resize_bilinear and peak_finding.py:135-190, instance_cropping.py:58-166).
struct CropAxis {
  int lo, hi;
  float lerp;
  bool ok;
};

__device__ __forceinline__ CropAxis crop_axis(int centre, int k, int i, int size) {
  const float cf = (float)centre;
  const float b1 = __fadd_rn(cf, __fmul_rn((float)(-k + 1), 0.5f));
  const float b2 = __fadd_rn(cf, __fmul_rn((float)(k - 1), 0.5f));
  const float sm1 = __fsub_rn((float)size, 1.0f);
  const float n1 = __fdiv_rn(b1, sm1), n2 = __fdiv_rn(b2, sm1);
  float in;
  if (k > 1) {
    const float scale = __fdiv_rn(__fmul_rn(__fsub_rn(n2, n1), sm1), (float)(k - 1));
    in = __fadd_rn(__fmul_rn(n1, sm1), __fmul_rn((float)i, scale));
  } else {
    in = __fmul_rn(__fmul_rn(0.5f, __fadd_rn(n1, n2)), sm1);
  }
  CropAxis a;
  a.ok = !(in < 0.0f || in > sm1);
  const float fl = floorf(in), ce = ceilf(in);
  a.lerp = __fsub_rn(in, fl);
  a.lo = min(max((int)fl, 0), size - 1);
  a.hi = min(max((int)ce, 0), size - 1);
  return a;
}

__device__ __forceinline__ float crop_sample(const float* __restrict__ img, int W, int C, int c,
                                             const CropAxis& ay, const CropAxis& ax) {
  if (!(ay.ok && ax.ok)) return 0.0f;
  const float tl = img[((size_t)ay.lo * W + ax.lo) * C + c];
  const float tr = img[((size_t)ay.lo * W + ax.hi) * C + c];
  const float bl = img[((size_t)ay.hi * W + ax.lo) * C + c];
  const float br = img[((size_t)ay.hi * W + ax.hi) * C + c];
  const float t = __fadd_rn(tl, __fmul_rn(__fsub_rn(tr, tl), ax.lerp));
  const float bt = __fadd_rn(bl, __fmul_rn(__fsub_rn(br, bl), ax.lerp));
  return __fadd_rn(t, __fmul_rn(__fsub_rn(bt, t), ay.lerp));
}

__device__ __forceinline__ float signf(float v) {
  return (v > 0.0f) ? 1.0f : ((v < 0.0f) ? -1.0f : v);  // sign(0)=0, sign(NaN)=NaN (tf.sign)
}

// offset (dx, dy) in grid units for the rough peak (x, y) of channel c in frame image `img`.
__device__ void refine_offset(const float* __restrict__ img, const float* __restrict__ off, int H,
                              int W, int C, int y, int x, int c, int mode, int k, float* dx,
                              float* dy) {
  if (mode == SA_REFINE_INTEGRAL) {
    // integral_regression on the k x k crop: gv = arange(k) - (k-1)/2 (peak_finding.py:311-334)
    float z = 0.0f, sx = 0.0f, sy = 0.0f;
    const float half = (float)(k - 1) * 0.5f;
    for (int i = 0; i < k; ++i) {
      const CropAxis ay = crop_axis(y, k, i, H);
      const float gy = __fsub_rn((float)i, half);
      for (int j = 0; j < k; ++j) {
        const CropAxis ax = crop_axis(x, k, j, W);
        const float p = crop_sample(img, W, C, c, ay, ax);
        const float gx = __fsub_rn((float)j, half);
        z = __fadd_rn(z, p);
        sx = __fadd_rn(sx, __fmul_rn(gx, p));
        sy = __fadd_rn(sy, __fmul_rn(gy, p));
      }
    }
    *dx = __fdiv_rn(sx, z);
    *dy = __fdiv_rn(sy, z);
  } else if (mode == SA_REFINE_LOCAL) {
    // find_offsets_local_direction on the 3 x 3 crop (peak_finding.py:78-132)
    const CropAxis y0 = crop_axis(y, 3, 0, H), y1 = crop_axis(y, 3, 1, H), y2 = crop_axis(y, 3, 2, H);
    const CropAxis x0 = crop_axis(x, 3, 0, W), x1 = crop_axis(x, 3, 1, W), x2 = crop_axis(x, 3, 2, W);
    const float right = crop_sample(img, W, C, c, y1, x2), left = crop_sample(img, W, C, c, y1, x0);
    const float bottom = crop_sample(img, W, C, c, y2, x1), top = crop_sample(img, W, C, c, y0, x1);
    *dx = __fmul_rn(signf(__fsub_rn(right, left)), 0.25f);
    *dy = __fmul_rn(signf(__fsub_rn(bottom, top)), 0.25f);
  } else if (mode == SA_REFINE_OFFSETS) {
    // offsets.reshape(B,H,W,C,2)[b,y,x,c,:] (peak_finding.py:690-704)
    const float* o = off + (((size_t)y * W + x) * C + c) * 2;
    *dx = o[0];
    *dy = o[1];
  } else {
    *dx = 0.0f;
    *dy = 0.0f;
  }
}

// One workgroup per frame: bitonic-sort the frame's keys (restores tf.where's row-major (y,x,c)
// order, which every downstream index depends on) and refine each peak.
__global__ void __launch_bounds__(256)
peaks_sort_refine_kernel(const float* __restrict__ cms, const float* __restrict__ offsets, int H,
                         int W, int C, int mode, int k, float xy_scale, int max_peaks,
                         const uint32_t* __restrict__ keys, int32_t* __restrict__ counts,
                         float* __restrict__ peak_xy, float* __restrict__ peak_val,
                         int32_t* __restrict__ peak_chan) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint32_t* sk = reinterpret_cast<uint32_t*>(smem_raw);
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int n = min(counts[b], max_peaks);
  int n2 = 1;
  while (n2 < n) n2 <<= 1;
  for (int i = tid; i < n2; i += nt) sk[i] = (i < n) ? keys[(size_t)b * max_peaks + i] : 0xFFFFFFFFu;
  __syncthreads();
  for (int size = 2; size <= n2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = tid; i < n2; i += nt) {
        const int j = i ^ stride;
        if (j > i) {
          const bool up = ((i & size) == 0);
          const uint32_t a = sk[i], c2 = sk[j];
          if ((a > c2) == up) {
            sk[i] = c2;
            sk[j] = a;
          }
        }
      }
      __syncthreads();
    }
  }
  const size_t plane = (size_t)H * W * C;
  const float* img = cms + (size_t)b * plane;
  const float* off = offsets ? offsets + (size_t)b * plane * 2 : nullptr;
  for (int i = tid; i < n; i += nt) {
    const uint32_t e = sk[i];
    const int c = (int)(e % C);
    const uint32_t p = e / C;
    const int x = (int)(p % W), y = (int)(p / W);
    float dx, dy;
    refine_offset(img, off, H, W, C, y, x, c, mode, k, &dx, &dy);
    const size_t o = (size_t)b * max_peaks + i;
    peak_xy[o * 2 + 0] = __fmul_rn(__fadd_rn((float)x, dx), xy_scale);
    peak_xy[o * 2 + 1] = __fmul_rn(__fadd_rn((float)y, dy), xy_scale);
    peak_val[o] = img[e];
    peak_chan[o] = c;
  }
  if (tid == 0) counts[b] = n;
}

// find_global_peaks_rough (+ refinement): one workgroup per (frame, channel).
__global__ void __launch_bounds__(256)
global_peaks_kernel(const float* __restrict__ cms, const float* __restrict__ offsets, int H, int W,
                    int C, float thr, int mode, int k, float xy_scale, float* __restrict__ peak_xy,
                    float* __restrict__ peak_val) {
  __shared__ float s_max[256];
  __shared__ int s_row[256], s_col[256];
  const int b = blockIdx.y, c = blockIdx.x, tid = threadIdx.x;
  const size_t plane = (size_t)H * W * C;
  const float* img = cms + (size_t)b * plane;
  const int npix = H * W;
  float m = -__builtin_huge_valf();
  for (int p = tid; p < npix; p += blockDim.x) {
    const float v = img[(size_t)p * C + c];
    if (v > m) m = v;
  }
  s_max[tid] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s && s_max[tid + s] > s_max[tid]) s_max[tid] = s_max[tid + s];
    __syncthreads();
  }
  m = s_max[0];
  // argmax over rows of (max over x) and over columns of (max over y): first index attaining m,
  // taken independently (peak_finding.py:215-221)
  int row = H, col = W;
  for (int p = tid; p < npix; p += blockDim.x) {
    if (img[(size_t)p * C + c] == m) {
      row = min(row, p / W);
      col = min(col, p % W);
    }
  }
  s_row[tid] = row;
  s_col[tid] = col;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      s_row[tid] = min(s_row[tid], s_row[tid + s]);
      s_col[tid] = min(s_col[tid], s_col[tid + s]);
    }
    __syncthreads();
  }
  if (tid == 0) {
    row = s_row[0] < H ? s_row[0] : 0;  // all-NaN plane: tf.argmax yields 0
    col = s_col[0] < W ? s_col[0] : 0;
    const float v = img[((size_t)row * W + col) * C + c];
    const size_t o = (size_t)b * C + c;
    peak_val[o] = v;
    if (v < thr) {
      peak_xy[o * 2 + 0] = __builtin_nanf("");
      peak_xy[o * 2 + 1] = __builtin_nanf("");
    } else {
      float dx = 0.0f, dy = 0.0f;
      const float* off = offsets ? offsets + (size_t)b * plane * 2 : nullptr;
      refine_offset(img, off, H, W, C, row, col, c, mode, k, &dx, &dy);
      peak_xy[o * 2 + 0] = __fmul_rn(__fadd_rn((float)col, dx), xy_scale);
      peak_xy[o * 2 + 1] = __fmul_rn(__fadd_rn((float)row, dy), xy_scale);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Instance crops for the top-down path: crop_bboxes (peak_finding.py:135-190) = tf.image.crop_and_resize
// (bilinear, extrapolation 0) of crop x crop boxes centred on FRACTIONAL centroids
// (make_centered_bboxes, instance_cropping.py:124-166), result cast back to the image dtype.
// Same float32 op sequence as the TF CPU kernel (normalised boxes, scale, floor/ceil/lerp).
// ------------------------------------------------------------------------------------------------
struct CropCoord {
  int lo, hi;
  float lerp;
  bool ok;
};

__device__ __forceinline__ CropCoord crop_coord(float centre, int k, int i, int size) {
  const float b1 = __fadd_rn(centre, __fmul_rn((float)(-k + 1), 0.5f));
  const float b2 = __fadd_rn(centre, __fmul_rn((float)(k - 1), 0.5f));
  const float sm1 = __fsub_rn((float)size, 1.0f);
  const float n1 = __fdiv_rn(b1, sm1), n2 = __fdiv_rn(b2, sm1);
  float in;
  if (k > 1) {
    const float scale = __fdiv_rn(__fmul_rn(__fsub_rn(n2, n1), sm1), (float)(k - 1));
    in = __fadd_rn(__fmul_rn(n1, sm1), __fmul_rn((float)i, scale));
  } else {
    in = __fmul_rn(__fmul_rn(0.5f, __fadd_rn(n1, n2)), sm1);
  }
  CropCoord a;
  a.ok = !(in < 0.0f || in > sm1);  // NaN centroids compare false twice -> "ok", handled by the caller's mask
  const float fl = floorf(in), ce = ceilf(in);
  a.lerp = __fsub_rn(in, fl);
  a.lo = min(max((int)fl, 0), size - 1);
  a.hi = min(max((int)ce, 0), size - 1);
  return a;
}

template <typename T>
__global__ void __launch_bounds__(256)
crop_and_resize_kernel(const T* __restrict__ images, int H, int W, int C, const float* __restrict__ centres_xy,
                       const int32_t* __restrict__ sample_inds, int n, int crop, T* __restrict__ out) {
  const size_t total = (size_t)n * crop * crop;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const int xx = (int)(t % crop);
    const int yy = (int)((t / crop) % crop);
    const int i = (int)(t / ((size_t)crop * crop));
    const float cx = centres_xy[2 * i], cy = centres_xy[2 * i + 1];
    const CropCoord ay = crop_coord(cy, crop, yy, H), ax = crop_coord(cx, crop, xx, W);
    const T* img = images + (size_t)sample_inds[i] * H * W * C;
    T* o = out + t * C;
    for (int c = 0; c < C; ++c) {
      float v = 0.0f;
      if (ay.ok && ax.ok) {
        const float tl = (float)img[((size_t)ay.lo * W + ax.lo) * C + c], tr = (float)img[((size_t)ay.lo * W + ax.hi) * C + c];
        const float bl = (float)img[((size_t)ay.hi * W + ax.lo) * C + c], br = (float)img[((size_t)ay.hi * W + ax.hi) * C + c];
        const float top = __fadd_rn(tl, __fmul_rn(__fsub_rn(tr, tl), ax.lerp));
        const float bot = __fadd_rn(bl, __fmul_rn(__fsub_rn(br, bl), ax.lerp));
        v = __fadd_rn(top, __fmul_rn(__fsub_rn(bot, top), ay.lerp));
      }
      o[c] = (T)v;  // tf.cast(crops, images.dtype): truncation for uint8
    }
  }
}

// ------------------------------------------------------------------------------------------------
// PAF scoring
// ------------------------------------------------------------------------------------------------

// One workgroup per frame. Phase 1 buckets the frame's peaks by node type (stable, so the s-th
// entry of a node is its s-th peak in (y,x) order == tf.argsort/top_k order, paf_grouping.py:105).
// Phase 2 walks the dense (edge, src, dst) candidate space; every candidate samples n_points
// nearest-pixel PAF vectors along src->dst and averages their projection on the unit vector.
__global__ void __launch_bounds__(256)
paf_score_kernel(const float* __restrict__ pafs, int Hp, int Wp, int E,
                 const float* __restrict__ peak_xy, const int32_t* __restrict__ peak_chan,
                 const int32_t* __restrict__ peak_count, int max_peaks,
                 const int32_t* __restrict__ edges, int N, int n_points, float pafs_stride,
                 float max_edge_length, float dist_penalty_weight, int NP,
                 int32_t* __restrict__ node_count, int32_t* __restrict__ node_peaks,
                 float* __restrict__ line_scores, int32_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  int32_t* s_cnt = reinterpret_cast<int32_t*>(smem_raw);  // [N]
  int32_t* s_list = s_cnt + N;                            // [N][NP]
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int n = min(peak_count[b], max_peaks);
  const int32_t* ch = peak_chan + (size_t)b * max_peaks;
  const float* xy = peak_xy + (size_t)b * max_peaks * 2;
  for (int i = tid; i < N; i += nt) s_cnt[i] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += nt) {
    const int c = ch[i];
    int rank = 0;
    for (int j = 0; j < i; ++j) rank += (ch[j] == c);
    if (rank < NP)
      s_list[c * NP + rank] = i;
    else
      atomicOr(&status[b], SA_STATUS_NODE_PEAK_OVERFLOW);
    atomicAdd(&s_cnt[c], 1);
  }
  __syncthreads();
  for (int i = tid; i < N; i += nt) {
    const int cnt = min(s_cnt[i], NP);
    s_cnt[i] = cnt;
    node_count[(size_t)b * N + i] = cnt;
  }
  __syncthreads();
  for (int i = tid; i < N * NP; i += nt)
    node_peaks[(size_t)b * N * NP + i] = ((i % NP) < s_cnt[i / NP]) ? s_list[i] : -1;

  const float* paf = pafs + (size_t)b * Hp * Wp * 2 * E;
  const int PC = 2 * E;
  const int total = E * NP * NP;
  bool oob = false;
  for (int idx = tid; idx < total; idx += nt) {
    const int k = idx / (NP * NP);
    const int s = (idx / NP) % NP, d = idx % NP;
    const int sn = edges[2 * k], dn = edges[2 * k + 1];
    if (s >= s_cnt[sn] || d >= s_cnt[dn]) continue;
    const int ps = s_list[sn * NP + s], pd = s_list[dn * NP + d];
    const float sx = xy[2 * ps], sy = xy[2 * ps + 1], ex = xy[2 * pd], ey = xy[2 * pd + 1];
    // spatial vector, tf.norm = sqrt(sum(v*v)) (paf_grouping.py:378-383)
    const float vx = __fsub_rn(ex, sx), vy = __fsub_rn(ey, sy);
    const float len = __fsqrt_rn(__fadd_rn(__fmul_rn(vx, vx), __fmul_rn(vy, vy)));
    const float ux = __fdiv_rn(vx, len), uy = __fdiv_rn(vy, len);
    // tf.linspace: delta = (stop - start) / (n - 1); start + delta * i; exact end points
    const int steps = max(n_points - 1, 1);
    const float ddx = __fdiv_rn(vx, (float)steps), ddy = __fdiv_rn(vy, (float)steps);
    float acc = 0.0f;
    for (int i = 0; i < n_points; ++i) {
      float px, py;
      if (i == 0) {
        px = sx;
        py = sy;
      } else if (i == n_points - 1) {
        px = ex;
        py = ey;
      } else {
        px = __fadd_rn(sx, __fmul_rn(ddx, (float)i));
        py = __fadd_rn(sy, __fmul_rn(ddy, (float)i));
      }
      // tf.round = half-to-even; no clipping in the reference (paf_grouping.py:192-197)
      const float rx = rintf(__fdiv_rn(px, pafs_stride)), ry = rintf(__fdiv_rn(py, pafs_stride));
      float fx = 0.0f, fy = 0.0f;
      if (rx >= 0.0f && rx < (float)Wp && ry >= 0.0f && ry < (float)Hp) {
        const float* q = paf + ((size_t)(int)ry * Wp + (int)rx) * PC + 2 * k;
        fx = q[0];
        fy = q[1];
      } else {
        oob = true;  // TF-CPU gather_nd raises; TF-GPU reads 0 -- we read 0 and flag it
      }
      acc = __fadd_rn(acc, __fadd_rn(__fmul_rn(fx, ux), __fmul_rn(fy, uy)));
    }
    const float mean = __fdiv_rn(acc, (float)n_points);
    // compute_distance_penalty (paf_grouping.py:278-322)
    const float pen = __fmul_rn(fminf(__fsub_rn(__fdiv_rn(max_edge_length, len), 1.0f), 0.0f),
                                dist_penalty_weight);
    line_scores[(((size_t)b * E + k) * NP + s) * NP + d] = __fadd_rn(mean, pen);
  }
  if (oob) atomicOr(&status[b], SA_STATUS_PAF_OOB);
}

// ------------------------------------------------------------------------------------------------
// Matching: one thread per (frame, edge) runs the rectangular LSA on its (n_src x n_dst) block.
// ------------------------------------------------------------------------------------------------
struct CostView {
  const float* s;  // [NP][NP] scores of this (frame, edge)
  int NP;
  bool transposed;
  __device__ double operator()(int i, int j) const {
    const float v = transposed ? s[j * NP + i] : s[i * NP + j];
    // cost = where(isnan(score), +inf, -score)  (paf_grouping.py:625-631), evaluated in f64 as
    // SciPy does after np.asarray(cost, dtype=float64)
    return (v != v) ? __builtin_huge_val() : -(double)v;
  }
};

__global__ void __launch_bounds__(64)
paf_match_kernel(const float* __restrict__ line_scores, const int32_t* __restrict__ node_count,
                 const int32_t* __restrict__ edges, int B, int E, int N, int NP,
                 int32_t* __restrict__ match_dst, float* __restrict__ match_score,
                 int32_t* __restrict__ status, unsigned char* __restrict__ workspace) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * E) return;
  const int b = t / E, k = t % E;
  const int n_src = node_count[(size_t)b * N + edges[2 * k]];
  const int n_dst = node_count[(size_t)b * N + edges[2 * k + 1]];
  int32_t* md = match_dst + (size_t)t * NP;
  float* ms = match_score + (size_t)t * NP;
  for (int i = 0; i < NP; ++i) {
    md[i] = -1;
    ms[i] = __builtin_nanf("");
  }
  if (n_src == 0 || n_dst == 0) return;
  const float* sc = line_scores + (size_t)t * NP * NP;
  // SciPy rejects NaN / -inf costs ("matrix contains invalid numeric entries"); NaN scores were
  // mapped to +inf, so only a +inf score (cost -inf) is invalid.
  for (int i = 0; i < n_src; ++i)
    for (int j = 0; j < n_dst; ++j)
      if (sc[i * NP + j] == __builtin_huge_valf()) {
        atomicOr(&status[b], SA_STATUS_LSA_INFEASIBLE);
        return;
      }
  sa::LsaWork w;
  w.bind(workspace + (size_t)t * sa::LsaWork::bytes(NP), NP);
  const bool tr = n_dst < n_src;
  CostView cv{sc, NP, tr};
  const int nr = tr ? n_dst : n_src, nc = tr ? n_src : n_dst;
  if (!sa::lsa_solve(nr, nc, cv, w)) {
    atomicOr(&status[b], SA_STATUS_LSA_INFEASIBLE);
    return;
  }
  for (int r = 0; r < nr; ++r) {
    const int c = w.col4row[r];
    const int s = tr ? c : r, d = tr ? r : c;
    md[s] = d;
    ms[s] = sc[s * NP + d];
  }
}

// ------------------------------------------------------------------------------------------------
// Grouping: one workgroup per frame; lane 0 runs the order-dependent greedy assembly on LDS tables.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
paf_group_kernel(const float* __restrict__ peak_xy, const float* __restrict__ peak_val,
                 const int32_t* __restrict__ node_count, const int32_t* __restrict__ node_peaks,
                 int max_peaks, const int32_t* __restrict__ match_dst,
                 const float* __restrict__ match_score, const int32_t* __restrict__ edges,
                 const int32_t* __restrict__ sorted_edge_inds, int n_sorted, int E, int N, int NP,
                 float min_line_scores, int min_instance_peaks, int max_instances,
                 float* __restrict__ instance_peaks, float* __restrict__ instance_peak_vals,
                 float* __restrict__ instance_scores, int32_t* __restrict__ n_instances,
                 int32_t* __restrict__ status, int32_t* __restrict__ workspace) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // tables live in LDS when they fit, otherwise in the caller's global workspace (3*N*NP+1 ints per frame)
  int32_t* assign = workspace ? workspace + (size_t)blockIdx.x * (3 * (size_t)N * NP + 1)
                              : reinterpret_cast<int32_t*>(smem_raw);  // [N*NP] instance id or -1
  int32_t* order = assign + N * NP;                        // [N*NP] peak ids in dict-insertion order
  int32_t* remap = order + N * NP;                         // [N*NP + 1] id -> contiguous index
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int NN = N * NP;
  for (int i = tid; i < NN; i += nt) assign[i] = -1;
  // NaN-fill outputs (make_predicted_instances: np.full(..., nan))
  const float qnan = __builtin_nanf("");
  float* ip = instance_peaks + (size_t)b * max_instances * N * 2;
  float* iv = instance_peak_vals + (size_t)b * max_instances * N;
  float* is = instance_scores + (size_t)b * max_instances;
  for (int i = tid; i < max_instances * N * 2; i += nt) ip[i] = qnan;
  for (int i = tid; i < max_instances * N; i += nt) iv[i] = qnan;
  for (int i = tid; i < max_instances; i += nt) is[i] = qnan;
  __syncthreads();
  if (tid != 0) return;

  const int32_t* md = match_dst + (size_t)b * E * NP;
  const float* msc = match_score + (size_t)b * E * NP;
  const int32_t* ncnt = node_count + (size_t)b * N;
  int n_order = 0;

  // ---- assign_connections_to_instances (paf_grouping.py:799-914)
  for (int q = 0; q < n_sorted; ++q) {
    const int k = sorted_edge_inds[q];
    const int sn = edges[2 * k], dn = edges[2 * k + 1];
    const int n_src = ncnt[sn];
    for (int s = 0; s < n_src; ++s) {
      const int d = md[k * NP + s];
      if (d < 0) continue;
      if (!(msc[k * NP + s] >= min_line_scores)) continue;  // group_instances_sample :1067
      const int src_id = sn * NP + s, dst_id = dn * NP + d;
      const int si = assign[src_id], di = assign[dst_id];
      if (si < 0 && di < 0) {
        int mx = -1;
        for (int i = 0; i < NN; ++i) mx = max(mx, assign[i]);
        assign[src_id] = mx + 1;
        order[n_order++] = src_id;
        if (dst_id != src_id) {
          assign[dst_id] = mx + 1;
          order[n_order++] = dst_id;
        }
      } else if (si >= 0 && di < 0) {
        assign[dst_id] = si;
        order[n_order++] = dst_id;
      } else if (si >= 0 && di >= 0) {
        assign[dst_id] = si;
        // node sets AFTER the reassignment above, as the reference computes them
        bool intersect = false;
        for (int nd = 0; nd < N && !intersect; ++nd) {
          bool hs = false, hd = false;
          for (int p = 0; p < NP; ++p) {
            const int a = assign[nd * NP + p];
            hs |= (a == si);
            hd |= (a == di);
          }
          intersect = hs && hd;
        }
        if (!intersect)
          for (int i = 0; i < NN; ++i)
            if (assign[i] == di) assign[i] = si;
      }
      // (src unassigned, dst assigned): the reference has no branch for it -> nothing happens
    }
  }
  // ---- optional min_instance_peaks filter (:887-913)
  if (min_instance_peaks > 0) {
    for (int i = 0; i < NN; ++i) remap[i] = 0;
    for (int i = 0; i < NN; ++i)
      if (assign[i] >= 0) remap[assign[i]]++;
    for (int i = 0; i < NN; ++i)
      if (assign[i] >= 0 && remap[assign[i]] < min_instance_peaks) assign[i] = -1 - NN;  // dropped
  }
  // ---- make_predicted_instances (:917-981): np.unique -> contiguous ids in ascending id order
  for (int i = 0; i <= NN; ++i) remap[i] = 0;
  for (int i = 0; i < NN; ++i)
    if (assign[i] >= 0) remap[assign[i]] = 1;
  int n_inst = 0;
  for (int i = 0; i <= NN; ++i) {
    const int present = remap[i];
    remap[i] = n_inst;
    n_inst += present;
  }
  if (n_inst > max_instances) atomicOr(&status[b], SA_STATUS_INSTANCE_OVERFLOW);
  const int n_out = min(n_inst, max_instances);
  n_instances[b] = n_out;
  for (int i = 0; i < n_out; ++i) is[i] = 0.0f;
  for (int q = 0; q < n_sorted; ++q) {
    const int k = sorted_edge_inds[q];
    const int sn = edges[2 * k];
    const int n_src = ncnt[sn];
    for (int s = 0; s < n_src; ++s) {
      const int d = md[k * NP + s];
      if (d < 0 || !(msc[k * NP + s] >= min_line_scores)) continue;
      const int a = assign[sn * NP + s];
      if (a < 0) continue;
      const int ii = remap[a];
      if (ii < n_out) is[ii] = __fadd_rn(is[ii], msc[k * NP + s]);
    }
  }
  const int32_t* np_list = node_peaks + (size_t)b * N * NP;
  const float* xy = peak_xy + (size_t)b * max_peaks * 2;
  const float* pv = peak_val + (size_t)b * max_peaks;
  for (int o = 0; o < n_order; ++o) {  // dict iteration order: later entries overwrite
    const int id = order[o];
    const int a = assign[id];
    if (a < 0) continue;
    const int ii = remap[a];
    if (ii >= n_out) continue;
    const int nd = id / NP;
    const int pk = np_list[id];
    ip[((size_t)ii * N + nd) * 2 + 0] = xy[2 * pk];
    ip[((size_t)ii * N + nd) * 2 + 1] = xy[2 * pk + 1];
    iv[(size_t)ii * N + nd] = pv[pk];
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

int sa_abi_version(void) { return SA_ABI_VERSION; }

const char* sa_storage_dtype(void) { return SA_HALF_NAME; }

const char* sa_last_error(void) { return sa::err_buf(); }

int sa_device_info(int device, int* n_cu, int* lds_bytes, int* wave_size, char* arch, int arch_len) {
  hipDeviceProp_t p;
  SA_HIP_CHECK(hipGetDeviceProperties(&p, device));
  if (n_cu) *n_cu = p.multiProcessorCount;
  if (lds_bytes) *lds_bytes = (int)p.sharedMemPerBlock;
  if (wave_size) *wave_size = p.warpSize;
  if (arch && arch_len > 0) {
    strncpy(arch, p.gcnArchName, arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  return SA_OK;
}

size_t sa_find_local_peaks_workspace(int B, int max_peaks) {
  return (size_t)B * max_peaks * sizeof(uint32_t);
}

int sa_find_local_peaks(const float* cms, const float* offsets, int B, int H, int W, int C,
                        float threshold, int refinement, int patch_size, float xy_scale,
                        int max_peaks, float* peak_xy, float* peak_val, int32_t* peak_chan,
                        int32_t* peak_count, int32_t* status, void* workspace, size_t ws_bytes,
                        sa_stream_t stream) {
  SA_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0, "sa_find_local_peaks: bad shape %dx%dx%dx%d", B, H, W, C);
  SA_REQUIRE((uint64_t)H * W * C < 0xFFFFFFFFull, "sa_find_local_peaks: H*W*C exceeds 32-bit keys");
  SA_REQUIRE(max_peaks > 0 && max_peaks <= 16384, "sa_find_local_peaks: max_peaks %d out of range", max_peaks);
  SA_REQUIRE(refinement >= 0 && refinement <= 3, "sa_find_local_peaks: bad refinement %d", refinement);
  SA_REQUIRE(refinement != SA_REFINE_OFFSETS || offsets, "sa_find_local_peaks: offsets is NULL");
  SA_REQUIRE(refinement != SA_REFINE_INTEGRAL || (patch_size >= 1 && (patch_size & 1)),
             "sa_find_local_peaks: integral patch size must be odd, got %d", patch_size);
  if (ws_bytes < sa_find_local_peaks_workspace(B, max_peaks))
    return sa::fail(SA_ERR_WORKSPACE, "sa_find_local_peaks: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  uint32_t* keys = (uint32_t*)workspace;
  SA_HIP_CHECK(hipMemsetAsync(peak_count, 0, sizeof(int32_t) * B, st));
  const size_t plane = (size_t)H * W * C;
  const int vec4 = (plane % 4 == 0) && (((uintptr_t)cms) % 16 == 0);
  const size_t work = vec4 ? plane / 4 : plane;
  int gx = (int)((work + 255) / 256);
  if (gx > 2048) gx = 2048;
  hipLaunchKernelGGL(nms_scan_kernel, dim3(gx, B), dim3(256), 0, st, cms, H, W, C, threshold, max_peaks,
                     keys, peak_count, status, vec4);
  SA_LAUNCH_CHECK();
  int n2 = 1;
  while (n2 < max_peaks) n2 <<= 1;
  hipLaunchKernelGGL(peaks_sort_refine_kernel, dim3(B), dim3(256), n2 * sizeof(uint32_t), st, cms,
                     offsets, H, W, C, refinement, patch_size, xy_scale, max_peaks, keys, peak_count,
                     peak_xy, peak_val, peak_chan);
  SA_LAUNCH_CHECK();
  return SA_OK;
}

int sa_find_global_peaks(const float* cms, const float* offsets, int B, int H, int W, int C,
                         float threshold, int refinement, int patch_size, float xy_scale,
                         float* peak_xy, float* peak_val, sa_stream_t stream) {
  SA_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0, "sa_find_global_peaks: bad shape");
  SA_REQUIRE(refinement >= 0 && refinement <= 3, "sa_find_global_peaks: bad refinement %d", refinement);
  SA_REQUIRE(refinement != SA_REFINE_OFFSETS || offsets, "sa_find_global_peaks: offsets is NULL");
  hipLaunchKernelGGL(global_peaks_kernel, dim3(C, B), dim3(256), 0, (hipStream_t)stream, cms, offsets,
                     H, W, C, threshold, refinement, patch_size, xy_scale, peak_xy, peak_val);
  SA_LAUNCH_CHECK();
  return SA_OK;
}

int sa_crop_and_resize(const void* images, int is_u8, int H, int W, int C, const float* centres_xy,
                       const int32_t* sample_inds, int n, int crop, void* out, sa_stream_t stream) {
  SA_REQUIRE(H > 1 && W > 1 && C > 0 && crop > 0 && n >= 0, "sa_crop_and_resize: bad shape");
  if (n == 0) return SA_OK;
  const size_t total = (size_t)n * crop * crop;
  int g = (int)((total + 255) / 256);
  if (g > 4096) g = 4096;
  if (is_u8)
    hipLaunchKernelGGL((crop_and_resize_kernel<uint8_t>), dim3(g), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t*)images, H, W, C, centres_xy, sample_inds, n, crop, (uint8_t*)out);
  else
    hipLaunchKernelGGL((crop_and_resize_kernel<float>), dim3(g), dim3(256), 0, (hipStream_t)stream,
                       (const float*)images, H, W, C, centres_xy, sample_inds, n, crop, (float*)out);
  SA_LAUNCH_CHECK();
  return SA_OK;
}

int sa_paf_score(const float* pafs, int B, int Hp, int Wp, int E, const float* peak_xy,
                 const int32_t* peak_chan, const int32_t* peak_count, int max_peaks,
                 const int32_t* edges, int N, int n_points, float pafs_stride,
                 float max_edge_length, float dist_penalty_weight, int max_node_peaks,
                 int32_t* node_count, int32_t* node_peaks, float* line_scores, int32_t* status,
                 sa_stream_t stream) {
  SA_REQUIRE(B > 0 && E >= 0 && N > 0 && N <= MAXNODES, "sa_paf_score: bad B/E/N (%d/%d/%d)", B, E, N);
  SA_REQUIRE(max_node_peaks > 0 && max_node_peaks <= MAXNP, "sa_paf_score: max_node_peaks must be in [1,%d]", MAXNP);
  SA_REQUIRE(n_points >= 1, "sa_paf_score: n_points must be >= 1");
  const size_t lds = sizeof(int32_t) * ((size_t)N + (size_t)N * max_node_peaks);
  SA_REQUIRE(lds <= 64 * 1024, "sa_paf_score: N * max_node_peaks too large for the LDS node tables");
  hipLaunchKernelGGL(paf_score_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, pafs, Hp, Wp, E,
                     peak_xy, peak_chan, peak_count, max_peaks, edges, N, n_points, pafs_stride,
                     max_edge_length, dist_penalty_weight, max_node_peaks, node_count, node_peaks,
                     line_scores, status);
  SA_LAUNCH_CHECK();
  return SA_OK;
}

size_t sa_paf_workspace(int B, int E, int N, int max_node_peaks) {
  const size_t match = (size_t)B * E * sa::LsaWork::bytes(max_node_peaks);
  const size_t group = (size_t)B * (3 * (size_t)N * max_node_peaks + 1) * sizeof(int32_t);
  return (match > group ? match : group) + 64;
}

int sa_paf_match(const float* line_scores, const int32_t* node_count, const int32_t* edges, int B,
                 int E, int N, int max_node_peaks, int32_t* match_dst, float* match_score,
                 int32_t* status, void* workspace, size_t ws_bytes, sa_stream_t stream) {
  SA_REQUIRE(max_node_peaks > 0 && max_node_peaks <= MAXNP, "sa_paf_match: max_node_peaks must be in [1,%d]", MAXNP);
  if (B * E == 0) return SA_OK;
  if (!workspace || ws_bytes < (size_t)B * E * sa::LsaWork::bytes(max_node_peaks))
    return sa::fail(SA_ERR_WORKSPACE, "sa_paf_match: workspace too small (see sa_paf_workspace)");
  const int nb = (B * E + 63) / 64;
  hipLaunchKernelGGL(paf_match_kernel, dim3(nb), dim3(64), 0, (hipStream_t)stream, line_scores,
                     node_count, edges, B, E, N, max_node_peaks, match_dst, match_score, status,
                     (unsigned char*)workspace);
  SA_LAUNCH_CHECK();
  return SA_OK;
}

int sa_paf_group(const float* peak_xy, const float* peak_val, const int32_t* node_count,
                 const int32_t* node_peaks, int max_peaks, const int32_t* match_dst,
                 const float* match_score, const int32_t* edges, const int32_t* sorted_edge_inds,
                 int n_sorted, int B, int E, int N, int max_node_peaks, float min_line_scores,
                 int min_instance_peaks, int max_instances, float* instance_peaks,
                 float* instance_peak_vals, float* instance_scores, int32_t* n_instances,
                 int32_t* status, void* workspace, size_t ws_bytes, sa_stream_t stream) {
  SA_REQUIRE(N > 0 && N <= MAXNODES && max_node_peaks > 0 && max_node_peaks <= MAXNP, "sa_paf_group: bad N/max_node_peaks");
  SA_REQUIRE(max_instances > 0, "sa_paf_group: max_instances must be > 0");
  const size_t nn = (size_t)N * max_node_peaks;
  size_t lds = sizeof(int32_t) * (3 * nn + 1);
  int32_t* ws = nullptr;
  if (lds > 48 * 1024) {  // tables too large for LDS: use the global workspace
    if (!workspace || ws_bytes < (size_t)B * lds)
      return sa::fail(SA_ERR_WORKSPACE, "sa_paf_group: workspace too small (see sa_paf_workspace)");
    ws = (int32_t*)workspace;
    lds = 0;
  }
  hipLaunchKernelGGL(paf_group_kernel, dim3(B), dim3(64), lds, (hipStream_t)stream, peak_xy, peak_val,
                     node_count, node_peaks, max_peaks, match_dst, match_score, edges, sorted_edge_inds,
                     n_sorted, E, N, max_node_peaks, min_line_scores, min_instance_peaks, max_instances,
                     instance_peaks, instance_peak_vals, instance_scores, n_instances, status, ws);
  SA_LAUNCH_CHECK();
  return SA_OK;
}

int sa_lsa_host(const double* cost, int nr, int nc, int64_t* row_ind, int64_t* col_ind) {
  if (nr <= 0 || nc <= 0) return 0;
  if (nr > 4096 || nc > 4096) return sa::fail(SA_ERR_INVALID_ARG, "sa_lsa_host: matrix too large");
  for (long i = 0; i < (long)nr * nc; ++i)
    if (cost[i] != cost[i] || cost[i] == -__builtin_huge_val()) return -1;
  const int nmax = nr > nc ? nr : nc;
  std::vector<unsigned char> mem(sa::LsaWork::bytes(nmax));
  sa::LsaWork wk;
  wk.bind(mem.data(), nmax);
  sa::LsaWork* w = &wk;
  const bool tr = nc < nr;
  const int R = tr ? nc : nr, Cn = tr ? nr : nc;
  auto cf = [=](int i, int j) { return tr ? cost[(size_t)j * nc + i] : cost[(size_t)i * nc + j]; };
  if (!sa::lsa_solve(R, Cn, cf, *w)) return -1;
  if (!tr) {
    for (int i = 0; i < R; ++i) {
      row_ind[i] = i;
      col_ind[i] = w->col4row[i];
    }
  } else {
    // rows of the transposed problem are original columns; emit sorted by original row
    int n = 0;
    for (int c = 0; c < Cn; ++c) {
      const int r = w->row4col[c];  // transposed row (= original column) assigned to original row c
      if (r >= 0) {
        row_ind[n] = c;
        col_ind[n] = r;
        ++n;
      }
    }
  }
  return R;
}

}  // extern "C"
