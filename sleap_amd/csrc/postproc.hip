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

#if !defined(SA_POSTPROC_PAR)
#define SA_POSTPROC_PAR 1  // round 6: the fused post-processing kernel's memory-latency chains taken apart (0: the round 2-5 form, A/B)
#endif
#if !defined(SA_NMS_ILP_DEFAULT)
#define SA_NMS_ILP_DEFAULT 1
#endif
#if !defined(SA_GROUP_LDS_EDGES)
#define SA_GROUP_LDS_EDGES 1  // grouping walk: edge list and edge order read from LDS (0: from global memory as before, A/B)
#endif
#if !defined(SA_GROUP_COMPACT)
#define SA_GROUP_COMPACT 1  // grouping: the connections as a compact list built in parallel (0: cursor over the match tables, A/B)
#endif
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
template <int ILP>
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
    // ILP 16-byte loads per thread are requested before the first is looked at (a workgroup's ILP x 256 loads are ILP coalesced
    // 4-KiB rows): with one load per thread and iteration the launch moved 2.6 TB/s alone -- wave launch and one round trip per
    // 16 bytes, not bandwidth
    for (size_t i0 = (size_t)blockIdx.x * blockDim.x * ILP + threadIdx.x; i0 < n4; i0 += stride * ILP) {
      float4 qs[ILP];
#pragma unroll
      for (int u = 0; u < ILP; ++u) {
        const size_t iu = i0 + (size_t)u * blockDim.x;
        qs[u] = iu < n4 ? img4[iu] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      }
#pragma unroll
      for (int u = 0; u < ILP; ++u) {
      const size_t i = i0 + (size_t)u * blockDim.x;
      if (i >= n4) continue;
      const float4 q = qs[u];
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
// order, which every downstream index depends on) and refine each peak. `sk`: LDS, next_pow2(max_peaks) words.
__device__ void frame_sort_refine(int b, uint32_t* sk, const float* __restrict__ cms, const float* __restrict__ offsets, int H,
                                  int W, int C, int mode, int k, float xy_scale, int max_peaks,
                                  const uint32_t* __restrict__ keys, int32_t* __restrict__ counts,
                                  float* __restrict__ peak_xy, float* __restrict__ peak_val,
                                  int32_t* __restrict__ peak_chan, int32_t* __restrict__ scan_counts = nullptr,
                                  float* samp = nullptr, int samp_words = 0) {
  // `scan_counts` (optional): the NMS scan's atomic counters when they are not `counts` itself; they are handed back ZEROED
  // for the next call (no memset launch in front of the scan)
  const int tid = threadIdx.x, nt = blockDim.x;
  const int n = min(scan_counts ? scan_counts[b] : counts[b], max_peaks);
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
  if (mode == SA_REFINE_INTEGRAL && samp && samp_words >= k * k) {
    // Round 6 (this kernel at small batches is a chain of memory latencies, not work): the k x k bilinear
    // samples of every peak are independent -- (peak, i, j) work items spread over the workgroup fetch them at once into LDS,
    // then one thread per peak adds them up IN THE REFERENCE'S ORDER (row-major, one fp32 add at a time: the same bits as the
    // serial loop of refine_offset, whose 25 dependent round trips to memory were the longest stage of the kernel).
    const int kk = k * k, chunk = min(samp_words / kk, max(n, 1));
    const float half = (float)(k - 1) * 0.5f;
    for (int base = 0; base < n; base += chunk) {
      const int m = min(chunk, n - base);
      for (int t = tid; t < m * kk; t += nt) {
        const int pi = t / kk, r = t - pi * kk, i = r / k, j = r - i * k;
        const uint32_t e = sk[base + pi];
        const int c = (int)(e % C);
        const uint32_t p = e / C;
        const int x = (int)(p % W), y = (int)(p / W);
        const CropAxis ay = crop_axis(y, k, i, H), ax = crop_axis(x, k, j, W);
        samp[t] = crop_sample(img, W, C, c, ay, ax);
      }
      __syncthreads();
      for (int pi = tid; pi < m; pi += nt) {
        const uint32_t e = sk[base + pi];
        const int c = (int)(e % C);
        const uint32_t p = e / C;
        const int x = (int)(p % W), y = (int)(p / W);
        float z = 0.0f, sx = 0.0f, sy = 0.0f;
        for (int i = 0; i < k; ++i) {
          const float gy = __fsub_rn((float)i, half);
          for (int j = 0; j < k; ++j) {
            const float pv = samp[pi * kk + i * k + j];
            const float gx = __fsub_rn((float)j, half);
            z = __fadd_rn(z, pv);
            sx = __fadd_rn(sx, __fmul_rn(gx, pv));
            sy = __fadd_rn(sy, __fmul_rn(gy, pv));
          }
        }
        const float dx = __fdiv_rn(sx, z), dy = __fdiv_rn(sy, z);
        const size_t o = (size_t)b * max_peaks + base + pi;
        peak_xy[o * 2 + 0] = __fmul_rn(__fadd_rn((float)x, dx), xy_scale);
        peak_xy[o * 2 + 1] = __fmul_rn(__fadd_rn((float)y, dy), xy_scale);
        peak_val[o] = img[e];
        peak_chan[o] = c;
      }
      __syncthreads();
    }
  } else
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
  __syncthreads();  // every thread has read the counter
  if (tid == 0) {
    counts[b] = n;
    if (scan_counts) scan_counts[b] = 0;
  }
}

__global__ void __launch_bounds__(256)
peaks_sort_refine_kernel(const float* __restrict__ cms, const float* __restrict__ offsets, int H,
                         int W, int C, int mode, int k, float xy_scale, int max_peaks,
                         const uint32_t* __restrict__ keys, int32_t* __restrict__ counts,
                         float* __restrict__ peak_xy, float* __restrict__ peak_val,
                         int32_t* __restrict__ peak_chan, int32_t* __restrict__ scan_counts) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  frame_sort_refine(blockIdx.x, reinterpret_cast<uint32_t*>(smem_raw), cms, offsets, H, W, C, mode, k, xy_scale, max_peaks, keys,
                    counts, peak_xy, peak_val, peak_chan, scan_counts);
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
// Top-down glue kept on the device (no host round trip between the centroid model and the instance model):
//   select_centroids  CentroidCrop.call after find_local_peaks (inference.py:1822-1916): un-scale the peaks
//                     ((p / input_scale) + 0.5 when input_scale != 1, x precrop_resize), keep each frame's peaks in their
//                     order -- or, with max_instances < count, tf.math.top_k's (value descending, ties: lower index first)
//                     -- into K fixed slots per frame; crop offsets = centroid - crop_size / 2
//   finish_instance_peaks  FindInstancePeaks.call after find_global_peaks (:2150-2178): (p / input_scale) + 0.5, plus
//                     crop_offset / input_scale, NaN rows for empty slots
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
select_centroids_kernel(const float* __restrict__ peak_xy, const float* __restrict__ peak_val,
                        const int32_t* __restrict__ peak_count, int max_peaks, int K, int max_instances, float input_scale,
                        float precrop_resize, float half_crop, float* __restrict__ cent_xy, float* __restrict__ cent_val,
                        float* __restrict__ crop_centre, float* __restrict__ crop_offset, int32_t* __restrict__ n_valid,
                        int32_t* __restrict__ status) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int n = min(peak_count[b], max_peaks);
  const float* xy = peak_xy + (size_t)b * max_peaks * 2;
  const float* pv = peak_val + (size_t)b * max_peaks;
  const bool topk = max_instances >= 0 && max_instances < n;
  const int keep = topk ? max_instances : n;
  const float qnan = __builtin_nanf("");
  for (int k = lane; k < K; k += 64) {
    const size_t o = (size_t)b * K + k;
    cent_xy[2 * o] = cent_xy[2 * o + 1] = qnan;
    cent_val[o] = qnan;
    crop_centre[2 * o] = crop_centre[2 * o + 1] = -8.0f * half_crop - 16.0f;  // a finite centre far outside: an all-zero crop
    crop_offset[2 * o] = crop_offset[2 * o + 1] = qnan;
  }
  __syncthreads();
  for (int i = lane; i < n; i += 64) {
    int slot = i;
    if (topk) {  // rank in (value descending, index ascending): top_k's output order
      const float v = pv[i];
      int rank = 0;
      for (int j = 0; j < n; ++j) rank += (pv[j] > v) || (pv[j] == v && j < i);
      slot = rank;
    }
    if (slot >= keep || slot >= K) continue;
    float x = xy[2 * i], y = xy[2 * i + 1];
    if (input_scale != 1.0f) {
      x = __fadd_rn(__fdiv_rn(x, input_scale), 0.5f);
      y = __fadd_rn(__fdiv_rn(y, input_scale), 0.5f);
    }
    if (precrop_resize != 1.0f) {
      x = __fmul_rn(x, precrop_resize);
      y = __fmul_rn(y, precrop_resize);
    }
    const size_t o = (size_t)b * K + slot;
    cent_xy[2 * o] = x, cent_xy[2 * o + 1] = y;
    crop_centre[2 * o] = x, crop_centre[2 * o + 1] = y;
    crop_offset[2 * o] = __fsub_rn(x, half_crop), crop_offset[2 * o + 1] = __fsub_rn(y, half_crop);
    cent_val[o] = pv[i];
  }
  if (lane == 0) {
    n_valid[b] = min(keep, K);
    if (keep > K) atomicOr(&status[b], SA_STATUS_INSTANCE_OVERFLOW);
  }
}

__global__ void __launch_bounds__(256)
finish_instance_peaks_kernel(float* __restrict__ peaks, float* __restrict__ vals, const float* __restrict__ crop_offset,
                             const int32_t* __restrict__ n_valid, int B, int K, int N, float input_scale) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * K * N) return;
  const int slot = t / N, b = slot / K, k = slot % K;
  if (k >= n_valid[b]) {
    peaks[2 * (size_t)t] = peaks[2 * (size_t)t + 1] = vals[t] = __builtin_nanf("");
    return;
  }
  float x = peaks[2 * (size_t)t], y = peaks[2 * (size_t)t + 1];
  if (input_scale != 1.0f) {
    x = __fadd_rn(__fdiv_rn(x, input_scale), 0.5f);
    y = __fadd_rn(__fdiv_rn(y, input_scale), 0.5f);
  }
  if (crop_offset) {
    x = __fadd_rn(x, __fdiv_rn(crop_offset[2 * (size_t)slot], input_scale));
    y = __fadd_rn(y, __fdiv_rn(crop_offset[2 * (size_t)slot + 1], input_scale));
  }
  peaks[2 * (size_t)t] = x, peaks[2 * (size_t)t + 1] = y;
}

// ------------------------------------------------------------------------------------------------
// PAF scoring
// ------------------------------------------------------------------------------------------------

// ---- arithmetic of one candidate line, shared by the fused scoring kernel and the stand-alone entry points behind the
// reference's module-level functions (make_line_subs / get_paf_lines / score_paf_lines / compute_distance_penalty)
struct LineDir {
  float vx, vy, len, ux, uy, ddx, ddy;
};

__device__ __forceinline__ LineDir line_dir(float sx, float sy, float ex, float ey, int n_points) {
  LineDir d;
  // spatial vector, tf.norm = sqrt(sum(v*v)) (paf_grouping.py:378-383)
  d.vx = __fsub_rn(ex, sx);
  d.vy = __fsub_rn(ey, sy);
  d.len = __fsqrt_rn(__fadd_rn(__fmul_rn(d.vx, d.vx), __fmul_rn(d.vy, d.vy)));
  d.ux = __fdiv_rn(d.vx, d.len);
  d.uy = __fdiv_rn(d.vy, d.len);
  // tf.linspace: delta = (stop - start) / (n - 1); start + delta * i; exact end points
  const int steps = max(n_points - 1, 1);
  d.ddx = __fdiv_rn(d.vx, (float)steps);
  d.ddy = __fdiv_rn(d.vy, (float)steps);
  return d;
}

__device__ __forceinline__ void line_point(const LineDir& d, float sx, float sy, float ex, float ey, int i, int n_points,
                                           float& px, float& py) {
  if (i == 0) {
    px = sx;
    py = sy;
  } else if (i == n_points - 1) {
    px = ex;
    py = ey;
  } else {
    px = __fadd_rn(sx, __fmul_rn(d.ddx, (float)i));
    py = __fadd_rn(sy, __fmul_rn(d.ddy, (float)i));
  }
}

// one line point's projection of the PAF vector on the unit vector (paf_lines @ spatial_vecs, :386-388)
__device__ __forceinline__ float line_dot(const LineDir& d, float fx, float fy) {
  return __fadd_rn(__fmul_rn(fx, d.ux), __fmul_rn(fy, d.uy));
}

// compute_distance_penalty (paf_grouping.py:278-322)
__device__ __forceinline__ float distance_penalty(float len, float max_edge_length, float dist_penalty_weight) {
  return __fmul_rn(fminf(__fsub_rn(__fdiv_rn(max_edge_length, len), 1.0f), 0.0f), dist_penalty_weight);
}

// mean over the line points + distance penalty (:399-401)
__device__ __forceinline__ float line_total(const LineDir& d, float acc, int n_points, float max_edge_length,
                                            float dist_penalty_weight) {
  return __fadd_rn(__fdiv_rn(acc, (float)n_points), distance_penalty(d.len, max_edge_length, dist_penalty_weight));
}

// One workgroup per frame. Phase 1 buckets the frame's peaks by node type (stable, so the s-th
// entry of a node is its s-th peak in (y,x) order == tf.argsort/top_k order, paf_grouping.py:105).
// Phase 2 walks the dense (edge, src, dst) candidate space; every candidate samples n_points
// nearest-pixel PAF vectors along src->dst and averages their projection on the unit vector.
__device__ void frame_score(int b, unsigned char* smem_raw, const float* __restrict__ pafs, int Hp, int Wp, int E,
                            const float* __restrict__ peak_xy, const int32_t* __restrict__ peak_chan,
                            const int32_t* __restrict__ peak_count, int max_peaks,
                            const int32_t* __restrict__ edges, int N, int n_points, float pafs_stride,
                            float max_edge_length, float dist_penalty_weight, int NP,
                            int32_t* __restrict__ node_count, int32_t* __restrict__ node_peaks,
                            float* __restrict__ line_scores, int32_t* __restrict__ status, int s_ch_words = 0) {

  int32_t* s_cnt = reinterpret_cast<int32_t*>(smem_raw);  // [N]
  int32_t* s_list = s_cnt + N;                            // [N][NP]
  const int tid = threadIdx.x, nt = blockDim.x;
  const int n = min(peak_count[b], max_peaks);
  const int32_t* ch = peak_chan + (size_t)b * max_peaks;
  const float* xy = peak_xy + (size_t)b * max_peaks * 2;
  for (int i = tid; i < N; i += nt) s_cnt[i] = 0;
  // (round 6) the frame's peak channels once into LDS: the rank loop below read ch[0 .. i) from global memory per thread
  int32_t* s_ch = s_ch_words > 0 ? s_cnt + N + N * NP + E + 1 : nullptr;
  const bool ch_lds = s_ch && n <= s_ch_words;
  if (ch_lds)
    for (int i = tid; i < n; i += nt) s_ch[i] = ch[i];
  __syncthreads();
  const int32_t* chr = ch_lds ? s_ch : ch;
  for (int i = tid; i < n; i += nt) {
    const int c = chr[i];
    int rank = 0;
    for (int j = 0; j < i; ++j) rank += (chr[j] == c);
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
  // candidates of edge k occupy [s_off[k], s_off[k+1]) of a flat list (n_src * n_dst each): only real candidates are walked
  // (the dense (edge, src, dst) space is 50-100x larger at typical counts)
  int32_t* s_off = s_list + N * NP;  // [E + 1]
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int k = 0; k < E; ++k) {
      s_off[k] = run;
      run += s_cnt[edges[2 * k]] * s_cnt[edges[2 * k + 1]];
    }
    s_off[E] = run;
  }
  __syncthreads();
  const int total = s_off[E];
  bool oob = false;
  for (int idx = tid; idx < total; idx += nt) {
    int k = 0;
    while (idx >= s_off[k + 1]) ++k;
    const int sn = edges[2 * k], dn = edges[2 * k + 1];
    const int local = idx - s_off[k], nd_ = s_cnt[dn];
    const int s = local / nd_, d = local - s * nd_;
    const int ps = s_list[sn * NP + s], pd = s_list[dn * NP + d];
    const float sx = xy[2 * ps], sy = xy[2 * ps + 1], ex = xy[2 * pd], ey = xy[2 * pd + 1];
    const LineDir ld = line_dir(sx, sy, ex, ey, n_points);
    float acc = 0.0f;
    if (SA_POSTPROC_PAR && n_points == 10) {
      // (round 6) the reference's default count with a compile-time trip count: the ten gathers are independent and go out
      // together; the sum stays one fp32 add at a time in the line's order
      float fxs[10], fys[10];
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        float px, py;
        line_point(ld, sx, sy, ex, ey, i, 10, px, py);
        const float rx = rintf(__fdiv_rn(px, pafs_stride)), ry = rintf(__fdiv_rn(py, pafs_stride));
        const bool in = rx >= 0.0f && rx < (float)Wp && ry >= 0.0f && ry < (float)Hp;
        const float* q = paf + ((size_t)(in ? (int)ry : 0) * Wp + (in ? (int)rx : 0)) * PC + 2 * k;
        const float2 v = *reinterpret_cast<const float2*>(q);
        fxs[i] = in ? v.x : 0.0f;
        fys[i] = in ? v.y : 0.0f;
        oob = oob || !in;
      }
#pragma unroll
      for (int i = 0; i < 10; ++i) acc = __fadd_rn(acc, line_dot(ld, fxs[i], fys[i]));
    } else
    for (int i = 0; i < n_points; ++i) {
      float px, py;
      line_point(ld, sx, sy, ex, ey, i, n_points, px, py);
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
      acc = __fadd_rn(acc, line_dot(ld, fx, fy));
    }
    line_scores[(((size_t)b * E + k) * NP + s) * NP + d] = line_total(ld, acc, n_points, max_edge_length, dist_penalty_weight);
  }
  if (oob) atomicOr(&status[b], SA_STATUS_PAF_OOB);
}


__global__ void __launch_bounds__(256)
paf_score_kernel(const float* __restrict__ pafs, int Hp, int Wp, int E,
                 const float* __restrict__ peak_xy, const int32_t* __restrict__ peak_chan,
                 const int32_t* __restrict__ peak_count, int max_peaks,
                 const int32_t* __restrict__ edges, int N, int n_points, float pafs_stride,
                 float max_edge_length, float dist_penalty_weight, int NP,
                 int32_t* __restrict__ node_count, int32_t* __restrict__ node_peaks,
                 float* __restrict__ line_scores, int32_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  frame_score(blockIdx.x, smem_raw, pafs, Hp, Wp, E, peak_xy, peak_chan, peak_count, max_peaks, edges, N, n_points, pafs_stride,
              max_edge_length, dist_penalty_weight, NP, node_count, node_peaks, line_scores, status);
}

// ---- stand-alone pieces of the scoring stage (the reference exposes them as module-level functions and tests them one by
// one; the hot path uses the fused kernel above, which shares the arithmetic)
// make_line_subs (paf_grouping.py:145-222): subs [K][n_points][2][3] = [row, col, 2*edge + {0, 1}]
__global__ void __launch_bounds__(256)
paf_line_subs_kernel(const float* __restrict__ peaks_xy, const int32_t* __restrict__ edge_peak_inds,
                     const int32_t* __restrict__ edge_inds, int K, int n_points, float pafs_stride, int32_t* __restrict__ subs) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= K * n_points) return;
  const int k = t / n_points, i = t % n_points;
  const int ps = edge_peak_inds[2 * k], pd = edge_peak_inds[2 * k + 1];
  const float sx = peaks_xy[2 * ps], sy = peaks_xy[2 * ps + 1], ex = peaks_xy[2 * pd], ey = peaks_xy[2 * pd + 1];
  const LineDir ld = line_dir(sx, sy, ex, ey, n_points);
  float px, py;
  line_point(ld, sx, sy, ex, ey, i, n_points, px, py);
  const int col = (int)rintf(__fdiv_rn(px, pafs_stride)), row = (int)rintf(__fdiv_rn(py, pafs_stride));  // tf.round: half to even
  int32_t* o = subs + (size_t)t * 6;
  o[0] = row, o[1] = col, o[2] = 2 * edge_inds[k];
  o[3] = row, o[4] = col, o[5] = 2 * edge_inds[k] + 1;
}

// tf.gather_nd(pafs_sample, line_subs) (get_paf_lines, :225-275); out-of-range subscripts read 0 and set the flag
__global__ void __launch_bounds__(256)
gather_nd3_kernel(const float* __restrict__ src, int H, int W, int C, const int32_t* __restrict__ subs, int n,
                  float* __restrict__ out, int32_t* __restrict__ status) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int r = subs[3 * t], c = subs[3 * t + 1], ch = subs[3 * t + 2];
  float v = 0.0f;
  if (r >= 0 && r < H && c >= 0 && c < W && ch >= 0 && ch < C)
    v = src[((size_t)r * W + c) * C + ch];
  else if (status)
    atomicOr(status, SA_STATUS_PAF_OOB);
  out[t] = v;
}

// score_paf_lines (:325-403) on already gathered lines [K][n_points][2]
__global__ void __launch_bounds__(256)
paf_line_scores_kernel(const float* __restrict__ paf_lines, const float* __restrict__ peaks_xy,
                       const int32_t* __restrict__ edge_peak_inds, int K, int n_points, float max_edge_length,
                       float dist_penalty_weight, float* __restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  const int ps = edge_peak_inds[2 * k], pd = edge_peak_inds[2 * k + 1];
  const LineDir ld = line_dir(peaks_xy[2 * ps], peaks_xy[2 * ps + 1], peaks_xy[2 * pd], peaks_xy[2 * pd + 1], n_points);
  float acc = 0.0f;
  for (int i = 0; i < n_points; ++i)
    acc = __fadd_rn(acc, line_dot(ld, paf_lines[((size_t)k * n_points + i) * 2], paf_lines[((size_t)k * n_points + i) * 2 + 1]));
  out[k] = line_total(ld, acc, n_points, max_edge_length, dist_penalty_weight);
}

__global__ void __launch_bounds__(256)
distance_penalty_kernel(const float* __restrict__ lengths, int n, float max_edge_length, float dist_penalty_weight,
                        float* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) out[t] = distance_penalty(lengths[t], max_edge_length, dist_penalty_weight);
}

// ------------------------------------------------------------------------------------------------
// Matching: one WAVEFRONT per (frame, edge) runs the rectangular LSA on its (n_src x n_dst) block (csrc/lsa.h:
// lsa_solve_wave -- column scans, dual updates and initialisation spread over the 64 lanes, SciPy's tie rule kept by an
// order-aware reduction; work arrays in LDS).
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

__host__ __device__ inline size_t score_lds_bytes(int N, int NP, int E) { return sizeof(int32_t) * ((size_t)N + (size_t)N * NP + E + 1); }
__host__ __device__ inline size_t match_lds_bytes(int NP) { return (sa::LsaWork::bytes(NP) + 15) & ~(size_t)15; }

// one (frame b, edge k) by the calling wave; `mem`: match_lds_bytes(NP) of LDS owned by this wave
__device__ void edge_match_wave(int b, int k, unsigned char* mem, const float* __restrict__ line_scores,
                                const int32_t* __restrict__ node_count, const int32_t* __restrict__ edges, int E, int N, int NP,
                                int32_t* __restrict__ match_dst, float* __restrict__ match_score, int32_t* __restrict__ status) {
  const int lane = threadIdx.x & 63;
  const size_t t = (size_t)b * E + k;
  const int n_src = node_count[(size_t)b * N + edges[2 * k]];
  const int n_dst = node_count[(size_t)b * N + edges[2 * k + 1]];
  int32_t* md = match_dst + t * NP;
  float* ms = match_score + t * NP;
  for (int i = lane; i < NP; i += 64) {
    md[i] = -1;
    ms[i] = __builtin_nanf("");
  }
  if (n_src == 0 || n_dst == 0) return;  // wave-uniform
  const float* sc = line_scores + t * NP * NP;
  // SciPy rejects NaN / -inf costs ("matrix contains invalid numeric entries"); NaN scores were
  // mapped to +inf, so only a +inf score (cost -inf) is invalid.
  bool bad = false;
  for (int idx = lane; idx < n_src * n_dst; idx += 64) bad |= (sc[(idx / n_dst) * NP + idx % n_dst] == __builtin_huge_valf());
  if (__ballot(bad)) {
    if (lane == 0) atomicOr(&status[b], SA_STATUS_LSA_INFEASIBLE);
    return;
  }
  sa::LsaWork w;
  const bool tr = n_dst < n_src;
  const int nr = tr ? n_dst : n_src, nc = tr ? n_src : n_dst;
  w.bind(mem, nc);
  CostView cv{sc, NP, tr};
  if (!sa::lsa_solve_wave(nr, nc, cv, w)) {
    if (lane == 0) atomicOr(&status[b], SA_STATUS_LSA_INFEASIBLE);
    return;
  }
  SA_WAVE_SYNC();
  for (int r = lane; r < nr; r += 64) {
    const int c = w.col4row[r];
    const int s_ = tr ? c : r, d_ = tr ? r : c;
    md[s_] = d_;
    ms[s_] = sc[s_ * NP + d_];
  }
}

__global__ void __launch_bounds__(256)
paf_match_kernel(const float* __restrict__ line_scores, const int32_t* __restrict__ node_count,
                 const int32_t* __restrict__ edges, int B, int E, int N, int NP,
                 int32_t* __restrict__ match_dst, float* __restrict__ match_score,
                 int32_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
  const int t = blockIdx.x * waves + wave;
  if (t >= B * E) return;
  edge_match_wave(t / E, t % E, smem_raw + (size_t)wave * match_lds_bytes(NP), line_scores, node_count, edges, E, N, NP, match_dst,
                  match_score, status);
}

// ------------------------------------------------------------------------------------------------
// Grouping: the order-dependent greedy assembly of one frame (assign_connections_to_instances + make_predicted_instances,
// paf_grouping.py:799-981). The walk over the connections is inherently sequential, but every step of it -- highest
// instance id, node-set intersection, relabelling a merged instance -- is a scan of the (node, peak) table: ONE WAVE runs the
// walk with those scans spread over its 64 lanes (tables in LDS). `frame_group_seq` is the single-lane form kept for tables
// that do not fit in LDS (global workspace).
// Connections come either from the match tables (hot path: edges in `sorted_edge_inds` order, sources ascending) or from an
// explicit list (conn_*: edge, src peak, dst peak, score in processing order; the dictionary form of the reference's
// assign_connections_to_instances). `assign_out` (optional) receives the raw instance id of every (node, peak) slot.
// ------------------------------------------------------------------------------------------------
struct GroupIn {
  const float* peak_xy;
  const float* peak_val;
  const int32_t* node_count;
  const int32_t* node_peaks;
  int max_peaks;
  const int32_t* match_dst;
  const float* match_score;
  const int32_t* edges;
  const int32_t* sorted_edge_inds;
  int n_sorted, E, N, NP;
  float min_line_scores;
  int min_instance_peaks, max_instances;
  float* instance_peaks;
  float* instance_peak_vals;
  float* instance_scores;
  int32_t* n_instances;
  int32_t* status;
  // explicit connection list (per frame: conn_count[b] entries of stride conn_stride) or nullptr
  const int32_t* conn_edge;
  const int32_t* conn_src;
  const int32_t* conn_dst;
  const float* conn_score;
  const int32_t* conn_count;
  int conn_stride;
  int32_t* assign_out;  // [B][N*NP] or nullptr
  // given assignments instead of the greedy walk (make_predicted_instances on a caller's dictionary): assign_in [B][N*NP]
  // (-1 = not assigned) and the dictionary's key order order_in [B][N*NP] (order_count[b] slot ids), or nullptr
  const int32_t* assign_in;
  const int32_t* order_in;
  const int32_t* order_count;
};

// assign, order, remap[+1], cell_last, + the frame's match tables and node counts staged for the sequential walk
__host__ __device__ inline size_t group_lds_bytes(int N, int NP, int max_instances, int E) {
  return sizeof(int32_t) * (3 * (size_t)N * NP + 1 + (size_t)max_instances * N + 2 * (size_t)E * NP + N + 3 * (size_t)E + 3 * (size_t)E * NP);  // (+ edges, edge order, the compact connection list)
}

__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
  return v;
}

// connection q of the frame in processing order -> (k, s, d, score); returns false when the walk is over. The table form is
// walked with a cursor (qe, s).
struct ConnCursor {
  int qe, s, q;
};

// `md`, `msc`, `ncnt`: this frame's match tables / node counts (global, or their LDS copies in the wave form: the walk is a
// chain of dependent reads, ~0.5 us each from global memory, ~30 ns from LDS)
__device__ __forceinline__ bool next_conn(const GroupIn& g, int b, const int32_t* md, const float* msc, const int32_t* ncnt,
                                          ConnCursor& c, int& k, int& s, int& d, float& sc) {
  if (g.conn_edge) {
    const int n = g.conn_count[b];
    if (c.q >= n) return false;
    const size_t o = (size_t)b * g.conn_stride + c.q++;
    k = g.conn_edge[o], s = g.conn_src[o], d = g.conn_dst[o], sc = g.conn_score[o];
    return true;
  }
  while (c.qe < g.n_sorted) {
    const int kk = g.sorted_edge_inds[c.qe];
    const int n_src = ncnt[g.edges[2 * kk]];
    while (c.s < n_src) {
      const int ss = c.s++;
      const int dd = md[kk * g.NP + ss];
      if (dd < 0) continue;
      k = kk, s = ss, d = dd, sc = msc[kk * g.NP + ss];
      return true;
    }
    c.s = 0;
    ++c.qe;
  }
  return false;
}

__device__ void frame_group_wave(const GroupIn& g_in, int b, int32_t* assign, int32_t* order, int32_t* remap, int32_t* cell_last) {
  const int lane = threadIdx.x & 63;
  const int N = g_in.N, NP = g_in.NP, NN = N * NP;
  // stage the tables the sequential walk reads (match_dst / match_score / node_count of this frame) in LDS
  int32_t* md = cell_last + g_in.max_instances * N;
  float* msc = reinterpret_cast<float*>(md + g_in.E * NP);
  int32_t* ncnt = md + 2 * g_in.E * NP;
  // ... and, since round 6, the skeleton's edge list and edge order: the walk below looked both up in GLOBAL memory on every one
  // of its ~50 strictly sequential steps (two or three dependent L2 round trips per connection: 75 of this kernel's 110 us at
  // 4 animals x 13 nodes, measured with s_memtime stamps), and so did every lane of the instance-score loop
  int32_t* s_edges = ncnt + N;
  int32_t* s_sorted = s_edges + 2 * g_in.E;
  for (int i = lane; i < 2 * g_in.E; i += 64) s_edges[i] = g_in.edges[i];
  for (int i = lane; i < g_in.n_sorted; i += 64) s_sorted[i] = g_in.sorted_edge_inds[i];
  GroupIn g = g_in;
  if (SA_GROUP_LDS_EDGES) {
    g.edges = s_edges;
    if (g_in.sorted_edge_inds) g.sorted_edge_inds = s_sorted;
  }
  if (!g.conn_edge) {
    for (int i = lane; i < g.E * NP; i += 64) {
      md[i] = g.match_dst[(size_t)b * g.E * NP + i];
      msc[i] = g.match_score[(size_t)b * g.E * NP + i];
    }
  }
  for (int i = lane; i < N; i += 64) ncnt[i] = g.node_count[(size_t)b * N + i];
  for (int i = lane; i < NN; i += 64) assign[i] = g.assign_in ? g.assign_in[(size_t)b * NN + i] : -1;
  int n_order = 0;
  if (g.assign_in) {
    n_order = g.order_count[b];
    for (int i = lane; i < n_order; i += 64) order[i] = g.order_in[(size_t)b * NN + i];
  }
  SA_WAVE_SYNC();
  // Round 6: the connections of the match tables as ONE compact list in processing order (sorted edges, sources ascending,
  // unmatched sources and scores below min_line_scores left out), built by the wave in parallel: (source slot | destination slot
  // << 16, score). Walking the tables with a cursor cost every step of the greedy walk -- and every step of every lane of the
  // instance-score loop -- about eight DEPENDENT LDS reads (edge order -> edge -> count -> match -> ...): 2-3 k cycles per
  // connection, two thirds of this kernel (s_memtime stamps, tools/pp_stamp_probe.py). Same connections, same order, same sums.
  int32_t* c_pair = s_sorted + g_in.E;
  float* c_sc = reinterpret_cast<float*>(c_pair + g_in.E * NP);
  int32_t* c_inst = reinterpret_cast<int32_t*>(c_sc + g_in.E * NP);
  const bool compact = SA_GROUP_COMPACT && !g.conn_edge && !g.assign_in && NN <= 65535;
  int T = 0;
  if (compact) {
    for (int base = 0; base < g.n_sorted; base += 64) {
      const int e = base + lane;
      int cnt = 0, kk = 0, n_src = 0, sn = 0, dn = 0;
      if (e < g.n_sorted) {
        kk = s_sorted[e];
        sn = s_edges[2 * kk], dn = s_edges[2 * kk + 1];
        n_src = ncnt[sn];
        for (int ss = 0; ss < n_src; ++ss) cnt += (md[kk * NP + ss] >= 0 && msc[kk * NP + ss] >= g.min_line_scores) ? 1 : 0;
      }
      int incl = cnt;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
      }
      const int total = __shfl(incl, 63, 64);
      int pos = T + incl - cnt;
      for (int ss = 0; ss < n_src; ++ss) {
        const int dd = md[kk * NP + ss];
        const float scv = msc[kk * NP + ss];
        if (dd >= 0 && scv >= g.min_line_scores) {
          c_pair[pos] = (sn * NP + ss) | ((dn * NP + dd) << 16);
          c_sc[pos] = scv;
          ++pos;
        }
      }
      T += total;
    }
    SA_WAVE_SYNC();
  }
  // ---- assign_connections_to_instances (paf_grouping.py:799-914)
  ConnCursor cur = {0, 0, 0};
  int k = 0, s = 0, d = 0, jc = 0;
  float sc = 0.0f;
  while (!g.assign_in && (compact ? jc < T : next_conn(g, b, md, msc, ncnt, cur, k, s, d, sc))) {
    int src_id, dst_id;
    if (compact) {
      const unsigned pr = (unsigned)c_pair[jc++];
      src_id = (int)(pr & 0xFFFFu), dst_id = (int)(pr >> 16);
    } else {
      if (!(sc >= g.min_line_scores)) continue;  // group_instances_sample :1067
      src_id = g.edges[2 * k] * NP + s, dst_id = g.edges[2 * k + 1] * NP + d;
    }
    const int si = assign[src_id], di = assign[dst_id];
    SA_WAVE_SYNC();
    if (si < 0 && di < 0) {
      int mx = -1;
      for (int i = lane; i < NN; i += 64) mx = max(mx, assign[i]);
      mx = wave_max_i32(mx);
      SA_WAVE_SYNC();
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
      SA_WAVE_SYNC();
      // node sets AFTER the reassignment above, as the reference computes them
      bool inter = false;
      for (int nd = lane; nd < N; nd += 64) {
        bool hs = false, hd = false;
        const int cnt = ncnt[nd];
        for (int p = 0; p < cnt; ++p) {
          const int a = assign[nd * NP + p];
          hs |= (a == si);
          hd |= (a == di);
        }
        inter |= hs && hd;
      }
      if (!__ballot(inter)) {
        SA_WAVE_SYNC();
        for (int i = lane; i < NN; i += 64)
          if (assign[i] == di) assign[i] = si;
      }
    }
    // (src unassigned, dst assigned): the reference has no branch for it -> nothing happens
    SA_WAVE_SYNC();
  }
  // ---- optional min_instance_peaks filter (:887-913)
  if (g.min_instance_peaks > 0 && !g.assign_in) {
    for (int i = lane; i < NN; i += 64) remap[i] = 0;
    SA_WAVE_SYNC();
    for (int i = lane; i < NN; i += 64)
      if (assign[i] >= 0) atomicAdd(&remap[assign[i]], 1);
    SA_WAVE_SYNC();
    for (int i = lane; i < NN; i += 64)
      if (assign[i] >= 0 && remap[assign[i]] < g.min_instance_peaks) assign[i] = -1 - NN;  // dropped
    SA_WAVE_SYNC();
  }
  if (g.assign_out)
    for (int i = lane; i < NN; i += 64) g.assign_out[(size_t)b * NN + i] = assign[i];
  // ---- make_predicted_instances (:917-981): np.unique -> contiguous ids in ascending id order
  for (int i = lane; i <= NN; i += 64) remap[i] = 0;
  SA_WAVE_SYNC();
  for (int i = lane; i < NN; i += 64)
    if (assign[i] >= 0) remap[assign[i]] = 1;
  SA_WAVE_SYNC();
  int n_inst;
  {  // exclusive prefix sum of the presence flags: a contiguous chunk per lane + a wave scan of the chunk totals
    const int chunk = (NN + 1 + 63) / 64, lo = lane * chunk, hi = min(lo + chunk, NN + 1);
    int cnt = 0;
    for (int i = lo; i < hi; ++i) cnt += remap[i];
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int v = __shfl_up(incl, off, 64);
      if (lane >= off) incl += v;
    }
    n_inst = __shfl(incl, 63, 64);
    int run = incl - cnt;
    for (int i = lo; i < hi; ++i) {
      const int present = remap[i];
      remap[i] = run;
      run += present;
    }
  }
  SA_WAVE_SYNC();
  if (n_inst > g.max_instances && lane == 0) atomicOr(&g.status[b], SA_STATUS_INSTANCE_OVERFLOW);
  const int n_out = min(n_inst, g.max_instances);
  if (lane == 0) g.n_instances[b] = n_out;
  float* ip = g.instance_peaks + (size_t)b * g.max_instances * N * 2;
  float* iv = g.instance_peak_vals + (size_t)b * g.max_instances * N;
  float* is = g.instance_scores + (size_t)b * g.max_instances;
  // instance score = sum of its matched edge scores IN CONNECTION ORDER (fp32): one lane per instance walks the list
  if (compact) {  // the output instance of every connection once, in parallel; then one lane per instance adds its scores in order
    for (int j = lane; j < T; j += 64) {
      const int a = assign[(unsigned)c_pair[j] & 0xFFFFu];
      c_inst[j] = a >= 0 ? remap[a] : -1;
    }
    SA_WAVE_SYNC();
    for (int ii = lane; ii < n_out; ii += 64) {
      float acc = 0.0f;
      for (int j = 0; j < T; ++j)
        if (c_inst[j] == ii) acc = __fadd_rn(acc, c_sc[j]);
      is[ii] = acc;
    }
  } else
  for (int ii = lane; ii < n_out; ii += 64) {
    float acc = 0.0f;
    ConnCursor c2 = {0, 0, 0};
    int k2, s2, d2;
    float sc2;
    while (next_conn(g, b, md, msc, ncnt, c2, k2, s2, d2, sc2)) {
      if (!(sc2 >= g.min_line_scores)) continue;
      const int a = assign[g.edges[2 * k2] * NP + s2];
      if (a >= 0 && remap[a] == ii) acc = __fadd_rn(acc, sc2);
    }
    is[ii] = acc;
  }
  // peaks: dict iteration order, later entries of the same (instance, node) cell overwrite -> the last one wins
  for (int i = lane; i < n_out * N; i += 64) cell_last[i] = -1;
  SA_WAVE_SYNC();
  for (int o = lane; o < n_order; o += 64) {
    const int id = order[o], a = assign[id];
    if (a < 0) continue;
    const int ii = remap[a];
    if (ii < n_out) atomicMax(&cell_last[ii * N + id / NP], o);
  }
  SA_WAVE_SYNC();
  const int32_t* np_list = g.node_peaks + (size_t)b * N * NP;
  const float* xy = g.peak_xy + (size_t)b * g.max_peaks * 2;
  const float* pv = g.peak_val + (size_t)b * g.max_peaks;
  for (int cell = lane; cell < n_out * N; cell += 64) {
    const int o = cell_last[cell];
    if (o < 0) continue;
    const int pk = np_list[order[o]];
    ip[(size_t)cell * 2 + 0] = xy[2 * pk];
    ip[(size_t)cell * 2 + 1] = xy[2 * pk + 1];
    iv[cell] = pv[pk];
  }
}

// single-lane form (tables in a global workspace when they exceed LDS)
__device__ void frame_group_seq(const GroupIn& g, int b, int32_t* assign, int32_t* order, int32_t* remap) {
  const int N = g.N, NP = g.NP, NN = N * NP;
  const int32_t* ncnt = g.node_count + (size_t)b * N;
  const int32_t* md = g.match_dst ? g.match_dst + (size_t)b * g.E * NP : nullptr;
  const float* msc = g.match_score ? g.match_score + (size_t)b * g.E * NP : nullptr;
  for (int i = 0; i < NN; ++i) assign[i] = g.assign_in ? g.assign_in[(size_t)b * NN + i] : -1;
  int n_order = 0;
  if (g.assign_in) {
    n_order = g.order_count[b];
    for (int i = 0; i < n_order; ++i) order[i] = g.order_in[(size_t)b * NN + i];
  }
  ConnCursor cur = {0, 0, 0};
  int k, s, d;
  float sc;
  while (!g.assign_in && next_conn(g, b, md, msc, ncnt, cur, k, s, d, sc)) {
    if (!(sc >= g.min_line_scores)) continue;
    const int sn = g.edges[2 * k], dn = g.edges[2 * k + 1];
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
      bool intersect = false;
      for (int nd = 0; nd < N && !intersect; ++nd) {
        bool hs = false, hd = false;
        for (int p = 0; p < ncnt[nd]; ++p) {
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
  }
  if (g.min_instance_peaks > 0 && !g.assign_in) {
    for (int i = 0; i < NN; ++i) remap[i] = 0;
    for (int i = 0; i < NN; ++i)
      if (assign[i] >= 0) remap[assign[i]]++;
    for (int i = 0; i < NN; ++i)
      if (assign[i] >= 0 && remap[assign[i]] < g.min_instance_peaks) assign[i] = -1 - NN;  // dropped
  }
  if (g.assign_out)
    for (int i = 0; i < NN; ++i) g.assign_out[(size_t)b * NN + i] = assign[i];
  for (int i = 0; i <= NN; ++i) remap[i] = 0;
  for (int i = 0; i < NN; ++i)
    if (assign[i] >= 0) remap[assign[i]] = 1;
  int n_inst = 0;
  for (int i = 0; i <= NN; ++i) {
    const int present = remap[i];
    remap[i] = n_inst;
    n_inst += present;
  }
  if (n_inst > g.max_instances) atomicOr(&g.status[b], SA_STATUS_INSTANCE_OVERFLOW);
  const int n_out = min(n_inst, g.max_instances);
  g.n_instances[b] = n_out;
  float* ip = g.instance_peaks + (size_t)b * g.max_instances * N * 2;
  float* iv = g.instance_peak_vals + (size_t)b * g.max_instances * N;
  float* is = g.instance_scores + (size_t)b * g.max_instances;
  for (int i = 0; i < n_out; ++i) is[i] = 0.0f;
  ConnCursor c2 = {0, 0, 0};
  while (next_conn(g, b, md, msc, ncnt, c2, k, s, d, sc)) {
    if (!(sc >= g.min_line_scores)) continue;
    const int a = assign[g.edges[2 * k] * NP + s];
    if (a < 0) continue;
    const int ii = remap[a];
    if (ii < n_out) is[ii] = __fadd_rn(is[ii], sc);
  }
  const int32_t* np_list = g.node_peaks + (size_t)b * N * NP;
  const float* xy = g.peak_xy + (size_t)b * g.max_peaks * 2;
  const float* pv = g.peak_val + (size_t)b * g.max_peaks;
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

// NaN-fill of a frame's outputs (make_predicted_instances: np.full(..., nan)) by the whole workgroup
__device__ void frame_group_fill(const GroupIn& g, int b) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const float qnan = __builtin_nanf("");
  float* ip = g.instance_peaks + (size_t)b * g.max_instances * g.N * 2;
  float* iv = g.instance_peak_vals + (size_t)b * g.max_instances * g.N;
  float* is = g.instance_scores + (size_t)b * g.max_instances;
  for (int i = tid; i < g.max_instances * g.N * 2; i += nt) ip[i] = qnan;
  for (int i = tid; i < g.max_instances * g.N; i += nt) iv[i] = qnan;
  for (int i = tid; i < g.max_instances; i += nt) is[i] = qnan;
}

__global__ void __launch_bounds__(64)
paf_group_kernel(const GroupIn g, int32_t* __restrict__ workspace) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int b = blockIdx.x, NN = g.N * g.NP;
  frame_group_fill(g, b);
  __syncthreads();
  if (workspace) {  // tables too large for LDS: single lane on the global workspace
    int32_t* assign = workspace + (size_t)b * (3 * (size_t)NN + 1);
    if (threadIdx.x == 0) frame_group_seq(g, b, assign, assign + NN, assign + 2 * NN);
    return;
  }
  int32_t* assign = reinterpret_cast<int32_t*>(smem_raw);
  frame_group_wave(g, b, assign, assign + NN, assign + 2 * NN, assign + 3 * NN + 1);
}

// ------------------------------------------------------------------------------------------------
// The whole per-frame post-processing in ONE launch (one workgroup of 4 waves per frame): sort + refine the NMS survivors,
// bucket + score the candidate connections, match every edge (one wave per edge, 4 at a time), assemble the instances
// (wave 0). The stages hand their tables over through global memory (the caller's buffers: they are outputs of the ABI
// anyway) with workgroup barriers in between; LDS is re-used stage by stage. Replaces four dependent launches
// (~20 us of dispatch + drain each at small batch sizes).
// ------------------------------------------------------------------------------------------------
#if defined(SA_POSTPROC_STAMP)
__device__ unsigned long long pp_stamp_dev[8];
#endif
struct FusedIn {
  const float* cms;
  const float* offsets;
  int H, W, C, mode, patch;
  float xy_scale;
  int max_peaks;
  const uint32_t* keys;
  int32_t* scan_counts;
  int32_t* peak_count;
  float* peak_xy;
  float* peak_val;
  int32_t* peak_chan;
  const float* pafs;
  int Hp, Wp, n_points;
  float pafs_stride, max_edge_length, dist_penalty_weight;
  int32_t* node_count;
  int32_t* node_peaks;
  float* line_scores;
  int32_t* match_dst;
  float* match_score;
  int sort_words, samp_words, ch_words;  // LDS behind the sort keys (sample buffer of the refinement) / behind the scoring tables
};

__global__ void __launch_bounds__(1024)
bottomup_postproc_kernel(const FusedIn f, const GroupIn g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int b = blockIdx.x;
  // SA_POSTPROC_STAMP (an instrumented A/B build, tools/pp_stamp_probe.py -- never the product library): s_memtime at the stage
  // boundaries, summed over the workgroups
#if defined(SA_POSTPROC_STAMP)
  unsigned long long ts[6];
#define PP_STAMP(i)                          \
  do {                                       \
    __syncthreads();                         \
    ts[i] = __builtin_amdgcn_s_memtime();    \
  } while (0)
#else
#define PP_STAMP(i)
#endif
  PP_STAMP(0);
  frame_sort_refine(b, reinterpret_cast<uint32_t*>(smem_raw), f.cms, f.offsets, f.H, f.W, f.C, f.mode, f.patch, f.xy_scale,
                    f.max_peaks, f.keys, f.peak_count, f.peak_xy, f.peak_val, f.peak_chan, f.scan_counts,
                    reinterpret_cast<float*>(smem_raw) + f.sort_words, f.samp_words);
  __syncthreads();  // (workgroup-scope release/acquire of the global tables written above)
  PP_STAMP(1);
  frame_score(b, smem_raw, f.pafs, f.Hp, f.Wp, g.E, f.peak_xy, f.peak_chan, f.peak_count, f.max_peaks, g.edges, g.N, f.n_points,
              f.pafs_stride, f.max_edge_length, f.dist_penalty_weight, g.NP, f.node_count, f.node_peaks, f.line_scores, g.status,
              f.ch_words);
  PP_STAMP(2);
  frame_group_fill(g, b);
  __syncthreads();
  PP_STAMP(3);
  const int wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
  for (int k = wave; k < g.E; k += n_waves)
    edge_match_wave(b, k, smem_raw + (size_t)wave * match_lds_bytes(g.NP), f.line_scores, f.node_count, g.edges, g.E, g.N, g.NP,
                    f.match_dst, f.match_score, g.status);
  __syncthreads();
  PP_STAMP(4);
  if (wave == 0) {
    const int NN = g.N * g.NP;
    int32_t* assign = reinterpret_cast<int32_t*>(smem_raw);
    frame_group_wave(g, b, assign, assign + NN, assign + 2 * NN, assign + 3 * NN + 1);
  }
  PP_STAMP(5);
#if defined(SA_POSTPROC_STAMP)
  if (threadIdx.x == 0) {
    for (int i = 0; i < 5; ++i) atomicAdd(&pp_stamp_dev[i], ts[i + 1] - ts[i]);
    atomicAdd(&pp_stamp_dev[5], 1ull);
  }
#endif
}

}  // namespace

// the scan's launch: ILP loads per thread (SA_NMS_ILP = 1 / 2 / 4 / 8 at run time for A/B; default below), at most SA_NMS_GX
// workgroups per frame
static void launch_nms_scan(hipStream_t st, const float* cms, int B, int H, int W, int C, float threshold, int max_peaks, uint32_t* keys,
                            int32_t* counts, int32_t* status) {
  static const int ilp = getenv("SA_NMS_ILP") ? atoi(getenv("SA_NMS_ILP")) : SA_NMS_ILP_DEFAULT;
  static const int gx_cap = getenv("SA_NMS_GX") ? atoi(getenv("SA_NMS_GX")) : 2048;
  const size_t plane = (size_t)H * W * C;
  const int vec4 = (plane % 4 == 0) && (((uintptr_t)cms) % 16 == 0);
  const size_t work = vec4 ? plane / 4 : plane;
  const int per = 256 * (vec4 ? ilp : 1);
  int gx = (int)((work + per - 1) / per);
  if (gx > gx_cap) gx = gx_cap;
  if (gx < 1) gx = 1;
#define SA_SCAN(I) hipLaunchKernelGGL(nms_scan_kernel<I>, dim3(gx, B), dim3(256), 0, st, cms, H, W, C, threshold, max_peaks, keys, counts, status, vec4)
  if (ilp >= 8) SA_SCAN(8);
  else if (ilp >= 4) SA_SCAN(4);
  else if (ilp >= 2) SA_SCAN(2);
  else SA_SCAN(1);
#undef SA_SCAN
}

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
  launch_nms_scan(st, cms, B, H, W, C, threshold, max_peaks, keys, peak_count, status);
  SA_LAUNCH_CHECK();
  int n2 = 1;
  while (n2 < max_peaks) n2 <<= 1;
  hipLaunchKernelGGL(peaks_sort_refine_kernel, dim3(B), dim3(256), n2 * sizeof(uint32_t), st, cms,
                     offsets, H, W, C, refinement, patch_size, xy_scale, max_peaks, keys, peak_count,
                     peak_xy, peak_val, peak_chan, (int32_t*)nullptr);
  SA_LAUNCH_CHECK();
  return SA_OK;
}

int sa_find_local_peaks_rough(const float* cms, int B, int H, int W, int C, float threshold, int max_peaks, uint32_t* keys,
                              int32_t* peak_count, int32_t* status, sa_stream_t stream) {
  SA_REQUIRE(cms && keys && peak_count && status, "sa_find_local_peaks_rough: NULL pointer");
  SA_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0, "sa_find_local_peaks_rough: bad shape %dx%dx%dx%d", B, H, W, C);
  SA_REQUIRE((uint64_t)H * W * C < 0xFFFFFFFFull, "sa_find_local_peaks_rough: H*W*C exceeds 32-bit keys");
  SA_REQUIRE(max_peaks > 0 && max_peaks <= 16384, "sa_find_local_peaks_rough: max_peaks %d out of range", max_peaks);
  hipStream_t st = (hipStream_t)stream;
  SA_HIP_CHECK(hipMemsetAsync(peak_count, 0, sizeof(int32_t) * B, st));
  launch_nms_scan(st, cms, B, H, W, C, threshold, max_peaks, keys, peak_count, status);
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

int sa_select_centroids(const float* peak_xy, const float* peak_val, const int32_t* peak_count, int B, int max_peaks, int K,
                        int max_instances, float input_scale, float precrop_resize, int crop_size, float* centroids,
                        float* centroid_vals, float* crop_centres, float* crop_offsets, int32_t* n_valid, int32_t* status,
                        sa_stream_t stream) {
  SA_REQUIRE(B > 0 && max_peaks > 0 && K > 0 && crop_size > 0, "sa_select_centroids: bad shape");
  SA_REQUIRE(peak_xy && peak_val && peak_count && centroids && centroid_vals && crop_centres && crop_offsets && n_valid && status,
             "sa_select_centroids: NULL pointer");
  hipLaunchKernelGGL(select_centroids_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, peak_xy, peak_val, peak_count, max_peaks,
                     K, max_instances, input_scale, precrop_resize, (float)crop_size / 2.0f, centroids, centroid_vals, crop_centres,
                     crop_offsets, n_valid, status);
  SA_LAUNCH_CHECK();
  return SA_OK;
}

int sa_finish_instance_peaks(float* peaks, float* vals, const float* crop_offsets, const int32_t* n_valid, int B, int K, int N,
                             float input_scale, sa_stream_t stream) {
  SA_REQUIRE(B > 0 && K > 0 && N > 0 && peaks && vals && n_valid, "sa_finish_instance_peaks: bad arguments");
  const int n = B * K * N;
  hipLaunchKernelGGL(finish_instance_peaks_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, peaks, vals,
                     crop_offsets, n_valid, B, K, N, input_scale);
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
  const size_t lds = score_lds_bytes(N, max_node_peaks, E);
  SA_REQUIRE(lds <= 64 * 1024, "sa_paf_score: N * max_node_peaks too large for the LDS node tables");
  hipLaunchKernelGGL(paf_score_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, pafs, Hp, Wp, E,
                     peak_xy, peak_chan, peak_count, max_peaks, edges, N, n_points, pafs_stride,
                     max_edge_length, dist_penalty_weight, max_node_peaks, node_count, node_peaks,
                     line_scores, status);
  SA_LAUNCH_CHECK();
  return SA_OK;
}

int sa_paf_line_subs(const float* peaks_xy, const int32_t* edge_peak_inds, const int32_t* edge_inds, int K, int n_points,
                     float pafs_stride, int32_t* subs, sa_stream_t stream) {
  SA_REQUIRE(K >= 0 && n_points >= 1, "sa_paf_line_subs: bad shape");
  if (K == 0) return SA_OK;
  SA_REQUIRE(peaks_xy && edge_peak_inds && edge_inds && subs, "sa_paf_line_subs: NULL pointer");
  const int n = K * n_points;
  hipLaunchKernelGGL(paf_line_subs_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, peaks_xy, edge_peak_inds,
                     edge_inds, K, n_points, pafs_stride, subs);
  SA_LAUNCH_CHECK();
  return SA_OK;
}

int sa_gather_nd3(const float* src, int H, int W, int C, const int32_t* subs, int n, float* out, int32_t* status,
                  sa_stream_t stream) {
  SA_REQUIRE(n >= 0 && H > 0 && W > 0 && C > 0, "sa_gather_nd3: bad shape");
  if (n == 0) return SA_OK;
  SA_REQUIRE(src && subs && out, "sa_gather_nd3: NULL pointer");
  hipLaunchKernelGGL(gather_nd3_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, H, W, C, subs, n, out,
                     status);
  SA_LAUNCH_CHECK();
  return SA_OK;
}

int sa_paf_line_scores(const float* paf_lines, const float* peaks_xy, const int32_t* edge_peak_inds, int K, int n_points,
                       float max_edge_length, float dist_penalty_weight, float* line_scores, sa_stream_t stream) {
  SA_REQUIRE(K >= 0 && n_points >= 1, "sa_paf_line_scores: bad shape");
  if (K == 0) return SA_OK;
  SA_REQUIRE(paf_lines && peaks_xy && edge_peak_inds && line_scores, "sa_paf_line_scores: NULL pointer");
  hipLaunchKernelGGL(paf_line_scores_kernel, dim3((K + 255) / 256), dim3(256), 0, (hipStream_t)stream, paf_lines, peaks_xy,
                     edge_peak_inds, K, n_points, max_edge_length, dist_penalty_weight, line_scores);
  SA_LAUNCH_CHECK();
  return SA_OK;
}

int sa_distance_penalty(const float* lengths, int n, float max_edge_length, float dist_penalty_weight, float* out,
                        sa_stream_t stream) {
  SA_REQUIRE(n >= 0, "sa_distance_penalty: bad shape");
  if (n == 0) return SA_OK;
  SA_REQUIRE(lengths && out, "sa_distance_penalty: NULL pointer");
  hipLaunchKernelGGL(distance_penalty_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, lengths, n,
                     max_edge_length, dist_penalty_weight, out);
  SA_LAUNCH_CHECK();
  return SA_OK;
}

size_t sa_paf_workspace(int B, int E, int N, int max_node_peaks) {
  const size_t match = (size_t)B * E * sa::LsaWork::bytes(max_node_peaks);
  const size_t group = (size_t)B * (3 * (size_t)N * max_node_peaks + 1) * sizeof(int32_t);
  return (match > group ? match : group) + 64;
}

static int launch_match(const float* line_scores, const int32_t* node_count, const int32_t* edges, int B, int E, int N, int NP,
                        int32_t* match_dst, float* match_score, int32_t* status, hipStream_t st) {
  const size_t per = match_lds_bytes(NP);
  const int waves = (4 * per <= 60 * 1024) ? 4 : 1;
  const int nb = (B * E + waves - 1) / waves;
  hipLaunchKernelGGL(paf_match_kernel, dim3(nb), dim3(64 * waves), waves * per, st, line_scores, node_count, edges, B, E, N, NP,
                     match_dst, match_score, status);
  SA_LAUNCH_CHECK();
  return SA_OK;
}

int sa_paf_match(const float* line_scores, const int32_t* node_count, const int32_t* edges, int B,
                 int E, int N, int max_node_peaks, int32_t* match_dst, float* match_score,
                 int32_t* status, void* workspace, size_t ws_bytes, sa_stream_t stream) {
  SA_REQUIRE(max_node_peaks > 0 && max_node_peaks <= MAXNP, "sa_paf_match: max_node_peaks must be in [1,%d]", MAXNP);
  (void)workspace, (void)ws_bytes;  // the solver's work arrays live in LDS (kept in the signature: ABI v1)
  if (B * E == 0) return SA_OK;
  return launch_match(line_scores, node_count, edges, B, E, N, max_node_peaks, match_dst, match_score, status, (hipStream_t)stream);
}

static int launch_group(const GroupIn& g, int B, void* workspace, size_t ws_bytes, hipStream_t st) {
  SA_REQUIRE(g.N > 0 && g.N <= MAXNODES && g.NP > 0 && g.NP <= MAXNP, "sa_paf_group: bad N/max_node_peaks");
  SA_REQUIRE(g.max_instances > 0, "sa_paf_group: max_instances must be > 0");
  size_t lds = group_lds_bytes(g.N, g.NP, g.max_instances, g.E);
  int32_t* ws = nullptr;
  if (lds > 60 * 1024) {  // tables too large for LDS: single-lane walk on the global workspace
    const size_t per = sizeof(int32_t) * (3 * (size_t)g.N * g.NP + 1);
    if (!workspace || ws_bytes < (size_t)B * per)
      return sa::fail(SA_ERR_WORKSPACE, "sa_paf_group: workspace too small (see sa_paf_workspace)");
    ws = (int32_t*)workspace;
    lds = 0;
  }
  hipLaunchKernelGGL(paf_group_kernel, dim3(B), dim3(64), lds, st, g, ws);
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
  GroupIn g = {};
  g.peak_xy = peak_xy, g.peak_val = peak_val, g.node_count = node_count, g.node_peaks = node_peaks, g.max_peaks = max_peaks;
  g.match_dst = match_dst, g.match_score = match_score, g.edges = edges, g.sorted_edge_inds = sorted_edge_inds;
  g.n_sorted = n_sorted, g.E = E, g.N = N, g.NP = max_node_peaks, g.min_line_scores = min_line_scores;
  g.min_instance_peaks = min_instance_peaks, g.max_instances = max_instances, g.instance_peaks = instance_peaks;
  g.instance_peak_vals = instance_peak_vals, g.instance_scores = instance_scores, g.n_instances = n_instances, g.status = status;
  return launch_group(g, B, workspace, ws_bytes, (hipStream_t)stream);
}

int sa_paf_group_connections(const float* peak_xy, const float* peak_val, const int32_t* node_count, const int32_t* node_peaks,
                             int max_peaks, const int32_t* conn_edge, const int32_t* conn_src, const int32_t* conn_dst,
                             const float* conn_score, const int32_t* conn_count, int conn_stride, const int32_t* edges, int B,
                             int E, int N, int max_node_peaks, float min_line_scores, int min_instance_peaks, int max_instances,
                             float* instance_peaks, float* instance_peak_vals, float* instance_scores, int32_t* n_instances,
                             int32_t* assign_out, const int32_t* assign_in, const int32_t* order_in, const int32_t* order_count,
                             int32_t* status, void* workspace, size_t ws_bytes, sa_stream_t stream) {
  SA_REQUIRE(conn_edge && conn_src && conn_dst && conn_score && conn_count && conn_stride >= 0,
             "sa_paf_group_connections: NULL connection list");
  SA_REQUIRE(!assign_in == !order_in && !assign_in == !order_count, "sa_paf_group_connections: assign_in, order_in and order_count come together");
  GroupIn g = {};
  g.peak_xy = peak_xy, g.peak_val = peak_val, g.node_count = node_count, g.node_peaks = node_peaks, g.max_peaks = max_peaks;
  g.edges = edges, g.E = E, g.N = N, g.NP = max_node_peaks, g.min_line_scores = min_line_scores;
  g.min_instance_peaks = min_instance_peaks, g.max_instances = max_instances, g.instance_peaks = instance_peaks;
  g.instance_peak_vals = instance_peak_vals, g.instance_scores = instance_scores, g.n_instances = n_instances, g.status = status;
  g.conn_edge = conn_edge, g.conn_src = conn_src, g.conn_dst = conn_dst, g.conn_score = conn_score, g.conn_count = conn_count;
  g.conn_stride = conn_stride, g.assign_out = assign_out;
  g.assign_in = assign_in, g.order_in = order_in, g.order_count = order_count;
  return launch_group(g, B, workspace, ws_bytes, (hipStream_t)stream);
}

static size_t pp_keys_bytes(int B, int max_peaks) { return (sa_find_local_peaks_workspace(B, max_peaks) + 255) & ~(size_t)255; }
static size_t pp_counts_bytes(int B) { return ((size_t)B * sizeof(int32_t) + 255) & ~(size_t)255; }

size_t sa_bottomup_postproc_workspace(int B, int max_peaks, int E, int N, int max_node_peaks) {
  return pp_counts_bytes(B) + pp_keys_bytes(B, max_peaks) + sa_paf_workspace(B, E, N, max_node_peaks);
}

int sa_bottomup_postproc(const float* cms, const float* offsets, int B, int H, int W, int C, float threshold, int refinement,
                         int patch_size, float xy_scale, int max_peaks, const float* pafs, int Hp, int Wp, int E,
                         const int32_t* edges, const int32_t* sorted_edge_inds, int n_sorted, int N, int n_points,
                         float pafs_stride, float max_edge_length, float dist_penalty_weight, int max_node_peaks,
                         float min_line_scores, int min_instance_peaks, int max_instances, float* peak_xy, float* peak_val,
                         int32_t* peak_chan, int32_t* peak_count, int32_t* node_count, int32_t* node_peaks, float* line_scores,
                         int32_t* match_dst, float* match_score, float* instance_peaks, float* instance_peak_vals,
                         float* instance_scores, int32_t* n_instances, int32_t* status, void* workspace, size_t ws_bytes,
                         sa_stream_t stream) {
  SA_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0, "sa_bottomup_postproc: bad shape %dx%dx%dx%d", B, H, W, C);
  SA_REQUIRE(C == N, "sa_bottomup_postproc: %d confidence-map channels for %d skeleton nodes", C, N);
  SA_REQUIRE((uint64_t)H * W * C < 0xFFFFFFFFull, "sa_bottomup_postproc: H*W*C exceeds 32-bit keys");
  SA_REQUIRE(max_peaks > 0 && max_peaks <= 16384, "sa_bottomup_postproc: max_peaks %d out of range", max_peaks);
  SA_REQUIRE(refinement >= 0 && refinement <= 3, "sa_bottomup_postproc: bad refinement %d", refinement);
  SA_REQUIRE(refinement != SA_REFINE_OFFSETS || offsets, "sa_bottomup_postproc: offsets is NULL");
  SA_REQUIRE(refinement != SA_REFINE_INTEGRAL || (patch_size >= 1 && (patch_size & 1)),
             "sa_bottomup_postproc: integral patch size must be odd, got %d", patch_size);
  SA_REQUIRE(E >= 0 && N > 0 && N <= MAXNODES && max_node_peaks > 0 && max_node_peaks <= MAXNP && n_points >= 1 &&
                 max_instances > 0, "sa_bottomup_postproc: bad skeleton / capacity arguments");
  if (ws_bytes < sa_bottomup_postproc_workspace(B, max_peaks, E, N, max_node_peaks))
    return sa::fail(SA_ERR_WORKSPACE, "sa_bottomup_postproc: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  // workspace = [scan counters | keys | paf scratch]. The counters are zero on entry BY CONTRACT (zeroed once by the caller
  // after allocation, handed back zeroed by every call): the scan needs no memset launch in front of it.
  int32_t* scan_counts = (int32_t*)workspace;
  uint32_t* keys = (uint32_t*)((unsigned char*)workspace + pp_counts_bytes(B));
  unsigned char* paf_ws = (unsigned char*)keys + pp_keys_bytes(B, max_peaks);
  const size_t paf_ws_bytes = sa_paf_workspace(B, E, N, max_node_peaks);
  launch_nms_scan(st, cms, B, H, W, C, threshold, max_peaks, keys, scan_counts, status);
  SA_LAUNCH_CHECK();
  int n2 = 1;
  while (n2 < max_peaks) n2 <<= 1;
  GroupIn g = {};
  g.peak_xy = peak_xy, g.peak_val = peak_val, g.node_count = node_count, g.node_peaks = node_peaks, g.max_peaks = max_peaks;
  g.match_dst = match_dst, g.match_score = match_score, g.edges = edges, g.sorted_edge_inds = sorted_edge_inds;
  g.n_sorted = n_sorted, g.E = E, g.N = N, g.NP = max_node_peaks, g.min_line_scores = min_line_scores;
  g.min_instance_peaks = min_instance_peaks, g.max_instances = max_instances, g.instance_peaks = instance_peaks;
  g.instance_peak_vals = instance_peak_vals, g.instance_scores = instance_scores, g.n_instances = n_instances, g.status = status;
  // small batches: few workgroups on a 256-CU chip, so each one gets a wave per edge (all edges matched at once); large
  // batches keep 4 waves per frame
  int n_waves = 4;
  if (B <= 16) n_waves = E < 4 ? 4 : (E > 16 ? 16 : E);
  while (n_waves > 4 && (size_t)n_waves * match_lds_bytes(max_node_peaks) > 60 * 1024) --n_waves;
  // (round 6) + a sample buffer for the parallel integral refinement (256 peaks x k x k floats at a time, fewer if k is large) and
  // the frame's peak channels behind the scoring tables
  const int kk = patch_size > 0 ? patch_size * patch_size : 1;
  const int samp_words = (SA_POSTPROC_PAR && refinement == SA_REFINE_INTEGRAL && kk <= 8192) ? (256 * kk < 8192 ? 256 * kk : 8192) : 0;
  const int ch_words = SA_POSTPROC_PAR ? max_peaks : 0;
  size_t lds = (size_t)n2 * sizeof(uint32_t) + (size_t)samp_words * sizeof(float);
  const size_t l2 = score_lds_bytes(N, max_node_peaks, E) + (size_t)ch_words * sizeof(int32_t), l3 = (size_t)n_waves * match_lds_bytes(max_node_peaks),
               l4 = group_lds_bytes(N, max_node_peaks, max_instances, E);
  lds = lds > l2 ? lds : l2;
  lds = lds > l3 ? lds : l3;
  lds = lds > l4 ? lds : l4;
  static const bool no_fuse = getenv("SA_POSTPROC_UNFUSED") && atoi(getenv("SA_POSTPROC_UNFUSED")) != 0;
  if (lds <= 60 * 1024 && !no_fuse) {
    FusedIn f = {};
    f.cms = cms, f.offsets = offsets, f.H = H, f.W = W, f.C = C, f.mode = refinement, f.patch = patch_size, f.xy_scale = xy_scale;
    f.max_peaks = max_peaks, f.keys = keys, f.scan_counts = scan_counts, f.peak_count = peak_count, f.peak_xy = peak_xy;
    f.peak_val = peak_val;
    f.peak_chan = peak_chan, f.pafs = pafs, f.Hp = Hp, f.Wp = Wp, f.n_points = n_points, f.pafs_stride = pafs_stride;
    f.max_edge_length = max_edge_length, f.dist_penalty_weight = dist_penalty_weight, f.node_count = node_count;
    f.node_peaks = node_peaks, f.line_scores = line_scores, f.match_dst = match_dst, f.match_score = match_score;
    f.sort_words = n2, f.samp_words = samp_words, f.ch_words = ch_words;
    hipLaunchKernelGGL(bottomup_postproc_kernel, dim3(B), dim3(64 * n_waves), lds, st, f, g);
    SA_LAUNCH_CHECK();
    return SA_OK;
  }
  // capacities beyond one workgroup's LDS: the same stages as separate launches
  hipLaunchKernelGGL(peaks_sort_refine_kernel, dim3(B), dim3(256), n2 * sizeof(uint32_t), st, cms, offsets, H, W, C, refinement,
                     patch_size, xy_scale, max_peaks, keys, peak_count, peak_xy, peak_val, peak_chan, scan_counts);
  SA_LAUNCH_CHECK();
  int rc = sa_paf_score(pafs, B, Hp, Wp, E, peak_xy, peak_chan, peak_count, max_peaks, edges, N, n_points, pafs_stride,
                        max_edge_length, dist_penalty_weight, max_node_peaks, node_count, node_peaks, line_scores, status, stream);
  if (rc != SA_OK) return rc;
  if (B * E > 0) {
    rc = launch_match(line_scores, node_count, edges, B, E, N, max_node_peaks, match_dst, match_score, status, st);
    if (rc != SA_OK) return rc;
  }
  return launch_group(g, B, paf_ws, paf_ws_bytes, st);
}

static int lsa_host_impl(const double* cost, int nr, int nc, int64_t* row_ind, int64_t* col_ind, bool wave);

int sa_lsa_host(const double* cost, int nr, int nc, int64_t* row_ind, int64_t* col_ind) {
  return lsa_host_impl(cost, nr, nc, row_ind, col_ind, false);
}

/* the wave-cooperative solver of the matching kernel (lsa_solve_wave) with its 64 lanes emulated on the host */
int sa_lsa_host_wave(const double* cost, int nr, int nc, int64_t* row_ind, int64_t* col_ind) {
  return lsa_host_impl(cost, nr, nc, row_ind, col_ind, true);
}

static int lsa_host_impl(const double* cost, int nr, int nc, int64_t* row_ind, int64_t* col_ind, bool wave) {
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
  if (!(wave ? sa::lsa_solve_wave(R, Cn, cf, *w) : sa::lsa_solve(R, Cn, cf, *w))) return -1;
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

#if defined(SA_POSTPROC_STAMP)
extern "C" int sa_pp_stamp_read(unsigned long long* out, int reset) {
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pp_stamp_dev), sizeof(z)) != hipSuccess) return -1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(pp_stamp_dev), z, sizeof(z)) != hipSuccess) return -1;
  return 0;
}
#endif
