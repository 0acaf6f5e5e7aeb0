// "Tap GEMM": convolution-like layers whose taps do not share a halo tile, on the CDNA4 matrix cores
// (v_mfma_f32_32x32x16_bf16, fp32 accumulate), bf16 NHWC activations:
//
//   out[b, ly*os + oy0, lx*os + ox0, co] = epi( sum_t sum_ci  x[b, ly*is + dy_t, lx*is + dx_t, ci] * w[t][ci][co] )
//
//   * Conv2D(k1, stride 1|2, valid)          -> one tap (0,0), is = stride            (resnet.py:196-229)
//   * Conv2DTranspose(k4|k3, s2, same)       -> os = 2, one launch per output phase (oy0, ox0) with the 1..4 taps that
//                                               hit that phase                          (upsampling.py:177-188)
//   epi = bias, ReLU, per-channel affine (BatchNormalization), residual add (Add), ReLU  -- the same extended epilogue
//   as sa_conv3x3_ex_bf16.
//
// GEMM view: A = packed weights (M = cout), B = pixels (N), K = taps x Cin. A workgroup (4 waves) owns 64*WM logical
// pixels of ONE frame x 64*WN output channels; every wave a 64 x 64 sub-tile (2 x 2 accumulators of 32x32). K runs in
// chunks of 64 channels of one tap: the pixel chunk (128 B per pixel) and the weight slabs are copied global -> LDS by
// buffer_load ... lds (double buffered; taps that fall outside the image and channel slots beyond CinP read zeros through
// the buffer bounds check), the 16-byte channel slots of a pixel are XOR-swizzled with (pixel >> 1) & 7 on the copy's
// SOURCE side so that the ds_read_b128 of a B fragment touches all 64 banks once per 16 lanes.
#include <cstdint>
#include <cstdlib>

#include "bf16.h"
#include "sa_common.h"

namespace {

using sa::h16x8_t;
using sa::mfma_h8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int MAX_TAPS = 16;
constexpr int MAX_WINDOW_TAPS = 81;  // sa_convk_bf16: up to 9 x 9 windows (their offsets are computed, not listed)

struct TapParams {
  const uint16_t* src;  // [B,Hs,Ws,CinP]
  const uint16_t* w;    // packed [co32][tap][k16][64][8]
  const float* bias;    // [CoutP]
  uint16_t* dst;        // [B,Ho,Wo,CoutP]
  const float* post_scale;
  const float* post_shift;
  const uint16_t* residual;  // [B,Ho,Wo,CoutP] or NULL
  int CinP, CoutP;
  int B, Hs, Ws;  // source size
  int Hl, Wl;     // logical output grid of this launch
  int Ho, Wo;     // destination tensor size
  int in_stride, out_stride, oy0, ox0;
  int n_taps;
  int tap_dy[MAX_TAPS], tap_dx[MAX_TAPS];
  // ksize > 0: the taps are the ksize x ksize window of a stride-1 "same" Conv2D, tap t = (ky, kx) = (t / ksize, t % ksize) at
  // offset (ky - kpad, kx - kpad) with kpad = (ksize - 1) / 2 (TF SAME: pad_before = pad_total / 2); the lists are ignored
  int ksize, kpad;
  int relu, relu_last;
  int m_tiles, co_tiles;
  // n_phases > 1: the output phases of a stride-2 transposed conv as ONE grid (phase = fastest block index): per phase its
  // packed weights, output offset and (<= 4) taps; the single-phase fields above are ignored then
  int n_phases;
  const uint16_t* ph_w[4];
  int ph_oy[4], ph_ox[4], ph_ntaps[4];
  int ph_dy[4][4], ph_dx[4][4];
  // 1: src / residual / dst are 16-channel planes [B, CP/16, H, W, 16] (SA_LAYOUT_PLANES16) instead of NHWC: a store instruction
  // of the epilogue then writes 32 pixels x 32 bytes = 1 KiB contiguously (NHWC: 32 separate 32-byte pieces, one per 128-byte
  // line -- the 64 -> 256 layers wrote at 2 TB/s), a copy instruction reads 32 pixels x 32 bytes of ONE plane
  int planar;
};

// m / w for 0 <= m < 2^23 without the ~40-instruction integer division: float reciprocal estimate + one correction step
__device__ __forceinline__ int fast_div(int m, int w, float inv_w) {
  int q = (int)(((float)m + 0.5f) * inv_w);
  if (q * w > m) --q;
  if ((q + 1) * w <= m) ++q;
  return q;
}

// PL: 16-channel planes (TapParams::planar). The LDS stage is then plane-major too -- [k16 step][pixel][32 B], the two 16-byte
// halves of a pixel record XOR-swizzled with (pixel >> 3) & 1 as in conv3x3_dma_kernel's 16-channel chunks -- because a copy
// writes LDS lane-linearly: one copy instruction = the 32 pixels of one pixel group in one plane (1 KiB on both sides).
// RES: the launch has a residual operand (ResNet shortcut Add). Only those instantiations carry the residual prefetch -- 32
// registers live across the K loop (every 1x1 / k x k / transposed launch used to pay for it).
// (Round 4, measured and dropped: three / four LDS stages with counted waits -- s_waitcnt vmcnt(k x copies per chunk) so that a
// wave does not drain its whole copy queue per chunk: 6.78 -> 6.95 / 7.33 ms on the ResNet-50 network, every multi-chunk 1x1 conv
// slower. The extra LDS costs resident workgroups, and those, not the depth of one workgroup's queue, are what hides the latency
// here; profiles/r04_tapconv_sweep.md.)
// CK: channels per K chunk (64 or 32). A stage is TP x 2 CK bytes of pixels + NCO32 x CK / 16 KiB of weights; the kernel is bound by
// memory latency, so what matters is how many workgroups a CU holds (LDS per workgroup = 1 or 2 stages, registers) -- round 4.
template <int WM, int WN, bool PL, bool RES, int CK>
__global__ void __launch_bounds__(256)
tapconv_kernel(const TapParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int TP = 64 * WM;          // pixels per workgroup
  constexpr int NCO32 = 2 * WN;        // 32-cout tiles per workgroup
  constexpr int KK = CK / 16;          // k16 steps per chunk
  constexpr int N_IN = TP * CK * 2 / 1024;  // 1 KiB copies per pixel chunk
  constexpr int IN_BYTES = N_IN * 1024;
  constexpr int N_W = NCO32 * KK;
  constexpr int STAGE = IN_BYTES + N_W * 1024;
  constexpr int IN_PER_WAVE = N_IN / 4;
  constexpr int W_PER_WAVE = N_W / 4;
  constexpr unsigned OOB = 0xFFFFFF00u;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  int bid;
  {  // XCD-aware order: consecutive logical tiles (cout tiles of one pixel tile first) share an L2
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  int ph = 0;
  if (p.n_phases > 1) {
    ph = bid % p.n_phases;
    bid /= p.n_phases;
  }
  const bool multi = p.n_phases > 1;
  const uint16_t* wsel = multi ? p.ph_w[ph] : p.w;
  const int n_taps = multi ? p.ph_ntaps[ph] : p.n_taps;
  const int oy0 = multi ? p.ph_oy[ph] : p.oy0, ox0 = multi ? p.ph_ox[ph] : p.ox0;
  const int co_t = bid % p.co_tiles;
  bid /= p.co_tiles;
  const int m_t = bid % p.m_tiles;
  const int b = bid / p.m_tiles;
  const int m0 = m_t * TP;
  const int ML = p.Hl * p.Wl;
  const int K16 = p.CinP / 16;
  const float inv_wl = 1.0f / (float)p.Wl;
  const int co32_n = (p.CoutP + 31) / 32;
  const int co32_0 = co_t * NCO32;
  const int chunks_per_tap = (p.CinP + CK - 1) / CK;
  const int n_chunks = n_taps * chunks_per_tap;

  const size_t fbytes = (size_t)p.Hs * p.Ws * p.CinP * 2;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const unsigned char*>(p.src) + b * fbytes), 0, (int)fbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
      (void*)wsel, 0, (int)((size_t)co32_n * n_taps * K16 * 1024), 0x00020000);

  // this lane's pixels in the copies it issues: copy i covers pixels 8i..8i+7, lane -> (pixel 8i + lane/8, slot lane%8)
  int py[IN_PER_WAVE], px[IN_PER_WAVE], pslot[IN_PER_WAVE];
#pragma unroll
  for (int j = 0; j < IN_PER_WAVE; ++j) {
    const int i = j * 4 + wave;
    // NHWC: copy i covers pixels 8i..8i+7 (8 slots of 16 bytes each); planes: copy i = pixel group i % (TP/32) of k16 step
    // i / (TP/32): pixels 32 pg .. 32 pg + 31, two 16-byte halves each
    const int pl = PL ? (i % (TP / 32)) * 32 + (lane >> 1) : i * 8 + (lane >> 3);
    const int m = m0 + pl;
    const bool ok = m < ML;
    const int ly = ok ? fast_div(m, p.Wl, inv_wl) : 0;
    py[j] = ok ? ly * p.in_stride : -(1 << 20);
    px[j] = ok ? (m - ly * p.Wl) * p.in_stride : 0;
    pslot[j] = PL ? ((lane & 1) ^ ((pl >> 3) & 1)) : ((lane & 7) ^ ((pl >> 1) & 7));  // global 16-byte slot this lane fetches
  }
  const unsigned plane_bytes_in = (unsigned)((size_t)p.Hs * p.Ws * 32);
  const unsigned wv = (unsigned)lane * 16;

  auto issue = [&](int chunk, int buf) {
    const int tap = chunk / chunks_per_tap, kc = chunk - tap * chunks_per_tap;
    int dy, dx;
    if (p.ksize > 0) {
      const int ky = tap / p.ksize;
      dy = ky - p.kpad;
      dx = tap - ky * p.ksize - p.kpad;
    } else {
      dy = multi ? p.ph_dy[ph][tap] : p.tap_dy[tap];
      dx = multi ? p.ph_dx[ph][tap] : p.tap_dx[tap];
    }
    unsigned char* stage = smem + buf * STAGE;
#pragma unroll
    for (int j = 0; j < IN_PER_WAVE; ++j) {
      const int i = j * 4 + wave;
      const int sy = py[j] + dy, sx = px[j] + dx;
      if constexpr (PL) {
        const int k16 = kc * KK + i / (TP / 32);  // the plane this copy reads (wave uniform)
        const bool ok = sy >= 0 && sy < p.Hs && sx >= 0 && sx < p.Ws && k16 < K16;
        const unsigned voff = ok ? (unsigned)((sy * p.Ws + sx) * 32 + pslot[j] * 16) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(stage + i * 1024), 16, voff, (int)((unsigned)k16 * plane_bytes_in), 0, 0);
        continue;
      }
      const bool ok = sy >= 0 && sy < p.Hs && sx >= 0 && sx < p.Ws && (kc * CK + pslot[j] * 8) < p.CinP;
      const unsigned voff = ok ? (unsigned)((sy * p.Ws + sx) * (p.CinP * 2) + pslot[j] * 16) : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(stage + i * 1024), 16, voff, kc * CK * 2, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < W_PER_WAVE; ++j) {
      const int k = j * 4 + wave;  // (co32 local, kk)
      const int c32 = k / KK, kk = k - c32 * KK;
      const int k16 = kc * KK + kk;
      const int soff = (co32_0 + c32 < co32_n && k16 < K16)
                           ? (((co32_0 + c32) * n_taps + tap) * K16 + k16) * 1024
                           : (int)0x7FFFF000;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)(stage + IN_BYTES + k * 1024), 16, wv, soff, 0, 0);
    }
  };

  f32x16 acc[2][2];  // [cout tile][pixel group]
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][r][i] = 0.0f;

  const int half = lane >> 5, lx = lane & 31;
  // ---- epilogue (same register layout as conv3x3_dma_kernel): a lane holds channels {0-3, 8-11, 16-19, 24-27} + 4*half
  // of its pixel; one exchange with lane ^ 32 per pair of groups -> every lane stores 8 consecutive channels (16 bytes).
  // (Round 3, measured and dropped: the wave's 64 x 64 float32 results transposed through LDS so that every global instruction
  // covers 8 pixels x 128 contiguous bytes instead of 32 pixels x 32 bytes: bitwise the same, 0.40 -> 0.47 ms on the
  // 64 -> 256 + residual layers -- the extra barrier and LDS round trip cost more than the coalescing gave;
  // profiles/r03_ab_session.md section 3.)
  uint16_t* drow[2];
  const uint16_t* rrow[2];
  bool pix_ok[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int m = m0 + (wm * 2 + r) * 32 + lx;
    pix_ok[r] = m < ML;
    const int ly = pix_ok[r] ? fast_div(m, p.Wl, inv_wl) : 0;
    const int lxx = pix_ok[r] ? m - ly * p.Wl : 0;
    // element offset of this pixel's channel 0: NHWC pixel record of CoutP elements; planes: frame base + 16 elements per pixel
    // inside a plane (channel co adds (co >> 4) planes + (co & 15), `chan_off`)
    const size_t opix = (size_t)(ly * p.out_stride + oy0) * p.Wo + (lxx * p.out_stride + ox0);
    const size_t off = PL ? (size_t)b * p.Ho * p.Wo * p.CoutP + opix * 16 : ((size_t)b * p.Ho * p.Wo + opix) * p.CoutP;
    drow[r] = p.dst + off;
    rrow[r] = RES ? p.residual + off : nullptr;
  }
  const size_t plane_elems_out = (size_t)p.Ho * p.Wo * 16;
  auto chan_off = [&](int co) -> size_t { return PL ? (size_t)(co >> 4) * plane_elems_out + (co & 15) : (size_t)co; };
  // the residual of the whole tile is requested BEFORE the K loop (32 registers): in the epilogue each load was a dependent HBM
  // round trip of its own. Round 4: 16-byte loads -- the lane fetches the 8 CONSECUTIVE channels it will store (16 pr + 8 half),
  // and one v_permlane32_swap per dword hands each half-wave the 4-channel groups of its accumulator layout (the store path's
  // exchange, backwards): 8 load instructions per lane instead of 16, each covering 32 bytes per pixel (planes: 1 KiB contiguous)
  // (the exchange itself waits for the data, so it is done in the epilogue: the K loop runs with the loads in flight)
  uint4 rq4[RES ? 2 : 1][2][2];
  auto load_res = [&](int mt) {
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int co = (co32_0 + wn * 2 + mt) * 32 + 16 * pr + 8 * half;
        rq4[mt][pr][r] = make_uint4(0u, 0u, 0u, 0u);
        if (pix_ok[r] && co < p.CoutP) rq4[mt][pr][r] = *reinterpret_cast<const uint4*>(rrow[r] + chan_off(co));
      }
  };
  if constexpr (RES) {  // (loading it in the epilogue instead, or a 128-register budget: measured, profiles/r04_tapconv_sweep.md)
    load_res(0);
    load_res(1);
  }
  issue(0, 0);
  int buf = 0;
  for (int chunk = 0; chunk < n_chunks; ++chunk) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (chunk + 1 < n_chunks) issue(chunk + 1, buf ^ 1);
    const unsigned char* in_tile = smem + buf * STAGE;
    const unsigned char* w_tile = in_tile + IN_BYTES;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      mfma_h8 a[2], bv[2];
#pragma unroll
      for (int m = 0; m < 2; ++m)
        a[m] = *reinterpret_cast<const mfma_h8*>(w_tile + ((wn * 2 + m) * KK + kk) * 1024 + lane * 16);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int pl = (wm * 2 + r) * 32 + lx;
        if constexpr (PL) {
          bv[r] = *reinterpret_cast<const mfma_h8*>(in_tile + kk * (TP * 32) + pl * 32 + ((half ^ ((pl >> 3) & 1)) * 16));
        } else {
          const int slot = (kk * 2 + half) ^ ((pl >> 1) & 7);
          bv[r] = *reinterpret_cast<const mfma_h8*>(in_tile + pl * (CK * 2) + slot * 16);
        }
      }
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 2; ++r) acc[m][r] = SA_MFMA_32x32x16(a[m], bv[r], acc[m][r], 0, 0, 0);
    }
    buf ^= 1;
  }

  const float lowv = p.relu ? 0.0f : -INFINITY, lowl = p.relu_last ? 0.0f : -INFINITY;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int cobase = (co32_0 + wn * 2 + mt) * 32;
    if (cobase >= p.CoutP) continue;
    uint2 pk[2][4];
    uint2 rq[4][2];  // residual, back in the accumulator layout: group g = channels 8 g + 4 half + 0..3
    if constexpr (RES) {
#pragma unroll
      for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const uint4 q = rq4[mt][pr][r];
          uint2 a = make_uint2(q.x, q.y), c = make_uint2(q.z, q.w);
          sa::swap32(a.x, c.x);
          sa::swap32(a.y, c.y);
          rq[2 * pr][r] = a;
          rq[2 * pr + 1][r] = c;
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int co = cobase + 8 * g + 4 * half;
      const bool cok = co < p.CoutP;
      // per-channel parameters: once per channel group, shared by both pixel groups
      float4 bq = make_float4(0.f, 0.f, 0.f, 0.f), sq = make_float4(1.f, 1.f, 1.f, 1.f), tq = bq;
      if (cok) {
        bq = *reinterpret_cast<const float4*>(p.bias + co);
        if (p.post_scale) {
          sq = *reinterpret_cast<const float4*>(p.post_scale + co);
          tq = *reinterpret_cast<const float4*>(p.post_shift + co);
        }
      }
      const float bb[4] = {bq.x, bq.y, bq.z, bq.w}, ss[4] = {sq.x, sq.y, sq.z, sq.w}, tt[4] = {tq.x, tq.y, tq.z, tq.w};
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        float rr[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (RES) {
          const uint2 q = rq[g][r];
          rr[0] = sa::h2f((uint16_t)(q.x & 0xffff)), rr[1] = sa::h2f((uint16_t)(q.x >> 16));
          rr[2] = sa::h2f((uint16_t)(q.y & 0xffff)), rr[3] = sa::h2f((uint16_t)(q.y >> 16));
        }
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float t = fmaxf(acc[mt][r][4 * g + j] + bb[j], lowv);
          t = fmaf(t, ss[j], tt[j]);
          if constexpr (RES) t += rr[j];  // (the same two roundings as before: fma, then the add)
          v[j] = fmaxf(t, lowl);
        }
        pk[r][g].x = sa::f2h2(v[0], v[1]);
        pk[r][g].y = sa::f2h2(v[2], v[3]);
      }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        uint2 a = pk[r][2 * pr], c = pk[r][2 * pr + 1];
        sa::swap32(a.x, c.x);
        sa::swap32(a.y, c.y);
        const int co = cobase + 16 * pr + 8 * half;
        if (pix_ok[r] && co < p.CoutP) *reinterpret_cast<uint4*>(drow[r] + chan_off(co)) = make_uint4(a.x, a.y, c.x, c.y);
      }
  }
#endif
}

// ---- 1 x 1 convs whose output channels span SEVERAL 128-channel tiles (ResNet's expand convs: 128 -> 512, 256 -> 1024), on planes:
// ONE workgroup per pixel tile walks all cout tiles. The pixel tile (TP pixels x all input channels) is copied into LDS once and
// stays; per cout tile only the weights stream (from L2), double buffered in 32-channel chunks, the first chunk of cout tile t + 1
// queued under the last chunk of t. Against one workgroup per (pixel tile, cout tile): the input is fetched once instead of
// co_tiles times, the per-chunk copy is 8 KiB of L2-resident weights instead of 8 + 8 KiB with an HBM leg, and the prologue is
// paid once. Same arithmetic in the same order per output: bitwise the tap-GEMM kernel's results.
template <int WM, int WN, bool RES>
__global__ void __launch_bounds__(256)
tapconv_coloop_kernel(const TapParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int CK = 32, KK = 2;
  constexpr int TP = 64 * WM, NCO32 = 2 * WN, PG = TP / 32;
  constexpr int N_W = NCO32 * KK, W_STAGE = N_W * 1024, W_PER_WAVE = N_W / 4;
  static_assert(N_W % 4 == 0, "weight pieces per wave");
  constexpr unsigned OOB = 0xFFFFFF00u;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  int bid;
  {
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int m_t = bid % p.m_tiles;
  const int b = bid / p.m_tiles;
  const int m0 = m_t * TP;
  const int ML = p.Hl * p.Wl;
  const int K16 = p.CinP / 16;
  const float inv_wl = 1.0f / (float)p.Wl;
  const int co32_n = (p.CoutP + 31) / 32;
  const int n_chunks = (p.CinP + CK - 1) / CK;   // per cout tile
  const int n_cot = p.co_tiles;
  const int in_bytes = K16 * TP * 32;            // the resident pixel tile: [k16][pixel][32 B]
  unsigned char* wst = smem + in_bytes;           // two weight stages behind it

  const size_t fbytes = (size_t)p.Hs * p.Ws * p.CinP * 2;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const unsigned char*>(p.src) + b * fbytes), 0, (int)fbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)((size_t)co32_n * K16 * 1024), 0x00020000);
  const unsigned plane_bytes_in = (unsigned)((size_t)p.Hs * p.Ws * 32);
  const unsigned wv = (unsigned)lane * 16;

  // ---- the pixel tile, once: piece i = pixel group i % PG of plane i / PG (1 KiB); wave w takes pieces w, w + 4, ...
  for (int i = wave; i < PG * K16; i += 4) {
    const int pg = i % PG, k16 = i / PG;
    const int pl = pg * 32 + (lane >> 1);
    const int m = m0 + pl;
    const bool okp = m < ML;
    const int ly = okp ? fast_div(m, p.Wl, inv_wl) : 0;
    const int sy = ly * p.in_stride, sx = (m - ly * p.Wl) * p.in_stride;
    const unsigned slot = (unsigned)((lane & 1) ^ ((pl >> 3) & 1));
    const unsigned voff = okp ? (unsigned)((sy * p.Ws + sx) * 32) + slot * 16u : OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + i * 1024), 16, voff, (int)((unsigned)k16 * plane_bytes_in), 0, 0);
  }
  // weight chunk `step` = (cout tile step / n_chunks, chunk step % n_chunks) -> stage step & 1
  auto issue_w = [&](int step) {
    const int ct = step / n_chunks, kc = step - ct * n_chunks;
    unsigned char* stage = wst + (step & 1) * W_STAGE;
#pragma unroll
    for (int j = 0; j < W_PER_WAVE; ++j) {
      const int k = j * 4 + wave;
      const int c32 = k / KK, kk = k - c32 * KK;
      const int k16 = kc * KK + kk;
      const int soff = (ct * NCO32 + c32 < co32_n && k16 < K16) ? ((ct * NCO32 + c32) * K16 + k16) * 1024 : (int)0x7FFFF000;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)(stage + k * 1024), 16, wv, soff, 0, 0);
    }
  };

  const int half = lane >> 5, lx = lane & 31;
  uint16_t* drow[2];
  const uint16_t* rrow[2];
  bool pix_ok[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int m = m0 + (wm * 2 + r) * 32 + lx;
    pix_ok[r] = m < ML;
    const int ly = pix_ok[r] ? fast_div(m, p.Wl, inv_wl) : 0;
    const int lxx = pix_ok[r] ? m - ly * p.Wl : 0;
    const size_t opix = (size_t)(ly * p.out_stride + p.oy0) * p.Wo + (lxx * p.out_stride + p.ox0);
    const size_t off = (size_t)b * p.Ho * p.Wo * p.CoutP + opix * 16;
    drow[r] = p.dst + off;
    rrow[r] = RES ? p.residual + off : nullptr;
  }
  const size_t plane_elems_out = (size_t)p.Ho * p.Wo * 16;
  auto chan_off = [&](int co) -> size_t { return (size_t)(co >> 4) * plane_elems_out + (co & 15); };
  const float lowv = p.relu ? 0.0f : -INFINITY, lowl = p.relu_last ? 0.0f : -INFINITY;

  const int total = n_cot * n_chunks;
  issue_w(0);
  int step = 0;
#pragma clang loop unroll(disable)
  for (int ct = 0; ct < n_cot; ++ct) {
    const int co32_0 = ct * NCO32;
    // this cout tile's residual: requested before its K loop (the loop's first wait covers it)
    uint4 rq4[RES ? 2 : 1][2][2];
    if constexpr (RES) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr)
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const int co = (co32_0 + wn * 2 + mt) * 32 + 16 * pr + 8 * half;
            rq4[mt][pr][r] = make_uint4(0u, 0u, 0u, 0u);
            if (pix_ok[r] && co < p.CoutP) rq4[mt][pr][r] = *reinterpret_cast<const uint4*>(rrow[r] + chan_off(co));
          }
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][r][i] = 0.0f;
#pragma clang loop unroll(disable)
    for (int kc = 0; kc < n_chunks; ++kc, ++step) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (step + 1 < total) issue_w(step + 1);
      const unsigned char* w_tile = wst + (step & 1) * W_STAGE;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        mfma_h8 a[2], bv[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) a[m] = *reinterpret_cast<const mfma_h8*>(w_tile + ((wn * 2 + m) * KK + kk) * 1024 + lane * 16);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int pl = (wm * 2 + r) * 32 + lx;
          bv[r] = *reinterpret_cast<const mfma_h8*>(smem + (kc * KK + kk) * (TP * 32) + pl * 32 + ((half ^ ((pl >> 3) & 1)) * 16));
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 2; ++r) acc[m][r] = SA_MFMA_32x32x16(a[m], bv[r], acc[m][r], 0, 0, 0);
      }
    }
    // ---- epilogue of this cout tile: the tap-GEMM kernel's, operation for operation
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int cobase = (co32_0 + wn * 2 + mt) * 32;
      if (cobase >= p.CoutP) continue;
      uint2 pk[2][4];
      uint2 rq[4][2];
      if constexpr (RES) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr)
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const uint4 q = rq4[mt][pr][r];
            uint2 a = make_uint2(q.x, q.y), c = make_uint2(q.z, q.w);
            sa::swap32(a.x, c.x);
            sa::swap32(a.y, c.y);
            rq[2 * pr][r] = a;
            rq[2 * pr + 1][r] = c;
          }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = cobase + 8 * g + 4 * half;
        const bool cok = co < p.CoutP;
        float4 bq = make_float4(0.f, 0.f, 0.f, 0.f), sq = make_float4(1.f, 1.f, 1.f, 1.f), tq = bq;
        if (cok) {
          bq = *reinterpret_cast<const float4*>(p.bias + co);
          if (p.post_scale) {
            sq = *reinterpret_cast<const float4*>(p.post_scale + co);
            tq = *reinterpret_cast<const float4*>(p.post_shift + co);
          }
        }
        const float bb[4] = {bq.x, bq.y, bq.z, bq.w}, ss[4] = {sq.x, sq.y, sq.z, sq.w}, tt[4] = {tq.x, tq.y, tq.z, tq.w};
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          float rr[4] = {0.0f, 0.0f, 0.0f, 0.0f};
          if constexpr (RES) {
            const uint2 q = rq[g][r];
            rr[0] = sa::h2f((uint16_t)(q.x & 0xffff)), rr[1] = sa::h2f((uint16_t)(q.x >> 16));
            rr[2] = sa::h2f((uint16_t)(q.y & 0xffff)), rr[3] = sa::h2f((uint16_t)(q.y >> 16));
          }
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float t = fmaxf(acc[mt][r][4 * g + j] + bb[j], lowv);
            t = fmaf(t, ss[j], tt[j]);
            if constexpr (RES) t += rr[j];
            v[j] = fmaxf(t, lowl);
          }
          pk[r][g].x = sa::f2h2(v[0], v[1]);
          pk[r][g].y = sa::f2h2(v[2], v[3]);
        }
      }
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          uint2 a = pk[r][2 * pr], c = pk[r][2 * pr + 1];
          sa::swap32(a.x, c.x);
          sa::swap32(a.y, c.y);
          const int co = cobase + 16 * pr + 8 * half;
          if (pix_ok[r] && co < p.CoutP) *reinterpret_cast<uint4*>(drow[r] + chan_off(co)) = make_uint4(a.x, a.y, c.x, c.y);
        }
    }
  }
#endif
}

template <int WM, int WN, bool RES>
int launch_tap_coloop(const TapParams& p0, hipStream_t st) {
  constexpr int TP = 64 * WM, NCO32 = 2 * WN;
  TapParams p = p0;
  p.m_tiles = (p.Hl * p.Wl + TP - 1) / TP;
  p.co_tiles = ((p.CoutP + 31) / 32 + NCO32 - 1) / NCO32;
  const size_t lds = (size_t)(p.CinP / 16) * TP * 32 + 2 * (size_t)NCO32 * 2 * 1024;
  const size_t nblk = (size_t)p.m_tiles * p.B;
  if (nblk == 0 || nblk > 0x7fffffffull) return sa::fail(SA_ERR_INVALID_ARG, "tapconv: bad grid");
  if ((size_t)p.Hl * p.Wl >= (1u << 23)) return sa::fail(SA_ERR_UNSUPPORTED, "tapconv: more than 2^23 output pixels per frame");
  static bool attr_set = false;
  if (!attr_set) {
    SA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tapconv_coloop_kernel<WM, WN, RES>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((tapconv_coloop_kernel<WM, WN, RES>), dim3((unsigned)nblk), dim3(256), lds, st, p);
  SA_LAUNCH_CHECK();
  return SA_OK;
}

template <int WM, int WN, bool PL, bool RES, int CK>
int launch_tap(const TapParams& p0, hipStream_t st) {
  constexpr int TP = 64 * WM, NCO32 = 2 * WN;
  constexpr size_t stage = (size_t)TP * CK * 2 + (size_t)NCO32 * (CK / 16) * 1024;
  TapParams p = p0;
  // single-chunk launches (one tap, Cin <= 64: the 64 -> 256 convs of ResNet's first stage) never touch the second stage: half
  // the LDS = twice the workgroups a CU can hold by LDS (round 3 measured "no change" for this -- with the accumulators in AGPRs
  // the register file capped a SIMD at two waves anyway; built VGPR-form since round 4: 124-156 registers, three waves)
  int max_taps = p.n_taps;
  for (int i = 0; i < p.n_phases; ++i) max_taps = p.ph_ntaps[i] > max_taps ? p.ph_ntaps[i] : max_taps;
  const size_t lds = (max_taps * ((p.CinP + CK - 1) / CK) == 1 ? 1 : 2) * stage;
  p.m_tiles = (p.Hl * p.Wl + TP - 1) / TP;
  p.co_tiles = ((p.CoutP + 31) / 32 + NCO32 - 1) / NCO32;
  const size_t nblk = (size_t)p.m_tiles * p.co_tiles * p.B * (p.n_phases > 1 ? p.n_phases : 1);
  if (nblk == 0 || nblk > 0x7fffffffull) return sa::fail(SA_ERR_INVALID_ARG, "tapconv: bad grid");
  if ((size_t)p.Hl * p.Wl >= (1u << 23))  // fast_div's float estimate is exact below 2^23
    return sa::fail(SA_ERR_UNSUPPORTED, "tapconv: more than 2^23 output pixels per frame");
  static bool attr_set = false;
  if (!attr_set && 2 * stage > 64 * 1024) {
    SA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tapconv_kernel<WM, WN, PL, RES, CK>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * stage)));
    attr_set = true;
  }
  // (a single LDS stage for single-chunk launches -- twice the resident workgroups -- measured no change: round 3, gpurun_out/r03n)
  if (PL && ((size_t)p.Hs * p.Ws * p.CinP * 2 >= 0xFFFFFF00ull || p.CinP % 16 || p.CoutP % 16))
    return sa::fail(SA_ERR_UNSUPPORTED, "tapconv: SA_LAYOUT_PLANES16 needs channel counts padded to 16 and frames below 4 GiB");
  hipLaunchKernelGGL((tapconv_kernel<WM, WN, PL, RES, CK>), dim3((unsigned)nblk), dim3(256), lds, st, p);
  SA_LAUNCH_CHECK();
  return SA_OK;
}

// Tile shape and chunk size. Defaults: measured on MI355X (profiles/r04_tapconv_sweep.md; the 64 px x 256 couts shape lost
// everywhere and is not built). SA_TAP_SHAPE = 1 / 2 forces 256 px x 64 couts / 128 x 128 per workgroup, SA_TAP_CK = 32 / 64
// the chunk size (A/B runs).
template <bool PL, bool RES, int CK>
int launch_tap_shape(const TapParams& p, hipStream_t st) {
  static const int force = [] {
    const char* v = getenv("SA_TAP_SHAPE");
    return v ? atoi(v) : 0;
  }();
  // <= 64 output channels: 256 pixels x 64 couts per workgroup -- unless that leaves the chip with fewer than two workgroups per
  // CU on a long K loop (the ResNet decoder's first transposed conv, 2048 -> 64 @32^2 x 16 frames: 256 workgroups of 128 chunks
  // each): then 128 pixels x 128 couts, half of them padding, for twice the workgroups (0.180 -> 0.157 ms)
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    SA_HIP_CHECK(hipGetDevice(&dev));
    SA_HIP_CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  }
  const size_t wg1 = (size_t)((p.Hl * p.Wl + 255) / 256) * p.B * (p.n_phases > 1 ? p.n_phases : 1);
  int max_taps = p.n_taps > 0 ? p.n_taps : 1;
  for (int i = 0; i < p.n_phases; ++i) max_taps = p.ph_ntaps[i] > max_taps ? p.ph_ntaps[i] : max_taps;
  const bool starved = wg1 < (size_t)2 * n_cu && (size_t)p.CinP * max_taps >= 2048;
  const int shape = force ? force : ((p.CoutP <= 64 && !starved) ? 1 : 2);
  if (shape == 1) return launch_tap<4, 1, PL, RES, CK>(p, st);
  return launch_tap<2, 2, PL, RES, CK>(p, st);
}

template <bool PL, bool RES>
int launch_tap_ck(const TapParams& p, hipStream_t st) {
  static const int force = [] {
    const char* v = getenv("SA_TAP_CK");
    return v ? atoi(v) : 0;
  }();
  // measured (profiles/r04_tapconv_sweep.md, ResNet-50 @1024^2, 16 frames): 32-channel chunks take 5-15 % off every 1x1 conv with
  // >= 128 input channels (a stage is half the LDS: more workgroups per CU for a latency-bound kernel); single-chunk launches
  // (Cin <= 64) and the multi-tap transposed / k x k convs are better with 64
  const int ck = force ? force : ((p.n_taps == 1 && p.n_phases <= 1 && p.CinP >= 128) ? 32 : 64);
  if constexpr (PL) {  // (the NHWC stage layout is written for 128-byte pixel records: 64-channel chunks only)
    if (ck == 32) return launch_tap_shape<PL, RES, 32>(p, st);
  }
  return launch_tap_shape<PL, RES, 64>(p, st);
}

int launch_tap_pick(const TapParams& p, hipStream_t st) {
  // stride-1 1 x 1 convs on planes with >= 2 cout tiles of 128 and <= 128 input channels (ResNet's 64 -> 256 and 128 -> 512):
  // one workgroup per pixel tile walks the cout tiles (tapconv_coloop_kernel; 0.194 -> 0.162 ms on 128 -> 512 + residual @128^2
  // x 16). Measured and left to the per-(pixel tile, cout tile) kernel: 256 input channels (80 KiB of LDS, one workgroup per CU:
  // 0.106 -> 0.108 ms; as 64-pixel x 256-cout tiles, 64 KiB: 0.116) and stride 2 (0.168 -> 0.178) -- profiles/r04_tapconv_sweep.md.
  // SA_TAP_COLOOP=0: off (A/B)
  static const bool coloop = [] {
    const char* v = getenv("SA_TAP_COLOOP");
    return !v || atoi(v) != 0;
  }();
  if (coloop && p.planar && p.n_taps == 1 && p.n_phases <= 1 && p.ksize == 0 && p.in_stride == 1 && p.CoutP >= 256 && p.CinP >= 64 &&
      p.CinP <= 128 && p.CinP % 32 == 0)
    return p.residual ? launch_tap_coloop<2, 2, true>(p, st) : launch_tap_coloop<2, 2, false>(p, st);
  if (p.planar) return p.residual ? launch_tap_ck<true, true>(p, st) : launch_tap_ck<true, false>(p, st);
  return p.residual ? launch_tap_ck<false, true>(p, st) : launch_tap_ck<false, false>(p, st);
}

int fill_common(TapParams& p, const void* src, int CinP, const void* w, const float* bias, int CoutP, int relu, int B,
                const float* post_scale, const float* post_shift, const void* residual, int relu_last, void* dst) {
  SA_REQUIRE(src && w && bias && dst, "tapconv: NULL pointer");
  SA_REQUIRE(CinP > 0 && CinP % 16 == 0 && CoutP > 0 && CoutP % 16 == 0, "tapconv: channels must be padded to multiples of 16");
  SA_REQUIRE(!post_scale == !post_shift, "tapconv: post_scale and post_shift come together");
  SA_REQUIRE(B > 0, "tapconv: bad batch");
  p.src = (const uint16_t*)src;
  p.w = (const uint16_t*)w;
  p.bias = bias;
  p.dst = (uint16_t*)dst;
  p.post_scale = post_scale;
  p.post_shift = post_shift;
  p.residual = (const uint16_t*)residual;
  p.CinP = CinP;
  p.CoutP = CoutP;
  p.B = B;
  p.relu = relu & 1;  // (`relu`: bit 0 = ReLU, SA_LAYOUT_PLANES16 = src / residual / dst are 16-channel planes)
  p.planar = (relu & SA_LAYOUT_PLANES16) ? 1 : 0;
  p.relu_last = relu_last;
  return SA_OK;
}

}  // namespace

extern "C" {

size_t sa_tapconv_packed_elems(int n_taps, int CinP, int CoutP) {
  return (size_t)((CoutP + 31) / 32) * n_taps * (CinP / 16) * 64 * 8;
}

int sa_pack_tapconv_weights(const float* w, int n_taps, int Cin, int CinP, int Cout, int CoutP, uint16_t* packed) {
  SA_REQUIRE(w && packed && n_taps > 0 && n_taps <= MAX_WINDOW_TAPS, "sa_pack_tapconv_weights: bad arguments");
  SA_REQUIRE(CinP % 16 == 0 && CoutP % 16 == 0 && Cin <= CinP && Cout <= CoutP, "sa_pack_tapconv_weights: bad channel padding");
  const int K16 = CinP / 16, co32_n = (CoutP + 31) / 32;
  for (int c32 = 0; c32 < co32_n; ++c32)
    for (int t = 0; t < n_taps; ++t)
      for (int k16 = 0; k16 < K16; ++k16)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            const int co = c32 * 32 + (lane & 31), ci = k16 * 16 + (lane >> 5) * 8 + j;
            float v = 0.0f;
            if (co < Cout && ci < Cin) v = w[((size_t)t * Cin + ci) * Cout + co];
            packed[((((size_t)c32 * n_taps + t) * K16 + k16) * 64 + lane) * 8 + j] = sa::f2h(v);
          }
  return SA_OK;
}

int sa_conv1x1_bf16(const void* src, int CinP, const void* w, const float* bias, int CoutP, int relu, int B, int Hs, int Ws,
                    int stride, const float* post_scale, const float* post_shift, const void* residual, int relu_last,
                    void* dst, sa_stream_t stream) {
  TapParams p = {};
  const int rc = fill_common(p, src, CinP, w, bias, CoutP, relu, B, post_scale, post_shift, residual, relu_last, dst);
  if (rc != SA_OK) return rc;
  SA_REQUIRE(Hs > 0 && Ws > 0 && (stride == 1 || stride == 2), "sa_conv1x1_bf16: bad shape / stride");
  SA_REQUIRE((size_t)Hs * Ws * CinP * 2 < 0x7fffffffull, "sa_conv1x1_bf16: one frame of the source must stay below 2 GiB");
  p.Hs = Hs;
  p.Ws = Ws;
  p.Hl = p.Ho = (Hs - 1) / stride + 1;  // Conv2D(k1, valid): floor((n - 1) / s) + 1
  p.Wl = p.Wo = (Ws - 1) / stride + 1;
  p.in_stride = stride;
  p.out_stride = 1;
  p.n_taps = 1;
  return launch_tap_pick(p, (hipStream_t)stream);
}

int sa_convk_bf16(const void* src, int CinP, const void* w, int ksize, const float* bias, int CoutP, int relu, int B, int H, int W,
                  const float* post_scale, const float* post_shift, const void* residual, int relu_last, void* dst,
                  sa_stream_t stream) {
  TapParams p = {};
  const int rc = fill_common(p, src, CinP, w, bias, CoutP, relu, B, post_scale, post_shift, residual, relu_last, dst);
  if (rc != SA_OK) return rc;
  SA_REQUIRE(H > 0 && W > 0 && ksize >= 1 && ksize * ksize <= MAX_WINDOW_TAPS, "sa_convk_bf16: bad shape / kernel size");
  SA_REQUIRE((size_t)H * W * CinP * 2 < 0x7fffffffull, "sa_convk_bf16: one frame of the source must stay below 2 GiB");
  p.Hs = p.Hl = p.Ho = H;
  p.Ws = p.Wl = p.Wo = W;
  p.in_stride = 1;
  p.out_stride = 1;
  p.n_taps = ksize * ksize;
  p.ksize = ksize;
  p.kpad = (ksize - 1) / 2;
  return launch_tap_pick(p, (hipStream_t)stream);
}

int sa_convt_s2_bf16(const void* src, int CinP, const void* const* w_phase, int ksize, const float* bias, int CoutP,
                     int relu, int B, int Hs, int Ws, const float* post_scale, const float* post_shift, int relu_last,
                     void* dst, sa_stream_t stream) {
  SA_REQUIRE(ksize == 3 || ksize == 4, "sa_convt_s2_bf16: kernel size must be 3 or 4");
  SA_REQUIRE(w_phase && Hs > 0 && Ws > 0, "sa_convt_s2_bf16: bad arguments");
  SA_REQUIRE((size_t)Hs * Ws * CinP * 2 < 0x7fffffffull, "sa_convt_s2_bf16: one frame of the source must stay below 2 GiB");
  // TF "same" Conv2DTranspose, stride 2: out[o] = sum_{i,k : 2i + k - c = o} x[i] w[k] with crop c = max(k - 2, 0) / 2
  // (k4: c = 1, k3: c = 0). For output parity a = o & 1 the contributing kernel rows are k = (a + c) & 1, +2, ... and
  // the input row is i = (o + c - k) / 2 = lo + (a + c - k) / 2.
  const int c = ksize == 4 ? 1 : 0;
  TapParams p = {};
  const int rc = fill_common(p, src, CinP, w_phase[0], bias, CoutP, relu, B, post_scale, post_shift, nullptr, relu_last, dst);
  if (rc != SA_OK) return rc;
  p.Hs = Hs;
  p.Ws = Ws;
  p.Hl = Hs;
  p.Wl = Ws;
  p.Ho = 2 * Hs;
  p.Wo = 2 * Ws;
  p.in_stride = 1;
  p.out_stride = 2;
  p.n_phases = 4;  // all four output phases in one grid: 4x the workgroups of a per-phase launch, no serialisation
  for (int a = 0; a < 2; ++a)
    for (int bq = 0; bq < 2; ++bq) {
      const int ph = a * 2 + bq;
      SA_REQUIRE(w_phase[ph], "sa_convt_s2_bf16: NULL phase weights");
      p.ph_w[ph] = (const uint16_t*)w_phase[ph];
      p.ph_oy[ph] = a;
      p.ph_ox[ph] = bq;
      int n = 0;
      for (int ky = (a + c) & 1; ky < ksize; ky += 2)
        for (int kx = (bq + c) & 1; kx < ksize; kx += 2) {
          p.ph_dy[ph][n] = (a + c - ky) / 2;  // exact: a + c - ky is even
          p.ph_dx[ph][n] = (bq + c - kx) / 2;
          ++n;
        }
      p.ph_ntaps[ph] = n;
    }
  p.n_taps = 4;
  return launch_tap_pick(p, (hipStream_t)stream);
}

int sa_convt_s2_phase_taps(int ksize, int phase, int* ky, int* kx) {
  // kernel taps (ky, kx) used by output phase (a, b) = (phase >> 1, phase & 1), in the order sa_convt_s2_bf16 expects
  // the slices of w_phase[phase]; returns the tap count
  if (ksize != 3 && ksize != 4) return 0;
  const int c = ksize == 4 ? 1 : 0, a = phase >> 1, b = phase & 1;
  int n = 0;
  for (int y = (a + c) & 1; y < ksize; y += 2)
    for (int x = (b + c) & 1; x < ksize; x += 2) {
      ky[n] = y;
      kx[n] = x;
      ++n;
    }
  return n;
}

}  // extern "C"
