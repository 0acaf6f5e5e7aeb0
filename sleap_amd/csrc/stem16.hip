// The two full-resolution layers of SLEAP's UNet encoder for filters <= 16 in ONE kernel, on 16x16x32 MFMA:
//   ensure_float (u8 * 1/255, normalization.py:49) -> Conv2D(k3, Cin in {1,3} -> C0<=16)+bias+ReLU
//   -> Conv2D(k3, C0 -> C1<=16)+bias+ReLU -> [full-resolution store] and/or [MaxPool2D(2) store]
// (encoder_decoder.py:109-131: block 0's two convs and block 1's leading max-pool).
//
// These layers carry 5 % of the network's FLOPs but would move ~40 % of its activation bytes if each stored its
// output; here the first conv's activation lives only in LDS and, when the consumer is the next block's pool,
// only the pooled quarter-size tensor is written.
//
// Why a dedicated kernel: with 16 output channels a 32x32x16 MFMA wastes half its rows; v_mfma_f32_16x16x32_bf16
// (M = 16 channels, N = 16 pixels, K = 32) fits exactly: conv0's 9*Cin <= 27 taps are ONE K=32 step, conv1 packs
// two taps x 16 channels per step (5 steps for 9 taps). All weights live in registers for the whole workgroup.
//
// conv0 precision: the B operand holds raw pixel values 0..255 (exact in bf16); the fp32 weights times 1/255 are
// split into three bf16 terms (hi + mid + lo carries the full 24-bit mantissa), three MFMAs with fp32 accumulation.
#include <cstdint>
#include <type_traits>

#include "bf16.h"
#include "sa_common.h"

namespace {

using sa::mfma_h8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
using sa::h16x8_t;

// conv0 weight terms per MFMA group: the fp32 weights (x 1/255 / U8_ACT_SCALE) enter as a sum of 16-bit terms. bf16 (8
// mantissa bits) needs hi + mid + lo for fp32's 24 bits; fp16 (11 bits) carries 22 in hi + mid -- what is left (2^-22
// relative) is 2000x below the fp16 rounding of the stored activation, so the fp16 build issues two MFMAs per group, not three.
#if defined(SA_STEM16_TERMS_OVERRIDE)
#define SA_STEM16_TERMS SA_STEM16_TERMS_OVERRIDE
#elif defined(SA_HALF_FP16)
#define SA_STEM16_TERMS 2
#else
#define SA_STEM16_TERMS 3
#endif

struct Stem16Params {
  const uint8_t* src;   // [B,H,W,CIN] u8
  const uint16_t* blob;  // sa_stem16_pack: wa[3][64][8] | wb[5][64][8] bf16 | bias0[16] | bias1[16] f32
  uint16_t* dst;        // [B,H,W,16] bf16 or nullptr
  uint16_t* dst_pool;   // [B,H/2,W/2,16] bf16 or nullptr
  int B, H, W, relu0, relu1, tiles_x, tiles_y;
};

#if !defined(SA_STEM16_FOLD)
#define SA_STEM16_FOLD 1  // gray kernel: hi + mid weight terms of conv0 in one MFMA (0: one MFMA per term, A/B)
#endif
#if !defined(SA_STEM16_PLANES8)
#define SA_STEM16_PLANES8 1  // gray kernel: conv0 activation tile as two 8-channel planes in LDS (0: pixel-major records, A/B)
#endif
#if !defined(SA_STEM16_TABLE_B128)
#define SA_STEM16_TABLE_B128 1  // gray kernel: the triplet table written 16 bytes at a time (0: 8 bytes, A/B)
#endif
#if !defined(SA_STEM16_READ2)
// gray kernel, folded conv0 (round 6): the B operand {triplet, triplet} comes from ONE `ds_read2_b64` whose two offsets are
// equal -- the LDS returns the same 8 bytes into both register pairs -- instead of a ds_read_b64 and two v_mov (18 of conv0's
// ~60 VALU instructions per wave and tile). MEASURED SLOWER (kernel trace, three alternating runs on one box: 490.9 against
// 488.5 us; with the DPP max below 479.7 against 476.4): the LDS serves the two addresses of a read2 one after the other, and
// the array (0.54 busy) is the more loaded unit. 0 = the round 2-5 form = the default; 1 kept as the A/B switch.
#define SA_STEM16_READ2 0
#endif
// conv1's K = 32 steps: step s multiplies the 16 channels of TWO taps (k-blocks 0,1 = tap PAIRS[s][0], k-blocks 2,3 = tap
// PAIRS[s][1]; -1 = zero weights). Round 4: the pairs follow the kernel's rows and columns instead of the tap index (round 1-3:
// taps 2s, 2s + 1), so that a B fragment is the SAME data for up to three output rows:
//   steps 0-2 ("G", kernel row dy): taps (dy,0) | (dy,1) -- halo row R, columns 0 | 1: used by output rows R, R-1, R-2
//   step 3   ("F0"): taps (0,2) | (1,2)                  -- column 2 of halo rows R | R+1: output row R
//   step 4   ("F1"): zero | (2,2)                        -- the same fragment: output row R-1
// A wave's four output rows then read 6 x 2 G + 5 x 2 F = 22 fragments from LDS instead of 40 for the same 40 MFMAs. The
// counters (profiles/r03_pmc_sq_counters.md) and a cycle count per tile put this kernel at the LDS array: ~1900 LDS cycles per
// tile (1280 of them conv1's fragment reads) against 800 matrix-core cycles per SIMD.
// (the round 1-3 pairing -- taps 2s | 2s + 1, one fragment read per MFMA -- measured against this one on one box: 0.502 vs 0.474 ms,
// profiles/r04_ab_session.md section 1; the A/B switch is gone)
__host__ __device__ constexpr int stem16_pair_tap(int s, int half) {
  return s < 3 ? 3 * s + half : (s == 3 ? (half ? 5 : 2) : (half ? 8 : -1));
}
// tile: 16 rows x 32 columns of output per workgroup (4 waves x 4 rows)
#define SA_STEM16_TH 16
#define SA_STEM16_TW 32

template <int CIN>
__global__ void __launch_bounds__(256)
stem16_kernel(const Stem16Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int TH = SA_STEM16_TH, TW = SA_STEM16_TW, PH = TH + 2, PW = TW + 2, RH = TH + 4, RW = TW + 4;
  // LDS: raw image tile as bf16 [RH][RW][CIN]; conv0 activation tile [PH*PW pixels][16 ch] bf16 with a pixel
  // stride of 48 B (32 B data + 16 B pad): 3 sixteen-byte slots per pixel is odd, so the 16 pixels of a
  // ds_read_b128 lane group fall on 16 distinct slots (conflict-free) and addresses stay base + immediate.
  constexpr int APIX = 48;
  __shared__ __attribute__((aligned(16))) uint16_t raw[RH * RW * CIN + 8];
  __shared__ __attribute__((aligned(16))) unsigned char act[PH * PW * APIX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n16 = lane & 15, kb = lane >> 4;
  int bid = blockIdx.x;
  const int tx_i = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int ty_i = bid % p.tiles_y;
  const int b = bid / p.tiles_y;
  const int x0 = tx_i * TW, y0 = ty_i * TH;
  const int H = p.H, W = p.W;

  // ---- raw tile (zero outside the image = conv0's SAME padding), pixel values x U8_ACT_SCALE in the storage type (exact for 0..255)
  if (CIN == 1 && (W & 3) == 0) {
    // aligned dword loads: columns x0-4 .. x0+35 (10 dwords per row); raw column tx corresponds to gx = x0 + tx - 2
    for (int i = tid; i < RH * 10; i += 256) {
      const int ty = i / 10, dq = i - ty * 10;
      const int gy = y0 + ty - 2, gx = x0 - 4 + dq * 4;
      unsigned v = 0;
      if (gy >= 0 && gy < H && gx >= 0 && gx < W)
        v = *reinterpret_cast<const unsigned*>(p.src + ((size_t)b * H + gy) * W + gx);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int tx = dq * 4 + e - 2;
        if (tx >= 0 && tx < RW) raw[ty * RW + tx] = sa::f2h((float)((v >> (8 * e)) & 0xFF) * sa::U8_ACT_SCALE);
      }
    }
  } else {
    for (int i = tid; i < RH * RW * CIN; i += 256) {
      const int c = i % CIN, px = i / CIN;
      const int ty = px / RW, tx = px - ty * RW;
      const int gy = y0 + ty - 2, gx = x0 + tx - 2;
      float v = 0.0f;
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = (float)p.src[(((size_t)b * H + gy) * W + gx) * CIN + c];
      raw[i] = sa::f2h(v * sa::U8_ACT_SCALE);
    }
  }
  // ---- weights: MFMA A fragments packed per lane on the host (sa_stem16_pack), register resident
  const uint4* blob4 = reinterpret_cast<const uint4*>(p.blob);
  mfma_h8 wa[3], wb[5];
#pragma unroll
  for (int i = 0; i < 3; ++i) wa[i] = __builtin_bit_cast(mfma_h8, blob4[i * 64 + lane]);
#pragma unroll
  for (int i = 0; i < 5; ++i) wb[i] = __builtin_bit_cast(mfma_h8, blob4[(3 + i) * 64 + lane]);
  const float* biases = reinterpret_cast<const float*>(p.blob + 8 * 64 * 8);
  const float4 q0 = *reinterpret_cast<const float4*>(biases + kb * 4);  // D rows of this lane: couts kb*4 .. kb*4+3
  const float4 q1 = *reinterpret_cast<const float4*>(biases + 16 + kb * 4);
  const float bias0[4] = {q0.x, q0.y, q0.z, q0.w}, bias1[4] = {q1.x, q1.y, q1.z, q1.w};
  __syncthreads();

  // ---- conv0 on the (PH x PW) halo tile, 16 pixels per MFMA group
  constexpr int NG0 = (PH * PW + 15) / 16;
  for (int g = wave; g < NG0; g += 4) {
    const int pl = g * 16 + n16;
    const bool valid = pl < PH * PW;
    const int plc = valid ? pl : 0;
    const int ty = plc / PW, tx = plc - ty * PW;
    h16x8_t bq = {0, 0, 0, 0, 0, 0, 0, 0};
    if (CIN == 1) {
      if (kb < 3) {
        const uint16_t* r = raw + (ty + kb) * RW + tx;
        bq[0] = r[0];
        bq[1] = r[1];
        bq[2] = r[2];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int f = kb * 8 + j;  // flat tap*CIN + c
        if (f < 9 * CIN) {
          const int tap = f / CIN, c = f - tap * CIN;
          bq[j] = raw[((ty + tap / 3) * RW + tx + tap % 3) * CIN + c];
        }
      }
    }
    const mfma_h8 bf = __builtin_bit_cast(mfma_h8, bq);
    f32x4 d = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int t = 0; t < SA_STEM16_TERMS; ++t) d = SA_MFMA_16x16x32(wa[t], bf, d, 0, 0, 0);
    const int gy = y0 + ty - 1, gx = x0 + tx - 1;
    const bool in_img = valid && gy >= 0 && gy < H && gx >= 0 && gx < W;  // outside: conv1's SAME padding = 0
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float t = d[j] + bias0[j];
      if (p.relu0) t = fmaxf(t, 0.0f);
      v[j] = in_img ? t : 0.0f;
    }
    if (valid) {
      uint2 o;
      o.x = sa::f2h2(v[0], v[1]);
      o.y = sa::f2h2(v[2], v[3]);
      *reinterpret_cast<uint2*>(act + pl * APIX + kb * 8) = o;
    }
  }
  __syncthreads();

  // ---- conv1: wave w owns rows 4w..4w+3, two 16-pixel groups per row
  f32x4 acc[4][2];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int h = 0; h < 2; ++h) acc[r][h] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int t0 = stem16_pair_tap(s, 0) < 0 ? 0 : stem16_pair_tap(s, 0), t1 = stem16_pair_tap(s, 1);  // (a zero-weight half reads any valid address)
    const int tap = (kb >> 1) ? t1 : t0;
    const int dy = tap / 3, dx = tap % 3;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int pl = (wave * 4 + r + dy) * PW + h * 16 + n16 + dx;
        const mfma_h8 bv = *reinterpret_cast<const mfma_h8*>(act + pl * APIX + (kb & 1) * 16);
        acc[r][h] = SA_MFMA_16x16x32(wb[s], bv, acc[r][h], 0, 0, 0);
      }
  }

  // ---- epilogue: lane holds couts kb*4..+3 of pixel (row, h*16 + n16)
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int gx = x0 + h * 16 + n16;
    float v[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float t = acc[r][h][j] + bias1[j];
        v[r][j] = p.relu1 ? fmaxf(t, 0.0f) : t;
      }
    if (p.dst) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gy = y0 + wave * 4 + r;
        if (gy < H && gx < W) {
          uint2 o;
          o.x = sa::f2h2(v[r][0], v[r][1]);
          o.y = sa::f2h2(v[r][2], v[r][3]);
          *reinterpret_cast<uint2*>(p.dst + (((size_t)b * H + gy) * W + gx) * 16 + kb * 4) = o;
        }
      }
    }
    if (p.dst_pool) {
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        float t4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float t = fmaxf(v[r][j], v[r + 1][j]);
          t4[j] = fmaxf(t, __shfl_xor(t, 1));
        }
        const int gy = y0 + wave * 4 + r;
        if (!(lane & 1) && gy < H && gx < W) {
          uint2 o;
          o.x = sa::f2h2(t4[0], t4[1]);
          o.y = sa::f2h2(t4[2], t4[3]);
          *reinterpret_cast<uint2*>(p.dst_pool + (((size_t)b * (H / 2) + gy / 2) * (W / 2) + gx / 2) * 16 + kb * 4) = o;
        }
      }
    }
  }
#endif
}


// ---- v2 (grayscale, W % 4 == 0): same arithmetic, re-laid-out for the LDS and VALU budgets the counters showed
// to be binding (profiles/r01_pmc_sq_counters.md: LDS array 83 % busy, VALU issue 84 %):
//   * the raw tile is stored as a table of horizontal tap triplets {p[x], p[x+1], p[x+2], 0} (8 bytes per entry, bf16),
//     so conv0's B operand is ONE aligned ds_read_b64 per lane instead of three 16-bit reads;
//   * the conv0 activation tile is dense (32 B per pixel): with the hardware's ds_read_b128 lane groups
//     ({0-3,12-15,20-27}, ...) the dense layout is conflict-free for conv1's reads, the 48-byte stride was not;
//   * pixel coordinates advance incrementally (no per-group division), bias1 is the accumulator's initial value, and
//     the ReLU of a pooled-only output is applied after the 2x2 max (max commutes with ReLU).
__global__ void __launch_bounds__(256)
stem16_gray_kernel(const Stem16Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int TH = SA_STEM16_TH, TW = SA_STEM16_TW, PH = TH + 2, PW = TW + 2, RH = TH + 4;
  constexpr int RS = 48;  // triplet-table row stride in entries: 96 dwords = 32 (mod 64) banks between kernel rows
  __shared__ __attribute__((aligned(16))) uint2 rawt[(RH + 1) * RS];  // row RH: entry 0 is the zero operand
#if SA_STEM16_PLANES8
  // conv0 activation tile as TWO planes of 8 channels, 16 bytes per pixel (round 3). The pixel-major record of 32 bytes put the
  // 16 lanes of a ds_write_b64 group (16 consecutive columns, one group of 4 couts) 8 banks apart: 4 distinct bank pairs, a
  // 4-way conflict on every conv0 store (16 LDS cycles instead of 6) -- the largest single item on the most loaded unit of this
  // kernel. With 16-byte records the group spans 64 banks (2-way: 8 cycles), and conv1's ds_read_b128 groups
  // ({0-3,12-15 | 20-27}: one k-half from plane 0, the other from plane 1) stay conflict-free because a plane is a
  // multiple of 256 bytes.
  constexpr int APLANE = ((PH * PW + 16) * 16 + 255) / 256 * 256;
  __shared__ __attribute__((aligned(16))) unsigned char act[2 * APLANE];
#else
  __shared__ __attribute__((aligned(16))) unsigned char act[(PH * PW + 16) * 32];
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kb = lane >> 4;
  // (3-D grid: tile column, tile row, frame -- two runtime divisions per workgroup, each a dependent ~100-cycle reciprocal
  //  sequence in FRONT of the first load, were the first thing every tile did)
  const int tx_i = blockIdx.x, ty_i = blockIdx.y, b = blockIdx.z;
  const int x0 = tx_i * TW, y0 = ty_i * TH;
  const int H = p.H, W = p.W;

  // ---- weights and biases first: the requests go out before the pixel loads below and come back under them. (They used to
  // follow the table build in program order, i.e. a second, serialized L2 round trip in every tile's prologue.)
  const uint4* blob4 = reinterpret_cast<const uint4*>(p.blob);
  mfma_h8 wa[3], wb[5];
#pragma unroll
  for (int i = 0; i < 3; ++i) wa[i] = __builtin_bit_cast(mfma_h8, blob4[i * 64 + lane]);
#pragma unroll
  for (int i = 0; i < 5; ++i) wb[i] = __builtin_bit_cast(mfma_h8, blob4[(3 + i) * 64 + lane]);
  const float* biases = reinterpret_cast<const float*>(p.blob + 8 * 64 * 8);
  const float4 q0 = *reinterpret_cast<const float4*>(biases + kb * 4);
  const float4 q1 = *reinterpret_cast<const float4*>(biases + 16 + kb * 4);
  const float bias0[4] = {q0.x, q0.y, q0.z, q0.w}, bias1[4] = {q1.x, q1.y, q1.z, q1.w};

  // ---- triplet table: thread (row, dq) loads the two aligned dwords covering image columns x0-4+4dq .. +7 and emits
  // the entries of raw columns tx = 4dq-2 .. 4dq+1 (raw column tx <-> image column x0 - 2 + tx; zero outside the image)
  if (tid == 255) rawt[RH * RS] = make_uint2(0u, 0u);
#if SA_STEM16_FOLD && SA_STEM16_READ2
  // entry RS - 1 of every table row (never a pixel: raw columns end at PW + 1 < RS - 1) is a zero operand that is reached with
  // the SAME row stride as the pixel entries, so that all lanes of a conv0 group read with one compile-time offset
  if (tid >= 192 && tid < 192 + RH) rawt[(tid - 192) * RS + RS - 1] = make_uint2(0u, 0u);
#endif
  if (tid < RH * 9) {
    const int ty = tid / 9, dq = tid - ty * 9;
    const int gy = y0 + ty - 2, gx = x0 - 4 + dq * 4;
    unsigned v0 = 0, v1 = 0;
    if (gy >= 0 && gy < H) {
      const uint8_t* row = p.src + ((size_t)b * H + gy) * W;
      if (gx >= 0 && gx < W) v0 = *reinterpret_cast<const unsigned*>(row + gx);
      if (gx + 4 >= 0 && gx + 4 < W) v1 = *reinterpret_cast<const unsigned*>(row + gx + 4);
    }
    unsigned h[6];  // storage bits of pixels 0..5 of the 8 (0..255 x U8_ACT_SCALE: exact in bf16 and fp16)
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      const unsigned byte = e < 4 ? (v0 >> (8 * e)) & 0xFF : (v1 >> (8 * (e - 4))) & 0xFF;
      h[e] = sa::f2h((float)byte * sa::U8_ACT_SCALE);
    }
#if SA_STEM16_TABLE_B128
    // (round 6) the four entries of a thread are 32 contiguous, 16-byte aligned bytes: two ds_write_b128 (lanes 32 bytes apart:
    // 2-way) instead of four ds_write_b64 (4-way: 144 of this kernel's 343 conflict cycles per tile). Entries come in valid pairs
    // (tx = -2, -1 at the left edge; tx <= 33 < PW + 2 at the right: entries 34, 35 of a row are never read).
#pragma unroll
    for (int e = 0; e < 4; e += 2) {
      const int tx = dq * 4 - 2 + e;
      if (tx >= 0 && tx < PW)
        *reinterpret_cast<uint4*>(&rawt[ty * RS + tx]) = make_uint4(h[e] | (h[e + 1] << 16), h[e + 2], h[e + 1] | (h[e + 2] << 16), h[e + 3]);
    }
#else
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int tx = dq * 4 - 2 + e;
      if (tx >= 0 && tx < PW) rawt[ty * RS + tx] = make_uint2(h[e] | (h[e + 1] << 16), h[e + 2]);
    }
#endif
  }
#if SA_STEM16_FOLD
  // A lane's conv0 K block holds one kernel row: 3 taps in 8 slots. The hi and the mid term of the weight split share ONE MFMA:
  // hi in slots 0-2, mid in slots 4-6, the pixel triplet in both halves of the B operand (products and fp32 accumulation are
  // the same, only their order inside the matrix core differs). fp16 build: 1 MFMA per 16 halo pixels instead of 2 (a third of
  // this kernel's matrix-core cycles went to conv0's 6 % of its FLOPs); bf16 build: 2 (hi+mid, lo) instead of 3.
  constexpr int NMF0 = SA_STEM16_TERMS > 1 ? SA_STEM16_TERMS - 1 : 1;
  mfma_h8 wf[NMF0];
  {
    const uint4 wh = __builtin_bit_cast(uint4, wa[0]), wm = __builtin_bit_cast(uint4, wa[SA_STEM16_TERMS > 1 ? 1 : 0]);
    wf[0] = SA_STEM16_TERMS > 1 ? __builtin_bit_cast(mfma_h8, make_uint4(wh.x, wh.y, wm.x, wm.y)) : wa[0];
    if constexpr (NMF0 > 1) wf[NMF0 - 1] = wa[2];
  }
#else
  constexpr int NMF0 = SA_STEM16_TERMS;
  const mfma_h8(&wf)[3] = wa;
#endif
  __syncthreads();

  // ---- conv0 on the (PH x PW) halo tile. Columns 0..31 of every halo row are two 16-pixel groups: wave w takes column
  // segment w & 1 of rows (w >> 1), (w >> 1) + 2, ... -- every per-lane quantity (table address, output address, column
  // validity) is loop invariant up to a compile-time stride and the row validity is wave uniform, so a group costs its three
  // MFMAs plus ~10 VALU instructions (the former "16 consecutive halo pixels per group" mapping needed 34 for its
  // lane-dependent row / column walk, and this loop was the VALU-bound half of the kernel: 0.80 -> 0.63-0.68 ms).
  // Columns 32, 33 (36 pixels) are three more groups, done by waves 0-2 afterwards. (Measured and dropped: pairing two
  // groups with v_permlane16_swap into one 16-byte store per lane -- 2-way instead of 2 x 4-way bank conflicts -- 0.60 -> 0.63 ms.)
  // VALU diet (round 2; the counters put this kernel at 0.72 of its VALU issue slots against 0.45 of the matrix-core cycles, and
  // this loop held 11.5 VALU instructions per group of two MFMAs):
  //   * tiles whose whole halo lies inside the image (all but the border ring) skip the per-value padding select;
  //   * fp16 build: ReLU on the PACKED pair after the conversion (one v_pk_max_f16 for two values instead of two v_max_f32);
  //   * the MFMA B operand is a register quad whose upper half is zero: kept as three persistent quads whose upper halves are
  //     zeroed once, the table read fills the lower halves (the compiler re-zeroed two registers per group before).
  const float low0 = p.relu0 ? 0.0f : -__builtin_huge_valf();
  const bool interior = x0 >= 1 && y0 >= 1 && x0 + PW - 1 <= W && y0 + PH - 1 <= H;  // wave (workgroup) uniform
#if SA_HAS_PK_MAX
  const uint32_t lowpk = p.relu0 ? 0u : SA_PK_NEG_INF;
#endif
  auto conv0_store = [&](const f32x4 d, bool masked, bool in_img, unsigned char* dstp) {
#if SA_HAS_PK_MAX
    uint32_t a = sa::pk_max(sa::f2h2(d[0], d[1]), lowpk), b = sa::pk_max(sa::f2h2(d[2], d[3]), lowpk);
#else
    uint32_t a = sa::f2h2(fmaxf(d[0], low0), fmaxf(d[1], low0)), b = sa::f2h2(fmaxf(d[2], low0), fmaxf(d[3], low0));
#endif
    if (masked) {  // (compile-time in the two copies of the loop below) outside the image: conv1's SAME padding = 0
      const unsigned m = in_img ? 0xFFFFFFFFu : 0u;
      a &= m;
      b &= m;
    }
    *reinterpret_cast<uint2*>(dstp) = make_uint2(a, b);
  };
  (void)low0;
  {
    const int seg = wave & 1, tx = seg * 16 + n16;
    const bool colok = (unsigned)(x0 + tx - 1) < (unsigned)W;
    // lanes kb == 3 carry K slots 24..31 (no taps): they read the zero entry with stride 0
#if SA_STEM16_FOLD && SA_STEM16_READ2
    const uint2* src = kb < 3 ? rawt + ((wave >> 1) + kb) * RS + tx : rawt + (wave >> 1) * RS + RS - 1;
    // LDS byte addresses of groups 0, 3, 6 (ds_read2_b64 offsets are 8 bits of 8-byte units: three groups of 2 * RS entries each)
    typedef __attribute__((address_space(3))) const void* lds_cptr_t;
    const unsigned sa0 = (unsigned)(uintptr_t)(lds_cptr_t)src;
    static_assert(2 * 2 * RS <= 255, "ds_read2_b64 offset range");
#else
    const uint2* src = kb < 3 ? rawt + ((wave >> 1) + kb) * RS + tx : rawt + RH * RS;
    const int sstep = kb < 3 ? 2 * RS : 0;
#endif
#if SA_STEM16_PLANES8
    unsigned char* dstp = act + (kb >> 1) * APLANE + ((wave >> 1) * PW + tx) * 16 + (kb & 1) * 8;
#else
    unsigned char* dstp = act + ((wave >> 1) * PW + tx) * 32 + kb * 8;
#endif
    constexpr int APIX = SA_STEM16_PLANES8 ? 16 : 32;  // bytes between neighbouring pixels of the activation tile
    auto rowok = [&](int it) { return (unsigned)(y0 + (wave >> 1) + 2 * it - 1) < (unsigned)H; };  // wave uniform
    // three groups at a time: the hi / mid / lo MFMAs of one group depend on each other, those of different groups do not
    uint4 opq[3] = {make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u)};
    (void)opq;
#if SA_STEM16_FOLD && SA_STEM16_READ2
    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
    u32x4v oqs[2][3];
#endif
    auto rows = [&](auto masked_c) {  // the loop exists twice: with and without the padding select (a REAL branch on `interior`)
      constexpr bool MASKED = decltype(masked_c)::value;
#pragma unroll
      for (int it = 0; it < PH / 2; it += 3) {
        f32x4 d[3];
#if SA_STEM16_FOLD && SA_STEM16_READ2
        // Three groups' operands are requested one batch AHEAD (under the previous batch's conversions and stores). The reads
        // are opaque to the compiler's wait-count pass: its own waits can only come out stricter (the counter retires in order);
        // ours is "at most the three newest LGKM operations outstanding" = the next batch's reads, whatever number of ds_write
        // instructions the compiler made of the stores in between, and it carries the operands it releases, so no MFMA can be
        // scheduled above it.
#define SA_STEM16_REQUEST(q, adr)                                                                                        \
  do {                                                                                                                   \
    const unsigned adr_ = (adr);                                                                                         \
    asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%2" : "=v"((q)[0]) : "v"(adr_), "n"(0) : "memory");            \
    asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%2" : "=v"((q)[1]) : "v"(adr_), "n"(2 * RS) : "memory");       \
    asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%2" : "=v"((q)[2]) : "v"(adr_), "n"(4 * RS) : "memory");       \
  } while (0)
        u32x4v(&oq)[3] = oqs[(it / 3) & 1];
        if (it == 0) SA_STEM16_REQUEST(oq, sa0);
        if (it + 3 < PH / 2) {
          SA_STEM16_REQUEST(oqs[(it / 3 + 1) & 1], sa0 + (it + 3) * 2 * RS * 8);
          asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(oq[0]), "+v"(oq[1]), "+v"(oq[2])::"memory");
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(oq[0]), "+v"(oq[1]), "+v"(oq[2])::"memory");
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) d[u] = (f32x4){bias0[0], bias0[1], bias0[2], bias0[3]};
#else
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const uint2 tq = src[(it + u) * sstep];
          opq[u].x = tq.x;
          opq[u].y = tq.y;
#if SA_STEM16_FOLD
          opq[u].z = tq.x;
          opq[u].w = tq.y;
#endif
          d[u] = (f32x4){bias0[0], bias0[1], bias0[2], bias0[3]};
        }
#endif
#pragma unroll
        for (int t = 0; t < NMF0; ++t)
#pragma unroll
          for (int u = 0; u < 3; ++u)
#if SA_STEM16_FOLD && SA_STEM16_READ2
            d[u] = SA_MFMA_16x16x32(wf[t], __builtin_bit_cast(mfma_h8, oq[u]), d[u], 0, 0, 0);
#else
            d[u] = SA_MFMA_16x16x32(wf[t], __builtin_bit_cast(mfma_h8, opq[u]), d[u], 0, 0, 0);
#endif
#pragma unroll
        for (int u = 0; u < 3; ++u) conv0_store(d[u], MASKED, MASKED && colok && rowok(it + u), dstp + (it + u) * (2 * PW * APIX));
      }
    };
    if (__builtin_amdgcn_readfirstlane((int)interior))
      rows(std::false_type{});
    else
      rows(std::true_type{});
    if (wave < 3) {  // columns 32, 33: pixel q = 16 * wave + n16 of the 36 -> row q >> 1, column 32 + (q & 1)
      const int q = wave * 16 + n16;
      if (q < 2 * PH) {
        const int ty2 = q >> 1, tx2 = 32 + (q & 1);
        const uint2 tq = kb < 3 ? rawt[(ty2 + kb) * RS + tx2] : make_uint2(0u, 0u);
        const mfma_h8 bf = __builtin_bit_cast(mfma_h8, SA_STEM16_FOLD ? make_uint4(tq.x, tq.y, tq.x, tq.y) : make_uint4(tq.x, tq.y, 0u, 0u));
        f32x4 d = {bias0[0], bias0[1], bias0[2], bias0[3]};
#pragma unroll
        for (int t = 0; t < NMF0; ++t) d = SA_MFMA_16x16x32(wf[t], bf, d, 0, 0, 0);
        conv0_store(d, true, (unsigned)(y0 + ty2 - 1) < (unsigned)H && (unsigned)(x0 + tx2 - 1) < (unsigned)W,
                    SA_STEM16_PLANES8 ? act + (kb >> 1) * (sizeof(act) / 2) + (ty2 * PW + tx2) * 16 + (kb & 1) * 8
                                      : act + (ty2 * PW + tx2) * 32 + kb * 8);  // columns 32, 33
      }
    }
  }
  __syncthreads();

  // ---- conv1: wave w owns rows 4w..4w+3, two 16-pixel groups per row; A fragment s covers the tap pair stem16_pair_tap(s, .)
  f32x4 acc[4][2];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int h = 0; h < 2; ++h) acc[r][h] = (f32x4){bias1[0], bias1[1], bias1[2], bias1[3]};
#if SA_STEM16_PLANES8
  const unsigned char* abase = act + (kb & 1) * APLANE + ((wave * 4) * PW + n16) * 16;
#else
  const unsigned char* abase = act + ((wave * 4) * PW + n16) * 32 + (kb & 1) * 16;
#endif
  constexpr int APIX1 = SA_STEM16_PLANES8 ? 16 : 32;
  // fragment reuse (see stem16_pair_tap): halo row R of the wave's six feeds output rows R, R-1, R-2
  const unsigned char* gb = abase + (kb >> 1) * APIX1;                   // G: columns 0 | 1 of halo row R
  const unsigned char* fb = abase + ((kb >> 1) ? PW + 2 : 2) * APIX1;    // F: column 2 of halo rows R | R + 1
#pragma unroll
  for (int R = 0; R < 6; ++R) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const mfma_h8 g = *reinterpret_cast<const mfma_h8*>(gb + (R * PW + h * 16) * APIX1);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int r = R - dy;
        if (r >= 0 && r < 4) acc[r][h] = SA_MFMA_16x16x32(wb[dy], g, acc[r][h], 0, 0, 0);
      }
      if (R < 5) {
        const mfma_h8 f = *reinterpret_cast<const mfma_h8*>(fb + (R * PW + h * 16) * APIX1);
        if (R < 4) acc[R][h] = SA_MFMA_16x16x32(wb[3], f, acc[R][h], 0, 0, 0);
        if (R >= 1) acc[R - 1][h] = SA_MFMA_16x16x32(wb[4], f, acc[R - 1][h], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: lane holds couts kb*4..+3 of pixel (row, h*16 + n16). ReLU is one v_max against a uniform bound
  // (0 or -inf); every address is one 64-bit base per lane plus compile-time offsets.
  const float low1 = p.relu1 ? 0.0f : -__builtin_huge_valf();
  const int gy0 = y0 + wave * 4, gx0 = x0 + n16;
  if (p.dst) {
    uint16_t* dp = p.dst + (((size_t)b * H + gy0) * W + gx0) * 16 + kb * 4;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (gy0 + r < H && gx0 + h * 16 < W)
          *reinterpret_cast<uint2*>(dp + ((size_t)r * W + h * 16) * 16) =
              make_uint2(sa::f2h2(fmaxf(acc[r][h][0], low1), fmaxf(acc[r][h][1], low1)),
                         sa::f2h2(fmaxf(acc[r][h][2], low1), fmaxf(acc[r][h][3], low1)));
  }
  if (p.dst_pool) {
    uint16_t* pp = p.dst_pool + (((size_t)b * (H / 2) + gy0 / 2) * (W / 2) + gx0 / 2) * 16 + kb * 4;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        float t4[4];
        float t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = fmaxf(fmaxf(acc[r][h][j], acc[r + 1][h][j]), low1);  // relu(max(.)) == max(relu(.))
        sa::max_xor1_x4(t, t4);
        if (!(lane & 1) && gy0 + r < H && gx0 + h * 16 < W)
          *reinterpret_cast<uint2*>(pp + ((size_t)(r / 2) * (W / 2) + h * 8) * 16) =
              make_uint2(sa::f2h2(t4[0], t4[1]), sa::f2h2(t4[2], t4[3]));
      }
  }
#endif
}

}  // namespace

extern "C" {

size_t sa_stem16_blob_bytes(void) { return 8 * 64 * 8 * 2 + 32 * 4; }

// HOST: Keras kernels conv0 (3,3,Cin,C0) f32 / conv1 (3,3,C0,C1) f32 + biases -> the per-lane MFMA A-fragment blob:
//   wa[3][64][8] bf16: conv0 weights * 1/255 / U8_ACT_SCALE split hi/mid/lo; lane l -> cout l&15, k = (l>>4)*8 + j
//                      (Cin=1: k-block = kernel row, j = kernel column; Cin=3: k = tap*3 + c)
//   wb[5][64][8] bf16: conv1, step s covers taps stem16_pair_tap(s, 0) (k-blocks 0,1 = channels 0-7, 8-15) and
//                      stem16_pair_tap(s, 1) (k-blocks 2,3); -1 = zero weights
//   bias0[16], bias1[16] f32
int sa_stem16_pack(const float* k0, const float* b0, int Cin, int C0, const float* k1, const float* b1, int C1,
                   void* blob) {
  SA_REQUIRE((Cin == 1 || Cin == 3) && C0 >= 1 && C0 <= 16 && C1 >= 1 && C1 <= 16, "sa_stem16_pack: bad channel counts");
  uint16_t* w = (uint16_t*)blob;
  for (int l = 0; l < 64; ++l) {
    const int m = l & 15, kb = l >> 4;
    for (int j = 0; j < 8; ++j) {
      const int k = kb * 8 + j;
      int f = -1;
      if (Cin == 1) {
        if (kb < 3 && j < 3) f = kb * 3 + j;
      } else if (k < 9 * Cin) {
        f = k;
      }
      float wv = 0.0f;
      if (f >= 0 && m < C0) wv = k0[(size_t)f * C0 + m] * (1.0f / 255.0f) * (1.0f / sa::U8_ACT_SCALE);
      const uint16_t h0 = sa::f2h(wv);
      const float r1 = wv - sa::h2f(h0);
      const uint16_t h1 = sa::f2h(r1);
      w[(0 * 64 + l) * 8 + j] = h0;
      w[(1 * 64 + l) * 8 + j] = h1;
      w[(2 * 64 + l) * 8 + j] = sa::f2h(r1 - sa::h2f(h1));
    }
    for (int s = 0; s < 5; ++s) {
      const int tap = stem16_pair_tap(s, kb >> 1);
      for (int j = 0; j < 8; ++j) {
        const int ci = (kb & 1) * 8 + j;
        float v = 0.0f;
        if (tap >= 0 && ci < C0 && m < C1) v = k1[((size_t)tap * C0 + ci) * C1 + m];
        w[((3 + s) * 64 + l) * 8 + j] = sa::f2h(v);
      }
    }
  }
  float* bb = reinterpret_cast<float*>(w + 8 * 64 * 8);
  for (int i = 0; i < 16; ++i) {
    bb[i] = i < C0 ? b0[i] : 0.0f;
    bb[16 + i] = i < C1 ? b1[i] : 0.0f;
  }
  return SA_OK;
}

int sa_stem16_u8_bf16(const void* src, int B, int H, int W, int Cin, const void* blob, int relu0, int relu1,
                      void* dst, void* dst_pool, sa_stream_t stream) {
  SA_REQUIRE(src && blob && (dst || dst_pool), "sa_stem16_u8_bf16: NULL pointer");
  SA_REQUIRE(Cin == 1 || Cin == 3, "sa_stem16_u8_bf16: Cin must be 1 or 3");
  SA_REQUIRE(B > 0 && H > 0 && W > 0, "sa_stem16_u8_bf16: bad shape");
  SA_REQUIRE(!dst_pool || (H % 2 == 0 && W % 2 == 0), "sa_stem16_u8_bf16: pooled output needs even H, W");
  Stem16Params p;
  p.src = (const uint8_t*)src;
  p.blob = (const uint16_t*)blob;
  p.dst = (uint16_t*)dst;
  p.dst_pool = (uint16_t*)dst_pool;
  p.B = B;
  p.H = H;
  p.W = W;
  p.relu0 = relu0;
  p.relu1 = relu1;
  p.tiles_x = (W + SA_STEM16_TW - 1) / SA_STEM16_TW;
  p.tiles_y = (H + SA_STEM16_TH - 1) / SA_STEM16_TH;
  const size_t nblk = (size_t)p.tiles_x * p.tiles_y * B;
  if (nblk > 0x7fffffffull) return sa::fail(SA_ERR_INVALID_ARG, "sa_stem16_u8_bf16: grid too large");
  if (Cin == 1 && (W & 3) == 0 && ((uintptr_t)src & 3) == 0 && p.tiles_y <= 65535 && B <= 65535)
    hipLaunchKernelGGL(stem16_gray_kernel, dim3((unsigned)p.tiles_x, (unsigned)p.tiles_y, (unsigned)B), dim3(256), 0, (hipStream_t)stream, p);
  else if (Cin == 1)
    hipLaunchKernelGGL((stem16_kernel<1>), dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL((stem16_kernel<3>), dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, p);
  SA_LAUNCH_CHECK();
  return SA_OK;
}

}  // extern "C"
