// One encoder block of SLEAP's UNet in ONE kernel: Conv2D(k3, 16 -> 32)+bias+ReLU -> Conv2D(k3, 32 -> 32)+bias+ReLU
// [-> full store] [-> MaxPool2D(2) store] (encoder_decoder.py:109-131, block 1 of baseline_medium_rf: filters 16, rate 2).
//
// Why: on their own these are the two worst-utilised layers of the network (matrix cores 19 % and 36 % busy,
// profiles/r01_v5_pmc_mfma_util.md): with Cin = 16 / 32 a tile issues 18 / 36 MFMAs per wave between an address set-up
// and an epilogue, and the 32-channel intermediate (1.07 GB per 64 frames) goes to HBM and back. Here the intermediate
// lives only in LDS as bf16 -- the same rounding the stored tensor would get, and the same MFMA accumulation order, so the
// result is bitwise what sa_conv3x3_bf16 twice produces -- and one tile carries 58 MFMAs per wave for one prologue and one
// epilogue.
//
// Workgroup = 8 waves, 16 x 32 output pixels. LDS (81152 B -> two workgroups per CU):
//   [0, 39168)        intermediate halo tile 18 x 34 px x 32 ch, 64 B per pixel, 16-byte slots XOR (pixel >> 2) & 3
//   [39168, 62720)    input halo tile 20 x 36 px x 16 ch, 32 B per pixel, slots XOR (pixel >> 3) & 1      (phase A only)
//   [62720, 81152)    conv-b weights, 18 slabs of 1 KiB (MFMA A fragments, sa_pack_conv3x3_weights)
// conv-a's nine A fragments are register resident (global loads at kernel start), which is what makes room for conv-b's
// weights beside the input tile: everything is requested up front, phase B starts without a third barrier or a mid-kernel
// wait for L2 (0.635 -> 0.58 ms per 64 frames of 512 x 512 against the version that re-used the input tile's space).
// Phase A: conv-a on all 612 halo pixels (20 groups of 32 over the 8 waves; +20 % conv-a FLOPs for the halo), written to
// the intermediate tile with zeros outside the image (= conv-b's SAME padding). Phase B: the usual 9-tap loop.
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "bf16.h"
#include "sa_common.h"

#if !defined(SA_PAIR_SWZ2)
#define SA_PAIR_SWZ2 1  // persistent pair kernel: conflict-free intermediate-tile stores (0: the round 1-5 swizzle, A/B)
#endif
namespace {

using sa::h16x8_t;
using sa::mfma_h8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct PairParams {
  const uint16_t* src;  // [B,H,W,16]
  const uint16_t* wa;   // packed (C0P 16 -> CoutP 32): [1][1][9][64][8]
  const uint16_t* wb;   // packed (C0P 32 -> CoutP 32): [1][2][9][64][8]
  const float* bias_a;  // [32]
  const float* bias_b;  // [32]
  uint16_t* dst;        // [B,H,W,32] or NULL
  uint16_t* dst_pool;   // [B,H/2,W/2,32] or NULL
  int B, H, W, relu_a, relu_b, tiles_x, tiles_y;
  int planar;  // dst / dst_pool as two 16-channel planes [B,2,H,W,16] (SA_LAYOUT_PLANES16) instead of [B,H,W,32]
};

// Lane mapping (round 2). The first version walked the 612 halo pixels of phase A as 20 groups of 32 CONSECUTIVE pixels; the
// counters showed it VALU-issue bound (60 % busy against 38 % for the matrix cores, profiles/r01_v8_pmc_sq_counters.md) and the
// ISA showed why: 146 VALU instructions per 9 MFMAs in phase A -- a lane-dependent row / column walk with a division per
// group, nine swizzled tap addresses, a per-value select for the padding -- and 100 per 36 in phase B for the swizzled
// fragment addresses. Now (53 and 36; 0.537 -> 0.497 ms per 64 frames of 512 x 512, same bits; profiles/r02_ab_session.md)
//   * phase A: wave w walks halo rows w, w + 8, (w + 16) over the 32 columns lane & 31: every per-lane quantity (the nine
//     tap offsets, the two store offsets, the column validity) is loop invariant up to a compile-time byte stride (the
//     swizzles are periodic in 8 rows), the row validity is wave uniform, and the SAME padding is one AND per packed dword;
//     halo columns 32, 33 (36 pixels) are two more groups, taken by waves 2 and 3, which have only two rows;
//   * phase B: the 24 fragment offsets (4 rows x 3 columns x 2 k-steps) are registers computed once.
__global__ void __launch_bounds__(512)
convpair_16_32_32_kernel(const PairParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NW = 8, R = 2, TH = NW * R, TW = 32;
  constexpr int PH = TH + 2, PW = TW + 2;
  constexpr int QH = TH + 4, QW = TW + 4;
  constexpr int INTER_BYTES = PH * PW * 64;
  constexpr int IN_BYTES = QH * QW * 32;
  constexpr int N_IN = (IN_BYTES + 1023) / 1024;
  constexpr int WA_OFF = INTER_BYTES + N_IN * 1024;
  constexpr unsigned OOB = 0xFFFFFF00u;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* inter = smem;
  unsigned char* in_tile = smem + INTER_BYTES;
  unsigned char* wb_tile = smem + WA_OFF;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, lx = lane & 31;
  int bid;
  {
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int tx_i = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int ty_i = bid % p.tiles_y;
  const int b = bid / p.tiles_y;
  const int x0 = tx_i * TW, y0 = ty_i * TH;
  const int H = p.H, W = p.W;

  const size_t fbytes = (size_t)H * W * 32;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const unsigned char*>(p.src) + b * fbytes), 0, (int)fbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwb = __builtin_amdgcn_make_buffer_rsrc((void*)p.wb, 0, 18 * 1024, 0x00020000);

#pragma unroll
  for (int j = 0; j < (N_IN + NW - 1) / NW; ++j) {
    const int i = j * NW + wave;
    if (i < N_IN) {
      const int o = i * 1024 + lane * 16;
      const int pl = o >> 5, s = (o >> 4) & 1;
      const int ty = pl / QW, tx = pl - ty * QW;
      const int gy = y0 + ty - 2, gx = x0 + tx - 2;
      const bool ok = pl < QH * QW && gy >= 0 && gy < H && gx >= 0 && gx < W;
      const unsigned voff = ok ? (unsigned)(gy * W + gx) * 32u + (unsigned)((s ^ ((pl >> 3) & 1)) * 16) : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(in_tile + i * 1024), 16, voff, 0, 0, 0);
    }
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int k = j * NW + wave;
    if (k < 18) __builtin_amdgcn_raw_ptr_buffer_load_lds(rwb, (lds_ptr_t)(wb_tile + k * 1024), 16, (unsigned)lane * 16, k * 1024, 0, 0);
  }
  mfma_h8 wa_reg[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
    wa_reg[tap] = *reinterpret_cast<const mfma_h8*>(p.wa + ((size_t)tap * 64 + lane) * 8);
  const float4 ba0 = *reinterpret_cast<const float4*>(p.bias_a + 4 * half);
  const float4 ba1 = *reinterpret_cast<const float4*>(p.bias_a + 8 + 4 * half);
  const float4 ba2 = *reinterpret_cast<const float4*>(p.bias_a + 16 + 4 * half);
  const float4 ba3 = *reinterpret_cast<const float4*>(p.bias_a + 24 + 4 * half);
  const float bias_a[16] = {ba0.x, ba0.y, ba0.z, ba0.w, ba1.x, ba1.y, ba1.z, ba1.w,
                            ba2.x, ba2.y, ba2.z, ba2.w, ba3.x, ba3.y, ba3.z, ba3.w};

  // ---- phase A set-up (overlaps the copies): offsets of halo row `wave`, column lx; row + 8 adds a compile-time stride
  // (input tile: 8 * 36 pixels * 32 B = 9216, and (pin >> 3) & 1 is unchanged by + 288; intermediate tile: 8 * 34 * 64 B =
  // 17408, and (pl >> 2) & 3 is unchanged by + 272)
  unsigned roff[9], woff[2];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int pin = (wave + tap / 3) * QW + lx + tap % 3;
    roff[tap] = (unsigned)(pin * 32 + ((half ^ ((pin >> 3) & 1)) * 16));
  }
  {
    const int pl = wave * PW + lx;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) woff[pr] = (unsigned)(pl * 64 + (((2 * pr + half) ^ ((pl >> 2) & 3)) * 16));
  }
  const bool colok = (unsigned)(x0 + lx - 1) < (unsigned)W;
  const float low_a = p.relu_a ? 0.0f : -INFINITY;

  // one group: 9 MFMAs on the fragments at rd[tap] (+ rimm), bias + ReLU, zero outside the image, two 16-byte stores
  auto group = [&](const unsigned (&rd)[9], int rimm, const unsigned (&wr)[2], int wimm, bool in_img, bool store) {
    f32x16 d;
#pragma unroll
    for (int i = 0; i < 16; ++i) d[i] = 0.0f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const mfma_h8 bv = *reinterpret_cast<const mfma_h8*>(in_tile + rd[tap] + rimm);
      d = SA_MFMA_32x32x16(wa_reg[tap], bv, d, 0, 0, 0);
    }
    const unsigned m = in_img ? 0xFFFFFFFFu : 0u;
    uint2 pk[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      pk[q].x = sa::f2h2(fmaxf(d[4 * q + 0] + bias_a[4 * q + 0], low_a), fmaxf(d[4 * q + 1] + bias_a[4 * q + 1], low_a)) & m;
      pk[q].y = sa::f2h2(fmaxf(d[4 * q + 2] + bias_a[4 * q + 2], low_a), fmaxf(d[4 * q + 3] + bias_a[4 * q + 3], low_a)) & m;
    }
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      uint2 x = pk[2 * pr], y = pk[2 * pr + 1];
      sa::swap32(x.x, y.x);
      sa::swap32(x.y, y.y);
      if (store) *reinterpret_cast<uint4*>(inter + wr[pr] + wimm) = make_uint4(x.x, x.y, y.x, y.y);
    }
  };

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- phase A: conv-a (16 -> 32) on the 18 x 34 halo pixels
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    const int ty = wave + 8 * it;
    if (ty < PH) {  // wave uniform (it == 2: waves 0, 1)
      const bool rowok = (unsigned)(y0 + ty - 1) < (unsigned)H;
      group(roff, it * (8 * QW * 32), woff, it * (8 * PW * 64), colok && rowok, true);
    }
  }
  if (wave == 2 || wave == 3) {  // halo columns 32, 33: pixel q of the 36 -> row q >> 1, column 32 + (q & 1)
    const int q = (wave - 2) * 32 + lx;
    const bool valid = q < 2 * PH;
    const int qc = valid ? q : 0;
    const int ty = qc >> 1, tx = 32 + (qc & 1);
    unsigned rd[9], wr[2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int pin = (ty + tap / 3) * QW + tx + tap % 3;
      rd[tap] = (unsigned)(pin * 32 + ((half ^ ((pin >> 3) & 1)) * 16));
    }
    const int pl = ty * PW + tx;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) wr[pr] = (unsigned)(pl * 64 + (((2 * pr + half) ^ ((pl >> 2) & 3)) * 16));
    const bool in_img = (unsigned)(y0 + ty - 1) < (unsigned)H && (unsigned)(x0 + tx - 1) < (unsigned)W;
    group(rd, 0, wr, 0, in_img, valid);
  }

  // ---- phase B set-up: fragment offsets of halo rows wave*2 .. wave*2+3, columns lx .. lx+2, both k-steps
  unsigned boff[R + 2][3][2];
#pragma unroll
  for (int rr = 0; rr < R + 2; ++rr)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int pl = (wave * R + rr) * PW + lx + dx;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        boff[rr][dx][kk] = (unsigned)(pl * 64 + (((kk * 2 + half) ^ ((pl >> 2) & 3)) * 16));
        asm volatile("" : "+v"(boff[rr][dx][kk]));
      }
    }
  float bb[4][4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 q = *reinterpret_cast<const float4*>(p.bias_b + 8 * g + 4 * half);
    bb[g][0] = q.x;
    bb[g][1] = q.y;
    bb[g][2] = q.z;
    bb[g][3] = q.w;
  }
  __syncthreads();  // the intermediate tile is complete

  // ---- phase B: conv-b (32 -> 32), wave owns rows wave*2, wave*2+1
  f32x16 acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[r][i] = 0.0f;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int dy = tap / 3, dx = tap % 3;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const mfma_h8 a = *reinterpret_cast<const mfma_h8*>(wb_tile + (kk * 9 + tap) * 1024 + lane * 16);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const mfma_h8 bv = *reinterpret_cast<const mfma_h8*>(inter + boff[r + dy][dx][kk]);
        acc[r] = SA_MFMA_32x32x16(a, bv, acc[r], 0, 0, 0);
      }
    }
  }

  // ---- epilogue (as conv3x3_dma_kernel)
  const float low_b = p.relu_b ? 0.0f : -INFINITY;
  const int gx = x0 + lx;
  auto act = [&](int r, int g, int j) { return fmaxf(acc[r][4 * g + j] + bb[g][j], low_b); };
  // store addressing as in conv3x3_dma_kernel: frame, 16-channel block and tile row in a wave-uniform base, ONE 32-bit byte offset
  // per lane and output (column, 8-channel half); piece pr = channels 16 pr + 8 half ..
  auto store_pieces = [&](unsigned char* row_base, size_t blk_bytes, unsigned lane_off, bool ok, const uint2 (&pk)[4]) {
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      uint2 x = pk[2 * pr], y = pk[2 * pr + 1];
      sa::swap32(x.x, y.x);
      sa::swap32(x.y, y.y);
      if (ok) *reinterpret_cast<uint4*>(row_base + (size_t)pr * blk_bytes + lane_off) = make_uint4(x.x, x.y, y.x, y.y);
    }
  };
  const unsigned pixb = p.planar ? 32u : 64u;  // bytes per pixel record: one 16-channel plane, or all 32 channels
  const bool col_in = gx < W;
  if (p.dst) {
    unsigned char* frame = reinterpret_cast<unsigned char*>(p.dst) + (size_t)b * H * W * 64;
    const size_t blk = p.planar ? (size_t)H * W * 32 : (size_t)32;
    const unsigned lane_off = (unsigned)gx * pixb + (unsigned)half * 16u;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int gy = y0 + wave * R + r;  // wave uniform
      uint2 pk[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        pk[g].x = sa::f2h2(act(r, g, 0), act(r, g, 1));
        pk[g].y = sa::f2h2(act(r, g, 2), act(r, g, 3));
      }
      store_pieces(frame + (size_t)gy * W * pixb, blk, lane_off, col_in && gy < H, pk);
    }
  }
  if (p.dst_pool) {
    const int gy = y0 + wave * R;  // wave uniform, even
    unsigned char* frame = reinterpret_cast<unsigned char*>(p.dst_pool) + (size_t)b * (H / 2) * (W / 2) * 64;
    const size_t blk = p.planar ? (size_t)(H / 2) * (W / 2) * 32 : (size_t)32;
    const unsigned lane_off = (unsigned)(gx >> 1) * pixb + (unsigned)half * 16u;
    uint2 pk[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float t4[4], t[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) t[j] = fmaxf(act(0, g, j), act(1, g, j));
      sa::max_xor1_x4(t, t4);
      pk[g].x = sa::f2h2(t4[0], t4[1]);
      pk[g].y = sa::f2h2(t4[2], t4[3]);
    }
    store_pieces(frame + (size_t)(gy >> 1) * (W / 2) * pixb, blk, lane_off, !(lane & 1) && col_in && gy < H, pk);
  }
#endif
}


// ---- round 3: the same block as a PERSISTENT workgroup with the next tile's input prefetched ------------------------------------
// What the one-tile-per-workgroup kernel above spends per tile besides its 58 MFMAs per wave (profiles/r02_pmc_sq_counters.md:
// matrix cores 48 % busy, VALU issue 62 %): a launch, ~150 VALU instructions of address set-up (DMA offsets, nine tap offsets,
// 24 fragment offsets), the reload of conv-a's nine A fragments and of conv-b's 18 KiB weight slab, and -- the largest part --
// an HBM round trip for the input tile with nothing to do meanwhile but the partner workgroup's work. Here a workgroup keeps
// everything tile-independent (both weight sets, every LDS offset) for its whole life and walks the tiles of its XCD's range;
// the input tile of tile t+1 is requested right after phase A of tile t has finished reading the input area (which phase B does
// not touch) and lands under phase B's 36 MFMAs per wave. The one wait for those copies sits BEFORE the epilogue's stores, so it
// never waits for a store (gfx9 counts loads and stores in one vmcnt; the stores of tile t then drain under phase A of t+1).
// Arithmetic and rounding are those of the kernel above (same bits). The biases live in LDS (256 B) instead of 32 registers.
__global__ void __launch_bounds__(512, 4)
convpair_persist_kernel(const PairParams p, int n_tiles) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NW = 8, R = 2, TH = NW * R, TW = 32;
  constexpr int PH = TH + 2, PW = TW + 2;
  constexpr int QH = TH + 4, QW = TW + 4;
  constexpr int INTER_BYTES = PH * PW * 64;
  constexpr int IN_BYTES = QH * QW * 32;
  constexpr int N_IN = (IN_BYTES + 1023) / 1024;
  constexpr int WA_OFF = INTER_BYTES + N_IN * 1024;
  constexpr int BIAS_OFF = WA_OFF + 18 * 1024;
  constexpr int IN_PER_WAVE = (N_IN + NW - 1) / NW;
  constexpr unsigned OOB = 0xFFFFFF00u;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* inter = smem;
  unsigned char* in_tile = smem + INTER_BYTES;
  unsigned char* wb_tile = smem + WA_OFF;
  const float* bias_lds = reinterpret_cast<const float*>(smem + BIAS_OFF);  // [0, 32): conv-a, [32, 64): conv-b

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, lx = lane & 31;
  const int H = p.H, W = p.W;
  // tile schedule: contiguous range per XCD (block i runs on XCD i % 8), the j-th workgroup of an XCD walks start + j, + g8, ...
  int L, L_end, L_step;
  {
    const int q = n_tiles >> 3, r = n_tiles & 7, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    L_step = ((int)gridDim.x - xcd + 7) >> 3;
    L = start + k;
    L_end = start + q + (xcd < r ? 1 : 0);
  }
  if (L >= L_end) return;  // workgroup uniform
  const size_t fbytes = (size_t)H * W * 32;
  const __amdgpu_buffer_rsrc_t rwb = __builtin_amdgcn_make_buffer_rsrc((void*)p.wb, 0, 18 * 1024, 0x00020000);

  // ---- once per workgroup: conv-b weights -> LDS, conv-a fragments -> registers, biases -> LDS, every tile-independent offset
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int k = j * NW + wave;
    if (k < 18) __builtin_amdgcn_raw_ptr_buffer_load_lds(rwb, (lds_ptr_t)(wb_tile + k * 1024), 16, (unsigned)lane * 16, k * 1024, 0, 0);
  }
  if (tid < 64) reinterpret_cast<float*>(smem + BIAS_OFF)[tid] = tid < 32 ? p.bias_a[tid] : p.bias_b[tid - 32];
  mfma_h8 wa_reg[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
    wa_reg[tap] = *reinterpret_cast<const mfma_h8*>(p.wa + ((size_t)tap * 64 + lane) * 8);
  // LDS layouts: as in the kernel above, but the 16-byte-slot swizzles depend on the tile COLUMN only ((column >> 3) & 1 for
  // the 32-byte input records, (column >> 2) & 3 for the 64-byte intermediate records): the lanes of a ds_read_b128 /
  // ds_write_b128 group always share their row, so this is as conflict-free as the pixel-index form, and a fragment offset
  // becomes (per-lane column part) + (compile-time row part) -- 3 + 6 + 2 offset registers instead of 9 + 24 + 2, which is what
  // lets everything tile-independent stay resident next to the 36 registers of conv-a's A fragments without spilling.
  // Round 6: the intermediate tile's slot swizzle is ((column >> 2) + 2 ((column >> 1) & 1)) & 3 instead of (column >> 2) & 3. The
  // reads do not care (a ds_read_b128 group's columns are 4 k apart: the added term is constant inside a group, the sum still runs
  // through 0, 3, 1, 2); the WRITES of phase A do: a ds_write_b128 group is eight consecutive columns of one k-slot, 64 bytes apart
  // = two 16-byte bank quads per column pair -- with the old swizzle columns c and c + 2 shared their quad (2-way on every
  // store: 14 % of this kernel's LDS-active cycles were conflicts, profiles/r06_pmc_sq_counters.md), now the eight are distinct.
  auto iswz = [](int c) { return SA_PAIR_SWZ2 ? (((c >> 2) + ((c >> 1) & 1) * 2) & 3) : ((c >> 2) & 3); };
  // phase B fragment offsets of halo row wave*2, columns lx + dx, both k-steps (row rr adds rr * PW * 64)
  unsigned boff[3][2];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      boff[dx][kk] = (unsigned)(((wave * R) * PW + lx + dx) * 64 + (((2 * kk + half) ^ iswz(lx + dx)) * 16));
      asm volatile("" : "+v"(boff[dx][kk]));
    }
  const float low_a = p.relu_a ? 0.0f : -INFINITY, low_b = p.relu_b ? 0.0f : -INFINITY;
  const unsigned pixb = p.planar ? 32u : 64u;

  struct Tile {
    int x0, y0, b;
  };
  auto decode = [&](int l) {
    Tile t;
    t.x0 = (l % p.tiles_x) * TW;
    l /= p.tiles_x;
    t.y0 = (l % p.tiles_y) * TH;
    t.b = l / p.tiles_y;
    return t;
  };
  auto issue_input = [&](const Tile& t) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const unsigned char*>(p.src) + t.b * fbytes), 0, (int)fbytes, 0x00020000);
    int ln = lane;  // (per-lane piece coordinates are re-derived per tile: nothing of them stays in registers across the phases)
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int j = 0; j < IN_PER_WAVE; ++j) {
      const int i = j * NW + wave;
      if (i < N_IN) {  // wave uniform
        const int o = i * 1024 + ln * 16;
        const int pl = o >> 5, s = (o >> 4) & 1;
        const int ty = pl / QW, tx = pl - ty * QW;
        const int gy = t.y0 + ty - 2, gx = t.x0 + tx - 2;
        const bool ok = pl < QH * QW && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        const unsigned voff = ok ? (unsigned)(gy * W + gx) * 32u + (unsigned)((s ^ ((tx >> 3) & 1)) * 16) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(in_tile + i * 1024), 16, voff, 0, 0, 0);
      }
    }
  };

  // one phase-A group: 9 MFMAs on the fragments at rd[tap] (+ rimm), bias + ReLU, zero outside the image, two 16-byte stores
  auto group = [&](const unsigned (&rd)[3], int rimm, const unsigned (&wr)[2], int wimm, auto masked_c, bool in_img, bool store,
                   int hoff4) {
    constexpr bool MASKED = decltype(masked_c)::value;
    f32x16 d;
#pragma unroll
    for (int i = 0; i < 16; ++i) d[i] = 0.0f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const mfma_h8 bv = *reinterpret_cast<const mfma_h8*>(in_tile + rd[tap % 3] + (tap / 3) * (QW * 32) + rimm);
      d = SA_MFMA_32x32x16(wa_reg[tap], bv, d, 0, 0, 0);
    }
    const unsigned m = (!MASKED || in_img) ? 0xFFFFFFFFu : 0u;
    uint2 pk[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 bq = *reinterpret_cast<const float4*>(bias_lds + 8 * q + hoff4);  // (one quad at a time: 4 registers, not 16)
      const float bj[4] = {bq.x, bq.y, bq.z, bq.w};
      pk[q].x = sa::f2h2(fmaxf(d[4 * q + 0] + bj[0], low_a), fmaxf(d[4 * q + 1] + bj[1], low_a));
      pk[q].y = sa::f2h2(fmaxf(d[4 * q + 2] + bj[2], low_a), fmaxf(d[4 * q + 3] + bj[3], low_a));
      if (MASKED) pk[q].x &= m, pk[q].y &= m;
    }
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      uint2 x = pk[2 * pr], y = pk[2 * pr + 1];
      sa::swap32(x.x, y.x);
      sa::swap32(x.y, y.y);
      if (store) *reinterpret_cast<uint4*>(inter + wr[pr] + wimm) = make_uint4(x.x, x.y, y.x, y.y);
    }
  };
  auto phase_a = [&](const Tile& t, auto masked_c) {
    // per-lane offsets of halo row `wave`, column lx: derived per tile from an opaque copy of the lane id, so that they are not
    // kept in 11 registers across phase B (with them the kernel needs > 128 registers and spills)
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int half = ln >> 5, lx = ln & 31;
    unsigned roff[3], woff[2];  // halo row `wave`, columns lx + dx; a tap's row dy adds dy * QW * 32
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) roff[dx] = (unsigned)((wave * QW + lx + dx) * 32 + ((half ^ (((lx + dx) >> 3) & 1)) * 16));
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) woff[pr] = (unsigned)((wave * PW + lx) * 64 + (((2 * pr + half) ^ iswz(lx)) * 16));
    const bool colok = (unsigned)(t.x0 + lx - 1) < (unsigned)W;
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int ty = wave + 8 * it;
      if (ty < PH) {  // wave uniform (it == 2: waves 0, 1)
        const bool rowok = (unsigned)(t.y0 + ty - 1) < (unsigned)H;
        group(roff, it * (8 * QW * 32), woff, it * (8 * PW * 64), masked_c, colok && rowok, true, 4 * half);
      }
    }
    if (wave == 2 || wave == 3) {  // halo columns 32, 33: pixel q of the 36 -> row q >> 1, column 32 + (q & 1)
      const int q2 = (wave - 2) * 32 + lx;
      const bool valid2 = q2 < 2 * PH;
      const int qc2 = valid2 ? q2 : 0, ty2 = qc2 >> 1, tx2 = 32 + (qc2 & 1);
      unsigned rd[3], wr[2];
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) rd[dx] = (unsigned)((ty2 * QW + tx2 + dx) * 32 + ((half ^ (((tx2 + dx) >> 3) & 1)) * 16));
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) wr[pr] = (unsigned)((ty2 * PW + tx2) * 64 + (((2 * pr + half) ^ iswz(tx2)) * 16));
      const bool in_img = (unsigned)(t.y0 + ty2 - 1) < (unsigned)H && (unsigned)(t.x0 + tx2 - 1) < (unsigned)W;
      group(rd, 0, wr, 0, masked_c, in_img, valid2, 4 * half);
    }
  };

  Tile cur = decode(L);
  issue_input(cur);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // first tile: weights + input (later tiles wait before their epilogue's stores)
#pragma clang loop unroll(disable)
  for (;;) {
    const int L_next = L + L_step;
    const bool more = L_next < L_end;  // workgroup uniform
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();  // every wave's copies of this tile's input landed (each waited for its own); phase B of the previous tile is done

    // ---- phase A: conv-a (16 -> 32) on the 18 x 34 halo pixels; tiles whose halo lies inside the image skip the padding mask
    const bool interior = cur.x0 >= 1 && cur.y0 >= 1 && cur.x0 + PW - 1 <= W && cur.y0 + PH - 1 <= H;
    if (__builtin_amdgcn_readfirstlane((int)interior))
      phase_a(cur, std::false_type{});
    else
      phase_a(cur, std::true_type{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();  // the intermediate tile is complete; nobody reads the input tile any more
    Tile nxt = cur;
    if (more) {
      nxt = decode(L_next);
      issue_input(nxt);  // lands under phase B
    }

    // ---- phase B: conv-b (32 -> 32), wave owns rows wave*2, wave*2+1
    f32x16 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[r][i] = 0.0f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap % 3;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const mfma_h8 a = *reinterpret_cast<const mfma_h8*>(wb_tile + (kk * 9 + tap) * 1024 + lane * 16);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const mfma_h8 bv = *reinterpret_cast<const mfma_h8*>(inter + boff[dx][kk] + (r + dy) * (PW * 64));
          acc[r] = SA_MFMA_32x32x16(a, bv, acc[r], 0, 0, 0);
        }
      }
    }

    // ---- epilogue. The prefetched copies were queued a whole phase B ago: waiting for them HERE, before the first store is
    // issued, costs next to nothing and keeps the stores out of every later wait.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int ln_e = lane;
    asm volatile("" : "+v"(ln_e));
    const int half_e = ln_e >> 5;
    const unsigned lane16 = (unsigned)half_e * 16u;
    const int gx = cur.x0 + (ln_e & 31);
    auto bias4 = [&](int g, float (&bj)[4]) {  // conv-b bias of this lane's channels 8g + 4 half .. +3
      const float4 q = *reinterpret_cast<const float4*>(bias_lds + 32 + 8 * g + 4 * half_e);
      bj[0] = q.x, bj[1] = q.y, bj[2] = q.z, bj[3] = q.w;
    };
    auto store_pieces = [&](unsigned char* row_base, size_t blk_bytes, unsigned lane_off, bool ok, const uint2 (&pk)[4]) {
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        uint2 x = pk[2 * pr], y = pk[2 * pr + 1];
        sa::swap32(x.x, y.x);
        sa::swap32(x.y, y.y);
        if (ok) *reinterpret_cast<uint4*>(row_base + (size_t)pr * blk_bytes + lane_off) = make_uint4(x.x, x.y, y.x, y.y);
      }
    };
    const bool col_in = gx < W;
    if (p.dst) {
      unsigned char* frame = reinterpret_cast<unsigned char*>(p.dst) + (size_t)cur.b * H * W * 64;
      const size_t blk = p.planar ? (size_t)H * W * 32 : (size_t)32;
      const unsigned lane_off = (unsigned)gx * pixb + lane16;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int gy = cur.y0 + wave * R + r;  // wave uniform
        uint2 pk[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float bj[4];
          bias4(g, bj);
          pk[g].x = sa::f2h2(fmaxf(acc[r][4 * g + 0] + bj[0], low_b), fmaxf(acc[r][4 * g + 1] + bj[1], low_b));
          pk[g].y = sa::f2h2(fmaxf(acc[r][4 * g + 2] + bj[2], low_b), fmaxf(acc[r][4 * g + 3] + bj[3], low_b));
        }
        store_pieces(frame + (size_t)gy * W * pixb, blk, lane_off, col_in && gy < H, pk);
      }
    }
    if (p.dst_pool) {
      const int gy = cur.y0 + wave * R;  // wave uniform, even
      unsigned char* frame = reinterpret_cast<unsigned char*>(p.dst_pool) + (size_t)cur.b * (H / 2) * (W / 2) * 64;
      const size_t blk = p.planar ? (size_t)(H / 2) * (W / 2) * 32 : (size_t)32;
      const unsigned lane_off = (unsigned)(gx >> 1) * pixb + lane16;
      uint2 pk[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float t4[4], t[4], bj[4];
        bias4(g, bj);
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = fmaxf(fmaxf(acc[0][4 * g + j] + bj[j], low_b), fmaxf(acc[1][4 * g + j] + bj[j], low_b));
        sa::max_xor1_x4(t, t4);
        pk[g].x = sa::f2h2(t4[0], t4[1]);
        pk[g].y = sa::f2h2(t4[2], t4[3]);
      }
      store_pieces(frame + (size_t)(gy >> 1) * (W / 2) * pixb, blk, lane_off, !(ln_e & 1) && col_in && gy < H, pk);
    }
    if (!more) break;
    cur = nxt;
    L = L_next;
  }
#endif
}

}  // namespace

// convpair64.hip: the 32 -> 64 -> 64 block (one persistent workgroup per CU)
int sa_convpair64_launch(const void* src, const void* wa, const float* bias_a, int relu_a, const void* wb, const float* bias_b,
                         int relu_b, int B, int H, int W, void* dst, void* dst_pool, int layout, hipStream_t stream);

extern "C" int sa_conv3x3_pair_bf16(const void* src, int C0P, const void* wa, const float* bias_a, int relu_a, int C1P,
                                    const void* wb, const float* bias_b, int relu_b, int C2P, int B, int H, int W, void* dst,
                                    void* dst_pool, int layout, sa_stream_t stream) {
  SA_REQUIRE(src && wa && wb && bias_a && bias_b && (dst || dst_pool), "sa_conv3x3_pair_bf16: NULL pointer");
  const bool is64 = C0P == 32 && C1P == 64 && C2P == 64;
  if (!is64 && (C0P != 16 || C1P != 32 || C2P != 32))
    return sa::fail(SA_ERR_UNSUPPORTED, "sa_conv3x3_pair_bf16: only the 16 -> 32 -> 32 and 32 -> 64 -> 64 blocks are implemented (got %d -> %d -> %d)",
                    C0P, C1P, C2P);
  SA_REQUIRE(B > 0 && H > 0 && W > 0, "sa_conv3x3_pair_bf16: bad shape");
  SA_REQUIRE(layout == SA_LAYOUT_NHWC || layout == SA_LAYOUT_PLANES16, "sa_conv3x3_pair_bf16: bad layout");
  SA_REQUIRE(!dst_pool || (H % 2 == 0 && W % 2 == 0), "sa_conv3x3_pair_bf16: pooled output needs even H, W");
  if (is64) return sa_convpair64_launch(src, wa, bias_a, relu_a, wb, bias_b, relu_b, B, H, W, dst, dst_pool, layout, (hipStream_t)stream);
  SA_REQUIRE((size_t)H * W * 32 < 0xFFFFFF00ull, "sa_conv3x3_pair_bf16: one frame must be smaller than 4 GiB");
  PairParams p;
  p.src = (const uint16_t*)src;
  p.wa = (const uint16_t*)wa;
  p.wb = (const uint16_t*)wb;
  p.bias_a = bias_a;
  p.bias_b = bias_b;
  p.dst = (uint16_t*)dst;
  p.dst_pool = (uint16_t*)dst_pool;
  p.B = B;
  p.H = H;
  p.W = W;
  p.relu_a = relu_a;
  p.relu_b = relu_b;
  p.planar = layout == SA_LAYOUT_PLANES16;
  p.tiles_x = (W + 31) / 32;
  p.tiles_y = (H + 15) / 16;
  const size_t nblk = (size_t)p.tiles_x * p.tiles_y * B;
  if (nblk > 0x7fffffffull) return sa::fail(SA_ERR_INVALID_ARG, "sa_conv3x3_pair_bf16: grid too large");
  constexpr int lds = 18 * 34 * 64 + 23 * 1024 + 18 * 1024;  // 81152: intermediate tile + input tile + conv-b weights
  // SA_CONVPAIR_PERSIST=0: one workgroup per tile (the round-2 kernel, kept for A/B runs); n > 0: n workgroups per CU
  static const int persist = [] {
    const char* v = getenv("SA_CONVPAIR_PERSIST");
    return v ? atoi(v) : -1;
  }();
  if (persist != 0) {
    constexpr int ldsp = lds + 256;  // + both bias vectors
    static int n_cu = 0, per_cu = 0;
    if (!n_cu) {
      int dev = 0, nb = 0;
      SA_HIP_CHECK(hipGetDevice(&dev));
      SA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&convpair_persist_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, ldsp));
      SA_HIP_CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
      SA_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&convpair_persist_kernel), 512, ldsp));
      per_cu = nb > 0 ? nb : 1;
    }
    size_t grid = (size_t)(persist > 0 ? persist : per_cu) * (size_t)n_cu;
    if (grid > nblk) grid = nblk;
    hipLaunchKernelGGL(convpair_persist_kernel, dim3((unsigned)grid), dim3(512), ldsp, (hipStream_t)stream, p, (int)nblk);
    SA_LAUNCH_CHECK();
    return SA_OK;
  }
  static bool attr_set = false;
  if (!attr_set) {
    SA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&convpair_16_32_32_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  hipLaunchKernelGGL(convpair_16_32_32_kernel, dim3((unsigned)nblk), dim3(512), lds, (hipStream_t)stream, p);
  SA_LAUNCH_CHECK();
  return SA_OK;
}
