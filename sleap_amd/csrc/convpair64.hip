// Encoder block 2 of SLEAP's UNet in ONE persistent kernel: Conv2D(k3, 32 -> 64)+bias+ReLU -> Conv2D(k3, 64 -> 64)+bias+ReLU
// [-> full store (the decoder's skip)] [-> MaxPool2D(2) store] (encoder_decoder.py:109-131, block 2 of baseline_medium_rf:
// filters 16, rate 2).
//
// Why (round 6): as two launches these layers are HBM-bound -- 32 -> 64 @256^2 moves its bytes at 5.1 TB/s, 64 -> 64 @256^2 + pool
// at 4.4 TB/s, matrix cores 0.41 / 0.55 busy (profiles/r05_pmc_dominant_per_layer.md) -- because the 64-channel 256^2
// intermediate (537 MB per 64 frames) is written and read back. Here it lives only in LDS, in the storage type (the rounding the
// stored tensor would get) and with the stand-alone kernels' MFMA accumulation order: the result is bitwise what two
// sa_conv3x3_bf16 calls produce.
//
// Shape of the kernel. The intermediate halo tile of a 16 x 32 output tile is 18 x 34 pixels x 64 channels = 76.5 KiB: with the
// 20 x 36 x 32-channel input halo tile (45 KiB) and two 18-KiB weight slots that is 158.5 of the CU's 160 KiB -- ONE workgroup
// per CU, so nothing a second workgroup would cover may be exposed. The workgroup (8 waves, up to 256 registers each) is
// persistent and walks its XCD's share of the tiles through six stages per tile, every stage = "wait for my copies, barrier,
// queue the copies of a LATER stage, multiply":
//   A0  conv-a, input channels  0-15 (input plane 0, slot 0)        queues: conv-a weights k-half 1 -> slot 1
//   A1  conv-a, input channels 16-31 (plane 1, slot 1), epilogue a  queues: conv-b chunk 0 -> slot 0, NEXT tile's input plane 0
//   B0..B3  conv-b, intermediate channels 16c..16c+15 (slot c & 1)  queues: chunk c+1 (B3: the next tile's conv-a k-half 0),
//                                                                           B0 also the next tile's input plane 1
//   epilogue b (stores) -- behind a wait for the copies queued in B3, so that no later wait ever sees a store (gfx9 counts
//   loads and stores in one vmcnt; the stores drain under the next tile's A0).
// LDS map (bytes):
//   [0, 78336)         intermediate tile as FOUR 16-channel planes of 18 x 34 pixels x 32 B (what a stage of conv3x3_dma_kernel
//                      holds: phase B's inner loop is that kernel's), 16-byte slot XOR (column >> 3) & 1
//   [78336, 125440)    input tile as TWO 16-channel planes of 20 x 36 pixels x 32 B (23 one-KiB copy pieces each), same swizzle
//   [125440, 162304)   two weight slots of 18 slabs x 1 KiB (MFMA A fragments of sa_pack_conv3x3_weights: [cout tile][tap])
//   [162304, 162816)   bias_a[64], bias_b[64] f32
// conv-a runs on all 612 halo pixels (+20 % of conv-a's FLOPs = +6.5 % of the block's): nine row pairs of 32 columns and the
// 36 pixels of columns 32, 33 as two more 32-pixel groups = 20 groups x 2 cout tiles = 40 units of 9 MFMAs per stage. Wave w takes
// halo rows 2w, 2w+1 x both cout tiles (4 units, fragments shared as in conv-b) and ONE extra unit: group 16 + (w >> 1) (halo rows
// 16, 17, column groups 0, 1) x cout tile w & 1 -- 45 MFMAs per wave and stage, the same code for every wave (only per-lane
// offsets differ), every SIMD equally loaded wherever its waves land. conv-b: wave w owns output rows 2w, 2w+1 x both cout tiles.
// B fragments are read ONCE per stage into registers (12 per wave: 4 halo rows x 3 columns) and reused by the taps that share
// them: 30 LDS fragment reads per 36 MFMAs instead of 36.
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "bf16.h"
#include "sa_common.h"

namespace {

using sa::mfma_h8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct Pair64Params {
  const uint16_t* src;  // 32 channels: [B,H,W,32] or planes [B,2,H,W,16]
  const uint16_t* wa;   // packed 32 -> 64: [2][2][9][64][8]
  const uint16_t* wb;   // packed 64 -> 64: [2][4][9][64][8]
  const float* bias_a;  // [64]
  const float* bias_b;  // [64]
  uint16_t* dst;        // 64 channels or NULL
  uint16_t* dst_pool;   // [.., H/2, W/2, ..] or NULL
  int B, H, W, relu_a, relu_b, tiles_x, tiles_y, planar, n_tiles;
};

constexpr int NW = 8, R = 2, TH = NW * R, TW = 32;
constexpr int PH = TH + 2, PW = TW + 2;  // intermediate halo tile
constexpr int QH = TH + 4, QW = TW + 4;  // input halo tile
constexpr int IP_ROW = PW * 32, IP_PLANE = PH * IP_ROW;            // 1088, 19584
constexpr int IN_ROW = QW * 32, IN_USED = QH * IN_ROW;             // 1152, 23040
constexpr int N_INP = (IN_USED + 1023) / 1024, IN_PLANE = N_INP * 1024;  // 23, 23552
constexpr int SLOT = 18 * 1024;
constexpr int INTER_OFF = 0, IN_OFF = 4 * IP_PLANE, RING_OFF = IN_OFF + 2 * IN_PLANE, BIAS_OFF = RING_OFF + 2 * SLOT;
constexpr int LDS_BYTES = BIAS_OFF + 128 * 4;
static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU: 160 KiB");
constexpr unsigned OOB = 0xFFFFFF00u;

__global__ void __launch_bounds__(NW * 64, 2)
convpair64_kernel(const Pair64Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, lx = lane & 31;
  const int H = p.H, W = p.W;
  const float* bias_lds = reinterpret_cast<const float*>(smem + BIAS_OFF);

  // ---- tile schedule: a contiguous range of the (frame, tile row, tile column) order per XCD (block i runs on XCD i % 8), the
  // j-th workgroup of an XCD walks start + j, start + j + g8, ...
  int L, L_end, L_step;
  {
    const int q = p.n_tiles >> 3, r = p.n_tiles & 7, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    L_step = ((int)gridDim.x - xcd + 7) >> 3;
    L = start + k;
    L_end = start + q + (xcd < r ? 1 : 0);
  }
  if (L >= L_end) return;  // workgroup uniform
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

  const size_t fbytes = (size_t)H * W * 64;                     // one frame of the source (both planes)
  const unsigned pixb_in = p.planar ? 32u : 64u;                // bytes between neighbouring pixels of the source
  const unsigned plane_in = p.planar ? (unsigned)((size_t)H * W * 32) : 32u;  // bytes between its two 16-channel blocks
  const __amdgpu_buffer_rsrc_t rwa = __builtin_amdgcn_make_buffer_rsrc((void*)p.wa, 0, 36 * 1024, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwb = __builtin_amdgcn_make_buffer_rsrc((void*)p.wb, 0, 72 * 1024, 0x00020000);
  const unsigned wv = (unsigned)lane * 16u;

  // conv-a weights of k-half k / conv-b weights of chunk c -> slot s: slab j = m * 9 + tap of the slot <- slab (m * K16 + k) * 9 + tap
  auto issue_w = [&](const __amdgpu_buffer_rsrc_t& rw, int k16n, int k, int s) {
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
      const int j = jj * NW + wave;  // wave uniform
      if (j < 18) {
        const int m = j >= 9 ? 1 : 0, tap = j - 9 * m;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(smem + RING_OFF + s * SLOT + j * 1024), 16, wv,
                                                 ((m * k16n + k) * 9 + tap) * 1024, 0, 0);
      }
    }
  };
  // per-lane source offsets of this wave's (up to) three copy pieces of an input plane (the same for both planes)
  auto make_voff = [&](const Tile& t, unsigned (&v)[3]) {
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
      const int i = jj * NW + wave;
      const int o = i * 1024 + lane * 16;
      const int pl = o >> 5, s = (o >> 4) & 1;
      const int ty = pl / QW, tx = pl - ty * QW;
      const int gy = t.y0 + ty - 2, gx = t.x0 + tx - 2;
      const bool ok = i < N_INP && pl < QH * QW && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      v[jj] = ok ? (unsigned)(gy * W + gx) * pixb_in + (unsigned)((s ^ ((tx >> 3) & 1)) * 16) : OOB;
    }
  };
  auto issue_in = [&](const Tile& t, const unsigned (&v)[3], int k) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const unsigned char*>(p.src) + t.b * fbytes), 0, (int)fbytes, 0x00020000);
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
      const int i = jj * NW + wave;  // wave uniform
      if (i < N_INP)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + IN_OFF + k * IN_PLANE + i * 1024), 16, v[jj],
                                                 (int)((unsigned)k * plane_in), 0, 2);  // read once: non-temporal
    }
  };

  // ---- tile-independent per-lane LDS offsets. Swizzles depend on the tile COLUMN only, so a row is a compile-time stride.
  unsigned aoff[3], boff[3];  // conv-a reads (input plane, halo row 2 * wave, column lx + dx); conv-b reads (intermediate plane)
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int c = lx + dx;
    aoff[dx] = (unsigned)(2 * wave * IN_ROW + c * 32 + ((half ^ ((c >> 3) & 1)) * 16));
    boff[dx] = (unsigned)(2 * wave * IP_ROW + c * 32 + ((half ^ ((c >> 3) & 1)) * 16));
  }
  // the extra unit of this wave: pixel group xg = 16 + (wave >> 1) -- 16, 17: halo rows 16, 17, columns lx; 18, 19: the 36 pixels of
  // columns 32, 33 (pixel q = 32 (xg - 18) + lx -> row q >> 1, column 32 + (q & 1)) -- x cout tile wave & 1
  const int m_x = wave & 1;
  const int xg = 16 + (wave >> 1);
  const bool x_rows = xg < 18;  // wave uniform
  const int xq = (xg - 18) * 32 + lx;
  const bool xvalid = x_rows || xq < 2 * PH;
  const int xqc = (!x_rows && xq < 2 * PH) ? xq : 0;
  const int xrow = x_rows ? xg : xqc >> 1, xcol = x_rows ? lx : 32 + (xqc & 1);
  unsigned xoff[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int c = xcol + dx;
    xoff[dx] = (unsigned)(xrow * IN_ROW + c * 32 + ((half ^ ((c >> 3) & 1)) * 16));
  }
  const unsigned xwoff = (unsigned)(xrow * IP_ROW + xcol * 32 + ((half ^ ((xcol >> 3) & 1)) * 16));
  const unsigned woff = (unsigned)(2 * wave * IP_ROW + lx * 32 + ((half ^ ((lx >> 3) & 1)) * 16));  // big item: row 2 wave, column lx
  const float low_a = p.relu_a ? 0.0f : -INFINITY, low_b = p.relu_b ? 0.0f : -INFINITY;
  const unsigned pixb_out = p.planar ? 32u : 128u;

  // Waits are the BUILTIN s_waitcnt (vmcnt in bits 3:0, expcnt 7 and lgkmcnt 15 = "do not wait"), not inline asm: the compiler's
  // own wait insertion sees them. It waits for every LDS-DMA it believes in flight in front of a C++ LDS access it cannot
  // prove disjoint (the bias reads, the intermediate tile's ds_writes); told that the counter was empty in front of the
  // epilogue's stores, it has no reason to wait for THEM at the next tile's first LDS read.
// Barriers inside the tile loop are the bare s_barrier: __syncthreads() carries a workgroup-scope fence, in front of which the
  // compiler drains vmcnt whenever an LDS-DMA is in flight (LDS-DMA completes through vmcnt) -- which is exactly what the
  // counted wait of stage B1 must not do. Every barrier is preceded by this wave's own waits (copies: vmcnt, LDS: lgkmcnt).
#define SA_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(0x0F70 | (n))
#define SA_WAIT_VM_LGKM0(n) __builtin_amdgcn_s_waitcnt(0x0070 | (n))
  // One stage of MFMAs: nine taps on the B fragments of halo rows 0..3 x columns 0..2 (read ONCE, up front) and the slot's A
  // fragments (read one tap ahead); XTRA (conv-a) adds the wave's extra unit: one more pixel group x one cout tile, its two
  // fragments read one tap ahead as well. sched_barrier(0) between the taps keeps the compiler from sinking the reads down to
  // their uses (it did: three fragments in flight, a wait in front of every MFMA pair).
  auto stage = [&](auto xtra_c, const unsigned char* bb, const unsigned (&off)[3], int row_bytes, const unsigned char* wt,
                   f32x16 (&acc)[2][R], f32x16& accx) {
    constexpr bool XTRA = decltype(xtra_c)::value;
    const unsigned char* wl = wt + lane * 16;
    mfma_h8 bfr[R + 2][3], a[2][2], xa[2], xb[2];
    a[0][0] = *reinterpret_cast<const mfma_h8*>(wl);
    a[0][1] = *reinterpret_cast<const mfma_h8*>(wl + 9 * 1024);
#pragma unroll
    for (int rr = 0; rr < R + 2; ++rr)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) bfr[rr][dx] = *reinterpret_cast<const mfma_h8*>(bb + off[dx] + rr * row_bytes);
    if constexpr (XTRA) {
      xa[0] = *reinterpret_cast<const mfma_h8*>(wl + (m_x * 9) * 1024);
      xb[0] = *reinterpret_cast<const mfma_h8*>(bb + xoff[0]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap % 3, cb = tap & 1, nb = cb ^ 1;
      if (tap < 8) {
        a[nb][0] = *reinterpret_cast<const mfma_h8*>(wl + (tap + 1) * 1024);
        a[nb][1] = *reinterpret_cast<const mfma_h8*>(wl + (9 + tap + 1) * 1024);
        if constexpr (XTRA) {
          xa[nb] = *reinterpret_cast<const mfma_h8*>(wl + (m_x * 9 + tap + 1) * 1024);
          xb[nb] = *reinterpret_cast<const mfma_h8*>(bb + xoff[(tap + 1) % 3] + ((tap + 1) / 3) * IN_ROW);
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        acc[0][r] = SA_MFMA_32x32x16(a[cb][0], bfr[r + dy][dx], acc[0][r], 0, 0, 0);
        acc[1][r] = SA_MFMA_32x32x16(a[cb][1], bfr[r + dy][dx], acc[1][r], 0, 0, 0);
      }
      if constexpr (XTRA) accx = SA_MFMA_32x32x16(xa[cb], xb[cb], accx, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  f32x16 accA[2][R], accX;
  auto init_a = [&]() {  // conv-a's accumulators start at its bias
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bq = *reinterpret_cast<const float4*>(bias_lds + m * 32 + 8 * g + 4 * half);
#pragma unroll
        for (int r = 0; r < R; ++r) accA[m][r][4 * g + 0] = bq.x, accA[m][r][4 * g + 1] = bq.y, accA[m][r][4 * g + 2] = bq.z, accA[m][r][4 * g + 3] = bq.w;
      }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 bq = *reinterpret_cast<const float4*>(bias_lds + m_x * 32 + 8 * g + 4 * half);
      accX[4 * g + 0] = bq.x, accX[4 * g + 1] = bq.y, accX[4 * g + 2] = bq.z, accX[4 * g + 3] = bq.w;
    }
  };

  // ---- once per workgroup: biases -> LDS, the first tile's copies
  if (tid < 128) reinterpret_cast<float*>(smem + BIAS_OFF)[tid] = tid < 64 ? p.bias_a[tid] : p.bias_b[tid - 64];
  Tile cur = decode(L);
  {
    unsigned v[3];
    make_voff(cur, v);
    issue_w(rwa, 2, 0, 0);
    issue_in(cur, v, 0);
    issue_in(cur, v, 1);
  }
  SA_WAIT_VM(0);
  __syncthreads();  // the biases are in LDS for every wave

#pragma clang loop unroll(disable)
  for (;;) {
    const int L_next = L + L_step;
    const bool more = L_next < L_end;  // workgroup uniform
    init_a();  // (LDS reads behind the previous tile's stores: the compiler knows that no LDS-DMA is in flight -- SA_WAIT_VM)

    // ================= phase A: conv-a (32 -> 64) on the 18 x 34 halo pixels, k-halves 0 and 1
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (k == 0) {
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0) only: the stores of the previous epilogue stay in flight
        __builtin_amdgcn_s_barrier();  // every wave's copies for this tile landed (each waited for its own in front of the previous epilogue / in
                          // the prologue); the previous tile's B3 is finished (slot 1, the intermediate tile)
        issue_w(rwa, 2, 1, 1);
      } else {
        SA_WAIT_VM_LGKM0(0);
        __builtin_amdgcn_s_barrier();  // k-half 1 landed; A0 is finished (slot 0)
        issue_w(rwb, 4, 0, 0);
      }
      const unsigned char* inp = smem + IN_OFF + k * IN_PLANE;
      const unsigned char* wt = smem + RING_OFF + k * SLOT;
      stage(std::true_type{}, inp, aoff, IN_ROW, wt, accA, accX);
    }
    // ---- epilogue a: ReLU, 16-bit pack, zero outside the image (= conv-b's SAME padding), 16-byte stores into the planes
    {
      auto put = [&](const f32x16& d, unsigned mask, bool store, unsigned char* base, int m) {
        uint2 pk[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          pk[g].x = sa::f2h2(fmaxf(d[4 * g + 0], low_a), fmaxf(d[4 * g + 1], low_a)) & mask;
          pk[g].y = sa::f2h2(fmaxf(d[4 * g + 2], low_a), fmaxf(d[4 * g + 3], low_a)) & mask;
        }
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          uint2 x = pk[2 * pr], y = pk[2 * pr + 1];
          sa::swap32(x.x, y.x);
          sa::swap32(x.y, y.y);
          if (store) *reinterpret_cast<uint4*>(base + (2 * m + pr) * IP_PLANE) = make_uint4(x.x, x.y, y.x, y.y);
        }
      };
      const bool colok = (unsigned)(cur.x0 + lx - 1) < (unsigned)W;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const bool rowok = (unsigned)(cur.y0 + 2 * wave + r - 1) < (unsigned)H;  // wave uniform
        const unsigned mask = (rowok && colok) ? 0xFFFFFFFFu : 0u;
#pragma unroll
        for (int m = 0; m < 2; ++m) put(accA[m][r], mask, true, smem + INTER_OFF + woff + r * IP_ROW, m);
      }
      {
        const bool in_img = (unsigned)(cur.y0 + xrow - 1) < (unsigned)H && (unsigned)(cur.x0 + xcol - 1) < (unsigned)W;
        put(accX, in_img ? 0xFFFFFFFFu : 0u, xvalid, smem + INTER_OFF + xwoff, m_x);
      }
    }

    // ================= phase B: conv-b (64 -> 64), wave owns rows 2 wave, 2 wave + 1 x both cout tiles
    f32x16 acc[2][R];
    Tile nxt = cur;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      // chunk c landed; the stage before is finished (c == 0: the intermediate tile is complete). c == 1: the next tile's input
      // planes, queued BEHIND chunk 1 in B0, may stay in flight (the counter retires in order): they have until B2.
      if (c == 1 && more) {
        if (wave < N_INP - 2 * NW) SA_WAIT_VM_LGKM0(6); else SA_WAIT_VM_LGKM0(4);  // (wave uniform) 2 x 3 or 2 x 2 input pieces per wave
      } else {
        SA_WAIT_VM_LGKM0(0);  // (c == 0: and this wave's ds_writes of the intermediate tile)
      }
      __builtin_amdgcn_s_barrier();
      if (c == 0) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 bq = *reinterpret_cast<const float4*>(bias_lds + 64 + m * 32 + 8 * g + 4 * half);
#pragma unroll
            for (int r = 0; r < R; ++r) acc[m][r][4 * g + 0] = bq.x, acc[m][r][4 * g + 1] = bq.y, acc[m][r][4 * g + 2] = bq.z, acc[m][r][4 * g + 3] = bq.w;
          }
      }
      if (c < 3) {
        issue_w(rwb, 4, c + 1, (c + 1) & 1);
        if (c == 0 && more) {  // A1 is finished: both input planes are free
          unsigned vnext[3];
          nxt = decode(L_next);
          make_voff(nxt, vnext);
          issue_in(nxt, vnext, 0);
          issue_in(nxt, vnext, 1);
        }
      } else if (more) {
        issue_w(rwa, 2, 0, 0);
      }
      stage(std::false_type{}, smem + INTER_OFF + c * IP_PLANE, boff, IP_ROW, smem + RING_OFF + (c & 1) * SLOT, acc, accX);
    }

    // ---- epilogue b (as conv3x3_dma_kernel's plain epilogue). The copies queued in B3 are waited for HERE, in front of the
    // stores, so that the next tile's first barrier needs no memory wait at all.
    SA_WAIT_VM(0);
    {
      const int gx = cur.x0 + lx;
      const bool colok = gx < W;
      auto act = [&](int m, int r, int i) { return fmaxf(acc[m][r][i], low_b); };
      auto store_pieces = [&](unsigned char* row_base, size_t blk_bytes, int m, unsigned lane_off, bool ok, const uint2 (&pk)[4]) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          uint2 x = pk[2 * pr], y = pk[2 * pr + 1];
          sa::swap32(x.x, y.x);
          sa::swap32(x.y, y.y);
          if (ok) *reinterpret_cast<uint4*>(row_base + (size_t)(2 * m + pr) * blk_bytes + lane_off) = make_uint4(x.x, x.y, y.x, y.y);
        }
      };
      if (p.dst) {
        unsigned char* frame = reinterpret_cast<unsigned char*>(p.dst) + (size_t)cur.b * H * W * 128;
        const size_t blk = p.planar ? (size_t)H * W * 32 : (size_t)32;
        const unsigned lane_off = (unsigned)gx * pixb_out + (unsigned)half * 16u;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const int gy = cur.y0 + wave * R + r;  // wave uniform
            uint2 pk[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              pk[g].x = sa::f2h2(act(m, r, 4 * g + 0), act(m, r, 4 * g + 1));
              pk[g].y = sa::f2h2(act(m, r, 4 * g + 2), act(m, r, 4 * g + 3));
            }
            store_pieces(frame + (size_t)gy * W * pixb_out, blk, m, lane_off, colok && gy < H, pk);
          }
      }
      if (p.dst_pool) {
        const int gy = cur.y0 + wave * R;  // wave uniform, even
        unsigned char* frame = reinterpret_cast<unsigned char*>(p.dst_pool) + (size_t)cur.b * (H / 2) * (W / 2) * 128;
        const size_t blk = p.planar ? (size_t)(H / 2) * (W / 2) * 32 : (size_t)32;
        const unsigned lane_off = (unsigned)(gx >> 1) * pixb_out + (unsigned)half * 16u;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          uint2 pk[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float t4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float t = fmaxf(act(m, 0, 4 * g + j), act(m, 1, 4 * g + j));
              t4[j] = fmaxf(t, sa::dpp_xor1(t));
            }
            pk[g].x = sa::f2h2(t4[0], t4[1]);
            pk[g].y = sa::f2h2(t4[2], t4[3]);
          }
          store_pieces(frame + (size_t)(gy >> 1) * (W / 2) * pixb_out, blk, m, lane_off, !(lane & 1) && colok && gy < H, pk);
        }
      }
    }
    if (!more) break;
    cur = nxt;
    L = L_next;
  }
#endif
}

}  // namespace

// the 32 -> 64 -> 64 form of sa_conv3x3_pair_bf16 (convpair.hip checks the arguments and dispatches here)
int sa_convpair64_launch(const void* src, const void* wa, const float* bias_a, int relu_a, const void* wb, const float* bias_b,
                         int relu_b, int B, int H, int W, void* dst, void* dst_pool, int layout, hipStream_t stream) {
  SA_REQUIRE((size_t)H * W * 128 < 0xFFFFFF00ull, "sa_conv3x3_pair_bf16: one output frame must be smaller than 4 GiB");
  Pair64Params p;
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
  p.tiles_x = (W + TW - 1) / TW;
  p.tiles_y = (H + TH - 1) / TH;
  const size_t nblk = (size_t)p.tiles_x * p.tiles_y * B;
  if (nblk > 0x7fffffffull) return sa::fail(SA_ERR_INVALID_ARG, "sa_conv3x3_pair_bf16: grid too large");
  p.n_tiles = (int)nblk;
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0, lds_max = 0;
    SA_HIP_CHECK(hipGetDevice(&dev));
    SA_HIP_CHECK(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev));
    if (lds_max < LDS_BYTES)
      return sa::fail(SA_ERR_UNSUPPORTED, "sa_conv3x3_pair_bf16: the 32 -> 64 -> 64 block needs %d bytes of LDS per workgroup (device: %d)",
                      LDS_BYTES, lds_max);
    SA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&convpair64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    SA_HIP_CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  }
  // one workgroup per CU (LDS); sa_conv3x3_set_grid_limit(n) launches at most n workgroups (tests: uneven tile shares)
  size_t grid = (size_t)n_cu;
  const int limit = sa_internal_grid_limit();
  if (limit > 0 && (size_t)limit < grid) grid = (size_t)limit;
  if (grid > nblk) grid = nblk;
  if (grid < 8 && nblk >= 8) grid = 8;  // the XCD schedule hands every XCD a range: at least one workgroup each
  hipLaunchKernelGGL(convpair64_kernel, dim3((unsigned)grid), dim3(NW * 64), LDS_BYTES, stream, p);
  SA_LAUNCH_CHECK();
  return SA_OK;
}
