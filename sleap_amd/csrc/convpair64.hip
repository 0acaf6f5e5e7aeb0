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
//   epilogue b -- ReLU, 2 x 2 max and rounding on packed pairs; the 12 store instructions per wave are DEFERRED into the next
//   tile's A0, one behind each tap's MFMAs (conv-a k-half 1 is queued in front of them and waited for with a counted vmcnt).
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

// the tap of a stage behind which the copies for a later stage are queued (-1: at the stage's head). Measured on one box, the
// benchmark network's data (profiles/r06_pair64.md): -1 0.417, 3 0.409, 1 0.404 ms -- the copy instructions (2-3 per wave, ~100
// cycles of issue each) go out under the wave's own MFMAs instead of in front of its first fragment reads.
#if !defined(SA_PAIR64_ITAP)
#define SA_PAIR64_ITAP 1
#endif

#if !defined(SA_PAIR64_SWZ2)
#define SA_PAIR64_SWZ2 1  // conflict-free stores of the intermediate planes (0: the first form, A/B)
#endif
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

constexpr int TH = 16, TW = 32;
constexpr int PH = TH + 2, PW = TW + 2;  // intermediate halo tile
constexpr int QH = TH + 4, QW = TW + 4;  // input halo tile
constexpr int IP_ROW = PW * 32, IP_PLANE = PH * IP_ROW;            // 1088, 19584
constexpr int IN_ROW = QW * 32, IN_USED = QH * IN_ROW;             // 1152, 23040
constexpr int N_INP = (IN_USED + 1023) / 1024, IN_PLANE = N_INP * 1024;  // 23, 23552
constexpr int SLOT = 18 * 1024;
[[maybe_unused]] constexpr int INTER_OFF = 0, IN_OFF = 4 * IP_PLANE, RING_OFF = IN_OFF + 2 * IN_PLANE, BIAS_OFF = RING_OFF + 2 * SLOT;
constexpr int LDS_BYTES = BIAS_OFF + 128 * 4;
static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU: 160 KiB");
[[maybe_unused]] constexpr unsigned OOB = 0xFFFFFF00u;

// SA_PAIR64_STAMP (an instrumented A/B build, tools/pair64_probe.py -- never the product library): every wave sums the shader
// cycles (s_memtime) of the segments of its tiles and adds them to g_stamp64 at its end: [0] waves, [1] tiles, [2] life,
// [3] A bodies, [4] epilogue a, [5] B bodies, [6] epilogue b (behind its wait), [8 + i] the wait + barrier in front of stage i
// (A0, A1, B0..B3), [14] the wait in front of epilogue b, [15] tile set-up (accumulator init, decode), [16 + i] the memory wait alone
// ([8 + i] is then the barrier alone), [22] epilogue b up to its first store.
#if defined(SA_PAIR64_STAMP)
__device__ unsigned long long g_stamp64[24];
#define ST_DECL unsigned long long st_t = __builtin_readcyclecounter(), st_t0 = st_t, st_a[24] = {0}
#define ST(i)                                                       \
  do {                                                              \
    const unsigned long long st_now = __builtin_readcyclecounter(); \
    st_a[i] += st_now - st_t;                                       \
    st_t = st_now;                                                  \
  } while (0)
#define ST_FLUSH                                                                            \
  do {                                                                                      \
    if ((threadIdx.x & 63) == 0) {                                                          \
      st_a[0] = 1, st_a[2] = __builtin_readcyclecounter() - st_t0;                          \
      for (int i_ = 0; i_ < 24; ++i_) atomicAdd(&g_stamp64[i_], st_a[i_]);                  \
    }                                                                                       \
  } while (0)
#else
#define ST_DECL
#define ST(i)
#define ST_FLUSH
#endif

// NW = waves per workgroup, R = TH / NW output rows per wave. NW = 8: two waves per SIMD (256 registers each); NW = 4: ONE wave per
// SIMD with 512 registers -- nothing arbitrates for the matrix pipe, a tap is 8-10 MFMAs (256-320 cycles) behind one set of
// fragment reads, and half the LDS fragment traffic (every A fragment feeds four rows instead of two).
template <int NW>
__global__ void __launch_bounds__(NW * 64, NW == 8 ? 2 : 1)
convpair64_kernel(const Pair64Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int R = TH / NW;
  constexpr int XM = NW == 8 ? 1 : 2;            // cout tiles of a wave's extra conv-a unit
  constexpr int NWJ = (18 + NW - 1) / NW;        // weight slabs per wave and copy (<=)
  constexpr int NIJ = (N_INP + NW - 1) / NW;     // input pieces per wave and plane (<=)
  // ITAP: the tap of a stage behind which the copies for a LATER stage are queued (-1: at the stage's head, in front of its first
  // fragment reads). YS = the stage's deferred stores that are queued behind those copies (store taps: 2, 6 / 1, 3, 5, 7).
  constexpr int ITAP = SA_PAIR64_ITAP;
  constexpr int YS = ITAP < 0 ? R : (R == 2 ? (ITAP <= 2 ? 2 : (ITAP <= 6 ? 1 : 0)) : (ITAP <= 1 ? 4 : (ITAP <= 3 ? 3 : (ITAP <= 5 ? 2 : (ITAP <= 7 ? 1 : 0)))));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, lx = lane & 31;
  const int H = p.H, W = p.W;

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
    for (int jj = 0; jj < NWJ; ++jj) {
      const int j = jj * NW + wave;  // wave uniform
      if (j < 18) {
        const int m = j >= 9 ? 1 : 0, tap = j - 9 * m;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(smem + RING_OFF + s * SLOT + j * 1024), 16, wv,
                                                 ((m * k16n + k) * 9 + tap) * 1024, 0, 0);
      }
    }
  };
  // per-lane source offsets of this wave's (up to) three copy pieces of an input plane (the same for both planes)
  auto make_voff = [&](const Tile& t, unsigned (&v)[NIJ]) {
    int ln = lane;  // (an opaque copy: what is derived from it is re-derived per tile, not kept in registers across the stages)
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int jj = 0; jj < NIJ; ++jj) {
      const int i = jj * NW + wave;
      const int o = i * 1024 + ln * 16;
      const int pl = o >> 5, s = (o >> 4) & 1;
      const int ty = pl / QW, tx = pl - ty * QW;
      const int gy = t.y0 + ty - 2, gx = t.x0 + tx - 2;
      const bool ok = i < N_INP && pl < QH * QW && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      v[jj] = ok ? (unsigned)(gy * W + gx) * pixb_in + (unsigned)((s ^ ((tx >> 3) & 1)) * 16) : OOB;
    }
  };
  auto issue_in = [&](const Tile& t, const unsigned (&v)[NIJ], int k) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const unsigned char*>(p.src) + t.b * fbytes), 0, (int)fbytes, 0x00020000);
#pragma unroll
    for (int jj = 0; jj < NIJ; ++jj) {
      const int i = jj * NW + wave;  // wave uniform
      if (i < N_INP)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + IN_OFF + k * IN_PLANE + i * 1024), 16, v[jj],
                                                 (int)((unsigned)k * plane_in), 0, 2);  // read once: non-temporal
    }
  };

  // ---- tile-independent per-lane LDS offsets. Swizzles depend on the tile COLUMN only, so a row is a compile-time stride.
  // Slot swizzle of the INTERMEDIATE planes (round 6, after the first profile set): ((c >> 2) ^ (c >> 3)) & 1 instead of the input
  // planes' (c >> 3) & 1. A ds_write_b128 group is eight consecutive columns of 32 bytes: with one slot for all eight, columns c and
  // c + 4 fall on the same 16-byte bank quad (2-way on every store of conv-a's output: 12 % of this kernel's LDS-active cycles);
  // flipping the slot every four columns makes the eight distinct, and the ds_read_b128 groups of conv-b (columns {0-3, 12-15,
  // 20-27} + dx and their complement) stay conflict-free -- the sixteen period-16 swizzles with both properties were enumerated.
  auto ipswz = [](int c) { return SA_PAIR64_SWZ2 ? (((c >> 2) ^ (c >> 3)) & 1) : ((c >> 3) & 1); };
  unsigned aoff[3], boff[3];  // conv-a reads (input plane, halo row 2 * wave, column lx + dx); conv-b reads (intermediate plane)
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int c = lx + dx;
    aoff[dx] = (unsigned)(R * wave * IN_ROW + c * 32 + ((half ^ ((c >> 3) & 1)) * 16));
    boff[dx] = (unsigned)(R * wave * IP_ROW + c * 32 + ((half ^ ipswz(c)) * 16));
  }
  // the extra unit of this wave: pixel group xg = 16 + (wave >> 1) -- 16, 17: halo rows 16, 17, columns lx; 18, 19: the 36 pixels of
  // columns 32, 33 (pixel q = 32 (xg - 18) + lx -> row q >> 1, column 32 + (q & 1)) -- x cout tile wave & 1
  // (NW = 4: group 16 + wave x BOTH cout tiles)
  const int m_x = NW == 8 ? (wave & 1) : 0;
  const int xg = NW == 8 ? 16 + (wave >> 1) : 16 + wave;
  const bool x_rows = xg < 18;  // wave uniform
  struct XPix {
    int row, col;
    bool valid;
  };
  auto xpix = [&](int lx_) {
    const int xq = (xg - 18) * 32 + lx_;
    const int xqc = (!x_rows && xq < 2 * PH) ? xq : 0;
    return XPix{x_rows ? xg : xqc >> 1, x_rows ? lx_ : 32 + (xqc & 1), x_rows || xq < 2 * PH};
  };
  unsigned xoff[3];
  {
    const XPix xp = xpix(lx);
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int c = xp.col + dx;
      xoff[dx] = (unsigned)(xp.row * IN_ROW + c * 32 + ((half ^ ((c >> 3) & 1)) * 16));
    }
  }
  [[maybe_unused]] const float low_a = p.relu_a ? 0.0f : -INFINITY, low_b = p.relu_b ? 0.0f : -INFINITY;
  const unsigned pixb_out = p.planar ? 32u : 128u;

  // Waits are the BUILTIN s_waitcnt (vmcnt in bits 3:0, expcnt 7 and lgkmcnt 15 = "do not wait"), not inline asm: the compiler's
  // own wait insertion sees them. It waits for every LDS-DMA it believes in flight in front of a C++ LDS access it cannot
  // prove disjoint (the bias reads, the intermediate tile's ds_writes); told that the counter was empty in front of the
  // epilogue's stores, it has no reason to wait for THEM at the next tile's first LDS read.
// Barriers inside the tile loop are the bare s_barrier: __syncthreads() carries a workgroup-scope fence, in front of which the
  // compiler drains vmcnt whenever an LDS-DMA is in flight (LDS-DMA completes through vmcnt) -- which is exactly what the
  // counted wait of stage B1 must not do. Every barrier is preceded by this wave's own waits (copies: vmcnt, LDS: lgkmcnt).
#define SA_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 15) | (((n) >> 4) << 14))
#define SA_WAIT_VM_LGKM0(n) __builtin_amdgcn_s_waitcnt(0x0070 | ((n) & 15) | (((n) >> 4) << 14))
  // One stage of MFMAs: nine taps on the B fragments of halo rows 0..3 x columns 0..2 (read ONCE, up front) and the slot's A
  // fragments (read one tap ahead); XTRA (conv-a) adds the wave's extra unit: one more pixel group x one cout tile, its two
  // fragments read one tap ahead as well; `between(tap)` runs behind the MFMAs of a tap. sched_barrier(0) between the taps keeps the compiler from sinking the reads down to
  // their uses (it did: three fragments in flight, a wait in front of every MFMA pair).
  auto stage = [&](auto xtra_c, const unsigned char* bb, const unsigned (&off)[3], int row_bytes, const unsigned char* wt,
                   f32x16 (&acc)[2][R], f32x16 (&accx)[XM], auto&& between) {
    constexpr bool XTRA = decltype(xtra_c)::value;
    const unsigned char* wl = wt + lane * 16;
    mfma_h8 bfr[R + 2][3], a[2][2], xa, xb;
    a[0][0] = *reinterpret_cast<const mfma_h8*>(wl);
    a[0][1] = *reinterpret_cast<const mfma_h8*>(wl + 9 * 1024);
    // halo rows 0 .. R-1 up front; rows R, R+1 one fragment per tap, three taps ahead of their first use
#pragma unroll
    for (int rr = 0; rr < R; ++rr)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) bfr[rr][dx] = *reinterpret_cast<const mfma_h8*>(bb + off[dx] + rr * row_bytes);
    between(-1);  // (while the first fragments are on their way)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap % 3, cb = tap & 1, nb = cb ^ 1;
      if (tap < 8) {
        a[nb][0] = *reinterpret_cast<const mfma_h8*>(wl + (tap + 1) * 1024);
        a[nb][1] = *reinterpret_cast<const mfma_h8*>(wl + (9 + tap + 1) * 1024);
      }
      if constexpr (XTRA) {  // the extra unit's fragments: read at the head of the tap, used behind its other MFMAs
        if constexpr (XM == 1) xa = *reinterpret_cast<const mfma_h8*>(wl + (m_x * 9 + tap) * 1024);
        xb = *reinterpret_cast<const mfma_h8*>(bb + xoff[dx] + dy * IN_ROW);
      }
      if (tap < 6) bfr[R + tap / 3][tap % 3] = *reinterpret_cast<const mfma_h8*>(bb + off[tap % 3] + (R + tap / 3) * row_bytes);
      __builtin_amdgcn_sched_barrier(0);  // (reads first: left alone the scheduler puts them behind the tap's third MFMA)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        acc[0][r] = SA_MFMA_32x32x16(a[cb][0], bfr[r + dy][dx], acc[0][r], 0, 0, 0);
        acc[1][r] = SA_MFMA_32x32x16(a[cb][1], bfr[r + dy][dx], acc[1][r], 0, 0, 0);
      }
      if constexpr (XTRA) {
        if constexpr (XM == 1) {
          accx[0] = SA_MFMA_32x32x16(xa, xb, accx[0], 0, 0, 0);
        } else {
          accx[0] = SA_MFMA_32x32x16(a[cb][0], xb, accx[0], 0, 0, 0);
          accx[1] = SA_MFMA_32x32x16(a[cb][1], xb, accx[1], 0, 0, 0);
        }
      }
      between(tap);  // (the previous tile's deferred stores: issued in the shadow of this tap's MFMAs)
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  f32x16 accA[2][R], accX[XM];
  // Bias reads are OPAQUE ds_reads (inline asm) behind one explicit lgkmcnt(0): in front of a C++ load from LDS the compiler drains
  // vmcnt whenever it believes a copy in flight whose target it cannot tell from the load's address -- it cannot follow the
  // counted waits below across the `pend` branches -- and with it the deferred stores.
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  const unsigned bias_addr = (unsigned)(uintptr_t)(lds_ptr_t)(smem + BIAS_OFF) + (unsigned)half * 16u;
#define SA_LDS_READ4(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
  auto init_a = [&]() {  // conv-a's accumulators start at its bias
    f32x4v bq[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      SA_LDS_READ4(bq[m][0], bias_addr, 0 + 128 * m);
      SA_LDS_READ4(bq[m][1], bias_addr, 32 + 128 * m);
      SA_LDS_READ4(bq[m][2], bias_addr, 64 + 128 * m);
      SA_LDS_READ4(bq[m][3], bias_addr, 96 + 128 * m);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < R; ++r) accA[m][r][4 * g + j] = bq[m][g][j];
        if constexpr (XM == 1) {
          accX[0][4 * g + j] = m_x ? bq[1][g][j] : bq[0][g][j];  // (wave uniform)
        } else {
          accX[0][4 * g + j] = bq[0][g][j];
          accX[XM - 1][4 * g + j] = bq[1][g][j];
        }
      }
  };
  auto init_b = [&](f32x16 (&acc)[2][R]) {  // conv-b's
    f32x4v bq[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      SA_LDS_READ4(bq[m][0], bias_addr, 256 + 128 * m);
      SA_LDS_READ4(bq[m][1], bias_addr, 288 + 128 * m);
      SA_LDS_READ4(bq[m][2], bias_addr, 320 + 128 * m);
      SA_LDS_READ4(bq[m][3], bias_addr, 352 + 128 * m);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < R; ++r) acc[m][r][4 * g + j] = bq[m][g][j];
  };

  // ---- once per workgroup: biases -> LDS, the first tile's copies
  if (tid < 128) reinterpret_cast<float*>(smem + BIAS_OFF)[tid] = tid < 64 ? p.bias_a[tid] : p.bias_b[tid - 64];
  Tile cur = decode(L);
  {
    unsigned v[NIJ];
    make_voff(cur, v);
    issue_w(rwa, 2, 0, 0);
    issue_in(cur, v, 0);
    issue_in(cur, v, 1);
  }
  SA_WAIT_VM(0);
  __syncthreads();  // the biases are in LDS for every wave
  ST_DECL;

  // ---- conv-b's outputs of a tile, packed to the storage type (ReLU and the 2 x 2 max applied), and their stores. The stores
  // of tile t are DEFERRED into stage A0 of tile t + 1, one 16-byte piece per lane behind each tap's MFMAs: as an epilogue of
  // their own they cost 5-7 k cycles per tile during which no wave of the CU issues an MFMA (the whole workgroup reaches the
  // epilogue together: there is no second workgroup to cover it -- profiles/r06_pair64_stamps.md).
  uint2 pkf[2][R][4];  // [cout tile][row][4-channel group]
  // max of two packed pairs of the storage type. Rounding is monotonic, so the max of ROUNDED values is the rounded max (the
  // bits the fp32 form produces, up to the sign of a zero): ReLU and the 2 x 2 max run on the packed values.
  auto pkmax = [](uint32_t u, uint32_t v) -> uint32_t {
#if SA_HAS_PK_MAX
    return sa::pk_max(u, v);
#else
    const float a0 = sa::h2f((uint16_t)u), a1 = sa::h2f((uint16_t)(u >> 16)), b0 = sa::h2f((uint16_t)v), b1 = sa::h2f((uint16_t)(v >> 16));
    return sa::f2h2(fmaxf(a0, b0), fmaxf(a1, b1));  // (bf16 -> f32 is exact: the same values back)
#endif
  };
  auto pack_b = [&](const f32x16 (&acc)[2][R]) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < R; ++r) {
#if SA_HAS_PK_MAX
          const uint32_t lowpk = p.relu_b ? 0u : SA_PK_NEG_INF;
          pkf[m][r][g].x = sa::pk_max(sa::f2h2(acc[m][r][4 * g + 0], acc[m][r][4 * g + 1]), lowpk);
          pkf[m][r][g].y = sa::pk_max(sa::f2h2(acc[m][r][4 * g + 2], acc[m][r][4 * g + 3]), lowpk);
#else
          pkf[m][r][g].x = sa::f2h2(fmaxf(acc[m][r][4 * g + 0], low_b), fmaxf(acc[m][r][4 * g + 1], low_b));
          pkf[m][r][g].y = sa::f2h2(fmaxf(acc[m][r][4 * g + 2], low_b), fmaxf(acc[m][r][4 * g + 3], low_b));
#endif
        }
  };
  auto pool2 = [&](uint32_t u, uint32_t v) -> uint32_t {  // rows: the wave's two; columns: lanes l, l ^ 1
    const uint32_t t = pkmax(u, v);
    return pkmax(t, (uint32_t)__builtin_amdgcn_update_dpp((int)t, (int)t, 0xB1, 0xF, 0xF, false));
  };
  const size_t blk_full = p.planar ? (size_t)H * W * 32 : (size_t)32, blk_pool = p.planar ? (size_t)(H / 2) * (W / 2) * 32 : (size_t)32;
  int ln_s = lane;  // (opaque per tile, see the loop: the stores' lane offsets are not kept across the stages)
  // piece i of the full-resolution output (4 R per wave): cout tile i / (2 R), row (i / 2) % R, 16-channel block i & 1 of the tile
  auto store_full = [&](const Tile& t, int i, auto checked_c) {
    const int m = i / (2 * R), r = (i >> 1) % R, pr = i & 1;
    const int gy = t.y0 + wave * R + r, gx = t.x0 + (ln_s & 31);  // gy wave uniform
    uint2 x = pkf[m][r][2 * pr], y = pkf[m][r][2 * pr + 1];
    sa::swap32(x.x, y.x);
    sa::swap32(x.y, y.y);
    unsigned char* base = reinterpret_cast<unsigned char*>(p.dst) + (size_t)t.b * H * W * 128 + (size_t)(2 * m + pr) * blk_full +
                          (size_t)gy * W * pixb_out;
    const unsigned lane_off = (unsigned)gx * pixb_out + (unsigned)(ln_s >> 5) * 16u;
    if (!decltype(checked_c)::value || (gx < W && gy < H)) *reinterpret_cast<uint4*>(base + lane_off) = make_uint4(x.x, x.y, y.x, y.y);
  };
  // piece i of the pooled output (2 R per wave): cout tile i / R, row pair (i / 2) % (R / 2), 16-channel block i & 1
  auto store_pool = [&](const Tile& t, int i, auto checked_c) {
    const int m = i / R, rp = (i >> 1) % (R / 2), pr = i & 1;
    const int gy = t.y0 + wave * R + 2 * rp, gx = t.x0 + (ln_s & 31);
    uint2 x, y;  // (computed here, from the full-resolution values: 16 registers less to carry into the next tile)
    x.x = pool2(pkf[m][2 * rp][2 * pr].x, pkf[m][2 * rp + 1][2 * pr].x), x.y = pool2(pkf[m][2 * rp][2 * pr].y, pkf[m][2 * rp + 1][2 * pr].y);
    y.x = pool2(pkf[m][2 * rp][2 * pr + 1].x, pkf[m][2 * rp + 1][2 * pr + 1].x), y.y = pool2(pkf[m][2 * rp][2 * pr + 1].y, pkf[m][2 * rp + 1][2 * pr + 1].y);
    sa::swap32(x.x, y.x);
    sa::swap32(x.y, y.y);
    unsigned char* base = reinterpret_cast<unsigned char*>(p.dst_pool) + (size_t)t.b * (H / 2) * (W / 2) * 128 + (size_t)(2 * m + pr) * blk_pool +
                          (size_t)(gy >> 1) * (W / 2) * pixb_out;
    const unsigned lane_off = (unsigned)(gx >> 1) * pixb_out + (unsigned)(ln_s >> 5) * 16u;
    if (!(ln_s & 1) && (!decltype(checked_c)::value || (gx < W && gy < H))) *reinterpret_cast<uint4*>(base + lane_off) = make_uint4(x.x, x.y, y.x, y.y);
  };
  const bool has_full = p.dst != nullptr, has_pool = p.dst_pool != nullptr;  // uniform

  bool pend = false;  // the previous tile's stores are still to be issued (workgroup uniform)
  Tile prev = cur;
#pragma clang loop unroll(disable)
  for (;;) {
    const int L_next = L + L_step;
    const bool more = L_next < L_end;  // workgroup uniform
#if defined(SA_PAIR64_STAMP)
    st_a[1] += 1;
#endif
    // Memory waits of the tile loop. A stage's copies are queued at its head, the R deferred stores of the stage behind them: the
    // wait in front of the NEXT stage is `vmcnt(R)` -- the copies, not the stores (the counter retires in order; a vmcnt(0)
    // would put a store's round trip to HBM in front of every barrier). Without deferred stores it is vmcnt(0).
    init_a();  // (this wave's copies for the tile -- conv-a k-half 0, the input planes -- were waited for at the end of the previous tile)
    ln_s = lane;
    asm volatile("" : "+v"(ln_s));
    // the deferred stores of the PREVIOUS tile (`prev`), R pieces per stage (behind taps 2, 6 / 1, 3, 5, 7): the 2 R pooled
    // pieces in A0 / A1, the 4 R full-resolution pieces in B0-B3
    Tile nxt = cur;
    // the copies queued during stage i: A0 conv-a k-half 1; A1 conv-b chunk 0; B0 chunk 1 + the NEXT tile's input planes (A1 is
    // finished: both are free); B1, B2 chunks 2, 3; B3 the next tile's conv-a k-half 0
    auto copies = [&](int stage_i) {
      if (stage_i == 0) {
        issue_w(rwa, 2, 1, 1);
      } else if (stage_i == 1) {
        issue_w(rwb, 4, 0, 0);
      } else if (stage_i < 5) {
        issue_w(rwb, 4, stage_i - 1, (stage_i - 1) & 1);
        if (stage_i == 2 && more) {
          unsigned vnext[NIJ];
          nxt = decode(L_next);
          make_voff(nxt, vnext);
          issue_in(nxt, vnext, 0);
          issue_in(nxt, vnext, 1);
        }
      } else if (more) {
        issue_w(rwa, 2, 0, 0);
      }
    };
    auto deferred = [&](int stage_i, int tap) {
      if (ITAP >= 0 && tap == ITAP) copies(stage_i);
      if (!pend || tap < 0) return;
      int j;
      if constexpr (R == 2) {
        if (tap != 2 && tap != 6) return;
        j = tap == 6 ? 1 : 0;
      } else {
        if (!(tap & 1)) return;
        j = tap >> 1;
      }
      if (stage_i < 2)
        store_pool(prev, R * stage_i + j, std::false_type{});
      else
        store_full(prev, R * (stage_i - 2) + j, std::false_type{});
    };

    // ================= phase A: conv-a (32 -> 64) on the 18 x 34 halo pixels, k-halves 0 and 1
    ST(16);
    __builtin_amdgcn_s_barrier();  // every wave's copies landed; the previous tile's B3 is finished (slot 1, the intermediate tile)
    ST(8);
    if (ITAP < 0) copies(0);
    stage(std::true_type{}, smem + IN_OFF, aoff, IN_ROW, smem + RING_OFF, accA, accX, [&](int tap) { deferred(0, tap); });
    ST(3);
    if (pend) SA_WAIT_VM_LGKM0(YS); else SA_WAIT_VM_LGKM0(0);  // conv-a k-half 1
    ST(17);
    __builtin_amdgcn_s_barrier();  // k-half 1 landed; A0 is finished (slot 0)
    ST(9);
    if (ITAP < 0) copies(1);
    stage(std::true_type{}, smem + IN_OFF + IN_PLANE, aoff, IN_ROW, smem + RING_OFF + SLOT, accA, accX, [&](int tap) { deferred(1, tap); });
    ST(3);
    // ---- epilogue a: ReLU, 16-bit pack, zero outside the image (= conv-b's SAME padding), 16-byte stores into the planes
    {
#if SA_HAS_PK_MAX
      const uint32_t lowpk_a = p.relu_a ? 0u : SA_PK_NEG_INF;
#endif
      auto put = [&](const f32x16& d, unsigned mask, bool store, unsigned char* base, int m) {
        uint2 pk[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#if SA_HAS_PK_MAX
          pk[g].x = sa::pk_max(sa::f2h2(d[4 * g + 0], d[4 * g + 1]), lowpk_a) & mask;
          pk[g].y = sa::pk_max(sa::f2h2(d[4 * g + 2], d[4 * g + 3]), lowpk_a) & mask;
#else
          pk[g].x = sa::f2h2(fmaxf(d[4 * g + 0], low_a), fmaxf(d[4 * g + 1], low_a)) & mask;
          pk[g].y = sa::f2h2(fmaxf(d[4 * g + 2], low_a), fmaxf(d[4 * g + 3], low_a)) & mask;
#endif
        }
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          uint2 x = pk[2 * pr], y = pk[2 * pr + 1];
          sa::swap32(x.x, y.x);
          sa::swap32(x.y, y.y);
          // (an opaque ds_write: in front of a C++ store to LDS the compiler drains vmcnt -- it cannot tell the target from that of
          //  the copy of conv-b's chunk 0, in flight since A1's head -- and with it the stage's deferred stores)
          if (store) {
            typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
            const u32x4v v = {x.x, x.y, y.x, y.y};
            const unsigned addr = (unsigned)(uintptr_t)(lds_ptr_t)base + (unsigned)((2 * m + pr) * IP_PLANE);
            asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
          }
        }
      };
      int ln = lane;  // (re-derived per tile: see make_voff)
      asm volatile("" : "+v"(ln));
      const int hf = ln >> 5, lxe = ln & 31;
      const unsigned woff = (unsigned)(R * wave * IP_ROW + lxe * 32 + ((hf ^ ipswz(lxe)) * 16));  // row R wave, column lx
      const bool colok = (unsigned)(cur.x0 + lxe - 1) < (unsigned)W;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const bool rowok = (unsigned)(cur.y0 + R * wave + r - 1) < (unsigned)H;  // wave uniform
        const unsigned mask = (rowok && colok) ? 0xFFFFFFFFu : 0u;
#pragma unroll
        for (int m = 0; m < 2; ++m) put(accA[m][r], mask, true, smem + INTER_OFF + woff + r * IP_ROW, m);
      }
      {
        const XPix xp = xpix(lxe);
        const unsigned xwoff = (unsigned)(xp.row * IP_ROW + xp.col * 32 + ((hf ^ ipswz(xp.col)) * 16));
        const bool in_img = (unsigned)(cur.y0 + xp.row - 1) < (unsigned)H && (unsigned)(cur.x0 + xp.col - 1) < (unsigned)W;
#pragma unroll
        for (int xm = 0; xm < XM; ++xm) put(accX[xm], in_img ? 0xFFFFFFFFu : 0u, xp.valid, smem + INTER_OFF + xwoff, XM == 1 ? m_x : xm);
      }
    }

    ST(4);
    // ================= phase B: conv-b (64 -> 64), wave owns rows 2 wave, 2 wave + 1 x both cout tiles
    f32x16 acc[2][R];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      // chunk c landed; the stage before is finished (c == 0: the intermediate tile is complete). c == 1: the next tile's input
      // planes, queued BEHIND chunk 1 in B0, may stay in flight (the counter retires in order): they have until B2.
      if (c == 1 && more) {  // (wave uniform) 2 x NIJ or 2 x (NIJ - 1) input pieces per wave, + the stage's R deferred stores
        if (wave < N_INP - (NIJ - 1) * NW) {
          if (pend) SA_WAIT_VM_LGKM0(2 * NIJ + YS); else SA_WAIT_VM_LGKM0(2 * NIJ);
        } else {
          if (pend) SA_WAIT_VM_LGKM0(2 * NIJ - 2 + YS); else SA_WAIT_VM_LGKM0(2 * NIJ - 2);
        }
      } else {
        if (pend) SA_WAIT_VM_LGKM0(YS); else SA_WAIT_VM_LGKM0(0);  // (c == 0: lgkmcnt(0) = this wave's ds_writes of the intermediate tile)
      }
      ST(18 + c);
      __builtin_amdgcn_s_barrier();
      ST(10 + c);
      if (c == 0) init_b(acc);
      if (ITAP < 0) copies(2 + c);
      stage(std::false_type{}, smem + INTER_OFF + c * IP_PLANE, boff, IP_ROW, smem + RING_OFF + (c & 1) * SLOT, acc, accX,
            [&](int tap) { deferred(2 + c, tap); });
      ST(5);
    }

    ln_s = lane;
    asm volatile("" : "+v"(ln_s));
    // ---- epilogue b: pack now; store now only where the stores cannot be deferred (the workgroup's last tile, ragged tiles)
    pack_b(acc);
    ST(14);
    const bool pend_next = more && has_full && has_pool && cur.x0 + TW <= W && cur.y0 + TH <= H;  // workgroup uniform
    if (!pend_next) {
      if (has_full) {
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) store_full(cur, i, std::true_type{});
      }
      if (has_pool) {
#pragma unroll
        for (int i = 0; i < 2 * R; ++i) store_pool(cur, i, std::true_type{});
      }
    }
    ST(6);
    if (!more) break;
    // the next tile's conv-a k-half 0 (queued in B3, in front of B3's R deferred stores -- governed by THIS tile's `pend`)
    if (pend && pend_next) SA_WAIT_VM_LGKM0(YS); else SA_WAIT_VM_LGKM0(0);
    pend = pend_next;
    prev = cur;
    cur = nxt;
    L = L_next;
  }
  ST_FLUSH;
#endif
}

}  // namespace

#if defined(SA_PAIR64_STAMP)
extern "C" int sa_pair64_stamp_reset() {
  unsigned long long z[24] = {0};
  SA_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp64), z, sizeof(z)));
  return SA_OK;
}
extern "C" int sa_pair64_stamp_read(unsigned long long* out) {
  SA_HIP_CHECK(hipDeviceSynchronize());
  SA_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp64), 24 * sizeof(unsigned long long)));
  return SA_OK;
}
#endif

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
    SA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&convpair64_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    SA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&convpair64_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    SA_HIP_CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  }
  // one workgroup per CU (LDS); sa_conv3x3_set_grid_limit(n) launches at most n workgroups (tests: uneven tile shares)
  size_t grid = (size_t)n_cu;
  const int limit = sa_internal_grid_limit();
  if (limit > 0 && (size_t)limit < grid) grid = (size_t)limit;
  if (grid > nblk) grid = nblk;
  if (grid < 8 && nblk >= 8) grid = 8;  // the XCD schedule hands every XCD a range: at least one workgroup each
  // default: 8 waves (two per SIMD x two rows each). SA_PAIR64_WAVES=4: one wave per SIMD x four rows (A/B: 0.49 vs 0.42 ms --
  // alone on its SIMD a wave's stage heads (five copy pieces, fourteen fragment reads) and epilogues are covered by nobody)
  const char* wv_env = getenv("SA_PAIR64_WAVES");  // (read per launch: the tests switch it)
  const int waves = wv_env && atoi(wv_env) == 4 ? 4 : 8;
  if (waves == 8)
    hipLaunchKernelGGL(convpair64_kernel<8>, dim3((unsigned)grid), dim3(512), LDS_BYTES, stream, p);
  else
    hipLaunchKernelGGL(convpair64_kernel<4>, dim3((unsigned)grid), dim3(256), LDS_BYTES, stream, p);
  SA_LAUNCH_CHECK();
  return SA_OK;
}
