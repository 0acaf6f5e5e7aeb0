// 3x3 stride-1 SAME convolution on bf16 NHWC activations as an implicit GEMM on the CDNA4 matrix
// cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate), with the surrounding Keras layers fused in:
//   - bias + ReLU epilogue                       (encoder_decoder.py:117-131 Conv2D -> Activation)
//   - MaxPool2D(2,s2) folded into the tile load  (encoder_decoder.py:109-114 pool_before_convs)
//   - Concatenate([skip, x]) as a two-source K loop, with UpSampling2D(2, bilinear) of x folded
//     into the tile load                         (encoder_decoder.py:335-339, 360-362)
//
// GEMM view: M = output channels (weights are the MFMA A operand), N = pixels of a 32-wide row
// segment (B operand), K = 9 taps x input channels. A workgroup (4 waves) owns a TH x 32 pixel tile
// for MT*32 output channels; the (TH+2) x 34 input halo tile of one CK-channel chunk is staged in
// LDS once and re-read by all 9 taps with shifted addresses (no im2col materialisation); each wave
// keeps R rows x MT cout-tiles of 32x32 fp32 accumulators in registers.
//
// LDS layout: pixel stride CK*2+16 bytes -> the 16 lanes of every ds_read_b128 service group hit 16
// distinct 16-byte slots (stride/16 is odd), so B-fragment reads are bank-conflict free; packed
// weights are stored lane-linear so A-fragment reads are consecutive 16 B per lane.
#include <cstdint>
#include <cstdlib>
#include <vector>

#include "bf16.h"
#include "sa_common.h"

// UPS kernels: the tap of a chunk's nine after which the low-resolution tile of chunk c+2 is requested (a rendezvous of the
// workgroup's waves precedes it; the copy must land before the next chunk's barrier)
#if !defined(SA_UPS_LOW_TAP)
#define SA_UPS_LOW_TAP 5
#endif

namespace {

int g_grid_limit = 0;  // sa_conv3x3_set_grid_limit
int g_persistent = -1;  // sa_conv3x3_set_persistent (-1: SA_CONV_PERS of the environment, default 0)

// SA_CONV_STAMP (an instrumented A/B build, tools/stall_probe.py -- never the product library): every wave of
// conv3x3_dma_kernel sums, in scalar registers, the shader cycles (s_memtime) it spends in five segments of its life -- tile
// prologue (entry / previous epilogue -> the first chunk's wait), the `s_waitcnt vmcnt` in front of a chunk, the workgroup
// barrier behind it, the chunk body (nine taps of MFMAs + the next chunk's copy instructions), the epilogue -- and adds them to
// g_stamp at its end: [0] waves, [1] prologue, [2] vmcnt wait, [3] barrier, [4] chunk body, [5] epilogue, [6] whole wave, [7] chunks.
#if defined(SA_CONV_STAMP)
__device__ unsigned long long g_stamp[8];
#define SA_STAMP_DECL unsigned long long st_t = __builtin_readcyclecounter(), st_t0 = st_t, st_pro = 0, st_vm = 0, st_bar = 0, st_body = 0, st_epi = 0, st_n = 0
#define SA_STAMP(acc)                                        \
  do {                                                       \
    const unsigned long long st_now = __builtin_readcyclecounter(); \
    acc += st_now - st_t;                                    \
    st_t = st_now;                                           \
  } while (0)
#define SA_STAMP_FLUSH                                                            \
  do {                                                                            \
    if ((threadIdx.x & 63) == 0) {                                                \
      atomicAdd(&g_stamp[0], 1ull), atomicAdd(&g_stamp[1], st_pro), atomicAdd(&g_stamp[2], st_vm); \
      atomicAdd(&g_stamp[3], st_bar), atomicAdd(&g_stamp[4], st_body), atomicAdd(&g_stamp[5], st_epi); \
      atomicAdd(&g_stamp[6], __builtin_readcyclecounter() - st_t0), atomicAdd(&g_stamp[7], st_n); \
    }                                                                             \
  } while (0)
#else
#define SA_STAMP_DECL
#define SA_STAMP(acc)
#define SA_STAMP_FLUSH
#endif

using sa::h16x8_t;
using sa::mfma_h8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct ConvParams {
  const uint16_t* src0;
  const uint16_t* src1;
  const uint16_t* w;  // packed [CoutP32/32][CinP/16][9][64][8]
  const float* bias;  // [CoutP]
  uint16_t* dst;
  int C0P, C1P, CoutP;
  int B, H, W;  // output size
  int relu;
  int tiles_x, tiles_y, co_tiles;
};

__device__ __forceinline__ h16x8_t zero8() {
  h16x8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
  return z;
}

__device__ __forceinline__ h16x8_t max8(h16x8_t a, h16x8_t b) {
  h16x8_t o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (sa::h2f(a[j]) >= sa::h2f(b[j])) ? a[j] : b[j];
  return o;
}

// MODE bits: 1 = src1 direct, 2 = src1 upsampled x2 (bilinear), 4 = src0 read through 2x2 max-pool
template <int MT, int R, int CK, int MODE>
__global__ void __launch_bounds__(256)
conv3x3_mfma_kernel(const ConvParams p) {
  constexpr int TH = 4 * R, TW = 32, PH = TH + 2, PW = TW + 2;
  constexpr int PIX = CK * 2 + 16;  // LDS bytes per pixel
  constexpr int IN_BYTES = PH * PW * PIX;
  constexpr int KK = CK / 16;
  constexpr int W_BYTES = MT * KK * 9 * 1024;
  constexpr int PARTS = CK / 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* in_tile = smem;
  unsigned char* w_tile = smem + IN_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bid = blockIdx.x;
  const int co_t = bid % p.co_tiles;
  bid /= p.co_tiles;
  const int tx_i = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int ty_i = bid % p.tiles_y;
  const int b = bid / p.tiles_y;
  const int x0 = tx_i * TW, y0 = ty_i * TH;
  const int H = p.H, W = p.W;
  const int CinP = p.C0P + p.C1P;
  const int K16 = CinP / 16;
  const int co32_0 = co_t * MT;
  const int co32_n = (p.CoutP + 31) / 32;

  f32x16 acc[MT][R];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][r][i] = 0.0f;

  const int n_chunks = CinP / CK;
  for (int chunk = 0; chunk < n_chunks; ++chunk) {
    __syncthreads();
    // ---------------- stage the input halo tile of this channel chunk
    const int c_lo = chunk * CK;
    const bool from1 = (MODE & 3) && (c_lo >= p.C0P);
    for (int piece = tid; piece < PH * PW * PARTS; piece += 256) {
      const int part = piece % PARTS;
      const int pix = piece / PARTS;
      const int tx = pix % PW, ty = pix / PW;
      const int gy = y0 + ty - 1, gx = x0 + tx - 1;
      h16x8_t v = zero8();
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
        if (!from1) {
          const int c = c_lo + part * 8;
          if (MODE & 4) {
            const int Ws = 2 * W;
            const uint16_t* s = p.src0 + (((size_t)b * 2 * H + 2 * gy) * Ws + 2 * gx) * p.C0P + c;
            const h16x8_t a = *reinterpret_cast<const h16x8_t*>(s);
            const h16x8_t a1 = *reinterpret_cast<const h16x8_t*>(s + p.C0P);
            const h16x8_t a2 = *reinterpret_cast<const h16x8_t*>(s + (size_t)Ws * p.C0P);
            const h16x8_t a3 = *reinterpret_cast<const h16x8_t*>(s + (size_t)Ws * p.C0P + p.C0P);
            v = max8(max8(a, a1), max8(a2, a3));
          } else {
            v = *reinterpret_cast<const h16x8_t*>(p.src0 + (((size_t)b * H + gy) * W + gx) * p.C0P + c);
          }
        } else {
          const int c = c_lo - p.C0P + part * 8;
          if (MODE & 2) {
            const int Hs = H / 2, Ws = W / 2;
            int yA, yB, xA, xB;
            float wy, wx;
            sa::up2_taps(gy, Hs, yA, yB, wy);
            sa::up2_taps(gx, Ws, xA, xB, wx);
            const uint16_t* base = p.src1 + (size_t)b * Hs * Ws * p.C1P + c;
            const h16x8_t tl = *reinterpret_cast<const h16x8_t*>(base + ((size_t)yA * Ws + xA) * p.C1P);
            const h16x8_t tr = *reinterpret_cast<const h16x8_t*>(base + ((size_t)yA * Ws + xB) * p.C1P);
            const h16x8_t bl = *reinterpret_cast<const h16x8_t*>(base + ((size_t)yB * Ws + xA) * p.C1P);
            const h16x8_t br = *reinterpret_cast<const h16x8_t*>(base + ((size_t)yB * Ws + xB) * p.C1P);
#pragma unroll
            for (int j = 0; j < 8; ++j)
              v[j] = sa::f2h(sa::up2_lerp(sa::h2f(tl[j]), sa::h2f(tr[j]), sa::h2f(bl[j]), sa::h2f(br[j]), wy, wx));
          } else {
            v = *reinterpret_cast<const h16x8_t*>(p.src1 + (((size_t)b * H + gy) * W + gx) * p.C1P + c);
          }
        }
      }
      *reinterpret_cast<h16x8_t*>(in_tile + pix * PIX + part * 16) = v;
    }
    // ---------------- stage the packed weights of this chunk (lane-linear 16 B pieces)
    for (int i = tid; i < W_BYTES / 16; i += 256) {
      const int m = i / (KK * 9 * 64);
      const int rest = i % (KK * 9 * 64);
      const int co32 = co32_0 + m;
      h16x8_t v = zero8();
      if (co32 < co32_n)
        v = *reinterpret_cast<const h16x8_t*>(p.w + (((size_t)co32 * K16 + chunk * KK) * 9 * 64 + rest) * 8);
      *reinterpret_cast<h16x8_t*>(w_tile + (size_t)i * 16) = v;
    }
    __syncthreads();
    // ---------------- 9 taps x KK k-steps of MFMA
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap % 3;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        mfma_h8 a[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
          a[m] = *reinterpret_cast<const mfma_h8*>(w_tile + ((m * KK + kk) * 9 + tap) * 1024 + lane * 16);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int row = wave * R + r;
          const int pix = (row + dy) * PW + (lane & 31) + dx;
          const mfma_h8 bv =
              *reinterpret_cast<const mfma_h8*>(in_tile + pix * PIX + kk * 32 + (lane >> 5) * 16);
#pragma unroll
          for (int m = 0; m < MT; ++m)
            acc[m][r] = SA_MFMA_32x32x16(a[m], bv, acc[m][r], 0, 0, 0);
        }
      }
    }
  }

  // ---------------- epilogue: bias + ReLU, bf16 pack, 8-byte stores (4 consecutive couts per group)
  const int gx = x0 + (lane & 31);
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int co_base = (co32_0 + m) * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int gy = y0 + wave * R + r;
      if (gy >= H || gx >= W) continue;
      uint16_t* out = p.dst + (((size_t)b * H + gy) * W + gx) * p.CoutP;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = co_base + 8 * g;
        if (co >= p.CoutP) continue;
        const float4 bq = *reinterpret_cast<const float4*>(p.bias + co);
        const float bb[4] = {bq.x, bq.y, bq.z, bq.w};
        sa::h16x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = acc[m][r][4 * g + j] + bb[j];
          if (p.relu) v = fmaxf(v, 0.0f);
          o[j] = sa::f2h(v);
        }
        *reinterpret_cast<sa::h16x4_t*>(out + co) = o;
      }
    }
  }
}

// ================================================================================================
// v2: asynchronous-DMA pipeline (global_load_lds), double-buffered LDS, dense XOR-swizzled tile.
//
// Sources: src0 (+ optional src1 = Concatenate([src0, src1]), same resolution). Per CK-channel chunk the
// (18 x 34)-pixel halo tile and the MT*(CK/16)*9 KiB packed weight slab are copied HBM/L2 -> LDS by
// global_load_lds_dwordx4 (16 B per lane, no VGPR round trip) into the buffer NOT being computed on, so
// the copy of chunk c+1 overlaps the 9-tap MFMA loop of chunk c; one barrier per chunk.
// A DMA writes LDS lane-linearly (base + lane*16), so bank-conflict avoidance cannot use padding: the
// tile is dense ([pixel][CK] bf16) and the 16-byte piece q of pixel p is stored at slot q ^ swz(p)
// (swz = (p>>2)&3 for CK=32, (p>>3)&1 for CK=16) by permuting the per-lane SOURCE address; B-fragment
// reads apply the same XOR. Any 16 pixels that are distinct mod 16 -- which the lane groups of
// ds_read_b128 always are here -- then hit 16 distinct 16-byte slots. Out-of-image halo pixels read a
// zero page. Epilogue: bias + ReLU, optional full-resolution store, optional fused MaxPool2D(2) store
// (rows pair inside the wave, columns pair across lanes l, l^1).
// ================================================================================================
struct ConvParams2 {
  const uint16_t* src0;
  const uint16_t* src1;
  const uint16_t* w;
  const float* bias;
  uint16_t* dst;       // [B,H,W,CoutP] or nullptr
  uint16_t* dst_pool;  // [B,H/2,W/2,CoutP] or nullptr
  const uint16_t* zeros;
  int C0P, C1P, CoutP;
  int B, H, W;
  int relu;
  int tiles_x, tiles_y, co_tiles;
  // fused 1x1 heads (Head.make_head, heads.py:42-62) computed from the fp32 accumulators of this conv
  int n_heads;
  const float* head_w[2];  // [NH][CoutP] f32
  const float* head_b[2];  // [NH]
  float* head_dst[2];      // [B,H,W,NH] f32
  int head_c[2];           // NH <= 32
  int head_act[2];         // 0 linear, 1 sigmoid
  // fused first layer (ensure_float + Conv2D(k3)+bias+ReLU on the raw image), computed on the VALU straight
  // into the LDS tile that the MFMA loop reads; src0 is unused then
  const void* stem_src;    // [B,H,W,CIN] u8 or f32
  const float* stem_w;     // [3][3][CIN][C0P] f32
  const float* stem_b;     // [C0P]
  int stem_is_u8, stem_relu;
  // extended epilogue (EXT kernels): v = acc + bias; relu?; v = v*post_scale + post_shift (BatchNormalization placed
  // AFTER the activation, hourglass.py:36-45); v += residual (Add layer; res_mode 1 = residual is half resolution and
  // read with nearest-neighbour x2 upsampling, hourglass.py:183-191); relu_last? (ResNet: relu(bn(conv) + shortcut))
  const float* post_scale;
  const float* post_shift;
  const uint16_t* residual;
  int res_mode, relu_last;
  // activation layout of src0 / src1 / dst / dst_pool: 0 = NHWC ([B,H,W,CP]), 1 = 16-channel planes ([B,CP/16,H,W,16]):
  // a CK = 16 chunk of a halo-tile row is then ONE contiguous run of 34 x 32 bytes instead of 34 slices of 32 bytes at a
  // pixel stride of 2 CP bytes, and a store instruction of the epilogue writes 1 KiB contiguously (CK = 16 kernels only)
  int planar;
  int pix_bytes0, pix_bytes1;  // bytes between consecutive pixels of src0 / src1 (NHWC: 2 CP; planes: 32)
  unsigned blk_bytes_in;       // bytes between consecutive 16-channel blocks of a source (NHWC: 32; planes: H W 32)
  unsigned blk_bytes_in1;      // the same for src1 when it is read at half resolution (UPS kernels: planes of H/2 x W/2 pixels)
  int out_pix_bytes;           // the same for the outputs: pixel stride (NHWC: 2 CoutP; planes: 32),
  unsigned out_blk_bytes, out_blk_bytes_pool;  // 16-channel block stride of dst / dst_pool (NHWC: 32; planes: pixels per frame x 32)
  int nt_in;  // input copies carry the non-temporal hint (layers whose input tiles are read by ONE cout tile: streamed once)
  // tile index -> (cout tile, tile column, tile row, frame): when the three tile counts are powers of two (every layer of the
  // 1024 x 1024 benchmark plan) the host hands over their log2 and the decode is three scalar shifts / masks; otherwise -1 and
  // the kernel divides (three runtime divisions = three ~35-instruction float-reciprocal sequences on the vector unit, in front
  // of the tile's FIRST copy. Measured against the division form on one box: no difference, the CU's partner workgroup covers that
  // microsecond -- profiles/r04_ab_session.md section 1; kept as the cheaper code)
  int sh_co, sh_tx, sh_ty;
  // ---- fused bottleneck tail (XP kernels, round 4; resnet.py:168-253). This conv's activation h (64 channels, never stored)
  // goes through the block's 1x1 EXPAND conv in the epilogue -- y = act1(affine1(W2 h + b2) + residual), stored -- and y (as
  // stored: rounded to 16 bits) through the NEXT block's 1x1 REDUCE conv -- z = act2(affine2(W1' y + b1')), stored. Both are
  // second / third MFMA stages fed straight from accumulator registers (the fused heads' trick: lane half h, element j <->
  // channel 16 s + 8 (j >> 2) + 4 h + (j & 3), the weights packed with the same map by sa_pack_pointwise_weights).
  const uint16_t* xp_w;       // [CoutX / 32][CoutP / 16][64][8]
  const float* xp_bias;       // [CoutX]
  const float* xp_scale;      // [CoutX] or nullptr
  const float* xp_shift;
  const uint16_t* xp_res;     // y-shaped residual or nullptr
  uint16_t* xp_dst;           // y: [B,H,W,CoutX] in the launch's layout
  int CoutX, xp_relu, xp_relu_last;
  const uint16_t* rd_w;       // [CoutR / 32][CoutX / 16][64][8] or nullptr (no reduce stage)
  const float* rd_bias;
  const float* rd_scale;
  const float* rd_shift;
  uint16_t* rd_dst;           // z: [B,H,W,CoutR], CoutR == 64
  int CoutR, rd_relu, rd_relu_last;
  int xp_pix_bytes;           // bytes between neighbouring pixels of y (NHWC: 2 CoutX; planes: 32)
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int CK>
__device__ __forceinline__ int swz(int p) {
  return (CK == 32) ? ((p >> 2) & 3) : ((p >> 3) & 1);
}

// MT: 32-cout tiles per workgroup; CK: channels per chunk; NW: waves per workgroup; R: rows per wave
// (tile = NW*R x 32 pixels); NBUF: LDS stages (2 = the copy of chunk c+1 overlaps the MFMAs of chunk c).
//   compute-bound (multi-chunk) layers: NW=8, R=2, NBUF=2 -> two waves per SIMD hide each other's LDS latency
//   single-chunk layers (Cin <= CK, HBM-bound): NW=4, R=2, NBUF=1 -> small footprint, many workgroups per CU
// Copies are buffer_load_dwordx4 ... lds (raw buffer descriptor per frame): the per-lane byte offset of a halo
// pixel that lies outside the image is set beyond num_records, for which the hardware writes zeros to LDS
// (verified by tools/probes/buffer_lds_oob.hip) -- SAME padding costs nothing; the chunk's channel offset and
// the weight-slab offset travel in the scalar offset, so issuing a copy is one SALU add + one VMEM instruction.
//
// UPS (SA_SRC1_UPSAMPLE2X on 16-channel planes): src1 is stored at HALF resolution and enters the convolution through
// UpSampling2D(2, bilinear) -- the decoder's Concatenate([skip, upsampled]) without the upsampled tensor ever existing in
// HBM (it is 4x the size of its source: a quarter of the plan's activation bytes, and 0.4 ms of bandwidth-bound upsampling
// kernels per 64-frame step). A src1 chunk is copied as the LOW-resolution tile that the halo tile touches ((TH/2 + 2) x
// (TW/2 + 2) pixels x 32 bytes = 6 one-KiB pieces instead of 20, edge pixels clamped in the copy's addresses) into the TAIL of
// the stage's input area; when the chunk's turn comes the workgroup expands it in place: every thread reads the 2 x 2
// low-resolution pixels of one 2 x 2 block of halo-tile pixels (one 8-channel half), barrier, then writes the four interpolated
// pixels (fp32 arithmetic in the operation order of upsample2x_bilinear_c16_kernel, rounded once: the SAME bits the
// materialised tensor would hold; halo pixels outside the image are the convolution's zero padding) where the copy engine
// would have put them. Two extra barriers per src1 chunk, no extra LDS.
//
// ITAP: where in a chunk the copies of the NEXT chunk are queued. -1: right after the chunk's barrier -- the most time to land,
// what the HBM-bound few-chunk layers want. t >= 0: after tap t of the chunk's nine -- by then the wave has MFMAs in flight and
// its copy instructions (a few SALU + one VMEM each, ~60-180 cycles of issue apiece) go out under its own matrix-core work
// instead of in front of it. Measured on MI355X (profiles/r02_ab_session.md section 13): t = 2..4 takes 3-7 % off every layer with
// >= 8 chunks (512->512 @32: 0.196 -> 0.182 ms, 768->256 @64: 0.557 -> 0.525), t >= 6 leaves the copy too little time, and the
// 2-4 chunk 256 x 256 layers lose 3-5 % with any t > 0.
// SA_CONV_EXT_WG2: the extended-epilogue kernels on 16-channel chunks keep to 128 registers (two workgroups per CU) like the
// plain ones (0: 148 registers, one workgroup per CU -- A/B builds)
#if !defined(SA_UPS_DX_SWAP)
#define SA_UPS_DX_SWAP 1  // UPS expansion: see the chunk loop (0: every lane writes column parity 0 first -- A/B)
#endif
#if !defined(SA_CONV_EXT_WG2)
#define SA_CONV_EXT_WG2 1
#endif
// PERS (round 5): the persistent tile loop for the TWO-workgroups-per-CU kernels (MT <= 2, plain epilogue). Round 2 measured that
// loop 1-9 % SLOWER there because the wait in front of a tile's first chunk -- `s_waitcnt vmcnt(0)` -- also waited for the previous
// tile's epilogue STORES (gfx9 counts loads and stores in one counter). The counter retires IN ORDER, though (LLVM's gfx9 model
// treats VMEM loads and stores as one in-order event type and emits counted waits across them), and the next tile's first-chunk
// copies are queued BEFORE the epilogue's stores: `s_waitcnt vmcnt(S)` with S = the number of store instructions this wave issued
// after them waits for exactly the copies. What else would touch the counter at a tile boundary is moved off it: the bias comes
// from LDS (staged once per workgroup) instead of global loads at every tile's start.
template <int MT, int CK, int NW, int R, int NBUF, bool HEADS, int STEM_CIN, bool EXT, bool UPS, int ITAP, bool XP = false, bool PERS = false>
__global__ void __launch_bounds__(NW * 64, (NW == 8 && MT <= 2 && NBUF == 2 && (!EXT || SA_CONV_EXT_WG2) && CK == 16 && STEM_CIN == 0 && !XP) ? 4 : 1)
conv3x3_dma_kernel(const ConvParams2 p) {
#if defined(__HIP_DEVICE_COMPILE__)  // the body uses device-only types (buffer resources); the host pass only needs the stub
  constexpr int TH = NW * R, TW = 32, PH = TH + 2, PW = TW + 2;
  constexpr int PIXB = CK * 2;
  constexpr int N_IN = (PH * PW * PIXB + 1023) / 1024;
  constexpr int IN_BYTES = N_IN * 1024;
  constexpr int KK = CK / 16;
  constexpr int N_W = MT * KK * 9;
  constexpr int W_BYTES = N_W * 1024;
  constexpr int STAGE = IN_BYTES + W_BYTES;
  constexpr int IN_PER_WAVE = (N_IN + NW - 1) / NW;
  constexpr int W_PER_WAVE = (N_W + NW - 1) / NW;
  constexpr unsigned OOB = 0xFFFFFF00u;
  // UPS: the low-resolution tile under the halo tile (LH x LW pixels of 32 bytes = 360 sixteen-byte pieces) has a buffer of its
  // own, so that the tile of chunk c+1 can be expanded into the idle stage WHILE chunk c is being multiplied (round 3; the round-2
  // form copied it into the stage it belonged to and expanded it in place between two extra barriers: ~2.6 k cycles per
  // source chunk without an MFMA). Two workgroups per CU leave 80 KiB each = the two stages + 4 KiB, and the tile needs
  // 5760 bytes: pieces 0..255 live in those 4 KiB (segment A, behind the stages), pieces 256..311 in the unused 896-byte tail
  // of stage 0's input area (B) and pieces 312..359 in the tail of stage 1's (C).
  constexpr int LH = TH / 2 + 2, LW = TW / 2 + 2, LOW_PIECES = LH * LW * 2;
  constexpr int IN_USED = PH * PW * PIXB, IN_TAIL = IN_BYTES - IN_USED;
  constexpr int LOW_A = NBUF * STAGE, LOW_NA = 256, LOW_NB = IN_TAIL / 16, LOW_NC = LOW_PIECES - LOW_NA - LOW_NB;
  constexpr int LOW_B = IN_USED, LOW_C = STAGE + IN_USED;
  static_assert(!UPS || (CK == 16 && NBUF == 2 && STEM_CIN == 0 && !EXT && !HEADS && TH % 2 == 0 && NW >= 6),
                "UPS: 16-channel chunks, two stages, plain epilogue");
  static_assert(!UPS || (LOW_NB > 0 && LOW_NB <= 64 && LOW_NC > 0 && LOW_NC <= LOW_NB), "UPS: the low tile's three segments");
  static_assert(!UPS || ((PH / 2) * (PW / 2) * 2 <= NW * 64 && PH % 2 == 0 && PW % 2 == 0), "UPS: one 2x2 block and half per thread");
  static_assert(!PERS || (NBUF == 2 && STEM_CIN == 0 && !EXT && !HEADS && !UPS && !XP && CK == 16 && MT <= 2 && R == 2),
                "PERS: the plain double-buffered 16-channel-chunk kernels");
  // LDS byte address of piece q of the low-resolution tile
  auto low_addr = [&](int q) -> int {
    return q < LOW_NA ? LOW_A + q * 16 : (q < LOW_NA + LOW_NB ? LOW_B + (q - LOW_NA) * 16 : LOW_C + (q - LOW_NA - LOW_NB) * 16);
  };
  (void)low_addr;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  SA_STAMP_DECL;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H, W = p.W;
  const int CinP = p.C0P + p.C1P;
  const int K16 = CinP / 16;
  const int co32_n = (p.CoutP + 31) / 32;
  // Tile schedule. Logical tiles 0..n_tiles-1 are ordered (frame, tile row, tile column, cout tile) and split into 8
  // contiguous ranges, one per XCD (hardware places block i on XCD i % 8): the cout tiles / neighbouring spatial tiles that
  // re-read the same input share one L2. A workgroup is PERSISTENT when the host launches fewer workgroups than tiles: the
  // j-th workgroup of an XCD (of g8 there) walks tiles start + j, start + j + g8, ... of its XCD's range, and while it runs
  // the last chunk of one tile the first chunk of its next tile is already being copied into the idle LDS stage -- the
  // epilogue (convert / permute / stores) of tile t overlaps the HBM latency of tile t+1 instead of being followed by a
  // workgroup teardown, a launch, address set-up and an exposed first copy (measured as ~2.9 chunk-times per tile: 72 % on
  // top of a 4-chunk layer). With gridDim == n_tiles every workgroup has exactly one tile (g8 == the range length).
  const int n_tiles = p.co_tiles * p.tiles_x * p.tiles_y * p.B;
  int L, L_end, L_step;
  {
    const int q = n_tiles >> 3, r = n_tiles & 7, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    L_step = ((int)gridDim.x - xcd + 7) >> 3;
    L = start + k;
    L_end = start + q + (xcd < r ? 1 : 0);
  }
  if (L >= L_end) return;  // wave-uniform
  struct Tile {
    int co32_0, x0, y0, b;
  };
  auto decode = [&](int l) {
    Tile t;
    if (p.sh_co >= 0) {  // (wave-uniform) power-of-two tile counts: scalar shifts and masks only
      t.co32_0 = (l & (p.co_tiles - 1)) * MT;
      l >>= p.sh_co;
      t.x0 = (l & (p.tiles_x - 1)) * TW;
      l >>= p.sh_tx;
      t.y0 = (l & (p.tiles_y - 1)) * TH;
      t.b = l >> p.sh_ty;
      return t;
    }
    t.co32_0 = (l % p.co_tiles) * MT;
    l /= p.co_tiles;
    t.x0 = (l % p.tiles_x) * TW;
    l /= p.tiles_x;
    t.y0 = (l % p.tiles_y) * TH;
    t.b = l / p.tiles_y;
    return t;
  };
  Tile cur = decode(L);

  // ---- buffer descriptors (wave-uniform): one frame of each source, the packed weights
  const size_t f0 = (size_t)H * W * p.C0P * 2, f1 = UPS ? (size_t)(H / 2) * (W / 2) * p.C1P * 2 : (size_t)H * W * p.C1P * 2;
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.w, 0, (int)((size_t)co32_n * K16 * 9 * 1024), 0x00020000);

  // ---- per-lane byte offsets of this wave's input copies inside a frame (same for every chunk of a tile)
  // `ln` = the lane id; inside the tile loop it is passed through an opaque asm so that the compiler re-derives these few
  // per-lane values per tile instead of hoisting a dozen of them into registers that stay live across the MFMA loop
  auto make_voff = [&](const Tile& t, int ln, unsigned (&v0)[IN_PER_WAVE], unsigned (&v1)[IN_PER_WAVE]) {
#pragma unroll
    for (int j = 0; j < IN_PER_WAVE; ++j) {
      const int i = j * NW + wave;
      const int o = i * 1024 + ln * 16;
      const int pl = o / PIXB, s = (o % PIXB) / 16;
      const int ty = pl / PW, tx = pl - ty * PW;
      const int gy = t.y0 + ty - 1, gx = t.x0 + tx - 1;
      const bool ok = (i < N_IN) && (pl < PH * PW) && gy >= 0 && gy < H && gx >= 0 && gx < W;
      const unsigned pix = (unsigned)(gy * W + gx);
      const unsigned q16 = (unsigned)((s ^ swz<CK>(pl)) * 16);
      v0[j] = ok ? pix * (unsigned)p.pix_bytes0 + q16 : OOB;
      v1[j] = UPS ? 0u : (ok ? pix * (unsigned)p.pix_bytes1 + q16 : OOB);  // (UPS: src1 never arrives at full resolution)
    }
  };
  // UPS: byte offset of this lane's 16-byte piece of the low-resolution tile inside a src1 plane. Waves 0-3 copy pieces
  // 64 w + lane (segment A), wave 4 pieces 256 + lane (lane < LOW_NB: segment B), wave 5 pieces 256 + LOW_NB + lane (segment C).
  auto make_voff_low = [&](const Tile& t, int ln) -> unsigned {
    const int q = wave < 4 ? wave * 64 + ln : (wave == 4 ? LOW_NA + ln : LOW_NA + LOW_NB + ln);
    const bool mine = wave < 4 || (wave == 4 && ln < LOW_NB) || (wave == 5 && ln < LOW_NC);
    const int lp = q >> 1, hf = q & 1;
    const int ly = lp / LW, lx = lp - ly * LW;
    const int gy = min(max((t.y0 >> 1) - 1 + ly, 0), (H >> 1) - 1), gx = min(max((t.x0 >> 1) - 1 + lx, 0), (W >> 1) - 1);
    return mine ? (unsigned)(gy * (W >> 1) + gx) * (unsigned)p.pix_bytes1 + (unsigned)hf * 16u : OOB;
  };
  unsigned voff0[IN_PER_WAVE], voff1[IN_PER_WAVE];
  make_voff(cur, lane, voff0, voff1);
  unsigned voff_low = 0;
  if constexpr (UPS) voff_low = make_voff_low(cur, lane);
  const unsigned wv = (unsigned)lane * 16;

  // this wave's pieces lo <= q < hi of the chunk (inputs are pieces 0 .. IN_PER_WAVE-1, the weights follow)
  auto issue = [&](const Tile& t, const unsigned (&v0)[IN_PER_WAVE], const unsigned (&v1)[IN_PER_WAVE], int chunk, int buf,
                   int lo = 0, int hi = 99) {
    const int c_lo = chunk * CK;
    const bool from1 = c_lo >= p.C0P;
    // byte offset of the chunk's first channel inside a pixel record (NHWC: 2 bytes per channel) or of its plane (planes:
    // H W 32 bytes per 16 channels); 32-bit unsigned arithmetic, a frame is < 4 GiB
    const int cc2 = (int)((unsigned)((from1 ? c_lo - p.C0P : c_lo) >> 4) * ((UPS && from1) ? p.blk_bytes_in1 : p.blk_bytes_in));
    unsigned char* stage = smem + buf * STAGE;
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(STEM_CIN ? reinterpret_cast<const unsigned char*>(p.w) : reinterpret_cast<const unsigned char*>(p.src0) + t.b * f0), 0,
        (int)f0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.C1P ? reinterpret_cast<const unsigned char*>(p.src1) + t.b * f1 : reinterpret_cast<const unsigned char*>(p.w)), 0,
        (int)f1, 0x00020000);
    if (UPS && from1) {
      // (wave-uniform) the input of this chunk is expanded from the low-resolution tile (issue_low / the chunk loop)
    } else {
#pragma unroll
      for (int j = 0; j < IN_PER_WAVE; ++j) {
        const int i = j * NW + wave;
        if (STEM_CIN == 0 && i < N_IN && lo <= j && j < hi) {
          if (!UPS && from1) {
            if (p.nt_in)  // (wave-uniform; the cache policy is an immediate of the instruction)
              __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lds_ptr_t)(stage + i * 1024), 16, v1[j], cc2, 0, 2);
            else
              __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lds_ptr_t)(stage + i * 1024), 16, v1[j], cc2, 0, 0);
          } else if (UPS && IN_TAIL > 0 && i == N_IN - 1) {
            // the tail of the input area belongs to the low-resolution tile: the last piece's lanes beyond the halo tile
            // stay switched off (an out-of-range offset would write zeros there)
            if (lane * 16 < IN_USED - (N_IN - 1) * 1024)
              __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lds_ptr_t)(stage + i * 1024), 16, v0[j], cc2, 0, 0);
          } else if (p.nt_in) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lds_ptr_t)(stage + i * 1024), 16, v0[j], cc2, 0, 2);
          } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lds_ptr_t)(stage + i * 1024), 16, v0[j], cc2, 0, 0);
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < W_PER_WAVE; ++j) {
      const int k = j * NW + wave;  // (m, kk, tap) slab index
      if (k < N_W && lo <= IN_PER_WAVE + j && IN_PER_WAVE + j < hi) {
        const int m = k / (KK * 9), rest = k - m * (KK * 9);
        // cout tiles beyond CoutP read out of range -> zeros
        const int soff = (t.co32_0 + m < co32_n) ? (((t.co32_0 + m) * K16 + chunk * KK) * 9 + rest) * 1024 : (int)0x7FFFF000;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)(stage + IN_BYTES + k * 1024), 16, wv, soff, 0, 0);
      }
    }
  };

  // Multi-chunk kernels (NBUF == 2) start the accumulators at the bias (the loads hide behind the first copy, the
  // epilogue saves one add per value); single-chunk kernels are latency-bound on their prologue and add it at the end.
  constexpr bool BIAS_INIT = (NBUF == 2);
  float* lds_bias = reinterpret_cast<float*>(smem + NBUF * STAGE);  // PERS only
  const unsigned lds_bias_addr = (unsigned)(uintptr_t)(lds_ptr_t)(smem + NBUF * STAGE);
  (void)lds_bias, (void)lds_bias_addr;
  f32x16 acc[MT][R];
  auto init_acc = [&](const Tile& t) {
    int ln_i = lane;
    asm volatile("" : "+v"(ln_i));
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 bq = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if constexpr (BIAS_INIT) {
          const int co = (t.co32_0 + m) * 32 + 8 * g + 4 * (ln_i >> 5);
          if constexpr (PERS) {
            // (an opaque ds_read: in front of a C++ load from LDS the compiler waits for EVERY outstanding VMEM operation --
            // it cannot tell the read from the copies' LDS writes -- which would put the epilogue's stores back on the path)
            if (t.co32_0 + m < co32_n) {  // (zero beyond CoutP)
              typedef float f32x4v __attribute__((ext_vector_type(4)));
              f32x4v v;
              const unsigned a = lds_bias_addr + (unsigned)co * 4u;
              asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
              bq = make_float4(v[0], v[1], v[2], v[3]);
            }
          } else {
            if (co < p.CoutP) bq = *reinterpret_cast<const float4*>(p.bias + co);
          }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
          acc[m][r][4 * g + 0] = bq.x;
          acc[m][r][4 * g + 1] = bq.y;
          acc[m][r][4 * g + 2] = bq.z;
          acc[m][r][4 * g + 3] = bq.w;
        }
      }
    }
  };

  const int n_chunks = CinP / CK;
  // PERS: the bias of EVERY output channel -> LDS behind the stages (<= 2 KiB), loaded before the first copy is queued so that the
  // wait for these loads waits for nothing else; the barrier below orders the writes before init_acc's reads
  if constexpr (PERS) {
    for (int i = tid; i < co32_n * 32; i += NW * 64) lds_bias[i] = i < p.CoutP ? p.bias[i] : 0.0f;
  }
  issue(cur, voff0, voff1, 0, 0);
  if constexpr (PERS) __syncthreads();
  if constexpr (XP) {
    // the expand / reduce weight fragments -> LDS behind the stages: CoutX * 128 bytes each = CoutX / 8 one-KiB pieces, wave w
    // takes pieces w, w + NW, ...; they land under the K loop (every chunk waits for vmcnt(0) and meets at a barrier)
    const int n_pc = p.CoutX >> 3;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)p.xp_w, 0, n_pc * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.rd_w ? p.rd_w : p.xp_w), 0, n_pc * 1024, 0x00020000);
    for (int i = wave; i < n_pc; i += NW) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr_t)(smem + NBUF * STAGE + i * 1024), 16, wv, i * 1024, 0, 0);
      if (p.rd_w) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsr, (lds_ptr_t)(smem + NBUF * STAGE + (n_pc + i) * 1024), 16, wv, i * 1024, 0, 0);
    }
  }
  // UPS: the low-resolution tile of source chunk `chunk` (>= C0P / CK) -> its own buffer, six wave-instructions
  auto issue_low = [&](const Tile& t, int chunk) {
    if constexpr (UPS) {
      const int cc2 = (int)((unsigned)((chunk * CK - p.C0P) >> 4) * p.blk_bytes_in1);
      const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(reinterpret_cast<const unsigned char*>(p.src1) + t.b * f1), 0, (int)f1, 0x00020000);
      if (wave < 4) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lds_ptr_t)(smem + LOW_A + wave * 1024), 16, voff_low, cc2, 0, 0);
      } else if (wave == 4) {
        if (lane < LOW_NB) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lds_ptr_t)(smem + LOW_B), 16, voff_low, cc2, 0, 0);
      } else if (wave == 5) {
        if (lane < LOW_NC) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lds_ptr_t)(smem + LOW_C), 16, voff_low, cc2, 0, 0);
      }
    }
  };
  if constexpr (UPS) {
    if (n_chunks > 1 && CK >= p.C0P) issue_low(cur, 1);  // chunk 1 is a source chunk already: expanded during chunk 0
  }
  const int x0 = cur.x0, y0 = cur.y0, b = cur.b;  // the fused first layer (STEM_CIN > 0) is never persistent
  (void)x0, (void)y0, (void)b;
  if constexpr (STEM_CIN > 0) {
    // First conv on the raw image, written as bf16 straight into the swizzled LDS tile of the second conv.
    // Halo pixels outside the image are ZERO (the second conv's SAME padding pads the first conv's OUTPUT); the
    // raw tile itself is zero outside the image (the first conv's SAME padding).
    constexpr int RW = PW + 2, RH = PH + 2, KT = 9 * STEM_CIN;
    if (p.stem_is_u8) {
      // uint8 path on the matrix cores: pixel values 0..255 are exact in bf16 (B operand); the fp32 weights times
      // 1/255 (ensure_float, normalization.py:49) are split into three bf16 terms hi+mid+lo (A operands), which
      // carries their full 24-bit mantissa, so three MFMAs per 16 taps give fp32-accurate products with fp32
      // accumulation. K index = tap*CIN + c, zero padded to a multiple of 16.
      constexpr int NK16 = (KT + 15) / 16;
      uint16_t* rawh = reinterpret_cast<uint16_t*>(smem + NBUF * STAGE);
      for (int i = tid; i < RH * RW * STEM_CIN; i += NW * 64) {
        const int c = i % STEM_CIN, px = i / STEM_CIN;
        const int ty = px / RW, tx = px - ty * RW;
        const int gy = y0 + ty - 2, gx = x0 + tx - 2;
        float v = 0.0f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W)
          v = (float)reinterpret_cast<const uint8_t*>(p.stem_src)[(((size_t)b * H + gy) * W + gx) * STEM_CIN + c];
        rawh[i] = sa::f2h(v * sa::U8_ACT_SCALE);
      }
      const int hf = lane >> 5, l32 = lane & 31;
      mfma_h8 wa[NK16][3];
#pragma unroll
      for (int ks = 0; ks < NK16; ++ks) {
        h16x8_t t0, t1, t2;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = ks * 16 + hf * 8 + j;
          float wv = 0.0f;
          if (k < KT && l32 < CK) wv = p.stem_w[k * CK + l32] * (1.0f / 255.0f) * (1.0f / sa::U8_ACT_SCALE);
          const uint16_t h0 = sa::f2h(wv);
          const float r1 = wv - sa::h2f(h0);
          const uint16_t h1 = sa::f2h(r1);
          const uint16_t h2 = sa::f2h(r1 - sa::h2f(h1));
          t0[j] = h0;
          t1[j] = h1;
          t2[j] = h2;
        }
        wa[ks][0] = __builtin_bit_cast(mfma_h8, t0);
        wa[ks][1] = __builtin_bit_cast(mfma_h8, t1);
        wa[ks][2] = __builtin_bit_cast(mfma_h8, t2);
      }
      __syncthreads();
      constexpr int NG = (PH * PW + 31) / 32;
      for (int g = wave; g < NG; g += NW) {
        const int pl = g * 32 + l32;
        const bool valid = pl < PH * PW;
        const int plc = valid ? pl : 0;
        const int ty = plc / PW, tx = plc - ty * PW;
        f32x16 d;
#pragma unroll
        for (int i = 0; i < 16; ++i) d[i] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < NK16; ++ks) {
          h16x8_t bq;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int k = ks * 16 + hf * 8 + j;
            // k is lane dependent only through hf: resolve both halves at compile time
            const int k0 = ks * 16 + j, k1 = ks * 16 + 8 + j;
            uint16_t v0 = 0, v1 = 0;
            if (k0 < KT) v0 = rawh[((ty + (k0 / STEM_CIN) / 3) * RW + tx + (k0 / STEM_CIN) % 3) * STEM_CIN + k0 % STEM_CIN];
            if (k1 < KT) v1 = rawh[((ty + (k1 / STEM_CIN) / 3) * RW + tx + (k1 / STEM_CIN) % 3) * STEM_CIN + k1 % STEM_CIN];
            bq[j] = hf ? v1 : v0;
            (void)k;
          }
          const mfma_h8 bf = __builtin_bit_cast(mfma_h8, bq);
#pragma unroll
          for (int part = 0; part < 3; ++part) d = SA_MFMA_32x32x16(wa[ks][part], bf, d, 0, 0, 0);
        }
        const int gy = y0 + ty - 1, gx = x0 + tx - 1;
        const bool in_img = valid && gy >= 0 && gy < H && gx >= 0 && gx < W;
        // lane holds couts (reg&3) + 8*(reg>>2) + 4*hf of pixel pl; piece q (couts 8q..8q+7) = regs 4q..4q+3 of
        // the hf=0 lane followed by the same regs of the hf=1 lane -> one exchange with the partner lane per pair
        // of pieces, then every lane stores one 16-byte piece per pair.
#pragma unroll
        for (int qp = 0; qp < CK / 16; ++qp) {
          unsigned pk[2][2];  // [piece in pair][dword]
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int q = 2 * qp + e;
            uint16_t h4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float t = d[4 * q + j] + p.stem_b[8 * q + 4 * hf + j];
              if (p.stem_relu) t = fmaxf(t, 0.0f);
              h4[j] = in_img ? sa::f2h(t) : (uint16_t)0;
            }
            pk[e][0] = (unsigned)h4[0] | ((unsigned)h4[1] << 16);
            pk[e][1] = (unsigned)h4[2] | ((unsigned)h4[3] << 16);
          }
          // hf=0 keeps piece 2qp (sends its half of piece 2qp+1); hf=1 keeps piece 2qp+1
          const unsigned s0 = hf ? pk[0][0] : pk[1][0], s1 = hf ? pk[0][1] : pk[1][1];
          const unsigned r0 = __shfl_xor(s0, 32), r1 = __shfl_xor(s1, 32);
          uint4 piece;
          if (hf) {
            piece = make_uint4(r0, r1, pk[1][0], pk[1][1]);
          } else {
            piece = make_uint4(pk[0][0], pk[0][1], r0, r1);
          }
          const int q = 2 * qp + hf;
          if (valid) *reinterpret_cast<uint4*>(smem + pl * PIXB + (q ^ swz<CK>(pl)) * 16) = piece;
        }
      }
    } else {
      // float32 images: VALU path, same fp32 FMA order as stem_conv3x3_kernel (bit-identical activations)
      float* raw = reinterpret_cast<float*>(smem + NBUF * STAGE);
      float* w0 = raw + RH * RW * STEM_CIN;  // [9*CIN][CK] then bias [CK]
      for (int i = tid; i < RH * RW * STEM_CIN; i += NW * 64) {
        const int c = i % STEM_CIN, px = i / STEM_CIN;
        const int ty = px / RW, tx = px - ty * RW;
        const int gy = y0 + ty - 2, gx = x0 + tx - 2;
        float v = 0.0f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W)
          v = reinterpret_cast<const float*>(p.stem_src)[(((size_t)b * H + gy) * W + gx) * STEM_CIN + c];
        raw[i] = v;
      }
      for (int i = tid; i < (KT + 1) * CK; i += NW * 64) w0[i] = (i < KT * CK) ? p.stem_w[i] : p.stem_b[i - KT * CK];
      __syncthreads();
      constexpr int PIECES = CK / 8;
      for (int it = tid; it < PH * PW * PIECES; it += NW * 64) {
        const int pl = it / PIECES, q = it - pl * PIECES;
        const int ty = pl / PW, tx = pl - ty * PW;
        const int gy = y0 + ty - 1, gx = x0 + tx - 1;
        h16x8_t o = zero8();
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
          float a8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) a8[j] = w0[KT * CK + q * 8 + j];
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
              for (int c = 0; c < STEM_CIN; ++c) {
                const float v = raw[((ty + dy) * RW + tx + dx) * STEM_CIN + c];
                const float* wr = w0 + ((dy * 3 + dx) * STEM_CIN + c) * CK + q * 8;
#pragma unroll
                for (int j = 0; j < 8; ++j) a8[j] = fmaf(v, wr[j], a8[j]);
              }
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = sa::f2h(p.stem_relu ? fmaxf(a8[j], 0.0f) : a8[j]);
        }
        *reinterpret_cast<h16x8_t*>(smem + pl * PIXB + (q ^ swz<CK>(pl)) * 16) = o;
      }
    }
  }
  int buf = 0;
  int stores_behind = 0;  // PERS: store instructions this wave certainly issued AFTER the copies that are in flight (wave-uniform)
  (void)stores_behind;
  // LDS byte offsets (inside a stage) of this lane's B fragments for k-step 0: halo row wave*R + (r + dy) in 0..R+1,
  // column lane&31 + dx; one register each (left to the compiler, pixel offset and swizzled slot are kept apart: 2 x 12
  // registers, which with the tile loop around everything spills into the MFMA loop). k-step kk flips bit 1 of the slot:
  // (2 kk + half) ^ swz == (half ^ swz) ^ 2 kk, i.e. the byte offset ^ 32 kk (pixel offsets are multiples of 2 CK >= 64 there).
  unsigned boff[R + 2][3];
  {
    const int half_m = lane >> 5, lx_m = lane & 31;
#pragma unroll
    for (int rr = 0; rr < R + 2; ++rr)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int pl = (wave * R + rr) * PW + lx_m + dx;
        boff[rr][dx] = (unsigned)(pl * PIXB + ((half_m ^ swz<CK>(pl)) * 16));
        asm volatile("" : "+v"(boff[rr][dx]));  // keep it ONE register (the optimiser re-splits the sum otherwise)
      }
  }
#pragma clang loop unroll(disable)
  for (;;) {  // ---- tiles of this workgroup
  init_acc(cur);
  const int L_next = L + L_step;
  // Persistent variants: only where ONE workgroup fits per CU (MT = 4: the 128-channel fused-head layer). Measured on MI355X
  // (profiles/r02_persistent_ab.md): there the cross-tile prefetch removes an exposed launch + first-copy latency per tile
  // (0.416 -> 0.393 ms); with two workgroups per CU the other workgroup already covers that gap, and the persistent loop
  // only adds a wait for the epilogue's STORES (vmcnt counts them together with the copies) in front of every tile: 2-9 %
  // slower on every MT <= 2 layer, most on the HBM-bound 256x256 ones. Those keep one tile per workgroup.
  constexpr bool PERSIST = (NBUF == 2) && (STEM_CIN == 0) && !EXT && (CK == 16) && (MT == 4 || PERS);
  const bool more = PERSIST && L_next < L_end;  // wave-uniform
  Tile nxt = cur;
#pragma clang loop unroll(disable)
  for (int chunk = 0; chunk < n_chunks; ++chunk) {
#if defined(SA_CONV_STAMP)
    if (chunk == 0) SA_STAMP(st_pro);
    st_n += 1;
#endif
    if constexpr (PERS) {
      // first chunk of a later tile: its copies were queued before the previous tile's `stores_behind` store instructions --
      // wait for the copies, not for the stores (in-order counter; a smaller immediate only waits for a few stores as well)
      if (chunk == 0 && stores_behind >= 4) {  // wave-uniform
        if (stores_behind >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (stores_behind >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (stores_behind >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    SA_STAMP(st_vm);
    __syncthreads();
    SA_STAMP(st_bar);
    // the copies of the next chunk: queued right after the barrier (ITAP < 0) or after tap ITAP of the MFMA sequence below
    auto prefetch = [&](int lo, int hi) {
      if constexpr (NBUF == 2) {
        if (chunk + 1 < n_chunks) {
          issue(cur, voff0, voff1, chunk + 1, buf ^ 1, lo, hi);
        } else if (more && !PERS) {  // cross-tile prefetch: chunk 0 of the next tile lands while this tile's epilogue runs
          if (lo <= 0) {
            nxt = decode(L_next);
            int ln = lane;
            asm volatile("" : "+v"(ln));
            make_voff(nxt, ln, voff0, voff1);  // this tile issues no more copies: its offsets are dead
          }
          issue(nxt, voff0, voff1, 0, buf ^ 1, lo, hi);
        }
      }
    };
    // inputs (HBM) after tap ISSUE_TAP, weights (L2) after tap W_TAP; -1 = right after the barrier
    constexpr int ISSUE_TAP = (NBUF == 2) ? ITAP : -1;
    // (the weights' copies at a tap of their own -- earlier or later than the inputs' -- measured within noise of this on
    // few- and many-chunk layers alike, gpurun_out/r02v)
    constexpr int W_TAP = ISSUE_TAP;
    if constexpr (ISSUE_TAP < 0) prefetch(0, W_TAP < 0 ? 99 : IN_PER_WAVE);
    if constexpr (ISSUE_TAP >= 0 && W_TAP < 0) prefetch(IN_PER_WAVE, 99);
    const unsigned char* in_tile = smem + buf * STAGE;
    const unsigned char* w_tile = in_tile + IN_BYTES;
    if constexpr (UPS) {
      // Source chunk c+1 is expanded NOW, from the low-resolution tile that landed before this chunk's barrier, into the idle
      // stage -- no barrier of its own: a wave goes on to its MFMAs of chunk c as soon as its share is written, and the waves
      // without a share (5-7) start at once, so the matrix pipe is never idle workgroup-wide. The expanded tile is read after
      // the next chunk's barrier.
      if (chunk + 1 < n_chunks && (chunk + 1) * CK >= p.C0P) {  // wave-uniform
        unsigned char* tile = smem + (buf ^ 1) * STAGE;
        // halo-tile rows 2rp, 2rp+1 are image rows y0-1+2rp (odd: weight 0.25 on the lower source row) and y0+2rp (even: 0.75);
        // both read low-resolution tile rows rp, rp+1; columns alike. (Source addresses are re-derived per chunk: the kernel
        // has no registers left to keep them.)
        int la[4];          // LDS addresses of the four source pieces (this thread's half of the low-resolution pixels)
        unsigned woff[4];   // (swizzled) offsets of the four pixels of the 2 x 2 block in the stage
        unsigned keepbits = 0;
        bool actv;
        // SA_UPS_DX_SWAP (round 6): which of its block's two columns a lane writes first alternates with bit 1 of the block index.
        // The eight lanes of a ds_write_b128 group are four blocks x two halves; with every lane on the SAME column parity their
        // pixels are 64 bytes apart and fall on two of the four 32-byte bank groups (2-way: the one source of LDS conflicts in
        // this kernel, 10 % of its LDS-active cycles); alternating, they cover all four.
        int sw;
        // tiles whose whole halo lies inside the image (all but the border ring) skip the padding masks (wave-uniform)
        const bool interior = __builtin_amdgcn_readfirstlane((int)(cur.x0 >= 1 && cur.y0 >= 1 && cur.x0 + PW - 1 <= W && cur.y0 + PH - 1 <= H));
        {
          int t_id = tid;
          asm volatile("" : "+v"(t_id));  // (derived per chunk, not kept in registers across the MFMA loop)
          constexpr int NBX = PW / 2, NBLK = (PH / 2) * NBX;  // 2x2 blocks of the halo tile: 9 rows of 17
          const int u = t_id < NBLK * 2 ? t_id : 0, hf = u & 1, blk = u >> 1;
          const int rp = blk / NBX, cp = blk - rp * NBX;
          sw = SA_UPS_DX_SWAP ? (blk >> 1) & 1 : 0;
          const int q0 = (rp * LW + cp) * 2 + hf;
          la[0] = low_addr(q0), la[1] = low_addr(q0 + 2), la[2] = low_addr(q0 + 2 * LW), la[3] = low_addr(q0 + 2 * LW + 2);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int ty = 2 * rp + (k >> 1), tx = 2 * cp + ((k & 1) ^ sw), pl = ty * PW + tx;
            woff[k] = (unsigned)(pl * PIXB + ((hf ^ swz<CK>(pl)) * 16));
            if (!interior) {
              const int gy = cur.y0 - 1 + ty, gx = cur.x0 - 1 + tx;
              if (gy >= 0 && gy < H && gx >= 0 && gx < W) keepbits |= 1u << k;  // outside the image: the conv's SAME padding
            }
          }
          actv = t_id < NBLK * 2;
        }
        // (memcpy, not __builtin_bit_cast: bit-casting an ELEMENT of an ext_vector to a half pair miscompiles with this
        // hipcc -- every element reads element 0; found in the ISA, reproduced in a ten-line kernel)
#if defined(SA_HALF_FP16) && !defined(SA_UPS_FP32)
        // Packed fp16 interpolation (round 3): the four source pieces stay packed pairs, every lerp is a v_pk_add_f16 +
        // v_pk_fma_f16 on two channels -- 12 packed operations per output dword quad instead of 8 conversions in, 24 fp32
        // operations and 4 conversions out. The horizontal blend is rounded to fp16 before the vertical one (the stand-alone
        // kernel rounds once, from fp32): values differ from the materialised tensor by <= 1.5 fp16 ulp, the size of the
        // storage rounding itself.
        typedef _Float16 hp2 __attribute__((ext_vector_type(2)));
        const hp2 q25 = {(_Float16)0.25f, (_Float16)0.25f}, q75 = {(_Float16)0.75f, (_Float16)0.75f};
        auto as_h2 = [](unsigned u) {
          hp2 h;
          __builtin_memcpy(&h, &u, 4);
          return h;
        };
        auto as_u = [](hp2 h) {
          unsigned u;
          __builtin_memcpy(&u, &h, 4);
          return u;
        };
        auto blend = [&](int dx, unsigned ua, unsigned ub, unsigned uc, unsigned ud, unsigned (&r)[2]) {
          const hp2 a = as_h2(ua), b_ = as_h2(ub), cc = as_h2(uc), d = as_h2(ud);
          const hp2 wx = (dx ^ sw) ? q75 : q25;                      // odd / even image column
          const hp2 t = a + (b_ - a) * wx, u_ = cc + (d - cc) * wx;  // source rows rp, rp + 1 at this column
          const hp2 dv = u_ - t;
          r[0] = as_u(t + dv * q25), r[1] = as_u(t + dv * q75);
        };
#else
        // fp32 interpolation in the order of the stand-alone kernel, one rounding: bitwise the materialised tensor
        auto blend = [&](int dx, unsigned ua, unsigned ub, unsigned uc, unsigned ud, unsigned (&r)[2]) {
          float v[2][2];
          const float wx = (dx ^ sw) ? 0.75f : 0.25f;
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const float a = sa::h2f((uint16_t)(ua >> (16 * e))), b_ = sa::h2f((uint16_t)(ub >> (16 * e)));
            const float cc = sa::h2f((uint16_t)(uc >> (16 * e))), d = sa::h2f((uint16_t)(ud >> (16 * e)));
            const float t = a + (b_ - a) * wx, u_ = cc + (d - cc) * wx;
            v[0][e] = t + (u_ - t) * 0.25f;
            v[1][e] = t + (u_ - t) * 0.75f;
          }
          r[0] = sa::f2h2(v[0][0], v[0][1]), r[1] = sa::f2h2(v[1][0], v[1][1]);
        };
#endif
        // Whole 16-byte pieces in and out: ds_read_b128 of the four source pieces (16 consecutive lanes read 256 contiguous
        // bytes: conflict-free) and ONE ds_write_b128 per output pixel (the 8 lanes of a write group cover 4 blocks: 2-way on
        // 32 banks). Measured on the way here (profiles/r03_ab_session.md): four-byte accesses (8-way conflicts on the writes)
        // cost 0.27 ms per decoder layer even though no barrier surrounds the expansion any more -- the LDS, not the VALU or
        // the barriers, is what the expansion competes with the MFMA loop for.
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4v;
        const u32x4v xa = *reinterpret_cast<const u32x4v*>(smem + la[0]), xb = *reinterpret_cast<const u32x4v*>(smem + la[1]);
        const u32x4v xc = *reinterpret_cast<const u32x4v*>(smem + la[2]), xd = *reinterpret_cast<const u32x4v*>(smem + la[3]);
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {  // the two output columns in turn: they share nothing but the sources
          u32x4v top, bot;                // output rows 2rp (dy = 0) and 2rp + 1 (dy = 1) at this column
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            unsigned r[2];
            blend(dx, xa[i], xb[i], xc[i], xd[i], r);
            top[i] = r[0];
            bot[i] = r[1];
          }
          if (!interior) {  // (wave-uniform) border tiles: pixels outside the image are the convolution's SAME padding
            const unsigned m0 = (unsigned)(((int)(keepbits << (31 - dx))) >> 31), m1 = (unsigned)(((int)(keepbits << (29 - dx))) >> 31);
            top &= m0;
            bot &= m1;
          }
          if (actv) {
            *reinterpret_cast<u32x4v*>(tile + woff[dx]) = top;
            *reinterpret_cast<u32x4v*>(tile + woff[2 + dx]) = bot;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap % 3;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        mfma_h8 a[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
          a[m] = *reinterpret_cast<const mfma_h8*>(w_tile + ((m * KK + kk) * 9 + tap) * 1024 + lane * 16);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const mfma_h8 bv = *reinterpret_cast<const mfma_h8*>(in_tile + (boff[r + dy][dx] ^ (unsigned)(kk * 32)));
#pragma unroll
          for (int m = 0; m < MT; ++m)
            acc[m][r] = SA_MFMA_32x32x16(a[m], bv, acc[m][r], 0, 0, 0);
        }
      }
#if defined(SA_CONV_SGB)
      // Explicit software pipeline of a scheduling region (A/B build): the operand fragments of tap t + 1 are read from LDS before
      // the MFMAs of tap t are issued -- two fragment sets in flight, whatever the register-pressure heuristic thinks
      if constexpr ((SA_CONV_SGB == 1 && PERS) || SA_CONV_SGB == 2 || (SA_CONV_SGB == 3 && !HEADS)) {
        constexpr int NR_ = (MT + R) * KK, NM_ = MT * R * KK;
        // (`tap` is a constant after unrolling, not a constant expression)
        const int first_ = (ISSUE_TAP >= 0 && tap > ISSUE_TAP) ? ISSUE_TAP + 1 : 0;
        const int last_ = (ISSUE_TAP >= 0 && tap <= ISSUE_TAP) ? ISSUE_TAP : 8;
        if (tap == last_) {
          __builtin_amdgcn_sched_group_barrier(0x100, NR_, 0);
#pragma unroll
          for (int t_ = first_; t_ < last_; ++t_) {
            __builtin_amdgcn_sched_group_barrier(0x100, NR_, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NM_, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, NM_, 0);
        }
      }
#endif
      if constexpr (ISSUE_TAP >= 0 || W_TAP >= 0) {
        if (tap == ISSUE_TAP || tap == W_TAP) {
          __builtin_amdgcn_sched_barrier(0);
          if (tap == ISSUE_TAP) prefetch(0, W_TAP == ISSUE_TAP ? 99 : IN_PER_WAVE);
          if (tap == W_TAP && W_TAP != ISSUE_TAP) prefetch(IN_PER_WAVE, 99);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr (UPS) {
        // the low-resolution tile of chunk c+2 may replace the one expanded at the top of this chunk once EVERY wave has read
        // its share: a rendezvous in the middle of the MFMA sequence (no memory wait attached), then six copies
        if (tap == SA_UPS_LOW_TAP && chunk + 2 < n_chunks && (chunk + 2) * CK >= p.C0P) {  // wave-uniform
          __builtin_amdgcn_sched_barrier(0);
          if ((chunk + 1) * CK >= p.C0P) __builtin_amdgcn_s_barrier();
          issue_low(cur, chunk + 2);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    SA_STAMP(st_body);
    if (NBUF == 2) {
      buf ^= 1;
    } else if (chunk + 1 < n_chunks) {  // single stage: refill after everyone finished reading it
      __syncthreads();
      issue(cur, voff0, voff1, chunk + 1, 0);
    }
  }
  if constexpr (PERS) {
    // Cross-tile prefetch of the two-workgroup kernels: queued HERE, between the last chunk's MFMAs and the epilogue, not inside
    // the chunk loop (round 5, first form: the branch in the middle of the last chunk cost the hot loop its register room -- 4
    // operand fragments in flight instead of 10, an `s_waitcnt lgkmcnt(0)` in front of every MFMA pair, every layer ~10 % slower
    // than one workgroup per tile). `buf` is the stage of chunk n-2 by now: every wave left it before the last chunk's barrier.
    // The copies are older than the epilogue's stores, which is what the counted wait of the next tile's first chunk needs.
    if (more) {
      nxt = decode(L_next);
      int ln = lane;
      asm volatile("" : "+v"(ln));
      make_voff(nxt, ln, voff0, voff1);  // this tile issues no more copies: its offsets are dead
      issue(nxt, voff0, voff1, 0, buf);
    }
  }
  const int co32_0 = cur.co32_0, x0 = cur.x0, y0 = cur.y0, b = cur.b;
  // the epilogue's per-lane constants are re-derived per tile (opaque copy of the lane id) for the same reason as above
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  const int half = lane_e >> 5, lx = lane_e & 31;

  // ---- epilogue: bias + ReLU, bf16 pack. A lane holds channels {0-3, 8-11, 16-19, 24-27} + 4*half of its pixel;
  // exchanging one 8-byte group with the partner lane (lane ^ 32) per pair of groups gives every lane 8 CONSECUTIVE
  // channels, i.e. one 16-byte store per pair instead of two 8-byte stores (fewer store instructions, full 64-byte
  // runs per pixel instead of interleaved partial writes).
  const int gx = x0 + lx;
  if constexpr (XP) {
    // ---- fused bottleneck tail: expand (+ BN + residual + ReLU) and the next block's reduce, see ConvParams2. One workgroup per
    // CU (the three stages' registers): the launch is HBM-bound -- 272 MFMAs per wave and tile against ~170 KB of tile traffic --
    // so what matters is that the residual loads and the weight fragments of co-tile t + 1 are in flight while t is computed.
    static_assert(!XP || (MT == 2 && EXT && !HEADS && !UPS && NBUF == 2 && CK == 16 && R == 2), "XP: the 64-channel extended kernel");
    const float lowv = p.relu ? 0.0f : -INFINITY;
    const bool colok = gx < W;
    // (1) h = this conv's activation (bias is the accumulators' initial value), as B fragments: k-step k = 2 m + s2
    mfma_h8 hb[4][R];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int c_lo = m * 32 + 16 * s2 + 4 * half;  // channels c_lo..c_lo+3 and c_lo+8..c_lo+11
        float sc[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f}, sh[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (p.post_scale) {
          const float4 a0 = *reinterpret_cast<const float4*>(p.post_scale + c_lo), a1 = *reinterpret_cast<const float4*>(p.post_scale + c_lo + 8);
          const float4 b0 = *reinterpret_cast<const float4*>(p.post_shift + c_lo), b1 = *reinterpret_cast<const float4*>(p.post_shift + c_lo + 8);
          sc[0] = a0.x, sc[1] = a0.y, sc[2] = a0.z, sc[3] = a0.w, sc[4] = a1.x, sc[5] = a1.y, sc[6] = a1.z, sc[7] = a1.w;
          sh[0] = b0.x, sh[1] = b0.y, sh[2] = b0.z, sh[3] = b0.w, sh[4] = b1.x, sh[5] = b1.y, sh[6] = b1.z, sh[7] = b1.w;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
          h16x8_t fq;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float t = fmaxf(acc[m][r][8 * s2 + j], lowv);  // the same operations as the extended epilogue's act()
            t = fmaf(t, sc[j], sh[j]);
            if (p.relu_last) t = fmaxf(t, 0.0f);
            fq[j] = sa::f2h(t);
          }
          hb[m * 2 + s2][r] = __builtin_bit_cast(mfma_h8, fq);
        }
      }
    // (2) per 32-channel tile t of y: expand GEMM (K = 64: 4 k-steps), epilogue, store; its rounded values feed the reduce GEMM
    const int n_t = p.CoutX >> 5, KX = p.CoutX >> 4;
    const bool has_rd = p.rd_w != nullptr;
    const unsigned ypix = (unsigned)p.xp_pix_bytes;
    const size_t yblk = p.planar ? (size_t)H * W * 32 : (size_t)32;  // bytes between 16-channel blocks of y
    unsigned char* yframe = reinterpret_cast<unsigned char*>(p.xp_dst) + (size_t)b * H * W * p.CoutX * 2;
    const unsigned char* rframe = p.xp_res ? reinterpret_cast<const unsigned char*>(p.xp_res) + (size_t)b * H * W * p.CoutX * 2 : nullptr;
    unsigned lane_off[R];
    bool ok[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int gy = y0 + wave * R + r;
      ok[r] = colok && gy < H;
      lane_off[r] = ok[r] ? (unsigned)(gy * W + gx) * ypix + (unsigned)half * 16u : 0u;
    }
    f32x16 a2[2][R];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) a2[m][r][i] = 0.0f;
    // the weight fragments of both stages sit in LDS behind the stages (copied by every workgroup at its start, under the 3x3
    // conv's K loop): expand [t][k = 0..3] at XW, reduce [m][k16] at XW + CoutX * 128. The residual pieces of co-tile t + 1 are
    // requested while t is computed (they come from HBM).
    const unsigned char* XW = smem + NBUF * STAGE;
    const unsigned char* RW = XW + (size_t)p.CoutX * 128;
    uint4 RQ[2][R];
    auto fetch = [&](int t, uint4 (&rq)[2][R]) {
#pragma unroll
      for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          rq[pr][r] = make_uint4(0u, 0u, 0u, 0u);
          if (rframe && ok[r]) rq[pr][r] = *reinterpret_cast<const uint4*>(rframe + (size_t)(2 * t + pr) * yblk + lane_off[r]);
        }
    };
    fetch(0, RQ);
    const float low1 = p.xp_relu ? 0.0f : -INFINITY, lowl1 = p.xp_relu_last ? 0.0f : -INFINITY;
#pragma clang loop unroll(disable)
    for (int t = 0; t < n_t; ++t) {
      uint4 nRQ[2][R];
      if (t + 1 < n_t) fetch(t + 1, nRQ);
      f32x16 a1[R];
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) a1[r][i] = 0.0f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const mfma_h8 a = *reinterpret_cast<const mfma_h8*>(XW + ((size_t)(t * 4 + k) * 64 + lane_e) * 16);
#pragma unroll
        for (int r = 0; r < R; ++r) a1[r] = SA_MFMA_32x32x16(a, hb[k][r], a1[r], 0, 0, 0);
      }
      // residual back into the accumulator layout (group g = channels 8 g + 4 half + 0..3), as tapconv_kernel does
      uint2 rq[4][R];
#pragma unroll
      for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          uint2 a = make_uint2(RQ[pr][r].x, RQ[pr][r].y), c = make_uint2(RQ[pr][r].z, RQ[pr][r].w);
          sa::swap32(a.x, c.x);
          sa::swap32(a.y, c.y);
          rq[2 * pr][r] = a;
          rq[2 * pr + 1][r] = c;
        }
      uint2 pk[R][4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = 32 * t + 8 * g + 4 * half;
        const float4 bq = *reinterpret_cast<const float4*>(p.xp_bias + co);
        float4 sq = make_float4(1.f, 1.f, 1.f, 1.f), tq = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.xp_scale) {
          sq = *reinterpret_cast<const float4*>(p.xp_scale + co);
          tq = *reinterpret_cast<const float4*>(p.xp_shift + co);
        }
        const float bb[4] = {bq.x, bq.y, bq.z, bq.w}, ss[4] = {sq.x, sq.y, sq.z, sq.w}, tt[4] = {tq.x, tq.y, tq.z, tq.w};
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const uint2 q = rq[g][r];
          const float rr[4] = {sa::h2f((uint16_t)(q.x & 0xffff)), sa::h2f((uint16_t)(q.x >> 16)), sa::h2f((uint16_t)(q.y & 0xffff)),
                               sa::h2f((uint16_t)(q.y >> 16))};
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {  // (the tap GEMM's epilogue, operation for operation)
            float u = fmaxf(a1[r][4 * g + j] + bb[j], low1);
            u = fmaf(u, ss[j], tt[j]);
            if (rframe) u += rr[j];
            v[j] = fmaxf(u, lowl1);
          }
          pk[r][g].x = sa::f2h2(v[0], v[1]);
          pk[r][g].y = sa::f2h2(v[2], v[3]);
        }
      }
      if (has_rd) {  // y as stored (rounded) is the reduce conv's input: k-steps 2 t, 2 t + 1
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const mfma_h8 yf = __builtin_bit_cast(mfma_h8, make_uint4(pk[r][2 * s2].x, pk[r][2 * s2].y, pk[r][2 * s2 + 1].x, pk[r][2 * s2 + 1].y));
#pragma unroll
            for (int m = 0; m < 2; ++m) {
              const mfma_h8 a = *reinterpret_cast<const mfma_h8*>(RW + ((size_t)(m * KX + 2 * t + s2) * 64 + lane_e) * 16);
              a2[m][r] = SA_MFMA_32x32x16(a, yf, a2[m][r], 0, 0, 0);
            }
          }
      }
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          uint2 a = pk[r][2 * pr], c = pk[r][2 * pr + 1];
          sa::swap32(a.x, c.x);
          sa::swap32(a.y, c.y);
          if (ok[r]) *reinterpret_cast<uint4*>(yframe + (size_t)(2 * t + pr) * yblk + lane_off[r]) = make_uint4(a.x, a.y, c.x, c.y);
        }
      if (t + 1 < n_t) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < R; ++j) RQ[i][j] = nRQ[i][j];
      }
    }
    // (3) z = the reduce conv's output (64 channels), in the layout / strides of this launch's 64-channel tensors
    if (has_rd) {
      const float low2 = p.rd_relu ? 0.0f : -INFINITY, lowl2 = p.rd_relu_last ? 0.0f : -INFINITY;
      unsigned char* zframe = reinterpret_cast<unsigned char*>(p.rd_dst) + (size_t)b * H * W * p.CoutR * 2;
      const unsigned zpix = (unsigned)p.out_pix_bytes;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        uint2 pk[R][4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int co = 32 * m + 8 * g + 4 * half;
          const float4 bq = *reinterpret_cast<const float4*>(p.rd_bias + co);
          float4 sq = make_float4(1.f, 1.f, 1.f, 1.f), tq = make_float4(0.f, 0.f, 0.f, 0.f);
          if (p.rd_scale) {
            sq = *reinterpret_cast<const float4*>(p.rd_scale + co);
            tq = *reinterpret_cast<const float4*>(p.rd_shift + co);
          }
          const float bb[4] = {bq.x, bq.y, bq.z, bq.w}, ss[4] = {sq.x, sq.y, sq.z, sq.w}, tt[4] = {tq.x, tq.y, tq.z, tq.w};
#pragma unroll
          for (int r = 0; r < R; ++r) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(fmaf(fmaxf(a2[m][r][4 * g + j] + bb[j], low2), ss[j], tt[j]), lowl2);
            pk[r][g].x = sa::f2h2(v[0], v[1]);
            pk[r][g].y = sa::f2h2(v[2], v[3]);
          }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int gy = y0 + wave * R + r;
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            uint2 a = pk[r][2 * pr], c = pk[r][2 * pr + 1];
            sa::swap32(a.x, c.x);
            sa::swap32(a.y, c.y);
            if (ok[r])
              *reinterpret_cast<uint4*>(zframe + (size_t)(2 * m + pr) * p.out_blk_bytes + (size_t)(gy * W + gx) * zpix + (unsigned)half * 16u) =
                  make_uint4(a.x, a.y, c.x, c.y);
          }
        }
      }
    }
  } else if constexpr (EXT) {
    // Extended epilogue, one PAIR of 4-channel groups (= one 16-byte store piece) at a time: post-scale, post-shift and the
    // residual of a whole cout tile live together cost 80 registers beside the accumulators (29 spilled at the 128 of two
    // workgroups per CU); per pair they cost 32. Same operations in the same order as before: same bits.
    const float lowv = p.relu ? 0.0f : -INFINITY;
    const bool colok = gx < W;
    const unsigned pixb = (unsigned)p.out_pix_bytes;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int cobase = (co32_0 + m) * 32;
      if (cobase >= p.CoutP) continue;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        float bb[2][4], ps[2][4], pt[2][4];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int co = cobase + 8 * (2 * pr + e) + 4 * half;
          float4 bq = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
          if (!BIAS_INIT && co < p.CoutP) bq = *reinterpret_cast<const float4*>(p.bias + co);
          float4 sq = make_float4(1.0f, 1.0f, 1.0f, 1.0f), tq = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
          if (p.post_scale && co < p.CoutP) {
            sq = *reinterpret_cast<const float4*>(p.post_scale + co);
            tq = *reinterpret_cast<const float4*>(p.post_shift + co);
          }
          bb[e][0] = bq.x, bb[e][1] = bq.y, bb[e][2] = bq.z, bb[e][3] = bq.w;
          ps[e][0] = sq.x, ps[e][1] = sq.y, ps[e][2] = sq.z, ps[e][3] = sq.w;
          pt[e][0] = tq.x, pt[e][1] = tq.y, pt[e][2] = tq.z, pt[e][3] = tq.w;
        }
        // the residual of this lane's outputs: four channels (8 bytes) per load, the R x 2 loads of the pair in flight together;
        // it has the output's layout (NHWC or 16-channel planes), at the output's resolution or at half of it (res_mode 1:
        // UpSampling2D(nearest) folded in)
        float rres[R][2][4];
        if (p.residual) {
          uint2 q[R][2];
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const int gy = y0 + wave * R + r;
            const int hr = p.res_mode ? H / 2 : H, wr = p.res_mode ? W / 2 : W;
            const size_t pix = p.res_mode ? (size_t)(gy >> 1) * wr + (gx >> 1) : (size_t)gy * wr + gx;
            const unsigned char* rf = reinterpret_cast<const unsigned char*>(p.residual) + (size_t)b * hr * wr * p.CoutP * 2 +
                                      pix * p.out_pix_bytes;
            const unsigned rblk = p.res_mode ? p.out_blk_bytes_pool : p.out_blk_bytes;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const int co = cobase + 8 * (2 * pr + e) + 4 * half;
              q[r][e] = make_uint2(0u, 0u);
              if (gy < H && gx < W && co < p.CoutP)
                q[r][e] = *reinterpret_cast<const uint2*>(rf + (size_t)(co >> 4) * rblk + (co & 15) * 2);
            }
          }
#pragma unroll
          for (int r = 0; r < R; ++r)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              rres[r][e][0] = sa::h2f((uint16_t)(q[r][e].x & 0xffff));
              rres[r][e][1] = sa::h2f((uint16_t)(q[r][e].x >> 16));
              rres[r][e][2] = sa::h2f((uint16_t)(q[r][e].y & 0xffff));
              rres[r][e][3] = sa::h2f((uint16_t)(q[r][e].y >> 16));
            }
        }
        auto act = [&](int r, int e, int j) {
          float t = acc[m][r][4 * (2 * pr + e) + j];
          if constexpr (!BIAS_INIT) t += bb[e][j];
          t = fmaxf(t, lowv);
          t = fmaf(t, ps[e][j], pt[e][j]);
          if (p.residual) t += rres[r][e][j];
          if (p.relu_last) t = fmaxf(t, 0.0f);
          return t;
        };
        // lower half-wave: own group 2pr (channels 0-3) + partner's (4-7); upper: partner's group 2pr+1 + own
        auto store_piece = [&](unsigned char* row_base, unsigned blk_bytes, unsigned lane_off, bool ok, uint2 a, uint2 c) {
          sa::swap32(a.x, c.x);
          sa::swap32(a.y, c.y);
          const int co16 = (cobase >> 4) + pr;  // 16-channel block of this piece (wave uniform)
          if (ok && co16 * 16 < p.CoutP)
            *reinterpret_cast<uint4*>(row_base + (size_t)co16 * blk_bytes + lane_off) = make_uint4(a.x, a.y, c.x, c.y);
        };
        if (p.dst) {
          unsigned char* frame = reinterpret_cast<unsigned char*>(p.dst) + (size_t)b * H * W * p.CoutP * 2;
          const unsigned lane_off = (unsigned)gx * pixb + (unsigned)half * 16u;
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const int gy = y0 + wave * R + r;  // wave uniform
            uint2 pk[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              pk[e].x = sa::f2h2(act(r, e, 0), act(r, e, 1));
              pk[e].y = sa::f2h2(act(r, e, 2), act(r, e, 3));
            }
            store_piece(frame + (size_t)gy * W * pixb, p.out_blk_bytes, lane_off, colok && gy < H, pk[0], pk[1]);
          }
        }
        if constexpr (R >= 2) if (p.dst_pool) {
          unsigned char* frame = reinterpret_cast<unsigned char*>(p.dst_pool) + (size_t)b * (H / 2) * (W / 2) * p.CoutP * 2;
          const unsigned lane_off = (unsigned)(gx >> 1) * pixb + (unsigned)half * 16u;
#pragma unroll
          for (int r = 0; r < R; r += 2) {
            const int gy = y0 + wave * R + r;  // wave uniform, even
            uint2 pk[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              float t4[4], t[4];
#pragma unroll
              for (int j = 0; j < 4; ++j) t[j] = fmaxf(act(r, e, j), act(r + 1, e, j));
              sa::max_xor1_x4(t, t4);
              pk[e].x = sa::f2h2(t4[0], t4[1]);
              pk[e].y = sa::f2h2(t4[2], t4[3]);
            }
            store_piece(frame + (size_t)(gy >> 1) * (W / 2) * pixb, p.out_blk_bytes_pool, lane_off, !(lane_e & 1) && colok && gy < H, pk[0], pk[1]);
          }
        }
      }
    }
  } else {
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int cobase = (co32_0 + m) * 32;
    if (cobase >= p.CoutP) continue;
    float bb[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int co = cobase + 8 * g + 4 * half;
      float4 bq = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (!BIAS_INIT && co < p.CoutP) bq = *reinterpret_cast<const float4*>(p.bias + co);
      bb[g][0] = bq.x;
      bb[g][1] = bq.y;
      bb[g][2] = bq.z;
      bb[g][3] = bq.w;
    }
    float ps[4][4], pt[4][4];
    if constexpr (EXT) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = cobase + 8 * g + 4 * half;
        float4 sq = make_float4(1.0f, 1.0f, 1.0f, 1.0f), tq = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (p.post_scale && co < p.CoutP) {
          sq = *reinterpret_cast<const float4*>(p.post_scale + co);
          tq = *reinterpret_cast<const float4*>(p.post_shift + co);
        }
        ps[g][0] = sq.x, ps[g][1] = sq.y, ps[g][2] = sq.z, ps[g][3] = sq.w;
        pt[g][0] = tq.x, pt[g][1] = tq.y, pt[g][2] = tq.z, pt[g][3] = tq.w;
      }
    }
    const float lowv = p.relu ? 0.0f : -INFINITY;  // ReLU as one v_max against a wave-uniform bound
    // EXT: the residual of this lane's outputs, four channels (8 bytes) per load, all R x 4 loads of the cout tile in flight
    // together (round 3: it was one 2-byte load per VALUE, each behind its own address arithmetic -- the residual 3x3 convs of
    // the hourglass ran at 0.79 PFLOP/s against 0.99 without a residual)
    float rres[EXT ? R : 1][4][4];
    if constexpr (EXT) {
      if (p.residual) {
        uint2 q[R][4];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int gy = y0 + wave * R + r;
          // the residual has the output's layout (NHWC or 16-channel planes: pixel stride / 16-channel-block stride), at the
          // output's resolution or at half of it (res_mode 1: UpSampling2D(nearest) folded in)
          const int hr = p.res_mode ? H / 2 : H, wr = p.res_mode ? W / 2 : W;
          const size_t pix = p.res_mode ? (size_t)(gy >> 1) * wr + (gx >> 1) : (size_t)gy * wr + gx;
          const unsigned char* rf = reinterpret_cast<const unsigned char*>(p.residual) + (size_t)b * hr * wr * p.CoutP * 2 +
                                    pix * p.out_pix_bytes;
          const unsigned rblk = p.res_mode ? p.out_blk_bytes_pool : p.out_blk_bytes;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int co = cobase + 8 * g + 4 * half;
            q[r][g] = make_uint2(0u, 0u);
            if (gy < H && gx < W && co < p.CoutP)
              q[r][g] = *reinterpret_cast<const uint2*>(rf + (size_t)(co >> 4) * rblk + (co & 15) * 2);
          }
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            rres[r][g][0] = sa::h2f((uint16_t)(q[r][g].x & 0xffff));
            rres[r][g][1] = sa::h2f((uint16_t)(q[r][g].x >> 16));
            rres[r][g][2] = sa::h2f((uint16_t)(q[r][g].y & 0xffff));
            rres[r][g][3] = sa::h2f((uint16_t)(q[r][g].y >> 16));
          }
      }
    }
    auto act = [&](int r, int g, int j) {
      float t = acc[m][r][4 * g + j];
      if constexpr (!BIAS_INIT) t += bb[g][j];  // (with the bias as the accumulators' initial value there is nothing to add:
      t = fmaxf(t, lowv);                       //  an `acc + 0.0f` is NOT dropped by the compiler -- 112 dead v_add per tile)
      if constexpr (EXT) {
        t = fmaf(t, ps[g][j], pt[g][j]);
        if (p.residual) t += rres[r][g][j];
        if (p.relu_last) t = fmaxf(t, 0.0f);
      }
      return t;
    };
    // Store addressing: everything that does not depend on the lane -- frame, 16-channel block, row of the tile -- goes into a
    // wave-uniform 64-bit base (scalar registers); the lane contributes ONE 32-bit byte offset per tile and output (its column,
    // its 8-channel half), so a store is `global_store_dwordx4 v_off, v_data, s[base]` with no vector address arithmetic (the
    // first version rebuilt a 64-bit address per store: ~100 VALU instructions per tile). A frame is < 4 GiB (checked on the host).
    // base(blk16, row) = frame + blk16 * blk_bytes + row * row_bytes;  NHWC: blk_bytes = 32, planes: npix * 32
    auto store_pieces = [&](unsigned char* row_base, unsigned blk_bytes, unsigned lane_off, bool ok, const uint2 (&pk)[4]) {
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        // lower half-wave: own group 2pr (channels 0-3) + partner's (4-7); upper: partner's group 2pr+1 + own
        uint2 a = pk[2 * pr], c = pk[2 * pr + 1];
        sa::swap32(a.x, c.x);
        sa::swap32(a.y, c.y);
        const int co16 = (cobase >> 4) + pr;  // 16-channel block of this piece (wave uniform)
        if (ok && co16 * 16 < p.CoutP)  // (non-temporal stores: measured, no effect -- profiles/r02_ab_session.md)
          *reinterpret_cast<uint4*>(row_base + (size_t)co16 * blk_bytes + lane_off) = make_uint4(a.x, a.y, c.x, c.y);
      }
    };
    const bool colok = gx < W;
    if (p.dst) {
      const unsigned pixb = (unsigned)p.out_pix_bytes;
      unsigned char* frame = reinterpret_cast<unsigned char*>(p.dst) + (size_t)b * H * W * p.CoutP * 2;
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
        store_pieces(frame + (size_t)gy * W * pixb, p.out_blk_bytes, lane_off, colok && gy < H, pk);
      }
    }
    if constexpr (R >= 2) if (p.dst_pool) {
      const unsigned pixb = (unsigned)p.out_pix_bytes;
      unsigned char* frame = reinterpret_cast<unsigned char*>(p.dst_pool) + (size_t)b * (H / 2) * (W / 2) * p.CoutP * 2;
      const unsigned lane_off = (unsigned)(gx >> 1) * pixb + (unsigned)half * 16u;
#pragma unroll
      for (int r = 0; r < R; r += 2) {
        const int gy = y0 + wave * R + r;  // wave uniform, even
        uint2 pk[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float t4[4], t[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) t[j] = fmaxf(act(r, g, j), act(r + 1, g, j));
          sa::max_xor1_x4(t, t4);
          pk[g].x = sa::f2h2(t4[0], t4[1]);
          pk[g].y = sa::f2h2(t4[2], t4[3]);
        }
        store_pieces(frame + (size_t)(gy >> 1) * (W / 2) * pixb, p.out_blk_bytes_pool, lane_off, !(lane_e & 1) && colok && gy < H, pk);
      }
    }
  }
  if constexpr (PERS) {
    // store instructions this wave has just issued, as a LOWER bound (a store whose lanes are all switched off may be branched
    // over): a tile that lies completely inside the image issues R (+ R / 2 pooled) stores per valid 16-channel piece
    int n_valid = 0;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int pr = 0; pr < 2; ++pr)
        if ((co32_0 + m) * 32 + pr * 16 < p.CoutP) ++n_valid;
    const bool full = x0 + TW <= W && y0 + TH <= H;
    stores_behind = full ? n_valid * ((p.dst ? R : 0) + (p.dst_pool ? R / 2 : 0)) : 0;
  }
  }

  // ---- fused 1x1 heads on the matrix cores: out[n, pixel] = act(b[n] + sum_co Wh[n][co] * f[co, pixel]) with
  // f = bf16(relu(conv + bias)), exactly the value the un-fused path stores and sa_conv1x1_head reads back.
  // This workgroup holds all CoutP (<= MT*32) channels of its pixels. GEMM view: A = head weights (rows n < 32),
  // B = features; the B fragment of k-step s of cout tile m is built straight from the accumulator registers
  // 8s..8s+7 of the lane (no cross-lane movement): lane half h, element j <-> channel 16s + 8*(j>>2) + 4h + (j&3) of
  // the tile, and the A fragment is gathered from the fp32 head weights with the same channel map. The weights enter
  // as hi + lo bf16 terms (two MFMAs), which keeps ~16 mantissa bits of the fp32 head kernel.
  if constexpr (HEADS) {
    for (int hd = 0; hd < p.n_heads; ++hd)
    for (int nb = 0; nb < p.head_c[hd]; nb += 32) {  // 32 head channels per pass (round 4: up to 64 -- 46 PAF channels of 23 edges)
      const int NH = p.head_c[hd];
      const int nrow = nb + (lane_e & 31);
      f32x16 hacc[R];
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) hacc[r][i] = 0.0f;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int cobase = (co32_0 + m) * 32;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int c_lo = cobase + 16 * s2 + 4 * half;  // channels c_lo..c_lo+3 and c_lo+8..c_lo+11
          if (cobase + 16 * s2 >= p.CoutP) continue;     // wave-uniform
          float4 w0 = make_float4(0.f, 0.f, 0.f, 0.f), w1 = w0;
          if (nrow < NH) {
            const float* wr = p.head_w[hd] + (size_t)nrow * p.CoutP + c_lo;
            w0 = *reinterpret_cast<const float4*>(wr);
            w1 = *reinterpret_cast<const float4*>(wr + 8);
          }
          const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
          float4 b0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), b1 = b0;
          if (!BIAS_INIT) {
            b0 = *reinterpret_cast<const float4*>(p.bias + c_lo);
            b1 = *reinterpret_cast<const float4*>(p.bias + c_lo + 8);
          }
          const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
          h16x8_t ahi, alo;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            ahi[j] = sa::f2h(wv[j]);
            alo[j] = sa::f2h(wv[j] - sa::h2f(ahi[j]));
          }
          // EXT: the heads see the extended epilogue's value (activation -> BatchNormalization affine -> final ReLU; no
          // residual with fused heads), operation for operation what act() above stores
          float sc[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f}, sh[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          if constexpr (EXT) {
            if (p.post_scale) {
              const float4 a0 = *reinterpret_cast<const float4*>(p.post_scale + c_lo), a1 = *reinterpret_cast<const float4*>(p.post_scale + c_lo + 8);
              const float4 t0 = *reinterpret_cast<const float4*>(p.post_shift + c_lo), t1 = *reinterpret_cast<const float4*>(p.post_shift + c_lo + 8);
              sc[0] = a0.x, sc[1] = a0.y, sc[2] = a0.z, sc[3] = a0.w, sc[4] = a1.x, sc[5] = a1.y, sc[6] = a1.z, sc[7] = a1.w;
              sh[0] = t0.x, sh[1] = t0.y, sh[2] = t0.z, sh[3] = t0.w, sh[4] = t1.x, sh[5] = t1.y, sh[6] = t1.z, sh[7] = t1.w;
            }
          }
#pragma unroll
          for (int r = 0; r < R; ++r) {
            h16x8_t fq;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              float t = acc[m][r][8 * s2 + j];
              if constexpr (!BIAS_INIT) t += bb[j];
              if constexpr (EXT) {
                t = fmaxf(t, p.relu ? 0.0f : -INFINITY);
                t = fmaf(t, sc[j], sh[j]);
                if (p.relu_last) t = fmaxf(t, 0.0f);
                fq[j] = sa::f2h(t);
              } else {
                fq[j] = sa::f2h(p.relu ? fmaxf(t, 0.0f) : t);
              }
            }
            const mfma_h8 bf = __builtin_bit_cast(mfma_h8, fq);
            hacc[r] = SA_MFMA_32x32x16(__builtin_bit_cast(mfma_h8, ahi), bf, hacc[r], 0, 0, 0);
            hacc[r] = SA_MFMA_32x32x16(__builtin_bit_cast(mfma_h8, alo), bf, hacc[r], 0, 0, 0);
          }
        }
      }
      // D layout: lane holds head channels (reg&3) + 8*(reg>>2) + 4*half of pixel lane&31, i.e. four runs of 4 CONSECUTIVE
      // channels: one 16-byte store per complete run (the rows of a [.., NH] f32 tensor are only 4-byte aligned for odd NH;
      // gfx950 global stores need dword alignment only), scalar stores for the ragged tail run. 13 channels: 2 stores per
      // lane instead of 8 / 5 -- the epilogue of the head layers was store-issue bound.
      float hb[4][4];
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = nb + j + 8 * g + 4 * half;
          hb[g][j] = n < NH ? p.head_b[hd][n] : 0.0f;
        }
      const bool sig = p.head_act[hd] == 1;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int gy = y0 + wave * R + r;
        const bool ok = gy < H && gx < W;
        float* out = p.head_dst[hd] + (((size_t)b * H + (ok ? gy : 0)) * W + (ok ? gx : 0)) * NH;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n0 = nb + 8 * g + 4 * half;
          float t[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            t[j] = hacc[r][4 * g + j] + hb[g][j];
            if (sig) t[j] = 1.0f / (1.0f + __expf(-t[j]));
          }
          if (!ok) continue;
          if (n0 + 4 <= NH) {
            *reinterpret_cast<sa::f32x4_unaligned*>(out + n0) = sa::f32x4_unaligned{t[0], t[1], t[2], t[3]};
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (n0 + j < NH) out[n0 + j] = t[j];
          }
        }
      }
    }
  }
  SA_STAMP(st_epi);
  // ---- next tile of this workgroup (its first chunk is in flight already when NBUF == 2)
  if (!more) break;
  cur = nxt;
  L = L_next;
  }  // tiles
  SA_STAMP_FLUSH;
#endif
}

template <int MT, int CK, int NW, int R, int NBUF, bool HEADS, int STEM_CIN = 0, bool EXT = false, bool UPS = false, int ITAP = -1, bool XP = false,
          bool PERS = false>
int launch2(const ConvParams2& p, hipStream_t st) {
  constexpr int TH = NW * R;
  constexpr int N_IN = ((TH + 2) * 34 * CK * 2 + 1023) / 1024;
  constexpr size_t lds = NBUF * ((size_t)N_IN * 1024 + (size_t)MT * (CK / 16) * 9 * 1024) +
                         (STEM_CIN ? ((size_t)(TH + 4) * 36 * STEM_CIN + (size_t)(9 * STEM_CIN + 1) * CK) * 4 : 0) +
                         (UPS ? 4096 : 0) +  // segment A of the low-resolution tile
                         (XP ? 65536 : 0) +  // expand + reduce weight fragments (CoutX <= 256)
                         (PERS ? 2048 : 0);  // the bias of every output channel (CoutP <= 512)
  if (PERS && p.CoutP > 512) return sa::fail(SA_ERR_UNSUPPORTED, "sa_conv3x3: the persistent two-workgroup kernels hold at most 512 biases in LDS");
  if (XP && (p.CoutX > 256 || p.CoutX % 32)) return sa::fail(SA_ERR_UNSUPPORTED, "sa_conv3x3_bneck_bf16: CoutX must be a multiple of 32, <= 256");
  ConvParams2 q = p;
  q.tiles_x = (p.W + 31) / 32;
  q.tiles_y = (p.H + TH - 1) / TH;
  const int co32_n = (p.CoutP + 31) / 32;
  q.co_tiles = (co32_n + MT - 1) / MT;
  // Non-temporal input copies (SA_CONV_NT=0 turns them off, A/B): a layer whose cout tiles all sit in ONE workgroup reads every
  // input tile once (plus the halo its neighbours share) -- streamed data. The guide measures issue -> landed -18 % for `nt` on
  // once-read LDS-DMA streams and a LOSS where several CUs re-read the same lines from L2, hence the co_tiles == 1 rule.
  static const bool nt_on = [] {
    const char* v = getenv("SA_CONV_NT");
    return !v || atoi(v) != 0;
  }();
  // (round 4: only layers with <= 4 chunks. On the ResNet decoder's concatenated convs -- 320 -> 64 @256^2, 20 chunks -- the hint
  //  cost 0.52 vs 0.35 ms: with many chunks the halo rows two vertically neighbouring tiles share are worth keeping in L2)
  static const int nt_max_chunks = [] {
    const char* v = getenv("SA_CONV_NT_MAX_CHUNKS");  // (A/B runs; 99 = round 3's rule)
    return v ? atoi(v) : 4;
  }();
  q.nt_in = (nt_on && q.co_tiles == 1 && !STEM_CIN && (p.C0P + p.C1P) / CK <= nt_max_chunks) ? 1 : 0;
  {
    auto lg = [](int v) { int k = 0; while ((1 << k) < v) ++k; return (1 << k) == v ? k : -1; };
    q.sh_co = lg(q.co_tiles), q.sh_tx = lg(q.tiles_x), q.sh_ty = lg(q.tiles_y);
    if (q.sh_co < 0 || q.sh_tx < 0 || q.sh_ty < 0) q.sh_co = q.sh_tx = q.sh_ty = -1;
  }
  if (p.planar && CK != 16 && !STEM_CIN) return sa::fail(SA_ERR_UNSUPPORTED, "sa_conv3x3: SA_LAYOUT_PLANES16 needs 16-channel chunks");
  q.pix_bytes0 = p.planar ? 32 : p.C0P * 2;
  q.pix_bytes1 = p.planar ? 32 : p.C1P * 2;
  q.blk_bytes_in = p.planar ? (unsigned)((size_t)p.H * p.W * 32) : 32u;
  q.blk_bytes_in1 = (unsigned)((size_t)(p.H / 2) * (p.W / 2) * 32);
  if (UPS && !(p.planar && CK == 16 && p.C1P > 0 && p.C0P > 0 && p.H % 2 == 0 && p.W % 2 == 0))
    return sa::fail(SA_ERR_UNSUPPORTED, "sa_conv3x3: the upsampling source mode of the DMA kernels needs SA_LAYOUT_PLANES16 and even H, W");
  q.out_pix_bytes = p.planar ? 32 : p.CoutP * 2;
  q.out_blk_bytes = p.planar ? (unsigned)((size_t)p.H * p.W * 32) : 32u;
  q.out_blk_bytes_pool = p.planar ? (unsigned)((size_t)(p.H / 2) * (p.W / 2) * 32) : 32u;
  if ((size_t)p.H * p.W * p.CoutP * 2 >= 0xFFFFFF00ull) return sa::fail(SA_ERR_UNSUPPORTED, "sa_conv3x3_bf16: one output frame must be smaller than 4 GiB");
  const size_t nblk = (size_t)q.tiles_x * q.tiles_y * q.co_tiles * p.B;
  if (nblk > 0x7fffffffull) return sa::fail(SA_ERR_INVALID_ARG, "sa_conv3x3_bf16: grid too large");
  if (!STEM_CIN && (size_t)p.H * p.W * (p.C0P > p.C1P ? p.C0P : p.C1P) * 2 >= 0xFFFFFF00ull)
    return sa::fail(SA_ERR_UNSUPPORTED, "sa_conv3x3_bf16: one frame must be smaller than 4 GiB");
  static bool attr_set = false;
  if (!attr_set && lds > 64 * 1024) {
    SA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_dma_kernel<MT, CK, NW, R, NBUF, HEADS, STEM_CIN, EXT, UPS, ITAP, XP, PERS>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  if (HEADS) {
    if (q.co_tiles != 1) return sa::fail(SA_ERR_UNSUPPORTED, "fused heads need all output channels in one workgroup (CoutP <= %d)", MT * 32);
    for (int hd = 0; hd < p.n_heads; ++hd)
      if (p.head_c[hd] > 64 || p.head_c[hd] < 1)
        return sa::fail(SA_ERR_UNSUPPORTED, "fused head %d: %d channels not supported", hd, p.head_c[hd]);
  }
  // Persistent launch for the double-buffered kernels: as many workgroups as the chip holds at once (occupancy x CUs), each
  // walking its share of the tiles with the next tile's first chunk prefetched across the tile boundary (see the kernel).
  // SA_CONV_PERSIST=0 launches one workgroup per tile (the pre-persistent behaviour, for A/B runs); SA_CONV_PERSIST=n (n > 0)
  // forces n workgroups per CU.
  size_t grid = nblk;
  if (NBUF == 2 && STEM_CIN == 0 && !EXT && CK == 16 && (MT == 4 || PERS)) {  // = PERSIST in the kernel
    static const int persist = [] {
      const char* v = getenv("SA_CONV_PERSIST");
      return v ? atoi(v) : -1;
    }();
    static int per_cu = 0, n_cu = 0;
    if (!n_cu) {
      int dev = 0, nb = 0;
      SA_HIP_CHECK(hipGetDevice(&dev));
      SA_HIP_CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
      SA_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(
          &nb, reinterpret_cast<const void*>(&conv3x3_dma_kernel<MT, CK, NW, R, NBUF, HEADS, STEM_CIN, EXT, UPS, ITAP, XP, PERS>), NW * 64, lds));
      per_cu = nb > 0 ? nb : 1;
      // (the occupancy API can answer one block too many -- MI355X_MICROARCH.md, correctness boundaries; LDS is the real limit here)
      int lds_cu = 0;
      SA_HIP_CHECK(hipDeviceGetAttribute(&lds_cu, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev));
      const int by_lds = lds ? (int)((size_t)lds_cu / lds) : 0;
      if (by_lds >= 1 && per_cu > by_lds) per_cu = by_lds;
    }
    if (g_grid_limit > 0) {
      if ((size_t)g_grid_limit < grid) grid = (size_t)g_grid_limit;
      // the tile schedule hands every XCD (workgroup index mod 8) one contiguous range of the tiles: fewer than 8 workgroups
      // would leave whole ranges unprocessed (round 5: the test hook allowed that, and stale buffer contents hid it)
      if (grid < 8) grid = nblk < 8 ? nblk : 8;
    } else if (g_grid_limit == 0 && persist != 0) {
      const size_t cap = (size_t)(persist > 0 ? persist : per_cu) * (size_t)n_cu;
      if (cap < grid) grid = cap;
    }
    // whatever policy chose the grid: the XCD tile schedule needs one workgroup per XCD range (or one per tile)
    if (grid < 8 && grid != nblk) grid = nblk < 8 ? nblk : 8;
  }
  hipLaunchKernelGGL((conv3x3_dma_kernel<MT, CK, NW, R, NBUF, HEADS, STEM_CIN, EXT, UPS, ITAP, XP, PERS>), dim3((unsigned)grid), dim3(NW * 64), lds, st, q);
  SA_LAUNCH_CHECK();
  return SA_OK;
}

#if !defined(SA_CONV_ITAP)
#define SA_CONV_ITAP 3
#endif
#if !defined(SA_CONV_ITAP_MIN_CHUNKS)
#define SA_CONV_ITAP_MIN_CHUNKS 5
#endif
// SA_CONV_LATE_ISSUE=0 queues every layer's copies right after the barrier (A/B runs)
bool late_issue(int n_chunks) {
  static const bool on = [] {
    const char* v = getenv("SA_CONV_LATE_ISSUE");
    return !v || atoi(v) != 0;
  }();
  return on && n_chunks >= SA_CONV_ITAP_MIN_CHUNKS;
}

template <int MT, int CK>
int launch2_pick(const ConvParams2& p, hipStream_t st) {
  if (p.post_scale || p.residual || p.relu_last) {  // extended epilogue
    if (p.n_heads > 0) {  // + fused heads (round 4: the ResNet decoder's Conv + BN + ReLU in front of its heads): 64-channel tiles only
      if constexpr (MT == 2 && CK == 16)
        return late_issue((p.C0P + p.C1P) / CK) ? launch2<2, 16, 8, 2, 2, true, 0, true, false, SA_CONV_ITAP>(p, st)
                                                : launch2<2, 16, 8, 2, 2, true, 0, true>(p, st);
      else
        return sa::fail(SA_ERR_UNSUPPORTED, "sa_conv3x3_ex_heads_bf16: fused heads behind the extended epilogue need 33-64 output channels and 16-channel chunks");
    }
    if ((p.C0P + p.C1P) == CK) return launch2<MT, CK, 4, 2, 1, false, 0, true>(p, st);
    // (CK == 16, planes: the mid-chunk variant holds 2 workgroups per CU -- 128 registers -- without spills, the other one
    //  spills 21, so every multi-chunk layer takes it; hourglass 3x3 convs: 1.0 -> see profiles/r03_ab_session.md section 5)
    if constexpr (CK == 16) return launch2<MT, CK, 8, 2, 2, false, 0, true, false, SA_CONV_ITAP>(p, st);
    return late_issue((p.C0P + p.C1P) / CK) ? launch2<MT, CK, 8, 2, 2, false, 0, true, false, SA_CONV_ITAP>(p, st)
                                            : launch2<MT, CK, 8, 2, 2, false, 0, true>(p, st);
  }
  // many-chunk layers queue the next chunk's copies in the middle of the chunk (ITAP, see the kernel)
  const bool mid = late_issue((p.C0P + p.C1P) / CK);
  if (p.n_heads > 0) return mid ? launch2<MT, CK, 8, 2, 2, true, 0, false, false, SA_CONV_ITAP>(p, st) : launch2<MT, CK, 8, 2, 2, true>(p, st);
  // single-chunk layers (Cin <= CK) are HBM-bound: small single-stage tiles, many workgroups per CU
  if ((p.C0P + p.C1P) == CK) return launch2<MT, CK, 4, 2, 1, false>(p, st);
  // (4-wave / 8x32 and 4-wave / 16x32 tiles were measured 4-6 % slower than 8 waves x 2 rows on every multi-chunk layer)
  // Round 5: the persistent tile loop with a counted wait (PERS, see the kernel) -- built, bitwise neutral, and SLOWER on every
  // layer in both of its forms (profiles/r05_ab_session.md section 1: 3-10 %), so one workgroup per tile stays the default.
  // SA_CONV_PERS=1 turns it on (A/B), 2 only for the few-chunk (ITAP < 0) layers, 3 only for the many-chunk ones.
  if constexpr (CK == 16) {
    static const int pers_env = [] {
      const char* v = getenv("SA_CONV_PERS");
      return v ? atoi(v) : 0;
    }();
    const int pers = g_persistent >= 0 ? g_persistent : pers_env;
    if (p.CoutP <= 512 && (pers == 1 || (pers == 2 && !mid) || (pers == 3 && mid)))
      return mid ? launch2<MT, CK, 8, 2, 2, false, 0, false, false, SA_CONV_ITAP, false, true>(p, st)
                 : launch2<MT, CK, 8, 2, 2, false, 0, false, false, -1, false, true>(p, st);
  }
  return mid ? launch2<MT, CK, 8, 2, 2, false, 0, false, false, SA_CONV_ITAP>(p, st) : launch2<MT, CK, 8, 2, 2, false>(p, st);
}

template <int MT, int R, int CK, int MODE>
int launch(const ConvParams& p, hipStream_t st) {
  constexpr int TH = 4 * R;
  constexpr size_t lds = (size_t)(TH + 2) * 34 * (CK * 2 + 16) + (size_t)MT * (CK / 16) * 9 * 1024;
  ConvParams q = p;
  q.tiles_x = (p.W + 31) / 32;
  q.tiles_y = (p.H + TH - 1) / TH;
  const int co32_n = (p.CoutP + 31) / 32;
  q.co_tiles = (co32_n + MT - 1) / MT;
  const size_t nblk = (size_t)q.tiles_x * q.tiles_y * q.co_tiles * p.B;
  if (nblk > 0x7fffffffull) return sa::fail(SA_ERR_INVALID_ARG, "sa_conv3x3_bf16: grid too large");
  static bool attr_set = false;
  if (!attr_set && lds > 64 * 1024) {
    SA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_mfma_kernel<MT, R, CK, MODE>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL((conv3x3_mfma_kernel<MT, R, CK, MODE>), dim3((unsigned)nblk), dim3(256), lds, st, q);
  SA_LAUNCH_CHECK();
  return SA_OK;
}

template <int MT, int R, int CK>
int launch_mode(const ConvParams& p, int mode, hipStream_t st) {
  switch (mode) {
    case 0: return launch<MT, R, CK, 0>(p, st);
    case 1: return launch<MT, R, CK, 1>(p, st);
    case 2: return launch<MT, R, CK, 2>(p, st);
    case 4: return launch<MT, R, CK, 4>(p, st);
    default: return sa::fail(SA_ERR_UNSUPPORTED, "sa_conv3x3_bf16: unsupported mode %d", mode);
  }
}

}  // namespace

int sa_internal_grid_limit() { return g_grid_limit; }

static int conv3x3_impl(const void* src0, int C0P, const void* src1, int C1P, int mode, const void* w,
                        const float* bias, int CoutP, int relu, int B, int H, int W, void* dst, void* dst_pool,
                        int n_heads, const float* const* head_w, const float* const* head_b, const int* head_c,
                        const int* head_act, float* const* head_dst, sa_stream_t stream,
                        const float* post_scale = nullptr, const float* post_shift = nullptr,
                        const void* residual = nullptr, int res_mode = 0, int relu_last = 0, const ConvParams2* xp = nullptr) {
  SA_REQUIRE(src0 && w && bias && (dst || dst_pool || n_heads > 0 || xp), "sa_conv3x3_bf16: NULL pointer");
  SA_REQUIRE(n_heads >= 0 && n_heads <= 2, "sa_conv3x3_bf16: at most 2 fused heads");
  SA_REQUIRE(C0P > 0 && C0P % 16 == 0 && C1P % 16 == 0 && CoutP > 0 && CoutP % 16 == 0,
             "sa_conv3x3_bf16: channels must be padded to multiples of 16 (C0P=%d C1P=%d CoutP=%d)", C0P, C1P, CoutP);
  SA_REQUIRE(B > 0 && H > 0 && W > 0, "sa_conv3x3_bf16: bad shape");
  SA_REQUIRE((mode & 3) == 0 || (src1 && C1P > 0), "sa_conv3x3_bf16: mode needs src1");
  SA_REQUIRE((mode & 3) != 0 || C1P == 0, "sa_conv3x3_bf16: C1P given without a src1 mode");
  SA_REQUIRE(!(mode & 2) || (H % 2 == 0 && W % 2 == 0), "sa_conv3x3_bf16: upsample mode needs even H, W");
  ConvParams p;
  p.src0 = (const uint16_t*)src0;
  p.src1 = (const uint16_t*)src1;
  p.w = (const uint16_t*)w;
  p.bias = bias;
  p.dst = (uint16_t*)dst;
  p.C0P = C0P;
  p.C1P = C1P;
  p.CoutP = CoutP;
  p.B = B;
  p.H = H;
  p.W = W;
  p.relu = relu;
  hipStream_t st = (hipStream_t)stream;
  // Chunk size. 16-channel chunks halve the LDS footprint of a stage (19.6 KB halo tile + MT*9 KB weights), so TWO
  // 8-wave workgroups fit on a CU and one's copies / epilogue overlap the other's MFMAs: measured 2-14 % faster than
  // 32-channel chunks (one workgroup per CU) on every multi-chunk layer of the benchmark model, 1.36-1.49 PFLOP/s on the
  // Cin >= 256 layers. The exception is the Cin = 32, Cout <= 32 layer, where one 32-channel chunk (no K loop at all) wins.
  // SA_CONV_VARIANT=ck32 / ck16 force either choice (tools/conv_bench.py experiments).
  static const int force_ck = [] {
    const char* v = getenv("SA_CONV_VARIANT");
    return !v ? 0 : !strcmp(v, "ck16") ? 16 : !strcmp(v, "ck32") ? 32 : 0;
  }();
  const bool can32 = (C0P % 32 == 0) && (C1P % 32 == 0);
  const int planar = (mode & SA_LAYOUT_PLANES16) ? 1 : 0;  // planes are 16 channels: 16-channel chunks only
  const bool ck32 = can32 && !planar && (force_ck == 32 || (force_ck == 0 && C0P + C1P == 32 && CoutP <= 32));
  const int co32_n = (CoutP + 31) / 32;
  const int src_mode = mode & 7;
  // On 16-channel planes the upsampling source mode runs on the DMA pipeline too (UPS kernels: the half-resolution tile is
  // expanded in LDS); on NHWC tensors it stays with the register-staged first-generation kernel below.
  const bool ups = src_mode == SA_SRC1_UPSAMPLE2X && planar;
  if (src_mode == SA_SRC1_NONE || src_mode == SA_SRC1_DIRECT || ups) {
    // v2 DMA pipeline; the pooled output (SA_DST_POOL2X) is only implemented here
    static const uint16_t* zeros = nullptr;
    if (!zeros) {
      void* z = nullptr;
      SA_HIP_CHECK(hipMalloc(&z, 256));
      SA_HIP_CHECK(hipMemset(z, 0, 256));
      zeros = (const uint16_t*)z;
    }
    SA_REQUIRE(!dst_pool || (H % 2 == 0 && W % 2 == 0), "sa_conv3x3_bf16: pooled output needs even H, W");
    ConvParams2 q = {};
    q.src0 = p.src0;
    q.src1 = p.src1;
    q.w = p.w;
    q.bias = bias;
    q.dst = (uint16_t*)dst;
    q.dst_pool = (uint16_t*)dst_pool;
    q.zeros = zeros;
    q.C0P = C0P;
    q.C1P = C1P;
    q.CoutP = CoutP;
    q.B = B;
    q.H = H;
    q.W = W;
    q.relu = relu;
    q.post_scale = post_scale;
    q.post_shift = post_shift;
    q.residual = (const uint16_t*)residual;
    q.res_mode = res_mode;
    q.relu_last = relu_last;
    q.planar = planar;
    SA_REQUIRE(!residual || n_heads == 0, "sa_conv3x3: a residual and fused heads are exclusive");
    SA_REQUIRE(!(post_scale || relu_last) || n_heads == 0 || (co32_n == 2 && !ck32 && !dst_pool),
               "sa_conv3x3_ex_heads_bf16: fused heads behind the extended epilogue need 33-64 (padded) output channels");
    SA_REQUIRE(!post_scale == !post_shift, "sa_conv3x3: post_scale and post_shift come together");
    SA_REQUIRE(!(residual && res_mode) || (H % 2 == 0 && W % 2 == 0), "sa_conv3x3: half-resolution residual needs even H, W");
    q.n_heads = n_heads;
    for (int hd = 0; hd < n_heads; ++hd) {
      q.head_w[hd] = head_w[hd];
      q.head_b[hd] = head_b[hd];
      q.head_c[hd] = head_c[hd];
      q.head_act[hd] = head_act[hd];
      q.head_dst[hd] = head_dst[hd];
    }
    if (xp) {  // fused bottleneck tail (sa_conv3x3_bneck_bf16): the 64-channel extended kernel + expand / reduce stages
      SA_REQUIRE(CoutP == 64 && C1P == 0 && src_mode == SA_SRC1_NONE && !residual && n_heads == 0 && !dst_pool,
                 "sa_conv3x3_bneck_bf16: a plain 3x3 conv with 64 (padded) output channels");
      q.xp_w = xp->xp_w, q.xp_bias = xp->xp_bias, q.xp_scale = xp->xp_scale, q.xp_shift = xp->xp_shift, q.xp_res = xp->xp_res;
      q.xp_dst = xp->xp_dst, q.CoutX = xp->CoutX, q.xp_relu = xp->xp_relu, q.xp_relu_last = xp->xp_relu_last;
      q.rd_w = xp->rd_w, q.rd_bias = xp->rd_bias, q.rd_scale = xp->rd_scale, q.rd_shift = xp->rd_shift, q.rd_dst = xp->rd_dst;
      q.CoutR = xp->CoutR, q.rd_relu = xp->rd_relu, q.rd_relu_last = xp->rd_relu_last;
      q.xp_pix_bytes = planar ? 32 : q.CoutX * 2;
      SA_REQUIRE((size_t)H * W * q.CoutX * 2 < 0xFFFFFF00ull, "sa_conv3x3_bneck_bf16: one frame of the expanded tensor must be smaller than 4 GiB");
      return late_issue((C0P + C1P) / 16) ? launch2<2, 16, 8, 2, 2, false, 0, true, false, SA_CONV_ITAP, true>(q, st)
                                          : launch2<2, 16, 8, 2, 2, false, 0, true, false, -1, true>(q, st);
    }
    if (n_heads > 0 && co32_n > 2) {
      // heads need every output channel in one workgroup: 4 cout tiles (<= 128 channels), CK = 16 keeps two LDS
      // stages within 160 KiB
      SA_REQUIRE(co32_n <= 4, "sa_conv3x3_heads_bf16: fused heads support at most 128 output channels");
      // (16 waves x 1 row on the same tile -- four waves per SIMD in the one workgroup a CU holds -- measured 8 % slower:
      // 0.327 -> 0.353 ms, profiles/r02_ab_session.md)
      return late_issue((C0P + C1P) / 16) ? launch2<4, 16, 8, 2, 2, true, 0, false, false, SA_CONV_ITAP>(q, st)
                                          : launch2<4, 16, 8, 2, 2, true>(q, st);
    }
    // Experiment switch (tools/ab runs): SA_CONV_MT4=n sends plain multi-chunk layers with >= n output channels (a multiple
    // of 128) to the 128-couts-per-workgroup persistent kernel (one workgroup per CU, half the input re-staging per MFMA).
    static const int mt4_min = [] {
      const char* v = getenv("SA_CONV_MT4");
      return v ? atoi(v) : 0;
    }();
    if (mt4_min > 0 && n_heads == 0 && !post_scale && !residual && !relu_last && CoutP % 128 == 0 && CoutP >= mt4_min &&
        (C0P + C1P) > 16 && !ck32)
      return launch2<4, 16, 8, 2, 2, false>(q, st);
    // Small launches (few frames, or the 32 x 32 / 64 x 64 layers of a small batch): with 64 output channels per workgroup the
    // grid does not fill the chip (two workgroups per CU); 32 channels per workgroup double the workgroup count at the price of
    // staging every input tile twice. Same arithmetic per output channel: bitwise the same results. SA_CONV_SMALL_MT1=0 turns
    // the rule off (A/B).
    static const bool small_mt1 = [] {
      const char* v = getenv("SA_CONV_SMALL_MT1");
      return !v || atoi(v) != 0;
    }();
    static int n_cu = 0;
    if (!n_cu) {
      int dev = 0;
      SA_HIP_CHECK(hipGetDevice(&dev));
      SA_HIP_CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    const size_t tiles2 = (size_t)B * ((H + 15) / 16) * ((W + 31) / 32) * ((co32_n + 1) / 2);
    const bool few = small_mt1 && co32_n >= 2 && n_heads == 0 && (C0P + C1P) > 16 && tiles2 < (size_t)2 * n_cu;
    if (ups) {
      SA_REQUIRE(n_heads == 0 && !post_scale && !residual && !relu_last,
                 "sa_conv3x3: fused heads / the extended epilogue are not available with the upsampling source mode");
      if (co32_n >= 2 && !few) return launch2<2, 16, 8, 2, 2, false, 0, false, true, SA_CONV_ITAP>(q, st);
      return launch2<1, 16, 8, 2, 2, false, 0, false, true, SA_CONV_ITAP>(q, st);
    }
    if (co32_n >= 2 && !few) return ck32 ? launch2_pick<2, 32>(q, st) : launch2_pick<2, 16>(q, st);
    return ck32 ? launch2_pick<1, 32>(q, st) : launch2_pick<1, 16>(q, st);
  }
  SA_REQUIRE(dst && !dst_pool && n_heads == 0 && !post_scale && !residual && !relu_last && !planar,
             "sa_conv3x3_bf16: pooled output / fused heads / extended epilogue / SA_LAYOUT_PLANES16 are not available with pool/upsample source modes");
  if (co32_n >= 2) {
    return ck32 ? launch_mode<2, 4, 32>(p, src_mode, st) : launch_mode<2, 4, 16>(p, src_mode, st);
  }
  return ck32 ? launch_mode<1, 4, 32>(p, src_mode, st) : launch_mode<1, 4, 16>(p, src_mode, st);
}


extern "C" {

#if defined(SA_CONV_STAMP)
// instrumented build only (not part of the ABI): zero / read the segment sums of the launches in between
int sa_conv3x3_stamp_reset() {
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  SA_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), z, sizeof(z)));
  return SA_OK;
}
int sa_conv3x3_stamp_read(unsigned long long* out) {
  SA_HIP_CHECK(hipDeviceSynchronize());
  SA_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp), 8 * sizeof(unsigned long long)));
  return SA_OK;
}
#endif

int sa_conv3x3_set_persistent(int mode) {
  const int prev = g_persistent;
  g_persistent = mode;
  return prev;
}

int sa_conv3x3_set_grid_limit(int n) {
  const int prev = g_grid_limit;
  g_grid_limit = n;
  return prev;
}

size_t sa_conv3x3_packed_elems(int C0P, int C1P, int CoutP) {
  const size_t co32 = (CoutP + 31) / 32, k16 = (size_t)(C0P + C1P) / 16;
  return co32 * k16 * 9 * 64 * 8;
}

int sa_pack_conv3x3_weights(const float* kk, int C0, int C0P, int C1, int C1P, int Cout, int CoutP,
                            uint16_t* packed) {
  SA_REQUIRE(C0P % 16 == 0 && C1P % 16 == 0 && C0 <= C0P && C1 <= C1P && Cout <= CoutP && CoutP % 16 == 0,
             "sa_pack_conv3x3_weights: channel counts must be padded to multiples of 16");
  const int Cin = C0 + C1, CinP = C0P + C1P, K16 = CinP / 16, co32_n = (CoutP + 31) / 32;
  for (int co32 = 0; co32 < co32_n; ++co32)
    for (int k16 = 0; k16 < K16; ++k16)
      for (int tap = 0; tap < 9; ++tap)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            const int co = co32 * 32 + (lane & 31);
            const int cp = k16 * 16 + (lane >> 5) * 8 + j;  // padded concat channel
            int ci = -1;                                     // Keras input channel
            if (cp < C0P) {
              if (cp < C0) ci = cp;
            } else if (cp - C0P < C1) {
              ci = C0 + (cp - C0P);
            }
            float v = 0.0f;
            if (co < Cout && ci >= 0) v = kk[((size_t)tap * Cin + ci) * Cout + co];  // (kh,kw,Cin,Cout)
            packed[((((size_t)co32 * K16 + k16) * 9 + tap) * 64 + lane) * 8 + j] = sa::f2h(v);
          }
  return SA_OK;
}

int sa_conv3x3_bf16(const void* src0, int C0P, const void* src1, int C1P, int mode, const void* w,
                    const float* bias, int CoutP, int relu, int B, int H, int W, void* dst, void* dst_pool,
                    sa_stream_t stream) {
  return conv3x3_impl(src0, C0P, src1, C1P, mode, w, bias, CoutP, relu, B, H, W, dst, dst_pool, 0, nullptr, nullptr,
                      nullptr, nullptr, nullptr, stream);
}

int sa_conv3x3_ex_bf16(const void* src0, int C0P, const void* src1, int C1P, int mode, const void* w,
                       const float* bias, int CoutP, int relu, int B, int H, int W, void* dst, void* dst_pool,
                       const float* post_scale, const float* post_shift, const void* residual, int res_mode,
                       int relu_last, sa_stream_t stream) {
  return conv3x3_impl(src0, C0P, src1, C1P, mode, w, bias, CoutP, relu, B, H, W, dst, dst_pool, 0, nullptr, nullptr,
                      nullptr, nullptr, nullptr, stream, post_scale, post_shift, residual, res_mode, relu_last);
}

int sa_conv3x3_ex_heads_bf16(const void* src0, int C0P, const void* src1, int C1P, int mode, const void* w, const float* bias,
                             int CoutP, int relu, int B, int H, int W, void* dst, const float* post_scale, const float* post_shift,
                             int relu_last, int n_heads, const float* const* head_w, const float* const* head_b,
                             const int* head_c, const int* head_act, float* const* head_dst, sa_stream_t stream) {
  SA_REQUIRE(n_heads >= 1 && head_w && head_b && head_c && head_act && head_dst, "sa_conv3x3_ex_heads_bf16: bad head arguments");
  return conv3x3_impl(src0, C0P, src1, C1P, mode, w, bias, CoutP, relu, B, H, W, dst, nullptr, n_heads, head_w, head_b, head_c,
                      head_act, head_dst, stream, post_scale, post_shift, nullptr, 0, relu_last);
}

int sa_stem_conv3x3x2_bf16(const void* src, int src_is_u8, int B, int H, int W, int Cin, const float* w0,
                           const float* bias0, int C0P, int relu0, const void* w1, const float* bias1, int CoutP,
                           int relu1, void* dst, void* dst_pool, int layout, sa_stream_t stream) {
  SA_REQUIRE(src && w0 && bias0 && w1 && bias1 && (dst || dst_pool), "sa_stem_conv3x3x2_bf16: NULL pointer");
  SA_REQUIRE(layout == SA_LAYOUT_NHWC || layout == SA_LAYOUT_PLANES16, "sa_stem_conv3x3x2_bf16: bad layout");
  SA_REQUIRE(Cin == 1 || Cin == 3, "sa_stem_conv3x3x2_bf16: Cin must be 1 or 3");
  SA_REQUIRE(C0P == 16 || C0P == 32, "sa_stem_conv3x3x2_bf16: the first conv must have 16 or 32 (padded) output channels");
  SA_REQUIRE(CoutP > 0 && CoutP % 16 == 0 && CoutP <= 64, "sa_stem_conv3x3x2_bf16: CoutP must be a multiple of 16, <= 64");
  SA_REQUIRE(!dst_pool || (H % 2 == 0 && W % 2 == 0), "sa_stem_conv3x3x2_bf16: pooled output needs even H, W");
  ConvParams2 q = {};
  q.src0 = nullptr;
  q.src1 = nullptr;
  q.w = (const uint16_t*)w1;
  q.bias = bias1;
  q.dst = (uint16_t*)dst;
  q.dst_pool = (uint16_t*)dst_pool;
  q.zeros = nullptr;
  q.C0P = C0P;
  q.C1P = 0;
  q.CoutP = CoutP;
  q.B = B;
  q.H = H;
  q.W = W;
  q.relu = relu1;
  q.n_heads = 0;
  q.stem_src = src;
  q.stem_w = w0;
  q.stem_b = bias0;
  q.stem_is_u8 = src_is_u8;
  q.stem_relu = relu0;
  q.planar = layout == SA_LAYOUT_PLANES16;
  hipStream_t st = (hipStream_t)stream;
  const bool two = CoutP > 32;
  if (C0P == 16) {
    if (Cin == 1) return two ? launch2<2, 16, 4, 2, 1, false, 1>(q, st) : launch2<1, 16, 4, 2, 1, false, 1>(q, st);
    return two ? launch2<2, 16, 4, 2, 1, false, 3>(q, st) : launch2<1, 16, 4, 2, 1, false, 3>(q, st);
  }
  if (Cin == 1) return two ? launch2<2, 32, 4, 2, 1, false, 1>(q, st) : launch2<1, 32, 4, 2, 1, false, 1>(q, st);
  return two ? launch2<2, 32, 4, 2, 1, false, 3>(q, st) : launch2<1, 32, 4, 2, 1, false, 3>(q, st);
}

size_t sa_pointwise_packed_elems(int CinP, int CoutP) { return (size_t)((CoutP + 31) / 32) * (CinP / 16) * 64 * 8; }

int sa_pack_pointwise_weights(const float* w, int Cin, int CinP, int Cout, int CoutP, void* packed) {
  // w: Keras 1x1 kernel [Cin][Cout] f32 -> MFMA A fragments [CoutP/32][CinP/16][64 lanes][8] of the storage type, the input
  // channels of a k-step in the order the fused stages' B fragments hold them: lane half h, element j <-> channel
  // 16 k + 8 (j >> 2) + 4 h + (j & 3) (the accumulator layout of v_mfma_f32_32x32x16, read back as an operand)
  SA_REQUIRE(w && packed && CinP % 16 == 0 && CoutP % 32 == 0 && Cin <= CinP && Cout <= CoutP, "sa_pack_pointwise_weights: bad arguments");
  uint16_t* o = static_cast<uint16_t*>(packed);
  const int K16 = CinP / 16;
  for (int c32 = 0; c32 < CoutP / 32; ++c32)
    for (int k = 0; k < K16; ++k)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
          const int co = c32 * 32 + (lane & 31), ci = 16 * k + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3);
          const float v = (co < Cout && ci < Cin) ? w[(size_t)ci * Cout + co] : 0.0f;
          o[(((size_t)c32 * K16 + k) * 64 + lane) * 8 + j] = sa::f2h(v);
        }
  return SA_OK;
}

int sa_conv3x3_bneck_bf16(const void* src, int CinP, int layout, const void* w, const float* bias, int relu, const float* post_scale,
                          const float* post_shift, int relu_last, int B, int H, int W, const void* xp_w, const float* xp_bias,
                          const float* xp_scale, const float* xp_shift, const void* xp_res, int xp_relu, int xp_relu_last,
                          int CoutX, void* xp_dst, const void* rd_w, const float* rd_bias, const float* rd_scale,
                          const float* rd_shift, int rd_relu, int rd_relu_last, int CoutR, void* rd_dst, sa_stream_t stream) {
  SA_REQUIRE(xp_w && xp_bias && xp_dst && CoutX >= 32 && CoutX % 32 == 0, "sa_conv3x3_bneck_bf16: expand stage: weights, bias, dst, CoutX %% 32 == 0");
  SA_REQUIRE(!xp_scale == !xp_shift && !rd_scale == !rd_shift, "sa_conv3x3_bneck_bf16: scale and shift come together");
  SA_REQUIRE(!rd_w || (rd_bias && rd_dst && CoutR == 64), "sa_conv3x3_bneck_bf16: reduce stage: bias, dst and 64 (padded) output channels");
  ConvParams2 x = {};
  x.xp_w = (const uint16_t*)xp_w, x.xp_bias = xp_bias, x.xp_scale = xp_scale, x.xp_shift = xp_shift, x.xp_res = (const uint16_t*)xp_res;
  x.xp_dst = (uint16_t*)xp_dst, x.CoutX = CoutX, x.xp_relu = xp_relu, x.xp_relu_last = xp_relu_last;
  x.rd_w = (const uint16_t*)rd_w, x.rd_bias = rd_bias, x.rd_scale = rd_scale, x.rd_shift = rd_shift, x.rd_dst = (uint16_t*)rd_dst;
  x.CoutR = CoutR, x.rd_relu = rd_relu, x.rd_relu_last = rd_relu_last;
  return conv3x3_impl(src, CinP, nullptr, 0, SA_SRC1_NONE | (layout & SA_LAYOUT_PLANES16), w, bias, 64, relu, B, H, W, nullptr, nullptr, 0,
                      nullptr, nullptr, nullptr, nullptr, nullptr, stream, post_scale, post_shift, nullptr, 0, relu_last, &x);
}

int sa_conv3x3_heads_bf16(const void* src0, int C0P, const void* src1, int C1P, int mode, const void* w,
                          const float* bias, int CoutP, int relu, int B, int H, int W, void* dst, int n_heads,
                          const float* const* head_w, const float* const* head_b, const int* head_c,
                          const int* head_act, float* const* head_dst, sa_stream_t stream) {
  SA_REQUIRE(n_heads >= 1 && head_w && head_b && head_c && head_act && head_dst, "sa_conv3x3_heads_bf16: bad head arguments");
  return conv3x3_impl(src0, C0P, src1, C1P, mode, w, bias, CoutP, relu, B, H, W, dst, nullptr, n_heads, head_w, head_b,
                      head_c, head_act, head_dst, stream);
}

}  // extern "C"
