// 16-bit storage type of the network kernels. One library build = one type:
//   default            bfloat16 (8-bit mantissa, fp32 range)                      -> libsleap_amd.so
//   -DSA_HALF_FP16=1   IEEE half (11-bit mantissa, max 65504)                     -> libsleap_amd_fp16.so
// Same kernels, same MFMA shapes and rates (v_mfma_f32_32x32x16_{bf16,f16}, v_mfma_f32_16x16x32_{bf16,f16}); only the
// conversions and the operand element type differ. Why both: tests/diagnostics/precision_probe.py -- with bf16 storage the
// heads are 1-4 % of their range away from an fp32 network, which flips marginal peak / matching decisions; with fp16 storage
// the difference is 0.1-0.5 % and the end-to-end results agree. bf16 never overflows; fp16 can (activations > 65504).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace sa {

typedef __attribute__((ext_vector_type(8))) unsigned short h16x8_t;  // 16 B = 8 channels (raw bits)
typedef __attribute__((ext_vector_type(4))) unsigned short h16x4_t;

#if defined(SA_HALF_FP16)
typedef _Float16 half_elem;
#define SA_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define SA_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define SA_HALF_NAME "fp16"
#else
typedef __bf16 half_elem;
#define SA_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define SA_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define SA_HALF_NAME "bf16"
#endif
typedef __attribute__((ext_vector_type(8))) half_elem mfma_h8;  // MFMA A / B operand: 8 consecutive k values

// uint8 pixels enter the matrix-core stems as the storage-type value `byte * U8_ACT_SCALE` (exact in both types), with
// the packed first-layer weights carrying `input_scale / U8_ACT_SCALE`, split into hi + lo (+ mid) terms. bf16 has fp32's
// exponent range, so the split terms of w / 255 are normal numbers. In fp16 the lo terms of w / 255 (~1e-3 * 2^-12) would
// be subnormal (spacing 6e-8, two or three significant bits left); moving 2^-8 of the scale to the pixel side keeps the
// absolute weight error at 3e-8 of weights that are now O(0.1 - 1).
#if defined(SA_HALF_FP16)
constexpr float U8_ACT_SCALE = 1.0f / 256.0f;
#else
constexpr float U8_ACT_SCALE = 1.0f;
#endif

// round-to-nearest-even float -> storage bits. Device code uses the hardware conversion (v_cvt_pk_bf16_f32 /
// v_cvt_pk_f16_f32 on gfx950); the host twin is used by the weight packers.
__host__ __device__ inline uint16_t f2h(float f) {
#if defined(SA_HALF_FP16)
  return __builtin_bit_cast(uint16_t, static_cast<_Float16>(f));
#else
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_bit_cast(uint16_t, static_cast<__bf16>(f));
#endif
  union { float f; uint32_t u; } v;
  v.f = f;
  if ((v.u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((v.u >> 16) | 0x40);
  const uint32_t lsb = (v.u >> 16) & 1u;
  return (uint16_t)((v.u + 0x7fffu + lsb) >> 16);
#endif
}

#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) half_elem half2_elem;
// two floats -> packed pair (lo in bits 0..15) with ONE v_cvt_pk_{bf16,f16}_f32
__device__ __forceinline__ uint32_t f2h2(float lo, float hi) {
  f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, half2_elem));
}
#endif

__host__ __device__ inline float h2f(uint16_t h) {
#if defined(SA_HALF_FP16)
  return (float)__builtin_bit_cast(_Float16, h);
#else
  union { float f; uint32_t u; } v;
  v.u = ((uint32_t)h) << 16;
  return v.f;
#endif
}

// UpSampling2D(2, bilinear) == tf.image.resize half-pixel centres: output index o reads source
// coordinate o/2 - 0.25 -> taps (i0, i1) with weight w1 on i1, edge-clamped.
__device__ __forceinline__ void up2_taps(int o, int n, int& i0, int& i1, float& w1) {
  const int i = o >> 1;
  if (o & 1) {
    i0 = i;
    i1 = min(i + 1, n - 1);
    w1 = 0.25f;
  } else {
    i0 = max(i - 1, 0);
    i1 = i;
    w1 = 0.75f;
  }
}

__device__ __forceinline__ float up2_lerp(float tl, float tr, float bl, float br, float wy, float wx) {
  const float t = tl + (tr - tl) * wx;
  const float b = bl + (br - bl) * wx;
  return t + (b - t) * wy;
}

#if defined(__HIP_DEVICE_COMPILE__)
// max of a packed pair of 16-bit values with a packed bound (ReLU: bound 0, none: -inf): ONE v_pk_max_f16 for two values in the
// fp16 build. Rounding to 16 bits is monotonic, so max(round(a), bound) == round(max(a, bound)) (up to the sign of a zero).
#if defined(SA_HALF_FP16)
#define SA_HAS_PK_MAX 1
__device__ __forceinline__ uint32_t pk_max(uint32_t v, uint32_t bound) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(h2, v), __builtin_bit_cast(h2, bound)));
}
#define SA_PK_NEG_INF 0xFC00FC00u
#else
#define SA_HAS_PK_MAX 0
#endif
#endif

// four floats at dword alignment: stored with one global_store_dwordx4 (gfx950 global accesses need dword alignment only)
struct __attribute__((packed, aligned(4))) f32x4_unaligned {
  float x, y, z, w;
};

#if defined(__HIP_DEVICE_COMPILE__)
// value of the neighbouring lane (lane ^ 1) through DPP quad_perm [1,0,3,2]: no LDS crossbar (ds_bpermute) involved
__device__ __forceinline__ float dpp_xor1(float v) {
  const int i = __float_as_int(v);
  return __int_as_float(__builtin_amdgcn_update_dpp(i, i, 0xB1, 0xF, 0xF, false));
}
// t4[j] = max(t[j], t[j] of lane ^ 1) for four values: `v_max_f32_dpp` takes the neighbour lane as an operand modifier (the
// compiler's DPP combine leaves dpp_xor1 + fmaxf as v_mov_b32_dpp + v_max_f32). `s_nop 1`: a VALU result read through DPP needs
// two wait states and the hazard recogniser does not look into asm. SA_DPPMAX=0: the two-instruction form (A/B).
#if !defined(SA_DPPMAX)
#define SA_DPPMAX 1
#endif
__device__ __forceinline__ void max_xor1_x4(const float (&t)[4], float (&t4)[4]) {
#if SA_DPPMAX
  asm("s_nop 1\n\t"
      "v_max_f32_dpp %0, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %1, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %2, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %3, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
      : "=&v"(t4[0]), "=&v"(t4[1]), "=&v"(t4[2]), "=&v"(t4[3])
      : "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]));
#else
#pragma unroll
  for (int j = 0; j < 4; ++j) t4[j] = fmaxf(t[j], dpp_xor1(t[j]));
#endif
}
// v_permlane32_swap: exchanges a's upper 32 lanes with b's lower 32 lanes. Afterwards lane l < 32 holds
// {a[l], a[l+32]} in (a, b) and lane l >= 32 holds {b[l-32], b[l]}: one instruction per dword turns the MFMA
// accumulator split (channels 0-3 in the lower half-wave, 4-7 in the upper) into 8 consecutive channels per lane.
__device__ __forceinline__ void swap32(unsigned& a, unsigned& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}
// v_permlane16_swap: exchanges a's odd rows of 16 lanes (lanes 16-31, 48-63) with b's even rows (0-15, 32-47).
__device__ __forceinline__ void swap16(unsigned& a, unsigned& b) {
  const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}
#endif

}  // namespace sa
