// bf16 helpers shared by the network kernels (device side).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace sa {

typedef __attribute__((ext_vector_type(8))) unsigned short bf16x8_t;  // 16 B = 8 bf16 channels
typedef __attribute__((ext_vector_type(4))) unsigned short bf16x4_t;

// round-to-nearest-even float -> bf16 (NaN kept quiet). Device code uses the hardware conversion
// (v_cvt_pk_bf16_f32 on gfx950, same rounding); the host twin is used by the weight packer.
__host__ __device__ inline uint16_t f2bf(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_bit_cast(uint16_t, static_cast<__bf16>(f));
#endif
  union { float f; uint32_t u; } v;
  v.f = f;
  if ((v.u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((v.u >> 16) | 0x40);
  const uint32_t lsb = (v.u >> 16) & 1u;
  return (uint16_t)((v.u + 0x7fffu + lsb) >> 16);
}

#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// two floats -> packed bf16 pair (lo in bits 0..15) with ONE v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t f2bf2(float lo, float hi) {
  f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
#endif

__host__ __device__ inline float bf2f(uint16_t h) {
  union { float f; uint32_t u; } v;
  v.u = ((uint32_t)h) << 16;
  return v.f;
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
// value of the neighbouring lane (lane ^ 1) through DPP quad_perm [1,0,3,2]: no LDS crossbar (ds_bpermute) involved
__device__ __forceinline__ float dpp_xor1(float v) {
  const int i = __float_as_int(v);
  return __int_as_float(__builtin_amdgcn_update_dpp(i, i, 0xB1, 0xF, 0xF, false));
}
// v_permlane32_swap: exchanges a's upper 32 lanes with b's lower 32 lanes. Afterwards lane l < 32 holds
// {a[l], a[l+32]} in (a, b) and lane l >= 32 holds {b[l-32], b[l]}: one instruction per dword turns the MFMA
// accumulator split (channels 0-3 in the lower half-wave, 4-7 in the upper) into 8 consecutive channels per lane.
__device__ __forceinline__ void swap32(unsigned& a, unsigned& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}
#endif

}  // namespace sa
