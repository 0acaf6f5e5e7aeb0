// Probe: does `buffer_load_dwordx4 ... lds` write ZEROS to LDS for out-of-range lanes (num_records check)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__global__ void k(const float* g, float* out, int n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* f = (float*)lds;
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) f[i] = -7.0f;  // poison
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, n * 4, 0x00020000);
  int voff = threadIdx.x * 16;
  if (threadIdx.x & 1) voff = 0x7fffff00;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(lds + wave * 1024), 16, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) out[i] = f[i];
}
int main() {
  float *g, *o, h[1024], r[1024];
  for (int i = 0; i < 1024; ++i) h[i] = i + 1;
  hipMalloc(&g, 4096); hipMalloc(&o, 4096);
  hipMemcpy(g, h, 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, g, o, 1024);
  hipMemcpy(r, o, 4096, hipMemcpyDeviceToHost);
  int ok = 1;
  for (int t = 0; t < 256; ++t) for (int j = 0; j < 4; ++j) {
    float want = (t & 1) ? 0.0f : h[t * 4 + j];
    if (r[t * 4 + j] != want) { if (ok) printf("mismatch at lane %d elem %d: got %f want %f\n", t, j, r[t*4+j], want); ok = 0; }
  }
  printf(ok ? "OOB lanes write zeros: YES\n" : "OOB lanes write zeros: NO\n");
  return !ok;
}
