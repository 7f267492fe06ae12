// Probe the lane semantics of gfx950's ds_read_b64_tr_b16 (transposing LDS read).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
// LDS holds lds[i] = i (as f16, i < 2048).  Each lane reads at byte address addr[lane]; out[lane*4 + j] = element j it received.
__global__ void k_tr(const int* addr, float* out) {
  __shared__ f16 lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (f16)(float)(i & 2047);
  __syncthreads();
  const unsigned base = (unsigned)(uintptr_t)lds;  // LDS byte offset of the array (generic->local offset is the low 32 bits for LDS apertures?)
  unsigned a = (unsigned)addr[threadIdx.x] + (unsigned)(size_t)(__attribute__((address_space(3))) f16*)lds;
  f16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (float)v[j];
}
extern "C" int trread(const int* addr, float* out, void* stream) {
  hipLaunchKernelGGL(k_tr, dim3(1), dim3(64), 0, (hipStream_t)stream, addr, out);
  return (int)hipGetLastError();
}
