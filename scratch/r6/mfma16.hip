// Round-6 probe: issue rates of the 16-wide MFMA shapes on gfx950 (16x16x32 f16, the legacy 16x16x16 f16) and how many plain VALU / exp
// fillers hide in their slots -- the design inputs of a 16x16-tile attention kernel.  Cycles by s_memtime (shader clock), one block per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define FMA(X) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(X) : "v"(c), "v"(d));
#define EXP(X) asm volatile("v_exp_f32 %0, %0" : "+v"(X));

template <int MODE, int FILL, bool EXPF>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters, float c, float d) {
  const int lane = threadIdx.x & 63;
  f32x16 big[4];
  f32x4 acc[8];
  for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) big[k][r] = 0.f;
  for (int k = 0; k < 8; ++k) for (int r = 0; r < 4; ++r) acc[k][r] = 0.f;
  f16x8 a, b; f16x4 a4, b4;
  for (int e = 0; e < 8; ++e) { a[e] = (f16)(lane * 0.01f + e); b[e] = (f16)(0.5f - e * 0.1f); }
  for (int e = 0; e < 4; ++e) { a4[e] = a[e]; b4[e] = b[e]; }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = lane * 0.001f + i * 0.01f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      if constexpr (MODE == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(big[i & 3]) : "v"(a), "v"(b));
      if constexpr (MODE == 1) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i & 7]) : "v"(a), "v"(b));
      if constexpr (MODE == 2) asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(acc[i & 7]) : "v"(a4), "v"(b4));
      if constexpr (MODE == 3) {
        if (i & 1) asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(acc[i & 7]) : "v"(a4), "v"(b4));
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i & 7]) : "v"(a), "v"(b));
      }
      if constexpr (MODE == 4) asm volatile("v_mfma_f32_32x32x8_f16 %0, %1, %2, %0" : "+v"(big[i & 3]) : "v"(a4), "v"(b4));
#pragma unroll
      for (int f = 0; f < FILL; ++f) { if (EXPF) { EXP(v[(i * FILL + f) & 7]) } else { FMA(v[(i * FILL + f) & 7]) } }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += big[k][r];
  for (int k = 0; k < 8; ++k) for (int r = 0; r < 4; ++r) s += acc[k][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
#define CASE(ID, M, F, E) case ID: hipLaunchKernelGGL((k<M, F, E>), dim3(blocks), dim3(threads), 0, s, out, cyc, iters, 0.999f, 0.001f); break;
extern "C" int mfma16(int id, float* out, unsigned long long* cyc, int blocks, int threads, int iters, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  switch (id) {
    CASE(0, 0, 0, false) CASE(1, 1, 0, false) CASE(2, 2, 0, false) CASE(3, 3, 0, false) CASE(4, 4, 0, false)
    CASE(5, 1, 1, false) CASE(6, 1, 2, false) CASE(7, 1, 3, false) CASE(8, 1, 4, false)
    CASE(9, 0, 4, false) CASE(10, 0, 6, false) CASE(11, 0, 8, false)
    CASE(12, 1, 1, true) CASE(13, 1, 2, true) CASE(14, 0, 2, true) CASE(15, 0, 4, true)
    CASE(16, 2, 1, false) CASE(17, 2, 2, false)
    default: return -1;
  }
  return (int)hipGetLastError();
}
