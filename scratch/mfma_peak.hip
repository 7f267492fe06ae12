// Sustained dense-fp16 MFMA rate of the whole chip: every SIMD issues independent v_mfma_f32_32x32x16_f16 back to back on non-trivial register
// data (operand toggling matters for power), for a chosen duration.  What clock / rate does the part hold under nothing but matrix work?
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ __launch_bounds__(256) void k_peak(float* out, int iters, unsigned long long* clk) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[8];
  for (int k = 0; k < 8; ++k)
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  f16x8 a[2], b[2];
  unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
  for (int q = 0; q < 2; ++q)
    for (int e = 0; e < 8; ++e) {
      h = h * 1664525u + 1013904223u; a[q][e] = (f16)(((int)(h >> 20) - 2048) * (1.f / 4096.f));
      h = h * 1664525u + 1013904223u; b[q][e] = (f16)(((int)(h >> 20) - 2048) * (1.f / 4096.f));
    }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k & 1], b[(k >> 1) & 1], acc[k & 7], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
  float s = 0.f;
  for (int k = 0; k < 8; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
  (void)lane;
}
extern "C" int mfma_peak(float* out, int blocks, int iters, unsigned long long* clk, hipStream_t s) {
  hipLaunchKernelGGL(k_peak, dim3(blocks), dim3(256), 0, s, out, iters, clk);
  return (int)hipGetLastError();
}
