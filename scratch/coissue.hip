// Probe: do VALU (v_exp + plain) and MFMA work of DIFFERENT waves on one SIMD overlap, and does a barrier-staggered ping-pong get it?
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ void mfma_phase(f32x16 (&acc)[4], const f16x8& a, const f16x8& b, int n) {
#pragma unroll
  for (int i = 0; i < 14; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i & 3], 0, 0, 0);
}
__device__ __forceinline__ void fma_phase(float (&v)[32], float c) {  // plain VALU only, 6 per element
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = fmaxf(v[i] * c + 0.25f, v[(i + 1) & 31]);
}
__device__ __forceinline__ void exp_phase(float (&v)[32], float c) {  // transcendental only
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]);
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i] = __builtin_amdgcn_exp2f(-v[i]);
}
__device__ __forceinline__ void valu_phase(float (&v)[32], float c) {
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i] = __builtin_amdgcn_exp2f(v[i] * c - 1.0f);
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i] = fmaxf(v[i], v[(i + 1) & 31]) + 0.5f;  // 2 more VALU per element
}
// MODE 0: every wave does MFMA phase then VALU phase (today's kernels). MODE 1: waves 0-3 only MFMA (x2), waves 4-7 only VALU (x2).
// MODE 2: as 0, with block barriers between phases and the second wave group shifted by one phase (ping-pong).
// MODE 3: MFMA only. MODE 4: VALU only.
template <int MODE>
__global__ __launch_bounds__(512) void k_mix(float* out, int iters, float c) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, grp = wave >> 2;
  f32x16 acc[4];
  for (int k = 0; k < 4; ++k)
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (f16)(lane * 0.01f + e); b[e] = (f16)(0.5f - e * 0.1f); }
  float v[32];
  for (int i = 0; i < 32; ++i) v[i] = lane * 0.001f + i * 0.01f;
  if (MODE == 2 && grp == 1) __builtin_amdgcn_s_barrier();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) { mfma_phase(acc, a, b, 14); valu_phase(v, c); }
    if (MODE == 1) { if (grp == 0) { mfma_phase(acc, a, b, 14); mfma_phase(acc, a, b, 14); } else { valu_phase(v, c); valu_phase(v, c); } }
    if (MODE == 2) { mfma_phase(acc, a, b, 14); __builtin_amdgcn_s_barrier(); valu_phase(v, c); __builtin_amdgcn_s_barrier(); }
    if (MODE == 5) { if ((wave & 1) == 0) { mfma_phase(acc, a, b, 14); mfma_phase(acc, a, b, 14); } else { valu_phase(v, c); valu_phase(v, c); } }
    if (MODE == 6) { if ((wave & 2) == 0) { mfma_phase(acc, a, b, 14); mfma_phase(acc, a, b, 14); } else { valu_phase(v, c); valu_phase(v, c); } }
    if (MODE == 7) { if (grp == 0) { mfma_phase(acc, a, b, 14); mfma_phase(acc, a, b, 14); } else { fma_phase(v, c); fma_phase(v, c); } }
    if (MODE == 8) { if (grp == 0) { mfma_phase(acc, a, b, 14); mfma_phase(acc, a, b, 14); } else { exp_phase(v, c); exp_phase(v, c); } }
    if (MODE == 9) fma_phase(v, c);
    if (MODE == 10) exp_phase(v, c);
    if (MODE == 3) mfma_phase(acc, a, b, 14);
    if (MODE == 4) valu_phase(v, c);
  }
  if (MODE == 2 && grp == 0) __builtin_amdgcn_s_barrier();
  float s = 0.f;
  for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
  for (int i = 0; i < 32; ++i) s += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
extern "C" int coissue(int mode, float* out, int blocks, int iters, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  static bool done = false;
  if (!done) {  // 100 KB of dynamic LDS per block: exactly one block (2 waves per SIMD) per CU
    hipFuncSetAttribute((const void*)k_mix<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)k_mix<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)k_mix<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)k_mix<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)k_mix<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)k_mix<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)k_mix<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)k_mix<7>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)k_mix<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)k_mix<9>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)k_mix<10>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    done = true;
  }
  switch (mode) {
    case 0: hipLaunchKernelGGL(k_mix<0>, dim3(blocks), dim3(512), 100 * 1024, s, out, iters, 0.7f); break;
    case 1: hipLaunchKernelGGL(k_mix<1>, dim3(blocks), dim3(512), 100 * 1024, s, out, iters, 0.7f); break;
    case 2: hipLaunchKernelGGL(k_mix<2>, dim3(blocks), dim3(512), 100 * 1024, s, out, iters, 0.7f); break;
    case 3: hipLaunchKernelGGL(k_mix<3>, dim3(blocks), dim3(512), 100 * 1024, s, out, iters, 0.7f); break;
    case 5: hipLaunchKernelGGL(k_mix<5>, dim3(blocks), dim3(512), 100 * 1024, s, out, iters, 0.7f); break;
    case 6: hipLaunchKernelGGL(k_mix<6>, dim3(blocks), dim3(512), 100 * 1024, s, out, iters, 0.7f); break;
    case 7: hipLaunchKernelGGL(k_mix<7>, dim3(blocks), dim3(512), 100 * 1024, s, out, iters, 0.7f); break;
    case 8: hipLaunchKernelGGL(k_mix<8>, dim3(blocks), dim3(512), 100 * 1024, s, out, iters, 0.7f); break;
    case 9: hipLaunchKernelGGL(k_mix<9>, dim3(blocks), dim3(512), 100 * 1024, s, out, iters, 0.7f); break;
    case 10: hipLaunchKernelGGL(k_mix<10>, dim3(blocks), dim3(512), 100 * 1024, s, out, iters, 0.7f); break;
    default: hipLaunchKernelGGL(k_mix<4>, dim3(blocks), dim3(512), 100 * 1024, s, out, iters, 0.7f); break;
  }
  return (int)hipGetLastError();
}
