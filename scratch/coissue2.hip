// Probe 2: WHEN do one SIMD's two waves overlap matrix and vector work?  (waves w and w+4 of a 512-thread block share a SIMD)
//   hypothesis: a wave whose NEXT instruction is an MFMA waiting for the busy matrix pipe holds the SIMD's VALU issue port, so a partner's
//   VALU only gets in when the MFMA wave has something else (fillers, s_nop) between its MFMAs.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define MF(ACC) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b));
#define FMA(X) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(X) : "v"(c), "v"(d));
#define EXP(X) asm volatile("v_exp_f32 %0, %0" : "+v"(X));
#define NOP16 asm volatile("s_nop 7\n\ts_nop 7");
#define NOP8 asm volatile("s_nop 7");

// 28 MFMAs per iteration on 4 rotating accumulators, K fillers (plain fma on 8 rotating registers) after each
template <int K, int NOPS, bool EXPF>
__device__ __forceinline__ void mfma_iter(f32x16 (&acc)[4], const f16x8& a, const f16x8& b, float (&v)[8], float c, float d) {
#pragma unroll
  for (int i = 0; i < 28; ++i) {
    MF(acc[i & 3])
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (EXPF) { EXP(v[(i * K + k) & 7]) } else { FMA(v[(i * K + k) & 7]) }
    }
    if (NOPS == 1) { NOP8 }
    if (NOPS == 2) { NOP16 }
  }
}
template <int N, bool EXPF>
__device__ __forceinline__ void valu_iter(float (&v)[8], float c, float d) {  // N plain VALU
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (EXPF) { EXP(v[i & 7]) } else { FMA(v[i & 7]) }
  }
}
// ROLE0 / ROLE1: what waves 0-3 / 4-7 do.  role encoding: 0 idle, 1 MFMA b2b, 2 MFMA + nop8, 3 MFMA + nop16, 4..: MFMA + (role-4) fma fillers,
// 20: 168 plain VALU, 21: 168 exp, 22: 336 plain VALU, 30+k: MFMA + k exp fillers
template <int ROLE>
__device__ __forceinline__ void role_iter(f32x16 (&acc)[4], const f16x8& a, const f16x8& b, float (&v)[8], float c, float d) {
  if constexpr (ROLE == 1) mfma_iter<0, 0, false>(acc, a, b, v, c, d);
  else if constexpr (ROLE == 2) mfma_iter<0, 1, false>(acc, a, b, v, c, d);
  else if constexpr (ROLE == 3) mfma_iter<0, 2, false>(acc, a, b, v, c, d);
  else if constexpr (ROLE >= 4 && ROLE < 20) mfma_iter<ROLE - 4, 0, false>(acc, a, b, v, c, d);
  else if constexpr (ROLE == 20) valu_iter<168, false>(v, c, d);
  else if constexpr (ROLE == 21) valu_iter<168, true>(v, c, d);
  else if constexpr (ROLE == 22) valu_iter<336, false>(v, c, d);
  else if constexpr (ROLE >= 30) mfma_iter<ROLE - 30, 0, true>(acc, a, b, v, c, d);
}
template <int ROLE0, int ROLE1, int PRIO0, int PRIO1>
__global__ __launch_bounds__(512) void k2(float* out, int iters, float c, float d) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x16 acc[4];
  for (int k = 0; k < 4; ++k)
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (f16)(lane * 0.01f + e); b[e] = (f16)(0.5f - e * 0.1f); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = lane * 0.001f + i * 0.01f;
  if (wave < 4) {
    if (PRIO0) __builtin_amdgcn_s_setprio(PRIO0);
    for (int it = 0; it < iters; ++it) role_iter<ROLE0>(acc, a, b, v, c, d);
  } else {
    if (PRIO1) __builtin_amdgcn_s_setprio(PRIO1);
    for (int it = 0; it < iters; ++it) role_iter<ROLE1>(acc, a, b, v, c, d);
  }
  float s = 0.f;
  for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
#define CASE(ID, R0, R1, P0, P1)                                                                                           \
  case ID: {                                                                                                               \
    hipFuncSetAttribute((const void*)k2<R0, R1, P0, P1>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);          \
    hipLaunchKernelGGL((k2<R0, R1, P0, P1>), dim3(blocks), dim3(512), 100 * 1024, s, out, iters, 0.999f, 0.001f);          \
    break;                                                                                                                 \
  }
extern "C" int coissue2(int id, float* out, int blocks, int iters, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  switch (id) {
    CASE(0, 1, 0, 0, 0)     // MFMA b2b | idle
    CASE(1, 0, 20, 0, 0)    // idle | 168 fma
    CASE(2, 1, 20, 0, 0)    // MFMA b2b | 168 fma
    CASE(3, 2, 0, 0, 0)     // MFMA+nop8 | idle
    CASE(4, 2, 20, 0, 0)    // MFMA+nop8 | 168 fma
    CASE(5, 3, 0, 0, 0)     // MFMA+nop16 | idle
    CASE(6, 3, 20, 0, 0)    // MFMA+nop16 | 168 fma
    CASE(7, 8, 0, 0, 0)     // MFMA + 4 fma fillers | idle
    CASE(8, 10, 0, 0, 0)    // MFMA + 6 fillers | idle
    CASE(9, 12, 0, 0, 0)    // MFMA + 8 fillers | idle
    CASE(10, 10, 10, 0, 0)  // both: MFMA + 6 fillers
    CASE(11, 8, 8, 0, 0)    // both: MFMA + 4 fillers
    CASE(12, 7, 7, 0, 0)    // both: MFMA + 3 fillers
    CASE(13, 1, 20, 0, 3)   // MFMA b2b | 168 fma at prio 3
    CASE(14, 1, 20, 3, 0)   // MFMA b2b prio 3 | 168 fma
    CASE(15, 1, 21, 0, 0)   // MFMA b2b | 168 exp
    CASE(16, 0, 21, 0, 0)   // idle | 168 exp
    CASE(17, 34, 0, 0, 0)   // MFMA + 4 exp fillers | idle
    CASE(18, 36, 0, 0, 0)   // MFMA + 6 exp fillers | idle
    CASE(19, 1, 1, 0, 0)    // both MFMA b2b
    CASE(20, 6, 6, 0, 0)    // both: MFMA + 2 fillers
    CASE(21, 1, 22, 0, 0)   // MFMA b2b | 336 fma
    CASE(22, 3, 22, 0, 0)   // MFMA+nop16 | 336 fma
    CASE(23, 6, 22, 0, 0)   // MFMA + 2 fillers | 336 fma
    default: return -1;
  }
  return (int)hipGetLastError();
}
