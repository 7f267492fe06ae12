// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of the TextBoost hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The library's 16-bit float: IEEE half by default; -DTB_BF16 builds the SAME sources with bfloat16 operands (libtextboost_hip_bf16.so: the
// reference's --mixed_precision bf16, train_textboost.py:298-308, :930-934 -- bf16 MFMA runs at the fp16 rate on gfx950, fp32 accumulation and
// statistics are unchanged).  `f16` therefore reads "the half type of this build" everywhere below.
#ifdef TB_BF16
typedef __bf16 f16;
#define TB_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define TB_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#else
typedef _Float16 f16;
#define TB_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define TB_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#endif
typedef __attribute__((ext_vector_type(2))) f16 f16x2;
typedef __attribute__((ext_vector_type(4))) f16 f16x4;
typedef __attribute__((ext_vector_type(8))) f16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define TB_OK 0
#define TB_EINVAL (-22)
#define TB_ELAUNCH (-5)

// last HIP error seen by a failed launch check (diagnostics only; see tb_last_hip_error in optim.hip)
extern "C" int tb_last_hip_error_code_;
#define TB_CHECK_LAUNCH()                                   \
  do {                                                      \
    hipError_t e__ = hipGetLastError();                     \
    if (e__ != hipSuccess) {                                \
      tb_last_hip_error_code_ = (int)e__;                   \
      return TB_ELAUNCH;                                    \
    }                                                       \
  } while (0)

// row index of accumulator register r (0..15) of a 32x32 MFMA tile for a lane with hi = lane >> 5
__device__ __forceinline__ int mfma32_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum for blockDim.x == 256 (4 waves); `red` is >= 4 floats of LDS. Result in all threads.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }
__device__ __forceinline__ float silu_grad_f(float x) {
  float s = 1.f / (1.f + __expf(-x));
  return s * (1.f + x * (1.f - s));
}
// Exact (erf) GELU without libm's erff (a two-branch routine of ~120 VALU instructions: the GEGLU epilogues were VALU-bound on it, 8.6 us of a
// 14.5 us tile).  Phi(x) = erfc(-x / sqrt 2) / 2 with the Chebyshev-fitted erfc(z) = t exp(-z^2 + P(t)), t = 1 / (1 + z / 2) of Numerical
// Recipes 6.2 (fractional error < 1.2e-7 for every z >= 0, i.e. relative accuracy also in the tail where 1 + erf cancels): one rcp, one exp
// and nine FMAs.  Against float64: relative error of gelu <= 3e-6 over |x| <= 12; after rounding to fp16 it differs in 158 of 2e6 inputs by 1 ulp.
__device__ __forceinline__ float normal_cdf_f(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(1.f + 0.5f * z);
  float q = 0.17087277f;
  q = q * t - 0.82215223f;
  q = q * t + 1.48851587f;
  q = q * t - 1.13520398f;
  q = q * t + 0.27886807f;
  q = q * t - 0.18628806f;
  q = q * t + 0.09678418f;
  q = q * t + 0.37409196f;
  q = q * t + 1.00002368f;
  q = q * t - 1.26551223f;
  const float half_erfc = 0.5f * t * __expf(q - z * z);
  return x < 0.f ? half_erfc : 1.f - half_erfc;
}
__device__ __forceinline__ float gelu_erf_f(float x) { return x * normal_cdf_f(x); }
__device__ __forceinline__ float gelu_erf_grad_f(float x) { return normal_cdf_f(x) + x * 0.3989422804014327f * __expf(-0.5f * x * x); }
__device__ __forceinline__ void gelu_erf_both_f(float x, float& g, float& dg) {  // gelu(x) and gelu'(x) off one Phi(x)
  const float c = normal_cdf_f(x);
  g = x * c;
  dg = c + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}
__device__ __forceinline__ float quick_gelu_f(float x) { return x / (1.f + __expf(-1.702f * x)); }
__device__ __forceinline__ float quick_gelu_grad_f(float x) {
  float s = 1.f / (1.f + __expf(-1.702f * x));
  return s * (1.f + 1.702f * x * (1.f - s));
}
