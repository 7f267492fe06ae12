// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of the TextBoost hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 f16;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
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
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad_f(float x) {
  return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}
__device__ __forceinline__ float quick_gelu_f(float x) { return x / (1.f + __expf(-1.702f * x)); }
__device__ __forceinline__ float quick_gelu_grad_f(float x) {
  float s = 1.f / (1.f + __expf(-1.702f * x));
  return s * (1.f + 1.702f * x * (1.f - s));
}
