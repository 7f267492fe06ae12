// Scratch probe: validates toolchain + MFMA fragment layout assumptions on a real gfx950.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void k_add(const float* a, const float* b, float* c, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) c[i] = a[i] + b[i];
}
// C[32][32] = A[32][16] * B[32][16]^T
__global__ void k_mfma32(const f16* A, const f16* B, float* C) {
  int l = threadIdx.x;
  f16x8 a = *(const f16x8*)(A + (l & 31) * 16 + 8 * (l >> 5));
  f16x8 b = *(const f16x8*)(B + (l & 31) * 16 + 8 * (l >> 5));
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    int col = l & 31;
    C[row * 32 + col] = c[r];
  }
}
// C[16][16] = A[16][32] * B[16][32]^T
__global__ void k_mfma16(const f16* A, const f16* B, float* C) {
  int l = threadIdx.x;
  f16x8 a = *(const f16x8*)(A + (l & 15) * 32 + 8 * (l >> 4));
  f16x8 b = *(const f16x8*)(B + (l & 15) * 32 + 8 * (l >> 4));
  f32x4 c = {0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) {
    int row = 4 * (l >> 4) + r;
    int col = l & 15;
    C[row * 16 + col] = c[r];
  }
}
extern "C" int probe_add(const float* a, const float* b, float* c, int n, hipStream_t s) {
  hipLaunchKernelGGL(k_add, dim3((n + 255) / 256), dim3(256), 0, s, a, b, c, n);
  return (int)hipGetLastError();
}
extern "C" int probe_mfma32(const void* A, const void* B, float* C, hipStream_t s) {
  hipLaunchKernelGGL(k_mfma32, dim3(1), dim3(64), 0, s, (const f16*)A, (const f16*)B, C);
  return (int)hipGetLastError();
}
extern "C" int probe_mfma16(const void* A, const void* B, float* C, hipStream_t s) {
  hipLaunchKernelGGL(k_mfma16, dim3(1), dim3(64), 0, s, (const f16*)A, (const f16*)B, C);
  return (int)hipGetLastError();
}
extern "C" int probe_props(int* out) {
  hipDeviceProp_t p; int e = hipGetDeviceProperties(&p, 0);
  out[0] = p.multiProcessorCount; out[1] = p.clockRate; out[2] = (int)(p.totalGlobalMem >> 30);
  out[3] = (int)p.sharedMemPerBlock; out[4] = p.maxSharedMemoryPerMultiProcessor; out[5] = p.warpSize;
  return e;
}
