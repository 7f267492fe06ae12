// Operand-delivery microbenchmark: bytes per clock per CU of global_load_lds_dwordx4 (LDS-DMA) against global_load_dwordx4 (+ ds_write_b128),
// one 512-thread workgroup per CU, source either L2-resident (small) or streaming.  hipcc --offload-arch=gfx950 -O3 -shared -fPIC
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(4))) float f4;

template <int MODE>
__global__ __launch_bounds__(512) void dma_kernel(const char* __restrict__ src, size_t span, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(128))) unsigned char lds[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // each wave-instruction moves 1 KB: lane -> 16 bytes; consecutive instructions walk the workgroup's private window of the source
  const size_t wg_base = ((size_t)blockIdx.x * 8 + wave) * 1024;
  f4 acc = {0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * 8 * 1024;
#pragma unroll 8
  for (int i = 0; i < iters; ++i) {
    const size_t off = (wg_base + (size_t)i * stride) & (span - 1);
    const char* p = src + off + lane * 16;
    if (MODE == 0) {
      __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(lds + ((wave * 16 + (i & 15)) * 1024)), 16, 0, 0);
    } else {
      const f4 v = *(const f4*)p;
      if (MODE == 1) acc += v;
      else *(f4*)(lds + ((wave * 16 + (i & 15)) * 1024) + lane * 16) = v;
    }
    if ((i & 7) == 7) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // keep ~16 loads in flight per wave
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (MODE != 1) acc[0] = ((float*)lds)[t];
  if (acc[0] == 12345.f) sink[t] = acc[0] + acc[1] + acc[2] + acc[3];
}

extern "C" int dma_run(int mode, const void* src, size_t span, int iters, int blocks, void* sink, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const size_t l = 128 * 1024;
  if (mode == 0) { hipFuncSetAttribute((const void*)dma_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, l); hipLaunchKernelGGL(dma_kernel<0>, dim3(blocks), dim3(512), l, s, (const char*)src, span, iters, (float*)sink); }
  else if (mode == 1) { hipFuncSetAttribute((const void*)dma_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, l); hipLaunchKernelGGL(dma_kernel<1>, dim3(blocks), dim3(512), l, s, (const char*)src, span, iters, (float*)sink); }
  else { hipFuncSetAttribute((const void*)dma_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, l); hipLaunchKernelGGL(dma_kernel<2>, dim3(blocks), dim3(512), l, s, (const char*)src, span, iters, (float*)sink); }
  return (int)hipGetLastError();
}
