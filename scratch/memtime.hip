#include <hip/hip_runtime.h>
__global__ void spin(unsigned long long ticks, unsigned long long* out) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long t1 = t0, c0 = clock64();
  while (t1 - t0 < ticks) t1 = __builtin_amdgcn_s_memtime();
  out[0] = t1 - t0; out[1] = clock64() - c0; out[2] = wall_clock64();
}
extern "C" int run_spin(unsigned long long ticks, unsigned long long* out, void* s) { hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, (hipStream_t)s, ticks, out); return (int)hipGetLastError(); }
