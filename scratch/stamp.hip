#include <hip/hip_runtime.h>
// device-side wall-clock stamps (100 MHz constant counter) placed between the launches of a captured step: a timeline without the tracer
__global__ void stamp_kernel(unsigned long long* out) { if (threadIdx.x == 0) out[0] = wall_clock64(); }
extern "C" int run_stamp(unsigned long long* out, void* s) { hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, out); return (int)hipGetLastError(); }
