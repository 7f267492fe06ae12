// Measurement aid, not part of the training path: the dense-fp16 matrix rate this chip SUSTAINS -- every SIMD issuing independent
// v_mfma_f32_32x32x16_f16 back to back on non-trivial register operands (operand toggling matters for power).  bench.py reports it next to
// the 2.5 PFLOP/s figure of MI355X_MICROARCH.md: under nothing but matrix work the part settles at ~1.5-1.75 GHz, not at its 2.4 GHz peak clock.
#include "common.h"
#include "../../include/textboost_hip.h"

namespace {
__global__ __launch_bounds__(256) void mfma_peak_kernel(float* __restrict__ out, int iters) {
  f32x16 acc[8];
  for (int k = 0; k < 8; ++k)
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  f16x8 a[2], b[2];
  unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
  for (int q = 0; q < 2; ++q)
    for (int e = 0; e < 8; ++e) {
      h = h * 1664525u + 1013904223u;
      a[q][e] = (f16)(((int)(h >> 20) - 2048) * (1.f / 4096.f));
      h = h * 1664525u + 1013904223u;
      b[q][e] = (f16)(((int)(h >> 20) - 2048) * (1.f / 4096.f));
    }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k & 7] = TB_MFMA_32x32x16(a[k & 1], b[(k >> 1) & 1], acc[k & 7]);
  }
  float s = 0.f;
  for (int k = 0; k < 8; ++k)
    for (int r = 0; r < 16; ++r) s += acc[k][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
}  // namespace

extern "C" int tb_mfma_peak_probe(float* out, int blocks, int iters, tb_stream_t stream) {
  (void)hipGetLastError();
  if (!out || blocks <= 0 || iters <= 0) return TB_EINVAL;
  hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
  TB_CHECK_LAUNCH();
  return TB_OK;
}
