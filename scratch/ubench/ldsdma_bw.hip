// L2 -> CU load-path throughput on gfx950: LDS-DMA (global_load_lds_dwordx4) against global_load_dwordx4 into VGPRs (+ ds_write_b128), per CU, as a function of
// waves per CU and loads in flight per wave.  Every workgroup streams the SAME 64-row x K operand panel pattern a GEMM tile does: each load instruction fetches
// 8 rows x 128 B (LDS-DMA form) from a buffer that fits the L2 (warm).  build: hipcc --offload-arch=gfx950 -O3 -o ldsdma_bw ldsdma_bw.hip ; run: ./ldsdma_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int DEPTH, int MODE>   // MODE 0: LDS-DMA, 1: VGPR load + ds_write, 2: VGPR load only
__global__ __launch_bounds__(512) void bw_kernel(const char* __restrict__ src, int64_t row_bytes, int rows_total, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  // a "piece" = 8 rows x 128 B; wave w takes pieces w, w + nw, ... of each k-step's (rows_per_block x 128 B) slab
  const int rows_per_block = nw * 8 * DEPTH;                       // one piece per wave per slot
  const int64_t r0 = ((int64_t)blockIdx.x * rows_per_block) % rows_total;
  const int rl = lane >> 3, cp = lane & 7;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int ksteps = (int)(row_bytes / 128);
  for (int it = 0; it < iters; ++it) {
    const int kt = it % ksteps;
    f32x4 v[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int64_t row = (r0 + (d * nw + wave) * 8 + rl) % rows_total;
      const char* p = src + row * row_bytes + (int64_t)kt * 128 + cp * 16;
      unsigned char* dst = smem + ((d * nw + wave) * 8) * 128;
      if (MODE == 0) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      } else {
        v[d] = *(const f32x4*)p;
      }
    }
    if (MODE == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        if (MODE == 1) *(f32x4*)(smem + ((d * nw + wave) * 8 + rl) * 128 + cp * 16) = v[d];
        else acc += v[d];
      }
      if (MODE == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  if (MODE != 2) acc[0] = ((float*)smem)[threadIdx.x];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}

template <int DEPTH, int MODE>
void run(const char* name, const char* src, int64_t row_bytes, int rows_total, float* sink, int waves, int blocks_per_cu) {
  const int cus = 256, iters = 2000;
  const int blocks = cus * blocks_per_cu;
  const size_t lds = (size_t)waves * 8 * DEPTH * 128;
  hipFuncSetAttribute((const void*)bw_kernel<DEPTH, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  bw_kernel<DEPTH, MODE><<<blocks, waves * 64, lds>>>(src, row_bytes, rows_total, 200, sink);
  hipDeviceSynchronize();
  hipEventRecord(a);
  bw_kernel<DEPTH, MODE><<<blocks, waves * 64, lds>>>(src, row_bytes, rows_total, iters, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)blocks * waves * DEPTH * 1024.0 * iters;
  printf("%-28s waves/blk %d blk/CU %d depth %d : %7.2f TB/s  %6.1f B/clk/CU (2.4 GHz)  %6.1f clk per 1 KB instr per CU\n", name, waves, blocks_per_cu, DEPTH,
         bytes / ms / 1e9, bytes / ms / 1e6 / 256 / 2.4e3 * 1e-3 * 1e3 / 1.0, 1024.0 / (bytes / ms / 1e6 / 256 / 2.4e3));
}

int main() {
  const int64_t row_bytes = 2560;   // K = 1280 halfs
  const int rows_total = 3328;      // 2048 + 1280 rows: 8.5 MB
  char* src; float* sink;
  hipMalloc(&src, row_bytes * rows_total); hipMemset(src, 1, row_bytes * rows_total);
  hipMalloc(&sink, 64);
  for (int pass = 0; pass < 2; ++pass) {
    if (pass) printf("---- second pass\n");
    run<1, 0>("LDS-DMA", src, row_bytes, rows_total, sink, 4, 2);
    run<2, 0>("LDS-DMA", src, row_bytes, rows_total, sink, 4, 2);
    run<4, 0>("LDS-DMA", src, row_bytes, rows_total, sink, 4, 2);
    run<4, 0>("LDS-DMA", src, row_bytes, rows_total, sink, 4, 3);
    run<8, 0>("LDS-DMA", src, row_bytes, rows_total, sink, 4, 2);
    run<4, 0>("LDS-DMA", src, row_bytes, rows_total, sink, 8, 1);
    run<8, 0>("LDS-DMA", src, row_bytes, rows_total, sink, 8, 1);
    run<8, 0>("LDS-DMA", src, row_bytes, rows_total, sink, 8, 2);
    run<1, 1>("VGPR load + ds_write", src, row_bytes, rows_total, sink, 4, 2);
    run<4, 1>("VGPR load + ds_write", src, row_bytes, rows_total, sink, 4, 2);
    run<8, 1>("VGPR load + ds_write", src, row_bytes, rows_total, sink, 4, 2);
    run<8, 1>("VGPR load + ds_write", src, row_bytes, rows_total, sink, 8, 1);
    run<8, 1>("VGPR load + ds_write", src, row_bytes, rows_total, sink, 8, 2);
    run<4, 2>("VGPR load only", src, row_bytes, rows_total, sink, 4, 2);
    run<8, 2>("VGPR load only", src, row_bytes, rows_total, sink, 8, 1);
    run<8, 2>("VGPR load only", src, row_bytes, rows_total, sink, 8, 2);
  }
  return 0;
}
