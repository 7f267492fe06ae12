// Round-6 probe: can ONE wave per SIMD with a 128 x 80 tile (13 fragment reads per 40 MFMAs, reads of the next k-step between the MFMAs of this
// one) keep the matrix pipe fed out of the LDS, and what do two waves per SIMD with 64 x 80 tiles (9 reads per 20 MFMAs each) reach without the
// phase structure?  One block per CU, shader clock by s_memtime.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define MF(acc, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))


#include <utility>
template <class F, int... I> __device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void sfor(F&& f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }
__host__ __device__ constexpr int cnt_reads(int mi, int nr, int span) { int c = mi * nr / span + 1; return c > nr ? nr : c; }

template <int R, bool ON, int S, int MT, int NT, int PITCH>
__device__ __forceinline__ void rd1(f16x8 (&af)[2][MT], f16x8 (&bf)[2][NT], unsigned a0, unsigned b0) {
  if constexpr (ON && R < MT) RD(af[S][R], a0, S * 64 + R * 16 * PITCH);
  if constexpr (ON && R >= MT && R < MT + NT) RD(bf[S][R - MT], b0, S * 64 + (R - MT) * 16 * PITCH);
}

// MT x NT fragments of 16 x 16, k = 32 per step; MODE bit 0: MFMAs, bit 1: fragment reads; PITCH: LDS row pitch in bytes (64 k-halfs + pad)
template <int MT, int NT, int MODE, int WAVES, int PITCH>
__global__ __launch_bounds__(WAVES * 64, 1) void k(float* out, unsigned long long* cyc, int iters, int fill, const char* __restrict__ gsrc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 150 * 1024 / 2; i += WAVES * 64) {   // zeros or N(0,1)-like halfs: the matrix pipe's power (and so the clock) depends on the data
    const unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
    const float u = ((int)(h >> 20) % 2048 - 1024) * (1.f / 512.f);
    ((f16*)smem)[i] = fill == 0 ? (f16)0.f : fill == 1 ? (f16)u : fill == 2 ? (f16)(u * 0.01f) : fill == 3 ? (f16)1.f : (f16)((float)((const f16*)gsrc)[(i + blockIdx.x * 4099) & 0x3fffff] * (fill == 5 && i >= 128 * 144 ? 0.0108f : 1.f));
  }
  __syncthreads();
  f32x4 acc[MT][NT];
  for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
  f16x8 af[2][MT], bf[2][NT];
  for (int s = 0; s < 2; ++s) {
    for (int i = 0; i < MT; ++i) af[s][i] = *(const f16x8*)(smem + (s * 13 + i) * 1024 + lane * 16);
    for (int j = 0; j < NT; ++j) bf[s][j] = *(const f16x8*)(smem + (s * 13 + 8 + j) * 1024 + lane * 16);
  }
  // A rows: the wave's MT x 16 pixel rows of a 256-row image; B rows: its NT x 16 channels of a 160-row weight tile behind it
  const int wm = WAVES == 4 ? (wave >> 1) : (wave & 3), wn = WAVES == 4 ? (wave & 1) : ((wave >> 2) & 1);
  unsigned a0 = (wm * MT * 16 + (lane & 15)) * PITCH + (lane >> 4) * 16;
  unsigned b0 = 256 * PITCH + (wn * NT * 16 + (lane & 15)) * PITCH + (lane >> 4) * 16;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
  constexpr int NP = 24 / WAVES;   // LDS-DMA pieces (1 KB each) per wave and tap: 20 KB of weights + ~5 KB of halo per tap and CU
  const char* gp = gsrc + (size_t)(blockIdx.x & 1) * 921600 + wave * 1024 + lane * 16;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)smem;
  f16x8 stage[NP];
  for (int q = 0; q < NP; ++q) stage[q] = af[0][0];
  for (int it = 0; it < iters; ++it) {
    const int tapi = it % 45;
    sfor<2>([&](auto sc) {   // two k = 32 steps of a 64-wide chunk; fragments of the other step are read between this one's MFMAs
      constexpr int s = decltype(sc)::value;
      constexpr int NR = MT + NT, NM = MT * NT, SPAN = NM * 3 / 4;
      if constexpr (s == 1 && (MODE & 12)) {
        if constexpr (MODE & 8) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      sfor<NM>([&](auto mc) {
        constexpr int mi = decltype(mc)::value, i = mi / NT, j = mi % NT;
        if (MODE & 1) MF(acc[i][j], af[s][i], bf[s][j]);
        if constexpr (s == 1 && (MODE & 16) && mi < NP) {
          const char* p = gp + (size_t)tapi * 20480 + (mi * WAVES) * 1024;
          const unsigned dst = lds0 + 100 * 1024 + (tapi % 3) * 16384 + (mi * WAVES + wave) * 1024 + lane * 16;
          if (mi == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          asm volatile("ds_write_b128 %0, %1" ::"v"(dst), "v"(stage[mi]) : "memory");
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(stage[mi]) : "v"(p) : "memory");
        }
        if constexpr (s == 1 && (MODE & 8) && mi < NP) {
          const char* p = gp + (size_t)tapi * 20480 + (mi * WAVES) * 1024;
          const unsigned dst = lds0 + 100 * 1024 + (tapi % 3) * 16384 + (mi * WAVES + wave) * 1024;
          const unsigned m = __builtin_amdgcn_readfirstlane(dst);
          asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(p), "s"(m) : "memory", "m0");
        }
        constexpr int lo = mi == 0 ? 0 : cnt_reads(mi - 1, NR, SPAN), hi = cnt_reads(mi, NR, SPAN);
        if constexpr (MODE & 2) {
          rd1<lo, lo < hi, s ^ 1, MT, NT, PITCH>(af, bf, a0, b0);
          rd1<lo + 1, lo + 1 < hi, s ^ 1, MT, NT, PITCH>(af, bf, a0, b0);
          rd1<lo + 2, lo + 2 < hi, s ^ 1, MT, NT, PITCH>(af, bf, a0, b0);
        }
      });
    });
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
  float sum = 0.f;
  for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) for (int r = 0; r < 4; ++r) sum += acc[i][j][r];
  for (int q = 0; q < NP; ++q) sum += (float)stage[q][0];
  for (int s = 0; s < 2; ++s) { for (int i = 0; i < MT; ++i) sum += (float)af[s][i][0]; for (int j = 0; j < NT; ++j) sum += (float)bf[s][j][0]; }
  out[blockIdx.x * 512 + threadIdx.x] = sum;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0, cyc[1] = w1 - w0;
}
#define CASE(ID, MT, NT, MODE, W, P) \
  case ID: hipFuncSetAttribute((const void*)k<MT, NT, MODE, W, P>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); \
    hipLaunchKernelGGL((k<MT, NT, MODE, W, P>), dim3(blocks), dim3(W * 64), 150 * 1024, s, out, cyc, iters, fill, gsrc); break;
extern "C" int ldsmfma(int id, float* out, unsigned long long* cyc, int blocks, int iters, int fill, const char* gsrc, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  switch (id) {
    CASE(0, 8, 5, 1, 4, 144) CASE(1, 8, 5, 2, 4, 144) CASE(2, 8, 5, 3, 4, 144)
    CASE(3, 4, 5, 1, 8, 144) CASE(4, 4, 5, 2, 8, 144) CASE(5, 4, 5, 3, 8, 144)
    CASE(6, 8, 5, 7, 4, 144) CASE(7, 8, 5, 15, 4, 144) CASE(8, 4, 5, 7, 8, 144) CASE(9, 4, 5, 15, 8, 144) CASE(10, 4, 5, 13, 8, 144) CASE(11, 8, 5, 13, 4, 144) CASE(12, 4, 5, 23, 8, 144) CASE(13, 8, 5, 23, 4, 144)
    default: return -1;
  }
  return (int)hipGetLastError();
}
