// Activation-stationary Linear for the short-K layers of the 64x64 maps (K = C = 320: attn1 qkv, attn2.to_q, proj_out and the K = 320 dgrads of the
// SD1.x transformer blocks behind train_textboost.py:1063-1067 / :1108), gfx950.  tb_gemm routes to it (tb_lin320_try); no entry point of its own.
//
// The 128 x 320 tiles of gemm8.hip spend their launch on five k-steps, each one exposed LDS-DMA round trip for 56 KB of operands, between a
// prologue and an epilogue (23 us for 6.7 GFLOP, 52 us for the N = 960 qkv projection).  Here -- the phase-A half of csrc/ff_fused.hip without the
// second product -- a workgroup (8 waves, one per CU, M / 128 of them) keeps its 128 x 320 activation tile in REGISTERS as MFMA operand fragments
// (wave (wm, wn): rows 32 wm .. + 31, all 320 k: 80 VGPRs, loaded once) and walks the N output columns in tiles of 64: the weight tile
// (64 rows x 320 k = 40 KB, five [64][128 B] slabs, chunk ^= row & 7) arrives by LDS-DMA TWO tiles ahead in a 3-slot ring, a tile is 40
// v_mfma_f32_16x16x32_f16 per wave behind ONE barrier, and its 32 x 32 outputs per wave leave straight from the accumulators (+ bias + residual,
// fp16, 8 bytes per lane).  Activations are read once, nothing but the weight stream goes through the LDS.
#include "gemm_epi.h"
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;
constexpr int L3_BM = 128, L3_K = 320, L3_TN = 64, L3_TILE = L3_TN * L3_K * 2, L3_SLAB = L3_TN * 128, L3_NS = 3;   // 40 KB weight tile, 3 slots
constexpr int L3_NI = L3_TILE / 1024 / 8;   // LDS-DMA instructions per wave and tile: 40 / 8 = 5
constexpr int L3_LDS = L3_NS * L3_TILE;

template <int OFF>
__device__ __forceinline__ f16x8 l3_read16(uint32_t addr) {
  f16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int N>
__device__ __forceinline__ void l3_wait_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
template <int I>
using l3c = std::integral_constant<int, I>;
template <int N, class F>
__device__ __forceinline__ void l3_unroll(F&& f) {
  if constexpr (N > 0) {
    l3_unroll<N - 1>(f);
    f(l3c<N - 1>{});
  }
}

template <bool HAS_R>
__global__ __launch_bounds__(512, 2) void lin320_kernel(const tb_gemm_desc p, int ntile) {
  extern __shared__ __attribute__((aligned(128))) unsigned char smem_raw[];
  const int t = threadIdx.x, lane = t & 63, l15 = lane & 15, lq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.x * L3_BM;

  // ---- the register-resident operand: rows 32 wm + 16 i + l15, k = 32 ks + 8 lq .. + 7 (requested first: every later wait covers it)
  f16x8 xf[2][10];
  {
    const f16* xr = (const f16*)p.A + (m0 + wm * 32 + l15) * p.lda + 8 * lq;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 10; ++ks) xf[i][ks] = *(const f16x8*)(xr + (int64_t)i * 16 * p.lda + 32 * ks);
  }
  // ---- LDS-DMA pieces of this wave: q = wave + 8 k (k = 0..4) of a tile's 40: slab s = q / 8, rows 8 (q % 8) .. + 7
  uint32_t d_off[L3_NI];
  int d_dst[L3_NI];
#pragma unroll
  for (int k = 0; k < L3_NI; ++k) {
    const int q = wave + 8 * k;
    const int s = q >> 3, rgp = q & 7;
    const int row = rgp * 8 + (lane >> 3), cc = (lane & 7) ^ (lane >> 3);
    d_off[k] = (uint32_t)(((int64_t)row * p.ldw + s * 64 + cc * 8) * 2);
    d_dst[k] = s * L3_SLAB + rgp * 1024;
  }
  const int64_t tile_stride = (int64_t)L3_TN * p.ldw * 2;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem_raw;
  auto issue_piece = [&](int tile, int slot, auto kc) {   // M0 = LDS address of the piece, then the load (inline asm: see ff_fused.hip)
    constexpr int k = decltype(kc)::value;
    const int tl = tile < ntile ? tile : ntile - 1;        // (past the end: a re-fetch nobody reads, keeps the counted waits uniform)
    const char* src = (const char*)p.W + tl * tile_stride + d_off[k];
    const uint32_t m = __builtin_amdgcn_readfirstlane(lds0 + slot * L3_TILE + d_dst[k]);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m), "v"(src) : "memory", "m0");
  };
  l3_unroll<L3_NI>([&](auto kc) { issue_piece(0, 0, kc); });
  l3_unroll<L3_NI>([&](auto kc) { issue_piece(1, 1, kc); });

  const int swz8 = l15 & 7;
  const uint32_t fa0 = lds0 + (wn * 32 + l15) * 128 + ((lq ^ swz8) << 4);   // weight row wn 32 + 16 j + l15, even ks; odd ks: ^ 64; slab (ks >> 1) * L3_SLAB
  const int64_t mrow = m0 + wm * 32 + l15;
  const int ncol = wn * 32 + 4 * lq;
  f16* const y_lane = (f16*)p.C + mrow * p.ldc + ncol;
  const f16* const r_lane = HAS_R ? (const f16*)p.R + mrow * p.ldr + ncol : nullptr;
  const float* const b_lane = p.bias ? p.bias + ncol : nullptr;
  const float* const zero4 = reinterpret_cast<const float*>(g_zero_line);   // (no bias: the loads below read zeros -- every wave issues the same count)
  const float alpha = p.alpha;

  // Barrier t releases tile t (every wave's pieces of it have landed) and the slot of tile t + 2 (last read in tile t - 1).  Per tile a wave
  // issues, in this order, [HAS_R: 4 residual loads] 2 bias loads, 5 pieces of tile t + 2, 4 output stores; the residual / bias loads are waited
  // for inside the tile (younger: the 5 pieces).  At barrier t the pieces of tile t (issued during tile t - 2) must have landed -- younger and
  // possibly outstanding: tile t - 2's stores, tile t - 1's pieces and stores = 13 (vector-memory operations retire in issue order).
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tiles 0 and 1 and the operand fragments (the counted wait assumes the steady state)
  int slot = 0;
  for (int tile = 0; tile < ntile; ++tile) {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(4 + L3_NI + 4) : "memory");
    // inline asm loads + ONE unconditional counted wait below (see ff_fused.hip: a plain load's compiler-inserted wait is vmcnt(0), which would
    // drain the prefetched weight tiles every step)
    f16x4 rv[2][2];
    if (HAS_R) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const f16* src = r_lane + (int64_t)i * 16 * p.ldr + tile * L3_TN + 16 * j;
          asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(rv[i][j]) : "v"(src) : "memory");
        }
    }
    f32x4_t bias[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float* src = b_lane ? b_lane + tile * L3_TN + 16 * j : zero4;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bias[j]) : "v"(src) : "memory");
    }
    const int nslot = slot == 0 ? 2 : slot - 1;   // slot of tile + 2 == slot of tile - 1
    const uint32_t a0 = fa0 + slot * L3_TILE, a1 = a0 ^ 64;
    f32x4_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    f16x8 wf[3][2];   // fragment ring: reads run two k-steps ahead of their MFMAs
    auto rd = [&](auto ksc) {
      constexpr int ks = decltype(ksc)::value;
      constexpr int off = (ks >> 1) * L3_SLAB;
      wf[ks % 3][0] = l3_read16<off>((ks & 1) ? a1 : a0);
      wf[ks % 3][1] = l3_read16<off + 16 * 128>((ks & 1) ? a1 : a0);
    };
    rd(l3c<0>{});
    rd(l3c<1>{});
    __builtin_amdgcn_sched_barrier(0);
    l3_unroll<10>([&](auto ksc) {
      constexpr int ks = decltype(ksc)::value;
      if constexpr (ks + 2 < 10) rd(l3c<ks + 2>{});
      constexpr int later = (ks + 2 < 10 ? 2 : 0) + (ks + 1 < 10 ? 2 : 0);
      l3_wait_lgkm<later>();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][j] = TB_MFMA_16x16x32(wf[ks % 3][j], xf[i][ks], acc[i][j]);
      if constexpr (ks < L3_NI) issue_piece(tile + 2, nslot, l3c<(ks < L3_NI ? ks : 0)>{});   // one piece per k-step, behind the step's MFMAs
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (HAS_R)
      asm volatile("s_waitcnt vmcnt(%6)" : "+v"(rv[0][0]), "+v"(rv[0][1]), "+v"(rv[1][0]), "+v"(rv[1][1]), "+v"(bias[0]), "+v"(bias[1]) : "n"(L3_NI));
    else
      asm volatile("s_waitcnt vmcnt(%2)" : "+v"(bias[0]), "+v"(bias[1]) : "n"(L3_NI));
    // ---- this lane's (row 16 i + l15, columns 16 j + 4 lq .. + 3) units: tb_gemm's epilogue arithmetic alpha * acc + bias + R
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc[i][j][e] * alpha + bias[j][e];
          if (HAS_R) v += (float)rv[i][j][e];
          o[e] = (f16)v;
        }
        *(f16x4*)(y_lane + (int64_t)i * 16 * p.ldc + tile * L3_TN + 16 * j) = o;
      }
    slot = slot == 2 ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tail's dummy prefetches
}

int g_lin320 = 1;   // tb_gemm_set_variant(9400 + {0,1})

}  // namespace

void tb_lin320_set(int on) { g_lin320 = on; }

// 1 = shape / epilogue not covered (tb_gemm goes on with its other kernels); TB_OK = launched
int tb_lin320_try(const tb_gemm_desc& d, hipStream_t s) {
  if (!g_lin320) return 1;
  if (d.a_mode != TB_A_LINEAR || d.K != L3_K || d.A2 || d.W2 || d.M % L3_BM || d.N % L3_TN || d.N < 128) return 1;
  if (d.M / L3_BM < 200) return 1;                                  // one workgroup per CU: the 64x64 maps at the metric batch
  if (d.act != TB_ACT_NONE || d.rowbias || d.C2 || d.c_dtype != TB_F16 || (d.R && d.r_dtype != TB_F16) || d.rs_out || d.rs_in) return 1;
  if (d.lda % 8 || d.ldw % 8 || d.ldc % 4 || (d.R && d.ldr % 4)) return 1;
  if (((uintptr_t)d.A) % 16 || ((uintptr_t)d.W) % 16 || ((uintptr_t)d.C) % 8 || ((uintptr_t)d.R) % 8 || ((uintptr_t)d.bias) % 16) return 1;
  if ((d.N * d.ldw) * 2 >= ((int64_t)1 << 32)) return 1;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)lin320_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, L3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)lin320_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, L3_LDS) != hipSuccess)
      return TB_ELAUNCH;
    attr_done = true;
  }
  const int ntile = (int)(d.N / L3_TN);
  if (d.R) hipLaunchKernelGGL(lin320_kernel<true>, dim3((unsigned)(d.M / L3_BM)), dim3(512), L3_LDS, s, d, ntile);
  else hipLaunchKernelGGL(lin320_kernel<false>, dim3((unsigned)(d.M / L3_BM)), dim3(512), L3_LDS, s, d, ntile);
  TB_CHECK_LAUNCH();
  return TB_OK;
}
