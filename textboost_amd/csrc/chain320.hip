// Linear -> LayerNorm -> Linear on the 64x64-map residual stream (C = 320) in ONE launch (round 6), gfx950.
//
// diffusers Transformer2DModel / BasicTransformerBlock on the SD1.x 64x64 maps (train_textboost.py:1063-1067):
//     proj_in (1x1 conv = Linear over NHWC rows) -> norm1 -> attn1.to_q|to_k|to_v          t0 = n0 Wp^T + bp;  qkv = LN1(t0) Wqkv^T
//     attn1.to_out.0 + residual              -> norm2 -> attn2.to_q                        t1 = o1 Wo^T + bo + t0;  q2 = LN2(t1) Wq^T
// As launches these were a Linear tile with the LayerNorm in its epilogue (tb_gemm TB_ACT_LN_FWD: writes t AND LN(t)) followed by an
// activation-stationary Linear (lin320.hip: reads LN(t) back): 30 + 42 us and 26 + 18 us per block at B = 8, of which the matrix work is < 10 us --
// they are bound by the 21 MB activation passes and by two launches' ramps.  A 128-row tile spans the 320-wide rows of all three layers, so here one
// workgroup (8 waves, one per CU; M / 128 = one chip round at the metric batch) keeps its rows on the CU from the first product to the last:
//   stage 1: X (80 registers of MFMA operand fragments) x ten 32-column weight tiles -> + bias + residual -> fp16 into the [128][320] LDS image
//   LayerNorm of the image rows in place; the un-normalised rows leave for T (the residual stream) in whole 640-byte rows, (mean, rstd) for stats
//   image -> operand fragments; stage 2: N2 / 32 weight tiles -> + bias -> fp16 into the image, which leaves for Y in whole rows every 320 columns
// (ff_chain.h: the 20 KB weight tiles stream through a 3-slot LDS ring two tiles ahead with counted waits).  LN(t) never exists in memory.
// Same arithmetic as the launches it replaces: fp16 operands, fp32 accumulation, t rounded to fp16 before the two-pass fp32 LayerNorm statistics.
#include "ff_chain.h"

namespace {

constexpr int CH_B2_OFF = FFC_BIAS_OFF;   // stage 2's bias: up to 2560 floats (this kernel has no packed proj bias there)
#ifndef CH_PROF
#define CH_PROF 0   // profiling build (TB_CFLAGS=-DCH_PROF=1): s_memtime stamps of waves 0 and 4 of workgroup 0 -> tb_chain320_debug buffer
#endif
__device__ unsigned long long* g_ch_dbg = nullptr;
#if CH_PROF
#define CH_PF(k) if (g_ch_dbg && blockIdx.x == 0 && (t == 0 || t == 256)) g_ch_dbg[(t ? 16 : 0) + (k)] = __builtin_amdgcn_s_memtime();
#else
#define CH_PF(k)
#endif

__global__ __launch_bounds__(512, 2) void chain320_kernel(const tb_chain_desc p) {
  extern __shared__ __attribute__((aligned(128))) unsigned char smem_raw[];
  const int t = threadIdx.x, lane = t & 63, l15 = lane & 15, lq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.x * FF_BM;
  const uint32_t lds0 = ff_lds_addr(smem_raw);
  const int64_t row0 = m0 + wm * 32 + l15;
  f16x8 xf[2][10];
  CH_PF(0)
  // ---------------------------------------------------------------- stage 1
  const FfChain ch1(smem_raw, p.W1, p.ldw1, wave, lane);
  ch1.issue(0);
  ch1.issue(1);
  {
    const f16* xr = (const f16*)p.X + row0 * p.ldx + 8 * lq;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 10; ++ks) xf[i][ks] = *(const f16x8*)(xr + (int64_t)i * 16 * p.ldx + 32 * ks);
  }
  // the residual of all ten tiles is requested before the tile loop (inside it only the weight DMA is on the vector-memory queue: loads return in
  // order, so the counted wait of a tile means "this tile has landed"); both stages' biases go through the LDS
  f16x4 rr[10][2];
#pragma unroll
  for (int tl = 0; tl < 10; ++tl)
#pragma unroll
    for (int i = 0; i < 2; ++i)
      rr[tl][i] = p.R1 ? *(const f16x4*)((const f16*)p.R1 + (row0 + 16 * i) * p.ldr1 + tl * 32 + wn * 16 + 4 * lq) : f16x4{0, 0, 0, 0};
  if (t < FF_C) reinterpret_cast<float*>(smem_raw + FFC_CB_OFF)[t] = p.b1 ? p.b1[t] : 0.f;
  ff_chain_stage_gamma_beta(smem_raw, t, p.gamma, p.beta);
  for (int i = t; i < p.N2; i += 512) reinterpret_cast<float*>(smem_raw + CH_B2_OFF)[i] = p.b2 ? p.b2[i] : 0.f;
  const uint32_t fa = lds0 + (wn * 16 + l15) * 128 + ((lq ^ (l15 & 7)) << 4);
  const uint32_t img_w = lds0 + FFC_IMG + (wm * 32 + l15) * FFC_PITCH + (wn * 16 + 4 * lq) * 2;   // + i * 16 rows, + 64 B per tile of a 320-column chunk
  const uint32_t cb1 = lds0 + FFC_CB_OFF + (wn * 16 + 4 * lq) * 4;                                  // + 128 B per tile
  CH_PF(1)
  auto run1 = [&](auto tlc) {   // (unrolled by hand: rr is a register array and must be indexed by constants)
    constexpr int tile = decltype(tlc)::value;
    if constexpr (tile + 1 < 10) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // tile landed for every wave; tile - 1's slot is free
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if constexpr (tile + 2 < 10) ch1.issue(tile + 2);
    FF_SB();
    const f32x4_t pb = ff_read16f<tile * 128>(cb1);   // (waited for with the k-loop's first fragments)
    f32x4_t acc[2];
    ff_chain_tile(fa + (tile % 3) * FFC_WT, xf, acc);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (f16)(acc[i][e] + pb[e] + (float)rr[tile][i][e]);   // tb_gemm's epilogue arithmetic
      if (i == 0) ff_write8<0>(img_w + tile * 64, o);
      else ff_write8<16 * FFC_PITCH>(img_w + tile * 64, o);
    }
  };
  run1(std::integral_constant<int, 0>{});
  CH_PF(2)
  run1(std::integral_constant<int, 1>{}); run1(std::integral_constant<int, 2>{});
  run1(std::integral_constant<int, 3>{}); run1(std::integral_constant<int, 4>{}); run1(std::integral_constant<int, 5>{});
  run1(std::integral_constant<int, 6>{}); run1(std::integral_constant<int, 7>{}); run1(std::integral_constant<int, 8>{});
  run1(std::integral_constant<int, 9>{});
  CH_PF(3)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the image is complete; the weight slots are free
  // ---------------------------------------------------------------- LayerNorm (stage 2's first two weight tiles stream in under it)
  const FfChain ch2(smem_raw, p.W2, p.ldw2, wave, lane);
  const int nt = p.N2 >> 5;
  ch2.issue(0);
  if (nt > 1) ch2.issue(1);
  else ch2.issue(0);   // (keeps "two tiles in flight" for the counted wait of tile 0)
  CH_PF(4)
  ff_chain_layernorm(smem_raw, t, m0, (f16*)p.T, p.ldt, p.eps, p.stats);
  CH_PF(5)
  // the stores of T are on the memory queue BEHIND the two tiles: the first counted wait below also waits for all but three of them -- stricter, not wrong
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LN(t) is in the image
  ff_chain_operand(smem_raw, wm, l15, lq, xf);
  CH_PF(6)
  // ---------------------------------------------------------------- stage 2
  const uint32_t cb2 = lds0 + CH_B2_OFF + (wn * 16 + 4 * lq) * 4;
  for (int tile = 0; tile < nt; ++tile) {
    // (tile 0: the barrier also orders every wave's operand reads of the image in front of the first output write into it)
    if (tile + 1 < nt) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (tile + 2 < nt) ch2.issue(tile + 2);
    FF_SB();
    const int ct = tile % 10;   // tile of the current 320-column chunk
    const f32x4_t pb = ff_read16f<0>(cb2 + tile * 128);
    f32x4_t acc[2];
    ff_chain_tile(fa + (tile % 3) * FFC_WT, xf, acc);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (f16)(acc[i][e] + pb[e]);
      if (i == 0) ff_write8<0>(img_w + ct * 64, o);
      else ff_write8<16 * FFC_PITCH>(img_w + ct * 64, o);
    }
    if (tile == 0) { CH_PF(7) }
    if (ct == 9) {   // a 320-column chunk of Y is complete: out in whole rows (the next tile's wait + barrier orders these reads before its writes)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      CH_PF(8)
      ff_chain_copy_out(smem_raw, t, m0, (f16*)p.Y, p.ldy, (tile / 10) * FF_C);
      CH_PF(9)
    }
  }
  CH_PF(10)
}

}  // namespace

extern "C" int tb_chain320_debug(void* buf32) {   // profiling builds only (CH_PROF): 32 x uint64 device buffer, NULL = off
  unsigned long long* v = (unsigned long long*)buf32;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_ch_dbg), &v, sizeof(v)) == hipSuccess ? TB_OK : TB_ELAUNCH;
}

extern "C" int tb_chain320_ok(int64_t M, int N2) { return M > 0 && M % FF_BM == 0 && N2 > 0 && N2 % FF_C == 0 && N2 <= 2 * FF_INNER; }

extern "C" int tb_chain320(const tb_chain_desc* dp, tb_stream_t stream) {
  (void)hipGetLastError();
  if (!dp) return TB_EINVAL;
  const tb_chain_desc d = *dp;
  if (!tb_chain320_ok(d.M, d.N2)) return TB_EINVAL;
  if (!d.X || !d.W1 || !d.W2 || !d.Y || !d.gamma || !d.beta) return TB_EINVAL;
  if (d.ldx % 8 || d.ldw1 % 8 || d.ldw2 % 8 || d.ldy % 8 || (d.T && d.ldt % 8) || (d.R1 && d.ldr1 % 4)) return TB_EINVAL;
  if (((uintptr_t)d.X) % 16 || ((uintptr_t)d.W1) % 16 || ((uintptr_t)d.W2) % 16 || ((uintptr_t)d.Y) % 16 || ((uintptr_t)d.T) % 16 ||
      ((uintptr_t)d.R1) % 8 || ((uintptr_t)d.gamma) % 16 || ((uintptr_t)d.beta) % 16 || ((uintptr_t)d.stats) % 8)
    return TB_EINVAL;
  const int64_t lim = (int64_t)1 << 31;
  if ((int64_t)FF_C * d.ldw1 * 2 >= lim || (int64_t)32 * d.ldw2 * 2 >= lim) return TB_EINVAL;   // 32-bit per-lane byte offsets inside a weight tile
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)chain320_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FFC_LDS) != hipSuccess) return TB_ELAUNCH;
    attr_done = true;
  }
  hipLaunchKernelGGL(chain320_kernel, dim3((unsigned)(d.M / FF_BM)), dim3(512), FFC_LDS, (hipStream_t)stream, d);
  TB_CHECK_LAUNCH();
  return TB_OK;
}
