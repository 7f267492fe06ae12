// Row-chain pieces shared by ff_fused.hip (the feed-forward's chained neighbours) and chain320.hip (Linear -> LayerNorm -> Linear on the 64x64 maps):
// LDS asm helpers, the 20 KB weight-tile stream of a [128 x 320] x [N x 320]^T stage, its k-loop, the [128][320] fp16 image.
#pragma once
#include "gemm_epi.h"
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int FF_C = 320, FF_INNER = 1280, FF_BM = 128, FF_NT = FF_INNER / 32;   // 40 tiles of 32 intermediate columns
constexpr int FF_SLOT = 60 * 1024;            // one ring slot: phase-A weight tile + phase-B weight tile
constexpr int FF_NDMA = 60;                   // LDS-DMA wave instructions (1 KB each) per tile
constexpr int FF_U_OFF = 2 * FF_SLOT;         // exchange buffer: forward [4][32][32] fp16 (8 KB), backward [4][32][64] fp16 (16 KB)
constexpr int FF_BIAS_OFF = FF_U_OFF + 16 * 1024;   // forward: the packed proj bias, 2560 floats
constexpr int FF_LDS = FF_BIAS_OFF + 2 * FF_INNER * 4;

__device__ __forceinline__ uint32_t ff_lds_addr(const void* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p; }
template <int OFF>
__device__ __forceinline__ f16x8 ff_read16(uint32_t addr) {
  f16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF>
__device__ __forceinline__ f32x4_t ff_read16f(uint32_t addr) {
  f32x4_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF>
__device__ __forceinline__ void ff_write8(uint32_t addr, f16x4 v) {
  asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void ff_wait_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
#define FF_SB() __builtin_amdgcn_sched_barrier(0)
// ---- round 6: the feed-forward's row-local neighbours in the same launch (tb_ff_desc.pre_W / post_W; forward only).
// A "chain stage" is one more [128 x 320] x [320 x 320]^T product on the workgroup's rows with the 128 x 320 operand in the SAME 80 registers the
// feed-forward keeps its input in: ten 32-column weight tiles (20 KB each, the backward's phase-A image: five [32 rows][128 B] slabs, chunk ^= row & 7)
// through two LDS slots at the bottom of ring slot 0, phase A's k-loop per tile (20 MFMAs per wave), a per-tile epilogue on (rows 32 wm + 16 i + l15,
// columns 32 tile + 16 wn + 4 lq .. + 3).  Results change hands through a [128][320] fp16 LDS IMAGE (656-byte pitch: the b128 reads of 16 consecutive
// rows at one column land on 16 distinct bank quads) behind ring slot 0, from which the next product's operand fragments are read back:
//   PRE : X -> t2 = X Wpre^T + b (+ R) -> pre_Y and the image -> two-pass LayerNorm of the image rows in place (+ stats) -> the feed-forward's operand
//   POST: t3 = ff + b2 + R -> the image (-> Y when asked for) -> operand -> post_Y = t3 Wpost^T + b (+ R)
constexpr int FFC_WT = 5 * 32 * 128;                      // one chain weight tile: 20 KB
constexpr int FFC_IMG = 60 * 1024, FFC_PITCH = 656;       // the image: [60 KB, 60 KB + 128 * 656) = up to 142 KB (over slot 1 and the exchange buffer)
constexpr int FFC_BIAS_OFF = 146 * 1024;                  // the packed proj bias of the chained launches sits behind it
constexpr int FFC_CB_OFF = FFC_BIAS_OFF + 2 * FF_INNER * 4;   // 156 KB: a chain stage's own bias (320 floats)
constexpr int FFC_GB_OFF = FFC_CB_OFF + FF_C * 4;         // the chained LayerNorm's gamma | beta (2 x 320 floats)
constexpr int FFC_LDS = FFC_GB_OFF + 2 * FF_C * 4;        // 159.75 KB
static_assert(FFC_IMG + FF_BM * FFC_PITCH <= FFC_BIAS_OFF && 3 * FFC_WT <= FFC_IMG && FFC_LDS <= 160 * 1024, "chain LDS map");
template <int OFF>
__device__ __forceinline__ void ff_write16(uint32_t addr, f16x8 v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}

// One chain stage's weight stream: every wave issues exactly THREE 1 KB pieces per 20 KB tile (q = wave, wave + 8, wave + 16; waves 4-7 have only
// two and repeat their second: same bytes to the same place), so that the counted waits below are wave-independent.  Three ring slots at the bottom
// of the LDS; tile t lives in slot t % 3 and is requested two tiles ahead: a tile's k-loop is ~0.3 us, an L2 round trip ~1 us.
struct FfChain {
  uint32_t off[3], dst[3];
  const char* W;
  int64_t tile_stride;
  unsigned char* smem;
  __device__ __forceinline__ FfChain(unsigned char* smem_raw, const void* W_, int64_t ldw, int wave, int lane) {
    smem = smem_raw, W = (const char*)W_, tile_stride = 32 * ldw * 2;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int q = wave + 8 * k < 20 ? wave + 8 * k : wave + 8 * (k - 1);
      const int sl = q >> 2, rgp = q & 3;
      const int row = rgp * 8 + (lane >> 3), cc = (lane & 7) ^ (lane >> 3);
      off[k] = (uint32_t)(((int64_t)row * ldw + sl * 64 + cc * 8) * 2);
      dst[k] = (uint32_t)(sl * 4096 + rgp * 1024);
    }
  }
  __device__ __forceinline__ void issue(int tile) const {
    // asm pieces: the builtin gets an `s_waitcnt vmcnt(0)` in front from hipcc (it cannot prove the DMA's LDS destination free of pending accesses),
    // which drained the two tiles in flight at every issue -- the stages ran one L2 round trip per tile (chain320 at N2 = 320: 44.7 us)
    const char* base = W + tile * tile_stride;
    const uint32_t slot = ff_lds_addr(smem) + (tile % 3) * FFC_WT;
#pragma unroll
    for (int k = 0; k < 3; ++k) glds16_asm(base + off[k], slot + dst[k]);
  }
};
// acc = X[rows 32 wm + 16 i + l15] . Wtile^T (columns 16 wn + 4 lq .. + 3 of the tile): phase A of the backward on the 20 KB image at `slot_addr`
__device__ __forceinline__ void ff_chain_tile(uint32_t a0, const f16x8 (&xf)[2][10], f32x4_t (&acc)[2]) {
  const uint32_t a1 = a0 ^ 64;
  acc[0] = f32x4_t{0.f, 0.f, 0.f, 0.f}, acc[1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  f16x8 wf[3];
  auto rd = [&](auto ksc) {
    constexpr int ks = decltype(ksc)::value;
    wf[ks % 3] = ff_read16<(ks >> 1) * 4096>((ks & 1) ? a1 : a0);
  };
  rd(std::integral_constant<int, 0>{});
  rd(std::integral_constant<int, 1>{});
  FF_SB();
  auto step = [&](auto ksc) {
    constexpr int ks = decltype(ksc)::value;
    if constexpr (ks + 2 < 10) rd(std::integral_constant<int, ks + 2>{});
    constexpr int later = (ks + 2 < 10 ? 1 : 0) + (ks + 1 < 10 ? 1 : 0);
    ff_wait_lgkm<later>();
    FF_SB();
#pragma unroll
    for (int i = 0; i < 2; ++i) acc[i] = TB_MFMA_16x16x32(wf[ks % 3], xf[i][ks], acc[i]);
    FF_SB();
  };
  step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
  step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
  step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
  step(std::integral_constant<int, 9>{});
}

// LayerNorm of the image rows, in place (callers: the image is complete and past a barrier): four threads per row (80 columns each),
// tb_layernorm_fwd's two-pass arithmetic on the fp16 image values; the un-normalised rows leave for `Y` (the residual stream) from here in 160-byte
// runs -- a whole 640-byte row per four lanes -- and (mean, rstd) for `stats`
// gamma / beta come from the LDS (ff_chain_stage_gamma_beta, called at kernel entry): read from memory inside this stage -- 40 dependent L2 round
// trips per thread in ten rounds -- they made it 11 us of a 44 us launch (s_memtime stamps, scratch/r6/chain_prof.py)
__device__ __forceinline__ void ff_chain_stage_gamma_beta(unsigned char* smem_raw, int t, const float* gamma, const float* beta) {
  float* gb = reinterpret_cast<float*>(smem_raw + FFC_GB_OFF);
  for (int i = t; i < 2 * FF_C; i += 512) gb[i] = i < FF_C ? gamma[i] : beta[i - FF_C];   // (512 threads, 640 values)
}
__device__ __forceinline__ void ff_chain_layernorm(unsigned char* smem_raw, int t, int64_t m0, f16* Y, int64_t ldy, float eps, float* stats) {
  const int lrow = t >> 2, lqr = t & 3;
  const uint32_t la = ff_lds_addr(smem_raw) + FFC_IMG + lrow * FFC_PITCH + lqr * 160;
  f16x8 v[10];
#pragma unroll
  for (int j = 0; j < 10; ++j) v[j] = ff_read16<0>(la + 16 * j);
  ff_wait_lgkm<0>();
  FF_SB();
  if (Y) {
    f16* yp = Y + (m0 + lrow) * ldy + lqr * 80;
#pragma unroll
    for (int j = 0; j < 10; ++j) *(f16x8*)(yp + 8 * j) = v[j];
  }
  float sm = 0.f;
#pragma unroll
  for (int j = 0; j < 10; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) sm += (float)v[j][e];
  sm += __shfl_xor(sm, 1, 64);
  sm += __shfl_xor(sm, 2, 64);
  const float mean = sm / (float)FF_C;
  float qq = 0.f;
#pragma unroll
  for (int j = 0; j < 10; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float dlt = (float)v[j][e] - mean;
      qq += dlt * dlt;
    }
  qq += __shfl_xor(qq, 1, 64);
  qq += __shfl_xor(qq, 2, 64);
  const float rstd = rsqrtf(qq / (float)FF_C + eps);
  if (lqr == 0 && stats) {
    stats[2 * (m0 + lrow)] = mean;
    stats[2 * (m0 + lrow) + 1] = rstd;
  }
  const uint32_t ga = ff_lds_addr(smem_raw) + FFC_GB_OFF + lqr * 320;   // this quarter row's 80 gamma values; beta 1280 B behind
#pragma unroll
  for (int j = 0; j < 10; ++j) {
    const f32x4_t g0 = ff_read16f<0>(ga + 32 * j), g1 = ff_read16f<16>(ga + 32 * j);
    const f32x4_t b0 = ff_read16f<FF_C * 4>(ga + 32 * j), b1 = ff_read16f<FF_C * 4 + 16>(ga + 32 * j);
    ff_wait_lgkm<0>();
    FF_SB();
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = (f16)(((float)v[j][e] - mean) * rstd * g0[e] + b0[e]);
      o[4 + e] = (f16)(((float)v[j][4 + e] - mean) * rstd * g1[e] + b1[e]);
    }
    ff_write16<0>(la + 16 * j, o);
  }
}
// the image rows -> Y[m0 + row][col0 .. col0 + 319] in whole rows (callers: the image is complete and past a barrier)
__device__ __forceinline__ void ff_chain_copy_out(unsigned char* smem_raw, int t, int64_t m0, f16* Y, int64_t ldy, int col0) {
  const int lrow = t >> 2, lqr = t & 3;
  const uint32_t la = ff_lds_addr(smem_raw) + FFC_IMG + lrow * FFC_PITCH + lqr * 160;
  f16x8 v[10];
#pragma unroll
  for (int j = 0; j < 10; ++j) v[j] = ff_read16<0>(la + 16 * j);
  ff_wait_lgkm<0>();
  FF_SB();
  f16* yp = Y + (m0 + lrow) * ldy + col0 + lqr * 80;
#pragma unroll
  for (int j = 0; j < 10; ++j) *(f16x8*)(yp + 8 * j) = v[j];
}
// the image -> the operand fragments of the next product (rows 32 wm + 16 i + l15, k = 32 ks + 8 lq .. + 7)
__device__ __forceinline__ void ff_chain_operand(unsigned char* smem_raw, int wm, int l15, int lq, f16x8 (&xf)[2][10]) {
  const uint32_t img_x = ff_lds_addr(smem_raw) + FFC_IMG + (wm * 32 + l15) * FFC_PITCH + lq * 16;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int ks = 0; ks < 10; ++ks) xf[i][ks] = ff_read16<0>(img_x + i * 16 * FFC_PITCH + ks * 64);
  ff_wait_lgkm<0>();
  FF_SB();
}

}  // namespace
