// Fused GEGLU feed-forward of diffusers' BasicTransformerBlock (ff.net.0.proj -> h * gelu(g) -> ff.net.2, train_textboost.py:1063-1067 forward,
// :1108 backward) for the 64x64-map transformer blocks of the SD1.x UNet (C = 320, inner = 1280, M = B * 4096 rows), gfx950.
//
// Why a kernel of its own: as two launches the layer is bound by what it moves, not by what it multiplies.  Forward, per block at B = 8:
// ff.net.0.proj reads 21 MB and writes the 84 MB gated tensor plus the 168 MB pre-gate projections the backward needs, ff.net.2 reads the
// 84 MB again (121 + 53 us measured); backward, the ff.net.2 dgrad writes 168 MB of d(proj) that the ff.net.0 dgrad reads back (97 + 93 us).
// The gated tensor (forward) and d(proj) (backward) only exist to carry data from one GEMM's epilogue to the next GEMM's operand fetch.
//
// Here one workgroup (8 waves, one per CU, M / 128 of them = exactly one chip round at the metric batch) owns 128 rows for the whole layer:
//   * ACTIVATION-STATIONARY: the 128 x 320 input tile (forward: LayerNorm(x); backward: d(out)) lives in REGISTERS as MFMA operand fragments
//     (wave (wm, wn) holds rows 32 wm .. 32 wm + 31 for all 320 k: 80 VGPRs), loaded once.
//   * the workgroup walks the 1280-wide intermediate in 40 tiles of 32 columns.  Per tile:
//       phase A  (K = 320 off the register-resident operand; weights from a 2-slot LDS ring filled by LDS-DMA one tile ahead)
//                forward : [h | g] = x W1_tile^T + b1 -> packed pre-gate projections to HBM (for the backward), u = h * gelu(g)
//                backward: du = d(out) W2^T_tile;  dh = du gelu(g), dg = du h gelu'(g)  (h, g read from the packed projections)
//       exchange the 128 x 32 (forward) / 128 x 64 (backward) fp16 result through 8 / 16 KB of LDS (the two waves of a row band own 16 columns each)
//       phase B  out[128, 320] += u W2[:, tile]^T   /   dx[128, 320] += [dh | dg] W1p^T[:, tile]^T   into 80 accumulator registers per lane
//   * epilogue once per workgroup: + bias + residual, fp16 store.
// HBM traffic per block: forward 21 + 168 + 21 (+ 21 residual) MB instead of 21 + 252 + 84 + 21 (+ 21); backward 21 + 168 + 21 instead of
// 21 + 168 + 168 + 168 + 21.  Matrix work is unchanged (80.5 GFLOP per direction), 60 v_mfma_f32_16x16x32_f16 per wave and tile.
//
// LDS image conventions (shared with gemm8.hip): operands arrive by global_load_lds_dwordx4 (lane-linear 1 KB per wave instruction), the bank
// swizzle lives in the per-lane SOURCE address and in the fragment reads; 128-byte rows use chunk ^= row & 7, the 64-byte rows of the forward's
// W2 / u tiles use chunk ^= SWZ4[(row >> 2) & 3] with SWZ4 = {0, 2, 3, 1} (every 16-lane group of a ds_read_b128 then touches all 64 banks once).
// All LDS accesses inside the main loop are inline asm: hipcc cannot prove that a read does not alias an LDS-DMA destination and drains
// vmcnt(0) -- the prefetched tile -- in front of every compiler-visible LDS access.
#include "ff_chain.h"

namespace {

#ifndef FF_DMAC
#define FF_DMAC 1   // 0 (TB_CFLAGS=-DFF_DMAC=0): the tile's LDS-DMA pieces in front of phase A, as in round 4
#endif
#ifndef FF_PROF
#define FF_PROF 0   // profiling build (TB_CFLAGS=-DFF_PROF=1): per-phase s_memtime sums of waves 0 and 4 of workgroup 0 -> tb_ff_debug buffer
#endif
__device__ unsigned long long* g_ff_dbg = nullptr;
#if FF_PROF
#define FF_PF(k)                                                  \
  {                                                               \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    pf_sum[k] += now_ - pf_t;                                     \
    pf_t = now_;                                                  \
  }
#else
#define FF_PF(k)
#endif

template <bool BWD, bool PRE = false, bool POST = false>
__global__ __launch_bounds__(512, 2) void ff_fused_kernel(const tb_ff_desc p) {
  static_assert(!BWD || (!PRE && !POST), "the chained stages exist in the forward only");
  constexpr bool CHAIN = PRE || POST;
  constexpr int BIAS_OFF = CHAIN ? FFC_BIAS_OFF : FF_BIAS_OFF;
  extern __shared__ __attribute__((aligned(128))) unsigned char smem_raw[];
  const int t = threadIdx.x, lane = t & 63, l15 = lane & 15, lq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.x * FF_BM;

  // phase-A weight tile: forward 64 rows ([h32 | g32]) x 320 k, backward 32 rows x 320 k: five [rows][128 B] slabs; phase-B tile behind it
  constexpr int A_ROWS = BWD ? 32 : 64, A_SLAB = A_ROWS * 128, A_BYTES = 5 * A_SLAB;
  constexpr int A_NI = A_BYTES / 1024;                  // 40 / 20 wave instructions
  constexpr int B_ROWB = BWD ? 128 : 64;                // phase-B tile: [320 rows][64 k] (backward) / [320 rows][32 k] (forward)
  constexpr int U_ROWB = B_ROWB, U_BAND = 32 * U_ROWB;  // exchange buffer: one 32-row band per wm

  // ---- the forward's packed projection bias through LDS (a tile's 8 values per lane seed the accumulators)
  if (!BWD) {
    float* bias_s = reinterpret_cast<float*>(smem_raw + BIAS_OFF);
    for (int i = t; i < 2 * FF_INNER; i += 512) bias_s[i] = p.b1 ? p.b1[i] : 0.f;
  }

  // ---- LDS-DMA sources of this lane: instruction q = wave + 8 k (k = 0..7, q < 60) of every tile
  uint32_t d_off[8];      // byte offset of the lane's 16 bytes from the operand base, tile 0
  uint32_t d_dst[8];      // byte offset of the instruction's 1 KB inside the slot
  bool d_isb[8];          // phase-B operand (W2)?
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    // (waves 4-7 have only 7 instructions of the 60: their eighth repeats the seventh -- same bytes to the same place -- so that EVERY wave issues
    // exactly 8 per tile and the counted wait of the backward's h / g loads below is one unconditional instruction)
    const int q = wave + 8 * k < FF_NDMA ? wave + 8 * k : wave + 8 * (k - 1);
    if (q < A_NI) {
      const int s = q / (A_ROWS / 8), rgp = q % (A_ROWS / 8);
      const int row = rgp * 8 + (lane >> 3), cc = (lane & 7) ^ (lane >> 3);
      d_off[k] = (uint32_t)(((int64_t)row * p.ldw1 + s * 64 + cc * 8) * 2);
      d_dst[k] = (uint32_t)(s * A_SLAB + rgp * 1024);
      d_isb[k] = false;
    } else {
      const int qq = q - A_NI;
      d_isb[k] = true;
      d_dst[k] = (uint32_t)(A_BYTES + qq * 1024);
      if (BWD) {  // [320][128 B]: 8 rows per instruction
        const int row = qq * 8 + (lane >> 3), cc = (lane & 7) ^ (lane >> 3);
        d_off[k] = (uint32_t)(((int64_t)(row < FF_C ? row : 0) * p.ldw2 + cc * 8) * 2);
      } else {    // [320][64 B]: 16 rows per instruction
        const int row = qq * 16 + (lane >> 2), cc = (lane & 3) ^ ((0x78 >> (2 * ((lane >> 4) & 3))) & 3);
        d_off[k] = (uint32_t)(((int64_t)(row < FF_C ? row : 0) * p.ldw2 + cc * 8) * 2);
      }
    }
  }
  const int64_t a_tile_stride = (int64_t)A_ROWS * p.ldw1 * 2;   // bytes between consecutive phase-A tiles (rows of W1)
  constexpr int b_tile_stride = BWD ? 128 : 64;                  // ... phase-B tiles (columns of W2)
  auto issue_tile = [&](int tile, int slot) {
    const char* wa = (const char*)p.W1 + tile * a_tile_stride;
    const char* wb = (const char*)p.W2 + (int64_t)tile * b_tile_stride;
    f16* const dst = reinterpret_cast<f16*>(smem_raw + slot * FF_SLOT);
#pragma unroll
    for (int k = 0; k < 8; ++k) glds16((const f16*)((d_isb[k] ? wb : wa) + d_off[k]), dst + (d_dst[k] >> 1));
  };
  f16x8 xf[2][10];
  if constexpr (!PRE) issue_tile(0, 0);

  // ---- the register-resident operand: rows 32 wm + 16 i + l15, k = 32 ks + 8 lq .. + 7
  if constexpr (PRE) {
    // ---- round 6: attn2.to_out + bias + residual -> t2 (pre_Y and the LDS image) -> norm3 in place -> the feed-forward's operand
    const FfChain ch(smem_raw, p.pre_W, p.ld_prew, wave, lane);
    ch.issue(0);
    ch.issue(1);
    const uint32_t lds0c = ff_lds_addr(smem_raw);
    const int64_t row0 = m0 + wm * 32 + l15;
    {
      const f16* xr = (const f16*)p.X + row0 * p.ldx + 8 * lq;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < 10; ++ks) xf[i][ks] = *(const f16x8*)(xr + (int64_t)i * 16 * p.ldx + 32 * ks);
    }
    // the residual of all ten tiles is requested before the tile loop (inside it only the weight DMA may be on the vector-memory queue: loads return
    // in order, so the counted wait of a tile means "this tile has landed"); the stage's bias goes through the LDS
    f16x4 rr[10][2];
#pragma unroll
    for (int tl = 0; tl < 10; ++tl)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        rr[tl][i] = p.pre_R ? *(const f16x4*)((const f16*)p.pre_R + (row0 + 16 * i) * p.ld_prer + tl * 32 + wn * 16 + 4 * lq) : f16x4{0, 0, 0, 0};
    if (t < FF_C) reinterpret_cast<float*>(smem_raw + FFC_CB_OFF)[t] = p.pre_b ? p.pre_b[t] : 0.f;
    ff_chain_stage_gamma_beta(smem_raw, t, p.pre_gamma, p.pre_beta);
    const uint32_t fa = lds0c + (wn * 16 + l15) * 128 + ((lq ^ (l15 & 7)) << 4);
    const uint32_t img_w = lds0c + FFC_IMG + (wm * 32 + l15) * FFC_PITCH + (wn * 16 + 4 * lq) * 2;   // + i * 16 rows, + 64 B per tile
    const uint32_t cb_a = lds0c + FFC_CB_OFF + (wn * 16 + 4 * lq) * 4;                               // + 128 B per tile
    // (the tile loop unrolled by hand: rr is a register array and must be indexed by constants)
    auto run = [&](auto tlc) {
      constexpr int tile = decltype(tlc)::value;
      if constexpr (tile + 1 < 10) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // tile landed for every wave (tile 0: so has the
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");                            // bias above); tile - 1's slot is free
      if constexpr (tile + 2 < 10) ch.issue(tile + 2);
      FF_SB();
      const f32x4_t pb = ff_read16f<tile * 128>(cb_a);   // (waited for with the k-loop's first fragments)
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
    run(std::integral_constant<int, 0>{}); run(std::integral_constant<int, 1>{}); run(std::integral_constant<int, 2>{});
    run(std::integral_constant<int, 3>{}); run(std::integral_constant<int, 4>{}); run(std::integral_constant<int, 5>{});
    run(std::integral_constant<int, 6>{}); run(std::integral_constant<int, 7>{}); run(std::integral_constant<int, 8>{});
    run(std::integral_constant<int, 9>{});
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the image is complete; the chain's weight slots are free
    issue_tile(0, 0);                                                   // the feed-forward's first tile streams in under the LayerNorm
    // ---- norm3 on the image rows, in place; the rows also leave for pre_Y from there
    ff_chain_layernorm(smem_raw, t, m0, (f16*)p.pre_Y, p.ld_prey, p.pre_eps, p.pre_stats);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // l3 is in the image
    {
      const uint32_t img_x = lds0c + FFC_IMG + (wm * 32 + l15) * FFC_PITCH + lq * 16;   // operand fragments: + i * 16 rows, + 64 B per k-step
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < 10; ++ks) xf[i][ks] = ff_read16<0>(img_x + i * 16 * FFC_PITCH + ks * 64);
      ff_wait_lgkm<0>();
      FF_SB();
    }
  } else {
    const f16* xr = (const f16*)p.X + (m0 + wm * 32 + l15) * p.ldx + 8 * lq;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 10; ++ks) xf[i][ks] = *(const f16x8*)(xr + (int64_t)i * 16 * p.ldx + 32 * ks);
  }
  f32x4_t acc_o[2][10];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jj = 0; jj < 10; ++jj) acc_o[i][jj] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // ---- fragment addresses (lane parts; + slot * FF_SLOT per tile)
  const uint32_t lds0 = ff_lds_addr(smem_raw);
  const int swz8 = l15 & 7, swz4 = (0x78 >> (2 * ((l15 >> 2) & 3))) & 3;
  const uint32_t fa0 = lds0 + (wn * 16 + l15) * 128 + ((lq ^ swz8) << 4);            // phase A, even ks; odd ks: ^ 64; slab (ks >> 1) * A_SLAB
  const uint32_t fb0 = lds0 + A_BYTES + (wn * 160 + l15) * B_ROWB + (BWD ? ((lq ^ swz8) << 4) : ((lq ^ swz4) << 4));   // + jj * 16 * B_ROWB
  const uint32_t fu0 = lds0 + FF_U_OFF + wm * U_BAND + l15 * U_ROWB + (BWD ? ((lq ^ swz8) << 4) : ((lq ^ swz4) << 4));  // + i * 16 * U_ROWB
  const int uc = wn * 2 + (lq >> 1);   // 16-byte chunk of this lane's four exchange columns (backward: the dg copy is chunk uc + 4, i.e. ^ 64)
  const uint32_t uw0 = lds0 + FF_U_OFF + wm * U_BAND + l15 * U_ROWB + ((uc ^ (BWD ? swz8 : swz4)) << 4) + (lq & 1) * 8;
  const uint32_t bias_a = lds0 + BIAS_OFF + (wn * 16 + 4 * lq) * 4;
  f16* const hg_lane = (f16*)p.HG + (m0 + wm * 32 + l15) * p.ldhg + wn * 16 + 4 * lq;

#if FF_PROF
  unsigned long long pf_sum[6] = {0, 0, 0, 0, 0, 0}, pf_t = __builtin_amdgcn_s_memtime();
  const unsigned long long pf_start = pf_t;
#endif
  for (int tile = 0; tile < FF_NT; ++tile) {
    const int slot = tile & 1;
    // tile `tile` has landed for every wave; every wave is done with tile - 1 (its ring slot and the exchange buffer are free again)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    f16x4 hin[2], gin[2];
    if (BWD) {  // the pre-gate projections of this tile's (row, column) units: consumed after phase A.  Inline asm + a counted wait below: the
                // compiler's own wait for a plain load here is vmcnt(0), which also drains the next tile's LDS-DMA issued right behind it
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const f16* src = hg_lane + (int64_t)i * 16 * p.ldhg + tile * 64;
        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(hin[i]) : "v"(src) : "memory");
        asm volatile("global_load_dwordx2 %0, %1, off offset:64" : "=v"(gin[i]) : "v"(src) : "memory");
      }
    }
    // (the last tile re-fetches itself into the free slot: never read, keeps the count at 8)
    const int ntile = tile + 1 < FF_NT ? tile + 1 : tile;
    const char* const wa_n = (const char*)p.W1 + ntile * a_tile_stride;
    const char* const wb_n = (const char*)p.W2 + (int64_t)ntile * b_tile_stride;
    const uint32_t dst_n = lds0 + (slot ^ 1) * FF_SLOT;
    // FF_DMAC (round 5, as DMAC in gemm8.hip): the next tile's eight 1 KB LDS-DMA pieces go out as asm statements BETWEEN the MFMAs of phase A (one
    // behind each of the first eight k-steps) instead of back to back in front of it, where every piece cost the wave 100 ... 185 issue cycles
    if (!FF_DMAC) issue_tile(ntile, slot ^ 1);
    FF_SB();

    // ---------------------------------------------------------------- phase A
    constexpr int NA = BWD ? 1 : 2;      // accumulator sets: backward du; forward h, g
    f32x4_t acc_a[NA][2];
    f32x4_t bias_h, bias_g;
    FF_PF(0)
    {
      const uint32_t a0 = fa0 + slot * FF_SLOT, a1 = a0 ^ 64;
#pragma unroll
      for (int a = 0; a < NA; ++a) acc_a[a][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}, acc_a[a][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      if (!BWD) bias_h = ff_read16f<0>(bias_a + tile * 256), bias_g = ff_read16f<128>(bias_a + tile * 256);   // (waited for with the first fragments)
      f16x8 wf[3][NA];   // fragment ring: reads run two k-steps ahead of their MFMAs
      auto rd = [&](auto ksc) {
        constexpr int ks = decltype(ksc)::value;
        constexpr int off = (ks >> 1) * A_SLAB;
        wf[ks % 3][0] = ff_read16<off>((ks & 1) ? a1 : a0);
        if (!BWD) wf[ks % 3][NA - 1] = ff_read16<off + 32 * 128>((ks & 1) ? a1 : a0);
      };
      rd(std::integral_constant<int, 0>{});
      rd(std::integral_constant<int, 1>{});
      FF_SB();
      auto step = [&](auto ksc) {
        constexpr int ks = decltype(ksc)::value;
        if constexpr (ks + 2 < 10) rd(std::integral_constant<int, ks + 2>{});
        constexpr int later = (ks + 2 < 10 ? NA : 0) + (ks + 1 < 10 ? NA : 0);   // reads issued after this k-step's
        ff_wait_lgkm<later>();
        FF_SB();
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
          for (int i = 0; i < 2; ++i) acc_a[a][i] = TB_MFMA_16x16x32(wf[ks % 3][a], xf[i][ks], acc_a[a][i]);
        if constexpr (FF_DMAC && ks < 8) glds16_asm((d_isb[ks] ? wb_n : wa_n) + d_off[ks], dst_n + d_dst[ks]);
        FF_SB();
      };
      step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
      step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
      step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
      step(std::integral_constant<int, 9>{});
    }
    FF_PF(1)
    // hin / gin have landed once at most the 8 LDS-DMA instructions issued after them are outstanding.  ONE unconditional asm: with the wait in
    // the arms of a branch hipcc placed the register copies that tie the operands in FRONT of the s_waitcnt in one arm (stale h / g for half the waves)
    if (BWD) asm volatile("s_waitcnt vmcnt(8)" : "+v"(hin[0]), "+v"(hin[1]), "+v"(gin[0]), "+v"(gin[1]));
    // ---- phase A epilogue: this lane's (row 16 i + l15, columns wn 16 + 4 lq .. + 3) units
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (!BWD) {
        f16x4 h4, g4, u4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          h4[e] = (f16)(acc_a[0][i][e] + bias_h[e]);     // tb_gemm's epilogue arithmetic: acc + bias, rounded to fp16
          g4[e] = (f16)(acc_a[NA - 1][i][e] + bias_g[e]);
          u4[e] = (f16)((float)h4[e] * gelu_erf_f((float)g4[e]));   // gate on the fp16-rounded projections, as an fp16 module would
        }
        f16* dst = hg_lane + (int64_t)i * 16 * p.ldhg + tile * 64;
        *(f16x4*)dst = h4;
        *(f16x4*)(dst + 32) = g4;
        if (i == 0) ff_write8<0>(uw0, u4);
        else ff_write8<16 * U_ROWB>(uw0, u4);
      } else {
        f16x4 dh, dg;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = acc_a[0][i][e];
          float ge, dge;
          gelu_erf_both_f((float)gin[i][e], ge, dge);
          dh[e] = (f16)(v * ge);
          dg[e] = (f16)(v * (float)hin[i][e] * dge);
        }
        if (i == 0) {
          ff_write8<0>(uw0, dh);
          ff_write8<0>(uw0 ^ 64, dg);
        } else {
          ff_write8<16 * U_ROWB>(uw0, dh);
          ff_write8<16 * U_ROWB>(uw0 ^ 64, dg);
        }
      }
    }
    FF_PF(2)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the exchange tile is complete
    FF_PF(3)

    // ---------------------------------------------------------------- phase B
    {
      constexpr int KS = BWD ? 2 : 1;    // 32-wide k-steps of the exchange tile
      const uint32_t b0 = fb0 + slot * FF_SLOT;
      f16x8 uf[KS][2];
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        uf[s][0] = ff_read16<0>(s ? (fu0 ^ 64) : fu0);
        uf[s][1] = ff_read16<16 * U_ROWB>(s ? (fu0 ^ 64) : fu0);
      }
      f16x8 wf[3][KS];
      auto rd = [&](auto jc) {
        constexpr int jj = decltype(jc)::value;
#pragma unroll
        for (int s = 0; s < KS; ++s) wf[jj % 3][s] = ff_read16<jj * 16 * B_ROWB>(s ? (b0 ^ 64) : b0);
      };
      rd(std::integral_constant<int, 0>{});
      rd(std::integral_constant<int, 1>{});
      FF_SB();
      auto step = [&](auto jc) {
        constexpr int jj = decltype(jc)::value;
        if constexpr (jj + 2 < 10) rd(std::integral_constant<int, jj + 2>{});
        constexpr int later = (jj + 2 < 10 ? KS : 0) + (jj + 1 < 10 ? KS : 0);
        ff_wait_lgkm<later>();
        FF_SB();
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
          for (int i = 0; i < 2; ++i) acc_o[i][jj] = TB_MFMA_16x16x32(wf[jj % 3][s], uf[s][i], acc_o[i][jj]);
        FF_SB();
      };
      step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
      step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
      step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
      step(std::integral_constant<int, 9>{});
    }
    FF_PF(4)
  }

#if FF_PROF
  if (g_ff_dbg && blockIdx.x == 0 && (t == 0 || t == 256)) {
    unsigned long long* o = g_ff_dbg + (t ? 8 : 0);
    for (int k = 0; k < 5; ++k) o[k] = pf_sum[k];
    o[5] = __builtin_amdgcn_s_memtime() - pf_start;
  }
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last tile's dummy prefetch
  // ---- epilogue: rows 32 wm + 16 i + l15, columns wn 160 + 16 jj + 4 lq .. + 3
  // (PRE + POST: the two stages compute the same lane constants; left to itself hipcc keeps PRE's copies alive ACROSS the main loop for POST --
  // the loop sits at 248 of 256 registers -- and spills inside it, +30 us.  Opaque copies of the lane indices cut the common subexpressions.)
  int l15e = l15, lqe = lq;
  if constexpr (PRE && POST) asm volatile("" : "+v"(l15e), "+v"(lqe));
  const int64_t mrow = m0 + wm * 32 + l15e;
  const int ncol = wn * 160 + 4 * lqe;
  if constexpr (BWD) {
    if (p.ln_x) {
      // LayerNorm backward of norm3 on the accumulators (round 5; it had been a launch of its own since the feed-forward was fused): the 320-wide row of
      // a tile row sits in 2 waves (wn) x 4 lanes (lq) x 40 accumulator values: lane partials of (sum g, sum g xhat), two cross-lane steps, one LDS
      // exchange between the two waves of the row band (added in the fixed order wn = 0, 1), then dx = rstd (g - mean g - xhat mean(g xhat)) + R --
      // tb_layernorm_bwd's arithmetic on the fp32 accumulators (one rounding fewer than through the fp16 dl3 tensor).
      typedef __attribute__((ext_vector_type(2))) float f32x2_t;
      // (the LayerNorm input x is read twice, here and in the output pass -- the second time from the L2: held in 40 registers across the exchange
      // the kernel spilled)
      f32x2_t st[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) st[i] = *(const f32x2_t*)(p.ln_stats + 2 * (mrow + 16 * i));
      float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
      for (int jj = 0; jj < 10; ++jj) {
        const f32x4_t gm = *(const f32x4_t*)(p.ln_gamma + ncol + 16 * jj);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const f16x4 x4 = *(const f16x4*)((const f16*)p.ln_x + (mrow + 16 * i) * p.ld_lnx + ncol + 16 * jj);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float g = acc_o[i][jj][e] * gm[e];
            acc_o[i][jj][e] = g;
            s1[i] += g;
            s2[i] += g * (((float)x4[e] - st[i][0]) * st[i][1]);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        s1[i] += __shfl_xor(s1[i], 16, 64), s2[i] += __shfl_xor(s2[i], 16, 64);
        s1[i] += __shfl_xor(s1[i], 32, 64), s2[i] += __shfl_xor(s2[i], 32, 64);
      }
      f32x2_t* const red = reinterpret_cast<f32x2_t*>(smem_raw);   // [8 waves][2][16] (the ring slots are free: every wave has left the main loop ...)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // ... after this barrier
      if (lq == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) red[(wave * 2 + i) * 16 + l15] = f32x2_t{s1[i], s2[i]};
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      constexpr float inv_c = 1.f / (float)FF_C;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const f32x2_t a = red[((wm * 2 + 0) * 2 + i) * 16 + l15], b = red[((wm * 2 + 1) * 2 + i) * 16 + l15];
        const float m1 = (a[0] + b[0]) * inv_c, m2 = (a[1] + b[1]) * inv_c;
#pragma unroll
        for (int jj = 0; jj < 10; ++jj) {
          f16x4 o, r4 = {0, 0, 0, 0};
          if (p.R) r4 = *(const f16x4*)((const f16*)p.R + (mrow + 16 * i) * p.ldr + ncol + 16 * jj);
          const f16x4 x4 = *(const f16x4*)((const f16*)p.ln_x + (mrow + 16 * i) * p.ld_lnx + ncol + 16 * jj);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xh = ((float)x4[e] - st[i][0]) * st[i][1];
            const float v = st[i][1] * (acc_o[i][jj][e] - m1 - xh * m2) + (float)r4[e];
            o[e] = (f16)v;
          }
          *(f16x4*)((f16*)p.Y + (mrow + 16 * i) * p.ldy + ncol + 16 * jj) = o;
        }
      }
      return;
    }
  }
  const uint32_t lds0c = ff_lds_addr(smem_raw);
  f16x4 qr[10][2];   // POST: proj_out's residual (the block input) for this lane's (row, tile) units -- requested here, waited for with rv below
  if constexpr (POST) {
    // every wave has left the main loop (ring slot 1 and the exchange buffer lie under the image) and the dummy prefetch into slot 0 has landed
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int tl = 0; tl < 10; ++tl)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        qr[tl][i] = p.post_R ? *(const f16x4*)((const f16*)p.post_R + (mrow + 16 * i) * p.ld_postr + tl * 32 + wn * 16 + 4 * lqe) : f16x4{0, 0, 0, 0};
    if (t < FF_C) reinterpret_cast<float*>(smem_raw + FFC_CB_OFF)[t] = p.post_b ? p.post_b[t] : 0.f;   // (t: live anyway)
  }
  f16x4 rv[2][10];
  if (p.R) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jj = 0; jj < 10; ++jj) rv[i][jj] = *(const f16x4*)((const f16*)p.R + (mrow + 16 * i) * p.ldr + ncol + 16 * jj);
  }
  const uint32_t img_o = lds0c + FFC_IMG + (wm * 32 + l15e) * FFC_PITCH + (wn * 160 + 4 * lqe) * 2;   // + i * 16 rows, + 32 B per jj
#pragma unroll
  for (int jj = 0; jj < 10; ++jj) {
    f32x4_t b = {0.f, 0.f, 0.f, 0.f};
    if (!BWD && p.b2) b = *(const f32x4_t*)(p.b2 + ncol + 16 * jj);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = acc_o[i][jj][e] + b[e];
        if (p.R) v += (float)rv[i][jj][e];
        o[e] = (f16)v;
      }
      if (!POST || p.Y) *(f16x4*)((f16*)p.Y + (mrow + 16 * i) * p.ldy + ncol + 16 * jj) = o;
      if constexpr (POST) {
        if (i == 0) ff_write8<0>(img_o + jj * 32, o);
        else ff_write8<16 * FFC_PITCH>(img_o + jj * 32, o);
      }
    }
  }
  if constexpr (POST) {
    // ---- proj_out + bias + the block input: t3 (fp16, as the separate launch would have read it back) -> operand -> ten tiles -> the image again
    // (the output tile by tile, fp16) -> post_Y in whole rows
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the image is complete; nothing of this wave is on the memory queue
    const FfChain ch(smem_raw, p.post_W, p.ld_postw, wave, lane);
    ch.issue(0);
    ch.issue(1);
    {
      const uint32_t img_x = lds0c + FFC_IMG + (wm * 32 + l15e) * FFC_PITCH + lqe * 16;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < 10; ++ks) xf[i][ks] = ff_read16<0>(img_x + i * 16 * FFC_PITCH + ks * 64);
      ff_wait_lgkm<0>();
      FF_SB();
    }
    const uint32_t fa = lds0c + (wn * 16 + l15e) * 128 + ((lqe ^ (l15e & 7)) << 4);
    const uint32_t img_w = lds0c + FFC_IMG + (wm * 32 + l15e) * FFC_PITCH + (wn * 16 + 4 * lqe) * 2;
    const uint32_t cb_a = lds0c + FFC_CB_OFF + (wn * 16 + 4 * lqe) * 4;
    auto run = [&](auto tlc) {
      constexpr int tile = decltype(tlc)::value;
      // (tile 0's barrier also orders every wave's operand reads of the image in front of the first output write into it)
      if constexpr (tile + 1 < 10) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if constexpr (tile + 2 < 10) ch.issue(tile + 2);
      FF_SB();
      const f32x4_t pb = ff_read16f<tile * 128>(cb_a);
      f32x4_t acc[2];
      ff_chain_tile(fa + (tile % 3) * FFC_WT, xf, acc);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (f16)(acc[i][e] + pb[e] + (float)qr[tile][i][e]);
        if (i == 0) ff_write8<0>(img_w + tile * 64, o);
        else ff_write8<16 * FFC_PITCH>(img_w + tile * 64, o);
      }
    };
    run(std::integral_constant<int, 0>{}); run(std::integral_constant<int, 1>{}); run(std::integral_constant<int, 2>{});
    run(std::integral_constant<int, 3>{}); run(std::integral_constant<int, 4>{}); run(std::integral_constant<int, 5>{});
    run(std::integral_constant<int, 6>{}); run(std::integral_constant<int, 7>{}); run(std::integral_constant<int, 8>{});
    run(std::integral_constant<int, 9>{});
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the output image is complete
    {
      int te = t;
      if constexpr (PRE && POST) asm volatile("" : "+v"(te));
      const int lrow = te >> 2, lqr = te & 3;
      const uint32_t la = lds0c + FFC_IMG + lrow * FFC_PITCH + lqr * 160;
      f16x8 v[10];
#pragma unroll
      for (int j = 0; j < 10; ++j) v[j] = ff_read16<0>(la + 16 * j);
      ff_wait_lgkm<0>();
      FF_SB();
      f16* yp = (f16*)p.post_Y + (m0 + lrow) * p.ld_posty + lqr * 80;
#pragma unroll
      for (int j = 0; j < 10; ++j) *(f16x8*)(yp + 8 * j) = v[j];
    }
  }
}

template <bool BWD, bool PRE = false, bool POST = false>
int ff_launch(const tb_ff_desc& d, hipStream_t s) {
  constexpr int LDS = (PRE || POST) ? FFC_LDS : FF_LDS;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)ff_fused_kernel<BWD, PRE, POST>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return TB_ELAUNCH;
    attr_done = true;
  }
  hipLaunchKernelGGL((ff_fused_kernel<BWD, PRE, POST>), dim3((unsigned)(d.M / FF_BM)), dim3(512), LDS, s, d);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

int ff_check(const tb_ff_desc* d, bool bwd) {
  if (!d || !tb_ff_fused_ok(d->M, d->C, d->inner)) return TB_EINVAL;
  if (!d->X || !d->W1 || !d->W2 || !d->HG || (!d->Y && !(d->post_W && !bwd))) return TB_EINVAL;
  if (bwd && (d->pre_W || d->post_W)) return TB_EINVAL;   // the chained stages exist in the forward only
  if (d->pre_W && (!d->pre_Y || !d->pre_gamma || !d->pre_beta || d->ld_prew % 8 || d->ld_prey % 8 || (d->pre_R && d->ld_prer % 4) ||
                   ((uintptr_t)d->pre_W) % 16 || ((uintptr_t)d->pre_Y) % 16 || ((uintptr_t)d->pre_R) % 8 || ((uintptr_t)d->pre_b) % 16 ||
                   ((uintptr_t)d->pre_gamma) % 16 || ((uintptr_t)d->pre_beta) % 16 || ((uintptr_t)d->pre_stats) % 8 ||
                   (int64_t)FF_C * d->ld_prew * 2 >= ((int64_t)1 << 31)))
    return TB_EINVAL;
  if (d->post_W && (!d->post_Y || d->ld_postw % 8 || d->ld_posty % 8 || (d->post_R && d->ld_postr % 4) || ((uintptr_t)d->post_W) % 16 ||
                    ((uintptr_t)d->post_Y) % 16 || ((uintptr_t)d->post_R) % 8 || ((uintptr_t)d->post_b) % 16 ||
                    (int64_t)FF_C * d->ld_postw * 2 >= ((int64_t)1 << 31)))
    return TB_EINVAL;
  if (d->ldx % 8 || d->ldw1 % 8 || d->ldw2 % 8 || d->ldhg % 4 || (d->Y && d->ldy % 4) || (d->R && d->ldr % 4)) return TB_EINVAL;
  if (((uintptr_t)d->X) % 16 || ((uintptr_t)d->W1) % 16 || ((uintptr_t)d->W2) % 16 || ((uintptr_t)d->HG) % 8 || ((uintptr_t)d->Y) % 8 ||
      ((uintptr_t)d->R) % 8 || ((uintptr_t)d->b2) % 16)
    return TB_EINVAL;
  // 32-bit per-lane byte offsets into the weight operands
  const int64_t lim = (int64_t)1 << 32;
  if ((bwd ? FF_INNER : 2 * FF_INNER) * d->ldw1 * 2 >= lim || (int64_t)FF_C * d->ldw2 * 2 >= lim) return TB_EINVAL;
  return TB_OK;
}

}  // namespace

extern "C" int tb_ff_debug(void* buf16) {   // profiling builds only (FF_PROF): 16 x uint64 device buffer, NULL = off
  unsigned long long* v = (unsigned long long*)buf16;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_ff_dbg), &v, sizeof(v)) == hipSuccess ? TB_OK : TB_ELAUNCH;
}

extern "C" int tb_ff_fused_ok(int64_t M, int C, int inner) { return M > 0 && M % FF_BM == 0 && C == FF_C && inner == FF_INNER; }

extern "C" int tb_ff_fwd(const tb_ff_desc* d, tb_stream_t stream) {
  (void)hipGetLastError();
  const int r = ff_check(d, false);
  if (r != TB_OK) return r;
  if (d->pre_W && d->post_W) return ff_launch<false, true, true>(*d, (hipStream_t)stream);
  if (d->pre_W) return ff_launch<false, true, false>(*d, (hipStream_t)stream);
  if (d->post_W) return ff_launch<false, false, true>(*d, (hipStream_t)stream);
  return ff_launch<false>(*d, (hipStream_t)stream);
}

extern "C" int tb_ff_bwd(const tb_ff_desc* d, tb_stream_t stream) {
  (void)hipGetLastError();
  const int r = ff_check(d, true);
  if (r != TB_OK) return r;
  return ff_launch<true>(*d, (hipStream_t)stream);
}
