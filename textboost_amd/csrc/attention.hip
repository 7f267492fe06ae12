// Flash attention forward + backward (dQ, dK, dV) for gfx950, fp16 in / fp32 accumulate, 32x32x16 MFMA.
//
// Serves the three attention shapes of the TextBoost step:
//   * UNet self-attention   (diffusers Attention + AttnProcessor2_0 -> F.scaled_dot_product_attention),
//     seq 4096/1024/256/64, head dim 40/80/160, non-causal          train_textboost.py:1063-1067
//   * UNet cross-attention  (K/V = 77 text tokens)                   same call site
//   * CLIP text self-attention (causal, 77 tokens, head dim 64)      train_textboost.py:1054-1059, :1099-1100
// and their autograd (accelerator.backward, :1108).
//
// Layout trick (cdna guide "swapped QK^T"): scores are computed TRANSPOSED, S^T = K Q^T, so each lane owns one
// query row (col = lane & 31) and the softmax statistics m, l, lse, delta are lane-local.  P^T is then already
// in B-operand layout for O^T = V^T P^T; the k-index permutation the accumulator layout implies is absorbed by
// reading the A operand (V^T, staged transposed in LDS) with the same permutation, so no cross-lane traffic is
// needed between the two matmuls.  The backward kernels use the same idea (lane owns a query in the dQ kernel,
// a key in the dK/dV kernel).
//
// The softmax side of these kernels is VALU-bound at the UNet's small head dims (rocprofv3 PMC, S = 4096, hd = 40: VALU busy 65 %,
// MFMA busy 28 %), so everything that can ride on the matrix pipe does:
//   * the lane-owned operand (Q, or K in the dK/dV kernel) is pre-multiplied by scale*log2(e) once, and the MFMA accumulator is
//     initialised with -m (forward), -lse (backward) or -delta (the dP product) instead of zero, so S arrives as the exp2 argument
//     and dP arrives as dP - delta: no per-score fma / subtract;
//   * the forward row sum l is an extra all-ones row of V^T in the head-dim padding (hd < 32*DT): O^T row hd accumulates sum_k P;
//   * the softmax scale of dS is applied once to the dQ / dK accumulators at write-out (as FlashAttention-2 does).
#define TB_ATTN_FUSED_DELTA 1  // A/B on MI355X: delta = rowsum(dO*O) inside the dQ kernel, +1.0 % steps/s vs its own launch
#include "common.h"
#include "../../include/textboost_hip.h"
#include "attn_il.h"

namespace {

constexpr int KVT = 64;      // keys (or queries, in the dK/dV kernel) per LDS tile
constexpr int TLD = KVT + 4; // row stride (halfs) of transposed tiles: 136 B -> conflict-free ds_read_b64 across d
// The transposed STORE (ds_write_b64, lanes = 8 column chunks x 8 row groups) hits each bank pair 4 times with that stride alone: rows 16
// apart (chunks c, c+2) land on the same banks (16 * TLD * 2 B = 16 words mod 32).  Every group of 16 rows is therefore skewed by another
// 4 words (TSK halfs): stores and reads are both conflict-free, and on the read side the skew is part of the lane's base address / the
// compile-time row-block offset -- no extra register or VALU op in the MFMA loops (an XOR swizzle made the forward kernel spill: +30 %).
// PMC before: SQ_LDS_BANK_CONFLICT = 192 cycles per K/V tile = 28 % of the forward kernel's LDS-active cycles.
#ifndef TB_ATTN_TRSKEW
#define TB_ATTN_TRSKEW 1
#endif
constexpr int TSK = TB_ATTN_TRSKEW ? 8 : 0;
__device__ __forceinline__ int tr_row(int d) { return d * TLD + (d >> 4) * TSK; }  // offset (halfs) of row d of a transposed tile
constexpr float LOG2E = 1.4426950408889634f;
__device__ __attribute__((aligned(16))) const f16 g_zero8[8] = {};  // out-of-range lanes load this line: no divergent branches
// single v_exp_f32 (no denormal-range fixup: softmax probabilities below 2^-126 may flush to 0)
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Workgroup -> (x block, head, batch).  The dispatcher deals consecutive workgroups round-robin over the 8 XCDs, so with the plain
// (x, h, b) grid the x blocks of ONE (batch, head) -- which all stream the same K / V (or Q / dO) rows -- land on 8 different L2s and every
// XCD sees every head's operands (42 MB at 64x64 maps against 4 MB of L2): the tile loads then run at the ~6.5 TB/s of the fabric.  With
// remap, XCD i owns the (batch, head) pairs i, i + 8, ..: all x blocks of a pair share one L2.  Only speed depends on the placement.
struct AttnBlk {
  int x, h, b;
};
__device__ __forceinline__ AttnBlk attn_block(int remap) {
  AttnBlk r;
  const int gx = gridDim.x, H = gridDim.y, B = gridDim.z;
  if (remap && ((H * B) & 7) == 0) {
    const int lin = blockIdx.x + gx * (blockIdx.y + H * blockIdx.z);
    const int xcd = lin & 7, k = lin >> 3;
    const int pair = (k / gx) * 8 + xcd;
    r.x = k - (k / gx) * gx;
    r.h = pair % H;
    r.b = pair / H;
  } else {
    r.x = blockIdx.x, r.h = blockIdx.y, r.b = blockIdx.z;
  }
  return r;
}

template <int WD>
struct RM {  // row-major [64][WD] tile, rows padded by 16 B
  static constexpr int LD = WD + 8;
  static constexpr int SIZE = KVT * LD;
};
template <int WD>
struct TR {  // transposed [WD][64] tile
  static constexpr int SIZE = WD * TLD + (WD / 16) * TSK;
};

// Staging of rows [row0, row0+64) x cols [0, WD) of a (rows x hd) matrix (row stride ld) into LDS, zero outside, split in
// two halves so the global loads of tile t+1 are in flight while tile t is multiplied (cdna guide T14):
//   tile_load : global -> registers.  One work item = 4 consecutive rows x one 16-byte column chunk.
//   tile_store: registers -> LDS, row-major (4 x ds_write_b128) and/or transposed (8 x ds_write_b64: for each of the 8
//               columns the 4 consecutive rows are adjacent in the [WD][64] transposed image).
template <int WD>
struct TileRegs {
  static constexpr int CPR = WD / 8;                  // 16-byte chunks per row
  static constexpr int ITEMS = (KVT / 4) * CPR;       // work items per tile
  static constexpr int NI = (ITEMS + 255) / 256;      // items per thread
  f16x8 v[NI][4];
  uint32_t off[NI];  // loop-invariant byte offset of this thread's first row / chunk from the tile's first row (tile_init)
};
// per-thread source offset of a tile: kg*4 * ld + chunk column, in bytes.  With it a full tile's loads are
// `global_load_dwordx4 v, v_off, s[base]` with the tile's row term (and the k-th row of the item) in the uniform base: no vector ALU
// work per load.
template <int WD>
__device__ __forceinline__ void tile_init(TileRegs<WD>& t, int64_t ld, int hd) {
  constexpr int CPR = TileRegs<WD>::CPR;
#pragma unroll
  for (int it = 0; it < TileRegs<WD>::NI; ++it) {
    const int idx = threadIdx.x + it * 256;
    const int kg = idx / CPR, ch = idx - kg * CPR;
    t.off[it] = (uint32_t)((kg * 4 * ld + (ch * 8 < hd ? ch * 8 : 0)) * 2);
  }
}
typedef const __attribute__((address_space(1))) f16x8* gvec8_t;  // explicit global address space: a select against the zero line must
                                                                  // not degrade the loads to flat_load (which also ticks lgkmcnt and
                                                                  // would make every LDS wait drain the prefetch)
// FAST (narrow heads, DT <= 2): rows past the end of the matrix are CLAMPED to its last row and chunks in the head-dim padding
// (ch*8 >= hd) read the row's first chunk; nothing is zero-filled.  Safe because every consumer neutralises them -- scores of keys >= Skv /
// queries >= Sq are masked to p = 0 (p * V_clamped = 0 with finite V), and padding columns only meet zero-filled fragments of the
// lane-owned operand (S, dP) or land in output rows >= hd that are never stored.  It buys: no per-row tests, no select, one address form
// (loop-invariant per-lane offset + wave-uniform row term).  The wide-head instantiations (AGPR-bound, one wave per SIMD) measured
// 5-7 % slower with it and keep the zero-line select (L1 self-attention backward 165 vs 177 us).
template <int WD, bool FAST>
__device__ __forceinline__ void tile_load(TileRegs<WD>& t, const f16* g, int64_t ld, int row0, int nrows, int hd) {
  constexpr int CPR = TileRegs<WD>::CPR;
  const bool full = row0 + KVT <= nrows;  // wave-uniform: every row of the tile exists
  const int rmax = nrows - 1 - row0;      // last existing row, tile-local (>= 0: callers never start a tile past the end)
#pragma unroll
  for (int it = 0; it < TileRegs<WD>::NI; ++it) {
    const int idx = threadIdx.x + it * 256;
    const int kg = idx / CPR, ch = idx - kg * CPR;
    if (FAST) {
      if (idx < TileRegs<WD>::ITEMS) {
        if (full) {  // wave-uniform branch: uniform base + precomputed 32-bit offsets
          const char* base = (const char*)(g + (int64_t)row0 * ld);
#pragma unroll
          for (int k = 0; k < 4; ++k) t.v[it][k] = *(gvec8_t)(base + (int64_t)k * ld * 2 + t.off[it]);
        } else {     // ragged last tile: clamp the rows
          const f16* base = g + (int64_t)row0 * ld + (ch * 8 < hd ? ch * 8 : 0);
#pragma unroll
          for (int k = 0; k < 4; ++k) t.v[it][k] = *(gvec8_t)(base + (int64_t)min(kg * 4 + k, rmax) * ld);
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int row = row0 + kg * 4 + k;
        const bool ok = idx < TileRegs<WD>::ITEMS && row < nrows && ch * 8 < hd;
        t.v[it][k] = *(ok ? (gvec8_t)(g + (int64_t)row * ld + ch * 8) : (gvec8_t)g_zero8);
      }
    }
  }
}
// ONES: row `ones_row` (a multiple of 8 inside the zero padding hd..WD-1) of the transposed image is all ones
// ONECOL: column `ones_row` (a multiple of 8 inside the padding hd..WD-1) of the row-major image is all ones (attn_fwd_kernel FOLDM)
template <int WD, bool ROWMAJOR, bool TRANSPOSED, bool ONES = false, bool ONECOL = false>
__device__ __forceinline__ void tile_store(const TileRegs<WD>& t, f16* rm, f16* tr, int ones_row = -1) {
  constexpr int CPR = TileRegs<WD>::CPR;
#pragma unroll
  for (int it = 0; it < TileRegs<WD>::NI; ++it) {
    const int idx = threadIdx.x + it * 256;
    if (idx >= TileRegs<WD>::ITEMS) continue;
    const int kg = idx / CPR, ch = idx - kg * CPR;
    if (ROWMAJOR) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        f16x8 v = t.v[it][k];
        if (ONECOL && ch * 8 == ones_row) v[0] = (f16)1.f;
        *(f16x8*)(rm + (kg * 4 + k) * RM<WD>::LD + ch * 8) = v;
      }
    }
    if (TRANSPOSED) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        f16x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = t.v[it][k][i];
        if (ONES && i == 0 && ch * 8 == ones_row) o = f16x4{(f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f};
        *(f16x4*)(tr + tr_row(ch * 8 + i) + kg * 4) = o;
      }
    }
  }
}

template <int WD>
__device__ __forceinline__ f16x8 frag_rm(const f16* rm, int row, int chunk) {
  return *(const f16x8*)(rm + row * RM<WD>::LD + chunk * 8);
}
// A-operand fragment from a transposed tile: row 32 dblk + l31, contraction indices {k0+4hi..+3, k0+8+4hi..+3}.  tr_row(32 dblk + l31) =
// tr_row(l31) + dblk * (32 TLD + 2 TSK): one per-lane base, everything else is an immediate of the unrolled loops.
__device__ __forceinline__ f16x8 frag_tr(const f16* tr, int dblk, int l31, int k0, int hi) {
  const f16* row = tr + tr_row(l31) + 4 * hi + (dblk * (32 * TLD + 2 * TSK) + k0);
  const f16x4 a = *(const f16x4*)(row);
  const f16x4 b = *(const f16x4*)(row + 8);
  f16x8 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    o[e] = a[e];
    o[4 + e] = b[e];
  }
  return o;
}
__device__ __forceinline__ f16x8 pack8(const f32x16& v, int r0) {
  f16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (f16)v[r0 + e];
  return o;
}
// B-operand (k = head-dim) register fragments of one row of a (rows x hd) matrix: chunk j covers cols 16j+8hi..+8
template <int KS>
__device__ __forceinline__ void load_row_frags(f16x8* f, const f16* g, int64_t ld, int row, int nrows, int hd, int hi) {
#pragma unroll
  for (int j = 0; j < KS; ++j) {
    const int col = 16 * j + 8 * hi;
    f[j] = *((row < nrows && col < hd) ? (gvec8_t)(g + (int64_t)row * ld + col) : (gvec8_t)g_zero8);
  }
}

#ifdef TB_ATTN_PRIO
#define TB_PRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define TB_PRIO(x)
#endif
#define ZERO16(x)                \
  _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) (x)[r_] = 0.f;
#define FILL16(x, v)             \
  _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) (x)[r_] = (v);

__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }  // v_max3_f32
// lane-owned B-operand fragments times a scalar (scale * log2 e), rounded back to fp16
template <int KS>
__device__ __forceinline__ void scale_frags(f16x8* f, float c) {
#pragma unroll
  for (int j = 0; j < KS; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) f[j][e] = (f16)((float)f[j][e] * c);
}

// ------------------------------------------------------------------------------------------------ forward
#ifndef TB_ATTN_FOLDM
#define TB_ATTN_FOLDM 1
#endif
#ifndef TB_ATTN_REBASE
#define TB_ATTN_REBASE 8.f
#endif
#ifdef TB_FWD_OCC2
#define TB_FWD_OCC3 0
#else
#define TB_FWD_OCC3 1
#endif
// FOLDM (hd a multiple of 8 with hd < 16 KS, i.e. SD1.x's hd = 40): the running max enters the score product through the head-dim PADDING --
// K's padding column hd is all ones in LDS and the lane-owned Q fragment carries -m (rounded to fp16, and m is kept fp16-representable so l, the
// re-base factor and the stored LSE all use exactly the value that was subtracted) in that slot -- so the score accumulators start from the
// inline constant 0: no 16-register -m tuple (the 168-VGPR kernel spilled with it) and no 32 accumulator-init moves per tile (PMC: 32 of the
// ~115 non-MFMA VALU ops per wave-tile).
template <int DT, int KS, bool ONES, bool FOLDM>
__global__ __launch_bounds__(256, (DT <= 2 && KS <= 3 && TB_FWD_OCC3 ? 3 : (DT <= 4 ? 2 : 1))) void attn_fwd_kernel(const tb_attn_desc p, int remap) {  // 3 blocks/CU only where 168 VGPRs hold without spills
  constexpr int WD = DT * 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  f16* Ks = reinterpret_cast<f16*>(smem_raw);
  f16* Vt = Ks + RM<WD>::SIZE;
  const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform (SGPR): the mask tests below become scalar branches
  const AttnBlk blk = attn_block(remap);
  const int b = blk.b, h = blk.h;
  const int qblk = blk.x * 128;
  const int q = qblk + wave * 32 + l31;
  const f16* Qg = (const f16*)p.Q + (int64_t)b * p.Sq * p.ldq + h * p.hd;
  const f16* Kg = (const f16*)p.K + (int64_t)b * p.Skv * p.ldk + h * p.hd;
  const f16* Vg = (const f16*)p.V + (int64_t)b * p.Skv * p.ldv + h * p.hd;
  f16x8 qf[KS];
  load_row_frags<KS>(qf, Qg, p.ldq, q, p.Sq, p.hd, hi);
  scale_frags<KS>(qf, p.scale * LOG2E);  // scores come out of the MFMA in the log2 domain
  f32x16 o[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d) ZERO16(o[d]);
  // running max m (log2 domain) enters the score MFMA as its accumulator input: s' = c * q.k - m is the exp2 argument
  float m = 0.f, l = 0.f;
  f32x16 negm;
  if (!FOLDM) ZERO16(negm);
  // FOLDM: column hd of the lane's Q row lives in element 0 of chunk hd / 16 of the hi == (hd / 8) & 1 lanes
  const int fj = p.hd >> 4;
  const bool fold_lane = FOLDM && hi == ((p.hd >> 3) & 1);
  int kv_end = p.Skv;
  if (p.causal) kv_end = min(p.Skv, qblk + 128);  // keys beyond the block's last query are never visible
  // K / V^T tiles are double-buffered in LDS (PF: DT <= 3): tile t+1 is written (from the registers its global loads landed in during
  // tile t-1) while tile t is multiplied, so one barrier per tile suffices and the LDS stores overlap the MFMAs
  constexpr bool PF = DT <= 3;  // wide heads keep the single-buffered, unprefetched schedule: the registers are needed for O
  constexpr int TILE = RM<WD>::SIZE + TR<WD>::SIZE;
  TileRegs<WD> kreg, vreg;
  tile_init<WD>(kreg, p.ldk, p.hd);
  tile_init<WD>(vreg, p.ldv, p.hd);
  if (PF) {
    tile_load<WD, (DT <= 2)>(kreg, Kg, p.ldk, 0, p.Skv, p.hd);
    tile_load<WD, (DT <= 2)>(vreg, Vg, p.ldv, 0, p.Skv, p.hd);
    tile_store<WD, true, false, false, FOLDM>(kreg, Ks, nullptr, p.hd);
    tile_store<WD, false, true, ONES>(vreg, nullptr, Vt, p.hd);
    if (KVT < kv_end) {
      tile_load<WD, (DT <= 2)>(kreg, Kg, p.ldk, KVT, p.Skv, p.hd);
      tile_load<WD, (DT <= 2)>(vreg, Vg, p.ldv, KVT, p.Skv, p.hd);
    }
    __syncthreads();
  }
  int cur = 0;
  for (int kv0 = 0; kv0 < kv_end; kv0 += KVT) {
    if (PF) {
      Ks = reinterpret_cast<f16*>(smem_raw) + cur * TILE;
      Vt = Ks + RM<WD>::SIZE;
      if (kv0 + KVT < kv_end) {  // next tile -> the other buffer (every wave left it at the barrier that ended the previous iteration)
        f16* Kn = reinterpret_cast<f16*>(smem_raw) + (cur ^ 1) * TILE;
        tile_store<WD, true, false, false, FOLDM>(kreg, Kn, nullptr, p.hd);
        tile_store<WD, false, true, ONES>(vreg, nullptr, Kn + RM<WD>::SIZE, p.hd);
        if (kv0 + 2 * KVT < kv_end) {
          tile_load<WD, (DT <= 2)>(kreg, Kg, p.ldk, kv0 + 2 * KVT, p.Skv, p.hd);
          tile_load<WD, (DT <= 2)>(vreg, Vg, p.ldv, kv0 + 2 * KVT, p.Skv, p.hd);
        }
      }
    } else {
      __syncthreads();
      tile_load<WD, (DT <= 2)>(kreg, Kg, p.ldk, kv0, p.Skv, p.hd);
      tile_load<WD, (DT <= 2)>(vreg, Vg, p.ldv, kv0, p.Skv, p.hd);
      tile_store<WD, true, false, false, FOLDM>(kreg, Ks, nullptr, p.hd);
      tile_store<WD, false, true, ONES>(vreg, nullptr, Vt, p.hd);
      __syncthreads();
    }
    f32x16 s[2];
    TB_PRIO(1);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      if (FOLDM) {
        ZERO16(s[kt]);  // the inline constant 0 as the first MFMA's accumulator input
      } else {
        s[kt] = negm;
      }
#pragma unroll
      for (int j = 0; j < KS; ++j)
        s[kt] = TB_MFMA_32x32x16(frag_rm<WD>(Ks, kt * 32 + l31, 2 * j + hi), qf[j], s[kt]);
    }
    TB_PRIO(0);
    // masking is needed only on the ragged last tile / the causal diagonal: wave-uniform test keeps it off the common path
    const bool need_mask = (kv0 + KVT > p.Skv) || (p.causal && kv0 + KVT - 1 > qblk + wave * 32);
    if (need_mask) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kv0 + kt * 32 + mfma32_row(r, hi);
          const bool ok = key < p.Skv && (!p.causal || key <= q);
          s[kt][r] = ok ? s[kt][r] : -INFINITY;
        }
    }
    float mx0 = s[0][0], mx1 = s[1][0];  // excess of this tile's scores over the running max
#pragma unroll
    for (int r = 1; r < 15; r += 2) {
      mx0 = max3f(mx0, s[0][r], s[0][r + 1]);
      mx1 = max3f(mx1, s[1][r], s[1][r + 1]);
    }
    float mx = max3f(mx0, mx1, fmaxf(s[0][15], s[1][15]));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const bool first = kv0 == 0;  // key 0 is visible to every query, so the first tile always has a finite maximum
    // Lazy re-base: the reference point m only has to stay within 2^REBASE of the true row maximum (p <= 2^REBASE = 256 is exact in the fp32
    // accumulators and well inside fp16 for the P operand; l and the stored LSE use the same m).  With "any excess > 0" 67 % of the L0 tiles
    // re-based (one of a wave's 32 rows sees a new maximum with probability ~ 1 - (1 - 1/t)^32 at tile t): ~70 VALU ops each (PMC:
    // SQ_INSTS_VALU_MUL_F32 / ADD_F32 ~ 11 per wave-tile).
    if (first || __any(mx > TB_ATTN_REBASE)) {
      float d = first ? mx : fmaxf(mx, 0.f);
      if (FOLDM) d = (float)(f16)(m + d) - m;  // keep m fp16-representable: the Q slot subtracts exactly m (the difference is exact in fp32)
      if (!first) {
        const float alpha = fast_exp2(-d);
        l *= alpha;
#pragma unroll
        for (int dd = 0; dd < DT; ++dd)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[dd][r] *= alpha;
      }
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kt][r] -= d;
      m += d;
      if (FOLDM) {
#pragma unroll
        for (int j = 0; j < KS; ++j)
          if (j == fj && fold_lane) qf[j][0] = (f16)(-m);
      } else {
        FILL16(negm, -m);
      }
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = fast_exp2(s[kt][r]);
        s[kt][r] = pv;
        if (!ONES) l += pv;
      }
    TB_PRIO(1);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const f16x8 pf = pack8(s[kt], 8 * jj);
#pragma unroll
        for (int d = 0; d < DT; ++d)
          o[d] = TB_MFMA_32x32x16(frag_tr(Vt, d, l31, kt * 32 + 16 * jj, hi), pf, o[d]);
      }
    TB_PRIO(0);
    if (PF) {
      __syncthreads();
      cur ^= 1;
    }
  }
  if (ONES) {
    // the all-ones row hd of V^T made O^T row hd the row sum; it sits in register 4*g of the hi == 0 lane of tile hd / 32
    float ls = 0.f;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (d * 32 + g * 8 == p.hd) ls = o[d][4 * g];
    l = __shfl(ls, l31, 64);
  } else {
    l += __shfl_xor(l, 32, 64);
  }
  if (q < p.Sq) {
    const float inv = 1.f / l;
    f16* Og = (f16*)p.O + ((int64_t)b * p.Sq + q) * p.ldo + h * p.hd;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int col = d * 32 + 8 * r4 + 4 * hi;  // 4 consecutive head-dim columns
        if (col < p.hd) {
          f16x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (f16)(o[d][4 * r4 + e] * inv);
          *(f16x4*)(Og + col) = v;
        }
      }
    if (p.LSE && hi == 0) p.LSE[((int64_t)b * p.H + h) * p.Sq + q] = m * (1.f / LOG2E) + logf(l);
  }
}

// ------------------------------------------------------------------------------------------------ forward, short key sequences
// Cross-attention on the prompt (train_textboost.py:1063-1067: 77 keys per head against 4096 / 1024 queries): Skv <= 96, non-causal, DT <= 2.
// In attn_fwd_kernel such a launch is two 64-key tiles (the second one 80 % masked) with a barrier, a V transpose and a re-base test per tile,
// for 128 queries per workgroup -- 29 us for 42 MB of Q / O traffic at the 64x64 maps.  Here the K / V^T images of ALL keys are staged once,
// K is held as MFMA operand fragments in registers, and each wave walks `qtiles` 32-query tiles with no barrier and a plain (single-pass)
// softmax; the next tile's Q fragments are requested before the current tile's products.  Same arithmetic as attn_fwd_kernel up to the
// reference point of the exponentials (the row maximum itself instead of a lazily re-based one): O to 1 fp16 ulp, LSE to ~1e-6.
template <int DT, int KS, int QT>
__global__ __launch_bounds__(256, 2) void attn_xs_fwd_kernel(const tb_attn_desc p, int remap) {
  constexpr int qtiles = QT;
  constexpr int WD = DT * 32;
  constexpr int TILE = RM<WD>::SIZE + TR<WD>::SIZE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  f16* const K0 = reinterpret_cast<f16*>(smem_raw);
  f16* const V0 = K0 + RM<WD>::SIZE;
  f16* const K1 = K0 + TILE;
  f16* const V1 = K1 + RM<WD>::SIZE;
  const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // Workgroup -> (query band, head, batch).  The K / V of a head are a few KB here; what matters is Q and O: a head's slice is hd of the H * hd
  // halfs of a row, so the H heads of one query band read and write the SAME cache lines.  remap: XCD i takes the (band, batch) pairs i, i + 8, ..
  // and runs their H heads back to back -- every line is fetched into ONE L2 and written back whole (attn_block's pairing, made for the long-key
  // kernels, puts the heads of a band on eight different XCDs).
  AttnBlk blk;
  {
    const int gx = gridDim.x, H = gridDim.y, B = gridDim.z;
    if (remap && ((gx * B) & 7) == 0) {
      const int lin = blockIdx.x + gx * (blockIdx.y + H * blockIdx.z);
      const int xcd = lin & 7, k = lin >> 3;
      const int xb = (k / H) * 8 + xcd;
      blk.h = k - (k / H) * H;
      blk.x = xb % gx;
      blk.b = xb / gx;
    } else {
      blk.x = blockIdx.x, blk.h = blockIdx.y, blk.b = blockIdx.z;
    }
  }
  const int b = blk.b, h = blk.h;
  const int qblk = blk.x * 128 * qtiles;
  const f16* Qg = (const f16*)p.Q + (int64_t)b * p.Sq * p.ldq + h * p.hd;
  const f16* Kg = (const f16*)p.K + (int64_t)b * p.Skv * p.ldk + h * p.hd;
  const f16* Vg = (const f16*)p.V + (int64_t)b * p.Skv * p.ldv + h * p.hd;
  const bool three = p.Skv > KVT;   // keys 64 .. 95 exist
  // the Q fragments of ALL of the wave's tiles are requested first: one exposed HBM round trip per wave, under the K / V staging
  f16x8 qall[QT][KS];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) load_row_frags<KS>(qall[qt], Qg, p.ldq, qblk + (qt * 4 + wave) * 32 + l31, p.Sq, p.hd, hi);
  {
    TileRegs<WD> kreg, vreg;
    tile_init<WD>(kreg, p.ldk, p.hd);
    tile_init<WD>(vreg, p.ldv, p.hd);
    tile_load<WD, (DT <= 2)>(kreg, Kg, p.ldk, 0, p.Skv, p.hd);
    tile_load<WD, (DT <= 2)>(vreg, Vg, p.ldv, 0, p.Skv, p.hd);
    tile_store<WD, true, false>(kreg, K0, nullptr);
    tile_store<WD, false, true>(vreg, nullptr, V0);
    if (three) {
      tile_load<WD, (DT <= 2)>(kreg, Kg, p.ldk, KVT, p.Skv, p.hd);
      tile_load<WD, (DT <= 2)>(vreg, Vg, p.ldv, KVT, p.Skv, p.hd);
      tile_store<WD, true, false>(kreg, K1, nullptr);
      tile_store<WD, false, true>(vreg, nullptr, V1);
    }
  }
  __syncthreads();
  f16x8 kf[3][KS];
#pragma unroll
  for (int j = 0; j < KS; ++j) {
    kf[0][j] = frag_rm<WD>(K0, l31, 2 * j + hi);
    kf[1][j] = frag_rm<WD>(K0, 32 + l31, 2 * j + hi);
    kf[2][j] = frag_rm<WD>(three ? K1 : K0, l31, 2 * j + hi);
  }
  const float c = p.scale * LOG2E;
#pragma unroll
  for (int qt = 0; qt < qtiles; ++qt) {
    const int q = qblk + (qt * 4 + wave) * 32 + l31;
    f16x8 (&qf)[KS] = qall[qt];
    scale_frags<KS>(qf, c);  // scores come out of the MFMA in the log2 domain
    f32x16 s[3];
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
      ZERO16(s[kt]);
      if (kt < 2 || three) {
#pragma unroll
        for (int j = 0; j < KS; ++j) s[kt] = TB_MFMA_32x32x16(kf[kt][j], qf[j], s[kt]);
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 3; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * 32 + mfma32_row(r, hi);
        s[kt][r] = key < p.Skv ? s[kt][r] : -INFINITY;
        mx = fmaxf(mx, s[kt][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));   // (key 0 exists: finite)
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < 3; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = fast_exp2(s[kt][r] - mx);
        s[kt][r] = pv;
        l += pv;
      }
    f32x16 o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) ZERO16(o[d]);
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
      if (kt < 2 || three) {
        const f16* Vt = kt < 2 ? V0 : V1;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const f16x8 pf = pack8(s[kt], 8 * jj);
#pragma unroll
          for (int d = 0; d < DT; ++d) o[d] = TB_MFMA_32x32x16(frag_tr(Vt, d, l31, (kt & 1) * 32 + 16 * jj, hi), pf, o[d]);
        }
      }
    }
    l += __shfl_xor(l, 32, 64);
    if (q < p.Sq) {
      const float inv = 1.f / l;
      f16* Og = (f16*)p.O + ((int64_t)b * p.Sq + q) * p.ldo + h * p.hd;
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int col = d * 32 + 8 * r4 + 4 * hi;  // 4 consecutive head-dim columns
          if (col < p.hd) {
            f16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (f16)(o[d][4 * r4 + e] * inv);
            *(f16x4*)(Og + col) = v;
          }
        }
      if (p.LSE && hi == 0) p.LSE[((int64_t)b * p.H + h) * p.Sq + q] = mx * (1.f / LOG2E) + logf(l);
    }
  }
}

// ------------------------------------------------------------------------------------------------ forward, LDS-DMA staged
// Same arithmetic as attn_fwd_kernel (swapped S^T = K Q^T, FOLDM / ONES tricks, lazy re-base), different operand path -- the register-staged
// kernel issues ~380 instructions per 14 MFMAs (tile loads into VGPRs, address arithmetic, f16 permutes and 12 LDS stores per thread for
// the transposed V image, exec-mask bookkeeping for the half-idle staging threads) and is bound by instruction issue, not by the matrix
// pipe (28 % busy) nor by the softmax VALU work alone:
//   * K and V tiles go HBM -> LDS by global_load_lds_dwordx4 (no staging registers, no LDS stores), BOTH row-major, 3-deep ring with
//     counted vmcnt (loads two tiles ahead of the multiplication);
//   * V^T fragments for O^T = V^T P^T are read with ds_read_b64_tr_b16 (the hardware transposing read) straight from the row-major
//     tile: lane j of a 16-lane group points at key row (j >> 2), columns 4 (j & 3) .. +3 and receives 4 consecutive keys of column j;
//   * rows are padded by ONE 16-byte chunk [1, 0, .. 0] (written once per stage; the DMA's pad lanes are switched off), which is the
//     all-ones column hd that FOLDM (K) and ONES (V) need.
// Requirements (else the register-staged kernel runs): non-causal, Sq % 128 == 0, Skv % 64 == 0, hd % 8 == 0, hd < 32 DT.
template <int OFF>
__device__ __forceinline__ f16x4 lds_tr_read_off(uint32_t addr) {
  f16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ void attn_wait_vmcnt(int n) {  // n wave-uniform
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
  }
}
typedef const __attribute__((address_space(1))) void* attn_gptr_t;
typedef __attribute__((address_space(3))) void* attn_lptr_t;

// QG: 32-query groups per wave (block = 4 waves x QG x 32 queries).  With QG = 2 every K / V fragment read from LDS, every DMA
// instruction, barrier and loop-control instruction serves two score tiles: the kernels are bound by instruction ISSUE (PMC: the waves of a
// SIMD are "active" -- issuing -- ~100 % of the time at ~4.6 cycles per instruction; MFMA busy 30 %), so what counts is instructions per MFMA.
// FP8 (BASELINE.json configs[4], opt-in through tb_attn_desc.fp8_ws): O^T = V^T P^T runs on v_mfma_scale_f32_32x32x64_f8f6f4 -- ONE matrix
// instruction per (query group, 32 head-dim rows) and 64-key tile instead of four fp16 ones.  P is rounded to e4m3 in registers
// (v_cvt_pk_fp8_f32; p <= 2^8 under the lazy re-base, e4m3 max 448), V arrives as the per-(batch, head) scaled, TRANSPOSED e4m3 image
// V8T[bh][48 rows: hd data rows, the all-ones row hd, zeros][Skv] that attn_v8t_kernel writes with the keys of every 64-key tile in the
// order the matrix instruction's k index meets them in the P registers (A and B fragments only have to agree on the k order), so a
// V^T fragment is 32 contiguous bytes per lane: two ds_read_b128, no transposing reads.  The row sum still rides on the all-ones row
// (it is the sum of the ROUNDED p: the output is normalised by what was actually accumulated); QK^T, the softmax and LSE stay fp16 / fp32.
constexpr int V8_ROWS = 48, V8_PITCH = 80;           // LDS image of a V8T tile: 64 rows (48 loaded, 16 zero) of 64 data bytes + a 16-byte pad chunk
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
template <int DT, int KS, int PC, bool FOLDM, int NST, int QG, bool FP8 = false>
__global__ __launch_bounds__(256, 2) void attn_fwd_dma_kernel(const tb_attn_desc p, int remap) {
  constexpr int PCB = PC * 16;                      // row pitch (bytes): hd / 8 data chunks + the pad chunk
  constexpr int TILE_B = KVT * PCB;                 // one 64-row tile
  constexpr int V8_B = 64 * V8_PITCH;
  constexpr int STAGE_B = FP8 ? TILE_B + V8_B : 2 * TILE_B + 64;  // K tile, V tile, slack for the V fragment reads past the last row's pad chunk
  static_assert(NST >= 3, "loads run two tiles ahead of the multiplication");
  static_assert(!FP8 || DT == 2, "fp8 path: hd = 40");
  constexpr int NV8 = (V8_ROWS * 5 + 63) / 64;      // fp8: load instructions of the V8T tile (48 rows x 5 chunks, the pad chunk's lanes off)
  constexpr int NI = FP8 ? PC + NV8 : 2 * PC;       // load instructions per stage (64 lanes x 16 B each)
  constexpr int WI = (NI + 3) / 4;                  // ... per wave
  constexpr int QB = 128 * QG;                      // queries per block
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const AttnBlk blk = attn_block(remap & 1);
  const int b = blk.b, h = blk.h;
  const int qblk = blk.x * QB;
  const int hd = p.hd, Skv = p.Skv;
  const int64_t ldk = p.ldk, ldv = p.ldv;
  const f16* Qg = (const f16*)p.Q + (int64_t)b * p.Sq * p.ldq + h * hd;
  const char* Kg = (const char*)((const f16*)p.K + (int64_t)b * Skv * ldk + h * hd);
  const char* Vg = (const char*)((const f16*)p.V + (int64_t)b * Skv * ldv + h * hd);
  const char* V8g = FP8 ? (const char*)p.fp8_ws + ((int64_t)b * p.H + h) * V8_ROWS * Skv : nullptr;
  f16x8 qf[QG][KS];
#pragma unroll
  for (int g = 0; g < QG; ++g) {
    load_row_frags<KS>(qf[g], Qg, p.ldq, qblk + (wave * QG + g) * 32 + l31, p.Sq, hd, hi);
    scale_frags<KS>(qf[g], p.scale * LOG2E);
  }
  // ---- this lane's part of a stage's loads: instruction t = wave + 4 i covers flat chunks t' * 64 + lane of tensor t / PC
  uint32_t g_off[WI];   // byte offset of the source chunk inside the tile's 64 rows (row * ld * 2 + c * 16)
  bool g_on[WI];        // false: pad chunk (keeps its constant) or no such instruction for this wave
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    const int t = wave + 4 * i;
    const int tensor = t >= PC ? 1 : 0;
    const int f = (t - tensor * PC) * 64 + lane;
    if (FP8 && tensor) {  // V8T tile: row = head-dim row (pitch Skv bytes in HBM), 4 data chunks of 16 keys + the pad chunk
      const int row = f / 5, c = f - row * 5;
      g_on[i] = t < NI && row < V8_ROWS && c < 4;
      g_off[i] = (uint32_t)((int64_t)row * Skv + c * 16);
    } else {
      const int row = f / PC, c = f - row * PC;
      g_on[i] = t < NI && c < PC - 1;
      g_off[i] = (uint32_t)((int64_t)row * (tensor ? ldv : ldk) * 2 + c * 16);
    }
  }
  int n_issued = 0;  // loads this wave issues per stage (wave-uniform): for the counted waits
#pragma unroll
  for (int i = 0; i < WI; ++i) n_issued += (wave + 4 * i < NI) ? 1 : 0;
  auto stage_loads = [&](int tile, int slot) {
    unsigned char* dst = smem_raw + slot * STAGE_B;
    const char* kb = Kg + (int64_t)tile * KVT * ldk * 2;
    const char* vb = FP8 ? V8g + (int64_t)tile * KVT : Vg + (int64_t)tile * KVT * ldv * 2;
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int t = wave + 4 * i;
      if (t < NI) {
        const char* src = (t >= PC ? vb : kb) + g_off[i];
        unsigned char* d_ = FP8 && t >= PC ? dst + TILE_B + (t - PC) * 1024 : dst + t * 1024;
        if (g_on[i]) __builtin_amdgcn_global_load_lds((attn_gptr_t)src, (attn_lptr_t)d_, 16, 0, 0);
      }
    }
  };
  // pad chunks [1, 0, .., 0] of every row of every stage (K and V): 2 * 64 * NST chunks (the first barrier of the loop publishes them)
  for (int u = threadIdx.x; u < 2 * KVT * NST; u += 256) {
    const int st = u / (2 * KVT), r = u - st * 2 * KVT;  // r < 64: K row r, else V row r - 64
    f16x8 one = {(f16)1.f, 0, 0, 0, 0, 0, 0, 0};
    if (FP8 && r >= KVT) {  // V8T image: zero the pad chunk of the loaded rows and the 16 rows that are never loaded
      const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      unsigned char* row = smem_raw + st * STAGE_B + TILE_B + (r - KVT) * V8_PITCH;
      *(f16x8*)(row + 64) = z;
      if (r - KVT >= V8_ROWS)
        for (int c = 0; c < 4; ++c) *(f16x8*)(row + c * 16) = z;
    } else {
      *(f16x8*)(smem_raw + st * STAGE_B + (r >= KVT ? TILE_B : 0) + (r & (KVT - 1)) * PCB + (PC - 1) * 16) = one;
    }
  }
  const int ntiles = Skv / KVT;
#pragma unroll
  for (int st = 0; st < NST - 1; ++st)
    if (st < ntiles) stage_loads(st, st);

  f32x16 o[QG][DT];
#pragma unroll
  for (int g = 0; g < QG; ++g)
#pragma unroll
    for (int d = 0; d < DT; ++d) ZERO16(o[g][d]);
  float m[QG], l[QG];
  f32x16 negm[QG];
#pragma unroll
  for (int g = 0; g < QG; ++g) {
    m[g] = 0.f, l[g] = 0.f;
    ZERO16(negm[g]);  // FOLDM: stays the zero tuple (the compiler feeds the inline constant 0 as the first MFMA's accumulator input)
  }
  const int fj = hd >> 4;
  const bool fold_lane = FOLDM && hi == ((hd >> 3) & 1);
  // fragment addressing: K row-major (lane = key row l31, chunk 2 j + hi); V transposing reads (lane group g = lane >> 4: d sub-block
  // 16 (g & 1), key offset 4 (g >> 1), lane j: key row j >> 2, columns 4 (j & 3))
  const uint32_t lds0 = (uint32_t)(uintptr_t)(attn_lptr_t)smem_raw;
  const uint32_t k_lane = l31 * PCB + hi * 16;
  const int g4 = lane >> 4, j16 = lane & 15;
  const uint32_t v_lane = TILE_B + (4 * (g4 >> 1) + (j16 >> 2)) * PCB + ((g4 & 1) * 16 + 4 * (j16 & 3)) * 2;
  int slot = 0, lslot = NST - 1;
#ifdef TB_ATTN_PROF
  unsigned long long pf_[5] = {0, 0, 0, 0, 0}, pt_ = __builtin_amdgcn_s_memtime();
#define TB_PF(k) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); pf_[k] += n_ - pt_; pt_ = n_; }
#else
#define TB_PF(k)
#endif
  for (int t = 0; t < ntiles; ++t) {
    // tile t has landed once at most the stages issued after it are outstanding (this wave's part); the barrier extends that to every
    // wave and frees the slot of tile t - 1 (its fragment reads were consumed by MFMAs of the previous iteration)
    {
      int later = ntiles - 1 - t;
      later = later > NST - 2 ? NST - 2 : later;
      attn_wait_vmcnt(later * n_issued);
    }
    TB_PF(3)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    TB_PF(4)
    if (t + NST - 1 < ntiles && !(remap & 16)) stage_loads((remap & 8) ? 0 : t + NST - 1, lslot);
    const unsigned char* Ks = smem_raw + slot * STAGE_B;
    f32x16 s[QG][2];
    {
      f16x8 kf[2][KS];
#pragma unroll
      for (int j = 0; j < KS; ++j)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) kf[kt][j] = *(const f16x8*)(Ks + k_lane + kt * 32 * PCB + j * 32);
#pragma unroll
      for (int g = 0; g < QG; ++g)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          s[g][kt] = TB_MFMA_32x32x16(kf[kt][0], qf[g][0], negm[g]);
#pragma unroll
          for (int j = 1; j < KS; ++j) s[g][kt] = TB_MFMA_32x32x16(kf[kt][j], qf[g][j], s[g][kt]);
        }
    }
    // V^T fragments of the whole tile (shared by the query groups): issued now, they arrive under the softmax arithmetic below
    f16x4 vt[2][2][DT][2];
    i32x8 v8[DT];
    if (FP8) {  // row d 32 + l31, bytes 32 hi .. +31: the lane's 32 k values of the whole tile
      const unsigned char* Vs8 = smem_raw + slot * STAGE_B + TILE_B + l31 * V8_PITCH + hi * 32;
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        const i32x4 a0 = *(const i32x4*)(Vs8 + d * 32 * V8_PITCH), a1 = *(const i32x4*)(Vs8 + d * 32 * V8_PITCH + 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) v8[d][e] = a0[e], v8[d][4 + e] = a1[e];
      }
    } else {
      const uint32_t va = lds0 + slot * STAGE_B + v_lane;
      __builtin_amdgcn_sched_barrier(0);
#define TB_VTR(KT, JJ, D, HH) vt[KT][JJ][D][HH] = lds_tr_read_off<((KT) * 32 + 16 * (JJ) + 8 * (HH)) * PCB + (D) * 64>(va);
#define TB_VTR_D(KT, JJ, HH)                      \
  TB_VTR(KT, JJ, 0, HH)                           \
  if (DT > 1) { TB_VTR(KT, JJ, (DT > 1 ? 1 : 0), HH) } \
  if (DT > 2) { TB_VTR(KT, JJ, (DT > 2 ? 2 : 0), HH) }
      TB_VTR_D(0, 0, 0) TB_VTR_D(0, 0, 1) TB_VTR_D(0, 1, 0) TB_VTR_D(0, 1, 1)
      TB_VTR_D(1, 0, 0) TB_VTR_D(1, 0, 1) TB_VTR_D(1, 1, 0) TB_VTR_D(1, 1, 1)
#undef TB_VTR_D
#undef TB_VTR
      __builtin_amdgcn_sched_barrier(0);
    }
    const bool first = t == 0;
    TB_PF(0)
#pragma unroll
    for (int g = 0; g < QG; ++g) {
      float mx0 = s[g][0][0], mx1 = s[g][1][0];
#pragma unroll
      for (int r = 1; r < 15; r += 2) {
        mx0 = max3f(mx0, s[g][0][r], s[g][0][r + 1]);
        mx1 = max3f(mx1, s[g][1][r], s[g][1][r + 1]);
      }
      float mx = max3f(mx0, mx1, fmaxf(s[g][0][15], s[g][1][15]));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      if (first || __any(mx > TB_ATTN_REBASE)) {
        float d = first ? mx : fmaxf(mx, 0.f);
        if (FOLDM) d = (float)(f16)(m[g] + d) - m[g];
        if (!first) {
          const float alpha = fast_exp2(-d);
          l[g] *= alpha;
#pragma unroll
          for (int dd = 0; dd < DT; ++dd)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[g][dd][r] *= alpha;
        }
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[g][kt][r] -= d;
        m[g] += d;
        if (FOLDM) {
#pragma unroll
          for (int j = 0; j < KS; ++j)
            if (j == fj && fold_lane) qf[g][j][0] = (f16)(-m[g]);
        } else {
          FILL16(negm[g], -m[g]);
        }
      }
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[g][kt][r] = fast_exp2(s[g][kt][r]);
      if (FP8) {  // LSE (the backward's softmax) from the exact row sum; O is normalised by the sum of the ROUNDED p (the all-ones row)
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) a0 += s[g][0][r], a1 += s[g][1][r];
        l[g] += a0 + a1;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the V^T fragments (inline-asm reads: not counted by the compiler)
    __builtin_amdgcn_sched_barrier(0);
    TB_PF(1)
    if (FP8) {
#pragma unroll
      for (int g = 0; g < QG; ++g) {
        i32x8 p8;  // byte 16 kt + r of the lane = register r of score tile kt: the k order attn_v8t_kernel stored the keys in
#pragma unroll
        for (int w = 0; w < 8; ++w) {
          const int kt = w >> 2, r = (w & 3) * 4;
          int x = 0;
          x = __builtin_amdgcn_cvt_pk_fp8_f32(s[g][kt][r], s[g][kt][r + 1], x, false);
          x = __builtin_amdgcn_cvt_pk_fp8_f32(s[g][kt][r + 2], s[g][kt][r + 3], x, true);
          p8[w] = x;
        }
#pragma unroll
        for (int d = 0; d < DT; ++d)
          o[g][d] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v8[d], p8, o[g][d], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
      }
    } else
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        f16x8 a[DT];
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            a[d][e] = vt[kt][jj][d][0][e];
            a[d][4 + e] = vt[kt][jj][d][1][e];
          }
#pragma unroll
        for (int g = 0; g < QG; ++g) {
          const f16x8 pf = pack8(s[g][kt], 8 * jj);
#pragma unroll
          for (int d = 0; d < DT; ++d) o[g][d] = TB_MFMA_32x32x16(a[d], pf, o[g][d]);
        }
      }
    TB_PF(2)
    slot = slot == NST - 1 ? 0 : slot + 1;
    lslot = lslot == NST - 1 ? 0 : lslot + 1;
  }
#ifdef TB_ATTN_PROF
  if (p.Delta && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (threadIdx.x & 63) == 0)
    for (int k = 0; k < 5; ++k) p.Delta[wave * 5 + k] = (float)pf_[k] / (float)ntiles;
#endif
#undef TB_PF
#pragma unroll
  for (int g = 0; g < QG; ++g) {
    // the all-ones column hd of V made O^T row hd the row sum: register 4 k of the hi == 0 lane of tile hd / 32
    float ls = 0.f;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4)
        if (d * 32 + k4 * 8 == hd) ls = o[g][d][4 * k4];
    const float lsum = __shfl(ls, l31, 64);
    float inv = 1.f / lsum;
    if (FP8) inv *= ((const float*)((const char*)p.fp8_ws + (int64_t)p.B * p.H * V8_ROWS * Skv))[b * p.H + h];  // 1 / (scale of this head's V image)
    const int q = qblk + (wave * QG + g) * 32 + l31;
    f16* Og = (f16*)p.O + ((int64_t)b * p.Sq + q) * p.ldo + h * hd;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int col = d * 32 + 8 * r4 + 4 * hi;
        if (col < hd) {
          f16x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (f16)(o[g][d][4 * r4 + e] * inv);
          *(f16x4*)(Og + col) = v;
        }
      }
    const float lexact = FP8 ? l[g] + __shfl_xor(l[g], 32, 64) : lsum;  // (each half-wave holds half of a query's keys)
    if (p.LSE && hi == 0) p.LSE[((int64_t)b * p.H + h) * p.Sq + q] = m[g] * (1.f / LOG2E) + logf(lexact);
  }
}

// ------------------------------------------------------------------------------------------------ dQ with LDS-DMA staging
// Same construction for dQ (hd = 40, Sq % 128 == 0, Skv % 64 == 0, non-causal): K and V tiles by LDS-DMA, row-major, NST-slot ring; the
// fragments of S^T = K Q^T and dP^T = V dO^T are ds_read_b128 of those rows, the K^T fragments of dQ^T += K^T dS^T come from the same K rows
// through the transposing read.  Also computes delta = rowsum(dO * O) and publishes -lse log2(e) / -delta for attn_bwd_dkv_dma_kernel.
// PAD = false (round 4, hd = 40): rows of hd / 8 chunks WITHOUT the pad chunk, i.e. an 80-byte pitch.  Nothing in this kernel needs the pad to hold
// a value -- the statistics enter as accumulator inputs, and the k-step that runs past a row's end (columns hd .. 16 KS - 1) meets the zeros of the
// lane-owned Q / dO fragments -- it only has to be FINITE: the next row's first chunk, or for the last row of the V tile the zeroed 64-byte slack
// behind the stage.  80-byte rows put the 16 rows of a ds_read_b128 lane group on 16 distinct 16-byte bank slots (the 96-byte rows were a
// 2-way conflict on every such read: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.50, profiles/r03_pmc_sq.txt) and the DMA moves 1/6 fewer bytes.
#ifndef TB_DQ_ASMDMA
#define TB_DQ_ASMDMA 0   // attn_bwd_dq_dma_kernel: 1 = the next tile's global -> LDS pieces go out as asm statements BETWEEN the tile's MFMAs (round 5 experiment:
                         // 913 -> 911 us for the 64x64-map backward pair, 110.6 = 110.6 us at 32x32 -- neutral, the kernel is not piece-issue bound); 0 = the
                         // builtin pieces in front of the tile's compute
#endif
// one global -> LDS piece (64 x 16 bytes, destination = wave-uniform LDS byte address + 16 lane) as an asm statement: may sit between MFMAs, where
// the builtin gets a vmcnt(0) and a VGPR round trip of M0 in front (csrc/gemm_epi.h glds16_asm)
__device__ __forceinline__ void attn_glds16_asm(const void* src, uint32_t lds_byte_addr) {
  const uint32_t m = __builtin_amdgcn_readfirstlane(lds_byte_addr);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m), "v"(src) : "memory", "m0");
}
template <int DT, int KS, int PC, int NST, bool PAD = true>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_dma_kernel(const tb_attn_desc p, int remap, int publish) {
  constexpr int PCB = PC * 16, TILE_B = KVT * PCB;
  constexpr int STAGE_B = 2 * TILE_B + 64;
  constexpr int NI = 2 * PC, WI = (NI + 3) / 4;
  static_assert(NST >= 3 && (DT == 2 || DT == 3), "hd = 40 / 80 instantiations; loads run two tiles ahead");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const AttnBlk blk = attn_block(remap);
  const int b = blk.b, h = blk.h, hd = p.hd, Skv = p.Skv;
  const int q = blk.x * 128 + wave * 32 + l31;
  const int64_t ldk = p.ldk, ldv = p.ldv;
  const f16* Qg = (const f16*)p.Q + (int64_t)b * p.Sq * p.ldq + h * hd;
  const f16* dOg = (const f16*)p.dO + (int64_t)b * p.Sq * p.lddo + h * hd;
  const char* Kg = (const char*)((const f16*)p.K + (int64_t)b * Skv * ldk + h * hd);
  const char* Vg = (const char*)((const f16*)p.V + (int64_t)b * Skv * ldv + h * hd);
  f16x8 qf[KS], dof[KS];
  load_row_frags<KS>(qf, Qg, p.ldq, q, p.Sq, hd, hi);
  load_row_frags<KS>(dof, dOg, p.lddo, q, p.Sq, hd, hi);
  scale_frags<KS>(qf, p.scale * LOG2E);
  const int64_t sidx = ((int64_t)b * p.H + h) * p.Sq + q;
  const float lse2 = p.LSE[sidx] * LOG2E;
  float delta;
  {
    const f16* Og = (const f16*)p.O + (int64_t)b * p.Sq * p.ldo + h * hd;
    float a = 0.f;
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      const int col = 16 * j + 8 * hi;
      if (col < hd) {
        const f16x8 ov = *(const f16x8*)(Og + (int64_t)q * p.ldo + col);
#pragma unroll
        for (int e = 0; e < 8; ++e) a += (float)ov[e] * (float)dof[j][e];
      }
    }
    delta = a + __shfl_xor(a, 32, 64);
    if (hi == 0) {
      p.Delta[sidx] = delta;
      if (publish) {
        p.ws[sidx] = -lse2;
        p.ws[(int64_t)p.B * p.H * p.Sq + sidx] = -delta;
      }
    }
  }
  f32x16 neg_lse, neg_delta;  // accumulator inputs of the two products
  FILL16(neg_lse, -lse2);
  FILL16(neg_delta, -delta);
  uint32_t g_off[WI];
  bool g_on[WI];
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    const int t = wave + 4 * i;
    const int tensor = t >= PC ? 1 : 0;
    const int f = (t - tensor * PC) * 64 + lane;
    const int row = f / PC, c = f - row * PC;
    g_on[i] = t < NI && (!PAD || c < PC - 1);
    g_off[i] = (uint32_t)((int64_t)row * (tensor ? ldv : ldk) * 2 + c * 16);
  }
  int n_issued = 0;
#pragma unroll
  for (int i = 0; i < WI; ++i) n_issued += (wave + 4 * i < NI) ? 1 : 0;
  auto stage_loads = [&](int tile, int slot) {
    unsigned char* dst = smem_raw + slot * STAGE_B;
    const char* kb = Kg + (int64_t)tile * KVT * ldk * 2;
    const char* vb = Vg + (int64_t)tile * KVT * ldv * 2;
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int t = wave + 4 * i;
      if (t < NI) {
        const char* src = (t >= PC ? vb : kb) + g_off[i];
        if (g_on[i]) __builtin_amdgcn_global_load_lds((attn_gptr_t)src, (attn_lptr_t)(dst + t * 1024), 16, 0, 0);
      }
    }
  };
  if (PAD) {
    for (int u = threadIdx.x; u < 2 * KVT * NST; u += 256) {  // pad chunks: zeros (finite)
      const int st = u / (2 * KVT), r = u - st * 2 * KVT;
      const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      *(f16x8*)(smem_raw + st * STAGE_B + (r >= KVT ? TILE_B : 0) + (r & (KVT - 1)) * PCB + (PC - 1) * 16) = z;
    }
  } else if (threadIdx.x < NST * 4) {  // the 64-byte slack behind every stage: what the last V row's over-long k-step reads
    const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    *(f16x8*)(smem_raw + (threadIdx.x >> 2) * STAGE_B + 2 * TILE_B + (threadIdx.x & 3) * 16) = z;
  }
  const int ntiles = Skv / KVT;
#pragma unroll
  for (int st = 0; st < NST - 1; ++st)
    if (st < ntiles) stage_loads(st, st);
  f32x16 dq[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d) ZERO16(dq[d]);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(attn_lptr_t)smem_raw;
  const uint32_t rm_lane = l31 * PCB + hi * 16;
  const int g4 = lane >> 4, j16 = lane & 15;
  const uint32_t tr_lane = (4 * (g4 >> 1) + (j16 >> 2)) * PCB + ((g4 & 1) * 16 + 4 * (j16 & 3)) * 2;
  int slot = 0, lslot = NST - 1;
  for (int t = 0; t < ntiles; ++t) {
    {
      int later = ntiles - 1 - t;
      later = later > NST - 2 ? NST - 2 : later;
      attn_wait_vmcnt(later * n_issued);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const bool ld_on = t + NST - 1 < ntiles;
    if (!TB_DQ_ASMDMA && ld_on) stage_loads(t + NST - 1, lslot);
    // TB_DQ_ASMDMA: piece i of the next stage behind MFMA group i of this tile (three groups: scores of the first half, its dQ products, scores of the second half)
    auto piece = [&](int i) {
      const int tt = wave + 4 * i;
      if (TB_DQ_ASMDMA && i < WI && ld_on && tt < NI && g_on[i]) {
        const char* src = (tt >= PC ? Vg + (int64_t)(t + NST - 1) * KVT * ldv * 2 : Kg + (int64_t)(t + NST - 1) * KVT * ldk * 2) + g_off[i];
        __builtin_amdgcn_sched_barrier(0);
        attn_glds16_asm(src, lds0 + lslot * STAGE_B + tt * 1024);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    const unsigned char* Ks = smem_raw + slot * STAGE_B;
    const uint32_t ka = lds0 + slot * STAGE_B + tr_lane;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      f32x16 s = neg_lse, dp = neg_delta;
#pragma unroll
      for (int j = 0; j < KS; ++j) {
        const f16x8 kfr = *(const f16x8*)(Ks + rm_lane + kt * 32 * PCB + j * 32);
        const f16x8 vfr = *(const f16x8*)(Ks + TILE_B + rm_lane + kt * 32 * PCB + j * 32);
        s = TB_MFMA_32x32x16(kfr, qf[j], s);
        dp = TB_MFMA_32x32x16(vfr, dof[j], dp);
      }
      piece(kt == 0 ? 0 : 2);
      f16x4 ktf[2][DT][2];  // K^T fragments of this 32-key half: issued behind the score products, they arrive under the exponentials
      __builtin_amdgcn_sched_barrier(0);
#define TB_TR(KT, JJ, D, HH) ktf[JJ][D][HH] = lds_tr_read_off<((KT) * 32 + 16 * (JJ) + 8 * (HH)) * PCB + (D) * 64>(ka);
#define TB_TR_D(KT, D) TB_TR(KT, 0, D, 0) TB_TR(KT, 0, D, 1) TB_TR(KT, 1, D, 0) TB_TR(KT, 1, D, 1)
#define TB_TR_ALL(KT)                              \
  TB_TR_D(KT, 0)                                   \
  TB_TR_D(KT, 1)                                   \
  if (DT > 2) { TB_TR_D(KT, (DT > 2 ? 2 : 0)) }
      if (kt == 0) { TB_TR_ALL(0) } else { TB_TR_ALL(1) }
#undef TB_TR_D
#undef TB_TR_ALL
#undef TB_TR
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = fast_exp2(s[r]) * dp[r];  // dS^T / scale
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const f16x8 dsf = pack8(s, 8 * jj);
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          f16x8 a;
#pragma unroll
          for (int e = 0; e < 4; ++e) a[e] = ktf[jj][d][0][e], a[4 + e] = ktf[jj][d][1][e];
          dq[d] = TB_MFMA_32x32x16(a, dsf, dq[d]);
        }
      }
      if (kt == 0) piece(1);
    }
    static_assert(!TB_DQ_ASMDMA || WI <= 3 || PAD, "three piece slots per tile");
    if (WI > 3) {
#pragma unroll
      for (int i = 3; i < WI; ++i) piece(i);
    }
    slot = slot == NST - 1 ? 0 : slot + 1;
    lslot = lslot == NST - 1 ? 0 : lslot + 1;
  }
  f16* dQg = (f16*)p.dQ + ((int64_t)b * p.Sq + q) * p.lddq + h * hd;
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int col = d * 32 + 8 * r4 + 4 * hi;
      if (col < hd) {
        f16x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (f16)(dq[d][4 * r4 + e] * p.scale);
        *(f16x4*)(dQg + col) = v;
      }
    }
}

// ------------------------------------------------------------------------------------------------ dK, dV with LDS-DMA staging
// The dK/dV kernel for the SD1.x self-attention shape (hd = 40, Sq % 64 == 0, Skv % 128 == 0, non-causal), built like attn_fwd_dma_kernel:
// the streamed Q and dO tiles go HBM -> LDS by global_load_lds_dwordx4, ROW-MAJOR only, through an NST-slot ring with counted vmcnt (one
// barrier per tile instead of two); the fragments of S^T = Q K^T and dP^T = dO V^T are ds_read_b128 from those rows, the Q^T / dO^T fragments
// of dK += dS^T Q and dV += P^T dO come from the SAME rows through the transposing read ds_read_b64_tr_b16 -- no second (transposed) LDS
// image, no register staging, no packing VALU (the register-staged kernel spends 32 v_perm + 24 LDS stores + the tile address math per
// tile and wave on it; beside the matrix pipe every plain VALU instruction is paid in full).  The per-query -lse log2(e) and -delta that
// enter the two products as accumulator inputs are DMA'd raw (dword loads) from the two [B, H, Sq] arrays the dQ kernel publishes in
// tb_attn_desc.ws: no arithmetic on them here.
template <int DT, int KS, int PC, int NST>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_dma_kernel(const tb_attn_desc p, int remap) {
  constexpr int PCB = PC * 16, TILE_B = KVT * PCB;
  constexpr int STAGE_B = 2 * TILE_B + 2 * KVT * 4 + 64;  // Q tile, dO tile, -lse2[64], -delta[64], slack for the tr reads past the last pad chunk
  constexpr int NI = 2 * PC + 2, WI = (NI + 3) / 4;
  static_assert(NST >= 3, "loads run two tiles ahead");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const AttnBlk blk = attn_block(remap);
  const int b = blk.b, h = blk.h, hd = p.hd;
  const int kblk = blk.x * 128;
  const int key = kblk + wave * 32 + l31;
  const int64_t ldq = p.ldq, lddo = p.lddo;
  const char* Qg = (const char*)((const f16*)p.Q + (int64_t)b * p.Sq * ldq + h * hd);
  const char* dOg = (const char*)((const f16*)p.dO + (int64_t)b * p.Sq * lddo + h * hd);
  const f16* Kg = (const f16*)p.K + (int64_t)b * p.Skv * p.ldk + h * hd;
  const f16* Vg = (const f16*)p.V + (int64_t)b * p.Skv * p.ldv + h * hd;
  const int64_t BHS = (int64_t)p.B * p.H * p.Sq;
  const char* NLg = (const char*)(p.ws + ((int64_t)b * p.H + h) * p.Sq);        // -lse * log2(e), written by the dQ kernel
  const char* NDg = (const char*)(p.ws + BHS + ((int64_t)b * p.H + h) * p.Sq);  // -delta
  f16x8 kf[KS], vf[KS];
  load_row_frags<KS>(kf, Kg, p.ldk, key, p.Skv, hd, hi);
  load_row_frags<KS>(vf, Vg, p.ldv, key, p.Skv, hd, hi);
  scale_frags<KS>(kf, p.scale * LOG2E);
  // ---- this lane's part of a stage's loads: instruction t = wave + 4 i; t < PC: Q rows, t < 2 PC: dO rows, then the two stat rows
  uint32_t g_off[WI];
  bool g_on[WI];
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    const int t = wave + 4 * i;
    const int tensor = t >= PC ? 1 : 0;
    const int f = (t - tensor * PC) * 64 + lane;
    const int row = f / PC, c = f - row * PC;
    g_on[i] = t < 2 * PC && c < PC - 1;
    g_off[i] = (uint32_t)((int64_t)row * (tensor ? lddo : ldq) * 2 + c * 16);
  }
  int n_issued = 0;
#pragma unroll
  for (int i = 0; i < WI; ++i) n_issued += (wave + 4 * i < NI) ? 1 : 0;
  auto stage_loads = [&](int tile, int slot) {
    unsigned char* dst = smem_raw + slot * STAGE_B;
    const char* qb = Qg + (int64_t)tile * KVT * ldq * 2;
    const char* ob = dOg + (int64_t)tile * KVT * lddo * 2;
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int t = wave + 4 * i;
      if (t < 2 * PC) {
        const char* src = (t >= PC ? ob : qb) + g_off[i];
        if (g_on[i]) __builtin_amdgcn_global_load_lds((attn_gptr_t)src, (attn_lptr_t)(dst + t * 1024), 16, 0, 0);
      } else if (t < NI) {  // 64 floats = one dword per lane
        const char* src = (t == 2 * PC ? NLg : NDg) + ((int64_t)tile * KVT + lane) * 4;
        __builtin_amdgcn_global_load_lds((attn_gptr_t)src, (attn_lptr_t)(dst + 2 * TILE_B + (t - 2 * PC) * 256), 4, 0, 0);
      }
    }
  };
  // pad chunks of every row of every stage: zeros (they meet the zero padding of the lane-owned K / V fragments, but must be finite)
  for (int u = threadIdx.x; u < 2 * KVT * NST; u += 256) {
    const int st = u / (2 * KVT), r = u - st * 2 * KVT;
    const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    *(f16x8*)(smem_raw + st * STAGE_B + (r >= KVT ? TILE_B : 0) + (r & (KVT - 1)) * PCB + (PC - 1) * 16) = z;
  }
  const int ntiles = p.Sq / KVT;
#pragma unroll
  for (int st = 0; st < NST - 1; ++st)
    if (st < ntiles) stage_loads(st, st);
  f32x16 dk[DT], dv[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d) {
    ZERO16(dk[d]);
    ZERO16(dv[d]);
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)(attn_lptr_t)smem_raw;
  const uint32_t rm_lane = l31 * PCB + hi * 16;   // row-major fragment: row l31 (+ 32 qt), chunk 2 j + hi
  const int g4 = lane >> 4, j16 = lane & 15;
  const uint32_t tr_lane = (4 * (g4 >> 1) + (j16 >> 2)) * PCB + ((g4 & 1) * 16 + 4 * (j16 & 3)) * 2;  // as the forward kernel's V^T reads
  int slot = 0, lslot = NST - 1;
  for (int t = 0; t < ntiles; ++t) {
    {
      int later = ntiles - 1 - t;
      later = later > NST - 2 ? NST - 2 : later;
      attn_wait_vmcnt(later * n_issued);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (t + NST - 1 < ntiles) stage_loads(t + NST - 1, lslot);
    const unsigned char* Qs = smem_raw + slot * STAGE_B;
    const float* lse_s = (const float*)(Qs + 2 * TILE_B);
    const float* del_s = lse_s + KVT;
    const uint32_t qa = lds0 + slot * STAGE_B + tr_lane;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      f32x16 s, dp;  // start from -lse2 / -delta of the tile's queries: rows of register quad g are 8 g + 4 hi + {0..3}
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const f32x4 lq = *(const f32x4*)(lse_s + qt * 32 + 8 * q4 + 4 * hi);
        const f32x4 dq4 = *(const f32x4*)(del_s + qt * 32 + 8 * q4 + 4 * hi);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s[4 * q4 + e] = lq[e];
          dp[4 * q4 + e] = dq4[e];
        }
      }
#pragma unroll
      for (int j = 0; j < KS; ++j) {
        const f16x8 qfr = *(const f16x8*)(Qs + rm_lane + qt * 32 * PCB + j * 32);
        const f16x8 dofr = *(const f16x8*)(Qs + TILE_B + rm_lane + qt * 32 * PCB + j * 32);
        s = TB_MFMA_32x32x16(qfr, kf[j], s);
        dp = TB_MFMA_32x32x16(dofr, vf[j], dp);
      }
      // transposed fragments of this 32-query half: issued behind the score products, they arrive under the exponentials
      f16x4 qtf[2][DT][2], dotf[2][DT][2];
      __builtin_amdgcn_sched_barrier(0);
#define TB_TR(ARR, BASE, QT, JJ, D, HH) ARR[JJ][D][HH] = lds_tr_read_off<(BASE) + ((QT) * 32 + 16 * (JJ) + 8 * (HH)) * PCB + (D) * 64>(qa);
#define TB_TR_D(ARR, BASE, QT, D) TB_TR(ARR, BASE, QT, 0, D, 0) TB_TR(ARR, BASE, QT, 0, D, 1) TB_TR(ARR, BASE, QT, 1, D, 0) TB_TR(ARR, BASE, QT, 1, D, 1)
#define TB_TR_ALL(QT)                                                                   \
  TB_TR_D(qtf, 0, QT, 0) TB_TR_D(qtf, 0, QT, 1)                                         \
  if (DT > 2) { TB_TR_D(qtf, 0, QT, (DT > 2 ? 2 : 0)) }                                 \
  TB_TR_D(dotf, TILE_B, QT, 0) TB_TR_D(dotf, TILE_B, QT, 1)                             \
  if (DT > 2) { TB_TR_D(dotf, TILE_B, QT, (DT > 2 ? 2 : 0)) }
      static_assert(DT == 2 || DT == 3, "hd = 40 / 80 instantiations");
      if (qt == 0) { TB_TR_ALL(0) } else { TB_TR_ALL(1) }
#undef TB_TR_D
#undef TB_TR_ALL
#undef TB_TR
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = fast_exp2(s[r]);
        s[r] = pv;            // P
        dp[r] = pv * dp[r];   // dS / scale
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the transposed fragments (inline-asm reads: not counted by the compiler)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const f16x8 pf = pack8(s, 8 * jj);
        const f16x8 dsf = pack8(dp, 8 * jj);
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          f16x8 a, c;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            a[e] = dotf[jj][d][0][e], a[4 + e] = dotf[jj][d][1][e];
            c[e] = qtf[jj][d][0][e], c[4 + e] = qtf[jj][d][1][e];
          }
          dv[d] = TB_MFMA_32x32x16(a, pf, dv[d]);
          dk[d] = TB_MFMA_32x32x16(c, dsf, dk[d]);
        }
      }
    }
    slot = slot == NST - 1 ? 0 : slot + 1;
    lslot = lslot == NST - 1 ? 0 : lslot + 1;
  }
  if (key < p.Skv) {
    f16* dKg = (f16*)p.dK + ((int64_t)b * p.Skv + key) * p.lddk + h * hd;
    f16* dVg = (f16*)p.dV + ((int64_t)b * p.Skv + key) * p.lddv + h * hd;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int col = d * 32 + 8 * r4 + 4 * hi;
        if (col < hd) {
          f16x4 a, bb;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            a[e] = (f16)(dk[d][4 * r4 + e] * p.scale);
            bb[e] = (f16)dv[d][4 * r4 + e];
          }
          *(f16x4*)(dKg + col) = a;
          *(f16x4*)(dVg + col) = bb;
        }
      }
  }
}

// ------------------------------------------------------------------------------------------------ fp8 image of V for attn_fwd_dma_kernel<.., FP8>
// workspace (tb_attn_desc.fp8_ws, tb_attention_fp8_ws_bytes): [B*H][48][Skv] e4m3 | [B*H] fp32 1/scale | [B*H][Skv/256] fp32 partial max
// pass 1: max |v| per (batch, head) and 256-key block; pass 2: scale = 384 / max (e4m3 max 448), V8T[bh][d][tile*64 + pos(key)] with
// pos = 32 hi + 16 kt + 4 q + e for key = 32 kt + 8 q + 4 hi + e -- the order in which a lane of the 32x32 score tile S^T holds its keys.
__global__ __launch_bounds__(256) void attn_v8t_kernel(const tb_attn_desc p, int pass) {
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H, nb = gridDim.x;
  const int key = blockIdx.x * 256 + threadIdx.x;
  const int64_t BH = (int64_t)p.B * p.H;
  unsigned char* v8 = (unsigned char*)p.fp8_ws + (int64_t)bh * V8_ROWS * p.Skv;
  float* inv_scale = (float*)((char*)p.fp8_ws + BH * V8_ROWS * p.Skv);
  float* partial = inv_scale + BH;
  const f16* vrow = (const f16*)p.V + ((int64_t)b * p.Skv + key) * p.ldv + h * p.hd;
  __shared__ float red[4];
  if (pass == 0) {
    float mx = 0.f;
    for (int c = 0; c < p.hd; c += 8) {
      const f16x8 v = *(const f16x8*)(vrow + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) mx = fmaxf(mx, fabsf((float)v[e]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) partial[(int64_t)bh * nb + blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    return;
  }
  float mx = 0.f;
  for (int i = 0; i < nb; ++i) mx = fmaxf(mx, partial[(int64_t)bh * nb + i]);
  const float scale = mx > 0.f ? 384.f / mx : 1.f;
  if (blockIdx.x == 0 && threadIdx.x == 0) inv_scale[bh] = 1.f / scale;
  const int kk = key & 63, kt = kk >> 5, rem = kk & 31;
  const int pos = (key & ~63) + 32 * ((rem >> 2) & 1) + 16 * kt + 4 * (rem >> 3) + (rem & 3);
  for (int c = 0; c < p.hd; c += 8) {
    const f16x8 v = *(const f16x8*)(vrow + c);
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      const int x = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[e] * scale, (float)v[e + 1] * scale, 0, false);
      v8[(int64_t)(c + e) * p.Skv + pos] = (unsigned char)(x & 0xff);
      v8[(int64_t)(c + e + 1) * p.Skv + pos] = (unsigned char)((x >> 8) & 0xff);
    }
  }
  v8[(int64_t)p.hd * p.Skv + pos] = 0x38;  // 1.0 in e4m3: the all-ones row that makes O^T row hd the row sum
  for (int d = p.hd + 1; d < V8_ROWS; ++d) v8[(int64_t)d * p.Skv + pos] = 0;
}

// ------------------------------------------------------------------------------------------------ delta = rowsum(dO * O)
__global__ __launch_bounds__(256) void attn_delta_kernel(const tb_attn_desc p) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over B*Sq*H
  const int64_t total = (int64_t)p.B * p.Sq * p.H;
  if (idx >= total) return;
  const int h = (int)(idx % p.H);
  const int64_t row = idx / p.H;  // b*Sq + q
  const f16* o = (const f16*)p.O + row * p.ldo + h * p.hd;
  const f16* d = (const f16*)p.dO + row * p.lddo + h * p.hd;
  float a = 0.f;
  for (int c = 0; c < p.hd; c += 8) {
    const f16x8 ov = *(const f16x8*)(o + c), dv = *(const f16x8*)(d + c);
#pragma unroll
    for (int e = 0; e < 8; ++e) a += (float)ov[e] * (float)dv[e];
  }
  const int b = (int)(row / p.Sq), q = (int)(row % p.Sq);
  p.Delta[((int64_t)b * p.H + h) * p.Sq + q] = a;
}

// ------------------------------------------------------------------------------------------------ dQ
#ifndef TB_DQ_OCC
#define TB_DQ_OCC 2
#endif
template <int DT, int KS>
__global__ __launch_bounds__(256, (DT <= 2 ? TB_DQ_OCC : 1)) void attn_bwd_dq_kernel(const tb_attn_desc p, int remap, int publish) {
  constexpr int WD = DT * 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  f16* Ks = reinterpret_cast<f16*>(smem_raw);
  f16* Vs = Ks + RM<WD>::SIZE;
  f16* Kt = Vs + RM<WD>::SIZE;
  const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform (SGPR): the mask tests below become scalar branches
  const AttnBlk blk = attn_block(remap);
  const int b = blk.b, h = blk.h;
  const int qblk = blk.x * 128;
  const int q = qblk + wave * 32 + l31;
  const f16* Qg = (const f16*)p.Q + (int64_t)b * p.Sq * p.ldq + h * p.hd;
  const f16* Kg = (const f16*)p.K + (int64_t)b * p.Skv * p.ldk + h * p.hd;
  const f16* Vg = (const f16*)p.V + (int64_t)b * p.Skv * p.ldv + h * p.hd;
  const f16* dOg = (const f16*)p.dO + (int64_t)b * p.Sq * p.lddo + h * p.hd;
  f16x8 qf[KS], dof[KS];
  load_row_frags<KS>(qf, Qg, p.ldq, q, p.Sq, p.hd, hi);
  load_row_frags<KS>(dof, dOg, p.lddo, q, p.Sq, p.hd, hi);
  scale_frags<KS>(qf, p.scale * LOG2E);  // as in the forward: scores arrive in the log2 domain
  const bool qok = q < p.Sq;
  const int64_t sidx = ((int64_t)b * p.H + h) * p.Sq + (qok ? q : 0);
  const float lse2 = p.LSE[sidx] * LOG2E;
#ifdef TB_ATTN_FUSED_DELTA
  float delta;  // rowsum(dO * O) computed here and published for the dK/dV kernel that runs next on the stream
  {
    const f16* Og = (const f16*)p.O + (int64_t)b * p.Sq * p.ldo + h * p.hd;
    float a = 0.f;
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      const int col = 16 * j + 8 * hi;
      if (qok && col < p.hd) {
        const f16x8 ov = *(const f16x8*)(Og + (int64_t)q * p.ldo + col);
#pragma unroll
        for (int e = 0; e < 8; ++e) a += (float)ov[e] * (float)dof[j][e];
      }
    }
    delta = a + __shfl_xor(a, 32, 64);
    if (qok && hi == 0) p.Delta[sidx] = delta;
  }
  if (publish && qok && hi == 0) {  // accumulator inputs of attn_bwd_dkv_dma_kernel, which DMAs them raw: -lse log2(e) | -delta in p.ws
    p.ws[sidx] = -lse2;
    p.ws[(int64_t)p.B * p.H * p.Sq + sidx] = -delta;
  }
#else
  const float delta = p.Delta[sidx];
#endif
  // FOLD: -lse and -delta enter the two products as accumulator inputs (S' = c q.k - lse is the exp2 argument, dP' = dO.v - delta);
  // costs 32 registers, so the wide-head instantiations (already AGPR-bound at one wave per SIMD) subtract explicitly instead
  constexpr bool FOLD = DT <= 2;
  f32x16 neg_lse, neg_delta;
  FILL16(neg_lse, FOLD ? -lse2 : 0.f);
  FILL16(neg_delta, FOLD ? -delta : 0.f);
  const float sub_l = FOLD ? 0.f : lse2, sub_d = FOLD ? 0.f : delta;
  f32x16 dq[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d) ZERO16(dq[d]);
  int kv_end = p.Skv;
  if (p.causal) kv_end = min(p.Skv, qblk + 128);
  constexpr bool PF = DT <= 2;
  TileRegs<WD> kreg, vreg;
  tile_init<WD>(kreg, p.ldk, p.hd);
  tile_init<WD>(vreg, p.ldv, p.hd);
  if (PF) {
    tile_load<WD, (DT <= 2)>(kreg, Kg, p.ldk, 0, p.Skv, p.hd);
    tile_load<WD, (DT <= 2)>(vreg, Vg, p.ldv, 0, p.Skv, p.hd);
  }
  for (int kv0 = 0; kv0 < kv_end; kv0 += KVT) {
    __syncthreads();
    if (!PF) {
      tile_load<WD, (DT <= 2)>(kreg, Kg, p.ldk, kv0, p.Skv, p.hd);
      tile_load<WD, (DT <= 2)>(vreg, Vg, p.ldv, kv0, p.Skv, p.hd);
    }
    tile_store<WD, true, true>(kreg, Ks, Kt);
    tile_store<WD, true, false>(vreg, Vs, nullptr);
    __syncthreads();
    if (PF && kv0 + KVT < kv_end) {
      tile_load<WD, (DT <= 2)>(kreg, Kg, p.ldk, kv0 + KVT, p.Skv, p.hd);
      tile_load<WD, (DT <= 2)>(vreg, Vg, p.ldv, kv0 + KVT, p.Skv, p.hd);
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      f32x16 s, dp;
      if (FOLD) {
        s = neg_lse;
        dp = neg_delta;
      } else {
        ZERO16(s);
        ZERO16(dp);
      }
#pragma unroll
      for (int j = 0; j < KS; ++j) {
        s = TB_MFMA_32x32x16(frag_rm<WD>(Ks, kt * 32 + l31, 2 * j + hi), qf[j], s);
        dp = TB_MFMA_32x32x16(frag_rm<WD>(Vs, kt * 32 + l31, 2 * j + hi), dof[j], dp);
      }
      const bool need_mask = (kv0 + kt * 32 + 32 > p.Skv) || (p.causal && kv0 + kt * 32 + 31 > qblk + wave * 32) || (qblk + wave * 32 + 32 > p.Sq);
      if (need_mask) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kv0 + kt * 32 + mfma32_row(r, hi);
          const bool ok = qok && key < p.Skv && (!p.causal || key <= q);
          const float pv = ok ? fast_exp2(FOLD ? s[r] : s[r] - sub_l) : 0.f;
          s[r] = pv * (FOLD ? dp[r] : dp[r] - sub_d);  // dS^T / scale
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = fast_exp2(FOLD ? s[r] : s[r] - sub_l) * (FOLD ? dp[r] : dp[r] - sub_d);
      }
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const f16x8 dsf = pack8(s, 8 * jj);
#pragma unroll
        for (int d = 0; d < DT; ++d)
          dq[d] = TB_MFMA_32x32x16(frag_tr(Kt, d, l31, kt * 32 + 16 * jj, hi), dsf, dq[d]);
      }
    }
  }
  if (qok) {
    f16* dQg = (f16*)p.dQ + ((int64_t)b * p.Sq + q) * p.lddq + h * p.hd;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int col = d * 32 + 8 * r4 + 4 * hi;
        if (col < p.hd) {
          f16x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (f16)(dq[d][4 * r4 + e] * p.scale);
          *(f16x4*)(dQg + col) = v;
        }
      }
  }
}

// ------------------------------------------------------------------------------------------------ dQ, short key sequences
// The dQ half of the cross-attention backward in the form of attn_xs_fwd_kernel: the K (row-major and transposed) and V images of all <= 96
// keys staged once, K / V held as register fragments, QT 32-query tiles per wave with all of their Q / dO / O rows requested up front, no
// barrier in the loop.  Same arithmetic as attn_bwd_dq_kernel (scores in the log2 domain with -lse and -delta as the products' accumulator
// inputs, dS^T = P (dP - delta), the softmax scale applied at write-out); also publishes delta = rowsum(dO * O) for the dK / dV launch.
template <int DT, int KS, int QT>
__global__ __launch_bounds__(256, 2) void attn_xs_bwd_dq_kernel(const tb_attn_desc p, int remap) {
  constexpr int WD = DT * 32;
  constexpr int TILE = 2 * RM<WD>::SIZE + TR<WD>::SIZE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  f16* const K0 = reinterpret_cast<f16*>(smem_raw);
  f16* const V0 = K0 + RM<WD>::SIZE;
  f16* const T0 = V0 + RM<WD>::SIZE;
  f16* const K1 = K0 + TILE;
  f16* const V1 = K1 + RM<WD>::SIZE;
  f16* const T1 = V1 + RM<WD>::SIZE;
  const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  AttnBlk blk;   // (query band, batch) pairs per XCD, heads back to back: see attn_xs_fwd_kernel
  {
    const int gx = gridDim.x, H = gridDim.y, B = gridDim.z;
    if (remap && ((gx * B) & 7) == 0) {
      const int lin = blockIdx.x + gx * (blockIdx.y + H * blockIdx.z);
      const int xcd = lin & 7, k = lin >> 3;
      const int xb = (k / H) * 8 + xcd;
      blk.h = k - (k / H) * H;
      blk.x = xb % gx;
      blk.b = xb / gx;
    } else {
      blk.x = blockIdx.x, blk.h = blockIdx.y, blk.b = blockIdx.z;
    }
  }
  const int b = blk.b, h = blk.h;
  const int qblk = blk.x * 128 * QT;
  const f16* Qg = (const f16*)p.Q + (int64_t)b * p.Sq * p.ldq + h * p.hd;
  const f16* Kg = (const f16*)p.K + (int64_t)b * p.Skv * p.ldk + h * p.hd;
  const f16* Vg = (const f16*)p.V + (int64_t)b * p.Skv * p.ldv + h * p.hd;
  const f16* dOg = (const f16*)p.dO + (int64_t)b * p.Sq * p.lddo + h * p.hd;
  const f16* Og = (const f16*)p.O + (int64_t)b * p.Sq * p.ldo + h * p.hd;
  const bool three = p.Skv > KVT;
  f16x8 qall[QT][KS], doall[QT][KS], oall[QT][KS];
  float lse2[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int q = qblk + (qt * 4 + wave) * 32 + l31;
    load_row_frags<KS>(qall[qt], Qg, p.ldq, q, p.Sq, p.hd, hi);
    load_row_frags<KS>(doall[qt], dOg, p.lddo, q, p.Sq, p.hd, hi);
    load_row_frags<KS>(oall[qt], Og, p.ldo, q, p.Sq, p.hd, hi);
    lse2[qt] = p.LSE[((int64_t)b * p.H + h) * p.Sq + (q < p.Sq ? q : 0)] * LOG2E;
  }
  {
    TileRegs<WD> kreg, vreg;
    tile_init<WD>(kreg, p.ldk, p.hd);
    tile_init<WD>(vreg, p.ldv, p.hd);
    tile_load<WD, true>(kreg, Kg, p.ldk, 0, p.Skv, p.hd);
    tile_load<WD, true>(vreg, Vg, p.ldv, 0, p.Skv, p.hd);
    tile_store<WD, true, true>(kreg, K0, T0);
    tile_store<WD, true, false>(vreg, V0, nullptr);
    if (three) {
      tile_load<WD, true>(kreg, Kg, p.ldk, KVT, p.Skv, p.hd);
      tile_load<WD, true>(vreg, Vg, p.ldv, KVT, p.Skv, p.hd);
      tile_store<WD, true, true>(kreg, K1, T1);
      tile_store<WD, true, false>(vreg, V1, nullptr);
    }
  }
  __syncthreads();
  f16x8 kf[3][KS], vf[3][KS];
#pragma unroll
  for (int j = 0; j < KS; ++j) {
    kf[0][j] = frag_rm<WD>(K0, l31, 2 * j + hi);
    kf[1][j] = frag_rm<WD>(K0, 32 + l31, 2 * j + hi);
    kf[2][j] = frag_rm<WD>(three ? K1 : K0, l31, 2 * j + hi);
    vf[0][j] = frag_rm<WD>(V0, l31, 2 * j + hi);
    vf[1][j] = frag_rm<WD>(V0, 32 + l31, 2 * j + hi);
    vf[2][j] = frag_rm<WD>(three ? V1 : V0, l31, 2 * j + hi);
  }
  const float c = p.scale * LOG2E;
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int q = qblk + (qt * 4 + wave) * 32 + l31;
    const bool qok = q < p.Sq;
    const int64_t sidx = ((int64_t)b * p.H + h) * p.Sq + (qok ? q : 0);
    f16x8 (&qf)[KS] = qall[qt];
    f16x8 (&dof)[KS] = doall[qt];
    scale_frags<KS>(qf, c);
    float a = 0.f;   // delta = rowsum(dO * O) (padding chunks of both rows are zero)
#pragma unroll
    for (int j = 0; j < KS; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) a += (float)oall[qt][j][e] * (float)dof[j][e];
    const float delta = a + __shfl_xor(a, 32, 64);
    if (qok && hi == 0) p.Delta[sidx] = delta;
    f32x16 dq[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) ZERO16(dq[d]);
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
      if (kt < 2 || three) {
        f32x16 sc, dp;
        FILL16(sc, -lse2[qt]);
        FILL16(dp, -delta);
#pragma unroll
        for (int j = 0; j < KS; ++j) {
          sc = TB_MFMA_32x32x16(kf[kt][j], qf[j], sc);
          dp = TB_MFMA_32x32x16(vf[kt][j], dof[j], dp);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * 32 + mfma32_row(r, hi);
          const bool ok = qok && key < p.Skv;
          sc[r] = ok ? fast_exp2(sc[r]) * dp[r] : 0.f;  // dS^T / scale
        }
        const f16* Kt = kt < 2 ? T0 : T1;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const f16x8 dsf = pack8(sc, 8 * jj);
#pragma unroll
          for (int d = 0; d < DT; ++d) dq[d] = TB_MFMA_32x32x16(frag_tr(Kt, d, l31, (kt & 1) * 32 + 16 * jj, hi), dsf, dq[d]);
        }
      }
    }
    if (qok) {
      f16* dQg = (f16*)p.dQ + ((int64_t)b * p.Sq + q) * p.lddq + h * p.hd;
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int col = d * 32 + 8 * r4 + 4 * hi;
          if (col < p.hd) {
            f16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (f16)(dq[d][4 * r4 + e] * p.scale);
            *(f16x4*)(dQg + col) = v;
          }
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------ dK, dV
// qsplit > 1: blockIdx.x = key_block * qsplit + q_slice; each block covers a slice of the queries and writes its partial
// dK/dV in fp32 to ws32[slice][{K,V}][B*Skv][H*hd]; a finalize kernel sums the slices in a fixed order (deterministic)
// and rounds to fp16.  Used when there are too few key blocks to fill the chip (cross-attention: 77 keys).
template <int DT, int KS>
__global__ __launch_bounds__(256, (DT <= 2 ? 2 : 1)) void attn_bwd_dkv_kernel(const tb_attn_desc p, int qsplit, int q_chunk, float* ws32, int remap) {
  constexpr int WD = DT * 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  f16* Qs = reinterpret_cast<f16*>(smem_raw);
  f16* dOs = Qs + RM<WD>::SIZE;
  f16* Qt = dOs + RM<WD>::SIZE;
  f16* dOt = Qt + TR<WD>::SIZE;
  float* lse_s = reinterpret_cast<float*>(dOt + TR<WD>::SIZE);
  float* del_s = lse_s + KVT;
  const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform (SGPR): the mask tests below become scalar branches
  const AttnBlk blk = attn_block(remap);
  const int b = blk.b, h = blk.h;
  const int kblk = (blk.x / qsplit) * 128;
  const int qslice = blk.x % qsplit;
  const int key = kblk + wave * 32 + l31;
  const bool kok = key < p.Skv;
  const f16* Qg = (const f16*)p.Q + (int64_t)b * p.Sq * p.ldq + h * p.hd;
  const f16* Kg = (const f16*)p.K + (int64_t)b * p.Skv * p.ldk + h * p.hd;
  const f16* Vg = (const f16*)p.V + (int64_t)b * p.Skv * p.ldv + h * p.hd;
  const f16* dOg = (const f16*)p.dO + (int64_t)b * p.Sq * p.lddo + h * p.hd;
  const float* LSEg = p.LSE + ((int64_t)b * p.H + h) * p.Sq;
  const float* DELg = p.Delta + ((int64_t)b * p.H + h) * p.Sq;
  f16x8 kf[KS], vf[KS];
  load_row_frags<KS>(kf, Kg, p.ldk, key, p.Skv, p.hd, hi);
  load_row_frags<KS>(vf, Vg, p.ldv, key, p.Skv, p.hd, hi);
  scale_frags<KS>(kf, p.scale * LOG2E);  // the lane-owned operand carries scale * log2(e) (K here, Q in the other two kernels)
  f32x16 dk[DT], dv[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d) {
    ZERO16(dk[d]);
    ZERO16(dv[d]);
  }
  int q_begin = qslice * q_chunk;
  const int q_end = min(p.Sq, q_begin + q_chunk);
  if (p.causal) q_begin = max(q_begin, (kblk / KVT) * KVT);  // queries before the block's first key see none of its keys
  constexpr bool PF = DT <= 2;
  TileRegs<WD> qreg, doreg;
  tile_init<WD>(qreg, p.ldq, p.hd);
  tile_init<WD>(doreg, p.lddo, p.hd);
  if (PF && q_begin < q_end) {
    tile_load<WD, (DT <= 2)>(qreg, Qg, p.ldq, q_begin, p.Sq, p.hd);
    tile_load<WD, (DT <= 2)>(doreg, dOg, p.lddo, q_begin, p.Sq, p.hd);
  }
  for (int q0 = q_begin; q0 < q_end; q0 += KVT) {
    __syncthreads();
    if (!PF) {
      tile_load<WD, (DT <= 2)>(qreg, Qg, p.ldq, q0, p.Sq, p.hd);
      tile_load<WD, (DT <= 2)>(doreg, dOg, p.lddo, q0, p.Sq, p.hd);
    }
    tile_store<WD, true, true>(qreg, Qs, Qt);
    tile_store<WD, true, true>(doreg, dOs, dOt);
    if (threadIdx.x < KVT) {
      const int qq = q0 + threadIdx.x;
      lse_s[threadIdx.x] = qq < p.Sq ? -LSEg[qq] * LOG2E : 0.f;  // negated: they are the accumulator inputs of the two products
      del_s[threadIdx.x] = qq < p.Sq ? -DELg[qq] : 0.f;
    }
    __syncthreads();
    if (PF && q0 + KVT < q_end) {
      tile_load<WD, (DT <= 2)>(qreg, Qg, p.ldq, q0 + KVT, p.Sq, p.hd);
      tile_load<WD, (DT <= 2)>(doreg, dOg, p.lddo, q0 + KVT, p.Sq, p.hd);
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      // per-row -lse / -delta: rows of register quad g are 8g + 4hi + {0..3} -> one 16-byte LDS read per quad.  FOLD (see the dQ
      // kernel): they are the accumulator inputs of the two products; otherwise they are added to the results below
      constexpr bool FOLD = DT <= 2;
      f32x16 s, dp, nl, nd;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 lq = *(const f32x4*)(lse_s + qt * 32 + 8 * g4 + 4 * hi);
        const f32x4 dq4 = *(const f32x4*)(del_s + qt * 32 + 8 * g4 + 4 * hi);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          nl[4 * g4 + e] = lq[e];
          nd[4 * g4 + e] = dq4[e];
        }
      }
      if (FOLD) {
        s = nl;
        dp = nd;
      } else {
        ZERO16(s);
        ZERO16(dp);
      }
#pragma unroll
      for (int j = 0; j < KS; ++j) {
        s = TB_MFMA_32x32x16(frag_rm<WD>(Qs, qt * 32 + l31, 2 * j + hi), kf[j], s);
        dp = TB_MFMA_32x32x16(frag_rm<WD>(dOs, qt * 32 + l31, 2 * j + hi), vf[j], dp);
      }
      const bool need_mask = (q0 + qt * 32 + 32 > p.Sq) || (kblk + wave * 32 + 32 > p.Skv) || (p.causal && kblk + wave * 32 + 31 > q0 + qt * 32);
      if (need_mask) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int qq = q0 + qt * 32 + mfma32_row(r, hi);
          const bool ok = kok && qq < p.Sq && (!p.causal || key <= qq);
          const float pv = ok ? fast_exp2(FOLD ? s[r] : s[r] + nl[r]) : 0.f;
          s[r] = pv;                                       // P
          dp[r] = pv * (FOLD ? dp[r] : dp[r] + nd[r]);     // dS / scale
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = fast_exp2(FOLD ? s[r] : s[r] + nl[r]);
          s[r] = pv;
          dp[r] = pv * (FOLD ? dp[r] : dp[r] + nd[r]);
        }
      }
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const f16x8 pf = pack8(s, 8 * jj);
        const f16x8 dsf = pack8(dp, 8 * jj);
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          dv[d] = TB_MFMA_32x32x16(frag_tr(dOt, d, l31, qt * 32 + 16 * jj, hi), pf, dv[d]);
          dk[d] = TB_MFMA_32x32x16(frag_tr(Qt, d, l31, qt * 32 + 16 * jj, hi), dsf, dk[d]);
        }
      }
    }
  }
  if (kok && qsplit > 1) {
    const int64_t C = (int64_t)p.H * p.hd;
    const int64_t plane = (int64_t)p.B * p.Skv * C;
    float* dK32 = ws32 + (int64_t)qslice * 2 * plane + ((int64_t)b * p.Skv + key) * C + h * p.hd;
    float* dV32 = dK32 + plane;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int col = d * 32 + 8 * r4 + 4 * hi;
        if (col < p.hd) {
          f32x4 a, bb;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            a[e] = dk[d][4 * r4 + e] * p.scale;
            bb[e] = dv[d][4 * r4 + e];
          }
          *(f32x4*)(dK32 + col) = a;
          *(f32x4*)(dV32 + col) = bb;
        }
      }
  } else if (kok) {
    f16* dKg = (f16*)p.dK + ((int64_t)b * p.Skv + key) * p.lddk + h * p.hd;
    f16* dVg = (f16*)p.dV + ((int64_t)b * p.Skv + key) * p.lddv + h * p.hd;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int col = d * 32 + 8 * r4 + 4 * hi;
        if (col < p.hd) {
          f16x4 a, bb;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            a[e] = (f16)(dk[d][4 * r4 + e] * p.scale);
            bb[e] = (f16)dv[d][4 * r4 + e];
          }
          *(f16x4*)(dKg + col) = a;
          *(f16x4*)(dVg + col) = bb;
        }
      }
  }
}

// ------------------------------------------------------------------------------------------------ dQ, dK, dV in ONE launch, short key sequences
// Round 5 (diffusers BasicTransformerBlock.attn2 backward, train_textboost.py:1108): the cross-attention backward on the 77 prompt tokens was three
// launches -- attn_xs_bwd_dq_kernel (reads Q, dO, O; publishes delta), attn_bwd_dkv_kernel over query slices (reads Q, dO again) and the finalize of its
// fp32 slices -- 72 / 51 us per layer on the 64x64 / 32x32 maps for work that is bounded by reading Q and dO once and writing dQ once.  Here a
// workgroup owns a query slice, stages each 64-query tile of Q and dO ONCE (row-major + transposed images, as attn_bwd_dkv_kernel) beside resident
// images of all <= 96 keys (K scaled by scale * log2 e, V, K transposed), and computes
//   * delta = rowsum(P * dP) instead of rowsum(dO * O)  (O = P V, so dO . O = sum_k P_k (dO . V_k) = sum_k P_k dP_k): the attention OUTPUT is not read;
//   * dQ in the lane-owns-a-query orientation (S^T = K Q^T, dQ^T = K^T dS^T): waves 0 / 1 make delta for query groups 0 / 1, waves 2 / 3 then dQ;
//   * dK / dV in the lane-owns-a-key orientation (waves 0 .. 2 = keys 0 .. 95), accumulated over the slice's tiles and left as fp32 partials
//     [slice][{K, V}][B * Skv][H * hd] for attn_dkv_finalize_kernel (same layout and order as attn_bwd_dkv_kernel's: deterministic).
// The scores are recomputed per orientation (77 keys: the matrix work is small beside the traffic).  Host: no causal mask, 32 < Skv <= 96, Sq % 64 == 0.
template <int DT, int KS, int PC, int NST>
__global__ __launch_bounds__(256, (DT <= 2 ? 2 : 1)) void attn_xs_bwd_kernel(const tb_attn_desc p, int q_chunk, float* ws32, int remap) {
  // Staging as attn_bwd_dkv_dma_kernel: the 64-query Q / dO tiles (and their LSE values) go HBM -> LDS by global_load_lds, row-major only, through
  // an NST-slot ring with counted vmcnt; row-major fragments by ds_read_b128, the transposed Q^T / dO^T operands of dK / dV by ds_read_b64_tr_b16
  // from the same rows.  (The first version staged through registers one tile ahead, like attn_bwd_dkv_kernel: 56 us at the 64x64 maps, the sum of
  // the two launches it replaced -- 10 KB per workgroup in flight is a quarter of what the memory latency needs.)
  constexpr int PCB = PC * 16, TILE_B = KVT * PCB;
  constexpr int STAGE_B = 2 * TILE_B + KVT * 4 + 64;   // Q tile, dO tile, lse[64], slack for the tr reads past the last pad chunk
  constexpr int NI = 2 * PC + 1, WI = (NI + 3) / 4;
  constexpr int TP = 100;                               // row pitch (halfs) of the transposed K image [32 DT][96 keys]
  static_assert(NST >= 3, "loads run two tiles ahead");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char* const Kc = smem_raw + NST * STAGE_B;   // [96][PCB]  K * scale * log2(e), zero outside Skv x hd
  unsigned char* const Vr = Kc + 96 * PCB;              // [96][PCB]
  f16* const Tk = reinterpret_cast<f16*>(Vr + 96 * PCB);   // [32 DT][TP]  K^T (unscaled)
  float* const del_s = reinterpret_cast<float*>(Tk + 32 * DT * TP);   // [64]  -delta of the current tile
  const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave_hw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // Roles by a ROTATED wave index: two workgroups share a CU and wave w of each sits on SIMD w -- with the same roles in both, the delta phase
  // (two waves) ran on SIMDs 0 / 1 of both while SIMDs 2 / 3 idled.  Workgroups 256 apart in launch order (the likely CU mates) rotate by two.
  const int wave = (wave_hw + 2 * (((blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) >> 8) & 1)) & 3;
  AttnBlk blk;   // (query slice, batch) pairs per XCD, heads back to back (as attn_xs_bwd_dq_kernel)
  {
    const int gx = gridDim.x, H = gridDim.y, B = gridDim.z;
    if (remap && ((gx * B) & 7) == 0) {
      const int lin = blockIdx.x + gx * (blockIdx.y + H * blockIdx.z);
      const int xcd = lin & 7, k = lin >> 3;
      const int xb = (k / H) * 8 + xcd;
      blk.h = k - (k / H) * H;
      blk.x = xb % gx;
      blk.b = xb / gx;
    } else {
      blk.x = blockIdx.x, blk.h = blockIdx.y, blk.b = blockIdx.z;
    }
  }
  const int b = blk.b, h = blk.h, qslice = blk.x, hd = p.hd;
  const int q_begin = qslice * q_chunk;
  const int q_end = min(p.Sq, q_begin + q_chunk);
  const int ntiles = (q_end - q_begin) / KVT;
  const int64_t ldq = p.ldq, lddo = p.lddo;
  const char* Qg = (const char*)((const f16*)p.Q + ((int64_t)b * p.Sq + q_begin) * ldq + h * hd);
  const char* dOg = (const char*)((const f16*)p.dO + ((int64_t)b * p.Sq + q_begin) * lddo + h * hd);
  const f16* Kg = (const f16*)p.K + (int64_t)b * p.Skv * p.ldk + h * hd;
  const f16* Vg = (const f16*)p.V + (int64_t)b * p.Skv * p.ldv + h * hd;
  const char* LSEg = (const char*)(p.LSE + ((int64_t)b * p.H + h) * p.Sq + q_begin);
  const float c = p.scale * LOG2E;
  // ---- this lane's part of a stage's loads: instruction t = wave + 4 i; t < PC: Q rows, t < 2 PC: dO rows, t == 2 PC: the 64 LSE values
  uint32_t g_off[WI];
  bool g_on[WI];
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    const int t = wave_hw + 4 * i;
    const int tensor = t >= PC ? 1 : 0;
    const int f = (t - tensor * PC) * 64 + lane;
    const int row = f / PC, cc = f - row * PC;
    g_on[i] = t < 2 * PC && cc < PC - 1;
    g_off[i] = (uint32_t)((int64_t)row * (tensor ? lddo : ldq) * 2 + cc * 16);
  }
  int n_issued = 0;
#pragma unroll
  for (int i = 0; i < WI; ++i) n_issued += (wave_hw + 4 * i < NI) ? 1 : 0;
  auto stage_loads = [&](int tile, int slot) {
    unsigned char* dst = smem_raw + slot * STAGE_B;
    const char* qb = Qg + (int64_t)tile * KVT * ldq * 2;
    const char* ob = dOg + (int64_t)tile * KVT * lddo * 2;
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int t = wave_hw + 4 * i;
      if (t < 2 * PC) {
        const char* src = (t >= PC ? ob : qb) + g_off[i];
        if (g_on[i]) __builtin_amdgcn_global_load_lds((attn_gptr_t)src, (attn_lptr_t)(dst + t * 1024), 16, 0, 0);
      } else if (t < NI) {  // 64 floats = one dword per lane
        const char* src = LSEg + ((int64_t)tile * KVT + lane) * 4;
        __builtin_amdgcn_global_load_lds((attn_gptr_t)src, (attn_lptr_t)(dst + 2 * TILE_B), 4, 0, 0);
      }
    }
  };
#pragma unroll
  for (int st = 0; st < NST - 1; ++st)
    if (st < ntiles) stage_loads(st, st);
  // pad chunks of every row of every stage: zeros (they meet the zero padding of the key images, but must be finite)
  for (int u = threadIdx.x; u < 2 * KVT * NST; u += 256) {
    const int st = u / (2 * KVT), r = u - st * 2 * KVT;
    const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    *(f16x8*)(smem_raw + st * STAGE_B + (r >= KVT ? TILE_B : 0) + (r & (KVT - 1)) * PCB + (PC - 1) * 16) = z;
  }
  // ---- resident key images: unit = (key, 16-byte chunk); zero outside Skv x hd (plain loads: older than nothing the counted waits protect)
  for (int u = threadIdx.x; u < 96 * PC; u += 256) {
    const int key = u / PC, ch = u - key * PC;
    const bool ok = key < p.Skv && ch * 8 < hd;
    const f16x8 kv = *(ok ? (gvec8_t)(Kg + (int64_t)key * p.ldk + ch * 8) : (gvec8_t)g_zero8);
    const f16x8 vv = *(ok ? (gvec8_t)(Vg + (int64_t)key * p.ldv + ch * 8) : (gvec8_t)g_zero8);
    f16x8 ks;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      ks[e] = (f16)((float)kv[e] * c);
      if (ch * 8 + e < 32 * DT) Tk[(ch * 8 + e) * TP + key] = kv[e];
    }
    *(f16x8*)(Kc + key * PCB + ch * 16) = ks;
    *(f16x8*)(Vr + key * PCB + ch * 16) = vv;
  }
  if (PC * 8 < 32 * DT) {   // rows of the transposed image behind the last chunk (hd = 40: rows 48 .. 63): finite
    for (int u = threadIdx.x; u < (32 * DT - PC * 8) * 96; u += 256) Tk[(PC * 8 + u / 96) * TP + (u % 96)] = (f16)0.f;
  }
  const int key_b = wave * 32 + l31;          // this lane's key in the dK / dV orientation (waves 0 .. 2)
  const bool kok = wave < 3 && key_b < p.Skv;
  f32x16 dk[DT], dv[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d) {
    ZERO16(dk[d]);
    ZERO16(dv[d]);
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)(attn_lptr_t)smem_raw;
  const uint32_t rm_lane = l31 * PCB + hi * 16;   // row-major fragment: row l31 (+ 32 g), chunk 2 j + hi
  const int g4l = lane >> 4, j16 = lane & 15;
  const uint32_t tr_lane = (4 * (g4l >> 1) + (j16 >> 2)) * PCB + ((g4l & 1) * 16 + 4 * (j16 & 3)) * 2;  // as attn_bwd_dkv_dma_kernel's Q^T / dO^T reads
  int slot = 0, lslot = NST - 1;
  for (int t = 0; t < ntiles; ++t) {
    {
      int later = ntiles - 1 - t;
      later = later > NST - 2 ? NST - 2 : later;
      attn_wait_vmcnt(later * n_issued);   // (waves 2 / 3 also have dQ stores in flight: the count then waits for a little more than it must)
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // tile t has landed; every wave has left tile t - 1 (first pass: the key images)
    if (t + NST - 1 < ntiles) stage_loads(t + NST - 1, lslot);
    const unsigned char* Qs = smem_raw + slot * STAGE_B;
    const float* lse_s = (const float*)(Qs + 2 * TILE_B);
    // ---- delta of query group g = wave (waves 0, 1): rowsum over the keys of P * dP, lane = query
    if (wave < 2) {
      const int g = wave;
      const float nl = -lse_s[g * 32 + l31] * LOG2E;
      float acc = 0.f;
#pragma unroll
      for (int kt = 0; kt < 3; ++kt) {
        f32x16 sc, dp;
        FILL16(sc, nl);
        ZERO16(dp);
#pragma unroll
        for (int j = 0; j < KS; ++j) {
          sc = TB_MFMA_32x32x16(*(const f16x8*)(Kc + (kt * 32 + l31) * PCB + (2 * j + hi) * 16), *(const f16x8*)(Qs + rm_lane + g * 32 * PCB + j * 32), sc);
          dp = TB_MFMA_32x32x16(*(const f16x8*)(Vr + (kt * 32 + l31) * PCB + (2 * j + hi) * 16), *(const f16x8*)(Qs + TILE_B + rm_lane + g * 32 * PCB + j * 32), dp);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * 32 + mfma32_row(r, hi);
          if (key < p.Skv) acc += fast_exp2(sc[r]) * dp[r];
        }
      }
      const float delta = acc + __shfl_xor(acc, 32, 64);
      if (hi == 0) del_s[g * 32 + l31] = -delta;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // ---- dQ of query group g = wave - 2 (waves 2, 3): dS^T = P (dP - delta), dQ^T = K^T dS^T
    if (wave >= 2) {
      const int g = wave - 2;
      const float nl = -lse_s[g * 32 + l31] * LOG2E, nd = del_s[g * 32 + l31];
      f32x16 dq[DT];
#pragma unroll
      for (int d = 0; d < DT; ++d) ZERO16(dq[d]);
#pragma unroll
      for (int kt = 0; kt < 3; ++kt) {
        f32x16 sc, dp;
        FILL16(sc, nl);
        FILL16(dp, nd);
#pragma unroll
        for (int j = 0; j < KS; ++j) {
          sc = TB_MFMA_32x32x16(*(const f16x8*)(Kc + (kt * 32 + l31) * PCB + (2 * j + hi) * 16), *(const f16x8*)(Qs + rm_lane + g * 32 * PCB + j * 32), sc);
          dp = TB_MFMA_32x32x16(*(const f16x8*)(Vr + (kt * 32 + l31) * PCB + (2 * j + hi) * 16), *(const f16x8*)(Qs + TILE_B + rm_lane + g * 32 * PCB + j * 32), dp);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * 32 + mfma32_row(r, hi);
          sc[r] = key < p.Skv ? fast_exp2(sc[r]) * dp[r] : 0.f;   // dS^T / scale
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const f16x8 dsf = pack8(sc, 8 * jj);
#pragma unroll
          for (int d = 0; d < DT; ++d) {
            const f16* row = Tk + (d * 32 + l31) * TP + kt * 32 + 16 * jj + 4 * hi;   // contraction indices {k0 + 4 hi .. + 3, k0 + 8 + 4 hi .. + 3}
            const f16x4 a0 = *(const f16x4*)row, a1 = *(const f16x4*)(row + 8);
            f16x8 a;
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = a0[e], a[4 + e] = a1[e];
            dq[d] = TB_MFMA_32x32x16(a, dsf, dq[d]);
          }
        }
      }
      f16* dQg = (f16*)p.dQ + ((int64_t)b * p.Sq + q_begin + t * KVT + g * 32 + l31) * p.lddq + h * hd;
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int col = d * 32 + 8 * r4 + 4 * hi;
          if (col < hd) {
            f16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (f16)(dq[d][4 * r4 + e] * p.scale);
            *(f16x4*)(dQg + col) = v;
          }
        }
    }
    // ---- dK / dV of keys 32 wave .. + 31 (waves 0 .. 2), both 32-query halves: the inner loop of attn_bwd_dkv_dma_kernel
    if (wave < 3) {
      f16x8 kf[KS], vf[KS];
#pragma unroll
      for (int j = 0; j < KS; ++j) {
        kf[j] = *(const f16x8*)(Kc + (wave * 32 + l31) * PCB + (2 * j + hi) * 16);
        vf[j] = *(const f16x8*)(Vr + (wave * 32 + l31) * PCB + (2 * j + hi) * 16);
      }
      const uint32_t qa = lds0 + slot * STAGE_B + tr_lane;
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        f32x16 s, dp;  // start from -lse2 / -delta of the tile's queries: rows of register quad g are 8 g + 4 hi + {0..3}
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const f32x4 lq = *(const f32x4*)(lse_s + qt * 32 + 8 * q4 + 4 * hi);
          const f32x4 dq4 = *(const f32x4*)(del_s + qt * 32 + 8 * q4 + 4 * hi);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            s[4 * q4 + e] = -lq[e] * LOG2E;
            dp[4 * q4 + e] = dq4[e];
          }
        }
#pragma unroll
        for (int j = 0; j < KS; ++j) {
          const f16x8 qfr = *(const f16x8*)(Qs + rm_lane + qt * 32 * PCB + j * 32);
          const f16x8 dofr = *(const f16x8*)(Qs + TILE_B + rm_lane + qt * 32 * PCB + j * 32);
          s = TB_MFMA_32x32x16(qfr, kf[j], s);
          dp = TB_MFMA_32x32x16(dofr, vf[j], dp);
        }
        f16x4 qtf[2][DT][2], dotf[2][DT][2];
        __builtin_amdgcn_sched_barrier(0);
#define TB_TR(ARR, BASE, QT, JJ, D, HH) ARR[JJ][D][HH] = lds_tr_read_off<(BASE) + ((QT) * 32 + 16 * (JJ) + 8 * (HH)) * PCB + (D) * 64>(qa);
#define TB_TR_D(ARR, BASE, QT, D) TB_TR(ARR, BASE, QT, 0, D, 0) TB_TR(ARR, BASE, QT, 0, D, 1) TB_TR(ARR, BASE, QT, 1, D, 0) TB_TR(ARR, BASE, QT, 1, D, 1)
#define TB_TR_ALL(QT)                                                                   \
  TB_TR_D(qtf, 0, QT, 0) TB_TR_D(qtf, 0, QT, 1)                                         \
  if (DT > 2) { TB_TR_D(qtf, 0, QT, (DT > 2 ? 2 : 0)) }                                 \
  TB_TR_D(dotf, TILE_B, QT, 0) TB_TR_D(dotf, TILE_B, QT, 1)                             \
  if (DT > 2) { TB_TR_D(dotf, TILE_B, QT, (DT > 2 ? 2 : 0)) }
        static_assert(DT == 2 || DT == 3, "hd = 40 / 64 / 80 instantiations");
        if (qt == 0) { TB_TR_ALL(0) } else { TB_TR_ALL(1) }
#undef TB_TR_D
#undef TB_TR_ALL
#undef TB_TR
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = kok ? fast_exp2(s[r]) : 0.f;
          s[r] = pv;            // P
          dp[r] = pv * dp[r];   // dS / scale
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the transposed fragments (inline-asm reads: not counted by the compiler)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const f16x8 pf = pack8(s, 8 * jj);
          const f16x8 dsf = pack8(dp, 8 * jj);
#pragma unroll
          for (int d = 0; d < DT; ++d) {
            f16x8 a, cq;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              a[e] = dotf[jj][d][0][e], a[4 + e] = dotf[jj][d][1][e];
              cq[e] = qtf[jj][d][0][e], cq[4 + e] = qtf[jj][d][1][e];
            }
            dv[d] = TB_MFMA_32x32x16(a, pf, dv[d]);
            dk[d] = TB_MFMA_32x32x16(cq, dsf, dk[d]);
          }
        }
      }
    }
    slot = slot == NST - 1 ? 0 : slot + 1;
    lslot = lslot == NST - 1 ? 0 : lslot + 1;
  }
  if (kok) {
    const int64_t C = (int64_t)p.H * hd;
    const int64_t plane = (int64_t)p.B * p.Skv * C;
    float* dK32 = ws32 + (int64_t)qslice * 2 * plane + ((int64_t)b * p.Skv + key_b) * C + h * hd;
    float* dV32 = dK32 + plane;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int col = d * 32 + 8 * r4 + 4 * hi;
        if (col < hd) {
          f32x4 a, bb;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            a[e] = dk[d][4 * r4 + e] * p.scale;
            bb[e] = dv[d][4 * r4 + e];
          }
          *(f32x4*)(dK32 + col) = a;
          *(f32x4*)(dV32 + col) = bb;
        }
      }
  }
}

__global__ __launch_bounds__(256) void attn_dkv_finalize_kernel(const float* __restrict__ ws32, f16* __restrict__ dK, int64_t lddk,
                                                                f16* __restrict__ dV, int64_t lddv, int64_t rows, int C, int qsplit) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t plane = rows * C;
  if (idx >= plane) return;
  const int64_t m = idx / C;
  const int c = (int)(idx - m * C);
  float a = 0.f, b = 0.f;
  for (int s = 0; s < qsplit; ++s) {
    a += ws32[(int64_t)s * 2 * plane + idx];
    b += ws32[(int64_t)s * 2 * plane + plane + idx];
  }
  dK[m * lddk + c] = (f16)a;
  dV[m * lddv + c] = (f16)b;
}

int g_attn_dma = 1;  // tb_attention_set_variant(bits): 1 = LDS-DMA staged forward kernel, 2 = also for hd 80, 4 = NO XCD-aware block remap,
                     // 1024 / 2048 = LDS-DMA kernels instead of the software-pipelined forward / dK-dV kernels (attention_il.hip), 4096 = the
                     // software-pipelined dQ kernel (opt-in), 16384 = the general flash kernel also for short key sequences (no attn_xs_fwd_kernel), 32768 = query slices of the cross-attention dK / dV launch until 1024 (not 512) workgroups,
                     // 32 = one query group per wave, 64 = XCD remap also in the backward kernels (measured: forward +14 %, backward -5 %)
                     // (A/B knobs; 8 / 16 = load-path ablations of the DMA kernel)

}  // namespace
extern "C" int64_t tb_attention_fp8_ws_bytes(int B, int H, int Skv) {
  const int64_t BH = (int64_t)B * H;
  return BH * V8_ROWS * Skv + BH * 4 + BH * ((Skv + 255) / 256) * 4;
}
namespace {

template <int DT, int KS>
int launch_fwd(const tb_attn_desc& d, hipStream_t s) {
  constexpr int WD = DT * 32;
  size_t lds = (RM<WD>::SIZE + TR<WD>::SIZE) * sizeof(f16) * (DT <= 3 ? 2 : 1);  // double-buffered K / V^T tiles
  dim3 grid((d.Sq + 127) / 128, d.H, d.B);
  if ((g_attn_dma & 1) && !d.causal && d.Sq % 128 == 0 && d.Skv % KVT == 0 && d.hd % 8 == 0 && d.hd < WD && (d.hd == 40 || d.hd == 80) &&
      d.ldk % 8 == 0 && d.ldv % 8 == 0 && (int64_t)KVT * (d.ldk > d.ldv ? d.ldk : d.ldv) * 2 < ((int64_t)1 << 31)) {
    // LDS-DMA staged kernel (see attn_fwd_dma_kernel): the SD1.x self-attention shapes, hd = 40 (64x64 maps) and 80 (32x32 maps)
    const int rm = (((g_attn_dma >> 2) & 1) ^ 1) | (g_attn_dma & 24);
    if (DT == 2 && KS == 3 && d.hd == 40 && d.fp8_ws && d.Sq % 256 == 0 && d.Skv % 256 == 0 &&
        d.fp8_ws_bytes >= tb_attention_fp8_ws_bytes(d.B, d.H, d.Skv) && (int64_t)V8_ROWS * d.Skv < ((int64_t)1 << 31)) {
      constexpr int PC = 6, NST = 4;
      const size_t lds8 = NST * (KVT * PC * 16 + 64 * V8_PITCH);
      hipLaunchKernelGGL(attn_v8t_kernel, dim3(d.Skv / 256, d.B * d.H), dim3(256), 0, s, d, 0);
      hipLaunchKernelGGL(attn_v8t_kernel, dim3(d.Skv / 256, d.B * d.H), dim3(256), 0, s, d, 1);
      hipLaunchKernelGGL((attn_fwd_dma_kernel<2, 3, PC, true, NST, 2, true>), dim3(d.Sq / 256, d.H, d.B), dim3(256), lds8, s, d, rm);
      TB_CHECK_LAUNCH();
      return TB_OK;
    }
    if (DT == 2 && KS == 3 && d.hd == 40 && !(g_attn_dma & 1024) && tb_attn_il_fwd_ok(d)) return tb_attn_il_fwd(d, s, rm);  // software-pipelined kernel
    if (DT == 2 && KS == 3 && d.hd == 40) {
      constexpr int PC = 6, NST = 4;
      const size_t lds = NST * (2 * KVT * PC * 16 + 64);
      if (d.Sq % 256 == 0 && !(g_attn_dma & 32))
        hipLaunchKernelGGL((attn_fwd_dma_kernel<2, 3, PC, true, NST, 2>), dim3(d.Sq / 256, d.H, d.B), dim3(256), lds, s, d, rm);
      else
        hipLaunchKernelGGL((attn_fwd_dma_kernel<2, 3, PC, true, NST, 1>), grid, dim3(256), lds, s, d, rm);
      TB_CHECK_LAUNCH();
      return TB_OK;
    }
    if (DT == 3 && KS == 5 && d.hd == 80 && (g_attn_dma & 2)) {
      constexpr int PC = 11, NST = 3;
      const size_t lds = NST * (2 * KVT * PC * 16 + 64);
      static bool attr_done = false;
      if (!attr_done) {
        if (hipFuncSetAttribute((const void*)attn_fwd_dma_kernel<3, 5, PC, false, NST, 1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
          return TB_ELAUNCH;
        attr_done = true;
      }
      hipLaunchKernelGGL((attn_fwd_dma_kernel<3, 5, PC, false, NST, 1>), grid, dim3(256), lds, s, d, rm);
      TB_CHECK_LAUNCH();
      return TB_OK;
    }
  }
  if constexpr (DT <= 3) if (!d.causal && d.Skv <= 96 && d.Skv > 32 && d.hd % 8 == 0 && d.Sq % 128 == 0 && d.Sq >= 1024 && !(g_attn_dma & 16384)) {
    // short key sequences (cross-attention on the prompt): attn_xs_fwd_kernel, 256 or 512 queries per workgroup
    const int qtiles = DT <= 2 && d.Sq % 512 == 0 && (int64_t)(d.Sq / 512) * d.H * d.B >= 512 ? 4 : (d.Sq % 256 == 0 ? 2 : 1);   // (hd > 64: four tiles of Q do not fit)
    const size_t ldsx = (RM<WD>::SIZE + TR<WD>::SIZE) * sizeof(f16) * 2;
    const int rmx = ((g_attn_dma >> 2) & 1) ^ 1;
    const dim3 gx((unsigned)(d.Sq / (128 * qtiles)), d.H, d.B);
    if (qtiles == 4) hipLaunchKernelGGL((attn_xs_fwd_kernel<DT, KS, (DT <= 2 ? 4 : 2)>), gx, dim3(256), ldsx, s, d, rmx);
    else if (qtiles == 2) hipLaunchKernelGGL((attn_xs_fwd_kernel<DT, KS, 2>), gx, dim3(256), ldsx, s, d, rmx);
    else hipLaunchKernelGGL((attn_xs_fwd_kernel<DT, KS, 1>), gx, dim3(256), ldsx, s, d, rmx);
    TB_CHECK_LAUNCH();
    return TB_OK;
  }
  if (TB_ATTN_FOLDM && DT <= 2 && d.hd < WD && d.hd < 16 * KS && d.hd % 8 == 0)  // padding in the head dim of BOTH products: see FOLDM
    hipLaunchKernelGGL((attn_fwd_kernel<DT, KS, true, (DT <= 2)>), grid, dim3(256), lds, s, d, ((g_attn_dma >> 2) & 1) ^ 1);
  else if (d.hd < WD)  // head-dim padding exists: the row sum rides on the PV product (all-ones row hd of V^T)
    hipLaunchKernelGGL((attn_fwd_kernel<DT, KS, true, false>), grid, dim3(256), lds, s, d, ((g_attn_dma >> 2) & 1) ^ 1);
  else
    hipLaunchKernelGGL((attn_fwd_kernel<DT, KS, false, false>), grid, dim3(256), lds, s, d, ((g_attn_dma >> 2) & 1) ^ 1);
  TB_CHECK_LAUNCH();
  return TB_OK;
}
template <int DT, int KS>
int launch_bwd(const tb_attn_desc& d, hipStream_t s) {
  constexpr int WD = DT * 32;
  if constexpr ((DT == 2 && KS == 3) || (DT == 2 && KS == 4) || (DT == 3 && KS == 5)) {
    // cross-attention on the prompt (32 < Skv <= 96; hd = 40 / 64 / 80): dQ, dK, dV in one launch over query slices + the finalize of the dK / dV
    // slices (tb_attention_set_variant bit 65536 = the three-launch path of round 4)
    constexpr int PC = KS == 3 ? 6 : (KS == 4 ? 9 : 11), NST = 3;
    const int64_t C = (int64_t)d.H * d.hd, plane2 = 2 * (int64_t)d.B * d.Skv * C;
    // measured (scratch/xbwd_time.py, B = 8, 77 keys): hd = 40 at 64x64 maps 66 -> 59 us per layer, hd = 80 at 32x32 maps 48 -> 53 us (three staged
    // images and three score products per tile on one workgroup per CU): the wide heads (hd = 64, 80) keep the three launches unless bit 131072 asks
    const bool wide_ok = KS <= 3 || (g_attn_dma & 131072);   // (hd = 64, SD2.x at 96x96 latents: 107 -> 136 us, 63 -> 78, 45 -> 51 us: off as well)
    if (!(g_attn_dma & 65536) && wide_ok && d.hd == (KS == 3 ? 40 : (KS == 4 ? 64 : 80)) && !d.causal && d.Skv <= 96 && d.Skv > 32 && d.Sq % KVT == 0 && d.Sq >= 256 &&
        d.ws && d.ws_floats >= 2 * plane2 && d.ldq % 8 == 0 && d.lddo % 8 == 0 && (int64_t)KVT * (d.ldq > d.lddo ? d.ldq : d.lddo) * 2 < ((int64_t)1 << 31)) {
      const int64_t bh = (int64_t)d.H * d.B;
      int nsl = (int)((512 + bh - 1) / bh);                     // ~512 workgroups
      const int max_sl = d.Sq / (2 * KVT);                      // >= 2 tiles per slice
      if (nsl > max_sl) nsl = max_sl;
      if ((int64_t)nsl * plane2 > d.ws_floats) nsl = (int)(d.ws_floats / plane2);
      if (nsl < 1) nsl = 1;
      int q_chunk = ((d.Sq + nsl - 1) / nsl + KVT - 1) / KVT * KVT;
      nsl = (d.Sq + q_chunk - 1) / q_chunk;
      const size_t lds = (size_t)NST * (2 * KVT * PC * 16 + KVT * 4 + 64) + 2 * 96 * PC * 16 + (size_t)32 * DT * 100 * 2 + KVT * 4;
      static bool attr_x = false;
      if (!attr_x && lds > 65536) {
        if (hipFuncSetAttribute((const void*)attn_xs_bwd_kernel<DT, KS, PC, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
          return TB_ELAUNCH;
        attr_x = true;
      }
      hipLaunchKernelGGL((attn_xs_bwd_kernel<DT, KS, PC, NST>), dim3(nsl, d.H, d.B), dim3(256), lds, s, d, q_chunk, d.ws, ((g_attn_dma >> 2) & 1) ^ 1);
      const int64_t rows = (int64_t)d.B * d.Skv;
      hipLaunchKernelGGL(attn_dkv_finalize_kernel, dim3((unsigned)((rows * C + 255) / 256)), dim3(256), 0, s, d.ws, (f16*)d.dK, d.lddk,
                         (f16*)d.dV, d.lddv, rows, (int)C, nsl);
      TB_CHECK_LAUNCH();
      return TB_OK;
    }
  }
  // LDS-DMA staged dK/dV kernel: the SD1.x 64x64-map self-attention shape; needs 2 * B * H * Sq floats of ws for the statistics the dQ
  // kernel publishes for it (else, and for every other shape, the register-staged kernel runs)
  const bool dkv_dma = ((DT == 2 && KS == 3 && d.hd == 40) || (DT == 2 && KS == 4 && d.hd == 64 && !(g_attn_dma & 512)) ||
                        (DT == 3 && KS == 5 && d.hd == 80 && !(g_attn_dma & 512))) && !d.causal && d.Sq % KVT == 0 && d.Skv % 128 == 0 && !(g_attn_dma & 128) &&
                       d.ws && d.ws_floats >= 2 * (int64_t)d.B * d.H * d.Sq && (int64_t)(d.Skv / 128) * d.H * d.B >= 512 &&
                       d.ldq % 8 == 0 && d.lddo % 8 == 0 && (int64_t)KVT * (d.ldq > d.lddo ? d.ldq : d.lddo) * 2 < ((int64_t)1 << 31);
#ifndef TB_ATTN_FUSED_DELTA
  {
    int64_t total = (int64_t)d.B * d.Sq * d.H;
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d);
  }
#endif
  {
    size_t lds = (2 * RM<WD>::SIZE + TR<WD>::SIZE) * sizeof(f16);
    static bool attr_done = false;
    if (!attr_done && lds > 65536) {
      if (hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<DT, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return TB_ELAUNCH;
      attr_done = true;
    }
    dim3 grid((d.Sq + 127) / 128, d.H, d.B);
    bool dq_dma = false;
    if constexpr ((DT == 2 && KS == 3) || (DT == 2 && KS == 4) || (DT == 3 && KS == 5)) {
      // hd = 40: 6-chunk rows, 4 slots; hd = 64 (SD2.x): 9-chunk rows, 4 slots (74 KB); hd = 80: 11-chunk rows, 3 slots (68 KB)
      constexpr int PC = KS == 3 ? 6 : (KS == 4 ? 9 : 11), NST = DT == 2 ? 4 : 3;
      dq_dma = d.hd == (KS == 3 ? 40 : (KS == 4 ? 64 : 80)) && !(KS != 3 && (g_attn_dma & 512)) && !d.causal && d.Sq % 128 == 0 && d.Skv % KVT == 0 &&
               !(g_attn_dma & 256) && d.ldk % 8 == 0 && d.ldv % 8 == 0 &&
               (int64_t)KVT * (d.ldk > d.ldv ? d.ldk : d.ldv) * 2 < ((int64_t)1 << 31) && d.Skv >= 512;
      if (dq_dma && DT == 2 && KS == 3 && (g_attn_dma & 4096) && tb_attn_il_dq_ok(d)) {  // hd = 40: the software-pipelined dQ kernel, opt-in
        // (bit-equal results; PMC: 714 k cycles vs the LDS-DMA kernel's 697 k per launch at S = 4096, B = 8 -- no gain for this product mix)
        const int rc = tb_attn_il_dq(d, s, (g_attn_dma >> 6) & 1, dkv_dma ? 1 : 0);
        if (rc) return rc;
      } else if (dq_dma && DT == 2 && KS == 3 && !(g_attn_dma & 8192)) {   // hd = 40: 80-byte rows without the pad chunk (bit-equal results)
        constexpr int PCN = 5;
        const size_t ldsq = NST * (2 * KVT * PCN * 16 + 64);
        hipLaunchKernelGGL((attn_bwd_dq_dma_kernel<DT, KS, PCN, NST, false>), grid, dim3(256), ldsq, s, d, (g_attn_dma >> 6) & 1, dkv_dma ? 1 : 0);
      } else if (dq_dma) {
        const size_t ldsq = NST * (2 * KVT * PC * 16 + 64);
        static bool attr_q = false;
        if (!attr_q && ldsq > 65536) {
          if (hipFuncSetAttribute((const void*)attn_bwd_dq_dma_kernel<DT, KS, PC, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsq) !=
              hipSuccess)
            return TB_ELAUNCH;
          attr_q = true;
        }
        hipLaunchKernelGGL((attn_bwd_dq_dma_kernel<DT, KS, PC, NST>), grid, dim3(256), ldsq, s, d, (g_attn_dma >> 6) & 1, dkv_dma ? 1 : 0);
      }
    }
    bool dq_xs = false;
    if constexpr (DT <= 2) {
      // short key sequences (cross-attention on the prompt): attn_xs_bwd_dq_kernel, 256 queries per workgroup (128 when Sq is an odd multiple)
      if (!dq_dma && !dkv_dma && !d.causal && d.Skv <= 96 && d.Skv > 32 && d.Sq % 128 == 0 && d.Sq >= 1024 && !(g_attn_dma & 16384)) {
        dq_xs = true;
        const size_t ldsx = 2 * (2 * RM<WD>::SIZE + TR<WD>::SIZE) * sizeof(f16);
        const int rmx = ((g_attn_dma >> 2) & 1) ^ 1;
        if (d.Sq % 256 == 0 && KS <= 3) hipLaunchKernelGGL((attn_xs_bwd_dq_kernel<DT, KS, (KS <= 3 ? 2 : 1)>), dim3(d.Sq / 256, d.H, d.B), dim3(256), ldsx, s, d, rmx);   // (KS = 4 with two tiles in flight spills)
        else hipLaunchKernelGGL((attn_xs_bwd_dq_kernel<DT, KS, 1>), grid, dim3(256), ldsx, s, d, rmx);
      }
    }
    if (!dq_dma && !dq_xs) hipLaunchKernelGGL((attn_bwd_dq_kernel<DT, KS>), grid, dim3(256), lds, s, d, (g_attn_dma >> 6) & 1, dkv_dma ? 1 : 0);
  }
  if (dkv_dma) {
    if constexpr (DT == 2 && KS == 3) {  // hd = 40: the software-pipelined kernel (attention_il.hip)
      if (!(g_attn_dma & 2048) && tb_attn_il_dkv_ok(d)) return tb_attn_il_dkv(d, s, (g_attn_dma >> 6) & 1);
    }
    if constexpr ((DT == 2 && KS == 3) || (DT == 2 && KS == 4) || (DT == 3 && KS == 5)) {
      constexpr int PC = KS == 3 ? 6 : (KS == 4 ? 9 : 11), NST = DT == 2 ? 4 : 3;
      const size_t lds = NST * (2 * KVT * PC * 16 + 2 * KVT * 4 + 64);
      static bool attr_kv = false;
      if (!attr_kv && lds > 65536) {
        if (hipFuncSetAttribute((const void*)attn_bwd_dkv_dma_kernel<DT, KS, PC, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
          return TB_ELAUNCH;
        attr_kv = true;
      }
      hipLaunchKernelGGL((attn_bwd_dkv_dma_kernel<DT, KS, PC, NST>), dim3(d.Skv / 128, d.H, d.B), dim3(256), lds, s, d, (g_attn_dma >> 6) & 1);
    }
    TB_CHECK_LAUNCH();
    return TB_OK;
  }
  {
    size_t lds = (2 * RM<WD>::SIZE + 2 * TR<WD>::SIZE) * sizeof(f16) + 2 * KVT * sizeof(float);
    static bool attr_done = false;
    if (!attr_done && lds > 65536) {
      if (hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<DT, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return TB_ELAUNCH;
      attr_done = true;
    }
    const int nkb = (d.Skv + 127) / 128;
    int qsplit = 1;
    const int64_t nblk = (int64_t)nkb * d.H * d.B;
    const int64_t C = (int64_t)d.H * d.hd;
    const int64_t plane2 = 2 * (int64_t)d.B * d.Skv * C;
    if (d.ws && d.ws_floats >= 2 * plane2 && !d.causal && nblk < 512) {
      // slices until ~512 workgroups exist: every slice costs a plane of fp32 partials that the finalize launch reads back (memory-bound: 25 MB
      // and 7.7 us per 64x64-map layer at 16 slices) -- sustained A/B of the step: 1024 -> 512 workgroups -0.13 ms, 256 -0.11, 128 +0.11
      const int target = (g_attn_dma & 32768) ? 1024 : 512;
      qsplit = (int)((target + nblk - 1) / nblk);
      const int max_split = (d.Sq + KVT - 1) / KVT;
      if (qsplit > max_split) qsplit = max_split;
      if ((int64_t)qsplit * plane2 > d.ws_floats) qsplit = (int)(d.ws_floats / plane2);
    }
    int q_chunk = ((d.Sq + qsplit - 1) / qsplit + KVT - 1) / KVT * KVT;
    if (qsplit > 1) qsplit = (d.Sq + q_chunk - 1) / q_chunk;
    dim3 grid(nkb * qsplit, d.H, d.B);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<DT, KS>), grid, dim3(256), lds, s, d, qsplit, q_chunk, d.ws, (g_attn_dma >> 6) & 1);
    if (qsplit > 1) {
      const int64_t rows = (int64_t)d.B * d.Skv;
      hipLaunchKernelGGL(attn_dkv_finalize_kernel, dim3((unsigned)((rows * C + 255) / 256)), dim3(256), 0, s, d.ws, (f16*)d.dK, d.lddk,
                         (f16*)d.dV, d.lddv, rows, (int)C, qsplit);
    }
  }
  TB_CHECK_LAUNCH();
  return TB_OK;
}

int check_desc(const tb_attn_desc& d, bool bwd) {
  if (d.B <= 0 || d.H <= 0 || d.Sq <= 0 || d.Skv <= 0 || d.hd <= 0 || d.hd % 8 || d.hd > 160) return TB_EINVAL;
  if (!d.Q || !d.K || !d.V || !d.O) return TB_EINVAL;
  if (d.ldq % 8 || d.ldk % 8 || d.ldv % 8 || d.ldo % 4) return TB_EINVAL;
  if (bwd) {
    if (!d.dO || !d.dQ || !d.dK || !d.dV || !d.LSE || !d.Delta) return TB_EINVAL;
    if (d.lddo % 8 || d.lddq % 4 || d.lddk % 4 || d.lddv % 4 || d.ldo % 8) return TB_EINVAL;
  }
  return TB_OK;
}

#define DISPATCH_HD(FN, d, s)                              \
  do {                                                     \
    const int ks_ = ((d).hd + 15) / 16;                    \
    switch (ks_) {                                         \
      case 1: case 2: return FN<1, 2>(d, s);               \
      case 3: return FN<2, 3>(d, s);                       \
      case 4: return FN<2, 4>(d, s);                       \
      case 5: return FN<3, 5>(d, s);                       \
      case 6: return FN<3, 6>(d, s);                       \
      case 7: case 8: return FN<4, 8>(d, s);               \
      case 9: case 10: return FN<5, 10>(d, s);             \
      default: return TB_EINVAL;                           \
    }                                                      \
  } while (0)

}  // namespace

extern "C" int tb_attention_set_variant(int bits) {
  const int old = g_attn_dma;
  g_attn_dma = bits;
  return old;
}

extern "C" int tb_attention_fwd(const tb_attn_desc* dp, tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!dp) return TB_EINVAL;
  const tb_attn_desc d = *dp;
  int rc = check_desc(d, false);
  if (rc) return rc;
  DISPATCH_HD(launch_fwd, d, (hipStream_t)stream);
}

extern "C" int tb_attention_bwd(const tb_attn_desc* dp, tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!dp) return TB_EINVAL;
  const tb_attn_desc d = *dp;
  int rc = check_desc(d, true);
  if (rc) return rc;
  // short sequences with hd = 64 (the CLIP text encoder's 77 x 77 self-attention): dQ, dK and dV in ONE launch (attention_small.hip)
  if (!(g_attn_dma & 8192) && tb_attn_small_bwd_ok(d)) return tb_attn_small_bwd(d, (hipStream_t)stream);
  DISPATCH_HD(launch_bwd, d, (hipStream_t)stream);
}
