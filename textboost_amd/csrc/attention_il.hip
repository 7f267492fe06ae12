// Software-pipelined ("interleaved") flash-attention kernels for the SD1.x 64x64-map self-attention shape (hd = 40, S = 4096, non-causal):
// the shape that dominates the TextBoost step's attention time (train_textboost.py:1063-1067 -> diffusers Attention + AttnProcessor2_0 ->
// F.scaled_dot_product_attention, and its autograd at :1108).
//
// Same arithmetic as attn_fwd_dma_kernel / attn_bwd_*_dma_kernel of attention.hip (swapped S^T = K Q^T so softmax statistics are lane-local,
// -m folded into the head-dim padding column, the row sum as an all-ones column of V, lazy re-base, LDS-DMA staged row-major tiles, transposing
// LDS reads) -- what changes is the INSTRUCTION STREAM.  Measured on MI355X (scratch/coissue2.*): a wave that issues an MFMA plus up to ~5
// other instructions per 32-cycle matrix slot hides them completely, and two such waves on one SIMD keep the matrix pipe 82-90 % busy
// (6 / 4 fillers per MFMA); the phase-structured kernels (all MFMAs, then all softmax VALU, behind one another inside a wave) sit at 42-47 %.
// So every wave owns TWO 32-query groups and runs them half an iteration apart: while the matrix pipe works on group A (P.V of tile t-1, then
// Q.K^T of tile t: 14 MFMAs) the vector ALU does group B's softmax of tile t-1 (max, exp2, fp16 pack: ~70 instructions, ~5 per MFMA), and
// vice versa.  __builtin_amdgcn_sched_barrier(0) after every [MFMA + its slice of the other group's softmax + the LDS reads two MFMAs ahead]
// pins the interleave; consecutive MFMAs alternate accumulators so no dependent pair is back to back.
#include "common.h"
#include "../../include/textboost_hip.h"
#include "attn_il.h"

namespace {

constexpr int KVT = 64;
constexpr float LOG2E = 1.4426950408889634f;
#ifndef TB_IL_REBASE
#define TB_IL_REBASE 8.f
#endif
__device__ __attribute__((aligned(16))) const f16 g_zero8_il[8] = {};
typedef const __attribute__((address_space(1))) f16x8* gvec8_t;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 h16x4;
typedef __attribute__((address_space(3))) h16x4* lds_h4p;
typedef __attribute__((address_space(3))) const f16x8* lds_f8p;
typedef __attribute__((ext_vector_type(2))) unsigned long long u64x2;

#define SB() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ f16x8 join8(h16x4 a, h16x4 b) {
  u64x2 t;
  t[0] = __builtin_bit_cast(unsigned long long, a);
  t[1] = __builtin_bit_cast(unsigned long long, b);
  return __builtin_bit_cast(f16x8, t);
}
// Transposing LDS read as inline asm: the builtin (__builtin_amdgcn_ds_read_tr16_b64_v4f16) makes hipcc drain vmcnt(0) -- the LDS-DMA prefetch --
// in front of it (it cannot prove the read does not alias the DMA's LDS destination); asm reads are invisible to that pass.  They are also
// invisible to its lgkmcnt bookkeeping: frag_wait<N>() below waits by hand, counting ONLY the asm reads issued after the wanted one (LDS
// operations return in order, so compiler-issued reads in between can only make the wait stricter, never too weak).
template <int OFF>
__device__ __forceinline__ h16x4 tr_read(uint32_t addr) {
  h16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int N>
__device__ __forceinline__ void frag_wait(f16x8& a) {  // a's reads have landed once at most N later asm reads are outstanding
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N));
}
__device__ __forceinline__ f16x8 rm_read(uint32_t addr) { return *(lds_f8p)(uintptr_t)addr; }
__device__ __forceinline__ float other_half_max(float v) {  // max over the two half-waves (lanes l and l ^ 32), VALU only
  // v_permlane32_swap a, b: lanes 32-63 of a <-> lanes 0-31 of b; with a = b = v every lane then holds its own value in one register and its
  // partner's in the other.  Inline asm: hipcc (ROCm 7.2) folds fmaxf(r[0], r[1]) of __builtin_amdgcn_permlane32_swap(v, v) to r[0] alone --
  // the maximum then only covers half of a query's keys and the re-base test misses the other half (found by the spiked-key test).
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}
__device__ __forceinline__ void wait_vmcnt(int n) {  // n wave-uniform
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
  }
}

struct Blk {
  int x, h, b;
};
__device__ __forceinline__ Blk block_of(int remap) {  // XCD-aware order: all x blocks of one (batch, head) on one L2 (attention.hip attn_block)
  Blk r;
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

// ------------------------------------------------------------------------------------------------ forward
// hd = 40: KS = 3 k-steps of 16 over the padded 48-wide rows, DT = 2 head-dim blocks of 32, rows of PC = 6 16-byte chunks in LDS (5 data
// chunks + the constant pad chunk [1, 0, .., 0]: the all-ones column that carries -m into Q.K^T and the row sum out of P.V).
constexpr int F_KS = 3, F_DT = 2, F_PC = 6, F_PCB = F_PC * 16, F_TILE_B = KVT * F_PCB, F_STAGE_B = 2 * F_TILE_B + 64, F_NST = 4;

struct FwdGroup {        // one 32-query group of a wave
  f32x16 o[F_DT];        // O^T accumulators (row hd = the row sum)
  f32x16 s[2];           // raw scores of the current tile (two 32-key halves), exp2 argument
  f16x8 pk[2][2];        // P^T of the previous tile as B-operand fragments [32-key half][16-key quarter]
  f16x8 qf[F_KS];        // lane-owned Q row, pre-scaled by scale * log2(e); slot hd carries -m
  float m;               // running reference maximum (log2 domain, fp16-representable)
};

// The softmax of one group's score tile, cut into 14 slices that ride behind the 14 MFMAs of the other group's phase.
//   slices 0..2 : the running-excess maximum (14 max3 + tail)
//   slice  3    : cross-half max, the (rare, wave-uniform) re-base branch
//   slices 3..13: 16 units of [2 exp2 + 1 fp16 pack]
struct SmTmp {
  float mx0, mx1;
};
template <int I>
__device__ __forceinline__ void sm_unit(FwdGroup& g) {  // unit I of 16: registers 2 * (I & 7), +1 of half I >> 3
  constexpr int kt = I >> 3, r = 2 * (I & 7);
  const float p0 = fast_exp2(g.s[kt][r]), p1 = fast_exp2(g.s[kt][r + 1]);
  g.pk[kt][r >> 3][r & 7] = (f16)p0;
  g.pk[kt][r >> 3][(r & 7) + 1] = (f16)p1;
}
template <int SLOT>
__device__ __forceinline__ void sm_slice(FwdGroup& g, SmTmp& t, bool first, bool fold_lane) {
  if constexpr (SLOT == 0) {
    t.mx0 = max3f(g.s[0][0], g.s[0][1], g.s[0][2]);
    t.mx1 = max3f(g.s[1][0], g.s[1][1], g.s[1][2]);
    t.mx0 = max3f(t.mx0, g.s[0][3], g.s[0][4]);
    t.mx1 = max3f(t.mx1, g.s[1][3], g.s[1][4]);
    t.mx0 = max3f(t.mx0, g.s[0][5], g.s[0][6]);
    t.mx1 = max3f(t.mx1, g.s[1][5], g.s[1][6]);
  } else if constexpr (SLOT == 1) {
    t.mx0 = max3f(t.mx0, g.s[0][7], g.s[0][8]);
    t.mx1 = max3f(t.mx1, g.s[1][7], g.s[1][8]);
    t.mx0 = max3f(t.mx0, g.s[0][9], g.s[0][10]);
    t.mx1 = max3f(t.mx1, g.s[1][9], g.s[1][10]);
    t.mx0 = max3f(t.mx0, g.s[0][11], g.s[0][12]);
    t.mx1 = max3f(t.mx1, g.s[1][11], g.s[1][12]);
  } else if constexpr (SLOT == 2) {
    t.mx0 = max3f(t.mx0, g.s[0][13], g.s[0][14]);
    t.mx1 = max3f(t.mx1, g.s[1][13], g.s[1][14]);
    t.mx0 = max3f(t.mx0, t.mx1, fmaxf(g.s[0][15], g.s[1][15]));
    t.mx0 = other_half_max(t.mx0);
  } else if constexpr (SLOT == 3) {
    // Lazy re-base (attention.hip attn_fwd_kernel): m only has to stay within 2^REBASE of the true row maximum; it stays fp16-representable so
    // that the Q slot subtracts exactly the m that l and the stored LSE use.
    const float mx = t.mx0;
    if (first || __any(mx > TB_IL_REBASE)) {
      float d = first ? mx : fmaxf(mx, 0.f);
      d = (float)(f16)(g.m + d) - g.m;
      if (!first) {
        const float alpha = fast_exp2(-d);
#pragma unroll
        for (int dd = 0; dd < F_DT; ++dd)
#pragma unroll
          for (int r = 0; r < 16; ++r) g.o[dd][r] *= alpha;
      }
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) g.s[kt][r] -= d;
      g.m += d;
      if (fold_lane) g.qf[2][0] = (f16)(-g.m);
    }
    sm_unit<0>(g);
  } else if constexpr (SLOT == 4) {
    sm_unit<1>(g);
  } else if constexpr (SLOT == 5) {
    sm_unit<2>(g);
    sm_unit<3>(g);
  } else if constexpr (SLOT == 6) {
    sm_unit<4>(g);
  } else if constexpr (SLOT == 7) {
    sm_unit<5>(g);
    sm_unit<6>(g);
  } else if constexpr (SLOT == 8) {
    sm_unit<7>(g);
  } else if constexpr (SLOT == 9) {
    sm_unit<8>(g);
    sm_unit<9>(g);
  } else if constexpr (SLOT == 10) {
    sm_unit<10>(g);
  } else if constexpr (SLOT == 11) {
    sm_unit<11>(g);
    sm_unit<12>(g);
  } else if constexpr (SLOT == 12) {
    sm_unit<13>(g);
  } else {
    sm_unit<14>(g);
    sm_unit<15>(g);
  }
}

// operand fetch of MFMA n of a phase (n = 0..7: P.V, A = V^T fragment by two transposing reads; n = 8..13: Q.K^T, A = K rows)
//   P.V   n -> (kt, jj, d) = (n >> 2, (n >> 1) & 1, n & 1)           (d alternates: consecutive MFMAs hit different accumulators)
//   Q.K^T n -> (j, kt)     = ((n - 8) >> 1, (n - 8) & 1)
template <int N>
__device__ __forceinline__ f16x8 fetch_a(uint32_t vaddr, uint32_t kaddr) {
  if constexpr (N < 8) {
    constexpr int kt = N >> 2, jj = (N >> 1) & 1, d = N & 1;
    constexpr int off = (kt * 32 + 16 * jj) * F_PCB + d * 64;
    return join8(tr_read<off>(vaddr), tr_read<off + 8 * F_PCB>(vaddr));
  } else {
    constexpr int j = (N - 8) >> 1, kt = (N - 8) & 1;
    return rm_read(kaddr + kt * 32 * F_PCB + j * 32);
  }
}
template <int N>
__device__ __forceinline__ void do_mfma(FwdGroup& g, const f16x8& a) {
  if constexpr (N < 8) {
    constexpr int kt = N >> 2, jj = (N >> 1) & 1, d = N & 1;
    g.o[d] = TB_MFMA_32x32x16(a, g.pk[kt][jj], g.o[d]);
  } else {
    constexpr int j = (N - 8) >> 1, kt = (N - 8) & 1;
    if constexpr (j == 0) {
      const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      g.s[kt] = TB_MFMA_32x32x16(a, g.qf[0], z);
    } else {
      g.s[kt] = TB_MFMA_32x32x16(a, g.qf[j], g.s[kt]);
    }
  }
}

// One phase: group G's matrix work (PV of the previous tile from the V tile at `vaddr`, QK of the current one from the K tile at `kaddr`)
// with group O's softmax interleaved.  LO..HI = the MFMA range that exists (prologue: 8..13, steady state: 0..13, epilogue: 0..7).
// KEEP: the first phase of an iteration reads the 14 operand fragments from LDS (two MFMAs ahead of their use) and keeps them; the second
// phase multiplies the SAME K / V^T fragments with the other group's operands, so it issues no LDS reads at all (half the LDS traffic and
// 22 fewer instructions per phase).
struct Frags {
  f16x8 f[14];
};
template <int N, int LO, int HI, bool SM, bool KEEP>
__device__ __forceinline__ void phase_step(FwdGroup& G, FwdGroup& O, SmTmp& t, Frags& F, uint32_t vaddr, uint32_t kaddr, bool first,
                                           bool fold_lane) {
  if constexpr (N <= HI) {
    if constexpr (KEEP) {
      if constexpr (N + 2 <= HI) F.f[N + 2] = fetch_a<N + 2>(vaddr, kaddr);  // operands two MFMAs ahead
      if constexpr (N < 8) {  // a transposing-read fragment: asm reads issued after it = those of fragments N+1, N+2 (two each while they are P.V ones)
        constexpr int later = ((N + 1 <= HI && N + 1 < 8) ? 2 : 0) + ((N + 2 <= HI && N + 2 < 8) ? 2 : 0);
        frag_wait<later>(F.f[N]);
      }
    }
    do_mfma<N>(G, F.f[N]);
    if constexpr (SM) sm_slice<N>(O, t, first, fold_lane);
    SB();
    phase_step<N + 1, LO, HI, SM, KEEP>(G, O, t, F, vaddr, kaddr, first, fold_lane);
  }
}
template <int LO, int HI, bool SM, bool KEEP>
__device__ __forceinline__ void phase(FwdGroup& G, FwdGroup& O, Frags& F, uint32_t vaddr, uint32_t kaddr, bool first, bool fold_lane) {
  SmTmp t;
  if constexpr (KEEP) {
    F.f[LO] = fetch_a<LO>(vaddr, kaddr);
    F.f[LO + 1] = fetch_a<LO + 1>(vaddr, kaddr);
    SB();
  }
  if constexpr (SM && LO > 0) {  // prologue / short phases: the slices that have no MFMA to ride behind run first
    sm_slice<0>(O, t, first, fold_lane);
    sm_slice<1>(O, t, first, fold_lane);
    sm_slice<2>(O, t, first, fold_lane);
    sm_slice<3>(O, t, first, fold_lane);
    sm_slice<4>(O, t, first, fold_lane);
    sm_slice<5>(O, t, first, fold_lane);
    sm_slice<6>(O, t, first, fold_lane);
    sm_slice<7>(O, t, first, fold_lane);
    SB();
  }
  phase_step<LO, LO, HI, SM, KEEP>(G, O, t, F, vaddr, kaddr, first, fold_lane);
  if constexpr (SM && HI < 13) {
    sm_slice<8>(O, t, first, fold_lane);
    sm_slice<9>(O, t, first, fold_lane);
    sm_slice<10>(O, t, first, fold_lane);
    sm_slice<11>(O, t, first, fold_lane);
    sm_slice<12>(O, t, first, fold_lane);
    sm_slice<13>(O, t, first, fold_lane);
    SB();
  }
}

__global__ __launch_bounds__(256, 2) void attn_fwd_il_kernel(const tb_attn_desc p, int remap) {
  constexpr int NI = 2 * F_PC, WI = NI / 4;  // 12 one-KB load instructions per stage, 3 per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const Blk blk = block_of(remap & 1);
  const int b = blk.b, h = blk.h, hd = p.hd, Skv = p.Skv;
  const int qblk = blk.x * 256;
  const int64_t ldk = p.ldk, ldv = p.ldv;
  const f16* Qg = (const f16*)p.Q + (int64_t)b * p.Sq * p.ldq + h * hd;
  const char* Kg = (const char*)((const f16*)p.K + (int64_t)b * Skv * ldk + h * hd);
  const char* Vg = (const char*)((const f16*)p.V + (int64_t)b * Skv * ldv + h * hd);
  FwdGroup A, Bg;
  Frags F;
  const float c = p.scale * LOG2E;
  auto load_q = [&](FwdGroup& g, int q) {
#pragma unroll
    for (int j = 0; j < F_KS; ++j) {
      const int col = 16 * j + 8 * hi;
      g.qf[j] = *(col < hd ? (gvec8_t)(Qg + (int64_t)q * p.ldq + col) : (gvec8_t)g_zero8_il);
#pragma unroll
      for (int e = 0; e < 8; ++e) g.qf[j][e] = (f16)((float)g.qf[j][e] * c);
    }
#pragma unroll
    for (int d = 0; d < F_DT; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) g.o[d][r] = 0.f;
    g.m = 0.f;
  };
  load_q(A, qblk + wave * 64 + l31);
  load_q(Bg, qblk + wave * 64 + 32 + l31);
  // ---- this lane's part of a stage's loads: instruction t = wave + 4 i covers flat chunks t' * 64 + lane of tensor t / PC
  uint32_t g_off[WI];
  bool g_on[WI];
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    const int t = wave + 4 * i;
    const int tensor = t >= F_PC ? 1 : 0;
    const int f = (t - tensor * F_PC) * 64 + lane;
    const int row = f / F_PC, cc = f - row * F_PC;
    g_on[i] = cc < F_PC - 1;  // the pad chunk keeps its constant
    g_off[i] = (uint32_t)((int64_t)row * (tensor ? ldv : ldk) * 2 + cc * 16);
  }
  auto stage_loads = [&](int tile, int slot) {
    unsigned char* dst = smem_raw + slot * F_STAGE_B;
    const char* kb = Kg + (int64_t)tile * KVT * ldk * 2;
    const char* vb = Vg + (int64_t)tile * KVT * ldv * 2;
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int t = wave + 4 * i;
      const char* src = (t >= F_PC ? vb : kb) + g_off[i];
      if (g_on[i]) __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + t * 1024), 16, 0, 0);
    }
  };
  for (int u = threadIdx.x; u < 2 * KVT * F_NST; u += 256) {  // pad chunks [1, 0, .., 0] of every row of every stage (K and V)
    const int st = u / (2 * KVT), r = u - st * 2 * KVT;
    const f16x8 one = {(f16)1.f, 0, 0, 0, 0, 0, 0, 0};
    *(f16x8*)(smem_raw + st * F_STAGE_B + (r >= KVT ? F_TILE_B : 0) + (r & (KVT - 1)) * F_PCB + (F_PC - 1) * 16) = one;
  }
  const int ntiles = Skv / KVT;  // >= 2 (launcher)
  stage_loads(0, 0);
  stage_loads(1, 1);
  const bool fold_lane = hi == ((hd >> 3) & 1);  // column hd of the lane's Q row = element 0 of chunk hd / 16 in these lanes
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem_raw;
  const uint32_t k_lane = lds0 + l31 * F_PCB + hi * 16;
  const int g4 = lane >> 4, j16 = lane & 15;
  const uint32_t v_lane = lds0 + F_TILE_B + (4 * (g4 >> 1) + (j16 >> 2)) * F_PCB + ((g4 & 1) * 16 + 4 * (j16 & 3)) * 2;
  // tile t lives in slot t % 4; iteration t multiplies K(t) and V(t-1) and has the loads of tiles t+1 (in flight) and t+2 (issued now)
#ifdef TB_IL_PROF
  unsigned long long pf_[5] = {0, 0, 0, 0, 0}, pt_ = __builtin_amdgcn_s_memtime();
  const unsigned long long pf_t0 = pt_, pf_r0 = wall_clock64();
#define TB_PF(k) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); pf_[k] += n_ - pt_; pt_ = n_; }
#else
#define TB_PF(k)
#endif
  auto sync_and_prefetch = [&](int t) {
    TB_PF(4)
    wait_vmcnt(t + 1 < ntiles ? WI : 0);  // tile t landed (this wave's part); tile t+1 may stay in flight
    TB_PF(0)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    TB_PF(1)
    if (t + 2 < ntiles) stage_loads(t + 2, (t + 2) & 3);  // into the slot of tile t-2: every wave is past its last read of it
    TB_PF(2)
  };
  // ---- prologue: QK_A(0); QK_B(0) || SM_A(0); then tile 1 with SM_B(0) as a first tile
  sync_and_prefetch(0);
  {
    const uint32_t ka = k_lane;
    phase<8, 13, false, true>(A, Bg, F, 0, ka, false, fold_lane);
    phase<8, 13, true, false>(Bg, A, F, 0, ka, true, fold_lane);
  }
  for (int t = 1; t < ntiles; ++t) {
    sync_and_prefetch(t);
    const uint32_t ka = k_lane + (t & 3) * F_STAGE_B;
    const uint32_t va = v_lane + ((t - 1) & 3) * F_STAGE_B;
    phase<0, 13, true, true>(A, Bg, F, va, ka, t == 1, fold_lane);  // PV_A(t-1), QK_A(t) || SM_B(t-1)
    TB_PF(3)
    phase<0, 13, true, false>(Bg, A, F, va, ka, false, fold_lane);  // PV_B(t-1), QK_B(t) || SM_A(t): the same fragments
  }
#ifdef TB_IL_PROF
  if (p.Delta && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (threadIdx.x & 63) == 0)
    for (int k = 0; k < 5; ++k) p.Delta[wave * 5 + k] = (float)pf_[k] / (float)ntiles;
  if (p.Delta && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) {
    p.Delta[20] = (float)(__builtin_amdgcn_s_memtime() - pf_t0);  // shader-clock ticks of the whole block
    p.Delta[21] = (float)(wall_clock64() - pf_r0);                 // 100 MHz constant clock over the same span
  }
  if (p.Delta && threadIdx.x == 0) {  // per block: start / end on the constant clock, shader ticks, hardware id
    const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    unsigned hwid = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
    unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
    double* o = (double*)(p.Delta + 64) + lin * 4;
    o[0] = (double)pf_r0; o[1] = (double)wall_clock64(); o[2] = (double)(__builtin_amdgcn_s_memtime() - pf_t0); o[3] = (double)(hwid | ((unsigned long long)(xcc & 15) << 32));
  }
#endif
  {  // ---- epilogue: the last tile's P.V (its V tile landed with the last barrier; no other wave writes LDS any more)
    const uint32_t va = v_lane + ((ntiles - 1) & 3) * F_STAGE_B;
    phase<0, 7, true, true>(A, Bg, F, va, 0, ntiles == 1, fold_lane);  // PV_A(n-1) || SM_B(n-1)
    phase<0, 7, false, false>(Bg, A, F, va, 0, false, fold_lane);      // PV_B(n-1)
  }
  auto store = [&](FwdGroup& g, int q) {
    // the all-ones column hd of V made O^T row hd the row sum: register 4 k of the hi == 0 lane of block hd / 32
    float ls = 0.f;
#pragma unroll
    for (int d = 0; d < F_DT; ++d)
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4)
        if (d * 32 + k4 * 8 == hd) ls = g.o[d][4 * k4];
    const float lsum = __shfl(ls, l31, 64);
    const float inv = 1.f / lsum;
    f16* Og = (f16*)p.O + ((int64_t)b * p.Sq + q) * p.ldo + h * hd;
#pragma unroll
    for (int d = 0; d < F_DT; ++d)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int col = d * 32 + 8 * r4 + 4 * hi;
        if (col < hd) {
          f16x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (f16)(g.o[d][4 * r4 + e] * inv);
          *(f16x4*)(Og + col) = v;
        }
      }
    if (p.LSE && hi == 0) p.LSE[((int64_t)b * p.H + h) * p.Sq + q] = g.m * (1.f / LOG2E) + logf(lsum);
  };
  store(A, qblk + wave * 64 + l31);
  store(Bg, qblk + wave * 64 + 32 + l31);
}

// ------------------------------------------------------------------------------------------------ dK, dV
// attn_bwd_dkv_dma_kernel's arithmetic (attention.hip: a lane owns a key; streamed Q / dO tiles row-major by LDS-DMA, S^T and dP^T from
// b128 row reads, dV += P^T dO and dK += dS^T Q from the same rows through the transposing read; the per-query -lse log2(e) / -delta the dQ
// kernel published enter as accumulator inputs) as a THREE-STAGE pipeline over 32-query halves inside one wave:
//     phase k:   matrix pipe  = dV, dK of half k-1 (8 MFMAs)  +  S^T, dP^T of half k+1 (6 MFMAs)
//                vector ALU   = p = exp2(s), ds = p * dp, fp16 packing of half k (48 instructions, 3.4 per MFMA)
// so every MFMA is followed by a slice of the softmax work of the half in between; score registers and packed fragments are double-buffered
// by the half's parity.  One barrier per 64-query tile, placed where the pipeline first touches the next tile.
// Rows of hd / 8 = 5 chunks WITHOUT a pad chunk (round 4; 80-byte pitch: the b128 row reads of a 16-lane group land on 16 distinct bank slots, the
// 96-byte rows were a 2-way conflict).  The k-step that runs past a row's end multiplies the zeros of the lane-owned K / V fragments, so what it
// reads only has to be finite: the next row, the dO tile behind the Q tile, and behind the dO tile a zeroed 64-byte gap in front of the statistics
// rows (raw fp32 bit patterns: NOT safe to read as fp16).  TB_IL_DKV_PAD=1 builds the padded layout of round 3.
#ifndef TB_IL_DKV_PAD
#define TB_IL_DKV_PAD 0
#endif
constexpr int D_PC = TB_IL_DKV_PAD ? 6 : 5, D_PCB = D_PC * 16, D_TILE_B = KVT * D_PCB, D_GAP = 64, D_STAGE_B = 2 * D_TILE_B + D_GAP + 2 * KVT * 4 + 64,
              D_NST = 4;

struct DkvState {
  f32x16 dk[2], dv[2];
  f32x16 s[2], dp[2];        // [parity of the half]
  f16x8 pp[2][2], ds[2][2];  // [parity][16-query quarter]: P^T and dS^T as B-operand fragments
  f16x8 kf[3], vf[3];        // lane-owned K (pre-scaled by scale * log2 e) and V rows
};
template <int U, int PAR>
__device__ __forceinline__ void dkv_unit(DkvState& st) {  // registers 2U, 2U+1 of the half with parity PAR
  constexpr int r = 2 * U;
  const float p0 = fast_exp2(st.s[PAR][r]), p1 = fast_exp2(st.s[PAR][r + 1]);
  const float d0 = p0 * st.dp[PAR][r], d1 = p1 * st.dp[PAR][r + 1];
  st.pp[PAR][r >> 3][r & 7] = (f16)p0;
  st.pp[PAR][r >> 3][(r & 7) + 1] = (f16)p1;
  st.ds[PAR][r >> 3][r & 7] = (f16)d0;
  st.ds[PAR][r >> 3][(r & 7) + 1] = (f16)d1;
}
template <int SLOT, int PAR>
__device__ __forceinline__ void dkv_slice(DkvState& st) {  // 8 units over the 14 MFMA slots of a phase
  if constexpr (SLOT == 0) dkv_unit<0, PAR>(st);
  else if constexpr (SLOT == 2) dkv_unit<1, PAR>(st);
  else if constexpr (SLOT == 4) dkv_unit<2, PAR>(st);
  else if constexpr (SLOT == 6) dkv_unit<3, PAR>(st);
  else if constexpr (SLOT == 8) dkv_unit<4, PAR>(st);
  else if constexpr (SLOT == 10) dkv_unit<5, PAR>(st);
  else if constexpr (SLOT == 12) dkv_unit<6, PAR>(st);
  else if constexpr (SLOT == 13) dkv_unit<7, PAR>(st);
}
// operand fetch of MFMA N of a phase whose halves k-1 / k+1 are the 32-query half KT of their tiles:
//   N = 0..7  (half k-1): (jj, d, which) = (N >> 2, (N >> 1) & 1, N & 1); which 0: dO^T fragment (dV), 1: Q^T fragment (dK): transposing reads
//   N = 8..13 (half k+1): (j, which) = ((N - 8) >> 1, (N - 8) & 1);       which 0: Q rows (S^T), 1: dO rows (dP^T): b128 reads
template <int N, int KT>
__device__ __forceinline__ f16x8 dkv_fetch(uint32_t prev_tr, uint32_t next_rm) {
  if constexpr (N < 8) {
    constexpr int jj = N >> 2, d = (N >> 1) & 1, which = N & 1;
    constexpr int off = (which ? 0 : D_TILE_B) + (KT * 32 + 16 * jj) * D_PCB + d * 64;
    return join8(tr_read<off>(prev_tr), tr_read<off + 8 * D_PCB>(prev_tr));
  } else {
    constexpr int j = (N - 8) >> 1, which = (N - 8) & 1;
    return rm_read(next_rm + (which ? D_TILE_B : 0) + KT * 32 * D_PCB + j * 32);
  }
}
template <int N, int PAR>
__device__ __forceinline__ void dkv_mfma(DkvState& st, const f16x8& a, const f32x16& s_init, const f32x16& dp_init) {
  constexpr int Q = PAR ^ 1;  // halves k-1 and k+1 have the other parity
  if constexpr (N < 8) {
    constexpr int jj = N >> 2, d = (N >> 1) & 1, which = N & 1;
    if constexpr (which == 0) st.dv[d] = TB_MFMA_32x32x16(a, st.pp[Q][jj], st.dv[d]);
    else st.dk[d] = TB_MFMA_32x32x16(a, st.ds[Q][jj], st.dk[d]);
  } else {
    constexpr int j = (N - 8) >> 1, which = (N - 8) & 1;
    if constexpr (which == 0) st.s[Q] = TB_MFMA_32x32x16(a, st.kf[j], j == 0 ? s_init : st.s[Q]);
    else st.dp[Q] = TB_MFMA_32x32x16(a, st.vf[j], j == 0 ? dp_init : st.dp[Q]);
  }
}
template <int N, int LO, int HI, int PAR, bool SM>
__device__ __forceinline__ void dkv_step(DkvState& st, f16x8& a0, f16x8& a1, uint32_t prev_tr, uint32_t next_rm, const f32x16& s_init,
                                         const f32x16& dp_init) {
  if constexpr (N <= HI) {
    constexpr int KT = PAR ^ 1;
    f16x8 nxt;
    if constexpr (N + 2 <= HI) nxt = dkv_fetch<N + 2, KT>(prev_tr, next_rm);
    if constexpr (N < 8) {
      constexpr int later = ((N + 1 <= HI && N + 1 < 8) ? 2 : 0) + ((N + 2 <= HI && N + 2 < 8) ? 2 : 0);
      frag_wait<later>(a0);
    }
    dkv_mfma<N, PAR>(st, a0, s_init, dp_init);
    if constexpr (SM) dkv_slice<N, PAR>(st);
    SB();
    a0 = a1;
    if constexpr (N + 2 <= HI) a1 = nxt;
    dkv_step<N + 1, LO, HI, PAR, SM>(st, a0, a1, prev_tr, next_rm, s_init, dp_init);
  }
}
// PAR = parity of the half whose softmax runs (k & 1); PREV / NEXT: the matrix work of halves k-1 / k+1 exists
template <int PAR, bool PREV, bool NEXT, bool SM>
__device__ __forceinline__ void dkv_phase(DkvState& st, uint32_t prev_tr, uint32_t next_rm, uint32_t next_stat, int hi) {
  constexpr int LO = PREV ? 0 : 8, HI = NEXT ? 13 : 7, KT = PAR ^ 1;
  f32x16 s_init, dp_init;
  if constexpr (NEXT) {  // -lse log2(e) / -delta of the next half's queries: rows of register quad g are 8 g + 4 hi + {0..3}
    const uint32_t sa = next_stat + (KT * 32 + 4 * hi) * 4;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const f32x4 lq = *(__attribute__((address_space(3))) const f32x4*)(uintptr_t)(sa + q4 * 32);
      const f32x4 dq = *(__attribute__((address_space(3))) const f32x4*)(uintptr_t)(sa + KVT * 4 + q4 * 32);
#pragma unroll
      for (int e = 0; e < 4; ++e) s_init[4 * q4 + e] = lq[e], dp_init[4 * q4 + e] = dq[e];
    }
  }
  f16x8 a0 = dkv_fetch<LO, KT>(prev_tr, next_rm), a1 = dkv_fetch<LO + 1, KT>(prev_tr, next_rm);
  SB();
  if constexpr (SM && LO > 0) {  // no MFMAs 0..7 to ride behind
    dkv_unit<0, PAR>(st); dkv_unit<1, PAR>(st); dkv_unit<2, PAR>(st); dkv_unit<3, PAR>(st);
    SB();
  }
  dkv_step<LO, LO, HI, PAR, SM>(st, a0, a1, prev_tr, next_rm, s_init, dp_init);
  if constexpr (SM && HI < 13) {
    dkv_unit<4, PAR>(st); dkv_unit<5, PAR>(st); dkv_unit<6, PAR>(st); dkv_unit<7, PAR>(st);
    SB();
  }
}

__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_il_kernel(const tb_attn_desc p, int remap) {
  constexpr int PC = D_PC, NI = 2 * PC + 2, WI = (NI + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const Blk blk = block_of(remap);
  const int b = blk.b, h = blk.h, hd = p.hd;
  const int key = blk.x * 128 + wave * 32 + l31;
  const int64_t ldq = p.ldq, lddo = p.lddo;
  const char* Qg = (const char*)((const f16*)p.Q + (int64_t)b * p.Sq * ldq + h * hd);
  const char* dOg = (const char*)((const f16*)p.dO + (int64_t)b * p.Sq * lddo + h * hd);
  const f16* Kg = (const f16*)p.K + (int64_t)b * p.Skv * p.ldk + h * hd;
  const f16* Vg = (const f16*)p.V + (int64_t)b * p.Skv * p.ldv + h * hd;
  const int64_t BHS = (int64_t)p.B * p.H * p.Sq;
  const char* NLg = (const char*)(p.ws + ((int64_t)b * p.H + h) * p.Sq);        // -lse * log2(e), written by the dQ kernel
  const char* NDg = (const char*)(p.ws + BHS + ((int64_t)b * p.H + h) * p.Sq);  // -delta
  DkvState st;
  {
    const float c = p.scale * LOG2E;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int col = 16 * j + 8 * hi;
      const bool ok = key < p.Skv && col < hd;
      st.kf[j] = *(ok ? (gvec8_t)(Kg + (int64_t)key * p.ldk + col) : (gvec8_t)g_zero8_il);
      st.vf[j] = *(ok ? (gvec8_t)(Vg + (int64_t)key * p.ldv + col) : (gvec8_t)g_zero8_il);
#pragma unroll
      for (int e = 0; e < 8; ++e) st.kf[j][e] = (f16)((float)st.kf[j][e] * c);
    }
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) st.dk[d][r] = 0.f, st.dv[d][r] = 0.f;
  }
  // ---- this lane's part of a stage's loads: instruction t = wave + 4 i; t < PC: Q rows, t < 2 PC: dO rows, then the two stat rows
  uint32_t g_off[WI];
  bool g_on[WI];
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    const int t = wave + 4 * i;
    const int tensor = t >= PC ? 1 : 0;
    const int f = (t - tensor * PC) * 64 + lane;
    const int row = f / PC, cc = f - row * PC;
    g_on[i] = t < 2 * PC && (!TB_IL_DKV_PAD || cc < PC - 1);
    g_off[i] = (uint32_t)((int64_t)row * (tensor ? lddo : ldq) * 2 + cc * 16);
  }
  int n_issued = 0;
#pragma unroll
  for (int i = 0; i < WI; ++i) n_issued += (wave + 4 * i < NI) ? 1 : 0;
  auto stage_loads = [&](int tile, int slot) {
    unsigned char* dst = smem_raw + slot * D_STAGE_B;
    const char* qb = Qg + (int64_t)tile * KVT * ldq * 2;
    const char* ob = dOg + (int64_t)tile * KVT * lddo * 2;
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int t = wave + 4 * i;
      if (t < 2 * PC) {
        const char* src = (t >= PC ? ob : qb) + g_off[i];
        if (g_on[i]) __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + t * 1024), 16, 0, 0);
      } else if (t < NI) {  // 64 floats = one dword per lane
        const char* src = (t == 2 * PC ? NLg : NDg) + ((int64_t)tile * KVT + lane) * 4;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + 2 * D_TILE_B + D_GAP + (t - 2 * PC) * 256), 4, 0, 0);
      }
    }
  };
  if (TB_IL_DKV_PAD) {
    for (int u = threadIdx.x; u < 2 * KVT * D_NST; u += 256) {  // pad chunks of every row of every stage: zeros (finite)
      const int s_ = u / (2 * KVT), r = u - s_ * 2 * KVT;
      const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      *(f16x8*)(smem_raw + s_ * D_STAGE_B + (r >= KVT ? D_TILE_B : 0) + (r & (KVT - 1)) * D_PCB + (PC - 1) * 16) = z;
    }
  }
  if (threadIdx.x < D_NST * 4) {  // the gap between the dO tile and the statistics rows of every stage
    const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    *(f16x8*)(smem_raw + (threadIdx.x >> 2) * D_STAGE_B + 2 * D_TILE_B + (threadIdx.x & 3) * 16) = z;
  }
  const int n = p.Sq / KVT;  // >= 2 (launcher)
#pragma unroll
  for (int t = 0; t < D_NST - 1; ++t)
    if (t < n) stage_loads(t, t);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem_raw;
  const uint32_t rm_lane = lds0 + l31 * D_PCB + hi * 16;
  const int g4 = lane >> 4, j16 = lane & 15;
  const uint32_t tr_lane = lds0 + (4 * (g4 >> 1) + (j16 >> 2)) * D_PCB + ((g4 & 1) * 16 + 4 * (j16 & 3)) * 2;
  const uint32_t stat0 = lds0 + 2 * D_TILE_B + D_GAP;
  // tile t in slot t & 3.  `sync(t)`: tile t has landed for every wave, every wave is done with tile t - 2 (-> the loads of tile t + 2 go there)
  auto sync = [&](int t) {
    int later = n - 1 - t;
    later = later > 1 ? 1 : later;      // loads issued after tile t's that may stay in flight: tile t + 1's
    if (t == 0) later = n - 1 > 2 ? 2 : n - 1;  // the prologue issued tiles 0, 1, 2
    switch (later * n_issued) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (t >= 1 && t + 2 < n) stage_loads(t + 2, (t + 2) & 3);
  };
#define TR_OF(t) (tr_lane + ((t) & 3) * D_STAGE_B)
#define RM_OF(t) (rm_lane + ((t) & 3) * D_STAGE_B)
#define ST_OF(t) (stat0 + ((t) & 3) * D_STAGE_B)
#ifdef TB_IL_PROF
  const unsigned long long pfd_t0 = __builtin_amdgcn_s_memtime(), pfd_r0 = wall_clock64();
#endif
  sync(0);
  dkv_phase<1, false, true, false>(st, 0, RM_OF(0), ST_OF(0), hi);          // k = -1: S, dP of half 0
  dkv_phase<0, false, true, true>(st, 0, RM_OF(0), ST_OF(0), hi);           // k = 0 : softmax(0); S, dP of half 1
  for (int t = 0; t + 1 < n; ++t) {
    sync(t + 1);
    dkv_phase<1, true, true, true>(st, TR_OF(t), RM_OF(t + 1), ST_OF(t + 1), hi);       // k = 2t+1: dV dK(2t) | softmax(2t+1) | S dP(2t+2)
    dkv_phase<0, true, true, true>(st, TR_OF(t), RM_OF(t + 1), ST_OF(t + 1), hi);       // k = 2t+2: dV dK(2t+1) | softmax(2t+2) | S dP(2t+3)
  }
  dkv_phase<1, true, false, true>(st, TR_OF(n - 1), 0, 0, hi);              // k = 2n-1: dV dK(2n-2) | softmax(2n-1)
  dkv_phase<0, true, false, false>(st, TR_OF(n - 1), 0, 0, hi);             // k = 2n  : dV dK(2n-1)
#ifdef TB_IL_PROF
  if (p.Delta && threadIdx.x == 0) {  // per block: shader-clock ticks and 100 MHz ticks of the main loop (the dQ kernel is done with Delta)
    const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    p.Delta[2 * lin] = (float)(__builtin_amdgcn_s_memtime() - pfd_t0);
    p.Delta[2 * lin + 1] = (float)(wall_clock64() - pfd_r0);
  }
#endif
#undef TR_OF
#undef RM_OF
#undef ST_OF
  if (key < p.Skv) {
    f16* dKg = (f16*)p.dK + ((int64_t)b * p.Skv + key) * p.lddk + h * hd;
    f16* dVg = (f16*)p.dV + ((int64_t)b * p.Skv + key) * p.lddv + h * hd;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int col = d * 32 + 8 * r4 + 4 * hi;
        if (col < hd) {
          f16x4 a, bb;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            a[e] = (f16)(st.dk[d][4 * r4 + e] * p.scale);
            bb[e] = (f16)st.dv[d][4 * r4 + e];
          }
          *(f16x4*)(dKg + col) = a;
          *(f16x4*)(dVg + col) = bb;
        }
      }
  }
}

// ------------------------------------------------------------------------------------------------ dQ
// attn_bwd_dq_dma_kernel's arithmetic (a lane owns a query; K / V tiles row-major by LDS-DMA; S^T = K Q^T and dP^T = V dO^T from b128 row
// reads with -lse log2(e) / -delta as accumulator inputs; dQ^T += K^T dS^T through the transposing read; also delta = rowsum(dO * O) and the
// statistics the dK/dV kernel reads) as the same three-stage pipeline over 32-key halves:
//     phase k:   matrix pipe = dQ of half k-1 (4 MFMAs) + S^T, dP^T of half k+1 (6 MFMAs);   vector ALU = dS = exp2(s) * dp, packing of half k
constexpr int Q_PCB = 96, Q_TILE_B = KVT * Q_PCB, Q_STAGE_B = 2 * Q_TILE_B + 64;  // (the opt-in dQ kernel keeps the padded 96-byte rows)
struct DqState {
  f32x16 dq[2];
  f32x16 s[2], dp[2];   // [parity of the half]
  f16x8 ds[2][2];       // [parity][16-key quarter]
  f16x8 qf[3], dof[3];  // lane-owned Q (pre-scaled) and dO rows
  f32x16 neg_lse, neg_delta;
};
template <int U, int PAR>
__device__ __forceinline__ void dq_unit(DqState& st) {
  constexpr int r = 2 * U;
  const float d0 = fast_exp2(st.s[PAR][r]) * st.dp[PAR][r], d1 = fast_exp2(st.s[PAR][r + 1]) * st.dp[PAR][r + 1];
  st.ds[PAR][r >> 3][r & 7] = (f16)d0;
  st.ds[PAR][r >> 3][(r & 7) + 1] = (f16)d1;
}
template <int SLOT, int PAR>
__device__ __forceinline__ void dq_slice(DqState& st) {  // 8 units over the 10 MFMA slots of a phase
  if constexpr (SLOT == 0) dq_unit<0, PAR>(st);
  else if constexpr (SLOT == 1) dq_unit<1, PAR>(st);
  else if constexpr (SLOT == 2) dq_unit<2, PAR>(st);
  else if constexpr (SLOT == 3) dq_unit<3, PAR>(st);
  else if constexpr (SLOT == 5) dq_unit<4, PAR>(st);
  else if constexpr (SLOT == 6) dq_unit<5, PAR>(st);
  else if constexpr (SLOT == 8) dq_unit<6, PAR>(st);
  else if constexpr (SLOT == 9) dq_unit<7, PAR>(st);
}
//   N = 0..3 (half k-1): (jj, d) = (N >> 1, N & 1): K^T fragment, transposing reads;   N = 4..9 (half k+1): (j, which) = ((N-4) >> 1, (N-4) & 1):
//   which 0: K rows (S^T), 1: V rows (dP^T)
template <int N, int KT>
__device__ __forceinline__ f16x8 dq_fetch(uint32_t prev_tr, uint32_t next_rm) {
  if constexpr (N < 4) {
    constexpr int jj = N >> 1, d = N & 1;
    constexpr int off = (KT * 32 + 16 * jj) * Q_PCB + d * 64;
    return join8(tr_read<off>(prev_tr), tr_read<off + 8 * Q_PCB>(prev_tr));
  } else {
    constexpr int j = (N - 4) >> 1, which = (N - 4) & 1;
    return rm_read(next_rm + (which ? Q_TILE_B : 0) + KT * 32 * Q_PCB + j * 32);
  }
}
template <int N, int PAR>
__device__ __forceinline__ void dq_mfma(DqState& st, const f16x8& a) {
  constexpr int Q = PAR ^ 1;
  if constexpr (N < 4) {
    constexpr int jj = N >> 1, d = N & 1;
    st.dq[d] = TB_MFMA_32x32x16(a, st.ds[Q][jj], st.dq[d]);
  } else {
    constexpr int j = (N - 4) >> 1, which = (N - 4) & 1;
    if constexpr (which == 0) st.s[Q] = TB_MFMA_32x32x16(a, st.qf[j], j == 0 ? st.neg_lse : st.s[Q]);
    else st.dp[Q] = TB_MFMA_32x32x16(a, st.dof[j], j == 0 ? st.neg_delta : st.dp[Q]);
  }
}
template <int N, int LO, int HI, int PAR, bool SM>
__device__ __forceinline__ void dq_step(DqState& st, f16x8& a0, f16x8& a1, uint32_t prev_tr, uint32_t next_rm) {
  if constexpr (N <= HI) {
    constexpr int KT = PAR ^ 1;
    f16x8 nxt;
    if constexpr (N + 2 <= HI) nxt = dq_fetch<N + 2, KT>(prev_tr, next_rm);
    if constexpr (N < 4) {
      constexpr int later = ((N + 1 <= HI && N + 1 < 4) ? 2 : 0) + ((N + 2 <= HI && N + 2 < 4) ? 2 : 0);
      frag_wait<later>(a0);
    }
    dq_mfma<N, PAR>(st, a0);
    if constexpr (SM) dq_slice<N, PAR>(st);
    SB();
    a0 = a1;
    if constexpr (N + 2 <= HI) a1 = nxt;
    dq_step<N + 1, LO, HI, PAR, SM>(st, a0, a1, prev_tr, next_rm);
  }
}
template <int PAR, bool PREV, bool NEXT, bool SM>
__device__ __forceinline__ void dq_phase(DqState& st, uint32_t prev_tr, uint32_t next_rm) {
  constexpr int LO = PREV ? 0 : 4, HI = NEXT ? 9 : 3, KT = PAR ^ 1;
  f16x8 a0 = dq_fetch<LO, KT>(prev_tr, next_rm), a1 = dq_fetch<LO + 1, KT>(prev_tr, next_rm);
  SB();
  if constexpr (SM && LO > 0) {
    dq_unit<0, PAR>(st); dq_unit<1, PAR>(st); dq_unit<2, PAR>(st); dq_unit<3, PAR>(st);
    SB();
  }
  dq_step<LO, LO, HI, PAR, SM>(st, a0, a1, prev_tr, next_rm);
  if constexpr (SM && HI < 9) {
    dq_unit<4, PAR>(st); dq_unit<5, PAR>(st); dq_unit<6, PAR>(st); dq_unit<7, PAR>(st);
    SB();
  }
}

__global__ __launch_bounds__(256, 2) void attn_bwd_dq_il_kernel(const tb_attn_desc p, int remap, int publish) {
  constexpr int PC = 6, NI = 2 * PC, WI = NI / 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const Blk blk = block_of(remap);
  const int b = blk.b, h = blk.h, hd = p.hd, Skv = p.Skv;
  const int q = blk.x * 128 + wave * 32 + l31;
  const int64_t ldk = p.ldk, ldv = p.ldv;
  const f16* Qg = (const f16*)p.Q + (int64_t)b * p.Sq * p.ldq + h * hd;
  const f16* dOg = (const f16*)p.dO + (int64_t)b * p.Sq * p.lddo + h * hd;
  const char* Kg = (const char*)((const f16*)p.K + (int64_t)b * Skv * ldk + h * hd);
  const char* Vg = (const char*)((const f16*)p.V + (int64_t)b * Skv * ldv + h * hd);
  DqState st;
  {
    const float c = p.scale * LOG2E;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int col = 16 * j + 8 * hi;
      const bool ok = col < hd;  // (Sq % 128 == 0: every query row exists)
      st.qf[j] = *(ok ? (gvec8_t)(Qg + (int64_t)q * p.ldq + col) : (gvec8_t)g_zero8_il);
      st.dof[j] = *(ok ? (gvec8_t)(dOg + (int64_t)q * p.lddo + col) : (gvec8_t)g_zero8_il);
    }
    const int64_t sidx = ((int64_t)b * p.H + h) * p.Sq + q;
    const float lse2 = p.LSE[sidx] * LOG2E;
    const f16* Og = (const f16*)p.O + (int64_t)b * p.Sq * p.ldo + h * hd;
    float a = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int col = 16 * j + 8 * hi;
      if (col < hd) {
        const f16x8 ov = *(const f16x8*)(Og + (int64_t)q * p.ldo + col);
#pragma unroll
        for (int e = 0; e < 8; ++e) a += (float)ov[e] * (float)st.dof[j][e];
      }
    }
    const float delta = a + __shfl_xor(a, 32, 64);
    if (hi == 0) {
      p.Delta[sidx] = delta;
      if (publish) {
        p.ws[sidx] = -lse2;
        p.ws[(int64_t)p.B * p.H * p.Sq + sidx] = -delta;
      }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) st.qf[j][e] = (f16)((float)st.qf[j][e] * c);
#pragma unroll
    for (int r = 0; r < 16; ++r) st.neg_lse[r] = -lse2, st.neg_delta[r] = -delta, st.dq[0][r] = 0.f, st.dq[1][r] = 0.f;
  }
  uint32_t g_off[WI];
  bool g_on[WI];
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    const int t = wave + 4 * i;
    const int tensor = t >= PC ? 1 : 0;
    const int f = (t - tensor * PC) * 64 + lane;
    const int row = f / PC, cc = f - row * PC;
    g_on[i] = cc < PC - 1;
    g_off[i] = (uint32_t)((int64_t)row * (tensor ? ldv : ldk) * 2 + cc * 16);
  }
  auto stage_loads = [&](int tile, int slot) {
    unsigned char* dst = smem_raw + slot * Q_STAGE_B;
    const char* kb = Kg + (int64_t)tile * KVT * ldk * 2;
    const char* vb = Vg + (int64_t)tile * KVT * ldv * 2;
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int t = wave + 4 * i;
      const char* src = (t >= PC ? vb : kb) + g_off[i];
      if (g_on[i]) __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + t * 1024), 16, 0, 0);
    }
  };
  for (int u = threadIdx.x; u < 2 * KVT * D_NST; u += 256) {  // pad chunks: zeros (finite)
    const int s_ = u / (2 * KVT), r = u - s_ * 2 * KVT;
    const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    *(f16x8*)(smem_raw + s_ * Q_STAGE_B + (r >= KVT ? Q_TILE_B : 0) + (r & (KVT - 1)) * Q_PCB + (PC - 1) * 16) = z;
  }
  const int n = Skv / KVT;  // >= 2 (launcher)
#pragma unroll
  for (int t = 0; t < D_NST - 1; ++t)
    if (t < n) stage_loads(t, t);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem_raw;
  const uint32_t rm_lane = lds0 + l31 * Q_PCB + hi * 16;
  const int g4 = lane >> 4, j16 = lane & 15;
  const uint32_t tr_lane = lds0 + (4 * (g4 >> 1) + (j16 >> 2)) * Q_PCB + ((g4 & 1) * 16 + 4 * (j16 & 3)) * 2;
  auto sync = [&](int t) {
    int later = n - 1 - t;
    later = later > 1 ? 1 : later;
    if (t == 0) later = n - 1 > 2 ? 2 : n - 1;
    switch (later) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (t >= 1 && t + 2 < n) stage_loads(t + 2, (t + 2) & 3);
  };
#define TR_OF(t) (tr_lane + ((t) & 3) * Q_STAGE_B)
#define RM_OF(t) (rm_lane + ((t) & 3) * Q_STAGE_B)
  sync(0);
  dq_phase<1, false, true, false>(st, 0, RM_OF(0));
  dq_phase<0, false, true, true>(st, 0, RM_OF(0));
  for (int t = 0; t + 1 < n; ++t) {
    sync(t + 1);
    dq_phase<1, true, true, true>(st, TR_OF(t), RM_OF(t + 1));
    dq_phase<0, true, true, true>(st, TR_OF(t), RM_OF(t + 1));
  }
  dq_phase<1, true, false, true>(st, TR_OF(n - 1), 0);
  dq_phase<0, true, false, false>(st, TR_OF(n - 1), 0);
#undef TR_OF
#undef RM_OF
  f16* dQg = (f16*)p.dQ + ((int64_t)b * p.Sq + q) * p.lddq + h * hd;
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int col = d * 32 + 8 * r4 + 4 * hi;
      if (col < hd) {
        f16x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (f16)(st.dq[d][4 * r4 + e] * p.scale);
        *(f16x4*)(dQg + col) = v;
      }
    }
}

}  // namespace

bool tb_attn_il_dq_ok(const tb_attn_desc& d) {
  return !d.causal && d.hd == 40 && d.Sq % 128 == 0 && d.Skv % KVT == 0 && d.Skv >= 512 && d.ldk % 8 == 0 && d.ldv % 8 == 0 &&
         (int64_t)KVT * (d.ldk > d.ldv ? d.ldk : d.ldv) * 2 < ((int64_t)1 << 31);
}
int tb_attn_il_dq(const tb_attn_desc& d, hipStream_t s, int remap, int publish) {
  const size_t lds = D_NST * Q_STAGE_B;
  hipLaunchKernelGGL(attn_bwd_dq_il_kernel, dim3(d.Sq / 128, d.H, d.B), dim3(256), lds, s, d, remap, publish);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

bool tb_attn_il_dkv_ok(const tb_attn_desc& d) {
  return !d.causal && d.hd == 40 && d.Sq % KVT == 0 && d.Sq >= 2 * KVT && d.Skv % 128 == 0 && d.ws && d.ws_floats >= 2 * (int64_t)d.B * d.H * d.Sq &&
         d.ldq % 8 == 0 && d.lddo % 8 == 0 && (int64_t)KVT * (d.ldq > d.lddo ? d.ldq : d.lddo) * 2 < ((int64_t)1 << 31);
}
int tb_attn_il_dkv(const tb_attn_desc& d, hipStream_t s, int remap) {
  const size_t lds = D_NST * D_STAGE_B;
  hipLaunchKernelGGL(attn_bwd_dkv_il_kernel, dim3(d.Skv / 128, d.H, d.B), dim3(256), lds, s, d, remap);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

bool tb_attn_il_fwd_ok(const tb_attn_desc& d) {
  return !d.causal && d.hd == 40 && d.Sq % 256 == 0 && d.Skv % KVT == 0 && d.Skv >= 2 * KVT && d.ldk % 8 == 0 && d.ldv % 8 == 0 && !d.fp8_ws &&
         (int64_t)KVT * (d.ldk > d.ldv ? d.ldk : d.ldv) * 2 < ((int64_t)1 << 31);
}
int tb_attn_il_fwd(const tb_attn_desc& d, hipStream_t s, int remap) {
  const size_t lds = F_NST * F_STAGE_B;
  hipLaunchKernelGGL(attn_fwd_il_kernel, dim3(d.Sq / 256, d.H, d.B), dim3(256), lds, s, d, remap);
  TB_CHECK_LAUNCH();
  return TB_OK;
}
