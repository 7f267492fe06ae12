// 8-wave wide-tile MFMA GEMM / 3x3 convolution for the large-M levels of the UNet (gfx950).
//
// Why a second kernel next to gemm.hip: at the 64x64 and 32x32 feature maps (M = 32768 / 8192 output rows) the 4-wave 128x64 tiles
// of gemm.hip are bound by the operand stream from L2 into the LDS, not by the matrix pipe -- a 128x64x64 step moves 24 KB for
// 1 MFLOP (43 FLOP/B), and the 3x3 convolution re-streams every weight tap once per 128 pixels (128 FLOP per weight byte).  Here one
// workgroup of 8 waves owns a 256-pixel x 160-channel (conv) or 128-row x 320-column (Linear) tile:
//   * conv:   the (R+2) x (W+2) input halo of 256 output pixels is resident in LDS per 64-channel chunk (double buffered, the next
//             chunk's halo is fetched piecewise under the current chunk's taps) and each weight tap tile is used by 256 pixels:
//             205 FLOP per byte moved into the LDS, 2.3x less operand traffic than conv_halo_kernel<64>;
//   * Linear: the tile spans 320 output columns, so for the C = 320 / 640 layers the activation matrix is read once
//             (not once per 64-column tile) -- these layers are HBM-bound and the re-reads were most of their L2 traffic.
// Each wave computes a (16 MT) x (16 NT) sub-tile with v_mfma_f32_16x16x32_f16 (the 16-wide shape fits N = 320 = 4 x 80 and
// 160 = 2 x 80 without padding), accumulators transposed like gemm.hip (a lane owns 4 consecutive output columns of one row).
// One workgroup per CU (2 waves per SIMD), grid sized to exactly one round of the chip.
// Operand path: global_load_lds_dwordx4 (no staging registers), 128-byte LDS rows with the 16-byte chunk XOR-swizzled by row & 7
// (conflict-free for the 16x16x32 fragment reads at ANY row offset, which the shifted halo reads of the conv need).
#include "gemm_epi.h"

#ifndef G8_PROF
#define G8_PROF 0  // profiling build: per-phase s_memtime sums of waves 0 and 4 of workgroup 0 into dbg[16..]
#endif
#ifndef G8_DMAC
#define G8_DMAC 9  // bit 3: DMACH (the 256 x 80 convolution tile on a 4-slot ring), bit 0: DMAC, bit 1: DMACC (measured SLOWER, off), bit 2: DMAC2 (measured neutral, off); A/B build switch (TB_CFLAGS=-DG8_DMAC=n) (TB_CFLAGS=-DG8_DMAC=0): the Linear tiles' global -> LDS pieces in the LOAD phase, as in round 4
#endif
#ifndef G8_DMACC_MASK
#define G8_DMACC_MASK 0xA  // DMACC: bit k = load slot k of a convolution step goes out in the COMPUTE phase of its half (slots 0 .. WI - 1 weight pieces, WI the halo piece)
#endif
#ifndef G8_DMAC_L
#define G8_DMAC_L 0  // DMAC tiles: load slots of a step that stay in the LOAD phase (experiment: balancing the two phases of the 128 x 160 tile)
#endif
#ifndef G8_PSTAMP
#define G8_PSTAMP 0  // profiling build: extra prologue stamps in dbg[3] (setup done), dbg[4] (prologue stages issued), dbg[5] (their wait done)
#endif
#ifndef G8_NT
#define G8_NT 0  // experiment (TB_CFLAGS=-DG8_NT=1): non-temporal stores in the lean / GEGLU epilogues
#endif
#if G8_NT
#define G8_ST(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define G8_ST(ptr, val) (*(ptr) = (val))
#endif
#ifndef G8_STREAM
#define G8_STREAM 1  // the 256 x 160 / 256 x 128 convolution tiles (MT >= 4, 3-slot weight ring) as ONE instruction stream per wave: the fragments of the next 32-wide k-step are read between the MFMAs of this one, one barrier per tap (0: the round-2 LOAD / COMPUTE phases of two barrier-shifted wave groups)
#endif
#ifndef G8_ABL
#define G8_ABL 0  // profiling builds (TB_CFLAGS=-DG8_ABL=bits): 1 = no MFMAs, 2 = no in-loop global->LDS loads, 4 = no fragment reads
#endif

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int BM, int BN>
struct G8Epi {
  static constexpr int LDC = BN + 4;                                           // padded fp32 staging pitch: conflict-free b128 stores
  // stage the tile in halves when it does not fit; the small Linear tiles stay under 80 KB so that two workgroups share a CU
  static constexpr int LIM = (BM * BN <= 128 * 192 ? 80 : 96) * 1024;
  static constexpr int PASSES = (BM * LDC * 4 <= LIM) ? 1 : 2;
  static constexpr int PR = BM / PASSES;
  static constexpr size_t BYTES = (size_t)PR * LDC * 4;
};

__device__ __forceinline__ uint32_t lds_addr(const void* p) {  // 32-bit LDS byte address of a __shared__ pointer
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
// ds_read_b128 the compiler does not count: completion is awaited with wait_lgkmcnt() + sched_barrier(0) (cdna guide 5.7, rule 18)
__device__ __forceinline__ f16x8 lds_read16(uint32_t addr) {
  f16x8 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
template <int STRIDE>
__device__ __forceinline__ f16x8 lds_read16_off(uint32_t addr, int j) {  // addr + j * STRIDE as the instruction's immediate offset
  f16x8 v;
  switch (j) {
    case 0: asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr)); break;
    case 1: asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(STRIDE)); break;
    case 2: asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(2 * STRIDE)); break;
    case 3: asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(3 * STRIDE)); break;
    case 4: asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(4 * STRIDE)); break;
    default: asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(5 * STRIDE)); break;
  }
  return v;
}
template <int N>
__device__ __forceinline__ void wait_lgkmcnt_c() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_lgkmcnt(int n) {  // n is a compile-time constant after unrolling
  switch (n) {
    case 0: wait_lgkmcnt_c<0>(); break;
    case 1: wait_lgkmcnt_c<1>(); break;
    case 2: wait_lgkmcnt_c<2>(); break;
    case 3: wait_lgkmcnt_c<3>(); break;
    case 4: wait_lgkmcnt_c<4>(); break;
    case 5: wait_lgkmcnt_c<5>(); break;
    case 6: wait_lgkmcnt_c<6>(); break;
    case 7: wait_lgkmcnt_c<7>(); break;
    case 8: wait_lgkmcnt_c<8>(); break;
    case 9: wait_lgkmcnt_c<9>(); break;
    case 10: wait_lgkmcnt_c<10>(); break;
    case 11: wait_lgkmcnt_c<11>(); break;
    case 12: wait_lgkmcnt_c<12>(); break;
    case 13: wait_lgkmcnt_c<13>(); break;
    case 14: wait_lgkmcnt_c<14>(); break;
    default: wait_lgkmcnt_c<15>(); break;
  }
}
__device__ __forceinline__ void wait_vmcnt(int n) {  // n is wave-uniform
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

__device__ __forceinline__ int64_t m0p_of(int tm, int bm) { return (int64_t)tm * bm; }  // first row of Linear row panel tm
// The workgroup -> tile arithmetic ran 9 integer divisions (~25 scalar instructions each, one dependent chain per wave) in front of the first load of
// EVERY launch: of the ~970 instructions in front of the first barrier (5 500 cycles = 2.9 us of a 20 us K = 640 launch, warm or cold operands alike --
// s_memtime stamps, scratch/lin_stamps.py) a quarter was division.  The divisors are launch constants: the host passes ceil(2^32 / d) and the kernel
// multiplies (exact for n, d < 2^16).
struct G8Magic {
  uint32_t S, tiles_n, rn, xn, tiles_img, tiles_x, hc;
};
__device__ __forceinline__ int mg_div(int n, uint32_t magic) { return (int)__umulhi((uint32_t)n, magic); }
inline uint32_t mg_of(int d) { return d > 1 ? (uint32_t)((((uint64_t)1 << 32) + (uint32_t)d - 1) / (uint32_t)d) : 0u; }   // (d == 1: callers skip the divide)

// WM x WN waves (WM * WN == 8), each wave (16 MT) x (16 NT) outputs; CONV: 3x3 stride-1 halo convolution, else A rows linear
//
// SUB (round 4): the nearest-x2 upsampler convolutions (diffusers Upsample2D = F.interpolate(x, 2, "nearest") + conv3x3: up_blocks.*.upsamplers.0) as
// SUB-PIXEL convolutions.  An output pixel (2Y + py, 2X + px) of the fine map only ever sees a 2 x 2 window of the COARSE map -- rows Y - 1 + py,
// Y + py, with the 3 x 3 filter's rows summed pairwise ({0}, {1, 2} for py = 0; {0, 1}, {2} for py = 1), the same in x -- so the convolution is four
// 2 x 2-tap convolutions with pre-summed (frozen) weights: 16 instead of 36 tap products per coarse pixel, 2.25x fewer FLOP, and the 4x map is
// never written.  The kernel stays the halo kernel: tiles of 256 COARSE pixels, the usual (R + 2) x (TW + 2) halo, four taps whose halo shifts
// depend on the class.
//   SUB = 1 (forward, tb_gemm_desc.upsample == 2): A = the coarse map; the output class (py, px) is the `slice` of the workgroup (S = 4); row r of
//           the tile is stored to the fine pixel (2 y + py, 2 x + px);  W = [4 classes][N][4 taps][Cin].
//   SUB = 2 (dgrad, upsample == 3): A = the fine gradient read as four strided VIEWS (py, px), the k-loop walks (view, channel chunk) pairs with
//           four taps each, the output is the coarse map; W = [N][4 views][4 taps][Cin]; split-K over the (view, chunk) list as usual.
// KH = 2 (round 5, Linear tiles with MT < 4): the 8 waves are WM x WN x 2 -- the two waves of a SIMD (w, w + 4) own the SAME (16 MT) x (16 NT)
// outputs and split every 64-wide k-step between them (32 k each), their accumulators are added through the LDS in front of the epilogue.  A wave
// tile twice as tall for the same register tile per wave: the 128 x 80 one-per-CU tile read (16 + 80) fragment rows per 5 MFMAs (LDS-read bound,
// 0.65 us per k-step at one tile per CU), as 4 x 1 x 2 waves it reads (32 + 80) rows per 10.
template <int WM, int WN, int MT, int NT, bool CONV, int NS, int SUB = 0, int KH = 1, bool ST = true>
__global__ __launch_bounds__(512, 2) void gemm8_kernel(const tb_gemm_desc p, int tiles_m, int tiles_n, int wshift, int a_rows8,
                                                          unsigned long long* dbg, int S, float* __restrict__ ws, int64_t npad, int xn, G8Magic mg) {
  static_assert(WM * WN * KH == 8, "8 waves");
  static_assert(KH == 1 || (KH == 2 && !CONV && MT < 4), "k-halves: Linear tiles whose step is one phase pair");
#define G8_STAMP(k)                                                                                  \
  if (dbg && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))                  \
    dbg[(blockIdx.x ? 8 : 0) + (k)] = __builtin_amdgcn_s_memtime();
  G8_STAMP(0)
  extern __shared__ __attribute__((aligned(128))) unsigned char smem_raw[];
  constexpr int BM = WM * MT * 16, BN = WN * NT * 16, BK = 64;
  static_assert(!SUB || CONV, "SUB is a convolution mode");
  constexpr int TAPS = CONV ? (SUB ? 4 : 9) : 1;
  constexpr int NI_W = BN / 8;               // weight-tile load instructions (8 rows x 128 B each) per step
  constexpr int WI = (NI_W + 7) / 8;         // ... per wave
  constexpr int MAXHI = CONV ? 7 : (BM / 64 > 0 ? BM / 64 : 1);  // A-panel load instructions per wave per chunk
  const int t = threadIdx.x, lane = t & 63, l15 = lane & 15, lq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int kh = KH == 2 ? wave / (WM * WN) : 0, wv = KH == 2 ? wave - kh * (WM * WN) : wave;   // (KH == 2: kh == the phase group, wave >> 2)
  const int wm = wv / WN, wn = wv - wm * WN;
  const bool stager = KH == 1 || kh == 0;   // the waves that hold the k-summed accumulators in the epilogue
  // LDS: [A panels | W stages]: conv 2 halo panels (a_rows8 pixels x 128 B), Linear NS tile-row panels; NS weight stages of BN rows x 128 B
  constexpr int NA = CONV ? 2 : NS;
  f16* const As = reinterpret_cast<f16*>(smem_raw);
  const int a_elems = a_rows8 * BK;
  f16* const Ws = As + NA * a_elems;

  // S > 1 (conv only): split-K over the 64-channel chunks -- the S slices of a tile are separate workgroups that store raw fp32 partials to
  // ws[slice][M][npad]; gemm.hip's splitk_reduce_kernel sums them and applies the epilogue.  For the 16x16 maps (M = 2048: 64 tiles of 256 x 160)
  // Workgroup -> tile, XCD-aware (workgroup b runs on XCD b & 7, each XCD has its own L2).  xn > 0: the tile grid is cut into (8 / xn) x xn
  // rectangles, one per XCD, so an XCD fetches 1 / (8 / xn) of the activations and 1 / xn of the weights (host: launch8 picks the cut with the
  // least fabric traffic -- with row panels per XCD, the 16x16-map layers pulled the whole 26..30 MB weight matrix into every L2: 215 MB for a
  // 31 MB GEGLU projection).  xn == 0: contiguous runs of row panels (grids that do not divide evenly).
  int tm, tn, slice = 0;
  if (xn > 0) {
    const int bid = blockIdx.x, xcd = bid & 7;
    int j = bid >> 3;
    if (S > 1) {
      const int jq = mg_div(j, mg.S);
      slice = j - jq * S;
      j = jq;
    }
    const int xs = xn == 8 ? 3 : (xn == 4 ? 2 : (xn == 2 ? 1 : 0));   // (xn is a power of two: launch8)
    const int rn = tiles_n >> xs, rm = (tiles_m << xs) >> 3;
    const int jm = rn > 1 ? mg_div(j, mg.rn) : j;
    tm = (xcd >> xs) * rm + jm;
    tn = (xcd & (xn - 1)) * rn + (j - jm * rn);
  } else {
    const int nwg = tiles_m * tiles_n * S;
    const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    if (S > 1) {
      const int tq = mg_div(tile, mg.S);
      slice = tile - tq * S;
      tile = tq;
    }
    tm = tiles_n > 1 ? mg_div(tile, mg.tiles_n) : tile, tn = tile - tm * tiles_n;  // the tiles_n column tiles of one row panel are adjacent (same XCD)
  }
  const int cls = SUB == 1 ? slice : 0;   // SUB == 1: the `slice` index is the output class 2 py + px, not a k-range
  if constexpr (SUB == 1) slice = 0;
  const int Ssplit = SUB == 1 ? 1 : S;
  const int64_t n0 = (int64_t)tn * BN;
  // Linear tiles: the BN bias values go through a small LDS array filled under the prologue loads.  Loaded at the point of use (after the main
  // loop) they queued behind the CU's in-flight stores and operand loads: 8300 cycles of a 35000-cycle 128x128x320 GEGLU tile were this wait.
  // (The conv tiles have no LDS to spare -- 160 KB exactly at 64-wide feature maps -- and amortise it over 45+ k-steps.)
  constexpr int OPB = CONV ? 0 : NS * (BM + BN) * 128;
  constexpr int EPB = (int)G8Epi<BM, BN>::BYTES;
  constexpr int REDB = (CONV || BM * BN > 128 * 160) ? 0 : (int)(G8Epi<BM, BN>::PR * (BN / 8) * 8);   // row-statistic partials of a producer launch (rs_out), behind the staging tile
  const int boff = (!CONV && p.rs_out && (OPB > EPB ? OPB : EPB) < EPB + REDB) ? EPB + REDB : (OPB > EPB ? OPB : EPB);   // (launch8 sizes the LDS the same way)
  float* const bias_s = reinterpret_cast<float*>(smem_raw + boff);
  float bias_v = 0.f;
  if (!CONV && t < BN && p.bias) bias_v = p.bias[n0 + t];
  // fused-LayerNorm epilogues (TB_ACT_LN_FWD / TB_ACT_LN_BWD; host: the tile spans the row, n0 == 0): gamma / beta take the same route
  const bool ln_epi = !CONV && (p.act == TB_ACT_LN_FWD || p.act == TB_ACT_LN_BWD);
  // LayerNorm folded into THIS Linear (tb_gemm_desc.rs_in): c1 = ln_gamma rides in the gamma slot; thread t < BM sums the producer's per-tile
  // (sum, sum of squares) partials of row m0 + t -- requested here, in front of the operand prologue, consumed next to bias_s below
  constexpr bool RS_OK = !CONV && BM * BN <= 128 * 160;   // (the 128x320 tile sits at 243 registers: no room; launch8 refuses the request there)
  const bool lnf = RS_OK && p.rs_in != nullptr;
  float gam_v = 0.f, bet_v = 0.f;
  if ((ln_epi || lnf) && t < BN) {
    gam_v = p.ln_gamma[n0 + t];
    if (p.act == TB_ACT_LN_FWD) bet_v = p.ln_beta[n0 + t];
  }
  typedef __attribute__((ext_vector_type(2))) float f32x2r_t;
  f32x2r_t rs_part[RS_OK ? 8 : 1];   // slots 0 .. 7 (slots 8 .. 15, when in use, are added where these are consumed)
  if constexpr (RS_OK) {
    if (lnf && t < BM) {
      const int64_t mrow = min(m0p_of(tm, BM) + t, p.M - 1);
      const f32x2r_t* src = reinterpret_cast<const f32x2r_t*>(p.rs_in) + mrow * p.rs_ld;
      const int rn = p.rs_n;
#pragma unroll
      for (int j = 0; j < 8; ++j) rs_part[j] = j < rn ? src[j] : f32x2r_t{0.f, 0.f};
    }
  }
  // epilogue unit geometry (used early by the residual prefetch): a thread keeps ONE 8-column group and walks rows
  using E = G8Epi<BM, BN>;
  constexpr int LDC = E::LDC, PASSES = E::PASSES, PR = E::PR;
  constexpr int UPR = BN / 8;            // 8-column groups per row
  constexpr int TPR = 512 / UPR;         // rows covered per sweep (threads beyond UPR * TPR idle)
  constexpr int NU = (PR + TPR - 1) / TPR;
  const int cg = t % UPR, rslot = t / UPR;
  const int64_t n = n0 + cg * 8;

  // ---- tile geometry
  const int TW = 1 << wshift, W = SUB == 1 ? p.Win : p.Wout, H = SUB == 1 ? p.Hin : p.Hout;  // the map the tile grid walks (SUB: the coarse one)
  const int R = BM >> wshift, HC = TW + 2, NH = CONV ? (R + 2) * HC : BM;
  const int NI_H = a_rows8 >> 3;
  int64_t m0;
  int y0 = 0, x0 = 0, bimg = 0;
  const int hw = H * W;
  if (CONV) {
    const int rshift = (BM == 256 ? 8 : 7) - wshift;                        // R = BM >> wshift rows per tile, a power of two
    const int tiles_x = W >> wshift, tiles_img = (H >> rshift) * tiles_x;   // (host: H % R == 0)
    bimg = tiles_img > 1 ? mg_div(tm, mg.tiles_img) : tm;
    const int trem = tm - bimg * tiles_img;
    const int ty = tiles_x > 1 ? mg_div(trem, mg.tiles_x) : trem;
    y0 = ty << rshift;
    x0 = (trem - ty * tiles_x) << wshift;
    m0 = (int64_t)bimg * hw + (int64_t)y0 * W + x0;
  } else {
    m0 = (int64_t)tm * BM;
  }
  // time-embedding row bias (rowbias[m / rows_per_group]): a convolution tile lies inside ONE image, so with rows_per_group == H W (every ResnetBlock2D
  // launch) the group is the tile's image index -- no 64-bit divisions (three of them, ~100 instructions each, sat in front of the first load of the
  // 256 x 80 tile; two in the 256 x 160 tile's epilogue)
  const bool rb_img = CONV && SUB == 0 && p.rowbias && p.rows_per_group == (int64_t)hw;
  auto rb_group0 = [&]() -> int64_t { return rb_img ? (int64_t)bimg : m0 / p.rows_per_group; };
  auto rb_is_uniform = [&](int64_t m_last) -> bool { return !p.rowbias || rb_img || (m0 / p.rows_per_group == m_last / p.rows_per_group); };
  // output row of tile row r: the tile's pixels in map order; SUB == 1: the class's pixels (2 y + py, 2 x + px) of the fine (2H x 2W) map
  const int64_t mo0 = SUB == 1 ? ((int64_t)bimg * 2 * H + 2 * y0 + (cls >> 1)) * (2 * W) + 2 * x0 + (cls & 1) : m0;
  const int oys = SUB == 1 ? 4 * W : W, oxs = SUB == 1 ? 2 : 1;
  auto out_row = [&](int r) -> int64_t {
    if constexpr (SUB == 1) return mo0 + (int64_t)(r >> wshift) * oys + (int64_t)(r & (TW - 1)) * oxs;
    else return CONV ? m0 + (int64_t)(r >> wshift) * W + (r & (TW - 1)) : m0 + r;
  };
  // Early epilogue operands (a global load issued after the main loop queues behind the CU's own stores and operand traffic for microseconds).
  // conv: bias, plus the time-embedding row bias when it is constant over the tile, stay in 8 registers across the main loop;
  // Linear: bias goes through bias_s (above) and the residual rows of the first staging pass are fetched now into NU x 4 registers.
  constexpr bool PRE_R = !CONV && BM * BN > 128 * 160;  // (the 128x128 tile must stay under 128 registers: two workgroups per CU)
  constexpr bool PRE_B = CONV && MT < 4;  // (the 256x160 conv tile sits at 254 of 256 registers already: its bias is loaded in the epilogue)
  float b8[8];
  f16x8 rv0[PRE_R ? NU : 1];
  bool fast = false;
  if (PRE_B) {
    const EpiFlags ef0 = epi_flags(p);
    const bool rb_uniform = rb_is_uniform(m0 + ((int64_t)(R - 1) * W + TW - 1));
    fast = p.c_dtype == TB_F16 && ef0.c_vec && (!p.R || (ef0.r_vec && p.r_dtype == TB_F16)) && (p.act == TB_ACT_NONE || p.act == TB_ACT_SILU) &&
           !p.C2 && rb_uniform;
    epi_load_bias8(p, n, b8);
    if (fast && p.rowbias) {
      const float* rb = p.rowbias + rb_group0() * p.ldrb + n;
#pragma unroll
      for (int e = 0; e < 8; ++e) b8[e] += rb[e];
    }
  } else if (PRE_R) {
    if (p.R && p.ldr % 8 == 0 && ((uintptr_t)p.R) % 16 == 0 && p.r_dtype == TB_F16 && (p.act == TB_ACT_NONE || p.act == TB_ACT_SILU || ln_epi) && rslot < TPR) {
#pragma unroll
      for (int it = 0; it < NU; ++it) {
        const int64_t m = min(m0 + min(rslot + it * TPR, PR - 1), p.M - 1);
        rv0[it] = *(const f16x8*)((const f16*)p.R + m * p.ldr + n);
      }
    }
  }

#if G8_PSTAMP
  G8_STAMP(7)
#endif
  const int cp = lane & 7, rl = lane >> 3;
  // STREAM: see the main loop
  constexpr bool STREAM = G8_STREAM && ST && CONV && SUB == 0 && MT >= 4 && NT <= 5 && NS == 3 && KH == 1 && !G8_PROF;
  // ---- A-panel sources of this lane (fixed across chunks, + 64 halfs per chunk)
  const f16* h_ptr[MAXHI];
  int h_step[MAXHI];
  uint32_t h_off[STREAM ? MAXHI : 1];   // STREAM keeps 32-bit byte offsets from p.A across its main loop (~0: a row outside the map) instead of pointer + step: 13 registers
  const f16* zero = g_zero_line;
#pragma unroll
  for (int i = 0; i < MAXHI; ++i) {
    const int j = wave + 8 * i;  // instruction index: rows j*8 .. j*8+7 of the panel
    const int hr = j * 8 + rl;
    bool ok;
    int64_t grow;
    if (CONV) {
      const int hy = mg_div(hr, mg.hc), hx = hr - hy * HC;   // (HC = TW + 2 >= 18: the multiply-high division)
      const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
      ok = j < NI_H && hr < NH && yy >= 0 && yy < H && xx >= 0 && xx < W;
      grow = SUB == 2 ? ((int64_t)bimg * 2 * H + 2 * yy) * (2 * W) + 2 * xx   // view (0, 0) of the fine gradient; other views: + a uniform offset
                      : (int64_t)bimg * hw + (int64_t)yy * W + xx;
    } else {
      ok = j < NI_H && m0 + hr < p.M;
      grow = m0 + hr;
    }
    h_ptr[i] = ok ? (const f16*)p.A + grow * p.lda + ((cp ^ (hr & 7)) << 3) : zero;
    h_step[i] = ok ? BK : 0;
    if constexpr (STREAM) h_off[i] = ok ? (uint32_t)((grow * p.lda + ((cp ^ (hr & 7)) << 3)) * 2) : ~0u;
  }
  uint32_t w_off[WI];
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    const int row = (wave + 8 * i) * 8 + rl;
    const int64_t n = n0 + (row < BN ? row : 0);
    w_off[i] = (uint32_t)((n * p.ldw + ((cp ^ (row & 7)) << 3)) * 2);
  }
  const int kpt = CONV ? p.Cin / BK : 1;                 // k-tiles per tap
  const int nchunk_all = CONV ? (SUB == 2 ? 4 * kpt : kpt) : (int)(p.K / BK);   // SUB == 2: (view, channel chunk) pairs, view-major
  const int c_begin = Ssplit > 1 ? nchunk_all * slice / Ssplit : 0;
  const int nchunk = Ssplit > 1 ? nchunk_all * (slice + 1) / Ssplit : nchunk_all;  // end of this workgroup's chunk range [c_begin, nchunk)
  // SUB == 2: chunk index -> (view, channel chunk), kept incrementally for the current chunk (_c) and the next one (_n): the halo source of a
  // view is view (0, 0)'s plus (py * Wsrc + px) pixels, its weights sit 4 * kpt k-tiles behind the previous view's
  int sv_view_c = 0, sv_lc_c = 0, sv_view_n = 0, sv_lc_n = 0;
  if (SUB == 2) {
    sv_view_c = c_begin / kpt, sv_lc_c = c_begin - sv_view_c * kpt;
    sv_view_n = sv_view_c, sv_lc_n = sv_lc_c + 1;
    if (sv_lc_n == kpt) sv_lc_n = 0, ++sv_view_n;
  }
  auto sub_a_off = [&](int view, int lcv) -> int64_t {  // element offset of (view, channel chunk) from view (0, 0), chunk 0 (in-map lanes only)
    return (int64_t)lcv * BK + ((int64_t)(view >> 1) * (2 * W) + (view & 1)) * p.lda;
  };

  auto stage_a_piece = [&](int c, int i, bool next) {  // one load instruction of chunk c's panel (SUB == 2: c is the current or the NEXT chunk)
    const int j = wave + 8 * i;
    if constexpr (SUB == 2) {
      if (j < NI_H) {
        const int64_t off = next ? sub_a_off(sv_view_n, sv_lc_n) : sub_a_off(sv_view_c, sv_lc_c);
        glds16(h_ptr[i] + (h_step[i] ? off : 0), As + (c & 1) * a_elems + j * 8 * BK);
      }
    } else {
      if (j < NI_H) glds16(h_ptr[i] + (int64_t)c * h_step[i], As + (c & 1) * a_elems + j * 8 * BK);
    }
  };

  // loads this wave issues per weight stage (wave-uniform), for the counted waits below
  int n_w = 0;
#pragma unroll
  for (int i = 0; i < WI; ++i) n_w += (wave + 8 * i < NI_W) ? 1 : 0;
  int n_a = 0;
#pragma unroll
  for (int i = 0; i < MAXHI; ++i) n_a += (wave + 8 * i < NI_H) ? 1 : 0;

  // ---- software pipeline: NS weight stages (and, for Linear, NS activation stages); the loads of step s + NS - 1 are issued in step s,
  // so NS - 2 steps of loads stay in flight across each barrier (counted vmcnt, raw s_barrier: cdna guide T3/T4).  An L2 round trip under
  // load is ~1 us, as long as one step of MFMAs: with NS = 2 every step waited for it.
  static_assert(!CONV || SUB || TAPS % NS == 0 || NS == 4, "the weight ring slot of a conv step is tap % NS (SUB and the 4-slot ring: run-time ring counters, like the Linear path)");
  // The stage loaded during step (c, tap) is the one NS - 1 steps ahead: (lc, ltap).  The tap loop is fully unrolled, so tap, ltap, the
  // ring slots and the tap's halo offset are compile-time; only the chunk index is a loop variable.  (With a rolled tap loop the LOAD
  // phase was ~100 instructions, half of them scalar index arithmetic, and took 420 cycles against 340 for the 20 MFMAs of the partner.)
  constexpr int NSLOT = CONV ? WI + (SUB ? 3 : 1) : WI + MAXHI;  // load slots per step: WI weight pieces, then the halo piece(s) (conv: one of the
                                                                 // next chunk's <= 7 per tap; SUB: up to three, see issue_slot) / MAXHI A pieces
  // dgrad (sign < 0) gathers with flipped offsets: walk the taps in reverse weight order instead, so that the gather offset of unrolled
  // step t is always (t / 3, t % 3)
  const int wtap0 = p.sign > 0 ? 0 : TAPS - 1, wtapd = p.sign > 0 ? 1 : -1;
  auto issue_slot = [&](int k, int lc, int ltap, int lslot, int c, int tap, int lnext) {  // lnext: the loaded stage belongs to chunk c + 1
    if (G8_ABL & 2) return;
    if (k < WI) {
      const int j = wave + 8 * k;
      if (lc < nchunk && j < NI_W) {
        if constexpr (SUB == 0) {
          const char* base = (const char*)p.W + (int64_t)((wtap0 + wtapd * ltap) * kpt + lc) * BK * 2;
          glds16((const f16*)(base + w_off[k]), Ws + lslot * (BN * BK) + j * 8 * BK);
        } else {
          int64_t kt;   // k-tile of the stage inside a weight row
          if (SUB == 2) kt = (int64_t)((lnext ? sv_view_n : sv_view_c) * 4 + ltap) * kpt + (lnext ? sv_lc_n : sv_lc_c);
          else kt = (int64_t)ltap * kpt + lc;
          const char* base = (const char*)p.W + (SUB == 1 ? (int64_t)cls * p.N * p.ldw * 2 : 0) + kt * BK * 2;
          glds16((const f16*)(base + w_off[k]), Ws + lslot * (BN * BK) + j * 8 * BK);
        }
      }
    } else if (CONV) {
      // SUB: the next chunk's halo pieces go out in taps 0 .. 2 only (pieces tap, tap + 3, tap + 6): the counted wait at the end of a step covers
      // everything EXCEPT that step's own loads, so a piece issued in the last tap (3) would not have landed when the next chunk's first tap reads it
      const int piece = SUB ? tap + 3 * (k - WI) : tap;
      if (tap >= 0 && (!SUB || tap < 3) && piece < MAXHI && c + 1 < nchunk) stage_a_piece(c + 1, piece, true);
    } else if (lc < nchunk) {
      const int i = k - WI, j = wave + 8 * i;
      if (j < NI_H) glds16(h_ptr[i] + (int64_t)lc * h_step[i], As + lslot * a_elems + j * 8 * BK);
    }
  };
  auto stage_count = [&](int lc, int c, int tap) -> int {  // loads this wave issues in step (c, tap): for the counted wait of the NEXT step
    int n = lc < nchunk ? n_w + (CONV ? 0 : n_a) : 0;
    if (!SUB && CONV && c + 1 < nchunk && tap < MAXHI && wave + 8 * tap < NI_H) ++n;
    if (SUB && c + 1 < nchunk && tap < 3) {
#pragma unroll
      for (int q = 0; q < 3; ++q)
        if (tap + 3 * q < MAXHI && wave + 8 * (tap + 3 * q) < NI_H) ++n;
    }
    return n;
  };
  // DMAC (round 5; Linear tiles whose step is one phase pair, rings of >= 4 stages): the step's global -> LDS pieces are issued in the COMPUTE phase,
  // between the MFMAs, not in the LOAD phase.  Measured on the 128 x 80 tile (scratch/kh_time.py with -DG8_ABL builds): a k-step took 0.625 us =
  // 0.26 (MFMAs + the two barriers) + 0.10 .. 0.17 (fragment reads) + 0.20 (the pieces) -- the phases ADDED: a LOAD phase carrying 12 fragment reads and
  // 3.25 pieces per wave ran ~600 cycles beside a COMPUTE phase of 160, and a piece costs 100 .. 185 issue cycles inside such a phase against ~60 among
  // bare MFMAs (microarch guide).  The counted wait stays at the end of the LOAD phase (so the second wave group's pieces have landed one interval
  // before the first group reads them): what must have landed is the next step's stage, issued two steps ago; the previous step's pieces stay in flight.
  constexpr bool DMAC = G8_DMAC && !CONV && MT < 4 && NS >= 4 && !(G8_ABL & 2);
  // DMACC: the same for the 9-tap convolution tiles whose step is TWO phase pairs (MT >= 4: 256 pixels x 160 / 128 channels, 3-slot weight ring): the
  // pieces of a step go out in its two COMPUTE phases; at the end of the second LOAD phase everything but this step's first-half pieces has landed,
  // i.e. the next step's weight stage (issued during the previous step) and every halo piece issued before this step
  // (DMACC measured, one MI355X, scratch/g8_time.py: every piece in the COMPUTE phases (G8_DMACC_MASK = 0xF) 320 -> 320 @ 64x64 57.7 -> 65.5 us, 960 -> 320
  // 171 -> 184 us, the step 28.9 -> 30.0 ms; one piece per half (0xA) 57.1 -> 55.7 / 166 -> 160 us but 29.12 = 29.15 ms in the step: the phase stamps
  // (-DG8_PROF) say LOAD 527 cycles against COMPUTE 365 per half-step, and the LOAD phase is its 36 KB of fragment reads (288 cycles of LDS transfer at
  // 128 B/clk), not the two pieces; off)
  // DMAC2: Linear tiles with TWO phase pairs per step on a 2-stage ring (the 128 x 128 GEGLU tile, 128 x 320): all pieces of the next stage go out in
  // the FIRST compute phase; the wait for them stays at the end of the second LOAD phase (vmcnt(0)), one interval before the other wave group reads them
  // (DMAC2 measured, scratch/geglu_time.py: GEGLU 8192x5120x640 126 -> 119..125 us, 32768x320x1280 46.7 -> 47.8 us, the step 29.00 = 29.00 ms: neutral; off)
  constexpr bool DMAC2 = (G8_DMAC & 4) && !CONV && MT >= 4 && NS == 2 && !(G8_ABL & 2);
  // DMACH (bit 3): the 256-pixel x 80-channel convolution tile (MT = 2: ONE phase pair per step) on a 4-slot weight ring, as DMAC: every piece of a step
  // between the MFMAs of its COMPUTE phase, the wait at the end of the LOAD phase leaves the previous step's pieces in flight.  Phase stamps of the
  // 3-slot version (-DG8_PROF, 640 -> 640 @ 32x32): LOAD 709 cycles against 355 for the 20 MFMAs -- the imbalance DMAC removed from the Linear tiles.
  constexpr bool DMACH = (G8_DMAC & 8) && CONV && SUB == 0 && MT < 4 && NS == 4 && !(G8_ABL & 2);
  constexpr bool DMACC = (G8_DMAC & 2) && CONV && SUB == 0 && MT >= 4 && NS == 3 && !(G8_ABL & 2);
  const uint32_t as_addr0 = lds_addr(As), ws_addr0 = lds_addr(Ws);
  auto conv_slot_issues = [&](int k, int lc, int c, int tap) -> int {   // DMACC: does this wave issue a piece for slot k in step (c, tap)?
    if (k < WI) return (lc < nchunk && wave + 8 * k < NI_W) ? 1 : 0;
    return (tap < MAXHI && c + 1 < nchunk && wave + 8 * tap < NI_H) ? 1 : 0;
  };
  auto issue_conv_slot_asm = [&](int k, int lc, int ltap, int lslot, int c, int tap) {   // DMACC: issue_slot's SUB == 0 convolution case as asm pieces
    if (k < WI) {
      const int j = wave + 8 * k;
      if (lc < nchunk && j < NI_W)
        glds16_asm_so((const char*)p.W + (int64_t)((wtap0 + wtapd * ltap) * kpt + lc) * BK * 2, w_off[k], ws_addr0 + (uint32_t)(lslot * (BN * BK) + j * 8 * BK) * 2);
    } else {
      const int j = wave + 8 * tap;
      if (tap < MAXHI && c + 1 < nchunk && j < NI_H)
        glds16_asm(h_ptr[tap] + (int64_t)(c + 1) * h_step[tap], as_addr0 + (uint32_t)(((c + 1) & 1) * a_elems + j * 8 * BK) * 2);
    }
  };
  uint32_t a_off[CONV ? 1 : MAXHI];   // Linear: 32-bit byte offsets of this lane's A-panel rows from p.A (rows past M: row 0 -- never stored); filled below
  auto issue_slot_asm = [&](int k, int lc, int lslot) {   // (Linear only) piece k of stage lc into ring slot lslot
    if (k < WI) {
      const int j = wave + 8 * k;
      if (lc < nchunk && j < NI_W) glds16_asm_so((const char*)p.W + (int64_t)lc * BK * 2, w_off[k], ws_addr0 + (uint32_t)(lslot * (BN * BK) + j * 8 * BK) * 2);
    } else if (lc < nchunk) {
      const int i = k - WI, j = wave + 8 * i;
      if (j < NI_H) glds16_asm_so((const char*)p.A + (int64_t)lc * BK * 2, a_off[CONV ? 0 : i], as_addr0 + (uint32_t)(lslot * a_elems + j * 8 * BK) * 2);
    }
  };
#if G8_PSTAMP
  G8_STAMP(3)
#endif
  int cnt_prev = 0;
  if (CONV) {
#pragma unroll
    for (int i = 0; i < MAXHI; ++i) stage_a_piece(c_begin, i, false);
  }
#pragma unroll
  for (int st = 0; st < (STREAM ? NS : NS - 1); ++st) {  // stages 0 .. NS-2 (STREAM: the whole ring); the youngest one's loads may stay in flight at step 0 (NS = 3)
    const int lc = c_begin + st / TAPS, ltap = st % TAPS;
    cnt_prev = lc < nchunk ? n_w + (CONV ? 0 : n_a) : 0;
#pragma unroll
    for (int k = 0; k < NSLOT; ++k)
      if (!CONV || k < WI) issue_slot(k, lc, ltap, st % NS, -2, -1, 0);
  }
  if (NS == 2) cnt_prev = 0;
  // ---- everything the first stages' loads do not need sits BEHIND their issue (it runs under their latency): accumulators, fragment addressing,
  // the DMAC path's 32-bit A offsets
  f32x4_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fragment addressing (bytes inside a panel / stage): row * 128 + ((chunk ^ (row & 7)) << 4), chunk = 4 s + lq for sub-step s
  int arow0[MT];  // panel row of this lane's output row for tap offset (0, 0)
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int ml = (wm * MT + i) * 16 + l15;
    arow0[i] = CONV ? (ml >> wshift) * HC + (ml & (TW - 1)) : ml;
  }
  uint32_t arow128[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) arow128[i] = (uint32_t)arow0[i] * 128;
  // W rows wn*NT*16 + 16 j + l15: row & 7 == l15 & 7 for every j, so one base + immediates
  const int brow = wn * NT * 16 + l15;
  const int boff0 = brow * 128 + ((lq ^ (brow & 7)) << 4), boff1 = boff0 ^ 64;

  if constexpr (!CONV) {
#pragma unroll
    for (int i = 0; i < MAXHI; ++i) {
      const int hr = (wave + 8 * i) * 8 + rl;
      const int64_t grow = m0 + hr < p.M ? m0 + hr : 0;
      a_off[i] = (uint32_t)((grow * p.lda + ((cp ^ (hr & 7)) << 3)) * 2);
    }
  }
  int cnt_hist = cnt_prev;   // loads this wave issued in the previous step (prologue: the youngest stage)
#if G8_PSTAMP
  G8_STAMP(4)
#endif
  wait_vmcnt(cnt_prev);                                              // stage 0 (and the first halo) have landed ...
#if G8_PSTAMP
  G8_STAMP(5)
#endif
  if (!CONV && t < BN) {
    bias_s[t] = bias_v;
    if (ln_epi || lnf) bias_s[BN + t] = gam_v, bias_s[2 * BN + t] = bet_v;
  }
  f32x2r_t* const rowst_s = reinterpret_cast<f32x2r_t*>(bias_s + 3 * BN);   // [BM] (mean, rstd) of the folded LayerNorm
  if constexpr (RS_OK) {
    if (lnf && t < BM) {
      float sx = 0.f, sq = 0.f;
      if (p.rs_n > 8) {   // (uniform; the 16x16-map producers' 16 tiles of 80 columns)
        const f32x2r_t* src = reinterpret_cast<const f32x2r_t*>(p.rs_in) + min(m0 + t, p.M - 1) * p.rs_ld;
        const int rn = p.rs_n;
        f32x2r_t hi8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) hi8[j] = 8 + j < rn ? src[8 + j] : f32x2r_t{0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) sx += rs_part[j][0], sq += rs_part[j][1];
#pragma unroll
        for (int j = 0; j < 8; ++j) sx += hi8[j][0], sq += hi8[j][1];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) sx += rs_part[j][0], sq += rs_part[j][1];
      }
      const float inv_k = 1.f / (float)p.K, mean = sx * inv_k;
      const float rstd = rsqrtf(fmaxf(sq * inv_k - mean * mean, 0.f) + p.ln_eps);
      rowst_s[t] = f32x2r_t{mean, rstd};
      if (tn == 0 && p.ln_stats && m0 + t < p.M) *reinterpret_cast<f32x2r_t*>(p.ln_stats + 2 * (m0 + t)) = f32x2r_t{mean, rstd};
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // ... for every wave
  G8_STAMP(1)

  // ---- main loop: the two waves of each SIMD (wave w and w + 4) run the SAME stream one phase apart.  A step is 2 * HALVES phases
  // separated by s_barrier: in its LOAD phase a wave issues the fragment reads of one half-step plus its share of the next stage's
  // global->LDS loads and waits for the reads; in its COMPUTE phase it only issues the half-step's MFMAs.  With the second wave group
  // shifted by one barrier, every SIMD always has one wave on the matrix pipe and one on the LDS / address / DMA-issue side.
  // (Measured before: run in lockstep, the load side (0.73 us per step) and the MFMA side (0.68 us) simply added up -- an LDS-DMA
  // instruction blocks its wave for ~100 issue cycles and the partner wave was doing exactly the same thing at the same time.)
  constexpr int HALVES = MT >= 4 ? 2 : 1;          // half-steps (32 of the 64 k) per phase pair; small wave tiles do the whole step at once
  constexpr int SPH = 2 / HALVES;                  // 32-wide sub-steps per half
  constexpr int SLOTS0 = NS == 2 ? NSLOT : (NSLOT + HALVES - 1) / HALVES;  // load slots issued in the first LOAD phase of a step
  const int group = wave >> 2;
#if G8_PROF
  unsigned long long pf_sum[4] = {0, 0, 0, 0}, pf_t = 0;
#define G8_PF(k)                                                     \
  {                                                                  \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();    \
    if ((k) >= 0) pf_sum[(k) < 0 ? 0 : (k)] += now_ - pf_t;          \
    pf_t = now_;                                                     \
  }
#else
#define G8_PF(k)
#endif
  const uint32_t as_addr = lds_addr(As), ws_addr = lds_addr(Ws);
  if constexpr (STREAM) {
    // ---- STREAM main loop (round 6).  scratch/r6/ldsmfma.hip: 9 fragment reads issued BETWEEN the 20 MFMAs of a 32-wide k-step cost nothing -- two waves per
    // SIMD run at 296 ns per step against 285 ns for the bare MFMAs (one wave per SIMD with 128 x 80 tiles the same), while the LOAD / COMPUTE phase
    // pairs below measured LOAD 527 cycles against COMPUTE 365.  So every wave runs ONE stream: the fragments of the next k-step are read into the
    // other register buffer while this step's MFMAs issue, and the only barrier is the one per tap that hands over a weight-ring slot:
    //   sub-step 0 of tap T: MFMAs (T, k 0..31)  | reads (T, k 32..63)
    //   sub-step 1 of tap T: wait for T + 1's weights, barrier -- every wave has now READ all of T's slot (its second half went into registers during
    //                        sub-step 0), so the slot is refilled at once with tap T + 3: the 3-slot ring runs three taps ahead, ~2 taps (2.6 k cycles)
    //                        of lead on the wait;  MFMAs (T, k 32..63) | the refill pieces, one halo piece of the next chunk, reads (T + 1, k 0..31)
    f16x8 af[2][MT], bf[2][NT];
    const uint32_t bb0 = ws_addr + boff0;   // (k 32..63: ^ 64)
    uint32_t a0[MT];
    auto frag_rows = [&](int cc, int tap) {   // a0[]: this lane's fragment addresses of (chunk cc, tap), k 0..31 (k 32..63: ^ 64)
      const int shift = (tap / 3) * HC + (tap % 3);
      const uint32_t ab = as_addr + (uint32_t)(cc & 1) * (uint32_t)(a_elems * 2) + (uint32_t)shift * 128u;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        uint32_t ar = (uint32_t)arow0[i];
        asm volatile("" : "+v"(ar));   // opaque: left to itself hipcc hoists the 9 x MT tap addresses out of the chunk loop (36 registers, spills)
        a0[i] = ab + (ar << 7) + ((lq ^ ((ar + shift) & 7)) << 4);
      }
    };
    int abl_c = c_begin;
    uint32_t wb = bb0;   // weight-fragment base of the sub-step being read (slot and k-half folded in, set once per sub-step)
    if (G8_ABL & 4) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int j = 0; j < NT; ++j) asm volatile("" : "=v"(bf[s][j]));
#pragma unroll
        for (int i = 0; i < MT; ++i) asm volatile("" : "=v"(af[s][i]));
      }
    }
    auto read_frag = [&](int r, int s) {   // fragment read r of a sub-step: r < NT weights, then activations
      if (G8_ABL & 4) return;
      if ((G8_ABL & 64) && abl_c > c_begin) return;   // (as bit 32) fragment reads in the first chunk only
      if (r < NT) bf[s][r] = lds_read16_off<2048>(wb, r);
      else af[s][r - NT] = lds_read16(s ? (a0[r - NT] ^ 64) : a0[r - NT]);
    };
    // Pieces of a tap, oldest first: the next chunk's halo piece, the weight pieces only some waves have (k >= NI_W / 8), then the NI_W / 8 every wave
    // has.  No run-time counting: past the last chunk the refill re-reads the last chunk's stage (never read again), so every wave issues at least NWALL
    // pieces per tap and the wait in front of the barrier is the immediate vmcnt(NWALL) -- a branch between MFMAs costs ~25 cycles of the stream (the
    // counted form had ~10 per tap: the switch over the count, the conditions of every piece).
    constexpr int NWALL = NI_W / 8;
    uint32_t hmask = 0, wmask = 0;   // wave-uniform: bit tap: this wave has halo piece `tap`; bit k: weight piece k
#pragma unroll
    for (int q = 0; q < MAXHI; ++q) hmask |= (wave + 8 * q < NI_H ? 1u : 0u) << q;
#pragma unroll
    for (int k = 0; k < WI; ++k) wmask |= (wave + 8 * k < NI_W ? 1u : 0u) << k;
    hmask = __builtin_amdgcn_readfirstlane(hmask), wmask = __builtin_amdgcn_readfirstlane(wmask);
    auto issue_piece = [&](int q, int lc, int ltap, int lslot, int c, int tap) {   // q: 0 the halo piece, then weight pieces WI - 1 .. 0
      if (G8_ABL & 2) return;
      if ((G8_ABL & 32) && c > c_begin) return;   // (timing builds with REAL operand data: the matrix pipe's clock depends on it) pieces in the first chunk only
      if (q == 0) {
        if (tap < MAXHI && ((hmask >> tap) & 1)) {
          const int j = wave + 8 * tap, cn = min(c + 1, nchunk - 1);
          const uint32_t ho = h_off[tap < MAXHI ? tap : 0];
          const char* srcp = ho != ~0u ? (const char*)p.A + ho + (int64_t)cn * (BK * 2) : (const char*)zero;
          glds16_asm(srcp, as_addr + (uint32_t)(((c + 1) & 1) * a_elems + j * 8 * BK) * 2);
        }
      } else {
        const int k = WI - q, j = wave + 8 * k;
        if (k < NWALL || ((wmask >> k) & 1))
          glds16_asm_so((const char*)p.W + (int64_t)((wtap0 + wtapd * ltap) * kpt + min(lc, nchunk - 1)) * BK * 2, w_off[k], ws_addr + (uint32_t)(lslot * (BN * BK) + j * 8 * BK) * 2);
      }
    };
    constexpr int NR = MT + NT, NM = MT * NT;
    frag_rows(c_begin, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < NR; ++r) read_frag(r, 0);
    __builtin_amdgcn_sched_barrier(0);
    for (int c = c_begin; c < nchunk; ++c) {
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) {
        const int wslot = tap % NS;
        if (G8_ABL & 64) abl_c = c;
        // ---- sub-step 0
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wb = (bb0 ^ 64u) + (uint32_t)(wslot * (BN * BK * 2));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            if (!(G8_ABL & 1)) acc[i][j] = TB_MFMA_16x16x32(bf[0][j], af[0][i], acc[i][j]);
            else asm volatile("" ::"v"(bf[0][j]), "v"(af[0][i]));
            const int mi = i * NT + j;
#pragma unroll
            for (int r = 0; r < NR; ++r)
              if (mi == (r * (NM - 4)) / NR) {   // spread over all but the last MFMAs
                __builtin_amdgcn_sched_barrier(0);
                read_frag(r, 1);
                __builtin_amdgcn_sched_barrier(0);
              }
          }
        __builtin_amdgcn_sched_barrier(0);
        // ---- sub-step 1
        const int ltap = (tap + NS) % TAPS, lc = c + (tap + NS) / TAPS;   // the stage that refills this tap's slot
        if ((G8_ABL & 2) || ((G8_ABL & 32) && c > c_begin)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else wait_vmcnt(NWALL);   // in flight: the previous tap's youngest pieces; landed (this wave's share): tap + 1's weights, every halo piece before them
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        frag_rows(tap + 1 < TAPS ? c : c + 1, tap + 1 < TAPS ? tap + 1 : 0);
        wb = bb0 + (uint32_t)(((tap + 1) % NS) * (BN * BK * 2));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            if (!(G8_ABL & 1)) acc[i][j] = TB_MFMA_16x16x32(bf[1][j], af[1][i], acc[i][j]);
            else asm volatile("" ::"v"(bf[1][j]), "v"(af[1][i]));
            const int mi = i * NT + j;
#pragma unroll
            for (int k = 0; k < NSLOT; ++k)
              if (mi == k) {   // the refill first: its latency is the ring's lead
                __builtin_amdgcn_sched_barrier(0);
                issue_piece(k, lc, ltap, wslot, c, tap);
                __builtin_amdgcn_sched_barrier(0);
              }
#pragma unroll
            for (int r = 0; r < NR; ++r)
              if (mi == NSLOT + (r * (NM - 4 - NSLOT)) / NR) {
                __builtin_amdgcn_sched_barrier(0);
                read_frag(r, 0);
                __builtin_amdgcn_sched_barrier(0);
              }
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else {
  if (group == 1) asm volatile("s_barrier" ::: "memory");  // phase shift of the second wave group
  int lin_slot = 0, lin_lslot = (NS - 1) % NS;  // Linear: ring slots of the current / the loaded stage (conv: tap % NS, compile-time)
  for (int c = c_begin; c < nchunk; ++c) {
    // SUB: halo position of the 2 x 2 window's first tap -- forward: (py, px) of the output class; dgrad: (1 - py, 1 - px) of the chunk's view
    int sub_shift0 = 0;
    if constexpr (SUB != 0) {
      const int sub_oy = SUB == 1 ? (cls >> 1) : 1 - (sv_view_c >> 1), sub_ox = SUB == 1 ? (cls & 1) : 1 - (sv_view_c & 1);
      sub_shift0 = sub_oy * HC + sub_ox;
    }
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      constexpr int dummy_ = 0;
      (void)dummy_;
      const int ltap = (tap + NS - 1) % TAPS, lc = c + (tap + NS - 1) / TAPS;
      constexpr bool RT_RING = !CONV || SUB != 0 || NS == 4;   // run-time ring counters
      const int wslot = RT_RING ? lin_slot : tap % NS, lslot = RT_RING ? lin_lslot : (tap + NS - 1) % NS;
      const int cnt_step = NS == 2 ? 0 : stage_count(lc, c, tap);
      int shift = CONV ? (tap / 3) * HC + (tap % 3) : 0;
      if constexpr (SUB != 0) shift = sub_shift0 + (tap >> 1) * HC + (tap & 1);
      const uint32_t ab_addr = as_addr + (CONV ? (c & 1) : wslot) * (a_elems * 2) + shift * 128;
      const uint32_t wb_addr = ws_addr + wslot * (BN * BK * 2);
      f16x8 af[2][MT], bf[2][NT];
      if (G8_ABL & 4) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
          for (int j = 0; j < NT; ++j) asm volatile("" : "=v"(bf[s][j]));
#pragma unroll
          for (int i = 0; i < MT; ++i) asm volatile("" : "=v"(af[s][i]));
        }
      }
      uint32_t a0[MT];
      const uint32_t b0 = wb_addr + boff0, b1 = wb_addr + boff1;
#pragma unroll
      for (int h = 0; h < HALVES; ++h) {
        // ---- LOAD phase (fragment reads are inline asm: the compiler neither waits for them nor moves them; the wait is below)
        G8_PF(-1)
        __builtin_amdgcn_sched_barrier(0);
        if (h == 0) {
#pragma unroll
          for (int i = 0; i < MT; ++i) a0[i] = ab_addr + arow128[i] + ((lq ^ ((arow0[i] + shift) & 7)) << 4);
        }
        if constexpr (KH == 2) {   // this wave's half of the k-step only: sub-step kh
#pragma unroll
          for (int j = 0; j < NT; ++j)
            if (!(G8_ABL & 4)) bf[0][j] = lds_read16_off<2048>(kh ? b1 : b0, j);
#pragma unroll
          for (int i = 0; i < MT; ++i)
            if (!(G8_ABL & 4)) af[0][i] = lds_read16(kh ? (a0[i] ^ 64) : a0[i]);
        } else {
#pragma unroll
        for (int q = 0; q < SPH; ++q) {
          const int s = h * SPH + q;
#pragma unroll
          for (int j = 0; j < NT; ++j)
            if (!(G8_ABL & 4)) bf[s][j] = lds_read16_off<2048>(s ? b1 : b0, j);
#pragma unroll
          for (int i = 0; i < MT; ++i)
            if (!(G8_ABL & 4)) af[s][i] = lds_read16(s ? (a0[i] ^ 64) : a0[i]);
        }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!DMAC && !DMAC2 && !DMACH) {
#pragma unroll
          for (int k = 0; k < NSLOT; ++k)
            if ((h == 0) == (k < SLOTS0) && !(DMACC && ((G8_DMACC_MASK >> k) & 1))) issue_slot(k, lc, ltap, lslot, c, tap, (tap + NS - 1) / TAPS);
        }
        int n_ld = 0;   // DMAC: pieces of this step issued here, in the LOAD phase (G8_DMAC_L slots)
        if constexpr (DMAC && G8_DMAC_L > 0 && MT * NT >= 10 && WN == 2) {
#pragma unroll
          for (int k = 0; k < NSLOT && k < G8_DMAC_L; ++k) {
            issue_slot(k, lc, ltap, lslot, c, tap, 0);
            n_ld += (lc < nchunk && (k < WI ? wave + 8 * k < NI_W : wave + 8 * (k - WI) < NI_H)) ? 1 : 0;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        // after the step's last issue: everything except THIS step's loads has landed (this wave's part), i.e. the next step's stage
        if (h == HALVES - 1) {
          // (NS >= 4, Linear: the stage two steps ahead may stay in flight as well -- what must have landed is the NEXT step's stage)
          if constexpr (DMACH) {
            wait_vmcnt(cnt_hist);
          } else if constexpr (DMAC) {
            wait_vmcnt(cnt_hist + n_ld);      // (this step's pieces go out in the COMPUTE phase below; in flight: the previous step's)
          } else if constexpr (DMACC) {
            int n0 = 0;                // in flight: this step's pieces issued so far (all but the second compute phase's)
#pragma unroll
            for (int k = 0; k < NSLOT; ++k)
              if (k < SLOTS0 || !((G8_DMACC_MASK >> k) & 1)) n0 += conv_slot_issues(k, lc, c, tap);
            wait_vmcnt(n0);
          } else {
            wait_vmcnt(!CONV && NS >= 4 ? cnt_step + cnt_hist : cnt_step);
            cnt_hist = cnt_step;
          }
        }
#if G8_PROF
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        G8_PF(0)
        asm volatile("s_barrier" ::: "memory");
        G8_PF(1)
#else
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
        __builtin_amdgcn_sched_barrier(0);
        // ---- COMPUTE phase
        constexpr int NMF = (KH == 2 ? 1 : SPH) * MT * NT;   // MFMAs of the phase; DMAC: piece k goes out behind MFMA (k + 1) NMF / (NSLOT + 1)
#pragma unroll
        for (int q = 0; q < (KH == 2 ? 1 : SPH); ++q) {
          const int s = KH == 2 ? 0 : h * SPH + q;
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {  // transposed: rows of D = output columns (from W), columns of D = output rows (from A)
              if (!(G8_ABL & 1)) acc[i][j] = TB_MFMA_16x16x32(bf[s][j], af[s][i], acc[i][j]);
              else asm volatile("" ::"v"(bf[s][j]), "v"(af[s][i]));
              if constexpr (DMACC) {   // this half's slots ((h == 0) == (k < SLOTS0)), spread over the phase's MT NT SPH MFMAs
                constexpr int NMFc = SPH * MT * NT;
                const int mi = (q * MT + i) * NT + j + 1;
                const int kf = h == 0 ? 0 : SLOTS0, kn = h == 0 ? SLOTS0 : NSLOT - SLOTS0;
#pragma unroll
                for (int k = 0; k < NSLOT; ++k)
                  if (k >= kf && k < kf + kn && ((G8_DMACC_MASK >> k) & 1) && mi == ((k - kf + 1) * NMFc) / (kn + 1)) {
                    __builtin_amdgcn_sched_barrier(0);
                    issue_conv_slot_asm(k, lc, ltap, lslot, c, tap);
                    __builtin_amdgcn_sched_barrier(0);
                  }
              }
              if constexpr (DMACH) {
                constexpr int NMFh = SPH * MT * NT;
                const int mi = (q * MT + i) * NT + j + 1;
#pragma unroll
                for (int k = 0; k < NSLOT; ++k)
                  if (mi == ((k + 1) * NMFh) / (NSLOT + 1)) {
                    __builtin_amdgcn_sched_barrier(0);
                    issue_conv_slot_asm(k, lc, ltap, lslot, c, tap);
                    __builtin_amdgcn_sched_barrier(0);
                  }
              }
              if constexpr (DMAC2) {   // first compute phase: every piece of the next stage
                constexpr int NMF2 = SPH * MT * NT;
                const int mi = (q * MT + i) * NT + j + 1;
#pragma unroll
                for (int k = 0; k < NSLOT; ++k)
                  if (h == 0 && mi == ((k + 1) * NMF2) / (NSLOT + 1)) {
                    __builtin_amdgcn_sched_barrier(0);
                    issue_slot_asm(k, lc, lslot);
                    __builtin_amdgcn_sched_barrier(0);
                  }
              }
              if constexpr (DMAC) {
                const int mi = (q * MT + i) * NT + j + 1;
                constexpr int KF = (G8_DMAC_L > 0 && MT * NT >= 10 && WN == 2) ? (G8_DMAC_L < NSLOT ? G8_DMAC_L : NSLOT) : 0;   // slots already issued in the LOAD phase
#pragma unroll
                for (int k = KF; k < NSLOT; ++k)
                  if (mi == ((k - KF + 1) * NMF) / (NSLOT - KF + 1)) {
                    __builtin_amdgcn_sched_barrier(0);
                    issue_slot_asm(k, lc, lslot);
                    __builtin_amdgcn_sched_barrier(0);
                  }
              }
            }
        }
        if constexpr (DMAC || DMACH) cnt_hist = cnt_step;
        __builtin_amdgcn_sched_barrier(0);
        G8_PF(2)
        asm volatile("s_barrier" ::: "memory");
        G8_PF(3)
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!CONV || SUB || NS == 4) {
        lin_slot = lin_slot + 1 == NS ? 0 : lin_slot + 1;
        lin_lslot = lin_lslot + 1 == NS ? 0 : lin_lslot + 1;
      }
    }
    if (SUB == 2) {  // (view, channel chunk) of the next iteration and of the one after it
      sv_view_c = sv_view_n, sv_lc_c = sv_lc_n;
      if (++sv_lc_n == kpt) sv_lc_n = 0, ++sv_view_n;
    }
  }
  if (group == 0) asm volatile("s_barrier" ::: "memory");  // pairs with the second group's last barrier
  }
#if G8_PROF
  if (dbg && blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256))
    for (int k = 0; k < 4; ++k) dbg[16 + group * 4 + k] = pf_sum[k];
#endif
  G8_STAMP(2)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave is done with the operand buffers: the epilogue reuses the LDS
  G8_STAMP(6)

  // ---- epilogue: accumulators -> padded fp32 tile in LDS -> (row, 8 columns) units with 16-byte global accesses.  A thread keeps ONE
  // column group (bias loaded once) and walks rows; all residual / auxiliary loads of its units are issued before the arithmetic.
  float* Cs = reinterpret_cast<float*>(smem_raw);
  if constexpr (KH == 2) {   // the upper k-half's accumulators go through the staging tile into the lower half's waves (host: one staging pass)
    static_assert(G8Epi<BM, BN>::PASSES == 1, "k-halves are added through a whole-tile staging pass");
    if (kh == 1) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          *(f32x4_t*)(Cs + ((wm * MT + i) * 16 + l15) * G8Epi<BM, BN>::LDC + (wn * NT + j) * 16 + 4 * lq) = acc[i][j];
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (kh == 0) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const f32x4_t o = *(const f32x4_t*)(Cs + ((wm * MT + i) * 16 + l15) * G8Epi<BM, BN>::LDC + (wn * NT + j) * 16 + 4 * lq);
          acc[i][j] += o;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  if (Ssplit > 1) {  // raw fp32 partial of this k-slice (Linear tiles: wshift = 30, so the row of tile row r is m0 + r)
    float* const dst0 = ws + (int64_t)slice * p.M * npad + n;
    const int64_t Mtot = p.M;
#pragma unroll 1
    for (int pass = 0; pass < PASSES; ++pass) {
      const int rp = pass * PR;
      if (stager && (PASSES == 1 || (wm * MT * 16) / PR == pass)) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            *(f32x4_t*)(Cs + ((wm * MT + i) * 16 + l15 - rp) * LDC + (wn * NT + j) * 16 + 4 * lq) = acc[i][j];
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (rslot < TPR) {
#pragma unroll
        for (int it = 0; it < NU; ++it) {
          const int row = rslot + it * TPR, r = rp + row;
          const int64_t m = m0 + (int64_t)(r >> wshift) * W + (r & (TW - 1));
          if (row < PR && m < Mtot) {
            float* dst = dst0 + m * npad;
            *(f32x4_t*)dst = *(const f32x4_t*)(Cs + row * LDC + cg * 8);
            *(f32x4_t*)(dst + 4) = *(const f32x4_t*)(Cs + row * LDC + cg * 8 + 4);
          }
        }
      }
      if (pass + 1 < PASSES) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    return;
  }
  const EpiFlags ef = epi_flags(p);
  const bool pre_r = PRE_R && p.R && ef.r_vec && p.r_dtype == TB_F16 && (p.act == TB_ACT_NONE || p.act == TB_ACT_SILU || ln_epi);
  if (!CONV && p.act != TB_ACT_GEGLU) {
    const f32x4_t u0 = *(const f32x4_t*)(bias_s + cg * 8), u1 = *(const f32x4_t*)(bias_s + cg * 8 + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) b8[e] = u0[e], b8[4 + e] = u1[e];
  }
  // Fast path (every UNet launch that reaches this kernel): fp16 output and residual with 16-byte accesses, activation none / SiLU, and a
  // row bias (time embedding) that is constant over the tile.  Everything it needs from the descriptor is copied into locals FIRST: read
  // through `p` inside the unit loop the fields were re-fetched from the kernarg segment (s_load + wait, a few hundred cycles each, ten per
  // unit) -- the generic epilogue8 path cost 8 us of a 23 us 32768x320x320 launch, against 3 us for the stores themselves.
  if (!PRE_B) {
    const int64_t m_last = CONV ? m0 + ((int64_t)(R - 1) * W + TW - 1) : m0 + BM - 1;
    const bool rb_uniform = rb_is_uniform(m_last);
    fast = p.c_dtype == TB_F16 && ef.c_vec && (!p.R || (ef.r_vec && p.r_dtype == TB_F16)) && (p.act == TB_ACT_NONE || p.act == TB_ACT_SILU) &&
           !p.C2 && rb_uniform;
    if (CONV) epi_load_bias8(p, n, b8);
  }
  // GEGLU forward (ff.net.0.proj, packed [h32 | g32] column blocks): unit = (row, 8 gate outputs); stores the fp16 projections (C2, for the
  // backward) and h * gelu(g).  Host guarantees: Linear tile (BN a multiple of 64), no residual / row bias, fp16 vector-aligned outputs.
  if (!CONV && p.act == TB_ACT_GEGLU) {
    constexpr int UG = BN / 16, TG = 512 / UG, NG = (PR + TG - 1) / TG;
    const int og = t % UG, rs = t / UG;
    const int hcol = (og >> 2) * 64 + (og & 3) * 8;           // tile-local packed column of h; g is + 32
    const float alpha = p.alpha;
    const int64_t nh = n0 + hcol, ldc = p.ldc, ldc2 = p.ldc2, Mtot = p.M;
    float bh[8], bg[8], ch[8], cgv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      bh[e] = bias_s[hcol + e];
      bg[e] = bias_s[hcol + 32 + e];
      ch[e] = lnf ? bias_s[BN + hcol + e] : 0.f;         // folded LayerNorm: c1 of the h / g columns
      cgv[e] = lnf ? bias_s[BN + hcol + 32 + e] : 0.f;
    }
    f16* const C2g = p.C2 ? (f16*)p.C2 + nh : nullptr;
    f16* const Cg = (f16*)p.C + (n0 >> 1) + (og >> 2) * 32 + (og & 3) * 8;
    G8_STAMP(7)
#pragma unroll 1
    for (int pass = 0; pass < PASSES; ++pass) {
      const int rp = pass * PR;
      if (stager && (PASSES == 1 || (wm * MT * 16) / PR == pass)) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            *(f32x4_t*)(Cs + ((wm * MT + i) * 16 + l15 - rp) * LDC + (wn * NT + j) * 16 + 4 * lq) = acc[i][j];
      }
      if (pass == 0) { G8_STAMP(4) }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (pass == 0) { G8_STAMP(5) }
      if (rs < TG) {
#pragma unroll
        for (int it = 0; it < NG; ++it) {
          const int row = rs + it * TG;
          const int64_t m = m0 + rp + row;
          if (row < PR && m < Mtot) {
            const float* cr = Cs + row * LDC + hcol;
            const f32x4_t h0 = *(const f32x4_t*)cr, h1 = *(const f32x4_t*)(cr + 4), g0 = *(const f32x4_t*)(cr + 32), g1 = *(const f32x4_t*)(cr + 36);
            f16x8 oh, og8, oo;
            const f32x2r_t rs = lnf ? rowst_s[rp + row] : f32x2r_t{0.f, 1.f};   // (mean, rstd); no fold: v = 1 * (acc - 0 * 0) + b
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              oh[e] = (f16)(rs[1] * (alpha * (e < 4 ? h0[e] : h1[e - 4]) - rs[0] * ch[e]) + bh[e]);
              og8[e] = (f16)(rs[1] * (alpha * (e < 4 ? g0[e] : g1[e - 4]) - rs[0] * cgv[e]) + bg[e]);
              oo[e] = (f16)((float)oh[e] * ((G8_ABL & 16) ? (float)og8[e] : gelu_erf_f((float)og8[e])));  // gate on the fp16-rounded projections, as an fp16 module would
            }
            if ((G8_ABL & 8) && alpha != 12345.f) continue;  // profiling: no global stores
            if (C2g) {
              G8_ST((f16x8*)(C2g + m * ldc2), oh);
              G8_ST((f16x8*)(C2g + m * ldc2 + 32), og8);
            }
            G8_ST((f16x8*)(Cg + m * ldc), oo);
          }
        }
      }
      if (pass + 1 < PASSES) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    G8_STAMP(3)
    return;
  }
  // GEGLU backward fused into the ff.net.2 dgrad GEMM: v = d(gated)[m, n..n+7]; C2 = packed pre-gate projections [M, 2N];
  // C = d(proj) [M, 2N] in the same packing:  d h = v gelu(g),  d g = v h gelu'(g)
  if (!CONV && p.act == TB_ACT_GEGLU_GRAD) {
    const float alpha = p.alpha;
    const int64_t pc = (n >> 5) * 64 + (n & 31), ldc = p.ldc, ldc2 = p.ldc2, Mtot = p.M;
    const f16* const C2g = (const f16*)p.C2 + pc;
    f16* const Cg = (f16*)p.C + pc;
#pragma unroll 1
    for (int pass = 0; pass < PASSES; ++pass) {
      const int rp = pass * PR;
      if (stager && (PASSES == 1 || (wm * MT * 16) / PR == pass)) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            *(f32x4_t*)(Cs + ((wm * MT + i) * 16 + l15 - rp) * LDC + (wn * NT + j) * 16 + 4 * lq) = acc[i][j];
      }
      f16x8 hv[NU], gv[NU];  // the pre-gate projections of the pass's units are in flight across the staging barrier
      if (rslot < TPR) {
#pragma unroll
        for (int it = 0; it < NU; ++it) {
          const int64_t m = min((int64_t)(m0 + rp + min(rslot + it * TPR, PR - 1)), Mtot - 1);
          hv[it] = *(const f16x8*)(C2g + m * ldc2);
          gv[it] = *(const f16x8*)(C2g + m * ldc2 + 32);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (rslot < TPR) {
#pragma unroll
        for (int it = 0; it < NU; ++it) {
          const int row = rslot + it * TPR;
          const int64_t m = m0 + rp + row;
          if (row < PR && m < Mtot) {
            const f32x4_t c0 = *(const f32x4_t*)(Cs + row * LDC + cg * 8), c1 = *(const f32x4_t*)(Cs + row * LDC + cg * 8 + 4);
            f16x8 dh, dg;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float v = (e < 4 ? c0[e] : c1[e - 4]) * alpha + b8[e];
              const float g = (float)gv[it][e];
              float ge, dge;
              gelu_erf_both_f(g, ge, dge);
              dh[e] = (f16)(v * ge);
              dg[e] = (f16)(v * (float)hv[it][e] * dge);
            }
            G8_ST((f16x8*)(Cg + m * ldc), dh);
            G8_ST((f16x8*)(Cg + m * ldc + 32), dg);
          }
        }
      }
      if (pass + 1 < PASSES) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    return;
  }
  // LayerNorm fused into the epilogue (Linear tiles that span the output row: BN == N == 320, one 64-row staging pass at a time).
  // Row statistics need all 40 column groups of a row, which sit in 40 different threads: partial sums go through `red` (one float2 per
  // (row, column group)), 8 threads per row add them (five each, then three shuffles), results come back through `rowst`.
  //   LN_FWD: the row written to C is normalised as well: two-pass statistics (mean, then centred squares) over the fp16-ROUNDED outputs,
  //           exactly the values tb_layernorm_fwd would read back; C2 = LN(row), ln_stats[m] = (mean, rstd).
  //   LN_BWD: tb_layernorm_bwd on the accumulators: g = acc * gamma, xhat from C2 (the LayerNorm's input) and ln_stats.
  if constexpr (!CONV && (PR == 64 || PR == 32) && BN == 320) {
    if (ln_epi) {
      typedef __attribute__((ext_vector_type(2))) float f32x2_t;
      const bool fwd = p.act == TB_ACT_LN_FWD;
      const float alpha = p.alpha, inv_n = 1.f / (float)BN, eps = p.ln_eps;
      f16* const Cg = (f16*)p.C + n;
      const f16* const Rg = p.R ? (const f16*)p.R + n : nullptr;
      f16* const C2g = (f16*)p.C2 + n;
      const int64_t ldc = p.ldc, ldr = p.ldr, ldc2 = p.ldc2, Mtot = p.M;
      float* const stats_g = p.ln_stats;
      f32x2_t* const red = reinterpret_cast<f32x2_t*>(smem_raw + EPB);   // [PR][UPR]
      f32x2_t* const rowst = red + PR * UPR;                             // [PR]
      float gm[8], bt[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) gm[e] = bias_s[BN + cg * 8 + e], bt[e] = bias_s[2 * BN + cg * 8 + e];
      const int rrow = t >> 3, rj = t & 7;  // reducer role: row (threads with rrow >= PR sit out: whole waves), five-partial slice
      auto reduce_rows = [&]() {  // sums the UPR partials of this thread's row; leaves the (x, y) totals in all 8 lanes of the row's group
        f32x2_t a = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < UPR / 8; ++k) {
          const f32x2_t v = red[rrow * UPR + rj * (UPR / 8) + k];
          a[0] += v[0], a[1] += v[1];
        }
#pragma unroll
        for (int sft = 1; sft < 8; sft <<= 1) {
          a[0] += __shfl_xor(a[0], sft, 64);
          a[1] += __shfl_xor(a[1], sft, 64);
        }
        return a;
      };
      f16x8 rv1[NU];
      if (PASSES == 2 && Rg && rslot < TPR) {
#pragma unroll
        for (int it = 0; it < NU; ++it) {
          const int64_t m = min(m0 + PR + min(rslot + it * TPR, PR - 1), Mtot - 1);
          rv1[it] = *(const f16x8*)(Rg + m * ldr);
        }
      }
#pragma unroll
      for (int pass = 0; pass < PASSES; ++pass) {
        const int rp = pass * PR;
        if (stager && (PASSES == 1 || (wm * MT * 16) / PR == pass)) {
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
              *(f32x4_t*)(Cs + ((wm * MT + i) * 16 + l15 - rp) * LDC + (wn * NT + j) * 16 + 4 * lq) = acc[i][j];
        }
        f16x8 rv[NU], xv[NU];
        f32x2_t st[NU];
        if (rslot < TPR) {
#pragma unroll
          for (int it = 0; it < NU; ++it) {
            const int64_t m = min(m0 + rp + min(rslot + it * TPR, PR - 1), Mtot - 1);
            if (pass == 1 || pre_r) rv[it] = pass ? rv1[it] : rv0[PRE_R ? it : 0];
            else if (Rg) rv[it] = *(const f16x8*)(Rg + m * ldr);
            if (!fwd) {
              xv[it] = *(const f16x8*)(C2g + m * ldc2);
              st[it] = *(const f32x2_t*)(stats_g + 2 * m);
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (fwd) {
          f16x8 o[NU];
          // ---- phase 1: the output row (stored), partial row sums of the rounded values
          if (rslot < TPR) {
#pragma unroll
            for (int it = 0; it < NU; ++it) {
              const int row = rslot + it * TPR;
              const int64_t m = m0 + rp + row;
              if (row < PR) {
                const f32x4_t c0 = *(const f32x4_t*)(Cs + row * LDC + cg * 8), c1 = *(const f32x4_t*)(Cs + row * LDC + cg * 8 + 4);
                float sum = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                  float x = (e < 4 ? c0[e] : c1[e - 4]) * alpha + b8[e];
                  if (Rg) x += (float)rv[it][e];
                  o[it][e] = (f16)x;
                  sum += (float)o[it][e];
                }
                if (m < Mtot) *(f16x8*)(Cg + m * ldc) = o[it];
                red[row * UPR + cg] = f32x2_t{sum, 0.f};
              }
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
          if (rrow < PR) {
            const f32x2_t a = reduce_rows();
            if (rj == 0) rowst[rrow] = f32x2_t{a[0] * inv_n, 0.f};
          }
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
          // ---- phase 2: centred squares
          if (rslot < TPR) {
#pragma unroll
            for (int it = 0; it < NU; ++it) {
              const int row = rslot + it * TPR;
              if (row < PR) {
                const float mean = rowst[row][0];
                float q = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                  const float d = (float)o[it][e] - mean;
                  q += d * d;
                }
                red[row * UPR + cg] = f32x2_t{q, 0.f};
              }
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
          if (rrow < PR) {
            const f32x2_t a = reduce_rows();
            if (rj == 0) {
              const float mean = rowst[rrow][0], rstd = rsqrtf(a[0] * inv_n + eps);
              rowst[rrow] = f32x2_t{mean, rstd};
              const int64_t m = m0 + rp + rrow;
              if (m < Mtot) *(f32x2_t*)(stats_g + 2 * m) = f32x2_t{mean, rstd};
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
          // ---- phase 3: the normalised row
          if (rslot < TPR) {
#pragma unroll
            for (int it = 0; it < NU; ++it) {
              const int row = rslot + it * TPR;
              const int64_t m = m0 + rp + row;
              if (row < PR && m < Mtot) {
                const f32x2_t ms = rowst[row];
                f16x8 y;
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = (f16)(((float)o[it][e] - ms[0]) * ms[1] * gm[e] + bt[e]);
                *(f16x8*)(C2g + m * ldc2) = y;
              }
            }
          }
        } else {
          // ---- phase 1: partial sums of g and g * xhat
          if (rslot < TPR) {
#pragma unroll
            for (int it = 0; it < NU; ++it) {
              const int row = rslot + it * TPR;
              if (row < PR) {
                const f32x4_t c0 = *(const f32x4_t*)(Cs + row * LDC + cg * 8), c1 = *(const f32x4_t*)(Cs + row * LDC + cg * 8 + 4);
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                  const float g = (e < 4 ? c0[e] : c1[e - 4]) * alpha * gm[e];
                  const float xh = ((float)xv[it][e] - st[it][0]) * st[it][1];
                  s1 += g;
                  s2 += g * xh;
                }
                red[row * UPR + cg] = f32x2_t{s1, s2};
              }
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
          if (rrow < PR) {
            const f32x2_t a = reduce_rows();
            if (rj == 0) rowst[rrow] = f32x2_t{a[0] * inv_n, a[1] * inv_n};
          }
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
          // ---- phase 2: dx = rstd (g - mean g - xhat mean(g xhat)) + add
          if (rslot < TPR) {
#pragma unroll
            for (int it = 0; it < NU; ++it) {
              const int row = rslot + it * TPR;
              const int64_t m = m0 + rp + row;
              if (row < PR && m < Mtot) {
                const f32x4_t c0 = *(const f32x4_t*)(Cs + row * LDC + cg * 8), c1 = *(const f32x4_t*)(Cs + row * LDC + cg * 8 + 4);
                const f32x2_t ss = rowst[row];
                f16x8 dx;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                  const float g = (e < 4 ? c0[e] : c1[e - 4]) * alpha * gm[e];
                  const float xh = ((float)xv[it][e] - st[it][0]) * st[it][1];
                  float v = st[it][1] * (g - ss[0] - xh * ss[1]);
                  if (Rg) v += (float)rv[it][e];
                  dx[e] = (f16)v;
                }
                *(f16x8*)(Cg + m * ldc) = dx;
              }
            }
          }
        }
        if (pass + 1 < PASSES) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
      return;
    }
  }
  if (fast) {
    const float alpha = p.alpha;
    const bool silu = p.act == TB_ACT_SILU;
    f16* const Cg = (f16*)p.C + n;
    const f16* const Rg = p.R ? (const f16*)p.R + n : nullptr;
    const int64_t ldc = p.ldc, ldr = p.ldr, Mtot = p.M;
    if (!PRE_B && p.rowbias) {  // (PRE_B: folded into b8 before the main loop)
      const float* rb = p.rowbias + rb_group0() * p.ldrb + n;
#pragma unroll
      for (int e = 0; e < 8; ++e) b8[e] += rb[e];
    }
    auto m_row = [&](int r) -> int64_t { return out_row(r); };
    // row statistics for the NEXT Linear's folded LayerNorm (producer, tb_gemm_desc.rs_out): per (row, 8-column unit) partials through `red`
    const bool rs_prod = RS_OK && p.rs_out != nullptr;
    f32x2r_t* const red = reinterpret_cast<f32x2r_t*>(smem_raw + EPB);   // [PR][UPR], behind the staging tile
    f16x8 rv1[NU];  // second pass: its residual rows are requested before the first pass is staged
    if (PASSES == 2 && Rg && rslot < TPR) {
#pragma unroll
      for (int it = 0; it < NU; ++it) {
        const int64_t m = min(m_row(PR + min(rslot + it * TPR, PR - 1)), Mtot - 1);
        rv1[it] = *(const f16x8*)(Rg + m * ldr);
      }
    }
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
      const int rp = pass * PR;
      if (stager && (PASSES == 1 || (wm * MT * 16) / PR == pass)) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const int row = (wm * MT + i) * 16 + l15 - rp;
            const int col = (wn * NT + j) * 16 + 4 * lq;
            *(f32x4_t*)(Cs + row * LDC + col) = acc[i][j];
          }
      }
      auto m_of = [&](int row) -> int64_t { return out_row(rp + row); };
      f16x8 rv[NU];
      if (pass == 1 || pre_r) {
#pragma unroll
        for (int it = 0; it < NU; ++it) rv[it] = pass ? rv1[it] : rv0[PRE_R ? it : 0];
      } else if (Rg && rslot < TPR) {  // conv: not prefetched (no registers to spare across the main loop)
#pragma unroll
        for (int it = 0; it < NU; ++it) {
          const int64_t m = min(m_of(min(rslot + it * TPR, PR - 1)), Mtot - 1);
          rv[it] = *(const f16x8*)(Rg + m * ldr);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // LDS-only: no vmcnt drain (residual loads, previous pass's stores)
      if (pass == 0) { G8_STAMP(5) }
      if (rslot < TPR) {
#pragma unroll
        for (int it = 0; it < NU; ++it) {
          const int row = rslot + it * TPR;
          const int64_t m = m_of(min(row, PR - 1));
          if (row < PR && m < Mtot) {
            const f32x4_t c0 = *(const f32x4_t*)(Cs + row * LDC + cg * 8), c1 = *(const f32x4_t*)(Cs + row * LDC + cg * 8 + 4);
            f16x8 o;
            float ssum = 0.f, ssq = 0.f;
            if (RS_OK && lnf) {
              const f32x2r_t rs = rowst_s[rp + row];
              // c1 of this thread's 8 columns: re-read from the LDS per unit (held in registers across the passes it cost the 128x160 tile its
              // second workgroup per CU: 133 VGPRs)
              const f32x4_t k0 = *(const f32x4_t*)(bias_s + BN + cg * 8), k1 = *(const f32x4_t*)(bias_s + BN + cg * 8 + 4);
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                float x = rs[1] * ((e < 4 ? c0[e] : c1[e - 4]) * alpha - rs[0] * (e < 4 ? k0[e] : k1[e - 4])) + b8[e];
                if (Rg) x += (float)rv[it][e];
                if (silu) x = silu_f(x);
                o[e] = (f16)x;
              }
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                float x = (e < 4 ? c0[e] : c1[e - 4]) * alpha + b8[e];
                if (Rg) x += (float)rv[it][e];
                if (silu) x = silu_f(x);
                o[e] = (f16)x;
                if (RS_OK) ssum += (float)o[e], ssq += (float)o[e] * (float)o[e];
              }
            }
            if (!(G8_ABL & 8) || alpha == 12345.f) G8_ST((f16x8*)(Cg + m * ldc), o);
            if (rs_prod) red[row * UPR + cg] = f32x2r_t{ssum, ssq};
          }
        }
      }
      if constexpr (RS_OK) {
        if (rs_prod) {   // the UPR unit partials of a row -> slot tn of the row's statistics
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
          if (t < PR) {
            const int64_t m = m0 + rp + t;
            float sx = 0.f, sq = 0.f;
#pragma unroll
            for (int k = 0; k < UPR; ++k) {
              const f32x2r_t v = red[t * UPR + k];
              sx += v[0], sq += v[1];
            }
            if (m < Mtot) reinterpret_cast<f32x2r_t*>(p.rs_out)[m * p.rs_ld + tn] = f32x2r_t{sx, sq};
          }
        }
      }
      if (pass == 0) { G8_STAMP(6) }
      if (pass + 1 < PASSES) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      G8_STAMP(3 + pass)
    }
    return;
  }
  if (ln_epi) __builtin_trap();  // (unreachable: every tile the host routes LN epilogues to has the block above)
  // ---- generic path (fp32 output / residual, GELU variants, unaligned rows): rolled loops around the shared epilogue8
#pragma unroll 1
  for (int pass = 0; pass < PASSES; ++pass) {
    const int rp = pass * PR;
    if (stager && (PASSES == 1 || (wm * MT * 16) / PR == pass)) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int row = (wm * MT + i) * 16 + l15 - rp;
          const int col = (wn * NT + j) * 16 + 4 * lq;
          *(f32x4_t*)(Cs + row * LDC + col) = acc[i][j];
        }
    }
    auto m_of = [&](int row) -> int64_t { return out_row(rp + row); };
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (rslot < TPR) {
#pragma unroll 1
      for (int it = 0; it < NU; ++it) {
        const int row = rslot + it * TPR;
        if (row >= PR) break;
        const int64_t m = m_of(row);
        if (m >= p.M) continue;
        float v[8], r8[8];
        epi_load_r8(p, ef, m, n, r8);
        const f16x8 aux = epi_load_aux8(p, ef, m, n);
        const f32x4_t c0 = *(const f32x4_t*)(Cs + row * LDC + cg * 8), c1 = *(const f32x4_t*)(Cs + row * LDC + cg * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = c0[e];
          v[4 + e] = c1[e];
        }
        epilogue8(p, ef, m, n, v, b8, r8, aux);
      }
    }
    if (pass + 1 < PASSES) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
#undef G8_STAMP
}

unsigned long long* g8_dbg = nullptr;  // profiling aid (tb_gemm8_debug): s_memtime stamps of the first and the last block
int g8_enable = 39;  // tb_gemm8_set(bits): 1 = convolutions, 2 = Linear layers, 4 = GEGLU / GEGLU-backward epilogues take the wide-tile path
thread_local int g8_split = 1;  // k-slices of the launch tb_gemm8_try is making on this thread (returned through its out-parameter)
thread_local bool g8_dry = false;   // tb_gemm_lnfold_ok: walk the dispatch rules without launching
int g8_last[7] = {0, 0, 0, 0, 0, 0, 0};  // [0] = 1 when the most recent tb_gemm went through gemm8_kernel<[1], [2], [3], [4], [5]>

template <int WM, int WN, int MT, int NT, bool CONV, int NS, int SUB = 0, int KH = 1>
int launch8(const tb_gemm_desc& d, hipStream_t s, int wshift, int S = 1) {
  constexpr int BM = WM * MT * 16, BN = WN * NT * 16;
  // SUB == 1: the tile grid walks the COARSE map (a quarter of the output rows) once per output class: S = 4 is the class count, not a k-split
  // (Linear: a ragged last row tile is allowed -- the panel loads clamp their rows to M - 1 and every epilogue masks m < M)
  const int tiles_m = CONV ? (int)((SUB == 1 ? d.M / 4 : d.M) / BM) : (int)((d.M + BM - 1) / BM), tiles_n = (int)(d.N / BN);
  const int64_t npad = (d.N + 7) & ~(int64_t)7;
  if (SUB == 1 && S != 4) return TB_EINVAL;
  if (SUB != 1 && S > 1 && (!d.ws || (size_t)S * (size_t)d.M * (size_t)npad * 4 > d.ws_bytes)) return 1;
  int a_rows;
  if (CONV) {
    const int TW = 1 << wshift, R = BM >> wshift;
    a_rows = (R + 2) * (TW + 2);
  } else {
    a_rows = BM;
  }
  const int a_rows8 = (a_rows + 7) & ~7;
  if (CONV && (a_rows8 >> 3) > 8 * 7) return 1;  // more panel load instructions than the kernel issues
  size_t lds = (size_t)(CONV ? 2 : NS) * a_rows8 * 128 + (size_t)NS * BN * 128;
  if (lds < G8Epi<BM, BN>::BYTES) lds = G8Epi<BM, BN>::BYTES;
  if ((d.rs_in || d.rs_out) && (CONV || BM * BN > 128 * 160)) return TB_EINVAL;   // (the kernel's RS_OK)
  if (!CONV) {
    // `red` ([rows per pass][8-column units] float2, the row-statistic partials of tb_gemm_desc.rs_out) sits behind the epilogue's staging tile
    const size_t red_end = G8Epi<BM, BN>::BYTES + (size_t)G8Epi<BM, BN>::PR * (BN / 8) * 8;
    if (d.rs_out && lds < red_end) lds = red_end;
    lds += BN * 12 + BM * 8;  // the tile's bias values (+ LayerNorm gamma / beta of the fused-LN epilogues, c1 of a folded one) + (mean, rstd) per tile row
  }
  if (lds > 160 * 1024) return 1;
  // tb_gemm8_set bit 1048576 (round 6, opt-in): the 256 x 160 / 256 x 128 convolution tiles as ONE instruction stream per wave (see STREAM in the
  // kernel) instead of the LOAD / COMPUTE phases of two barrier-shifted wave groups.  Isolated launches 57.8 -> 54.5 us (320 -> 320 @ 64 x 64) and
  // 162.4 -> 156 us (960 -> 320): 1800 -> 1546 shader cycles per tap against 1304 of MFMAs -- and the clock inside the launch falls from 1.63 to
  // 1.46 GHz: on real operand data these tiles sit at the board's power limit, cycles saved come back as clock.  In the sustained step
  // (scratch/ab_step.py): 27.92 ms with the phases, 27.99 ms with the stream.  Default: the phases.
  constexpr bool CAN_ST = G8_STREAM && CONV && SUB == 0 && MT >= 4 && NT <= 5 && NS == 3 && KH == 1;
  const bool phases = CAN_ST && !(g8_enable & 1048576);
  static bool attr_done = false;
  if (!attr_done && !g8_dry) {
    if (hipFuncSetAttribute((const void*)gemm8_kernel<WM, WN, MT, NT, CONV, NS, SUB, KH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
        hipSuccess)
      return TB_ELAUNCH;
    if constexpr (CAN_ST) {
      if (hipFuncSetAttribute((const void*)gemm8_kernel<WM, WN, MT, NT, CONV, NS, SUB, KH, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024) != hipSuccess)
        return TB_ELAUNCH;
    }
    attr_done = true;
  }
  // XCD cut (see the kernel): fabric bytes = xn * A + (8 / xn) * W over the cuts that divide the tile grid
  int xn = 0;
  if (!(g8_enable & 64)) {
    const double a_bytes = 2.0 * (double)d.M * (CONV ? d.Cin : d.K), w_bytes = 2.0 * (double)d.N * d.K;
    double best = (a_bytes + 8 * w_bytes) / 1.5;  // row panels per XCD (xn = 0 / 1) unless a cut saves a third of the traffic: on the
    for (int c = 2; c <= 8; c *= 2) {             // 32x32 maps, where the two operands are the same size, the 4 x 2 cut measured 8 % slower
      if (tiles_n % c || tiles_m % (8 / c)) continue;
      const double cost = c * a_bytes + (8 / c) * w_bytes;
      if (cost < best) xn = c, best = cost;
    }
  }
  // producer of folded-LayerNorm statistics: slot tn < tiles_n of every row is written, the consumer sums rs_n slots -- both sides must mean the
  // same tile width (it follows mutable dispatch knobs: tb_gemm8_set), and a row of rs_out holds rs_ld slots
  if (d.rs_out && (tiles_n > d.rs_ld || (d.rs_n > 0 && tiles_n != d.rs_n))) return TB_EINVAL;
  G8Magic mg;
  {
    const int W_ = SUB == 1 ? d.Win : d.Wout, H_ = SUB == 1 ? d.Hin : d.Hout;
    const int tiles_x = CONV ? (W_ >> wshift) : 1, R_ = CONV ? (BM >> wshift) : 1;
    mg.S = mg_of(S), mg.tiles_n = mg_of(tiles_n), mg.rn = mg_of(xn > 0 ? tiles_n / xn : 1), mg.xn = mg_of(xn);
    mg.tiles_img = mg_of(CONV ? (H_ / R_) * tiles_x : 1), mg.tiles_x = mg_of(tiles_x), mg.hc = mg_of(CONV ? (1 << wshift) + 2 : 1);
    if ((int64_t)tiles_m * tiles_n * S >= 65536) return 1;   // (the multiply-high division is exact below 2^16)
  }
  if (g8_dry) {
    g8_last[0] = 1, g8_last[1] = WM, g8_last[2] = WN, g8_last[3] = MT, g8_last[4] = NT, g8_last[5] = CONV, g8_last[6] = NS;
    return TB_OK;
  }
  if (phases)
    hipLaunchKernelGGL((gemm8_kernel<WM, WN, MT, NT, CONV, NS, SUB, KH, !CAN_ST>), dim3((unsigned)(tiles_m * tiles_n * S)), dim3(512), lds, s, d, tiles_m,
                       tiles_n, wshift, a_rows8, g8_dbg, S, (float*)d.ws, npad, xn, mg);
  else
    hipLaunchKernelGGL((gemm8_kernel<WM, WN, MT, NT, CONV, NS, SUB, KH>), dim3((unsigned)(tiles_m * tiles_n * S)), dim3(512), lds, s, d, tiles_m, tiles_n,
                       wshift, a_rows8, g8_dbg, S, (float*)d.ws, npad, xn, mg);
  TB_CHECK_LAUNCH();
  g8_split = SUB == 1 ? 1 : S;
  g8_last[0] = 1, g8_last[1] = WM, g8_last[2] = WN, g8_last[3] = MT, g8_last[4] = NT, g8_last[5] = CONV, g8_last[6] = NS;
  return TB_OK;
}

inline int halo_wshift8(int W) {  // tile width: the largest power of two (16..64) dividing W
  int sh = 0;
  while (sh < 6 && !(W & (1 << sh))) ++sh;
  return sh >= 4 ? sh : 0;
}

}  // namespace

void tb_gemm8_clear_last() { g8_last[0] = 0; }   // (tb_gemm took another kernel family for this launch)

extern "C" int tb_gemm8_set(int v) {
  const int old = g8_enable;
  g8_enable = v;
  return old;
}
extern "C" int tb_gemm8_debug(void* stamps16) {  // device buffer of 16 x u64 (or NULL): s_memtime at start / prologue done / loop done /
  g8_dbg = (unsigned long long*)stamps16;       // epilogue passes done, for the first ([0..7]) and the last ([8..15]) workgroup
  return TB_OK;
}
extern "C" int tb_gemm8_last(int* out5 /* 6 ints */) {
  if (out5)
    for (int i = 0; i < 6; ++i) out5[i] = g8_last[1 + i];
  return g8_last[0];
}

// can the nearest-x2 + conv3x3 pair over a coarse [B, Hc, Wc, Cin] map run as sub-pixel convolutions (tb_gemm_desc.upsample == 2 / 3)?
extern "C" int tb_gemm_subpixel_ok(int B, int Hc, int Wc, int Cin, int N) {
  if (!g8_enable) return 0;                          // the sub-pixel descs exist in this kernel family only (tb_gemm8_set(0): callers keep the 9-tap form)
  if (B <= 0 || Hc <= 0 || Wc <= 0 || Cin <= 0 || Cin % 64 || N <= 0 || N % 80) return 0;
  int sh = 0;
  while (sh < 6 && !(Wc & (1 << sh))) ++sh;
  if (sh < 4) return 0;                              // tile width: the largest power of two (16 .. 64) dividing Wc
  const int TW = 1 << sh;
  if (Wc % TW || 256 % TW || Hc % (256 / TW)) return 0;
  const int64_t Mc = (int64_t)B * Hc * Wc;
  if (Mc % 256) return 0;
  if ((Mc / 256) * (N / 80) * 4 < 128) return 0;     // not enough tiles to be worth it
  return 1;
}

// LayerNorm folded into the consuming Linear (tb_gemm_desc.rs_out / rs_in) for a transformer block of width C over M rows: 1 when the producers of
// the residual stream (proj_in, attn1.to_out, attn2.to_out: M x C x C with a residual) take an 8-wave tile that writes row statistics AND the three
// consumers (qkv M x 3C x C, attn2.to_q M x C x C, the GEGLU projection M x 8C x C) take a tile whose epilogue applies the fold; *slots = the
// producers' column tiles per row (rs_n).  Walks the dispatch rules themselves (dry run), so it cannot drift from them.
static int gemm8_try(const tb_gemm_desc& d, hipStream_t s);
extern "C" int tb_gemm_lnfold_ok(int64_t M, int64_t C, int* slots) {
  if (M <= 0 || C <= 0 || C % 64 || M % 128) return 0;
  char* const fake = (char*)(uintptr_t)0x10000;   // never dereferenced: the dry run stops in front of the launch
  tb_gemm_desc d = {};
  d.M = M, d.K = d.K1 = C, d.lda = C, d.ldw = C, d.a_mode = TB_A_LINEAR, d.alpha = 1.f, d.A = fake, d.W = fake, d.C = fake, d.c_dtype = TB_F16;
  g8_dry = true;
  int ok = 1, nslots = 0;
  {  // producer
    tb_gemm_desc p = d;
    p.N = C, p.ldc = C, p.R = fake, p.ldr = C, p.r_dtype = TB_F16, p.bias = (const float*)fake, p.rs_out = (float*)fake, p.rs_ld = 16;
    if (gemm8_try(p, nullptr) != TB_OK) ok = 0;
    else nslots = (int)(C / (g8_last[2] * g8_last[4] * 16));
    if (nslots < 1 || nslots > 16) ok = 0;
  }
  const int64_t ns[3] = {3 * C, C, 8 * C};
  for (int i = 0; i < 3 && ok; ++i) {
    tb_gemm_desc c = d;
    c.N = ns[i], c.ldc = i == 2 ? 4 * C : ns[i], c.rs_in = (const float*)fake, c.rs_ld = 16, c.rs_n = nslots, c.ln_gamma = (const float*)fake;
    if (i == 2) c.act = TB_ACT_GEGLU, c.C2 = fake, c.ldc2 = ns[i], c.bias = (const float*)fake;
    const int r = gemm8_try(c, nullptr);
    if (r != TB_OK && !(r == 1 && i != 2)) ok = 0;   // (1 = the 4-wave kernel's lean epilogue takes it: plain Linear only)
  }
  g8_dry = false;
  g8_last[0] = 0;
  if (slots) *slots = ok ? nslots : 0;
  return ok;
}

extern "C" int tb_gemm_ln_epilogue_ok(int64_t M, int64_t N, int64_t K) {
  if (N != 320 || K <= 0 || K % 64 || M <= 0) return 0;
  if (M % 128 == 0 && M / 128 >= 200) return 1;  // 128 x 320 tiles
  if (M % 64 == 0 && M / 64 >= 200) return 1;    // 64 x 320 tiles
  return 0;
}

// returns TB_OK when the launch was taken, 1 when the shape is not covered (the caller falls back to gemm.hip), < 0 on error

static int gemm8_try(const tb_gemm_desc& d, hipStream_t s);
// *split_out = k-slices of the launch (> 1: fp32 partials are in d.ws and the caller runs the reducer)
int tb_gemm8_try(const tb_gemm_desc& d, hipStream_t s, int* split_out) {
  const int rc = gemm8_try(d, s);
  if (split_out) *split_out = g8_split;
  return rc;
}
static int gemm8_try(const tb_gemm_desc& d, hipStream_t s) {
  g8_last[0] = 0;
  g8_split = 1;
  if (!g8_enable) return 1;
  if (d.A2 || d.W2) return 1;
  if (d.act == TB_ACT_GEGLU || d.act == TB_ACT_GEGLU_GRAD) {  // lean fused epilogues exist for the aligned fp16 Linear case only
    if (!(g8_enable & 4) || d.a_mode != TB_A_LINEAR || d.R || d.rowbias || d.c_dtype != TB_F16) return 1;
    if (d.ldc % 8 || ((uintptr_t)d.C) % 16) return 1;
    if (d.C2 && (d.ldc2 % 8 || ((uintptr_t)d.C2) % 16)) return 1;
    if (d.act == TB_ACT_GEGLU && (d.N % 640 || ((uintptr_t)d.C) % 16)) return 1;       // whole [h32 | g32] blocks per 320-wide tile
    if (d.act == TB_ACT_GEGLU_GRAD && (!d.C2 || d.N % 32 || d.bias)) return 1;
  }
  if (d.rs_out || d.rs_in) {  // folded-LayerNorm statistics (producer / consumer): the lean fp16 epilogues only -- anything else is the caller's error
    const bool vec = d.c_dtype == TB_F16 && d.ldc % 8 == 0 && ((uintptr_t)d.C) % 16 == 0 &&
                     (!d.R || (d.r_dtype == TB_F16 && d.ldr % 8 == 0 && ((uintptr_t)d.R) % 16 == 0));
    if (d.a_mode != TB_A_LINEAR || d.rowbias || !vec || d.M % 128) return TB_EINVAL;
    if (d.rs_out && (d.act != TB_ACT_NONE || d.C2 || d.rs_in || ((uintptr_t)d.rs_out) % 8)) return TB_EINVAL;
    if (d.rs_in && ((d.act != TB_ACT_NONE && d.act != TB_ACT_GEGLU) || (d.act == TB_ACT_NONE && d.C2) || !d.ln_gamma || d.rs_n < 1 || d.rs_n > 16 ||
                    d.rs_ld < d.rs_n || ((uintptr_t)d.rs_in) % 8 || ((uintptr_t)d.ln_stats) % 8))
      return TB_EINVAL;
  }
  const bool ln_act = d.act == TB_ACT_LN_FWD || d.act == TB_ACT_LN_BWD;
  if (ln_act) {  // only the row-spanning Linear tiles implement these: anything else is the caller's error (tb_gemm_ln_epilogue_ok)
    if (!(g8_enable & 2) || d.a_mode != TB_A_LINEAR || !tb_gemm_ln_epilogue_ok(d.M, d.N, d.K) || d.rowbias || d.c_dtype != TB_F16 || !d.C2 ||
        !d.ln_gamma || !d.ln_stats || (d.act == TB_ACT_LN_FWD && !d.ln_beta) || d.ldc % 8 || ((uintptr_t)d.C) % 16 || d.ldc2 % 8 ||
        ((uintptr_t)d.C2) % 16 || ((uintptr_t)d.ln_stats) % 8 || (d.R && (d.r_dtype != TB_F16 || d.ldr % 8 || ((uintptr_t)d.R) % 16)))
      return TB_EINVAL;
  }
  if (d.K % 64 || d.N % 8) return 1;
  const int64_t lim = (int64_t)1 << 32;
  if (((d.N * (d.upsample == 2 ? 4 : 1) - 1) * d.ldw + d.K) * 2 >= lim) return 1;
  if (d.a_mode == TB_A_CONV3X3 && (d.upsample == 2 || d.upsample == 3)) {  // sub-pixel upsampler convolution (forward / dgrad), see the kernel
    if (!tb_gemm_subpixel_ok(d.B, d.upsample == 2 ? d.Hin : d.Hout, d.upsample == 2 ? d.Win : d.Wout, d.Cin, (int)d.N)) return TB_EINVAL;
    const int Wc = d.upsample == 2 ? d.Win : d.Wout;
    const int wshift = halo_wshift8(Wc);
    const int64_t Mc = d.upsample == 2 ? d.M / 4 : d.M;
    const int64_t t160 = (Mc / 256) * (d.N / 160), t80 = (Mc / 256) * (d.N / 80);
    if (d.upsample == 2) {
      if (d.N % 160 == 0 && t160 * 4 >= 200) return launch8<4, 2, 4, 5, true, 3, 1>(d, s, wshift, 4);
      return launch8<8, 1, 2, 5, true, 3, 1>(d, s, wshift, 4);
    }
    if (d.N % 160 == 0 && t160 >= 200) return launch8<4, 2, 4, 5, true, 3, 2>(d, s, wshift);
    if (t80 >= 200) return launch8<8, 1, 2, 5, true, 3, 2>(d, s, wshift);
    // too few tiles: one VIEW of the fine gradient per k-slice (fp32 partials + the caller's reducer)
    if (d.N % 160 == 0 && t160 * 4 >= 128) return launch8<4, 2, 4, 5, true, 3, 2>(d, s, wshift, 4);
    return launch8<8, 1, 2, 5, true, 3, 2>(d, s, wshift, 4);
  }
  if (d.a_mode == TB_A_CONV3X3) {
    if (!(g8_enable & 1)) return 1;
    if (d.stride != 1 || d.upsample || d.transposed || d.shift || d.Hin != d.Hout || d.Win != d.Wout) return 1;
    const int wshift = halo_wshift8(d.Wout);
    if (!wshift) return 1;
    const int TW = 1 << wshift;
    if (d.Wout % TW) return 1;
    // 256 pixels x 160 channels (one round of >= 200 blocks), else 256 x 80
    if (256 % TW == 0 && d.Hout % (256 / TW) == 0 && d.M % 256 == 0) {
      if (d.N % 160 == 0 && (d.M / 256) * (d.N / 160) >= 200) return launch8<4, 2, 4, 5, true, 3>(d, s, wshift);
      // (524288, round-6 experiment: the 32x32-map convolutions -- 128 tiles of 256 x 160 -- as TWO k-slices of the 256 x 160 tile (the 64 x 80 wave
      //  tile: 9 fragment reads per 20 MFMAs) instead of one round of 256 x 80 tiles (32 x 80 wave tiles: 7 per 10); fp32 partials + the consumer's /
      //  the caller's reduction, as on the 16x16 maps)
      if ((g8_enable & 524288) && (g8_enable & 32) && d.N % 160 == 0 && d.ws) {
        const int tiles = (int)((d.M / 256) * (d.N / 160)), kpt = d.Cin / 64;
        if (tiles >= 100 && tiles < 200 && kpt >= 4) return launch8<4, 2, 4, 5, true, 3>(d, s, wshift, 2);
      }
      if (d.N % 80 == 0 && (d.M / 256) * (d.N / 80) >= 200) {
        if (!(g8_enable & 32768)) {   // 4-slot weight ring with the pieces between the MFMAs (DMACH) where the LDS holds it (maps up to 32 wide)
          const int r = launch8<8, 1, 2, 5, true, 4>(d, s, wshift);
          if (r != 1) return r;
        }
        return launch8<8, 1, 2, 5, true, 3>(d, s, wshift);
      }
      // 256 pixels x 128 channels (wave tiles 64 x 64): the VAE's 128 / 256 / 512-channel convolutions (AutoencoderKL encoder / decoder,
      // train_textboost.py:1036-1037 and log_validation), which the 80-wide wave tiles do not divide
      if (d.N % 128 == 0 && (d.M / 256) * (d.N / 128) >= 200 && !(g8_enable & 1024)) return launch8<4, 2, 4, 4, true, 3>(d, s, wshift);
      // too few tiles for the chip (16x16 maps: 8 x 8 tiles of 256 x 160): split the channel chunks over S workgroups per tile
      // (65536, experiment: the same maps as 256 x 80 tiles -- twice the tiles, so HALF the k-slices: the GroupNorm behind the convolution reads
      //  2 fp32 slices instead of 4 and this launch writes half the partial bytes, against 61 instead of 98 FLOP per operand byte)
      if (d.N % 160 == 0 && (g8_enable & 32) && (g8_enable & 65536)) {
        const int tiles = (int)((d.M / 256) * (d.N / 80)), kpt = d.Cin / 64;
        int S = (230 + tiles - 1) / tiles;
        while (S > 1 && kpt / S < 2) --S;
        if (S > 1 && tiles * S >= 128) {
          const int r = launch8<8, 1, 2, 5, true, 4>(d, s, wshift, S);
          if (r != 1) return r;
          return launch8<8, 1, 2, 5, true, 3>(d, s, wshift, S);
        }
      }
      if (d.N % 160 == 0 && (g8_enable & 32)) {
        const int tiles = (int)((d.M / 256) * (d.N / 160)), kpt = d.Cin / 64;
        int S = (230 + tiles - 1) / tiles;
        while (S > 1 && kpt / S < 2) --S;
        if (S > 1 && tiles * S >= 128) return launch8<4, 2, 4, 5, true, 3>(d, s, wshift, S);
      }
    }
    return 1;
  }
  if (!(g8_enable & 2)) return 1;
  // GEGLU / GEGLU-backward layers (K = 320 .. 1280, N = 8 K or 4 K): 5 .. 20 k-steps in front of an epilogue that moves 3 .. 5 bytes per
  // output, eight or more tiles per CU.  128x128 tiles at 64 KB of LDS put TWO workgroups on a CU, so one's epilogue (stores, gelu) runs
  // under the other's main loop -- with one resident 128x320 tile per CU the two phases simply alternated (21 us per tile, 3 of them MFMA).
  if ((d.act == TB_ACT_GEGLU || d.act == TB_ACT_GEGLU_GRAD) && !(g8_enable & 16) && d.N % 128 == 0 && d.M % 128 == 0 &&
      (d.M / 128) * (d.N / 128) >= 512)
    return launch8<2, 4, 4, 2, false, 2>(d, s, 30);
  // short-K layers wider than one 320-column tile (qkv N = 3 C, K = C): several tiles per CU, each a handful of k-steps between a prologue and
  // an epilogue -- 128x160 tiles at 72 KB keep two workgroups on the CU so those phases overlap (as for the GEGLU layers above)
  if (!(g8_enable & 128) && !ln_act && d.act != TB_ACT_GEGLU && d.act != TB_ACT_GEGLU_GRAD && d.K <= 640 && (d.N > 320 || (g8_enable & 256)) && d.N % 160 == 0 && d.M % 128 == 0 &&
      (d.M / 128) * (d.N / 160) >= ((g8_enable & 512) ? 256 : 512))
    return launch8<4, 2, 2, 5, false, 2>(d, s, 30);
  // One 128 x 80 tile per CU with a 4-stage ring for the 16x16-map Linear layers (M = 2048, N = 1280: 16 x 16 tiles): their launches are bound by
  // the bytes that go through the L2 -> LDS path (scratch/lin_ablate.py, scratch/ubench/ldsdma_bw.hip: ~100 GB/s per CU whatever is in flight) --
  // 2.5 resident 64x64 tiles of the four-wave kernel move 820 KB per CU at K = 1280, this tile 532 KB
  if (!(g8_enable & 2048) && !ln_act && (d.act == TB_ACT_NONE || d.act == TB_ACT_SILU) && d.N % 80 == 0 && d.M % 128 == 0 && d.K >= 640 && d.K <= ((g8_enable & 262144) ? 2560 : 5120)) {
    // K limits (bit 262144 = 2560 for both, as until round 5): 3840 for the 128 x 80 tile (the 16x16-map qkv dgrad, K = 3 C: 50 -> 34 us; at
    // K = 5120 the two-slice route below is the faster one there), 5120 for the 128 x 160 tile (the 32x32-map GEGLU-projection dgrad, 8192 x 640 x
    // 5120, sat on 64 x 320 tiles of the 3-stage ring)
    const int64_t tiles = (d.M / 128) * (d.N / 80);
    if (tiles >= 200 && tiles <= 256 && d.K <= 3840)   // (16384: the 8 x 1 waves of round 4 instead of 4 x 1 x 2 k-halves)
      return (g8_enable & 16384) ? launch8<8, 1, 1, 5, false, 4>(d, s, 30) : launch8<4, 1, 2, 5, false, 4, 0, 2>(d, s, 30);
    // ... and 128 x 160 for the 32x32-map layers (M = 8192, N = 640: 64 x 4 tiles; 369 KB per CU at K = 640 against 491 KB for the 64 x 320 tile)
    const int64_t tiles160 = d.N % 160 == 0 ? (d.M / 128) * (d.N / 160) : 0;
    if (!(g8_enable & 4096) && tiles160 >= 200 && tiles160 <= 256) return launch8<4, 2, 2, 5, false, 4>(d, s, 30);
  }
  // long-K layers of the 16x16 maps (ff.net.2 and the GEGLU-projection dgrad at C = 1280: M = 2048, N = 1280, K = 5120 / 10240): 16 x 8 tiles of
  // 128 x 160 (14 fragment reads per 20 MFMAs) cut into two k-slices = one workgroup per CU; fp32 partials + the caller's reducer
  if (!(g8_enable & 8192) && !ln_act && (d.act == TB_ACT_NONE || d.act == TB_ACT_SILU) && d.N % 160 == 0 && d.M % 128 == 0 && d.K >= 4096 && d.ws) {
    const int64_t tiles160 = (d.M / 128) * (d.N / 160);
    if (tiles160 * 2 >= 200 && tiles160 * 2 <= 256) {
      const int r = launch8<4, 2, 2, 5, false, 4>(d, s, 30, 2);
      if (r != 1) return r;
    }
  }
  // the text encoder's wide layers (CLIP fc1 and the dgrad of fc2: M = 1848 token rows, N = 3072, K = 768): 15 x 24 tiles of 128 x 128, two
  // workgroups per CU, the last row tile ragged -- against 720 two-stage 128 x 64 tiles of the 4-wave kernel (bit 131072)
  if ((g8_enable & 131072) && !ln_act && d.K <= 1024 && d.N % 128 == 0 && d.M >= 1024 && d.M < 4096 && ((d.M + 127) / 128) * (d.N / 128) >= 300)
    return launch8<2, 4, 4, 2, false, 2>(d, s, 30);
  if (d.N % 320) return 1;
  if (ln_act && (d.M / 128) * (d.N / 320) >= 200 && d.M % 128 == 0) return launch8<2, 4, 4, 5, false, 2>(d, s, 30);
  if (d.M % 128 == 0 && (d.M / 128) * (d.N / 320) >= 200 && !(g8_enable & 8)) return launch8<2, 4, 4, 5, false, 2>(d, s, 30);
  // (not the 16x16-map qkv projection, M = 2048 x N = 3840 x K = 1280: 384 of these 64x320 tiles take 51.9 us cold against 41.3 us for one round of
  // 128x128 four-wave tiles -- scratch/shape_sweep.py; the 32x32-map N = 640 layers at M = 8192 are what this tile is for)
  if (d.M % 64 == 0 && (d.M / 64) * (d.N / 320) >= 200 && (d.M >= 4096 || d.N <= 1280)) return launch8<2, 4, 2, 5, false, 3>(d, s, 30);
  return 1;
}
