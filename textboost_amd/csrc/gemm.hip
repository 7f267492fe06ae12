// MFMA GEMM family for gfx950: C[M,N] = A[M,K] * W[N,K]^T with fused epilogues.
//
// One kernel body serves nn.Linear / 1x1 conv (A rows read linearly) and 3x3 conv as implicit GEMM over
// NHWC activations (A rows gathered per tap, incl. stride-2, nearest-x2-upsample-folded and
// stride-2-transposed (dgrad) gathers).  dgrad of every layer is the same kernel on host-pre-transposed
// frozen weights, so no NN/TN variants exist.
//
// Structure (v2):
//  * BMxBNx64 block tile, 256 threads = 4 waves (2x2), each wave (BM/2)x(BN/2) as 32x32x16 f16 MFMA tiles, fp32 acc.
//  * operands go HBM -> LDS directly (global_load_lds_dwordx4, 16 B per lane, no staging VGPRs); the gather (conv taps,
//    ragged rows, zero padding) lives in the per-lane SOURCE address -- out-of-range lanes read a 16-byte zero line.
//  * LDS image is lane-linear per wave instruction (8 rows x 128 B); the bank swizzle chunk ^= (row>>1)&7 is applied to the
//    source address and to the ds_read_b128 fragment reads (cdna guide rule 21), which makes every 16-lane read group hit 16
//    distinct 16-byte slots (conflict free).
//  * two LDS stages, the next k-tile's loads are in flight while the current one is multiplied; one barrier per k-tile.
//  * accumulators are kept TRANSPOSED (D = W_tile * A_tile^T) so each lane owns 4 consecutive output columns of one row:
//    the epilogue reads residuals / writes C with 8-byte (fp16) or 16-byte (fp32) vectors.
//  * blockIdx is remapped so that consecutive tiles of one A row-panel run on the same XCD (shared L2).
#include "common.h"
#include "../../include/textboost_hip.h"

#include "gemm_epi.h"

int tb_lin320_try(const tb_gemm_desc& d, hipStream_t s);  // lin320.hip: activation-stationary tiles for the K = 320 layers of the 64x64 maps; 1 = not covered
void tb_lin320_set(int on);
void tb_gemm8_clear_last();

namespace {

// split-K second pass: C = epilogue(sum_s ws[s][m][n]); ws is fp32 [S][M][Npad] with Npad = N rounded up to 8
// 16 bytes of a partial written by ANOTHER workgroup of the same launch (in-kernel reduction): agent-scope loads, coherent at the memory side
__device__ __forceinline__ f32x4 load_partial_coherent(const float* src) {
  typedef unsigned long long u64;
  const u64 a = __hip_atomic_load((const u64*)src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const u64 b = __hip_atomic_load((const u64*)src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  f32x4 v;
  v[0] = __uint_as_float((unsigned)a), v[1] = __uint_as_float((unsigned)(a >> 32));
  v[2] = __uint_as_float((unsigned)b), v[3] = __uint_as_float((unsigned)(b >> 32));
  return v;
}
__device__ __forceinline__ void store_partial_coherent(float* dst, const f32x4& v) {
  typedef unsigned long long u64;
  __hip_atomic_store((u64*)dst, (u64)__float_as_uint(v[0]) | ((u64)__float_as_uint(v[1]) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store((u64*)dst + 1, (u64)__float_as_uint(v[2]) | ((u64)__float_as_uint(v[3]) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one (row, 8-column) unit of the split-K second pass: C = epilogue(sum_s ws[s][m][n]) in slice order (deterministic)
template <bool COHERENT>
__device__ __forceinline__ void splitk_reduce_unit(const tb_gemm_desc& p, const EpiFlags& f, const float* __restrict__ ws, int S, int64_t npad,
                                                   int64_t m, int64_t n) {
  float r8[8], b8[8];
  epi_load_r8(p, f, m, n, r8);
  const f16x8 aux = epi_load_aux8(p, f, m, n);
  epi_load_bias8(p, n, b8);
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float* src = ws + m * npad + n;
  const int64_t plane = p.M * npad;
  int s = 0;
  for (; s + 4 <= S; s += 4) {  // 4 slices per trip: 8 independent 16-byte loads in flight, summed in slice order
    f32x4 x[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      x[2 * k] = COHERENT ? load_partial_coherent(src + (s + k) * plane) : *(const f32x4*)(src + (s + k) * plane);
      x[2 * k + 1] = COHERENT ? load_partial_coherent(src + (s + k) * plane + 4) : *(const f32x4*)(src + (s + k) * plane + 4);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] += x[2 * k][e];
        v[4 + e] += x[2 * k + 1][e];
      }
  }
  for (; s < S; ++s) {
    const f32x4 x0 = COHERENT ? load_partial_coherent(src + s * plane) : *(const f32x4*)(src + s * plane);
    const f32x4 x1 = COHERENT ? load_partial_coherent(src + s * plane + 4) : *(const f32x4*)(src + s * plane + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] += x0[e];
      v[4 + e] += x1[e];
    }
  }
  epilogue8(p, f, m, n, v, b8, r8, aux);
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const tb_gemm_desc p, const float* __restrict__ ws, int S, int64_t npad) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t groups = npad >> 3;
  if (idx >= p.M * groups) return;
  const int64_t m = idx / groups;
  const int64_t n = (idx - m * groups) << 3;
  const EpiFlags f = epi_flags(p);
  splitk_reduce_unit<false>(p, f, ws, S, npad, m, n);
}

// conv_halo_kernel on 8-wide maps (round 5): lane l31 of a 32-column MFMA tile owns output pixel halo_perm32(l31) of the tile's 32-pixel block, not
// pixel l31.  A ds_read_b128 is served in 16-lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32) and, with 128-byte LDS rows and the
// (row >> 1) & 7 chunk swizzle, is conflict-free iff a group's 16 rows are distinct mod 16.  With the pixels in lane order and the 10-pixel halo pitch
// the rows of a group are {R..R+3, R+14..R+17, R+24..R+27, R+30..R+33}: 3-way conflicts (PMC: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.47, the
// worst of the step).  With a 12-pixel halo pitch the four image rows of a block sit at 0, 12, 24, 36 (mod 16: 0, 12, 8, 4) and giving group 0 the
// image rows 0 / 2 and group 1 the rows 1 / 3 makes every group's rows {R..R+7, R+24..R+31}: all residues once.
__device__ __forceinline__ int halo_perm32(int l) {
  return l < 4 ? l : (l < 12 ? l + 4 : (l < 16 ? l - 8 : (l < 20 ? l + 8 : (l < 28 ? l - 4 : l))));
}

// Shared epilogue of the MFMA kernels (gemm_kernel, conv_halo_kernel): accumulators -> LDS -> coalesced global writes.
// PASSES = 2 stages the tile in two halves of BM/2 rows (the rows of the waves with wm == pass), which halves the LDS the epilogue
// needs: with 32-wide k-tiles the operand stages then bound the block's LDS and a third / fourth block fits on the CU.
template <int BM, int BN, int TM, int TN, int PASSES>
__device__ __forceinline__ void tile_epilogue(const tb_gemm_desc& p, f32x16 (&acc)[TM][TN], unsigned char* smem_raw, int64_t m0, int64_t n0,
                                              int wm, int wn, int S, int slice, float* __restrict__ ws, int64_t npad, int mshift = 30,
                                              int mstride = 0, bool lean_ok = true, int tile_id = 0, bool perm32 = false) {
  // tile row r (0..BM-1) is output row  m0 + (r >> mshift) * mstride + (r & (2^mshift - 1)):  contiguous rows for the GEMMs (defaults),
  // a 2^mshift-pixel-wide block of image rows (mstride = image width) for the LDS-halo conv
  constexpr int WTM = BM / 2, WTN = BN / 2;
  constexpr int PR = BM / PASSES;  // rows staged per pass
  static_assert(PASSES == 1 || PASSES == 2, "one or two passes");
  const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, hi = lane >> 5;
  // ---- epilogue, staged through LDS so every global access is a full 16-byte, row-contiguous vector:
  //  (a) each lane owns row m = .. + l31 and, per register quad r4, 4 consecutive columns: dump raw fp32 accumulators into
  //      Cs[PR][BN] (16-byte chunks XOR-swizzled by row & 7: conflict-free for these writes and for the row reads below);
  //  (b) barrier; (c) thread t walks (row, 8-column group) units: bias / residual / activation / stores in 16-byte vectors.
  float* Cs = reinterpret_cast<float*>(smem_raw);
  const EpiFlags ef = epi_flags(p);
  // LayerNorm folded into this Linear (tb_gemm_desc.rs_in, see the header): thread t < BM turns the producer's per-tile (sum, sum of squares)
  // partials of tile row t into (mean, rstd), kept behind the staging tile (launch_v adds the bytes); the staging barrier below publishes them
  typedef __attribute__((ext_vector_type(2))) float f32x2e;
  const bool lnf = p.rs_in != nullptr && S == 1;
  f32x2e* const rowst = reinterpret_cast<f32x2e*>(smem_raw + (size_t)(BM / PASSES) * BN * sizeof(float));
  if (lnf && t < BM) {
    const int64_t mr = m0 + t < p.M ? m0 + t : p.M - 1;
    const f32x2e* src = reinterpret_cast<const f32x2e*>(p.rs_in) + mr * p.rs_ld;
    const int rn = p.rs_n;
    f32x2e pv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) pv[j] = j < rn ? src[j] : f32x2e{0.f, 0.f};
    float sx = 0.f, sq = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) sx += pv[j][0], sq += pv[j][1];
    const float inv_k = 1.f / (float)p.K, mean = sx * inv_k;
    const float rstd = rsqrtf(fmaxf(sq * inv_k - mean * mean, 0.f) + p.ln_eps);
    rowst[t] = f32x2e{mean, rstd};
    if (n0 == 0 && p.ln_stats && m0 + t < p.M) *reinterpret_cast<f32x2e*>(p.ln_stats + 2 * (m0 + t)) = f32x2e{mean, rstd};
  }
  // lean-path eligibility and the descriptor fields it uses (block-uniform; see the `fast` branch below)
  const int64_t Mtot = p.M, ldc = p.ldc, ldr = p.ldr;
  const int64_t m_first = m0, m_last_ = m0 + (int64_t)((BM - 1) >> mshift) * mstride + ((BM - 1) & ((1 << mshift) - 1));
  const int64_t m_last = m_last_ < Mtot ? m_last_ : Mtot - 1;
  const bool rb_uniform = !p.rowbias || (m_first / p.rows_per_group == m_last / p.rows_per_group);
  const bool act_gelu = p.act == TB_ACT_QUICK_GELU || p.act == TB_ACT_GELU;                 // C2 (optional) receives the fp16 pre-activation
  const bool act_grad = p.act == TB_ACT_QUICK_GELU_GRAD || p.act == TB_ACT_GELU_GRAD;       // C2 holds the saved pre-activation
  const bool fast = lean_ok && S == 1 && (p.N % 8 == 0) && ef.c_vec && (!p.R || ef.r_vec) && rb_uniform &&
                    ((p.act == TB_ACT_NONE || p.act == TB_ACT_SILU) ? !p.C2 : ((act_gelu && (!p.C2 || ef.c2_vec)) || (act_grad && ef.c2_vec)));
  const float alpha = p.alpha;
  const bool silu = p.act == TB_ACT_SILU, c_f32 = p.c_dtype == TB_F32, r_f32 = p.r_dtype == TB_F32;
  const bool quick = p.act == TB_ACT_QUICK_GELU || p.act == TB_ACT_QUICK_GELU_GRAD;
  f16* const C2b = (f16*)p.C2;
  const int64_t ldc2 = p.ldc2;
  void* const Cb = p.C;
  const void* const Rb = p.R;
#pragma unroll
  for (int pass = 0; pass < PASSES; ++pass) {
    if (PASSES == 1 || wm == pass) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const int row = (PASSES == 1 ? wm * WTM : 0) + i * 32 + l31;
            const int ch = (wn * WTN + j * 32 + 8 * r4 + 4 * hi) >> 2;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = acc[i][j][4 * r4 + e];
            *(f32x4*)(Cs + row * BN + ((ch ^ (row & 7)) << 2)) = o;
          }
    }
    __syncthreads();
    const int rp = pass * PR;  // first tile row of this pass
    auto m_of = [&](int row) -> int64_t {
      const int r0_ = rp + row;
      const int r = perm32 ? ((r0_ & ~31) | halo_perm32(r0_ & 31)) : r0_;   // (conv_halo_kernel on 8-wide maps: the pixel a tile row's lane owns)
      const int64_t m = m0 + (int64_t)(r >> mshift) * mstride + (r & ((1 << mshift) - 1));
      if (p.a_mode == TB_A_CONV3X3 && p.transposed == 2) {  // phase-ordered tile rows (gemm_kernel) back to map order
        const int64_t mq = p.M >> 2;
        if (m >= p.M) return m;
        const int cls = (int)(m / mq), q = 3 - cls;
        const int64_t rr = m - (int64_t)cls * mq;
        const int hc = p.Hout >> 1, wc = p.Wout >> 1;
        const int b = (int)(rr / (hc * wc));
        const int rem = (int)(rr - (int64_t)b * hc * wc);
        const int i2 = rem / wc, j2 = rem - i2 * wc;
        return ((int64_t)b * p.Hout + 2 * i2 + (q >> 1)) * p.Wout + 2 * j2 + (q & 1);
      }
      return m;
    };
    if (p.act == TB_ACT_GEGLU) {
      // packed columns: [h0..31 | g0..31] per 64; unit = (row, 8 gate outputs); out column = packed_h_column / 2 (+ j)
      constexpr int UPR = BN / 16;  // units per row
      for (int u = t; u < PR * UPR; u += 256) {
        const int row = u / UPR, og = u - row * UPR;
        const int64_t m = m_of(row);
        if (m >= p.M) continue;
        const int hcol = (og >> 2) * 64 + (og & 3) * 8;  // tile-local packed column of h; g is +32
        float vh[8], vg[8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const f32x4 a = *(const f32x4*)(Cs + row * BN + ((((hcol >> 2) + q) ^ (row & 7)) << 2));
          const f32x4 b = *(const f32x4*)(Cs + row * BN + (((((hcol + 32) >> 2) + q) ^ (row & 7)) << 2));
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            vh[4 * q + e] = a[e];
            vg[4 * q + e] = b[e];
          }
        }
        const int64_t nh = n0 + hcol;
        f16x8 oh, og8, oo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          oh[e] = (f16)(p.alpha * vh[e] + (p.bias ? p.bias[nh + e] : 0.f));
          og8[e] = (f16)(p.alpha * vg[e] + (p.bias ? p.bias[nh + 32 + e] : 0.f));
          // gate on the fp16-rounded projections, as a fp16 module would (diffusers GEGLU on fp16 tensors)
          oo[e] = (f16)((float)oh[e] * gelu_erf_f((float)og8[e]));
        }
        if (p.C2) {
          f16* c2 = (f16*)p.C2 + m * p.ldc2 + nh;
          if (ef.c2_vec) {
            *(f16x8*)c2 = oh;
            *(f16x8*)(c2 + 32) = og8;
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              c2[e] = oh[e];
              c2[32 + e] = og8[e];
            }
          }
        }
        f16* c = (f16*)p.C + m * p.ldc + (n0 >> 1) + (og >> 2) * 32 + (og & 3) * 8;
        if (ef.c_vec) *(f16x8*)c = oo;
        else {
#pragma unroll
          for (int e = 0; e < 8; ++e) c[e] = oo[e];
        }
      }
    } else {
      constexpr int UPR = BN / 8;    // (row, 8-column) units per row
      constexpr int RS = 256 / UPR;  // rows covered per sweep of the 256 threads
      constexpr int NU = PR / RS;    // units per thread: same column group, rows row0 + it*RS
      const int cg = t % UPR, row0 = t / UPR;
      const int64_t n = n0 + cg * 8;
      if (n < p.N) {
        if (S > 1) {  // split-K: raw fp32 partial, the reducer applies the epilogue
#pragma unroll
          for (int it = 0; it < NU; ++it) {
            const int row = row0 + it * RS;
            const int64_t m = m_of(row);
            if (m >= p.M) continue;
            float* dst = ws + ((int64_t)slice * p.M + m) * npad + n;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              const f32x4 v = *(const f32x4*)(Cs + row * BN + (((cg * 2 + q) ^ (row & 7)) << 2));
              if (p.sync) store_partial_coherent(dst + 4 * q, v);  // read back by the last slice of this launch (below)
              else *(f32x4*)(dst + 4 * q) = v;
            }
          }
        } else if (fast) {
          // Lean path (every UNet launch and the CLIP out / fc2 projections): the descriptor fields are in locals -- read through `p` in
          // the unit loop they were re-fetched from the kernarg segment by scalar loads (one dependent wait each, ~10 per unit):
          // measured on the 8-wave kernel, 8 us of a 23 us 32768x320x320 launch against 3 us for the stores themselves.
          float b8[8];
          epi_load_bias8(p, n, b8);
          if (p.rowbias) {
            const float* rb = p.rowbias + (m_first / p.rows_per_group) * p.ldrb + n;
#pragma unroll
            for (int e = 0; e < 8; ++e) b8[e] += rb[e];
          }
          float c18[8];   // folded LayerNorm: c1 of this thread's 8 columns
#pragma unroll
          for (int e = 0; e < 8; ++e) c18[e] = lnf ? p.ln_gamma[n + e] : 0.f;
          float r8[NU][8];
#pragma unroll
          for (int it = 0; it < NU; ++it) {
            const int64_t m = m_of(row0 + it * RS);
            const int64_t mm = m < Mtot ? m : Mtot - 1;
            if (!Rb) {
#pragma unroll
              for (int e = 0; e < 8; ++e) r8[it][e] = 0.f;
            } else if (r_f32) {
              const f32x4 x0 = *(const f32x4*)((const float*)Rb + mm * ldr + n), x1 = *(const f32x4*)((const float*)Rb + mm * ldr + n + 4);
#pragma unroll
              for (int e = 0; e < 4; ++e) r8[it][e] = x0[e], r8[it][4 + e] = x1[e];
            } else {
              const f16x8 x = *(const f16x8*)((const f16*)Rb + mm * ldr + n);
#pragma unroll
              for (int e = 0; e < 8; ++e) r8[it][e] = (float)x[e];
            }
          }
          f16x8 aux[NU];
          if (act_grad) {
#pragma unroll
            for (int it = 0; it < NU; ++it) {
              const int64_t m = m_of(row0 + it * RS);
              aux[it] = *(const f16x8*)(C2b + (m < Mtot ? m : Mtot - 1) * ldc2 + n);
            }
          }
#pragma unroll
          for (int it = 0; it < NU; ++it) {
            const int row = row0 + it * RS;
            const int64_t m = m_of(row);
            if (m >= Mtot) continue;
            float v[8];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              const f32x4 a = *(const f32x4*)(Cs + row * BN + (((cg * 2 + q) ^ (row & 7)) << 2));
#pragma unroll
              for (int e = 0; e < 4; ++e) v[4 * q + e] = a[e];
            }
            if (lnf) {
              const f32x2e rs = rowst[rp + row];
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = rs[1] * (v[e] * alpha - rs[0] * c18[e]) + b8[e] + r8[it][e];
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = v[e] * alpha + b8[e] + r8[it][e];
            }
            if (silu) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
            }
            if (act_gelu) {  // the fp16-rounded linear output feeds the activation, as under autocast; it is what the backward re-reads
              f16x8 pre;
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                pre[e] = (f16)v[e];
                v[e] = quick ? quick_gelu_f((float)pre[e]) : gelu_erf_f((float)pre[e]);
              }
              if (C2b) *(f16x8*)(C2b + m * ldc2 + n) = pre;
            } else if (act_grad) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] *= quick ? quick_gelu_grad_f((float)aux[it][e]) : gelu_erf_grad_f((float)aux[it][e]);
            }
            if (c_f32) {
              f32x4 o0, o1;
#pragma unroll
              for (int e = 0; e < 4; ++e) o0[e] = v[e], o1[e] = v[4 + e];
              *(f32x4*)((float*)Cb + m * ldc + n) = o0;
              *(f32x4*)((float*)Cb + m * ldc + n + 4) = o1;
            } else {
              f16x8 o;
#pragma unroll
              for (int e = 0; e < 8; ++e) o[e] = (f16)v[e];
              *(f16x8*)((f16*)Cb + m * ldc + n) = o;
            }
          }
        } else {
          float b8[8];
          epi_load_bias8(p, n, b8);
          float r8[NU][8];
          f16x8 aux[NU];
#pragma unroll
          for (int it = 0; it < NU; ++it) {  // all residual / aux loads first: their (cold) latency overlaps
            const int64_t m = m_of(row0 + it * RS);
            const int64_t mm = m < p.M ? m : p.M - 1;
            epi_load_r8(p, ef, mm, n, r8[it]);
            aux[it] = epi_load_aux8(p, ef, mm, n);
          }
#pragma unroll
          for (int it = 0; it < NU; ++it) {
            const int row = row0 + it * RS;
            const int64_t m = m_of(row);
            if (m >= p.M) continue;
            float v[8];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              const f32x4 a = *(const f32x4*)(Cs + row * BN + (((cg * 2 + q) ^ (row & 7)) << 2));
#pragma unroll
              for (int e = 0; e < 4; ++e) v[4 * q + e] = a[e];
            }
            epilogue8(p, ef, m, n, v, b8, r8[it], aux[it]);
          }
        }
      }
    }
    if (pass + 1 < PASSES) __syncthreads();  // the next half overwrites the staging tile
  }
  // In-kernel split-K reduction (p.sync: one zeroed counter per tile, left zeroed): the slice that arrives LAST at the tile's counter adds the S
  // partials in slice order -- the arithmetic of splitk_reduce_kernel, bit for bit -- and applies the epilogue; no second launch (a graph node
  // costs ~4.8 us before it does anything).  No cache-wide fences: partials are written and read with agent-scope accesses
  // (coherent at the memory side whichever XCD a slice ran on); the arrival counter is an agent-scope atomic issued after the stores completed.
  if (S > 1 && p.sync) {
    __shared__ int last_slice;
    __syncthreads();  // (s_waitcnt vmcnt(0): every thread's partial stores have been acknowledged)
    if (t == 0) {
      const unsigned old = __hip_atomic_fetch_add(p.sync + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last_slice = old == (unsigned)(S - 1);
      if (last_slice) __hip_atomic_store(p.sync + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!last_slice) return;
    constexpr int UPRT = BN / 8;
    for (int u = t; u < BM * UPRT; u += 256) {
      const int r0_ = u / UPRT, cgu = u - r0_ * UPRT;
      const int r = perm32 ? ((r0_ & ~31) | halo_perm32(r0_ & 31)) : r0_;
      int64_t m = m0 + (int64_t)(r >> mshift) * mstride + (r & ((1 << mshift) - 1));
      const int64_t n = n0 + cgu * 8;
      if (m >= p.M || n >= p.N) continue;
      if (p.a_mode == TB_A_CONV3X3 && p.transposed == 2) {  // phase-ordered tile rows back to map order (as m_of above)
        const int64_t mq = p.M >> 2;
        const int cls = (int)(m / mq), q = 3 - cls;
        const int64_t rr = m - (int64_t)cls * mq;
        const int hc = p.Hout >> 1, wc = p.Wout >> 1;
        const int b = (int)(rr / (hc * wc));
        const int rem = (int)(rr - (int64_t)b * hc * wc);
        const int i2 = rem / wc, j2 = rem - i2 * wc;
        m = ((int64_t)b * p.Hout + 2 * i2 + (q >> 1)) * p.Wout + 2 * j2 + (q & 1);
      }
      splitk_reduce_unit<true>(p, ef, ws, S, npad, m, n);
    }
  }
}

// BKT: halfs per k-tile (64 -> 128-byte LDS rows, 32 -> 64-byte rows); NST: LDS stages (2, or 3 with counted vmcnt)
template <int BM, int BN, int MODE, int BKT, int NST>
__global__ __launch_bounds__(256, (BKT == 32 ? 3 : (NST >= 4 && BM == 128 ? 1 : 2))) void gemm_kernel(const tb_gemm_desc p, int tiles_m, int tiles_n, int S, float* __restrict__ ws, int64_t npad, int abl,
                                                                              int m_fastest) {
  extern __shared__ __attribute__((aligned(128))) unsigned char smem_raw[];
  f16* smem = reinterpret_cast<f16*>(smem_raw);
  constexpr int BK = BKT;
  constexpr int A_TILE = BM * BK, B_TILE = BN * BK, STAGE = A_TILE + B_TILE;
  constexpr int CPR = BK / 8;          // 16-byte chunks per LDS row
  constexpr int RPI = 64 / CPR;        // rows filled by one wave-wide global_load_lds
#define SWZ(row) (BKT == 64 ? (((row) >> 1) & 7) : (((row) >> 2) & 3))

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  constexpr int WTM = BM / 2, WTN = BN / 2, TM = WTM / 32, TN = WTN / 32;
  constexpr int AI = BM / 4 / RPI, BI = BN / 4 / RPI;  // global_load_lds instructions per wave per k-tile

  // ---- XCD-aware tile order: blocks b, b+8, b+16, ... share an XCD (and its L2); give each XCD a contiguous run of tiles
  const int nwg = tiles_m * tiles_n * S;
  int tile, slice;
  {
    const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
#ifdef TB_GEMM_NO_XCD
    const int logical = bid;
    (void)xcd; (void)q; (void)r;
#else
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
#endif
    tile = logical / S;       // the S k-slices of one tile are adjacent (same XCD)
    slice = logical - tile * S;
  }
  // tile order (host cost model, see launch_v): keep one operand's panel L2-resident while the other streams through the XCD
  int tm, tn;
  if (m_fastest == 3) {  // grouped: 8 M-tiles x all N-tiles, M fastest inside the group -> the ~64 blocks that run together on one
                         // XCD form an 8 x 8 patch of the tile grid (8 A-panel + 8 W-panel k-slices shared through its L2)
    const int gsize = 8 * tiles_n;
    const int gid = tile / gsize, first_m = gid * 8;
    const int gm = min(tiles_m - first_m, 8);
    const int in_g = tile - gid * gsize;
    tm = first_m + in_g % gm;
    tn = in_g / gm;
  } else if (m_fastest) {
    tm = tile % tiles_m;
    tn = tile / tiles_m;
  } else {
    tm = tile / tiles_n;
    tn = tile % tiles_n;
  }
  const int64_t m0 = (int64_t)tm * BM;
  const int64_t n0 = (int64_t)tn * BN;

  const int cp = lane % CPR;  // 16-byte chunk position inside the LDS row this lane fills
  const int rl = lane / CPR;  // row within the row group of one load instruction

  // ---- per-lane source descriptors (rows are fixed across k-tiles)
  const f16* a_ptr[AI];
  const f16* a2_ptr[AI];
  bool a_ok[AI];
  int py[AI], px[AI], a_sw[AI], a_step[AI];
  int64_t pbase[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int row = wave * (BM / 4) + i * RPI + rl;
    const int64_t m = m0 + row;
    a_ok[i] = m < p.M;
    const int64_t mm = a_ok[i] ? m : 0;
    a_sw[i] = (cp ^ SWZ(row)) * 8;  // source chunk (halfs) that belongs at LDS chunk position cp of this row
    if (MODE == TB_A_LINEAR) {
      a_ptr[i] = (const f16*)p.A + mm * p.lda + a_sw[i];
      a2_ptr[i] = p.A2 ? (const f16*)p.A2 + mm * p.lda2 + a_sw[i] : nullptr;
    } else if (p.transposed == 2) {
      // phase-ordered rows (see phase_q below): row m' = class * (M / 4) + ((b * Hout/2 + i) * Wout/2 + j)  ->  output pixel (2 i + qy, 2 j + qx)
      const int64_t mq = p.M >> 2;
      const int cls = (int)(mm / mq);
      const int q = 3 - cls;
      const int64_t r = mm - (int64_t)cls * mq;
      const int hc = p.Hout >> 1, wc = p.Wout >> 1;
      const int b = (int)(r / (hc * wc));
      const int rem = (int)(r - (int64_t)b * hc * wc);
      const int i2 = rem / wc;
      py[i] = 2 * i2 + (q >> 1);
      px[i] = 2 * (rem - i2 * wc) + (q & 1);
      pbase[i] = (int64_t)b * p.Hin * p.Win;
    } else {
      const int hw = p.Hout * p.Wout;
      const int b = (int)(mm / hw);
      const int rem = (int)(mm - (int64_t)b * hw);
      py[i] = rem / p.Wout;
      px[i] = rem - py[i] * p.Wout;
      pbase[i] = (int64_t)b * p.Hin * p.Win;
    }
  }
  // Transposed (stride-2 dgrad) gather, phase mode (p.transposed == 2, set by tb_gemm when the map is even and M / 4 is a whole number of tiles):
  // an output pixel only receives the taps with ky = y + 1 (mod 2), kx = x + 1 (mod 2) -- 1, 2, 2 or 4 of the 9, by the parity class (y & 1, x & 1).
  // With rows in map order every tile holds all four classes and each tap's k-tiles were multiplied for every row (3 of 4 against the zero
  // line: 60 of the 80 GFLOP of a 320-channel 64x64 map).  Here the tile grid walks the rows class by class (heaviest class first), so a tile
  // only issues its class's taps: 2.25 taps per pixel on average instead of 9.  The epilogue maps tile rows back to map order.
  const int phase_q = (MODE == TB_A_CONV3X3 && p.transposed == 2) ? 3 - (int)(m0 / (p.M >> 2)) : -1;
  const int ph_nky = phase_q >= 0 && (phase_q >> 1) ? 2 : 1, ph_nkx = phase_q >= 0 && (phase_q & 1) ? 2 : 1;
  const f16* w_ptr[BI];
  const f16* w2_ptr[BI];
  bool w_ok[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int row = wave * (BN / 4) + i * RPI + rl;
    const int64_t n = n0 + row;
    w_ok[i] = n < p.N;
    const int64_t nn = w_ok[i] ? n : 0;
    const int sw = (cp ^ SWZ(row)) * 8;
    w_ptr[i] = (const f16*)p.W + nn * p.ldw + sw;
    w2_ptr[i] = p.W2 ? (const f16*)p.W2 + nn * p.ldw2 + sw : nullptr;
  }
#ifndef TB_GEMM_OLD_ADDR
  // Fast addressing (operands below 4 GiB, checked by the host): every lane keeps a 32-bit byte offset from the operand base and the
  // k advance lives in the uniform base, so a k-tile's loads cost no vector ALU work at all (plain VALU does not overlap the MFMAs of
  // the co-resident waves).  Rows past M / N are clamped to the last row instead of the zero line: they only
  // feed output rows / columns that are never stored.  (Conv A operands keep per-tap pointers: their padding taps must read zeros.)
  uint32_t a_off[AI], a2_off[AI], w_off[BI], w2_off[BI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int row = wave * (BM / 4) + i * RPI + rl;
    const int64_t mm = min((int64_t)(m0 + row), p.M - 1);
    a_off[i] = (uint32_t)((mm * p.lda + a_sw[i]) * 2);
    a2_off[i] = (uint32_t)((mm * p.lda2 + a_sw[i]) * 2);
  }
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int row = wave * (BN / 4) + i * RPI + rl;
    const int64_t nn = min((int64_t)(n0 + row), p.N - 1);
    const int sw = (cp ^ SWZ(row)) * 8;
    w_off[i] = (uint32_t)((nn * p.ldw + sw) * 2);
    w2_off[i] = (uint32_t)((nn * p.ldw2 + sw) * 2);
  }
#endif

  const int kpt = (MODE == TB_A_CONV3X3) ? p.Cin / BK : 1;  // k-tiles per tap
  const int nk_all = phase_q >= 0 ? ph_nky * ph_nkx * kpt : (int)(p.K / BK);
  const int nk1 = phase_q >= 0 ? nk_all : (int)(p.K1 / BK);
  const int kt_begin = (int)((int64_t)nk_all * slice / S), kt_end = (int)((int64_t)nk_all * (slice + 1) / S);
  const int nk = kt_end - kt_begin;  // k-tiles of this block; stage()/loops below index them relative to kt_begin
  const f16* zero = g_zero_line;

  auto stage = [&](int kt_rel, int buf) {
    const int kt = kt_begin + kt_rel;
    f16* As = smem + buf * STAGE + (wave * (BM / 4)) * BK;
    f16* Bs = smem + buf * STAGE + A_TILE + (wave * (BN / 4)) * BK;
    if (MODE == TB_A_LINEAR) {
      const bool second = kt >= nk1;  // wave-uniform: which K-source this k-tile comes from
      const int koff = (second ? kt - nk1 : kt) * BK;
#ifdef TB_GEMM_OLD_ADDR
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        const f16* src = a_ok[i] ? (second ? a2_ptr[i] : a_ptr[i]) + koff : zero;
        glds16(src, As + i * RPI * BK);
      }
#else
      if (second) {
        const char* base = (const char*)p.A2 + (int64_t)koff * 2;
#pragma unroll
        for (int i = 0; i < AI; ++i) glds16((const f16*)(base + a2_off[i]), As + i * RPI * BK);
      } else {
        const char* base = (const char*)p.A + (int64_t)koff * 2;
#pragma unroll
        for (int i = 0; i < AI; ++i) glds16((const f16*)(base + a_off[i]), As + i * RPI * BK);
      }
#endif
    } else {
      const int tap = kt / kpt;   // (phase mode: index into the class's tap list)
      const int cc = kt - tap * kpt;
      if (cc == 0 || kt_rel == 0) {  // tap changed (or first tile of this block): recompute the gathered row pointers
        int ky = tap / 3, kx = tap - ky * 3;
        if (phase_q >= 0) {
          const int ty_ = tap / ph_nkx, tx_ = tap - ty_ * ph_nkx;
          ky = ph_nky == 2 ? 2 * ty_ : 1;
          kx = ph_nkx == 2 ? 2 * tx_ : 1;
        }
#pragma unroll
        for (int i = 0; i < AI; ++i) {
          int sy, sx;
          bool ok;
          if (p.upsample) {
            const int uy = py[i] + ky - 1, ux = px[i] + kx - 1;
            ok = uy >= 0 && ux >= 0 && uy < 2 * p.Hin && ux < 2 * p.Win;
            sy = uy >> 1;
            sx = ux >> 1;
          } else if (p.transposed) {
            const int ty = py[i] + 1 - ky, tx = px[i] + 1 - kx;
            ok = ty >= 0 && tx >= 0 && !(ty & 1) && !(tx & 1);
            sy = ty >> 1;
            sx = tx >> 1;
            ok = ok && sy < p.Hin && sx < p.Win;
          } else {
            sy = py[i] * p.stride + p.sign * (ky - 1) + p.shift;
            sx = px[i] * p.stride + p.sign * (kx - 1) + p.shift;
            ok = sy >= 0 && sx >= 0 && sy < p.Hin && sx < p.Win;
          }
          ok = ok && a_ok[i];
          // invalid taps read the zero line with a zero per-tile stride
          a_ptr[i] = ok ? (const f16*)p.A + (pbase[i] + (int64_t)sy * p.Win + sx) * p.lda + a_sw[i] : zero;
          a_step[i] = ok ? BK : 0;
        }
      }
#pragma unroll
      for (int i = 0; i < AI; ++i) glds16(a_ptr[i] + (int64_t)cc * a_step[i], As + i * RPI * BK);
    }
    {
      const bool second = kt >= nk1;
      int koff = (second ? kt - nk1 : kt) * BK;
      if (MODE == TB_A_CONV3X3 && phase_q >= 0) {  // the k-tile of the class's tap list -> its place in the 9-tap weight row
        const int tl = kt / kpt, cc = kt - tl * kpt;
        const int ty_ = tl / ph_nkx, tx_ = tl - ty_ * ph_nkx;
        const int ky = ph_nky == 2 ? 2 * ty_ : 1, kx = ph_nkx == 2 ? 2 * tx_ : 1;
        koff = ((ky * 3 + kx) * kpt + cc) * BK;
      }
#ifdef TB_GEMM_OLD_ADDR
#pragma unroll
      for (int i = 0; i < BI; ++i) {
        const f16* src = w_ok[i] ? (second ? w2_ptr[i] : w_ptr[i]) + koff : zero;
        glds16(src, Bs + i * RPI * BK);
      }
#else
      if (second) {
        const char* base = (const char*)p.W2 + (int64_t)koff * 2;
#pragma unroll
        for (int i = 0; i < BI; ++i) glds16((const f16*)(base + w2_off[i]), Bs + i * RPI * BK);
      } else {
        const char* base = (const char*)p.W + (int64_t)koff * 2;
#pragma unroll
        for (int i = 0; i < BI; ++i) glds16((const f16*)(base + w_off[i]), Bs + i * RPI * BK);
      }
#endif
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;


  const int l31 = lane & 31, hi = lane >> 5;
  auto compute = [&](int buf) {
    const f16* Ab = smem + buf * STAGE + (wm * WTM) * BK;
    const f16* Bb = smem + buf * STAGE + A_TILE + (wn * WTN) * BK;
#ifdef TB_GEMM_OLD_ADDR
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      f16x8 af[TM], bf[TN];
      const int ch = kk * 2 + hi;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = i * 32 + l31;
        af[i] = *(const f16x8*)(Ab + row * BK + ((ch ^ SWZ(row)) << 3));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = j * 32 + l31;
        bf[j] = *(const f16x8*)(Bb + row * BK + ((ch ^ SWZ(row)) << 3));
      }
      __builtin_amdgcn_s_setprio(1);  // +0.2..0.5 % in situ: MFMA bursts win arbitration over the co-resident block's loads
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)  // transposed accumulator: rows = n (from W), cols = m (from A)
          acc[i][j] = TB_MFMA_32x32x16(bf[j], af[i], acc[i][j]);
      __builtin_amdgcn_s_setprio(0);
    }
#else
    // all fragment reads of the k-tile are issued before its first MFMA (one exposed LDS latency per k-tile instead of one per 16-wide
    // k-step; s_setprio inside the loop had been a scheduling barrier that kept the compiler from doing this itself)
    constexpr int NKK = BK / 16;
    f16x8 af[NKK][TM], bf[NKK][TN];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      const int ch = kk * 2 + hi;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = i * 32 + l31;
        af[kk][i] = *(const f16x8*)(Ab + row * BK + ((ch ^ SWZ(row)) << 3));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = j * 32 + l31;
        bf[kk][j] = *(const f16x8*)(Bb + row * BK + ((ch ^ SWZ(row)) << 3));
      }
    }
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)  // transposed accumulator: rows = n (from W), cols = m (from A)
          acc[i][j] = TB_MFMA_32x32x16(bf[kk][j], af[kk][i], acc[i][j]);
#endif
  };
  if constexpr (NST == 2) {
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < nk && !(abl & 1)) stage(kt + 1, cur ^ 1);
      if (!(abl & 2)) compute(cur);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  } else {
    // NST stages: tiles kt+1 .. kt+NST-2 stay in flight while tile kt is multiplied.  Each wave issues AI+BI loads per tile,
    // so "tile kt has landed" == at most (tiles still in flight) * (AI+BI) younger loads outstanding: counted vmcnt + raw
    // barrier (cdna guide T3/T4) -- a plain __syncthreads() would drain the queue.
    constexpr int LPT = AI + BI;
#pragma unroll
    for (int i = 0; i < NST - 1; ++i)
      if (i < nk) stage(i, i);
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
      const int ahead = min(nk - kt - 1, NST - 2);  // tiles issued after kt that may still be in flight
      if (NST >= 6 && ahead >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * LPT) : "memory");
      else if (NST >= 5 && ahead == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPT) : "memory");
      else if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      // buffer (kt-1) % NST was last read in iteration kt-1, which every wave has left (barrier above)
      if (kt + NST - 1 < nk && !(abl & 1)) stage(kt + NST - 1, buf == 0 ? NST - 1 : buf - 1);
      if (!(abl & 2)) compute(buf);
      buf = buf == NST - 1 ? 0 : buf + 1;
    }
    __syncthreads();  // all waves done with the operand tiles before the epilogue reuses the LDS
  }
#undef SWZ

  if (abl & 4) return;  // profiling: no epilogue
  tile_epilogue<BM, BN, TM, TN, (BKT == 32 ? 2 : 1)>(p, acc, smem_raw, m0, n0, wm, wn, S, slice, ws, npad, 30, 0, !(abl & 8), tm * tiles_n + tn);
}

// ------------------------------------------------------------------------------------------------------------------------
// 3x3 stride-1 convolution with an LDS-resident input HALO tile (used when a 128-pixel output tile is a whole number of image
// rows or, for wider / non-power-of-two maps, a (128/TW) x TW block with TW the largest power of two <= 64 dividing W): for each 64-channel chunk the (R+2) x (W+2) input pixels the tile needs are fetched ONCE and all 9 taps read their A
// fragments from that halo at shifted row addresses; only the weight tiles stream per tap.  Compared with gemm_kernel's
// per-tap gather this moves 4.4x fewer activation bytes through the L2->LDS path that bounds the kernel.
template <int BN, int NSW = 2>
__global__ __launch_bounds__(256, (NSW > 2 ? 1 : 2)) void conv_halo_kernel(const tb_gemm_desc p, int tiles_m, int tiles_n, int wshift, int S,
                                                             float* __restrict__ ws, int64_t npad, int perm8) {
  extern __shared__ __attribute__((aligned(128))) unsigned char smem_raw[];
  f16* smem = reinterpret_cast<f16*>(smem_raw);
  constexpr int BM = 128, BK = 64;
  constexpr int WTM = BM / 2, WTN = BN / 2, TM = WTM / 32, TN = WTN / 32;
  constexpr int BI = BN / 32;  // weight-tile load instructions per wave per tap
  constexpr int MAXHI = 9;     // halo load instructions per wave (<= ceil(33 / 4): 4 x 66 halo pixels)
#define SWZ(row) ((((row) >> 1) & 7))
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // tile = R image rows x TW = min(W, 64) columns (R * TW = 128 output pixels): whole rows for W <= 64, a 2 x 64 block for wider maps
  // (the same 4 x 66 halo and 3 blocks per CU as the 64-wide case; a 1 x 128 row segment would need 3 x 130 halo pixels and 66 KB)
  // Maps smaller than a tile (8x8: H * W = 64, round 4): the tile is 128 / (H W) WHOLE images, the halo a stack of per-image (H + 2) x (W + 2) blocks
  // (each image keeps its own zero border: a tap must not reach into the neighbouring image) -- the per-tap gather of gemm_kernel moved 9 x 16 KB of
  // activations per channel chunk through the LDS-DMA path that bounds these launches, the halo is 25 KB.
  const int TW = 1 << wshift, W = p.Wout, H = p.Hout, R = BM >> wshift;
  const int hw_img = H * W;
  const bool multi = hw_img < BM;                       // host: then W == TW and BM % hw_img == 0
  const bool perm = multi && TW == 8 && perm8;   // 8-wide maps: 12-pixel halo pitch + permuted lanes (halo_perm32); the host's choice (launch_halo)
  const int HC = perm ? 12 : TW + 2, HIMG = (H + 2) * HC;           // halo pixels per image (multi)
  const int NH = multi ? (BM / hw_img) * HIMG : (R + 2) * HC, NH8 = (NH + 7) & ~7, NI = NH8 >> 3;
  // NSW > 2 (round 5, the 8x8-map launches): TWO halo panels and an NSW-slot weight ring with counted vmcnt, one workgroup per CU -- see the
  // flattened step loop below
  constexpr int NHB = NSW > 2 ? 2 : 1;
  f16* Hs = smem;
  f16* Wst = smem + NHB * NH8 * BK;  // NSW weight stages of BN x 64 halfs

  const int nwg = tiles_m * tiles_n * S;
  int tile, slice;
  {
    const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    tile = logical / S;  // the S channel-chunk slices of one tile are adjacent (same XCD)
    slice = logical - tile * S;
  }
  int tm, tn;
  {
    const int gsize = 8 * tiles_n;
    const int gid = tile / gsize, first_m = gid * 8;
    const int gm = min(tiles_m - first_m, 8);
    const int in_g = tile - gid * gsize;
    tm = first_m + in_g % gm;
    tn = in_g / gm;
  }
  const int64_t n0 = (int64_t)tn * BN;
  const int hw = H * W;
  const int tiles_x = W >> wshift, tiles_img = (H / R) * tiles_x;  // tiles per image row-block / per image
  const int b = multi ? tm * (BM / hw_img) : tm / tiles_img;
  const int trem = multi ? 0 : tm - b * tiles_img;
  const int y0 = (trem / tiles_x) * R, x0 = (trem % tiles_x) << wshift;  // first image row / column of the tile
  const int64_t m0 = (int64_t)b * hw + (int64_t)y0 * W + x0;             // output row of the tile's first pixel

  const int cp = lane & 7, rl = lane >> 3;
  // ---- halo sources of this lane (fixed across channel chunks, + 64 halfs per chunk)
  const f16* h_ptr[MAXHI];
  int h_step[MAXHI];
  const f16* zero = g_zero_line;
#pragma unroll
  for (int i = 0; i < MAXHI; ++i) {
    const int j = wave + 4 * i;  // instruction index; rows j*8 .. j*8+7 of the halo image
    const int hr = j * 8 + rl;
    const int img = multi ? hr / HIMG : 0, hr1 = hr - img * HIMG;   // (multi: image of the stack, halo pixel inside it)
    const int hy = hr1 / HC, hx = hr1 - hy * HC;
    const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
    const bool ok = j < NI && hr < NH && yy >= 0 && yy < H && xx >= 0 && xx < W;
    h_ptr[i] = ok ? (const f16*)p.A + ((int64_t)(b + img) * hw + (int64_t)yy * W + xx) * p.lda + ((cp ^ SWZ(hr)) << 3) : zero;
    h_step[i] = ok ? BK : 0;
  }
  const f16* w_ptr[BI];
  bool w_ok[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int row = wave * (BN / 4) + i * 8 + rl;
    const int64_t n = n0 + row;
    w_ok[i] = n < p.N;
    w_ptr[i] = (const f16*)p.W + (w_ok[i] ? n : 0) * p.ldw + ((cp ^ SWZ(row)) << 3);
  }
  const int kpt = p.Cin / BK;
  const int c_begin = (int)((int64_t)kpt * slice / S), c_end = (int)((int64_t)kpt * (slice + 1) / S);
  uint32_t w_voff[BI];   // (NSW > 2) byte offsets of this lane's weight rows from p.W; rows past N read row 0: those output columns are never stored
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int row = wave * (BN / 4) + i * 8 + rl;
    w_voff[i] = (uint32_t)(((w_ok[i] ? n0 + row : 0) * p.ldw + ((cp ^ SWZ(row)) << 3)) * 2);
  }

  auto stage_halo = [&](int c, int hb = 0) {
#pragma unroll
    for (int i = 0; i < MAXHI; ++i) {
      const int j = wave + 4 * i;
      if (j < NI) glds16(h_ptr[i] + (int64_t)c * h_step[i], Hs + hb * NH8 * BK + j * 8 * BK);
    }
  };
  auto stage_w = [&](int tap, int c, int buf) {
    f16* dst = Wst + buf * (BN * BK) + (wave * (BN / 4)) * BK;
    const int koff = (tap * kpt + c) * BK;
#pragma unroll
    for (int i = 0; i < BI; ++i) glds16(w_ok[i] ? w_ptr[i] + koff : zero, dst + i * 8 * BK);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int l31 = lane & 31, hi = lane >> 5;
  int hrow0[TM];  // halo row of this lane's output pixel for tap offset (0, 0)
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int ml = wm * WTM + i * 32 + (perm ? halo_perm32(l31) : l31);
    const int img = multi ? ml / hw_img : 0, ml1 = ml - img * hw_img;
    hrow0[i] = img * HIMG + (ml1 >> wshift) * HC + (ml1 & (TW - 1));
  }

  if constexpr (NSW > 2) {
    // One workgroup per CU streams its (channel chunk, tap) steps through an NSW-slot weight ring: NSW - 2 weight tiles (and the next chunk's halo)
    // stay in flight across every barrier.  The 2-slot loop below exposes one memory round trip per tap -- at the 8x8 maps every weight byte is
    // HBM-cold and read once: 18 taps x ~1.3 us per workgroup for 0.3 us of matrix work each (30 us for 1280 -> 1280, 1 TB/s of weights).
    const int nstep = (c_end - c_begin) * 9;
    stage_halo(c_begin, 0);
#pragma unroll
    for (int i = 0; i < NSW - 1; ++i)
      if (i < nstep) stage_w(i % 9, c_begin + i / 9, i);
    int slot = 0, tap = 0, c = c_begin, ltap = (NSW - 1) % 9, lc = c_begin + (NSW - 1) / 9, lslot = NSW - 1;
    for (int st = 0; st < nstep; ++st) {
      {   // stage st (issued NSW - 1 steps ago) has landed: at most the NSW - 2 younger weight stages stay in flight (a halo issued among them
          // makes the count wait for a little more than it must -- never less)
        int later = nstep - 1 - st;
        later = later > NSW - 2 ? NSW - 2 : later;
        switch (later) {
          case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
          case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BI) : "memory"); break;
          case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * BI) : "memory"); break;
          default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * BI) : "memory"); break;
        }
        static_assert(NSW <= 5, "counted waits up to three stages in flight");
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // ... for every wave, and every wave has left step st - 1
      if (tap == 0 && c + 1 < c_end) stage_halo(c + 1, (c + 1 - c_begin) & 1);   // (that panel was last read in chunk c - 1)
      // the BI weight pieces of stage st + NSW - 1 (its slot was read in step st - 1) go out BETWEEN the MFMAs below, as asm (a builtin piece in front
      // of the MFMAs blocks its wave for 100+ cycles; with one wave per SIMD nothing else runs meanwhile: 42 us against the 2-slot kernel's 38)
      const bool w_on = st + NSW - 1 < nstep;
      const char* w_base = (const char*)p.W + (int64_t)(ltap * kpt + lc) * BK * 2;
      const uint32_t w_dst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)(Wst + lslot * (BN * BK) + (wave * (BN / 4)) * BK);
      const int ky = tap / 3, kx = tap - ky * 3;
      const int dy = p.sign > 0 ? ky : 2 - ky, dx = p.sign > 0 ? kx : 2 - kx;
      const int shift = dy * HC + dx;
      const f16* Hb = Hs + ((c - c_begin) & 1) * NH8 * BK;
      const f16* Bb = Wst + slot * (BN * BK) + (wn * WTN) * BK;
#pragma unroll
      for (int kk = 0; kk < BK / 16; ++kk) {
        f16x8 af[TM], bf[TN];
        const int ch = kk * 2 + hi;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int row = hrow0[i] + shift;
          af[i] = *(const f16x8*)(Hb + row * BK + ((ch ^ SWZ(row)) << 3));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int row = j * 32 + l31;
          bf[j] = *(const f16x8*)(Bb + row * BK + ((ch ^ SWZ(row)) << 3));
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = TB_MFMA_32x32x16(bf[j], af[i], acc[i][j]);
        static_assert(BI == BK / 16, "one weight piece per 16-wide k-step");
        if (w_on) {
          __builtin_amdgcn_sched_barrier(0);
          glds16_asm_so(w_base, w_voff[kk], w_dst + kk * 8 * BK * 2);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      slot = slot + 1 == NSW ? 0 : slot + 1;
      lslot = lslot + 1 == NSW ? 0 : lslot + 1;
      if (++tap == 9) tap = 0, ++c;
      if (++ltap == 9) ltap = 0, ++lc;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the epilogue reuses the LDS
  } else
  for (int c = c_begin; c < c_end; ++c) {
    // every wave has left the previous chunk's last tap (barrier there), so the halo and weight stage 0 may be overwritten
    stage_halo(c);
    stage_w(0, c, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      const int cur = tap & 1;
      if (tap + 1 < 9) stage_w(tap + 1, c, cur ^ 1);
      const int ky = tap / 3, kx = tap - ky * 3;
      const int dy = p.sign > 0 ? ky : 2 - ky, dx = p.sign > 0 ? kx : 2 - kx;  // dgrad gathers with flipped offsets
      const int shift = dy * HC + dx;
      const f16* Bb = Wst + cur * (BN * BK) + (wn * WTN) * BK;
#pragma unroll
      for (int kk = 0; kk < BK / 16; ++kk) {
        f16x8 af[TM], bf[TN];
        const int ch = kk * 2 + hi;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int row = hrow0[i] + shift;
          af[i] = *(const f16x8*)(Hs + row * BK + ((ch ^ SWZ(row)) << 3));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int row = j * 32 + l31;
          bf[j] = *(const f16x8*)(Bb + row * BK + ((ch ^ SWZ(row)) << 3));
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = TB_MFMA_32x32x16(bf[j], af[i], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
#undef SWZ
  tile_epilogue<BM, BN, TM, TN, 1>(p, acc, smem_raw, m0, n0, wm, wn, S, slice, ws, npad, wshift, W, true, tm * tiles_n + tn, perm);
}


// tb_gemm_desc.split_out: the caller adds the k-slices itself (a consumer kernel that reads the fp32 partials: tb_groupnorm_*_splitk) -- no reducer launch
inline bool defer_reduce(const tb_gemm_desc& d, int S) {
  return S > 1 && d.split_out && d.act == TB_ACT_NONE && d.c_dtype == TB_F16 && d.alpha == 1.f && !d.C2 && (!d.R || d.r_dtype == TB_F16) &&
         !(d.a_mode == TB_A_CONV3X3 && d.transposed == 2);
}

int g_inkernel_reduce = 0;  // split-K launches with tb_gemm_desc.sync reduce in the kernel (tb_gemm_set_variant(9800 + {0,1})).  OFF: bit-equal to the
                            // reducer launch, but the agent-scope partial stores / loads it needs make the step 35.1 ms against 31.3 (scratch/ab_step.py)
int g_last_cfg[5] = {0, 0, 0, 0, 0};  // BM, BN, MODE, k-tile, split of the most recent launch (bench.py names kernels by it)
int g_order = 0;    // tile order: 0 = 8-row groups (default), 1 = n-fastest, 2 = m-fastest (tb_gemm_set_variant(3000 + v))
int g_halo = 11;    // (bit 3 = 8-wide maps on the 12-pixel halo pitch with permuted lanes: conflict-free fragment reads; bit 2, the 5-slot ring, measured slower: off)
                    // 3x3 stride-1 convs use conv_halo_kernel: bit 0 = tiles of whole image rows, bit 1 = tiles of whole small images (8x8 maps), bit 2 = the 8x8-map launches on the 5-slot weight ring (one workgroup per CU) (tb_gemm_set_variant(7000 + bits))
int g_ablate = 0;   // profiling only (tb_gemm_set_variant(2000 + bits)): 1 = skip k-loop loads, 2 = skip k-loop MFMAs
int g_variant = 0;  // tuning knob (tb_gemm_set_variant): 0 = BK64 x 2 stages, 1 = BK32 x 3 stages, 2 = BK32 x 2 stages

template <int BM, int BN, int MODE, int BKT, int NST>
int launch_v(const tb_gemm_desc& d, hipStream_t s, int S) {
  const int tiles_m = (int)((d.M + BM - 1) / BM), tiles_n = (int)((d.N + BN - 1) / BN);
  g_last_cfg[0] = BM, g_last_cfg[1] = BN, g_last_cfg[2] = MODE, g_last_cfg[3] = BKT * 10 + NST, g_last_cfg[4] = S;
  const int64_t npad = (d.N + 7) / 8 * 8;
  size_t lds = (size_t)NST * (BM + BN) * BKT * sizeof(f16);
  const size_t epi = (size_t)BM * BN * sizeof(float) / (BKT == 32 ? 2 : 1);  // epilogue stages the fp32 tile in LDS (two halves for BK32)
  if (lds < epi) lds = epi;
  if (d.rs_in) {   // (mean, rstd) per tile row behind the epilogue's staging tile
    if (S > 1) return TB_EINVAL;
    if (lds < epi + BM * 8) lds = epi + BM * 8;
  }
  static bool attr_done = false;
  if (!attr_done && lds + 1024 > 65536) {
    if (hipFuncSetAttribute((const void*)gemm_kernel<BM, BN, MODE, BKT, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds + 1024)) !=
        hipSuccess)
      return TB_ELAUNCH;
    attr_done = true;
  }
  // tile order inside each XCD's contiguous range (A/B in situ on one MI355X: n-fastest 22.39, m-fastest 21.62, a bytes-to-fabric
  // cost model 22.11, 8-row groups 22.55 steps/s): what matters is that the ~64 blocks co-resident on an XCD share few k-slices
  const int m_fastest = g_order == 0 ? 3 : (g_order == 2 ? 1 : 0);
  tb_gemm_desc dk = d;  // the kernel reduces the k-slices itself when the caller gave it one zeroed counter per tile (tb_gemm_desc.sync)
  const bool defer = defer_reduce(d, S);
  const bool inkernel = S > 1 && !defer && g_inkernel_reduce && d.sync && (int64_t)tiles_m * tiles_n <= d.sync_count;
  if (!inkernel) dk.sync = nullptr;
  hipLaunchKernelGGL((gemm_kernel<BM, BN, MODE, BKT, NST>), dim3((unsigned)(tiles_m * tiles_n * S)), dim3(256), lds, s, dk, tiles_m,
                     tiles_n, S, (float*)d.ws, npad, g_ablate, m_fastest);
  if (defer) *d.split_out = S;
  else if (S > 1 && !inkernel)
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((d.M * (npad / 8) + 255) / 256)), dim3(256), 0, s, d, (const float*)d.ws, S,
                       npad);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

int g_variant64 = -1;  // k-tile depth of the 64x64 tile (tb_gemm_set_variant(10 + v)): -1 = by grid size, 0 = BK64 x 2 stages, 4 = x 3, 5 = x 4

template <int BM, int BN, int MODE>
int launch(const tb_gemm_desc& d, hipStream_t s, int S = 1) {
  if (BM == 64 && BN == 64) {
    // The small tile runs latency-bound GEMMs (a k-tile of MFMA work is ~1/8 of an L2 round trip).  When the grid is at most two
    // blocks per CU the LDS is free for a 4-stage ring (3 k-tiles in flight); larger grids keep 2 stages and 5 blocks per CU
    // (scratch/tile64.py, one MI355X: 512x1280x1280 16.2 -> 14.1 us, 1232x768x768 12.3 -> 10.8; 2048x1280x1280 (640 blocks) 20.7 vs 27.9)
    const int64_t nblk = ((d.M + 63) / 64) * ((d.N + 63) / 64) * S;
    const int v = g_variant64 >= 0 ? g_variant64 : (nblk <= 512 ? 5 : 0);
    switch (v) {
      case 4: return launch_v<BM, BN, MODE, 64, 3>(d, s, S);
      case 5: return launch_v<BM, BN, MODE, 64, 4>(d, s, S);
      default: return launch_v<BM, BN, MODE, 64, 2>(d, s, S);
    }
  }
  switch (g_variant) {
    case 1: return launch_v<BM, BN, MODE, 32, 3>(d, s, S);
    case 2: return launch_v<BM, BN, MODE, 32, 2>(d, s, S);
    case 3: return launch_v<BM, BN, MODE, 32, 4>(d, s, S);
    case 4: return launch_v<BM, BN, MODE, 64, 3>(d, s, S);
    case 5: return launch_v<BM, BN, MODE, 64, 4>(d, s, S);
    case 6: return BN == 64 ? launch_v<BM, BN, MODE, 64, 6>(d, s, S) : launch_v<BM, BN, MODE, 64, 4>(d, s, S);  // one block per CU, deep ring
    default: return launch_v<BM, BN, MODE, 64, 2>(d, s, S);
  }
}

int g_phase = 1;             // stride-2 dgrad gathers walk the rows parity class by parity class (tb_gemm_set_variant(9900 + {0,1}))
int g_conv_narrow = 0;       // experiment: bit 0 = conv_halo_kernel<64> always, bit 1 = convs do NOT follow the linear tile rule (9000 + bits)
int g_force_tile = 0;        // profiling: 1 = 64x64, 2 = 128x64, 3 = 128x128 for every un-split launch (tb_gemm_set_variant(8000 + v))
int g_split_min_tiles = 8;   // k-tiles per slice lower bound (tb_gemm_set_variant(4000 + n))
int g_split_blocks = 256;    // split only when the un-split grid has fewer blocks than this (5000 + n)
int g_split_minnk = 32;      // ... and at least this many k-tiles (6000 + n)
int g_nosplit64 = 1;         // see dispatch_tile (tb_gemm_set_variant(9600 + {0,1}))
int g_split_target = 384;  // split K until about this many blocks exist (A/B: 256 22.7, 384 23.05, 512 22.6, 768 22.6, off 20.2 steps/s)  (tb_gemm_set_variant(1000 + n))

template <int BN, int NSW = 2>
int launch_halo(const tb_gemm_desc& d, hipStream_t s, int wshift, int S) {
  const int tiles_m = (int)(d.M / 128), tiles_n = (int)((d.N + BN - 1) / BN);
  const int TW = 1 << wshift, R = 128 >> wshift;
  const int hw_img = d.Hout * d.Wout;
  const bool perm8 = hw_img < 128 && TW == 8 && (g_halo & 8);
  const int nh8 = ((hw_img < 128 ? (128 / hw_img) * (d.Hout + 2) * (perm8 ? 12 : TW + 2) : (R + 2) * (TW + 2)) + 7) & ~7;
  size_t lds = (size_t)(NSW > 2 ? 2 : 1) * nh8 * 128 + (size_t)NSW * BN * 128;
  if (lds < (size_t)128 * BN * sizeof(float)) lds = (size_t)128 * BN * sizeof(float);
  if (lds > 160 * 1024) return 1;
  g_last_cfg[0] = 128, g_last_cfg[1] = BN, g_last_cfg[2] = 2, g_last_cfg[3] = 640 + NSW, g_last_cfg[4] = S;
  const int64_t npad = (d.N + 7) / 8 * 8;
  static bool attr_done = false;
  if (!attr_done && lds > 65536) {
    if (hipFuncSetAttribute((const void*)conv_halo_kernel<BN, NSW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return TB_ELAUNCH;
    attr_done = true;
  }
  tb_gemm_desc dk = d;
  const bool defer = defer_reduce(d, S);
  const bool inkernel = S > 1 && !defer && g_inkernel_reduce && d.sync && (int64_t)tiles_m * tiles_n <= d.sync_count;
  if (!inkernel) dk.sync = nullptr;
  hipLaunchKernelGGL((conv_halo_kernel<BN, NSW>), dim3((unsigned)(tiles_m * tiles_n * S)), dim3(256), lds, s, dk, tiles_m, tiles_n, wshift, S,
                     (float*)d.ws, npad, perm8 ? 1 : 0);
  if (defer) *d.split_out = S;
  else if (S > 1 && !inkernel)
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((d.M * (npad / 8) + 255) / 256)), dim3(256), 0, s, d, (const float*)d.ws, S,
                       npad);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

// log2 of the halo kernel's tile width for a W x H map, or 0 when it does not apply: the tile is (128 / TW) rows x TW columns with TW the
// largest power of two (8..64) dividing W, and tiles must not straddle images (64 -> 2 x 64, 96 -> 4 x 32, 48 -> 8 x 16, 16 -> 8 x 16)
inline int halo_wshift(int W, int H) {
  int sh = 0;
  while (sh < 6 && !(W & (1 << sh))) ++sh;
  if (W & ((1 << sh) - 1)) return 0;
  if (sh < 3) return 0;
  const int R = 128 >> sh;
  if (H * W < 128) return (W == (1 << sh) && 128 % (H * W) == 0) ? sh : 0;   // whole small images per tile (8x8 maps: two), see conv_halo_kernel
  return (H % R == 0) ? sh : 0;
}

template <int MODE>
int dispatch_tile(const tb_gemm_desc& d, hipStream_t s) {
  // BN = 64 tiles N that is an odd multiple of 64 (320, 960, ...) exactly.  It is also the faster tile whenever the 128x128 grid
  // is large enough not to be split and is not a single full round: 3 blocks of 48 KB LDS fit on a CU instead of 2, and these launches are bound by
  // exposed latency (prologue, per-k-tile drain, residual loads), not by MFMA rate (scratch/tiles.py, one MI355X: 8192x640x640
  // 26.2 -> 21.3 us, 2048x5120x1280 78 -> 57 us, 32768x1280x320 115 -> 95 us; 2048x3840x1280 (480 tiles = one round) 39.8 vs 46.8;
  // halo conv 640->640 @32x32 94.5 -> 81.5 us, 1280->640 168 -> 151 us)
#ifdef TB_NO_NARROW
  bool narrow = false;
  const bool by_rule = false;
#else
  const bool odd64 = (d.N % 128) != 0 && (d.N % 128) <= 64;
  bool narrow = odd64;
  if ((MODE == TB_A_LINEAR || !(g_conv_narrow & 2)) && g_force_tile != 4) {
    const int64_t b128 = ((d.M + 127) / 128) * ((d.N + 127) / 128);
    if (b128 >= 256 && !(b128 >= 384 && b128 <= 512)) narrow = true;
  }
  const bool by_rule = narrow && !odd64;  // such grids are never small enough for the 64x64 fallback below
#endif
  const int64_t blocks = ((d.M + 127) / 128) * ((d.N + (narrow ? 63 : 127)) / (narrow ? 64 : 128));
  if (MODE == TB_A_CONV3X3 && g_halo && d.stride == 1 && !d.upsample && !d.transposed && !d.shift && d.Hin == d.Hout && d.Win == d.Wout &&
      halo_wshift(d.Wout, d.Hout) > 0 && ((int64_t)d.Hout * d.Wout >= 128 ? (g_halo & 1) : ((g_halo & 2) && d.M % 128 == 0))) {
    // too few tiles for 256 CUs: split the 64-channel chunks over S blocks per tile (fp32 partials + the split-K reducer)
    int Sh = 1;
    const int kpt = d.Cin / 64;
    if (blocks < 256 && d.ws) {
      int64_t want = (g_split_target + blocks - 1) / blocks;
      if (want > kpt / 2) want = kpt / 2;  // >= 2 chunks (18 k-tiles) per slice
      const int64_t fit = d.ws_bytes / (int64_t)(d.M * ((d.N + 7) / 8 * 8) * sizeof(float));
      if (want > fit) want = fit;
      if (want > 1) Sh = (int)want;
    }
    const int wshift = halo_wshift(d.Wout, d.Hout);
    if ((int64_t)d.Hout * d.Wout < 128 && !narrow && !(g_conv_narrow & 1) && (g_halo & 4) && d.ws && blocks < 256) {
      // 8x8 maps (round 5): one workgroup per CU on a 5-slot weight ring -- as few k-slices as still give ~200 workgroups (half the fp32 partials)
      int Sd = (int)((220 + blocks - 1) / blocks);
      if (Sd > kpt / 2) Sd = kpt / 2;
      const int64_t fit = d.ws_bytes / (int64_t)(d.M * ((d.N + 7) / 8 * 8) * sizeof(float));
      if (Sd > fit) Sd = (int)fit;
      if (Sd >= 1 && blocks * Sd >= 128) {
        const int r = launch_halo<128, 5>(d, s, wshift, Sd);
        if (r != 1) return r;
      }
    }
    if (blocks * Sh >= 200) {
      return (narrow || (g_conv_narrow & 1)) ? launch_halo<64>(d, s, wshift, Sh) : launch_halo<128>(d, s, wshift, Sh);
    }
  }
  int S = 1;
  if (blocks < 384 && d.ws && g_split_target > 0) {
    // too few tiles to fill 256 CUs: split K across blocks (fp32 partials in ws, fixed-order reduction -> deterministic)
    const int64_t nk = d.K / 64;
    const int64_t npad = (d.N + 7) / 8 * 8;
    // short K: a second pass costs more.  When the un-split 64x64 grid already has >= 2 blocks per CU only very long K pays
    // (2048x1280: K=2560 30.6 us un-split vs 36.1 split, K=3840 41.0 vs 44.1, K=5120 62.3 vs 50.1)
    const int64_t minnk = (!narrow && blocks * 4 >= 512 && g_split_minnk == 32) ? 80 : g_split_minnk;
    int64_t want = blocks < g_split_blocks && nk >= minnk ? (g_split_target + blocks - 1) / blocks : 1;
    if (want > nk / g_split_min_tiles) want = nk / g_split_min_tiles;
    if (want > 16) want = 16;
    const int64_t fit = d.ws_bytes / (int64_t)(d.M * npad * sizeof(float));
    if (want > fit) want = fit;
    if (want > 1) S = (int)want;
    // Long-K Linear layers whose un-split 64x64 grid is most of one chip round (224..512 blocks: the text encoder's N = 768 projections at
    // M = 1232 / 1848, K = 2368 / 3072) run faster on the 4-stage 64x64 ring than as 128x128 split-K slices plus the reducer launch
    // (scratch/te_gemm_ab.py, cold weights: 1848x768x3072 29.2 -> 24.6 us, 1232x768x3072 25.2 -> 20.9, 1232x768x2368 23.2 -> 17.4; smaller grids --
    // 512x1280x5120: 160 blocks -- and larger ones -- 2048x1280x5120: 640 blocks, 2 stages -- keep the split: 27.3 vs 30.9, 47.0 vs 67.5 us)
    if (MODE == TB_A_LINEAR && S > 1 && g_nosplit64) {
      const int64_t b64 = ((d.M + 63) / 64) * ((d.N + 63) / 64);
      if (b64 >= 224 && b64 <= 512) return launch<64, 64, MODE>(d, s);
    }
  }
#ifndef TB_SMALL_TILE_BLOCKS
#define TB_SMALL_TILE_BLOCKS 320  // A/B on one MI355X: 96 -1.6 %, 192 base, 320 +1.0 %, 512 -1.0 % steps/s
#endif
  if (g_force_tile && g_force_tile < 4 && S == 1)
    return g_force_tile == 1 ? launch<64, 64, MODE>(d, s) : (g_force_tile == 2 ? launch<128, 64, MODE>(d, s) : launch<128, 128, MODE>(d, s));
  if (S == 1 && !by_rule && blocks < TB_SMALL_TILE_BLOCKS) return launch<64, 64, MODE>(d, s);  // small problems without workspace: smaller tiles
  return narrow ? launch<128, 64, MODE>(d, s, S) : launch<128, 128, MODE>(d, s, S);
}

}  // namespace

extern "C" void tb_gemm_last_config(int* out5) {
  for (int i = 0; i < 5; ++i) out5[i] = g_last_cfg[i];
}

extern "C" int tb_gemm_set_variant(int v) {
  const int old = g_variant;
  if (v >= 9900) g_phase = v - 9900;
  else if (v >= 9400 && v < 9500) tb_lin320_set(v - 9400);
  else if (v >= 9600 && v < 9700) g_nosplit64 = v - 9600;
  else if (v >= 9800) g_inkernel_reduce = v - 9800;
  else if (v >= 9000) g_conv_narrow = v - 9000;
  else if (v >= 8000) g_force_tile = v - 8000;
  else if (v >= 7000) g_halo = v - 7000;
  else if (v >= 6000) g_split_minnk = v - 6000;
  else if (v >= 5000) g_split_blocks = v - 5000;
  else if (v >= 4000) g_split_min_tiles = v - 4000;
  else if (v >= 3000) g_order = v - 3000;
  else if (v >= 2000) g_ablate = v - 2000;
  else if (v >= 1000) g_split_target = v - 1000;  // 1000 disables split-K, 1512 = default target of 512 blocks
  else if (v >= 9 && v < 20) g_variant64 = v - 10;  // 9 = automatic
  else g_variant = v;
  return old;
}

int tb_gemm8_try(const tb_gemm_desc& d, hipStream_t s, int* split_out);  // gemm8.hip: 8-wave wide tiles for the large-M levels; 1 = shape not covered;
                                                                         // *split_out = k-slices of the launch (> 1: fp32 partials in d.ws)

extern "C" int tb_gemm(const tb_gemm_desc* dp, tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!dp) return TB_EINVAL;
  tb_gemm_desc d = *dp;
  hipStream_t s = (hipStream_t)stream;
  if (d.M <= 0 || d.N <= 0 || d.K <= 0 || d.K % BK) return TB_EINVAL;
  if (!d.A || !d.W || !d.C) return TB_EINVAL;
  if (d.split_out) *d.split_out = 1;
  if (!d.A2 && !d.W2) d.K1 = d.K;
  if (d.K1 % BK || d.K1 > d.K || d.K1 <= 0) return TB_EINVAL;
  if ((d.A2 == nullptr) != (d.W2 == nullptr)) return TB_EINVAL;
  if (d.lda % 8 || d.ldw % 8 || (d.A2 && (d.lda2 % 8 || d.ldw2 % 8))) return TB_EINVAL;  // 16-byte vector loads
  if (((uintptr_t)d.A) % 16 || ((uintptr_t)d.W) % 16 || ((uintptr_t)d.A2) % 16 || ((uintptr_t)d.W2) % 16) return TB_EINVAL;
  {  // 32-bit per-lane byte offsets inside the kernels: every operand row range must lie below 4 GiB from its base
    const int64_t lim = (int64_t)1 << 32;
    const int64_t a_rows = d.a_mode == TB_A_LINEAR ? d.M : 0;
    if (a_rows && ((a_rows - 1) * d.lda + d.K1) * 2 >= lim) return TB_EINVAL;
    if (d.A2 && ((d.M - 1) * d.lda2 + (d.K - d.K1)) * 2 >= lim) return TB_EINVAL;
    if (((d.N - 1) * d.ldw + d.K1) * 2 >= lim) return TB_EINVAL;
    if (d.W2 && ((d.N - 1) * d.ldw2 + (d.K - d.K1)) * 2 >= lim) return TB_EINVAL;
  }
  if (d.rowbias && d.rows_per_group <= 0) return TB_EINVAL;
  if (d.rowbias && d.ldrb < d.N) d.ldrb = d.N;
  if ((d.act == TB_ACT_QUICK_GELU_GRAD || d.act == TB_ACT_GELU_GRAD || d.act == TB_ACT_GEGLU_GRAD) && !d.C2) return TB_EINVAL;
  if (d.act == TB_ACT_GEGLU_GRAD && (d.N % 32 || d.c_dtype != TB_F16)) return TB_EINVAL;
  if (d.act < 0 || d.act > TB_ACT_LN_BWD) return TB_EINVAL;
  if (d.a_mode == TB_A_CONV3X3 && (d.upsample == 2 || d.upsample == 3)) {
    // sub-pixel form of nearest-x2 + conv3x3 (gemm8.hip): upsample == 2 forward (A coarse [B, Hin, Win, Cin], C fine [B, 2 Hin, 2 Win, N],
    // W [4 classes][N][4 taps][Cin], K = 4 Cin), upsample == 3 dgrad (A fine [B, Hin, Win, Cin], C coarse [B, Hin / 2, Win / 2, N], W [N][16 Cin])
    if (d.A2 || d.Cin <= 0 || d.Cin % BK || d.K != (d.upsample == 2 ? 4 : 16) * (int64_t)d.Cin) return TB_EINVAL;
    if (d.M != (int64_t)d.B * d.Hout * d.Wout || d.stride != 1 || d.transposed || d.shift || d.rowbias || d.C2 || d.act != TB_ACT_NONE ||
        d.c_dtype != TB_F16)
      return TB_EINVAL;
    // a residual rides in the forward form only (the stride-2 convolution's dgrad written as a sub-pixel convolution adds the skip gradient), as
    // 16-byte fp16 rows
    if (d.R && (d.upsample != 2 || d.r_dtype != TB_F16 || d.ldr % 8 || ((uintptr_t)d.R) % 16)) return TB_EINVAL;
    if (d.upsample == 2 ? (d.Hout != 2 * d.Hin || d.Wout != 2 * d.Win) : (d.Hin != 2 * d.Hout || d.Win != 2 * d.Wout)) return TB_EINVAL;
  } else if (d.a_mode == TB_A_CONV3X3) {
    if (d.A2 || d.Cin <= 0 || d.Cin % BK || d.K != 9 * (int64_t)d.Cin) return TB_EINVAL;
    if (d.M != (int64_t)d.B * d.Hout * d.Wout) return TB_EINVAL;
    if (d.sign != 1 && d.sign != -1) return TB_EINVAL;
    if (d.stride < 1 || (d.shift && (d.upsample || d.transposed))) return TB_EINVAL;
  } else if (d.a_mode != TB_A_LINEAR) {
    return TB_EINVAL;
  }
  if (d.a_mode == TB_A_CONV3X3 && d.transposed) {
    // phase mode of the transposed gather (gemm_kernel): even maps whose quarter is a whole number of 128-row tiles, no per-row-group bias
    d.transposed = (g_phase && !(d.Hout & 1) && !(d.Wout & 1) && (d.M % 512) == 0 && !d.rowbias && d.Hin * 2 >= d.Hout && d.Win * 2 >= d.Wout) ? 2 : 1;
  }
  {
    const int r3 = tb_lin320_try(d, s);
    if (r3 != 1) {
      tb_gemm8_clear_last();
      g_last_cfg[0] = 128, g_last_cfg[1] = 64, g_last_cfg[2] = 3, g_last_cfg[3] = 643, g_last_cfg[4] = 1;   // (mode 3 = lin320_kernel)
      return r3;
    }
  }
  {
    int split8 = 1;
    const int r8 = tb_gemm8_try(d, s, &split8);
    if (r8 == TB_OK && defer_reduce(d, split8)) {
      *d.split_out = split8;
      return TB_OK;
    }
    if (r8 == TB_OK && split8 > 1) {
      const int64_t npad = (d.N + 7) & ~(int64_t)7;
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((d.M * (npad / 8) + 255) / 256)), dim3(256), 0, s, d, (const float*)d.ws,
                         split8, npad);
      TB_CHECK_LAUNCH();
      return TB_OK;
    }
    if (r8 != 1) return r8;
  }
  if (d.act == TB_ACT_LN_FWD || d.act == TB_ACT_LN_BWD) return TB_EINVAL;  // exist in the row-spanning wide tiles only
  if (d.rs_out) return TB_EINVAL;                                           // row statistics are written by the 8-wave Linear tiles only
  if (d.rs_in) {   // folded LayerNorm (consumer): the lean epilogue of the un-split 4-wave tiles (the 16x16-map qkv projection)
    if (d.act != TB_ACT_NONE || d.a_mode != TB_A_LINEAR || d.A2 || d.C2 || d.rowbias || d.c_dtype != TB_F16 || d.ldc % 8 || ((uintptr_t)d.C) % 16 ||
        (d.R && (d.ldr % 8 || ((uintptr_t)d.R) % 16)) || d.N % 8 || !d.ln_gamma || d.rs_n < 1 || d.rs_n > 16 || d.rs_ld < d.rs_n ||
        ((uintptr_t)d.rs_in) % 8 || ((uintptr_t)d.ln_stats) % 8)
      return TB_EINVAL;
    // the fold lives in the lean (`fast`) branch of tile_epilogue only; the conditions above imply it unless a profiling knob switches that branch
    // or whole phases off (tb_gemm_set_variant(2000 + bits)): refuse rather than store un-normalised x W'^T + c2
    if (g_ablate) return TB_EINVAL;
    d.ws = nullptr, d.ws_bytes = 0;   // never split K: the reducer does not know the fold
  }
  // the sub-pixel descs exist in the 8-wave kernel only: the 4-wave kernel would read `upsample != 0` as the folded 9-tap gather (wrong K / weight layout)
  if (d.a_mode == TB_A_CONV3X3 && (d.upsample == 2 || d.upsample == 3)) return TB_EINVAL;
  if (d.act == TB_ACT_GEGLU) {
    if (d.N % 128 || d.R || d.rowbias || d.c_dtype != TB_F16) return TB_EINVAL;
    if (d.a_mode != TB_A_LINEAR) return TB_EINVAL;
    // packed [h32|g32] column blocks of 64 fit both tile widths; the narrow tile only pays for the shortest K (one MI355X:
    // 32768x2560x320 207 -> 182 us, but 8192x5120x640 134 -> 145 us, 2048x10240x1280 109 -> 118 us)
    const int64_t b128 = ((d.M + 127) / 128) * (d.N / 128);
    const bool narrow = g_force_tile != 4 && g_force_tile != 3 && b128 > 512 && d.K <= 320;
    return narrow ? launch<128, 64, TB_A_LINEAR>(d, s, 1) : launch<128, 128, TB_A_LINEAR>(d, s, 1);
  }
  return d.a_mode == TB_A_LINEAR ? dispatch_tile<TB_A_LINEAR>(d, s) : dispatch_tile<TB_A_CONV3X3>(d, s);
}
