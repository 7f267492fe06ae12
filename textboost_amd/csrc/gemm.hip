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

namespace {

constexpr int BK = 64;  // halfs per k-tile (128 B per tile row)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __attribute__((aligned(128))) const f16 g_zero_line[64] = {};

__device__ __forceinline__ void glds16(const f16* src, f16* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}

template <int BM, int BN, int MODE>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const tb_gemm_desc p, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(128))) unsigned char smem_raw[];
  f16* smem = reinterpret_cast<f16*>(smem_raw);
  constexpr int A_TILE = BM * BK, B_TILE = BN * BK, STAGE = A_TILE + B_TILE;

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  constexpr int WTM = BM / 2, WTN = BN / 2, TM = WTM / 32, TN = WTN / 32;
  constexpr int AI = BM / 32, BI = BN / 32;  // global_load_lds instructions per wave per k-tile (8 rows each)

  // ---- XCD-aware tile order: blocks b, b+8, b+16, ... share an XCD (and its L2); give each XCD a contiguous run of tiles
  const int nwg = tiles_m * tiles_n;
  int tile;
  {
    const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int64_t m0 = (int64_t)(tile / tiles_n) * BM;
  const int64_t n0 = (int64_t)(tile % tiles_n) * BN;

  const int cp = lane & 7;   // 16-byte chunk position inside the 128-byte LDS row this lane fills
  const int rl = lane >> 3;  // row within the 8-row group of one load instruction

  // ---- per-lane source descriptors (rows are fixed across k-tiles)
  const f16* a_ptr[AI];
  const f16* a2_ptr[AI];
  bool a_ok[AI];
  int py[AI], px[AI], a_sw[AI];
  int64_t pbase[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int row = wave * (BM / 4) + i * 8 + rl;
    const int64_t m = m0 + row;
    a_ok[i] = m < p.M;
    const int64_t mm = a_ok[i] ? m : 0;
    a_sw[i] = (cp ^ ((row >> 1) & 7)) * 8;  // source chunk (halfs) that belongs at LDS chunk position cp of this row
    if (MODE == TB_A_LINEAR) {
      a_ptr[i] = (const f16*)p.A + mm * p.lda + a_sw[i];
      a2_ptr[i] = p.A2 ? (const f16*)p.A2 + mm * p.lda2 + a_sw[i] : nullptr;
    } else {
      const int hw = p.Hout * p.Wout;
      const int b = (int)(mm / hw);
      const int rem = (int)(mm - (int64_t)b * hw);
      py[i] = rem / p.Wout;
      px[i] = rem - py[i] * p.Wout;
      pbase[i] = (int64_t)b * p.Hin * p.Win;
    }
  }
  const f16* w_ptr[BI];
  const f16* w2_ptr[BI];
  bool w_ok[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int row = wave * (BN / 4) + i * 8 + rl;
    const int64_t n = n0 + row;
    w_ok[i] = n < p.N;
    const int64_t nn = w_ok[i] ? n : 0;
    const int sw = (cp ^ ((row >> 1) & 7)) * 8;
    w_ptr[i] = (const f16*)p.W + nn * p.ldw + sw;
    w2_ptr[i] = p.W2 ? (const f16*)p.W2 + nn * p.ldw2 + sw : nullptr;
  }

  const int nk = (int)(p.K / BK);
  const int nk1 = (int)(p.K1 / BK);
  const int kpt = (MODE == TB_A_CONV3X3) ? p.Cin / BK : 1;  // k-tiles per tap
  const f16* zero = g_zero_line;

  auto stage = [&](int kt, int buf) {
    f16* As = smem + buf * STAGE + (wave * (BM / 4)) * BK;
    f16* Bs = smem + buf * STAGE + A_TILE + (wave * (BN / 4)) * BK;
    if (MODE == TB_A_LINEAR) {
      const bool second = kt >= nk1;
      const int koff = (second ? kt - nk1 : kt) * BK;
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        const f16* src = a_ok[i] ? (second ? a2_ptr[i] : a_ptr[i]) + koff : zero;
        glds16(src, As + i * 8 * BK);
      }
    } else {
      const int tap = kt / kpt;
      const int cc = kt - tap * kpt;
      const int ky = tap / 3, kx = tap - ky * 3;
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
          sy = py[i] * p.stride + p.sign * (ky - 1);
          sx = px[i] * p.stride + p.sign * (kx - 1);
          ok = sy >= 0 && sx >= 0 && sy < p.Hin && sx < p.Win;
        }
        ok = ok && a_ok[i];
        const f16* src = ok ? (const f16*)p.A + (pbase[i] + (int64_t)sy * p.Win + sx) * p.lda + cc * BK + a_sw[i] : zero;
        glds16(src, As + i * 8 * BK);
      }
    }
    {
      const bool second = kt >= nk1;
      const int koff = (second ? kt - nk1 : kt) * BK;
#pragma unroll
      for (int i = 0; i < BI; ++i) {
        const f16* src = w_ok[i] ? (second ? w2_ptr[i] : w_ptr[i]) + koff : zero;
        glds16(src, Bs + i * 8 * BK);
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int l31 = lane & 31, hi = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(kt + 1, cur ^ 1);
    const f16* Ab = smem + cur * STAGE + (wm * WTM) * BK;
    const f16* Bb = smem + cur * STAGE + A_TILE + (wn * WTN) * BK;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      f16x8 af[TM], bf[TN];
      const int ch = kk * 2 + hi;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = i * 32 + l31;
        af[i] = *(const f16x8*)(Ab + row * BK + ((ch ^ ((row >> 1) & 7)) << 3));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = j * 32 + l31;
        bf[j] = *(const f16x8*)(Bb + row * BK + ((ch ^ ((row >> 1) & 7)) << 3));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)  // transposed accumulator: rows = n (from W), cols = m (from A)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue: lane owns row m = .. + l31 and, per register quad r4, columns n = .. + 8*r4 + 4*hi + {0,1,2,3}
  const float alpha = p.alpha;
  const bool c_vec = (p.ldc % 4 == 0) && (((uintptr_t)p.C) % 16 == 0);
  if (p.act == TB_ACT_GEGLU) {
    if constexpr (TN == 2) {
      const bool c2_vec = p.C2 && (p.ldc2 % 4 == 0) && (((uintptr_t)p.C2) % 8 == 0);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int64_t m = m0 + wm * WTM + i * 32 + l31;
        if (m >= p.M) continue;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int nl = 8 * r4 + 4 * hi;                 // 0..31 inside the 32-wide h (and g) block
          const int64_t nh = n0 + wn * WTN + nl;          // packed column of h; g is nh + 32
          const int64_t nout = (n0 + wn * WTN) / 2 + nl;
          f16x4 oh, og, oo;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float vh = alpha * acc[i][0][4 * r4 + e] + (p.bias ? p.bias[nh + e] : 0.f);
            const float vg = alpha * acc[i][1][4 * r4 + e] + (p.bias ? p.bias[nh + 32 + e] : 0.f);
            oh[e] = (f16)vh;
            og[e] = (f16)vg;
            // gate on the fp16-rounded projections, as a fp16 module would (diffusers GEGLU on fp16 tensors)
            oo[e] = (f16)((float)oh[e] * gelu_erf_f((float)og[e]));
          }
          if (p.C2) {
            f16* c2 = (f16*)p.C2 + m * p.ldc2;
            if (c2_vec) {
              *(f16x4*)(c2 + nh) = oh;
              *(f16x4*)(c2 + nh + 32) = og;
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                c2[nh + e] = oh[e];
                c2[nh + 32 + e] = og[e];
              }
            }
          }
          f16* c = (f16*)p.C + m * p.ldc + nout;
          if (c_vec) *(f16x4*)c = oo;
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e) c[e] = oo[e];
          }
        }
      }
    }
    return;
  }
  const bool r_vec = p.R && (p.ldr % 4 == 0) && (((uintptr_t)p.R) % 16 == 0);
  const bool c2_vec = p.C2 && (p.ldc2 % 4 == 0) && (((uintptr_t)p.C2) % 8 == 0);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int64_t m = m0 + wm * WTM + i * 32 + l31;
    if (m >= p.M) continue;
    const float* rb = p.rowbias ? p.rowbias + (m / p.rows_per_group) * p.ldrb : nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int64_t n = n0 + wn * WTN + j * 32 + 8 * r4 + 4 * hi;
        if (n >= p.N) continue;
        const bool full = n + 3 < p.N;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = alpha * acc[i][j][4 * r4 + e];
          if (full || n + e < p.N) {
            if (p.bias) v[e] += p.bias[n + e];
            if (rb) v[e] += rb[n + e];
          }
        }
        if (p.R) {
          if (p.r_dtype == TB_F32) {
            const float* rp = (const float*)p.R + m * p.ldr + n;
            if (full && r_vec) {
              const f32x4 rv = *(const f32x4*)rp;
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += rv[e];
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (n + e < p.N) v[e] += rp[e];
            }
          } else {
            const f16* rp = (const f16*)p.R + m * p.ldr + n;
            if (full && r_vec) {
              const f16x4 rv = *(const f16x4*)rp;
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (n + e < p.N) v[e] += (float)rp[e];
            }
          }
        }
        if (p.act == TB_ACT_QUICK_GELU) {
          f16x4 pre;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            pre[e] = (f16)v[e];
            v[e] = quick_gelu_f((float)pre[e]);  // fp16 linear output feeds the activation, as under autocast
          }
          if (p.C2) {
            f16* c2 = (f16*)p.C2 + m * p.ldc2 + n;
            if (full && c2_vec) *(f16x4*)c2 = pre;
            else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (n + e < p.N) c2[e] = pre[e];
            }
          }
        } else if (p.act == TB_ACT_SILU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
        } else if (p.act == TB_ACT_QUICK_GELU_GRAD) {
          const f16* c2 = (const f16*)p.C2 + m * p.ldc2 + n;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < p.N) v[e] *= quick_gelu_grad_f((float)c2[e]);
        }
        if (p.c_dtype == TB_F32) {
          float* c = (float*)p.C + m * p.ldc + n;
          if (full && c_vec) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = v[e];
            *(f32x4*)c = o;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n + e < p.N) c[e] = v[e];
          }
        } else {
          f16* c = (f16*)p.C + m * p.ldc + n;
          if (full && c_vec) {
            f16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (f16)v[e];
            *(f16x4*)c = o;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n + e < p.N) c[e] = (f16)v[e];
          }
        }
      }
  }
}

template <int BM, int BN, int MODE>
int launch(const tb_gemm_desc& d, hipStream_t s) {
  const int tiles_m = (int)((d.M + BM - 1) / BM), tiles_n = (int)((d.N + BN - 1) / BN);
  const size_t lds = 2 * (size_t)(BM + BN) * BK * sizeof(f16);
  hipLaunchKernelGGL((gemm_kernel<BM, BN, MODE>), dim3((unsigned)(tiles_m * tiles_n)), dim3(256), lds, s, d, tiles_m, tiles_n);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

template <int MODE>
int dispatch_tile(const tb_gemm_desc& d, hipStream_t s) {
  // N that is an odd multiple of 64 (320, 960, ...) tiles exactly with BN = 64
  const bool narrow = (d.N % 128) != 0 && (d.N % 128) <= 64;
  const int64_t blocks128 = ((d.M + 127) / 128) * ((d.N + (narrow ? 63 : 127)) / (narrow ? 64 : 128));
  if (blocks128 < 192) return launch<64, 64, MODE>(d, s);  // small problems: more, smaller tiles to fill 256 CUs
  return narrow ? launch<128, 64, MODE>(d, s) : launch<128, 128, MODE>(d, s);
}

}  // namespace

extern "C" int tb_gemm(const tb_gemm_desc* dp, tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!dp) return TB_EINVAL;
  tb_gemm_desc d = *dp;
  hipStream_t s = (hipStream_t)stream;
  if (d.M <= 0 || d.N <= 0 || d.K <= 0 || d.K % BK) return TB_EINVAL;
  if (!d.A || !d.W || !d.C) return TB_EINVAL;
  if (!d.A2 && !d.W2) d.K1 = d.K;
  if (d.K1 % BK || d.K1 > d.K || d.K1 <= 0) return TB_EINVAL;
  if ((d.A2 == nullptr) != (d.W2 == nullptr)) return TB_EINVAL;
  if (d.lda % 8 || d.ldw % 8 || (d.A2 && (d.lda2 % 8 || d.ldw2 % 8))) return TB_EINVAL;  // 16-byte vector loads
  if (((uintptr_t)d.A) % 16 || ((uintptr_t)d.W) % 16 || ((uintptr_t)d.A2) % 16 || ((uintptr_t)d.W2) % 16) return TB_EINVAL;
  if (d.rowbias && d.rows_per_group <= 0) return TB_EINVAL;
  if (d.rowbias && d.ldrb < d.N) d.ldrb = d.N;
  if (d.act == TB_ACT_QUICK_GELU_GRAD && !d.C2) return TB_EINVAL;
  if (d.act < 0 || d.act > TB_ACT_QUICK_GELU_GRAD) return TB_EINVAL;
  if (d.split_k > 1) return TB_EINVAL;
  if (d.a_mode == TB_A_CONV3X3) {
    if (d.A2 || d.Cin <= 0 || d.Cin % BK || d.K != 9 * (int64_t)d.Cin) return TB_EINVAL;
    if (d.M != (int64_t)d.B * d.Hout * d.Wout) return TB_EINVAL;
    if (d.sign != 1 && d.sign != -1) return TB_EINVAL;
    if (d.stride < 1) return TB_EINVAL;
  } else if (d.a_mode != TB_A_LINEAR) {
    return TB_EINVAL;
  }
  if (d.act == TB_ACT_GEGLU) {
    if (d.N % 128 || d.R || d.rowbias || d.c_dtype != TB_F16) return TB_EINVAL;
    return d.a_mode == TB_A_LINEAR ? launch<128, 128, TB_A_LINEAR>(d, s) : TB_EINVAL;
  }
  return d.a_mode == TB_A_LINEAR ? dispatch_tile<TB_A_LINEAR>(d, s) : dispatch_tile<TB_A_CONV3X3>(d, s);
}
