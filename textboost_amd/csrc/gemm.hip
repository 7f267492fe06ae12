// MFMA GEMM family for gfx950: C[M,N] = A[M,K] * W[N,K]^T with fused epilogues.
//
// One kernel body serves nn.Linear / 1x1 conv (A rows read linearly) and 3x3 conv as implicit GEMM over
// NHWC activations (A rows gathered per tap, incl. stride-2, nearest-x2-upsample-folded and
// stride-2-transposed (dgrad) gathers).  dgrad of every layer is the same kernel on host-pre-transposed
// frozen weights, so no NN/TN variants exist.
//
// Structure (v1): BMxBNx64 block tile, 256 threads = 4 waves (2x2), each wave (BM/2)x(BN/2) as 32x32x16 f16
// MFMA tiles accumulating in fp32; global->register->LDS staging (needed for the gather), LDS double
// buffered with ONE barrier per k-tile, 16-byte chunks XOR-swizzled by (row & 7) so ds_read_b128 fragment
// reads are <= 2-way bank conflicted (cdna guide T2).
#include "common.h"
#include "../../include/textboost_hip.h"

namespace {

constexpr int BK = 64;  // halfs per k-tile (128 B per tile row)

template <int BM, int BN, int MODE>
__global__ __launch_bounds__(256) void gemm_kernel(const tb_gemm_desc p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  f16* smem = reinterpret_cast<f16*>(smem_raw);
  constexpr int A_TILE = BM * BK, B_TILE = BN * BK;
  f16* As[2] = {smem, smem + A_TILE + B_TILE};
  f16* Bs[2] = {smem + A_TILE, smem + 2 * A_TILE + B_TILE};

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  constexpr int WTM = BM / 2, WTN = BN / 2, TM = WTM / 32, TN = WTN / 32;
  constexpr int AR = BM / 32, BR = BN / 32;  // rows per thread in the staging pass

  const int tiles_n = (int)((p.N + BN - 1) / BN);
  const int64_t m0 = (int64_t)(blockIdx.x / tiles_n) * BM;
  const int64_t n0 = (int64_t)(blockIdx.x % tiles_n) * BN;

  const int c = t & 7;    // 16-byte chunk within the 128-byte tile row
  const int r0 = t >> 3;  // 0..31

  // ---- per-thread A row descriptors
  const f16* a_ptr[AR];   // linear: row base pointer (first K source)
  const f16* a2_ptr[AR];
  bool a_ok[AR];
  int py[AR], px[AR];
  int64_t pbase[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    int64_t m = m0 + r0 + 32 * i;
    a_ok[i] = m < p.M;
    int64_t mm = a_ok[i] ? m : 0;
    if (MODE == TB_A_LINEAR) {
      a_ptr[i] = (const f16*)p.A + mm * p.lda + c * 8;
      a2_ptr[i] = p.A2 ? (const f16*)p.A2 + mm * p.lda2 + c * 8 : nullptr;
    } else {
      int hw = p.Hout * p.Wout;
      int b = (int)(mm / hw);
      int rem = (int)(mm - (int64_t)b * hw);
      py[i] = rem / p.Wout;
      px[i] = rem - py[i] * p.Wout;
      pbase[i] = (int64_t)b * p.Hin * p.Win;
    }
  }
  const f16* w_ptr[BR];
  const f16* w2_ptr[BR];
  bool w_ok[BR];
#pragma unroll
  for (int i = 0; i < BR; ++i) {
    int64_t n = n0 + r0 + 32 * i;
    w_ok[i] = n < p.N;
    int64_t nn = w_ok[i] ? n : 0;
    w_ptr[i] = (const f16*)p.W + nn * p.ldw + c * 8;
    w2_ptr[i] = p.W2 ? (const f16*)p.W2 + nn * p.ldw2 + c * 8 : nullptr;
  }

  const int nk = (int)(p.K / BK);
  const int nk1 = (int)(p.K1 / BK);
  const int kpt = (MODE == TB_A_CONV3X3) ? p.Cin / BK : 1;  // k-tiles per tap

  f16x8 a_reg[AR], b_reg[BR];
  const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  auto load_tiles = [&](int kt) {
    if (MODE == TB_A_LINEAR) {
      const bool second = kt >= nk1;
      const int koff = (second ? kt - nk1 : kt) * BK;
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        const f16* src = (second ? a2_ptr[i] : a_ptr[i]) + koff;
        a_reg[i] = a_ok[i] ? *(const f16x8*)src : zero8;
      }
    } else {
      const int tap = kt / kpt;
      const int cc = kt - tap * kpt;
      const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        int sy, sx;
        bool ok;
        if (p.upsample) {
          int uy = py[i] + ky - 1, ux = px[i] + kx - 1;
          ok = uy >= 0 && ux >= 0 && uy < 2 * p.Hin && ux < 2 * p.Win;
          sy = uy >> 1;
          sx = ux >> 1;
        } else if (p.transposed) {
          int ty = py[i] + 1 - ky, tx = px[i] + 1 - kx;
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
        const f16* src = (const f16*)p.A + (pbase[i] + (int64_t)sy * p.Win + sx) * p.lda + cc * BK + c * 8;
        a_reg[i] = ok ? *(const f16x8*)src : zero8;
      }
    }
    {
      const bool second = kt >= nk1;
      const int koff = (second ? kt - nk1 : kt) * BK;
#pragma unroll
      for (int i = 0; i < BR; ++i) {
        const f16* src = (second ? w2_ptr[i] : w_ptr[i]) + koff;
        b_reg[i] = w_ok[i] ? *(const f16x8*)src : zero8;
      }
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      int row = r0 + 32 * i;
      *(f16x8*)(As[buf] + row * BK + ((c ^ (row & 7)) << 3)) = a_reg[i];
    }
#pragma unroll
    for (int i = 0; i < BR; ++i) {
      int row = r0 + 32 * i;
      *(f16x8*)(Bs[buf] + row * BK + ((c ^ (row & 7)) << 3)) = b_reg[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  load_tiles(0);
  store_tiles(0);
  __syncthreads();

  const int l31 = lane & 31, hi = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tiles(kt + 1);
    const f16* Ab = As[cur] + (wm * WTM) * BK;
    const f16* Bb = Bs[cur] + (wn * WTN) * BK;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      f16x8 af[TM], bf[TN];
      const int ch = kk * 2 + hi;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        int row = i * 32 + l31;
        af[i] = *(const f16x8*)(Ab + row * BK + ((ch ^ (row & 7)) << 3));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        int row = j * 32 + l31;
        bf[j] = *(const f16x8*)(Bb + row * BK + ((ch ^ (row & 7)) << 3));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tiles(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue
  const float alpha = p.alpha;
  if (p.act == TB_ACT_GEGLU) {
    if constexpr (TN == 2) {
      const int64_t nh = n0 + wn * WTN + l31;  // packed column of h; g is nh + 32
      const int64_t nout = (n0 + wn * WTN) / 2 + l31;
      const float bh = p.bias ? p.bias[nh] : 0.f, bg = p.bias ? p.bias[nh + 32] : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int64_t m = m0 + wm * WTM + i * 32 + mfma32_row(r, hi);
          if (m < p.M) {
            float vh = alpha * acc[i][0][r] + bh, vg = alpha * acc[i][1][r] + bg;
            if (p.C2) {
              f16* c2 = (f16*)p.C2 + m * p.ldc2;
              c2[nh] = (f16)vh;
              c2[nh + 32] = (f16)vg;
            }
            // gate on the fp16-rounded projections, as a fp16 module would (diffusers GEGLU on fp16 tensors)
            float o = (float)(f16)vh * gelu_erf_f((float)(f16)vg);
            ((f16*)p.C)[m * p.ldc + nout] = (f16)o;
          }
        }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int64_t n = n0 + wn * WTN + j * 32 + l31;
    if (n >= p.N) continue;
    const float bn = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int64_t m = m0 + wm * WTM + i * 32 + mfma32_row(r, hi);
        if (m >= p.M) continue;
        float v = alpha * acc[i][j][r] + bn;
        if (p.rowbias) v += p.rowbias[(m / p.rows_per_group) * p.ldrb + n];
        if (p.R) v += (p.r_dtype == TB_F32) ? ((const float*)p.R)[m * p.ldr + n] : (float)((const f16*)p.R)[m * p.ldr + n];
        if (p.act == TB_ACT_QUICK_GELU) {
          if (p.C2) ((f16*)p.C2)[m * p.ldc2 + n] = (f16)v;
          v = quick_gelu_f((float)(f16)v);  // fp16 linear output feeds the activation, as under autocast
        } else if (p.act == TB_ACT_SILU) {
          v = silu_f(v);
        } else if (p.act == TB_ACT_QUICK_GELU_GRAD) {
          v *= quick_gelu_grad_f((float)((const f16*)p.C2)[m * p.ldc2 + n]);
        }
        if (p.c_dtype == TB_F32) ((float*)p.C)[m * p.ldc + n] = v;
        else ((f16*)p.C)[m * p.ldc + n] = (f16)v;
      }
  }
}

template <int BM, int BN, int MODE>
int launch(const tb_gemm_desc& d, hipStream_t s) {
  int64_t tiles = ((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN);
  size_t lds = 2 * (size_t)(BM + BN) * BK * sizeof(f16);
  hipLaunchKernelGGL((gemm_kernel<BM, BN, MODE>), dim3((unsigned)tiles), dim3(256), lds, s, d);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

}  // namespace

extern "C" int tb_gemm(const tb_gemm_desc* dp, tb_stream_t stream) {
  if (!dp) return TB_EINVAL;
  tb_gemm_desc d = *dp;
  hipStream_t s = (hipStream_t)stream;
  if (d.M <= 0 || d.N <= 0 || d.K <= 0 || d.K % BK) return TB_EINVAL;
  if (!d.A || !d.W || !d.C) return TB_EINVAL;
  if (!d.A2 && !d.W2) d.K1 = d.K;
  if (d.K1 % BK || d.K1 > d.K || d.K1 <= 0) return TB_EINVAL;
  if ((d.A2 == nullptr) != (d.W2 == nullptr)) return TB_EINVAL;
  if (d.lda % 8 || d.ldw % 8 || (d.A2 && (d.lda2 % 8 || d.ldw2 % 8))) return TB_EINVAL;  // 16-byte vector loads
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
  const bool narrow = (d.N % 128) != 0 && (d.N % 128) <= 64;
  if (d.a_mode == TB_A_LINEAR) return narrow ? launch<128, 64, TB_A_LINEAR>(d, s) : launch<128, 128, TB_A_LINEAR>(d, s);
  return narrow ? launch<128, 64, TB_A_CONV3X3>(d, s) : launch<128, 128, TB_A_CONV3X3>(d, s);
}
