// fp32 (no-AMP) numeric mode of the TextBoost step on gfx950.
//
// The reference trains in full fp32 unless `--mixed_precision fp16` is given (train_textboost.py:298-308: default None; :930-939: weight_dtype
// stays float32, no GradScaler) -- that is what its README command runs (README.md:58-76).  This file is the device side of that mode: every
// activation, weight and gradient is fp32 in HBM and every contraction runs on the exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32
// (one rounding per product, fp32 accumulation: bitwise an fmaf chain), so results differ from an fp32 CPU run only by summation order.
// Correctness first: one generic GEMM kernel serves every Linear / 1x1 / 3x3 convolution (the same tb_gemm_desc, incl. the gathers and
// epilogues of the fp16 family) and, batched over (batch, head) with transposed operand views, the four products of attention; the
// remaining kernels are plain streaming kernels.  The matrix rate of fp32 MFMA is 1/16 of fp16 (157 TFLOP/s peak), the mode is expected
// to run an order of magnitude below the fp16 one.
#include "common.h"
#include "../../include/textboost_hip.h"

namespace {

constexpr int GB = 64;    // tile: 64 x 64 outputs per 256-thread workgroup (four waves, one 32x32 MFMA accumulator each)
constexpr int GK = 16;    // k-slice per LDS stage
constexpr int GLD = 20;   // LDS row pitch (floats): 80 B rows keep the 32-byte fragment reads spread over the banks

struct GemmExt {  // operand views beyond tb_gemm_desc: transposed operands and a batch of independent problems
  int a_trans, w_trans;      // A given as [K][M] (element (m, k) at k * lda + m); W given as [K][N]
  int batch, bdiv;           // grid.z = batch; problem z = (z / bdiv, z % bdiv) -> offset i0 * s0 + i1 * s1 per operand
  int64_t a_s0, a_s1, w_s0, w_s1, c_s0, c_s1, r_s0, r_s1;
};

__device__ __forceinline__ float act_fwd(int act, float v) {
  if (act == TB_ACT_QUICK_GELU) return quick_gelu_f(v);
  if (act == TB_ACT_GELU) return gelu_erf_f(v);
  if (act == TB_ACT_SILU) return silu_f(v);
  return v;
}

__global__ __launch_bounds__(256) void gemm_f32_kernel(const tb_gemm_desc p, const GemmExt x) {
  __shared__ __attribute__((aligned(16))) float As[GB * GLD];
  __shared__ __attribute__((aligned(16))) float Ws[GB * GLD];
  __shared__ __attribute__((aligned(16))) float Cs[GB * (GB + 1)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.y * GB, n0 = (int64_t)blockIdx.x * GB;
  const int z = blockIdx.z, z0 = z / x.bdiv, z1 = z - z0 * x.bdiv;
  const float* A = (const float*)p.A + z0 * x.a_s0 + z1 * x.a_s1;
  const float* W = (const float*)p.W + z0 * x.w_s0 + z1 * x.w_s1;
  float* C = (float*)p.C + z0 * x.c_s0 + z1 * x.c_s1;
  const float* R = p.R ? (const float*)p.R + z0 * x.r_s0 + z1 * x.r_s1 : nullptr;
  const float* A2 = (const float*)p.A2;
  const float* W2 = (const float*)p.W2;
  // staging role: thread -> (row of the tile, 4 consecutive k)
  const int srow = tid >> 2, sk = (tid & 3) * 4;
  const int64_t am = m0 + srow, wnrow = n0 + srow;
  const bool a_in = am < p.M, w_in = wnrow < p.N;
  // conv: decode the output pixel of this thread's A row once
  int cb = 0, cy = 0, cx = 0;
  if (p.a_mode == TB_A_CONV3X3 && a_in) {
    const int hw = p.Hout * p.Wout;
    cb = (int)(am / hw);
    const int rem = (int)(am - (int64_t)cb * hw);
    cy = rem / p.Wout;
    cx = rem - cy * p.Wout;
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  auto load_a = [&](int64_t k0, float* v) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = 0.f;
    const int64_t k = k0 + sk;
    if (!a_in || k >= p.K) return;
    if (p.a_mode == TB_A_CONV3X3) {  // k = tap * Cin + ci, Cin % 4 == 0: the four k share one tap
      const int tap = (int)(k / p.Cin), ci = (int)(k - (int64_t)tap * p.Cin);
      const int ky = tap / 3, kx = tap - ky * 3;
      int sy, sx;
      bool ok;
      if (p.upsample) {
        const int uy = cy + ky - 1, ux = cx + kx - 1;
        ok = uy >= 0 && ux >= 0 && uy < 2 * p.Hin && ux < 2 * p.Win;
        sy = uy >> 1, sx = ux >> 1;
      } else if (p.transposed) {
        const int ty = cy + 1 - ky, tx = cx + 1 - kx;
        ok = ty >= 0 && tx >= 0 && !(ty & 1) && !(tx & 1);
        sy = ty >> 1, sx = tx >> 1;
        ok = ok && sy < p.Hin && sx < p.Win;
      } else {
        sy = cy * p.stride + p.sign * (ky - 1) + p.shift;
        sx = cx * p.stride + p.sign * (kx - 1) + p.shift;
        ok = sy >= 0 && sx >= 0 && sy < p.Hin && sx < p.Win;
      }
      if (!ok) return;
      const float* src = A + (((int64_t)cb * p.Hin + sy) * p.Win + sx) * p.lda + ci;
      const f32x4 t = *(const f32x4*)src;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = t[e];
      return;
    }
    if (x.a_trans) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (k + e < p.K) v[e] = A[(k + e) * p.lda + am];
      return;
    }
    const float* src = k < p.K1 ? A + am * p.lda + k : A2 + am * p.lda2 + (k - p.K1);  // K1 % 4 == 0: no straddling
    const int64_t lim = k < p.K1 ? p.K1 : p.K;
    if (k + 3 < lim && (((uintptr_t)src) & 15) == 0) {
      const f32x4 t = *(const f32x4*)src;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = t[e];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (k + e < lim) v[e] = src[e];
    }
  };
  auto load_w = [&](int64_t k0, float* v) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = 0.f;
    const int64_t k = k0 + sk;
    if (!w_in || k >= p.K) return;
    if (x.w_trans) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (k + e < p.K) v[e] = W[(k + e) * p.ldw + wnrow];
      return;
    }
    const float* src = k < p.K1 ? W + wnrow * p.ldw + k : W2 + wnrow * p.ldw2 + (k - p.K1);
    const int64_t lim = k < p.K1 ? p.K1 : p.K;
    if (k + 3 < lim && (((uintptr_t)src) & 15) == 0) {
      const f32x4 t = *(const f32x4*)src;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = t[e];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (k + e < lim) v[e] = src[e];
    }
  };
  float ra[4], rw[4];
  load_a(0, ra);
  load_w(0, rw);
  for (int64_t k0 = 0; k0 < p.K; k0 += GK) {
    __syncthreads();  // the previous slice's fragment reads are done
    *(f32x4*)(As + srow * GLD + sk) = f32x4{ra[0], ra[1], ra[2], ra[3]};
    *(f32x4*)(Ws + srow * GLD + sk) = f32x4{rw[0], rw[1], rw[2], rw[3]};
    __syncthreads();
    if (k0 + GK < p.K) {  // next slice in flight under the multiplication
      load_a(k0 + GK, ra);
      load_w(k0 + GK, rw);
    }
    // MFMA j of the slice multiplies k = 8 hi + j (any order both operands agree on): a lane's 8 values are contiguous in its row
    const float* ap = As + (wm * 32 + l31) * GLD + hi * 8;
    const float* wp = Ws + (wn * 32 + l31) * GLD + hi * 8;
    const f32x4 a0 = *(const f32x4*)ap, a1 = *(const f32x4*)(ap + 4);
    const f32x4 w0 = *(const f32x4*)wp, w1 = *(const f32x4*)(wp + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], w0[j], acc, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], w1[j], acc, 0, 0, 0);
  }
  // ---- epilogue through an LDS image of the tile: every global access below is row-contiguous
#pragma unroll
  for (int r = 0; r < 16; ++r) Cs[(wm * 32 + mfma32_row(r, hi)) * (GB + 1) + wn * 32 + l31] = acc[r];
  __syncthreads();
  const int act = p.act;
  if (act == TB_ACT_GEGLU) {
    // W rows are interleaved in 32-row blocks [h | g] (textboost_amd/unet.py pack_geglu_rows): tile columns 0..31 = h, 32..63 = g of gate
    // columns n0 / 2 .. + 31;  C[m, n0/2 + j] = h * gelu(g) and C2[m, n0 + ..] keeps both pre-gate projections for the backward
    float* C2 = (float*)p.C2;
    for (int i = tid; i < GB * 32; i += 256) {
      const int r = i >> 5, j = i & 31;
      const int64_t m = m0 + r;
      if (m >= p.M) continue;
      const float h = Cs[r * (GB + 1) + j] * p.alpha + (p.bias ? p.bias[n0 + j] : 0.f);
      const float g = Cs[r * (GB + 1) + 32 + j] * p.alpha + (p.bias ? p.bias[n0 + 32 + j] : 0.f);
      if (C2) {
        C2[m * p.ldc2 + n0 + j] = h;
        C2[m * p.ldc2 + n0 + 32 + j] = g;
      }
      C[m * p.ldc + (n0 >> 1) + j] = h * gelu_erf_f(g);
    }
    return;
  }
  if (act == TB_ACT_GEGLU_GRAD) {
    // v = d(gated)[m, n]; C2 = packed pre-gate [M, 2N]; C = d(proj) [M, 2N] in the same packing: dh = v gelu(g), dg = v h gelu'(g)
    const float* C2 = (const float*)p.C2;
    for (int i = tid; i < GB * GB; i += 256) {
      const int r = i >> 6, j = i & 63;
      const int64_t m = m0 + r, n = n0 + j;
      if (m >= p.M || n >= p.N) continue;
      const float v = Cs[r * (GB + 1) + j] * p.alpha;
      const int64_t pc = (n >> 5) * 64 + (n & 31);
      const float h = C2[m * p.ldc2 + pc], g = C2[m * p.ldc2 + pc + 32];
      float ge, dge;
      gelu_erf_both_f(g, ge, dge);
      C[m * p.ldc + pc] = v * ge;
      C[m * p.ldc + pc + 32] = v * h * dge;
    }
    return;
  }
  for (int i = tid; i < GB * GB; i += 256) {
    const int r = i >> 6, j = i & 63;
    const int64_t m = m0 + r, n = n0 + j;
    if (m >= p.M || n >= p.N) continue;
    float v = Cs[r * (GB + 1) + j] * p.alpha;
    if (p.bias) v += p.bias[n];
    if (p.rowbias) v += p.rowbias[(m / p.rows_per_group) * p.ldrb + n];
    if (R) v += R[m * p.ldr + n];
    if (act == TB_ACT_QUICK_GELU || act == TB_ACT_GELU) {
      if (p.C2) ((float*)p.C2)[m * p.ldc2 + n] = v;  // pre-activation for the backward
      v = act_fwd(act, v);
    } else if (act == TB_ACT_SILU) {
      v = silu_f(v);
    } else if (act == TB_ACT_QUICK_GELU_GRAD) {
      v *= quick_gelu_grad_f(((const float*)p.C2)[m * p.ldc2 + n]);
    } else if (act == TB_ACT_GELU_GRAD) {
      v *= gelu_erf_grad_f(((const float*)p.C2)[m * p.ldc2 + n]);
    }
    C[m * p.ldc + n] = v;
  }
}

int check_gemm_f32(const tb_gemm_desc& d, const GemmExt& x) {
  if (!d.A || !d.W || !d.C || d.M <= 0 || d.N <= 0 || d.K <= 0) return TB_EINVAL;
  if (d.c_dtype != TB_F32 || (d.R && d.r_dtype != TB_F32)) return TB_EINVAL;
  if ((d.A2 == nullptr) != (d.W2 == nullptr)) return TB_EINVAL;
  if (d.A2 && (d.K1 <= 0 || d.K1 >= d.K || d.K1 % 4 || x.a_trans || x.w_trans || d.a_mode != TB_A_LINEAR)) return TB_EINVAL;
  if (d.rowbias && d.rows_per_group <= 0) return TB_EINVAL;
  if ((d.act == TB_ACT_QUICK_GELU_GRAD || d.act == TB_ACT_GELU_GRAD || d.act == TB_ACT_GEGLU_GRAD) && !d.C2) return TB_EINVAL;
  if ((d.act == TB_ACT_GEGLU || d.act == TB_ACT_GEGLU_GRAD) && (d.R || d.rowbias || x.batch != 1)) return TB_EINVAL;
  if (d.act == TB_ACT_GEGLU && d.N % 64) return TB_EINVAL;
  if (d.act == TB_ACT_GEGLU_GRAD && d.N % 32) return TB_EINVAL;
  if (d.act < 0 || d.act > TB_ACT_GEGLU_GRAD) return TB_EINVAL;
  if (d.a_mode == TB_A_CONV3X3) {
    if (d.A2 || x.a_trans || d.Cin <= 0 || d.Cin % 4 || d.lda % 4 || d.K != 9 * (int64_t)d.Cin || ((uintptr_t)d.A) % 16) return TB_EINVAL;
    if (d.M != (int64_t)d.B * d.Hout * d.Wout) return TB_EINVAL;
    if (d.sign != 1 && d.sign != -1) return TB_EINVAL;
    if (d.stride < 1 || (d.shift && (d.upsample || d.transposed))) return TB_EINVAL;
  } else if (d.a_mode != TB_A_LINEAR) {
    return TB_EINVAL;
  }
  return TB_OK;
}
int launch_gemm_f32(tb_gemm_desc d, const GemmExt& x, hipStream_t s) {
  if (!d.A2) d.K1 = d.K;
  if (d.rowbias && d.ldrb < d.N) d.ldrb = d.N;
  const int rc = check_gemm_f32(d, x);
  if (rc) return rc;
  dim3 grid((unsigned)((d.N + GB - 1) / GB), (unsigned)((d.M + GB - 1) / GB), (unsigned)x.batch);
  hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, s, d, x);
  TB_CHECK_LAUNCH();
  return TB_OK;
}
GemmExt plain_ext() {
  GemmExt x = {};
  x.batch = 1;
  x.bdiv = 1;
  return x;
}

// ------------------------------------------------------------------------------------------------ attention pieces
// scores S fp32 [Z][Sq][Skv] (Z = B * H), already scaled.  One wave per row.
__global__ __launch_bounds__(256) void softmax_f32_kernel(float* __restrict__ S, float* __restrict__ lse, int64_t rows, int Sq, int Skv,
                                                          int causal) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int q = (int)(row % Sq);
  const int nvis = causal ? min(Skv, q + 1) : Skv;
  float* s = S + row * Skv;
  float mx = -INFINITY;
  for (int k = lane; k < nvis; k += 64) mx = fmaxf(mx, s[k]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int k = lane; k < nvis; k += 64) sum += expf(s[k] - mx);
  sum = wave_sum(sum);
  const float l = mx + logf(sum);
  for (int k = lane; k < Skv; k += 64) s[k] = k < nvis ? expf(s[k] - l) : 0.f;
  if (lane == 0 && lse) lse[row] = l;
}
// backward: P = exp(S - lse) (masked), in place
__global__ __launch_bounds__(256) void probs_f32_kernel(float* __restrict__ S, const float* __restrict__ lse, int64_t rows, int Sq, int Skv,
                                                        int causal) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int q = (int)(row % Sq);
  const int nvis = causal ? min(Skv, q + 1) : Skv;
  float* s = S + row * Skv;
  const float l = lse[row];
  for (int k = lane; k < Skv; k += 64) s[k] = k < nvis ? expf(s[k] - l) : 0.f;
}
// dS = P * (dP - delta[row]), in place of dP
__global__ __launch_bounds__(256) void ds_f32_kernel(float* __restrict__ dP, const float* __restrict__ P, const float* __restrict__ delta,
                                                     int64_t rows, int Skv) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float dl = delta[row];
  for (int k = lane; k < Skv; k += 64) dP[row * Skv + k] = P[row * Skv + k] * (dP[row * Skv + k] - dl);
}
// delta[b, h, q] = sum_d dO * O
__global__ __launch_bounds__(256) void delta_f32_kernel(const tb_attn_desc p) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)p.B * p.H * p.Sq;
  if (idx >= total) return;
  const int q = (int)(idx % p.Sq);
  const int h = (int)((idx / p.Sq) % p.H), b = (int)(idx / ((int64_t)p.Sq * p.H));
  const float* o = (const float*)p.O + ((int64_t)b * p.Sq + q) * p.ldo + h * p.hd;
  const float* d = (const float*)p.dO + ((int64_t)b * p.Sq + q) * p.lddo + h * p.hd;
  float a = 0.f;
  for (int c = 0; c < p.hd; ++c) a += o[c] * d[c];
  p.Delta[idx] = a;
}

// batched product over z = (b, h): A rows from a [B*Sa, H*hd]-style buffer or from the score buffer
int attn_gemm(hipStream_t s, const float* A, int64_t lda, int a_trans, int64_t a_s0, int64_t a_s1, const float* W, int64_t ldw, int w_trans,
              int64_t w_s0, int64_t w_s1, float* C, int64_t ldc, int64_t c_s0, int64_t c_s1, int64_t M, int64_t N, int64_t K, float alpha,
              int Bn, int Hn) {
  tb_gemm_desc d = {};
  d.M = M, d.N = N, d.K = K, d.K1 = K;
  d.A = A, d.lda = lda, d.W = W, d.ldw = ldw, d.C = C, d.ldc = ldc, d.c_dtype = TB_F32;
  d.a_mode = TB_A_LINEAR, d.alpha = alpha, d.act = TB_ACT_NONE;
  GemmExt x = {};
  x.a_trans = a_trans, x.w_trans = w_trans, x.batch = Bn * Hn, x.bdiv = Hn;
  x.a_s0 = a_s0, x.a_s1 = a_s1, x.w_s0 = w_s0, x.w_s1 = w_s1, x.c_s0 = c_s0, x.c_s1 = c_s1;
  return launch_gemm_f32(d, x, s);
}

// ------------------------------------------------------------------------------------------------ GroupNorm, fp32 NHWC
// one workgroup per (image, group); two passes over the slice for the statistics (mean, then centred variance), one to apply
__global__ __launch_bounds__(256) void gn_f32_fwd_kernel(const float* __restrict__ X, int64_t ldx, float* __restrict__ Y, int64_t ldy,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ stats, int HW, int C, int G, float eps, int silu) {
  __shared__ float red[4];
  const int b = blockIdx.x / G, g = blockIdx.x - b * G, cg = C / G;
  const float* x = X + (int64_t)b * HW * ldx + g * cg;
  float* y = Y + (int64_t)b * HW * ldy + g * cg;
  const int n = HW * cg;
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += x[(int64_t)(i / cg) * ldx + (i % cg)];
  const float mean = block_sum_256(s, red) / (float)n;
  float q = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float d = x[(int64_t)(i / cg) * ldx + (i % cg)] - mean;
    q += d * d;
  }
  const float rstd = rsqrtf(block_sum_256(q, red) / (float)n + eps);
  if (threadIdx.x == 0) {
    stats[blockIdx.x * 2] = mean;
    stats[blockIdx.x * 2 + 1] = rstd;
  }
  for (int i = threadIdx.x; i < n; i += 256) {
    const int c = i % cg;
    const int64_t r = i / cg;
    float v = (x[r * ldx + c] - mean) * rstd * gamma[g * cg + c] + beta[g * cg + c];
    if (silu) v = silu_f(v);
    y[r * ldy + c] = v;
  }
}
// dx = rstd * (gy - mean(gy) - xhat * mean(gy * xhat)) [+ add], gy = dy * silu'(z) * gamma, z = xhat * gamma + beta
__global__ __launch_bounds__(256) void gn_f32_bwd_kernel(const float* __restrict__ dY, int64_t lddy, const float* __restrict__ X, int64_t ldx,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ stats, const float* __restrict__ add, int64_t ldadd,
                                                         float* __restrict__ dX, int64_t lddx, int HW, int C, int G, int silu) {
  __shared__ float red[4];
  const int b = blockIdx.x / G, g = blockIdx.x - b * G, cg = C / G;
  const int64_t base = (int64_t)b * HW;
  const int n = HW * cg;
  const float mean = stats[blockIdx.x * 2], rstd = stats[blockIdx.x * 2 + 1];
  auto gy_of = [&](int64_t r, int c, float& xh) {
    xh = (X[(base + r) * ldx + g * cg + c] - mean) * rstd;
    float d = dY[(base + r) * lddy + g * cg + c];
    if (silu) d *= silu_grad_f(xh * gamma[g * cg + c] + beta[g * cg + c]);
    return d * gamma[g * cg + c];
  };
  float s1 = 0.f, s2 = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    float xh;
    const float gy = gy_of(i / cg, i % cg, xh);
    s1 += gy;
    s2 += gy * xh;
  }
  const float m1 = block_sum_256(s1, red) / (float)n;
  const float m2 = block_sum_256(s2, red) / (float)n;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int c = i % cg;
    const int64_t r = i / cg;
    float xh;
    const float gy = gy_of(r, c, xh);
    float v = rstd * (gy - m1 - xh * m2);
    if (add) v += add[(base + r) * ldadd + g * cg + c];
    dX[(base + r) * lddx + g * cg + c] = v;
  }
}

// ------------------------------------------------------------------------------------------------ streaming kernels
__global__ void add_noise_f32_kernel(const float* __restrict__ x0, const float* __restrict__ noise, const int64_t* __restrict__ t,
                                     const float* __restrict__ acp, float* __restrict__ noisy, float* __restrict__ velocity,
                                     int64_t per_sample, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const float a = acp[t[i / per_sample]];
  const float sa = sqrtf(a), sb = sqrtf(1.f - a);
  noisy[i] = sa * x0[i] + sb * noise[i];
  if (velocity) velocity[i] = sa * noise[i] - sb * x0[i];
}
__global__ void timestep_embed_f32_kernel(const int64_t* __restrict__ t, float* __restrict__ out, int B, int dim) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * dim) return;
  const int b = i / dim, j = i - b * dim, half = dim / 2;
  const int k = j < half ? j : j - half;
  const float f = expf(-9.210340371976184f * (float)k / (float)half);
  const float arg = (float)t[b] * f;
  out[i] = j < half ? cosf(arg) : sinf(arg);
}
// 3x3 conv with CIN (<= 4) channels on the NCHW fp32 side -> NHWC fp32 [B*H*W, Cout]; one thread per (pixel, output channel)
__global__ __launch_bounds__(256) void conv4_to_nhwc_f32_kernel(const float* __restrict__ in, int CIN, const float* __restrict__ Wp,
                                                                const float* __restrict__ bias, float* __restrict__ out, int64_t ldo, int B,
                                                                int H, int W, int Cout, int sign, float in_scale) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)B * H * W * Cout;
  if (idx >= total) return;
  const int co = (int)(idx % Cout);
  const int64_t m = idx / Cout;
  const int x = (int)(m % W), y = (int)((m / W) % H), b = (int)(m / ((int64_t)W * H));
  float acc = bias ? bias[co] : 0.f;
  for (int ky = 0; ky < 3; ++ky) {
    const int sy = y + sign * (ky - 1);
    if (sy < 0 || sy >= H) continue;
    for (int kx = 0; kx < 3; ++kx) {
      const int sx = x + sign * (kx - 1);
      if (sx < 0 || sx >= W) continue;
      for (int ci = 0; ci < CIN; ++ci)
        acc += in[(((int64_t)b * CIN + ci) * H + sy) * W + sx] * in_scale * Wp[(int64_t)((ky * 3 + kx) * CIN + ci) * Cout + co];
    }
  }
  out[m * ldo + co] = acc;
}
// conv_out forward: NHWC fp32 [M, C] -> NCHW fp32 [B, 4, H, W]; one wave per output pixel.  Wp fp32 [4][9][C]
__global__ __launch_bounds__(256) void conv_to4_f32_kernel(const float* __restrict__ in, int64_t ldi, const float* __restrict__ Wp,
                                                           const float* __restrict__ bias, float* __restrict__ out, int B, int H, int W, int C) {
  const int lane = threadIdx.x & 63;
  const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t M = (int64_t)B * H * W;
  if (m >= M) return;
  const int b = (int)(m / (H * W));
  const int rem = (int)(m - (int64_t)b * H * W);
  const int y = rem / W, x = rem - y * W;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int tap = 0; tap < 9; ++tap) {
    const int ky = tap / 3, kx = tap - ky * 3;
    const int sy = y + ky - 1, sx = x + kx - 1;
    if (sy < 0 || sx < 0 || sy >= H || sx >= W) continue;
    const float* src = in + (((int64_t)b * H + sy) * W + sx) * ldi;
    for (int c = lane; c < C; c += 64) {
      const float v = src[c];
#pragma unroll
      for (int co = 0; co < 4; ++co) acc[co] += v * Wp[((int64_t)co * 9 + tap) * C + c];
    }
  }
#pragma unroll
  for (int co = 0; co < 4; ++co) {
    const float s = wave_sum(acc[co]);
    if (lane == 0) out[(((int64_t)b * 4 + co) * H + y) * W + x] = s + (bias ? bias[co] : 0.f);
  }
}
__global__ void upsample2x_f32_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ u, int64_t ldu, int B, int H, int W, int C) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)B * 2 * H * 2 * W * C;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  const int64_t m = idx / C;
  const int ux = (int)(m % (2 * W)), uy = (int)((m / (2 * W)) % (2 * H)), b = (int)(m / ((int64_t)4 * W * H));
  u[m * ldu + c] = x[(((int64_t)b * H + (uy >> 1)) * W + (ux >> 1)) * ldx + c];
}
__global__ void pool2x2_sum_f32_kernel(const float* __restrict__ du, int64_t ldu, float* __restrict__ dx, int64_t ldx, int B, int H, int W, int C) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)B * H * W * C;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  const int64_t m = idx / C;
  const int x = (int)(m % W), y = (int)((m / W) % H), b = (int)(m / ((int64_t)W * H));
  const int64_t r0 = ((int64_t)b * 2 * H + 2 * y) * 2 * W + 2 * x;
  dx[m * ldx + c] = du[r0 * ldu + c] + du[(r0 + 1) * ldu + c] + du[(r0 + 2 * W) * ldu + c] + du[(r0 + 2 * W + 1) * ldu + c];
}
__global__ void add_f32_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ b, int64_t ldb, float* __restrict__ o,
                               int64_t ldo, int64_t M, int C) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * C) return;
  const int64_t m = idx / C;
  const int c = (int)(idx - m * C);
  o[m * ldo + c] = a[m * lda + c] + b[m * ldb + c];
}
// mean((pred - target)^2) and its gradient (:1085-1090), fp32 prediction; deterministic two-stage reduction
__global__ __launch_bounds__(256) void mse_f32_partial_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                              float* __restrict__ dpred, const float* __restrict__ loss_scale, int64_t N,
                                                              float* __restrict__ partial) {
  __shared__ float red[4];
  const float sc = loss_scale[0] * 2.f / (float)N;
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
    const float d = pred[i] - target[i];
    s += d * d;
    dpred[i] = d * sc;
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ void mse_f32_final_kernel(const float* __restrict__ partial, int n, float inv, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += partial[i];
    out[0] = s * inv;
  }
}

// K-extension operand of the hoisted cross-attention K/V GEMM with rank-r adapters on every to_k / to_v (--unet_params_to_train crossattn_kv,
// train_textboost.py:712-721): W2[n, col_base[n] + j] = scaling * B[n, j], zero elsewhere (row n = output feature of the concatenated
// projections, col_base[n] = first column of that row's adapter in the concatenated down projections t = ehs A_all^T)
__global__ void kv_lora_pack_f32_kernel(const float* __restrict__ B, const int* __restrict__ col_base, float* __restrict__ W2, int64_t rows,
                                        int ncols, int r, float scaling) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * ncols) return;
  const int64_t n = idx / ncols;
  const int j = (int)(idx - n * ncols) - col_base[n];
  W2[idx] = (j >= 0 && j < r) ? scaling * B[n * r + j] : 0.f;
}

}  // namespace

// ================================================================================================ C ABI
extern "C" int tb_kv_lora_pack_f32(const float* B, const int32_t* col_base, float* w2, int64_t rows, int ncols, int r, float scaling,
                                   tb_stream_t stream) {
  (void)hipGetLastError();
  if (!B || !col_base || !w2 || rows <= 0 || ncols <= 0 || r <= 0) return TB_EINVAL;
  hipLaunchKernelGGL(kv_lora_pack_f32_kernel, dim3((unsigned)((rows * ncols + 255) / 256)), dim3(256), 0, (hipStream_t)stream, B, col_base, w2, rows,
                     ncols, r, scaling);
  TB_CHECK_LAUNCH();
  return TB_OK;
}
extern "C" int tb_gemm_f32(const tb_gemm_desc* dp, tb_stream_t stream) {
  (void)hipGetLastError();
  if (!dp) return TB_EINVAL;
  return launch_gemm_f32(*dp, plain_ext(), (hipStream_t)stream);
}
extern "C" int tb_gemm_f32_t(const tb_gemm_desc* dp, int a_trans, int w_trans, tb_stream_t stream) {
  (void)hipGetLastError();
  if (!dp) return TB_EINVAL;
  GemmExt x = plain_ext();
  x.a_trans = a_trans, x.w_trans = w_trans;
  return launch_gemm_f32(*dp, x, (hipStream_t)stream);
}

extern "C" int64_t tb_attention_f32_ws_floats(int B, int H, int Sq, int Skv) { return 2 * (int64_t)B * H * Sq * Skv; }

static int check_attn_f32(const tb_attn_desc& d, const float* ws, int64_t ws_floats, bool bwd) {
  if (d.B <= 0 || d.H <= 0 || d.Sq <= 0 || d.Skv <= 0 || d.hd <= 0 || !d.Q || !d.K || !d.V || !d.O || !ws) return TB_EINVAL;
  if (ws_floats < (bwd ? 2 : 1) * (int64_t)d.B * d.H * d.Sq * d.Skv) return TB_EINVAL;
  if (bwd && (!d.dO || !d.dQ || !d.dK || !d.dV || !d.LSE || !d.Delta)) return TB_EINVAL;
  return TB_OK;
}
// O = softmax(scale Q K^T [+ causal]) V with fp32 Q / K / V / O (column slices of [B*S, *] buffers like the fp16 entry point)
extern "C" int tb_attention_f32_fwd(const tb_attn_desc* dp, float* ws, int64_t ws_floats, tb_stream_t stream) {
  (void)hipGetLastError();
  if (!dp) return TB_EINVAL;
  const tb_attn_desc d = *dp;
  int rc = check_attn_f32(d, ws, ws_floats, false);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int64_t SS = (int64_t)d.Sq * d.Skv, rows = (int64_t)d.B * d.H * d.Sq;
  // S[z] = scale Q K^T
  rc = attn_gemm(s, (const float*)d.Q, d.ldq, 0, (int64_t)d.Sq * d.ldq, d.hd, (const float*)d.K, d.ldk, 0, (int64_t)d.Skv * d.ldk, d.hd, ws, d.Skv,
                 (int64_t)d.H * SS, SS, d.Sq, d.Skv, d.hd, d.scale, d.B, d.H);
  if (rc) return rc;
  hipLaunchKernelGGL(softmax_f32_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, ws, d.LSE, rows, d.Sq, d.Skv, d.causal);
  // O[z] = P V   (W = V^T: w_trans)
  rc = attn_gemm(s, ws, d.Skv, 0, (int64_t)d.H * SS, SS, (const float*)d.V, d.ldv, 1, (int64_t)d.Skv * d.ldv, d.hd, (float*)d.O, d.ldo,
                 (int64_t)d.Sq * d.ldo, d.hd, d.Sq, d.hd, d.Skv, 1.f, d.B, d.H);
  if (rc) return rc;
  TB_CHECK_LAUNCH();
  return TB_OK;
}
extern "C" int tb_attention_f32_bwd(const tb_attn_desc* dp, float* ws, int64_t ws_floats, tb_stream_t stream) {
  (void)hipGetLastError();
  if (!dp) return TB_EINVAL;
  const tb_attn_desc d = *dp;
  int rc = check_attn_f32(d, ws, ws_floats, true);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int64_t SS = (int64_t)d.Sq * d.Skv, rows = (int64_t)d.B * d.H * d.Sq, ZS = (int64_t)d.H * SS;
  float* P = ws;
  float* dP = ws + (int64_t)d.B * d.H * SS;
  const int64_t qb = (int64_t)d.Sq * d.ldq, kb = (int64_t)d.Skv * d.ldk, vb = (int64_t)d.Skv * d.ldv, ob = (int64_t)d.Sq * d.lddo;
  hipLaunchKernelGGL(delta_f32_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, d);
  // P = exp(scale Q K^T - lse)
  rc = attn_gemm(s, (const float*)d.Q, d.ldq, 0, qb, d.hd, (const float*)d.K, d.ldk, 0, kb, d.hd, P, d.Skv, ZS, SS, d.Sq, d.Skv, d.hd, d.scale,
                 d.B, d.H);
  if (rc) return rc;
  hipLaunchKernelGGL(probs_f32_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, P, d.LSE, rows, d.Sq, d.Skv, d.causal);
  // dV = P^T dO   (A = P^T: a_trans, W = dO^T: w_trans)
  rc = attn_gemm(s, P, d.Skv, 1, ZS, SS, (const float*)d.dO, d.lddo, 1, ob, d.hd, (float*)d.dV, d.lddv, (int64_t)d.Skv * d.lddv, d.hd, d.Skv,
                 d.hd, d.Sq, 1.f, d.B, d.H);
  if (rc) return rc;
  // dP = dO V^T
  rc = attn_gemm(s, (const float*)d.dO, d.lddo, 0, ob, d.hd, (const float*)d.V, d.ldv, 0, vb, d.hd, dP, d.Skv, ZS, SS, d.Sq, d.Skv, d.hd, 1.f,
                 d.B, d.H);
  if (rc) return rc;
  hipLaunchKernelGGL(ds_f32_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, dP, P, d.Delta, rows, d.Skv);
  // dQ = scale dS K   (W = K^T: w_trans)
  rc = attn_gemm(s, dP, d.Skv, 0, ZS, SS, (const float*)d.K, d.ldk, 1, kb, d.hd, (float*)d.dQ, d.lddq, (int64_t)d.Sq * d.lddq, d.hd, d.Sq, d.hd,
                 d.Skv, d.scale, d.B, d.H);
  if (rc) return rc;
  // dK = scale dS^T Q
  rc = attn_gemm(s, dP, d.Skv, 1, ZS, SS, (const float*)d.Q, d.ldq, 1, qb, d.hd, (float*)d.dK, d.lddk, (int64_t)d.Skv * d.lddk, d.hd, d.Skv,
                 d.hd, d.Sq, d.scale, d.B, d.H);
  if (rc) return rc;
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int tb_groupnorm_f32_fwd(const float* x, int64_t ldx, float* y, int64_t ldy, const float* gamma, const float* beta, float* stats,
                                    int B, int HW, int C, int G, float eps, int silu, tb_stream_t stream) {
  (void)hipGetLastError();
  if (!x || !y || !gamma || !beta || !stats || B <= 0 || HW <= 0 || C <= 0 || G <= 0 || C % G) return TB_EINVAL;
  hipLaunchKernelGGL(gn_f32_fwd_kernel, dim3(B * G), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, gamma, beta, stats, HW, C, G, eps, silu);
  TB_CHECK_LAUNCH();
  return TB_OK;
}
extern "C" int tb_groupnorm_f32_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma, const float* beta,
                                    const float* stats, const float* add, int64_t ldadd, float* dx, int64_t lddx, int B, int HW, int C, int G,
                                    int silu, tb_stream_t stream) {
  (void)hipGetLastError();
  if (!dy || !x || !gamma || !beta || !stats || !dx || B <= 0 || HW <= 0 || C <= 0 || G <= 0 || C % G) return TB_EINVAL;
  hipLaunchKernelGGL(gn_f32_bwd_kernel, dim3(B * G), dim3(256), 0, (hipStream_t)stream, dy, lddy, x, ldx, gamma, beta, stats, add, ldadd, dx, lddx,
                     HW, C, G, silu);
  TB_CHECK_LAUNCH();
  return TB_OK;
}
extern "C" int tb_add_noise_f32(const float* x0, const float* noise, const int64_t* timesteps, const float* acp, float* noisy, float* velocity,
                                int B, int64_t per_sample, tb_stream_t stream) {
  (void)hipGetLastError();
  if (!x0 || !noise || !timesteps || !acp || !noisy || B <= 0 || per_sample <= 0) return TB_EINVAL;
  const int64_t total = (int64_t)B * per_sample;
  hipLaunchKernelGGL(add_noise_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x0, noise, timesteps, acp,
                     noisy, velocity, per_sample, total);
  TB_CHECK_LAUNCH();
  return TB_OK;
}
extern "C" int tb_timestep_embed_f32(const int64_t* timesteps, float* out, int B, int dim, tb_stream_t stream) {
  (void)hipGetLastError();
  if (!timesteps || !out || B <= 0 || dim <= 0 || dim % 2) return TB_EINVAL;
  hipLaunchKernelGGL(timestep_embed_f32_kernel, dim3((B * dim + 255) / 256), dim3(256), 0, (hipStream_t)stream, timesteps, out, B, dim);
  TB_CHECK_LAUNCH();
  return TB_OK;
}
extern "C" int tb_conv4_to_nhwc_f32(const float* in, int Cin, const float* w_packed, const float* bias, float* out, int64_t ldo, int B, int H,
                                    int W, int Cout, int sign, float in_scale, tb_stream_t stream) {
  (void)hipGetLastError();
  if (!in || !w_packed || !out || Cin <= 0 || Cin > 4 || B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || (sign != 1 && sign != -1)) return TB_EINVAL;
  const int64_t total = (int64_t)B * H * W * Cout;
  hipLaunchKernelGGL(conv4_to_nhwc_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, Cin, w_packed, bias,
                     out, ldo, B, H, W, Cout, sign, in_scale);
  TB_CHECK_LAUNCH();
  return TB_OK;
}
extern "C" int tb_conv_to4_f32(const float* in, int64_t ldi, const float* w_packed, const float* bias, float* out, int B, int H, int W, int C,
                               tb_stream_t stream) {
  (void)hipGetLastError();
  if (!in || !w_packed || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0) return TB_EINVAL;
  const int64_t M = (int64_t)B * H * W;
  hipLaunchKernelGGL(conv_to4_f32_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, in, ldi, w_packed, bias, out, B, H, W, C);
  TB_CHECK_LAUNCH();
  return TB_OK;
}
extern "C" int tb_upsample2x_f32(const float* x, int64_t ldx, float* u, int64_t ldu, int B, int H, int W, int C, tb_stream_t stream) {
  (void)hipGetLastError();
  if (!x || !u || B <= 0 || H <= 0 || W <= 0 || C <= 0) return TB_EINVAL;
  const int64_t total = (int64_t)B * 4 * H * W * C;
  hipLaunchKernelGGL(upsample2x_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, u, ldu, B, H, W, C);
  TB_CHECK_LAUNCH();
  return TB_OK;
}
extern "C" int tb_pool2x2_sum_f32(const float* du, int64_t ldu, float* dx, int64_t ldx, int B, int H, int W, int C, tb_stream_t stream) {
  (void)hipGetLastError();
  if (!du || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0) return TB_EINVAL;
  const int64_t total = (int64_t)B * H * W * C;
  hipLaunchKernelGGL(pool2x2_sum_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, du, ldu, dx, ldx, B, H, W, C);
  TB_CHECK_LAUNCH();
  return TB_OK;
}
extern "C" int tb_add_f32(const float* a, int64_t lda, const float* b, int64_t ldb, float* out, int64_t ldo, int64_t M, int C, tb_stream_t stream) {
  (void)hipGetLastError();
  if (!a || !b || !out || M <= 0 || C <= 0) return TB_EINVAL;
  hipLaunchKernelGGL(add_f32_kernel, dim3((unsigned)((M * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, out, ldo, M, C);
  TB_CHECK_LAUNCH();
  return TB_OK;
}
extern "C" int tb_mse_loss_f32(const float* pred, const float* target, float* dpred, float* loss_out, const float* loss_scale, int64_t N,
                               float* ws /* >= 256 floats */, tb_stream_t stream) {
  (void)hipGetLastError();
  if (!pred || !target || !dpred || !loss_out || !loss_scale || !ws || N <= 0) return TB_EINVAL;
  int blocks = (int)((N + 255) / 256);
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(mse_f32_partial_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pred, target, dpred, loss_scale, N, ws);
  hipLaunchKernelGGL(mse_f32_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ws, blocks, 1.f / (float)N, loss_out);
  TB_CHECK_LAUNCH();
  return TB_OK;
}
