// Text-encoder side small kernels (gfx950): token/position embedding gather + sparse gradient, TextBoost pins,
// rank-r LoRA down-projection / weight packing / parameter gradients.  All are latency-class (M = B*77 rows).
#include "common.h"
#include "../../include/textboost_hip.h"

namespace {

// ---- CLIPTextEmbeddings (transformers): h[m] = tok[ids[m]] + pos[m % T]       (text_encoder(...) :1054, :1099)
template <typename TT, typename TO>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* __restrict__ ids, const TT* __restrict__ tok,
                                                        const TT* __restrict__ pos, TO* __restrict__ out, int64_t M, int T, int D) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * D) return;
  const int64_t m = idx / D;
  const int c = (int)(idx - m * D);
  // a fp16 module adds in fp16 (teacher); the fp32 module adds in fp32
  const TT a = tok[ids[m] * D + c], b = pos[(m % T) * D + c];
  out[idx] = (TO)(TT)((float)a + (float)b);
}

// ---- sparse embedding gradient: only rows >= first_added receive gradient (:1109-1117 zeroes the rest).
// block a owns token id first_added + a; deterministic accumulation over the M positions.
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ dh, const int64_t* __restrict__ ids,
                                                        float* __restrict__ g, int64_t M, int D, int64_t first_added) {
  // positions holding this block's token are collected in ascending order (-> deterministic sum) by wave 0 with ballots
  __shared__ int pos[4096];
  __shared__ int npos;
  const int64_t tok = first_added + blockIdx.x;
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    int count = 0;
    for (int64_t base = 0; base < M; base += 64) {
      const int64_t m = base + lane;
      const bool hit = m < M && ids[m] == tok;
      const unsigned long long mask = __ballot(hit);
      if (hit) pos[count + __popcll(mask & ((1ull << lane) - 1ull))] = (int)m;
      count += __popcll(mask);
    }
    if (lane == 0) npos = count;
  }
  __syncthreads();
  const int n = npos;
  if (n == 0) return;
  for (int c = threadIdx.x; c < D; c += 256) {
    float a = g[(int64_t)blockIdx.x * D + c];
    for (int i = 0; i < n; ++i) a += dh[(int64_t)pos[i] * D + c];
    g[(int64_t)blockIdx.x * D + c] = a;
  }
}

// ---- TextBoostModel.forward pins (textboost/text_encoder.py:71-86)
template <typename T>
__global__ __launch_bounds__(256) void pin_fwd_kernel(T* __restrict__ h, const int64_t* __restrict__ ids, const float* __restrict__ null,
                                                      int B, int Tn, int D, int use_fixed, int64_t eos_id) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)B * Tn * D) return;
  const int64_t m = idx / D;
  const int c = (int)(idx - m * D);
  const int b = (int)(m / Tn), t = (int)(m - (int64_t)b * Tn);
  const bool is_null = ids[(int64_t)b * Tn + 1] == eos_id;
  if (is_null) h[idx] = (T)null[t * D + c];
  else if (use_fixed && t == 0) h[idx] = (T)null[c];
}
__global__ __launch_bounds__(256) void pin_bwd_kernel(float* __restrict__ dh, const int64_t* __restrict__ ids, int B, int Tn, int D,
                                                      int use_fixed, int64_t eos_id) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)B * Tn * D) return;
  const int64_t m = idx / D;
  const int b = (int)(m / Tn), t = (int)(m - (int64_t)b * Tn);
  const bool is_null = ids[(int64_t)b * Tn + 1] == eos_id;
  if (is_null || (use_fixed && t == 0)) dh[idx] = 0.f;
}

// ---- LoRA down projection t[m, j] = sum_k x[m,k] * fp16(A[j,k]),  j < R (R = 3r: q,k,v adapters stacked)
// one wave per row; the row of x is read ONCE and dotted against all R adapter rows (A is tiny and L1/L2 resident)
__global__ __launch_bounds__(256) void lora_down_kernel(const f16* __restrict__ x, int64_t ldx, const float* __restrict__ A,
                                                        f16* __restrict__ t, int64_t ldt, int64_t M, int K, int R) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float acc[24];
#pragma unroll
  for (int j = 0; j < 24; ++j) acc[j] = 0.f;
  for (int k = lane * 8; k < K; k += 512) {
    const f16x8 xv = *(const f16x8*)(x + row * ldx + k);
#pragma unroll
    for (int j = 0; j < 24; ++j)
      if (j < R) {
        const f32x4 a0 = *(const f32x4*)(A + (int64_t)j * K + k), a1 = *(const f32x4*)(A + (int64_t)j * K + k + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[j] += (float)xv[e] * (float)(f16)a0[e] + (float)xv[4 + e] * (float)(f16)a1[e];
      }
  }
#pragma unroll
  for (int j = 0; j < 24; ++j)
    if (j < R) {
      const float v = wave_sum(acc[j]);
      if (lane == 0) t[row * ldt + j] = (f16)v;
    }
}

// ---- W2[p*D + n, p*r + j] = scaling * B[(p*D + n)*r + j] (fp16, block diagonal, zero elsewhere; 64 columns)
__global__ __launch_bounds__(256) void lora_pack_b_kernel(const float* __restrict__ Bcat, f16* __restrict__ W2, int D, int r, int P,
                                                          float scaling) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)P * D * 64) return;
  const int64_t n = idx / 64;
  const int j = (int)(idx - n * 64);
  const int p = (int)(n / D);
  float v = 0.f;
  if (j >= p * r && j < (p + 1) * r) v = scaling * Bcat[n * r + (j - p * r)];
  W2[idx] = (f16)v;
}
// ---- W2d[k, j] = A[j*K + k] for j < R else 0 (fp16 [K, 64]) : second K-source of the qkv dgrad GEMM
__global__ __launch_bounds__(256) void lora_pack_at_kernel(const float* __restrict__ A, f16* __restrict__ W2d, int K, int R) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)K * 64) return;
  const int64_t k = idx / 64;
  const int j = (int)(idx - k * 64);
  W2d[idx] = (f16)(j < R ? A[(int64_t)j * K + k] : 0.f);
}

// ---- LoRA backward, three tiny reductions.  dY is the gradient of the fused qkv projection output [M, P*D].
// (a) dt[m, p*r + j] = scaling * sum_n dY[m, p*D + n] * fp16(B[(p*D+n)*r + j])      (wave per row)
__global__ __launch_bounds__(256) void lora_bwd_dt_kernel(const f16* __restrict__ dY, int64_t lddy, const float* __restrict__ Bcat,
                                                          f16* __restrict__ dt, int64_t lddt, int64_t M, int D, int r, int P,
                                                          float scaling) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  for (int p = 0; p < P; ++p)
    for (int j = 0; j < r; ++j) {
      float a = 0.f;
      for (int n = lane; n < D; n += 64) a += (float)dY[row * lddy + p * D + n] * (float)(f16)Bcat[((int64_t)p * D + n) * r + j];
      a = wave_sum(a) * scaling;
      if (lane == 0) dt[row * lddt + p * r + j] = (f16)a;
    }
}
// (b)/(c) are reductions over the M rows; M is split into slabs of LORA_RS rows across blockIdx.y so the chip is
// filled, per-slab partials go to a workspace and a second kernel sums them in a fixed order (deterministic).
constexpr int LORA_RS = 16;
// (b) partB[s][n*r + j] = sum_{m in slab s} dY[m, n] * t[m, p*r+j]                    (thread per output column n)
__global__ __launch_bounds__(256) void lora_bwd_db_kernel(const f16* __restrict__ dY, int64_t lddy, const f16* __restrict__ t,
                                                          int64_t ldt, float* __restrict__ part, int64_t M, int D, int r, int P) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over P*D
  if (n >= (int64_t)P * D) return;
  const int p = (int)(n / D);
  const int64_t m0 = (int64_t)blockIdx.y * LORA_RS, m1 = m0 + LORA_RS < M ? m0 + LORA_RS : M;
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t m = m0; m < m1; ++m) {
    const float d = (float)dY[m * lddy + n];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < r) a[j] += d * (float)t[m * ldt + p * r + j];
  }
  float* o = part + (int64_t)blockIdx.y * P * D * r + n * r;
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (j < r) o[j] = a[j];
}
// (c) partA[s][j*K + k] = sum_{m in slab s} dt[m, j] * x[m, k]                          (thread per k, j < R <= 24)
__global__ __launch_bounds__(256) void lora_bwd_da_kernel(const f16* __restrict__ dt, int64_t lddt, const f16* __restrict__ x,
                                                          int64_t ldx, float* __restrict__ part, int64_t M, int K, int R) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  const int64_t m0 = (int64_t)blockIdx.y * LORA_RS, m1 = m0 + LORA_RS < M ? m0 + LORA_RS : M;
  float a[24];
#pragma unroll
  for (int j = 0; j < 24; ++j) a[j] = 0.f;
  for (int64_t m = m0; m < m1; ++m) {
    const float xv = (float)x[m * ldx + k];
#pragma unroll
    for (int j = 0; j < 24; ++j)
      if (j < R) a[j] += (float)dt[m * lddt + j] * xv;
  }
  float* o = part + (int64_t)blockIdx.y * R * K;
#pragma unroll
  for (int j = 0; j < 24; ++j)
    if (j < R) o[(int64_t)j * K + k] = a[j];
}
// out[i] += scale * sum_s part[s*n + i]
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int64_t n, int nslab,
                                                          float scale) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float a = 0.f;
  for (int s = 0; s < nslab; ++s) a += part[(int64_t)s * n + i];
  out[i] += scale * a;
}

}  // namespace

#define GRID1D(n) dim3((unsigned)(((n) + 255) / 256))

extern "C" int tb_embed_fwd(const int64_t* ids, const void* tok, const void* pos, int table_dtype, void* out, int out_dtype, int64_t M,
                            int T, int D, tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!ids || !tok || !pos || !out || M <= 0) return TB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (table_dtype == TB_F32 && out_dtype == TB_F32)
    hipLaunchKernelGGL((embed_fwd_kernel<float, float>), GRID1D(M * D), dim3(256), 0, s, ids, (const float*)tok, (const float*)pos,
                       (float*)out, M, T, D);
  else if (table_dtype == TB_F16 && out_dtype == TB_F16)
    hipLaunchKernelGGL((embed_fwd_kernel<f16, f16>), GRID1D(M * D), dim3(256), 0, s, ids, (const f16*)tok, (const f16*)pos, (f16*)out,
                       M, T, D);
  else
    return TB_EINVAL;
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int tb_embed_bwd(const float* dh, const int64_t* ids, float* g_added, int64_t M, int D, int64_t first_added, int n_added,
                            tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!dh || !ids || !g_added || n_added <= 0 || M > 4096) return TB_EINVAL;
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(n_added), dim3(256), 0, (hipStream_t)stream, dh, ids, g_added, M, D, first_added);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int tb_textboost_pin_fwd(void* h, int h_dtype, const int64_t* ids, const float* null_embedding, int B, int T, int D,
                                    int use_fixed, int64_t eos_id, tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!h || !ids || !null_embedding) return TB_EINVAL;
  const int64_t n = (int64_t)B * T * D;
  if (h_dtype == TB_F32)
    hipLaunchKernelGGL(pin_fwd_kernel<float>, GRID1D(n), dim3(256), 0, (hipStream_t)stream, (float*)h, ids, null_embedding, B, T, D,
                       use_fixed, eos_id);
  else
    hipLaunchKernelGGL(pin_fwd_kernel<f16>, GRID1D(n), dim3(256), 0, (hipStream_t)stream, (f16*)h, ids, null_embedding, B, T, D,
                       use_fixed, eos_id);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int tb_textboost_pin_bwd(float* dh, const int64_t* ids, int B, int T, int D, int use_fixed, int64_t eos_id,
                                    tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!dh || !ids) return TB_EINVAL;
  hipLaunchKernelGGL(pin_bwd_kernel, GRID1D((int64_t)B * T * D), dim3(256), 0, (hipStream_t)stream, dh, ids, B, T, D, use_fixed, eos_id);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int tb_lora_down(const void* x, int64_t ldx, const float* A, void* t, int64_t ldt, int64_t M, int K, int R,
                            tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!x || !A || !t || R <= 0 || R > 24 || K % 8 || ldx % 8) return TB_EINVAL;
  hipLaunchKernelGGL(lora_down_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const f16*)x, ldx, A, (f16*)t,
                     ldt, M, K, R);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int tb_lora_pack(const float* A, const float* Bcat, void* w2_fwd, void* w2_dgrad, int D, int K, int r, int P, float scaling,
                            tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!A || !Bcat || !w2_fwd || !w2_dgrad || P * r > 64) return TB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(lora_pack_b_kernel, GRID1D((int64_t)P * D * 64), dim3(256), 0, s, Bcat, (f16*)w2_fwd, D, r, P, scaling);
  hipLaunchKernelGGL(lora_pack_at_kernel, GRID1D((int64_t)K * 64), dim3(256), 0, s, A, (f16*)w2_dgrad, K, P * r);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int64_t tb_lora_bwd_ws_floats(int64_t M, int D, int K, int r, int P) {
  const int64_t nslab = (M + LORA_RS - 1) / LORA_RS;
  return nslab * ((int64_t)P * D * r + (int64_t)P * r * K);
}

extern "C" int tb_lora_bwd(const void* dY, int64_t lddy, const void* x, int64_t ldx, const void* t, int64_t ldt, const float* Bcat,
                           void* dt, int64_t lddt, float* dA, float* dB, float* ws, int64_t M, int D, int K, int r, int P,
                           float scaling, tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!dY || !x || !t || !Bcat || !dt || !dA || !dB || !ws || r > 8 || P * r > 24) return TB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int nslab = (int)((M + LORA_RS - 1) / LORA_RS);
  float* partB = ws;
  float* partA = ws + (int64_t)nslab * P * D * r;
  hipLaunchKernelGGL(lora_bwd_dt_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, (const f16*)dY, lddy, Bcat, (f16*)dt, lddt, M,
                     D, r, P, scaling);
  hipLaunchKernelGGL(lora_bwd_db_kernel, dim3((unsigned)(((int64_t)P * D + 255) / 256), nslab), dim3(256), 0, s, (const f16*)dY, lddy,
                     (const f16*)t, ldt, partB, M, D, r, P);
  hipLaunchKernelGGL(lora_bwd_da_kernel, dim3((unsigned)((K + 255) / 256), nslab), dim3(256), 0, s, (const f16*)dt, lddt,
                     (const f16*)x, ldx, partA, M, K, P * r);
  hipLaunchKernelGGL(slab_reduce_kernel, GRID1D((int64_t)P * D * r), dim3(256), 0, s, partB, dB, (int64_t)P * D * r, nslab, scaling);
  hipLaunchKernelGGL(slab_reduce_kernel, GRID1D((int64_t)P * r * K), dim3(256), 0, s, partA, dA, (int64_t)P * r * K, nslab, 1.f);
  TB_CHECK_LAUNCH();
  return TB_OK;
}
