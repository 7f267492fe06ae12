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

// ---- one launch packs the fp16 K-extension operands of ALL layers (the stacks are contiguous over layers):
//   W2 [l][p*D + n, p*r + j] = scaling * B[l][(p*D + n)*r + j]   (block diagonal, zero elsewhere; 64 columns)
//   W2d[l][k, j]             = A[l][j*K + k] for j < R else 0    (fp16 [K, 64]: second K-source of the qkv dgrad GEMM)
template <typename TW>  // f16: K-extension operands of the fp16 GEMMs; float: the fp32 (no-AMP) mode
__global__ __launch_bounds__(256) void lora_pack_kernel(const float* __restrict__ A, const float* __restrict__ Bcat, TW* __restrict__ W2,
                                                        TW* __restrict__ W2d, int D, int K, int r, int P, int layers, float scaling) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t nb = (int64_t)layers * P * D * 64, na = (int64_t)layers * K * 64;
  if (idx < nb) {
    const int64_t n = idx / 64;  // row over layers * P * D
    const int j = (int)(idx - n * 64);
    const int p = (int)((n / D) % P);
    float v = 0.f;
    if (j >= p * r && j < (p + 1) * r) v = scaling * Bcat[n * r + (j - p * r)];
    W2[idx] = (TW)v;
  } else if (idx < nb + na) {
    const int64_t i = idx - nb;
    const int64_t lk = i / 64;  // layer * K + k
    const int j = (int)(i - lk * 64);
    const int64_t l = lk / K, k = lk - l * K;
    const int R = P * r;
    W2d[i] = (TW)(j < R ? A[(l * R + j) * K + k] : 0.f);
  }
}

// ---- LoRA backward.  dY is the gradient of the fused qkv projection output [M, P*D]; per adapter p (q, k, v):
//   (a) dt[m, p*r + j]  = scaling * sum_n dY[m, p*D + n] * fp16(B[(p*D+n)*r + j])
//   (b) dB[(p*D+n)*r+j] += scaling * sum_m dY[m, p*D + n] * t[m, p*r + j]
//   (c) dA[j*K + k]     += sum_m dt[m, j] * x[m, k]
// One kernel per slab of LORA_RS rows does all three: (a) lands in LDS (and in dt for the dgrad GEMM), (b)/(c) are per-slab
// partial sums written to a workspace; blockIdx.y = 0 takes the (b) units, 1 the (c) units so that M / LORA_RS * 2 blocks fill the
// chip.  A second kernel adds the partials in slab order (deterministic).
#ifndef TB_LORA_RS
#define TB_LORA_RS 8
#endif
constexpr int LORA_RS = TB_LORA_RS;
constexpr int LORA_MAXR = 24;  // P * r
template <int RR>  // RR = r when r is 4 or 8 (16-byte vector loads of the B rows, fully unrolled), 0 = generic r <= 8
__global__ __launch_bounds__(256) void lora_bwd_fused_kernel(const f16* __restrict__ dY, int64_t lddy, const f16* __restrict__ x, int64_t ldx,
                                                             const f16* __restrict__ t, int64_t ldt, const float* __restrict__ Bcat,
                                                             f16* __restrict__ dt, int64_t lddt, float* __restrict__ partB,
                                                             float* __restrict__ partA, int64_t M, int D, int K, int r, int P,
                                                             float scaling) {
  __shared__ float ts[LORA_RS][LORA_MAXR];
  __shared__ float dts[LORA_RS][LORA_MAXR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int R = P * r;
  const int64_t m0 = (int64_t)blockIdx.x * LORA_RS;
  const int rows = (int)(M - m0 < LORA_RS ? M - m0 : LORA_RS);
  for (int u = tid; u < LORA_RS * LORA_MAXR; u += 256) {
    const int i = u / LORA_MAXR, j = u - i * LORA_MAXR;
    ts[i][j] = (i < rows && j < R) ? (float)t[(m0 + i) * ldt + j] : 0.f;
    dts[i][j] = 0.f;  // columns >= R stay zero for (c)'s 4-row units
  }
  __syncthreads();
  // ---- (a): each wave owns LORA_RS / 4 rows; a chunk of B (8 columns x r) is loaded once into registers and used for all of them
  constexpr int RPW = LORA_RS / 4;
  for (int p = 0; p < P; ++p) {
    float acc[RPW][8];
#pragma unroll
    for (int ii = 0; ii < RPW; ++ii)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[ii][j] = 0.f;
    for (int n = lane * 8; n < D; n += 512) {
      const float* bp = Bcat + ((int64_t)p * D + n) * r;
      float bw[8][8];  // [e][j], fp16-rounded as the forward used them
      if (RR) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
          for (int q = 0; q < RR / 4; ++q) {
            const f32x4 b = *(const f32x4*)(bp + RR * e + 4 * q);
#pragma unroll
            for (int j = 0; j < 4; ++j) bw[e][4 * q + j] = (float)(f16)b[j];
          }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
          for (int j = 0; j < 8; ++j) bw[e][j] = j < r ? (float)(f16)bp[e * r + j] : 0.f;
      }
#pragma unroll
      for (int ii = 0; ii < RPW; ++ii) {
        const int i = wave + 4 * ii;
        if (i < rows) {
          const f16x8 dy = *(const f16x8*)(dY + (m0 + i) * lddy + (int64_t)p * D + n);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = (float)dy[e];
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (j < (RR ? RR : 8)) acc[ii][j] += d * bw[e][j];
          }
        }
      }
    }
#pragma unroll
    for (int ii = 0; ii < RPW; ++ii) {
      const int i = wave + 4 * ii;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < r) {
          const float v = wave_sum(acc[ii][j]) * scaling;
          if (lane == 0) {
            const f16 h = (f16)v;
            dts[i][p * r + j] = (float)h;  // (c) consumes the fp16 value the dgrad GEMM sees
            if (i < rows && blockIdx.y == 0) dt[(m0 + i) * lddt + p * r + j] = h;
          }
        }
    }
  }
  __syncthreads();
  if (blockIdx.y == 0) {
    // ---- (b): unit = 8 consecutive columns n of dY; acc[e][j] over the slab rows
    const int units = P * D / 8;
    float* o = partB + (int64_t)blockIdx.x * P * D * r;
    for (int u = tid; u < units; u += 256) {
      const int n = u * 8, p = n / D;
      float acc[8][8];
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[e][j] = 0.f;
      for (int i = 0; i < rows; ++i) {
        const f16x8 dy = *(const f16x8*)(dY + (m0 + i) * lddy + n);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (j < (RR ? RR : r)) {
            const float tv = ts[i][p * r + j];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e][j] += (float)dy[e] * tv;
          }
      }
      if (RR) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
          for (int q = 0; q < RR / 4; ++q) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = acc[e][4 * q + j];
            *(f32x4*)(o + (int64_t)(n + e) * RR + 4 * q) = v;
          }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (j < r) o[(int64_t)(n + e) * r + j] = acc[e][j];
      }
    }
  } else {
    // ---- (c): unit = (4 adapter rows j, 8 consecutive k)
    const int kg = K / 8, jgs = (R + 3) / 4;
    float* o = partA + (int64_t)blockIdx.x * R * K;
    for (int u = tid; u < kg * jgs; u += 256) {
      const int jg = u / kg, k = (u - jg * kg) * 8;
      float acc[4][8];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[j][e] = 0.f;
      for (int i = 0; i < rows; ++i) {
        const f16x8 xv = *(const f16x8*)(x + (m0 + i) * ldx + k);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float d = dts[i][jg * 4 + j];  // columns >= R hold zeros (LORA_MAXR is a multiple of 4)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[j][e] += d * (float)xv[e];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (jg * 4 + j < R) {
          f32x4 v0, v1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v0[e] = acc[j][e];
            v1[e] = acc[j][4 + e];
          }
          float* dst = o + (int64_t)(jg * 4 + j) * K + k;
          *(f32x4*)dst = v0;
          *(f32x4*)(dst + 4) = v1;
        }
    }
  }
}
// dB[i] += scaleB * sum_s partB[s][i] (i < nB);  dA[i] += sum_s partA[s][i] (i < nA): one launch for both, 4 floats per thread
__global__ __launch_bounds__(256) void lora_slab_reduce_kernel(const float* __restrict__ partB, const float* __restrict__ partA,
                                                               float* __restrict__ dB, float* __restrict__ dA, int64_t nB, int64_t nA,
                                                               int nslab, float scaleB) {
  int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const float* part;
  float* out;
  int64_t n;
  float scale;
  if (i < nB) part = partB, out = dB, n = nB, scale = scaleB;
  else if (i - nB < nA) i -= nB, part = partA, out = dA, n = nA, scale = 1.f;
  else return;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  int s = 0;
  for (; s + 4 <= nslab; s += 4) {  // 4 independent loads in flight, summed in slab order
    const f32x4 v0 = *(const f32x4*)(part + (int64_t)s * n + i), v1 = *(const f32x4*)(part + (int64_t)(s + 1) * n + i),
                v2 = *(const f32x4*)(part + (int64_t)(s + 2) * n + i), v3 = *(const f32x4*)(part + (int64_t)(s + 3) * n + i);
    a += v0;
    a += v1;
    a += v2;
    a += v3;
  }
  for (; s < nslab; ++s) a += *(const f32x4*)(part + (int64_t)s * n + i);
  f32x4 o = *(f32x4*)(out + i);
  o += a * scale;
  *(f32x4*)(out + i) = o;
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

extern "C" int tb_lora_pack(const float* A, const float* Bcat, void* w2_fwd, void* w2_dgrad, int D, int K, int r, int P, int layers,
                            float scaling, tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!A || !Bcat || !w2_fwd || !w2_dgrad || P * r > 64 || layers <= 0) return TB_EINVAL;
  const int64_t n = (int64_t)layers * ((int64_t)P * D + K) * 64;
  hipLaunchKernelGGL(lora_pack_kernel<f16>, GRID1D(n), dim3(256), 0, (hipStream_t)stream, A, Bcat, (f16*)w2_fwd, (f16*)w2_dgrad, D, K, r, P,
                     layers, scaling);
  TB_CHECK_LAUNCH();
  return TB_OK;
}
// the same operands in fp32 for the no-AMP mode (train_textboost.py:298-308): w2_fwd fp32 [layers][P*D, 64], w2_dgrad fp32 [layers][K, 64]
extern "C" int tb_lora_pack_f32(const float* A, const float* Bcat, float* w2_fwd, float* w2_dgrad, int D, int K, int r, int P, int layers,
                                float scaling, tb_stream_t stream) {
  (void)hipGetLastError();
  if (!A || !Bcat || !w2_fwd || !w2_dgrad || P * r > 64 || layers <= 0) return TB_EINVAL;
  const int64_t n = (int64_t)layers * ((int64_t)P * D + K) * 64;
  hipLaunchKernelGGL(lora_pack_kernel<float>, GRID1D(n), dim3(256), 0, (hipStream_t)stream, A, Bcat, w2_fwd, w2_dgrad, D, K, r, P, layers,
                     scaling);
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
  if (!dY || !x || !t || !Bcat || !dt || !dA || !dB || !ws || r > 8 || P * r > LORA_MAXR) return TB_EINVAL;
  if (D % 8 || K % 8 || lddy % 8 || ldx % 8 || (P * D * r) % 4 || (P * r * K) % 4) return TB_EINVAL;  // 16-byte vector accesses
  if (((uintptr_t)dY) % 16 || ((uintptr_t)x) % 16 || ((uintptr_t)dA) % 16 || ((uintptr_t)dB) % 16 || ((uintptr_t)ws) % 16 ||
      ((r == 4 || r == 8) && ((uintptr_t)Bcat) % 16))
    return TB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int nslab = (int)((M + LORA_RS - 1) / LORA_RS);
  const int64_t nB = (int64_t)P * D * r, nA = (int64_t)P * r * K;
  float* partB = ws;
  float* partA = ws + (int64_t)nslab * nB;
#define TB_LORA_BWD(RR)                                                                                                                   \
  hipLaunchKernelGGL(lora_bwd_fused_kernel<RR>, dim3(nslab, 2), dim3(256), 0, s, (const f16*)dY, lddy, (const f16*)x, ldx, (const f16*)t, \
                     ldt, Bcat, (f16*)dt, lddt, partB, partA, M, D, K, r, P, scaling)
  if (r == 4) TB_LORA_BWD(4);
  else if (r == 8) TB_LORA_BWD(8);
  else TB_LORA_BWD(0);
#undef TB_LORA_BWD
  hipLaunchKernelGGL(lora_slab_reduce_kernel, GRID1D((nB + nA) / 4), dim3(256), 0, s, partB, partA, dB, dA, nB, nA, nslab, scaling);
  TB_CHECK_LAUNCH();
  return TB_OK;
}
