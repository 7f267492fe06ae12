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
// Two launches, no partial-sum workspace (round 3; before: 8-row slabs writing 11 MB of per-slab partials + a reducer, 43 us per layer):
//   launch 1: blocks [0, M/16)       -> (a) for 16 rows each (16 threads per row, B staged once per block in LDS as the fp16 values the forward used)
//             blocks [M/16, +P*D/32) -> (b) for a 32-column panel of dY over ALL rows: 64 row slots x 4 column groups accumulate in registers,
//                                       the 64 slots are summed through LDS in slot order (deterministic) and added to dB
//   launch 2: P * K/32 blocks        -> (c) the same panel product with (x, dt) in place of (dY, t)
constexpr int LORA_MAXR = 24;  // P * r
constexpr int LORA_DT_ROWS = 16, LORA_PANEL = 32, LORA_SLOTS = 64, LORA_CG = LORA_PANEL / 8;

// red[slot][c][j] = sum over the rows m = slot (mod LORA_SLOTS) of L[m, c0 + c] * S[m, s0 + j]  (c < LORA_PANEL, j < r <= RR); thread (slot = tid / LORA_CG,
// g = tid % LORA_CG) owns columns 8 g .. 8 g + 7; the callers add the LORA_SLOTS slot sums in slot order
template <int RR>
__device__ __forceinline__ void lora_panel_dot(const f16* __restrict__ Lm, int64_t ldl, int64_t c0, const f16* __restrict__ Sm, int64_t lds_, int s0,
                                               int r, int64_t M, float* red /* [LORA_SLOTS][LORA_PANEL][RR] */) {
  const int tid = threadIdx.x, slot = tid / LORA_CG, g = tid % LORA_CG;
  float acc[8][RR];
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int j = 0; j < RR; ++j) acc[e][j] = 0.f;
  // rows in batches of U: all loads of a batch are issued before its arithmetic (one row at a time the loop was a chain of L2 latencies:
  // 39 dependent round trips per thread made the whole backward 40 us)
  constexpr int U = 10;
  const f16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t mb = slot; mb < M; mb += LORA_SLOTS * U) {
    f16x8 lv[U];
    float sv[U][RR];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t m = mb + (int64_t)u * LORA_SLOTS;
      const bool ok = m < M;
      lv[u] = ok ? *(const f16x8*)(Lm + m * ldl + c0 + g * 8) : z8;
      if (r == RR && ((lds_ | s0) % RR) == 0) {  // one aligned vector per row (r = 4: 8 bytes, r = 8: 16 bytes)
        typedef __attribute__((ext_vector_type(RR))) f16 hvec;
        hvec h;
#pragma unroll
        for (int j = 0; j < RR; ++j) h[j] = (f16)0.f;
        if (ok) h = *(const hvec*)(Sm + m * lds_ + s0);
#pragma unroll
        for (int j = 0; j < RR; ++j) sv[u][j] = (float)h[j];
      } else {
#pragma unroll
        for (int j = 0; j < RR; ++j) sv[u][j] = (ok && j < r) ? (float)Sm[m * lds_ + s0 + j] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float l = (float)lv[u][e];
#pragma unroll
        for (int j = 0; j < RR; ++j) acc[e][j] += l * sv[u][j];
      }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int j = 0; j < RR; ++j) red[(slot * LORA_PANEL + g * 8 + e) * RR + j] = acc[e][j];
  __syncthreads();
}

template <int RR>
__device__ __forceinline__ void lora_da_panel(const f16* __restrict__ x, int64_t ldx, const f16* __restrict__ dt, int64_t lddt,
                                              float* __restrict__ dA, int64_t M, int K, int r, int block, float* lora_smem) {
  const int tid = threadIdx.x;
  const int panels = K / LORA_PANEL;
  const int p = block / panels;
  const int64_t k0 = (int64_t)(block - p * panels) * LORA_PANEL;
  lora_panel_dot<RR>(x, ldx, k0, dt, lddt, p * r, r, M, lora_smem);
  for (int u = tid; u < LORA_PANEL * RR; u += 256) {
    const int j = u / LORA_PANEL, c = u - j * LORA_PANEL;   // consecutive threads -> consecutive k of one adapter row
    if (j < r) {
      float a = 0.f;
#pragma unroll 8
      for (int sl = 0; sl < LORA_SLOTS; ++sl) a += lora_smem[(sl * LORA_PANEL + c) * RR + j];
      dA[(int64_t)(p * r + j) * K + k0 + c] += a;
    }
  }
}

template <int RR, int NG>  // RR = 4 (r <= 4) or 8 (r <= 8); NG = D / 128 when the slabs hold every dY vector of a thread at once (P <= 3), else 0
__global__ __launch_bounds__(256) void lora_bwd_dt_db_kernel(const f16* __restrict__ dY, int64_t lddy, const f16* __restrict__ t, int64_t ldt,
                                                             const float* __restrict__ Bcat, f16* __restrict__ dt, int64_t lddt,
                                                             float* __restrict__ dB, int64_t M, int D, int r, int P, float scaling, int nslab,
                                                             const f16* __restrict__ pend_x, int64_t pend_ldx, const f16* __restrict__ pend_dt,
                                                             int64_t pend_lddt, float* __restrict__ pend_dA, int K) {
  extern __shared__ __attribute__((aligned(16))) float lora_smem[];
  const int tid = threadIdx.x;
  const int nb_panels = P * D / LORA_PANEL;
  if ((int)blockIdx.x >= nslab + nb_panels) {
    // ---- (c): the dA panels of the adapter set whose dt the PREVIOUS launch of the chain wrote (tb_lora_bwd_chain): they depend on nothing this
    // launch produces, and slabs + panels of one text-encoder layer do not fill the chip
    lora_da_panel<RR>(pend_x, pend_ldx, pend_dt, pend_lddt, pend_dA, M, K, r, (int)blockIdx.x - nslab - nb_panels, lora_smem);
    return;
  }
  if ((int)blockIdx.x < nslab) {
    // ---- (a): 16 rows, 16 threads per row; thread c of a row takes the 8-column groups n = 8 c (mod 128)
    // Every global load of the block is issued before anything waits: the thread's dY vectors of all P adapters (<= LORA_DYV), then its share
    // of B.  (As loops of [load, use] the block was a chain of 2 + 2 P dependent round trips, and with 150-190 blocks on 256 CUs nothing else
    // covers them: 15.4 us per launch on the text encoder's critical path.)
    const int i = tid >> 4, c = tid & 15;
    const int64_t m = (int64_t)blockIdx.x * LORA_DT_ROWS + i;
    const bool live = m < M;
    constexpr bool hoist = NG > 0;   // (host: D == 128 NG, P <= 3)
    constexpr int NGN = NG ? NG : 1;
    const f16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
    f16x8 dyv[3][NGN];
    if (hoist) {
#pragma unroll
      for (int pq = 0; pq < 3; ++pq)
#pragma unroll
        for (int u = 0; u < NGN; ++u)
          dyv[pq][u] = (live && pq < P) ? *(const f16x8*)(dY + m * lddy + (int64_t)pq * D + c * 8 + 128 * u) : z8;
    }
    float* Bs = lora_smem;  // [P * D][RR], fp16-rounded
    if (r == RR) {  // same layout: a 16-byte copy, twelve vectors in flight per thread
      const int nvec = P * D * RR / 4;
      for (int u0 = tid; u0 < nvec; u0 += 256 * 12) {
        f32x4 v[12];
#pragma unroll
        for (int q = 0; q < 12; ++q) {
          const int u = u0 + 256 * q;
          v[q] = u < nvec ? *(const f32x4*)(Bcat + (int64_t)u * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int q = 0; q < 12; ++q) {
          const int u = u0 + 256 * q;
          if (u < nvec) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[q][e] = (float)(f16)v[q][e];
            *(f32x4*)(Bs + (int64_t)u * 4) = v[q];
          }
        }
      }
    } else {
      for (int u = tid; u < P * D * RR; u += 256) {
        const int n = u / RR, j = u - n * RR;
        Bs[u] = j < r ? (float)(f16)Bcat[(int64_t)n * r + j] : 0.f;
      }
    }
    __syncthreads();
    for (int p = 0; p < P; ++p) {
      float acc[RR];
#pragma unroll
      for (int j = 0; j < RR; ++j) acc[j] = 0.f;
      if (live && hoist) {
        // (same order as the loop below: the thread's column groups ascending, e, j)
#pragma unroll
        for (int u = 0; u < NGN; ++u) {
          const float* bp = Bs + ((int64_t)p * D + c * 8 + 128 * u) * RR;
          const f16x8 dv = p == 0 ? dyv[0][u] : (p == 1 ? dyv[1][u] : dyv[2][u]);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = (float)dv[e];
#pragma unroll
            for (int j = 0; j < RR; ++j) acc[j] += d * bp[e * RR + j];
          }
        }
      } else if (live) {
        for (int nb = c * 8; nb < D; nb += 128 * 4) {  // four column groups per batch, loads first
          f16x8 dy[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int n = nb + 128 * u;
            dy[u] = n < D ? *(const f16x8*)(dY + m * lddy + (int64_t)p * D + n) : z8;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int n = nb + 128 * u;
            if (n < D) {
              const float* bp = Bs + ((int64_t)p * D + n) * RR;
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const float d = (float)dy[u][e];
#pragma unroll
                for (int j = 0; j < RR; ++j) acc[j] += d * bp[e * RR + j];
              }
            }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < RR; ++j) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) acc[j] += __shfl_xor(acc[j], o, 64);
      }
      if (live && c < r) {
        float v = acc[0];
#pragma unroll
        for (int j = 1; j < RR; ++j) v = c == j ? acc[j] : v;
        dt[m * lddt + p * r + c] = (f16)(v * scaling);
      }
    }
    return;
  }
  // ---- (b): one LORA_PANEL-column panel of dY (inside one adapter: D % LORA_PANEL == 0) against that adapter's columns of t
  const int64_t n0 = (int64_t)((int)blockIdx.x - nslab) * LORA_PANEL;
  const int p = (int)(n0 / D);
  lora_panel_dot<RR>(dY, lddy, n0, t, ldt, p * r, r, M, lora_smem);
  for (int u = tid; u < LORA_PANEL * RR; u += 256) {
    const int c = u / RR, j = u - c * RR;
    if (j < r) {
      float a = 0.f;
#pragma unroll 8
      for (int sl = 0; sl < LORA_SLOTS; ++sl) a += lora_smem[(sl * LORA_PANEL + c) * RR + j];
      dB[(n0 + c) * r + j] += a * scaling;
    }
  }
}

template <int RR>
__global__ __launch_bounds__(256) void lora_bwd_da_kernel(const f16* __restrict__ x, int64_t ldx, const f16* __restrict__ dt, int64_t lddt,
                                                          float* __restrict__ dA, int64_t M, int K, int r, int P) {
  extern __shared__ __attribute__((aligned(16))) float lora_smem[];
  (void)P;
  lora_da_panel<RR>(x, ldx, dt, lddt, dA, M, K, r, (int)blockIdx.x, lora_smem);
}

int g_lora_hoist = 1;   // tb_lora_set_variant: 0 = the dt slabs load dY in loops (A/B)
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
  (void)M, (void)D, (void)K, (void)r, (void)P;
  return 4;  // the round-3 kernels need no scratch (kept in the ABI: callers still pass a pointer)
}

extern "C" int tb_lora_set_variant(int v) {   // A/B knob; returns the previous value
  const int old = g_lora_hoist;
  g_lora_hoist = v;
  return old;
}

// One adapter set's backward as a link of a chain (the text encoder walks its layers last to first): the dt / dB launch of THIS set also carries
// the dA panels of the PENDING set (pend_x / pend_dt / pend_dA of the previous link: same M, K, r, P; null = none), whose inputs are complete
// and which nothing later in the layer waits for; da_now != 0 launches this set's own dA panels behind it (the last link, or a caller
// that does not chain).  Arithmetic and summation order per set are those of tb_lora_bwd.
extern "C" int tb_lora_bwd_chain(const void* dY, int64_t lddy, const void* x, int64_t ldx, const void* t, int64_t ldt, const float* Bcat,
                                 void* dt, int64_t lddt, float* dA, float* dB, int64_t M, int D, int K, int r, int P, float scaling,
                                 const void* pend_x, int64_t pend_ldx, const void* pend_dt, int64_t pend_lddt, float* pend_dA, int da_now,
                                 tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!dY || !x || !t || !Bcat || !dt || !dA || !dB || r <= 0 || r > 8 || P <= 0 || P * r > LORA_MAXR || M <= 0) return TB_EINVAL;
  if (D % LORA_PANEL || K % LORA_PANEL || lddy % 8 || ldx % 8) return TB_EINVAL;  // 16-byte vector accesses, panels inside one adapter
  if (((uintptr_t)dY) % 16 || ((uintptr_t)x) % 16) return TB_EINVAL;
  const bool pend = pend_x || pend_dt || pend_dA;
  if (pend && (!pend_x || !pend_dt || !pend_dA || pend_ldx % 8 || ((uintptr_t)pend_x) % 16 || pend_dt == dt)) return TB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int nslab = (int)((M + LORA_DT_ROWS - 1) / LORA_DT_ROWS);
  const int RR = r <= 4 ? 4 : 8;
  const size_t red_bytes = (size_t)LORA_SLOTS * LORA_PANEL * RR * sizeof(float);
  const size_t b_bytes = (size_t)P * D * RR * sizeof(float);
  const size_t lds1 = red_bytes > b_bytes ? red_bytes : b_bytes;
  if (lds1 > 160 * 1024) return TB_EINVAL;
  const int nda = P * (K / LORA_PANEL);
  const int NG = (g_lora_hoist && P <= 3 && (D == 768 || D == 1024)) ? D / 128 : 0;
#define TB_LORA_BWD(RRV, NGV)                                                                                                                   \
  {                                                                                                                                             \
    if (lds1 > 64 * 1024 &&                                                                                                                     \
        hipFuncSetAttribute((const void*)lora_bwd_dt_db_kernel<RRV, NGV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=            \
            hipSuccess)                                                                                                                         \
      return TB_ELAUNCH;                                                                                                                        \
    hipLaunchKernelGGL((lora_bwd_dt_db_kernel<RRV, NGV>), dim3(nslab + P * D / LORA_PANEL + (pend ? nda : 0)), dim3(256), lds1, s,              \
                       (const f16*)dY, lddy, (const f16*)t, ldt, Bcat, (f16*)dt, lddt, dB, M, D, r, P, scaling, nslab, (const f16*)pend_x,      \
                       pend_ldx, (const f16*)pend_dt, pend_lddt, pend_dA, K);                                                                   \
    if (da_now)                                                                                                                                 \
      hipLaunchKernelGGL(lora_bwd_da_kernel<RRV>, dim3(nda), dim3(256), red_bytes, s, (const f16*)x, ldx, (const f16*)dt, lddt, dA, M, K, r,    \
                         P);                                                                                                                    \
  }
  if (RR == 4) {
    if (NG == 6) TB_LORA_BWD(4, 6) else if (NG == 8) TB_LORA_BWD(4, 8) else TB_LORA_BWD(4, 0)
  } else {
    if (NG == 6) TB_LORA_BWD(8, 6) else TB_LORA_BWD(8, 0)   // (<8, 8>: 32 dY vectors + the B staging spill)
  }
#undef TB_LORA_BWD
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int tb_lora_bwd(const void* dY, int64_t lddy, const void* x, int64_t ldx, const void* t, int64_t ldt, const float* Bcat,
                           void* dt, int64_t lddt, float* dA, float* dB, float* ws, int64_t M, int D, int K, int r, int P,
                           float scaling, tb_stream_t stream) {
  (void)ws;
  return tb_lora_bwd_chain(dY, lddy, x, ldx, t, ldt, Bcat, dt, lddt, dA, dB, M, D, K, r, P, scaling, nullptr, 0, nullptr, 0, nullptr, 1, stream);
}
