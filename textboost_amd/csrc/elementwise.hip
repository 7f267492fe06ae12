// Small streaming kernels of the TextBoost step on gfx950: scheduler noise, 4-channel convs at the UNet boundary,
// losses, GEGLU backward, nearest-upsample backward, residual adds, timestep embedding.
// Reference call sites are cited per kernel (all in /root/reference/train_textboost.py unless noted).
#include "common.h"
#include "../../include/textboost_hip.h"

namespace {

// ---- :1052 noise_scheduler.add_noise (+ :1073 get_velocity): fp32 NCHW in, fp16 NCHW noisy out
__global__ void add_noise_kernel(const float* __restrict__ x0, const float* __restrict__ noise, const int64_t* __restrict__ t,
                                 const float* __restrict__ acp, f16* __restrict__ noisy, float* __restrict__ velocity,
                                 int64_t per_sample, int64_t total) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int b = (int)(i / per_sample);
  const float a = acp[t[b]];
  const float sa = sqrtf(a), sb = sqrtf(1.f - a);
  const float x = x0[i], n = noise[i];
  noisy[i] = (f16)(sa * x + sb * n);
  if (velocity) velocity[i] = sa * n - sb * x;
}

// ---- Timesteps(320, flip_sin_to_cos=True, shift 0) of the UNet time embedding: out fp16 [B, dim] = [cos | sin]
__global__ void timestep_embed_kernel(const int64_t* __restrict__ t, f16* __restrict__ out, int B, int dim) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * dim) return;
  const int b = i / dim, j = i - b * dim, half = dim / 2;
  const int k = j < half ? j : j - half;
  const float f = expf(-9.210340371976184f * (float)k / (float)half);
  const float arg = (float)t[b] * f;
  out[i] = (f16)(j < half ? cosf(arg) : sinf(arg));
}

// ---- 3x3 conv with 4 channels on the NCHW side (UNet conv_in forward; conv_out input-gradient), fp32 accumulate.
// out[m, co] (NHWC fp16, row stride ldo) = bias[co] + sum_{tap, ci<4} in[b, ci, y+sign*(ky-1), x+sign*(kx-1)] * Wp[(tap*4+ci)*Cout + co]
// A thread owns 8 output channels of PX consecutive pixels of one image row: the 8 weights of a (tap, input channel) are loaded once for
// all PX pixels and the row's inputs are shared between neighbouring taps (one thread per pixel issued 108 loads per 288 FMAs: load-issue bound).
template <typename TIN, int CIN, int PX>
__global__ __launch_bounds__(256) void conv4_to_nhwc_kernel(const TIN* __restrict__ in, const float* __restrict__ Wp,
                                                            const float* __restrict__ bias, f16* __restrict__ out, int64_t ldo,
                                                            int B, int H, int W, int Cout, int sign, float in_scale) {
  const int groups = Cout >> 3;
  const int wq = (W + PX - 1) / PX;  // pixel groups per row
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)B * H * wq * groups;
  if (idx >= total) return;
  const int cg = (int)(idx % groups);
  const int64_t pg = idx / groups;
  const int xq = (int)(pg % wq);
  const int64_t by = pg / wq;
  const int y = (int)(by % H), b = (int)(by / H);
  const int x0 = xq * PX;
  float acc[PX][8];
#pragma unroll
  for (int px = 0; px < PX; ++px)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[px][e] = bias ? bias[cg * 8 + e] : 0.f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int sy = y + sign * (ky - 1);
    if (sy < 0 || sy >= H) continue;
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
      const TIN* row = in + (((int64_t)b * CIN + ci) * H + sy) * W;
      float v[PX + 2];  // inputs at x0 - 1 .. x0 + PX
#pragma unroll
      for (int j = 0; j < PX + 2; ++j) {
        const int sx = x0 - 1 + j;
        v[j] = (sx >= 0 && sx < W) ? (float)row[sx] * in_scale : 0.f;
      }
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const float* w = Wp + (int64_t)((ky * 3 + kx) * CIN + ci) * Cout + cg * 8;
        const f32x4 w0 = *(const f32x4*)w, w1 = *(const f32x4*)(w + 4);
#pragma unroll
        for (int px = 0; px < PX; ++px) {
          const float a = v[px + 1 + sign * (kx - 1)];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[px][e] += a * w0[e];
            acc[px][4 + e] += a * w1[e];
          }
        }
      }
    }
  }
#pragma unroll
  for (int px = 0; px < PX; ++px)
    if (x0 + px < W) {
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (f16)acc[px][e];
      *(f16x8*)(out + (((int64_t)b * H + y) * W + x0 + px) * ldo + cg * 8) = o;
    }
}

// ---- UNet conv_out forward: NHWC fp16 [M, C] -> NCHW fp16 [B, 4, H, W]; one wave per output pixel.
// (A version with the 46 KB of filters staged in LDS and 16-byte loads measured SLOWER, 96 vs 71 us: 72 LDS reads per pixel and wave.)
// Wp fp32 [4][9][C]
__global__ __launch_bounds__(256) void conv_to4_kernel(const f16* __restrict__ in, int64_t ldi, const float* __restrict__ Wp,
                                                       const float* __restrict__ bias, f16* __restrict__ out, int B, int H, int W,
                                                       int C) {
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
    const f16* src = in + (((int64_t)b * H + sy) * W + sx) * ldi;
    for (int c = lane; c < C; c += 64) {
      const float v = (float)src[c];
#pragma unroll
      for (int co = 0; co < 4; ++co) acc[co] += v * Wp[((int64_t)co * 9 + tap) * C + c];
    }
  }
#pragma unroll
  for (int co = 0; co < 4; ++co) {
    const float s = wave_sum(acc[co]);
    if (lane == 0) out[(((int64_t)b * 4 + co) * H + y) * W + x] = (f16)(s + (bias ? bias[co] : 0.f));
  }
}

// ---- the same two boundary convolutions on the matrix cores (round 5).  The kernels above are load-ISSUE bound, not memory bound: conv_to4 spends
// 225 scalar-width load instructions per pixel and wave (74 us for 21 MB of input), conv4_to_nhwc 144 per thread (59 us for 21 MB of output).  As
// 16 x 16 x 32 MFMA tiles over 16 consecutive pixels of an image row a wave issues one 16-byte load per lane and MFMA, the 16-bit weights wait
// in the LDS (staged once per workgroup from the caller's fp32 pack: exact for a model cast by unet.to(fp16), train_textboost.py:937), fp32
// accumulation as before.  Shapes outside (W % 16, C % 32) keep the kernels above.
//
// conv_to4: D[co, px] = sum_k W[co, k] X[px, k], k = (tap, channel).  srcA = weight rows (4 real, 12 zero), srcB = pixel rows: lane l holds
// pixel l % 16 and its accumulator rows 4 (l / 16) .. + 3 are output channels -- lanes 0..15 own the 4 x 16 results, one 32-byte run per channel.
constexpr int C4_GROUPS = 1;   // 16-pixel groups per wave (64 pixels per workgroup: 512 workgroups at 64 x 64 x 8, two per CU)
// KCT = C / 32 at compile time (10: the 320-channel UNets) unrolls the whole 9 x KCT tile loop -- straight-line code, so the compiler issues the
// 16-byte pixel loads many MFMAs ahead; 0 = run-time channel count (rolled loops)
template <int KCT>
__global__ __launch_bounds__(256) void conv_to4_mfma_kernel(const f16* __restrict__ in, int64_t ldi, const float* __restrict__ Wp,
                                                            const float* __restrict__ bias, f16* __restrict__ out, int B, int H, int W, int C) {
  extern __shared__ __attribute__((aligned(16))) unsigned char c4_smem[];
  f16* ws = reinterpret_cast<f16*>(c4_smem);   // [tap][kc][co 4][32]
  const int KC = KCT ? KCT : C >> 5, t = threadIdx.x;
  {
    constexpr int NIT = KCT ? (9 * KCT * 32 + 255) / 256 : 1;
    const int units = 9 * KC * 4 * 8;   // units of 4 weights
    auto put = [&](int i, const f32x4& w) {
      const int q = i & 7, co = (i >> 3) & 3, r = i >> 5, kc = r % KC, tap = r / KC;
      f16x4 h;
#pragma unroll
      for (int e = 0; e < 4; ++e) h[e] = (f16)w[e];
      *(f16x4*)(ws + ((tap * KC + kc) * 4 + co) * 32 + q * 4) = h;
    };
    auto src_of = [&](int i) {
      const int q = i & 7, co = (i >> 3) & 3, r = i >> 5, kc = r % KC, tap = r / KC;
      return Wp + ((int64_t)co * 9 + tap) * C + kc * 32 + q * 4;
    };
    if constexpr (KCT != 0) {
      f32x4 w[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int i = t + it * 256;
        w[it] = *(const f32x4*)src_of(i < units ? i : 0);
      }
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int i = t + it * 256;
        if (i < units) put(i, w[it]);
      }
    } else {
      for (int i = t; i < units; i += 256) put(i, *(const f32x4*)src_of(i));
    }
  }
  __syncthreads();
  const int lane = t & 63, wave = t >> 6, p = lane & 15, quad = lane >> 4;
  const int gpr = W >> 4;   // groups per image row
  const int64_t ngroups = (int64_t)B * H * gpr;
  const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 1
  for (int g = 0; g < C4_GROUPS; ++g) {
    const int64_t grp = ((int64_t)blockIdx.x * 4 + wave) * C4_GROUPS + g;
    if (grp >= ngroups) break;
    const int xg = (int)(grp % gpr);
    const int64_t by = grp / gpr;
    const int y = (int)(by % H), b = (int)(by / H), x = xg * 16 + p;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const f16* wt0 = ws + (p & 3) * 32 + quad * 8;
    if constexpr (KCT != 0) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const int sy = y + ky - 1, sx = x + kx - 1;
        const bool ok = sx >= 0 && sx < W && sy >= 0 && sy < H;   // (a row outside the image costs its MFMAs on zeros: two of 64 rows)
        const f16* src = in + (((int64_t)b * H + (ok ? sy : y)) * W + (ok ? sx : x)) * ldi + quad * 8;
#pragma unroll
        for (int kc = 0; kc < KCT; ++kc) {
          f16x8 xv = *(const f16x8*)(src + kc * 32);
          f16x8 wv = *(const f16x8*)(wt0 + (tap * KCT + kc) * 128);
          if (!ok) xv = zero8;
          if (p >= 4) wv = zero8;
          if (kc & 1) acc1 = TB_MFMA_16x16x32(wv, xv, acc1);
          else acc0 = TB_MFMA_16x16x32(wv, xv, acc0);
        }
      }
    } else {
#pragma unroll 1
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const int sy = y + ky - 1, sx = x + kx - 1;
        if (sy < 0 || sy >= H) continue;   // (wave-uniform)
        const bool ok = sx >= 0 && sx < W;
        const f16* src = in + (((int64_t)b * H + sy) * W + (ok ? sx : x)) * ldi + quad * 8;
        const f16* wt = wt0 + tap * KC * 128;
#pragma unroll 2
        for (int kc = 0; kc < KC; kc += 2) {
          f16x8 x0 = *(const f16x8*)(src + kc * 32), x1 = kc + 1 < KC ? *(const f16x8*)(src + kc * 32 + 32) : zero8;
          f16x8 w0 = *(const f16x8*)(wt + kc * 128), w1 = kc + 1 < KC ? *(const f16x8*)(wt + kc * 128 + 128) : zero8;
          if (!ok) x0 = zero8, x1 = zero8;
          if (p >= 4) w0 = zero8, w1 = zero8;
          acc0 = TB_MFMA_16x16x32(w0, x0, acc0);
          acc1 = TB_MFMA_16x16x32(w1, x1, acc1);
        }
      }
    }
    if (quad == 0) {
#pragma unroll
      for (int co = 0; co < 4; ++co)
        out[(((int64_t)b * 4 + co) * H + y) * W + x] = (f16)(acc0[co] + acc1[co] + (bias ? bias[co] : 0.f));
    }
  }
}

// conv4_to_nhwc: D[co, px] = sum_{k < 9 CIN} W[k, co] P[px, k], P = the 3 x 3 x CIN patch of the NCHW side (k = tap * CIN + ci, zero-padded to 64).
// The weight rows of a 32-channel block are permuted so that a lane's accumulators of the block's two 16-row tiles are 8 CONSECUTIVE output
// channels of its pixel: MFMA row a of tile h is channel 32 J + 8 (a / 4) + 4 h + (a % 4) -> one 16-byte store per lane, block and pixel.
constexpr int CI_LDW = 72;   // f16 per weight row in the LDS (64 k + pad: conflict-free 16-byte reads of 16 consecutive rows)
template <typename TIN, int CIN>
__global__ __launch_bounds__(256) void conv4_to_nhwc_mfma_kernel(const TIN* __restrict__ in, const float* __restrict__ Wp,
                                                                 const float* __restrict__ bias, f16* __restrict__ out, int64_t ldo, int B, int H,
                                                                 int W, int Cout, int sign, float in_scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char c4_smem[];
  f16* ws = reinterpret_cast<f16*>(c4_smem);                       // [Cout][CI_LDW]: row co = the 9 CIN weights of output channel co over k, zeros behind
  float* bs = reinterpret_cast<float*>(ws + (size_t)Cout * CI_LDW);   // [Cout] bias (zeros when absent)
  constexpr int KR = 9 * CIN, KZ = (KR + 3) & ~3;   // real k, first k of the 8-byte zero fill
  const int t = threadIdx.x, c4n = Cout >> 2;
  for (int i = t; i < KR * c4n; i += 256) {   // 4 channels of one k per thread: 16-byte global loads, a row of the pack is contiguous over channels
    const int k = i / c4n, c4 = i - k * c4n;
    const f32x4 w = *(const f32x4*)(Wp + (int64_t)k * Cout + c4 * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) ws[(c4 * 4 + e) * CI_LDW + k] = (f16)w[e];
  }
  for (int i = t; i < Cout * ((64 - KZ) / 4 + 1); i += 256) {   // the padding k: KR .. KZ - 1 one by one (slot 0), then 8-byte zeros
    const int per = (64 - KZ) / 4 + 1, co = i / per, j = i - co * per;
    if (j == 0) {
      for (int k = KR; k < KZ; ++k) ws[co * CI_LDW + k] = (f16)0.f;
    } else {
      *(f16x4*)(ws + co * CI_LDW + KZ + (j - 1) * 4) = f16x4{0, 0, 0, 0};
    }
  }
  for (int i = t; i < Cout; i += 256) bs[i] = bias ? bias[i] : 0.f;
  const int lane = t & 63, wave = t >> 6, p = lane & 15, quad = lane >> 4;
  const int gpr = W >> 4;
  const int64_t ngroups = (int64_t)B * H * gpr;
  // this lane's slice of its first pixel's patch is fetched before the barrier (independent of the staged weights)
  auto patch = [&](int64_t grp, f16x8 (&pf)[2], int& b, int& y, int& x) {
    const int xg = (int)(grp % gpr);
    const int64_t by = grp / gpr;
    y = (int)(by % H), b = (int)(by / H), x = xg * 16 + p;
    float v[2][8];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = s2 * 32 + quad * 8 + e;
        const int tap = k / CIN, ci = k - tap * CIN, ky = tap / 3, kx = tap - ky * 3;
        const int sy = y + sign * (ky - 1), sx = x + sign * (kx - 1);
        const bool ok = k < KR && sy >= 0 && sy < H && sx >= 0 && sx < W;
        const TIN raw = in[ok ? (((int64_t)b * CIN + ci) * H + sy) * W + sx : 0];   // (clamped address: no branch around the load)
        v[s2][e] = ok ? (float)raw * in_scale : 0.f;
      }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int e = 0; e < 8; ++e) pf[s2][e] = (f16)v[s2][e];
  };
  const int64_t grp0 = ((int64_t)blockIdx.x * 4 + wave) * C4_GROUPS;
  f16x8 pf[2];
  int b = 0, y = 0, x = 0;
  if (grp0 < ngroups) patch(grp0, pf, b, y, x);
  __syncthreads();
#pragma unroll 1
  for (int g = 0; g < C4_GROUPS; ++g) {
    const int64_t grp = grp0 + g;
    if (grp >= ngroups) break;
    if (g > 0) patch(grp, pf, b, y, x);
    f16* dst = out + (((int64_t)b * H + y) * W + x) * ldo + quad * 8;
#pragma unroll 2
    for (int J = 0; J < Cout; J += 32) {
      f32x4 a[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int co = J + 8 * (p >> 2) + 4 * h + (p & 3);   // channel of MFMA row p of tile h
        const f16* wr = ws + co * CI_LDW + quad * 8;
        a[h] = f32x4{0.f, 0.f, 0.f, 0.f};
        a[h] = TB_MFMA_16x16x32(*(const f16x8*)wr, pf[0], a[h]);
        if (KR > 32) a[h] = TB_MFMA_16x16x32(*(const f16x8*)(wr + 32), pf[1], a[h]);
      }
      // accumulator i of tile h = channel J + 8 quad + 4 h + i of pixel p
      const f32x4 b0 = *(const f32x4*)(bs + J + 8 * quad), b1 = *(const f32x4*)(bs + J + 8 * quad + 4);
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (f16)(a[0][e] + b0[e]), o[4 + e] = (f16)(a[1][e] + b1[e]);
      *(f16x8*)(dst + J) = o;
    }
  }
}

// ---- :1085-1090 loss = mean((pred.float() - target.float())^2); also d(loss*loss_scale)/d pred.
// Single block (N = B*4*h*w is small); loss_out[0] = mse (unscaled). dpred fp32 NCHW.
__global__ __launch_bounds__(1024) void mse_loss_kernel(const f16* __restrict__ pred, const float* __restrict__ target,
                                                        float* __restrict__ dpred, float* __restrict__ loss_out,
                                                        const float* __restrict__ loss_scale, int64_t N) {
  __shared__ float red[16];
  const float ls = loss_scale ? loss_scale[0] : 1.f;
  const float gcoef = 2.f / (float)N * ls;
  float a = 0.f;
  for (int64_t i = threadIdx.x; i < N; i += 1024) {
    const float d = (float)pred[i] - target[i];
    a += d * d;
    if (dpred) dpred[i] = gcoef * d;
  }
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += red[i];
    loss_out[0] = s / (float)N;
  }
}

// multi-block first stage of the same loss: block b writes its sum of squared differences to part[b]; tb_mse_loss then runs sum_kernel
// (deterministic two-stage reduction; the single 1024-thread block above needs ~64 us for the metric's 131k elements, this ~8 us)
__global__ __launch_bounds__(256) void mse_partial_kernel(const f16* __restrict__ pred, const float* __restrict__ target,
                                                          float* __restrict__ dpred, float* __restrict__ part,
                                                          const float* __restrict__ loss_scale, int64_t N) {
  __shared__ float red[4];
  const float gcoef = 2.f / (float)N * (loss_scale ? loss_scale[0] : 1.f);
  float a = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
    const float d = (float)pred[i] - target[i];
    a += d * d;
    if (dpred) dpred[i] = gcoef * d;
  }
  a = block_sum_256(a, red);
  if (threadIdx.x == 0) part[blockIdx.x] = a;
}

// ---- :1099-1106 knowledge-preservation loss, cos variant: mean_rows(1 - cos(h, h0)); grad wrt h (fp32).
// One wave per row; partial[row] = 1 - cos;  dh = -(weight*loss_scale/M) * (h0/(|h||h0|) - cos * h/|h|^2)
template <typename T0>
__global__ __launch_bounds__(256) void kpl_cos_kernel(const float* __restrict__ h, int64_t ldh, const T0* __restrict__ h0,
                                                      int64_t ldh0, float* __restrict__ dh, int64_t lddh, float* __restrict__ partial,
                                                      const float* __restrict__ loss_scale, float weight, int64_t M, int D) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float dot = 0.f, n1 = 0.f, n2 = 0.f;
  for (int c = lane; c < D; c += 64) {
    const float a = h[row * ldh + c], b = (float)h0[row * ldh0 + c];
    dot += a * b;
    n1 += a * a;
    n2 += b * b;
  }
  dot = wave_sum(dot);
  n1 = wave_sum(n1);
  n2 = wave_sum(n2);
  const float eps = 1e-8f;
  const float na = fmaxf(sqrtf(n1), eps), nb = fmaxf(sqrtf(n2), eps);
  const float cosv = dot / (na * nb);
  if (lane == 0) partial[row] = 1.f - cosv;
  if (dh) {
    const float coef = -(weight * (loss_scale ? loss_scale[0] : 1.f)) / (float)M;
    for (int c = lane; c < D; c += 64) {
      const float a = h[row * ldh + c], b = (float)h0[row * ldh0 + c];
      dh[row * lddh + c] = coef * (b / (na * nb) - cosv * a / (na * na));
    }
  }
}

// ---- :1104-1105 KPL mse variant: mean over all M*D elements of (h - h0)^2; one wave per row, partial[row] = row sum / D
template <typename T0>
__global__ __launch_bounds__(256) void kpl_mse_kernel(const float* __restrict__ h, int64_t ldh, const T0* __restrict__ h0, int64_t ldh0,
                                                      float* __restrict__ dh, int64_t lddh, float* __restrict__ partial,
                                                      const float* __restrict__ loss_scale, float weight, int64_t M, int D) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float coef = 2.f * weight * (loss_scale ? loss_scale[0] : 1.f) / ((float)M * (float)D);
  float a = 0.f;
  for (int c = lane; c < D; c += 64) {
    const float d = h[row * ldh + c] - (float)h0[row * ldh0 + c];
    a += d * d;
    if (dh) dh[row * lddh + c] = coef * d;
  }
  a = wave_sum(a);
  if (lane == 0) partial[row] = a / (float)D;
}

// deterministic sum of n floats by one block; out[0] = scale * sum
__global__ __launch_bounds__(256) void sum_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n, float scale) {
  __shared__ float red[4];
  float a = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 256) a += x[i];
  a = block_sum_256(a, red);
  if (threadIdx.x == 0) out[0] = a * scale;
}

// ---- GEGLU backward on the packed [h32|g32] column order the forward GEMM epilogue saved:
// dproj[:, blk*64 + j] = dout[:, blk*32+j] * gelu(g);  dproj[:, blk*64+32+j] = dout * h * gelu'(g)
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const f16* __restrict__ dout, int64_t lddo, const f16* __restrict__ raw,
                                                        int64_t ldr, f16* __restrict__ dproj, int64_t lddp, int64_t M, int inner) {
  const int vecs = inner >> 3;  // 8 gate outputs per thread
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * vecs) return;
  const int64_t m = idx / vecs;
  const int v = (int)(idx - m * vecs);
  const int col = v * 8;              // output column (0..inner)
  const int blk = col >> 5, j = col & 31;
  const f16x8 d = *(const f16x8*)(dout + m * lddo + col);
  const f16x8 hh = *(const f16x8*)(raw + m * ldr + blk * 64 + j);
  const f16x8 gg = *(const f16x8*)(raw + m * ldr + blk * 64 + 32 + j);
  f16x8 dh, dg;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float g = (float)gg[e], dd = (float)d[e];
    float ge, dge;
    gelu_erf_both_f(g, ge, dge);
    dh[e] = (f16)(dd * ge);
    dg[e] = (f16)(dd * (float)hh[e] * dge);
  }
  *(f16x8*)(dproj + m * lddp + blk * 64 + j) = dh;
  *(f16x8*)(dproj + m * lddp + blk * 64 + 32 + j) = dg;
}

// ---- F.interpolate(nearest, x2) materialised: U[b, y, x, :] = X[b, y >> 1, x >> 1, :] (NHWC fp16, H x W = the INPUT map).  tb_gemm can fold
// the upsampling into its 3x3 gather, but only the 4-wave kernel does (0.8 PFLOP/s); writing the 4x map (42 MB at 64x64 x 640) costs 12 us and
// lets the halo-resident wide-tile kernel (1.3 PFLOP/s) take the convolution.
__global__ __launch_bounds__(256) void upsample2x_kernel(const f16* __restrict__ X, int64_t ldx, f16* __restrict__ U, int64_t ldu, int B, int H,
                                                         int W, int C) {
  const int vecs = C >> 3;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t M = (int64_t)B * 4 * H * W;
  if (idx >= M * vecs) return;
  const int64_t m = idx / vecs;
  const int v = (int)(idx - m * vecs);
  const int b = (int)(m / (4 * H * W));
  const int rem = (int)(m - (int64_t)b * 4 * H * W);
  const int y = rem / (2 * W), x = rem - y * 2 * W;
  const int64_t src = ((int64_t)b * H + (y >> 1)) * W + (x >> 1);
  *(f16x8*)(U + m * ldu + v * 8) = *(const f16x8*)(X + src * ldx + v * 8);
}

// ---- backward of F.interpolate(nearest, x2): dX[b,y,x,:] = sum of the 2x2 block of dU (NHWC fp16)
__global__ __launch_bounds__(256) void pool2x2_sum_kernel(const f16* __restrict__ dU, int64_t ldu, f16* __restrict__ dX, int64_t ldx,
                                                          int B, int H, int W, int C) {
  const int vecs = C >> 3;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t M = (int64_t)B * H * W;
  if (idx >= M * vecs) return;
  const int64_t m = idx / vecs;
  const int v = (int)(idx - m * vecs);
  const int b = (int)(m / (H * W));
  const int rem = (int)(m - (int64_t)b * H * W);
  const int y = rem / W, x = rem - y * W;
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const int64_t um = ((int64_t)b * 2 * H + 2 * y + dy) * 2 * W + 2 * x + dx;
      const f16x8 u = *(const f16x8*)(dU + um * ldu + v * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += (float)u[e];
    }
  f16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (f16)a[e];
  *(f16x8*)(dX + m * ldx + v * 8) = o;
}

// ---- out = a + b (fp16 rows with strides); gradient merges at residual / skip fan-outs
__global__ __launch_bounds__(256) void add_f16_kernel(const f16* __restrict__ a, int64_t lda, const f16* __restrict__ b, int64_t ldb,
                                                      f16* __restrict__ out, int64_t ldo, int64_t M, int C) {
  const int vecs = C >> 3;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * vecs) return;
  const int64_t m = idx / vecs;
  const int v = (int)(idx - m * vecs);
  const f16x8 x = *(const f16x8*)(a + m * lda + v * 8);
  const f16x8 y = *(const f16x8*)(b + m * ldb + v * 8);
  f16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (f16)((float)x[e] + (float)y[e]);
  *(f16x8*)(out + m * ldo + v * 8) = o;
}

// ---- dtype conversion with strides (fp32 <-> fp16), optional scale: used for ehs.to(fp16) (:1066) and its gradient
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void convert_kernel(const TI* __restrict__ in, int64_t ldi, TO* __restrict__ out, int64_t ldo,
                                                      int64_t M, int C, float scale) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * C) return;
  const int64_t m = idx / C;
  const int c = (int)(idx - m * C);
  out[m * ldo + c] = (TO)((float)in[m * ldi + c] * scale);
}

// ---- row softmax of fp32 scores -> fp16 probabilities (the VAE mid-block attention: one head of 512 channels over 4096 pixels
// does not fit the flash kernels' head-dim range, so it runs as GEMM -> this -> GEMM).  One block per row; upcast_softmax=True.
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S, int64_t lds, f16* __restrict__ P, int64_t ldp,
                                                           int cols) {
  __shared__ float red[4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* s = S + (int64_t)blockIdx.x * lds;
  f16* p = P + (int64_t)blockIdx.x * ldp;
  float mx = -INFINITY;
  for (int c = t * 4; c < cols; c += 1024) {
    const f32x4 v = *(const f32x4*)(s + c);
    mx = fmaxf(fmaxf(mx, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int c = t * 4; c < cols; c += 1024) {
    const f32x4 v = *(const f32x4*)(s + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) sum += __expf(v[e] - mx);
  }
  sum = wave_sum(sum);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  const float inv = 1.f / (red[0] + red[1] + red[2] + red[3]);
  for (int c = t * 4; c < cols; c += 1024) {
    const f32x4 v = *(const f32x4*)(s + c);
    f16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (f16)(__expf(v[e] - mx) * inv);
    *(f16x4*)(p + c) = o;
  }
}

// ---- DiagonalGaussianDistribution.sample() * scaling_factor (train_textboost.py:1036-1037):
// moments fp32 NHWC [B*HW, 2L] = (mean | logvar);  latents NCHW fp32 [B, L, HW] = (mean + exp(0.5 clamp(logvar, -30, 20)) * eps) * scale
__global__ __launch_bounds__(256) void vae_sample_kernel(const float* __restrict__ mom, int64_t ldm, const float* __restrict__ eps,
                                                         float* __restrict__ out, int B, int HW, int Lc, float scale) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over B * L * HW (NCHW order)
  if (idx >= (int64_t)B * Lc * HW) return;
  const int pix = (int)(idx % HW);
  const int c = (int)((idx / HW) % Lc);
  const int b = (int)(idx / ((int64_t)HW * Lc));
  const float* m = mom + ((int64_t)b * HW + pix) * ldm;
  const float lv = fminf(fmaxf(m[Lc + c], -30.f), 20.f);
  out[idx] = (m[c] + __expf(0.5f * lv) * eps[idx]) * scale;
}

// ---- sampling path (train_textboost.py:453-531 log_validation, inference.py) ------------------------------------
// 1x1 conv on a few channels of an NCHW fp32 tensor: out[b,o,p] = scale * sum_c W[o,c] in[b,c,p] + bias[o]   (the VAE's post_quant_conv,
// with the pipeline's latents / scaling_factor folded into `scale`)
__global__ __launch_bounds__(256) void chan_mix_kernel(const float* __restrict__ in, const float* __restrict__ Wm,
                                                       const float* __restrict__ bias, float* __restrict__ out, int B, int C, int HW,
                                                       float scale) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over B * HW
  if (idx >= (int64_t)B * HW) return;
  const int b = (int)(idx / HW), p = (int)(idx - (int64_t)b * HW);
  float v[8];
  for (int c = 0; c < C; ++c) v[c] = in[((int64_t)b * C + c) * HW + p];
  for (int o = 0; o < C; ++o) {
    float a = 0.f;
    for (int c = 0; c < C; ++c) a += Wm[o * C + c] * v[c];
    out[((int64_t)b * C + o) * HW + p] = scale * a + (bias ? bias[o] : 0.f);
  }
}
// One DPM-Solver++(2M) update with classifier-free guidance, elementwise over the latents x [B, n] (fp32):
//   eps = e_u + g (e_c - e_u)            (e = UNet output fp16 [2B, n]: rows 0..B-1 unconditional, B..2B-1 conditional)
//   m0  = (x - sigma_t eps) / alpha_t ;  x <- ca x + cb m0 + cc m_prev ;  m_prev <- m0 ;  x2 (fp16 [2B, n]) <- (x, x) for the next UNet call
__global__ __launch_bounds__(256) void dpm_step_kernel(float* __restrict__ x, const f16* __restrict__ e, float* __restrict__ m_prev,
                                                       f16* __restrict__ x2, int64_t n_per_b, int B, float g, float alpha_t,
                                                       float sigma_t, float ca, float cb, float cc) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t tot = n_per_b * B;
  if (i >= tot) return;
  const float eu = (float)e[i], ec = (float)e[tot + i];
  const float eps = eu + g * (ec - eu);
  const float xv = x[i];
  const float m0 = (xv - sigma_t * eps) / alpha_t;
  const float xn = ca * xv + cb * m0 + cc * m_prev[i];
  m_prev[i] = m0;
  x[i] = xn;
  x2[i] = (f16)xn;
  x2[tot + i] = (f16)xn;
}
// image = (decoded / 2 + 0.5).clamp(0, 1): NHWC fp32 [B*HW, ld >= C] -> NCHW fp32 [B, C, HW]   (StableDiffusionPipeline post-processing)
__global__ __launch_bounds__(256) void vae_image_kernel(const float* __restrict__ in, int64_t ld, float* __restrict__ out, int B, int HW,
                                                        int C) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over B * C * HW (NCHW order: coalesced stores)
  if (idx >= (int64_t)B * C * HW) return;
  const int p = (int)(idx % HW);
  const int c = (int)((idx / HW) % C);
  const int b = (int)(idx / ((int64_t)HW * C));
  const float v = in[((int64_t)b * HW + p) * ld + c] * 0.5f + 0.5f;
  out[idx] = fminf(fmaxf(v, 0.f), 1.f);
}

}  // namespace

static int g_boundary_mfma = 1;   // tb_boundary_conv_set_variant: 1 = the matrix-core boundary convolutions (round 5), 0 = the VALU kernels
extern "C" int tb_boundary_conv_set_variant(int v) {
  const int old = g_boundary_mfma;
  g_boundary_mfma = v;
  return old;
}
#define GRID1D(n) dim3((unsigned)(((n) + 255) / 256))

extern "C" int tb_add_noise(const float* x0, const float* noise, const int64_t* timesteps, const float* alphas_cumprod, void* noisy,
                            float* velocity, int B, int64_t per_sample, tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!x0 || !noise || !timesteps || !alphas_cumprod || !noisy || B <= 0 || per_sample <= 0) return TB_EINVAL;
  const int64_t total = (int64_t)B * per_sample;
  hipLaunchKernelGGL(add_noise_kernel, GRID1D(total), dim3(256), 0, (hipStream_t)stream, x0, noise, timesteps, alphas_cumprod,
                     (f16*)noisy, velocity, per_sample, total);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int tb_timestep_embed(const int64_t* timesteps, void* out, int B, int dim, tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!timesteps || !out || B <= 0 || dim <= 0 || dim % 2) return TB_EINVAL;
  hipLaunchKernelGGL(timestep_embed_kernel, GRID1D((int64_t)B * dim), dim3(256), 0, (hipStream_t)stream, timesteps, (f16*)out, B, dim);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int tb_convin_to_nhwc(const void* in, int in_dtype, int Cin, const float* w_packed, const float* bias, void* out, int64_t ldo,
                                 int B, int H, int W, int Cout, int sign, float in_scale, tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!in || !w_packed || !out || Cout % 8 || ldo % 8 || (sign != 1 && sign != -1) || (Cin != 3 && Cin != 4)) return TB_EINVAL;
  if (((uintptr_t)w_packed) % 16) return TB_EINVAL;
  const int64_t n = (int64_t)B * H * ((W + 3) / 4) * (Cout / 8);
  hipStream_t s = (hipStream_t)stream;
  if (g_boundary_mfma && W % 16 == 0 && Cout % 32 == 0 && Cout <= 416) {   // (<= 64 KB of LDS: weights + bias)   // matrix-core form (see conv4_to_nhwc_mfma_kernel)
    const int64_t groups = (int64_t)B * H * (W / 16);
    const dim3 grid((unsigned)((groups + 4 * C4_GROUPS - 1) / (4 * C4_GROUPS)));
    const size_t lds = (size_t)Cout * CI_LDW * 2 + (size_t)Cout * 4;
#define TB_CONVIN_M(T, CI)                                                                                                                     \
  hipLaunchKernelGGL((conv4_to_nhwc_mfma_kernel<T, CI>), grid, dim3(256), lds, s, (const T*)in, w_packed, bias, (f16*)out, ldo, B, H, W, Cout, sign, \
                     in_scale)
    if (in_dtype == TB_F32) {
      if (Cin == 4) TB_CONVIN_M(float, 4);
      else TB_CONVIN_M(float, 3);
    } else {
      if (Cin == 4) TB_CONVIN_M(f16, 4);
      else TB_CONVIN_M(f16, 3);
    }
#undef TB_CONVIN_M
    TB_CHECK_LAUNCH();
    return TB_OK;
  }
#define TB_CONVIN(T, CI) \
  hipLaunchKernelGGL((conv4_to_nhwc_kernel<T, CI, 4>), GRID1D(n), dim3(256), 0, s, (const T*)in, w_packed, bias, (f16*)out, ldo, B, H, W, Cout, sign, in_scale)
  if (in_dtype == TB_F32) {
    if (Cin == 4) TB_CONVIN(float, 4);
    else TB_CONVIN(float, 3);
  } else {
    if (Cin == 4) TB_CONVIN(f16, 4);
    else TB_CONVIN(f16, 3);
  }
#undef TB_CONVIN
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int tb_conv4_to_nhwc(const void* in, int in_dtype, const float* w_packed, const float* bias, void* out, int64_t ldo, int B,
                                int H, int W, int Cout, int sign, float in_scale, tb_stream_t stream) {
  return tb_convin_to_nhwc(in, in_dtype, 4, w_packed, bias, out, ldo, B, H, W, Cout, sign, in_scale, stream);
}

extern "C" int tb_conv_to4(const void* in, int64_t ldi, const float* w_packed, const float* bias, void* out, int B, int H, int W, int C,
                           tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!in || !w_packed || !out || B <= 0) return TB_EINVAL;
  const int64_t M = (int64_t)B * H * W;
  if (g_boundary_mfma && W % 16 == 0 && C % 32 == 0 && C <= 1280 && ldi % 8 == 0 && ((uintptr_t)in) % 16 == 0 && ((uintptr_t)w_packed) % 16 == 0) {
    const int64_t groups = (int64_t)B * H * (W / 16);
    const size_t lds = (size_t)9 * C * 4 * 2;
    const dim3 grid((unsigned)((groups + 4 * C4_GROUPS - 1) / (4 * C4_GROUPS)));
    if (C == 320) {
      hipLaunchKernelGGL(conv_to4_mfma_kernel<10>, grid, dim3(256), lds, (hipStream_t)stream, (const f16*)in, ldi, w_packed, bias, (f16*)out, B, H, W, C);
    } else {
      if (lds > 48 * 1024) {
        static bool attr_done = false;
        if (!attr_done) {
          if (hipFuncSetAttribute((const void*)conv_to4_mfma_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess)
            return TB_ELAUNCH;
          attr_done = true;
        }
      }
      hipLaunchKernelGGL(conv_to4_mfma_kernel<0>, grid, dim3(256), lds, (hipStream_t)stream, (const f16*)in, ldi, w_packed, bias, (f16*)out, B, H, W, C);
    }
    TB_CHECK_LAUNCH();
    return TB_OK;
  }
  hipLaunchKernelGGL(conv_to4_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const f16*)in, ldi, w_packed,
                     bias, (f16*)out, B, H, W, C);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int tb_mse_loss(const void* pred, const float* target, float* dpred, float* loss_out, const float* loss_scale, int64_t N,
                           float* ws, tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!pred || !target || !loss_out || N <= 0) return TB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (ws && N >= 16384) {  // two-stage: up to 128 blocks of partial sums (ws >= 128 floats), then one block adds them in order
    int nb = (int)((N + 1023) / 1024);
    if (nb > 128) nb = 128;
    hipLaunchKernelGGL(mse_partial_kernel, dim3(nb), dim3(256), 0, s, (const f16*)pred, target, dpred, ws, loss_scale, N);
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, s, ws, loss_out, (int64_t)nb, 1.f / (float)N);
  } else {
    hipLaunchKernelGGL(mse_loss_kernel, dim3(1), dim3(1024), 0, s, (const f16*)pred, target, dpred, loss_out, loss_scale, N);
  }
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int tb_kpl_cos(const float* h, int64_t ldh, const void* h0, int64_t ldh0, int h0_dtype, float* dh, int64_t lddh,
                          float* partial, float* loss_out, const float* loss_scale, float weight, int64_t M, int D, tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!h || !h0 || !partial || !loss_out || M <= 0 || D <= 0) return TB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)((M + 3) / 4));
  if (h0_dtype == TB_F32)
    hipLaunchKernelGGL(kpl_cos_kernel<float>, grid, dim3(256), 0, s, h, ldh, (const float*)h0, ldh0, dh, lddh, partial, loss_scale,
                       weight, M, D);
  else
    hipLaunchKernelGGL(kpl_cos_kernel<f16>, grid, dim3(256), 0, s, h, ldh, (const f16*)h0, ldh0, dh, lddh, partial, loss_scale, weight,
                       M, D);
  hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, s, partial, loss_out, M, 1.f / (float)M);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int tb_kpl_mse(const float* h, int64_t ldh, const void* h0, int64_t ldh0, int h0_dtype, float* dh, int64_t lddh,
                          float* partial, float* loss_out, const float* loss_scale, float weight, int64_t M, int D, tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!h || !h0 || !partial || !loss_out || M <= 0 || D <= 0) return TB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)((M + 3) / 4));
  if (h0_dtype == TB_F32)
    hipLaunchKernelGGL(kpl_mse_kernel<float>, grid, dim3(256), 0, s, h, ldh, (const float*)h0, ldh0, dh, lddh, partial, loss_scale,
                       weight, M, D);
  else
    hipLaunchKernelGGL(kpl_mse_kernel<f16>, grid, dim3(256), 0, s, h, ldh, (const f16*)h0, ldh0, dh, lddh, partial, loss_scale, weight,
                       M, D);
  hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, s, partial, loss_out, M, 1.f / (float)M);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int tb_geglu_bwd(const void* dout, int64_t lddo, const void* raw, int64_t ldr, void* dproj, int64_t lddp, int64_t M,
                            int inner, tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!dout || !raw || !dproj || inner % 32 || lddo % 8 || ldr % 8 || lddp % 8) return TB_EINVAL;
  hipLaunchKernelGGL(geglu_bwd_kernel, GRID1D(M * (inner / 8)), dim3(256), 0, (hipStream_t)stream, (const f16*)dout, lddo,
                     (const f16*)raw, ldr, (f16*)dproj, lddp, M, inner);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int tb_pool2x2_sum(const void* du, int64_t ldu, void* dx, int64_t ldx, int B, int H, int W, int C, tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!du || !dx || C % 8 || ldu % 8 || ldx % 8) return TB_EINVAL;
  hipLaunchKernelGGL(pool2x2_sum_kernel, GRID1D((int64_t)B * H * W * (C / 8)), dim3(256), 0, (hipStream_t)stream, (const f16*)du, ldu,
                     (f16*)dx, ldx, B, H, W, C);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int tb_upsample2x(const void* x, int64_t ldx, void* u, int64_t ldu, int B, int H, int W, int C, tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!x || !u || C % 8 || ldu % 8 || ldx % 8) return TB_EINVAL;
  hipLaunchKernelGGL(upsample2x_kernel, GRID1D((int64_t)B * 4 * H * W * (C / 8)), dim3(256), 0, (hipStream_t)stream, (const f16*)x, ldx, (f16*)u,
                     ldu, B, H, W, C);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int tb_add_f16(const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t ldo, int64_t M, int C,
                          tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!a || !b || !out || C % 8 || lda % 8 || ldb % 8 || ldo % 8) return TB_EINVAL;
  hipLaunchKernelGGL(add_f16_kernel, GRID1D(M * (C / 8)), dim3(256), 0, (hipStream_t)stream, (const f16*)a, lda, (const f16*)b, ldb,
                     (f16*)out, ldo, M, C);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int tb_convert(const void* in, int64_t ldi, int in_dtype, void* out, int64_t ldo, int out_dtype, int64_t M, int C, float scale,
                          tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!in || !out || M <= 0 || C <= 0) return TB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid = GRID1D(M * C);
  if (in_dtype == TB_F32 && out_dtype == TB_F16)
    hipLaunchKernelGGL((convert_kernel<float, f16>), grid, dim3(256), 0, s, (const float*)in, ldi, (f16*)out, ldo, M, C, scale);
  else if (in_dtype == TB_F16 && out_dtype == TB_F32)
    hipLaunchKernelGGL((convert_kernel<f16, float>), grid, dim3(256), 0, s, (const f16*)in, ldi, (float*)out, ldo, M, C, scale);
  else if (in_dtype == TB_F32 && out_dtype == TB_F32)
    hipLaunchKernelGGL((convert_kernel<float, float>), grid, dim3(256), 0, s, (const float*)in, ldi, (float*)out, ldo, M, C, scale);
  else
    hipLaunchKernelGGL((convert_kernel<f16, f16>), grid, dim3(256), 0, s, (const f16*)in, ldi, (f16*)out, ldo, M, C, scale);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int tb_softmax_rows(const float* scores, int64_t lds, void* probs, int64_t ldp, int64_t rows, int cols, tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!scores || !probs || rows <= 0 || cols <= 0 || cols % 4 || lds % 4 || ldp % 4) return TB_EINVAL;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, scores, lds, (f16*)probs, ldp, cols);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int tb_vae_sample(const float* moments, int64_t ldm, const float* eps, float* latents, int B, int HW, int L, float scale,
                             tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!moments || !eps || !latents || B <= 0 || HW <= 0 || L <= 0 || ldm < 2 * L) return TB_EINVAL;
  hipLaunchKernelGGL(vae_sample_kernel, GRID1D((int64_t)B * L * HW), dim3(256), 0, (hipStream_t)stream, moments, ldm, eps, latents, B, HW,
                     L, scale);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int tb_chan_mix(const float* in, const float* W, const float* bias, float* out, int B, int C, int HW, float scale,
                           tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!in || !W || !out || B <= 0 || C <= 0 || C > 8 || HW <= 0) return TB_EINVAL;
  hipLaunchKernelGGL(chan_mix_kernel, GRID1D((int64_t)B * HW), dim3(256), 0, (hipStream_t)stream, in, W, bias, out, B, C, HW, scale);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int tb_dpm_step(float* x, const void* eps2, float* m_prev, void* x2, int64_t n_per_b, int B, float guidance, float alpha_t,
                           float sigma_t, float ca, float cb, float cc, tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!x || !eps2 || !m_prev || !x2 || n_per_b <= 0 || B <= 0 || alpha_t == 0.f) return TB_EINVAL;
  hipLaunchKernelGGL(dpm_step_kernel, GRID1D(n_per_b * B), dim3(256), 0, (hipStream_t)stream, x, (const f16*)eps2, m_prev, (f16*)x2, n_per_b, B,
                     guidance, alpha_t, sigma_t, ca, cb, cc);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int tb_vae_image(const float* decoded, int64_t ld, float* image, int B, int HW, int C, tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!decoded || !image || B <= 0 || HW <= 0 || C <= 0 || ld < C) return TB_EINVAL;
  hipLaunchKernelGGL(vae_image_kernel, GRID1D((int64_t)B * C * HW), dim3(256), 0, (hipStream_t)stream, decoded, ld, image, B, HW, C);
  TB_CHECK_LAUNCH();
  return TB_OK;
}
