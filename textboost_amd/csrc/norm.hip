// GroupNorm(+SiLU) and LayerNorm, forward and input-gradient, for NHWC / row-major fp16 activations on gfx950.
// All four are HBM-bound streaming kernels: 16-byte vector loads (8 halfs per lane), fp32 statistics,
// deterministic two-stage reductions (no float atomics), explicit row strides so tensors may be column slices
// of wider (concat) buffers.
//
// Replaces torch.nn.GroupNorm / F.silu / torch.nn.LayerNorm (and their autograd) that diffusers' ResnetBlock2D,
// Transformer2DModel, BasicTransformerBlock and transformers' CLIPEncoderLayer dispatch from
// train_textboost.py:1063-1067 (forward) and :1108 (backward).
#define TB_GN_INBLOCK 1  // A/B on MI355X (same box, bench.py): in-block finalize +0.6 % steps/s vs a separate finalize launch
#include "common.h"
#include "../../include/textboost_hip.h"

namespace {

constexpr int GN_MAXC = 4096;

struct GnMap {  // thread -> (column-vector, row-lane) mapping shared by the four GroupNorm kernels
  int cols, cw, nrl, col0, rl;
  bool active;
  __device__ GnMap(int C) {
    cols = C >> 3;
    cw = cols < 256 ? cols : 256;
    nrl = 256 / cw;
    int t = threadIdx.x;
    active = t < cw * nrl;
    col0 = t % cw;
    rl = t / cw;
  }
};

// Row chunks per image.  Each thread covers r rows of its column vector per chunk; r is the largest of {16, 8, 4} that still
// leaves >= 640 blocks over the batch (fatter blocks amortise the block reduction and shorten the per-block finalize loop over the
// chunk partials; A/B on one MI355X, fwd+bwd: 4096x320 77 -> 61 us, 4096x640 113 -> 94 us, 1024x1280 85 -> 61 us; fewer than
// ~600 blocks under-fills the 256 CUs and loses again)
#ifndef TB_GN_RU
#define TB_GN_RU 1   // rows per trip of the two-pass backward kernels' row loops (A/B build switch)
#endif
__host__ __device__ inline int gn_chunks(int B, int HW, int C) {
  int cols = C >> 3;
  int cw = cols < 256 ? cols : 256;
  int nrl = 256 / cw;
#ifdef TB_GN_ROWS
  int n = (HW + nrl * TB_GN_ROWS - 1) / (nrl * TB_GN_ROWS);
#else
  int n = (HW + nrl * 4 - 1) / (nrl * 4);
  for (int r = 16; r > 4; r >>= 1) {
    const int nr = (HW + nrl * r - 1) / (nrl * r);
    if ((long long)nr * (B > 0 ? B : 1) >= 640) {
      n = nr;
      break;
    }
  }
#endif
  int cap = 2048 / (B > 0 ? B : 1);
  if (cap < 1) cap = 1;
  if (n > cap) n = cap;
  if (n < 1) n = 1;
  return n;
}

// Reduce per-thread per-channel partial sums (two quantities) to per-group sums and write them to
// part[(b*nchunks + chunk)*G*2 + g*2 + {0,1}].
__device__ __forceinline__ void gn_block_group_reduce(const GnMap& mp, float (*s)[8], float (*q)[8], int C, int G,
                                                      float* lds /* 2*nrl*C floats */, float* part_out) {
  const int gs = C / G;
  float* ls = lds;
  float* lq = lds + mp.nrl * C;
  if (mp.active) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int cc = mp.col0 + k * mp.cw;
      if (cc >= mp.cols) break;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        ls[mp.rl * C + cc * 8 + e] = s[k][e];
        lq[mp.rl * C + cc * 8 + e] = q[k][e];
      }
    }
  }
  __syncthreads();
  const int t = threadIdx.x;
  if (t < 2 * G) {
    const int g = t >> 1;
    const float* src = (t & 1) ? lq : ls;
    float a = 0.f;
    for (int r = 0; r < mp.nrl; ++r)
      for (int ch = g * gs; ch < (g + 1) * gs; ++ch) a += src[r * C + ch];
    part_out[g * 2 + (t & 1)] = a;
  }
}

__global__ __launch_bounds__(256) void gn_stats_kernel(const f16* __restrict__ X, int64_t ldx, float* __restrict__ part, int HW,
                                                       int C, int G, int nchunks) {
  __shared__ float lds[2 * GN_MAXC];
  const GnMap mp(C);
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int rpc = (HW + nchunks - 1) / nchunks;
  const int r_begin = chunk * rpc, r_end = min(HW, r_begin + rpc);
  float s[2][8], q[2][8];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) s[k][e] = q[k][e] = 0.f;
  if (mp.active) {
    const f16* base = X + (int64_t)b * HW * ldx;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int cc = mp.col0 + k * mp.cw;
      if (cc >= mp.cols) break;
      for (int r = r_begin + mp.rl; r < r_end; r += mp.nrl) {
        f16x8 v = *(const f16x8*)(base + (int64_t)r * ldx + cc * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float f = (float)v[e];
          s[k][e] += f;
          q[k][e] += f * f;
        }
      }
    }
  }
  gn_block_group_reduce(mp, s, q, C, G, lds, part + ((int64_t)b * nchunks + chunk) * G * 2);
}

// per-(b, group) totals of the chunk partials, 8 lanes per group + LDS tree (fixed order -> deterministic).
// mode 0 (forward): out = (mean, rstd);  mode 1 (backward): out = (sum1 / n, sum2 / n)
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ part, float* __restrict__ out, int HW, int C, int G,
                                                          int nchunks, float eps, int mode) {
  __shared__ float ls[8][64], lq[8][64];
  const int b = blockIdx.x;
  const int g = threadIdx.x & 31, sub = threadIdx.x >> 5;  // G <= 32 handled per pass
  for (int g0 = 0; g0 < G; g0 += 32) {
    const int gg = g0 + g;
    float s = 0.f, q = 0.f;
    if (gg < G)
      for (int c = sub; c < nchunks; c += 8) {
        const float* p = part + ((int64_t)b * nchunks + c) * G * 2 + gg * 2;
        s += p[0];
        q += p[1];
      }
    ls[sub][g] = s;
    lq[sub][g] = q;
    __syncthreads();
    if (sub == 0 && gg < G) {
      float S = 0.f, Q = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        S += ls[k][g];
        Q += lq[k][g];
      }
      const float n = (float)(C / G) * (float)HW;
      float o0, o1;
      if (mode == 0) {
        const float mean = S / n;
        const float var = fmaxf(Q / n - mean * mean, 0.f);
        o0 = mean;
        o1 = rsqrtf(var + eps);
      } else {
        o0 = S / n;
        o1 = Q / n;
      }
      out[((int64_t)b * G + gg) * 2 + 0] = o0;
      out[((int64_t)b * G + gg) * 2 + 1] = o1;
    }
    __syncthreads();
  }
}

#ifdef TB_GN_INBLOCK
// In-block version of the finalize step (saves a launch): all 256 threads sum the chunk partials of this image's groups
__device__ __forceinline__ void gn_block_finalize(const float* __restrict__ part, int b, int HW, int C, int G, int nchunks, float eps,
                                                  int mode, float* o0_s, float* o1_s, float (*ls)[64], float (*lq)[64]) {
  const int g = threadIdx.x & 31, sub = threadIdx.x >> 5;
  for (int g0 = 0; g0 < G; g0 += 32) {
    const int gg = g0 + g;
    float s = 0.f, q = 0.f;
    if (gg < G)
      for (int c = sub; c < nchunks; c += 8) {
        const float* p = part + ((int64_t)b * nchunks + c) * G * 2 + gg * 2;
        s += p[0];
        q += p[1];
      }
    ls[sub][g] = s;
    lq[sub][g] = q;
    __syncthreads();
    if (sub == 0 && gg < G) {
      float S = 0.f, Q = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        S += ls[k][g];
        Q += lq[k][g];
      }
      const float n = (float)(C / G) * (float)HW;
      if (mode == 0) {
        const float mean = S / n;
        o0_s[gg] = mean;
        o1_s[gg] = rsqrtf(fmaxf(Q / n - mean * mean, 0.f) + eps);
      } else {
        o0_s[gg] = S / n;
        o1_s[gg] = Q / n;
      }
    }
    __syncthreads();
  }
}
#endif

__global__ __launch_bounds__(256) void gn_apply_kernel(const f16* __restrict__ X, int64_t ldx, f16* __restrict__ Y, int64_t ldy,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ part, float* __restrict__ stats, int HW, int C,
                                                       int G, int nchunks, float eps, int silu) {
  __shared__ float mean_s[64], rstd_s[64];
  const GnMap mp(C);
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int gs = C / G;
#ifdef TB_GN_INBLOCK
  __shared__ float ls[8][64], lq[8][64];
  gn_block_finalize(part, b, HW, C, G, nchunks, eps, 0, mean_s, rstd_s, ls, lq);
  if (chunk == 0 && threadIdx.x < G) {
    stats[((int64_t)b * G + threadIdx.x) * 2 + 0] = mean_s[threadIdx.x];
    stats[((int64_t)b * G + threadIdx.x) * 2 + 1] = rstd_s[threadIdx.x];
  }
#else
  if (threadIdx.x < G) {
    mean_s[threadIdx.x] = stats[((int64_t)b * G + threadIdx.x) * 2 + 0];
    rstd_s[threadIdx.x] = stats[((int64_t)b * G + threadIdx.x) * 2 + 1];
  }
  __syncthreads();
#endif
  if (!mp.active) return;
  const int rpc = (HW + nchunks - 1) / nchunks;
  const int r_begin = chunk * rpc, r_end = min(HW, r_begin + rpc);
  const f16* xb = X + (int64_t)b * HW * ldx;
  f16* yb = Y + (int64_t)b * HW * ldy;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int cc = mp.col0 + k * mp.cw;
    if (cc >= mp.cols) break;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int ch = cc * 8 + e;
      int g = ch / gs;
      sc[e] = rstd_s[g] * gamma[ch];
      sh[e] = beta[ch] - mean_s[g] * sc[e];
    }
    for (int r = r_begin + mp.rl; r < r_end; r += mp.nrl) {
      f16x8 v = *(const f16x8*)(xb + (int64_t)r * ldx + cc * 8);
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float z = (float)v[e] * sc[e] + sh[e];
        o[e] = (f16)(silu ? silu_f(z) : z);
      }
      *(f16x8*)(yb + (int64_t)r * ldy + cc * 8) = o;
    }
  }
}

// backward pass 1: per (b, group) sums of dyh = dz*gamma and dyh*xhat, where dz = dOut * silu'(z)
__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(const f16* __restrict__ dY, int64_t lddy, const f16* __restrict__ X,
                                                           int64_t ldx, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ stats,
                                                           float* __restrict__ part, int HW, int C, int G, int nchunks, int silu) {
  __shared__ float lds[2 * GN_MAXC];
  const GnMap mp(C);
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int gs = C / G;
  const int rpc = (HW + nchunks - 1) / nchunks;
  const int r_begin = chunk * rpc, r_end = min(HW, r_begin + rpc);
  float s[2][8], q[2][8];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) s[k][e] = q[k][e] = 0.f;
  if (mp.active) {
    const f16* xb = X + (int64_t)b * HW * ldx;
    const f16* dyb = dY + (int64_t)b * HW * lddy;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int cc = mp.col0 + k * mp.cw;
      if (cc >= mp.cols) break;
      float mu[8], rs[8], ga[8], be[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        int ch = cc * 8 + e;
        int g = ch / gs;
        mu[e] = stats[((int64_t)b * G + g) * 2];
        rs[e] = stats[((int64_t)b * G + g) * 2 + 1];
        ga[e] = gamma[ch];
        be[e] = beta[ch];
      }
      auto body = [&](const f16x8& xv, const f16x8& dv) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float xh = ((float)xv[e] - mu[e]) * rs[e];
          float d = (float)dv[e];
          if (silu) d *= silu_grad_f(xh * ga[e] + be[e]);
          d *= ga[e];
          s[k][e] += d;
          q[k][e] += d * xh;
        }
      };
      int r = r_begin + mp.rl;
#if TB_GN_RU > 1
      // TB_GN_RU rows per trip, every load of the trip issued before the first use: a thread of the rolled loop kept one row of x and dy in
      // flight (32 B), i.e. ~20 KB per CU -- at ~2 us of memory latency that alone bounds the pass near 2.8 TB/s
      for (; r + (TB_GN_RU - 1) * mp.nrl < r_end; r += TB_GN_RU * mp.nrl) {
        f16x8 xv[TB_GN_RU], dv[TB_GN_RU];
#pragma unroll
        for (int u = 0; u < TB_GN_RU; ++u) {
          xv[u] = *(const f16x8*)(xb + (int64_t)(r + u * mp.nrl) * ldx + cc * 8);
          dv[u] = *(const f16x8*)(dyb + (int64_t)(r + u * mp.nrl) * lddy + cc * 8);
        }
#pragma unroll
        for (int u = 0; u < TB_GN_RU; ++u) body(xv[u], dv[u]);
      }
#endif
      for (; r < r_end; r += mp.nrl) {
        f16x8 xv = *(const f16x8*)(xb + (int64_t)r * ldx + cc * 8);
        f16x8 dv = *(const f16x8*)(dyb + (int64_t)r * lddy + cc * 8);
        body(xv, dv);
      }
    }
  }
  gn_block_group_reduce(mp, s, q, C, G, lds, part + ((int64_t)b * nchunks + chunk) * G * 2);
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const f16* __restrict__ dY, int64_t lddy, const f16* __restrict__ X,
                                                           int64_t ldx, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ stats,
                                                           const float* __restrict__ part, const f16* __restrict__ add,
                                                           int64_t ldadd, f16* __restrict__ dX, int64_t lddx, int HW, int C, int G,
                                                           int nchunks, int silu) {
  __shared__ float s1_s[64], s2_s[64];
  const GnMap mp(C);
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int gs = C / G;
#ifdef TB_GN_INBLOCK
  __shared__ float ls[8][64], lq[8][64];
  gn_block_finalize(part, b, HW, C, G, nchunks, 0.f, 1, s1_s, s2_s, ls, lq);
#else
  if (threadIdx.x < G) {  // part = finalized per-(b, group) means of dyh and dyh*xhat
    s1_s[threadIdx.x] = part[((int64_t)b * G + threadIdx.x) * 2 + 0];
    s2_s[threadIdx.x] = part[((int64_t)b * G + threadIdx.x) * 2 + 1];
  }
  __syncthreads();
#endif
  if (!mp.active) return;
  const int rpc = (HW + nchunks - 1) / nchunks;
  const int r_begin = chunk * rpc, r_end = min(HW, r_begin + rpc);
  const f16* xb = X + (int64_t)b * HW * ldx;
  const f16* dyb = dY + (int64_t)b * HW * lddy;
  const f16* ab = add ? add + (int64_t)b * HW * ldadd : nullptr;
  f16* dxb = dX + (int64_t)b * HW * lddx;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int cc = mp.col0 + k * mp.cw;
    if (cc >= mp.cols) break;
    float mu[8], rs[8], ga[8], be[8], m1[8], m2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int ch = cc * 8 + e;
      int g = ch / gs;
      mu[e] = stats[((int64_t)b * G + g) * 2];
      rs[e] = stats[((int64_t)b * G + g) * 2 + 1];
      ga[e] = gamma[ch];
      be[e] = beta[ch];
      m1[e] = s1_s[g];
      m2[e] = s2_s[g];
    }
    auto body = [&](int r, const f16x8& xv, const f16x8& dv, const f16x8& av) {
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float xh = ((float)xv[e] - mu[e]) * rs[e];
        float d = (float)dv[e];
        if (silu) d *= silu_grad_f(xh * ga[e] + be[e]);
        d *= ga[e];
        float dx = rs[e] * (d - m1[e] - xh * m2[e]);
        if (ab) dx += (float)av[e];
        o[e] = (f16)dx;
      }
      *(f16x8*)(dxb + (int64_t)r * lddx + cc * 8) = o;
    };
    int r = r_begin + mp.rl;
#if TB_GN_RU > 1
    for (; r + (TB_GN_RU - 1) * mp.nrl < r_end; r += TB_GN_RU * mp.nrl) {   // (see gn_bwd_stats_kernel)
      f16x8 xv[TB_GN_RU], dv[TB_GN_RU], av[TB_GN_RU];
#pragma unroll
      for (int u = 0; u < TB_GN_RU; ++u) {
        xv[u] = *(const f16x8*)(xb + (int64_t)(r + u * mp.nrl) * ldx + cc * 8);
        dv[u] = *(const f16x8*)(dyb + (int64_t)(r + u * mp.nrl) * lddy + cc * 8);
        if (ab) av[u] = *(const f16x8*)(ab + (int64_t)(r + u * mp.nrl) * ldadd + cc * 8);
      }
#pragma unroll
      for (int u = 0; u < TB_GN_RU; ++u) body(r + u * mp.nrl, xv[u], dv[u], av[u]);
    }
#endif
    for (; r < r_end; r += mp.nrl) {
      f16x8 xv = *(const f16x8*)(xb + (int64_t)r * ldx + cc * 8);
      f16x8 dv = *(const f16x8*)(dyb + (int64_t)r * lddy + cc * 8);
      f16x8 av;
      if (ab) av = *(const f16x8*)(ab + (int64_t)r * ldadd + cc * 8);
      body(r, xv, dv, av);
    }
  }
}

// ------------------------------------------------------------------------------------------- GroupNorm, one workgroup per (image, group)
// Small maps with 8-aligned groups (C = 1280 / 2560 at 16x16 and 8x8: 40 / 80 channels per group): the whole (image, group) slice -- HW rows
// of C / G / 8 16-byte vectors, <= GNF_VPT per thread -- is loaded ONCE into registers, reduced in the block and normalised from the
// registers: one launch and one pass over x (and dy) instead of the statistics pass + the apply pass.  At these sizes the two-pass kernels
// are launch latency (6 - 13 us each for 1 - 5 MB tensors).
constexpr int GNF_VPT = 10;
__device__ __forceinline__ void gnf_block_sum2(float& a, float& b, float* red /* [8] */) {
  a = wave_sum(a);
  b = wave_sum(b);
  const int wave = threadIdx.x >> 6;
  __syncthreads();  // (red may still be read from a previous call)
  if ((threadIdx.x & 63) == 0) red[wave] = a, red[4 + wave] = b;
  __syncthreads();
  a = (red[0] + red[1]) + (red[2] + red[3]);
  b = (red[4] + red[5]) + (red[6] + red[7]);
}

// Workgroup -> (image, group), XCD-aware (workgroup w runs on XCD w & 7): a group's slice of a pixel row is 80 ... 320 bytes of a 128-byte-line
// tensor, so neighbouring groups share lines; with the (G, B) grid of the first version neighbours sat on DIFFERENT XCDs and every shared line
// was fetched into two L2s (1.8x the bytes from HBM for 40-channel groups of fp32 k-slices, 2.6x for the fp16 rows).  Here every XCD owns a
// contiguous run of slices (a whole image at B = 8), the lines meet in one L2.  TB_GNF_XCD=0 at build time restores the plain order (A/B).
#ifndef TB_GNF_XCD
#define TB_GNF_XCD 1
#endif
__device__ __forceinline__ void gnf_slice_of_block(int G, int& b, int& g) {
  const int n = gridDim.x, bid = blockIdx.x;
  int sl = bid;
  if (TB_GNF_XCD && (n & 7) == 0) sl = (bid & 7) * (n >> 3) + (bid >> 3);
  b = sl / G;
  g = sl - b * G;
}

// Split-K source of a fused GroupNorm (tb_groupnorm_*_splitk): the fp32 k-slices of the producing convolution and its epilogue operands.
// gnf_splitk_load8 is splitk_reduce_unit + epilogue8 of gemm.hip for act NONE / fp16 C / alpha 1, operation for operation (slice order, then
// + bias, + residual, + row bias, one rounding to fp16), so the fused launch is bit-identical to reducer + GroupNorm.
struct GnSplitK {
  const float* part;      // [S][M][npad]
  int S;
  int64_t npad, plane;    // plane = M * npad
  const float* bias;      // [C] or null
  const float* rowbias;   // [B, ldrb] or null (time-embedding projection of the image)
  int64_t ldrb;
  const f16* R;           // residual [M, ldr] or null
  int64_t ldr;
};
// All of a thread's items at once: the loads of every item of a slice (or of two slices, when few items are active) are issued before the first
// add, so a workgroup has 10-20 16-byte loads per thread in flight instead of one dependent round trip per item (the first version of this
// kernel ran exactly as long as the reducer launch it replaced).  Inactive items read item 0's address (valid memory, result ignored): no
// branches around the loads.
template <int VPT>
__device__ __forceinline__ void gnf_splitk_load_all(const GnSplitK& k, int b, int HW, int g, int gs, int cv, int items, f16x8 (&out)[GNF_VPT]) {
  float acc[VPT][8];
  int64_t off[VPT], roff[VPT];
  int col[VPT];
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int it = threadIdx.x + i * 256;
    const int itc = it < items ? it : 0;
    const int r = itc / cv, c = itc - r * cv;
    col[i] = g * gs + c * 8;
    off[i] = ((int64_t)b * HW + r) * k.npad + col[i];
#ifdef TB_GNF_FAKE  // timing experiment only (wrong results): the slices as if laid out [b][g][HW][gs], contiguous per workgroup
    off[i] = ((int64_t)b * (k.npad / gs) + g) * HW * gs + (int64_t)itc * 8;
#endif
    roff[i] = ((int64_t)b * HW + r) * k.ldr + col[i];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[i][e] = 0.f;
  }
  f16x8 rv[VPT];
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    if (k.R) rv[i] = *(const f16x8*)(k.R + roff[i]);
    else rv[i] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }
  constexpr int CH = VPT <= 2 ? 8 : (VPT <= 3 ? 6 : (VPT <= 5 ? 4 : 1));  // slices per round: <= 160 registers of loads in flight
  int s = 0;
  for (; s + CH <= k.S; s += CH) {
    f32x4 x[VPT][CH][2];
#pragma unroll
    for (int j = 0; j < CH; ++j)
#pragma unroll
      for (int i = 0; i < VPT; ++i) {
        const float* src = k.part + off[i] + (s + j) * k.plane;
        x[i][j][0] = *(const f32x4*)src;
        x[i][j][1] = *(const f32x4*)(src + 4);
      }
#pragma unroll
    for (int j = 0; j < CH; ++j)  // slice order
#pragma unroll
      for (int i = 0; i < VPT; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[i][e] += x[i][j][0][e];
          acc[i][4 + e] += x[i][j][1][e];
        }
  }
  for (; s < k.S; ++s) {
    f32x4 x[VPT][2];
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const float* src = k.part + off[i] + s * k.plane;
      x[i][0] = *(const f32x4*)src;
      x[i][1] = *(const f32x4*)(src + 4);
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[i][e] += x[i][0][e];
        acc[i][4 + e] += x[i][1][e];
      }
  }
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = acc[i][e] * 1.f + (k.bias ? k.bias[col[i] + e] : 0.f) + (float)rv[i][e];
      if (k.rowbias) t += k.rowbias[(int64_t)b * k.ldrb + col[i] + e];
      out[i][e] = (f16)t;
    }
  }
}

template <bool SPLITK>
__global__ __launch_bounds__(256) void gn_fused_fwd_kernel(const f16* __restrict__ X, int64_t ldx, f16* __restrict__ Y, int64_t ldy,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ stats, int HW, int C, int G, float eps, int silu,
                                                           const GnSplitK sk) {
  __shared__ float red[8];
  int b, g;
  gnf_slice_of_block(G, b, g);
  const int gs = C / G, cv = gs >> 3;
  const int items = HW * cv;
  const f16* xb = X + (int64_t)b * HW * ldx + g * gs;
  f16* yb = Y + (int64_t)b * HW * ldy + g * gs;
  f16x8 v[GNF_VPT];
  float s = 0.f, q = 0.f;
  if (SPLITK) {  // x = the convolution's output, formed here from its k-slices and written for the backward / the skip connections
    if (items <= 2 * 256) gnf_splitk_load_all<2>(sk, b, HW, g, gs, cv, items, v);
    else if (items <= 3 * 256) gnf_splitk_load_all<3>(sk, b, HW, g, gs, cv, items, v);
    else if (items <= 5 * 256) gnf_splitk_load_all<5>(sk, b, HW, g, gs, cv, items, v);
    else gnf_splitk_load_all<GNF_VPT>(sk, b, HW, g, gs, cv, items, v);
#pragma unroll
    for (int i = 0; i < GNF_VPT; ++i) {
      const int it = threadIdx.x + i * 256;
      if (it < items) {
        const int r = it / cv, c = it - r * cv;
        *(f16x8*)(const_cast<f16*>(xb) + (int64_t)r * ldx + c * 8) = v[i];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < GNF_VPT; ++i) {
    const int it = threadIdx.x + i * 256;
    if (it < items) {
      const int r = it / cv, c = it - r * cv;
      if (!SPLITK) v[i] = *(const f16x8*)(xb + (int64_t)r * ldx + c * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float f = (float)v[i][e];
        s += f;
        q += f * f;
      }
    }
  }
  gnf_block_sum2(s, q, red);
  const float n = (float)gs * (float)HW;
  const float mean = s / n;
  const float rstd = rsqrtf(fmaxf(q / n - mean * mean, 0.f) + eps);
  if (threadIdx.x == 0) {
    stats[((int64_t)b * G + g) * 2 + 0] = mean;
    stats[((int64_t)b * G + g) * 2 + 1] = rstd;
  }
#pragma unroll
  for (int i = 0; i < GNF_VPT; ++i) {
    const int it = threadIdx.x + i * 256;
    if (it < items) {
      const int r = it / cv, c = it - r * cv;
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ch = g * gs + c * 8 + e;
        const float sc = rstd * gamma[ch];
        const float z = (float)v[i][e] * sc + (beta[ch] - mean * sc);
        o[e] = (f16)(silu ? silu_f(z) : z);
      }
      *(f16x8*)(yb + (int64_t)r * ldy + c * 8) = o;
    }
  }
}

template <bool SPLITK>
__global__ __launch_bounds__(256) void gn_fused_bwd_kernel(const f16* __restrict__ dY, int64_t lddy, const f16* __restrict__ X, int64_t ldx,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ stats, const f16* __restrict__ add, int64_t ldadd,
                                                           f16* __restrict__ dX, int64_t lddx, int HW, int C, int G, int silu,
                                                           const GnSplitK sk) {
  __shared__ float red[8];
  int b, g;
  gnf_slice_of_block(G, b, g);
  const int gs = C / G, cv = gs >> 3;
  const int items = HW * cv;
  const f16* xb = X + (int64_t)b * HW * ldx + g * gs;
  const f16* dyb = SPLITK ? nullptr : dY + (int64_t)b * HW * lddy + g * gs;
  const f16* ab = add ? add + (int64_t)b * HW * ldadd + g * gs : nullptr;
  f16* dxb = dX + (int64_t)b * HW * lddx + g * gs;
  const float mu = stats[((int64_t)b * G + g) * 2], rs = stats[((int64_t)b * G + g) * 2 + 1];
  f16x8 xh[GNF_VPT], dh[GNF_VPT];  // the slice's x and dy rows (both passes below work from these registers)
  float s1 = 0.f, s2 = 0.f;
  if (SPLITK) {  // dy = the dgrad convolution's k-slices: never stored
    if (items <= 2 * 256) gnf_splitk_load_all<2>(sk, b, HW, g, gs, cv, items, dh);
    else if (items <= 3 * 256) gnf_splitk_load_all<3>(sk, b, HW, g, gs, cv, items, dh);
    else if (items <= 5 * 256) gnf_splitk_load_all<5>(sk, b, HW, g, gs, cv, items, dh);
    else gnf_splitk_load_all<GNF_VPT>(sk, b, HW, g, gs, cv, items, dh);
  }
#pragma unroll
  for (int i = 0; i < GNF_VPT; ++i) {
    const int it = threadIdx.x + i * 256;
    if (it < items) {
      const int r = it / cv, c = it - r * cv;
      xh[i] = *(const f16x8*)(xb + (int64_t)r * ldx + c * 8);
      if (!SPLITK) dh[i] = *(const f16x8*)(dyb + (int64_t)r * lddy + c * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ch = g * gs + c * 8 + e;
        const float ga = gamma[ch];
        const float x_ = ((float)xh[i][e] - mu) * rs;
        float d = (float)dh[i][e];
        if (silu) d *= silu_grad_f(x_ * ga + beta[ch]);
        d *= ga;
        s1 += d;
        s2 += d * x_;
      }
    }
  }
  gnf_block_sum2(s1, s2, red);
  const float n = (float)gs * (float)HW;
  const float m1 = s1 / n, m2 = s2 / n;
#pragma unroll
  for (int i = 0; i < GNF_VPT; ++i) {
    const int it = threadIdx.x + i * 256;
    if (it < items) {
      const int r = it / cv, c = it - r * cv;
      f16x8 av;
      if (ab) av = *(const f16x8*)(ab + (int64_t)r * ldadd + c * 8);
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ch = g * gs + c * 8 + e;
        const float ga = gamma[ch];
        const float x_ = ((float)xh[i][e] - mu) * rs;
        float d = (float)dh[i][e];
        if (silu) d *= silu_grad_f(x_ * ga + beta[ch]);
        d *= ga;
        float dx = rs * (d - m1 - x_ * m2);
        if (ab) dx += (float)av[e];
        o[e] = (f16)dx;
      }
      *(f16x8*)(dxb + (int64_t)r * lddx + c * 8) = o;
    }
  }
}
// ------------------------------------------------------------------------------------------- GroupNorm, one pass, any even group width (round 4)
// The large maps (64x64 / 32x32, groups of 10 .. 60 channels) ran the two-pass kernels above: a statistics launch that reads x (and dy), then an
// apply launch that reads them AGAIN -- at 10-35 us per launch these passes were 2.2 ms of the step at 2.4-2.9 TB/s.  Here one 1024-thread
// workgroup owns an (image, group) slice at DWORD granularity (two channels per item; a pixel's slice is gs / 2 consecutive dwords, 4-byte
// aligned for any even gs), holds it in registers (<= NV dwords per thread), reduces in the block and normalises from the registers: x (and dy)
// are read once and there is one launch.  A wave's 64 consecutive items cover 64 / (gs / 2) pixels, i.e. a handful of 128-byte lines per load
// instruction instead of two -- paid in the CU's L1, which has nothing else to do here.  The lines of a pixel row are shared by the 32 groups of
// the image: the block -> slice map gives every XCD a contiguous run of slices (a whole image at B = 8), so they meet in one L2 and HBM sees the
// tensor once.
constexpr int GNS_T = 1024, GNS_W = GNS_T / 64, GNS_MAXGS = 128;
__device__ __forceinline__ void gns_block_sum2(float& a, float& b, float* red /* [2 * GNS_W] */) {
  a = wave_sum(a);
  b = wave_sum(b);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[wave] = a, red[GNS_W + wave] = b;
  __syncthreads();
  float x = 0.f, y = 0.f;
#pragma unroll
  for (int w = 0; w < GNS_W; ++w) x += red[w], y += red[GNS_W + w];
  a = x, b = y;
}
__device__ __forceinline__ void gns_slice_of_block(int B, int G, int& b, int& g) {  // XCD-aware: workgroup w runs on XCD w & 7
  const int n = B * G, bid = blockIdx.x;
  int sl = bid;
  if ((n & 7) == 0) sl = (bid & 7) * (n >> 3) + (bid >> 3);
  b = sl / G;
  g = sl - b * G;
}
// Thread -> items: with dpp = gs / 2 dwords per pixel only the first T' = (1024 / dpp) * dpp threads work; thread t owns dword column j = t % dpp
// (so its two channels -- gamma, beta -- are loop constants) of the pixels p0 + k * PS, p0 = t / dpp, PS = T' / dpp: consecutive threads read
// consecutive dwords of a pixel's slice, then the next pixel's; the k-th access is a constant stride behind the (k-1)-th.  No per-item index
// state survives between the passes (the first version, a flat item walk, kept 2 NV index registers alive and spilled kilobytes per lane).
struct GnsMap {
  int j, p0, ps;
  bool active;
  __device__ GnsMap(int dpp) {
    const int per = GNS_T / dpp;  // pixels per sweep
    ps = per;
    p0 = threadIdx.x / dpp;
    j = threadIdx.x - p0 * dpp;
    active = p0 < per;
  }
};
__host__ __device__ inline int gns_items_per_thread(int HW, int gs) {
  const int per = GNS_T / (gs >> 1);
  return (HW + per - 1) / per;
}
__device__ __forceinline__ float2 gns_unpack(uint32_t v) {
  const f16x2 h = __builtin_bit_cast(f16x2, v);
  return make_float2((float)h[0], (float)h[1]);
}
__device__ __forceinline__ uint32_t gns_pack(float a, float b) {
  f16x2 h;
  h[0] = (f16)a, h[1] = (f16)b;
  return __builtin_bit_cast(uint32_t, h);
}

template <int NV>
__global__ __launch_bounds__(GNS_T) void gn_slice_fwd_kernel(const f16* __restrict__ X, int64_t ldx, f16* __restrict__ Y, int64_t ldy,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float* __restrict__ stats, int B, int HW, int C, int G, float eps, int silu) {
  __shared__ float red[2 * GNS_W];
  int b, g;
  gns_slice_of_block(B, G, b, g);
  const int gs = C / G;
  const GnsMap mp(gs >> 1);
  const uint32_t* xb = reinterpret_cast<const uint32_t*>(X + (int64_t)b * HW * ldx + g * gs) + mp.j;
  uint32_t* yb = reinterpret_cast<uint32_t*>(Y + (int64_t)b * HW * ldy + g * gs) + mp.j;
  const uint32_t ldxw = (uint32_t)(ldx >> 1), ldyw = (uint32_t)(ldy >> 1);
  uint32_t v[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int p = mp.p0 + k * mp.ps;
    v[k] = xb[(mp.active && p < HW) ? (uint32_t)p * ldxw : 0u];  // (clamped: no branch around the load; ignored below)
  }
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    if (mp.active && mp.p0 + k * mp.ps < HW) {
      const float2 f = gns_unpack(v[k]);
      s += f.x + f.y;
      q += f.x * f.x + f.y * f.y;
    }
  }
  gns_block_sum2(s, q, red);
  const float n = (float)gs * (float)HW;
  const float mean = s / n;
  const float rstd = rsqrtf(fmaxf(q / n - mean * mean, 0.f) + eps);
  if (threadIdx.x == 0) {
    stats[((int64_t)b * G + g) * 2 + 0] = mean;
    stats[((int64_t)b * G + g) * 2 + 1] = rstd;
  }
  if (!mp.active) return;
  const int ch = g * gs + 2 * mp.j;
  const float sc0 = rstd * gamma[ch], sc1 = rstd * gamma[ch + 1];
  const float of0 = beta[ch] - mean * sc0, of1 = beta[ch + 1] - mean * sc1;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int p = mp.p0 + k * mp.ps;
    if (p < HW) {
      const float2 f = gns_unpack(v[k]);
      float z0 = f.x * sc0 + of0, z1 = f.y * sc1 + of1;
      if (silu) z0 = silu_f(z0), z1 = silu_f(z1);
      yb[(uint32_t)p * ldyw] = gns_pack(z0, z1);
    }
  }
}

// KEEP = true would hold d = dy silu'(z) gamma in registers between the two passes; hipcc then spills kilobytes per lane (it hoists the second
// pass's arithmetic above the block reduction), so the launches use KEEP = false: d is recomputed from the raw x / dy dwords, as the two-pass
// kernels did
template <int NV, bool KEEP>
__global__ __launch_bounds__(GNS_T) void gn_slice_bwd_kernel(const f16* __restrict__ dY, int64_t lddy, const f16* __restrict__ X, int64_t ldx,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const float* __restrict__ stats, const f16* __restrict__ add, int64_t ldadd,
                                                               f16* __restrict__ dX, int64_t lddx, int B, int HW, int C, int G, int silu) {
  __shared__ float red[2 * GNS_W];
  int b, g;
  gns_slice_of_block(B, G, b, g);
  const int gs = C / G;
  const GnsMap mp(gs >> 1);
  const uint32_t* xb = reinterpret_cast<const uint32_t*>(X + (int64_t)b * HW * ldx + g * gs) + mp.j;
  const uint32_t* dyb = reinterpret_cast<const uint32_t*>(dY + (int64_t)b * HW * lddy + g * gs) + mp.j;
  const uint32_t* ab = add ? reinterpret_cast<const uint32_t*>(add + (int64_t)b * HW * ldadd + g * gs) + mp.j : nullptr;
  uint32_t* dxb = reinterpret_cast<uint32_t*>(dX + (int64_t)b * HW * lddx + g * gs) + mp.j;
  const uint32_t ldxw = (uint32_t)(ldx >> 1), lddyw = (uint32_t)(lddy >> 1), ldaw = (uint32_t)(ldadd >> 1), lddxw = (uint32_t)(lddx >> 1);
  const float mu = stats[((int64_t)b * G + g) * 2], rs = stats[((int64_t)b * G + g) * 2 + 1];
  uint32_t xv[NV], dv[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int p = mp.p0 + k * mp.ps;
    const bool ok = mp.active && p < HW;
    xv[k] = xb[ok ? (uint32_t)p * ldxw : 0u];
    dv[k] = dyb[ok ? (uint32_t)p * lddyw : 0u];
  }
  const int ch = g * gs + 2 * (mp.active ? mp.j : 0);
  const float g0 = gamma[ch], g1 = gamma[ch + 1], b0 = beta[ch], b1 = beta[ch + 1];
  float d0[KEEP ? NV : 1], d1[KEEP ? NV : 1];
  float s1 = 0.f, s2 = 0.f;
  auto grad_of = [&](int k, float& h0, float& h1, float& a0, float& a1) {
    const float2 x = gns_unpack(xv[k]), dy = gns_unpack(dv[k]);
    h0 = (x.x - mu) * rs, h1 = (x.y - mu) * rs;
    a0 = dy.x, a1 = dy.y;
    if (silu) a0 *= silu_grad_f(h0 * g0 + b0), a1 *= silu_grad_f(h1 * g1 + b1);
    a0 *= g0, a1 *= g1;
  };
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    if (KEEP) d0[k] = d1[k] = 0.f;
    if (mp.active && mp.p0 + k * mp.ps < HW) {
      float h0, h1, a0, a1;
      grad_of(k, h0, h1, a0, a1);
      if (KEEP) d0[k] = a0, d1[k] = a1;
      s1 += a0 + a1;
      s2 += a0 * h0 + a1 * h1;
    }
  }
  gns_block_sum2(s1, s2, red);
  if (!mp.active) return;
  const float n = (float)gs * (float)HW;
  const float m1 = s1 / n, m2 = s2 / n;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int p = mp.p0 + k * mp.ps;
    if (p < HW) {
      float h0, h1, a0, a1;
      if (KEEP) {
        const float2 x = gns_unpack(xv[k]);
        h0 = (x.x - mu) * rs, h1 = (x.y - mu) * rs;
        a0 = d0[k], a1 = d1[k];
      } else {
        grad_of(k, h0, h1, a0, a1);
      }
      float o0 = rs * (a0 - m1 - h0 * m2), o1 = rs * (a1 - m1 - h1 * m2);
      if (ab) {
        const float2 a = gns_unpack(ab[(uint32_t)p * ldaw]);
        o0 += a.x, o1 += a.y;
      }
      dxb[(uint32_t)p * lddxw] = gns_pack(o0, o1);
    }
  }
}
// NV (dwords per thread) the slice kernels would need, 0 when they do not apply: even groups of <= 128 channels, strides even, enough slices
// to occupy the chip, and the slice must fit the registers of 1024 threads (forward <= 44, backward <= 22 dwords per thread: x and dy live)
int g_gn_fused = 11;  // tb_groupnorm_set_variant bits: 1 = one-pass 256-thread kernels for the small maps with 8-aligned groups, 2 = one-pass slice
                     // kernels for the large maps (round 4), 4 = slice backward at 22 dwords per thread (opt-in), 8 = the hoisted-load LayerNorm + LoRA-down
                     // kernel (ln_fwd_kernel<.., 12>); 0 = always the two-pass kernels
inline int gn_slice_nv(int B, int HW, int C, int G, bool bwd) {
  if (!(g_gn_fused & 2) || G <= 0 || C % G) return 0;
  const int gs = C / G;
  if ((gs & 1) || gs > GNS_MAXGS || (int64_t)B * G < 128) return 0;
  // measured on one MI355X (scratch/gn_slice_time.py, B = 8; two-pass -> one-pass us): forward 4096x320 24.8 -> 20.9, 4096x640 34.7 -> 26.6,
  // 1024x640 18.3 -> 10.1, 1024x1280 24.8 -> 15.7, 1024x1920 28.8 -> 21.0, 256x1920 15.4 -> 7.9; backward 4096x320 39.3 -> 35.3, 1024x640 25.1 -> 17.2,
  // 1024x960 29.8 -> 23.7, 256x640 16.0 -> 8.3 -- but 44 dwords of x AND dy per thread (4096x640 backward) spill and ran 66 -> 101 us, and 64
  // (4096x960 forward) 48 -> 60: those shapes keep the two-pass kernels
  const int nv = gns_items_per_thread(HW, gs);
  if (nv <= 11) return 11;
  if (bwd && !(g_gn_fused & 4)) return 0;   // backward with 22 dwords of x and dy per thread (4096x320, 1024x1280): 39 -> 35 us on one box, 39 -> 45 on
                                            // another; opt-in (bit 4) until the step says otherwise
  if (nv <= 22) return 22;
  if (bwd) return 0;   // (33 dwords of x and of dy per thread: 1024x1920 backward 50.5 -> 60.1 us)
  if (nv <= 33) return 33;
  if (nv <= 44) return 44;
  return 0;
}

inline bool gn_fused_ok(int B, int HW, int C, int G) {
  const int gs = C / G;
  return (g_gn_fused & 1) && gs % 8 == 0 && (int64_t)HW * (gs / 8) <= 256 * GNF_VPT && (int64_t)B * G >= 128;
}

// ------------------------------------------------------------------------------------------- LayerNorm
constexpr int LN_MAXV = 3;  // vectors of 8 per lane -> C <= 1536

template <typename T>
__device__ __forceinline__ void load8(const T* p, float* out);
template <>
__device__ __forceinline__ void load8<f16>(const f16* p, float* out) {
  f16x8 v = *(const f16x8*)p;
#pragma unroll
  for (int e = 0; e < 8; ++e) out[e] = (float)v[e];
}
template <>
__device__ __forceinline__ void load8<float>(const float* p, float* out) {
  f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    out[e] = a[e];
    out[4 + e] = b[e];
  }
}
template <typename T>
__device__ __forceinline__ void store8(T* p, const float* v);
template <>
__device__ __forceinline__ void store8<f16>(f16* p, const float* v) {
  f16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (f16)v[e];
  *(f16x8*)p = o;
}
template <>
__device__ __forceinline__ void store8<float>(float* p, const float* v) {
  f32x4 a, b;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    a[e] = v[e];
    b[e] = v[4 + e];
  }
  *(f32x4*)p = a;
  *(f32x4*)(p + 4) = b;
}

// RF = 0: the generic kernel.  RF = 12 (R <= 12 adapter rows, C <= 1024: CLIP-L's q / k / v adapters at r = 4): every adapter-row load of the
// fused down projection is issued at the top, next to the row's own loads, so the LayerNorm statistics run under their latency (one wave per
// row and two waves per SIMD: nothing else hides it), and the 12 cross-lane reductions are one halving butterfly (17 shuffles, not 72).
template <typename T, typename TO, int RF = 0>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ X, int64_t ldx, TO* __restrict__ Y, int64_t ldy,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ stats, int64_t M, int C, float eps,
                                                     const float* __restrict__ loraA, int R, f16* __restrict__ tdown, int64_t ldt, int64_t lora_rows) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nv = C >> 3;
  float v[LN_MAXV][8];
  float s = 0.f;
  constexpr int RFN = RF ? RF : 1, RFK = 2;   // (RF: nv <= 128 -- two vectors of 8 per lane)
  f32x4 fa0[RFN][RFK], fa1[RFN][RFK];
  const bool fast_lora = RF && loraA && row < lora_rows;
  if (RF) {
#pragma unroll
    for (int k = 0; k < RFK; ++k) {
      const int vi = lane + 64 * k;
#pragma unroll
      for (int j = 0; j < RFN; ++j) {
        fa0[j][k] = f32x4{0.f, 0.f, 0.f, 0.f};
        fa1[j][k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (fast_lora && vi < nv) {
          const float* ap = loraA + (int64_t)(j < R ? j : R - 1) * C + vi * 8;   // (clamped duplicate, not stored)
          fa0[j][k] = *(const f32x4*)ap;
          fa1[j][k] = *(const f32x4*)(ap + 4);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < LN_MAXV; ++k) {
    int vi = lane + 64 * k;
    if (vi < nv) {
      load8<T>(X + row * ldx + vi * 8, v[k]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[k][e];
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAXV; ++k) {
    int vi = lane + 64 * k;
    if (vi < nv) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float d = v[k][e] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
  if (lane == 0 && stats) {
    stats[row * 2] = mean;
    stats[row * 2 + 1] = rstd;
  }
#pragma unroll
  for (int k = 0; k < LN_MAXV; ++k) {
    int vi = lane + 64 * k;
    if (vi < nv) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (v[k][e] - mean) * rstd * gamma[vi * 8 + e] + beta[vi * 8 + e];
      store8<TO>(Y + row * ldy + vi * 8, o);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[k][e] = (float)(TO)o[e];  // what a consumer of Y reads
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[k][e] = 0.f;
    }
  }
  if (RF && fast_lora) {
    // per-lane partial sums in the generic path's order (k ascending, the two halves of a vector pairwise), then the halving butterfly: at
    // distance o a lane keeps the half of its values its bit selects and hands the other half over -- the same pairs are added at every level
    // as in the 6-step butterfly of each value on its own, so the sums are the same bit for bit; value j ends in the lanes with (lane >> 2) == j
    float a[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = 0.f;
#pragma unroll
    for (int j = 0; j < RFN; ++j)
#pragma unroll
      for (int k = 0; k < RFK; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) a[j] += v[k][e] * (float)(f16)fa0[j][k][e] + v[k][4 + e] * (float)(f16)fa1[j][k][e];
#define TB_HALVE(N_, O_)                                                                   \
  {                                                                                        \
    const bool up = lane & (O_);                                                           \
    _Pragma("unroll") for (int i = 0; i < (N_); ++i) {                                     \
      const float send = up ? a[i] : a[i + (N_)], keep = up ? a[i + (N_)] : a[i];          \
      a[i] = keep + __shfl_xor(send, (O_), 64);                                            \
    }                                                                                      \
  }
    TB_HALVE(8, 32)
    TB_HALVE(4, 16)
    TB_HALVE(2, 8)
    TB_HALVE(1, 4)
#undef TB_HALVE
    a[0] += __shfl_xor(a[0], 2, 64);
    a[0] += __shfl_xor(a[0], 1, 64);
    const int j = lane >> 2;
    if ((lane & 3) == 0 && j < R) tdown[row * ldt + j] = (f16)a[0];
  } else if (loraA && row < lora_rows) {  // (rows >= lora_rows: a frozen batch riding along -- their rows of tdown are zeroed below)
    // fused LoRA down projection (lora_A of peft lora.Linear on the normalised row): tdown[row, j] = sum_k y[row,k] * fp16(A[j,k]);
    // the row is still in registers, the R <= 24 adapter rows are L2 resident
    // four adapter rows at a time: their loads and their butterfly reductions are independent, so the load latencies and the six dependent
    // cross-lane steps of a reduction are paid once per FOUR rows (one row at a time, the 12 serial [load -> 6 shuffles] chains of r = 4 on q/k/v
    // made this launch 17.4 us against 5.4 us for the plain LayerNorm of the same rows).  Per-row arithmetic and summation order are unchanged.
    for (int j0 = 0; j0 < R; j0 += 4) {
      float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j = j0 + q < R ? j0 + q : R - 1;  // (clamped duplicate, not stored)
#pragma unroll
        for (int k = 0; k < LN_MAXV; ++k) {
          int vi = lane + 64 * k;
          if (vi < nv) {
            const f32x4 a0 = *(const f32x4*)(loraA + (int64_t)j * C + vi * 8), a1 = *(const f32x4*)(loraA + (int64_t)j * C + vi * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) a[q] += v[k][e] * (float)(f16)a0[e] + v[k][4 + e] * (float)(f16)a1[e];
          }
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] += __shfl_xor(a[q], o, 64);
      }
      if (lane < 4 && j0 + lane < R) {
        const float r = lane == 0 ? a[0] : lane == 1 ? a[1] : lane == 2 ? a[2] : a[3];
        tdown[row * ldt + j0 + lane] = (f16)r;
      }
    }
  } else if (loraA && lane < R) {
    // a frozen row riding along (KPL teacher): its K-extension operand must be zero whatever an earlier call left in the buffer
    tdown[row * ldt + lane] = (f16)0.f;
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)) [+ add],  g = dy * gamma.   T = dtype of x / add / dx.
template <typename T, typename TDY>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const TDY* __restrict__ dY, int64_t lddy, const T* __restrict__ X, int64_t ldx,
                                                     const float* __restrict__ gamma, const float* __restrict__ stats,
                                                     const T* __restrict__ add, int64_t ldadd, T* __restrict__ dX, int64_t lddx,
                                                     f16* __restrict__ dX16, int64_t lddx16, int64_t M, int C) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nv = C >> 3;
  const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
  float g[LN_MAXV][8], xh[LN_MAXV][8];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAXV; ++k) {
    int vi = lane + 64 * k;
    if (vi < nv) {
      float xv[8], dv[8];
      load8<T>(X + row * ldx + vi * 8, xv);
      load8<TDY>(dY + row * lddy + vi * 8, dv);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        xh[k][e] = (xv[e] - mean) * rstd;
        g[k][e] = dv[e] * gamma[vi * 8 + e];
        s1 += g[k][e];
        s2 += g[k][e] * xh[k][e];
      }
    }
  }
  s1 = wave_sum(s1) / (float)C;
  s2 = wave_sum(s2) / (float)C;
#pragma unroll
  for (int k = 0; k < LN_MAXV; ++k) {
    int vi = lane + 64 * k;
    if (vi < nv) {
      float o[8];
      if (add) load8<T>(add + row * ldadd + vi * 8, o);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += rstd * (g[k][e] - s1 - xh[k][e] * s2);
      store8<T>(dX + row * lddx + vi * 8, o);
      if (dX16) store8<f16>(dX16 + row * lddx16 + vi * 8, o);  // the fp16 operand of the next dgrad GEMM
    }
  }
}

}  // namespace

extern "C" int64_t tb_groupnorm_ws_floats(int B, int HW, int C, int G) {
  return (int64_t)B * gn_chunks(B, HW, C) * G * 2 + (int64_t)B * G * 2;
}

extern "C" int tb_groupnorm_fwd(const void* x, int64_t ldx, void* y, int64_t ldy, const float* gamma, const float* beta,
                                float* stats, float* ws, int B, int HW, int C, int G, float eps, int silu, tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!x || !y || !gamma || !beta || !stats || !ws) return TB_EINVAL;
  if (C % 8 || C > GN_MAXC || G <= 0 || G > 64 || C % G || ldx % 8 || ldy % 8 || B <= 0 || HW <= 0) return TB_EINVAL;
  const int nch = gn_chunks(B, HW, C);
  hipStream_t s = (hipStream_t)stream;
  if (gn_fused_ok(B, HW, C, G)) {
    hipLaunchKernelGGL(gn_fused_fwd_kernel<false>, dim3(G * B), dim3(256), 0, s, (const f16*)x, ldx, (f16*)y, ldy, gamma, beta, stats, HW, C, G,
                       eps, silu, GnSplitK{});
    TB_CHECK_LAUNCH();
    return TB_OK;
  }
  if (const int nv = gn_slice_nv(B, HW, C, G, false)) {
#define TB_GNS_FWD(NV_)                                                                                                                       \
  hipLaunchKernelGGL(gn_slice_fwd_kernel<NV_>, dim3(B * G), dim3(GNS_T), 0, s, (const f16*)x, ldx, (f16*)y, ldy, gamma, beta, stats, B, HW, C, G, \
                     eps, silu)
    if (nv == 11) TB_GNS_FWD(11);
    else if (nv == 22) TB_GNS_FWD(22);
    else if (nv == 33) TB_GNS_FWD(33);
    else TB_GNS_FWD(44);
#undef TB_GNS_FWD
    TB_CHECK_LAUNCH();
    return TB_OK;
  }
  hipLaunchKernelGGL(gn_stats_kernel, dim3(nch, B), dim3(256), 0, s, (const f16*)x, ldx, ws, HW, C, G, nch);
#ifndef TB_GN_INBLOCK
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(256), 0, s, ws, stats, HW, C, G, nch, eps, 0);
#endif
  hipLaunchKernelGGL(gn_apply_kernel, dim3(nch, B), dim3(256), 0, s, (const f16*)x, ldx, (f16*)y, ldy, gamma, beta, ws, stats, HW, C,
                     G, nch, eps, silu);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int tb_groupnorm_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* gamma, const float* beta,
                                const float* stats, const void* add, int64_t ldadd, void* dx, int64_t lddx, float* ws, int B, int HW,
                                int C, int G, int silu, tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!dy || !x || !gamma || !beta || !stats || !dx || !ws) return TB_EINVAL;
  if (C % 8 || C > GN_MAXC || G <= 0 || G > 64 || C % G || ldx % 8 || lddy % 8 || lddx % 8 || (add && ldadd % 8)) return TB_EINVAL;
  const int nch = gn_chunks(B, HW, C);
  hipStream_t s = (hipStream_t)stream;
  if (gn_fused_ok(B, HW, C, G)) {
    hipLaunchKernelGGL(gn_fused_bwd_kernel<false>, dim3(G * B), dim3(256), 0, s, (const f16*)dy, lddy, (const f16*)x, ldx, gamma, beta, stats,
                       (const f16*)add, ldadd, (f16*)dx, lddx, HW, C, G, silu, GnSplitK{});
    TB_CHECK_LAUNCH();
    return TB_OK;
  }
  if (const int nv = gn_slice_nv(B, HW, C, G, true)) {
#define TB_GNS_BWD(NV_)                                                                                                                        \
  hipLaunchKernelGGL((gn_slice_bwd_kernel<NV_, false>), dim3(B * G), dim3(GNS_T), 0, s, (const f16*)dy, lddy, (const f16*)x, ldx, gamma, beta, stats,      \
                     (const f16*)add, ldadd, (f16*)dx, lddx, B, HW, C, G, silu)
    if (nv == 11) TB_GNS_BWD(11);
    else TB_GNS_BWD(22);
#undef TB_GNS_BWD
    TB_CHECK_LAUNCH();
    return TB_OK;
  }
  hipLaunchKernelGGL(gn_bwd_stats_kernel, dim3(nch, B), dim3(256), 0, s, (const f16*)dy, lddy, (const f16*)x, ldx, gamma, beta, stats,
                     ws, HW, C, G, nch, silu);
#ifdef TB_GN_INBLOCK
  float* fin = ws;
#else
  float* fin = ws + (int64_t)B * nch * G * 2;  // finalized sums live behind the partials
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(256), 0, s, ws, fin, HW, C, G, nch, 0.f, 1);
#endif
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(nch, B), dim3(256), 0, s, (const f16*)dy, lddy, (const f16*)x, ldx, gamma, beta, stats,
                     fin, (const f16*)add, ldadd, (f16*)dx, lddx, HW, C, G, nch, silu);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int tb_groupnorm_splitk_ok(int B, int HW, int C, int G) {
  return (G > 0 && C % 8 == 0 && C % G == 0 && gn_fused_ok(B, HW, C, G)) ? 1 : 0;
}

extern "C" int tb_groupnorm_fwd_splitk(const float* part, int S, int64_t npad, const float* bias, const float* rowbias, int64_t ldrb,
                                       const void* R, int64_t ldr, void* x, int64_t ldx, void* y, int64_t ldy, const float* gamma,
                                       const float* beta, float* stats, int B, int HW, int C, int G, float eps, int silu, tb_stream_t stream) {
  (void)hipGetLastError();
  if (!part || S < 1 || !x || !y || !gamma || !beta || !stats) return TB_EINVAL;
  if (C % 8 || C > GN_MAXC || G <= 0 || G > 64 || C % G || ldx % 8 || ldy % 8 || B <= 0 || HW <= 0 || npad < C || npad % 8) return TB_EINVAL;
  if (((uintptr_t)part) % 16 || (R && (ldr % 8 || ((uintptr_t)R) % 16)) || ((uintptr_t)x) % 16 || ((uintptr_t)y) % 16) return TB_EINVAL;
  if (!gn_fused_ok(B, HW, C, G)) return TB_EINVAL;
  GnSplitK sk{part, S, npad, (int64_t)B * HW * npad, bias, rowbias, ldrb, (const f16*)R, ldr};
  hipLaunchKernelGGL(gn_fused_fwd_kernel<true>, dim3(G * B), dim3(256), 0, (hipStream_t)stream, (const f16*)x, ldx, (f16*)y, ldy, gamma, beta,
                     stats, HW, C, G, eps, silu, sk);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int tb_groupnorm_bwd_splitk(const float* part, int S, int64_t npad, const void* x, int64_t ldx, const float* gamma,
                                       const float* beta, const float* stats, const void* add, int64_t ldadd, void* dx, int64_t lddx, int B,
                                       int HW, int C, int G, int silu, tb_stream_t stream) {
  (void)hipGetLastError();
  if (!part || S < 1 || !x || !gamma || !beta || !stats || !dx) return TB_EINVAL;
  if (C % 8 || C > GN_MAXC || G <= 0 || G > 64 || C % G || ldx % 8 || lddx % 8 || (add && ldadd % 8) || npad < C || npad % 8) return TB_EINVAL;
  if (((uintptr_t)part) % 16) return TB_EINVAL;
  if (!gn_fused_ok(B, HW, C, G)) return TB_EINVAL;
  GnSplitK sk{part, S, npad, (int64_t)B * HW * npad, nullptr, nullptr, 0, nullptr, 0};
  hipLaunchKernelGGL(gn_fused_bwd_kernel<true>, dim3(G * B), dim3(256), 0, (hipStream_t)stream, (const f16*)nullptr, (int64_t)0, (const f16*)x,
                     ldx, gamma, beta, stats, (const f16*)add, ldadd, (f16*)dx, lddx, HW, C, G, silu, sk);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

static int ln_fwd_launch(const void* x, int64_t ldx, int x_dtype, void* y, int64_t ldy, int y_dtype, const float* gamma,
                         const float* beta, float* stats, int64_t M, int C, float eps, const float* loraA, int R, f16* tdown, int64_t ldt, int64_t lora_rows,
                         tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!x || !y || !gamma || !beta || M <= 0) return TB_EINVAL;
  if (C % 8 || C > 64 * 8 * LN_MAXV || ldx % 8 || ldy % 8) return TB_EINVAL;
  if (loraA && (!tdown || R <= 0 || R > 64 || ((uintptr_t)loraA) % 16)) return TB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)((M + 3) / 4));
  if (x_dtype == TB_F32 && y_dtype == TB_F16 && loraA && R <= 12 && C <= 1024 && (g_gn_fused & 8))   // (bit 8 of tb_groupnorm_set_variant)
    hipLaunchKernelGGL((ln_fwd_kernel<float, f16, 12>), grid, dim3(256), 0, s, (const float*)x, ldx, (f16*)y, ldy, gamma, beta, stats, M, C, eps, loraA, R, tdown, ldt, lora_rows);
  else if (x_dtype == TB_F32 && y_dtype == TB_F16)
    hipLaunchKernelGGL((ln_fwd_kernel<float, f16>), grid, dim3(256), 0, s, (const float*)x, ldx, (f16*)y, ldy, gamma, beta, stats, M, C, eps, loraA, R, tdown, ldt, lora_rows);
  else if (x_dtype == TB_F32 && y_dtype == TB_F32)
    hipLaunchKernelGGL((ln_fwd_kernel<float, float>), grid, dim3(256), 0, s, (const float*)x, ldx, (float*)y, ldy, gamma, beta, stats, M, C, eps, loraA, R, tdown, ldt, lora_rows);
  else if (x_dtype == TB_F16 && y_dtype == TB_F16)
    hipLaunchKernelGGL((ln_fwd_kernel<f16, f16>), grid, dim3(256), 0, s, (const f16*)x, ldx, (f16*)y, ldy, gamma, beta, stats, M, C, eps, loraA, R, tdown, ldt, lora_rows);
  else
    return TB_EINVAL;
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int tb_layernorm_fwd(const void* x, int64_t ldx, int x_dtype, void* y, int64_t ldy, int y_dtype, const float* gamma,
                                const float* beta, float* stats, int64_t M, int C, float eps, tb_stream_t stream) {
  return ln_fwd_launch(x, ldx, x_dtype, y, ldy, y_dtype, gamma, beta, stats, M, C, eps, nullptr, 0, nullptr, 0, 0, stream);
}

extern "C" int tb_layernorm_lora_fwd(const void* x, int64_t ldx, int x_dtype, void* y, int64_t ldy, int y_dtype, const float* gamma,
                                     const float* beta, float* stats, int64_t M, int C, float eps, const float* loraA, int R, void* t,
                                     int64_t ldt, tb_stream_t stream) {
  if (!loraA) return TB_EINVAL;
  return ln_fwd_launch(x, ldx, x_dtype, y, ldy, y_dtype, gamma, beta, stats, M, C, eps, loraA, R, (f16*)t, ldt, M, stream);
}

extern "C" int tb_layernorm_lora_rows_fwd(const void* x, int64_t ldx, int x_dtype, void* y, int64_t ldy, int y_dtype, const float* gamma,
                                          const float* beta, float* stats, int64_t M, int C, float eps, const float* loraA, int R, void* t,
                                          int64_t ldt, int64_t lora_rows, tb_stream_t stream) {
  if (!loraA || lora_rows < 0 || lora_rows > M) return TB_EINVAL;
  return ln_fwd_launch(x, ldx, x_dtype, y, ldy, y_dtype, gamma, beta, stats, M, C, eps, loraA, R, (f16*)t, ldt, lora_rows, stream);
}

extern "C" int tb_layernorm_bwd(const void* dy, int64_t lddy, int dy_dtype, const void* x, int64_t ldx, int x_dtype,
                                const float* gamma, const float* stats, const void* add, int64_t ldadd, void* dx, int64_t lddx,
                                void* dx16, int64_t lddx16, int64_t M, int C, tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!dy || !x || !gamma || !stats || !dx || M <= 0) return TB_EINVAL;
  if (C % 8 || C > 64 * 8 * LN_MAXV || ldx % 8 || lddy % 8 || lddx % 8 || (add && ldadd % 8) || (dx16 && lddx16 % 8)) return TB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)((M + 3) / 4));
  if (x_dtype == TB_F32 && dy_dtype == TB_F16)
    hipLaunchKernelGGL((ln_bwd_kernel<float, f16>), grid, dim3(256), 0, s, (const f16*)dy, lddy, (const float*)x, ldx, gamma, stats,
                       (const float*)add, ldadd, (float*)dx, lddx, (f16*)dx16, lddx16, M, C);
  else if (x_dtype == TB_F32 && dy_dtype == TB_F32)
    hipLaunchKernelGGL((ln_bwd_kernel<float, float>), grid, dim3(256), 0, s, (const float*)dy, lddy, (const float*)x, ldx, gamma,
                       stats, (const float*)add, ldadd, (float*)dx, lddx, (f16*)dx16, lddx16, M, C);
  else if (x_dtype == TB_F16 && dy_dtype == TB_F16)
    hipLaunchKernelGGL((ln_bwd_kernel<f16, f16>), grid, dim3(256), 0, s, (const f16*)dy, lddy, (const f16*)x, ldx, gamma, stats,
                       (const f16*)add, ldadd, (f16*)dx, lddx, (f16*)dx16, lddx16, M, C);
  else
    return TB_EINVAL;
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int tb_groupnorm_set_variant(int fused) {  // A/B: 0 = the two-pass kernels for every shape; returns the previous value
  const int old = g_gn_fused;
  g_gn_fused = fused;
  return old;
}
