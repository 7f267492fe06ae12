// On-device optimizer tail of the TextBoost step (gfx950), no host round trip:
//   GradScaler unscale + inf check + dynamic scale update   (accelerate fp16; train_textboost.py:1108, :1134)
//   clip_grad_norm_(text_model.encoder.parameters(), 1.0)     (:1128-1133; only the LoRA tensors have grads there)
//   AdamW group 0 (token embedding, lr emb_lr) / group 1 (LoRA, lr)  (:828-854, :1134)
//   embedding re-normalisation of the added rows               (:1138-1149)
// State lives in one small fp32 device array `st` (see TB_ST_* in the header).
#include "common.h"
#include "../../include/textboost_hip.h"

// No FMA contraction in this file: torch's AdamW / clip / GradScaler arithmetic is one rounding per operation, and the two forms of the tail below
// (ten launches / two launches) must round identically whatever the compiler would fuse in either context (found by the bit-equality test: 1 ulp).
#pragma clang fp contract(off)

namespace {

// deterministic sum of squares: 64 blocks write partials, the LAST-launched tiny kernel adds them in a fixed order
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ part) {
  __shared__ float red[4];
  float a = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = x[i];
    a += v * v;
  }
  a = block_sum_256(a, red);
  if (threadIdx.x == 0) part[blockIdx.x] = a;
}
__global__ void sumsq_final_kernel(const float* __restrict__ part, int nb, float* __restrict__ out) {
  const int lane = threadIdx.x;  // 64 threads
  float a = lane < nb ? part[lane] : 0.f;
  a = wave_sum(a);
  if (lane == 0) out[0] = a;
}

// one thread: derive every per-step scalar from the two gradient sums of squares
__global__ void scaler_update_kernel(float* __restrict__ st, float max_norm, float beta1, float beta2, float growth_factor,
                                     float backoff_factor, float growth_interval, int use_scaler, float grad_div) {
  const float scale = st[TB_ST_LOSS_SCALE];
  const float ss_lora = st[TB_ST_SUMSQ_LORA], ss_emb = st[TB_ST_SUMSQ_EMB];
  // GradScaler.unscale_'s inf check exists only under the scaler (fp16 mode: text-encoder groups only, the UNet adapters need the fp32 mode); without
  // it (no-AMP run) torch's optimizer.step() is never skipped, whatever the gradients hold
  const bool found_inf = use_scaler && !(isfinite(ss_lora) && isfinite(ss_emb));
  // grad_div = data-parallel world size: the gradient buffer holds the all-reduce SUM, DDP's mean is folded into the coefficients here
  const float inv = 1.f / (scale * grad_div);
  const float total = sqrtf(ss_lora) * inv;                      // norm of the unscaled LoRA grads
  const float clip = fminf(max_norm / (total + 1e-6f), 1.f);    // torch clip_grad_norm_
  st[TB_ST_FOUND_INF] = found_inf ? 1.f : 0.f;
  st[TB_ST_GRAD_NORM] = total;
  st[TB_ST_COEF_LORA] = inv * clip;
  st[TB_ST_COEF_EMB] = inv;
  if (!found_inf) {
    const float step = st[TB_ST_STEP] + 1.f;
    st[TB_ST_STEP] = step;
    st[TB_ST_BC1] = 1.f - powf(beta1, step);
    st[TB_ST_BC2] = 1.f - powf(beta2, step);
  }
  if (use_scaler) {  // torch.cuda.amp.GradScaler.update()
    if (found_inf) {
      st[TB_ST_LOSS_SCALE] = scale * backoff_factor;
      st[TB_ST_GROWTH_TRACKER] = 0.f;
    } else {
      const float tr = st[TB_ST_GROWTH_TRACKER] + 1.f;
      if (tr >= growth_interval) {
        st[TB_ST_LOSS_SCALE] = scale * growth_factor;
        st[TB_ST_GROWTH_TRACKER] = 0.f;
      } else {
        st[TB_ST_GROWTH_TRACKER] = tr;
      }
    }
  }
}

// lr_scheduler.step() (diffusers get_scheduler LambdaLR, :911-916, :1135) on the device: the multiplier of the NEXT optimizer step is
// lambda(k), k = optimizer steps that were NOT skipped so far (accelerate's AcceleratedScheduler does not advance on an overflow step)
__global__ void lr_from_table_kernel(float* __restrict__ st, const float* __restrict__ table, int n) {
  int k = (int)st[TB_ST_STEP];
  k = k < 0 ? 0 : (k >= n ? n - 1 : k);
  st[TB_ST_LR_MULT] = table[k] - 1.f;
}

// torch.optim.AdamW (decoupled decay first, then Adam update with bias corrections), grads pre-multiplied by coef
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, float lr, float beta1, float beta2, float eps,
                                                    float wd, const float* __restrict__ st, int coef_slot) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (st[TB_ST_FOUND_INF] != 0.f) return;
  const float gi = g[i] * st[coef_slot];
  const float bc1 = st[TB_ST_BC1], bc2 = st[TB_ST_BC2];
  lr *= 1.f + st[TB_ST_LR_MULT];  // LambdaLR: every group's lr is its base lr times lambda(step) (slot holds lambda - 1)
  float pi = p[i] * (1.f - lr * wd);
  const float mi = beta1 * m[i] + (1.f - beta1) * gi;
  const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
  pi -= (lr / bc1) * (mi / denom);
  p[i] = pi;
}

// rows that never receive gradient: AdamW still applies p *= (1 - lr*wd) every (non-skipped) step
__global__ __launch_bounds__(256) void decay_kernel(float* __restrict__ p, int64_t n4, float factor, const float* __restrict__ st) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  if (st[TB_ST_FOUND_INF] != 0.f) return;
  f32x4 x = ((f32x4*)p)[i];
  x *= 1.f - (1.f - factor) * (1.f + st[TB_ST_LR_MULT]);  // factor = 1 - lr * wd at the base lr
  ((f32x4*)p)[i] = x;
}

// w[a] <- w[a] * min(mean_norm, |w[a]|) / |w[a]| ; norms[a] = |w[a]| before clamping  (:1143-1149)
__global__ __launch_bounds__(256) void renorm_rows_kernel(float* __restrict__ w, int D, float mean_norm, float* __restrict__ norms) {
  __shared__ float red[4];
  float* row = w + (int64_t)blockIdx.x * D;
  float a = 0.f;
  for (int c = threadIdx.x; c < D; c += 256) a += row[c] * row[c];
  const float nrm = sqrtf(block_sum_256(a, red));
  const float f = fminf(mean_norm, nrm) / nrm;
  for (int c = threadIdx.x; c < D; c += 256) row[c] *= f;
  if (threadIdx.x == 0 && norms) norms[blockIdx.x] = nrm;
}

// mean over rows of the L2 row norm of a [rows, D] fp32 table (:1017 mean_norm); two-stage deterministic
__global__ __launch_bounds__(256) void row_norm_kernel(const float* __restrict__ w, int D, float* __restrict__ norms) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  float a = 0.f;
  for (int c = lane; c < D; c += 64) {
    const float x = w[row * D + c];
    a += x * x;
  }
  a = wave_sum(a);
  if (lane == 0) norms[row] = sqrtf(a);
}

// ---------------------------------------------------------------------------------------------- the tail as two launches (tb_optimizer_tail)
constexpr int OPT_SNAP = 128;      // ws[128..131]: loss scale, step, growth tracker of BEFORE the step
constexpr int OPT_DECAY_V4 = 16;   // float4 per thread of a decay slab (64 KB per workgroup)

// launch 1: the two sumsq_partial_kernel passes in one grid of 64 blocks (same strides, same block_sum: bit-equal partials) + the snapshot
__global__ __launch_bounds__(256) void opt_reduce_kernel(const tb_opt_desc d) {
  __shared__ float red[4];
  const float* x = d.grad;
  float a = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < d.n_lora; i += (int64_t)gridDim.x * 256) {
    const float v = x[i];
    a += v * v;
  }
  a = block_sum_256(a, red);
  if (threadIdx.x == 0) d.ws[blockIdx.x] = a;
  const float* e = d.grad + d.n_lora;
  const int64_t ne = (int64_t)d.n_added * d.D;
  float b = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < ne; i += (int64_t)gridDim.x * 256) {
    const float v = e[i];
    b += v * v;
  }
  b = block_sum_256(b, red);
  if (threadIdx.x == 0) d.ws[64 + blockIdx.x] = b;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    d.ws[OPT_SNAP + 0] = d.state[TB_ST_LOSS_SCALE];
    d.ws[OPT_SNAP + 1] = d.state[TB_ST_STEP];
    d.ws[OPT_SNAP + 2] = d.state[TB_ST_GROWTH_TRACKER];
  }
}

struct OptScalars {
  float found_inf, coef_lora, coef_emb, bc1, bc2, lr_mult, total, ss_lora, ss_emb, step, scale;
};
// what sumsq_final_kernel + lr_from_table_kernel + scaler_update_kernel leave in the state, from the partials and the snapshot (every lane of
// every wave computes the same values: the wave_sum order is sumsq_final_kernel's)
__device__ __forceinline__ OptScalars opt_scalars(const tb_opt_desc& d) {
  const int lane = threadIdx.x & 63;
  OptScalars o;
  o.ss_lora = d.n_lora > 0 ? wave_sum(d.ws[lane]) : d.state[TB_ST_SUMSQ_LORA];          // (a group that does not exist keeps its slot, as the
  o.ss_emb = d.n_added > 0 ? wave_sum(d.ws[64 + lane]) : d.state[TB_ST_SUMSQ_EMB];       //  separate entry points would: never written, zero)
  o.scale = d.ws[OPT_SNAP + 0];
  const float step0 = d.ws[OPT_SNAP + 1];
  if (d.lr_table) {
    int k = (int)step0;
    k = k < 0 ? 0 : (k >= d.lr_table_n ? d.lr_table_n - 1 : k);
    o.lr_mult = d.lr_table[k] - 1.f;
  } else {
    o.lr_mult = d.state[TB_ST_LR_MULT];   // host-written only: no workgroup of this launch writes it in this case
  }
  const bool found_inf = d.use_scaler && !(isfinite(o.ss_lora) && isfinite(o.ss_emb));
  const float inv = 1.f / (o.scale * d.grad_div);
  o.total = sqrtf(o.ss_lora) * inv;
  const float clip = fminf(d.max_norm / (o.total + 1e-6f), 1.f);
  o.found_inf = found_inf ? 1.f : 0.f;
  o.coef_lora = inv * clip;
  o.coef_emb = inv;
  o.step = found_inf ? step0 : step0 + 1.f;
  o.bc1 = o.bc2 = 1.f;
  return o;
}
__device__ __forceinline__ void opt_bias_corrections(OptScalars& o, const tb_opt_desc& d) {   // (the roles that run AdamW, and the publisher)
  o.bc1 = 1.f - powf(d.beta1, o.step);
  o.bc2 = 1.f - powf(d.beta2, o.step);
}
__device__ __forceinline__ void opt_adamw_elem(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                               int64_t i, float lr, float coef, const OptScalars& o, const tb_opt_desc& d) {  // = adamw_kernel
  const float gi = g[i] * coef;
  lr *= 1.f + o.lr_mult;
  float pi = p[i] * (1.f - lr * d.wd);
  const float mi = d.beta1 * m[i] + (1.f - d.beta1) * gi;
  const float vi = d.beta2 * v[i] + (1.f - d.beta2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / sqrtf(o.bc2) + d.eps;
  pi -= (lr / o.bc1) * (mi / denom);
  p[i] = pi;
}
// launch 2.  Workgroup roles by index: [0, nbL) LoRA elements | [nbL, nbL + n_added) one added row each | [.., + nbU) UNet adapter elements |
// the rest: decay slabs.  Workgroup 0 also publishes the state (whatever its role is).
__global__ __launch_bounds__(256) void opt_apply_kernel(const tb_opt_desc d, int nbL, int nbU) {
  __shared__ float red[4];
  OptScalars o = opt_scalars(d);
  const int b = blockIdx.x, t = threadIdx.x;
  if (b < nbL + d.n_added + nbU || b == 0) opt_bias_corrections(o, d);   // (block-uniform)
  if (b == 0 && t == 0) {   // scaler_update_kernel's stores (+ the sums of squares sumsq_final_kernel stored, + lr_from_table_kernel's slot)
    float* st = d.state;
    st[TB_ST_SUMSQ_LORA] = o.ss_lora;
    st[TB_ST_SUMSQ_EMB] = o.ss_emb;
    if (d.lr_table) st[TB_ST_LR_MULT] = o.lr_mult;
    st[TB_ST_FOUND_INF] = o.found_inf;
    st[TB_ST_GRAD_NORM] = o.total;
    st[TB_ST_COEF_LORA] = o.coef_lora;
    st[TB_ST_COEF_EMB] = o.coef_emb;
    if (o.found_inf == 0.f) {
      st[TB_ST_STEP] = o.step;
      st[TB_ST_BC1] = o.bc1;
      st[TB_ST_BC2] = o.bc2;
    }
    if (d.use_scaler) {
      const float tr0 = d.ws[OPT_SNAP + 2];
      if (o.found_inf != 0.f) {
        st[TB_ST_LOSS_SCALE] = o.scale * d.backoff_factor;
        st[TB_ST_GROWTH_TRACKER] = 0.f;
      } else {
        const float tr = tr0 + 1.f;
        if (tr >= d.growth_interval) {
          st[TB_ST_LOSS_SCALE] = o.scale * d.growth_factor;
          st[TB_ST_GROWTH_TRACKER] = 0.f;
        } else {
          st[TB_ST_GROWTH_TRACKER] = tr;
        }
      }
    }
  }
  if (b < nbL) {
    const int64_t i = (int64_t)b * 256 + t;
    if (i < d.n_lora && o.found_inf == 0.f) opt_adamw_elem(d.p_lora, d.grad, d.m_lora, d.v_lora, i, d.lr, o.coef_lora, o, d);
    return;
  }
  if (b < nbL + d.n_added) {   // AdamW on the row, then renorm_rows_kernel on it (an overflow step skips the update, never the clamp)
    const int r = b - nbL;
    float* row = d.p_added + (int64_t)r * d.D;
    if (o.found_inf == 0.f)
      for (int c = t; c < d.D; c += 256) {
        const int64_t i = (int64_t)r * d.D + c;
        opt_adamw_elem(d.p_added, d.grad + d.n_lora, d.m_emb, d.v_emb, i, d.emb_lr, o.coef_emb, o, d);
      }
    float a = 0.f;
    for (int c = t; c < d.D; c += 256) a += row[c] * row[c];   // (a thread re-reads exactly the elements it wrote)
    const float nrm = sqrtf(block_sum_256(a, red));
    const float f = fminf(d.mean_norm, nrm) / nrm;
    for (int c = t; c < d.D; c += 256) row[c] *= f;
    if (t == 0 && d.added_norms) d.added_norms[r] = nrm;
    return;
  }
  if (b < nbL + d.n_added + nbU) {
    const int64_t i = (int64_t)(b - nbL - d.n_added) * 256 + t;
    const int64_t ne = (int64_t)d.n_added * d.D;
    if (i < d.n_unet && o.found_inf == 0.f) opt_adamw_elem(d.p_unet, d.grad + d.n_lora + ne, d.m_unet, d.v_unet, i, d.lr, o.coef_emb, o, d);
    return;
  }
  if (o.found_inf != 0.f) return;
  {   // decay_kernel over a slab: x *= 1 - (1 - factor) (1 + lr_mult), factor = d.decay_factor = 1 - emb_lr * wd at the base lr
    const float mul = 1.f - (1.f - d.decay_factor) * (1.f + o.lr_mult);
    const int64_t n4 = d.n_decay >> 2;
    const int64_t base = (int64_t)(b - nbL - d.n_added - nbU) * (256 * OPT_DECAY_V4) + t;
    f32x4* p4 = (f32x4*)d.p_decay;
    f32x4 x[OPT_DECAY_V4];
#pragma unroll
    for (int k = 0; k < OPT_DECAY_V4; ++k) {
      const int64_t i = base + (int64_t)k * 256;
      if (i < n4) x[k] = p4[i];
    }
#pragma unroll
    for (int k = 0; k < OPT_DECAY_V4; ++k) {
      const int64_t i = base + (int64_t)k * 256;
      if (i < n4) {
        x[k] *= mul;
        p4[i] = x[k];
      }
    }
  }
}

}  // namespace

#define GRID1D(n) dim3((unsigned)(((n) + 255) / 256))

extern "C" int tb_last_hip_error_code_ = 0;
extern "C" const char* tb_last_hip_error(void) { return hipGetErrorString((hipError_t)tb_last_hip_error_code_); }

extern "C" int tb_sumsq(const float* x, int64_t n, float* out, float* ws64, tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!x || !out || !ws64 || n <= 0) return TB_EINVAL;
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, x, n, ws64);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ws64, 64, out);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int tb_scaler_update(float* state, float max_norm, float beta1, float beta2, float growth_factor, float backoff_factor,
                                float growth_interval, int use_scaler, float grad_div, tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!state || !(grad_div >= 1.f)) return TB_EINVAL;
  hipLaunchKernelGGL(scaler_update_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, max_norm, beta1, beta2, growth_factor,
                     backoff_factor, growth_interval, use_scaler, grad_div);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int tb_lr_from_table(float* state, const float* table, int n, tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!state || !table || n <= 0) return TB_EINVAL;
  hipLaunchKernelGGL(lr_from_table_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, table, n);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int tb_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float wd,
                        const float* state, int coef_slot, tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!p || !g || !m || !v || !state || n <= 0) return TB_EINVAL;
  if (coef_slot != TB_ST_COEF_LORA && coef_slot != TB_ST_COEF_EMB) return TB_EINVAL;
  hipLaunchKernelGGL(adamw_kernel, GRID1D(n), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps, wd, state,
                     coef_slot);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int tb_weight_decay(float* p, int64_t n, float factor, const float* state, tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!p || !state || n <= 0 || n % 4) return TB_EINVAL;
  hipLaunchKernelGGL(decay_kernel, GRID1D(n / 4), dim3(256), 0, (hipStream_t)stream, p, n / 4, factor, state);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int tb_renorm_rows(float* rows, int n_rows, int D, float mean_norm, float* norms, tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!rows || n_rows <= 0 || D <= 0) return TB_EINVAL;
  hipLaunchKernelGGL(renorm_rows_kernel, dim3(n_rows), dim3(256), 0, (hipStream_t)stream, rows, D, mean_norm, norms);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int tb_row_norms(const float* w, int64_t rows, int D, float* norms, tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!w || !norms || rows <= 0 || rows % 4) return TB_EINVAL;
  hipLaunchKernelGGL(row_norm_kernel, dim3((unsigned)(rows / 4)), dim3(256), 0, (hipStream_t)stream, w, D, norms);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int tb_optimizer_tail(const tb_opt_desc* dp, tb_stream_t stream) {
  (void)hipGetLastError();  // drop any stale (non-sticky) error left by an earlier runtime call
  if (!dp) return TB_EINVAL;
  const tb_opt_desc d = *dp;
  if (!d.state || !d.grad || !d.ws || !(d.grad_div >= 1.f) || d.n_lora < 0 || d.n_added < 0 || d.n_unet < 0 || d.n_decay < 0 || d.n_decay % 4) return TB_EINVAL;
  if (d.n_lora && (!d.p_lora || !d.m_lora || !d.v_lora)) return TB_EINVAL;
  if (d.n_added && (!d.p_added || !d.m_emb || !d.v_emb || d.D <= 0)) return TB_EINVAL;
  if (d.n_unet && (!d.p_unet || !d.m_unet || !d.v_unet)) return TB_EINVAL;
  if (d.n_decay && (!d.p_decay || ((uintptr_t)d.p_decay) % 16)) return TB_EINVAL;
  if (d.lr_table && d.lr_table_n <= 0) return TB_EINVAL;
  const int64_t nbL = (d.n_lora + 255) / 256, nbU = (d.n_unet + 255) / 256;
  const int64_t nbD = ((d.n_decay >> 2) + 256 * OPT_DECAY_V4 - 1) / (256 * OPT_DECAY_V4);
  const int64_t nb = nbL + d.n_added + nbU + nbD;
  if (nb <= 0 || nb >= ((int64_t)1 << 31)) return TB_EINVAL;
  hipLaunchKernelGGL(opt_reduce_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, d);
  hipLaunchKernelGGL(opt_apply_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, d, (int)nbL, (int)nbU);
  TB_CHECK_LAUNCH();
  return TB_OK;
}
