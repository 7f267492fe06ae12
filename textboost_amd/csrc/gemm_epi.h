// Shared pieces of the MFMA GEMM / conv kernels (gemm.hip, gemm8.hip): HBM->LDS direct loads and the fused epilogue of one
// (row, 8 consecutive columns) output unit.
#pragma once
#include "common.h"
#include "../../include/textboost_hip.h"

namespace {

constexpr int BK = 64;  // K granularity required of callers (halfs)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __attribute__((aligned(128))) const f16 g_zero_line[64] = {};

__device__ __forceinline__ void glds16(const f16* src, f16* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}

// global -> LDS DMA of 64 x 16 bytes as ONE asm statement (destination = the wave-uniform LDS byte address in M0 + 16 * lane).  Unlike the builtin it
// may sit between MFMAs: there the builtin gets an `s_waitcnt vmcnt(0)` and a VGPR round trip of its M0 value in front (ff_fused.hip / lin320.hip)
__device__ __forceinline__ void glds16_asm(const void* src, uint32_t lds_byte_addr) {
  const uint32_t m = __builtin_amdgcn_readfirstlane(lds_byte_addr);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m), "v"(src) : "memory", "m0");
}
// ... with a wave-uniform 64-bit base (SGPR pair) + a 32-bit per-lane byte offset: no 64-bit VGPR address (the 256 x 160 convolution tile has none to spare)
__device__ __forceinline__ void glds16_asm_so(const void* sbase, uint32_t voff, uint32_t lds_byte_addr) {
  const uint32_t m = __builtin_amdgcn_readfirstlane(lds_byte_addr);
  const uint64_t b = (uint64_t)(uintptr_t)sbase;
  const uint64_t bs = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32)) << 32) |
                      (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b);   // (the builtin returns int: no sign extension into the high dword)
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m), "v"(voff), "s"(bs) : "memory", "m0");
}

// Epilogue for 8 consecutive output columns n..n+7 of row m (shared by the MFMA kernel and the split-K reducer):
// v = alpha*acc + bias + rowbias + R ; activation ; store C (fp16 | fp32), optionally C2 (pre-activation).
// All global accesses are 16-byte vectors when the row pitch / base alignment allow (EpiFlags), else scalar.
struct EpiFlags {
  bool c_vec, r_vec, c2_vec;
};
__device__ __forceinline__ EpiFlags epi_flags(const tb_gemm_desc& p) {
  EpiFlags f;
  f.c_vec = (p.ldc % 8 == 0) && (((uintptr_t)p.C) % 16 == 0);
  f.r_vec = p.R && (p.ldr % 8 == 0) && (((uintptr_t)p.R) % 16 == 0);
  f.c2_vec = p.C2 && (p.ldc2 % 8 == 0) && (((uintptr_t)p.C2) % 16 == 0);
  return f;
}
// residual (R) and the QUICK_GELU_GRAD pre-activation (C2) are fetched by the two loaders below so that callers can issue the
// loads of all their units back to back BEFORE the arithmetic (cold HBM latency is paid once, not once per unit).
__device__ __forceinline__ void epi_load_r8(const tb_gemm_desc& p, const EpiFlags& f, int64_t m, int64_t n, float* r) {
  const bool full = n + 7 < p.N;
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = 0.f;
  if (!p.R) return;
  if (p.r_dtype == TB_F32) {
    const float* rp = (const float*)p.R + m * p.ldr + n;
    if (full && f.r_vec) {
      const f32x4 r0 = *(const f32x4*)rp, r1 = *(const f32x4*)(rp + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        r[e] = r0[e];
        r[4 + e] = r1[e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (n + e < p.N) r[e] = rp[e];
    }
  } else {
    const f16* rp = (const f16*)p.R + m * p.ldr + n;
    if (full && f.r_vec) {
      const f16x8 rv = *(const f16x8*)rp;
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] = (float)rv[e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (n + e < p.N) r[e] = (float)rp[e];
    }
  }
}
// second auxiliary vector: the g half of the packed pre-gate projections for GEGLU_GRAD (h comes through epi_load_aux8)
__device__ __forceinline__ f16x8 epi_load_aux8b(const tb_gemm_desc& p, const EpiFlags& f, int64_t m, int64_t n) {
  f16x8 a = {0, 0, 0, 0, 0, 0, 0, 0};
  if (p.act != TB_ACT_GEGLU_GRAD) return a;
  const f16* c2 = (const f16*)p.C2 + m * p.ldc2 + (n >> 5) * 64 + (n & 31) + 32;
  if (f.c2_vec) return *(const f16x8*)c2;
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] = c2[e];
  return a;
}
__device__ __forceinline__ f16x8 epi_load_aux8(const tb_gemm_desc& p, const EpiFlags& f, int64_t m, int64_t n) {
  f16x8 a = {0, 0, 0, 0, 0, 0, 0, 0};
  if (p.act == TB_ACT_GEGLU_GRAD) {
    const f16* c2 = (const f16*)p.C2 + m * p.ldc2 + (n >> 5) * 64 + (n & 31);
    if (f.c2_vec) return *(const f16x8*)c2;
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = c2[e];
    return a;
  }
  if (p.act != TB_ACT_QUICK_GELU_GRAD && p.act != TB_ACT_GELU_GRAD) return a;
  const f16* c2 = (const f16*)p.C2 + m * p.ldc2 + n;
  if (n + 7 < p.N && f.c2_vec) return *(const f16x8*)c2;
#pragma unroll
  for (int e = 0; e < 8; ++e)
    if (n + e < p.N) a[e] = c2[e];
  return a;
}
// bias8: bias[n..n+7] (0 when absent / out of range), r: preloaded residual, aux: preloaded pre-activation
__device__ __forceinline__ void epilogue8(const tb_gemm_desc& p, const EpiFlags& f, int64_t m, int64_t n, float* v, const float* bias8,
                                          const float* r, const f16x8& aux) {
  const bool full = n + 7 < p.N;
  const float* rb = p.rowbias ? p.rowbias + (m / p.rows_per_group) * p.ldrb : nullptr;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    v[e] = v[e] * p.alpha + bias8[e] + r[e];
    if (rb && (full || n + e < p.N)) v[e] += rb[n + e];
  }
  if (p.act == TB_ACT_QUICK_GELU) {
    f16x8 pre;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      pre[e] = (f16)v[e];
      v[e] = quick_gelu_f((float)pre[e]);  // fp16 linear output feeds the activation, as under autocast
    }
    if (p.C2) {
      f16* c2 = (f16*)p.C2 + m * p.ldc2 + n;
      if (full && f.c2_vec) *(f16x8*)c2 = pre;
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (n + e < p.N) c2[e] = pre[e];
      }
    }
  } else if (p.act == TB_ACT_GELU) {  // erf GELU (OpenCLIP-H text MLP of SD2.x), same save-pre-activation contract as QUICK_GELU
    f16x8 pre;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      pre[e] = (f16)v[e];
      v[e] = gelu_erf_f((float)pre[e]);
    }
    if (p.C2) {
      f16* c2 = (f16*)p.C2 + m * p.ldc2 + n;
      if (full && f.c2_vec) *(f16x8*)c2 = pre;
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (n + e < p.N) c2[e] = pre[e];
      }
    }
  } else if (p.act == TB_ACT_SILU) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
  } else if (p.act == TB_ACT_QUICK_GELU_GRAD) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= quick_gelu_grad_f((float)aux[e]);
  } else if (p.act == TB_ACT_GELU_GRAD) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= gelu_erf_grad_f((float)aux[e]);
  }
  if (p.act == TB_ACT_GEGLU_GRAD) {
    // v = d(gated)[m, n..n+7]; C2 = packed pre-gate projections [M, 2N] ([h32|g32] blocks); C = d(proj) [M, 2N], same packing:
    // d h = v * gelu(g),  d g = v * h * gelu'(g)          (backward of diffusers GEGLU fused into the ff.net.2 dgrad GEMM)
    const int64_t pc = (n >> 5) * 64 + (n & 31);  // packed column of h for gate column n
    f16* c = (f16*)p.C + m * p.ldc + pc;
    const f16x8 hh = aux, gg = epi_load_aux8b(p, f, m, n);
    f16x8 dh, dg;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float g = (float)gg[e];
      float ge, dge;
      gelu_erf_both_f(g, ge, dge);
      dh[e] = (f16)(v[e] * ge);
      dg[e] = (f16)(v[e] * (float)hh[e] * dge);
    }
    if (f.c_vec) {
      *(f16x8*)c = dh;
      *(f16x8*)(c + 32) = dg;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        c[e] = dh[e];
        c[32 + e] = dg[e];
      }
    }
    return;
  }
  if (p.c_dtype == TB_F32) {
    float* c = (float*)p.C + m * p.ldc + n;
    if (full && f.c_vec) {
      f32x4 o0, o1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o0[e] = v[e];
        o1[e] = v[4 + e];
      }
      *(f32x4*)c = o0;
      *(f32x4*)(c + 4) = o1;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (n + e < p.N) c[e] = v[e];
    }
  } else {
    f16* c = (f16*)p.C + m * p.ldc + n;
    if (full && f.c_vec) {
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (f16)v[e];
      *(f16x8*)c = o;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (n + e < p.N) c[e] = (f16)v[e];
    }
  }
}

__device__ __forceinline__ void epi_load_bias8(const tb_gemm_desc& p, int64_t n, float* b) {
#pragma unroll
  for (int e = 0; e < 8; ++e) b[e] = (p.bias && n + e < p.N) ? p.bias[n + e] : 0.f;
}

}  // namespace
