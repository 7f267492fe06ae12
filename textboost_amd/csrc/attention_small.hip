// Attention backward for SHORT sequences in one launch: the CLIP text encoder's causal 77 x 77 self-attention (hd = 64; CLIP-L 12 heads,
// OpenCLIP-H 16) -- transformers CLIPAttention under train_textboost.py:1054-1059 / :1108.  The generic kernels of attention.hip run it as a
// dQ launch and a dK/dV launch that each recompute S and dP for 77 keys (9.5 + 10.2 us per layer, almost all of it the ~4.8 us a graph node
// costs before it does anything plus staging); here one workgroup per (batch, head) holds everything in LDS:
//     S = Q K^T, dP = dO V^T  (3 x 3 tiles of 32 x 32, once)  ->  P = exp(scale S - lse), dS = scale P (dP - delta)
//     dQ = dS K,  dK = dS^T Q,  dV = P^T dO                    (18 tile products over the sixteen waves)
// Every MFMA operand is read as "row with the contraction index contiguous" (b128 LDS reads): the products that contract over queries or keys
// take their operands from transposed copies (K^T, Q^T, dO^T, P^T, dS^T) written once -- at this size simplicity beats a transposing read.
#include "common.h"
#include "../../include/textboost_hip.h"

namespace {

constexpr int SM_T = 96;          // padded sequence length (3 tiles of 32)
constexpr int SM_HD = 64;
constexpr int SM_LDR = SM_HD + 8;  // row-major [t][d] pitch (halfs): 144 B rows, conflict-free b128 reads
constexpr int SM_LDT = SM_T + 8;   // transposed [d][t] and score [t][t] pitch (halfs)

struct SmallLds {
  f16 q[SM_T][SM_LDR], k[SM_T][SM_LDR], v[SM_T][SM_LDR], dO[SM_T][SM_LDR];   // row-major
  f16 qT[SM_HD][SM_LDT], kT[SM_HD][SM_LDT], dOT[SM_HD][SM_LDT];               // transposed
  f16 dS[SM_T][SM_LDT], dST[SM_T][SM_LDT], pT[SM_T][SM_LDT];                  // [q][k], [k][q], [k][q]
  float lse[SM_T], delta[SM_T];
};

// D[i][j] = sum_k X[i0 + i][k] * Y[j0 + j][k] over K (a multiple of 16): lane (l31, hi) ends with column j = l31, rows mfma32_row(r, hi)
template <int K>
__device__ __forceinline__ f32x16 tile_xyT(const f16* X, int ldx, int i0, const f16* Y, int ldy, int j0, int l31, int hi) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < K / 16; ++ks) {
    const f16x8 a = *(const f16x8*)(X + (i0 + l31) * ldx + ks * 16 + 8 * hi);
    const f16x8 b = *(const f16x8*)(Y + (j0 + l31) * ldy + ks * 16 + 8 * hi);
    acc = TB_MFMA_32x32x16(a, b, acc);
  }
  return acc;
}

__global__ __launch_bounds__(1024) void attn_bwd_small_kernel(const tb_attn_desc p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  SmallLds& L = *reinterpret_cast<SmallLds*>(smem_raw);
  const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int h = blockIdx.x, b = blockIdx.y, T = p.Sq, hd = SM_HD;
  const f16* Qg = (const f16*)p.Q + (int64_t)b * T * p.ldq + h * hd;
  const f16* Kg = (const f16*)p.K + (int64_t)b * T * p.ldk + h * hd;
  const f16* Vg = (const f16*)p.V + (int64_t)b * T * p.ldv + h * hd;
  const f16* Og = (const f16*)p.O + (int64_t)b * T * p.ldo + h * hd;
  const f16* dOg = (const f16*)p.dO + (int64_t)b * T * p.lddo + h * hd;
  // ---- stage: rows (row, 8-half chunk) units; rows >= T are zero.  delta[row] = sum_d dO * O from the same loads
  for (int u = t; u < SM_T * 8; u += 1024) {  // (768 units: taken by whole waves only)
    const int row = u >> 3, c = (u & 7) * 8;
    f16x8 q8 = {0, 0, 0, 0, 0, 0, 0, 0}, k8 = q8, v8 = q8, d8 = q8, o8 = q8;
    if (row < T) {
      q8 = *(const f16x8*)(Qg + (int64_t)row * p.ldq + c);
      k8 = *(const f16x8*)(Kg + (int64_t)row * p.ldk + c);
      v8 = *(const f16x8*)(Vg + (int64_t)row * p.ldv + c);
      d8 = *(const f16x8*)(dOg + (int64_t)row * p.lddo + c);
      o8 = *(const f16x8*)(Og + (int64_t)row * p.ldo + c);
    }
    *(f16x8*)&L.q[row][c] = q8;
    *(f16x8*)&L.k[row][c] = k8;
    *(f16x8*)&L.v[row][c] = v8;
    *(f16x8*)&L.dO[row][c] = d8;
    float ds = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      L.qT[c + e][row] = q8[e];
      L.kT[c + e][row] = k8[e];
      L.dOT[c + e][row] = d8[e];
      ds += (float)d8[e] * (float)o8[e];
    }
    // the 8 chunks of a row sit in 8 consecutive lanes
    ds += __shfl_xor(ds, 1, 64);
    ds += __shfl_xor(ds, 2, 64);
    ds += __shfl_xor(ds, 4, 64);
    if ((u & 7) == 0) {
      L.delta[row] = ds;
      L.lse[row] = row < T ? p.LSE[((int64_t)b * p.H + h) * T + row] : 0.f;
      if (p.Delta && row < T) p.Delta[((int64_t)b * p.H + h) * T + row] = ds;
    }
  }
  __syncthreads();
  // ---- S and dP tiles (i = query tile, j = key tile), P and dS with the causal / length mask; 9 tile jobs over 16 waves
  const float scale = p.scale;
  for (int job = wave; job < 9; job += 16) {
    const int i0 = (job / 3) * 32, j0 = (job % 3) * 32;
    if (p.causal && j0 > i0 + 31) {  // wholly above the diagonal: zeros
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = i0 + (r & 3) + 8 * (r >> 2) + 4 * hi, k = j0 + l31;
        L.dS[q][k] = (f16)0.f;
        L.dST[k][q] = (f16)0.f;
        L.pT[k][q] = (f16)0.f;
      }
      continue;
    }
    const f32x16 s = tile_xyT<SM_HD>(&L.q[0][0], SM_LDR, i0, &L.k[0][0], SM_LDR, j0, l31, hi);
    const f32x16 dp = tile_xyT<SM_HD>(&L.dO[0][0], SM_LDR, i0, &L.v[0][0], SM_LDR, j0, l31, hi);
    const int k = j0 + l31;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      f16x4 pv, dv;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * r4 + e;
        const int q = i0 + 8 * r4 + 4 * hi + e;   // mfma32_row(r, hi) = (r & 3) + 8 * (r >> 2) + 4 * hi
        const bool ok = q < T && k < T && (!p.causal || k <= q);
        const float pe = ok ? __expf(scale * s[r] - L.lse[q]) : 0.f;
        const float de = pe * (dp[r] - L.delta[q]) * scale;
        pv[e] = (f16)pe;
        dv[e] = (f16)de;
        L.dS[q][k] = dv[e];
      }
      *(f16x4*)&L.pT[k][i0 + 8 * r4 + 4 * hi] = pv;    // 4 consecutive queries of this key
      *(f16x4*)&L.dST[k][i0 + 8 * r4 + 4 * hi] = dv;
    }
  }
  __syncthreads();
  // ---- dQ = dS K (X = dS [q][k], Y = K^T [d][k]); dK = dS^T Q (X = dS^T [k][q], Y = Q^T [d][q]); dV = P^T dO (X = P^T [k][q], Y = dO^T [d][q])
  for (int job = wave; job < 18; job += 16) {
    const int which = job / 6, rem = job - which * 6;
    const int i0 = (rem >> 1) * 32, j0 = (rem & 1) * 32;
    const f16* X = which == 0 ? &L.dS[0][0] : (which == 1 ? &L.dST[0][0] : &L.pT[0][0]);
    const f16* Y = which == 0 ? &L.kT[0][0] : (which == 1 ? &L.qT[0][0] : &L.dOT[0][0]);
    const f32x16 acc = tile_xyT<SM_T>(X, SM_LDT, i0, Y, SM_LDT, j0, l31, hi);
    f16* out = which == 0 ? (f16*)p.dQ : (which == 1 ? (f16*)p.dK : (f16*)p.dV);
    const int64_t ldo = which == 0 ? p.lddq : (which == 1 ? p.lddk : p.lddv);
    out += (int64_t)b * T * ldo + h * hd + j0 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = i0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (row < T) out[(int64_t)row * ldo] = (f16)acc[r];
    }
  }
}

}  // namespace

// true when the short-sequence kernel takes this backward (attention.hip asks before its generic dispatch)
static int g_small_lds_ok = -1;   // does the device grant sizeof(SmallLds) of dynamic LDS to the kernel?  (asked once; a part with 64 KB says no and
                                  // attention.hip's generic dQ + dK/dV launches take the call instead of a TB_ELAUNCH)
bool tb_attn_small_bwd_ok(const tb_attn_desc& d) {
  if (g_small_lds_ok < 0)
    g_small_lds_ok = hipFuncSetAttribute((const void*)attn_bwd_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SmallLds)) == hipSuccess;
  if (!g_small_lds_ok) {
    (void)hipGetLastError();
    return false;
  }
  return d.hd == SM_HD && d.Sq == d.Skv && d.Sq <= SM_T && d.Sq >= 1 && d.ldq % 8 == 0 && d.ldk % 8 == 0 && d.ldv % 8 == 0 && d.ldo % 8 == 0 &&
         d.lddo % 8 == 0 && ((uintptr_t)d.Q) % 16 == 0 && ((uintptr_t)d.K) % 16 == 0 && ((uintptr_t)d.V) % 16 == 0 && ((uintptr_t)d.O) % 16 == 0 &&
         ((uintptr_t)d.dO) % 16 == 0 && d.LSE && d.dQ && d.dK && d.dV;
}
int tb_attn_small_bwd(const tb_attn_desc& d, hipStream_t s) {
  const size_t lds = sizeof(SmallLds);   // (granted: tb_attn_small_bwd_ok set the attribute)
  hipLaunchKernelGGL(attn_bwd_small_kernel, dim3(d.H, d.B), dim3(1024), lds, s, d);
  TB_CHECK_LAUNCH();
  return TB_OK;
}
