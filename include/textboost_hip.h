/* libtextboost_hip.so -- C ABI of the MI355X (gfx950) kernels behind the TextBoost training step.
 *
 * The reference (nahyeonkaty/textboost) has no FFI: its hot path train_textboost.py:1024-1150 calls
 * three Python modules (text_encoder, unet, optimizer) whose arithmetic executes inside
 * torch/diffusers/transformers/peft kernels.  Each entry point below replaces the device work of one
 * class of those library ops; the comment on each cites the reference call site whose work it does.
 * The Python host (textboost_amd/) binds them with ctypes -- see INTEGRATION.md.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (PyTorch tensors); no function allocates,
 *    frees, synchronises or throws.  Return 0 on success, negative errno-style code otherwise
 *    (-22 invalid argument, -5 launch failure).
 *  - `stream` is a hipStream_t; all work is enqueued on it (graph-capture safe).
 *  - activations are row-major 2-D [rows, channels] with an explicit row stride `ld*` in ELEMENTS
 *    (UNet: NHWC, rows = b*H*W + y*W + x); fp16 unless stated; statistics/optimizer state fp32.
 */
#ifndef TEXTBOOST_HIP_H
#define TEXTBOOST_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* tb_stream_t; /* hipStream_t */

/* ---- dtype / activation codes ------------------------------------------------------------- */
enum { TB_F16 = 0, TB_F32 = 1 };
enum { TB_ACT_NONE = 0, TB_ACT_QUICK_GELU = 1, TB_ACT_GEGLU = 2 };
enum { TB_A_LINEAR = 0, TB_A_CONV3X3 = 1 };

/* ---- MFMA GEMM family: C[M,N] = A[M,K] * W[N,K]^T (+ epilogue), fp16 in, fp32 accumulate ------
 * Replaces every nn.Linear / 1x1 conv / 3x3 conv (implicit GEMM, NHWC) forward and input-gradient
 * (dgrad; weights pre-transposed by the host) the reference triggers through
 * unet(...) train_textboost.py:1063-1067, text_encoder(...) :1054-1059/:1099-1100 and
 * accelerator.backward(loss) :1108.  LoRA (peft lora.Linear, :700-722) rides as a second K-source:
 * A = [x | xA^T], W = [W | B].
 */
typedef struct tb_gemm_desc {
  int64_t M, N, K;            /* K % 64 == 0 */
  const void* A;  int64_t lda;   /* linear: fp16 [M, >=K1] ; conv: NHWC [B,Hin,Win,Cin], pixel stride lda */
  const void* A2; int64_t lda2;  /* optional second K source for k >= K1 (fp16 [M, K-K1]) */
  int64_t K1;                    /* = K when A2 == NULL; K1 % 64 == 0 */
  const void* W;  int64_t ldw;   /* fp16 [N, K1] (conv: [N][tap=ky*3+kx][Cin]) */
  const void* W2; int64_t ldw2;  /* fp16 [N, K-K1] */
  int32_t a_mode;                /* TB_A_LINEAR | TB_A_CONV3X3 */
  /* conv gather geometry: output pixel (b,y,x) of an Hout x Wout map, tap (ky,kx):
   *   plain      : src = (y*stride + sign*(ky-1), x*stride + sign*(kx-1))          in Hin x Win
   *   upsample   : u = (y + ky-1, x + kx-1) in 2Hin x 2Win, src = u >> 1  (nearest x2 folded into the gather)
   *   transposed : dgrad of a stride-2 conv: src = ((y+1-ky)/2, (x+1-kx)/2) where both are even & in range */
  int32_t B, Hin, Win, Cin, Hout, Wout, stride, sign, upsample, transposed;
  /* epilogue: v = alpha*acc + bias[n] + rowbias[(m / rows_per_group)*N + n] + R[m,n]; v = act(v) */
  float alpha;
  const float* bias;             /* fp32 [N] or NULL */
  const float* rowbias;          /* fp32 [M/rows_per_group, N] or NULL (ResnetBlock2D time_emb_proj term) */
  int64_t rows_per_group;
  const void* R; int64_t ldr; int32_t r_dtype; /* residual, fp16 or fp32, or NULL */
  int32_t act;                   /* GEGLU: W rows interleaved in 32-row blocks [h|g]; C is [M, N/2] */
  void* C; int64_t ldc; int32_t c_dtype;
  void* C2; int64_t ldc2;        /* GEGLU only: raw (pre-gate) fp16 [M,N] in packed column order, or NULL */
  int32_t split_k;               /* reserved, must be 0 or 1 */
} tb_gemm_desc;

int tb_gemm(const tb_gemm_desc* d, tb_stream_t stream);

/* ---- GroupNorm(+SiLU) over NHWC fp16 [B, HW, C] (row stride ld), G groups, fp32 statistics -------
 * Replaces torch.nn.GroupNorm + F.silu inside diffusers ResnetBlock2D / Transformer2DModel /
 * conv_norm_out (train_textboost.py:1063-1067) and their autograd (:1108); gamma/beta are frozen so only
 * the input gradient is produced.  stats = [B, G, 2] (mean, rstd) written by fwd, read by bwd.
 * ws = tb_groupnorm_ws_floats(...) floats of scratch.  bwd: dx = GN'(dy) (+ add). */
int64_t tb_groupnorm_ws_floats(int B, int HW, int C, int G);
int tb_groupnorm_fwd(const void* x, int64_t ldx, void* y, int64_t ldy, const float* gamma, const float* beta,
                     float* stats, float* ws, int B, int HW, int C, int G, float eps, int silu, tb_stream_t stream);
int tb_groupnorm_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* gamma, const float* beta,
                     const float* stats, const void* add, int64_t ldadd, void* dx, int64_t lddx, float* ws,
                     int B, int HW, int C, int G, int silu, tb_stream_t stream);

/* ---- LayerNorm over rows [M, C]; x fp16 (UNet) or fp32 (CLIP residual stream), y fp16 -----------
 * Replaces torch.nn.LayerNorm in diffusers BasicTransformerBlock and transformers CLIPEncoderLayer /
 * final_layer_norm.  stats = [M, 2] (mean, rstd).  bwd: dx = LN'(dy) (+ add), dx/add in x's dtype. */
int tb_layernorm_fwd(const void* x, int64_t ldx, int x_dtype, void* y, int64_t ldy, const float* gamma, const float* beta,
                     float* stats, int64_t M, int C, float eps, tb_stream_t stream);
int tb_layernorm_bwd(const void* dy, int64_t lddy, int dy_dtype, const void* x, int64_t ldx, int x_dtype,
                     const float* gamma, const float* stats, const void* add, int64_t ldadd, void* dx, int64_t lddx,
                     int64_t M, int C, tb_stream_t stream);

/* ---- Flash attention, O = softmax(scale * Q K^T [+ causal]) V, and its backward -------------------
 * Replaces F.scaled_dot_product_attention in diffusers AttnProcessor2_0 (UNet attn1/attn2) and the eager
 * bmm-softmax-bmm of transformers CLIPAttention, plus their autograd.  Q/K/V/O are column slices of
 * row-major [B*S, *] fp16 buffers: head h occupies columns [h*hd, (h+1)*hd) from the given pointer.
 * hd % 8 == 0, hd <= 160.  LSE/Delta are fp32 [B, H, Sq]. */
typedef struct tb_attn_desc {
  int32_t B, H, Sq, Skv, hd, causal;
  float scale;
  const void* Q; int64_t ldq;
  const void* K; int64_t ldk;
  const void* V; int64_t ldv;
  void* O; int64_t ldo;
  float* LSE;
  /* backward only */
  const void* dO; int64_t lddo;
  float* Delta;
  void* dQ; int64_t lddq;
  void* dK; int64_t lddk;
  void* dV; int64_t lddv;
} tb_attn_desc;
int tb_attention_fwd(const tb_attn_desc* d, tb_stream_t stream);
int tb_attention_bwd(const tb_attn_desc* d, tb_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
