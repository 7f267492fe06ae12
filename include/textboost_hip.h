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
/* TB_F16 = the 16-bit float of the BUILD: IEEE half in libtextboost_hip.so, bfloat16 in libtextboost_hip_bf16.so (the same sources compiled with
 * -DTB_BF16: the reference's --mixed_precision bf16, train_textboost.py:298-308 / :930-934).  Every "fp16" in the comments below reads "the
 * build's half type"; fp32 accumulation, statistics, losses and the optimizer are identical in both builds. */
enum { TB_F16 = 0, TB_F32 = 1 };
enum { TB_ACT_NONE = 0, TB_ACT_QUICK_GELU = 1, TB_ACT_GEGLU = 2, TB_ACT_SILU = 3, TB_ACT_QUICK_GELU_GRAD = 4, TB_ACT_GELU = 5,
       TB_ACT_GELU_GRAD = 6, TB_ACT_GEGLU_GRAD = 7,
       /* LayerNorm fused into the epilogue of a Linear whose tile spans the whole output row (N = 320: the 64x64-map transformer blocks of
        * diffusers BasicTransformerBlock, train_textboost.py:1063-1067 / :1108); only where tb_gemm_ln_epilogue_ok() says so:
        *   LN_FWD: t = alpha*acc + bias + R -> C (fp16);  ln_stats[m] = (mean, rstd) of the fp16 row t;  C2 = fp16 LN(t) = (t - mean) rstd ln_gamma + ln_beta
        *           (the producer of the residual stream writes the next layer's normalised input: no tb_layernorm_fwd launch, t is not re-read)
        *   LN_BWD: g = (alpha*acc) ln_gamma, xhat = (C2[m,n] - mean) rstd with (mean, rstd) = ln_stats[m] (READ), C2 = the LayerNorm's fp16 INPUT (READ);
        *           C = fp16( rstd (g - mean_n g - xhat mean_n(g xhat)) + R )  -- tb_layernorm_bwd applied to the dgrad GEMM's fp32 accumulators */
       TB_ACT_LN_FWD = 8, TB_ACT_LN_BWD = 9 };
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
   *   plain      : src = (y*stride + sign*(ky-1) + shift, x*stride + sign*(kx-1) + shift)   in Hin x Win
   *                (shift = 1, stride = 2: the VAE's Downsample2D, F.pad(x, (0,1,0,1)) + conv(stride 2, padding 0))
   *   upsample   : u = (y + ky-1, x + kx-1) in 2Hin x 2Win, src = u >> 1  (nearest x2 folded into the gather)
   *   transposed : dgrad of a stride-2 conv: src = ((y+1-ky)/2, (x+1-kx)/2) where both are even & in range */
  int32_t B, Hin, Win, Cin, Hout, Wout, stride, sign, upsample, transposed, shift;
  /* epilogue: v = alpha*acc + bias[n] + rowbias[(m / rows_per_group)*N + n] + R[m,n]; v = act(v) */
  float alpha;
  const float* bias;             /* fp32 [N] or NULL */
  const float* rowbias;          /* fp32 [M/rows_per_group, N] or NULL (ResnetBlock2D time_emb_proj term) */
  int64_t rows_per_group; int64_t ldrb; /* row stride of rowbias (>= N) */
  const void* R; int64_t ldr; int32_t r_dtype; /* residual, fp16 or fp32, or NULL */
  int32_t act;                   /* GEGLU: W rows interleaved in 32-row blocks [h|g]; C is [M, N/2] */
  void* C; int64_t ldc; int32_t c_dtype;
  void* C2; int64_t ldc2;        /* fp16 [M,N] aux: GEGLU: raw pre-gate output (packed order), written;
                                  * QUICK_GELU / GELU: pre-activation, written (if non-NULL);
                                  * QUICK_GELU_GRAD / GELU_GRAD: pre-activation, READ: v *= act'(C2[m,n]);
                                  * GEGLU_GRAD: packed pre-gate [M,2N], READ; C becomes d(proj) [M,2N] fp16 (packed) */
  void* ws; int64_t ws_bytes;    /* optional scratch: lets small-M / long-K problems split K over blocks (fp32 partials,
                                  * fixed-order reduction => deterministic); NULL disables */
  /* TB_ACT_LN_FWD / TB_ACT_LN_BWD only (torch.nn.LayerNorm of BasicTransformerBlock fused into the neighbouring Linear) */
  const float* ln_gamma; const float* ln_beta; /* fp32 [N]; ln_beta unused by LN_BWD */
  float* ln_stats;               /* fp32 [M, 2] (mean, rstd): written by LN_FWD, read by LN_BWD */
  float ln_eps;
  /* optional: sync_count uint32 counters, ZERO before the first call and left zero by every call (one buffer per stream that runs tb_gemm
   * concurrently): a split-K launch (see `ws`) with at most sync_count output tiles then adds its k-slices inside the kernel -- the slice that
   * arrives last at its tile's counter reduces, same arithmetic and order as the separate reducer -- instead of launching a second kernel.
   * Opt-in (tb_gemm_set_variant(9801)): measured SLOWER than the reducer launch (the partials must be written and read coherently) */
  uint32_t* sync; int64_t sync_count;
  /* optional HOST pointer (round 4): when non-NULL, a launch that splits K leaves its fp32 partials in `ws` -- layout [S][M][Npad], Npad = N
   * rounded up to 8, the epilogue NOT applied and C not written -- and writes S to *split_out instead of running the reducer; the caller then
   * hands (ws, S, Npad, bias, rowbias, R) to a consumer that adds the slices itself (tb_groupnorm_fwd_splitk / tb_groupnorm_bwd_splitk: the
   * reduction, the conv epilogue and the GroupNorm in one launch).  *split_out = 1 means the launch was not split and C holds the result as
   * usual.  Honoured for act NONE, fp16 C, alpha == 1 and no C2; never for the phase-ordered stride-2 dgrad. */
  int32_t* split_out;
  /* LayerNorm folded into the Linear that CONSUMES it (round 5; torch.nn.LayerNorm norm1 / norm2 / norm3 of diffusers BasicTransformerBlock in front
   * of attn1.to_q/k/v, attn2.to_q and ff.net.0.proj, train_textboost.py:1063-1067; gamma / beta are frozen):
   *     LN(x) W^T + b  =  rstd (x W'^T - mean c1) + c2,   W' = gamma (.) W (fp16, packed once by the host), c1[n] = sum_k W'[n,k], c2 = b + W beta
   * so the consumer multiplies the RAW residual stream and the LayerNorm is two per-row scalars in its epilogue -- no LayerNorm launch, no
   * normalised copy in memory.  The row statistics come from the launch that PRODUCES x:
   *   rs_out (producer): fp32 [M][rs_ld][2]; the launch writes, for its column tile tn, slot tn of row m = (sum, sum of squares) of the fp16-ROUNDED
   *          output row over the tile's columns.  8-wave Linear tiles only (act NONE, fp16 C; tb_gemm8_last tells the tile width = N / slots in use);
   *          tb_gemm returns -22 when the launch would take another kernel, when its column tiles do not fit a row of rs_out (> rs_ld), or
   *          when rs_n > 0 and the launch would fill another number of slots than the rs_n its consumer is going to sum.
   *   rs_in  (consumer): the producer's rs_out with rs_n slots per row in use (row stride rs_ld slots).  Per row: mean = S / K, var = Q / K - mean^2
   *          (K = the LayerNorm width), rstd = rsqrt(var + ln_eps); epilogue v = rstd (alpha acc - mean ln_gamma[n]) + bias[n] (+ R) for act NONE,
   *          the same in front of the gate for GEGLU; ln_gamma = c1 (fp32 [N], in W's row order), bias = c2.  When ln_stats != NULL the column
   *          tile 0 of every row panel writes (mean, rstd) to ln_stats[m] for tb_layernorm_bwd.  rs_n <= 16; no second K source. */
  float* rs_out; const float* rs_in; int64_t rs_ld; int32_t rs_n;
} tb_gemm_desc;

int tb_gemm(const tb_gemm_desc* d, tb_stream_t stream);
/* 1 when a fp16 Linear of this shape takes a tile that spans the whole output row, i.e. TB_ACT_LN_FWD / TB_ACT_LN_BWD are available for it
 * (N == 320, K % 64 == 0, M a multiple of the tile height with at least one chip round of tiles); tb_gemm returns -22 for them otherwise */
int tb_gemm_ln_epilogue_ok(int64_t M, int64_t N, int64_t K);
/* 1 when the three LayerNorms of a BasicTransformerBlock of width C over M rows can be folded into the Linear layers behind them
 * (tb_gemm_desc.rs_out / rs_in): the residual-stream producers (M x C x C) write row statistics and qkv (M x 3C x C), attn2.to_q (M x C x C) and
 * the GEGLU projection (M x 8C x C) take tiles that apply the fold; *slots = statistic slots per row the producers fill (the consumers' rs_n) */
int tb_gemm_lnfold_ok(int64_t M, int64_t C, int* slots);
/* 1 when the pair nearest-x2 upsampling + conv3x3 (diffusers Upsample2D, up_blocks.*.upsamplers.0; train_textboost.py:1063-1067 forward, :1108
 * backward) over a coarse [B, Hc, Wc, Cin] map can run as four 2 x 2-tap SUB-PIXEL convolutions with pre-summed weights (2.25x fewer FLOP, the 4x
 * map is never written): tb_gemm with a_mode = CONV3X3 and
 *   upsample = 2 (forward): A = coarse [B, Hin, Win, Cin], C = fine fp16 [B, 2 Hin, 2 Win, N], K = 4 Cin, W = fp16 [4 classes][N][4 taps][Cin]
 *                           (class 2 py + px = parity of the output pixel, tap 2 a + c = window row / column; ldw = 4 Cin), bias optional;
 *   upsample = 3 (dgrad):   A = fine gradient [B, Hin, Win, Cin], C = coarse fp16 [B, Hin / 2, Win / 2, N], K = 16 Cin,
 *                           W = fp16 [N][4 views][4 taps][Cin].
 * The fp16 rounding of the summed filter rows is the one stated divergence from the 9-tap arithmetic (tests bound it). */
int tb_gemm_subpixel_ok(int B, int Hc, int Wc, int Cin, int N);
/* tuning knob for the k-tile / pipeline-depth variant of tb_gemm (returns the previous value); 0 is the default */
int tb_gemm_set_variant(int v);
/* {BM, BN, a_mode, k_tile*10 + stages, split_k} of the most recent tb_gemm launch (profiling aid) */
void tb_gemm_last_config(int* out5);
/* 8-wave wide-tile path of tb_gemm (csrc/gemm8.hip; 256-pixel x 160-channel halo convolutions and 128 x 320 Linear tiles for the
 * 64x64 / 32x32 / 16x16 feature maps): tb_gemm8_set(bits) -- 1 = convolutions, 2 = Linear layers, 4 = fused GEGLU epilogues, 32 = split-K for
 * small conv grids (default 39); A/B switches: 8 = 64x320 instead of 128x320 Linear tiles, 16 = GEGLU layers on 128x320 tiles, 64 = no XCD rectangle
 * cut, 128 = no 128x160 tiles, 256 = 128x160 tiles also for N = 320, 1024 = no 256x128 convolution tiles (the VAE's 128 / 256 / 512-channel convolutions stay on the 4-wave halo kernel), 512 = 128x160 tiles from 256 (not 512) tiles on (the 32x32-map N = 640 layers: measured +0.27 ms), 2048 = no one-per-CU 128x80 / 128x160 four-stage Linear tiles (the 16x16- / 32x32-map layers with K <= 2560), 4096 = no 128x160 one only, 8192 = the long-K 16x16-map Linear layers (K >= 4096) stay on the 4-wave kernel's 128x128 split-K launches instead of two k-slices of 128x160 tiles, 131072 = CLIP's wide Linear layers on ragged 128x128 tiles (round 5, opt-in), 524288 = the 32x32-map convolutions as two k-slices of the 256x160 tile (round 6, measured +0.15 ms), 1048576 = the 256x160 / 256x128 convolution tiles as one instruction stream per wave (round 6: -4..6 % isolated, no gain in the sustained step -- power-limited; bit-equal results) -- returns the previous value;
 * tb_gemm8_last returns 1 when the most recent tb_gemm launched gemm8_kernel<WM, WN, MT, NT, CONV, NS> and writes those SIX ints to out5 */
int tb_gemm8_set(int bits);
int tb_gemm8_last(int* out5);
/* profiling aid: device buffer of 16 x uint64 (NULL = off) that receives s_memtime stamps of the first / last workgroup of gemm8 launches */
int tb_gemm8_debug(void* stamps16);

/* ---- fused GEGLU feed-forward of diffusers BasicTransformerBlock (ff.net.0.proj -> h * gelu(g) -> ff.net.2; train_textboost.py:1063-1067 forward,
 * :1108 backward) for the C = 320 transformer blocks of the 64x64 maps (csrc/ff_fused.hip): one launch per direction, the 128-row input tile stays
 * in registers, the 4C-wide intermediate never goes to memory (forward: the gated tensor; backward: d(proj)); only the packed pre-gate
 * projections HG (what tb_gemm's TB_ACT_GEGLU writes to C2, same [h32 | g32] packing) are written by the forward and read by the backward.
 *   tb_ff_fwd: HG = X W1^T + b1 (fp16);  Y = fp16((h * gelu(g)) W2^T + b2 + R)            X = LayerNorm output [M, C]
 *              W1 = ff.net.0.proj.weight with rows packed [h32 | g32] ([2 inner, C]), W2 = ff.net.2.weight [C, inner]
 *   tb_ff_bwd: du = X W1^T;  dh = du gelu(g), dg = du h gelu'(g);  Y = fp16([dh | dg] W2^T + R)   X = d(out) [M, C]
 *              W1 = ff.net.2.weight transposed [inner, C], W2 = the packed ff.net.0.proj.weight transposed [C, 2 inner]
 * Same arithmetic as the two tb_gemm launches it replaces (fp16 operands, fp32 accumulation, the gate evaluated on the fp16-rounded projections);
 * only the fp32 summation order of the second product differs.  tb_ff_fused_ok: M % 128 == 0, C == 320, inner == 1280. */
typedef struct tb_ff_desc {
  int64_t M; int32_t C, inner;
  const void* X; int64_t ldx;      /* fp16 [M, C], 16-byte aligned rows */
  const void* W1; int64_t ldw1;    /* fp16, see above */
  const void* W2; int64_t ldw2;
  const float* b1; const float* b2;/* forward: fp32 [2 inner] (packed like W1's rows) / [C], or NULL; backward: ignored */
  void* HG; int64_t ldhg;          /* fp16 [M, 2 inner]: written by tb_ff_fwd, read by tb_ff_bwd */
  const void* R; int64_t ldr;      /* optional fp16 [M, C] added to Y */
  void* Y; int64_t ldy;            /* fp16 [M, C] */
  /* tb_ff_bwd only (round 5): the LayerNorm backward of norm3 applied to the accumulators -- Y = tb_layernorm_bwd(dy = [dh | dg] W2^T, ln_x, ln_gamma,
   * ln_stats) + R instead of Y = dy + R, i.e. the block's d(residual stream) directly (the tile spans the 320-wide row).  NULL ln_x: off. */
  const void* ln_x; int64_t ld_lnx;   /* fp16 [M, C]: the LayerNorm's input */
  const float* ln_stats;              /* fp32 [M, 2] (mean, rstd) */
  const float* ln_gamma;              /* fp32 [C] */
  /* tb_ff_fwd only (round 6): the row-local neighbours of the feed-forward in the SAME launch (a 128-row tile spans the 320-wide rows of all of them).
   *   pre_W != NULL: X is the input of attn2.to_out (the cross-attention output); the launch first computes
   *       t2 = X pre_W^T + pre_b (+ pre_R), stores it to pre_Y (the residual stream: set R = pre_Y for the feed-forward's own residual),
   *       l3 = LayerNorm(t2; pre_gamma, pre_beta, pre_eps) with (mean, rstd) to pre_stats -- and feeds l3 to ff.net.0.proj straight from the LDS:
   *       diffusers BasicTransformerBlock: attn2.to_out.0 + residual, norm3, ff  (train_textboost.py:1063-1067).
   *   post_W != NULL: the block's output t3 = ff(...) + b2 + R goes through one more Linear before it leaves the launch,
   *       post_Y = t3 post_W^T + post_b (+ post_R): Transformer2DModel.proj_out + the block input (1x1 conv = Linear over NHWC rows);
   *       Y may then be NULL (nothing reads t3: no weight gradients exist in this path).
   * Weights fp16 [C, C] row-major (out, in), 16-byte aligned rows.  Same arithmetic as the launches this replaces: every intermediate is rounded to
   * fp16 where they stored it; the LayerNorm statistics are two-pass fp32 over the fp16-rounded t2. */
  const void* pre_W; int64_t ld_prew; const float* pre_b; const void* pre_R; int64_t ld_prer; void* pre_Y; int64_t ld_prey;
  const float* pre_gamma; const float* pre_beta; float* pre_stats; float pre_eps;
  const void* post_W; int64_t ld_postw; const float* post_b; const void* post_R; int64_t ld_postr; void* post_Y; int64_t ld_posty;
} tb_ff_desc;
int tb_ff_fused_ok(int64_t M, int C, int inner);
int tb_ff_fwd(const tb_ff_desc* d, tb_stream_t stream);
int tb_ff_bwd(const tb_ff_desc* d, tb_stream_t stream);
/* Linear -> LayerNorm -> Linear on 128-row tiles of a C = 320 residual stream in ONE launch (round 6, csrc/chain320.hip): the pairs
 * proj_in -> norm1 -> attn1.to_q|to_k|to_v and attn1.to_out.0 + residual -> norm2 -> attn2.to_q of diffusers Transformer2DModel /
 * BasicTransformerBlock on the SD1.x 64x64 maps (train_textboost.py:1063-1067); replaces tb_gemm(TB_ACT_LN_FWD) + tb_gemm.
 *     T = X W1^T + b1 (+ R1)                      fp16 [M, 320], stored (the residual stream; NULL: not stored)
 *     stats[m] = (mean, rstd) of the fp16 row T[m]   two-pass, fp32 (NULL: not stored)
 *     Y = LayerNorm(T; gamma, beta, eps) W2^T + b2   fp16 [M, N2];  LayerNorm(T) itself never exists in memory
 * W1 fp16 [320, 320], W2 fp16 [N2, 320] row-major (out, in), rows 16-byte aligned.  tb_chain320_ok: M % 128 == 0, N2 a multiple of 320 (<= 2560). */
typedef struct tb_chain_desc {
  int64_t M;
  const void* X; int64_t ldx;
  const void* W1; int64_t ldw1; const float* b1; const void* R1; int64_t ldr1;
  void* T; int64_t ldt;
  const float* gamma; const float* beta; float eps; float* stats;
  const void* W2; int64_t ldw2; int32_t N2; const float* b2;
  void* Y; int64_t ldy;
} tb_chain_desc;
int tb_chain320_ok(int64_t M, int N2);
int tb_chain320(const tb_chain_desc* d, tb_stream_t stream);
/* profiling aid (builds with -DFF_PROF=1 only): device buffer of 16 x uint64 (NULL = off) that receives per-phase s_memtime sums of waves 0 and 4 of workgroup 0 */
int tb_ff_debug(void* buf16);

/* measurement aid (bench.py `roofline.sustained_peak`): `blocks` workgroups of 4 waves issue iters * 16 independent v_mfma_f32_32x32x16_f16 each on
 * random register operands; FLOP = blocks * 4 * iters * 16 * 32768.  out: blocks * 256 floats (sink). */
int tb_mfma_peak_probe(float* out, int blocks, int iters, tb_stream_t stream);

/* text of the HIP error behind the most recent -5 (launch failure) return; diagnostics only */
const char* tb_last_hip_error(void);

/* ---- GroupNorm(+SiLU) over NHWC fp16 [B, HW, C] (row stride ld), G groups, fp32 statistics -------
 * Replaces torch.nn.GroupNorm + F.silu inside diffusers ResnetBlock2D / Transformer2DModel /
 * conv_norm_out (train_textboost.py:1063-1067) and their autograd (:1108); gamma/beta are frozen so only
 * the input gradient is produced.  stats = [B, G, 2] (mean, rstd) written by fwd, read by bwd.
 * ws = tb_groupnorm_ws_floats(...) floats of scratch.  bwd: dx = GN'(dy) (+ add). */
int64_t tb_groupnorm_ws_floats(int B, int HW, int C, int G);
/* A/B knob: 1 (default) = small maps with 8-aligned groups run one-pass per-(image, group) kernels, 0 = always the two-pass kernels */
int tb_groupnorm_set_variant(int fused);
int tb_groupnorm_fwd(const void* x, int64_t ldx, void* y, int64_t ldy, const float* gamma, const float* beta,
                     float* stats, float* ws, int B, int HW, int C, int G, float eps, int silu, tb_stream_t stream);
int tb_groupnorm_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* gamma, const float* beta,
                     const float* stats, const void* add, int64_t ldadd, void* dx, int64_t lddx, float* ws,
                     int B, int HW, int C, int G, int silu, tb_stream_t stream);
/* GroupNorm straight off the fp32 split-K partials of the convolution that produces its input (tb_gemm_desc.split_out): one workgroup per
 * (image, group) adds the S slices in slice order, applies the convolution's epilogue -- x = fp16(((sum + bias[n]) + R[m,n]) + rowbias[b,n]), the
 * arithmetic of tb_gemm's reducer bit for bit -- writes x (the tensor the backward and the skip connections read), and normalises from registers.
 * Replaces the reducer launch + tb_groupnorm_fwd of ResnetBlock2D's conv1 -> norm2 and conv2 -> next norm1 on the 16x16 / 8x8 maps.
 * Available when tb_groupnorm_splitk_ok (the one-pass per-(image, group) kernels: C / G a multiple of 8, the slice fits a workgroup's registers).
 * part: fp32 [S][B*HW][npad]; bias fp32 [C] or NULL; rowbias fp32 [B, ldrb] or NULL; R fp16 [B*HW, ldr] or NULL. */
int tb_groupnorm_splitk_ok(int B, int HW, int C, int G);
int tb_groupnorm_fwd_splitk(const float* part, int S, int64_t npad, const float* bias, const float* rowbias, int64_t ldrb, const void* R,
                            int64_t ldr, void* x, int64_t ldx, void* y, int64_t ldy, const float* gamma, const float* beta, float* stats,
                            int B, int HW, int C, int G, float eps, int silu, tb_stream_t stream);
/* backward twin: dy = fp16(sum of the dgrad convolution's slices) never goes to memory; dx = GN'(dy) (+ add) */
int tb_groupnorm_bwd_splitk(const float* part, int S, int64_t npad, const void* x, int64_t ldx, const float* gamma, const float* beta,
                            const float* stats, const void* add, int64_t ldadd, void* dx, int64_t lddx, int B, int HW, int C, int G,
                            int silu, tb_stream_t stream);

/* ---- LayerNorm over rows [M, C]; x fp16 (UNet) or fp32 (CLIP residual stream), y fp16|fp32 ----
 * Replaces torch.nn.LayerNorm in diffusers BasicTransformerBlock and transformers CLIPEncoderLayer /
 * final_layer_norm.  stats = [M, 2] (mean, rstd).  bwd: dx = LN'(dy) (+ add), dx/add in x's dtype. */
int tb_layernorm_fwd(const void* x, int64_t ldx, int x_dtype, void* y, int64_t ldy, int y_dtype, const float* gamma,
                     const float* beta, float* stats, int64_t M, int C, float eps, tb_stream_t stream);
/* same, plus the rank-R LoRA down projection of the normalised row while it is in registers (peft lora_A on q/k/v,
 * train_textboost.py:700-722): t[m, j] = sum_k y[m,k] * fp16(loraA[j*C + k]), j < R (what tb_lora_down computes from y) */
int tb_layernorm_lora_fwd(const void* x, int64_t ldx, int x_dtype, void* y, int64_t ldy, int y_dtype, const float* gamma,
                          const float* beta, float* stats, int64_t M, int C, float eps, const float* loraA, int R, void* t /* fp16 */,
                          int64_t ldt, tb_stream_t stream);
/* same, with the down projection only for rows < lora_rows (a frozen batch without adapters behind them: the KPL teacher rows) */
int tb_layernorm_lora_rows_fwd(const void* x, int64_t ldx, int x_dtype, void* y, int64_t ldy, int y_dtype, const float* gamma,
                               const float* beta, float* stats, int64_t M, int C, float eps, const float* loraA, int R, void* t /* fp16 */,
                               int64_t ldt, int64_t lora_rows, tb_stream_t stream);
/* dx16 (optional, may be NULL): an fp16 copy of dx for the next dgrad GEMM (saves a conversion pass over the fp32 stream) */
int tb_layernorm_bwd(const void* dy, int64_t lddy, int dy_dtype, const void* x, int64_t ldx, int x_dtype,
                     const float* gamma, const float* stats, const void* add, int64_t ldadd, void* dx, int64_t lddx,
                     void* dx16, int64_t lddx16, int64_t M, int C, tb_stream_t stream);

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
  float* ws; int64_t ws_floats;  /* optional scratch (n * 2*B*Skv*H*hd floats, n >= 2): lets the dK/dV kernel split the query
                                  * range over up to n blocks per key block (per-slice fp32 partials summed in a fixed order)
                                  * when Skv is too short to fill the chip (cross-attention).  With >= 2*B*H*Sq floats it also enables the
                                  * LDS-DMA staged dK/dV kernel of the hd = 40 / 64 / 80 self-attention shapes (Sq % 64 == 0, Skv % 128 == 0,
                                  * >= 512 key blocks): the dQ kernel publishes -lse*log2(e) and -delta there for it.  The two uses exclude
                                  * each other (the split needs < 512 key blocks) */
  void* fp8_ws; int64_t fp8_ws_bytes;  /* forward only, opt-in (BASELINE.json configs[4]): non-NULL with >= tb_attention_fp8_ws_bytes(B, H, Skv)
                                        * bytes makes the hd = 40 self-attention forward (Sq % 256 == 0, Skv % 256 == 0, non-causal) compute
                                        * P V with e4m3 operands on v_mfma_scale_f32_32x32x64_f8f6f4 (P rounded in registers, V through a
                                        * per-(batch, head) scaled transposed image in this workspace); other shapes ignore it.  Tolerance
                                        * against fp32 attention: rel-L2 <= 5e-2 on O (3.6e-2 measured on random data, tests/test_gpu_norm_attn.py) vs 2e-3 for the fp16 path */
} tb_attn_desc;
int64_t tb_attention_fp8_ws_bytes(int B, int H, int Skv);
int tb_attention_fwd(const tb_attn_desc* d, tb_stream_t stream);
int tb_attention_bwd(const tb_attn_desc* d, tb_stream_t stream);
/* A/B knob (default 1): bit 0 = LDS-DMA staged forward kernel for the hd = 40 self-attention shape, 2 = also hd = 80, 4 = no XCD block remap in
 * the forward, 64 = XCD remap in the backward, 128 / 256 = register-staged dK/dV / dQ kernels instead of the DMA ones, 512 = DMA backward only
 * for hd = 40, 1024 / 2048 = the LDS-DMA forward / dK-dV kernels instead of the software-pipelined ones (csrc/attention_il.hip) for the hd = 40
 * 64x64-map shape, 4096 = the software-pipelined dQ kernel (opt-in), 8192 = the generic dQ + dK/dV launches also for short hd = 64 sequences (the CLIP
 * encoder's 77 x 77 attention otherwise takes csrc/attention_small.hip: the whole backward in one launch), 16384 = the general flash forward also for
 * short key sequences (cross-attention on the prompt, 33 .. 96 keys, otherwise attn_xs_fwd_kernel / attn_xs_bwd_dq_kernel), 32768 = the query-split dK / dV launch of short key sequences cuts until 1024 (not 512)
 * workgroups exist; returns the previous value */
int tb_attention_set_variant(int bits);

/* ---- scheduler / boundary / loss / misc streaming kernels ------------------------------------------ */
/* noise_scheduler.add_noise (:1052) [+ get_velocity (:1073) when velocity != NULL]; fp32 NCHW in, fp16 noisy out */
int tb_add_noise(const float* x0, const float* noise, const int64_t* timesteps, const float* alphas_cumprod, void* noisy,
                 float* velocity, int B, int64_t per_sample, tb_stream_t stream);
/* diffusers Timesteps(dim, flip_sin_to_cos=True, freq_shift=0): out fp16 [B, dim] = [cos | sin] */
int tb_timestep_embed(const int64_t* timesteps, void* out, int B, int dim, tb_stream_t stream);
/* 3x3 conv whose NCHW side has 4 channels -> NHWC fp16 [B*H*W, Cout]: UNet conv_in forward (sign=+1) and conv_out
 * input-gradient (sign=-1).  w_packed fp32 [(tap*4 + c4)*Cout + co]. */
int tb_conv4_to_nhwc(const void* in, int in_dtype, const float* w_packed, const float* bias, void* out, int64_t ldo, int B,
                     int H, int W, int Cout, int sign, float in_scale, tb_stream_t stream);
/* same with Cin = 3 or 4 input channels (w_packed fp32 [(tap*Cin + ci)*Cout + co]): the VAE encoder's conv_in on RGB pixels */
int tb_convin_to_nhwc(const void* in, int in_dtype, int Cin, const float* w_packed, const float* bias, void* out, int64_t ldo, int B,
                      int H, int W, int Cout, int sign, float in_scale, tb_stream_t stream);
/* UNet conv_out forward: NHWC fp16 [B*H*W, C] -> NCHW fp16 [B,4,H,W]; w_packed fp32 [4][9][C] */
int tb_conv_to4(const void* in, int64_t ldi, const float* w_packed, const float* bias, void* out, int B, int H, int W, int C,
                tb_stream_t stream);
/* the three boundary convolutions above run on the matrix cores when the map width is a multiple of 16 and the channel count of 32 (16-pixel
 * MFMA tiles; the fp32 pack is consumed in the library's 16-bit type, exact for a model cast by unet.to(fp16), fp32 accumulation): 1 (default);
 * 0 = the fp32 VALU kernels for every shape.  Returns the old value (A/B and test knob). */
int tb_boundary_conv_set_variant(int v);
/* ---- VAE encoder pieces (vae.encode(x).latent_dist.sample() * scaling_factor, train_textboost.py:1036-1037) ----
 * row softmax of fp32 scores -> fp16 probabilities (single-head 512-channel mid-block attention runs as GEMM, softmax, GEMM) */
int tb_softmax_rows(const float* scores, int64_t lds, void* probs /* fp16 */, int64_t ldp, int64_t rows, int cols, tb_stream_t stream);
/* moments fp32 NHWC [B*HW, ldm >= 2L] = (mean | logvar) -> latents fp32 NCHW [B, L, HW] = (mean + exp(0.5 clamp(logvar,-30,20)) eps) * scale */
int tb_vae_sample(const float* moments, int64_t ldm, const float* eps, float* latents, int B, int HW, int L, float scale,
                  tb_stream_t stream);
/* ---- validation / inference sampling (log_validation train_textboost.py:453-531, inference.py:73-100) ----
 * out[b,o,p] = scale * sum_c W[o,c] in[b,c,p] + bias[o] on NCHW fp32, C <= 8: AutoencoderKL.post_quant_conv with latents / scaling_factor */
int tb_chan_mix(const float* in, const float* W, const float* bias, float* out, int B, int C, int HW, float scale, tb_stream_t stream);
/* one DPMSolverMultistepScheduler(dpmsolver++, order 2) step with classifier-free guidance: eps2 = UNet output fp16 [2B, n] (uncond | cond);
 * eps = e_u + g (e_c - e_u); m0 = (x - sigma_t eps) / alpha_t; x = ca x + cb m0 + cc m_prev; m_prev = m0; x2 = fp16 (x | x) for the next call */
int tb_dpm_step(float* x, const void* eps2, float* m_prev, void* x2, int64_t n_per_b, int B, float guidance, float alpha_t, float sigma_t,
                float ca, float cb, float cc, tb_stream_t stream);
/* (decoded / 2 + 0.5).clamp(0, 1): NHWC fp32 [B*HW, ld >= C] -> NCHW fp32 [B, C, HW] */
int tb_vae_image(const float* decoded, int64_t ld, float* image, int B, int HW, int C, tb_stream_t stream);
/* F.mse_loss(pred.float(), target.float()).mean() (:1085-1090); dpred = loss_scale[0] * dloss/dpred (fp32) */
int tb_mse_loss(const void* pred, const float* target, float* dpred, float* loss_out, const float* loss_scale, int64_t N,
                float* ws /* >= 128 floats of scratch for the two-stage reduction, or NULL: one block */, tb_stream_t stream);
/* knowledge-preservation loss, cos variant (:1099-1106): loss_out = mean_rows(1 - cos(h, h0));
 * dh = weight * loss_scale[0] * dloss/dh.  partial = M floats scratch. */
int tb_kpl_cos(const float* h, int64_t ldh, const void* h0, int64_t ldh0, int h0_dtype, float* dh, int64_t lddh,
               float* partial, float* loss_out, const float* loss_scale, float weight, int64_t M, int D, tb_stream_t stream);
/* backward of diffusers GEGLU on the packed [h32|g32] layout tb_gemm(TB_ACT_GEGLU) saved in C2 */
/* knowledge-preservation loss, mse variant (--kpl_type mse, :1104-1105): loss_out = mean((h - h0)^2) */
int tb_kpl_mse(const float* h, int64_t ldh, const void* h0, int64_t ldh0, int h0_dtype, float* dh, int64_t lddh,
               float* partial, float* loss_out, const float* loss_scale, float weight, int64_t M, int D, tb_stream_t stream);
int tb_geglu_bwd(const void* dout, int64_t lddo, const void* raw, int64_t ldr, void* dproj, int64_t lddp, int64_t M,
                 int inner, tb_stream_t stream);
/* backward of F.interpolate(scale 2, nearest): dx[b,y,x,:] = sum of the 2x2 block of du (NHWC fp16) */
int tb_pool2x2_sum(const void* du, int64_t ldu, void* dx, int64_t ldx, int B, int H, int W, int C, tb_stream_t stream);
/* F.interpolate(scale_factor=2, mode="nearest") of diffusers Upsample2D, materialised: u[B, 2H, 2W, C] from x[B, H, W, C] (NHWC fp16) */
int tb_upsample2x(const void* x, int64_t ldx, void* u, int64_t ldu, int B, int H, int W, int C, tb_stream_t stream);
int tb_add_f16(const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t ldo, int64_t M, int C, tb_stream_t stream);
int tb_convert(const void* in, int64_t ldi, int in_dtype, void* out, int64_t ldo, int out_dtype, int64_t M, int C, float scale,
               tb_stream_t stream);

/* ---- text-encoder small kernels ------------------------------------------------------------------- */
/* CLIPTextEmbeddings: out[m] = tok[ids[m]] + pos[m % T]; (fp32 table -> fp32 out) or (fp16 -> fp16, KPL teacher) */
int tb_embed_fwd(const int64_t* ids, const void* tok, const void* pos, int table_dtype, void* out, int out_dtype, int64_t M,
                 int T, int D, tb_stream_t stream);
/* g_added[a] += sum over positions with ids == first_added + a of dh (rows below first_added get no gradient, :1109-1117) */
int tb_embed_bwd(const float* dh, const int64_t* ids, float* g_added, int64_t M, int D, int64_t first_added, int n_added,
                 tb_stream_t stream);
/* TextBoostModel.forward pins, textboost/text_encoder.py:71-86, and their (zero) gradient */
int tb_textboost_pin_fwd(void* h, int h_dtype, const int64_t* ids, const float* null_embedding, int B, int T, int D,
                         int use_fixed, int64_t eos_id, tb_stream_t stream);
int tb_textboost_pin_bwd(float* dh, const int64_t* ids, int B, int T, int D, int use_fixed, int64_t eos_id, tb_stream_t stream);
/* peft lora.Linear pieces (train_textboost.py:700-722): P adapters (q,k,v) stacked: A fp32 [P*r, K], Bcat fp32 [P*D, r] */
int tb_lora_down(const void* x, int64_t ldx, const float* A, void* t, int64_t ldt, int64_t M, int K, int R, tb_stream_t stream);
/* `layers` stacked adapters in one launch: A [layers][P*r, K], Bcat [layers][P*D, r] -> w2_fwd [layers][P*D, 64], w2_dgrad [layers][K, 64] */
int tb_lora_pack(const float* A, const float* Bcat, void* w2_fwd /*fp16 [P*D,64]*/, void* w2_dgrad /*fp16 [K,64]*/, int D, int K,
                 int r, int P, int layers, float scaling, tb_stream_t stream);
int64_t tb_lora_bwd_ws_floats(int64_t M, int D, int K, int r, int P);
int tb_lora_bwd(const void* dY, int64_t lddy, const void* x, int64_t ldx, const void* t, int64_t ldt, const float* Bcat,
                void* dt, int64_t lddt, float* dA /* += */, float* dB /* += */, float* ws, int64_t M, int D, int K, int r, int P,
                float scaling, tb_stream_t stream);
/* the same backward as one link of a chain over adapter sets of equal shape (CLIP's layers, last to first; accelerator.backward,
   train_textboost.py:1108): this set's dt / dB launch also carries the dA panels of the PENDING set (the previous link's x / dt / dA; all
   three null = none; pend_dt must not be this link's dt), da_now != 0 launches this set's own dA panels behind it (the last link).
   Per-set arithmetic and summation order are tb_lora_bwd's. */
int tb_lora_bwd_chain(const void* dY, int64_t lddy, const void* x, int64_t ldx, const void* t, int64_t ldt, const float* Bcat,
                      void* dt, int64_t lddt, float* dA /* += */, float* dB /* += */, int64_t M, int D, int K, int r, int P,
                      float scaling, const void* pend_x, int64_t pend_ldx, const void* pend_dt, int64_t pend_lddt,
                      float* pend_dA /* += */, int da_now, tb_stream_t stream);
int tb_lora_set_variant(int v);   /* profiling: 0 = the dt slabs of tb_lora_bwd load dY in loops, 1 (default) = all loads up front; returns the old value */

/* ---- optimizer tail: all scalars stay on the device in `state` (fp32[TB_ST_COUNT]) ------------------- */
enum { TB_ST_LOSS_SCALE = 0, TB_ST_GROWTH_TRACKER = 1, TB_ST_STEP = 2, TB_ST_FOUND_INF = 3, TB_ST_COEF_LORA = 4,
       TB_ST_COEF_EMB = 5, TB_ST_BC1 = 6, TB_ST_BC2 = 7, TB_ST_GRAD_NORM = 8, TB_ST_SUMSQ_LORA = 9, TB_ST_SUMSQ_EMB = 10,
       TB_ST_LOSS_MSE = 11, TB_ST_LOSS_KPL = 12,
       TB_ST_LR_MULT = 13, /* lr_scheduler (diffusers get_scheduler, :911-916): holds lambda(step) - 1, written by the host or by tb_lr_from_table, so a zeroed
                              state means the constant schedule; every group's lr = base lr * (1 + state[13]) */
       TB_ST_COUNT = 16 };
int tb_sumsq(const float* x, int64_t n, float* out, float* ws64 /* 64 floats scratch */, tb_stream_t stream);
/* GradScaler unscale/inf-check/update + clip_grad_norm_ coefficient + Adam bias corrections (:1108, :1128-1134).  grad_div >= 1 is the
 * data-parallel world size: the gradient buffers then hold the all-reduce SUM and DDP's division (:919-926) is folded into the unscale
 * coefficients state[TB_ST_COEF_*] and the reported norm -- no separate pass over the gradients */
int tb_scaler_update(float* state, float max_norm, float beta1, float beta2, float growth_factor, float backoff_factor,
                     float growth_interval, int use_scaler, float grad_div, tb_stream_t stream);
/* lr_scheduler on the device: state[TB_ST_LR_MULT] = table[min(state[TB_ST_STEP], n - 1)] - 1 (table[k] = lambda(k) of diffusers
 * get_scheduler, :911-916); indexed by the number of optimizer steps that were not skipped, like accelerate's AcceleratedScheduler */
int tb_lr_from_table(float* state, const float* table, int n, tb_stream_t stream);
/* torch.optim.AdamW step on a flat fp32 buffer; g is multiplied by state[coef_slot]; skipped when state says inf */
int tb_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float wd,
             const float* state, int coef_slot, tb_stream_t stream);
int tb_weight_decay(float* p, int64_t n, float factor, const float* state, tb_stream_t stream);
/* added-row norm clamp (:1138-1149) and row norms for mean_norm (:1017) */
int tb_renorm_rows(float* rows, int n_rows, int D, float mean_norm, float* norms, tb_stream_t stream);
int tb_row_norms(const float* w, int64_t rows, int D, float* norms, tb_stream_t stream);
/* The whole optimizer tail above (train_textboost.py:1128-1149: unscale + inf check + GradScaler.update, clip_grad_norm_, lr_scheduler, AdamW of
 * both text-encoder groups [+ the UNet adapter group], the decoupled decay of the rows that never get a gradient, the added-row norm clamp) as TWO
 * launches instead of ten (round 6; same arithmetic in the same order: bit-equal to the separate entry points, which stay as the reference form):
 *   launch 1: per-block partial sums of squares of both gradient segments of `grad` ([0, n_lora) and [n_lora, n_lora + n_emb)) and a SNAPSHOT of the
 *             state fields launch 2 is going to overwrite (loss scale, step count, growth tracker);
 *   launch 2: every workgroup derives the step's scalars from the partials + the snapshot itself (so no workgroup reads a state field another one
 *             writes), then does its share: LoRA elements, one added row each (AdamW, then the norm clamp of that row), UNet adapter elements,
 *             a slab of the decay-only rows; workgroup 0 publishes the new state.
 * ws = 2 * 64 + 4 floats of scratch.  lr_table may be NULL (state[TB_ST_LR_MULT] as the host left it). */
typedef struct {
  float* state;                       /* fp32[TB_ST_COUNT] */
  const float* grad;                  /* flat [lora | added rows | unet adapters], loss-scaled (and summed over ranks) */
  float* p_lora; float* m_lora; float* v_lora; int64_t n_lora;        /* group 1 (clipped) */
  float* p_added; float* m_emb; float* v_emb; int32_t n_added; int32_t D;   /* group 0, trainable rows [n_added][D]; n_emb = n_added * D */
  float* p_unet; float* m_unet; float* v_unet; int64_t n_unet;        /* group 2 (unclipped, lr), may be empty */
  float* p_decay; int64_t n_decay;    /* group 0, rows that only see the decoupled decay (n_decay % 4 == 0), may be empty */
  float decay_factor;                 /* 1 - emb_lr * wd at the base lr, as tb_weight_decay takes it */
  float* added_norms;                 /* [n_added] or NULL: row norms before the clamp */
  const float* lr_table; int32_t lr_table_n;
  float lr, emb_lr, beta1, beta2, eps, wd, max_norm, mean_norm;
  float growth_factor, backoff_factor, growth_interval; int32_t use_scaler; float grad_div;
  float* ws;
} tb_opt_desc;
int tb_optimizer_tail(const tb_opt_desc* d, tb_stream_t stream);

/* ---- fp32 (no-AMP) numeric mode ------------------------------------------------------------------------------------------------------
 * The reference runs in full fp32 unless --mixed_precision fp16 is given (train_textboost.py:298-308 default None; :930-939 weight_dtype
 * float32, no GradScaler): the README command (README.md:58-76).  Same layouts and semantics as the fp16 entry points above, every tensor fp32;
 * contractions on the exact-fp32 matrix instruction (csrc/f32_path.hip).  LayerNorm, embedding, pins, KPL, convert and the optimizer entry
 * points above already take fp32 tensors.
 * tb_gemm_f32: the tb_gemm descriptor with fp32 A / A2 / W / W2 / R / C / C2 (c_dtype and r_dtype must be TB_F32; K, N, M unrestricted;
 * conv: Cin % 4 == 0); ws is ignored.  tb_gemm_f32_t: linear only, A given as [K][M] (a_trans) and / or W given as [K][N] (w_trans) -- the
 * weight-gradient-shaped products of the LoRA adapters (dB = dY^T t, dA = dt^T x; train_textboost.py:700-722) */
int tb_gemm_f32(const tb_gemm_desc* d, tb_stream_t stream);
int tb_gemm_f32_t(const tb_gemm_desc* d, int a_trans, int w_trans, tb_stream_t stream);
/* attention with fp32 Q / K / V / O / dO / dQ / dK / dV (same descriptor; fp8_ws and ws of the descriptor are ignored).  The score matrix is
 * materialised: ws = tb_attention_f32_ws_floats(B, H, Sq, Skv) floats (forward uses the first half) */
int64_t tb_attention_f32_ws_floats(int B, int H, int Sq, int Skv);
int tb_attention_f32_fwd(const tb_attn_desc* d, float* ws, int64_t ws_floats, tb_stream_t stream);
int tb_attention_f32_bwd(const tb_attn_desc* d, float* ws, int64_t ws_floats, tb_stream_t stream);
int tb_groupnorm_f32_fwd(const float* x, int64_t ldx, float* y, int64_t ldy, const float* gamma, const float* beta, float* stats, int B, int HW,
                         int C, int G, float eps, int silu, tb_stream_t stream);
int tb_groupnorm_f32_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma, const float* beta, const float* stats,
                         const float* add, int64_t ldadd, float* dx, int64_t lddx, int B, int HW, int C, int G, int silu, tb_stream_t stream);
int tb_add_noise_f32(const float* x0, const float* noise, const int64_t* timesteps, const float* alphas_cumprod, float* noisy, float* velocity,
                     int B, int64_t per_sample, tb_stream_t stream);
int tb_timestep_embed_f32(const int64_t* timesteps, float* out, int B, int dim, tb_stream_t stream);
int tb_conv4_to_nhwc_f32(const float* in, int Cin, const float* w_packed, const float* bias, float* out, int64_t ldo, int B, int H, int W,
                         int Cout, int sign, float in_scale, tb_stream_t stream);
int tb_conv_to4_f32(const float* in, int64_t ldi, const float* w_packed, const float* bias, float* out, int B, int H, int W, int C,
                    tb_stream_t stream);
int tb_upsample2x_f32(const float* x, int64_t ldx, float* u, int64_t ldu, int B, int H, int W, int C, tb_stream_t stream);
int tb_pool2x2_sum_f32(const float* du, int64_t ldu, float* dx, int64_t ldx, int B, int H, int W, int C, tb_stream_t stream);
int tb_add_f32(const float* a, int64_t lda, const float* b, int64_t ldb, float* out, int64_t ldo, int64_t M, int C, tb_stream_t stream);
int tb_lora_pack_f32(const float* A, const float* Bcat, float* w2_fwd /* fp32 [layers][P*D, 64] */, float* w2_dgrad /* fp32 [layers][K, 64] */,
                     int D, int K, int r, int P, int layers, float scaling, tb_stream_t stream);
/* UNet cross-attention K/V adapters (--unet_params_to_train crossattn_kv, train_textboost.py:712-721; fp32 mode only, as in the reference):
 * block-structured K-extension operand of the hoisted K/V GEMM: w2[n, col_base[n] + j] = scaling * B[n, j] (B fp32 [rows, r]) */
int tb_kv_lora_pack_f32(const float* B, const int32_t* col_base, float* w2 /* fp32 [rows, ncols] */, int64_t rows, int ncols, int r,
                        float scaling, tb_stream_t stream);
int tb_mse_loss_f32(const float* pred, const float* target, float* dpred, float* loss_out, const float* loss_scale, int64_t N,
                    float* ws /* >= 256 floats */, tb_stream_t stream);

/* ---- image augmentation + feeder --------------------------------------------------------------------------------
 * Replaces the Pillow / torchvision calls of textboost/augment/paired_augmentation.py:20-277 and textboost/dataset.py:324-381 (Resize(LANCZOS),
 * crop, ToImage / ToDtype / Normalize).  Device images are RGBX u8, one uint32 per pixel, [H][stride] pixels.  Bit-exact with Pillow.
 * The three table builders run on the HOST (plain host pointers, O(W + H) work); everything else takes device pointers + a stream. */
#define TB_IMG_LANCZOS 1 /* PIL.Image.Resampling ids */
#define TB_IMG_BICUBIC 3
/* Resample.c precompute_coeffs + normalize_coeffs_8bpc for a full-width box: returns ksize (odd), or a negative error */
int tb_resample_ksize(int in_size, int out_size, int filter);
int tb_resample_coeffs(int in_size, int out_size, int filter, int32_t* bounds /* host [out_size * 2]: (first, count) */,
                       int32_t* kk /* host [out_size * ksize], 22-bit fixed point */);
/* Geometry.c ImagingScaleAffine column / row tables for an axis-aligned inverse matrix a[6] (a[1] == a[3] == 0); -1 = zero fill */
int tb_affine_nearest_tables(const double* a, int in_w, int in_h, int out_w, int out_h, int32_t* xt /* host [out_w] */,
                             int32_t* yt /* host [out_h] */);
/* one pass of Image.resize: vertical = 0 resamples the width sw -> out_size (dst is [sh][out_size]), 1 the height sh -> out_size */
int tb_img_resample(const uint32_t* src, int64_t sstride, int sw, int sh, uint32_t* dst, int64_t dstride, int out_size,
                    const int32_t* bounds /* device */, const int32_t* kk /* device */, int ksize, int vertical,
                    int coeffs_fit_24bit /* 1: every |kk| < 2^23 (checked by the caller on the host table): full-rate 24-bit multiplies */,
                    tb_stream_t stream);
/* dst[y][x] = xt[x] >= 0 && yt[y] >= 0 ? src[yt[y]][xt[x]] : 0 (crop, pad, flip, collage tiling, NEAREST affine); gray = convert("L") luma.
 * The caller owns the tables: entries must be -1 or valid column / row indices of src (they cannot be range-checked without reading them back). */
int tb_img_gather(const uint32_t* src, int64_t sstride, uint32_t* dst, int64_t dstride, int dw, int dh, const int32_t* xt /* device [dw] */,
                  const int32_t* yt /* device [dh] */, int gray, tb_stream_t stream);
/* Image.transform(AFFINE, a, BICUBIC) of the image edge-padded by (pad_x, pad_y), window [oy, oy + dh) x [ox, ox + dw) of its output (zeros
 * outside it): adjust_scale's pad + affine + center_crop in one launch */
int tb_img_affine_bicubic(const uint32_t* src, int64_t sstride, int sw, int sh, int pad_x, int pad_y, uint32_t* dst, int64_t dstride, int dw,
                          int dh, int ox, int oy, const double* a /* host [6] */, tb_stream_t stream);
/* crop [y0, y0 + R) x [x0, x0 + R) + ToImage + ToDtype(float32, scale=True) + Normalize(0.5, 0.5) -> fp32 [3, R, R] */
int tb_img_to_pixels(const uint32_t* src, int64_t sstride, int sw, int sh, int x0, int y0, float* dst, int R, tb_stream_t stream);
/* RGB u8 [n][3] -> RGBX, and back ([h][w][3]) */
int tb_img_pack_rgb(const uint8_t* rgb, uint32_t* dst, int64_t n_pixels, tb_stream_t stream);
int tb_img_unpack_rgb(const uint32_t* src, int64_t sstride, int w, int h, uint8_t* rgb, tb_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
