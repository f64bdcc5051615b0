/*
 * crg_hip.h - C-ABI of libcrg_hip.so: hand-written HIP/CDNA4 (gfx950) kernels for the
 * Stable Diffusion denoising path of HowToSD/cremage (UNet step + VAE decode/encode).
 *
 * The reference has NO native FFI: its "kernels" are PyTorch call sites inside the L6
 * compute modules.  Each entry point below replaces one family
 * of those call sites; the reference interface it replaces is cited per function
 * (paths relative to the reference root).  The Python binding a maintainer adds is
 * cremage_amd/_lib.py (ctypes); see INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a raw DEVICE pointer (tensor.data_ptr()), caller-owned;
 *   - `stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream);
 *   - activations are channels-last: images [N][H][W][C] ("NHWC"), tokens [B][T][C];
 *   - all kernels are asynchronous on `stream`; no call synchronises, allocates or frees
 *     (graph-capture safe) except crg_ctx_create/destroy/reserve;
 *   - return value: 0 = ok, negative = error; text via crg_last_error(ctx).  The library
 *     never aborts (the reference's ML process has no handler, mp/mp.py:125).
 *   - one context per device, not re-entrant (the reference is single-threaded,
 *     mp/mp.py:32-127).
 */
#ifndef CRG_HIP_H
#define CRG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRG_VERSION 103

typedef struct crg_ctx crg_ctx;

enum crg_dtype { CRG_BF16 = 0, CRG_F32 = 1, CRG_F16 = 2 };

/* GEMM/conv arithmetic:
 *   CRG_PREC_BF16   operands rounded to bf16, one MFMA pass, fp32 accumulate;
 *   CRG_PREC_BF16X3 operands split hi+lo bf16, three MFMA passes (hi*hi + hi*lo + lo*hi),
 *                   fp32 accumulate: ~2^-17 relative operand error ("fp32-class"), used for
 *                   the VAE (pixel L-inf <= 1e-3 target) and the fp32 parity configuration. */
enum crg_prec { CRG_PREC_BF16 = 0, CRG_PREC_BF16X3 = 1,
                /* fp32-class 3x3 conv in TWO matrix passes' worth of cycles (round 4): operands as MX planes (crg_split_mx / crg_groupnorm_mx /
                 * crg_pack_weight_mx) - a fp16 plane and a plane of e4m3 pairs; main product fp16 x fp16, the two cross terms at fp8 precision
                 * on the block-scaled MX matrix instruction; fp32 accumulate and output.  Emulated pixel L-inf of the VAE decode 8e-5 (bound 1e-3) */
                CRG_PREC_F16MX = 2 };

enum crg_epilogue {
  CRG_EPI_NONE = 0,
  CRG_EPI_SILU = 1,  /* y = silu(acc + bias)                                  */
  CRG_EPI_GEGLU = 2  /* y[:, j] = (acc_v + b_v) * gelu_erf(acc_g + b_g); W packed with crg_pack_geglu */
};

enum crg_bias_mode { CRG_BIAS_NONE = 0, CRG_BIAS_COL = 1 /* bias[n] */, CRG_BIAS_ROW = 2 /* bias[m] */ };

/* ---- context ------------------------------------------------------------------------- */
int crg_version(void);
/* Which 16-bit element type this build of the library computes in: 0 = bfloat16 (libcrg_hip.so, the default and what
 * BASELINE.json configs[1] names), 1 = IEEE fp16 (libcrg_hip_f16.so: the same kernels with the _f16 matrix instructions - the
 * operand type of the reference's own GPU flow, image_generator.py:489-493,748-751).  In either build `CRG_BF16` in a dtype argument
 * means "the library's half type"; the caller must hand it tensors of that type (cremage_amd.ops does: ops.HALF). */
int crg_half_kind(void);
int crg_ctx_create(int device, crg_ctx** out);
void crg_ctx_destroy(crg_ctx* ctx);
const char* crg_last_error(crg_ctx* ctx);
/* Pre-size the context's scratch (GroupNorm partial statistics, split-K slabs). Synchronous. */
int crg_ctx_reserve(crg_ctx* ctx, size_t bytes);

/* ---- per-kernel timing (bench.py roofline: HIP events on the launch stream) ------------ */
/* Between begin/end every kernel launched by a crg_* call is bracketed by hipEvents on its launch stream and
 * tagged with its kernel slot (one slot per kernel symbol as rocprofv3 prints it, see crg_kernel_name) and the
 * algorithmic FLOPs / bytes of the call.  crg_profile_end synchronises and fills `out`. */
enum crg_kernel_slot {
  CRG_K_GEMM_W1 = 0, CRG_K_GEMM_W4 = 1, CRG_K_GEMM_W5 = 2, /* gemm_glds_kernel<WNT, *, CONV=false>  bf16 LDS-DMA GEMM      */
  CRG_K_GEMM_X3 = 3,                                        /* gemm_kernel<*, *, *, *, false>        fp32 / split-bf16 GEMM */
  CRG_K_CONV_W1 = 4, CRG_K_CONV_W4 = 5, CRG_K_CONV_W5 = 6, /* conv3_pp_kernel<WNT> / conv3_rowhalo_kernel<WNT> / gemm_glds_kernel<WNT, *, CONV=true>  bf16 implicit-GEMM conv */
  CRG_K_CONV_X3 = 7,                                        /* gemm_kernel<*, *, *, *, true>         fp32-class conv (VAE)  */
  CRG_K_SPLITK = 8, CRG_K_ATTN = 9, CRG_K_GN_STATS = 10, CRG_K_GN_APPLY = 11, CRG_K_LAYERNORM = 12,
  CRG_K_ELEMENTWISE = 13, CRG_K_CONV_SMALL = 14, CRG_K_SOFTMAX = 15,
  CRG_K_LNGEMM = 16,                                        /* lngemm_kernel<WNT, KT, PAIR>          LayerNorm fused into the consuming GEMM */
  CRG_K_SLOTS = 17
};
typedef struct {
  double ms[CRG_K_SLOTS];     /* summed device time per kernel slot          */
  double flops[CRG_K_SLOTS];  /* summed algorithmic FLOPs (2*MAC)            */
  double bytes[CRG_K_SLOTS];  /* summed algorithmic (minimum) HBM bytes      */
  int64_t launches[CRG_K_SLOTS];
} crg_profile;
/* Kernel symbol (as it appears in a rocprofv3 kernel trace, template arguments abbreviated) of a slot. */
const char* crg_kernel_name(int slot);
int crg_profile_begin(crg_ctx* ctx);
int crg_profile_end(crg_ctx* ctx, void* stream, crg_profile* out);

/* ---- GroupNorm (+SiLU) ------------------------------------------------------------------
 * Replaces GroupNorm32.forward (modules/ldm/modules/diffusionmodules/util.py:214-216, eps 1e-5,
 * fp32 statistics) + nn.SiLU (openaimodel.py:207,231,754), Normalize (attention.py:189-190,
 * model.py:45-46, eps 1e-6) + `nonlinearity` (model.py:40-42).
 * x,y: [N][HW][C] of `dtype`; gamma,beta: fp32 [C].  Statistics and affine in fp32.
 * If x2 != NULL the input is the virtual channel concat [x | x2] with C = C1 + C2
 * (th.cat([h, hs.pop()], dim=1), openaimodel.py:808) and C1 % (C/groups) == 0. */
int crg_groupnorm(crg_ctx* ctx, void* stream, const void* x, const void* x2, int C1, const float* gamma,
                  const float* beta, void* y, int N, int HW, int C, int groups, float eps, int fuse_silu,
                  int dtype);
/* GroupNorm (+SiLU) of an fp32 image whose result is written as TWO bf16 planes, hi = bf16(y) and lo = bf16(y - hi) (each
 * [N][HW][C]): the operand format of the fp32-class conv (crg_conv_args.x_lo).  Same statistics and arithmetic as
 * crg_groupnorm with dtype CRG_F32; used by the VAE ResnetBlocks (model.py:128-148: norm -> swish -> conv). */
int crg_groupnorm_split(crg_ctx* ctx, void* stream, const void* x, const void* x2, int C1, const float* gamma,
                        const float* beta, void* y_hi, void* y_lo, int N, int HW, int C, int groups, float eps,
                        int fuse_silu);
/* crg_groupnorm_split for an fp32 input whose PRODUCER handed over its statistics (round 3: the fp32-class convs of the VAE emit them from
 * their epilogue like the bf16 path does, crg_conv_args.gn_stats with y_dtype = CRG_F32): `stats` = fp32 [2][N * HW / 32][C], plane 0 the
 * per 32-row block and channel sums of the conv's outputs, plane 1 of their squares.  Replaces the same `Normalize` + swish
 * (model.py:116-121 / :99-113) without the statistics pass over the fp32 tensor. */
int crg_groupnorm_pre_split(crg_ctx* ctx, void* stream, const void* x, const float* stats, const float* gamma, const float* beta,
                            void* y_hi, void* y_lo, int N, int HW, int C, int groups, float eps, int fuse_silu);
/* GroupNorm (+SiLU) whose STATISTICS come from the producer of x (and of x2): the conv / GEMM launch that wrote the tensor also wrote
 * `gn_stats` (crg_conv_args / crg_gemm_args): per 32-row block and channel the sum and the sum of squares of its rounded outputs,
 * planes [2][N * HW / 32][C1].  This call folds them per (sample, group) - one tiny launch instead of a read of the whole tensor -
 * and applies the normalisation.  Same result as crg_groupnorm up to the summation order of the statistics (fp32 partials of 32
 * values, fp64 fold).  bf16 only; HW % 32 == 0.  Reference call sites as crg_groupnorm (util.py:214-216, openaimodel.py:205-209,
 * 229-236: the GroupNorm32 + SiLU in front of each ResBlock conv).  stats2 pairs with x2 (virtual concat).  rows1 / rows2 (0 = 32): rows per
 * partial of stats1 / stats2 as their producers reported them (crg_conv_args.gn_stats_rows); with tile partials (>= 64 rows) on every
 * input the fold happens inside the normalising launch (HW / rows x C x 2 floats per sample and block) and the finalise launch is gone. */
int crg_groupnorm_pre(crg_ctx* ctx, void* stream, const void* x, const void* x2, int C1, const float* stats1,
                      const float* stats2, const float* gamma, const float* beta, void* y, int N, int HW, int C,
                      int groups, float eps, int fuse_silu, int dtype, int rows1, int rows2);
/* MX planes (CRG_PREC_F16MX, round 4) of an fp32 image [pixels][C], C % 64 == 0: x16 = fp16 [pixels][C]; x8 = [pixels][C / 64][128] bytes - per
 * 64-channel chunk 64 OCP e4m3 values of half(x) * 2^hi_log2 followed by 64 of (x - half(x)) * 2^lo_log2 (saturating).  crg_split_mx: a
 * plain tensor; crg_groupnorm_mx: GroupNorm(+SiLU) of x written in that form (`stats` = the producing conv's statistics side channel as in
 * crg_groupnorm_pre_split, or NULL for a statistics pass over x) - the VAE's Normalize + swish in front of a 3x3 conv (model.py:99-121);
 * crg_pack_weight_mx: a 3x3 conv weight [Cout][Cin][3][3] (Cin % 64 == 0) in the conv's K order, w16 [Cout][9 Cin], w8 [Cout][9 Cin / 64][128]. */
int crg_split_mx(crg_ctx* ctx, void* stream, const void* x, void* x16, void* x8, int64_t pixels, int C, int hi_log2, int lo_log2);
int crg_groupnorm_mx(crg_ctx* ctx, void* stream, const void* x, const float* stats, const float* gamma, const float* beta, void* y16, void* y8,
                     int N, int HW, int C, int groups, float eps, int fuse_silu, int hi_log2, int lo_log2);
int crg_pack_weight_mx(crg_ctx* ctx, void* stream, const void* src, int src_dtype, int n_out, int n_in, void* dst16, void* dst8, int hi_log2,
                       int lo_log2);
/* fp32 tensor -> the same two bf16 planes (n elements, n % 8 == 0): for fp32-class convs whose input does not come from a
 * GroupNorm (the VAE's Upsample / Downsample convs on the residual stream, model.py:60-64,79-86). */
int crg_split_bf16(crg_ctx* ctx, void* stream, const void* x, void* hi, void* lo, int64_t n);

/* ---- LayerNorm --------------------------------------------------------------------------
 * Replaces nn.LayerNorm(dim) in BasicTransformerBlock (attention.py:900-902,909-911).
 * x,y: [rows][dim] of `dtype`; gamma,beta fp32 [dim]. */
int crg_layernorm(crg_ctx* ctx, void* stream, const void* x, const float* gamma, const float* beta, void* y,
                  int64_t rows, int dim, float eps, int dtype);

/* ---- LayerNorm fused into the GEMM that consumes it: Y = epi(LN(X) (MxK) * W^T (NxK) + bias) --------------------
 * Replaces nn.LayerNorm + the Linear(s) behind it in BasicTransformerBlock._forward (attention.py:908-912):
 * norm1 -> to_q | to_k | to_v (:614,629,636; one launch over the row-stacked weights), norm2 -> to_q, norm3 -> the GEGLU
 * projection (:88-96).  A block keeps its 128 rows resident in LDS, normalises them in place and walks every n-tile of W:
 * X is read once and LN(X) never reaches HBM.  bf16 in / out, fp32 statistics (two-pass, as crg_layernorm).
 *   x: bf16 [M][ldx];  gamma, beta: fp32 [K];  w: bf16 [N][ldw] (crg_pack_weight; GEGLU: CRG_PACK_GEGLU);  bias: fp32 [N]
 *   (GEGLU: crg_pack_geglu_bias) or NULL;  y: bf16 [M][ldy];  epilogue: CRG_EPI_NONE | CRG_EPI_GEGLU (then y is [M][N/2]).
 * gamma == beta == NULL: no LayerNorm (plain GEMM on the same row-resident kernel, e.g. to_out + residual).
 * Built for K == 320 (the 64x64 level of SD1.5, where these GEMMs have five k-tiles and are launch / latency bound);
 * other K return an error: the caller uses crg_layernorm + crg_gemm. */
typedef struct {
  const void* x; int64_t ldx;
  const float* gamma; const float* beta; float eps;
  const void* w; int64_t ldw;
  const float* bias;
  void* y; int64_t ldy;
  int M, N, K;
  int epilogue;
  /* optional: columns n >= vt_n0 (tile aligned: multiple of 160 when N % 160 == 0, else 128) go TRANSPOSED to
   * vt[(m / vt_tokens) * (N - vt_n0) + (n - vt_n0)][m % vt_tokens] (row length vt_ld) instead of y - the V^T operand of
   * crg_attention, emitted by the same launch as Q | K.  vt == NULL: every column goes to y. */
  void* vt; int vt_n0; int vt_tokens; int64_t vt_ld;
  /* optional residual [M][ldr] (bf16) added after the bias (plain epilogue only): to_out / proj_out + skip (attention.py:685-693,1049-1057).
   * With gamma == beta == NULL the LayerNorm is skipped and the kernel is a row-resident GEMM for K = 320. */
  const void* residual; int64_t ldr;
} crg_lngemm_args;
int crg_ln_gemm(crg_ctx* ctx, void* stream, const crg_lngemm_args* args);

/* ---- GEMM: Y[b] = epi(A[b] (MxK) * W[b]^T (NxK) + bias) + residual ------------------------
 * Replaces F.linear / 1x1 nn.Conv2d call sites: to_q/to_k/to_v (attention.py:614,629,636),
 * to_out (:685), GEGLU proj + net[2] (:88-96,157-168), proj_in/proj_out (:1036,1049),
 * time_embed / emb_layers (openaimodel.py:538-543,222-228), skip_connection 1x1 (:245),
 * VAE q/k/v/proj_out/nin_shortcut (model.py:163-182,122-126), quant/post_quant_conv
 * (autoencoder.py:302-303).  Both operands are K-contiguous ("NT").
 *   a: `a_dtype` [M][lda];  w: packed by crg_pack_weight (bf16 planes) [N][ldw];
 *   bias fp32; residual/y: `y_dtype`;  cvec fp32 [M / rows_per_cvec][N] added per row group.
 * Swapping the roles of a and w yields transposed outputs (used to emit V^T for attention). */
typedef struct {
  const void* a; int64_t lda; int64_t a_bstride;
  const void* w; int64_t ldw; int64_t w_bstride;   /* hi plane                      */
  const void* w_lo;                                 /* lo plane (BF16X3) or NULL     */
  const float* bias; int bias_mode;
  const void* residual; int64_t ldr; int64_t r_bstride;
  void* y; int64_t ldy; int64_t y_bstride;
  int M, N, K, batch;
  int epilogue;
  int a_dtype, y_dtype, prec;
  int a_is_weight;   /* BF16X3 only: `a` is a packed weight (hi plane) and a_lo its lo plane */
  const void* a_lo;
  /* optional GroupNorm statistics side channel (bf16 in / out, batch 1, N % 8 == 0, plain epilogue): fp32 [2][ceil(M / 32)][N],
   * plane 0 = per 32-row block and column the sum of the rounded outputs y, plane 1 the sum of their squares (consumed by
   * crg_groupnorm_pre when y - e.g. proj_out + residual, attention.py:1049-1057 - feeds the next ResBlock's GroupNorm) */
  float* gn_stats;
  /* optional transposed column range (plain bf16 GEMM, unbatched, K below the split-K rule): output columns n >= vt_n0 - a multiple
   * of the tile width: 160 when N % 160 == 0, else 128 - go to vt[(m / vt_tokens) * (N - vt_n0) + (n - vt_n0)][m % vt_tokens] (bf16,
   * row length vt_ld >= vt_tokens) instead of y, which then has vt_n0 columns: `to_q | to_k | to_v` of a self-attention
   * (attention.py:614,629,636) as ONE launch whose V third comes out as the V^T operand of crg_attention */
  void* vt; int vt_n0; int vt_tokens; int64_t vt_ld;
  /* ---- nn.LayerNorm folded into the GEMM that consumes it, any K (attention.py:900-912: norm1 -> to_q | to_k | to_v, norm2 -> to_q,
   * norm3 -> the GEGLU projection; sgm/modules/attention.py:724-847 for SDXL) -------------------------------------------------------
   * PRODUCER side, `row_stats` (bf16 in / out, unbatched, plain epilogue, N % 8 == 0): the launch that writes the LayerNorm's input
   * (proj_in, to_out + residual, net[2] + residual) also writes, per row m, the sum and the sum of squares of its rounded outputs as
   * row_stats_parts = 2 * ceil(N / tile) column partials (tile = 160 when N % 160 == 0, else 128): fp32 [M][row_stats_parts][2]
   * (N <= 1280: at most 16 partials, which is what the consumer side takes).
   * CONSUMER side, `ln_stats` (bf16 in / out, unbatched, no residual, epilogue NONE or GEGLU, a transposed range allowed): `a` holds the
   * RAW rows x (the LayerNorm input), `w` the weight scaled by gamma and `bias` the folded bias (crg_pack_ln_weight), and
   *     y = rstd_m * (a W'^T - mean_m * ln_colsum[n]) + bias[n]  =  LayerNorm(x) W^T + b
   * with (mean_m, rstd_m) folded from the ln_parts partials per row of the producer's row_stats (fp32, eps = ln_eps, biased
   * variance as nn.LayerNorm).  The normalised tensor never exists and no LayerNorm launch runs.  K = the LayerNorm width. */
  float* row_stats; int row_stats_parts;
  const float* ln_stats; int ln_parts; const float* ln_colsum; float ln_eps;
} crg_gemm_args;
int crg_gemm(crg_ctx* ctx, void* stream, const crg_gemm_args* args);

/* ---- conv2d as implicit GEMM ---------------------------------------------------------------
 * Replaces conv_nd(2, Cin, Cout, 3, padding=1) (openaimodel.py:208,234,551,755), stride-2
 * Downsample (:155-157), Upsample = nearest-2x + conv (:120-122; VAE model.py:60-64), the VAE
 * convs (model.py:99-113,494-498,536-540) and its asymmetric-pad stride-2 Downsample
 * (model.py:79-83, pad (0,1,0,1)).  Fused: skip concat as a second input pointer
 * (openaimodel.py:808), nearest-2x upsample in the gather, bias, per-sample channel vector
 * (the timestep-embedding add, openaimodel.py:268-277) and residual add (:279).
 *   x: [N][H][W][C1] (+ x2: [N][H][W][C2]) of `x_dtype`; w: packed [Cout][kh*kw*(C1+C2)];
 *   y/residual: [N][Ho][Wo][Cout] of `y_dtype`; cvec fp32 [N][Cout]; bias fp32 [Cout]. */
typedef struct {
  const void* x; const void* x2; int C1, C2;
  const void* w; const void* w_lo;
  const float* bias; const float* cvec;
  int64_t cvec_ld;                  /* row stride of cvec in floats (0 = Cout) */
  const void* residual;
  void* y;
  int N, H, W, Cout, Ho, Wo;
  int ksize, stride, pad_t, pad_l;  /* pad_b / pad_r are implied by Ho/Wo */
  int upsample2x;
  int x_dtype, y_dtype, prec;
  const void* x_lo;                 /* BF16X3 with PRE-SPLIT activations: `x` is the bf16 hi plane, `x_lo` the bf16 lo plane
                                       (same layout; written by crg_groupnorm_split / crg_split_bf16); x_dtype = CRG_BF16,
                                       y_dtype = CRG_F32; NULL otherwise */
  float* gn_stats;                  /* optional GroupNorm statistics side channel of y (see crg_gemm_args.gn_stats; M = N * Ho * Wo rows,
                                       Ho * Wo % 32 == 0): the conv in front of a GroupNorm (openaimodel.py:208 -> :229-231, :234 -> the
                                       next block's :205-207) hands it the statistics, so the tensor is not read once more for them */
  /* optional GroupNorm(+SiLU) of the finished output - the `normalization(channels)` + `SiLU` that follows the conv inside a ResBlock
   * (openaimodel.py:208 -> :229-231; util.py:214-216): gn_y (same layout and dtype as y) = silu?(GroupNorm(y) * gn_gamma + gn_beta).
   * y is still written.  When the conv is split along K and a (sample, group) slab of the output fits one block (the 8x8 / 16x16
   * UNet levels) the launch that sums the K slices normalises as well; otherwise this is crg_conv2d followed by crg_groupnorm.
   * The result is bitwise that of the two separate calls. */
  const float* gn_gamma; const float* gn_beta; void* gn_y;
  int gn_groups; int gn_silu; float gn_eps;
  /* prec = CRG_PREC_F16MX: x / w are the fp16 planes, x_lo / w_lo the e4m3 pair planes; mx_log2 = {w hi8, w lo8, x hi8, x lo8}: the power of
   * two each fp8 half was multiplied by when it was written (crg_pack_weight_mx / crg_split_mx / crg_groupnorm_mx take the same numbers) */
  int mx_log2[4];
  /* optional OUT (host memory, written before the call returns; NULL: 32-row partials, as ever): the granularity of gn_stats this launch
   * wrote - rows per partial.  32, or the TILE height (256) where the kernel folds its waves' sums itself (round 4: the 256-pixel-tile 3x3
   * conv with the paired epilogue, Ho * Wo % 256 == 0, no K slices): plane p, partial r, channel c at gn_stats[p * plane + r * Cout + c] with
   * the SAME plane stride as the 32-row layout (ceil(M / 32) * Cout floats), only the first M / rows partials of each plane used.  Hand
   * the value to crg_groupnorm_pre (rows1 / rows2): with tile partials on every input it folds them inside the normalising launch. */
  int* gn_stats_rows;
} crg_conv_args;
int crg_conv2d(crg_ctx* ctx, void* stream, const crg_conv_args* args);

/* ---- weight packing -------------------------------------------------------------------------
 * src: a torch parameter as the checkpoint stores it (fp32/bf16/fp16):
 *   CRG_PACK_LINEAR   [N][K]            -> [N][K]             (nn.Linear / 1x1 conv)
 *   CRG_PACK_CONV     [Cout][Cin][k][k] -> [Cout][k*k*Cin]    K order = [tap][Cin], or, for 3x3 with Cin % 64 == 0,
 *                                                             [Cin/64][tap][64] (chunk-major: the 9 taps of one
 *                                                             64-channel slab are consecutive k-tiles -> L1 reuse)
 *   CRG_PACK_GEGLU    [2*F][K]          -> rows interleaved in 16-row groups [v0-15|g0-15|v16-31|..]
 * dst_hi (and dst_lo when non-NULL: the bf16 residual src - hi) are bf16, caller-allocated. */
enum crg_pack_kind { CRG_PACK_LINEAR = 0, CRG_PACK_CONV = 1, CRG_PACK_GEGLU = 2 };
int crg_pack_weight(crg_ctx* ctx, void* stream, const void* src, int src_dtype, int kind, int n_out, int n_in,
                    int ksize, void* dst_hi, void* dst_lo);
/* bias for GEGLU packed the same way (fp32 in, fp32 out) */
int crg_pack_geglu_bias(crg_ctx* ctx, void* stream, const float* src, int n_out2, float* dst);
/* Operands of a GEMM that carries the LayerNorm in front of it as an epilogue correction (crg_gemm_args.ln_stats):
 *   dst_w[o][k]   = half( W[r(o)][k] * gamma[k] )                    bf16 [n_out][n_in]
 *   dst_colsum[o] = sum_k float(dst_w[o][k])                          fp32 [n_out]   (over the ROUNDED weight: what the MFMA multiplies,
 *                                                                     so that mean * colsum cancels the mean's share of the product exactly)
 *   dst_bias[o]   = sum_k W[r(o)][k] * beta[k] + bias[r(o)]           fp32 [n_out]   (bias may be NULL)
 * r(o) = o for CRG_PACK_LINEAR, the GEGLU row interleave of crg_pack_weight for CRG_PACK_GEGLU.  src: [n_out][n_in] of src_dtype.
 * Replaces the `weight` / `bias` of nn.LayerNorm (attention.py:900-902) together with the Linear behind it. */
int crg_pack_ln_weight(crg_ctx* ctx, void* stream, const void* src, int src_dtype, const float* gamma, const float* beta,
                       const float* bias, int kind, int n_out, int n_in, void* dst_w, float* dst_colsum, float* dst_bias);

/* ---- attention: O = softmax(Q K^T * scale) V ---------------------------------------------------
 * Replaces the attention core of CrossAttentionOriginal.forward (attention.py:644-658), the sliced
 * CUDA variant (:415-424) and xformers (:811): heads split 'b n (h d) -> (b h) n d', softmax over
 * keys, merge.  Flash-style (never materialises Nq x Nk).  bf16 in/out, fp32 softmax/accumulate.
 *   q: [B][Nq][ldq] (head h at column h*Dh), k: [B][Nk][ldk], vt: [B][H*Dh][ldvt] (V transposed:
 *   row = channel, column = key), o: [B][Nq][ldo].  Dh % 8 == 0, Dh <= 160. */
int crg_attention(crg_ctx* ctx, void* stream, const void* q, int64_t ldq, const void* k, int64_t ldk,
                  const void* vt, int64_t ldvt, void* o, int64_t ldo, int B, int H, int Nq, int Nk, int Dh,
                  float scale, int dtype);
/* Same attention with V ROW-MAJOR: v: [B][Nk][ldv] (head h at column h*Dh) - the layout of a fused Q|K|V (or K|V) projection
 * output, so that `to_q / to_k / to_v` of a self-attention (attention.py:614,629,636) are ONE GEMM launch over stacked
 * weights and q, k, v are column slices of its output (ldq = ldk = ldv = 3*H*Dh).  The kernel transposes V on the LDS read. */
int crg_attention_v(crg_ctx* ctx, void* stream, const void* q, int64_t ldq, const void* k, int64_t ldk,
                    const void* v, int64_t ldv, void* o, int64_t ldo, int B, int H, int Nq, int Nk, int Dh,
                    float scale, int dtype);

/* ---- row softmax (VAE AttnBlock, model.py:197-199: softmax(w * c^-0.5, dim=2)) ----------------
 * x,y: [rows][cols] of `dtype` (in place allowed), y = softmax(x * scale) per row. */
int crg_softmax_rows(crg_ctx* ctx, void* stream, const void* x, void* y, int64_t rows, int cols, int64_t ld,
                     float scale, int dtype);

/* ---- small-channel direct conv ----------------------------------------------------------------
 * conv_in (4->320, openaimodel.py:551; VAE 4->512 model.py:494; encoder 3->128 :390), conv_out
 * (320->4, :755; VAE 128->3 :536; encoder 512->8 :435) and the 1x1 quant_conv / post_quant_conv
 * (autoencoder.py:302-303): Cin <= 8 or Cout <= 8, ksize 3 (pad 1) or 1, stride 1.
 * x: [N][H][W][Cin] x_dtype, w: fp32 [Cout][Cin][k][k] (checkpoint layout), bias fp32, y y_dtype. */
int crg_conv_small(crg_ctx* ctx, void* stream, const void* x, const float* w, const float* bias, void* y,
                   int N, int H, int W, int Cin, int Cout, int ksize, int x_dtype, int y_dtype);

/* ---- elementwise / layout ------------------------------------------------------------------------ */
/* timestep_embedding (util.py:151-171): out[b] = [cos(t_b f_i) | sin(t_b f_i)], f_i = exp(-ln(1e4) i/half) */
int crg_timestep_embedding(crg_ctx* ctx, void* stream, const float* t, void* out, int B, int dim, int dtype);
/* y = silu(x) (nn.SiLU in emb_layers / time_embed, openaimodel.py:222-228,538-543) */
int crg_silu(crg_ctx* ctx, void* stream, const void* x, void* y, int64_t n, int dtype);
/* NCHW (src_dtype) -> NHWC (dst_dtype) and back; used once at the UNet/VAE boundary */
int crg_nchw_to_nhwc(crg_ctx* ctx, void* stream, const void* src, void* dst, int N, int C, int HW, int src_dtype,
                     int dst_dtype);
int crg_nhwc_to_nchw(crg_ctx* ctx, void* stream, const void* src, void* dst, int N, int C, int HW, int src_dtype,
                     int dst_dtype);
/* y = a*x + b elementwise with dtype conversion (latent scaling z/0.18215, clamp((x+1)/2,0,1)) */
int crg_affine_cast(crg_ctx* ctx, void* stream, const void* x, void* y, int64_t n, float a, float b, float lo,
                    float hi, int src_dtype, int dst_dtype);

/* One k-diffusion Euler / Euler-ancestral step of an eps-prediction model under classifier-free guidance, fp32, in place on x
 * (the scalings, the guidance and the update that sit either side of the UNet call, as one kernel):
 *   den_u = x + eps_u * (-sigma), den_c = x + eps_c * (-sigma)      CompVisDenoiser.forward      k_diffusion/external.py:111-114
 *   den   = den_u + cfg_scale * (den_c - den_u)                      LDMWrapperForKDiffusion      ldm_wrapper_for_k_diffusion.py:99
 *   d     = (x - den) / sigma;   x += d * dt                         sample_euler(_ancestral)     k_diffusion/sampling.py:134-142,157-160
 *   x    += noise * noise_scale   (noise may be NULL)                ancestral noise              sampling.py:161-162
 * eps: [2][n] fp32 (unconditional half first - the batch-doubled UNet output), x / noise: [n] fp32. */
int crg_cfg_euler_step(crg_ctx* ctx, void* stream, void* x, const void* eps, const void* noise, int64_t n, float sigma,
                       float dt, float cfg_scale, float noise_scale);

/* y = a*x + b*y elementwise (IP-Adapter FaceID: out + ipa_scale * out_ipa, attention.py:681;
 * ControlNet residual adds, cldm.py:57-65) */
int crg_axpby(crg_ctx* ctx, void* stream, const void* x, void* y, int64_t n, float a, float b, int dtype);

#ifdef __cplusplus
}
#endif
#endif /* CRG_HIP_H */
