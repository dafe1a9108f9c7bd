/*
 * coin_hip.h - C ABI of libcoin_hip.so: the MI355X (gfx950) kernels behind COIN's
 * adaptation-training hot path.
 *
 * The reference (Flashkong/COIN) has no native boundary of its own: its hot path sits
 * behind Python registries and bottoms out in torchvision / cuDNN / cuBLAS.
 * This header is the boundary a maintainer binds instead (ctypes stub: INTEGRATION.md).
 * Each entry point names the reference call site it replaces (file:line under
 * /root/reference).
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and sizes; no C++ / torch types.
 *   - every pointer is a DEVICE pointer owned by the caller and kept alive by the caller
 *     until the stream has passed the call; the library allocates nothing and keeps no
 *     global state (re-entrant, thread-safe per stream).
 *   - asynchronous on `stream` (a hipStream_t passed as void*; NULL = the null stream).
 *   - returns 0 on success, a NEGATIVE COIN_E* code for bad arguments, or a POSITIVE
 *     hipError_t if the launch failed.  Never throws, never aborts.
 *   - `dtype`: COIN_F32 = float, COIN_BF16 = bfloat16 (round-to-nearest-even on store);
 *     accumulation is always fp32.
 */
#ifndef COIN_HIP_H
#define COIN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a prototype below changes (2: round 4's signature changes -- coin_sgd_step gate, coin_bn_* ReLU mask, coin_anchor_match
 * candidate sets, ... -- and round 5's additions; 3: round 6 -- coin_conv_gemm_stats_tile_rows, the tile height argument of
 * coin_conv_gemm_stats_finalize, COIN_BN_MAX_PARTS 768).  The Python binding refuses a library whose version differs (coin_amd/_lib.py). */
#define COIN_ABI_VERSION 3

enum { COIN_F32 = 0, COIN_BF16 = 1 };
enum { COIN_NCHW = 0, COIN_NHWC = 1 };
enum { COIN_ACT_NONE = 0, COIN_ACT_LEAKY_RELU = 1, COIN_ACT_RELU = 2 };

enum {
  COIN_OK = 0,
  COIN_EINVAL = -1,   /* null pointer / negative size / unknown enum */
  COIN_ESHAPE = -2,   /* shape not supported by this kernel (see the entry point) */
  COIN_EALIGN = -3    /* pointer or leading dimension not aligned as required */
};

/* Library / ABI version and the offload arch the code objects were built for ("gfx950"). */
int coin_abi_version(void);
const char* coin_build_arch(void);
/* Returns and CLEARS the calling thread's last HIP runtime error (hipGetLastError): after a failed stream capture the error is sticky and
 * the next entry point would report it as its own launch failure (coin_amd/graphs.py calls this when a capture is abandoned). */
int coin_clear_last_error(void);

/* ------------------------------------------------------------------------------------------
 * RoIAlign   (replaces coin/modeling/roi_heads/clip_roi_heads.py:172-176 `self.pooler(...)`
 *             -> detectron2 ROIPooler -> torchvision.ops.roi_align(aligned, sampling_ratio))
 *
 * feat  : [N,C,H,W] (COIN_NCHW) or [N,H,W,C] (COIN_NHWC), dtype `dtype`
 * rois  : [R,5] float32 rows (batch_index, x0, y0, x1, y1) in input-image pixels
 * out   : [R,C,ph,pw] (COIN_NCHW) or [R,ph,pw,C] (COIN_NHWC), dtype `dtype`
 * sampling_ratio <= 0 selects the adaptive grid ceil(roi_size / pooled_size).
 * NHWC requires C % 8 == 0 (bf16) / C % 4 == 0 (f32) and 16-byte aligned feat/out.
 * ---------------------------------------------------------------------------------------- */
int coin_roi_align_fwd(const void* feat, int N, int C, int H, int W, int layout,
                       const float* rois, int R, int ph, int pw, float spatial_scale,
                       int sampling_ratio, int aligned, void* out, int dtype, void* stream);

/* Adjoint of coin_roi_align_fwd (torchvision roi_align backward).
 * grad_out : same shape/layout/dtype as `out` above.
 * grad_feat: [N,C,H,W] / [N,H,W,C] FLOAT32, fully OVERWRITTEN (no need to zero it; R == 0 writes zeros).
 *            NHWC with ph, pw <= 16 runs the atomic-free tiled gather: every element is written once, RoIs are
 *            summed in index order (bit-reproducible).  Larger bins and NCHW fall back to zero-fill + float
 *            atomics (summation order not fixed).
 * workspace: none. */
int coin_roi_align_bwd(const void* grad_out, int N, int C, int H, int W, int layout,
                       const float* rois, int R, int ph, int pw, float spatial_scale,
                       int sampling_ratio, int aligned, float* grad_feat, int dtype, void* stream);

/* Multi-level RoIAlign of the FPN extension (coin_amd/modeling/fpn.py; the reference's pooler has a single level,
 * clip_roi_heads.py:172-176): RoI r is pooled from pyramid level roi_level[r] (device int32, clamped to [0, nlevels)) -- ONE launch for
 * all levels, every output row written exactly once.  Channels-last maps only; `levels` is a HOST array of nlevels <= COIN_ROI_MAX_LEVELS
 * entries (device map pointer, its height / width, its spatial scale), all maps [N, H_l, W_l, C].  The backward is run once per level:
 * coin_roi_align_bwd_level overwrites that level's float32 gradient map with the contributions of the RoIs whose roi_level == level
 * (the atomic-free tile gather; bins <= 16 x 16). */
#define COIN_ROI_MAX_LEVELS 4
typedef struct coin_roi_level {
  const void* feat;
  int H, W;
  float spatial_scale;
} coin_roi_level;
int coin_roi_align_fwd_levels(const coin_roi_level* levels, int nlevels, int N, int C, const float* rois, const int* roi_level, int R,
                              int ph, int pw, int sampling_ratio, int aligned, void* out, int dtype, void* stream);
int coin_roi_align_bwd_level(const void* grad_out, int N, int C, int H, int W, const float* rois, const int* roi_level, int level, int R,
                             int ph, int pw, float spatial_scale, int sampling_ratio, int aligned, float* grad_feat, int dtype,
                             void* stream);

/* ------------------------------------------------------------------------------------------
 * Box-head GEMM   (replaces the nn.Linear calls of FastRCNNOutputLayers.forward,
 *                  coin/modeling/roi_heads/fast_rcnn.py:331-337: `trans` MLP, `cls_score`,
 *                  `bbox_pred`; and their autograd backward)
 *
 *   C[M,N] = act( A[M,K] . B[N,K]^T + bias[N] )          ("NT": both operands K-contiguous)
 *
 * A, B: dtype `dtype`, row-major with leading dimensions lda, ldb (elements).
 * C   : dtype `out_dtype`, leading dimension ldc.  bias: float32 or NULL.
 * act : COIN_ACT_*; leaky slope is `act_alpha` (0.01 for nn.LeakyReLU()).
 * bf16 path: MFMA 16x16x32 bf16, fp32 accumulate; requires K % 64 == 0, lda/ldb % 8 == 0,
 *            16-byte aligned A/B.  M, N arbitrary.
 * f32 path : MFMA 16x16x4 f32 (exact fp32 FMA chain in k order); requires K % 16 == 0.
 * ---------------------------------------------------------------------------------------- */
int coin_gemm_nt(const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                 int M, int N, int K, const float* bias, int act, float act_alpha,
                 int dtype, int out_dtype, void* stream);

/* out[N,M] = in[M,N]^T (row-major, contiguous), same dtype; used to present the NN / TN
 * products of the backward pass (dX = dY.W, dW = dY^T.X) to coin_gemm_nt. */
int coin_transpose2d(const void* in, void* out, int M, int N, int dtype, void* stream);

/* Backward of bias + activation for C = act(Z + bias):
 *   dZ[M,N] = dC * act'(C)   (leaky/relu derivative recovered from the sign of C)
 *   dbias[N] += sum_m dZ[m,n]  (float32, ACCUMULATED; may be NULL)
 * dC, C, dZ share dtype `dtype` and leading dimension ld.  workspace: ceil(M / 256) * N floats (required when dbias is given): the
 * column sums of every 256-row block, added to dbias in block order by a second launch -- no float atomics, bit-reproducible. */
int coin_bias_act_bwd(const void* dC, const void* C, void* dZ, int ld, int M, int N,
                      float* dbias, int act, float act_alpha, int dtype, void* workspace, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused train-mode BatchNorm (+ residual) (+ ReLU) (+ 2x2 average pool), channels-last
 *   (replaces the nn.BatchNorm2d / `out += identity` / ReLU / nn.AvgPool2d chains of the CLIP Bottleneck,
 *    coin/modeling/utils.py:77-90, for the trainable stages incl. res5 on the RoI tiles)
 *
 * x [N,H,W,C] dtype `dtype`; per-channel float32 vectors of length C; C % 8 == 0 (bf16) / % 4 (f32), and
 * C/8 (resp. C/4) <= 256 or a multiple of 256.
 *
 * coin_bn_stats     : batch mean and 1/sqrt(biased var + eps) over N*H*W (fp32, pivoted sums), and the
 *                     nn.BatchNorm2d running-statistics update (momentum, unbiased variance) when
 *                     running_mean/var are non-NULL.  sums_workspace: COIN_BN_MAX_PARTS*2*C floats (contents
 *                     undefined): per-workgroup partial sums combined in a fixed order (no atomics, reproducible).
 * coin_bn_apply_fwd : y = pool( relu?( (x-mean)*rstd*gamma + beta [+ residual] ) ); pool in {0,1,2};
 *                     pool == 2 writes y [N,H/2,W/2,C] (floor, as nn.AvgPool2d(2)); residual requires pool != 2.
 *                     pool == 0 writes only the spatial mean y [N,C] (the RoI head's `x.mean(dim=[2,3])`,
 *                     coin/modeling/roi_heads/clip_roi_heads.py:207-208): the activation itself is never stored.
 *                     num_batches_tracked (may be NULL): the module's int64 counter, incremented by the same launch that
 *                     updates the running statistics (nn.BatchNorm2d does `num_batches_tracked += 1` as a launch of its own).
 *                     relu_mask (may be NULL; pool 1 with a residual, pool 0): [N*H*W][C/8] bytes (C/4 for float32), bit i = channel i of
 *                     the 16-byte channel group passed the ReLU.  Given to coin_bn_bwd it replaces `y` (the saved output / the
 *                     residual input): the backward of a residual block then reads 1/16 of those bytes, in both of its passes.
 * coin_bn_bwd       : given dy (shape of y) computes dsums[0..C) = dbeta, dsums[C..2C) = dgamma (dsums must hold
 *                     (COIN_BN_MAX_PARTS+1)*2*C floats: the result followed by the partial sums), dx (shape of x)
 *                     and, if d_residual != NULL, d_residual = dy * relu'  (shape of y, pool == 1).
 *                     pool == 0: dy is [N,C], `y` must be the forward's RESIDUAL input (or NULL if there was none)
 *                     and d_residual, if requested, has the shape of x.
 *                     `y` (the saved forward output) supplies the ReLU mask when relu && pool == 1 and the forward
 *                     added a residual; pass y = NULL when it did not: the mask is then recomputed from x and the
 *                     output tensor is not read at all.
 * ---------------------------------------------------------------------------------------- */
#define COIN_BN_MAX_PARTS 768   /* ABI 3 (round 6): 512 before -- coin_bn_bwd reduces res5-sized tensors over 768 partial sums; workspaces are sized by this constant */
int coin_bn_stats(const void* x, int N, int H, int W, int C, float eps, float momentum, float* sums_workspace,
                  float* mean, float* rstd, float* running_mean, float* running_var, int64_t* num_batches_tracked, int dtype,
                  void* stream);
int coin_bn_apply_fwd(const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                      const void* residual, void* y, uint8_t* relu_mask, int N, int H, int W, int C, int relu, int pool, int dtype,
                      void* stream);
int coin_bn_bwd(const void* x, const void* dy, const void* y, const uint8_t* relu_mask, const float* mean, const float* rstd,
                const float* gamma, const float* beta, int N, int H, int W, int C, int relu, int pool, float* dsums, void* dx,
                void* d_residual, int dtype, void* stream);

/* nn.AvgPool2d(2) forward / backward on channels-last tensors (the anti-aliased shortcut of utils.py:71-75). */
int coin_avgpool2_fwd(const void* x, void* y, int N, int H, int W, int C, int dtype, void* stream);
int coin_avgpool2_bwd(const void* dy, void* dx, int N, int H, int W, int C, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Cosine-similarity classifier  (replaces FastRCNNOutputLayers.do_classify,
 *                                fast_rcnn.py:343-346)
 *   scores[r,k] = <f_r/|f_r|, t_k/|t_k|> * inv_scale          (inv_scale = 1/logit_scale = 100)
 * feats [R,D] dtype `dtype`; text [Kc,D] float32; scores [R,Kc] float32;
 * inv_norm_f [R] float32 (saved for backward; may be NULL in inference).  Kc <= 64.
 * ---------------------------------------------------------------------------------------- */
int coin_cosine_logits_fwd(const void* feats, int ldf, const float* text, int R, int D, int Kc,
                           float inv_scale, float* scores, float* inv_norm_f, int dtype,
                           void* stream);

/* d_feats[R,D] (dtype `dtype`) and d_text[Kc,D] (float32, ACCUMULATED) given d_scores[R,Kc].  workspace (required when d_text is
 * given): ceil(R / 16) * Kc * D floats -- every 16-row block's partial text gradient, added to d_text in block order by a second
 * launch (no float atomics: bit-reproducible). */
int coin_cosine_logits_bwd(const float* d_scores, const void* feats, int ldf, const float* text,
                           const float* scores, const float* inv_norm_f, int R, int D, int Kc,
                           float inv_scale, void* d_feats, float* d_text, int dtype, void* workspace, void* stream);

/* ------------------------------------------------------------------------------------------
 * bf16 GEMM / implicit-GEMM convolution of the res5 bottlenecks on the RoI tiles (coin/modeling/utils.py:77-90,184-186 as run by
 * coin/modeling/roi_heads/clip_roi_heads.py:172-176): replaces the cuDNN convolutions of `backbone.layer4` forward and
 * data-gradient (the weight gradient stays a library contraction).
 *   C[M,N] (bf16, row stride ldc) = Aop[M,K] . B[N,K]^T      fp32 accumulation, bf16 operands, K % 64 == 0
 *   mode 0: Aop = A, a row-major [M,K] matrix (row stride lda): 1x1 convolution on NHWC activations, nn.Linear;
 *   mode 1: implicit 3x3 / pad 1 / stride 1 convolution: A = NHWC activation [M/(H*W), H, W, Cin], K = 9*Cin ordered
 *           (ky, kx, ci) = the channels-last weight layout [Cout][3][3][Cin]; taps outside the image read zeros.
 * The data-gradient is the same contraction with the weight re-laid as [Cin][flipped tap][Cout].
 * R (optional, bf16 [M,N], row stride ldr): C = bf16(bf16(Aop.B^T) + R) -- the other branch's gradient where a Bottleneck's input
 *   fans out to conv1 and to the identity path (coin/modeling/utils.py:77-90): the sum autograd would form with a separate pass.
 * stats (optional): per (row tile, column) statistics of the STORED bf16 outputs over rows < stats_rows, as
 *   stats[tile][0][n] = pivot (the tile's first row), [1][n] = sum(x - pivot), [2][n] = sum((x - pivot)^2);
 *   coin_conv_gemm_stats_bytes(M, N) bytes (enough for either tile height).  A row tile is 256 rows on the persistent 256 x 256 core and
 *   128 rows on the small-map core (ABI 3; conv_gemm_s4.hip: launches whose 256 x 256 tiling has fewer than three rounds of tiles for
 *   the chip, and every N < 256): coin_conv_gemm_stats_tile_rows, called with the launch's own arguments, says which -- it is the
 *   dispatch function itself.  coin_conv_gemm_stats_finalize(tile_rows) turns the partials into the train-mode BatchNorm
 *   statistics (mean, 1/sqrt(var + eps), running statistics as nn.BatchNorm2d) in a fixed summation order: the separate
 *   statistics pass over the activation (coin_bn_stats) is not needed after a convolution run through this entry point. */
size_t coin_conv_gemm_stats_bytes(int M, int N);
int coin_conv_gemm_stats_tile_rows(int lda, int mode, int Cin, int ldb, int M, int N, int K);
int coin_conv_gemm_bf16(const void* A, int lda, int mode, int H, int W, int Cin, const void* B, int ldb,
                        void* C, int ldc, const void* R, int ldr, int M, int N, int K, float* stats, int64_t stats_rows,
                        void* stream);
/* The same with a caller-owned workspace (coin_conv_gemm_workspace_bytes; may be NULL / 0): when the output tiles do not fill the last
 * round of the persistent grid, the leftover tiles are cut along K, their fp32 partial tiles go through the workspace and are summed in
 * a fixed order (bit-reproducible).  Results equal coin_conv_gemm_bf16's up to the fp32 summation order of those tiles. */
size_t coin_conv_gemm_workspace_bytes(int M, int N, int K);
int coin_conv_gemm_bf16_ws(const void* A, int lda, int mode, int H, int W, int Cin, const void* B, int ldb,
                           void* C, int ldc, const void* R, int ldr, int M, int N, int K, float* stats, int64_t stats_rows,
                           void* workspace, size_t workspace_bytes, void* stream);

/* coin_conv_gemm_bf16_rpool: as coin_conv_gemm_bf16_ws without statistics, with C = bf16(bf16(A.B^T) + 0.25 * R[n][h / 2][w / 2]) for the
 * output pixel (n, h, w) of the grid [M / (out_h * out_w), out_h, out_w]; R: [M / (out_h * out_w) * (out_h / 2) * (out_w / 2), N] bf16 (floor
 * pooling: pixels whose h / 2 or w / 2 falls off the pooled map add nothing).  R = the gradient of `AvgPool2d(2)` applied to the tensor whose
 * data gradient this GEMM produces: the downsample branch of the CLIP Bottleneck (coin/modeling/utils.py:60-75, 84-88: `avgpool` +
 * `downsample`) -- the pool's backward pass folded into the residual add.  COIN_ESHAPE when neither the persistent nor the small-map kernel
 * serves the shape (K % 64, N % 8 ...): the caller then materialises the pool gradient (coin_avgpool2_bwd) and passes it as a plain residual. */
int coin_conv_gemm_bf16_rpool(const void* A, int lda, int mode, int H, int W, int Cin, const void* B, int ldb, void* C, int ldc, const void* R,
                              int ldr, int out_h, int out_w, int M, int N, int K, void* workspace, size_t workspace_bytes, void* stream);
int coin_conv_gemm_stats_finalize(const float* partials, int M, int N, int64_t rows, int tile_rows, float eps, float momentum,
                                  float* mean, float* rstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                  void* stream);

/* Weight gradient of the same convolutions: dW[Cout][Ktot] (fp32, OVERWRITTEN) = gy[M,Cout]^T . Acol[M,Ktot] with Acol as in
 * coin_conv_gemm_bf16 (mode 0: Ktot = Cin; mode 1: Ktot = 9*Cin, (ky, kx, ci) = the channels-last weight layout).  gy and x are the
 * NHWC tensors of the forward pass (bf16, row strides Cout / Cin); Cout % 128 == 0 and Cin % 128 == 0 (odd multiples of 128 -- half-valid
 * edge tiles, a 3x3 K-tile that spans two taps -- on the persistent kernel only, whose 32-bit buffer offsets bound the pixel count:
 * COIN_ESHAPE when (M + 320) * max(Cout, Cin) * 2 >= 0x7f000000).  Also the weight gradient of a
 * linear layer (mode 0: dW[N,K] = dZ[M,N]^T . X[M,K]).  The contraction over the M pixels is cut into per-XCD segments whose fp32
 * partial tiles go to `workspace` (coin_conv_wgrad_workspace_bytes) and are summed in a fixed order (no atomics: bit-reproducible). */
size_t coin_conv_wgrad_workspace_bytes(int M, int Cout, int Ktot);
int coin_conv_wgrad_bf16(const void* gy, const void* x, int mode, int H, int W, int Cin, int M, int Cout, int Ktot,
                         float* dW, void* workspace, void* stream);

/* Window attention forward of the Swin-T student (FPN extension, coin_amd/modeling/swin.py; no counterpart in the reference):
 *   out[b][i][h*32 + d] = sum_j softmax_j(scale * q[b][i][h].k[b][j][h] + bias[h][i][j] + mask[b % windows_per_image][i][j]) * v[b][j][h][d]
 * qkv bf16 [num_windows][tokens][3][heads][32]; bias float32 [heads][64][64] with columns >= tokens <= -1e30 (rows / columns padded to
 * 64); mask float32 [windows_per_image][64][64] or NULL; out bf16 [num_windows][tokens][heads*32].  tokens <= 64, head_dim == 32. */
int coin_window_attn_fwd(const void* qkv, const float* bias, const float* mask, void* out, int num_windows, int windows_per_image,
                         int heads, int tokens, int head_dim, float scale, void* stream);

/* Its backward (replaces the autograd of the same published formulation; no reference counterpart): given dout bf16
 * [num_windows][tokens][heads*32] -> dqkv bf16 (qkv's layout, every element written) and dbias float32 [heads][tokens][tokens]
 * (OVERWRITTEN; the sum over all windows, reduced through `workspace` in a fixed order: no atomics).  P is recomputed from qkv / bias /
 * mask with the forward's arithmetic; dQ, dK, dV, dP run on MFMA.  workspace: coin_window_attn_bwd_workspace_bytes, 16-byte aligned. */
size_t coin_window_attn_bwd_workspace_bytes(int num_windows, int heads);
int coin_window_attn_bwd(const void* qkv, const float* bias, const float* mask, const void* dout, void* dqkv, float* dbias,
                         void* workspace, int num_windows, int windows_per_image, int heads, int tokens, int head_dim, float scale,
                         void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused losses: each computes the scalar loss AND the gradient w.r.t. its differentiable
 * input for a unit upstream gradient, in one launch.  `loss` is a single float32 that is
 * OVERWRITTEN.  Rows are independent; per-row reductions use wavefront shuffles.
 * ---------------------------------------------------------------------------------------- */

/* MIL soft-target cross entropy (coin/utils/losses.py:13-34; call sites fast_rcnn.py:462-467,
 * 580).  NO max-subtraction (reference quirk, losses.py:15-18):
 *   p = exp(x) / sum(exp(x));  l_r = -log( sum_c t*p / (avg_positives ? sum_c t + 1e-6 : 1) ) * w_r
 *   loss = mean_r l_r  (reduction_mean) or sum_r l_r;   R == 0 -> loss = 0.
 * x [R,C] float32 (row stride ldx); exactly one of {target [R,C] float32, labels [R] int64
 * (one-hot target)} is non-NULL; weights [R] float32 or NULL;  grad_x [R,C] float32 or NULL. */
int coin_mil_ce_fwd_bwd(const float* x, int ldx, const float* target, const int64_t* labels,
                        const float* weights, int R, int C, int avg_positives, int reduction_mean,
                        float* loss, float* grad_x, void* stream);

/* MIL focal loss (coin/utils/losses.py:36-73; selected by CLOUD.LOSS_TYPE "MILFocalLoss", fast_rcnn.py:468-469,581-582,595-596):
 *   alpha_r = sum_c t*alpha_c / (sum_c t + 1e-6);  P = sum_c t*p / (avg_positives ? sum_c t + 1e-6 : 1),  p = softmax(x) without
 *   max-subtraction;  l_r = -alpha_r (1 - P)^gamma log P * w_r;  loss = mean_r l_r (reduction_mean; R == 0 -> NaN: torch's mean of
 *   an empty tensor) or sum_r l_r.  The reference passes no row weights (weights NULL); the fixed-shape sampler uses them as the
 *   row-validity mask.  alpha [C] float32 (the class weights); other arguments as coin_mil_ce_fwd_bwd. */
int coin_mil_focal_fwd_bwd(const float* x, int ldx, const float* target, const int64_t* labels,
                           const float* weights, const float* alpha, float gamma, int R, int C,
                           int avg_positives, int reduction_mean, float* loss, float* grad_x, void* stream);

/* nn.KLDivLoss(reduction='mean')(log(p + eps), q)  -  ELEMENT mean over R*C, not batchmean
 * (fast_rcnn.py:273,526,538,544; rpn.py:335):   loss = 1/(R*C) * sum q * (log q - log(p+eps)),
 * terms with q == 0 contribute 0.
 * mode 0: p = softmax(x) over C (x are logits);       grad_x = d loss / d x
 * mode 1: p = x (already probabilities);              grad_x = d loss / d p
 * mode 2: binary, C must be 2 on the q side: x [R] logits, p = [sigmoid(x), 1-sigmoid(x)],
 *         q [R] teacher prob -> [q, 1-q]  (rpn.py:331-335); grad_x [R].
 * row_mask [R] uint8 or NULL selects rows (masked rows are excluded from R in the mean). */
int coin_kl_div_fwd_bwd(const float* x, int ldx, const float* q, int ldq, const uint8_t* row_mask,
                        int R, int C, int mode, float eps, float* loss, float* grad_x,
                        void* stream);

/* Box regression L1 (FastRCNNOutputLayers.box_reg_loss, fast_rcnn.py:601-646, smooth_l1 with
 * beta = 0; Box2BoxTransform.get_deltas with weights (wx,wy,ww,wh)):
 *   fg rows: 0 <= gt_classes[r] < num_fg_classes
 *   loss = sum_{fg r} sum_j | pred_deltas[r,j] - get_deltas(proposal_r, gt_r)_j | / normalizer
 * proposals, gt_boxes [R,4] float32 xyxy; pred_deltas [R,4] float32 (class-agnostic);
 * grad_deltas [R,4] float32 (zero on non-fg rows) or NULL. */
int coin_box_reg_l1_fwd_bwd(const float* proposals, const float* gt_boxes, const float* pred_deltas,
                            const int64_t* gt_classes, int R, int num_fg_classes,
                            float wx, float wy, float ww, float wh, float normalizer,
                            float* loss, float* grad_deltas, void* stream);

/* mean |a - b| over n elements (nn.L1Loss(reduction='mean'), fast_rcnn.py:351 loss_text_align);
 * grad_a = sign(a-b)/n or NULL. */
int coin_l1_mean_fwd_bwd(const float* a, const float* b, int64_t n, float* loss, float* grad_a,
                         void* stream);

/* RPN objectness BCE-with-logits, sum over anchors with labels >= min_label, and the anchor
 * L1 localisation loss over labels == 1 (DualTeacherRPN.losses, rpn.py:300-324):
 *   loss_cls = sum_{label>=min_label} bce(logit, label) ; loss_loc = sum_{label==1} |d - get_deltas(anchor, gt)|_1
 * Box2BoxTransform weights (1,1,1,1) (detectron2 MODEL.RPN.BBOX_REG_WEIGHTS default).
 * logits [A_total] float32, labels [A_total] int8 (-1 ignore, 0 neg, 1 pos), deltas [A_total,4],
 * anchors [A_per_image,4] (broadcast over images), matched_gt [A_total,4].
 * Outputs are SUMS (caller divides by batch_size_per_image * num_images).
 * grad_logits [A_total], grad_deltas [A_total,4] (unit upstream for each loss) or NULL. */
#define COIN_RPN_LOSS_MAX_BLOCKS 1024
/* workspace: 2 * COIN_RPN_LOSS_MAX_BLOCKS floats (per-block partial sums, joined in block order: no float atomics). */
int coin_rpn_losses_fwd_bwd(const float* logits, const int8_t* labels, const float* deltas,
                            const float* anchors, const float* matched_gt, int64_t A_total,
                            int64_t A_per_image, int min_label, float* loss_cls, float* loss_loc,
                            float* grad_logits, float* grad_deltas, void* workspace, void* stream);

/* ------------------------------------------------------------------------------------------
 * Batched greedy NMS  (replaces torchvision.ops.nms / batched_nms under detectron2
 *                      find_top_rpn_proposals, coin/modeling/proposal_generator/rpn.py:113-115, and
 *                      fast_rcnn_inference_single_image, coin/modeling/roi_heads/fast_rcnn.py:164)
 * boxes   : [B, n_max, 4] float32 xyxy, each image's first counts[b] rows sorted by DESCENDING score
 * counts  : [B] int32 (device)
 * keep    : [B, n_max] int32: the first num_keep[b] entries are the surviving row indices, in score
 *           order, truncated to max_keep;  num_keep: [B] int32.
 * A box is suppressed when IoU > iou_threshold with an earlier surviving box.
 * workspace: coin_nms_workspace_bytes(B, n_max) bytes of device memory (contents undefined).
 * n_max <= 16384.  Everything stays on the device (no host round trip).
 * ---------------------------------------------------------------------------------------- */
size_t coin_nms_workspace_bytes(int B, int n_max);
int coin_nms_batched(const float* boxes, const int* counts, int B, int n_max, float iou_threshold,
                     int max_keep, void* workspace, int* keep, int* num_keep, void* stream);

/* ------------------------------------------------------------------------------------------
 * Anchor / proposal labelling  (replaces detectron2 Matcher(pairwise_iou(gt, anchors)) and the fg/bg subsampling under
 *                               DualTeacherRPN.label_and_sample_anchors, coin/modeling/proposal_generator/rpn.py:120-254,
 *                               and label_and_sample_proposals, coin/modeling/roi_heads/clip_roi_heads.py:283-399)
 * coin_anchor_match: for every image i and anchor a
 *   matched[i][a] = argmax_g IoU(gt_i[g], anchor_a)  (lowest g among equal maxima, as torch.max), and labels[i][a] =
 *   label_lo if max < lo, label_mid if lo <= max < hi, label_hi if max >= hi (pass lo == hi for a single threshold);
 *   with allow_low_quality: labels = 1 where IoU(gt_i[g], anchor_a) == max_a' IoU(gt_i[g], anchor_a') for some g.
 *   IoU in fp32 with individually rounded operations in pairwise_iou's order: labels and indices are bit-exact with the
 *   composed torch ops.  An image without boxes gets matched = 0, labels = empty_label, matched_boxes = 0.
 * gt_boxes        : [sum G_i, 4] float32 xyxy (device), the images' boxes concatenated;
 * gt_offsets_host : HOST array of num_images + 1 ints, image i owns rows [off[i], off[i+1]); G_i <= 512, num_images <= 64
 * anchors         : [A, 4] float32 (device), shared by the images -- or, with anchors_per_image != 0, [num_images, A, 4]: image i is
 *                   matched against its own set (the RoI samplers: every image's proposals ++ its teacher boxes, one launch per batch)
 * matched         : [num_images, A] int64;  labels: [num_images, A] int8;  matched_boxes (optional): [num_images, A, 4]
 * workspace       : 4 * sum G_i bytes of device memory when allow_low_quality (contents undefined), else may be NULL.
 *
 * coin_sample_labels: per image the k_pos = min(#pos, pos_cap) positives and k_neg = min(#neg, num_samples - k_pos)
 *   negatives with the smallest random key (ties: lowest index) -- the subsets an ascending stable sort by key would rank
 *   first, found by a radix select.  Keys are compared as floor(key * 2^24) (torch.rand's resolution; keys in [0, 1)).  cls [num_images, M] int8 or int64: -1 ignore, bg_label negative, else positive;
 *   keys [num_images, M] float32 >= 0;  out [num_images, M] int8 = 1 (chosen positive) / 0 (chosen negative) / -1.
 * ---------------------------------------------------------------------------------------- */
int coin_anchor_match(const float* gt_boxes, const int* gt_offsets_host, int num_images, const float* anchors, int A,
                      int anchors_per_image, float lo, float hi, int label_lo, int label_mid, int label_hi, int empty_label,
                      int allow_low_quality, int64_t* matched, int8_t* labels, float* matched_boxes, void* workspace,
                      void* stream);
int coin_sample_labels(const void* cls, int cls_is_int64, const float* keys, int num_images, int M, int bg_label,
                       int num_samples, int pos_cap, int8_t* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Two-view input augmentation on the device  (replaces the Pillow calls under DatasetMapperUnsupervised.__call__,
 *   coin/data/dataset_mapper.py:363-450: detectron2 ResizeShortestEdge / RandomFlip -> PIL.Image.resize(BILINEAR), np.flip;
 *   coin/data/detection_utils.py:22-45 strong augmentation: torchvision ColorJitter / RandomGrayscale on PIL images ->
 *   ImageEnhance.{Brightness,Contrast,Color}, HSV round trip, convert("L");  coin/data/transforms/augmentation_impl.py:64-92:
 *   ImageFilter.GaussianBlur, ImageOps.solarize)
 * Images are uint8 [H, W, 3] RGB interleaved in device memory.  Byte / integer work: every result is BIT-EXACT with Pillow
 * (restated and pinned against Pillow 12.2 in oracle/augment.py).
 * coin_aug_resize_bilinear_u8: Image.resize((out_w, out_h), BILINEAR) (+ horizontal flip of the result when flip_h);
 *   tmp: H * out_w * 3 bytes, needed when both sizes change.
 * coin_aug_point_op_u8: one of the COIN_AUG_* point operations; fparam = enhancement factor (brightness / contrast /
 *   saturation), iparam = hue shift in [0, 255] (= uint8(hue_factor * 255)) or the solarize threshold; contrast needs 16 bytes
 *   of 8-byte aligned device workspace (the image's mean luma is reduced on the device, no host round trip);
 *   out_chw != 0 writes [3, H, W] planes (the layout coin_normalize_pad reads) instead of interleaved pixels.
 * coin_aug_gaussian_blur_u8: ImageFilter.GaussianBlur(radius) = 3 horizontal + 3 vertical extended box blurs;
 *   tmp: H * W * 3 bytes; src != dst.
 * ---------------------------------------------------------------------------------------- */
typedef enum {
  COIN_AUG_COPY = 0, COIN_AUG_BRIGHTNESS = 1, COIN_AUG_CONTRAST = 2, COIN_AUG_SATURATION = 3, COIN_AUG_HUE = 4,
  COIN_AUG_GRAYSCALE = 5, COIN_AUG_SOLARIZE = 6
} coin_aug_op;
int coin_aug_resize_bilinear_u8(const uint8_t* src, int H, int W, uint8_t* dst, int out_h, int out_w, int flip_h,
                                uint8_t* tmp, void* stream);
int coin_aug_point_op_u8(const uint8_t* src, uint8_t* dst, int H, int W, int op, float fparam, int iparam,
                         void* workspace, int out_chw, void* stream);
int coin_aug_gaussian_blur_u8(const uint8_t* src, uint8_t* dst, int H, int W, float radius, uint8_t* tmp, void* stream);

/* ------------------------------------------------------------------------------------------
 * Input normalisation  (replaces OpenVocabularyRCNN.preprocess_image, clip_rcnn.py:287-298:
 *                       ToTensor + Normalize + ImageList.from_tensors zero padding)
 * img   : [3,h,w] uint8 (CHW, as the dataset mapper emits); mean/std_ are HOST arrays of 3 floats
 * out   : image `n` of a batch [Nb,3,Hp,Wp] (COIN_NCHW) or [Nb,Hp,Wp,3] (COIN_NHWC), dtype
 *         `dtype`; pixels outside h x w are written as 0.
 * ---------------------------------------------------------------------------------------- */
int coin_normalize_pad(const uint8_t* img, int h, int w, const float mean[3], const float std_[3],
                       void* out, int n, int Hp, int Wp, int layout, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused SGD with momentum over a table of parameter tensors (replaces the per-tensor loop of
 * torch.optim.SGD over ~170 param groups, coin/solver/build.py:96-103, engine/pre_train.py:201):
 *   g = grad * inv_loss_scale + wd * p ; buf = momentum * buf + g (buf = g on first step) ; p -= lr * lr_scale * buf
 * (`lr_scale` = the schedule factor common to all groups, so the device table is uploaded once, not every step.)
 * `table` is a DEVICE array of coin_sgd_tensor descriptors; one launch updates all of them.
 * Optionally also refreshes a bf16 shadow copy of each parameter (shadow may be NULL).
 * `gate` (DEVICE float, may be NULL): when *gate == 0 the launch changes nothing -- the data-parallel CKG update (trainer.py:192-197
 * under DistributedDataParallel, :66-72) is taken by every rank or by none, and "did any rank see B boxes" is the all-reduced count
 * that rides in the gradient arena: the decision never returns to the host.
 * ---------------------------------------------------------------------------------------- */
typedef struct coin_sgd_tensor {
  float* param;
  const float* grad;
  float* momentum_buf;
  uint16_t* bf16_shadow; /* may be NULL */
  int64_t numel;
  float lr;
  float weight_decay;
} coin_sgd_tensor;

int coin_sgd_step(const coin_sgd_tensor* table, int num_tensors, int64_t max_numel, float momentum,
                  float inv_loss_scale, float lr_scale, int first_step, const float* gate, void* stream);

/* Data-gradient layout of convolution / linear weights, all tensors of a DEVICE table in one launch (replaces the per-call
 * flip + permute + copy of the weight in the backward of the res5 / RPN / box-head layers, coin/modeling/utils.py:77-90 under autograd):
 *   dst[ci][ks-1-ky][ks-1-kx][co] = src[co][ky][kx][ci]     bf16; ks = 1 is the plain transpose of a linear weight [N, K].
 * cout % 8 == 0, cin % 8 == 0, 16-byte aligned src / dst.  max_tiles >= ks*ks*ceil(cout/64)*ceil(cin/64) of every entry. */
typedef struct coin_wd_tensor {
  const uint16_t* src;
  uint16_t* dst;
  int32_t cout, cin, ks, reserved;
} coin_wd_tensor;
int coin_weight_dgrad_layout(const coin_wd_tensor* table, int num_tensors, int max_tiles, void* stream);

/* ------------------------------------------------------------------------------------------
 * Teacher EMA (replaces EnsembleTSModel.update_params, coin/modeling/meta_arch/ts_ensemble.py:39-69)
 *   teacher = student * (1 - keep) + teacher * keep      over a table of float32 tensors.
 * ---------------------------------------------------------------------------------------- */
typedef struct coin_ema_tensor {
  float* teacher;
  const float* student;
  int64_t numel;
} coin_ema_tensor;

int coin_ema_update(const coin_ema_tensor* table, int num_tensors, int64_t max_numel, float keep,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif /* COIN_HIP_H */
