/* mtn.h -- C ABI of the MI355X-native MultiTaskNet hot path (libmtn_hip.so).
 *
 * The reference (yingkunwu/hand-gesture-recognition) has no FFI layer: its boundary is the
 * Python nn.Module `MultiTaskNet` (model/multitasknet.py:8-29) whose forward issues PyTorch
 * ATen ops.  Each entry point below replaces the ATen op sequence of the cited reference lines;
 * the Python host side (hand-gesture-recognition_amd/) binds them with ctypes.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero otherwise (mtn_last_error() has the text);
 *     no C++ exception crosses the boundary, nothing calls exit().
 *   - enqueue-only on the caller's HIP stream (`stream` is a hipStream_t passed as void*);
 *     no allocation, no synchronisation, no retained pointers -> safe under hipGraph capture.
 *   - the caller (PyTorch) owns every buffer incl. workspaces (`*_workspace_bytes` sizes them).
 *   - dtype: MTN_DTYPE_F32 = fp32 storage + exact-f32 MFMA (validation mode),
 *            MTN_DTYPE_BF16 = bf16 storage + bf16 MFMA with fp32 accumulation (throughput mode),
 *            MTN_DTYPE_F32X3 = SPLIT storage, 4 bytes per element (ABI 5): every 32-channel block of a row -- 128
 *            bytes -- holds [32 bf16 hi][32 bf16 lo], hi = bf16_rne(x), lo = bf16_rne(x - hi), channels in natural
 *            order (x = hi + lo to ~2^-17).  Every entry point that writes an activation / gradient tensor under
 *            this dtype writes that layout (each element is split ONCE, by its producer), every one that reads
 *            one expects it; the GEMM-shaped entry points (mtn_conv_fwd / _dgrad / _dgrad_bn / _wgrad,
 *            mtn_linear_fwd / _gelu_bwd / _wgrad, mtn_attention_*) take their MFMA operands from it without any
 *            conversion and multiply with three bf16 MFMAs per product (hi*hi + hi*lo + lo*hi, fp32
 *            accumulate; ~2^-17 relative error per product) -- the 1e-3 tolerance of the reference's fp32
 *            training (train.py:19,230: fp32 with TF32 matmuls allowed) at throughput.  Tensors start on a 16-byte
 *            boundary at a channel that is a multiple of 32; channel counts and row strides are multiples of 32.  mtn_x3_convert moves
 *            a tensor between this layout and plain f32 (module boundary).  EXCEPTIONS, which take / write PLAIN
 *            f32 under this dtype: the image, every f32-typed argument (statistics, parameters, logits,
 *            heat-maps, attention map, lse), the unfused stem entries mtn_conv1_fwd / _wgrad, and the token
 *            tensors of mtn_pose_head_fwd / _bwd (convert at that boundary).
 *     Statistics, gradients of parameters, optimizer state and losses are always fp32.
 *   - activations are NHWC ("channels last") / token-major [rows][channels]; `ld*` is the
 *     element stride between consecutive rows (pixels), so an op can read or write a channel
 *     slice of a wider buffer (this is how torch.cat / chunk of model/gelan.py:138,142 are free).
 *   - pointers and row strides must keep 16-byte alignment (split storage: row strides of 32 elements, above).
 */
#ifndef MTN_H_
#define MTN_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MTN_DTYPE_F32 0
#define MTN_DTYPE_BF16 1
#define MTN_DTYPE_F32X3 2
#define MTN_ACT_NONE 0
#define MTN_ACT_SILU 1
#define MTN_ACT_GELU 2

const char* mtn_last_error(void);
int mtn_abi_version(void);

/* ---- GELAN backbone: Conv2d(bias=False) -> BatchNorm2d(train) -> SiLU  (model/gelan.py:37-56) */

/* number of row tiles whose partial BN statistics mtn_conv_fwd writes: stats is [tiles][2][Cout]
 * (the tile shape is chosen per layer from its whole geometry).  mtn_conv_stats_tiles(M, Cout, K) is
 * the geometry-free form for 1x1 convolutions / Linear-shaped calls (M = B*OH*OW, K = Cin). */
int mtn_conv_fwd_stats_tiles(int dtype, int B, int H, int W, int Cin, int Cout, int ksize, int stride);
int mtn_conv_stats_tiles(int M, int Cout, int K);

/* y_raw = conv(x) (k in {1,3}, pad k/2, stride in {1,2}); w packed [Cout][k][k][Cin];
 * stats (optional) receives per-tile per-channel sum / sum-of-squares of y_raw; stats_tiles is the
 * row count the caller sized it for and must equal mtn_conv_fwd_stats_tiles(...) (checked:
 * the kernel writes exactly that many rows and mtn_bn_finalize reads them). */
int mtn_conv_fwd(const void* x, int ldx, const void* w, void* y, int ldy, float* stats, int stats_tiles,
                 int dtype, int B, int H, int W, int Cin, int Cout, int ksize, int stride,
                 void* stream);

/* Inference form of a Conv unit with its BatchNorm folded away (model/gelan.py:56, 78-87 in eval()):
 * y = act(conv(x) + bias [+ res]), act in {NONE, SILU}; w = the folded weights (mtn_pack_weights with a
 * scale), bias f32 [Cout] = beta - running_mean * scale.  One launch per unit, no raw output, no BN pass. */
int mtn_conv_bias_act_fwd(const void* x, int ldx, const void* w, const float* bias, int act, const void* res,
                          int ldres, void* y, int ldy, int dtype, int B, int H, int W, int Cin, int Cout,
                          int ksize, int stride, void* stream);

/* dx = conv_transpose(dy) [+ res]; wt packed [Cin][k][k][Cout]; (H, W) is the INPUT size. */
int mtn_conv_dgrad(const void* dy, int lddy, const void* wt, void* dx, int lddx,
                   const void* res, int ldres, int dtype, int B, int H, int W, int Cin, int Cout,
                   int ksize, int stride, void* stream);

/* The same data gradient when dx [+ res] is the OUTPUT gradient of another Conv unit
 * act(bn(conv)) [+ its residual] (model/gelan.py:56,78-87; in the backward pass nearly every dx is):
 * while each tile of dx is still on chip its epilogue also reads that unit's raw conv output bn_y
 * (same [B][H][W][Cin] rows as dx) and writes the partial sums of its BatchNorm backward --
 * sum dz, sum dz*xhat with dz = dx * act'(bn_y*scale + shift [+ bn_res]) -- to
 * bn_partial[bn_tiles][2][Cin] (allocate mtn_slab_rows(bn_tiles) rows).  mtn_bn_act_bwd called with
 * partial_rows = bn_tiles then skips its own reduce pass over (dout, y): one read of dx and one
 * launch less per layer.  bn_tiles must equal mtn_conv_dgrad_bn_tiles(...) (checked). */
int mtn_conv_dgrad_bn_tiles(int dtype, int B, int H, int W, int Cin, int Cout, int ksize, int stride);
/* (ABI 6) what a successful mtn_conv_dgrad_bn wrote, in memory the CALLER owns: it hands the record to the mtn_bn_act_bwd that
 * consumes the rows, which checks its own arguments against it and clears it (a hand-over is used once).  The library keeps
 * nothing between the two calls (until ABI 5 it kept this record itself, per thread). */
typedef struct mtn_bn_handover { const float* partial; int tiles; int channels; int dtype; } mtn_bn_handover;
int mtn_conv_dgrad_bn(const void* dy, int lddy, const void* wt, void* dx, int lddx, const void* res,
                      int ldres, const void* bn_y, int bn_ldy, const void* bn_res, int bn_ldres,
                      const float* bn_scale_shift, const float* bn_mean_rstd, int bn_act,
                      float* bn_partial, int bn_tiles, mtn_bn_handover* record, int dtype, int B, int H, int W,
                      int Cin, int Cout, int ksize, int stride, void* stream);

/* Stem Conv(3,64,k3,s2) (model/gelan.py:155): reads the NCHW fp32 image batch directly, writes
 * the NHWC raw output [B][H/2][W/2][64] + stats[mtn_conv1_stats_tiles][2][64]; w packed [64][3][3][3]
 * as (co, kh, kw, ci).  The image needs no gradient, so there is no conv1 dgrad. */
int mtn_conv1_stats_tiles(int B, int H, int W);
int mtn_conv1_fwd(const float* x_nchw, const void* w, void* y, float* stats, int dtype, int B, int H,
                  int W, void* stream);
/* dw f32 [64][3][3][3] OIHW (overwritten); partial: f32 [mtn_conv1_wgrad_partial_rows][1728]. */
int mtn_conv1_wgrad_partial_rows(int B, int H, int W);
int mtn_conv1_wgrad(const float* x_nchw, const void* dy, float* dw, float* partial, int dtype, int B,
                    int H, int W, void* stream);

/* Fused stem for training: the RAW conv output (the largest tensor of the network) is never stored.
 *   mtn_conv1_stats   conv -> stats[mtn_conv1_stats_tiles][2][64] only (BN partial sums of the fp32 results)
 *   mtn_conv1_bn_act  conv -> out = act(conv * scale + shift), NHWC [B][H/2][W/2][64]  (scale_shift[2][64]
 *                     from mtn_bn_finalize, or mtn_bn_eval_scale_shift in eval mode)
 *   mtn_conv1_bwd     backward of act(bn(conv(x))) in ONE pass over (x, dout): recomputes the conv,
 *                     dgamma / dbeta [64] and dw f32 [64][3][3][3] OIHW (all overwritten) -- replaces
 *                     BN-backward reduce + apply + mtn_conv1_wgrad.  partial: f32
 *                     [mtn_conv1_bwd_partial_rows][mtn_conv1_bwd_partial_width] workspace. */
int mtn_conv1_stats(const float* x_nchw, const void* w, float* stats, int dtype, int B, int H, int W,
                    void* stream);
int mtn_conv1_bn_act(const float* x_nchw, const void* w, const float* scale_shift, void* out, int act,
                     int dtype, int B, int H, int W, void* stream);
int mtn_conv1_bwd_partial_rows(int B, int H, int W);
int mtn_conv1_bwd_partial_width(void);
int mtn_conv1_bwd(const float* x_nchw, const void* w, const void* dout, const float* scale_shift,
                  const float* mean_rstd, float* partial, float* dgamma, float* dbeta, float* dw,
                  int act, int dtype, int B, int H, int W, void* stream);
/* Gradient of the IMAGE through the stem (autograd parity with the reference's plain module, multitasknet.py:24-29; the
 * training step never asks for it): dx f32 NCHW [B][3][H][W], overwritten.  w_oihw: the fp32 master weights [64][3][3][3];
 * dout / scale_shift / mean_rstd as for mtn_conv1_bwd; dgamma / dbeta [64]: what mtn_conv1_bwd wrote for the same (x, dout). */
int mtn_conv1_dx(const float* x_nchw, const float* w_oihw, const void* dout, const float* scale_shift,
                 const float* mean_rstd, const float* dgamma, const float* dbeta, float* dx_nchw, int act, int dtype,
                 int B, int H, int W, void* stream);

/* ---- nn.Linear / 1x1 projections  (model/transformer.py:34-37,56-58; multitasknet.py:13) ------
 * out = act(x @ w^T + bias) + res ; w is [N][K]; aux_out (optional) gets the pre-activation. */
int mtn_linear_fwd(const void* x, int ldx, const void* w, const float* bias, int act,
                   void* aux_out, int ld_aux, const void* res, int ldres, void* out, int ldo,
                   int dtype, int M, int N, int K, void* stream);

/* ---- weight gradients (autograd of the Conv2d / Linear sites above) ----------------------------
 * dw is written (not accumulated) in the reference layout: Conv2d OIHW [Cout][Cin][k][k],
 * Linear [N][K].  x is the forward input (NHWC), dy the gradient of the raw conv output.
 * The pixel axis is split over workgroups; each writes its f32-accumulated partial tile to the workspace and a second
 * launch sums the splits in f32 in a fixed order.  MTN_DTYPE_BF16 stores the partial tiles as bf16 (one rounding of an
 * f32 sum, the precision class of a bf16 autocast weight gradient); F32 / F32X3 store them as f32. */
size_t mtn_conv_wgrad_workspace_bytes(int dtype, int B, int H, int W, int Cin, int Cout, int ksize,
                                      int stride);
int mtn_conv_wgrad(const void* x, int ldx, const void* dy, int lddy, float* dw, void* workspace,
                   size_t ws_bytes, int dtype, int B, int H, int W, int Cin, int Cout, int ksize,
                   int stride, void* stream);
size_t mtn_linear_wgrad_workspace_bytes(int dtype, int M, int N, int K);
int mtn_linear_wgrad(const void* x, int ldx, const void* dy, int lddy, float* dw, void* workspace,
                     size_t ws_bytes, int dtype, int M, int N, int K, void* stream);
/* (ABI 7) the Linear weight gradients of a whole backward pass -- the `.weight.grad` of every nn.Linear of the four Transformer
 * layers (model/transformer.py:44,58-60: to_qkv, to_out, the two FeedForward Linears), which autograd produces one by one --
 * as ONE launch plus one split sum: dw_i[N_i][K_i] = dy_i[M][N_i]^T x_i[M][K_i], i < count, all over the same M rows.
 * Nothing downstream reads a weight gradient before the optimizer step, so the caller keeps every (x_i, dy_i) alive until
 * this call.  Results are written (not accumulated); the split sums run in a fixed order.  Groups the kernel cannot take in one
 * launch (a side that is no multiple of 128) run as `count` mtn_linear_wgrad launches on the same workspace; more than 16
 * problems run 16 at a time. */
typedef struct mtn_wgrad_item {
    const void* x;  int ldx;      /* [M][ldx]  forward input, K_i columns used */
    const void* dy; int lddy;     /* [M][lddy] output gradient, N_i columns used */
    float* dw;                    /* [N_i][K_i] f32 */
    int N, K;
} mtn_wgrad_item;
/* (ABI 7) ... and the weight gradients of `count` Conv2d(k=3, s=1, p=1) layers of ONE geometry -- the four 3x3 convolutions of the two
 * residual blocks of a GELAN stage (model/gelan.py:69-70,78-87,127-142) -- likewise: items[i].N = Cout, items[i].K = Cin for every i.
 * Other geometries run as `count` mtn_conv_wgrad launches inside the call. */
size_t mtn_conv_wgrad_group_workspace_bytes(int dtype, int count, int B, int H, int W, int Cin, int Cout, int ksize, int stride);
int mtn_conv_wgrad_group(const mtn_wgrad_item* items, int count, void* workspace, size_t ws_bytes, int dtype, int B, int H, int W,
                         int Cin, int Cout, int ksize, int stride, void* stream);
size_t mtn_linear_wgrad_group_workspace_bytes(int dtype, int M, const mtn_wgrad_item* items, int count);
int mtn_linear_wgrad_group(const mtn_wgrad_item* items, int count, int M, void* workspace, size_t ws_bytes, int dtype,
                           void* stream);

/* ---- train-mode BatchNorm2d + SiLU (+ residual)  (model/gelan.py:46,56,78-87) -------------------
 * Slabs of f32 partial sums ([rows][W]) are reduced in row order by the library; a slab that the
 * caller allocates for `rows` producer rows must have mtn_slab_rows(rows) rows (scratch tail). */
int mtn_slab_rows(int rows);

/* stats[tiles][2][C] (from mtn_conv_fwd / mtn_conv1_fwd) -> scale_shift[2][C] = (gamma*rstd,
 * beta - mean*gamma*rstd), mean_rstd[2][C]; updates running_mean/var (unbiased var, momentum) and
 * num_batches_tracked (+1) in place when given.  count = B*OH*OW. */
int mtn_bn_finalize(float* stats, int tiles, int C, long long count, const float* gamma,
                    const float* beta, float* running_mean, float* running_var,
                    long long* num_batches_tracked, float momentum, float eps, float* scale_shift,
                    float* mean_rstd, void* stream);

/* eval mode (module.eval()): scale_shift from the running statistics instead of batch statistics. */
int mtn_bn_eval_scale_shift(const float* gamma, const float* beta, const float* running_mean,
                            const float* running_var, float eps, float* scale_shift, int C,
                            void* stream);

/* out = act(y*scale + shift [+ res]), act in {NONE, SILU}. */
int mtn_bn_act_fwd(const void* y, int ldy, const float* scale_shift, const void* res, int ldres,
                   void* out, int ldo, int dtype, int M, int C, int act, void* stream);

/* Backward of the above + BatchNorm: dout -> dy (gradient of the raw conv output), dgamma, dbeta;
 * dres (optional) = gradient of the residual input [+ extra].  partial: f32
 * [mtn_bn_bwd_partial_rows(M)][2][C]; coef: f32 [2][C] scratch.  partial_rows = 0: the sums are
 * reduced here from (dout, y); partial_rows > 0: partial already holds that many rows of them
 * (mtn_conv_dgrad_bn wrote dout and the sums; the slab has mtn_slab_rows(partial_rows) rows).
 * `record` (may be NULL when partial_rows = 0): the mtn_bn_handover that mtn_conv_dgrad_bn filled.  A partial_rows > 0
 * call whose slab / rows / C differ from the record, or whose record is empty (no such launch, or already consumed), is
 * REFUSED; the record is cleared by the call either way. */
int mtn_bn_bwd_partial_rows(int M);
int mtn_bn_act_bwd(const void* dout, int lddo, const void* y, int ldy, const void* res, int ldres,
                   const float* scale_shift, const float* mean_rstd, float* partial, float* coef,
                   float* dgamma, float* dbeta, void* dy, int lddy, void* dres, int lddres,
                   const void* extra, int ldextra, int partial_rows, mtn_bn_handover* record, int dtype, int M,
                   int C, int act, void* stream);

/* ---- ViT encoder (model/transformer.py), dim 256 ------------------------------------------------ */

/* (ABI 6) the 1x1 projection (model/multitasknet.py:13,26) and the token assembly (model/transformer.py:132-139) as ONE launch
 * (SURVEY K5): tok f32/T [B][H*W + 1][Cout] receives, per image, the class token in row 0 and conv1x1(x)[p] + pos[p] in row
 * 1 + p; pos f32 [H*W][Cout], cls f32 [Cout]; x [B][H][W][Cin] with pixel stride ldx, w packed [Cout][Cin].  Cout % 128 == 0.
 * (= mtn_conv_fwd(k = 1) into a feature tensor + mtn_tokens_fwd, without the feature tensor's round trip.) */
int mtn_proj_tokens_fwd(const void* x, int ldx, const void* w, const float* pos, const float* cls, void* tok, int dtype,
                        int B, int H, int W, int Cin, int Cout, void* stream);
/* tokens[b][0] = cls_token, tokens[b][1+p] = feat[b*P+p] + pos[p]   (transformer.py:132-139) */
int mtn_tokens_fwd(const void* feat, const float* pos, const float* cls, void* tok, int dtype, int B,
                   int P, void* stream);
/* dfeat[b*P+p] = dtok[b][1+p]; dcls[256] = sum_b dtok[b][0] */
int mtn_tokens_bwd(const void* dtok, void* dfeat, float* dcls, int dtype, int B, int P, void* stream);

/* nn.LayerNorm(256) (transformer.py:33,54): y = (x-mean)*rstd*gamma+beta; mean_rstd f32 [M][2]. */
int mtn_layernorm_fwd(const void* x, int ldx, const float* gamma, const float* beta, void* y, int ldy,
                      float* mean_rstd, int dtype, int M, float eps, void* stream);
/* rows of the f32 partial slab ([rows][512] for LayerNorm, [rows][256] for bias gradients) */
int mtn_rowslab_rows(int M);
/* ... of which the kernels themselves write the first mtn_rowslab_producer_rows(M) (the rest is folding scratch) */
int mtn_rowslab_producer_rows(int M);
/* dx = LN'(dy) [+ dres] (dx may alias dres); dgamma_dbeta f32 [2][256] (weight grad, then bias grad) */
int mtn_layernorm_bwd(const void* dy, int lddy, const void* x, int ldx, const float* mean_rstd,
                      const float* gamma, const void* dres, int lddres, void* dx, int lddx,
                      float* partial, float* dgamma_dbeta, int dtype, int M, void* stream);
/* da = dh * gelu'(a), exact erf GELU (transformer.py:35); n elements, contiguous */
int mtn_gelu_bwd(const void* dh, const void* a, void* da, int dtype, long long n, void* stream);
/* The backward of `Linear -> GELU -> Linear` (FeedForward, transformer.py:34-37) around the GELU in one
 * launch: da = (dh @ wt^T) * gelu'(a) -- the data gradient of the second Linear (wt = its weight packed
 * [N][K], dh [M][K]) with the GELU backward applied in its epilogue -- and dbias[N] = column sums of da
 * as stored (the bias gradient of the first Linear).  Replaces mtn_linear_fwd + mtn_gelu_bwd +
 * mtn_bias_grad.  partial: f32 [mtn_slab_rows(tiles)][2][N], tiles = mtn_linear_gelu_bwd_tiles(M, N, K). */
int mtn_linear_gelu_bwd_tiles(int M, int N, int K);
int mtn_linear_gelu_bwd(const void* dh, int lddh, const void* wt, const void* a, int lda, void* da,
                        int ldda, float* partial, int tiles, float* dbias, int dtype, int M, int N,
                        int K, void* stream);
/* Deferred parameter-gradient folds.  mtn_layernorm_bwd (dgamma_dbeta = NULL), mtn_bias_grad and mtn_linear_gelu_bwd
 * (dbias = NULL) then leave their per-workgroup partial sums in `partial` -- these gradients are leaves of the backward
 * graph -- and ONE call folds any number of such slabs: out[c] = sum over rows of in[r][c], in a fixed order (two levels:
 * blocks of mtn_colsum_rows_per_block() rows into the slab's scratch tail, then the rest; slabs of at most
 * mtn_colsum_direct_rows() rows skip level 1).  jobs_l1 / jobs_l2: device tables, one 32-byte record per slab:
 * {const float* in; float* out; int32 rows, W, ld, rows_per_block}; the level-1 record of a slab writes
 * out = in + rows*ld (its scratch tail, [ceil(rows / rpb)][W] dense), its level-2 record reads that.  max_w* /
 * max_rowblocks1: the largest W / row-block count of a table (grid extents).  Replaces two launches per slab by two in all. */
int mtn_colsum_rows_per_block(void);
int mtn_colsum_direct_rows(void);
int mtn_colsum_grouped(const void* jobs_l1, int n1, int max_w1, int max_rowblocks1, const void* jobs_l2, int n2,
                       int max_w2, void* stream);
/* dbias[256] = column sums of dy[M][256] */
int mtn_bias_grad(const void* dy, int lddy, float* partial, float* dbias, int dtype, int M, void* stream);

/* softmax(q k^T * scale) v per (batch, head), heads merged in the output (transformer.py:62-74).
 * qkv [B*N][3*H*D]; out [B*N][H*D]; lse f32 [B*H][N]; attn f32 [B][H][N][N] or NULL. D must be 32. */
int mtn_attention_fwd(const void* qkv, void* out, float* lse, float* attn, int dtype, int B, int N,
                      int H, int D, float scale, void* stream);
/* dqkv from dout (flash-style recompute from lse); delta f32 [B*H][N] scratch */
int mtn_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                      float* delta, void* dqkv, int dtype, int B, int N, int H, int D, float scale,
                      void* stream);

/* mlp_head = LayerNorm + Linear(256->C) on the cls token (transformer.py:113-116,142-144);
 * tok_stride = elements between consecutive samples' cls rows; xhat f32 [B][256], mean_rstd [B][2]. */
int mtn_cls_head_fwd(const void* tok, long long tok_stride, const float* ln_w, const float* ln_b,
                     const float* W, const float* bias, float* logits, float* xhat, float* mean_rstd,
                     int dtype, int B, int C, float eps, void* stream);
/* writes dtok cls rows, dW [C][256], dbias [C], dln_w_b [2][256]; partial f32 [mtn_slab_rows(B)][512] */
int mtn_cls_head_bwd(const float* dlogits, const float* W, const float* ln_w, const float* ln_b,
                     const float* xhat, const float* mean_rstd, void* dtok, long long tok_stride,
                     float* dW, float* dbias, float* dln_w_b, float* partial, int dtype, int B, int C,
                     void* stream);

/* pose head: bilinear x4 (align_corners) -> ReLU -> 1x1 conv 256->J + bias (transformer.py:146-150);
 * tok [B][h*w+1][256] (row 0 = cls, skipped); hm f32 NCHW [B][J][4h][4w]. */
int mtn_pose_head_fwd(const void* tok, const float* W, const float* bias, float* hm, int dtype, int B,
                      int h, int w, int J, void* stream);
/* writes dtok rows 1.. and dW_dbias = [dW [J][256] | dbias [J]] (one contiguous f32 block);
 * slab f32 [mtn_pose_slab_floats(B, h, w, J)]: weight-gradient partials per image and band of source rows, their
 * folding rows, and the f32 partials of the token-gradient rows two bands share (small batches of large images
 * are cut into up to 4 bands per image so that the launch fills the chip). */
size_t mtn_pose_slab_floats(int B, int h, int w, int J);
int mtn_pose_head_bwd(const void* tok, const float* W, const float* dhm, void* dtok, float* dW_dbias,
                      float* slab, int dtype, int B, int h, int w, int J, void* stream);

/* ---- loss + its gradients (libs/loss.py:4-40, train.py:63-64,75) --------------------------------
 * losses[3] = (total, ce_weight*CE, JointsMSE); dlogits [B][C], dhm like hm (either may be NULL);
 * partial f32 [mtn_loss_partial_len(B*J*HW)]. */
int mtn_loss_partial_len(long long n);
int mtn_loss_fwd_bwd(const float* logits, const long long* label, const float* hm, const float* target,
                     const float* target_weight, float* dlogits, float* dhm, float* partial,
                     float* losses, int B, int C, int J, int HW, float ce_weight, void* stream);

/* ---- optimizer (train.py:50-51: AdamW defaults) --------------------------------------------------
 * state f32[16]: lr, beta1, beta2, eps, weight_decay, step, bias_corr1, bias_corr2, grad_scale.
 * One call = step += 1, bias corrections, decoupled decay and the Adam update over n elements. */
int mtn_adamw_step(float* p, const float* g, float* m, float* v, long long n, float* state, void* stream);
/* repack fp32 master weights (OIHW) into kernel operand layouts; desc int32 [ntensors][7] =
 * {src_off, dst_off, O, I, taps, mode, scale_off}; mode 0: [O][tap][I] (forward), 1: [I][tap][O] (dgrad).
 * scale_off >= 0 (mode 0 only, scale != NULL): the weights of output channel o are multiplied by
 * scale[scale_off + o] -- inference with BatchNorm folded into the convolution (the reference runs its
 * model in eval() for export / evaluation, export.py:41-45): scale = gamma / sqrt(running_var + eps) from
 * mtn_bn_eval_scale_shift, the matching shift goes to mtn_conv_bias_act_fwd as its bias.  -1: no factor.
 * dtype MTN_DTYPE_F32X3: the packed weights are written in split storage like the activations -- every 32-k block of a
 * packed row as [32 bf16 hi][32 bf16 lo], k in natural order; mode + 2 keeps a tensor in plain fp32 (the stem: mtn_conv1_*
 * multiply in exact f32).  Packed row lengths (I*taps for mode 0, O*taps for mode 1) must then be multiples of 32. */
int mtn_pack_weights(const float* src, const float* scale, void* dst, const int* desc, int ntensors,
                     int max_elems, int dtype, void* stream);

/* split storage (MTN_DTYPE_F32X3) <-> plain f32: rows of C channels (C, ld_src, ld_dst multiples of 32; 16-byte aligned),
 * to_split != 0: src plain f32 -> dst split; 0: src split -> dst plain f32.  Used where a split tensor crosses the module
 * boundary (backbone features / their gradient) and around the pose head. */
int mtn_x3_convert(const void* src, int ld_src, void* dst, int ld_dst, int to_split, long long M, int C, void* stream);

/* ---- step metrics on the device (reference: host numpy / sklearn per step) ------
 * get_max_preds (libs/utils.py:4-32): preds f32 [maps][2] = (x, y) of the first maximum, zeroed
 * where the maximum is <= 0; maxvals f32 [maps] (optional).  maps = B*J. */
int mtn_heatmap_argmax(const float* hm, float* preds, float* maxvals, int maps, int H, int W, void* stream);
/* pose_accuracy (libs/metrics.py:31-62) from decoded coordinates: acc f32 [J+2] =
 * (mean over valid joints, per-joint accuracy or -1 ..., number of valid joints). */
int mtn_pose_accuracy(const float* pred_xy, const float* target_xy, float* acc, int B, int J, int H,
                      int W, float thr, void* stream);
/* argmax of the logits (train.py:66) and confusion counts cm[pred][label] += 1 (int32 [C][C],
 * accumulated; either output may be NULL) */
int mtn_cls_confusion(const float* logits, const long long* label, long long* pred_label, int* cm, int B,
                      int C, void* stream);
/* macro F1 over the labels present (sklearn f1_score(average='macro') as used at train.py:67-69) */
int mtn_macro_f1(const int* cm, float* f1, int C, void* stream);

/* ---- input side on the device ------------------------------------------------
 * Gaussian heat-map targets of libs/load.py:148-206: joints f32 [maps][2] in image pixels,
 * vis f32 [maps] -> target f32 [maps][HH][HW] (peak 1, (6*sigma+1)^2 patch), weight f32 [maps]. */
int mtn_render_targets(const float* joints, const float* vis, float* target, float* weight, int maps,
                       int HH, int HW, int img_h, int img_w, int sigma, void* stream);
/* uint8 HWC image batch (cv2 order) -> fp32 NCHW, /255 then ImageNet mean/std (libs/load.py:46-50) */
int mtn_normalize_image(const unsigned char* img_hwc_u8, float* out_nchw, int B, int H, int W, void* stream);

/* test / tuning hooks.  mtn_debug_set: key 1 = ds_read_b64_tr_b16 operand reads in wgrad (default 1);
 * key 3 = igemm pipeline (1: register-staged single LDS buffer; 2: register-staged double buffer;
 *         3: LDS-DMA double buffer; 4: LDS-DMA triple buffer with counted vmcnt; 5: window-resident
 *         activations for stride-1 3x3 layers, LDS-DMA double buffer elsewhere);
 * keys 4-14: tile / split / kernel-choice overrides of the tuning tools (tools/bench_kernels.py);
 * key 15 = dgrad2 (all four parities of a 3x3 stride-2 data gradient per workgroup; 0: igemm's
 *          parity-major rows); key 16 = igemm image tiles (bit 0 on, bit 1 ignore the tile-count floor);
 * key 18 = kernel-row tiles of the 64-channel 3x3 weight gradient (0: one tap per tile);
 * key 3 = 7 / 8 and key 6 = 4 / 5 / 6: measured-and-rejected igemm pipelines kept for their tests (ping-pong wave
 *          halves, weights through registers, 256x128 tiles: DESIGN.md section 7d); keys 19 / 20: split-bf16 tile /
 *          LDS-stage variants; key 21 = retired (accepted, no effect); key 22 = persistent pointwise kernel (0 off,
 *          2: also K = 512); key 24 = channel-chunk order of bn_act_fwd; key 25 = bands per image of the pose-head
 *          backward (0 = by batch size); keys 26 / 27 = measured-and-rejected tail variants (experiments builds: profiles/r04_tail_ab.txt);
 *          key 28 = 0: f32 partial tiles in the split-K workspace of the bf16 mode (default: bf16 tiles, summed in f32);
 *          key 29 = 0: the narrow step of the 3x3 window weight gradient for every map -- 128 slots bf16, 64 split storage
 *                      (default: 256 for bf16 maps up to 48 wide, 128 for split-storage maps up to 24 wide);
 *          key 30 = 0: mtn_linear_wgrad_group / mtn_conv_wgrad_group run their problems one launch each;
 *          key 8 = 7 / 8: the 256x64 window tile of the 64-channel 3x3 layers whatever the row count / never.
 * Every key is refused (MTN_ERR_UNSUPPORTED, "disabled") unless the process set MTN_ALLOW_DEBUG_KEYS=1 before its first
 * call into the library: the product path has no mutable configuration.
 * The environment variable MTN_DEBUG_KEYS="k=v,k=v" applies keys when the binding loads the library.
 * mtn_debug_read: key 1 = in-kernel phase stamps of the last igemm launch (diagnostic builds with
 * -DMTN_STAMPS only; returns non-zero otherwise). */
int mtn_debug_set(int key, int value);
int mtn_debug_read(int key, void* dst, size_t bytes);
/* (ABI 6) 1 iff this build contains the measured-and-rejected kernel variants (-DMTN_EXPERIMENTS: the CPU executor
 * library of the tests, diagnostic GPU builds); the product library returns 0.  mtn_debug_key_is_experiment: 1 iff
 * mtn_debug_set(key, value) selects such a variant (it is refused on a build without them) -- the test suite uses it to
 * keep the experiment-only parametrisations out of a product-library run instead of skipping them one by one. */
int mtn_has_experiments(void);
int mtn_debug_key_is_experiment(int key, int value);
/* (ABI 6) which kernel source serves a conv / Linear-shaped product under the launch rules of this build, for profilers
 * and bench.py's per-family roofline (no launch, no pointers): dgrad = 0 forward (a Linear is B = rows, H = W = 1,
 * ksize = stride = 1), 1 data gradient (geometry of the FORWARD conv); has_res / has_bn: the call has a second epilogue
 * operand / a BatchNorm-backward hand-over.  Returns MTN_KERNEL_*, or -1 on bad arguments. */
#define MTN_KERNEL_IGEMM 0        /* igemm_kernel, streamed A and W tiles */
#define MTN_KERNEL_IGEMM_WINDOW 1 /* igemm_kernel, stride-1 3x3 window mode */
#define MTN_KERNEL_IGEMM_IMAGE 2  /* igemm_kernel, one image per tile (stage-3 3x3 layers) */
#define MTN_KERNEL_PW 3           /* pw_kernel: persistent weight-stationary pointwise walk */
#define MTN_KERNEL_DGRAD2 4       /* dgrad2_kernel: 3x3 stride-2 data gradient */
#define MTN_KERNEL_WIN3 5         /* win3_kernel (experiments builds only) */
int mtn_conv_kernel_family(int dtype, int dgrad, int B, int H, int W, int Cin, int Cout, int ksize, int stride,
                           int has_res, int has_bn);

#ifdef __cplusplus
}
#endif
#endif /* MTN_H_ */
