// mtn_kernels.h -- internal launch interfaces between the C ABI (capi.hip) and the kernel files.
#pragma once
#include "mtn.h"
#include "mtn_rt.h"

#define MTN_F32 MTN_DTYPE_F32
#define MTN_BF16 MTN_DTYPE_BF16
#define MTN_F32X3 MTN_DTYPE_F32X3
// the kernels that have no split-storage form (the stem's unfused ABI entries, the pose head) take PLAIN f32 operands
// when they are called with MTN_DTYPE_F32X3: the caller converts at that boundary (mtn_x3_convert)
static inline int mtn_storage_dtype(int dtype) { return dtype == MTN_DTYPE_F32X3 ? MTN_DTYPE_F32 : dtype; }
#define MTN_ACT_GELU_GRAD_ 100     // igemm epilogue only: multiply by gelu'(saved pre-activation)

struct IgemmArgs {
    const void* A;        // NHWC activations, pixel stride lda (elements)
    const void* Wp;       // packed weights [N][K], K ordered (kh, kw, ci)
    void* Out;            // [M][ldc]
    int lda, ldc;
    int B, H, W, Cin;     // gather source geometry
    int OH, OW;           // rows: m = (b*OH + oy)*OW + ox
    int KH, KW;
    int mul, sgn, off, div;
    int M, N, K;
    const float* bias;    // [N] or null
    int act;              // MTN_ACT_NONE / MTN_ACT_GELU (before res) / MTN_ACT_SILU (AFTER res: folded-BatchNorm inference) /
                          // MTN_ACT_GELU_GRAD_ (internal: out *= gelu'(aux), aux_out is READ)
    void* aux_out;        // pre-activation copy [M][ld_aux] (GELU backward) or null
    int ld_aux;
    const void* res;      // residual [M][ldr] added after act, or null (may alias Out)
    int ldr;
    float* stats;         // [tiles_m][2][N] BN partial sums, or null
    int stats_rows;       // rows the caller allocated for stats: must equal the tile rows of the launch
    // Data gradients only: Out is the OUTPUT gradient of a BatchNorm unit (raw conv output bn_y, pre-activation
    // bn_y*scale+shift [+ bn_res]).  stats then receives that unit's backward sums -- sum dz, sum dz*xhat,
    // dz = Out * act'(.) -- instead of sum / sum of squares: the reduce pass of its BN backward (bn.hip).
    const void* bn_y; int bn_ldy;
    const void* bn_res; int bn_ldr;
    const float* bn_ss;   // [2][N] scale, shift
    const float* bn_mr;   // [2][N] mean, rstd
    int bn_act;
    // Token assembly in the epilogue (pointwise launches only): Out is the ViT token tensor [B][tok_P + 1][N]; row
    // m = (b, p) is written to token row b*(tok_P+1) + 1 + p with tok_pos[p][n] added, and the rows p = 0 also write the class
    // token tok_cls[n] to row b*(tok_P+1) (model/transformer.py:132-139).  tok_magic: multiply-high divisor of m / tok_P.
    const float* tok_pos; const float* tok_cls; int tok_P; unsigned tok_magic;
    int pm;               // rows in parity-major order (stride-2 dgrad tap skipping)
    int win_cs;           // filled at launch (window mode): resident 128-byte channel slabs of the activation window
    unsigned magic_w, magic_h;   // filled by mtn_igemm_run: multiply-high divisors of the row decode
    unsigned magic_tn;           // filled at launch: multiply-high divisor of the tile index (N / BN)
    unsigned a_bytes, w_bytes;   // filled by mtn_igemm_run: byte extents of A / Wp (buffer descriptors)
    int dbg;                     // diagnostic builds (debug key 8): 4 = stamp the last 256 workgroups instead of the first
};

int mtn_igemm_run(const IgemmArgs& p, int dtype, hipStream_t stream);
int mtn_igemm_tiles_m(int M, int N, int K);
bool mtn_dgrad2_ok(const IgemmArgs& p, int dtype);      // dgrad2.hip: all four parities of a 3x3 stride-2 data gradient per workgroup
int mtn_dgrad2_run(const IgemmArgs& p, int dtype, hipStream_t stream);
int mtn_dgrad2_tiles_m(const IgemmArgs& p);
int mtn_igemm_stats_tiles(const IgemmArgs& p, int dtype);   // rows of the statistics slab this launch would write
void mtn_dgrad2_set(int on);
int mtn_igemm_family(const IgemmArgs& p, int dtype);     // MTN_KERNEL_* (include/mtn.h): the kernel mtn_igemm_run would launch
bool mtn_pw_ok(const IgemmArgs& p, int dtype);          // pw.hip: persistent weight-stationary kernel for short-K pointwise products
int mtn_pw_run(const IgemmArgs& p, int dtype, hipStream_t stream);
int mtn_pw_rows(const IgemmArgs& p);                    // rows of the statistics slab it writes (one per workgroup of an n-tile)
void mtn_pw_set(int on);
bool mtn_win3_ok(const IgemmArgs& p, int dtype);        // win3.hip: persistent weight-stationary kernel for the 64 -> 64 3x3 stride-1 layers
int mtn_win3_run(const IgemmArgs& p, int dtype, hipStream_t stream);
int mtn_win3_rows(const IgemmArgs& p);
void mtn_win3_set(int on);

struct WgradArgs {
    const void* dY; int lddy;     // [M][lddy] output-pixel gradients
    const void* X; int ldx;       // NHWC input of the forward conv
    float* partial;               // [splits][N][K] f32 workspace
    float* dW;                    // [N][Cin][KH][KW] f32 (reference OIHW layout)
    int B, H, W, Cin, OH, OW, KH, KW, stride, pad;
    int M, N, K;
    int splits, rows_per_split;   // win: split of the PADDED SLOT axis, rows_per_split = slots per split
    int win;                      // 1: 3x3 stride-1 window kernel (wgrad3_kernel); 2: kernel-row tiles (wgrad_row3_kernel)
    unsigned y_bytes, x_bytes;    // filled by mtn_wgrad_run: byte extents of dY / X (buffer descriptors)
    unsigned magic_ow, magic_oh;  // filled by mtn_wgrad_run: multiply-high divisors of the pixel walk
    int slab_bf16;                // filled by mtn_wgrad_run: the partial tiles are stored as bf16 (bf16 mode only)
};
// several pointwise problems (same row count) in one launch: wgrad.hip, wgrad_dma2_group_kernel
constexpr int MTN_WGRAD_GROUP_MAX = 16;
struct WgradGroupArgs {
    int count;
    int first[MTN_WGRAD_GROUP_MAX + 1];       // workgroup ranges of the problems (multiples of 8)
    int rfirst[MTN_WGRAD_GROUP_MAX + 1];      // block ranges of the grouped split sum
    WgradArgs item[MTN_WGRAD_GROUP_MAX];
};
int mtn_wgrad_group_plan(int dtype, int M, const int* N, const int* K, int count, int* rows_per_split);
size_t mtn_wgrad_group_bytes(int dtype, int M, const int* N, const int* K, int count);
int mtn_wgrad_group_run(WgradGroupArgs& g, int dtype, int M, void* workspace, size_t ws_bytes, hipStream_t stream);
void mtn_wgrad_set_group(int on);
int mtn_wgrad3_group_splits(int B, int H, int W, int N, int Cin, int dtype, int count, int* slots_per_split);
size_t mtn_wgrad3_group_bytes(int B, int H, int W, int N, int Cin, int dtype, int count);
int mtn_wgrad3_group_run(WgradGroupArgs& g, int dtype, int B, int H, int W, int Cin, int N, void* workspace, size_t ws_bytes, hipStream_t stream);
int mtn_wgrad_splits(int M, int N, int K, int Cin, int dtype, int* rows_per_split);
int mtn_wgrad_run(const WgradArgs& p, int dtype, hipStream_t stream);
void mtn_wgrad_set_tr16(int on);
void mtn_wgrad_set_ks256(int on);
void mtn_conv1_set_x3(int v);
void mtn_wgrad_set_x3_sep(int on);

struct BnBwdArgs {
    const void* dout; int lddo;       // gradient of the block output act(bn(y) [+res])
    const void* y; int ldy;           // raw conv output saved by the forward
    const void* res; int ldr;         // residual input of the forward (or null)
    const float* scale_shift;         // [2][C] from bn_finalize
    const float* mean_rstd;           // [2][C]
    float* partial;                   // [mtn_reduce_rows_needed(mtn_bn_bwd_blocks(M))][2][C]
    int partial_rows;                 // 0: run the reduce pass; > 0: partial already holds that many rows of
                                      // (sum dz, sum dz*xhat) -- written by the data gradient that produced dout
    float* coef;                      // [2][C] scratch
    float* dgamma; float* dbeta;      // [C] parameter gradients (overwritten)
    void* dy; int lddy;               // gradient of the raw conv output
    void* dres; int lddres;           // gradient of the residual input (+ extra), or null
    const void* extra; int ldextra;   // optional tensor added into dres (may alias dres)
    int M, C, act;
};
int mtn_reduce_rows_needed(int rows);
int mtn_bn_finalize_run(float* stats, int tiles, int C, long count, const float* gamma, const float* beta,
                        float* rmean, float* rvar, long long* nbt, float momentum, float eps, float* scale_shift,
                        float* mean_rstd, hipStream_t stream);
int mtn_bn_act_fwd_run(const void* y, int ldy, const float* ss, const void* res, int ldr, void* out, int ldo, int dtype,
                       int M, int C, int act, hipStream_t stream);
int mtn_bn_bwd_blocks(int M);
void mtn_bn_set_fwd_order(int descending);
void mtn_bn_set_ticket(int on);
void mtn_wgrad_set_skip_reduce(int on);
void mtn_wgrad_set_slab_bf16(int on);
int mtn_bn_bwd_run(const BnBwdArgs& a, int dtype, hipStream_t stream);
int mtn_colsum_run(float* slab, int rows, int W, float* out, hipStream_t stream);
int mtn_colsum_ld_run(float* slab, int rows, int W, int ld, float* out, hipStream_t stream);
int mtn_colsum_grouped_run(const void* jobs_l1, int n1, int max_w1, int max_rowblocks1, const void* jobs_l2, int n2, int max_w2,
                           hipStream_t stream);
int mtn_fold_rows_per_block(void);
int mtn_fold_direct_rows(void);

int mtn_conv1_tiles(long M);
int mtn_conv1_fwd_run(const float* x, const void* w, void* y, float* stats, int dtype, int B, int H, int W,
                      hipStream_t stream);
int mtn_conv1_wgrad_blocks(long M);
int mtn_conv1_stats_run(const float* x, const void* w, float* stats, int dtype, int B, int H, int W, hipStream_t stream);
int mtn_conv1_bn_act_run(const float* x, const void* w, const float* scale_shift, void* out, int act, int dtype, int B, int H,
                         int W, hipStream_t stream);
int mtn_conv1_bwd_blocks(long M);
int mtn_conv1_bwd_width(void);
int mtn_conv1_bwd_run(const float* x, const void* w, const void* dout, const float* scale_shift, const float* mean_rstd,
                      float* partial, float* dgamma, float* dbeta, float* dw, int act, int dtype, int B, int H, int W,
                      hipStream_t stream);
int mtn_conv1_dx_run(const float* x, const float* w, const void* dact, const float* scale_shift, const float* mean_rstd,
                     const float* dgamma, const float* dbeta, float* dx, int act, int dtype, int B, int H, int W, hipStream_t stream);
int mtn_conv1_wgrad_run(const float* x, const void* dy, float* dw, float* partial, int dtype, int B, int H, int W,
                        hipStream_t stream);

struct AttnArgs {
    const void* qkv;      // [B*N][3*H*D]  (q | k | v) x (h d)
    void* out;            // [B*N][H*D] attention output (forward: written; backward: read)
    const void* dout;     // [B*N][H*D] gradient of out (backward)
    float* lse;           // [B*H][N] log-sum-exp of the scaled scores
    float* delta;         // [B*H][N] rowsum(dout*out) scratch (backward)
    float* attn;          // [B][H][N][N] probabilities or null (forward)
    void* dqkv;           // [B*N][3*H*D] (backward)
    int B, N, H, D;
    float scale;
};
int mtn_attn_fwd_run(const AttnArgs& a, int dtype, hipStream_t stream);
int mtn_attn_bwd_run(const AttnArgs& a, int dtype, hipStream_t stream);

int mtn_x3_convert_run(const void* src, int ld_src, void* dst, int ld_dst, int to_split, long M, int C, hipStream_t stream);
int mtn_tokens_fwd_run(const void* feat, const float* pos, const float* cls, void* tok, int dtype, int B, int P, hipStream_t stream);
int mtn_tokens_bwd_run(const void* dtok, void* dfeat, float* dcls, int dtype, int B, int P, hipStream_t stream);
int mtn_ln_fwd_run(const void* x, int ldx, const float* gamma, const float* beta, void* y, int ldy, float* mean_rstd, int dtype,
                   int M, float eps, hipStream_t stream);
int mtn_rowslab_blocks(int M);
int mtn_ln_bwd_run(const void* dy, int lddy, const void* x, int ldx, const float* mean_rstd, const float* gamma, const void* dres,
                   int lddres, void* dx, int lddx, float* partial, float* dgamma_dbeta, int dtype, int M, hipStream_t stream);
int mtn_gelu_bwd_run(const void* dh, const void* a, void* da, int dtype, long n, hipStream_t stream);
int mtn_bias_grad_run(const void* dy, int lddy, float* partial, float* dbias, int dtype, int M, hipStream_t stream);
int mtn_cls_head_fwd_run(const void* tok, long tok_stride, const float* lg, const float* lb, const float* W, const float* bias,
                         float* logits, float* xhat, float* mean_rstd, int dtype, int B, int C, float eps, hipStream_t stream);
int mtn_cls_head_bwd_run(const float* dlogits, const float* W, const float* lg, const float* lb, const float* xhat,
                         const float* mean_rstd, void* dtok, long tok_stride, float* dW, float* dbias, float* dlg_dlb,
                         float* partial, int dtype, int B, int C, hipStream_t stream);
int mtn_pose_fwd_run(const void* tok, const float* W, const float* bias, float* hm, int dtype, int B, int h, int w, int J,
                     hipStream_t stream);
int mtn_pose_bwd_run(const void* tok, const float* W, const float* dhm, void* dtok, float* dW_dbias, float* slab,
                     int dtype, int B, int h, int w, int J, hipStream_t stream);
int mtn_loss_blocks(long n);
int mtn_loss_run(const float* logits, const long long* label, const float* hm, const float* tgt, const float* tw, float* dlogits,
                 float* dhm, float* partial, float* losses, int B, int C, int J, int HW, float ce_weight, hipStream_t stream);
int mtn_adamw_run(float* p, const float* g, float* m, float* v, long n, float* state, hipStream_t stream);
int mtn_pack_run(const float* src, const float* scale, void* dst, const int* desc, int ntensors, int max_elems, int dtype, hipStream_t stream);
int mtn_bn_eval_run(const float* gamma, const float* beta, const float* rmean, const float* rvar, float eps,
                    float* scale_shift, int C, hipStream_t stream);
void mtn_pose_set_dbg(int v);
void mtn_pose_set_row(int on);
void mtn_pose_set_bands(int n);
size_t mtn_pose_slab_floats_run(int B, int h, int w, int J);
void mtn_igemm_set_stages(int n);
int mtn_igemm_has_experiments(void);      // 1 in -DMTN_EXPERIMENTS builds (CPU executor, diagnostics): the rejected variants exist

int mtn_hm_argmax_run(const float* hm, float* preds, float* maxvals, int maps, int H, int W, hipStream_t stream);
int mtn_pose_acc_run(const float* pred, const float* tgt, float* acc, int B, int J, int H, int W, float thr, hipStream_t stream);
int mtn_cls_confusion_run(const float* logits, const long long* label, long long* pred_label, int* cm, int B, int C,
                          hipStream_t stream);
int mtn_macro_f1_run(const int* cm, float* f1, int C, hipStream_t stream);
int mtn_render_targets_run(const float* joints, const float* vis, float* target, float* weight, int maps, int HH, int HW,
                           int img_h, int img_w, int sigma, hipStream_t stream);
int mtn_normalize_image_run(const unsigned char* img, float* out, int B, int H, int W, hipStream_t stream);
void mtn_igemm_set_tile_n64(int rows);
void mtn_wgrad_set_dma(int on);
void mtn_igemm_set_tile(int t);
void mtn_igemm_set_dbg(int v);
void mtn_igemm_set_x3_waves(int v);
void mtn_igemm_set_window_limit(int kb);
void mtn_igemm_set_image_tiles(int on);
int mtn_igemm_read_stamps(void* dst, size_t bytes);
void mtn_wgrad_set_tile(int t);
void mtn_wgrad_set_slots(int n);
void mtn_wgrad_set_win(int on);
void mtn_wgrad_set_row3(int on);
int mtn_wgrad_row3_ok(int dtype, int N, int Cin, int ksize, int stride);
int mtn_wgrad_row3_splits(int M, int N, int Cin, int dtype, int* rows_per_split);
int mtn_wgrad3_ok(int dtype, int H, int W, int ksize, int stride);
int mtn_wgrad3_splits(int B, int H, int W, int N, int Cin, int dtype, int* slots_per_split);
void mtn_conv1_set_grid_cap(int n);
void mtn_conv1_set_bwd_px(int n);
