// capi.hip -- the extern "C" boundary (include/mtn.h): argument checks, error text, dispatch.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include "mtn.h"
#include "mtn_kernels.h"

static thread_local char g_err[512] = "";

void mtn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int mtn_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        mtn_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return MTN_ERR_LAUNCH;
    }
    return MTN_OK;
}

extern "C" {

const char* mtn_last_error(void) { return g_err; }
int mtn_abi_version(void) { return 7; }

int mtn_conv_stats_tiles(int M, int Cout, int K) { return mtn_igemm_tiles_m(M, Cout, K); }

static int conv_fwd_args(IgemmArgs& p, const char* who, int B, int H, int W, int Cin, int Cout, int ksize, int stride) {
    MTN_REQUIRE(ksize == 1 || ksize == 3, "%s: ksize=%d", who, ksize);
    MTN_REQUIRE(stride == 1 || stride == 2, "%s: stride=%d", who, stride);
    const int pad = ksize / 2;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin;
    p.OH = (H + 2 * pad - ksize) / stride + 1;
    p.OW = (W + 2 * pad - ksize) / stride + 1;
    p.KH = p.KW = ksize;
    p.mul = stride; p.sgn = 1; p.off = -pad; p.div = 1;
    p.M = B * p.OH * p.OW; p.N = Cout; p.K = ksize * ksize * Cin;
    return MTN_OK;
}

int mtn_conv_fwd_stats_tiles(int dtype, int B, int H, int W, int Cin, int Cout, int ksize, int stride) {
    IgemmArgs p{};
    if (conv_fwd_args(p, "conv_fwd_stats_tiles", B, H, W, Cin, Cout, ksize, stride)) return -1;
    return mtn_igemm_stats_tiles(p, dtype);
}

int mtn_conv_fwd(const void* x, int ldx, const void* w, void* y, int ldy, float* stats, int stats_tiles, int dtype,
                 int B, int H, int W, int Cin, int Cout, int ksize, int stride, void* stream) {
    IgemmArgs p{};
    const int rc = conv_fwd_args(p, "conv_fwd", B, H, W, Cin, Cout, ksize, stride);
    if (rc) return rc;
    p.A = x; p.Wp = w; p.Out = y; p.lda = ldx; p.ldc = ldy;
    p.stats = stats;
    p.stats_rows = stats_tiles;
    return mtn_igemm_run(p, dtype, (hipStream_t)stream);
}

int mtn_proj_tokens_fwd(const void* x, int ldx, const void* w, const float* pos, const float* cls, void* tok, int dtype, int B, int H,
                        int W, int Cin, int Cout, void* stream) {
    IgemmArgs p{};
    const int rc = conv_fwd_args(p, "proj_tokens_fwd", B, H, W, Cin, Cout, 1, 1);
    if (rc) return rc;
    MTN_REQUIRE(pos && cls && tok, "proj_tokens_fwd: null operand");
    p.A = x; p.Wp = w; p.Out = tok; p.lda = ldx; p.ldc = Cout;
    p.tok_pos = pos; p.tok_cls = cls; p.tok_P = H * W;
    return mtn_igemm_run(p, dtype, (hipStream_t)stream);
}

int mtn_conv_bias_act_fwd(const void* x, int ldx, const void* w, const float* bias, int act, const void* res, int ldres,
                          void* y, int ldy, int dtype, int B, int H, int W, int Cin, int Cout, int ksize, int stride,
                          void* stream) {
    MTN_REQUIRE(act == MTN_ACT_NONE || act == MTN_ACT_SILU, "conv_bias_act_fwd: act=%d", act);
    IgemmArgs p{};
    const int rc = conv_fwd_args(p, "conv_bias_act_fwd", B, H, W, Cin, Cout, ksize, stride);
    if (rc) return rc;
    p.A = x; p.Wp = w; p.Out = y; p.lda = ldx; p.ldc = ldy;
    p.bias = bias; p.act = act; p.res = res; p.ldr = ldres;
    return mtn_igemm_run(p, dtype, (hipStream_t)stream);
}

static int conv_dgrad_args(IgemmArgs& p, const char* who, int B, int H, int W, int Cin, int Cout, int ksize, int stride) {
    MTN_REQUIRE(ksize == 1 || ksize == 3, "%s: ksize=%d", who, ksize);
    MTN_REQUIRE(stride == 1 || stride == 2, "%s: stride=%d", who, stride);
    const int pad = ksize / 2;
    p.B = B;
    p.H = (H + 2 * pad - ksize) / stride + 1;     // gather source = dy
    p.W = (W + 2 * pad - ksize) / stride + 1;
    p.Cin = Cout;
    p.OH = H; p.OW = W;                             // rows = input pixels
    p.KH = p.KW = ksize;
    p.mul = 1; p.sgn = -1; p.off = pad; p.div = stride;
    p.M = B * H * W; p.N = Cin; p.K = ksize * ksize * Cout;
    p.pm = (stride == 2 && ksize == 3 && H % 2 == 0 && W % 2 == 0) ? 1 : 0;
    return MTN_OK;
}

int mtn_conv_dgrad(const void* dy, int lddy, const void* wt, void* dx, int lddx, const void* res,
                   int ldres, int dtype, int B, int H, int W, int Cin, int Cout, int ksize,
                   int stride, void* stream) {
    IgemmArgs p{};
    const int rc = conv_dgrad_args(p, "conv_dgrad", B, H, W, Cin, Cout, ksize, stride);
    if (rc) return rc;
    p.A = dy; p.Wp = wt; p.Out = dx; p.lda = lddy; p.ldc = lddx;
    p.res = res; p.ldr = ldres;
    return mtn_igemm_run(p, dtype, (hipStream_t)stream);
}

int mtn_conv_dgrad_bn_tiles(int dtype, int B, int H, int W, int Cin, int Cout, int ksize, int stride) {
    IgemmArgs p{};
    if (conv_dgrad_args(p, "conv_dgrad_bn_tiles", B, H, W, Cin, Cout, ksize, stride)) return -1;
    p.bn_y = &p;                 // (the launch this sizes has a BatchNorm hand-over: the kernel choice depends on it)
    return mtn_igemm_stats_tiles(p, dtype);
}

// BatchNorm-backward hand-over: mtn_conv_dgrad_bn leaves `tiles` rows of (sum dz, sum dz*xhat) in bn_partial and the
// mtn_bn_act_bwd of the unit they belong to consumes them instead of reducing (dout, y) itself.  The consumer cannot see
// what the producer wrote, so the producer describes it in a record the CALLER owns (mtn_bn_handover, include/mtn.h) and the
// consumer requires its partial_rows / slab / channel count to match that record, then clears it: a hand-over is used once.
// (Until ABI 5 the record was a thread-local of the library holding a device pointer between two calls; SURVEY 8b: the
//  library retains nothing past a call.)
int mtn_conv_dgrad_bn(const void* dy, int lddy, const void* wt, void* dx, int lddx, const void* res, int ldres,
                      const void* bn_y, int bn_ldy, const void* bn_res, int bn_ldres, const float* bn_scale_shift,
                      const float* bn_mean_rstd, int bn_act, float* bn_partial, int bn_tiles, mtn_bn_handover* record, int dtype,
                      int B, int H, int W, int Cin, int Cout, int ksize, int stride, void* stream) {
    IgemmArgs p{};
    const int rc = conv_dgrad_args(p, "conv_dgrad_bn", B, H, W, Cin, Cout, ksize, stride);
    if (rc) return rc;
    MTN_REQUIRE(bn_y && bn_scale_shift && bn_mean_rstd && bn_partial, "conv_dgrad_bn: null BatchNorm operand");
    p.A = dy; p.Wp = wt; p.Out = dx; p.lda = lddy; p.ldc = lddx;
    p.res = res; p.ldr = ldres;
    p.bn_y = bn_y; p.bn_ldy = bn_ldy; p.bn_res = bn_res; p.bn_ldr = bn_ldres;
    p.bn_ss = bn_scale_shift; p.bn_mr = bn_mean_rstd; p.bn_act = bn_act;
    p.stats = bn_partial; p.stats_rows = bn_tiles;
    MTN_REQUIRE(record, "conv_dgrad_bn: null hand-over record");
    *record = mtn_bn_handover{nullptr, 0, 0, 0};
    const int rc2 = mtn_igemm_run(p, dtype, (hipStream_t)stream);
    if (rc2 == MTN_OK) *record = mtn_bn_handover{bn_partial, bn_tiles, Cin, dtype};
    return rc2;
}

int mtn_linear_fwd(const void* x, int ldx, const void* w, const float* bias, int act, void* aux_out,
                   int ld_aux, const void* res, int ldres, void* out, int ldo, int dtype, int M,
                   int N, int K, void* stream) {
    MTN_REQUIRE(act == MTN_ACT_NONE || act == MTN_ACT_GELU, "linear_fwd: act=%d", act);
    IgemmArgs p{};
    p.A = x; p.Wp = w; p.Out = out; p.lda = ldx; p.ldc = ldo;
    p.B = M; p.H = 1; p.W = 1; p.Cin = K; p.OH = 1; p.OW = 1; p.KH = p.KW = 1;
    p.mul = 1; p.sgn = 1; p.off = 0; p.div = 1;
    p.M = M; p.N = N; p.K = K;
    p.bias = bias; p.act = act; p.aux_out = aux_out; p.ld_aux = ld_aux; p.res = res; p.ldr = ldres;
    return mtn_igemm_run(p, dtype, (hipStream_t)stream);
}

int mtn_linear_gelu_bwd_tiles(int M, int N, int K) { return mtn_igemm_tiles_m(M, N, K); }

int mtn_linear_gelu_bwd(const void* dh, int lddh, const void* wt, const void* a, int lda, void* da, int ldda,
                        float* partial, int tiles, float* dbias, int dtype, int M, int N, int K, void* stream) {
    MTN_REQUIRE(a && partial, "linear_gelu_bwd: null operand");
    IgemmArgs p{};
    p.A = dh; p.Wp = wt; p.Out = da; p.lda = lddh; p.ldc = ldda;
    p.B = M; p.H = 1; p.W = 1; p.Cin = K; p.OH = 1; p.OW = 1; p.KH = p.KW = 1;
    p.mul = 1; p.sgn = 1; p.off = 0; p.div = 1;
    p.M = M; p.N = N; p.K = K;
    p.act = MTN_ACT_GELU_GRAD_; p.aux_out = const_cast<void*>(a); p.ld_aux = lda;
    p.stats = partial; p.stats_rows = tiles;
    const int rc = mtn_igemm_run(p, dtype, (hipStream_t)stream);
    if (rc) return rc;
    if (!dbias) return MTN_OK;                            // the caller folds the slab later (mtn_colsum_grouped)
    return mtn_colsum_ld_run(partial, tiles, N, 2 * N, dbias, (hipStream_t)stream);    // the sum half of each [2][N] row
}

static int conv_wgrad_args(WgradArgs& p, int dtype, int B, int H, int W, int Cin, int Cout, int ksize, int stride) {
    const int pad = ksize / 2;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin;
    p.OH = (H + 2 * pad - ksize) / stride + 1;
    p.OW = (W + 2 * pad - ksize) / stride + 1;
    p.KH = p.KW = ksize; p.stride = stride; p.pad = pad;
    p.M = B * p.OH * p.OW; p.N = Cout; p.K = ksize * ksize * Cin;
    if (Cout % 64 != 0 || Cin % 64 != 0) { mtn_set_error("wgrad: Cout=%d, Cin=%d must be multiples of 64", Cout, Cin); return 1; }
    p.win = mtn_wgrad3_ok(dtype, H, W, ksize, stride);
    if (p.win) p.splits = mtn_wgrad3_splits(B, H, W, p.N, Cin, dtype, &p.rows_per_split);
    else if (mtn_wgrad_row3_ok(dtype, p.N, Cin, ksize, stride)) {
        p.win = 2;
        p.splits = mtn_wgrad_row3_splits(p.M, p.N, Cin, dtype, &p.rows_per_split);
    } else p.splits = mtn_wgrad_splits(p.M, p.N, p.K, Cin, dtype, &p.rows_per_split);
    return 0;
}

size_t mtn_conv_wgrad_workspace_bytes(int dtype, int B, int H, int W, int Cin, int Cout, int ksize, int stride) {
    WgradArgs p{};
    if (conv_wgrad_args(p, dtype, B, H, W, Cin, Cout, ksize, stride)) return 0;
    return (size_t)p.splits * p.N * p.K * sizeof(float);
}

int mtn_conv_wgrad(const void* x, int ldx, const void* dy, int lddy, float* dw, void* workspace,
                   size_t ws_bytes, int dtype, int B, int H, int W, int Cin, int Cout, int ksize,
                   int stride, void* stream) {
    MTN_REQUIRE(ksize == 1 || ksize == 3, "conv_wgrad: ksize=%d", ksize);
    MTN_REQUIRE(stride == 1 || stride == 2, "conv_wgrad: stride=%d", stride);
    WgradArgs p{};
    if (conv_wgrad_args(p, dtype, B, H, W, Cin, Cout, ksize, stride)) return MTN_ERR_BAD_ARG;
    MTN_REQUIRE(ws_bytes >= (size_t)p.splits * p.N * p.K * sizeof(float), "conv_wgrad: workspace too small");
    p.dY = dy; p.lddy = lddy; p.X = x; p.ldx = ldx; p.partial = (float*)workspace; p.dW = dw;
    return mtn_wgrad_run(p, dtype, (hipStream_t)stream);
}

size_t mtn_linear_wgrad_workspace_bytes(int dtype, int M, int N, int K) {
    return mtn_conv_wgrad_workspace_bytes(dtype, 1, 1, M, K, N, 1, 1);
}

int mtn_linear_wgrad(const void* x, int ldx, const void* dy, int lddy, float* dw, void* workspace,
                     size_t ws_bytes, int dtype, int M, int N, int K, void* stream) {
    // the M rows as ONE image of width M: the kernels' pixel walk then never crosses an image
    // boundary (as M images of 1x1 it would, on every row of every step)
    return mtn_conv_wgrad(x, ldx, dy, lddy, dw, workspace, ws_bytes, dtype, 1, 1, M, K, N, 1, 1, stream);
}

// The Linear weight gradients of a whole backward pass in one launch + one split sum (wgrad.hip): `count` problems
// dw_i[N_i][K_i] = dy_i[M][N_i]^T x_i[M][K_i] over the same M rows.  Groups the kernel cannot take in one launch (a side that
// is no multiple of 128, more than MTN_WGRAD_GROUP_MAX problems) run as `count` mtn_linear_wgrad calls on the same workspace.
size_t mtn_linear_wgrad_group_workspace_bytes(int dtype, int M, const mtn_wgrad_item* items, int count) {
    if (!items || count <= 0 || M <= 0) return 0;
    size_t need = 0;
    for (int i = 0; i < count; ++i) {
        const size_t b = mtn_linear_wgrad_workspace_bytes(dtype, M, items[i].N, items[i].K);
        if (!b) return 0;
        need = b > need ? b : need;
    }
    for (int i0 = 0; i0 < count; i0 += MTN_WGRAD_GROUP_MAX) {
        const int n = count - i0 < MTN_WGRAD_GROUP_MAX ? count - i0 : MTN_WGRAD_GROUP_MAX;
        int Ns[MTN_WGRAD_GROUP_MAX], Ks[MTN_WGRAD_GROUP_MAX];
        for (int i = 0; i < n; ++i) { Ns[i] = items[i0 + i].N; Ks[i] = items[i0 + i].K; }
        const size_t b = mtn_wgrad_group_bytes(dtype, M, Ns, Ks, n);
        need = b > need ? b : need;
    }
    return need;
}

int mtn_linear_wgrad_group(const mtn_wgrad_item* items, int count, int M, void* workspace, size_t ws_bytes, int dtype, void* stream) {
    MTN_REQUIRE(items && count > 0 && M > 0 && workspace, "linear_wgrad_group: bad arguments (count=%d, M=%d)", count, M);
    for (int i0 = 0; i0 < count; i0 += MTN_WGRAD_GROUP_MAX) {
        const int n = count - i0 < MTN_WGRAD_GROUP_MAX ? count - i0 : MTN_WGRAD_GROUP_MAX;
        int Ns[MTN_WGRAD_GROUP_MAX], Ks[MTN_WGRAD_GROUP_MAX], rps = 0;
        for (int i = 0; i < n; ++i) { Ns[i] = items[i0 + i].N; Ks[i] = items[i0 + i].K; }
        if (mtn_wgrad_group_plan(dtype, M, Ns, Ks, n, &rps) > 0) {
            WgradGroupArgs g{};
            g.count = n;
            for (int i = 0; i < n; ++i) {
                const mtn_wgrad_item& it = items[i0 + i];
                g.item[i].dY = it.dy; g.item[i].lddy = it.lddy; g.item[i].X = it.x; g.item[i].ldx = it.ldx;
                g.item[i].dW = it.dw; g.item[i].N = it.N; g.item[i].K = it.K;
            }
            const int rc = mtn_wgrad_group_run(g, dtype, M, workspace, ws_bytes, (hipStream_t)stream);
            if (rc) return rc;
            continue;
        }
        for (int i = 0; i < n; ++i) {
            const mtn_wgrad_item& it = items[i0 + i];
            const int rc = mtn_linear_wgrad(it.x, it.ldx, it.dy, it.lddy, it.dw, workspace, ws_bytes, dtype, M, it.N, it.K, stream);
            if (rc) return rc;
        }
    }
    return MTN_OK;
}

// The weight gradients of `count` Conv2d(k=3, s=1, p=1) layers of ONE geometry -- the four 3x3 convolutions of a GELAN stage's
// residual blocks (model/gelan.py:69-70,78-87) -- in one launch + one split sum (wgrad.hip, wgrad3_group_kernel).  Geometries
// the window kernel does not serve (or more than MTN_WGRAD_GROUP_MAX problems) run as `count` mtn_conv_wgrad calls.  items[i].N / K
// must be Cout / Cin.
size_t mtn_conv_wgrad_group_workspace_bytes(int dtype, int count, int B, int H, int W, int Cin, int Cout, int ksize, int stride) {
    const size_t one = mtn_conv_wgrad_workspace_bytes(dtype, B, H, W, Cin, Cout, ksize, stride);
    if (!one || count <= 0) return 0;
    const size_t grp = (ksize == 3 && stride == 1) ? mtn_wgrad3_group_bytes(B, H, W, Cout, Cin, dtype, count < MTN_WGRAD_GROUP_MAX ? count : MTN_WGRAD_GROUP_MAX) : 0;
    return grp > one ? grp : one;
}

int mtn_conv_wgrad_group(const mtn_wgrad_item* items, int count, void* workspace, size_t ws_bytes, int dtype, int B, int H, int W,
                         int Cin, int Cout, int ksize, int stride, void* stream) {
    MTN_REQUIRE(items && count > 0 && workspace, "conv_wgrad_group: bad arguments (count=%d)", count);
    MTN_REQUIRE(ksize == 1 || ksize == 3, "conv_wgrad_group: ksize=%d", ksize);
    MTN_REQUIRE(stride == 1 || stride == 2, "conv_wgrad_group: stride=%d", stride);
    for (int i = 0; i < count; ++i)
        MTN_REQUIRE(items[i].N == Cout && items[i].K == Cin, "conv_wgrad_group: problem %d is %d x %d, the geometry says %d x %d", i, items[i].N, items[i].K, Cout, Cin);
    for (int i0 = 0; i0 < count; i0 += MTN_WGRAD_GROUP_MAX) {
        const int n = count - i0 < MTN_WGRAD_GROUP_MAX ? count - i0 : MTN_WGRAD_GROUP_MAX;
        int sps = 0;
        if (ksize == 3 && stride == 1 && mtn_wgrad3_group_splits(B, H, W, Cout, Cin, dtype, n, &sps) > 0) {
            WgradGroupArgs g{};
            g.count = n;
            for (int i = 0; i < n; ++i) {
                const mtn_wgrad_item& it = items[i0 + i];
                g.item[i].dY = it.dy; g.item[i].lddy = it.lddy; g.item[i].X = it.x; g.item[i].ldx = it.ldx; g.item[i].dW = it.dw;
            }
            const int rc = mtn_wgrad3_group_run(g, dtype, B, H, W, Cin, Cout, workspace, ws_bytes, (hipStream_t)stream);
            if (rc) return rc;
            continue;
        }
        for (int i = 0; i < n; ++i) {
            const mtn_wgrad_item& it = items[i0 + i];
            const int rc = mtn_conv_wgrad(it.x, it.ldx, it.dy, it.lddy, it.dw, workspace, ws_bytes, dtype, B, H, W, Cin, Cout, ksize, stride, stream);
            if (rc) return rc;
        }
    }
    return MTN_OK;
}

int mtn_conv1_stats_tiles(int B, int H, int W) { return mtn_conv1_tiles((long)B * (H / 2) * (W / 2)); }

int mtn_conv1_fwd(const float* x_nchw, const void* w, void* y, float* stats, int dtype, int B, int H, int W,
                  void* stream) {
    return mtn_conv1_fwd_run(x_nchw, w, y, stats, mtn_storage_dtype(dtype), B, H, W, (hipStream_t)stream);
}

int mtn_conv1_wgrad_partial_rows(int B, int H, int W) {
    return mtn_reduce_rows_needed(mtn_conv1_wgrad_blocks((long)B * (H / 2) * (W / 2)));
}

int mtn_conv1_wgrad(const float* x_nchw, const void* dy, float* dw, float* partial, int dtype, int B, int H,
                    int W, void* stream) {
    return mtn_conv1_wgrad_run(x_nchw, dy, dw, partial, mtn_storage_dtype(dtype), B, H, W, (hipStream_t)stream);
}

int mtn_conv1_stats(const float* x_nchw, const void* w, float* stats, int dtype, int B, int H, int W, void* stream) {
    return mtn_conv1_stats_run(x_nchw, w, stats, mtn_storage_dtype(dtype), B, H, W, (hipStream_t)stream);
}
int mtn_conv1_bn_act(const float* x_nchw, const void* w, const float* scale_shift, void* out, int act, int dtype, int B,
                     int H, int W, void* stream) {
    return mtn_conv1_bn_act_run(x_nchw, w, scale_shift, out, act, dtype, B, H, W, (hipStream_t)stream);
}
int mtn_conv1_dx(const float* x_nchw, const float* w_oihw, const void* dout, const float* scale_shift, const float* mean_rstd,
                 const float* dgamma, const float* dbeta, float* dx_nchw, int act, int dtype, int B, int H, int W, void* stream) {
    return mtn_conv1_dx_run(x_nchw, w_oihw, dout, scale_shift, mean_rstd, dgamma, dbeta, dx_nchw, act, dtype, B, H, W, (hipStream_t)stream);
}
int mtn_conv1_bwd_partial_rows(int B, int H, int W) {
    const int blocks = mtn_conv1_bwd_blocks((long)B * (H / 2) * (W / 2));
    // producer rows (+ the column-sum scratch the library appends) + one row for the sums themselves
    return mtn_reduce_rows_needed(blocks) + 1;
}
int mtn_conv1_bwd_partial_width(void) { return mtn_conv1_bwd_width(); }
int mtn_conv1_bwd(const float* x_nchw, const void* w, const void* dout, const float* scale_shift, const float* mean_rstd,
                  float* partial, float* dgamma, float* dbeta, float* dw, int act, int dtype, int B, int H, int W,
                  void* stream) {
    return mtn_conv1_bwd_run(x_nchw, w, dout, scale_shift, mean_rstd, partial, dgamma, dbeta, dw, act, dtype, B, H, W,
                             (hipStream_t)stream);
}

int mtn_slab_rows(int rows) { return mtn_reduce_rows_needed(rows); }

int mtn_bn_finalize(float* stats, int tiles, int C, long long count, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, long long* num_batches_tracked, float momentum,
                    float eps, float* scale_shift, float* mean_rstd, void* stream) {
    return mtn_bn_finalize_run(stats, tiles, C, (long)count, gamma, beta, running_mean, running_var,
                               num_batches_tracked, momentum, eps, scale_shift, mean_rstd, (hipStream_t)stream);
}

int mtn_bn_eval_scale_shift(const float* gamma, const float* beta, const float* running_mean,
                            const float* running_var, float eps, float* scale_shift, int C, void* stream) {
    return mtn_bn_eval_run(gamma, beta, running_mean, running_var, eps, scale_shift, C, (hipStream_t)stream);
}

int mtn_bn_act_fwd(const void* y, int ldy, const float* scale_shift, const void* res, int ldres, void* out,
                   int ldo, int dtype, int M, int C, int act, void* stream) {
    return mtn_bn_act_fwd_run(y, ldy, scale_shift, res, ldres, out, ldo, dtype, M, C, act, (hipStream_t)stream);
}

int mtn_bn_bwd_partial_rows(int M) { return mtn_reduce_rows_needed(mtn_bn_bwd_blocks(M)); }

int mtn_bn_act_bwd(const void* dout, int lddo, const void* y, int ldy, const void* res, int ldres,
                   const float* scale_shift, const float* mean_rstd, float* partial, float* coef,
                   float* dgamma, float* dbeta, void* dy, int lddy, void* dres, int lddres,
                   const void* extra, int ldextra, int partial_rows, mtn_bn_handover* record, int dtype, int M, int C, int act,
                   void* stream) {
    BnBwdArgs a{};
    if (partial_rows > 0) {
        MTN_REQUIRE(record, "bn_act_bwd: partial_rows=%d needs the hand-over record mtn_conv_dgrad_bn filled", partial_rows);
        const mtn_bn_handover h = *record;
        *record = mtn_bn_handover{nullptr, 0, 0, 0};     // consumed (or refused) exactly once
        MTN_REQUIRE(h.partial != nullptr, "bn_act_bwd: partial_rows=%d but no mtn_conv_dgrad_bn launch left "
                    "BatchNorm-backward sums in this record (or they were already consumed)", partial_rows);
        MTN_REQUIRE(h.partial == partial && h.tiles == partial_rows && h.channels == C && h.dtype == dtype,
                    "bn_act_bwd: partial_rows=%d rows of %d channels do not match what the preceding mtn_conv_dgrad_bn "
                    "wrote (%d rows of %d channels%s)", partial_rows, C, h.tiles, h.channels, h.partial == partial ? "" : ", another slab");
    }
    a.partial_rows = partial_rows;
    a.dout = dout; a.lddo = lddo; a.y = y; a.ldy = ldy; a.res = res; a.ldr = ldres;
    a.scale_shift = scale_shift; a.mean_rstd = mean_rstd; a.partial = partial; a.coef = coef;
    a.dgamma = dgamma; a.dbeta = dbeta; a.dy = dy; a.lddy = lddy; a.dres = dres; a.lddres = lddres;
    a.extra = extra; a.ldextra = ldextra; a.M = M; a.C = C; a.act = act;
    return mtn_bn_bwd_run(a, dtype, (hipStream_t)stream);
}

int mtn_colsum_rows_per_block(void) { return mtn_fold_rows_per_block(); }
int mtn_colsum_direct_rows(void) { return mtn_fold_direct_rows(); }
int mtn_colsum_grouped(const void* jobs_l1, int n1, int max_w1, int max_rowblocks1, const void* jobs_l2, int n2, int max_w2,
                       void* stream) {
    return mtn_colsum_grouped_run(jobs_l1, n1, max_w1, max_rowblocks1, jobs_l2, n2, max_w2, (hipStream_t)stream);
}

int mtn_tokens_fwd(const void* feat, const float* pos, const float* cls, void* tok, int dtype, int B, int P, void* stream) {
    return mtn_tokens_fwd_run(feat, pos, cls, tok, dtype, B, P, (hipStream_t)stream);
}
int mtn_tokens_bwd(const void* dtok, void* dfeat, float* dcls, int dtype, int B, int P, void* stream) {
    return mtn_tokens_bwd_run(dtok, dfeat, dcls, dtype, B, P, (hipStream_t)stream);
}
int mtn_layernorm_fwd(const void* x, int ldx, const float* gamma, const float* beta, void* y, int ldy, float* mean_rstd,
                      int dtype, int M, float eps, void* stream) {
    return mtn_ln_fwd_run(x, ldx, gamma, beta, y, ldy, mean_rstd, dtype, M, eps, (hipStream_t)stream);
}
int mtn_rowslab_rows(int M) { return mtn_reduce_rows_needed(mtn_rowslab_blocks(M)); }
int mtn_rowslab_producer_rows(int M) { return mtn_rowslab_blocks(M); }
int mtn_layernorm_bwd(const void* dy, int lddy, const void* x, int ldx, const float* mean_rstd, const float* gamma,
                      const void* dres, int lddres, void* dx, int lddx, float* partial, float* dgamma_dbeta, int dtype,
                      int M, void* stream) {
    return mtn_ln_bwd_run(dy, lddy, x, ldx, mean_rstd, gamma, dres, lddres, dx, lddx, partial, dgamma_dbeta, dtype, M,
                          (hipStream_t)stream);
}
int mtn_gelu_bwd(const void* dh, const void* a, void* da, int dtype, long long n, void* stream) {
    return mtn_gelu_bwd_run(dh, a, da, dtype, (long)n, (hipStream_t)stream);
}
int mtn_bias_grad(const void* dy, int lddy, float* partial, float* dbias, int dtype, int M, void* stream) {
    return mtn_bias_grad_run(dy, lddy, partial, dbias, dtype, M, (hipStream_t)stream);
}
int mtn_attention_fwd(const void* qkv, void* out, float* lse, float* attn, int dtype, int B, int N, int H, int D,
                      float scale, void* stream) {
    AttnArgs a{};
    a.qkv = qkv; a.out = out; a.lse = lse; a.attn = attn; a.B = B; a.N = N; a.H = H; a.D = D; a.scale = scale;
    return mtn_attn_fwd_run(a, dtype, (hipStream_t)stream);
}
int mtn_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                      int dtype, int B, int N, int H, int D, float scale, void* stream) {
    AttnArgs a{};
    a.qkv = qkv; a.out = const_cast<void*>(out); a.dout = dout; a.lse = const_cast<float*>(lse); a.delta = delta;
    a.dqkv = dqkv; a.B = B; a.N = N; a.H = H; a.D = D; a.scale = scale;
    return mtn_attn_bwd_run(a, dtype, (hipStream_t)stream);
}
int mtn_cls_head_fwd(const void* tok, long long tok_stride, const float* ln_w, const float* ln_b, const float* W,
                     const float* bias, float* logits, float* xhat, float* mean_rstd, int dtype, int B, int C, float eps,
                     void* stream) {
    return mtn_cls_head_fwd_run(tok, (long)tok_stride, ln_w, ln_b, W, bias, logits, xhat, mean_rstd, dtype, B, C, eps,
                                (hipStream_t)stream);
}
int mtn_cls_head_bwd(const float* dlogits, const float* W, const float* ln_w, const float* ln_b, const float* xhat,
                     const float* mean_rstd, void* dtok, long long tok_stride, float* dW, float* dbias, float* dln_w_b,
                     float* partial, int dtype, int B, int C, void* stream) {
    return mtn_cls_head_bwd_run(dlogits, W, ln_w, ln_b, xhat, mean_rstd, dtok, (long)tok_stride, dW, dbias, dln_w_b, partial,
                                dtype, B, C, (hipStream_t)stream);
}
int mtn_pose_head_fwd(const void* tok, const float* W, const float* bias, float* hm, int dtype, int B, int h, int w, int J,
                      void* stream) {
    return mtn_pose_fwd_run(tok, W, bias, hm, dtype, B, h, w, J, (hipStream_t)stream);     // (F32X3: plain f32 tokens, split-bf16 products)
}
size_t mtn_pose_slab_floats(int B, int h, int w, int J) {
    if (B <= 0 || h <= 0 || w <= 0 || J <= 0) return 0;
    return mtn_pose_slab_floats_run(B, h, w, J);
}
int mtn_pose_head_bwd(const void* tok, const float* W, const float* dhm, void* dtok, float* dW_dbias, float* slab,
                      int dtype, int B, int h, int w, int J, void* stream) {
    return mtn_pose_bwd_run(tok, W, dhm, dtok, dW_dbias, slab, dtype, B, h, w, J, (hipStream_t)stream);
}
int mtn_loss_partial_len(long long n) { return mtn_loss_blocks((long)n); }
int mtn_loss_fwd_bwd(const float* logits, const long long* label, const float* hm, const float* target,
                     const float* target_weight, float* dlogits, float* dhm, float* partial, float* losses, int B, int C,
                     int J, int HW, float ce_weight, void* stream) {
    return mtn_loss_run(logits, label, hm, target, target_weight, dlogits, dhm, partial, losses, B, C, J, HW, ce_weight,
                        (hipStream_t)stream);
}
int mtn_adamw_step(float* p, const float* g, float* m, float* v, long long n, float* state, void* stream) {
    return mtn_adamw_run(p, g, m, v, (long)n, state, (hipStream_t)stream);
}
int mtn_pack_weights(const float* src, const float* scale, void* dst, const int* desc, int ntensors, int max_elems, int dtype,
                     void* stream) {
    return mtn_pack_run(src, scale, dst, desc, ntensors, max_elems, dtype, (hipStream_t)stream);
}

int mtn_x3_convert(const void* src, int ld_src, void* dst, int ld_dst, int to_split, long long M, int C, void* stream) {
    return mtn_x3_convert_run(src, ld_src, dst, ld_dst, to_split, (long)M, C, (hipStream_t)stream);
}

int mtn_heatmap_argmax(const float* hm, float* preds, float* maxvals, int maps, int H, int W, void* stream) {
    return mtn_hm_argmax_run(hm, preds, maxvals, maps, H, W, (hipStream_t)stream);
}
int mtn_pose_accuracy(const float* pred_xy, const float* target_xy, float* acc, int B, int J, int H, int W, float thr,
                      void* stream) {
    return mtn_pose_acc_run(pred_xy, target_xy, acc, B, J, H, W, thr, (hipStream_t)stream);
}
int mtn_cls_confusion(const float* logits, const long long* label, long long* pred_label, int* cm, int B, int C,
                      void* stream) {
    return mtn_cls_confusion_run(logits, label, pred_label, cm, B, C, (hipStream_t)stream);
}
int mtn_macro_f1(const int* cm, float* f1, int C, void* stream) { return mtn_macro_f1_run(cm, f1, C, (hipStream_t)stream); }
int mtn_render_targets(const float* joints, const float* vis, float* target, float* weight, int maps, int HH, int HW,
                       int img_h, int img_w, int sigma, void* stream) {
    return mtn_render_targets_run(joints, vis, target, weight, maps, HH, HW, img_h, img_w, sigma, (hipStream_t)stream);
}
int mtn_normalize_image(const unsigned char* img_hwc_u8, float* out_nchw, int B, int H, int W, void* stream) {
    return mtn_normalize_image_run(img_hwc_u8, out_nchw, B, H, W, (hipStream_t)stream);
}

// The diagnostic keys below change tile / pipeline choices process-wide.  They exist for A/B measurements and
// for tests that must drive every kernel variant; a product process never sets them, and so that "never" is a
// property of the library and not of its callers, they are refused unless the process opted in through its
// environment BEFORE the first call (read once): without MTN_ALLOW_DEBUG_KEYS=1 the launch heuristics are
// immutable for the life of the process (no mutable globals behind the C ABI).
static bool debug_keys_allowed() {
    static const bool ok = [] {
        const char* e = std::getenv("MTN_ALLOW_DEBUG_KEYS");
        return e != nullptr && e[0] == '1';
    }();
    return ok;
}

int mtn_has_experiments(void) { return mtn_igemm_has_experiments() != 0; }

// keys / values that select a measured-and-rejected variant exist in -DMTN_EXPERIMENTS builds only (the CPU executor library,
// diagnostic GPU builds): the product library does not contain those kernels and says so.  (Keys 2 / 14 choose an older
// generation of the pose-head kernels where a newer one applies: the older ones ship only as the fallback for the map
// sizes the newer ones do not take.)
int mtn_debug_key_is_experiment(int key, int value) {
    return (key == 1 && value == 0) || (key == 3 && value != 5) || (key == 4 && value == 256) || (key == 5 && value == 0) ||
           (key == 6 && value >= 4) || (key == 7 && value != 0 && value != 64) || (key == 19 && value != 0 && value != 8) ||
           (key == 20 && value != 0) || (key == 8 && value == 5) || (key == 2 && value != 0) || (key == 14 && value == 0) ||
           ((key == 26 || key == 27) && value != 0) || key == 23;
}

int mtn_conv_kernel_family(int dtype, int dgrad, int B, int H, int W, int Cin, int Cout, int ksize, int stride, int has_res,
                           int has_bn) {
    IgemmArgs p{};
    if (dgrad ? conv_dgrad_args(p, "conv_kernel_family", B, H, W, Cin, Cout, ksize, stride)
              : conv_fwd_args(p, "conv_kernel_family", B, H, W, Cin, Cout, ksize, stride))
        return -1;
    if (has_res) p.res = &p;            // (only their presence matters to the launch rules)
    if (has_bn) p.bn_y = &p;
    return mtn_igemm_family(p, dtype);
}

int mtn_debug_set(int key, int value) {
    if (!debug_keys_allowed()) {
        mtn_set_error("debug_set: diagnostic keys are disabled in this process (set MTN_ALLOW_DEBUG_KEYS=1 in the environment "
                      "before the first call); the product path has no mutable configuration");
        return MTN_ERR_UNSUPPORTED;
    }
    const bool ex = mtn_igemm_has_experiments() != 0;
    const bool needs_ex = mtn_debug_key_is_experiment(key, value) != 0;
    if (needs_ex && !ex) {
        mtn_set_error("debug_set: key %d = %d selects a measured-and-rejected kernel variant that this build does not contain "
                      "(built without -DMTN_EXPERIMENTS)", key, value);
        return MTN_ERR_UNSUPPORTED;
    }
    if (key == 1) { mtn_wgrad_set_tr16(value); return MTN_OK; }
    if (key == 2) { mtn_pose_set_dbg(value); return MTN_OK; }
    if (key == 3) { mtn_igemm_set_stages(value); return MTN_OK; }
    if (key == 4) { mtn_igemm_set_tile_n64(value); return MTN_OK; }
    if (key == 5) { mtn_wgrad_set_dma(value); return MTN_OK; }
    if (key == 6) { mtn_igemm_set_tile(value); return MTN_OK; }
    if (key == 7) { mtn_wgrad_set_tile(value); return MTN_OK; }
    if (key == 8) { mtn_igemm_set_dbg(value); return MTN_OK; }
    if (key == 9) { mtn_igemm_set_window_limit(value); return MTN_OK; }
    if (key == 10) { mtn_wgrad_set_slots(value); return MTN_OK; }
    if (key == 11) { mtn_conv1_set_grid_cap(value); return MTN_OK; }
    if (key == 12) { mtn_conv1_set_bwd_px(value); return MTN_OK; }
    if (key == 13) { mtn_wgrad_set_win(value); return MTN_OK; }
    if (key == 14) { mtn_pose_set_row(value); return MTN_OK; }
    if (key == 15) { mtn_dgrad2_set(value); return MTN_OK; }
    if (key == 16) { mtn_igemm_set_image_tiles(value); return MTN_OK; }
    if (key == 18) { mtn_wgrad_set_row3(value); return MTN_OK; }
    if (key == 19) { mtn_igemm_set_x3_waves(value); return MTN_OK; }
    if (key == 20) { mtn_wgrad_set_x3_sep(value); return MTN_OK; }
    if (key == 21) { mtn_conv1_set_x3(value); return MTN_OK; }
    if (key == 22) { mtn_pw_set(value); return MTN_OK; }
    if (key == 24) { mtn_bn_set_fwd_order(value); return MTN_OK; }
    if (key == 25) { mtn_pose_set_bands(value); return MTN_OK; }
    if (key == 28) { mtn_wgrad_set_slab_bf16(value); return MTN_OK; }
    if (key == 29) { mtn_wgrad_set_ks256(value); return MTN_OK; }
    if (key == 30) { mtn_wgrad_set_group(value); return MTN_OK; }
    if (key == 26 || key == 27) {
#ifdef MTN_EXPERIMENTS
        if (key == 26) mtn_bn_set_ticket(value);
        else mtn_wgrad_set_skip_reduce(value);              // timing ablation (wrong results)
        return MTN_OK;
#else
        if (value == 0) return MTN_OK;
        mtn_set_error("debug_set: key %d selects a measured-and-rejected variant (profiles/r04_tail_ab.txt) that this build does not contain (-DMTN_EXPERIMENTS)", key);
        return MTN_ERR_UNSUPPORTED;
#endif
    }
    if (key == 23) {
#ifdef MTN_EXPERIMENTS
        mtn_win3_set(value);
        return MTN_OK;
#else
        mtn_set_error("debug_set: key 23 selects a measured-and-rejected kernel (win3.hip) that this build does not contain (-DMTN_EXPERIMENTS)");
        return MTN_ERR_UNSUPPORTED;
#endif
    }
    mtn_set_error("debug_set: unknown key %d", key);
    return MTN_ERR_BAD_ARG;
}

int mtn_debug_read(int key, void* dst, size_t bytes) {
    if (key == 1) return mtn_igemm_read_stamps(dst, bytes) ? MTN_ERR_BAD_ARG : MTN_OK;
    mtn_set_error("debug_read: unknown key %d", key);
    return MTN_ERR_BAD_ARG;
}

}  // extern "C"
