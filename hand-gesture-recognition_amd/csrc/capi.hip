// capi.hip -- the extern "C" boundary (include/mtn.h): argument checks, error text, dispatch.
#include <cstdarg>
#include <cstdio>
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
int mtn_abi_version(void) { return 1; }

int mtn_conv_stats_tiles(int M, int Cout) { return mtn_igemm_tiles_m(M, Cout); }

int mtn_conv_fwd(const void* x, int ldx, const void* w, void* y, int ldy, float* stats, int dtype,
                 int B, int H, int W, int Cin, int Cout, int ksize, int stride, void* stream) {
    MTN_REQUIRE(ksize == 1 || ksize == 3, "conv_fwd: ksize=%d", ksize);
    MTN_REQUIRE(stride == 1 || stride == 2, "conv_fwd: stride=%d", stride);
    const int pad = ksize / 2;
    IgemmArgs p{};
    p.A = x; p.Wp = w; p.Out = y; p.lda = ldx; p.ldc = ldy;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin;
    p.OH = (H + 2 * pad - ksize) / stride + 1;
    p.OW = (W + 2 * pad - ksize) / stride + 1;
    p.KH = p.KW = ksize;
    p.mul = stride; p.sgn = 1; p.off = -pad; p.div = 1;
    p.M = B * p.OH * p.OW; p.N = Cout; p.K = ksize * ksize * Cin;
    p.stats = stats;
    return mtn_igemm_run(p, dtype, (hipStream_t)stream);
}

int mtn_conv_dgrad(const void* dy, int lddy, const void* wt, void* dx, int lddx, const void* res,
                   int ldres, int dtype, int B, int H, int W, int Cin, int Cout, int ksize,
                   int stride, void* stream) {
    MTN_REQUIRE(ksize == 1 || ksize == 3, "conv_dgrad: ksize=%d", ksize);
    MTN_REQUIRE(stride == 1 || stride == 2, "conv_dgrad: stride=%d", stride);
    const int pad = ksize / 2;
    IgemmArgs p{};
    p.A = dy; p.Wp = wt; p.Out = dx; p.lda = lddy; p.ldc = lddx;
    p.B = B;
    p.H = (H + 2 * pad - ksize) / stride + 1;     // gather source = dy
    p.W = (W + 2 * pad - ksize) / stride + 1;
    p.Cin = Cout;
    p.OH = H; p.OW = W;                             // rows = input pixels
    p.KH = p.KW = ksize;
    p.mul = 1; p.sgn = -1; p.off = pad; p.div = stride;
    p.M = B * H * W; p.N = Cin; p.K = ksize * ksize * Cout;
    p.res = res; p.ldr = ldres;
    return mtn_igemm_run(p, dtype, (hipStream_t)stream);
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

}  // extern "C"
