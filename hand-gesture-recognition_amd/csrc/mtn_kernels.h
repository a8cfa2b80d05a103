// mtn_kernels.h -- internal launch interfaces between the C ABI (capi.hip) and the kernel files.
#pragma once
#include "mtn.h"
#include "mtn_rt.h"

#define MTN_F32 MTN_DTYPE_F32
#define MTN_BF16 MTN_DTYPE_BF16

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
    int act;              // MTN_ACT_NONE / MTN_ACT_GELU
    void* aux_out;        // pre-activation copy [M][ld_aux] (GELU backward) or null
    int ld_aux;
    const void* res;      // residual [M][ldr] added after act, or null (may alias Out)
    int ldr;
    float* stats;         // [tiles_m][2][N] BN partial sums, or null
};

int mtn_igemm_run(const IgemmArgs& p, int dtype, hipStream_t stream);
int mtn_igemm_tiles_m(int M, int N);
