// conv1.hip -- the stem Conv(3, 64, k=3, s=2) of GELANNet (model/gelan.py:155) and its weight
// gradient.  The layer is HBM bound (20 flop/B); both kernels gather the 27-tap patch straight
// from the dataloader's NCHW fp32 image (no layout pass over the input; libs/load.py:46-50 batch
// contract), pad K to 32 and run one-K-step MFMA GEMMs; the forward writes the NHWC activation the
// rest of the backbone uses.  No data gradient: the image needs none.
#include "mtn_types.h"
#include "mtn_kernels.h"

namespace {

constexpr int CO = 64, KK = 27, PXB = 256;    // output channels, taps*cin, pixels per block

// y[m][64] = sum_{kh,kw,ci} x[b][ci][2oy-1+kh][2ox-1+kw] * w[co][kh][kw][ci]; stats[tile][2][64]
// As a GEMM: M = 256 output pixels per workgroup, N = 64, K = 27 zero-padded to 32 (one bf16 MFMA
// K step).  Each thread gathers the 27-tap patch of ITS pixel from the NCHW image (lanes walk
// along ox: stride-2 fp32 reads), packs it to T and writes one padded LDS row; the weight
// matrix is 64 x 32 in LDS.  The C tile is staged through LDS in T so the 32 KB of NHWC output
// of a workgroup is one fully coalesced block of 16-byte stores and the BN partial statistics
// are column sums of exactly the stored values.
template <class T>
__global__ __launch_bounds__(PXB) void conv1_fwd_kernel(const float* __restrict__ x, const T* __restrict__ w,
                                                        T* __restrict__ y, float* __restrict__ stats, int B, int H,
                                                        int W, int OH, int OW) {
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr int KP = 32;                                   // padded K
    constexpr int PA = KP + 16 / (int)sizeof(T);             // LDS pitch of A / W rows (elements)
    constexpr int PC = CO + 16 / (int)sizeof(T);             // LDS pitch of the C tile (elements)
    constexpr int KI = Mma<T>::KI;
    MTN_DYN_SMEM(smem);
    T* sA = reinterpret_cast<T*>(smem);                      // [256][PA]
    T* sB = sA + PXB * PA;                                   // [64][PA]
    T* sC = reinterpret_cast<T*>(smem);                      // [256][PC] (after the MFMAs)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // weights: thread -> (row n = tid/4, 8 consecutive k)
    {
        const int n = tid >> 2, k0 = (tid & 3) * 8;
        float v[8];
        MTN_UNROLL for (int e = 0; e < 8; ++e) v[e] = (k0 + e < KK) ? to_f<T>(w[n * KK + k0 + e]) : 0.f;
        MTN_UNROLL for (int c = 0; c < 8; c += VEC) *reinterpret_cast<chunk16*>(sB + n * PA + k0 + c) = pack<T>(v + c);
    }
    const long M = (long)B * OH * OW;
    const long m = (long)blockIdx.x * PXB + tid;
    {
        float v[KP];
        MTN_UNROLL for (int k = 0; k < KP; ++k) v[k] = 0.f;
        if (m < M) {
            const int ohw = OH * OW;
            const int b = (int)(m / ohw), rem = (int)(m - (long)b * ohw);
            const int oy = rem / OW, ox = rem - oy * OW;
            const float* xb = x + (long)b * 3 * H * W;
            MTN_UNROLL for (int kh = 0; kh < 3; ++kh) {
                const int iy = 2 * oy - 1 + kh;
                MTN_UNROLL for (int kw = 0; kw < 3; ++kw) {
                    const int ix = 2 * ox - 1 + kw;
                    const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
                    MTN_UNROLL for (int ci = 0; ci < 3; ++ci)
                        v[(kh * 3 + kw) * 3 + ci] = ok ? xb[((long)ci * H + iy) * W + ix] : 0.f;
                }
            }
        }
        MTN_UNROLL for (int c = 0; c < KP; c += VEC) *reinterpret_cast<chunk16*>(sA + tid * PA + c) = pack<T>(v + c);
    }
    __syncthreads();
    f32x4_t acc[4][4];
    MTN_UNROLL for (int i = 0; i < 4; ++i)
        MTN_UNROLL for (int j = 0; j < 4; ++j) acc[i][j] = zero4();
    {
        const T* a = sA + (wid * 64 + (lane & 15)) * PA + (lane >> 4) * (KI / 4);
        const T* bq = sB + (lane & 15) * PA + (lane >> 4) * (KI / 4);
        MTN_UNROLL for (int ks = 0; ks < KP / KI; ++ks) {
            typename Mma<T>::Frag fa[4], fb[4];
            MTN_UNROLL for (int i = 0; i < 4; ++i) fa[i] = Mma<T>::load(a + i * 16 * PA + ks * KI);
            MTN_UNROLL for (int j = 0; j < 4; ++j) fb[j] = Mma<T>::load(bq + j * 16 * PA + ks * KI);
            MTN_UNROLL for (int i = 0; i < 4; ++i)
                MTN_UNROLL for (int j = 0; j < 4; ++j) acc[i][j] = Mma<T>::mma(fa[i], fb[j], acc[i][j]);
        }
    }
    __syncthreads();
    MTN_UNROLL for (int i = 0; i < 4; ++i)
        MTN_UNROLL for (int j = 0; j < 4; ++j)
            MTN_UNROLL for (int r = 0; r < 4; ++r)
                sC[(wid * 64 + i * 16 + 4 * (lane >> 4) + r) * PC + j * 16 + (lane & 15)] = from_f<T>(acc[i][j][r]);
    __syncthreads();
    constexpr int CPR = CO / VEC;
    for (int idx = tid; idx < PXB * CPR; idx += PXB) {
        const int row = idx / CPR, col = (idx - row * CPR) * VEC;
        const long mm = (long)blockIdx.x * PXB + row;
        if (mm < M) *reinterpret_cast<chunk16*>(y + mm * CO + col) = *reinterpret_cast<const chunk16*>(sC + row * PC + col);
    }
    if (stats) {
        const int c = tid & 63, part = tid >> 6;
        float s = 0.f, ss = 0.f;
        long rows = M - (long)blockIdx.x * PXB;
        if (rows > PXB) rows = PXB;
        for (int r = part * 64; r < part * 64 + 64 && r < rows; ++r) { const float v = to_f<T>(sC[r * PC + c]); s += v; ss += v * v; }
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);         // [2][4][64]
        red[part * 64 + c] = s;
        red[256 + part * 64 + c] = ss;
        __syncthreads();
        if (tid < 128) {
            const int which = tid >> 6;
            const float* q = red + which * 256 + c;
            stats[((long)blockIdx.x * 2 + which) * CO + c] = q[0] + q[64] + q[128] + q[192];
        }
    }
}

// Weight gradient of the stem as a TN GEMM on MFMA: dW[co][k] = sum_px dY[px][co] * A[px][k], A = the
// 27-tap im2col patch (padded to 32) gathered from the NCHW fp32 image like in the forward.
// Per step a workgroup stages 64 pixels of dY ([px][64], 16-byte coalesced) and of A ([px][32]) in
// LDS and reads both operands down the pixel axis (ds_read_b64_tr_b16 in bf16); wave w owns output
// channels 16w..16w+15 x all 32 k.  partial[block][co*27 + ci*9 + kh*3 + kw] (reference OIHW order).
constexpr int WPX = 64;        // pixels staged per step
template <class T, bool TR>
__global__ __launch_bounds__(256) void conv1_wgrad_kernel(const float* __restrict__ x, const T* __restrict__ dy,
                                                          float* __restrict__ partial, int B, int H, int W, int OH,
                                                          int OW, int px_per_block) {
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr int KI = Mma<T>::KI;
    constexpr int PY = CO + 16 / (int)sizeof(T), PA = 32 + 16 / (int)sizeof(T);
    __shared__ __attribute__((aligned(16))) T sY[WPX * PY];
    __shared__ __attribute__((aligned(16))) T sA[WPX * PA];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const long M = (long)B * OH * OW;
    const long m0 = (long)blockIdx.x * px_per_block;
    long m1 = m0 + px_per_block;
    if (m1 > M) m1 = M;
    const int ohw = OH * OW;
    f32x4_t acc0 = zero4(), acc1 = zero4();
    for (long mb = m0; mb < m1; mb += WPX) {
        // dY rows: 64 px x 64 ch
        for (int i = tid; i < WPX * (CO / VEC); i += 256) {
            const int r = i / (CO / VEC), c = (i - r * (CO / VEC)) * VEC;
            *reinterpret_cast<chunk16*>(sY + r * PY + c) =
                (mb + r < m1) ? *reinterpret_cast<const chunk16*>(dy + (mb + r) * CO + c) : zero_chunk();
        }
        // im2col patch: thread -> (pixel tid/4, k = 8*(tid%4) .. +7)
        {
            const int r = tid >> 2, k0 = (tid & 3) * 8;
            const long m = mb + r;
            float v[8];
            MTN_UNROLL for (int e = 0; e < 8; ++e) v[e] = 0.f;
            if (m < m1) {
                const int b = (int)(m / ohw), rem = (int)(m - (long)b * ohw);
                const int oy = rem / OW, ox = rem - oy * OW;
                MTN_UNROLL for (int e = 0; e < 8; ++e) {
                    const int kk = k0 + e;
                    const int iy = 2 * oy - 1 + kk / 9, ix = 2 * ox - 1 + (kk / 3) % 3;
                    if (kk < KK && iy >= 0 && iy < H && ix >= 0 && ix < W) v[e] = x[(((long)b * 3 + kk % 3) * H + iy) * W + ix];
                }
            }
            MTN_UNROLL for (int c = 0; c < 8; c += VEC) *reinterpret_cast<chunk16*>(sA + r * PA + k0 + c) = pack<T>(v + c);
        }
        __syncthreads();
        MTN_UNROLL for (int ks = 0; ks < WPX / KI; ++ks) {
            const typename Mma<T>::Frag fa = ColFrag<T, TR>::load(sY + ks * KI * PY + wid * 16, PY, lane);
            acc0 = Mma<T>::mma(fa, ColFrag<T, TR>::load(sA + ks * KI * PA, PA, lane), acc0);
            acc1 = Mma<T>::mma(fa, ColFrag<T, TR>::load(sA + ks * KI * PA + 16, PA, lane), acc1);
        }
        __syncthreads();
    }
    float* out = partial + (long)blockIdx.x * CO * KK;
    MTN_UNROLL for (int r = 0; r < 4; ++r) {
        const int co = wid * 16 + 4 * (lane >> 4) + r;
        const int k0 = lane & 15, k1 = 16 + (lane & 15);
        out[co * KK + (k0 % 3) * 9 + (k0 / 9) * 3 + (k0 / 3) % 3] = acc0[r];
        if (k1 < KK) out[co * KK + (k1 % 3) * 9 + (k1 / 9) * 3 + (k1 / 3) % 3] = acc1[r];
    }
}

}  // namespace

int mtn_conv1_tiles(long M) { return (int)((M + PXB - 1) / PXB); }

int mtn_conv1_fwd_run(const float* x, const void* w, void* y, float* stats, int dtype, int B, int H, int W,
                      hipStream_t stream) {
    MTN_REQUIRE(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "conv1_fwd: bad geometry %dx%dx%d", B, H, W);
    const int OH = H / 2, OW = W / 2;
    const long M = (long)B * OH * OW;
    MTN_REQUIRE(M * CO < (1L << 31), "conv1_fwd: tensor too large");
    const int blocks = mtn_conv1_tiles(M);
    if (dtype == MTN_BF16) {
        const size_t lds = (size_t)PXB * (CO + 8) * 2;       // C tile (bf16) >= A + W tiles
        MTN_LAUNCH(conv1_fwd_kernel<bf16_t>, dim3(blocks), dim3(PXB), lds, stream, x, (const bf16_t*)w, (bf16_t*)y, stats, B,
                   H, W, OH, OW);
    } else {
        const size_t lds = (size_t)PXB * (CO + 4) * 4;
#ifndef MTN_EMU
        static MtnLdsAttr attr;
        mtn_set_max_lds(attr, &conv1_fwd_kernel<float>, lds);
#endif
        MTN_LAUNCH(conv1_fwd_kernel<float>, dim3(blocks), dim3(PXB), lds, stream, x, (const float*)w, (float*)y, stats, B, H,
                   W, OH, OW);
    }
    return mtn_check_launch("conv1_fwd");
}

static const int kWgradPx = 2048;
int mtn_conv1_wgrad_blocks(long M) { return (int)((M + kWgradPx - 1) / kWgradPx); }

int mtn_conv1_wgrad_run(const float* x, const void* dy, float* dw, float* partial, int dtype, int B, int H, int W,
                        hipStream_t stream) {
    MTN_REQUIRE(B > 0 && H % 2 == 0 && W % 2 == 0, "conv1_wgrad: bad geometry");
    const int OH = H / 2, OW = W / 2;
    const long M = (long)B * OH * OW;
    const int blocks = mtn_conv1_wgrad_blocks(M);
    if (dtype == MTN_BF16)
        MTN_LAUNCH((conv1_wgrad_kernel<bf16_t, true>), dim3(blocks), dim3(256), 0, stream, x, (const bf16_t*)dy, partial, B, H, W,
                   OH, OW, kWgradPx);
    else
        MTN_LAUNCH((conv1_wgrad_kernel<float, false>), dim3(blocks), dim3(256), 0, stream, x, (const float*)dy, partial, B, H, W, OH,
                   OW, kWgradPx);
    int rc = mtn_check_launch("conv1_wgrad");
    if (rc) return rc;
    return mtn_colsum_run(partial, blocks, CO * KK, dw, stream);
}
