// conv1.hip -- the stem Conv(3, 64, k=3, s=2) of GELANNet (model/gelan.py:155) and its weight
// gradient.  K = 27 is far too shallow for MFMA and the layer is HBM/issue bound (23 flop/B), so
// it is a direct VALU convolution that reads the dataloader's NCHW fp32 image as is (no layout
// pass over the input; libs/load.py:46-50 batch contract) and writes the NHWC activation the
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

// partial[block][co*27 + ci*9 + kh*3 + kw]  (already in the reference's OIHW order)
constexpr int WPX = 64;        // pixels staged per step
template <class T>
__global__ __launch_bounds__(256) void conv1_wgrad_kernel(const float* __restrict__ x, const T* __restrict__ dy,
                                                          float* __restrict__ partial, int B, int H, int W, int OH,
                                                          int OW, int px_per_block) {
    constexpr int VEC = 16 / (int)sizeof(T);
    __shared__ float sdy[WPX][CO];
    __shared__ float sx[WPX][32];
    const int tid = threadIdx.x;
    const int cg = tid >> 5, k = tid & 31;       // 8 channel groups x 32 tap slots (27 used)
    const long M = (long)B * OH * OW;
    const long m0 = (long)blockIdx.x * px_per_block;
    long m1 = m0 + px_per_block;
    if (m1 > M) m1 = M;
    const int ohw = OH * OW;
    float acc[8];
    MTN_UNROLL for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    const int kh = k / 9, kw = (k / 3) % 3, ci = k % 3;    // k = (kh, kw, ci), ci fastest
    for (long mb = m0; mb < m1; mb += WPX) {
        // stage dy rows: WPX x 64 values, 16-byte chunks
        for (int i = tid; i < WPX * (CO / VEC); i += 256) {
            const int r = i / (CO / VEC), c = (i - r * (CO / VEC)) * VEC;
            float v[VEC];
            if (mb + r < m1) unpack<T>(*reinterpret_cast<const chunk16*>(dy + (mb + r) * CO + c), v);
            else { MTN_UNROLL for (int e = 0; e < VEC; ++e) v[e] = 0.f; }
            MTN_UNROLL for (int e = 0; e < VEC; ++e) sdy[r][c + e] = v[e];
        }
        // stage the 27-tap input patch of each pixel
        for (int i = tid; i < WPX * 32; i += 256) {
            const int r = i >> 5, kk = i & 31;
            float v = 0.f;
            const long m = mb + r;
            if (kk < KK && m < m1) {
                const int b = (int)(m / ohw), rem = (int)(m - (long)b * ohw);
                const int oy = rem / OW, ox = rem - oy * OW;
                const int iy = 2 * oy - 1 + kk / 9, ix = 2 * ox - 1 + (kk / 3) % 3;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[(((long)b * 3 + kk % 3) * H + iy) * W + ix];
            }
            sx[r][kk] = v;
        }
        __syncthreads();
        for (int r = 0; r < WPX; ++r) {
            const float xv = sx[r][k];
            MTN_UNROLL for (int e = 0; e < 8; ++e) acc[e] = fmaf(sdy[r][cg * 8 + e], xv, acc[e]);
        }
        __syncthreads();
    }
    if (k < KK) {
        float* out = partial + (long)blockIdx.x * CO * KK;
        MTN_UNROLL for (int e = 0; e < 8; ++e) out[(cg * 8 + e) * KK + ci * 9 + kh * 3 + kw] = acc[e];
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
        static bool once = false;
        if (!once) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1_fwd_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); once = true; }
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
        MTN_LAUNCH(conv1_wgrad_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, x, (const bf16_t*)dy, partial, B, H, W,
                   OH, OW, kWgradPx);
    else
        MTN_LAUNCH(conv1_wgrad_kernel<float>, dim3(blocks), dim3(256), 0, stream, x, (const float*)dy, partial, B, H, W, OH,
                   OW, kWgradPx);
    int rc = mtn_check_launch("conv1_wgrad");
    if (rc) return rc;
    return mtn_colsum_run(partial, blocks, CO * KK, dw, stream);
}
