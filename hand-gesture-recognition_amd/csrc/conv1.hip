// conv1.hip -- the stem Conv(3, 64, k=3, s=2) of GELANNet (model/gelan.py:155) and its weight
// gradient.  The layer is HBM bound (20 flop/B); both kernels gather the 27-tap patch straight
// from the dataloader's NCHW fp32 image (no layout pass over the input; libs/load.py:46-50 batch
// contract), pad K to 32 and run one-K-step MFMA GEMMs; the forward writes the NHWC activation the
// rest of the backbone uses.  No data gradient: the image needs none.
//
// Fused stem (what the training plan runs): the layer is so cheap to recompute (K = 27) and its
// output so large (B x 96 x 96 x 64: 302 MB in bf16 at B = 256, the largest tensor of the network)
// that the RAW conv output is never stored:
//   conv1_fused<STATS>   pass 1: conv -> BN partial statistics only          (reads the image)
//   conv1_fused<BNACT>   pass 2: conv -> scale/shift -> SiLU -> NHWC activation (reads the image again)
//   conv1_bwd            ONE backward pass over (image, d activation): recomputes the conv, forms
//                        dz = g * silu'(z) and contracts over the pixel axis
//                            P[n][k] = sum_px [dz | a][px][n] * a[px][k]      a = 27-tap patch, a[27] = 1
//                        which holds sum dz*x (the weight-gradient core), sum dz (dbeta), the patch Gram
//                        matrix G = sum a a^T and sum a; with S2 = sum dz*xhat (dgamma) the BatchNorm
//                        backward folds into a closed form (conv1_bwd_finalize):
//                            d raw = scale (dz - m1 - xhat m2),  xhat = rstd (W a - mean)
//                            dW[c][k] = scale_c ( P[c][k] - m1_c Sa[k] - m2_c rstd_c ( (W G)[c][k] - mean_c Sa[k] ) )
//                        -- no d raw tensor, no separate BN reduce / apply passes (they were 3 + 2 sweeps
//                        over 302 MB tensors), and the weight gradient sees un-rounded d raw.
#include <type_traits>
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

// Fused-stem forward passes (MODE 1: BN partial statistics from the fp32 accumulators, nothing
// stored; MODE 2: y = act(conv * scale + shift)).  A workgroup walks tiles of 256 pixels; the 27
// image taps of a thread's NEXT pixel are fetched into registers before the MFMAs / epilogue of the
// current tile, so the stride-2 fp32 gather (the cost of these HBM-bound passes) overlaps them.
// S = storage type of the output (T, or split storage f32x3_t over T = float: the tile is staged as f32 and split on the way out)
template <class T, int MODE, class S = T>
__global__ __launch_bounds__(PXB) void conv1_fused_kernel(const float* __restrict__ x, const T* __restrict__ w,
                                                          S* __restrict__ y, float* __restrict__ stats, int B, int H,
                                                          int W, int OH, int OW, const float* __restrict__ ss, int act) {
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr int KP = 32;
    constexpr int PA = KP + 16 / (int)sizeof(T);
    constexpr int PC = CO + 16 / (int)sizeof(T);
    constexpr int KI = Mma<T>::KI;
    MTN_DYN_SMEM(smem);
    T* sB = reinterpret_cast<T*>(smem);                      // [64][PA] weights, resident
    T* sA = sB + CO * PA;                                    // [256][PA] patches of the tile
    T* sC = sA;                                              // [256][PC] output tile (MODE 2, after the MFMAs)
    float* red = reinterpret_cast<float*>(sA);               // [2][4][64] (MODE 1, after the MFMAs)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    {
        const int n = tid >> 2, k0 = (tid & 3) * 8;
        float v[8];
        MTN_UNROLL for (int e = 0; e < 8; ++e) v[e] = (k0 + e < KK) ? to_f<T>(w[n * KK + k0 + e]) : 0.f;
        MTN_UNROLL for (int c = 0; c < 8; c += VEC) *reinterpret_cast<chunk16*>(sB + n * PA + k0 + c) = pack<T>(v + c);
    }
    const unsigned M = (unsigned)((long)B * OH * OW), ohw = (unsigned)(OH * OW);
    const int ntiles = (int)((M + PXB - 1) / PXB);
    float v[KP];
    MTN_UNROLL for (int k = KK; k < KP; ++k) v[k] = 0.f;
    auto gather = [&](int tile) {
        const unsigned m = (unsigned)tile * PXB + tid;
        MTN_UNROLL for (int k = 0; k < KK; ++k) v[k] = 0.f;
        if (m < M) {
            const unsigned b = m / ohw, rem = m - b * ohw;
            const int oy = (int)(rem / (unsigned)OW), ox = (int)(rem - (unsigned)oy * OW);
            const float* xb = x + (long)b * 3 * H * W;
            // taps are loaded UNDER their validity condition.  (Measured both ways, B = 256: loading every
            // tap from a clamped address and selecting afterwards -- the usual cure for hipcc's branch
            // around a conditional load -- was 15 % slower on the two forward passes and 40 % slower on the
            // backward pass: the clamp / select arithmetic and the extra loads of the 5 padding slots
            // cost more than the branches, which are wave-uniform almost everywhere.)
            MTN_UNROLL for (int kh = 0; kh < 3; ++kh) {
                const int iy = 2 * oy - 1 + kh;
                const int iyc = iy < 0 ? 0 : (iy >= H ? H - 1 : iy);
                MTN_UNROLL for (int kw = 0; kw < 3; ++kw) {
                    const int ix = 2 * ox - 1 + kw;
                    const int ixc = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
                    const bool ok = iy == iyc && ix == ixc;
                    MTN_UNROLL for (int ci = 0; ci < 3; ++ci) {
                        v[(kh * 3 + kw) * 3 + ci] = 0.f;
                        if (ok) v[(kh * 3 + kw) * 3 + ci] = xb[(ci * H + iy) * W + ix];
                    }
                }
            }
        }
    };
    float k_sc[4], k_sh[4];
    if (MODE == 2) {
        MTN_UNROLL for (int j = 0; j < 4; ++j) { k_sc[j] = ss[j * 16 + (lane & 15)]; k_sh[j] = ss[CO + j * 16 + (lane & 15)]; }
    }
    int tile = blockIdx.x;
    if (tile < ntiles) gather(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        MTN_UNROLL for (int c = 0; c < KP; c += VEC) *reinterpret_cast<chunk16*>(sA + tid * PA + c) = pack<T>(v + c);
        MTN_BARRIER_LDS();
        if (tile + (int)gridDim.x < ntiles) gather(tile + gridDim.x);       // in flight under the rest of this tile
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
        MTN_BARRIER_LDS();                                     // every wave is done with the patch tile
        if (MODE == 1) {
            // column sums straight from the accumulators: lane holds pixels 4*(lane>>4)+r of fragment i,
            // channel j*16 + (lane&15); rows past M are all-zero patches and add nothing
            MTN_UNROLL for (int j = 0; j < 4; ++j) {
                float sv = 0.f, sq = 0.f;
                MTN_UNROLL for (int i = 0; i < 4; ++i)
                    MTN_UNROLL for (int r = 0; r < 4; ++r) { const float t = acc[i][j][r]; sv += t; sq += t * t; }
                sv += __shfl_xor(sv, 16, 64); sq += __shfl_xor(sq, 16, 64);
                sv += __shfl_xor(sv, 32, 64); sq += __shfl_xor(sq, 32, 64);
                if (lane < 16) { red[wid * 64 + j * 16 + lane] = sv; red[256 + wid * 64 + j * 16 + lane] = sq; }
            }
            MTN_BARRIER_LDS();
            if (tid < 128) {
                const int which = tid >> 6, c = tid & 63;
                const float* q = red + which * 256 + c;
                stats[((long)tile * 2 + which) * CO + c] = q[0] + q[64] + q[128] + q[192];
            }
        } else {
            MTN_UNROLL for (int i = 0; i < 4; ++i)
                MTN_UNROLL for (int j = 0; j < 4; ++j)
                    MTN_UNROLL for (int r = 0; r < 4; ++r) {
                        const float z = acc[i][j][r] * k_sc[j] + k_sh[j];
                        sC[(wid * 64 + i * 16 + 4 * (lane >> 4) + r) * PC + j * 16 + (lane & 15)] =
                            from_f<T>(act == MTN_ACT_SILU ? siluf_(z) : z);
                    }
            MTN_BARRIER_LDS();
            constexpr int EVS = Ev<S>::N, CPR = CO / EVS;
            for (int idx = tid; idx < PXB * CPR; idx += PXB) {
                const int row = idx / CPR, col = (idx - row * CPR) * EVS;
                const unsigned mm = (unsigned)tile * PXB + row;
                if (mm >= M) continue;
                if constexpr (std::is_same<S, T>::value) {
                    *reinterpret_cast<chunk16*>(y + (long)mm * CO + col) = *reinterpret_cast<const chunk16*>(sC + row * PC + col);
                } else {
                    float f[EVS];
                    MTN_UNROLL for (int e = 0; e < EVS; ++e) f[e] = to_f<T>(sC[row * PC + col + e]);
                    st_vec<S>(y + (long)mm * CO, col, f);
                }
            }
        }
        MTN_BARRIER_LDS();                                     // red / sC are read before the next tile's patches land
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

// ---- fused backward of the stem (see the header) ---------------------------------------------------
constexpr int BW_N = CO + 32;                 // contraction columns [dz (64) | patch (32)]
constexpr int BW_W = BW_N * 32 + CO;          // floats per partial row: P[96][32] then S2[64]
constexpr int ONE_K = KK;                     // patch slot that carries the constant 1

// Work split per 64-pixel step (256 threads): the loaders stage the d-activation tile and the im2col
// patches through registers one step AHEAD (the global latency sits under the previous step's math);
// wave w recomputes raw[ch 16w..16w+15][64 px] (weight fragment as the MFMA row operand, loaded once:
// a lane's quad is 4 consecutive channels of one pixel, so it needs only 4 x 4 BatchNorm constants),
// forms dz, keeps sum dz*xhat in 4 registers and writes dz to LDS for the pixel contraction.
template <class T, bool TR, bool SILU, class S = T>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 3 : 2) void conv1_bwd_kernel(const float* __restrict__ x, const T* __restrict__ w,
                                                        const S* __restrict__ dout, const float* __restrict__ ss,
                                                        const float* __restrict__ mr, float* __restrict__ partial, int B,
                                                        int H, int W, int OH, int OW, int px_per_block) {
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr int KI = Mma<T>::KI;
    constexpr int PY = CO + 16 / (int)sizeof(T), PA = 32 + 16 / (int)sizeof(T);
    constexpr int EVS = Ev<S>::N;                                // channels per load of the d-activation (split storage: 8)
    constexpr int GCH = WPX * (CO / EVS) / 256;                  // d-activation loads per thread per step
    __shared__ __attribute__((aligned(16))) T sA[WPX * PA];      // patches [px][32]
    __shared__ __attribute__((aligned(16))) T sB[CO * PA];       // weights [co][32]
    __shared__ __attribute__((aligned(16))) T sG[WPX * PY];      // d activation [px][64]
    __shared__ __attribute__((aligned(16))) T sD[WPX * PY];      // dz [px][64]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    {
        const int n = tid >> 2, k0 = (tid & 3) * 8;
        float v[8];
        MTN_UNROLL for (int e = 0; e < 8; ++e) v[e] = (k0 + e < KK) ? to_f<T>(w[n * KK + k0 + e]) : 0.f;
        MTN_UNROLL for (int c = 0; c < 8; c += VEC) *reinterpret_cast<chunk16*>(sB + n * PA + k0 + c) = pack<T>(v + c);
    }
    const unsigned M = (unsigned)((long)B * OH * OW);             // < 2^31 (host-checked)
    const unsigned m0 = blockIdx.x * (unsigned)px_per_block;
    unsigned m1 = m0 + (unsigned)px_per_block;
    if (m1 > M) m1 = M;
    const unsigned ohw = (unsigned)(OH * OW);
    // this lane's 4 channels and their constants: z = raw*sc + sh, xhat = raw*rs + b2
    const int c0 = wid * 16 + 4 * (lane >> 4);
    float k_sc[4], k_sh[4], k_rs[4], k_b2[4], s2[4];
    MTN_UNROLL for (int r = 0; r < 4; ++r) {
        const float mean = mr[c0 + r], rstd = mr[CO + c0 + r];
        k_sc[r] = ss[c0 + r]; k_sh[r] = ss[CO + c0 + r]; k_rs[r] = rstd; k_b2[r] = -mean * rstd; s2[r] = 0.f;
    }
    // loader state: d-activation chunks (row, column) and the patch slice (pixel tid/4, k = 8*(tid%4) .. +7)
    RawVec<S> rg[GCH];
    float rv[8];
    const int pr = tid >> 2, pk0 = (tid & 3) * 8;
    // this thread's 8 taps: row / column offset and channel plane (loop invariant)
    int t_dy[8], t_dx[8], t_pl[8];
    MTN_UNROLL for (int e = 0; e < 8; ++e) {
        const int kk = pk0 + e, kkc = kk < KK ? kk : 0;
        t_dy[e] = kkc / 9 - 1; t_dx[e] = (kkc / 3) % 3 - 1; t_pl[e] = (kkc % 3) * H * W;
    }
    auto fetch = [&](unsigned mb) {
        MTN_UNROLL for (int i = 0; i < GCH; ++i) {                // rows past the end: loaded from the last row, zeroed by select
            const int q = tid + i * 256, r = q / (CO / EVS), c = (q - r * (CO / EVS)) * EVS;
            const unsigned mr_ = mb + r < m1 ? mb + r : m1 - 1;
            const RawVec<S> t = ld_raw<S>(dout + (long)mr_ * CO, c);
            rg[i] = mb + r < m1 ? t : zero_raw<S>();
        }
        // taps are loaded under their validity condition (see the forward gather: measured faster than
        // unconditional loads from clamped addresses + selects)
        const unsigned m = mb + pr;
        const unsigned mc = m < m1 ? m : m1 - 1;
        const unsigned b = mc / ohw, rem = mc - b * ohw;
        const int oy = (int)(rem / (unsigned)OW), ox = (int)(rem - (unsigned)oy * OW);
        const float* xb = x + (long)b * 3 * H * W;
        const bool live = m < m1;
        MTN_UNROLL for (int e = 0; e < 8; ++e) {
            const int kk = pk0 + e;
            const int iy = 2 * oy + t_dy[e], ix = 2 * ox + t_dx[e];
            const int iyc = iy < 0 ? 0 : (iy >= H ? H - 1 : iy), ixc = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
            const bool ok = live && kk < KK && iy == iyc && ix == ixc;
            rv[e] = 0.f;
            if (ok) rv[e] = xb[t_pl[e] + iy * W + ix];
            if (kk == ONE_K) rv[e] = live ? 1.f : 0.f;
        }
    };
    auto stage = [&]() {
        MTN_UNROLL for (int i = 0; i < GCH; ++i) {
            const int q = tid + i * 256, r = q / (CO / EVS), c = (q - r * (CO / EVS)) * EVS;
            if constexpr (std::is_same<S, T>::value) {
                *reinterpret_cast<chunk16*>(sG + r * PY + c) = rg[i].c;
            } else {                                             // split storage -> the f32 tile the math reads
                float f[EVS];
                unpack_raw<S>(rg[i], f);
                MTN_UNROLL for (int e = 0; e < EVS; e += VEC) *reinterpret_cast<chunk16*>(sG + r * PY + c + e) = pack<T>(f + e);
            }
        }
        MTN_UNROLL for (int c = 0; c < 8; c += VEC) *reinterpret_cast<chunk16*>(sA + pr * PA + pk0 + c) = pack<T>(rv + c);
    };
    f32x4_t out3[3];
    out3[0] = out3[1] = out3[2] = zero4();
    __syncthreads();                                              // sB
    // the weight fragment of this wave's 16 channels: the row operand of every raw MFMA
    typename Mma<T>::Frag fw[32 / KI];
    MTN_UNROLL for (int ks = 0; ks < 32 / KI; ++ks)
        fw[ks] = Mma<T>::load(sB + (wid * 16 + (lane & 15)) * PA + ks * KI + (lane >> 4) * (KI / 4));
    if (m0 < m1) fetch(m0);
    for (unsigned mb = m0; mb < m1; mb += WPX) {
        stage();
        MTN_BARRIER_LDS();
        if (mb + WPX < m1) fetch(mb + WPX);                       // in flight during this step's math
        MTN_UNROLL for (int i = 0; i < 4; ++i) {                  // pixel fragment i: pixel p = 16 i + (lane & 15)
            const int p = i * 16 + (lane & 15);
            f32x4_t raw = zero4();
            MTN_UNROLL for (int ks = 0; ks < 32 / KI; ++ks)
                raw = Mma<T>::mma(fw[ks], Mma<T>::load(sA + p * PA + ks * KI + (lane >> 4) * (KI / 4)), raw);
            float g[4];
            MTN_UNROLL for (int r = 0; r < 4; ++r) g[r] = to_f<T>(sG[p * PY + c0 + r]);
            f32x4_t dz;
            MTN_UNROLL for (int r = 0; r < 4; ++r) {
                const float z = raw[r] * k_sc[r] + k_sh[r];
                const float d = SILU ? g[r] * silu_grad_(z) : g[r];
                s2[r] += d * (raw[r] * k_rs[r] + k_b2[r]);
                dz[r] = d;
            }
            store4<T>(sD + p * PY + c0, dz);
        }
        MTN_BARRIER_LDS();
        // contraction over the 64 pixels of the step: out[n][k], fragments id = 3*wid + q -> (n = id/2, k = id%2)
        MTN_UNROLL for (int ks = 0; ks < WPX / KI; ++ks) {
            MTN_UNROLL for (int q = 0; q < 3; ++q) {
                const int id = 3 * wid + q, nf = id >> 1, kf = id & 1;
                const typename Mma<T>::Frag fa = nf < 4 ? ColFrag<T, TR>::load(sD + ks * KI * PY + nf * 16, PY, lane)
                                                        : ColFrag<T, TR>::load(sA + ks * KI * PA + (nf - 4) * 16, PA, lane);
                out3[q] = Mma<T>::mma(fa, ColFrag<T, TR>::load(sA + ks * KI * PA + kf * 16, PA, lane), out3[q]);
            }
        }
        MTN_BARRIER_LDS();
    }
    float* out = partial + (long)blockIdx.x * BW_W;
    MTN_UNROLL for (int q = 0; q < 3; ++q) {
        const int id = 3 * wid + q, nf = id >> 1, kf = id & 1;
        MTN_UNROLL for (int r = 0; r < 4; ++r)
            out[(nf * 16 + 4 * (lane >> 4) + r) * 32 + kf * 16 + (lane & 15)] = out3[q][r];
    }
    MTN_UNROLL for (int r = 0; r < 4; ++r) {                      // sum dz*xhat: over the 16 pixel lanes of the wave
        float v = s2[r];
        MTN_UNROLL for (int m = 8; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        if ((lane & 15) == 0) out[BW_N * 32 + c0 + r] = v;        // one writer per channel
    }
}

// sums[BW_W] (column sums of the partial slab) -> dgamma, dbeta, dW (OIHW); one block
template <class T>
__global__ __launch_bounds__(256) void conv1_bwd_finalize_kernel(const float* __restrict__ sums, const T* __restrict__ w,
                                                                 const float* __restrict__ ss, const float* __restrict__ mr,
                                                                 double count, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta, float* __restrict__ dw) {
    for (int t = threadIdx.x; t < CO * KK; t += 256) {
        const int c = t / KK, k = t - c * KK;
        const double S1 = sums[c * 32 + ONE_K], S2 = sums[BW_N * 32 + c];
        const double Sa = sums[(CO + ONE_K) * 32 + k];                       // sum of patch element k
        double WG = 0.0;                                                      // sum_px raw[c] * a[k] = (W G)[c][k]
        for (int q = 0; q < KK; ++q) WG += (double)to_f<T>(w[c * KK + q]) * (double)sums[(CO + q) * 32 + k];
        const double mean = mr[c], rstd = mr[CO + c];
        const double m1 = S1 / count, m2 = S2 / count;
        const double v = (double)ss[c] * ((double)sums[c * 32 + k] - m1 * Sa - m2 * rstd * (WG - mean * Sa));
        dw[c * KK + (k % 3) * 9 + (k / 9) * 3 + (k / 3) % 3] = (float)v;
        if (k == 0) { dbeta[c] = (float)S1; dgamma[c] = (float)S2; }
    }
}


// ---- gradient of the IMAGE through the stem (autograd parity; not part of the training step) ----------------------
// MultiTaskNet is a plain autograd module in the reference (model/multitasknet.py:24-29): x.grad exists when the caller asks
// for it.  The training step never does, so the fused backward above produces no d raw tensor; this kernel recomputes it for
// its own tile and scatters it back through the 3 x 3 / stride-2 taps as a gather:
//   phase 1  work item = (output pixel of the tile's halo, 16 channels): raw = W a (fp32 VALU on the fp32 master weights),
//            z = raw scale + shift, dz = g silu'(z), d raw = scale (dz - m1 - xhat m2)  (m1 = dbeta / M, m2 = dgamma / M:
//            the sums the fused backward left in the BatchNorm gradients)                       -> LDS [pixels][64] f32
//   phase 2  thread = input pixel: dx[c] = sum over its <= 4 output pixels and 64 channels of d raw * W[o][c][ky][kx]
// A tile is 16 x 32 input pixels (9 x 17 output pixels).  One writer per dx element: deterministic.
constexpr int DX_TH = 16, DX_TW = 32, DX_OH = DX_TH / 2 + 1, DX_OW = DX_TW / 2 + 1, DX_OP = DX_OH * DX_OW;
template <class S, bool SILU>
__global__ __launch_bounds__(256) void conv1_dx_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const S* __restrict__ dact, const float* __restrict__ ss,
                                                       const float* __restrict__ mr, const float* __restrict__ dgamma,
                                                       const float* __restrict__ dbeta, float* __restrict__ dx, int B, int H,
                                                       int W, float inv_count) {
    __shared__ float sW[CO * KK];                 // [o][c*9 + ky*3 + kx]  (OIHW as stored)
    __shared__ float sK[6][CO];                   // scale, shift, mean, rstd, m1, m2
    __shared__ float sD[DX_OP][CO + 1];           // d raw of the tile's output pixels (+1: conflict-free column walks)
    const int tid = threadIdx.x;
    const int OH = H / 2, OW = W / 2;
    const int tiles_x = (W + DX_TW - 1) / DX_TW, tiles_y = (H + DX_TH - 1) / DX_TH;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int iy0 = ty * DX_TH, ix0 = tx * DX_TW, oy0 = iy0 / 2, ox0 = ix0 / 2;
    for (int i = tid; i < CO * KK; i += 256) sW[i] = w[i];
    if (tid < CO) {
        sK[0][tid] = ss[tid]; sK[1][tid] = ss[CO + tid]; sK[2][tid] = mr[tid]; sK[3][tid] = mr[CO + tid];
        sK[4][tid] = dbeta[tid] * inv_count; sK[5][tid] = dgamma[tid] * inv_count;
    }
    __syncthreads();
    const float* xb = x + (long)b * 3 * H * W;
    for (int it = tid; it < DX_OP * 4; it += 256) {
        const int p = it >> 2, q = it & 3;
        const int oy = oy0 + p / DX_OW, ox = ox0 + p % DX_OW;
        const bool ok = oy < OH && ox < OW;
        float a[KK];
        MTN_UNROLL for (int c = 0; c < 3; ++c)
            MTN_UNROLL for (int ky = 0; ky < 3; ++ky)
                MTN_UNROLL for (int kx = 0; kx < 3; ++kx) {
                    const int iy = 2 * oy - 1 + ky, ix = 2 * ox - 1 + kx;
                    a[c * 9 + ky * 3 + kx] = (ok && iy >= 0 && iy < H && ix >= 0 && ix < W) ? xb[((long)c * H + iy) * W + ix] : 0.f;
                }
        float g[16];
        if (ok) {
            const S* gp = dact + (((long)b * OH + oy) * OW + ox) * CO;
            MTN_UNROLL for (int e = 0; e < 16; e += Ev<S>::N) ld_vec<S>(gp, q * 16 + e, g + e);
        }
        for (int j = 0; j < 16; ++j) {
            const int o = q * 16 + j;
            float raw = 0.f;
            MTN_UNROLL for (int k = 0; k < KK; ++k) raw += sW[o * KK + k] * a[k];
            const float z = raw * sK[0][o] + sK[1][o];
            float dz = ok ? g[j] : 0.f;
            if (SILU) {
                const float sg = 1.f / (1.f + __expf(-z));
                dz *= sg * (1.f + z * (1.f - sg));
            }
            const float xh = (raw - sK[2][o]) * sK[3][o];
            sD[p][o] = ok ? sK[0][o] * (dz - sK[4][o] - xh * sK[5][o]) : 0.f;
        }
    }
    __syncthreads();
    for (int i = tid; i < DX_TH * DX_TW; i += 256) {
        const int iy = iy0 + i / DX_TW, ix = ix0 + i % DX_TW;
        if (iy >= H || ix >= W) continue;
        float acc[3] = {0.f, 0.f, 0.f};
        MTN_UNROLL for (int ky = 0; ky < 3; ++ky) {
            const int ny = iy + 1 - ky;                       // = 2 oy
            if (ny < 0 || (ny & 1) || ny / 2 >= OH) continue;
            MTN_UNROLL for (int kx = 0; kx < 3; ++kx) {
                const int nx = ix + 1 - kx;
                if (nx < 0 || (nx & 1) || nx / 2 >= OW) continue;
                const float* d = sD[(ny / 2 - oy0) * DX_OW + (nx / 2 - ox0)];
                for (int o = 0; o < CO; ++o) {
                    const float v = d[o];
                    MTN_UNROLL for (int c = 0; c < 3; ++c) acc[c] += v * sW[o * KK + c * 9 + ky * 3 + kx];
                }
            }
        }
        MTN_UNROLL for (int c = 0; c < 3; ++c) dx[(((long)b * 3 + c) * H + iy) * W + ix] = acc[c];
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

// ---- fused stem -------------------------------------------------------------------------------------
static int g_conv1_grid_cap = 0;       // debug key 11: cap on the fused kernels' grid (tests: several tiles per workgroup)
void mtn_conv1_set_grid_cap(int n) { g_conv1_grid_cap = n > 0 ? n : 0; }

template <class T, int MODE, class S = T>
static int conv1_fused_launch_t(const float* x, const void* w, void* y, float* stats, const float* ss, int act, int B, int H,
                                int W, hipStream_t stream) {
    const int OH = H / 2, OW = W / 2;
    const long M = (long)B * OH * OW;
    const int tiles = mtn_conv1_tiles(M);
    // resident weights + the larger of the patch tile and (MODE 2) the output tile / (MODE 1) the reduction scratch
    const size_t ea = (size_t)PXB * (32 * sizeof(T) + 16), ec = MODE == 2 ? (size_t)PXB * (CO * sizeof(T) + 16) : 2048;
    const size_t lds = (size_t)CO * (32 * sizeof(T) + 16) + (ea > ec ? ea : ec);
    // one round of resident workgroups, each walking tiles (prefetching the next one's taps)
    const int per_cu = (int)(160 * 1024 / lds) < 8 ? (int)(160 * 1024 / lds) : 8;
    int grid = tiles < 256 * per_cu ? tiles : 256 * per_cu;
    if (g_conv1_grid_cap && grid > g_conv1_grid_cap) grid = g_conv1_grid_cap;
#ifndef MTN_EMU
    static MtnLdsAttr attr;
    mtn_set_max_lds(attr, &conv1_fused_kernel<T, MODE, S>, lds);
#endif
    MTN_LAUNCH((conv1_fused_kernel<T, MODE, S>), dim3(grid), dim3(PXB), lds, stream, x, (const T*)w, (S*)y, stats, B, H, W, OH, OW, ss, act);
    return mtn_check_launch("conv1_fused");
}

template <int MODE>
static int conv1_fused_launch(const float* x, const void* w, void* y, float* stats, const float* ss, int act, int dtype,
                              int B, int H, int W, hipStream_t stream) {
    MTN_REQUIRE(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "conv1: bad geometry %dx%dx%d", B, H, W);
    MTN_REQUIRE(dtype == MTN_BF16 || dtype == MTN_F32 || dtype == MTN_F32X3, "conv1: bad dtype %d", dtype);
    MTN_REQUIRE((long)B * (H / 2) * (W / 2) * CO < (1L << 31), "conv1: tensor too large");
    // split storage: the stem multiplies in exact f32 (its weights stay plain f32, optim.hip) and writes a split output
    if (dtype == MTN_F32X3) return conv1_fused_launch_t<float, MODE, f32x3_t>(x, w, y, stats, ss, act, B, H, W, stream);
    return dtype == MTN_BF16 ? conv1_fused_launch_t<bf16_t, MODE>(x, w, y, stats, ss, act, B, H, W, stream)
                             : conv1_fused_launch_t<float, MODE>(x, w, y, stats, ss, act, B, H, W, stream);
}

int mtn_conv1_stats_run(const float* x, const void* w, float* stats, int dtype, int B, int H, int W, hipStream_t stream) {
    MTN_REQUIRE(x && w && stats, "conv1_stats: null operand");
    return conv1_fused_launch<1>(x, w, nullptr, stats, nullptr, 0, dtype, B, H, W, stream);
}

int mtn_conv1_bn_act_run(const float* x, const void* w, const float* scale_shift, void* out, int act, int dtype, int B, int H,
                         int W, hipStream_t stream) {
    MTN_REQUIRE(x && w && scale_shift && out, "conv1_bn_act: null operand");
    MTN_REQUIRE(act == MTN_ACT_NONE || act == MTN_ACT_SILU, "conv1_bn_act: act=%d", act);
    return conv1_fused_launch<2>(x, w, out, nullptr, scale_shift, act, dtype, B, H, W, stream);
}

void mtn_conv1_set_x3(int v) { (void)v; }     // (debug key 21 -- the stem backward on split-bf16 products -- retired in round 4)
static int kBwdPx = 3072;           // pixels per workgroup of the fused backward (debug key 12): 768 workgroups = one round at B = 256
void mtn_conv1_set_bwd_px(int n) { kBwdPx = n >= 64 ? (n / 64) * 64 : 3072; }
int mtn_conv1_bwd_blocks(long M) { return (int)((M + kBwdPx - 1) / kBwdPx); }
int mtn_conv1_bwd_width(void) { return BW_W; }

// partial: f32 [mtn_reduce_rows_needed(mtn_conv1_bwd_blocks(M)) + 1][BW_W] (the last row receives the column sums)
int mtn_conv1_bwd_run(const float* x, const void* w, const void* dout, const float* scale_shift, const float* mean_rstd,
                      float* partial, float* dgamma, float* dbeta, float* dw, int act, int dtype, int B, int H, int W,
                      hipStream_t stream) {
    MTN_REQUIRE(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "conv1_bwd: bad geometry");
    MTN_REQUIRE(x && w && dout && scale_shift && mean_rstd && partial && dgamma && dbeta && dw, "conv1_bwd: null operand");
    MTN_REQUIRE(act == MTN_ACT_NONE || act == MTN_ACT_SILU, "conv1_bwd: act=%d", act);
    const int OH = H / 2, OW = W / 2;
    const long M = (long)B * OH * OW;
    const int blocks = mtn_conv1_bwd_blocks(M);
#define MTN_CONV1_BWD(T, TR, SILU, S)                                                                                       \
    MTN_LAUNCH((conv1_bwd_kernel<T, TR, SILU, S>), dim3(blocks), dim3(256), 0, stream, x, (const T*)w, (const S*)dout, scale_shift, \
               mean_rstd, partial, B, H, W, OH, OW, kBwdPx)
    MTN_REQUIRE(dtype == MTN_BF16 || dtype == MTN_F32 || dtype == MTN_F32X3, "conv1_bwd: bad dtype %d", dtype);
    if (dtype == MTN_BF16) {
        if (act == MTN_ACT_SILU) MTN_CONV1_BWD(bf16_t, true, true, bf16_t); else MTN_CONV1_BWD(bf16_t, true, false, bf16_t);
    } else if (dtype == MTN_F32X3) {      // split storage: exact-f32 math on a split output gradient
        if (act == MTN_ACT_SILU) MTN_CONV1_BWD(float, false, true, f32x3_t); else MTN_CONV1_BWD(float, false, false, f32x3_t);
    } else {
        if (act == MTN_ACT_SILU) MTN_CONV1_BWD(float, false, true, float); else MTN_CONV1_BWD(float, false, false, float);
    }
#undef MTN_CONV1_BWD
    int rc = mtn_check_launch("conv1_bwd");
    if (rc) return rc;
    float* sums = partial + (long)mtn_reduce_rows_needed(blocks) * BW_W;     // after the slab and its folding scratch
    rc = mtn_colsum_run(partial, blocks, BW_W, sums, stream);
    if (rc) return rc;
    if (dtype == MTN_BF16)
        MTN_LAUNCH(conv1_bwd_finalize_kernel<bf16_t>, dim3(1), dim3(256), 0, stream, (const float*)sums, (const bf16_t*)w, scale_shift,
                   mean_rstd, (double)M, dgamma, dbeta, dw);
    else
        MTN_LAUNCH(conv1_bwd_finalize_kernel<float>, dim3(1), dim3(256), 0, stream, (const float*)sums, (const float*)w, scale_shift,
                   mean_rstd, (double)M, dgamma, dbeta, dw);
    return mtn_check_launch("conv1_bwd_finalize");
}

// dx f32 NCHW [B][3][H][W] (overwritten).  w: the fp32 MASTER weights OIHW [64][3][3][3]; dact: gradient of the stem's activation
// (NHWC, the mode's storage); dgamma / dbeta: as left by mtn_conv1_bwd_run for the same (x, dact).
int mtn_conv1_dx_run(const float* x, const float* w, const void* dact, const float* scale_shift, const float* mean_rstd,
                     const float* dgamma, const float* dbeta, float* dx, int act, int dtype, int B, int H, int W, hipStream_t stream) {
    MTN_REQUIRE(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "conv1_dx: bad geometry");
    MTN_REQUIRE(x && w && dact && scale_shift && mean_rstd && dgamma && dbeta && dx, "conv1_dx: null operand");
    MTN_REQUIRE(act == MTN_ACT_NONE || act == MTN_ACT_SILU, "conv1_dx: act=%d", act);
    MTN_REQUIRE(dtype == MTN_BF16 || dtype == MTN_F32 || dtype == MTN_F32X3, "conv1_dx: bad dtype %d", dtype);
    MTN_REQUIRE((uintptr_t)dact % 16 == 0, "conv1_dx: dact must be 16-byte aligned");
    const long tiles = (long)B * ((H + DX_TH - 1) / DX_TH) * ((W + DX_TW - 1) / DX_TW);
    MTN_REQUIRE(tiles < (1L << 31), "conv1_dx: too many tiles");
    const float inv = 1.0f / (float)((long)B * (H / 2) * (W / 2));
#define MTN_CONV1_DX(S, SILU) \
    MTN_LAUNCH((conv1_dx_kernel<S, SILU>), dim3((unsigned)tiles), dim3(256), 0, stream, x, w, (const S*)dact, scale_shift, mean_rstd, \
               dgamma, dbeta, dx, B, H, W, inv)
    const bool silu = act == MTN_ACT_SILU;
    if (dtype == MTN_BF16) { if (silu) MTN_CONV1_DX(bf16_t, true); else MTN_CONV1_DX(bf16_t, false); }
    else if (dtype == MTN_F32X3) { if (silu) MTN_CONV1_DX(f32x3_t, true); else MTN_CONV1_DX(f32x3_t, false); }
    else { if (silu) MTN_CONV1_DX(float, true); else MTN_CONV1_DX(float, false); }
#undef MTN_CONV1_DX
    return mtn_check_launch("conv1_dx");
}
