// heads.hip -- pose head and the fused training loss.
//
// Pose head (model/transformer.py:118-127,146-150): tokens[:,1:] viewed as [B,256,h,w] ->
//   F.interpolate(x4, bilinear, align_corners=True) -> ReLU -> Conv2d(256 -> J, 1x1, bias) -> NCHW fp32.
// The 16x larger upsampled tensor is never materialised: a workgroup interpolates + ReLUs an 8x8
// output tile into LDS and contracts it with the J x 256 weight matrix from LDS.  The backward
// recomputes the upsampled features per (image, channel chunk), accumulates the bilinear scatter
// in an LDS-resident token-gradient tile and writes dW partials to a per-image slab.
//
// Loss (libs/loss.py:4-40, train.py:63-64,75): total = 0.001 * CrossEntropy(mean) + JointsMSE,
// JointsMSE == 0.5 * mean_{b,j,p} (w_bj (p - t))^2; the kernel emits the three scalars and both
// gradients (dlogits, dheatmap) in one pass.
#include <type_traits>
#include "mtn_types.h"
#include "mtn_kernels.h"

namespace {

constexpr int DIM = 256;
constexpr int MAXJ = 24;

// PyTorch's align_corners=True source index: src = dst * (in-1)/(out-1), in fp32
struct Lerp { int i0, i1; float w1; };
__device__ __forceinline__ Lerp lerp_coord(int dst, int in, int out) {
    const float r = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    const float s = r * (float)dst;
    Lerp l;
    l.i0 = (int)s;
    if (l.i0 > in - 1) l.i0 = in - 1;
    l.i1 = l.i0 + (l.i0 < in - 1 ? 1 : 0);
    l.w1 = s - (float)l.i0;
    return l;
}

constexpr int TP = 8;                 // output tile edge

// Forward: a workgroup owns an 8x8 tile of output pixels of one image.  Phase 1 (VALU) builds
// U[64 px][256] = ReLU(bilinear(tokens)) in LDS in T; phase 2 contracts it with W (J x 256, padded
// to 32 rows, T) on MFMA: M = 64 pixels (one 16-row tile per wave), N = 32, K = 256.
template <class T>
__global__ __launch_bounds__(256) void pose_fwd_kernel(const T* __restrict__ tok, const float* __restrict__ W,
                                                       const float* __restrict__ bias, float* __restrict__ hm, int h, int w,
                                                       int J) {
    constexpr int PU = DIM + 16 / (int)sizeof(T);
    constexpr int KI = Mma<T>::KI;
    MTN_DYN_SMEM(smem);
    T* sU = reinterpret_cast<T*>(smem);                 // [64][PU]
    T* sW = sU + TP * TP * PU;                          // [32][PU]
    const int OH = 4 * h, OW = 4 * w;
    const int tiles_x = (OW + TP - 1) / TP;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    for (int i = tid; i < 32 * (DIM / 4); i += 256) {
        const int j = i / (DIM / 4), c = (i - j * (DIM / 4)) * 4;
        float v[4];
        MTN_UNROLL for (int e = 0; e < 4; ++e) v[e] = j < J ? W[j * DIM + c + e] : 0.f;
        MTN_UNROLL for (int e = 0; e < 4; ++e) sW[j * PU + c + e] = from_f<T>(v[e]);
    }
    {   // phase 1: thread = (pixel lane, 4 channels)
        const int cg = tid & 63, pq = tid >> 6;
        const T* base = tok + ((long)b * (h * w + 1) + 1) * DIM + cg * 4;
        for (int px = pq; px < TP * TP; px += 4) {
            const int oy = ty * TP + px / TP, ox = tx * TP + px % TP;
            float u[4] = {0.f, 0.f, 0.f, 0.f};
            if (oy < OH && ox < OW) {
                const Lerp ly = lerp_coord(oy, h, OH), lx = lerp_coord(ox, w, OW);
                float t0[4], t1[4], t2[4], t3[4];
                const T* p00 = base + (long)(ly.i0 * w + lx.i0) * DIM;
                const T* p01 = base + (long)(ly.i0 * w + lx.i1) * DIM;
                const T* p10 = base + (long)(ly.i1 * w + lx.i0) * DIM;
                const T* p11 = base + (long)(ly.i1 * w + lx.i1) * DIM;
                MTN_UNROLL for (int e = 0; e < 4; ++e) {
                    t0[e] = to_f<T>(p00[e]); t1[e] = to_f<T>(p01[e]); t2[e] = to_f<T>(p10[e]); t3[e] = to_f<T>(p11[e]);
                }
                const float wy0 = 1.f - ly.w1, wx0 = 1.f - lx.w1;
                MTN_UNROLL for (int e = 0; e < 4; ++e) {
                    const float v = wy0 * (wx0 * t0[e] + lx.w1 * t1[e]) + ly.w1 * (wx0 * t2[e] + lx.w1 * t3[e]);
                    u[e] = v > 0.f ? v : 0.f;
                }
            }
            MTN_UNROLL for (int e = 0; e < 4; ++e) sU[px * PU + cg * 4 + e] = from_f<T>(u[e]);
        }
    }
    __syncthreads();
    // phase 2: wave w -> pixels 16w..16w+15, joints 0..31
    const int lane = tid & 63, wid = tid >> 6, g = lane >> 4, li = lane & 15;
    f32x4_t a0 = zero4(), a1 = zero4();
    const T* ap = sU + (wid * 16 + li) * PU + g * (KI / 4);
    const T* bp = sW + li * PU + g * (KI / 4);
    MTN_UNROLL for (int ks = 0; ks < DIM / KI; ++ks) {
        const typename Mma<T>::Frag fa = Mma<T>::load(ap + ks * KI);
        a0 = Mma<T>::mma(fa, Mma<T>::load(bp + ks * KI), a0);
        a1 = Mma<T>::mma(fa, Mma<T>::load(bp + 16 * PU + ks * KI), a1);
    }
    MTN_UNROLL for (int r = 0; r < 4; ++r) {
        const int px = wid * 16 + 4 * g + r;
        const int oy = ty * TP + px / TP, ox = tx * TP + px % TP;
        if (oy < OH && ox < OW) {
            if (li < J) hm[(((long)b * J + li) * OH + oy) * OW + ox] = a0[r] + bias[li];
            if (16 + li < J) hm[(((long)b * J + 16 + li) * OH + oy) * OW + ox] = a1[r] + bias[16 + li];
        }
    }
}

// Backward.  block = (channel chunk of CC, image b), 256 threads = CC channels x PL lanes.
// The heat-map gradient of the image streams through LDS RB output rows at a time.
//   phase 1 (thread = channel x group of 4 consecutive pixels): recompute u, g = sum_j d_j W_jc with
//            one ds_read_b128 of d per joint per 4 pixels, dW[j][c] += d_j * relu(u) in registers,
//            du = g * [u > 0] -> LDS;
//   phase 2 (thread = channel x OWNED source column): gather the transposed bilinear stencil from
//            du: every token-gradient cell has exactly one writer, so there are no atomics and the
//            result is deterministic.
// Interpolation coordinates come from per-block LDS tables (PyTorch's align_corners=True map).
// slab row b = [dW (J*256) | dbias (J)].
constexpr int RB = 4;
constexpr int MAXTAP = 12;      // an output column range of at most 2*4+2 touches one source column at scale 4
template <class T, int CC, int JT>     // JT = compile-time joint count (0: runtime J <= MAXJ)
__global__ __launch_bounds__(256) void pose_bwd_kernel(const T* __restrict__ tok, const float* __restrict__ W,
                                                       const float* __restrict__ dhm, T* __restrict__ dtok,
                                                       float* __restrict__ slab, int h, int w, int Jrt) {
    constexpr int PL = 256 / CC;
    constexpr int JL = JT ? JT : MAXJ;                  // unrolled loop bound
    const int J = JT ? JT : Jrt;
    MTN_DYN_SMEM(smem);
    const int hw = h * w, OH = 4 * h, OW = 4 * w, OHW = OH * OW, RP = RB * OW;
    float* sS = reinterpret_cast<float*>(smem);         // source tokens [hw][CC]
    float* sG = sS + hw * CC;                           // token gradient [hw][CC]
    float* sD = sG + hw * CC;                           // [J][RP] rows of dhm
    float* sU = sD + J * RP;                            // du [RP][CC]; sD|sU are reused as [PL][J][CC] at the end
    const int tail = (J * RP + RP * CC > PL * J * CC) ? J * RP + RP * CC : PL * J * CC;
    int* tI0 = reinterpret_cast<int*>(sD + tail);       // [OH + OW] lerp tables
    int* tI1 = tI0 + OH + OW;
    float* tW = reinterpret_cast<float*>(tI1 + OH + OW);
    int* tLo = reinterpret_cast<int*>(tW + OH + OW);    // [w] first / last output column touching a source column
    int* tHi = tLo + w;
    float* tTap = reinterpret_cast<float*>(tHi + w);    // [w][MAXTAP] x-weights of output columns lo.. on column sx
    // XCD-aware order: the DIM/CC channel-chunk workgroups of one image all stream the same heat-map
    // gradient; keep them on one XCD group (id % 8) and adjacent in time so it is fetched once.
    constexpr int NCH = DIM / CC;
    int b, chunk;
    {
        const int L = blockIdx.x, B = gridDim.x / NCH;
        if (B % 8 == 0) { const int xcd = L & 7, j = L >> 3; b = (j / NCH) * 8 + xcd; chunk = j % NCH; }
        else { b = L / NCH; chunk = L % NCH; }
    }
    const int c0 = chunk * CC;
    const int c = threadIdx.x % CC, pl = threadIdx.x / CC;
    const T* base = tok + ((long)b * (hw + 1) + 1) * DIM + c0;
    for (int i = threadIdx.x; i < hw * CC; i += 256) {
        const int p = i / CC, cc = i - p * CC;
        sS[i] = to_f<T>(base[(long)p * DIM + cc]);
        sG[i] = 0.f;
    }
    for (int i = threadIdx.x; i < OH + OW; i += 256) {
        const Lerp l = i < OH ? lerp_coord(i, h, OH) : lerp_coord(i - OH, w, OW);
        tI0[i] = l.i0; tI1[i] = l.i1; tW[i] = l.w1;
    }
    __syncthreads();
    for (int sx = threadIdx.x; sx < w; sx += 256) {
        int lo = OW, hi = -1;
        for (int ox = 0; ox < OW; ++ox)
            if (tI0[OH + ox] == sx || tI1[OH + ox] == sx) { if (ox < lo) lo = ox; hi = ox; }
        tLo[sx] = lo; tHi[sx] = hi;
        for (int k = 0; k < MAXTAP; ++k) {
            const int ox = lo + k;
            float wgt = 0.f;
            if (ox <= hi) {
                const int x0 = tI0[OH + ox], x1 = tI1[OH + ox];
                const float wx = tW[OH + ox];
                wgt = (x0 == sx ? 1.f - wx : 0.f) + ((x1 == sx && x1 != x0) ? wx : 0.f);
            }
            tTap[sx * MAXTAP + k] = wgt;
        }
    }
    float wj[JL], dw[JL], db[JL];
    MTN_UNROLL for (int j = 0; j < JL; ++j) { wj[j] = (JT || j < J) ? W[j * DIM + c0 + c] : 0.f; dw[j] = 0.f; db[j] = 0.f; }
    const float* dh = dhm + (long)b * J * OHW;
    const bool do_bias = chunk == 0;
    for (int r0 = 0; r0 < OH; r0 += RB) {
        __syncthreads();
        for (int i = threadIdx.x; i < J * RP; i += 256) {
            const int j = i / RP, p = i - j * RP;
            sD[i] = (r0 * OW + p < OHW) ? dh[(long)j * OHW + r0 * OW + p] : 0.f;
        }
        __syncthreads();
        if (do_bias) {
            for (int p = threadIdx.x; p < RP; p += 256)
                MTN_UNROLL for (int j = 0; j < JL; ++j)
                    if (JT || j < J) db[j] += sD[j * RP + p];
        }
        // phase 1
        for (int gq = pl; gq < RP / 4; gq += PL) {
            const int p0 = gq * 4, r = p0 / OW, ox0 = p0 - r * OW, oy = r0 + r;
            float u[4], gs[4];
            if (oy < OH) {
                const int y0 = tI0[oy] * w, y1 = tI1[oy] * w;
                const float wy = tW[oy];
                MTN_UNROLL for (int e = 0; e < 4; ++e) {
                    const int x0 = tI0[OH + ox0 + e], x1 = tI1[OH + ox0 + e];
                    const float wx = tW[OH + ox0 + e];
                    u[e] = (1.f - wy) * ((1.f - wx) * sS[(y0 + x0) * CC + c] + wx * sS[(y0 + x1) * CC + c]) +
                           wy * ((1.f - wx) * sS[(y1 + x0) * CC + c] + wx * sS[(y1 + x1) * CC + c]);
                    gs[e] = 0.f;
                }
                MTN_UNROLL for (int j = 0; j < JL; ++j) {
                    if (JT || j < J) {
                        const float4 d = *reinterpret_cast<const float4*>(sD + j * RP + p0);
                        const float ww = wj[j];
                        gs[0] += d.x * ww; gs[1] += d.y * ww; gs[2] += d.z * ww; gs[3] += d.w * ww;
                        dw[j] += d.x * fmaxf(u[0], 0.f) + d.y * fmaxf(u[1], 0.f) + d.z * fmaxf(u[2], 0.f) + d.w * fmaxf(u[3], 0.f);
                    }
                }
                MTN_UNROLL for (int e = 0; e < 4; ++e) sU[(p0 + e) * CC + c] = u[e] > 0.f ? gs[e] : 0.f;
            } else {
                MTN_UNROLL for (int e = 0; e < 4; ++e) sU[(p0 + e) * CC + c] = 0.f;
            }
        }
        __syncthreads();
        // phase 2: thread (c, owner pl) owns source columns pl, pl+PL, ...
        for (int sx = pl; sx < w; sx += PL) {
            const int lo = tLo[sx], hi = tHi[sx];
            const int nt = hi - lo + 1;
            for (int r = 0; r < RB && r0 + r < OH; ++r) {
                float sum = 0.f;
                const float* up = sU + (r * OW + lo) * CC + c;
                for (int k = 0; k < nt; ++k) sum += tTap[sx * MAXTAP + k] * up[k * CC];
                const int oy = r0 + r;
                const float wy = tW[oy];
                sG[(tI0[oy] * w + sx) * CC + c] += (1.f - wy) * sum;
                if (tI1[oy] != tI0[oy]) sG[(tI1[oy] * w + sx) * CC + c] += wy * sum;
            }
        }
    }
    __syncthreads();
    MTN_UNROLL for (int j = 0; j < JL; ++j)
        if (JT || j < J) sD[(pl * J + j) * CC + c] = dw[j];
    __syncthreads();
    for (int i = threadIdx.x; i < hw * CC; i += 256) {
        const int p = i / CC, cc = i - p * CC;
        dtok[((long)b * (hw + 1) + 1 + p) * DIM + c0 + cc] = from_f<T>(sG[i]);
    }
    float* row = slab + (long)b * J * (DIM + 1);
    for (int i = threadIdx.x; i < J * CC; i += 256) {
        const int j = i / CC, cc = i - j * CC;
        float s = 0.f;
        for (int q = 0; q < PL; ++q) s += sD[(q * J + j) * CC + cc];
        row[j * DIM + c0 + cc] = s;
    }
    if (do_bias) {
        __syncthreads();
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
        MTN_UNROLL for (int j = 0; j < JL; ++j) {
            if (JT || j < J) {
                const float t = wave_sum(db[j]);
                if (lane == 0) sD[wid * J + j] = t;
            }
        }
        __syncthreads();
        if (threadIdx.x < J) row[J * DIM + threadIdx.x] = sD[threadIdx.x] + sD[J + threadIdx.x] + sD[2 * J + threadIdx.x] + sD[3 * J + threadIdx.x];
    }
}

// Backward on MFMA.  block = (32-channel chunk, image); the heat-map gradient streams through LDS
// RB output rows (<= 96 pixels) at a time and every contraction runs on the matrix cores:
//   Y_r[sx][c]  = (1-wy) S[y0][sx][c] + wy S[y1][sx][c]          VALU (rows of the bilinear map)
//   R_r[ox][c]  = relu(sum_sx X[ox][sx] Y_r[sx][c])              MFMA in f32 (exact products: the sign of u
//                                                                  decides the ReLU mask, a rounded operand
//                                                                  would flip it where u ~ 0), K = source columns
//   dW[j][c]   += sum_p D[j][p] R[p][c]                           MFMA, K = pixels   (accumulates in registers)
//   G[p][c]     = [R > 0] sum_j D[j][p] W[j][c]                   MFMA, K = 32 joints (zero padded)
//   T_r[sx][c]  = sum_ox X[ox][sx] G_r[ox][c]                     MFMA, K = output columns (transposed stencil)
//   dS[y0] += (1-wy) T_r, dS[y1] += wy T_r                        registers: output rows arrive in order and
//        touch source rows y0, y0+1 with y0 non-decreasing in steps of one, so two running sums per
//        owned (sx, c) suffice and each source row is written to dtok exactly once.
// X is the [OW x w] x-interpolation matrix (align_corners=True), kept in LDS in T; operands that
// are contracted along their LDS rows are read with the transposing fragment read (ColFrag).
// One writer per output element everywhere: deterministic, no atomics, no block-level reduction.
constexpr int PB_CC = 32;             // channels per workgroup
constexpr int PB_MAXPX = 96;          // pixels per streamed tile (RB rows of OW)
constexpr int PB_MAXRB = 4;
template <class T> struct PoseLay {   // LDS pitches (elements): +16 bytes per row
    static constexpr int PAD = 16 / (int)sizeof(T);
    static constexpr int PR = PB_CC + PAD;        // rows of channels
    static constexpr int PD = PB_MAXPX + PAD;     // rows of pixels
    static constexpr int PW = 32 + PAD;           // rows of joints / source columns
};
// LDS bytes: T-typed tiles + the f32 interpolation operands + the y tap tables
template <class T>
static size_t pose_bwd_mma_bytes(int h, int w, int rb) {
    const int OW = 4 * w, OWK = (OW + 31) & ~31, OWM = (OW + 15) & ~15, WK4 = (w + 3) & ~3;
    const size_t t_elems = (size_t)h * w * PB_CC + (size_t)OWK * PoseLay<T>::PW + (size_t)PB_MAXPX * PoseLay<T>::PR +
                           (size_t)(PB_MAXPX + 32) * PoseLay<T>::PR + 32 * (size_t)PoseLay<T>::PD + PB_CC * (size_t)PoseLay<T>::PW;
    const size_t f_elems = (size_t)rb * WK4 * (PB_CC + 4) + (size_t)OWM * (WK4 + 4) + 3 * (size_t)(4 * h);
    return t_elems * sizeof(T) + f_elems * 4;
}
template <class T>
__global__ __launch_bounds__(256) void pose_bwd_mma_kernel(const T* __restrict__ tok, const float* __restrict__ W,
                                                           const float* __restrict__ dhm, T* __restrict__ dtok,
                                                           float* __restrict__ slab, int h, int w, int J, int RB_, int skip_) {
#ifdef MTN_ABLATE          // timing ablations (tools/bench_kernels.py, diagnostic builds only): phases can be skipped
    const int skip = skip_;
#else
    constexpr int skip = 0;
    (void)skip_;
#endif
    constexpr int CC = PB_CC;
    constexpr int PR = PoseLay<T>::PR, PD = PoseLay<T>::PD, PW = PoseLay<T>::PW;
    constexpr int KI = Mma<T>::KI;
    constexpr bool TR = sizeof(T) == 2;
    MTN_DYN_SMEM(smem);
    const int hw = h * w, OH = 4 * h, OW = 4 * w, OHW = OH * OW;
    const int RP = RB_ * OW, RPK = (RP + 31) & ~31;     // pixels per tile, padded to whole K steps
    const int OWM = (OW + 15) & ~15, OWK = (OW + 31) & ~31, MW = (w + 15) >> 4, WK4 = (w + 3) & ~3;
    constexpr int PYF = CC + 4;                         // f32 pitches
    const int PXF = WK4 + 1;                            // odd pitch: the 16 rows of an A fragment land in 16 banks
    float* sYf = reinterpret_cast<float*>(smem);        // y-interpolated rows [RB][WK4 sx][PYF] f32 (sx >= w: zero)
    float* sXf = sYf + RB_ * WK4 * PYF;                 // x-interpolation matrix [OWM ox][PXF] f32 (zero padded)
    int* tY0 = reinterpret_cast<int*>(sXf + OWM * PXF); // [OH] y taps
    int* tY1 = tY0 + OH;
    float* tWy = reinterpret_cast<float*>(tY1 + OH);
    T* sS = reinterpret_cast<T*>(tWy + OH);             // source tokens [hw][CC]
    T* sX = sS + hw * CC;                               // x-interpolation matrix in T [OWK ox][PW] (scatter operand)
    T* sR = sX + OWK * PW;                              // relu(u) [MAXPX][PR]
    T* sG = sR + PB_MAXPX * PR;                         // masked G [MAXPX + 32][PR] (tail rows: zero)
    T* sD = sG + (PB_MAXPX + 32) * PR;                  // heat-map gradient [32 j][PD]
    T* sW = sD + 32 * PD;                               // weights [CC][PW] (channel-major, joints contiguous)
    constexpr int NCH = DIM / CC;
    int b, chunk;
    {
        const int L = blockIdx.x, B = gridDim.x / NCH;
        if (B % 8 == 0) { const int xcd = L & 7, j = L >> 3; b = (j / NCH) * 8 + xcd; chunk = j % NCH; }
        else { b = L / NCH; chunk = L % NCH; }
    }
    const int c0 = chunk * CC;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const T* base = tok + ((long)b * (hw + 1) + 1) * DIM + c0;
    for (int i = tid; i < hw * (CC / 2); i += 256) {    // pairs of channels keep the copy 4 / 8 bytes wide
        const int p = i / (CC / 2), cc = (i - p * (CC / 2)) * 2;
        sS[p * CC + cc] = base[(long)p * DIM + cc];
        sS[p * CC + cc + 1] = base[(long)p * DIM + cc + 1];
    }
    for (int i = tid; i < CC * 32; i += 256) {          // W^T chunk, joints zero padded to 32
        const int cc = i >> 5, j = i & 31;
        sW[cc * PW + j] = from_f<T>(j < J ? W[j * DIM + c0 + cc] : 0.f);
    }
    for (int i = tid; i < OWK * 32; i += 256) {         // X[ox][sx]
        const int ox = i >> 5, sx = i & 31;
        float v = 0.f;
        if (ox < OW && sx < w) {
            const Lerp l = lerp_coord(ox, w, OW);
            v = (l.i0 == sx ? 1.f - l.w1 : 0.f) + ((l.i1 == sx && l.i1 != l.i0) ? l.w1 : 0.f);
        }
        sX[ox * PW + sx] = from_f<T>(v);
        if (ox < OWM && sx < WK4) sXf[ox * PXF + sx] = v;
    }
    for (int i = tid; i < RB_ * WK4 * PYF; i += 256) sYf[i] = 0.f;
    for (int i = tid; i < (PB_MAXPX + 32) * PR; i += 256) sG[i] = from_f<T>(0.f);
    for (int i = tid; i < OH; i += 256) {
        const Lerp l = lerp_coord(i, h, OH);
        tY0[i] = l.i0; tY1[i] = l.i1; tWy[i] = l.w1;
    }
    const float* dh = dhm + (long)b * J * OHW;
    const bool do_bias = chunk == 0;
    const int frow = lane & 15, fk = (lane >> 4) * (KI / 4);
    const int jt = wid >> 1, ct = wid & 1;              // this wave's dW fragment
    f32x4_t acc_dw = zero4();
    float db[4] = {0.f, 0.f, 0.f, 0.f};                 // joints (tid >> 5) + 8 q
    // D rows of the NEXT tile travel in registers while this tile is processed (8 joints x 32 pixel
    // lanes per pass: 128-byte runs of each joint's row); the global latency is off the tile's path
    constexpr int DK = PB_MAXPX / 32;
    float dn[4][DK];
    auto fetch_d = [&](int r0n) {
        MTN_UNROLL for (int q = 0; q < 4; ++q) {
            const int j = q * 8 + (tid >> 5);
            MTN_UNROLL for (int k = 0; k < DK; ++k) {
                const int px = (tid & 31) + 32 * k;
                dn[q][k] = (j < J && px < RP && r0n * OW + px < OHW) ? dh[(long)j * OHW + r0n * OW + px] : 0.f;
            }
        }
    };
    fetch_d(0);
    // scatter state (waves 0 and 1: channel fragment nt = wid): running token gradients of source
    // rows cur, cur + 1 for source columns mt*16 + 4*(lane>>4) + r, channel nt*16 + (lane&15)
    f32x4_t g0[2], g1[2];
    g0[0] = g0[1] = g1[0] = g1[1] = zero4();
    int cur = 0;
    T* const drow = dtok + ((long)b * (hw + 1) + 1) * DIM + c0 + wid * 16 + (lane & 15);
    auto flush = [&](const f32x4_t* g, int sy) {
        MTN_UNROLL for (int mt = 0; mt < 2; ++mt)
            MTN_UNROLL for (int r = 0; r < 4; ++r) {
                const int sx = mt * 16 + 4 * (lane >> 4) + r;
                if (mt < MW && sx < w) drow[(long)(sy * w + sx) * DIM] = from_f<T>(g[mt][r]);
            }
    };
    for (int r0 = 0; r0 < OH; r0 += RB_) {
        __syncthreads();                                // setup done / previous tile's MFMA phases are done with sD, sY
        // D tile from the prefetched registers, then start the next tile's fetch
        MTN_UNROLL for (int q = 0; q < 4; ++q) {
            const int j = q * 8 + (tid >> 5);
            MTN_UNROLL for (int k = 0; k < DK; ++k) {
                const int px = (tid & 31) + 32 * k;
                if (px < RPK) sD[j * PD + px] = from_f<T>(dn[q][k]);
                db[q] += dn[q][k];
            }
        }
        if (r0 + RB_ < OH && !(skip & 8)) fetch_d(r0 + RB_);
        // Y rows: thread = channel x (row, source column) pairs
        if (!(skip & 1))
        for (int i = tid >> 5; i < RB_ * w; i += 8) {
            const int r = i / w, sx = i - r * w, oy = r0 + r, c = tid & 31;
            float v = 0.f;
            if (oy < OH) {
                const float wy = tWy[oy];
                v = (1.f - wy) * to_f<T>(sS[(tY0[oy] * w + sx) * CC + c]) + wy * to_f<T>(sS[(tY1[oy] * w + sx) * CC + c]);
            }
            sYf[(r * WK4 + sx) * PYF + c] = v;
        }
        __syncthreads();
        // R = relu(X Y_r): fragments (row r, 16 output columns, 16 channels)
        if (!(skip & 1))
        for (int f = wid; f < RB_ * (OWM / 16) * 2; f += 4) {
            const int nt = f & 1, mt = (f >> 1) % (OWM / 16), r = (f >> 1) / (OWM / 16);
            f32x4_t u = zero4();
            for (int k0 = 0; k0 < WK4; k0 += 4) {
                const float fa = Mma<float>::load(sXf + (mt * 16 + frow) * PXF + k0 + (lane >> 4));
                const float fb = ColFrag<float, false>::load(sYf + (r * WK4 + k0) * PYF + nt * 16, PYF, lane);
                u = Mma<float>::mma(fa, fb, u);
            }
            MTN_UNROLL for (int q = 0; q < 4; ++q) {
                const int ox = mt * 16 + 4 * (lane >> 4) + q;
                if (ox < OW) sR[(r * OW + ox) * PR + nt * 16 + (lane & 15)] = from_f<T>(u[q] > 0.f ? u[q] : 0.f);
            }
        }
        for (int i = RP * CC + tid; i < RPK * CC; i += 256) sR[(i / CC) * PR + (i % CC)] = from_f<T>(0.f);   // K padding rows
        __syncthreads();
        if (!(skip & 2)) {
            // dW[jt][ct] += D[jt rows][pixels] x R[pixels][ct columns]
            for (int k0 = 0; k0 < RPK; k0 += KI) {
                const typename Mma<T>::Frag fa = Mma<T>::load(sD + (jt * 16 + frow) * PD + k0 + fk);
                const typename Mma<T>::Frag fb = ColFrag<T, TR>::load(sR + k0 * PR + ct * 16, PR, lane);
                acc_dw = Mma<T>::mma(fa, fb, acc_dw);
            }
            // G[pixels][channels] = D^T x W, masked by R > 0  (accumulator: row = pixel, column = channel)
            for (int f = wid; f < (RPK / 16) * 2; f += 4) {
                const int mt = f >> 1, nt = f & 1;
                f32x4_t g = zero4();
                MTN_UNROLL for (int k0 = 0; k0 < 32; k0 += KI) {
                    const typename Mma<T>::Frag fa = ColFrag<T, TR>::load(sD + k0 * PD + mt * 16, PD, lane);
                    const typename Mma<T>::Frag fb = Mma<T>::load(sW + (nt * 16 + frow) * PW + k0 + fk);
                    g = Mma<T>::mma(fa, fb, g);
                }
                MTN_UNROLL for (int q = 0; q < 4; ++q) {
                    const int px = mt * 16 + 4 * (lane >> 4) + q, cc = nt * 16 + (lane & 15);
                    sG[px * PR + cc] = from_f<T>(to_f<T>(sR[px * PR + cc]) > 0.f ? g[q] : 0.f);
                }
            }
        }
        __syncthreads();
        // scatter: T_r = X^T G_r on waves 0 / 1, then the y taps into the running sums
        if (wid < 2 && !(skip & 4))
        for (int r = 0; r < RB_ && r0 + r < OH; ++r) {
            const int oy = r0 + r;
            const float wy = tWy[oy];
            const int i0 = tY0[oy], i1 = tY1[oy];
            if (i0 != cur) {                            // source row `cur` is complete
                flush(g0, cur);
                g0[0] = g1[0]; g0[1] = g1[1];
                g1[0] = g1[1] = zero4();
                cur = i0;
            }
            MTN_UNROLL for (int mt = 0; mt < 2; ++mt) {
                if (mt < MW) {
                    f32x4_t t = zero4();
                    for (int k0 = 0; k0 < OWK; k0 += KI) {
                        const typename Mma<T>::Frag fa = ColFrag<T, TR>::load(sX + k0 * PW + mt * 16, PW, lane);
                        const typename Mma<T>::Frag fb = ColFrag<T, TR>::load(sG + (r * OW + k0) * PR + wid * 16, PR, lane);
                        t = Mma<T>::mma(fa, fb, t);
                    }
                    MTN_UNROLL for (int q = 0; q < 4; ++q) {
                        g0[mt][q] += (1.f - wy) * t[q];
                        if (i1 != i0) g1[mt][q] += wy * t[q];
                    }
                }
            }
        }
    }
    if (wid < 2) {                                      // the last one or two source rows
        flush(g0, cur);
        if (cur + 1 < h) flush(g1, cur + 1);
    }
    float* row = slab + (long)b * J * (DIM + 1);
    MTN_UNROLL for (int r = 0; r < 4; ++r) {            // accumulator: row = joint, column = channel
        const int j = jt * 16 + 4 * (lane >> 4) + r;
        if (j < J) row[j * DIM + c0 + ct * 16 + (lane & 15)] = acc_dw[r];
    }
    if (do_bias) {
        MTN_UNROLL for (int q = 0; q < 4; ++q) {
            float v = db[q];
            MTN_UNROLL for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);   // the 32 pixel lanes of a joint
            const int j = q * 8 + (tid >> 5);
            if ((tid & 31) == 0 && j < J) row[J * DIM + j] = v;
        }
    }
}

// Forward on MFMA.  block = (image, RB output rows): for each 32-channel chunk the bilinear map is
// one small f32 matrix product (Y_r rows by VALU straight from the tokens, then R = relu(X Y_r) on
// the exact-f32 MFMA -- the same arithmetic as the backward, so both see the same ReLU mask), and
// the 1x1 head accumulates hm[p][j] += R[p][c] W[j][c] over the eight chunks in registers.
constexpr int PF_TILES = 1;          // row tiles per workgroup of the MFMA forward (4 measured 7 % slower: fewer workgroups in flight)
template <class T>
__global__ __launch_bounds__(256) void pose_fwd_mma_kernel(const T* __restrict__ tok, const float* __restrict__ W,
                                                           const float* __restrict__ bias, float* __restrict__ hm,
                                                           int h, int w, int J, int RB_) {
    constexpr int CC = PB_CC;
    constexpr int PR = PoseLay<T>::PR;
    constexpr int PWC = DIM + 16 / (int)sizeof(T);      // weight rows: all 256 channels
    constexpr int KI = Mma<T>::KI;
    constexpr int PYF = CC + 4;
    MTN_DYN_SMEM(smem);
    const int hw = h * w, OH = 4 * h, OW = 4 * w, OHW = OH * OW;
    const int RP = RB_ * OW, RPK = (RP + 15) & ~15;
    const int OWM = (OW + 15) & ~15, WK4 = (w + 3) & ~3, PXF = WK4 + 1;      // odd pitch: the 16 rows of an A fragment land in 16 banks
    float* sYf = reinterpret_cast<float*>(smem);        // [RB][WK4][PYF]
    float* sXf = sYf + RB_ * WK4 * PYF;                 // [OWM][PXF]
    T* sR = reinterpret_cast<T*>(sXf + OWM * PXF);      // relu(u) of the chunk [MAXPX][PR]
    T* sW = sR + PB_MAXPX * PR;                         // [32 j][PWC]
    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int frow = lane & 15, fk = (lane >> 4) * (KI / 4);
    for (int i = tid; i < 32 * (DIM / 4); i += 256) {   // 16-byte loads: the fill is on every workgroup's critical path
        const int j = i / (DIM / 4), c = (i - j * (DIM / 4)) * 4;
        f32x4_t v = zero4();
        if (j < J) v = *reinterpret_cast<const f32x4_t*>(W + j * DIM + c);
        store4<T>(sW + j * PWC + c, v);
    }
    for (int i = tid; i < OWM * PXF; i += 256) {
        const int ox = i / PXF, sx = i - ox * PXF;
        float v = 0.f;
        if (ox < OW && sx < w) {
            const Lerp l = lerp_coord(ox, w, OW);
            v = (l.i0 == sx ? 1.f - l.w1 : 0.f) + ((l.i1 == sx && l.i1 != l.i0) ? l.w1 : 0.f);
        }
        sXf[i] = v;
    }
    for (int i = tid; i < RB_ * WK4 * PYF; i += 256) sYf[i] = 0.f;
    const T* base = tok + ((long)b * (hw + 1) + 1) * DIM;
    const int nfr = (RPK / 16) * 2;                     // head fragments (pixels x joints), <= 12: 3 per wave
    // a workgroup walks PF_TILES row tiles of its image: the weight / X setup above is paid once
    for (int tile = blockIdx.x * PF_TILES; tile < (int)(blockIdx.x + 1) * PF_TILES && tile * RB_ < OH; ++tile) {
    const int r0 = tile * RB_;
    f32x4_t acc[3];
    acc[0] = acc[1] = acc[2] = zero4();
    // this thread's (row, source column) pairs of the y mix: tap rows and weight do not depend on the chunk
    int yo0[3], yo1[3], ydst[3];
    float ywy[3];
    MTN_UNROLL for (int q = 0; q < 3; ++q) {
        const int i = (tid >> 5) + 8 * q;
        ydst[q] = -1;
        if (i < RB_ * w) {
            const int r = i / w, sx = i - r * w, oy = r0 + r;
            if (oy < OH) {
                const Lerp l = lerp_coord(oy, h, OH);
                yo0[q] = (l.i0 * w + sx) * DIM; yo1[q] = (l.i1 * w + sx) * DIM; ywy[q] = l.w1;
                ydst[q] = (r * WK4 + sx) * PYF;
            }
        }
    }
    // the two token rows of each pair travel one chunk ahead in registers (global latency off the chunk's path)
    T ta[3], tb[3];
    auto fetch_tokens = [&](int c0) {
        MTN_UNROLL for (int q = 0; q < 3; ++q)
            if (ydst[q] >= 0) { ta[q] = base[yo0[q] + c0 + (tid & 31)]; tb[q] = base[yo1[q] + c0 + (tid & 31)]; }
    };
    fetch_tokens(0);
    for (int c0 = 0; c0 < DIM; c0 += CC) {
        __syncthreads();                                // setup done / the previous chunk's MFMAs are done with sYf
        MTN_UNROLL for (int q = 0; q < 3; ++q)
            if (ydst[q] >= 0) sYf[ydst[q] + (tid & 31)] = (1.f - ywy[q]) * to_f<T>(ta[q]) + ywy[q] * to_f<T>(tb[q]);
        if (c0 + CC < DIM) fetch_tokens(c0 + CC);
        __syncthreads();
        for (int f = wid; f < RB_ * (OWM / 16) * 2; f += 4) {       // R = relu(X Y_r), exact f32
            const int nt = f & 1, mt = (f >> 1) % (OWM / 16), r = (f >> 1) / (OWM / 16);
            f32x4_t u = zero4();
            for (int k0 = 0; k0 < WK4; k0 += 4) {
                const float fa = Mma<float>::load(sXf + (mt * 16 + frow) * PXF + k0 + (lane >> 4));
                const float fb = ColFrag<float, false>::load(sYf + (r * WK4 + k0) * PYF + nt * 16, PYF, lane);
                u = Mma<float>::mma(fa, fb, u);
            }
            MTN_UNROLL for (int q = 0; q < 4; ++q) {
                const int ox = mt * 16 + 4 * (lane >> 4) + q;
                if (ox < OW) sR[(r * OW + ox) * PR + nt * 16 + (lane & 15)] = from_f<T>(u[q] > 0.f ? u[q] : 0.f);
            }
        }
        __syncthreads();
        MTN_UNROLL for (int q = 0; q < 3; ++q) {        // hm[pixels][joints] += R[pixels][chunk] W[joints][chunk]
            const int f = wid + 4 * q;
            if (f < nfr) {
                const int mt = f >> 1, nt = f & 1;
                MTN_UNROLL for (int k0 = 0; k0 < CC; k0 += KI) {
                    const typename Mma<T>::Frag fa = Mma<T>::load(sR + (mt * 16 + frow) * PR + k0 + fk);
                    const typename Mma<T>::Frag fb = Mma<T>::load(sW + (nt * 16 + frow) * PWC + c0 + k0 + fk);
                    acc[q] = Mma<T>::mma(fa, fb, acc[q]);
                }
            }
        }
    }
    // accumulator: row = pixel 4*(lane>>4)+r (4 consecutive pixels of one output row), column = joint
    MTN_UNROLL for (int q = 0; q < 3; ++q) {
        const int f = wid + 4 * q;
        if (f < nfr) {
            const int mt = f >> 1, j = (f & 1) * 16 + (lane & 15);
            const int p = mt * 16 + 4 * (lane >> 4);
            const int r = p / OW, ox = p - r * OW, oy = r0 + r;
            if (j < J && p < RP && oy < OH) {
                f32x4_t o = acc[q];
                const float bj = bias[j];
                MTN_UNROLL for (int e = 0; e < 4; ++e) o[e] += bj;
                store4<float>(hm + ((long)(b * J + j) * OH + oy) * OW + ox, o);
            }
        }
    }
    }
}

// ---- row-streaming pose head (round 2) ------------------------------------------------------------
// The kernels above cut an image into 32-channel chunks x 96-pixel tiles and pay 3-4 workgroup
// barriers per (chunk, tile) for a few hundred cycles of work between them; they ran 5-10x above
// the head's memory traffic.  Here a workgroup owns ALL 256 channels of an image (a band of its
// output rows in the forward pass) and walks the output rows:
//   Y[sx][c]  = (1-wy) S[y0][sx][c] + wy S[y1][sx][c]        thread = channel, global -> LDS (f32, own column)
//   R[ox][c]  = relu((1-wx) Y[x0][c] + wx Y[x1][c])          thread = channel, two-register window over x0
//   hm[j][ox] = sum_c R[ox][c] W[j][c] + bias[j]             MFMA, K = 256
// The interpolation stays exact f32 arithmetic on the VALU (the sign of u is the ReLU mask that
// forward and backward must agree on): 4 FLOP per element against the dense x-interpolation matrix
// product (2 w FLOP per element on the 1/16-rate f32 MFMA) of the kernels above.
// The map width is a template parameter: the x loop is fully unrolled, so the y-mixed row lives in
// registers and every x tap index / weight is a literal (a first version read them from LDS tables
// in a rolled loop: three dependent LDS round trips per output element, 10x slower).
// Split mode (T = f32s_t: plain f32 tokens, head products as three bf16 MFMAs): the weight and relu(u) tiles sit in LDS as
// a hi and a lo bf16 PLANE each, split once when they are written (round 5).  The first version kept f32 tiles and split
// every fragment as it was read -- ~28 VALU operations per fragment per K-step, the weights again for every output row:
// VALU-bound at 3x the bf16 kernel's time.  The planes hold the same bits the read-time split produced.  Weight rows beyond
// the J joints share ONE zero row (row J), which brings the workgroup under 80 KB: two per CU instead of one.
template <class T, int WT>
__global__ __launch_bounds__(256) void pose_fwd_row_kernel(const T* __restrict__ tok, const float* __restrict__ W,
                                                           const float* __restrict__ bias, float* __restrict__ hm, int h,
                                                           int J, int rows_per_band) {
    constexpr bool SP = std::is_same<T, f32s_t>::value;
    constexpr int PW = DIM + 16 / (int)sizeof(T);
    constexpr int PB = DIM + 8;                                  // pitch of a bf16 plane (split mode)
    constexpr int KI = Mma<T>::KI;
    constexpr int w = WT, OW = 4 * WT, OWM = (OW + 15) & ~15;
    MTN_DYN_SMEM(smem);
    const int hw = h * w, OH = 4 * h;
    T* sW = reinterpret_cast<T*>(smem);                          // [32][PW]
    T* sR = sW + 32 * PW;                                        // [OWM][PW]
    const int WR = J < 32 ? J + 1 : 32;                          // split mode: weight rows held (row J = zeros)
    bf16_t* pWh = reinterpret_cast<bf16_t*>(smem);               // [WR][PB] hi, lo
    bf16_t* pWl = pWh + WR * PB;
    bf16_t* pRh = pWl + WR * PB;                                 // [OWM][PB] hi, lo
    bf16_t* pRl = pRh + OWM * PB;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int b = blockIdx.y;
    const int r_begin = blockIdx.x * rows_per_band;
    int r_end = r_begin + rows_per_band;
    if (r_end > OH) r_end = OH;
    if constexpr (SP) {
        for (int i = tid; i < WR * (DIM / 4); i += 256) {
            const int j = i / (DIM / 4), c = (i - j * (DIM / 4)) * 4;
            f32x4_t v = zero4();
            if (j < J) v = *reinterpret_cast<const f32x4_t*>(W + j * DIM + c);
            MTN_UNROLL for (int e = 0; e < 4; ++e) {
                const bf16_t hi = f2bf(v[e]);
                pWh[j * PB + c + e] = hi;
                pWl[j * PB + c + e] = f2bf(v[e] - bf2f(hi));
            }
        }
        for (int i = tid; i < (OWM - OW) * DIM; i += 256) {                                                       // M padding rows
            pRh[(OW + i / DIM) * PB + (i % DIM)] = f2bf(0.f);
            pRl[(OW + i / DIM) * PB + (i % DIM)] = f2bf(0.f);
        }
    } else {
        for (int i = tid; i < 32 * (DIM / 4); i += 256) {
            const int j = i / (DIM / 4), c = (i - j * (DIM / 4)) * 4;
            f32x4_t v = zero4();
            if (j < J) v = *reinterpret_cast<const f32x4_t*>(W + j * DIM + c);
            store4<T>(sW + j * PW + c, v);
        }
        for (int i = tid; i < (OWM - OW) * DIM; i += 256) sR[(OW + i / DIM) * PW + (i % DIM)] = from_f<T>(0.f);   // M padding rows
    }
    const T* base = tok + ((long)b * (hw + 1) + 1) * DIM + tid;   // this thread's channel
    const int frow = lane & 15, fk = (lane >> 4) * (KI / 4);
    const int nfr = (OWM / 16) * 2;                               // head fragments (pixels x joint halves)
    // the two token rows of the current y interval stay in registers: four output rows share them
    float t0[WT], t1[WT];
    int cur_y0 = -1;
    for (int oy = r_begin; oy < r_end; ++oy) {
        __syncthreads();                                          // setup / the previous row's MFMAs are done with sR
        const Lerp ly = lerp_coord(oy, h, OH);
        const float wy = ly.w1;
        if (ly.i0 != cur_y0) {                                    // wave-uniform
            const T* s0 = base + (long)ly.i0 * w * DIM;
            const T* s1 = base + (long)ly.i1 * w * DIM;
            MTN_UNROLL for (int sx = 0; sx < WT; ++sx) { t0[sx] = to_f<T>(s0[(long)sx * DIM]); t1[sx] = to_f<T>(s1[(long)sx * DIM]); }
            cur_y0 = ly.i0;
        }
        float ya[WT];
        MTN_UNROLL for (int sx = 0; sx < WT; ++sx) ya[sx] = (1.f - wy) * t0[sx] + wy * t1[sx];
        MTN_UNROLL for (int ox = 0; ox < OW; ++ox) {
            const Lerp lx = lerp_coord(ox, WT, OW);               // literals after unrolling
            const float u = (1.f - lx.w1) * ya[lx.i0] + lx.w1 * ya[lx.i1];
            const float r = u > 0.f ? u : 0.f;
            if constexpr (SP) {
                const bf16_t hi = f2bf(r);
                pRh[ox * PB + tid] = hi;
                pRl[ox * PB + tid] = f2bf(r - bf2f(hi));
            } else {
                sR[ox * PW + tid] = from_f<T>(r);
            }
        }
        __syncthreads();
        for (int f = wid; f < nfr; f += 4) {                      // hm[j][px] : the weight fragment is the row operand
            const int pf = f >> 1, jf = f & 1;
            f32x4_t acc = zero4();
            if constexpr (SP) {
                const int wr = jf * 16 + frow < WR ? jf * 16 + frow : WR - 1;
                const bf16_t* aw = pWh + wr * PB + fk;
                const bf16_t* ar = pRh + (pf * 16 + frow) * PB + fk;
                const int lo_w = WR * PB, lo_r = OWM * PB;        // hi plane -> lo plane
                MTN_UNROLL for (int k0 = 0; k0 < DIM; k0 += KI) {
                    Frag3 fa, fb;
                    fa.hi = load_bf16x8(aw + k0); fa.lo = load_bf16x8(aw + lo_w + k0);
                    fb.hi = load_bf16x8(ar + k0); fb.lo = load_bf16x8(ar + lo_r + k0);
                    acc = Mma<f32x3_t>::mma(fa, fb, acc);
                }
            } else {
                MTN_UNROLL for (int k0 = 0; k0 < DIM; k0 += KI)
                    acc = Mma<T>::mma(Mma<T>::load(sW + (jf * 16 + frow) * PW + k0 + fk), Mma<T>::load(sR + (pf * 16 + frow) * PW + k0 + fk), acc);
            }
            const int ox = pf * 16 + (lane & 15);
            MTN_UNROLL for (int r = 0; r < 4; ++r) {
                const int j = jf * 16 + 4 * (lane >> 4) + r;
                if (j < J && ox < OW) hm[(((long)b * J + j) * OH + oy) * OW + ox] = acc[r] + bias[j];
            }
        }
    }
}

static bool pose_row_width(int w);
// Backward, row-streaming: workgroup = image, thread = channel (256).  Per output row:
//   recompute Y, u, R = relu(u) exactly as the forward (registers), keep the ReLU mask as bits,
//   R -> LDS (T), the row of the heat-map gradient D[j][ox] -> LDS (T, prefetched a row ahead);
//   dW[j][c] += sum_ox D[j][ox] R[ox][c]          MFMA, K = output columns; 8 accumulator fragments per wave
//   G[ox][c]  = sum_j  D[j][ox] W[j][c]           MFMA, K = joints (padded to 32) -> LDS in f32
//   t[sx]    += x-taps of mask * G[ox][c]         VALU, thread = channel: literal tap indices / weights
//   g0 / g1  += (1-wy) t, wy t                    running token-gradient rows y0, y0+1 in registers; a
//        source row is written to dtok exactly once (coalesced: 256 threads = 256 channels), when the
//        output rows move past it.  One writer per element everywhere: deterministic, no atomics.
template <class T, int WT>
__global__ __launch_bounds__(512) void pose_bwd_row_kernel(const T* __restrict__ tok, const float* __restrict__ W,
                                                           const float* __restrict__ dhm, T* __restrict__ dtok,
                                                           float* __restrict__ slab, float* __restrict__ seam, int h, int J,
                                                           int nb) {
    // Bands (round 4): an image is cut into nb bands of source rows [a_k, a_k+1), a_k = h k / nb, one workgroup each
    // (384x384 images at B = 128 left half the chip idle with one workgroup per image).  Band k walks the output rows
    // whose upper source row i0 lies in its range; its last row's lower tap a_k+1 is the first row of band k+1, so
    // exactly ONE source row is shared by neighbours: both write their f32 partial of it to seam[b][k][0 / 1] and
    // pose_seam_kernel adds the two (one writer per dtok element, fixed order: deterministic).  The weight-gradient
    // slab has one row per (image, band).
    // 512 threads = 256 channels x 2 halves of the output columns: thread (c, half) recomputes / scatters
    // the columns [half OW/2, (half+1) OW/2) of its channel (literal tap indices need a compile-time
    // column, hence the two instantiations of the column loops below); both halves keep partial
    // running rows g0 / g1 and combine them through LDS when a source row is flushed (12 times per image).
    constexpr int KI = Mma<T>::KI;
    constexpr bool TR = sizeof(T) == 2;
    constexpr int PAD = 16 / (int)sizeof(T);
    constexpr int w = WT, OW = 4 * WT, OH2 = OW / 2, OWM = (OW + 15) & ~15, OWK = (OW + 31) & ~31;
    constexpr int PR = DIM + PAD;                    // sR rows (channels contiguous)
    constexpr int PD = OWK + PAD;                    // sD rows (output columns contiguous)
    constexpr int PWT = 32 + PAD;                    // sWt rows (joints contiguous)
    // G is staged in f32, except for the 96-column maps of 384x384 images in bf16 mode: there the f32 tile
    // (100 KB) would push the workgroup past 160 KB of LDS (and the kernel back to the 12x slower chunk kernel),
    // so it is staged in T -- one rounding to bf16 of a term of a token gradient that is stored in bf16 anyway
    typedef typename std::conditional<(WT >= 24 && sizeof(T) == 2), T, float>::type GT;
    constexpr int PG = DIM + 16 / (int)sizeof(GT);   // sG rows
    constexpr int NT = 512;
    constexpr int ND = (32 * OW + NT - 1) / NT;      // heat-map gradient elements per thread per row (joints padded to 32)
    constexpr int NPF = OWM / 16;
    static_assert(OH2 <= 64, "ReLU mask bits");
    MTN_DYN_SMEM(smem);
    GT* sG = reinterpret_cast<GT*>(smem);            // [OWM][PG]
    T* sR = reinterpret_cast<T*>(sG + OWM * PG);     // [OWK][PR]  relu(u) (rows >= OW: zero)
    T* sD = sR + OWK * PR;                           // [32][PD]   D (columns >= OW and rows >= J: zero)
    T* sWt = sD + 32 * PD;                           // [DIM][PWT] W^T (joints >= J: zero)
    float* sF = reinterpret_cast<float*>(sR);        // [WT][DIM] f32 flush staging (aliases sR between rows)
    const int hw = h * w, OH = 4 * h, OHW = OH * OW;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int ch = tid & 255, half = tid >> 8;
    const int b = blockIdx.x / nb, band = blockIdx.x - b * nb;
    const int a_lo = (h * band) / nb, a_hi = (h * (band + 1)) / nb;          // this band's upper-tap source rows [a_lo, a_hi)
    int oy_lo = 0, oy_hi = OH;                                              // ... and its output rows (i0 is monotone in oy)
    if (band > 0) { while (lerp_coord(oy_lo, h, OH).i0 < a_lo) ++oy_lo; }
    if (band < nb - 1) { oy_hi = oy_lo; while (lerp_coord(oy_hi, h, OH).i0 < a_hi) ++oy_hi; }
    float* const seam_lo = band > 0 ? seam + (((long)b * (nb - 1) + band - 1) * 2 + 1) * w * DIM + ch : nullptr;
    float* const seam_hi = band < nb - 1 ? seam + (((long)b * (nb - 1) + band) * 2 + 0) * w * DIM + ch : nullptr;
    for (int i = tid; i < DIM * 32; i += NT) {
        const int c = i >> 5, j = i & 31;
        sWt[c * PWT + j] = from_f<T>(j < J ? W[j * DIM + c] : 0.f);
    }
    for (int i = tid; i < (OWK - OW) * DIM; i += NT) sR[(OW + i / DIM) * PR + (i % DIM)] = from_f<T>(0.f);
    for (int i = tid; i < 32 * PD; i += NT) sD[i] = from_f<T>(0.f);
    __syncthreads();                                 // the zero fill is ordered before the first row's writes
    const T* base = tok + ((long)b * (hw + 1) + 1) * DIM + ch;
    T* const drow = dtok + ((long)b * (hw + 1) + 1) * DIM + ch;
    const float* dh = dhm + (long)b * J * OHW;
    const int frow = lane & 15, fk = (lane >> 4) * (KI / 4);
    // this wave's dW fragments: joint half jf = wid >> 2, channel fragments (wid & 3) * 4 .. + 3
    const int jf = wid >> 2, cf0 = (wid & 3) * 4;
    f32x4_t acc_dw[4];
    MTN_UNROLL for (int q = 0; q < 4; ++q) acc_dw[q] = zero4();
    // heat-map gradient: element i = tid + 512 k of the [32][OW] row tile -> (joint i / OW, column i % OW)
    float dn[ND], db[ND];
    MTN_UNROLL for (int k = 0; k < ND; ++k) db[k] = 0.f;
    auto fetch_d = [&](int oy) {
        MTN_UNROLL for (int k = 0; k < ND; ++k) {
            const int i = tid + NT * k, j = i / OW, ox = i - j * OW;
            dn[k] = (j < J && oy < oy_hi) ? dh[(long)j * OHW + oy * OW + ox] : 0.f;
        }
    };
    fetch_d(oy_lo);
    float t0[WT], t1[WT], g0[WT], g1[WT], ya[WT], tx[WT];
    MTN_UNROLL for (int sx = 0; sx < WT; ++sx) { g0[sx] = 0.f; g1[sx] = 0.f; }
    unsigned long long mask = 0ull;                  // bit o: u > 0 at column HALF*OW/2 + o
    // columns of one half: recompute u, R -> LDS, ReLU mask bits
    auto xphase = [&](auto hc) {
        constexpr int HALF = decltype(hc)::value;
        mask = 0ull;
        MTN_UNROLL for (int o = 0; o < OH2; ++o) {
            const int ox = HALF * OH2 + o;
            const Lerp lx = lerp_coord(ox, WT, OW);  // literals after unrolling
            const float u = (1.f - lx.w1) * ya[lx.i0] + lx.w1 * ya[lx.i1];
            mask |= (u > 0.f ? 1ull : 0ull) << o;
            sR[ox * PR + ch] = from_f<T>(u > 0.f ? u : 0.f);
        }
    };
    // transposed x stencil of one half's columns into tx
    auto scatter = [&](auto hc) {
        constexpr int HALF = decltype(hc)::value;
        MTN_UNROLL for (int o = 0; o < OH2; ++o) {
            const int ox = HALF * OH2 + o;
            const Lerp lx = lerp_coord(ox, WT, OW);
            const float g = ((mask >> o) & 1ull) ? to_f<GT>(sG[ox * PG + ch]) : 0.f;
            tx[lx.i0] += (1.f - lx.w1) * g;
            if (lx.i1 != lx.i0) tx[lx.i1] += lx.w1 * g;
        }
    };
    typedef std::integral_constant<int, 0> C0;
    typedef std::integral_constant<int, 1> C1;
    // source row `sy` is complete: the second half hands its partial sums over, the first half writes
    auto flush = [&](const float* g, int sy) {
        __syncthreads();                             // everyone is past its reads of sR (the staging aliases it)
        if (half == 1) { MTN_UNROLL for (int sx = 0; sx < WT; ++sx) sF[sx * DIM + ch] = g[sx]; }
        __syncthreads();
        if (half == 0) {
            float* const sm = (sy == a_lo && seam_lo) ? seam_lo : ((sy == a_hi && seam_hi) ? seam_hi : nullptr);   // workgroup-uniform
            if (sm) { MTN_UNROLL for (int sx = 0; sx < WT; ++sx) sm[sx * DIM] = g[sx] + sF[sx * DIM + ch]; }
            else { MTN_UNROLL for (int sx = 0; sx < WT; ++sx) drow[(long)(sy * w + sx) * DIM] = from_f<T>(g[sx] + sF[sx * DIM + ch]); }
        }
        __syncthreads();                             // staging consumed before the next row's R lands in sR
    };
    int cur_y0 = -1, cur = a_lo;
    for (int oy = oy_lo; oy < oy_hi; ++oy) {
        const Lerp ly = lerp_coord(oy, h, OH);
        const float wy = ly.w1;
        if (ly.i0 != cur_y0) {                       // workgroup-uniform: new y interval -> its two token rows
            const T* s0 = base + (long)ly.i0 * w * DIM;
            const T* s1 = base + (long)ly.i1 * w * DIM;
            MTN_UNROLL for (int sx = 0; sx < WT; ++sx) { t0[sx] = to_f<T>(s0[(long)sx * DIM]); t1[sx] = to_f<T>(s1[(long)sx * DIM]); }
            cur_y0 = ly.i0;
        }
        if (ly.i0 != cur) {                          // source row `cur` is complete (before sR is rewritten)
            flush(g0, cur);
            MTN_UNROLL for (int sx = 0; sx < WT; ++sx) { g0[sx] = g1[sx]; g1[sx] = 0.f; }
            cur = ly.i0;
        }
        MTN_UNROLL for (int sx = 0; sx < WT; ++sx) ya[sx] = (1.f - wy) * t0[sx] + wy * t1[sx];
        if (half == 0) xphase(C0{}); else xphase(C1{});
        MTN_UNROLL for (int k = 0; k < ND; ++k) {    // D row tile from the prefetched registers
            const int i = tid + NT * k, j = i / OW, ox = i - j * OW;
            if (i < 32 * OW) sD[j * PD + ox] = from_f<T>(dn[k]);
            db[k] += dn[k];
        }
        fetch_d(oy + 1);                             // the next row's gradient travels under this row's work
        __syncthreads();                             // sR, sD complete
        // dW[j][c] += D[j][ox] R[ox][c]
        for (int k0 = 0; k0 < OWK; k0 += KI) {
            const typename Mma<T>::Frag fa = Mma<T>::load(sD + (jf * 16 + frow) * PD + k0 + fk);
            typename Mma<T>::Frag fr[4];
            MTN_UNROLL for (int q = 0; q < 4; ++q) fr[q] = ColFrag<T, TR>::load(sR + k0 * PR + (cf0 + q) * 16, PR, lane);
            MTN_UNROLL for (int q = 0; q < 4; ++q) acc_dw[q] = Mma<T>::mma(fa, fr[q], acc_dw[q]);
        }
        // G[ox][c] = D^T W : the W^T fragment is the row operand -> a lane's quad is 4 consecutive channels.
        // Wave w owns channel fragments 2w, 2w+1 for every pixel fragment; all operand fragments of a K
        // step are read before its MFMAs issue.
        {
            f32x4_t g[NPF][2];
            MTN_UNROLL for (int pf = 0; pf < NPF; ++pf) { g[pf][0] = zero4(); g[pf][1] = zero4(); }
            MTN_UNROLL for (int k0 = 0; k0 < 32; k0 += KI) {
                typename Mma<T>::Frag fw[2], fd[NPF];
                MTN_UNROLL for (int q = 0; q < 2; ++q) fw[q] = Mma<T>::load(sWt + ((wid * 2 + q) * 16 + frow) * PWT + k0 + fk);
                MTN_UNROLL for (int pf = 0; pf < NPF; ++pf) fd[pf] = ColFrag<T, TR>::load(sD + k0 * PD + pf * 16, PD, lane);
                MTN_UNROLL for (int pf = 0; pf < NPF; ++pf)
                    MTN_UNROLL for (int q = 0; q < 2; ++q) g[pf][q] = Mma<T>::mma(fw[q], fd[pf], g[pf][q]);
            }
            MTN_UNROLL for (int pf = 0; pf < NPF; ++pf)
                MTN_UNROLL for (int q = 0; q < 2; ++q)
                    store4<GT>(sG + (pf * 16 + (lane & 15)) * PG + (wid * 2 + q) * 16 + 4 * (lane >> 4), g[pf][q]);
        }
        __syncthreads();                             // sG complete; everyone is done with sR / sD
        MTN_UNROLL for (int sx = 0; sx < WT; ++sx) tx[sx] = 0.f;
        if (half == 0) scatter(C0{}); else scatter(C1{});
        MTN_UNROLL for (int sx = 0; sx < WT; ++sx) {
            g0[sx] += (1.f - wy) * tx[sx];
            if (ly.i1 != ly.i0) g1[sx] += wy * tx[sx];
        }
    }
    flush(g0, cur);                                  // the last one or two source rows
    if (cur + 1 < h) flush(g1, cur + 1);
    float* row = slab + (long)blockIdx.x * J * (DIM + 1);
    MTN_UNROLL for (int q = 0; q < 4; ++q)
        MTN_UNROLL for (int r = 0; r < 4; ++r) {     // accumulator: row = joint, column = channel
            const int j = jf * 16 + 4 * (lane >> 4) + r;
            if (j < J) row[j * DIM + (cf0 + q) * 16 + (lane & 15)] = acc_dw[q][r];
        }
    // dbias[j] = sum of this image's gradient row j: element i always belongs to joint i / OW
    __syncthreads();
    float* red = reinterpret_cast<float*>(sG);       // [32 * OW] f32 (12 KB at most: inside the G tile either way)
    MTN_UNROLL for (int k = 0; k < ND; ++k) { const int i = tid + NT * k; if (i < 32 * OW) red[i] = db[k]; }
    __syncthreads();
    if (tid < J) {
        float v = 0.f;
        for (int ox = 0; ox < OW; ++ox) v += red[tid * OW + ox];
        row[J * DIM + tid] = v;
    }
}

// dtok row a_k+1 of image b = the two band partials of it (pose_bwd_row_kernel); thread = 4 channels of one token
template <class T>
__global__ __launch_bounds__(256) void pose_seam_kernel(const float* __restrict__ seam, T* __restrict__ dtok, int B, int h, int w,
                                                        int nb) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;             // over [B][nb-1][w][DIM/4]
    const long n = (long)B * (nb - 1) * w * (DIM / 4);
    if (i >= n) return;
    const int c = (int)(i % (DIM / 4)) * 4;
    const int sx = (int)((i / (DIM / 4)) % w);
    const int k = (int)((i / ((long)(DIM / 4) * w)) % (nb - 1));
    const int b = (int)(i / ((long)(DIM / 4) * w * (nb - 1)));
    const float* p = seam + (((long)b * (nb - 1) + k) * 2) * w * DIM + (long)sx * DIM + c;
    const f32x4_t u = *reinterpret_cast<const f32x4_t*>(p), v = *reinterpret_cast<const f32x4_t*>(p + (long)w * DIM);
    const int sy = (h * (k + 1)) / nb;
    f32x4_t o;
    MTN_UNROLL for (int e = 0; e < 4; ++e) o[e] = u[e] + v[e];
    store4<T>(dtok + ((long)b * (h * w + 1) + 1 + sy * w + sx) * DIM + c, o);
}

// bands per image of the row-streaming backward: fill the chip when the batch alone does not (>= 3 source rows per band)
static int pose_bwd_bands(int B, int h) {
    int nb = 1;
    while (nb < 4 && B * nb * 2 <= 256 && h / (nb * 2) >= 3) nb *= 2;
    return nb;
}

template <class T>
static size_t pose_bwd_row_lds(int w) {
    const int OW = 4 * w, OWM = (OW + 15) & ~15, OWK = (OW + 31) & ~31, pad = 16 / (int)sizeof(T);
    const size_t gsz = (w >= 24 && sizeof(T) == 2) ? sizeof(T) : 4;       // element size of the staged G tile (GT in the kernel)
    return (size_t)OWM * (DIM + 16 / gsz) * gsz + ((size_t)OWK * (DIM + pad) + 32 * (size_t)(OWK + pad) + (size_t)DIM * (32 + pad)) * sizeof(T);
}
template <class T, int WT>
static void pose_bwd_row_launch(int B, size_t lds, hipStream_t stream, const void* tok, const float* W, const float* dhm, void* dtok,
                                float* slab, float* seam, int h, int J, int nb) {
#ifndef MTN_EMU
    static MtnLdsAttr attr;
    mtn_set_max_lds(attr, &pose_bwd_row_kernel<T, WT>, 160 * 1024);
#endif
    MTN_LAUNCH((pose_bwd_row_kernel<T, WT>), dim3(B * nb), dim3(512), lds, stream, (const T*)tok, W, dhm, (T*)dtok, slab, seam, h, J, nb);
}
// S = storage type of the token gradient (f32s_t computes on plain f32 rows)
template <class T, class S>
static void pose_bwd_row_dispatch(int w, int B, size_t lds, hipStream_t stream, const void* tok, const float* W, const float* dhm,
                                  void* dtok, float* slab, float* seam, int h, int J, int nb) {
    if (w == 6) pose_bwd_row_launch<T, 6>(B, lds, stream, tok, W, dhm, dtok, slab, seam, h, J, nb);
    else if (w == 12) pose_bwd_row_launch<T, 12>(B, lds, stream, tok, W, dhm, dtok, slab, seam, h, J, nb);
    else pose_bwd_row_launch<T, 24>(B, lds, stream, tok, W, dhm, dtok, slab, seam, h, J, nb);
    if (nb > 1) {
        const long n = (long)B * (nb - 1) * w * (DIM / 4);
        MTN_LAUNCH((pose_seam_kernel<S>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, seam, (S*)dtok, B, h, w, nb);
    }
}

template <class T>
static size_t pose_fwd_row_lds(int w, int J) {
    const int OW = 4 * w, OWM = (OW + 15) & ~15;
    if (std::is_same<T, f32s_t>::value) return (size_t)((J < 32 ? J + 1 : 32) + OWM) * 2 * (DIM + 8) * sizeof(bf16_t);   // hi + lo planes
    return (size_t)(32 + OWM) * (DIM * sizeof(T) + 16);
}
// instantiated map widths: 12 (192^2 images), 24 (384^2), 6 (96^2; also the CPU kernel-logic tests)
static bool pose_row_width(int w) { return w == 6 || w == 12 || w == 24; }
template <class T, int WT>
static void pose_fwd_row_launch(dim3 grid, size_t lds, hipStream_t stream, const void* tok, const float* W, const float* bias,
                                float* hm, int h, int J, int rpb) {
#ifndef MTN_EMU
    static MtnLdsAttr attr;
    mtn_set_max_lds(attr, &pose_fwd_row_kernel<T, WT>, 160 * 1024);
#endif
    MTN_LAUNCH((pose_fwd_row_kernel<T, WT>), grid, dim3(256), lds, stream, (const T*)tok, W, bias, hm, h, J, rpb);
}
template <class T>
static void pose_fwd_row_dispatch(int w, dim3 grid, size_t lds, hipStream_t stream, const void* tok, const float* W,
                                  const float* bias, float* hm, int h, int J, int rpb) {
    if (w == 6) pose_fwd_row_launch<T, 6>(grid, lds, stream, tok, W, bias, hm, h, J, rpb);
    else if (w == 12) pose_fwd_row_launch<T, 12>(grid, lds, stream, tok, W, bias, hm, h, J, rpb);
    else pose_fwd_row_launch<T, 24>(grid, lds, stream, tok, W, bias, hm, h, J, rpb);
}

// ---- loss ------------------------------------------------------------------------------------------
// dhm = w^2 (p - t) / n ; partial[block] = sum 0.5 (w (p - t))^2 / n,  n = B*J*HW
__global__ __launch_bounds__(256) void jmse_kernel(const float* __restrict__ hm, const float* __restrict__ tgt,
                                                   const float* __restrict__ tw, float* __restrict__ dhm,
                                                   float* __restrict__ partial, long n, int HW) {
    __shared__ float red[256];
    const float inv = 1.0f / (float)n;
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float wgt = tw[i / HW];
        const float d = wgt * (hm[i] - tgt[i]);
        s += 0.5f * d * d;
        if (dhm) dhm[i] = wgt * d * inv;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] * inv;
}

// single block: cross entropy over the batch + final sum of the heat-map partials
// losses = (total, 0.001*CE, JMSE); dlogits = ce_weight/B (softmax - onehot)
__global__ __launch_bounds__(256) void ce_finish_kernel(const float* __restrict__ logits, const long long* __restrict__ label,
                                                        float* __restrict__ dlogits, const float* __restrict__ partial,
                                                        int nparts, float* __restrict__ losses, int B, int C,
                                                        float ce_weight) {
    // A label outside [0, C) is IGNORED, the way torch's CrossEntropyLoss treats ignore_index
    // (libs/loss.py:36 keeps the default -100): no loss, zero gradient, and the mean runs over the
    // valid samples only (0/0 = NaN if there is none, as in torch).  Nothing is ever read or
    // written at an index derived from such a label.
    __shared__ float red[2][256];
    __shared__ int s_cnt[256];
    int cnt = 0;
    for (int b = threadIdx.x; b < B; b += 256) {
        const long long y = label[b];
        cnt += (y >= 0 && y < (long long)C) ? 1 : 0;
    }
    s_cnt[threadIdx.x] = cnt;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) s_cnt[threadIdx.x] += s_cnt[threadIdx.x + st];
        __syncthreads();
    }
    const float valid = (float)s_cnt[0];
    float ce = 0.f, jm = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
        const float* l = logits + (long)b * C;
        const long long yl = label[b];
        const bool ok = yl >= 0 && yl < (long long)C;
        const int y = ok ? (int)yl : -1;
        float mx = l[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, l[c]);
        float sum = 0.f;
        for (int c = 0; c < C; ++c) sum += __expf(l[c] - mx);
        if (ok) ce += mx + __logf(sum) - l[y];
        if (dlogits) {
            const float k = ok ? ce_weight / valid : 0.f, inv = 1.0f / sum;
            for (int c = 0; c < C; ++c) dlogits[(long)b * C + c] = k * (__expf(l[c] - mx) * inv - (c == y ? 1.f : 0.f));
        }
    }
    for (int i = threadIdx.x; i < nparts; i += 256) jm += partial[i];
    red[0][threadIdx.x] = ce;
    red[1][threadIdx.x] = jm;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) { red[0][threadIdx.x] += red[0][threadIdx.x + st]; red[1][threadIdx.x] += red[1][threadIdx.x + st]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float cl = ce_weight * red[0][0] / valid;
        losses[1] = cl;
        losses[2] = red[1][0];
        losses[0] = cl + red[1][0];
    }
}

}  // namespace

static int g_pose_mma = 1;      // debug key 2: 1 selects the VALU kernels
static int g_pose_row = 1;      // debug key 14: 0 selects the round-1 chunk/tile kernels
void mtn_pose_set_row(int on) { g_pose_row = on; }
static int g_pose_bands = 0;    // debug key 25: bands per image of the row-streaming backward (0 = by batch size)
void mtn_pose_set_bands(int n) { g_pose_bands = n; }
static int pose_bwd_rb(int h, int w);

template <class T>
static size_t pose_fwd_mma_lds(int h, int w, int rb) {
    const int OW = 4 * w, OWM = (OW + 15) & ~15, WK4 = (w + 3) & ~3;
    return ((size_t)rb * WK4 * (PB_CC + 4) + (size_t)OWM * (WK4 + 4)) * 4 +
           ((size_t)PB_MAXPX * PoseLay<T>::PR + 32 * (size_t)(DIM + 16 / sizeof(T))) * sizeof(T);
}

int mtn_pose_fwd_run(const void* tok, const float* W, const float* bias, float* hm, int dtype, int B, int h, int w, int J,
                     hipStream_t stream) {
    MTN_REQUIRE(B > 0 && h > 0 && w > 0 && J > 0 && J <= MAXJ, "pose_fwd: bad sizes (J=%d, max %d)", J, MAXJ);
    const int OH = 4 * h, OW = 4 * w;
    {
        const bool x3 = dtype == MTN_F32X3;      // plain f32 tokens, head products as split-bf16 MFMAs (f32s_t, mtn_types.h)
        if (x3) dtype = MTN_F32;
        const size_t lds = dtype == MTN_BF16 ? pose_fwd_row_lds<bf16_t>(w, J) : (x3 ? pose_fwd_row_lds<f32s_t>(w, J) : pose_fwd_row_lds<float>(w, J));
        if (g_pose_row && pose_row_width(w) && lds <= 160 * 1024) {   // row-streaming kernel: all 256 channels per workgroup
            // >= 768 workgroups (three per CU at B = 256: 42 KB of LDS each); split mode: 512 (two per CU, 74 KB each)
            int bands = mtn_cdiv(x3 ? 512 : 768, B);
            if (bands > OH) bands = OH;
            const int rpb = mtn_cdiv(OH, bands);
            const dim3 grid(mtn_cdiv(OH, rpb), B);
            if (dtype == MTN_BF16) pose_fwd_row_dispatch<bf16_t>(w, grid, lds, stream, tok, W, bias, hm, h, J, rpb);
            else if (x3) pose_fwd_row_dispatch<f32s_t>(w, grid, lds, stream, tok, W, bias, hm, h, J, rpb);
            else pose_fwd_row_dispatch<float>(w, grid, lds, stream, tok, W, bias, hm, h, J, rpb);
            return mtn_check_launch("pose_fwd_row");
        }
    }
    if (g_pose_mma && 4 * w <= PB_MAXPX && w <= 32) {    // MFMA path: rows of the map as matrix products
        const int rb = pose_bwd_rb(h, w);
        const dim3 grid(mtn_cdiv(mtn_cdiv(OH, rb), PF_TILES), B);
#ifndef MTN_EMU
        static MtnLdsAttr attr_b, attr_f;
        mtn_set_max_lds(attr_b, &pose_fwd_mma_kernel<bf16_t>, 160 * 1024);
        mtn_set_max_lds(attr_f, &pose_fwd_mma_kernel<float>, 160 * 1024);
#endif
        if (dtype == MTN_BF16)
            MTN_LAUNCH(pose_fwd_mma_kernel<bf16_t>, grid, dim3(256), pose_fwd_mma_lds<bf16_t>(h, w, rb), stream, (const bf16_t*)tok, W, bias, hm, h, w, J, rb);
        else
            MTN_LAUNCH(pose_fwd_mma_kernel<float>, grid, dim3(256), pose_fwd_mma_lds<float>(h, w, rb), stream, (const float*)tok, W, bias, hm, h, w, J, rb);
        return mtn_check_launch("pose_fwd");
    }
    const dim3 grid(mtn_cdiv(OH, TP) * mtn_cdiv(OW, TP), B);
    const size_t esz = dtype == MTN_BF16 ? 2 : 4;
    const size_t lds = (size_t)(TP * TP + 32) * (DIM * esz + 16);
#ifndef MTN_EMU
    static MtnLdsAttr attr_b, attr_f;
    mtn_set_max_lds(attr_b, &pose_fwd_kernel<bf16_t>, 160 * 1024);
    mtn_set_max_lds(attr_f, &pose_fwd_kernel<float>, 160 * 1024);
#endif
    if (dtype == MTN_BF16)
        MTN_LAUNCH(pose_fwd_kernel<bf16_t>, grid, dim3(256), lds, stream, (const bf16_t*)tok, W, bias, hm, h, w, J);
    else
        MTN_LAUNCH(pose_fwd_kernel<float>, grid, dim3(256), lds, stream, (const float*)tok, W, bias, hm, h, w, J);
    return mtn_check_launch("pose_fwd");
}

template <class T, int CC, int JT>
static int pose_bwd_launch(const void* tok, const float* W, const float* dhm, void* dtok, float* slab, int B, int h, int w,
                           int J, hipStream_t stream) {
    const int OH = 4 * h, OW = 4 * w, RP = RB * OW, PL = 256 / CC;
    const size_t tail = (size_t)(J * RP + RP * CC > PL * J * CC ? J * RP + RP * CC : PL * J * CC);
    const size_t lds = ((size_t)2 * h * w * CC + tail + 3 * (size_t)(OH + OW) + (2 + MAXTAP) * (size_t)w) * 4;
    MTN_REQUIRE(lds <= 160 * 1024, "pose_bwd: feature map %dx%d too large for LDS (%zu bytes)", h, w, lds);
#ifndef MTN_EMU
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pose_bwd_kernel<T, CC, JT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#endif
    MTN_LAUNCH((pose_bwd_kernel<T, CC, JT>), dim3((DIM / CC) * B), dim3(256), lds, stream, (const T*)tok, W, dhm, (T*)dtok, slab, h, w, J);
    return mtn_check_launch("pose_bwd");
}

// (g_pose_mma is declared above the forward launcher)
static int g_pose_skip = 0;     // 2 * bits: timing ablations of the MFMA kernel in -DMTN_ABLATE builds (1 interpolation, 2 MFMA, 4 scatter, 8 prefetch)
void mtn_pose_set_dbg(int v) { g_pose_mma = v == 1 ? 0 : 1; g_pose_skip = v > 1 ? v >> 1 : 0; }

// rows of the heat map per streamed tile
static int pose_bwd_rb(int h, int w) {
    int rb = PB_MAXPX / (4 * w);
    if (rb > 4 * h) rb = 4 * h;
    return rb > PB_MAXRB ? PB_MAXRB : rb;
}

// LDS bytes of the MFMA backward for an h x w token map, or 0 when it does not apply
template <class T>
static size_t pose_bwd_mma_lds(int h, int w) {
    if (4 * w > PB_MAXPX || w > 32) return 0;
    const size_t bytes = pose_bwd_mma_bytes<T>(h, w, pose_bwd_rb(h, w));
    return bytes <= 160 * 1024 ? bytes : 0;
}

template <class T>
static int pose_bwd_mma_launch(const void* tok, const float* W, const float* dhm, void* dtok, float* slab, int B, int h, int w,
                               int J, size_t lds, hipStream_t stream) {
#ifndef MTN_EMU
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pose_bwd_mma_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#endif
    MTN_LAUNCH((pose_bwd_mma_kernel<T>), dim3((DIM / PB_CC) * B), dim3(256), lds, stream, (const T*)tok, W, dhm, (T*)dtok, slab, h, w, J,
               pose_bwd_rb(h, w), g_pose_skip);
    return mtn_check_launch("pose_bwd");
}

int mtn_pose_bwd_run(const void* tok, const float* W, const float* dhm, void* dtok, float* dW_dbias, float* slab,
                     int dtype, int B, int h, int w, int J, hipStream_t stream) {
    MTN_REQUIRE(B > 0 && h > 0 && w > 0 && J > 0 && J <= MAXJ, "pose_bwd: bad sizes (J=%d, max %d)", J, MAXJ);
    int rc;
    const bool x3 = dtype == MTN_F32X3;          // plain f32 tokens, products as split-bf16 MFMAs (f32s_t, mtn_types.h)
    if (x3) dtype = MTN_F32;
    const size_t mma_lds = !g_pose_mma ? 0 : (dtype == MTN_BF16 ? pose_bwd_mma_lds<bf16_t>(h, w) : pose_bwd_mma_lds<float>(h, w));
    const size_t row_lds = dtype == MTN_BF16 ? pose_bwd_row_lds<bf16_t>(w) : pose_bwd_row_lds<float>(w);
    int slab_rows = B;
    if (g_pose_row && pose_row_width(w) && row_lds <= 160 * 1024 && J <= 32) {
        const int nb = g_pose_bands > 0 ? (g_pose_bands <= 4 && h / g_pose_bands >= 3 ? g_pose_bands : 1) : pose_bwd_bands(B, h);
        slab_rows = B * nb;
        float* seam = slab + (size_t)mtn_reduce_rows_needed(slab_rows) * J * (DIM + 1);       // after the slab and its folding rows
        if (dtype == MTN_BF16) pose_bwd_row_dispatch<bf16_t, bf16_t>(w, B, row_lds, stream, tok, W, dhm, dtok, slab, seam, h, J, nb);
        else if (x3) pose_bwd_row_dispatch<f32s_t, float>(w, B, row_lds, stream, tok, W, dhm, dtok, slab, seam, h, J, nb);
        else pose_bwd_row_dispatch<float, float>(w, B, row_lds, stream, tok, W, dhm, dtok, slab, seam, h, J, nb);
        rc = mtn_check_launch("pose_bwd_row");
    } else if (mma_lds) {
        if (dtype == MTN_BF16) rc = pose_bwd_mma_launch<bf16_t>(tok, W, dhm, dtok, slab, B, h, w, J, mma_lds, stream);
        else rc = pose_bwd_mma_launch<float>(tok, W, dhm, dtok, slab, B, h, w, J, mma_lds, stream);
    } else if (J == 21) {      // hagrid.yaml: num_joints = 21 -> fully unrolled joint loops
        // VALU kernel, 16-channel chunks: ~48 KB of LDS at 12x12 tokens -> 3 workgroups per CU
        if (dtype == MTN_BF16) rc = pose_bwd_launch<bf16_t, 16, 21>(tok, W, dhm, dtok, slab, B, h, w, J, stream);
        else rc = pose_bwd_launch<float, 16, 21>(tok, W, dhm, dtok, slab, B, h, w, J, stream);
    } else {
        if (dtype == MTN_BF16) rc = pose_bwd_launch<bf16_t, 16, 0>(tok, W, dhm, dtok, slab, B, h, w, J, stream);
        else rc = pose_bwd_launch<float, 16, 0>(tok, W, dhm, dtok, slab, B, h, w, J, stream);
    }
    if (rc) return rc;
    return mtn_colsum_run(slab, slab_rows, J * (DIM + 1), dW_dbias, stream);
}

// f32 elements of the scratch mtn_pose_bwd_run needs: [rows + folding rows][J*257] of weight-gradient partials (one row per
// image and band) followed by the seam rows [B][bands-1][2][w][256]; sized for the most bands any batch of this size takes
size_t mtn_pose_slab_floats_run(int B, int h, int w, int J) {
    const int nb = 4;
    return (size_t)mtn_reduce_rows_needed(B * nb) * J * (DIM + 1) + (size_t)B * (nb - 1) * 2 * w * DIM;
}

int mtn_loss_blocks(long n) {
    long b = (n + 1023) / 1024;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

int mtn_loss_run(const float* logits, const long long* label, const float* hm, const float* tgt, const float* tw,
                 float* dlogits, float* dhm, float* partial, float* losses, int B, int C, int J, int HW, float ce_weight,
                 hipStream_t stream) {
    MTN_REQUIRE(B > 0 && C > 0 && J > 0 && HW > 0, "loss: bad sizes");
    MTN_REQUIRE(logits && label && hm && tgt && tw && partial && losses, "loss: null operand");
    MTN_REQUIRE((uintptr_t)label % 8 == 0 && ((uintptr_t)logits | (uintptr_t)hm | (uintptr_t)tgt | (uintptr_t)tw) % 4 == 0,
                "loss: label must be int64-aligned, the float operands 4-byte aligned");
    const long n = (long)B * J * HW;
    const int blocks = mtn_loss_blocks(n);
    MTN_LAUNCH(jmse_kernel, dim3(blocks), dim3(256), 0, stream, hm, tgt, tw, dhm, partial, n, HW);
    MTN_LAUNCH(ce_finish_kernel, dim3(1), dim3(256), 0, stream, logits, label, dlogits, partial, blocks, losses, B, C, ce_weight);
    return mtn_check_launch("loss");
}
