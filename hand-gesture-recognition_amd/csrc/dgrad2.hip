// dgrad2.hip -- data gradient of the 3x3 stride-2 pad-1 convolutions (conv2, down1, down2, down3:
// model/gelan.py:156-164; autograd of nn.Conv2d), all four output parities from one workgroup.
//
//   dx[b][2a+qy][2c+qx][n] = sum over the taps (kh, kw) with kh = 1 (qy = 0) or kh in {0, 2} (qy = 1),
//                            likewise kw / qx, of  dy[b][a + (kh == 0)][c + (kw == 0)][:] . W[:][n][kh][kw]
//
// A row of the tile is one dy pixel (b, a, c) = one 2x2 block of dx.  The nine taps of that block read
// only FOUR dy pixels -- (a, c), (a, c+1), (a+1, c), (a+1, c+1) -- which in the flattened (b, a, c)
// order are rows m, m+1, m+Wd, m+Wd+1: the BM rows of a tile need the CONTIGUOUS dy rows
// [m0, m0 + BM + Wd + 1).  That window is DMA-ed into LDS once per 128-byte x 2 channel slab and every
// tap reads its operand fragments from it at a row offset (a zero row where c+1 / a+1 leave the image);
// each tap accumulates into the accumulator set of the parity class it produces, so the dy operand
// crosses L2 -> LDS once instead of once per tap and class (igemm.hip's parity-major rows: 2.25 taps
// x 4 classes, each class its own short-K workgroup with its own prologue, epilogue and scattered
// 128-byte stores).  Only the 64 x 128-byte weight tiles stream per tap.  The epilogue writes whole
// 2x2 blocks: both x parities of an output row are adjacent in memory.
//
// 8 waves (4 row groups x 2 column groups; 32 x 32 per wave and class), 2 workgroups per CU.
#include <type_traits>
#include "mtn_types.h"
#include "mtn_kernels.h"

namespace {

#ifdef MTN_EMU
#define MTN_D2_WAVES(n)
#else
#define MTN_D2_WAVES(n) __attribute__((amdgpu_waves_per_eu(n, n)))
#endif

constexpr int D2_BM = 128, D2_BN = 64, D2_NT = 512, D2_CS = 2;

template <class T>
__global__ __launch_bounds__(D2_NT) MTN_D2_WAVES(sizeof(T) == 2 ? 4 : 2)
void dgrad2_kernel(IgemmArgs p) {
    constexpr int BM = D2_BM, BN = D2_BN, NT = D2_NT, CS = D2_CS;
    constexpr int BK = 128 / (int)sizeof(T);            // elements per 128-byte LDS row
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr int KI = Mma<T>::KI, KS = BK / KI, NS = CS * KS;
    constexpr int FM = 2, FN = 2;
    constexpr int NQ = sizeof(T) == 2 ? 2 : 1;          // output-row parities staged per epilogue round
    typedef typename StageT<T>::type CT;                // staged element (split storage stages f32 and splits on the way out)
    constexpr int EV = Ev<T>::N;                        // channels per thread and access in the epilogue
    constexpr int CPR = NQ * 2 * BN / EV;               // channel groups of one staged row ([qy][qx][n])
    constexpr int CP = NQ * 2 * BN + 16 / (int)sizeof(T);   // its pitch in elements
    static_assert(NT % CPR == 0, "store mapping");

    MTN_DYN_SMEM(smem);
    const int Wd = p.W, Hd = p.H, Cd = p.Cin;
    const int Md = p.B * Hd * Wd;
    const int WR8 = (BM + Wd + 1 + 7) & ~7;
    T* const sW = reinterpret_cast<T*>(smem);           // [2 stages][CS][BN][BK]
    T* const sWin = sW + 2 * CS * BN * BK;              // [CS][WR8][BK]
    const int zero_off = CS * WR8 * BK;                 // (relative to sWin) one zero row
    CT* const sC = reinterpret_cast<CT*>(smem);         // [BM][CP] staged classes (after the K loop)

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid & 3, wn = wid >> 2;
    const int frow = lane & 15, fg = lane >> 4;

    // XCD-aware tile order as in igemm.hip: the N/BN column tiles of one window share an XCD's L2
    const int tiles_n = p.N / BN, nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    auto decode = [&](int m, int& b, int& a, int& c) {
        const unsigned yt = Wd == 1 ? (unsigned)m : __umulhi((unsigned)m, p.magic_w);     // m / Wd
        c = m - (int)yt * Wd;
        b = Hd == 1 ? (int)yt : (int)__umulhi(yt, p.magic_h);                             // yt / Hd
        a = (int)yt - b * Hd;
    };
    // per fragment row of this lane: may the row read its right / lower neighbour?
    bool f_vx[FM], f_vy[FM];
    MTN_UNROLL for (int i = 0; i < FM; ++i) {
        const int m = m0 + wm * 32 + i * 16 + frow;
        int b, a, c;
        decode(m < Md ? m : 0, b, a, c);
        f_vx[i] = m < Md && c + 1 < Wd;
        f_vy[i] = m < Md && a + 1 < Hd;
    }
    if (tid < 128 / 4) reinterpret_cast<unsigned*>(sWin + zero_off)[tid] = 0u;

    const BufDesc bufA = make_buf(p.A, p.a_bytes), bufW = make_buf(p.Wp, p.w_bytes);
    const int dchunk = ((lane & 7) ^ (lane >> 3)) * VEC;          // source chunk of this lane (XOR swizzle by row & 7)
    auto dma_window = [&](int slab) {
        const int ppc = WR8 >> 3;                                  // 1 KiB pieces per 128-byte channel slab
        for (int q = wid; q < CS * ppc; q += NT / 64) {
            const int cs = q / ppc, pr = q - cs * ppc;
            const int g = m0 + pr * 8 + (lane >> 3);
            const unsigned e = (unsigned)(g * p.lda + (slab * CS + cs) * BK + dchunk) * (unsigned)sizeof(T);
            buf_glds16(bufA, g < Md ? e : 0xFFFFFFF0u, reinterpret_cast<char*>(sWin + (cs * WR8 + pr * 8) * BK));
        }
    };
    auto dma_w = [&](int slab, int tap, int stage) {
        MTN_UNROLL for (int q = wid; q < CS * (BN / 8); q += NT / 64) {
            const int cs = q / (BN / 8), pr = q - cs * (BN / 8);
            const int n = n0 + pr * 8 + (lane >> 3);
            const unsigned e = (unsigned)(n * p.K + tap * Cd + (slab * CS + cs) * BK + dchunk) * (unsigned)sizeof(T);
            buf_glds16(bufW, e, reinterpret_cast<char*>(sW + ((stage * CS + cs) * BN + pr * 8) * BK));
        }
    };

    f32x4_t acc[4][FM][FN];
    MTN_UNROLL for (int q = 0; q < 4; ++q)
        MTN_UNROLL for (int i = 0; i < FM; ++i)
            MTN_UNROLL for (int j = 0; j < FN; ++j) acc[q][i][j] = zero4();

    // one tap: class (kh != 1, kw != 1) += W[kh][kw] x window rows shifted by (kh == 0, kw == 0)
    auto compute = [&](auto TAP, int stage) {
        constexpr int t = decltype(TAP)::value, kh = t / 3, kw = t % 3;
        constexpr int cls = (kh != 1 ? 2 : 0) + (kw != 1 ? 1 : 0);
        constexpr bool da = kh == 0, dc = kw == 0;
        const T* wb = sW + (stage * CS * BN + wn * 32 + frow) * BK;
        const int swb = frow & 7;
        int aoff[FM], acs[FM], asw[FM];
        MTN_UNROLL for (int i = 0; i < FM; ++i) {
            const int j = wm * 32 + i * 16 + frow + (da ? Wd : 0) + (dc ? 1 : 0);
            const bool ok = (!da || f_vy[i]) && (!dc || f_vx[i]);
            aoff[i] = ok ? j * BK : zero_off;
            acs[i] = ok ? WR8 * BK : 0;
            asw[i] = j & 7;
        }
        typename Mma<T>::Frag fa[2][FM], fb[2][FN];
        auto load = [&](int s, int buf) {
            const int cs = s / KS, ks = s % KS;
            MTN_UNROLL for (int j = 0; j < FN; ++j)
                fb[buf][j] = RowFragW<T>::load(wb + (cs * BN + j * 16) * BK, ks, fg, swb);
            MTN_UNROLL for (int i = 0; i < FM; ++i)
                fa[buf][i] = RowFrag<T>::load(sWin + aoff[i] + cs * acs[i], ks, fg, asw[i]);
        };
        load(0, 0);
        MTN_UNROLL for (int s = 0; s < NS; ++s) {
            if (s + 1 < NS) load(s + 1, (s + 1) & 1);
            MTN_UNROLL for (int i = 0; i < FM; ++i)
                MTN_UNROLL for (int j = 0; j < FN; ++j)
                    acc[cls][i][j] = Mma<T>::mma(fb[s & 1][j], fa[s & 1][i], acc[cls][i][j]);
        }
    };
    // tap t of a slab: the next tap's weight tile streams in while this one is multiplied
    auto tap_step = [&](auto TAP, int slab) {
        constexpr int t = decltype(TAP)::value;
        if (t + 1 < 9) dma_w(slab, t + 1, (t + 1) & 1);
        compute(TAP, t & 1);
        __syncthreads();
    };

    const int slabs = Cd / (CS * BK);
    for (int slab = 0; slab < slabs; ++slab) {
        if (slab) __syncthreads();
        dma_window(slab);
        dma_w(slab, 0, 0);
        __syncthreads();
        tap_step(std::integral_constant<int, 0>{}, slab);
        tap_step(std::integral_constant<int, 1>{}, slab);
        tap_step(std::integral_constant<int, 2>{}, slab);
        tap_step(std::integral_constant<int, 3>{}, slab);
        tap_step(std::integral_constant<int, 4>{}, slab);
        tap_step(std::integral_constant<int, 5>{}, slab);
        tap_step(std::integral_constant<int, 6>{}, slab);
        tap_step(std::integral_constant<int, 7>{}, slab);
        tap_step(std::integral_constant<int, 8>{}, slab);
    }

    // ---- epilogue: classes staged as [row][qy][qx][n] -> 16-byte stores (both x parities of an output row are adjacent) ----
    // bf16: all four classes in one round; fp32: one output-row parity per round (LDS).  A thread keeps ONE
    // (qy, qx, channel chunk) for all its rows.
    T* __restrict__ gO = reinterpret_cast<T*>(p.Out);
    const T* __restrict__ gR = reinterpret_cast<const T*>(p.res);
    const int ccol = (tid % CPR) * EV, trow = tid / CPR;
    const int qyl = ccol / (2 * BN), qx = (ccol / BN) & 1, col = ccol % BN;
    constexpr int RPT = BM / (NT / CPR);
    // BatchNorm backward sums of the unit whose output gradient dx is (IgemmArgs::bn_y), as in igemm.hip
    const T* __restrict__ gBy = reinterpret_cast<const T*>(p.bn_y);
    const T* __restrict__ gBr = reinterpret_cast<const T*>(p.bn_res);
    float st_s[EV], st_q[EV];
    MTN_UNROLL for (int e = 0; e < EV; ++e) { st_s[e] = 0.f; st_q[e] = 0.f; }
    MTN_UNROLL for (int r = 0; r < 2 / NQ; ++r) {
        if (r) __syncthreads();
        MTN_UNROLL for (int q = 0; q < NQ; ++q)
            MTN_UNROLL for (int x = 0; x < 2; ++x)
                MTN_UNROLL for (int i = 0; i < FM; ++i)
                    MTN_UNROLL for (int j = 0; j < FN; ++j)
                        store4<CT>(sC + (wm * 32 + i * 16 + frow) * CP + (q * 2 + x) * BN + wn * 32 + j * 16 + 4 * (lane >> 4),
                                   acc[(r * NQ + q) * 2 + x][i][j]);
        // (the accumulators of this round are dead now: room for the row prefetch)
        // this thread's rows: pixel (b, 2a + qy, 2c + qx) of each, -1 beyond Md
        const int qy = r * NQ + qyl;
        int opix[RPT];
        MTN_UNROLL for (int i = 0; i < RPT; ++i) {
            const int m = m0 + trow + i * (NT / CPR);
            int b = 0, a = 0, c = 0;
            if (m < Md) decode(m, b, a, c);
            opix[i] = m < Md ? (b * p.OH + 2 * a + qy) * p.OW + 2 * c + qx : -1;
        }
        // raw-output values of the BatchNorm unit: in flight across the barrier and the first rows
        RawVec<T> bn_yv[RPT];
        if (gBy) {
            MTN_UNROLL for (int i = 0; i < RPT; ++i) {
                const long o = opix[i] < 0 ? 0 : opix[i];
                bn_yv[i] = ld_raw<T>(gBy + o * p.bn_ldy + n0, col);
            }
        }
        __syncthreads();
        float bn_sc[EV], bn_sh[EV], bn_mu[EV], bn_rs[EV];
        if (gBy) {
            MTN_UNROLL for (int e = 0; e < EV; ++e) {
                bn_sc[e] = p.bn_ss[n0 + col + e]; bn_sh[e] = p.bn_ss[p.N + n0 + col + e];
                bn_mu[e] = p.bn_mr[n0 + col + e]; bn_rs[e] = p.bn_mr[p.N + n0 + col + e];
            }
        }
        MTN_UNROLL for (int i = 0; i < RPT; ++i) {
            if (opix[i] < 0) continue;
            const int row = trow + i * (NT / CPR);
            const long o = opix[i];
            RawVec<T> v;
            if constexpr (std::is_same<CT, T>::value) {           // staged in the storage type: the chunk leaves as it is
                v.c = *reinterpret_cast<const chunk16*>(sC + row * CP + ccol);
                if (gR) {
                    float f[EV], rr[EV];
                    unpack<T>(v.c, f);
                    ld_vec<T>(gR + o * p.ldr + n0, col, rr);
                    MTN_UNROLL for (int e = 0; e < EV; ++e) f[e] += rr[e];
                    v = pack_raw<T>(f);
                }
            } else {                                              // split storage: f32 staging, split on the way out
                float f[EV];
                MTN_UNROLL for (int e = 0; e < EV; ++e) f[e] = sC[row * CP + ccol + e];
                if (gR) {
                    float rr[EV];
                    ld_vec<T>(gR + o * p.ldr + n0, col, rr);
                    MTN_UNROLL for (int e = 0; e < EV; ++e) f[e] += rr[e];
                }
                v = pack_raw<T>(f);
            }
            st_raw<T>(gO + o * p.ldc + n0, col, v);
            if (gBy) {
                float w[EV], yv[EV], rr[EV], dz[EV], xh[EV];
                unpack_raw<T>(v, w);
                unpack_raw<T>(bn_yv[i], yv);
                if (gBr) ld_vec<T>(gBr + o * p.bn_ldr + n0, col, rr);
                bn_dz_math<EV>(w, yv, rr, gBr != nullptr, bn_sc, bn_sh, bn_mu, bn_rs, p.bn_act == MTN_ACT_SILU, dz, xh);
                MTN_UNROLL for (int e = 0; e < EV; ++e) { st_s[e] += dz[e]; st_q[e] += dz[e] * xh[e]; }
            }
        }
    }
    if (gBy) {
        constexpr int TRN = NT / CPR, RW = NQ * 2 * BN;
        __syncthreads();                                   // everyone is done reading the staged tile
        float* red = reinterpret_cast<float*>(smem);       // [2][TRN][RW]
        MTN_UNROLL for (int e = 0; e < EV; ++e) {
            red[trow * RW + ccol + e] = st_s[e];
            red[(TRN + trow) * RW + ccol + e] = st_q[e];
        }
        __syncthreads();
        for (int i = tid; i < 2 * BN; i += NT) {
            const int which = i / BN, cc = i - which * BN;
            float t = 0.f;
            for (int q = 0; q < TRN; ++q)
                MTN_UNROLL for (int x = 0; x < NQ * 2; ++x) t += red[(which * TRN + q) * RW + x * BN + cc];
            p.stats[((long)tile_m * 2 + which) * p.N + n0 + cc] = t;
        }
    }
}

size_t dgrad2_lds(const IgemmArgs& p, int esz) {
    const size_t wr8 = (size_t)((D2_BM + p.W + 1 + 7) & ~7);
    const size_t loop = (2 * (size_t)D2_BN + wr8) * D2_CS * 128 + 128;
    const size_t epi = (size_t)D2_BM * ((esz == 2 ? 2 : 1) * 2 * D2_BN * esz + 16);
    return loop > epi ? loop : epi;
}

int g_dgrad2 = 1;     // debug key 15 (0: off, 1: on, n >= 2: on with an LDS limit of n KB)
int g_dgrad2_lds_kb = 96;    // two workgroups per CU up to 80 KB; the 96-column maps of 384x384 images (92 KB, one per CU) still beat the generic parity-major igemm (config 4: -0.12 ms)

template <class T>
int launch_dgrad2(const IgemmArgs& p, hipStream_t stream) {
    const size_t lds = dgrad2_lds(p, (int)sizeof(T));
#ifndef MTN_EMU
    static MtnLdsAttr attr;
    mtn_set_max_lds(attr, &dgrad2_kernel<T>, 160 * 1024);
#endif
    const int tiles_m = mtn_cdiv((long)p.B * p.H * p.W, D2_BM);
    MTN_REQUIRE(!p.stats || p.stats_rows == tiles_m, "dgrad2: the statistics slab was sized for %d row tiles, this launch writes %d",
                p.stats_rows, tiles_m);
    const long tiles = (long)tiles_m * (p.N / D2_BN);
    MTN_LAUNCH(dgrad2_kernel<T>, dim3((unsigned)tiles), dim3(D2_NT), lds, stream, p);
    return mtn_check_launch("dgrad2");
}

}  // namespace

void mtn_dgrad2_set(int on) {
    g_dgrad2 = on ? 1 : 0;
    g_dgrad2_lds_kb = on >= 2 && on <= 160 ? on : 96;
}

// the fused form serves 3x3 stride-2 pad-1 data gradients whose window still lets two workgroups share a CU
bool mtn_dgrad2_ok(const IgemmArgs& p, int dtype) {
    const int esz = dtype == MTN_BF16 ? 2 : 4, bk = 128 / esz;
    return g_dgrad2 && p.pm && p.KH == 3 && p.KW == 3 && p.div == 2 && p.mul == 1 && p.sgn == -1 && p.off == 1 &&
           p.OH == 2 * p.H && p.OW == 2 * p.W && p.Cin % (D2_CS * bk) == 0 && p.N % D2_BN == 0 && !p.bias && (!p.stats || p.bn_y) &&
           p.act == MTN_ACT_NONE && !p.aux_out && dgrad2_lds(p, esz) <= (size_t)g_dgrad2_lds_kb * 1024;
}

int mtn_dgrad2_tiles_m(const IgemmArgs& p) { return mtn_cdiv((long)p.B * p.H * p.W, D2_BM); }

int mtn_dgrad2_run(const IgemmArgs& p, int dtype, hipStream_t stream) {
    if (dtype == MTN_F32X3) return launch_dgrad2<f32x3_t>(p, stream);
    return dtype == MTN_BF16 ? launch_dgrad2<bf16_t>(p, stream) : launch_dgrad2<float>(p, stream);
}
