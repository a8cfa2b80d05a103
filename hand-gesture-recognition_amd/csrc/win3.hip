// win3.hip -- persistent, weight-stationary kernel for the 64 -> 64 channel 3x3 stride-1 convolutions of stage 1, forward and
// data gradient (the ResBasicBlocks of cspelan1: model/gelan.py:69-70,78-87 and their autograd).
//
//   Out[m][n] = sum over the 9 taps and 64 channels of  In[m + dy*W + dx][c] * Wp[n][tap][c]       m = (b, y, x), 128 rows per tile
//
// These 8 launches per step were 2.9x off their floor in igemm.hip's window mode (86 us against ~28 us of HBM time and ~23 us of
// MFMA time): a 128 x 64 tile there re-streams the nine 8 KB weight tiles through the L2 -> LDS path for
// every tile -- 72 KB of weights beside 29 KB of activations -- with a barrier per tap, and 40 % of a tile's life is fixed cost.
// Here (the structure of pw.hip, which see):
//   * ONE workgroup per CU walks its tiles; the WEIGHTS of all nine taps live in registers (wave = 32 x 32 of the tile: 36
//     fragments, 144 VGPRs) -- no weight traffic and no barrier inside a tile's 72 MFMAs per wave;
//   * the activation WINDOW of a tile (its 128 rows and W+1 halo rows either side, contiguous in the flattened pixel order)
//     arrives by LDS-DMA into a ring of NSLOT slots, two tiles ahead, under counted s_waitcnt vmcnt and raw s_barrier;
//   * every tap reads its fragments from the resident window at a row offset; rows whose tap leaves the image read a zero row;
//   * the epilogue's second operands (data gradient: residual, the raw output / residual of the BatchNorm unit whose backward
//     sums this launch pre-reduces) are requested for the NEXT tile before the next window's DMA is issued, so that the wait at
//     their first use never has to wait for a DMA younger than they are;
//   * forward statistics / BatchNorm-backward sums are accumulated over the workgroup's tiles and written once: one slab row per
//     workgroup (<= 256 rows: bn_finalize / bn_bwd_finalize take them without a fold launch).
//
// MEASURED (profiles/r04_win3_ab.txt): ties the window mode without second operands (75.9 vs 78.5 us), loses with them.  Built only
// with -DMTN_EXPERIMENTS (the CPU executor and diagnostic GPU builds); the product library dispatches none of this.
#include <type_traits>
#include "mtn_types.h"
#include "mtn_kernels.h"

#ifndef MTN_EXPERIMENTS
void mtn_win3_set(int on) { (void)on; }
bool mtn_win3_ok(const IgemmArgs& p, int dtype) { (void)p; (void)dtype; return false; }
int mtn_win3_rows(const IgemmArgs& p) { (void)p; return 0; }
int mtn_win3_run(const IgemmArgs& p, int dtype, hipStream_t stream) { (void)p; (void)dtype; (void)stream; return MTN_ERR_UNSUPPORTED; }
#else

namespace {

constexpr int W3_BM = 128, W3_BN = 64, W3_NT = 512;

#define W3_INL __attribute__((always_inline))
#ifdef MTN_EMU
#define W3_WAIT_VM(n) ((void)0)
#define W3_BARRIER() __syncthreads()
#else
#define W3_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define W3_BARRIER()                                       \
    do {                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  \
        __builtin_amdgcn_s_barrier();                      \
        asm volatile("" ::: "memory");                     \
    } while (0)
#endif

// PWV = LDS-DMA pieces (8 rows x 128 bytes) per wave and tile: the window slot has 64 * PWV rows >= 128 + 2 (W + 1);
// EPI = 0: forward / plain data gradient (optional sum / sum-of-squares statistics), 1: data gradient with second operands
template <int PWV, int NSLOT, int EPI>
__global__ __launch_bounds__(W3_NT) void win3_kernel(IgemmArgs p, int tiles_m, int G) {
    typedef bf16_t T;
    constexpr int BM = W3_BM, BN = W3_BN, NT = W3_NT, BK = 64, VEC = 8, KS = 2, KI = 32;
    constexpr int FM = 2, FN = 2;                         // wave tile 32 x 32, waves 4 (rows) x 2 (columns)
    constexpr int WR = 64 * PWV;                          // window rows of a slot
    constexpr int SLOT = (WR + 8) * BK;                   // ... + a block of zero rows (index WR): the padding taps read it
    constexpr int CP = BN + 8;
    constexpr int EV = 8, CPR = BN / EV, TR = NT / CPR, RPT = BM / TR;      // 8 column groups x 64 thread rows, 2 rows per thread
    __shared__ __attribute__((aligned(16))) T sA0[SLOT];
    __shared__ __attribute__((aligned(16))) T sA1[SLOT];
    __shared__ __attribute__((aligned(16))) T sA2[NSLOT == 3 ? SLOT : 8];
    __shared__ __attribute__((aligned(16))) T sC[BM * CP];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int frow = lane & 15, fg = lane >> 4;
    const int walker = blockIdx.x;
    const int halo = p.W + 1;
    const int total = p.B * p.H * p.W;                    // rows of the gather source (= M: stride 1)

    const T* __restrict__ gW = reinterpret_cast<const T*>(p.Wp);
    T* __restrict__ gO = reinterpret_cast<T*>(p.Out);
    const T* __restrict__ gR = reinterpret_cast<const T*>(p.res);
    const T* __restrict__ gBy = reinterpret_cast<const T*>(p.bn_y);
    const T* __restrict__ gBr = reinterpret_cast<const T*>(p.bn_res);

    // ---- weights of all nine taps -> registers, once ------------------------------------------------------------------------
    bf16x8_t wq[9][KS][FN];
    {
        const T* w = gW + (long)(wn * 32 + frow) * p.K + fg * VEC;
        MTN_UNROLL for (int tap = 0; tap < 9; ++tap)
            MTN_UNROLL for (int ks = 0; ks < KS; ++ks)
                MTN_UNROLL for (int jj = 0; jj < FN; ++jj) wq[tap][ks][jj] = Mma<T>::load(w + (long)jj * 16 * p.K + tap * BK + ks * KI);
    }
    const int ccol = (tid % CPR) * EV, trow = tid / CPR;
    float bn_sc[EV], bn_sh[EV], bn_mu[EV], bn_rs[EV];
    if (EPI == 1 && gBy) {
        MTN_UNROLL for (int e = 0; e < EV; ++e) {
            bn_sc[e] = p.bn_ss[ccol + e]; bn_sh[e] = p.bn_ss[p.N + ccol + e];
            bn_mu[e] = p.bn_mr[ccol + e]; bn_rs[e] = p.bn_mr[p.N + ccol + e];
        }
    }
    float st_s[EV], st_q[EV];
    MTN_UNROLL for (int e = 0; e < EV; ++e) { st_s[e] = 0.f; st_q[e] = 0.f; }
    // ---- window ring --------------------------------------------------------------------------------------------------------
    const BufDesc bufA = make_buf(p.A, p.a_bytes);
    // zero rows of every slot: one out-of-range LDS-DMA piece each (no ordinary LDS store ever touches a slot: hipcc then keeps the
    // ring's counted waits apart from the staging tile's stores)
    if (wid == 0) {
        buf_glds16(bufA, 0xFFFFFFF0u, reinterpret_cast<char*>(sA0 + WR * BK));
        buf_glds16(bufA, 0xFFFFFFF0u, reinterpret_cast<char*>(sA1 + WR * BK));
        if (NSLOT == 3) buf_glds16(bufA, 0xFFFFFFF0u, reinterpret_cast<char*>(sA2 + WR * BK));
    }
#ifndef MTN_EMU
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // ... and the ordinary loads above are done before the ring starts
#endif

    const int prow = lane >> 3, pchunk = ((lane & 7) ^ (lane >> 3)) * VEC;
    auto dma_tile = [&](int tm, T* slot) W3_INL {
        // window row r = input row tm*128 - halo + r; rows outside [0, total) -- and every row of a tile beyond the walk -- are
        // zero filled by the buffer range check
        const int g0 = tm * BM - halo + wid * PWV * 8 + prow;
        MTN_UNROLL for (int i = 0; i < PWV; ++i) {
            const int g = g0 + i * 8;
            const bool ok = tm < tiles_m && g >= 0 && g < total;
            const unsigned off = (unsigned)(g * p.lda + pchunk) * 2u;
            buf_glds16(bufA, ok ? off : 0xFFFFFFF0u, reinterpret_cast<char*>(slot + (wid * PWV + i) * 8 * BK));
        }
    };

    // per-tile tap validity of this lane's two fragment rows: bit kh of vy / bit kw of vx set iff that tap row / column is inside
    // the image (multiply-high row decode as in igemm.hip)
    auto tap_bits = [&](int tm, unsigned* vy, unsigned* vx) W3_INL {
        MTN_UNROLL for (int i = 0; i < FM; ++i) {
            const unsigned m = (unsigned)(tm * BM + wm * 32 + i * 16 + frow);
            const unsigned yt = p.W == 1 ? m : __umulhi(m, p.magic_w);           // m / W
            const int ox = (int)(m - yt * (unsigned)p.W);
            const unsigned bi = p.H == 1 ? yt : __umulhi(yt, p.magic_h);         // yt / H
            const int oy = (int)(yt - bi * (unsigned)p.H);
            unsigned a = 0, b = 0;
            MTN_UNROLL for (int t = 0; t < 3; ++t) {
                const int ty = oy + t * p.sgn + p.off, tx = ox + t * p.sgn + p.off;
                if (ty >= 0 && ty < p.H) a |= 1u << t;
                if (tx >= 0 && tx < p.W) b |= 1u << t;
            }
            vy[i] = a; vx[i] = b;
        }
    };

    f32x4_t acc[FM][FN];
    auto mma_tile = [&](const T* slot, const unsigned* vy, const unsigned* vx) W3_INL {
        MTN_UNROLL for (int i = 0; i < FM; ++i)
            MTN_UNROLL for (int jj = 0; jj < FN; ++jj) acc[i][jj] = zero4();
        // (the fragment addresses of the nine taps are tile-invariant up to the padding select: left alone, hipcc computes all 36 of
        //  them once, keeps them beside the 144 weight registers and spills; an opaque copy of the lane's row keeps the arithmetic
        //  -- a handful of VALU per tap -- inside the tile)
        int fr = frow;
#ifndef MTN_EMU
        asm volatile("" : "+v"(fr));
#endif
        MTN_UNROLL for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap - kh * 3;
            const int delta = (kh * p.sgn + p.off) * p.W + (kw * p.sgn + p.off) + halo;
            int row[FM];
            MTN_UNROLL for (int i = 0; i < FM; ++i) {
                const int jr = wm * 32 + i * 16 + fr + delta;
                const bool ok = ((vy[i] >> kh) & (vx[i] >> kw)) & 1u;
                row[i] = ok ? jr : WR;
            }
            MTN_UNROLL for (int ks = 0; ks < KS; ++ks) {
                bf16x8_t fa[FM];
                MTN_UNROLL for (int i = 0; i < FM; ++i) fa[i] = RowFrag<T>::load(slot + row[i] * BK, ks, fg, row[i] & 7);
                MTN_UNROLL for (int i = 0; i < FM; ++i)
                    MTN_UNROLL for (int jj = 0; jj < FN; ++jj) acc[i][jj] = Mma<T>::mma(wq[tap][ks][jj], fa[i], acc[i][jj]);
            }
            // (144 registers hold the weights: the fragment reads of a tap are not hoisted over the MFMAs of the tap before --
            //  the partner wave of the SIMD covers their latency -- or hipcc's read-ahead spills into scratch, whose loads and
            //  stores are vector-memory operations that would also break the counted waits of the ring)
            MTN_SCHED_FENCE();
        }
    };
    auto stage_c = [&]() W3_INL {
        MTN_UNROLL for (int i = 0; i < FM; ++i)
            MTN_UNROLL for (int jj = 0; jj < FN; ++jj)
                store4<T>(sC + (wm * 32 + i * 16 + frow) * CP + wn * 32 + jj * 16 + 4 * fg, acc[i][jj]);
    };
    // second operands of the epilogue for tile tm (EPI == 1), requested one tile ahead
    chunk16 o_res[RPT], o_by[RPT], o_br[RPT];
    auto fetch_ops = [&](int tm) W3_INL {
        if (EPI != 1) return;
        MTN_UNROLL for (int i = 0; i < RPT; ++i) {
            long m = (long)tm * BM + trow + i * TR;
            if (tm >= tiles_m) m = 0;
            if (gR) o_res[i] = *reinterpret_cast<const chunk16*>(gR + m * p.ldr + ccol);
            if (gBy) o_by[i] = *reinterpret_cast<const chunk16*>(gBy + m * p.bn_ldy + ccol);
            if (gBr) o_br[i] = *reinterpret_cast<const chunk16*>(gBr + m * p.bn_ldr + ccol);
        }
    };
    auto epilogue = [&](int tm) W3_INL {
        MTN_UNROLL for (int i = 0; i < RPT; ++i) {
            const int row = trow + i * TR;
            const long m = (long)tm * BM + row;
            chunk16 o = *reinterpret_cast<const chunk16*>(sC + row * CP + ccol);
            float w[EV];
            if (EPI == 1 && gR) {
                float rr[EV];
                unpack<T>(o, w);
                unpack<T>(o_res[i], rr);
                MTN_UNROLL for (int e = 0; e < EV; ++e) w[e] += rr[e];
                o = pack<T>(w);
            }
            *reinterpret_cast<chunk16*>(gO + m * p.ldc + ccol) = o;
            if (p.stats) {      // of the values exactly as stored
                unpack<T>(o, w);
                if (EPI == 1 && gBy) {
                    float yv[EV], rr[EV], dz[EV], xh[EV];
                    unpack<T>(o_by[i], yv);
                    if (gBr) unpack<T>(o_br[i], rr);
                    bn_dz_math<EV>(w, yv, rr, gBr != nullptr, bn_sc, bn_sh, bn_mu, bn_rs, p.bn_act == MTN_ACT_SILU, dz, xh);
                    MTN_UNROLL for (int e = 0; e < EV; ++e) { st_s[e] += dz[e]; st_q[e] += dz[e] * xh[e]; }
                } else {
                    MTN_UNROLL for (int e = 0; e < EV; ++e) { st_s[e] += w[e]; st_q[e] += w[e] * w[e]; }
                }
            }
        }
    };
    // Vector-memory operations of a wave per step, in issue order:
    //   EPI 0:  [window pieces of t + NSLOT: PWV] [stores of t: RPT]                       (as pw.hip)
    //   EPI 1:  [stores of t: RPT] [operand loads of t + 1: NL] [window pieces of t + NSLOT: PWV]
    // "Tile t has landed" = at most the operations issued after its pieces are outstanding: the NSLOT - 1 younger windows and the
    // stores (+ operand loads) of the SEEN steps in between -- min(t, NSLOT) of them for EPI 0, min(t, NSLOT - 1) for EPI 1.
    const int nl = EPI == 1 ? RPT * ((gR ? 1 : 0) + (gBy ? 1 : 0) + (gBr ? 1 : 0)) : 0;
    auto wait_tile = [&](auto SEEN) W3_INL {
#ifndef MTN_EMU
        constexpr int seen = decltype(SEEN)::value;
        constexpr int base = (NSLOT - 1) * PWV + seen * RPT;
        if (EPI != 1 || nl == 0) W3_WAIT_VM(base);
        else if (nl == RPT) W3_WAIT_VM(base + seen * RPT);
        else if (nl == 2 * RPT) W3_WAIT_VM(base + seen * 2 * RPT);
        else W3_WAIT_VM(base + seen * 3 * RPT);
#endif
    };
    auto step = [&](int t, T* cur, auto SEEN) W3_INL {
        const int tm = walker + t * G;
        unsigned vy[FM], vx[FM];
        tap_bits(tm, vy, vx);
        wait_tile(SEEN);
        W3_BARRIER();                                      // window t is visible to every wave; everyone is done with sC (epilogue t-1)
        mma_tile(cur, vy, vx);
        stage_c();
        W3_BARRIER();                                      // `cur` is consumed, sC is written
        if (EPI == 0) {
            dma_tile(tm + NSLOT * G, cur);
            epilogue(tm);
        } else {
            epilogue(tm);
            fetch_ops(tm + G);                             // second operands of the next tile: older than the DMA below
            dma_tile(tm + NSLOT * G, cur);
        }
    };
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    typedef std::integral_constant<int, 2> S2;
    typedef std::integral_constant<int, (EPI == 0 ? NSLOT : NSLOT - 1)> SN;

    const int my_tiles = walker < tiles_m ? (tiles_m - walker + G - 1) / G : 0;
    if (my_tiles > 0) {
        fetch_ops(walker);
#ifndef MTN_EMU
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        dma_tile(walker, sA0);
        dma_tile(walker + G, sA1);
        if (NSLOT == 3) dma_tile(walker + 2 * G, sA2);
        // whole rounds of NSLOT unconditional steps in the steady state (pw.hip: a skippable step makes hipcc drain the ring)
        if (my_tiles >= NSLOT) {
            step(0, sA0, S0{});
            step(1, sA1, S1{});
            if (NSLOT == 3) step(2, sA2, S2{});
            int t = NSLOT;
            for (; t + NSLOT <= my_tiles; t += NSLOT) {
                step(t, sA0, SN{});
                step(t + 1, sA1, SN{});
                if (NSLOT == 3) step(t + 2, sA2, SN{});
            }
            if (t < my_tiles) {
                step(t, sA0, SN{});
                if (t + 1 < my_tiles) step(t + 1, sA1, SN{});
            }
        } else {
            step(0, sA0, S0{});
            if (1 < my_tiles) step(1, sA1, S1{});
        }
    }
#ifndef MTN_EMU
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the ring's tail has landed before the LDS is given up
#endif
    if (p.stats) {
        // 64 thread rows: the 8 of a wave by shuffles (lanes l, l + 8, ..., l + 56 hold the same channels), the 8 waves through LDS
        MTN_UNROLL for (int e = 0; e < EV; ++e) {
            MTN_UNROLL for (int s = 8; s <= 32; s <<= 1) {
                st_s[e] += __shfl_xor(st_s[e], s, 64);
                st_q[e] += __shfl_xor(st_q[e], s, 64);
            }
        }
        __syncthreads();
        float* red = reinterpret_cast<float*>(sC);         // [2][8 waves][BN] f32 = 4 KB
        if (lane < 8) {
            MTN_UNROLL for (int e = 0; e < EV; ++e) {
                red[wid * BN + ccol + e] = st_s[e];
                red[(8 + wid) * BN + ccol + e] = st_q[e];
            }
        }
        __syncthreads();
        for (int i = tid; i < 2 * BN; i += NT) {
            const int which = i / BN, c = i - which * BN;
            float t = 0.f;
            MTN_UNROLL for (int q = 0; q < 8; ++q) t += red[(which * 8 + q) * BN + c];
            p.stats[((long)walker * 2 + which) * p.N + c] = t;
        }
    }
}

int g_win3 = 0;                 // debug key 23 (experiments build): 1 = on for long walks, 2 = on for any launch (tests)

int win3_walkers(const IgemmArgs& p) {
    const int tiles_m = p.M / W3_BM;
    return tiles_m < 256 ? tiles_m : 256;
}

template <int PWV, int NSLOT>
int launch_win3(const IgemmArgs& p, hipStream_t stream) {
    const int G = win3_walkers(p), tiles_m = p.M / W3_BM;
    MTN_REQUIRE(!p.stats || p.stats_rows == G, "win3: the statistics slab was sized for %d rows, this launch writes %d", p.stats_rows, G);
    if (p.res || p.bn_y) MTN_LAUNCH((win3_kernel<PWV, NSLOT, 1>), dim3(G), dim3(W3_NT), 0, stream, p, tiles_m, G);
    else MTN_LAUNCH((win3_kernel<PWV, NSLOT, 0>), dim3(G), dim3(W3_NT), 0, stream, p, tiles_m, G);
    return mtn_check_launch("win3");
}

}  // namespace

void mtn_win3_set(int on) { g_win3 = (on == 2) ? 2 : (on ? 1 : 0); }

// 64 -> 64 channel 3x3 stride-1 layers in bf16 whose window (128 rows + 2 (W + 1) halo rows) fits a slot, whole 128-row tiles,
// walks long enough to amortise the 36 weight fragments and the ring fill
bool mtn_win3_ok(const IgemmArgs& p, int dtype) {
    if (!g_win3 || dtype != MTN_BF16) return false;
    if (p.KH != 3 || p.KW != 3 || p.mul != 1 || p.div != 1 || p.pm || p.OH != p.H || p.OW != p.W) return false;
    if (p.Cin != 64 || p.N != 64 || p.M % W3_BM) return false;
    if (W3_BM + 2 * (p.W + 1) > 384) return false;
    if (p.bias || p.aux_out || p.act != MTN_ACT_NONE) return false;
    if (g_win3 != 2 && p.M / W3_BM < 8 * 256) return false;
    if ((size_t)p.M * p.lda * 2 >= (1ull << 31)) return false;
    return true;
}
int mtn_win3_rows(const IgemmArgs& p) { return win3_walkers(p); }

int mtn_win3_run(const IgemmArgs& p, int dtype, hipStream_t stream) {
    (void)dtype;
    if (W3_BM + 2 * (p.W + 1) <= 256) return launch_win3<4, 3>(p, stream);
    return launch_win3<6, 2>(p, stream);
}
#endif  // MTN_EXPERIMENTS
