// pw.hip -- persistent, weight-stationary kernel for the short-K pointwise products (1x1 convolutions at stride 1, their data
// gradients, nn.Linear): model/gelan.py:131-143 (cv1 / cv4), model/multitasknet.py:13,26 (proj), model/transformer.py:34-37,56-58.
//
//   Out[m][n] = epilogue( sum_k A[m][k] * Wp[n][k] ),      K = Cin in {128, 256, 512},  N % 128 == 0
//
// Why its own kernel: these launches are HBM-bound (64 .. 256 flop/B) and igemm.hip ran them at
// 0.36 of the HBM roof -- a tile there is a serial chain (prologue -> cold first load -> 2..8 K steps -> LDS staging -> stores)
// that only a second resident workgroup overlaps, and a loop around that kernel made hipcc hoist 70 registers of tile state.
// Here ONE workgroup per CU (8 waves, up to 256 VGPRs) walks its tiles itself:
//   * the WEIGHTS live in registers for the whole launch: wave (wm, wn) keeps the fragments of its 32 output columns over all
//     K (16 x 4 VGPRs at K = 256) -- no weight tile in LDS, no weight traffic per tile;
//   * the ACTIVATIONS stream through a ring of NSLOT LDS slots by LDS-DMA: the 64 x K tile t+NSLOT-1 is in flight while tile t
//     is multiplied, with counted s_waitcnt vmcnt and raw s_barrier (a __syncthreads() would drain the ring);
//   * the output tile of t leaves through a separate LDS staging tile while the MFMAs of t+1 run: stores retire under loads;
//   * per-tile state is one row offset: rows are contiguous (pointwise), no pixel decode, no tap machinery.
// BatchNorm partial statistics (forward) are accumulated across a workgroup's tiles in registers and written ONCE per
// workgroup: the slab has gridDim rows (<= 256) instead of one per tile, so bn_finalize needs no fold launch.
//
// Epilogues served: bias, GELU (+ pre-activation copy), SiLU (folded-BatchNorm inference), statistics.  Epilogues that READ
// a second tensor (residual, BatchNorm hand-over, GELU') stay in igemm.hip: an ordinary global load beside in-flight LDS-DMA
// makes hipcc wait vmcnt(0) at its first use and the ring would drain every tile (cdna_hip_programming.md section 5).
#include <type_traits>
#include "mtn_types.h"
#include "mtn_kernels.h"

namespace {

constexpr int PW_BM = 64, PW_BN = 128, PW_NT = 512;

// (every lambda of the kernel must inline: one that is called from several places and left out of line gets its captured
//  register arrays through scratch memory)
#define PW_INL __attribute__((always_inline))
#ifdef MTN_EMU
#define PW_WAIT_VM(n) ((void)0)
#define PW_BARRIER() __syncthreads()
#else
#define PW_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
// LDS traffic of this wave retired, then the workgroup barrier; vector memory (the ring, the stores) stays in flight
#define PW_BARRIER()                                      \
    do {                                                  \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_s_barrier();                     \
        asm volatile("" ::: "memory");                    \
    } while (0)
#endif

// NSLAB = K / (128 bytes of K per row), NSLOT = ring depth (3 where LDS allows, else 2)
template <class T, int NSLAB, int NSLOT>
__global__ __launch_bounds__(PW_NT) void pw_kernel(IgemmArgs p, int tiles_m, int G) {
    constexpr int BM = PW_BM, BN = PW_BN, NT = PW_NT;
    constexpr int BK = 128 / (int)sizeof(T), VEC = 16 / (int)sizeof(T);
    constexpr int KI = Mma<T>::KI, KS = BK / KI;
    constexpr int FM = 2, FN = 2;                         // wave tile 32 x 32, waves 2 (rows) x 4 (columns)
    constexpr int SLOT = NSLAB * BM * BK;                 // elements of one ring slot: [NSLAB][BM][BK], rows XOR-swizzled
    typedef typename StageT<T>::type CT;
    constexpr int CP = BN + 16 / (int)sizeof(CT);
    constexpr int EV = Ev<T>::N, CPR = BN / EV, TR = NT / CPR, RPT = BM / TR;
    static_assert(NT % CPR == 0 && BM % TR == 0, "store mapping");
    constexpr int PA = NSLAB;                             // LDS-DMA pieces per wave and tile: wave w fills rows 8w .. 8w+7 of every slab
    // three distinct objects: the compiler can tell the slot being read from the slots being filled
    __shared__ __attribute__((aligned(16))) T sA0[SLOT];
    __shared__ __attribute__((aligned(16))) T sA1[SLOT];
    __shared__ __attribute__((aligned(16))) T sA2[NSLOT == 3 ? SLOT : 8];
    __shared__ __attribute__((aligned(16))) CT sC[BM * CP];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 2, wn = wid & 3;
    const int frow = lane & 15, fg = lane >> 4;

    // (n-tile, walker) of this workgroup: the n-tiles of one m-tile sequence sit on one XCD (id % 8) and walk it together,
    // so the second and later readers of an activation tile hit that XCD's L2
    const int tiles_n = p.N / BN;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int tile_n = j % tiles_n, walker = (j / tiles_n) * 8 + xcd;
    const int n0 = tile_n * BN;

    const T* __restrict__ gW = reinterpret_cast<const T*>(p.Wp);
    T* __restrict__ gO = reinterpret_cast<T*>(p.Out);
    T* __restrict__ gAux = reinterpret_cast<T*>(p.aux_out);

    // ---- weights -> registers, once ---------------------------------------------------------------------------------
    typename Mma<T>::Frag wq[NSLAB][KS][FN];
    {
        const T* w = gW + (long)(n0 + wn * 32 + frow) * p.K + fg * VEC;      // lane group g: the 16-byte slot g of each K sub-step
        MTN_UNROLL for (int s = 0; s < NSLAB; ++s)
            MTN_UNROLL for (int ks = 0; ks < KS; ++ks)
                MTN_UNROLL for (int jj = 0; jj < FN; ++jj) wq[s][ks][jj] = Mma<T>::load(w + (long)jj * 16 * p.K + s * BK + ks * KI);
    }
    // per-thread epilogue constants (the n-tile never changes)
    const int ccol = (tid % CPR) * EV, trow = tid / CPR;
    float kb[EV];
    MTN_UNROLL for (int e = 0; e < EV; ++e) kb[e] = p.bias ? p.bias[n0 + ccol + e] : 0.f;
    float st_s[EV], st_q[EV];
    MTN_UNROLL for (int e = 0; e < EV; ++e) { st_s[e] = 0.f; st_q[e] = 0.f; }
#ifndef MTN_EMU
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the ordinary loads above are done before the first LDS-DMA is issued
#endif

    // ---- activation ring ----------------------------------------------------------------------------------------------
    const BufDesc bufA = make_buf(p.A, p.a_bytes);
    // lane's row / source chunk inside a 1 KiB piece (8 rows x 128 bytes, chunk swizzled by the row)
    const unsigned lane_off = (unsigned)(((wid * 8 + (lane >> 3)) * p.lda + (((lane & 7) ^ (lane >> 3)) * VEC)) * (int)sizeof(T));
    auto dma_tile = [&](int tm, T* slot) PW_INL {
        // rows tm*64 .. +63; a tile beyond the last one is all out of range: zero fill, nothing is stored from it
        const unsigned base = tm < tiles_m ? lane_off + (unsigned)tm * (unsigned)(BM * p.lda * (int)sizeof(T)) : 0xFFFFF000u;
        MTN_UNROLL for (int s = 0; s < NSLAB; ++s)
            buf_glds16(bufA, base + (unsigned)(s * 128), reinterpret_cast<char*>(slot + (s * BM + wid * 8) * BK));
    };

    f32x4_t acc[FM][FN];
    auto mma_tile = [&](const T* slot) PW_INL {
        MTN_UNROLL for (int i = 0; i < FM; ++i)
            MTN_UNROLL for (int jj = 0; jj < FN; ++jj) acc[i][jj] = zero4();
        const T* a = slot + (wm * 32 + frow) * BK;
        const int sw = frow & 7;
        MTN_UNROLL for (int s = 0; s < NSLAB; ++s)
            MTN_UNROLL for (int ks = 0; ks < KS; ++ks) {
                typename Mma<T>::Frag fa[FM];
                MTN_UNROLL for (int i = 0; i < FM; ++i) fa[i] = RowFrag<T>::load(a + (s * BM + i * 16) * BK, ks, fg, sw);
                MTN_UNROLL for (int i = 0; i < FM; ++i)
                    MTN_UNROLL for (int jj = 0; jj < FN; ++jj) acc[i][jj] = Mma<T>::mma(wq[s][ks][jj], fa[i], acc[i][jj]);
            }
    };
    auto stage_c = [&]() PW_INL {      // (the W fragment is the MFMA's row operand: a lane's quad is 4 consecutive channels of one row)
        MTN_UNROLL for (int i = 0; i < FM; ++i)
            MTN_UNROLL for (int jj = 0; jj < FN; ++jj)
                store4<CT>(sC + (wm * 32 + i * 16 + frow) * CP + wn * 32 + jj * 16 + 4 * fg, acc[i][jj]);
    };
    auto epilogue = [&](int tm) PW_INL {
        MTN_UNROLL for (int i = 0; i < RPT; ++i) {
            const int row = trow + i * TR;
            const long m = (long)tm * BM + row;
            float v[EV];
            MTN_UNROLL for (int e = 0; e < EV; ++e) v[e] = to_f<CT>(sC[row * CP + ccol + e]) + kb[e];
            if (p.act == MTN_ACT_GELU) {
                if (gAux) st_vec<T>(gAux + m * p.ld_aux + n0, ccol, v);
                MTN_UNROLL for (int e = 0; e < EV; ++e) v[e] = geluf_(v[e]);
            } else if (p.act == MTN_ACT_SILU) {
                MTN_UNROLL for (int e = 0; e < EV; ++e) v[e] = siluf_(v[e]);
            }
            const RawVec<T> o = pack_raw<T>(v);
            st_raw<T>(gO + m * p.ldc + n0, ccol, o);
            if (p.stats) {      // statistics of the values exactly as stored
                float w[EV];
                unpack_raw<T>(o, w);
                MTN_UNROLL for (int e = 0; e < EV; ++e) { st_s[e] += w[e]; st_q[e] += w[e] * w[e]; }
            }
        }
    };
    // VM operations a wave issues per tile besides its PA ring pieces: the epilogue's stores (M is a multiple of 64 and only
    // real tiles are stepped: every row is stored, the count is exact)
    constexpr int RAWS = std::is_same<T, f32x3_t>::value ? 2 : 1;      // 16-byte stores per st_raw
    constexpr int NST = RPT * RAWS;                                    // ... and twice that when the GELU pre-activation is kept
    const bool aux2 = p.act == MTN_ACT_GELU && gAux != nullptr;

    // One tile: `cur` holds tile t (its DMA was issued NSLOT tiles of work ago) and is refilled with tile t + NSLOT as soon as
    // everyone has read it.  A wave's vector-memory operations retire in issue order -- per step [ring pieces of t + NSLOT]
    // [stores of t] -- so "tile t has landed" is "at most the operations issued after its pieces are outstanding":
    //     (NSLOT - 1) ring tiles  +  the stores of the min(t, NSLOT) steps since     (SEEN = min(t, NSLOT), compile time)
    auto step = [&](int t, T* cur, auto SEEN) PW_INL {
        constexpr int seen = decltype(SEEN)::value;
        const int tm = walker + t * G;
#ifndef MTN_EMU
        if (aux2) PW_WAIT_VM((NSLOT - 1) * PA + seen * 2 * NST);
        else PW_WAIT_VM((NSLOT - 1) * PA + seen * NST);
#endif
        PW_BARRIER();                                      // tile t is visible to every wave; everyone is done with sC (epilogue t-1)
        mma_tile(cur);
        stage_c();
        PW_BARRIER();                                      // `cur` is consumed, sC is written
        dma_tile(tm + NSLOT * G, cur);                     // (a tile index beyond the walk: zero fill, never multiplied)
        epilogue(tm);
    };
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    typedef std::integral_constant<int, 2> S2;
    typedef std::integral_constant<int, NSLOT> SN;

    const int my_tiles = walker < tiles_m ? (tiles_m - walker + G - 1) / G : 0;
    if (my_tiles > 0) {
        dma_tile(walker, sA0);
        dma_tile(walker + G, sA1);
        if (NSLOT == 3) dma_tile(walker + 2 * G, sA2);
        // The first NSLOT steps have fewer stores behind them than the steady state.  The steady-state loop runs whole rounds
        // of NSLOT UNCONDITIONAL steps: with a skippable step inside it hipcc's own wait-count pass (it tracks the LDS-DMA of
        // each __shared__ object exactly) sees a path with fewer operations behind a slot's refill and waits vmcnt(2) at the
        // top of every round -- the whole ring drained, seen in the ISA.
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the ring's zero-fill tail has landed before the LDS is given up
#endif
    if (p.stats) {
        // the workgroup's sums over all its tiles: the four thread rows of a wave by shuffles (lanes l, l+16, l+32, l+48 hold
        // the same channels), the eight waves through LDS in a fixed order -- one slab row per workgroup, deterministic
        MTN_UNROLL for (int e = 0; e < EV; ++e) {
            st_s[e] += __shfl_xor(st_s[e], 16, 64); st_q[e] += __shfl_xor(st_q[e], 16, 64);
            st_s[e] += __shfl_xor(st_s[e], 32, 64); st_q[e] += __shfl_xor(st_q[e], 32, 64);
        }
        __syncthreads();
        float* red = reinterpret_cast<float*>(sC);         // [2][8 waves][BN] f32 = 8 KB
        if (lane < 16) {
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
            p.stats[((long)walker * 2 + which) * p.N + n0 + c] = t;
        }
    }
}

int g_pw = 1;                   // debug key 22: 0 = every pointwise product through igemm.hip, 2 = also launches of a few tiles (tests)

// workgroups that walk one n-tile: a multiple of 8 (XCD groups), one workgroup per CU in all
int pw_walkers(const IgemmArgs& p) {
    const int tiles_n = p.N / PW_BN;
    int g = (256 / tiles_n) & ~7;
    if (g < 8) g = 8;
    const int tiles_m = p.M / PW_BM;
    while (g > 8 && g - 8 >= tiles_m) g -= 8;             // never more walkers than tiles (rounded up to an XCD group)
    return g;
}

template <class T, int NSLAB, int NSLOT>
int launch_pw(const IgemmArgs& p, hipStream_t stream) {
    const int G = pw_walkers(p), tiles_n = p.N / PW_BN, tiles_m = p.M / PW_BM;
    MTN_REQUIRE(!p.stats || p.stats_rows == G, "pw: the statistics slab was sized for %d rows, this launch writes %d", p.stats_rows, G);
    MTN_LAUNCH((pw_kernel<T, NSLAB, NSLOT>), dim3(G * tiles_n), dim3(PW_NT), 0, stream, p, tiles_m, G);
    return mtn_check_launch("pw");
}

}  // namespace

void mtn_pw_set(int on) { g_pw = (on == 2) ? 2 : (on ? 1 : 0); }

// Shapes this kernel serves: pointwise taps, whole 64-row tiles, K of 2 / 4 LDS slabs, epilogues without a second input tensor
// (header) -- and only where it measured faster than igemm.hip's tiles (B = 256, profiles/r04_pointwise_ab.txt): the long walks
// of stages 1 and 2 (>= 16 tiles per workgroup: s1.cv1 77 -> 57 us, s1.cv4 103 -> 85, its data gradient 124 -> 94).  Short
// walks -- the 36,864-row stage-3 layers, the 37,120-row Linears: 4 .. 14 tiles per workgroup -- do not amortise the weight
// load and the ring fill and a lone workgroup per CU has nobody to cover its per-tile chain (qkv 27.7 -> 31.3 us); K = 512
// (MFMA-heavier, 248 VGPRs, two-slot ring) ties at best.  Those stay with igemm.hip.
bool mtn_pw_ok(const IgemmArgs& p, int dtype) {
    if (!g_pw || (dtype != MTN_BF16 && dtype != MTN_F32X3)) return false;
    if (p.KH * p.KW != 1 || p.mul != 1 || p.off != 0 || p.div != 1 || p.pm) return false;
    if (p.N % PW_BN || p.M % PW_BM) return false;
    if (dtype == MTN_BF16 ? (p.K != 128 && p.K != 256 && !(p.K == 512 && g_pw == 2)) : p.K != 128) return false;
    if (g_pw != 2 && p.M / PW_BM < 16 * pw_walkers(p)) return false;
    if (p.res || p.bn_y || p.act == MTN_ACT_GELU_GRAD_ || p.tok_pos) return false;
    if ((size_t)p.M * p.lda * (dtype == MTN_BF16 ? 2 : 4) >= (1ull << 32) - (1u << 20)) return false;      // 32-bit buffer offsets
    return true;
}
int mtn_pw_rows(const IgemmArgs& p) { return pw_walkers(p); }

int mtn_pw_run(const IgemmArgs& p, int dtype, hipStream_t stream) {
    if (dtype == MTN_F32X3) return launch_pw<f32x3_t, 4, 3>(p, stream);      // split storage, K = 128: four 32-channel slabs
    if (p.K == 128) return launch_pw<bf16_t, 2, 3>(p, stream);
    if (p.K == 256) return launch_pw<bf16_t, 4, 3>(p, stream);
    return launch_pw<bf16_t, 8, 2>(p, stream);
}
