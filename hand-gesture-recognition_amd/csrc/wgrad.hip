// wgrad.hip -- weight gradients of Conv2d / Linear on MFMA, reduction over the pixel axis.
//
//   dW[n][tap][ci] = sum_m dY[m][n] * X(m; tap, ci)        m = (b, oy, ox) output pixels
//
// (autograd of model/gelan.py:37-45, model/multitasknet.py:13, model/transformer.py:34-37,56-58.)
// This is a "TN" GEMM: both operands are stored pixel-major (NHWC), the reduction index m is the
// slow axis of both.  A workgroup owns a BN x BC tile of dW for ONE tap (n0.., ci0..) and one
// slice of the pixel axis (split-M); per step it stages BMR pixels of dY[:, n-tile] and of the
// tap-shifted X[:, ci-tile] into LDS as loaded (pixel-major, 16-byte coalesced) and reads the
// MFMA operands DOWN the pixel axis: in bf16 with ds_read_b64_tr_b16 (hardware transpose read,
// cdna_hip_programming.md T10), in f32 one element per lane.
// Partial tiles go to a [split][N][K] f32 slab; wgrad_reduce sums the splits in order
// (deterministic, no float atomics) and writes the gradient in the reference's OIHW layout.
#include "mtn_types.h"
#include "mtn_kernels.h"

namespace {

constexpr int PADB = 16;      // LDS row padding in bytes

template <class T, int BN, int BC, int WGM, int WGN, bool TR>
__global__ __launch_bounds__(WGM* WGN * 64) void wgrad_kernel(WgradArgs p) {
    constexpr int NT = WGM * WGN * 64;
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr int BMR = 128 / (int)sizeof(T);          // pixels per step: 64 bf16 / 32 f32
    constexpr int KI = Mma<T>::KI;
    constexpr int PY = BN + PADB / (int)sizeof(T);     // LDS pitches (elements)
    constexpr int PX = BC + PADB / (int)sizeof(T);
    constexpr int CPRY = BN / VEC, CPRX = BC / VEC;    // 16-byte chunks per row
    constexpr int RY = BMR * CPRY / NT, RX = BMR * CPRX / NT;
    constexpr int WM = BN / WGM, WN = BC / WGN;
    constexpr int FM = WM / 16, FN = WN / 16;
    static_assert(BMR * CPRY % NT == 0 && BMR * CPRX % NT == 0, "loader mapping");

    MTN_DYN_SMEM(smem);
    T* const sY = reinterpret_cast<T*>(smem);          // [BMR][PY]
    T* const sX = sY + BMR * PY;                       // [BMR][PX]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WGN, wn = wid % WGN;

    // XCD-aware order: the (n-tile, tap/ci-tile) workgroups of ONE pixel split re-read the same dY /
    // X rows (9 taps x N/BN tiles), so they must share an L2.  Workgroups are dealt round-robin
    // over the 8 XCDs (id % 8 labels the XCD group): XCD group x walks splits x, x+8, ... and runs
    // the tiles of a split back to back.  (PMC: without this the 64x64-tile layers pulled 1.25 GB
    // per launch through the fabric, 8x their operands.)
    const int tiles_n = p.N / BN;
    const int tiles = tiles_n * (p.K / BC);
    int tile, split;
    {
        const int L = blockIdx.x;
        if (p.splits % 8 == 0 && tiles <= 24) {       // many-tile layers already share operands within an XCD
            const int xcd = L & 7, j = L >> 3;
            split = (j / tiles) * 8 + xcd;
            tile = j % tiles;
        } else {
            split = L / tiles;
            tile = L % tiles;
        }
    }
    const int tile_n = tile % tiles_n, tile_k = tile / tiles_n;
    const int n0 = tile_n * BN;
    const int kcol0 = tile_k * BC;                     // column in [0, K)
    const int tap = kcol0 / p.Cin, c0 = kcol0 - tap * p.Cin;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const long m_begin = (long)split * p.rows_per_split;
    long m_end = m_begin + p.rows_per_split;
    if (m_end > p.M) m_end = p.M;
    const int steps = (int)((m_end - m_begin + BMR - 1) / BMR);

    const T* __restrict__ gY = reinterpret_cast<const T*>(p.dY);
    const T* __restrict__ gX = reinterpret_cast<const T*>(p.X);
    const int ohw = p.OH * p.OW;

    // per-thread loader rows
    int yrow[RY], ycc[RY];
    MTN_UNROLL for (int i = 0; i < RY; ++i) { const int c = tid + i * NT; yrow[i] = c / CPRY; ycc[i] = c % CPRY; }
    int xrow[RX], xcc[RX], xb[RX], xrem[RX];
    MTN_UNROLL for (int i = 0; i < RX; ++i) {
        const int c = tid + i * NT; xrow[i] = c / CPRX; xcc[i] = c % CPRX;
        const long m = m_begin + xrow[i];
        xb[i] = (int)(m / ohw); xrem[i] = (int)(m - (long)xb[i] * ohw);
    }

    chunk16 ry[RY], rx[RX];
    auto load_tiles = [&](int st) {
        const long mb = m_begin + (long)st * BMR;
        MTN_UNROLL for (int i = 0; i < RY; ++i) {
            const long m = mb + yrow[i];
            ry[i] = (m < m_end) ? *reinterpret_cast<const chunk16*>(gY + m * p.lddy + n0 + ycc[i] * VEC) : zero_chunk();
        }
        MTN_UNROLL for (int i = 0; i < RX; ++i) {
            const long m = mb + xrow[i];
            const int oy = xrem[i] / p.OW, ox = xrem[i] - oy * p.OW;
            const int iy = oy * p.stride - p.pad + kh, ix = ox * p.stride - p.pad + kw;
            const bool ok = (m < m_end) && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            rx[i] = ok ? *reinterpret_cast<const chunk16*>(gX + ((long)(xb[i] * p.H + iy) * p.W + ix) * p.ldx + c0 + xcc[i] * VEC)
                       : zero_chunk();
            xrem[i] += BMR;                                  // advance this row to the next step
            while (xrem[i] >= ohw) { xrem[i] -= ohw; ++xb[i]; }
        }
    };
    auto store_tiles = [&](int stage) {
        T* y = sY + stage * BMR * PY;
        T* x = sX + stage * BMR * PX;
        MTN_UNROLL for (int i = 0; i < RY; ++i) *reinterpret_cast<chunk16*>(y + yrow[i] * PY + ycc[i] * VEC) = ry[i];
        MTN_UNROLL for (int i = 0; i < RX; ++i) *reinterpret_cast<chunk16*>(x + xrow[i] * PX + xcc[i] * VEC) = rx[i];
    };

    f32x4_t acc[FM][FN];
    MTN_UNROLL for (int i = 0; i < FM; ++i)
        MTN_UNROLL for (int j = 0; j < FN; ++j) acc[i][j] = zero4();

    // single LDS buffer + register prefetch (two barriers per step): ~35 KB of LDS per workgroup
    // -> 4 workgroups per CU, whose load / MFMA phases overlap each other
    if (steps > 0) load_tiles(0);
    for (int st = 0; st < steps; ++st) {
        store_tiles(0);
        __syncthreads();
        if (st + 1 < steps) load_tiles(st + 1);
        const T* y = sY + wm * WM;
        const T* x = sX + wn * WN;
        MTN_UNROLL for (int ks = 0; ks < BMR / KI; ++ks) {
            typename Mma<T>::Frag fa[FM], fb[FN];
            MTN_UNROLL for (int i = 0; i < FM; ++i) fa[i] = ColFrag<T, TR>::load(y + ks * KI * PY + i * 16, PY, lane);
            MTN_UNROLL for (int j = 0; j < FN; ++j) fb[j] = ColFrag<T, TR>::load(x + ks * KI * PX + j * 16, PX, lane);
            MTN_UNROLL for (int i = 0; i < FM; ++i)
                MTN_UNROLL for (int j = 0; j < FN; ++j) acc[i][j] = Mma<T>::mma(fa[i], fb[j], acc[i][j]);
        }
        __syncthreads();
    }

    // partial tile -> slab[split][n][k]   (row = n, 16 consecutive k per 16 lanes)
    float* __restrict__ out = p.partial + (long)split * p.N * p.K;
    MTN_UNROLL for (int i = 0; i < FM; ++i)
        MTN_UNROLL for (int j = 0; j < FN; ++j)
            MTN_UNROLL for (int r = 0; r < 4; ++r) {
                const int n = n0 + wm * WM + i * 16 + 4 * (lane >> 4) + r;
                const int k = kcol0 + wn * WN + j * 16 + (lane & 15);
                out[(long)n * p.K + k] = acc[i][j][r];
            }
}

// dW_oihw[o][ci][kh][kw] = sum_s partial[s][o][(kh*KW+kw)*Cin + ci]; block = 4 split lanes x 64 elements
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                           int splits, int N, int K, int Cin, int taps) {
    __shared__ float red[4][64];
    const int sl = threadIdx.x >> 6, el = threadIdx.x & 63;
    const long total = (long)N * K;
    const long idx = (long)blockIdx.x * 64 + el;
    float s = 0.f;
    if (idx < total)
        for (int q = sl; q < splits; q += 4) s += partial[(long)q * total + idx];
    red[sl][el] = s;
    __syncthreads();
    if (sl == 0 && idx < total) {
        const int n = (int)(idx / K), k = (int)(idx - (long)n * K);
        const int tap = k / Cin, ci = k - tap * Cin;
        dw[((long)n * Cin + ci) * taps + tap] = red[0][el] + red[1][el] + red[2][el] + red[3][el];
    }
}

template <class T, int BN, int BC, int WGM, int WGN, bool TR>
int launch_wgrad(const WgradArgs& p, hipStream_t stream) {
    constexpr int NT = WGM * WGN * 64;
    constexpr int BMR = 128 / (int)sizeof(T);
    constexpr size_t lds = (size_t)BMR * ((BN + BC) * sizeof(T) + 2 * PADB);
#ifndef MTN_EMU
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<T, BN, BC, WGM, WGN, TR>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
#endif
    const int tiles = (p.N / BN) * (p.K / BC);
    MTN_LAUNCH((wgrad_kernel<T, BN, BC, WGM, WGN, TR>), dim3(tiles * p.splits), dim3(NT), lds, stream, p);
    return mtn_check_launch("wgrad");
}

}  // namespace

static int g_use_tr16 = 1;
void mtn_wgrad_set_tr16(int on) { g_use_tr16 = on; }

// tile of dW owned by one workgroup: BN output channels x BC input channels (of one tap)
static void wgrad_tile(int N, int Cin, int* bn, int* bc) {
    *bn = N % 128 == 0 ? 128 : 64;
    *bc = Cin % 128 == 0 ? 128 : 64;
    if (*bn == 64 && *bc == 128) *bc = 64;          // (64,128) is not instantiated
}

// pixels per split: multiple of the step size, >= ~4096 rows, but enough splits to fill the chip
int mtn_wgrad_splits(int M, int N, int K, int Cin, int dtype, int* rows_per_split) {
    const int bmr = dtype == MTN_BF16 ? 64 : 32;
    int bn, bc;
    wgrad_tile(N, Cin, &bn, &bc);
    const long tiles = (long)(N / bn) * (K / bc);
    long want = (M + 4095) / 4096;
    const long fill = (512 + tiles - 1) / tiles;
    if (want < fill) want = fill;
    long rps = (M + want - 1) / want;
    rps = (rps + bmr - 1) / bmr * bmr;
    if (rps < bmr) rps = bmr;
    *rows_per_split = (int)rps;
    long splits = (M + rps - 1) / rps;
    if (splits >= 8) splits = (splits + 7) / 8 * 8;        // whole splits per XCD group (empty ones add zeros)
    return (int)splits;
}

template <class T, bool TR>
static int wgrad_dispatch(const WgradArgs& p, hipStream_t stream) {
    int bn, bc;
    wgrad_tile(p.N, p.Cin, &bn, &bc);
    if (bn == 128 && bc == 128) return launch_wgrad<T, 128, 128, 2, 2, TR>(p, stream);
    if (bn == 128) return launch_wgrad<T, 128, 64, 4, 1, TR>(p, stream);
    return launch_wgrad<T, 64, 64, 2, 2, TR>(p, stream);
}

int mtn_wgrad_run(const WgradArgs& p0, int dtype, hipStream_t stream) {
    WgradArgs p = p0;
    const int bmr = dtype == MTN_BF16 ? 64 : 32;
    MTN_REQUIRE(dtype == MTN_BF16 || dtype == MTN_F32, "wgrad: bad dtype %d", dtype);
    MTN_REQUIRE(p.N % 64 == 0 && p.Cin % 64 == 0, "wgrad: N=%d and Cin=%d must be multiples of 64", p.N, p.Cin);
    MTN_REQUIRE(p.K == p.KH * p.KW * p.Cin, "wgrad: K mismatch");
    MTN_REQUIRE(p.M == p.B * p.OH * p.OW, "wgrad: M mismatch");
    MTN_REQUIRE(p.splits >= 1 && p.rows_per_split % bmr == 0 && (long)p.splits * p.rows_per_split >= p.M,
                "wgrad: bad split (%d x %d rows for M=%d)", p.splits, p.rows_per_split, p.M);
    MTN_REQUIRE(((uintptr_t)p.dY | (uintptr_t)p.X) % 16 == 0, "wgrad: pointers must be 16-byte aligned");
    int rc;
    if (dtype == MTN_BF16) rc = g_use_tr16 ? wgrad_dispatch<bf16_t, true>(p, stream) : wgrad_dispatch<bf16_t, false>(p, stream);
    else rc = wgrad_dispatch<float, false>(p, stream);
    if (rc) return rc;
    const long total = (long)p.N * p.K;
    const int blocks = (int)((total + 63) / 64);
    MTN_LAUNCH(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, stream, p.partial, p.dW, p.splits, p.N, p.K,
               p.Cin, p.KH * p.KW);
    return mtn_check_launch("wgrad_reduce");
}
