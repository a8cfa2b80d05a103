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
#include <type_traits>
#include "mtn_types.h"
#include "mtn_kernels.h"

static int g_wgrad_tile = 0;     // debug key 7: 64 forces 64x64 tiles
static int g_wgrad_dma = 1;      // LDS-DMA pipeline (debug key 5 selects the register-staged one)
static int g_wgrad_slots = 0;    // debug key 10: workgroups per launch the pixel split aims at (0 = per-tile default)
static int g_x3_wgrad_sep = 0;   // debug key 20: 1 = separate LDS stages for the split-bf16 128x128 weight gradient, 2 = in every mode
                                 // (measured SLOWER: 22.1 vs 21.25 ms per bf16x3 step -- DESIGN.md 7d)

namespace {

// pixel walk of the X loader: (image, oy, ox) of a row, advanced by a fixed number of rows per K
// step with multiply-high divisions (exact while operand * divisor < 2^32, host-checked)
struct PixelPos { int b, oy, ox; };
__device__ __forceinline__ unsigned wg_div(unsigned x, int d, unsigned magic) { return d == 1 ? x : __umulhi(x, magic); }
__device__ __forceinline__ PixelPos pixel_of(long m, const WgradArgs& p) {
    const unsigned yt = wg_div((unsigned)m, p.OW, p.magic_ow);
    PixelPos q;
    q.ox = (int)((unsigned)m - yt * (unsigned)p.OW);
    q.b = (int)wg_div(yt, p.OH, p.magic_oh);
    q.oy = (int)yt - q.b * p.OH;
    return q;
}
__device__ __forceinline__ void pixel_advance(PixelPos& q, int rows, const WgradArgs& p) {
    q.ox += rows;
    const int cy = (int)wg_div((unsigned)q.ox, p.OW, p.magic_ow);
    q.ox -= cy * p.OW;
    q.oy += cy;
    const int cb = (int)wg_div((unsigned)q.oy, p.OH, p.magic_oh);
    q.oy -= cb * p.OH;
    q.b += cb;
}

// A partial tile of the split-K slab [split][N][K]: f32, or -- bf16 mode, debug key 28 -- bf16 (slab_bf16): half the slab's
// write and read-back bytes; the splits are still summed in f32 (wgrad_reduce_kernel).
__device__ __forceinline__ void slab_store(const WgradArgs& p, int split, int n, int k, const f32x4_t& v) {
    const long off = ((long)split * p.N + n) * p.K + k;
    if (p.slab_bf16) store4<bf16_t>(reinterpret_cast<bf16_t*>(p.partial) + off, v);
    else store4<float>(p.partial + off, v);
}

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
    const bool pointwise = p.KH * p.KW == 1 && p.stride == 1 && p.pad == 0;
    const long m_begin = (long)split * p.rows_per_split;
    long m_end = m_begin + p.rows_per_split;
    if (m_end > p.M) m_end = p.M;
    const int steps = (int)((m_end - m_begin + BMR - 1) / BMR);

    const T* __restrict__ gY = reinterpret_cast<const T*>(p.dY);
    const T* __restrict__ gX = reinterpret_cast<const T*>(p.X);

    // per-thread loader rows
    int yrow[RY], ycc[RY];
    MTN_UNROLL for (int i = 0; i < RY; ++i) { const int c = tid + i * NT; yrow[i] = c / CPRY; ycc[i] = c % CPRY; }
    int xrow[RX], xcc[RX];
    PixelPos xpos[RX];
    MTN_UNROLL for (int i = 0; i < RX; ++i) {
        const int c = tid + i * NT; xrow[i] = c / CPRX; xcc[i] = c % CPRX;
        xpos[i] = pointwise ? PixelPos{0, 0, 0} : pixel_of(m_begin + xrow[i], p);
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
            if (pointwise) {                                 // 1x1 taps at stride 1: input pixel m itself
                rx[i] = (m < m_end) ? *reinterpret_cast<const chunk16*>(gX + m * p.ldx + c0 + xcc[i] * VEC) : zero_chunk();
                continue;
            }
            const int iy = xpos[i].oy * p.stride - p.pad + kh, ix = xpos[i].ox * p.stride - p.pad + kw;
            const bool ok = (m < m_end) && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            rx[i] = ok ? *reinterpret_cast<const chunk16*>(gX + ((long)(xpos[i].b * p.H + iy) * p.W + ix) * p.ldx + c0 + xcc[i] * VEC)
                       : zero_chunk();
            pixel_advance(xpos[i], BMR, p);                  // this row at the next step
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
                MTN_UNROLL for (int j = 0; j < FN; ++j) acc[i][j] = Mma<T>::mma(fb[j], fa[i], acc[i][j]);
        }
        __syncthreads();
    }

    // partial tile -> slab[split][n][k]   (row = n, 16 consecutive k per 16 lanes)
    MTN_UNROLL for (int i = 0; i < FM; ++i)
        MTN_UNROLL for (int j = 0; j < FN; ++j)
            {   // the X fragment is the MFMA's row operand: a lane's quad is 4 consecutive k of one n
                const int n = n0 + wm * WM + i * 16 + (lane & 15);
                const int k = kcol0 + wn * WN + j * 16 + 4 * (lane >> 4);
                slab_store(p, split, n, k, acc[i][j]);
            }
}

// ---- LDS-DMA variant ------------------------------------------------------------------------------
// Same tiling, but the dY / X tiles go global -> LDS by buffer_load ... lds (16 B per lane, no
// staging VGPRs, no ds_write; out-of-range offsets deliver the zero padding) into two buffers.
// The LDS image is lane-linear, so the bank-conflict fix of the column (ds_read_b64_tr_b16) reads
// is an XOR of the 32-byte unit index applied to the lane's SOURCE chunk and to the read address
// (cdna_hip_programming.md rule 21): a half-wave's tr read touches rows {r..r+3, r+8..r+11} at one
// unit column, and f2(row) sends those 8 rows to 8 different bank ranges.
template <class T, int ROWE> struct Swz {
    static constexpr int EPU = 32 / (int)sizeof(T);              // elements per 32-byte unit
    static constexpr int ROWB = ROWE * (int)sizeof(T);
    static constexpr int UNITS = ROWB / 32;
    static_assert(ROWB == 128 || ROWB % 256 == 0, "row bytes");
    static __device__ __forceinline__ int f2(int row) {
        return ROWB == 128 ? (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) : ((row & 3) | (((row >> 3) & 1) << 2));
    }
    static __device__ __forceinline__ int off(int row, int col) {
        const int unit = col / EPU;
        return row * ROWE + ((unit ^ f2(row)) * EPU) + (col - unit * EPU);
    }
    static __device__ __forceinline__ int fetch_chunk(int row, int pos) { return (((pos >> 1) ^ f2(row)) << 1) | (pos & 1); }
};

// operand fragment read down the rows of a swizzled tile: rows k0.., column col0 + (lane & 15)
template <class T, int ROWE> struct ColFragS;
template <int ROWE> struct ColFragS<float, ROWE> {
    static __device__ __forceinline__ float load(const float* tile, int k0, int col0, int lane) {
        return tile[Swz<float, ROWE>::off(k0 + (lane >> 4), col0 + (lane & 15))];
    }
};
template <int ROWE> struct ColFragS<bf16_t, ROWE> {
    static __device__ __forceinline__ bf16x8_t load(const bf16_t* tile, int k0, int col0, int lane) {
        const int g = lane >> 4, i = lane & 15;
        const int r0 = k0 + 8 * g + (i >> 2), c = col0 + 4 * (i & 3);
        const bf16_t* p0 = tile + Swz<bf16_t, ROWE>::off(r0, c);
        const bf16_t* p1 = tile + Swz<bf16_t, ROWE>::off(r0 + 4, c);
        bf16x8_t f;
#ifdef MTN_EMU
        emu::u16x4 a = emu::ds_read_tr16_b64(p0), b = emu::ds_read_tr16_b64(p1);
        MTN_UNROLL for (int j = 0; j < 4; ++j) { f[j] = a[j]; f[4 + j] = b[j]; }
#else
        typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) bf16x4_t* lds_ptr;
        const bf16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(p0));
        const bf16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(p1));
        f = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
#endif
        return f;
    }
};

// Split storage: a tile row of ROWE channels is ROWE / 32 blocks [32 hi][32 lo] = a bf16 row of 2 ROWE columns in which channel
// c sits at column x3_pos(c) (hi) and x3_pos(c) + 32 (lo).  The DMA moved the bytes as they are and swizzled 32-byte units
// by row (Swz<f32x3_t, ROWE> and Swz<bf16_t, 2 ROWE> are the same byte map), so the hi and the lo fragment of 16 channels are
// the bf16 mode's transposing reads at two columns: lane group g gets pixels k0 + 8g .. 8g+7, no conversion.
template <int ROWE> struct ColFragS<f32x3_t, ROWE> {
    static __device__ __forceinline__ Frag3 load(const f32x3_t* tile, int k0, int col0, int lane) {
        const bf16_t* t = reinterpret_cast<const bf16_t*>(tile);
        const int hc = x3_pos(col0);
        Frag3 f;
        f.hi = ColFragS<bf16_t, 2 * ROWE>::load(t, k0, hc, lane);
        f.lo = ColFragS<bf16_t, 2 * ROWE>::load(t, k0, hc + 32, lane);
        return f;
    }
};

template <class T, int BN, int BC, int WGM, int WGN>
__global__ __launch_bounds__(WGM* WGN * 64) void wgrad_dma_kernel(WgradArgs p) {
    constexpr int NW = WGM * WGN;
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr int BMR = 128 / (int)sizeof(T);
    constexpr int KI = Mma<T>::KI;
    constexpr int CPRY = BN / VEC, CPRX = BC / VEC;                 // 16-byte slots per row
    constexpr int RPBY = 64 / CPRY, RPBX = 64 / CPRX;               // rows per 1 KiB DMA block
    constexpr int IY = BMR / RPBY / NW, IX = BMR / RPBX / NW;       // DMA instructions per wave per step
    constexpr int WM = BN / WGM, WN = BC / WGN;
    constexpr int FM = WM / 16, FN = WN / 16;
    static_assert(BMR % (RPBY * NW) == 0 && BMR % (RPBX * NW) == 0, "DMA mapping");

    // (One dynamic array with a runtime stage offset: hipcc then cannot tell the stage being filled from
    // the stage being read and waits vmcnt(0) before the first fragment read after every DMA issue, i.e.
    // a workgroup does not overlap its own DMA with its own MFMAs -- the 2..4 co-resident workgroups
    // overlap each other instead.  Two distinct __shared__ stages + a step loop unrolled by two remove
    // that wait (wgrad3_kernel below does it), but here it measured SLOWER on 10 of 13 layers (conv2
    // +24 %, down1/2 +12 %, cv4 +17 %), and even the unrolled loop alone costs the 128x64 tile 60
    // registers and a resident workgroup.)
    MTN_DYN_SMEM(smem);
    T* const sY = reinterpret_cast<T*>(smem);          // [2][BMR][BN]
    T* const sX = sY + 2 * BMR * BN;                   // [2][BMR][BC]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WGN, wn = wid % WGN;

    const int tiles_n = p.N / BN;
    const int tiles = tiles_n * (p.K / BC);
    int tile, split;
    {
        const int L = blockIdx.x, nwg = gridDim.x;
        if (p.splits % 8 == 0 && tiles <= 24) {
            const int xcd = L & 7, j = L >> 3;
            split = (j / tiles) * 8 + xcd;
            tile = j % tiles;
        } else {
            // many tiles per split: give each XCD group (id % 8) a contiguous range of the
            // (split, tile) order, so the tiles of one split -- which all stream the same rows of
            // dY and X -- share one L2 instead of spreading over all eight
            const int q = nwg >> 3, r = nwg & 7, xcd = L & 7;
            const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
            split = lin / tiles;
            tile = lin % tiles;
        }
    }
    const int tile_n = tile % tiles_n, tile_k = tile / tiles_n;
    const int n0 = tile_n * BN;
    const int kcol0 = tile_k * BC;
    const int tap = kcol0 / p.Cin, c0 = kcol0 - tap * p.Cin;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const bool pointwise = p.KH * p.KW == 1 && p.stride == 1 && p.pad == 0;
    const long m_begin = (long)split * p.rows_per_split;
    long m_end = m_begin + p.rows_per_split;
    if (m_end > p.M) m_end = p.M;
    const int steps = (int)((m_end - m_begin + BMR - 1) / BMR);

    const BufDesc bufY = make_buf(p.dY, p.y_bytes), bufX = make_buf(p.X, p.x_bytes);
    // this lane's row / source chunk in each of its DMA instructions
    int yrow[IY], ychunk[IY];
    MTN_UNROLL for (int i = 0; i < IY; ++i) {
        yrow[i] = (wid * IY + i) * RPBY + lane / CPRY;
        ychunk[i] = Swz<T, BN>::fetch_chunk(yrow[i], lane % CPRY);
    }
    int xrow[IX], xchunk[IX];
    PixelPos xpos[IX];
    MTN_UNROLL for (int i = 0; i < IX; ++i) {
        xrow[i] = (wid * IX + i) * RPBX + lane / CPRX;
        xchunk[i] = Swz<T, BC>::fetch_chunk(xrow[i], lane % CPRX);
        xpos[i] = pointwise ? PixelPos{0, 0, 0} : pixel_of(m_begin + xrow[i], p);
    }
    auto dma_tiles = [&](int st, int stage) {
        const long mb = m_begin + (long)st * BMR;
        char* y = reinterpret_cast<char*>(sY + stage * BMR * BN);
        char* x = reinterpret_cast<char*>(sX + stage * BMR * BC);
        MTN_UNROLL for (int i = 0; i < IY; ++i) {
            const long m = mb + yrow[i];
            const unsigned off = (unsigned)(m * p.lddy + n0 + ychunk[i] * VEC) * (unsigned)sizeof(T);
            buf_glds16(bufY, m < m_end ? off : 0xFFFFFFF0u, y + (wid * IY + i) * 1024);
        }
        MTN_UNROLL for (int i = 0; i < IX; ++i) {
            const long m = mb + xrow[i];
            if (pointwise) {                                 // 1x1 taps at stride 1: input pixel m itself
                const unsigned off = (unsigned)(m * p.ldx + c0 + xchunk[i] * VEC) * (unsigned)sizeof(T);
                buf_glds16(bufX, m < m_end ? off : 0xFFFFFFF0u, x + (wid * IX + i) * 1024);
                continue;
            }
            const int iy = xpos[i].oy * p.stride - p.pad + kh, ix = xpos[i].ox * p.stride - p.pad + kw;
            const bool ok = (m < m_end) && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            const unsigned off = (unsigned)(((xpos[i].b * p.H + iy) * p.W + ix) * p.ldx + c0 + xchunk[i] * VEC) * (unsigned)sizeof(T);
            buf_glds16(bufX, ok ? off : 0xFFFFFFF0u, x + (wid * IX + i) * 1024);
            pixel_advance(xpos[i], BMR, p);
        }
    };

    f32x4_t acc[FM][FN];
    MTN_UNROLL for (int i = 0; i < FM; ++i)
        MTN_UNROLL for (int j = 0; j < FN; ++j) acc[i][j] = zero4();

    if (steps > 0) dma_tiles(0, 0);
    __syncthreads();
    for (int st = 0; st < steps; ++st) {
        const int cur = st & 1;
        if (st + 1 < steps) dma_tiles(st + 1, cur ^ 1);
        const T* y = sY + cur * BMR * BN;
        const T* x = sX + cur * BMR * BC;
        MTN_UNROLL for (int ks = 0; ks < BMR / KI; ++ks) {
            typename Mma<T>::Frag fa[FM], fb[FN];
            MTN_UNROLL for (int i = 0; i < FM; ++i) fa[i] = ColFragS<T, BN>::load(y, ks * KI, wm * WM + i * 16, lane);
            MTN_UNROLL for (int j = 0; j < FN; ++j) fb[j] = ColFragS<T, BC>::load(x, ks * KI, wn * WN + j * 16, lane);
            MTN_UNROLL for (int i = 0; i < FM; ++i)
                MTN_UNROLL for (int j = 0; j < FN; ++j) acc[i][j] = Mma<T>::mma(fb[j], fa[i], acc[i][j]);
        }
        __syncthreads();
    }
    MTN_UNROLL for (int i = 0; i < FM; ++i)
        MTN_UNROLL for (int j = 0; j < FN; ++j)
            {   // the X fragment is the MFMA's row operand: a lane's quad is 4 consecutive k of one n
                const int n = n0 + wm * WM + i * 16 + (lane & 15);
                const int k = kcol0 + wn * WN + j * 16 + 4 * (lane >> 4);
                slab_store(p, split, n, k, acc[i][j]);
            }
}

// The same kernel with its step loop unrolled by two (each copy names its read / fill stage by a
// pointer fixed outside the loop): measured 5 % faster on the 8-wave 128x128 tile, and slower on the
// 128x64 tile (the second copy of the loader state costs it 50 registers and a resident workgroup),
// so only the former uses it.
// SEP: the two stages are two DISTINCT __shared__ objects: hipcc then knows that the LDS-DMA of step st+1 cannot alias the
// fragment reads of step st and drops the `s_waitcnt vmcnt(0)` it otherwise puts in front of the first fragment read (seen
// in the ISA of the one-array form: a workgroup then never overlaps its own DMA with its own MFMAs).  That overlap measured
// slower in bf16 (round 2) AND in the split-bf16 mode (round 3: 22.1 vs 21.25 ms per step): the co-resident workgroups
// already cover each other, and DMA in flight contends with the fragment reads.  Kept behind debug key 20.
// (the body takes the workgroup's index in ITS problem's grid and that grid's size: wgrad_dma2_group_kernel below runs several
//  problems in one launch)
template <class T, int BN, int BC, int WGM, int WGN, bool SEP>
__device__ __forceinline__ void wgrad_dma2_body(const WgradArgs& p, const int L, const int nwg) {
    constexpr int NW = WGM * WGN;
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr int BMR = 128 / (int)sizeof(T);
    constexpr int KI = Mma<T>::KI;
    constexpr int CPRY = BN / VEC, CPRX = BC / VEC;                 // 16-byte slots per row
    constexpr int RPBY = 64 / CPRY, RPBX = 64 / CPRX;               // rows per 1 KiB DMA block
    constexpr int IY = BMR / RPBY / NW, IX = BMR / RPBX / NW;       // DMA instructions per wave per step
    constexpr int WM = BN / WGM, WN = BC / WGN;
    constexpr int FM = WM / 16, FN = WN / 16;
    static_assert(BMR % (RPBY * NW) == 0 && BMR % (RPBX * NW) == 0, "DMA mapping");

    constexpr int STAGE = 128 / (int)sizeof(T) * (BN + BC);       // elements of one stage: [BMR][BN] dY then [BMR][BC] X
    __shared__ __attribute__((aligned(16))) T sSep0[SEP ? STAGE : 1];
    __shared__ __attribute__((aligned(16))) T sSep1[SEP ? STAGE : 1];
    MTN_DYN_SMEM(smem);
    T* const sStage0 = SEP ? sSep0 : reinterpret_cast<T*>(smem);
    T* const sStage1 = SEP ? sSep1 : sStage0 + STAGE;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WGN, wn = wid % WGN;

    const int tiles_n = p.N / BN;
    const int tiles = tiles_n * (p.K / BC);
    int tile, split;
    {
        if (p.splits % 8 == 0 && tiles <= 24) {
            const int xcd = L & 7, j = L >> 3;
            split = (j / tiles) * 8 + xcd;
            tile = j % tiles;
        } else {
            // many tiles per split: give each XCD group (id % 8) a contiguous range of the
            // (split, tile) order, so the tiles of one split -- which all stream the same rows of
            // dY and X -- share one L2 instead of spreading over all eight
            const int q = nwg >> 3, r = nwg & 7, xcd = L & 7;
            const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
            split = lin / tiles;
            tile = lin % tiles;
        }
    }
    const int tile_n = tile % tiles_n, tile_k = tile / tiles_n;
    const int n0 = tile_n * BN;
    const int kcol0 = tile_k * BC;
    const int tap = kcol0 / p.Cin, c0 = kcol0 - tap * p.Cin;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const bool pointwise = p.KH * p.KW == 1 && p.stride == 1 && p.pad == 0;
    const long m_begin = (long)split * p.rows_per_split;
    long m_end = m_begin + p.rows_per_split;
    if (m_end > p.M) m_end = p.M;
    const int steps = (int)((m_end - m_begin + BMR - 1) / BMR);

    const BufDesc bufY = make_buf(p.dY, p.y_bytes), bufX = make_buf(p.X, p.x_bytes);
    // this lane's row / source chunk in each of its DMA instructions
    int yrow[IY], ychunk[IY];
    MTN_UNROLL for (int i = 0; i < IY; ++i) {
        yrow[i] = (wid * IY + i) * RPBY + lane / CPRY;
        ychunk[i] = Swz<T, BN>::fetch_chunk(yrow[i], lane % CPRY);
    }
    int xrow[IX], xchunk[IX];
    PixelPos xpos[IX];
    MTN_UNROLL for (int i = 0; i < IX; ++i) {
        xrow[i] = (wid * IX + i) * RPBX + lane / CPRX;
        xchunk[i] = Swz<T, BC>::fetch_chunk(xrow[i], lane % CPRX);
        xpos[i] = pointwise ? PixelPos{0, 0, 0} : pixel_of(m_begin + xrow[i], p);
    }
    auto dma_tiles = [&](int st, T* stage) {
        const long mb = m_begin + (long)st * BMR;
        char* y = reinterpret_cast<char*>(stage);
        char* x = reinterpret_cast<char*>(stage + BMR * BN);
        MTN_UNROLL for (int i = 0; i < IY; ++i) {
            const long m = mb + yrow[i];
            const unsigned off = (unsigned)(m * p.lddy + n0 + ychunk[i] * VEC) * (unsigned)sizeof(T);
            buf_glds16(bufY, m < m_end ? off : 0xFFFFFFF0u, y + (wid * IY + i) * 1024);
        }
        MTN_UNROLL for (int i = 0; i < IX; ++i) {
            const long m = mb + xrow[i];
            if (pointwise) {                                 // 1x1 taps at stride 1: input pixel m itself
                const unsigned off = (unsigned)(m * p.ldx + c0 + xchunk[i] * VEC) * (unsigned)sizeof(T);
                buf_glds16(bufX, m < m_end ? off : 0xFFFFFFF0u, x + (wid * IX + i) * 1024);
                continue;
            }
            const int iy = xpos[i].oy * p.stride - p.pad + kh, ix = xpos[i].ox * p.stride - p.pad + kw;
            const bool ok = (m < m_end) && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            const unsigned off = (unsigned)(((xpos[i].b * p.H + iy) * p.W + ix) * p.ldx + c0 + xchunk[i] * VEC) * (unsigned)sizeof(T);
            buf_glds16(bufX, ok ? off : 0xFFFFFFF0u, x + (wid * IX + i) * 1024);
            pixel_advance(xpos[i], BMR, p);
        }
    };

    f32x4_t acc[FM][FN];
    MTN_UNROLL for (int i = 0; i < FM; ++i)
        MTN_UNROLL for (int j = 0; j < FN; ++j) acc[i][j] = zero4();

    auto step = [&](int st, const T* rd, T* wr) {
        if (st + 1 < steps) dma_tiles(st + 1, wr);
        const T* y = rd;
        const T* x = rd + BMR * BN;
        MTN_UNROLL for (int ks = 0; ks < BMR / KI; ++ks) {
            typename Mma<T>::Frag fa[FM], fb[FN];
            MTN_UNROLL for (int i = 0; i < FM; ++i) fa[i] = ColFragS<T, BN>::load(y, ks * KI, wm * WM + i * 16, lane);
            MTN_UNROLL for (int j = 0; j < FN; ++j) fb[j] = ColFragS<T, BC>::load(x, ks * KI, wn * WN + j * 16, lane);
            MTN_UNROLL for (int i = 0; i < FM; ++i)
                MTN_UNROLL for (int j = 0; j < FN; ++j) acc[i][j] = Mma<T>::mma(fb[j], fa[i], acc[i][j]);
        }
        __syncthreads();
    };
    if (steps > 0) dma_tiles(0, sStage0);
    __syncthreads();
    for (int st = 0; st < steps; st += 2) {
        step(st, sStage0, sStage1);
        if (st + 1 < steps) step(st + 1, sStage1, sStage0);
    }
    MTN_UNROLL for (int i = 0; i < FM; ++i)
        MTN_UNROLL for (int j = 0; j < FN; ++j)
            {   // the X fragment is the MFMA's row operand: a lane's quad is 4 consecutive k of one n
                const int n = n0 + wm * WM + i * 16 + (lane & 15);
                const int k = kcol0 + wn * WN + j * 16 + 4 * (lane >> 4);
                slab_store(p, split, n, k, acc[i][j]);
            }
}

template <class T, int BN, int BC, int WGM, int WGN, bool SEP = false>
__global__ __launch_bounds__(WGM* WGN * 64) void wgrad_dma2_kernel(WgradArgs p) {
    wgrad_dma2_body<T, BN, BC, WGM, WGN, SEP>(p, (int)blockIdx.x, (int)gridDim.x);
}

// ---- several pointwise problems in ONE launch (round 5) ----------------------------------------------------------------
// The sixteen Linear weight gradients of the ViT are 37,120-row reductions: as sixteen launches each fills the chip's 512
// workgroup slots with 40 .. 120 splits of ~10 K-steps -- prologue, ring fill and a 64 KB partial tile per workgroup for
// ten steps of work (410 TFLOP/s, and 31 MB of slab per launch written and read back) where the same kernel reaches
// 650 TFLOP/s on a long reduction axis (tools/probes/linear_wgrad_scaling.py).  Nothing downstream needs a weight gradient
// before the optimizer, so the plan defers them and runs all of them here: workgroup L belongs to the problem whose block
// range [first[i], first[i+1]) holds it (ranges are multiples of 8, so L & 7 is still the XCD inside the problem's grid);
// 128 tiles x 4 splits of 145 steps each, 33 MB of slab in all, one grouped reduce.
template <class T>
__global__ __launch_bounds__(512) void wgrad_dma2_group_kernel(WgradGroupArgs g) {
    const int L = (int)blockIdx.x;
    int prob = 0;
    for (int i = 1; i < g.count; ++i) prob += L >= g.first[i] ? 1 : 0;
    const WgradArgs p = g.item[prob];
    const int l = L - g.first[prob];
    const int nwg = (p.N / 128) * (p.K / 128) * p.splits;      // (the range is padded to a multiple of 8 workgroups)
    if (l >= nwg) return;
    wgrad_dma2_body<T, 128, 128, 4, 2, false>(p, l, nwg);
}


// ---- 3x3 taps, any stride, 64-channel input slabs: the three taps of ONE kernel row per workgroup --------------
// wgrad_dma_kernel gives every tap its own workgroup, so a layer with Cin = 64 (conv2: 128 x 64 tiles, nine
// of them per split) streams the same dY rows nine times through L2 -> LDS and spends two 8-row gather
// instructions per wave and step on 16 MFMAs (the gathered X tile costs 2.5x the dY
// tile).  Here a workgroup owns the taps (kh, 0..2) of one 64-channel slab: ONE dY tile per step feeds three
// X tiles -- adjacent pixels of the same input rows -- and 48 MFMAs per wave; a third of the workgroups,
// each with three times the reduction work per DMA instruction.  128 x (3 x 64) output tile, 4 waves
// (32 dY columns each, all 192 X columns), two stages of [64 rows][128 + 192 columns] = 80 KB: 2 workgroups / CU.
template <class T>
__global__ __launch_bounds__(256) void wgrad_row3_kernel(WgradArgs p) {
    constexpr int BN = 128, BC = 64, NW = 4, TAPS = 3;
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr int BMR = 128 / (int)sizeof(T);
    constexpr int KI = Mma<T>::KI;
    constexpr int CPRY = BN / VEC, CPRX = BC / VEC;
    constexpr int RPBY = 64 / CPRY, RPBX = 64 / CPRX;
    constexpr int IY = BMR / RPBY / NW, IX = BMR / RPBX / NW;
    constexpr int WM = BN / NW, FM = WM / 16, FN = BC / 16;
    static_assert(BMR % (RPBY * NW) == 0 && BMR % (RPBX * NW) == 0, "DMA mapping");

    MTN_DYN_SMEM(smem);
    T* const sY = reinterpret_cast<T*>(smem);          // [2][BMR][BN]
    T* const sX = sY + 2 * BMR * BN;                   // [2][TAPS][BMR][BC]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;

    const int tiles_n = p.N / BN, slabs = p.Cin / BC;
    const int tiles = tiles_n * p.KH * slabs;
    int tile, split;
    {   // (split, tile) order as in wgrad_dma_kernel: the tiles of one split share an XCD's L2
        const int L = blockIdx.x;
        if (p.splits % 8 == 0 && tiles <= 24) {
            const int xcd = L & 7, j = L >> 3;
            split = (j / tiles) * 8 + xcd;
            tile = j % tiles;
        } else {
            const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = L & 7;
            const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
            split = lin / tiles;
            tile = lin % tiles;
        }
    }
    const int tile_n = tile % tiles_n, tk = tile / tiles_n;
    const int kh = tk / slabs, c0 = (tk - kh * slabs) * BC;
    const int n0 = tile_n * BN;
    const long m_begin = (long)split * p.rows_per_split;
    long m_end = m_begin + p.rows_per_split;
    if (m_end > p.M) m_end = p.M;
    const int steps = (int)((m_end - m_begin + BMR - 1) / BMR);

    const BufDesc bufY = make_buf(p.dY, p.y_bytes), bufX = make_buf(p.X, p.x_bytes);
    int yrow[IY], ychunk[IY];
    MTN_UNROLL for (int i = 0; i < IY; ++i) {
        yrow[i] = (wid * IY + i) * RPBY + lane / CPRY;
        ychunk[i] = Swz<T, BN>::fetch_chunk(yrow[i], lane % CPRY);
    }
    int xrow[IX], xchunk[IX];
    PixelPos xpos[IX];
    MTN_UNROLL for (int i = 0; i < IX; ++i) {
        xrow[i] = (wid * IX + i) * RPBX + lane / CPRX;
        xchunk[i] = Swz<T, BC>::fetch_chunk(xrow[i], lane % CPRX);
        xpos[i] = pixel_of(m_begin + xrow[i], p);
    }
    auto dma_tiles = [&](int st, int stage) {
        const long mb = m_begin + (long)st * BMR;
        char* y = reinterpret_cast<char*>(sY + stage * BMR * BN);
        MTN_UNROLL for (int i = 0; i < IY; ++i) {
            const long m = mb + yrow[i];
            const unsigned off = (unsigned)(m * p.lddy + n0 + ychunk[i] * VEC) * (unsigned)sizeof(T);
            buf_glds16(bufY, m < m_end ? off : 0xFFFFFFF0u, y + (wid * IY + i) * 1024);
        }
        MTN_UNROLL for (int i = 0; i < IX; ++i) {
            const long m = mb + xrow[i];
            const int iy = xpos[i].oy * p.stride - p.pad + kh, ix0 = xpos[i].ox * p.stride - p.pad;
            const bool oky = (m < m_end) && iy >= 0 && iy < p.H;
            const unsigned base = (unsigned)(((xpos[i].b * p.H + iy) * p.W + ix0) * p.ldx + c0 + xchunk[i] * VEC) * (unsigned)sizeof(T);
            MTN_UNROLL for (int t = 0; t < TAPS; ++t) {
                const int ix = ix0 + t;
                char* x = reinterpret_cast<char*>(sX + (stage * TAPS + t) * BMR * BC);
                buf_glds16(bufX, (oky && ix >= 0 && ix < p.W) ? base + (unsigned)(t * p.ldx) * (unsigned)sizeof(T) : 0xFFFFFFF0u,
                           x + (wid * IX + i) * 1024);
            }
            pixel_advance(xpos[i], BMR, p);
        }
    };

    f32x4_t acc[TAPS][FM][FN];
    MTN_UNROLL for (int t = 0; t < TAPS; ++t)
        MTN_UNROLL for (int i = 0; i < FM; ++i)
            MTN_UNROLL for (int j = 0; j < FN; ++j) acc[t][i][j] = zero4();

    if (steps > 0) dma_tiles(0, 0);
    __syncthreads();
    for (int st = 0; st < steps; ++st) {
        const int cur = st & 1;
        if (st + 1 < steps) dma_tiles(st + 1, cur ^ 1);
        const T* y = sY + cur * BMR * BN;
        MTN_UNROLL for (int ks = 0; ks < BMR / KI; ++ks) {
            typename Mma<T>::Frag fa[FM];
            MTN_UNROLL for (int i = 0; i < FM; ++i) fa[i] = ColFragS<T, BN>::load(y, ks * KI, wid * WM + i * 16, lane);
            MTN_UNROLL for (int t = 0; t < TAPS; ++t) {
                const T* x = sX + (cur * TAPS + t) * BMR * BC;
                typename Mma<T>::Frag fb[FN];
                MTN_UNROLL for (int j = 0; j < FN; ++j) fb[j] = ColFragS<T, BC>::load(x, ks * KI, j * 16, lane);
                MTN_UNROLL for (int i = 0; i < FM; ++i)
                    MTN_UNROLL for (int j = 0; j < FN; ++j) acc[t][i][j] = Mma<T>::mma(fb[j], fa[i], acc[t][i][j]);
            }
        }
        __syncthreads();
    }
    MTN_UNROLL for (int t = 0; t < TAPS; ++t)
        MTN_UNROLL for (int i = 0; i < FM; ++i)
            MTN_UNROLL for (int j = 0; j < FN; ++j) {
                const int n = n0 + wid * WM + i * 16 + (lane & 15);
                const int k = (kh * p.KW + t) * p.Cin + c0 + j * 16 + 4 * (lane >> 4);
                slab_store(p, split, n, k, acc[t][i][j]);
            }
}

// ---- 3x3, stride 1: all nine taps from ONE window of the input -------------------------------------
// The tiles above re-stream dY and the tap-shifted X once per tap: nine times the bytes through the
// L2 -> LDS path for the same FLOPs, and that path (64 B/clk/CU) is what bounds them.  Here a
// workgroup owns dW[64 n][9 taps][64 ci] (36 accumulator fragments per wave) and walks the pixel axis
// in PADDED SLOT coordinates: every image row gets one extra (zero) slot and every image one extra
// (zero) row, so that
//     slot s = (b*(H+1) + y)*(W+1) + x,      tap (kh, kw) of slot s reads slot s + (kh-1)*(W+1) + (kw-1)
// and a tap that leaves the image lands on a zero slot: no masks, no per-tap validity arithmetic --
// the zero slots are produced by the LDS-DMA's out-of-range zero fill, for dY as well (a zero dY slot
// times anything is nothing).  Per step a workgroup stages KS slots of dY[:, n-tile] and the
// KS + 2(W+2) slots of X[:, ci-tile] around them (lane-linear, swizzled on the source chunk like the
// tiles above) and every tap reads its operand fragments down the slot axis at a ROW OFFSET of that
// one window: the offsets (swizzle included) are nine constants per lane, the K-sub-step is an
// immediate.  Bytes through the DMA path per FLOP: 1/3.3 (W = 48) .. 1/4.5 (W = 12) of the per-tap
// tiles; cost: (W+1)(H+1)/(WH) - 1 = 4 % (48^2) .. 17 % (12^2) of MFMA work on zero slots.
template <class T> struct Wg3 {
    static constexpr int KS = sizeof(T) == 2 ? 128 : 64;         // slots per step
    static constexpr int VEC = 16 / (int)sizeof(T);
    static constexpr int CPR = 64 / VEC;                          // 16-byte slots per 64-channel row
    static constexpr int RPB = 64 / CPR;                          // rows per 1 KiB DMA piece (8 bf16 / 4 f32)
    static constexpr int MAXW = sizeof(T) == 2 ? 96 : 48;         // widest feature map whose two window stages fit 160 KB
    static constexpr int STAGE = (2 * KS + 2 * (MAXW + 2) + RPB) * 64;   // elements: dY [KS][64] + window [KS + 2(W+2)][64]
};

// 8 waves: wave = (input-channel fragment w & 3, half of the 64 output channels w >> 2), 18 accumulator
// fragments each; ONE workgroup per CU with two LDS stages: the DMA of step t+1 is issued piece by
// piece between the K-sub-steps of step t.  (One workgroup per CU, not two: every workgroup writes its
// 64 x 576 f32 partial tile to the slab, so the slab -- written here, read back by wgrad_reduce -- is
// 147 KB x workgroups; at 512 workgroups that was 75 MB per layer, as many bytes as the operands.)
// KSX / WMAX (round 5): slots per step and the widest map the two stages are sized for -- 256 slots per step for bf16 maps up to
// 48 wide (2 x 79 KB of stages): halo 1.39x instead of 1.78x of the step's slots, half the barriers and DMA batches per slot
// (68-72 us per layer instead of 72-79, step -0.02 ms: profiles/r05_misc_ab.txt)
// (the body takes the workgroup's index in its problem's grid and that grid's size: wgrad3_group_kernel runs several problems
//  in one launch)
template <class T, int KSX, int WMAX>
__device__ __forceinline__ void wgrad3_body(const WgradArgs& p, const int L, const int nwg) {
    constexpr int KS = KSX, VEC = Wg3<T>::VEC, CPR = Wg3<T>::CPR, RPB = Wg3<T>::RPB;
    constexpr int STG = (2 * KS + 2 * (WMAX + 2) + RPB) * 64;
    constexpr int KI = Mma<T>::KI;
    constexpr int SUB = 32;                                       // slots per K-sub-step of the MFMA loop
    constexpr int NSUB = KS / SUB;
    constexpr int NW = 8;
    typedef Swz<T, 64> SW;
    // The two LDS stages are two DISTINCT __shared__ objects, and the step loop below is unrolled by
    // two so that every LDS-DMA names one object and every fragment read the other.  With one array
    // and a runtime stage offset hipcc cannot tell them apart and puts `s_waitcnt vmcnt(0)` in front
    // of the first fragment read after every DMA issue (an LDS-DMA is an LDS write to the compiler):
    // the prefetch then hides nothing (cdna_hip_programming.md section 5, trap 4a).
    __shared__ __attribute__((aligned(16))) T sStage0[STG];
    __shared__ __attribute__((aligned(16))) T sStage1[STG];
    const int W1 = p.W + 1, H1 = p.H + 1, HALO = W1 + 1;
    const int WR = (KS + 2 * HALO + RPB - 1) / RPB * RPB;         // window rows, whole DMA pieces
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int cf = wid & 3, nh = wid >> 2;

    // (tile, split): the tiles of one split stream the same rows -> keep them on one XCD group
    const int tiles_n = p.N / 64, tiles = tiles_n * (p.Cin / 64);
    int tile, split;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = L & 7;
        const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
        split = lin / tiles;
        tile = lin - split * tiles;
    }
    const int n0 = (tile % tiles_n) * 64, c0 = (tile / tiles_n) * 64;
    const unsigned S = (unsigned)p.B * H1 * W1;                   // padded slots
    const unsigned s_begin = (unsigned)split * p.rows_per_split;
    unsigned s_end = s_begin + p.rows_per_split;
    if (s_end > S) s_end = S;
    const int steps = s_begin < s_end ? (int)((s_end - s_begin + KS - 1) / KS) : 0;

    const BufDesc bufY = make_buf(p.dY, p.y_bytes), bufX = make_buf(p.X, p.x_bytes);
    // byte offset of (slot s, 16-byte chunk) in a pixel-major tensor, or the zero-fill offset
    auto src = [&](unsigned s, int ld, int col0, int chunk) -> unsigned {
        const unsigned R = __umulhi(s, p.magic_ow), col = s - R * (unsigned)W1;       // s / W1
        const unsigned b = __umulhi(R, p.magic_oh), y = R - b * (unsigned)H1;         // R / H1
        const unsigned px = (b * (unsigned)p.H + y) * (unsigned)p.W + col;
        const bool ok = s < S && col < (unsigned)p.W && y < (unsigned)p.H;
        return ok ? (px * (unsigned)ld + (unsigned)(col0 + chunk * VEC)) * (unsigned)sizeof(T) : 0xFFFFFFF0u;
    };
    const int prow = lane / CPR, pslot = lane % CPR;              // this lane's row / 16-byte slot inside a DMA piece
    const int PY = KS / RPB, PT = PY + WR / RPB;                  // pieces of the dY tile / of the whole stage
    // piece q of the stage that starts at slot s0 (q < PY: dY rows, else window rows)
    auto dma_piece = [&](int q, unsigned s0, T* stage) {
        char* base = reinterpret_cast<char*>(stage);
        if (q < PY) {
            const int row = q * RPB + prow;
            buf_glds16(bufY, src(s0 + row, p.lddy, n0, SW::fetch_chunk(row, pslot)), base + q * 1024);
        } else if (q < PT) {
            const int row = (q - PY) * RPB + prow;
            buf_glds16(bufX, src(s0 - HALO + row, p.ldx, c0, SW::fetch_chunk(row, pslot)), base + q * 1024);
        }
    };

    // accumulators: [tap][n fragment of this wave's half]
    f32x4_t acc[9][2];
    MTN_UNROLL for (int t = 0; t < 9; ++t) { acc[t][0] = zero4(); acc[t][1] = zero4(); }
    // operand fragment offsets (elements) at K-sub-step 0: rows advance by SUB per sub-step, which
    // leaves the swizzle (a function of row bits 0..3) unchanged -> the sub-step is an address immediate
    // (split storage: offsets in bf16 units of the SAME byte image -- a 64-channel row is 128 bf16 columns, channel c's hi half
    //  at column x3_pos(c); the lo half sits two 32-byte units further, which under the XOR swizzle of the unit index is the
    //  hi offset ^ 32: the unit index of a hi half has bit 1 clear)
    constexpr bool X3 = std::is_same<T, f32x3_t>::value;
    constexpr int RE = X3 ? 128 : 64;                             // bf16 columns of a tile row
    typedef Swz<bf16_t, RE> SWB;
    int yoff[2][2], xoff[9][2];
    if (sizeof(T) == 2 || X3) {
        const int g = lane >> 4, i16 = lane & 15;
        const int r0 = 8 * g + (i16 >> 2), cc = 4 * (i16 & 3);
        MTN_UNROLL for (int i = 0; i < 2; ++i) {
            const int col = X3 ? x3_pos((nh * 2 + i) * 16) : (nh * 2 + i) * 16;
            yoff[i][0] = SWB::off(r0, col + cc);
            yoff[i][1] = SWB::off(r0 + 4, col + cc);
        }
        MTN_UNROLL for (int t = 0; t < 9; ++t) {
            const int sh = HALO + (t / 3 - 1) * W1 + (t % 3 - 1);
            const int col = X3 ? x3_pos(cf * 16) : cf * 16;
            xoff[t][0] = KS * RE + SWB::off(r0 + sh, col + cc);
            xoff[t][1] = KS * RE + SWB::off(r0 + 4 + sh, col + cc);
        }
    }
    auto tr_pair = [&](const bf16_t* q0, const bf16_t* q1) -> bf16x8_t {
        bf16x8_t f;
#ifdef MTN_EMU
        emu::u16x4 a = emu::ds_read_tr16_b64(q0), b = emu::ds_read_tr16_b64(q1);
        MTN_UNROLL for (int j = 0; j < 4; ++j) { f[j] = a[j]; f[4 + j] = b[j]; }
#else
        typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) bf16x4_t* lds_ptr;
        const bf16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(q0));
        const bf16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(q1));
        f = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
#endif
        return f;
    };
    auto frag = [&](const T* tile, int e0, int e1, int row, int col) -> typename Mma<T>::Frag {
        if constexpr (X3) {
            (void)row; (void)col;
            const bf16_t* t = reinterpret_cast<const bf16_t*>(tile);
            Frag3 f;
            f.hi = tr_pair(t + e0, t + e1);
            f.lo = tr_pair(t + (e0 ^ 32), t + (e1 ^ 32));
            return f;
        } else if constexpr (sizeof(T) == 2) {
            (void)row; (void)col;
            const bf16_t* q0 = reinterpret_cast<const bf16_t*>(tile) + e0;
            const bf16_t* q1 = reinterpret_cast<const bf16_t*>(tile) + e1;
            bf16x8_t f;
#ifdef MTN_EMU
            emu::u16x4 a = emu::ds_read_tr16_b64(q0), b = emu::ds_read_tr16_b64(q1);
            MTN_UNROLL for (int j = 0; j < 4; ++j) { f[j] = a[j]; f[4 + j] = b[j]; }
#else
            typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(3))) bf16x4_t* lds_ptr;
            const bf16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(q0));
            const bf16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(q1));
            f = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
#endif
            return f;
        } else {
            (void)e0; (void)e1;
            return ColFragS<T, 64>::load(tile, row, col, lane);
        }
    };

    // one step: multiply the stage `rd` while the DMA of step st+1 lands in `wr`
    auto step = [&](int st, const T* rd, T* wr) {
        if (st + 1 < steps) {
            const unsigned s_next = s_begin + (unsigned)(st + 1) * KS;
            for (int q = wid; q < PT; q += NW) dma_piece(q, s_next, wr);
        }
        if constexpr (sizeof(T) == 2) {
            // fragments of sub-step j+1 are read while the MFMAs of sub-step j issue (two register sets)
            typename Mma<T>::Frag fy[2][2], fx[2][9];
            auto load_sub = [&](int j, int set) {
                MTN_UNROLL for (int i = 0; i < 2; ++i) fy[set][i] = frag(rd, yoff[i][0] + j * SUB * 64, yoff[i][1] + j * SUB * 64, 0, 0);
                MTN_UNROLL for (int t = 0; t < 9; ++t) fx[set][t] = frag(rd, xoff[t][0] + j * SUB * 64, xoff[t][1] + j * SUB * 64, 0, 0);
            };
            load_sub(0, 0);
            MTN_UNROLL for (int j = 0; j < NSUB; ++j) {
                if (j + 1 < NSUB) load_sub(j + 1, (j + 1) & 1);
                MTN_SCHED_FENCE();
                MTN_UNROLL for (int t = 0; t < 9; ++t)
                    MTN_UNROLL for (int i = 0; i < 2; ++i) acc[t][i] = Mma<T>::mma(fx[j & 1][t], fy[j & 1][i], acc[t][i]);
                MTN_SCHED_FENCE();
            }
        } else if constexpr (X3) {
            // one register set (a Frag3 is 8 registers: two sets of 11 would not leave room for the 18 accumulator quads)
            // (the sub-step loop is unrolled by two at most: with four unrolled sub-steps -- 128 slots per step -- hipcc hoists
            //  fragment reads across them and spills 17 registers into the loop)
            constexpr int UJ = NSUB > 2 ? 2 : NSUB;
            _Pragma("clang loop unroll(disable)") for (int jj = 0; jj < NSUB / UJ; ++jj) {
                const T* rj = rd + jj * (UJ * SUB * 64);
                MTN_UNROLL for (int j = 0; j < UJ; ++j) {
                    Frag3 fy[2];
                    MTN_UNROLL for (int i = 0; i < 2; ++i) fy[i] = frag(rj, yoff[i][0] + j * SUB * RE, yoff[i][1] + j * SUB * RE, 0, 0);
                    MTN_UNROLL for (int t = 0; t < 9; ++t) {
                        const Frag3 fx = frag(rj, xoff[t][0] + j * SUB * RE, xoff[t][1] + j * SUB * RE, 0, 0);
                        MTN_UNROLL for (int i = 0; i < 2; ++i) acc[t][i] = Mma<T>::mma(fx, fy[i], acc[t][i]);
                    }
                }
            }
        } else {
            MTN_UNROLL for (int j = 0; j < NSUB; ++j) {
                MTN_UNROLL for (int kk = 0; kk < SUB / KI; ++kk) {
                    const int k0 = j * SUB + kk * KI;
                    typename Mma<T>::Frag fy[2];
                    MTN_UNROLL for (int i = 0; i < 2; ++i) fy[i] = frag(rd, 0, 0, k0, (nh * 2 + i) * 16);
                    MTN_UNROLL for (int t = 0; t < 9; ++t) {
                        const int sh = HALO + (t / 3 - 1) * W1 + (t % 3 - 1);
                        const typename Mma<T>::Frag fx = frag(rd + KS * 64, 0, 0, k0 + sh, cf * 16);
                        MTN_UNROLL for (int i = 0; i < 2; ++i) acc[t][i] = Mma<T>::mma(fx, fy[i], acc[t][i]);
                    }
                }
            }
        }
        __syncthreads();                                          // own DMA landed (vmcnt 0) + everyone is done with `rd`
    };
    if (steps > 0)
        for (int q = wid; q < PT; q += NW) dma_piece(q, s_begin, sStage0);
    __syncthreads();
    for (int st = 0; st < steps; st += 2) {
        step(st, sStage0, sStage1);
        if (st + 1 < steps) step(st + 1, sStage1, sStage0);
    }
    // partial tile -> slab[split][n][tap*Cin + ci]: the X fragment is the MFMA's row operand, so a lane's
    // quad is 4 consecutive ci of one n
    MTN_UNROLL for (int t = 0; t < 9; ++t)
        MTN_UNROLL for (int i = 0; i < 2; ++i) {
            const int n = n0 + (nh * 2 + i) * 16 + (lane & 15);
            const int k = t * p.Cin + c0 + cf * 16 + 4 * (lane >> 4);
            slab_store(p, split, n, k, acc[t][i]);
        }
}

template <class T, int KSX = Wg3<T>::KS, int WMAX = Wg3<T>::MAXW>
__global__ __launch_bounds__(512) void wgrad3_kernel(WgradArgs p) {
    wgrad3_body<T, KSX, WMAX>(p, (int)blockIdx.x, (int)gridDim.x);
}

// The 3x3 stride-1 weight gradients of ONE stage (its four residual-block convolutions: same map, same channel counts) in one
// launch (round 5).  Each alone fills the 256 CUs with splits of ~10 window steps and a 147 KB partial tile per workgroup
// (580-630 TFLOP/s); at four times the batch the same kernel runs at 710-770 TFLOP/s (tools/probes/wgrad3_scaling.py), which is
// what four problems sharing the CUs get: a quarter of the splits, four times the steps per workgroup, a quarter of the slab.
template <class T, int KSX, int WMAX>
__global__ __launch_bounds__(512) void wgrad3_group_kernel(WgradGroupArgs g) {
    const int L = (int)blockIdx.x;
    int prob = 0;
    for (int i = 1; i < g.count; ++i) prob += L >= g.first[i] ? 1 : 0;
    const WgradArgs p = g.item[prob];
    const int l = L - g.first[prob];
    const int nwg = (p.N / 64) * (p.Cin / 64) * p.splits;         // (the range is padded to a multiple of 8 workgroups)
    if (l >= nwg) return;
    wgrad3_body<T, KSX, WMAX>(p, l, nwg);
}

// dW_oihw[o][ci][kh][kw] = sum_s partial[s][o][(kh*KW+kw)*Cin + ci]
// block = 8 split lanes x 32 columns of FOUR consecutive elements: 16-byte loads, several in flight per
// thread (the first version read 4 bytes per lane with one load in flight: 3 TB/s on a slab that sits in
// the memory-side cache); the 8 lanes are summed in a fixed order -> deterministic.
template <class PT>
__device__ __forceinline__ void wgrad_reduce_body(const PT* __restrict__ partial, float* __restrict__ dw, int splits, int N, int K,
                                                  int Cin, int taps, const int block) {
    __shared__ f32x4_t red[8][32];
    const int sl = threadIdx.x >> 5, el = threadIdx.x & 31;
    const long total = (long)N * K;                        // multiple of 4 (host-checked)
    const long idx = ((long)block * 32 + el) * 4;
    f32x4_t s = zero4();
    if (idx < total) {
        const PT* src = partial + idx;
#pragma unroll 4
        for (int q = sl; q < splits; q += 8) {
            float v[4];
            ld4<PT>(src + (long)q * total, 0, v);
            MTN_UNROLL for (int e = 0; e < 4; ++e) s[e] += v[e];
        }
    }
    red[sl][el] = s;
    __syncthreads();
    if (sl == 0 && idx < total) {
        f32x4_t t = red[0][el];
        MTN_UNROLL for (int q = 1; q < 8; ++q) { const f32x4_t v = red[q][el]; MTN_UNROLL for (int e = 0; e < 4; ++e) t[e] += v[e]; }
        const int n = (int)(idx / K), k = (int)(idx - (long)n * K);
        const int tap = k / Cin, ci = k - tap * Cin;       // the 4 elements share n and tap (Cin % 4 == 0)
        MTN_UNROLL for (int e = 0; e < 4; ++e) dw[((long)n * Cin + ci + e) * taps + tap] = t[e];
    }
}
template <class PT>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const PT* __restrict__ partial, float* __restrict__ dw,
                                                           int splits, int N, int K, int Cin, int taps) {
    wgrad_reduce_body<PT>(partial, dw, splits, N, K, Cin, taps, (int)blockIdx.x);
}
// the split sums of every problem of a grouped launch (block ranges rfirst[])
template <class PT>
__global__ __launch_bounds__(256) void wgrad_reduce_group_kernel(WgradGroupArgs g) {
    const int L = (int)blockIdx.x;
    int prob = 0;
    for (int i = 1; i < g.count; ++i) prob += L >= g.rfirst[i] ? 1 : 0;
    const WgradArgs p = g.item[prob];
    wgrad_reduce_body<PT>(reinterpret_cast<const PT*>(p.partial), p.dW, p.splits, p.N, p.K, p.Cin, p.KH * p.KW, L - g.rfirst[prob]);
}

template <class T, int BN, int BC, int WGM, int WGN, bool TR>
int launch_wgrad(const WgradArgs& p, hipStream_t stream) {
    constexpr int NT = WGM * WGN * 64;
    constexpr int BMR = 128 / (int)sizeof(T);
    constexpr size_t lds = (size_t)BMR * ((BN + BC) * sizeof(T) + 2 * PADB);
#ifndef MTN_EMU
    static MtnLdsAttr attr;
    mtn_set_max_lds(attr, &wgrad_kernel<T, BN, BC, WGM, WGN, TR>, lds);
#endif
    const int tiles = (p.N / BN) * (p.K / BC);
    MTN_LAUNCH((wgrad_kernel<T, BN, BC, WGM, WGN, TR>), dim3(tiles * p.splits), dim3(NT), lds, stream, p);
    return mtn_check_launch("wgrad");
}

template <class T, int BN, int BC, int WGM, int WGN>
int launch_wgrad_dma(const WgradArgs& p, hipStream_t stream) {
    constexpr int NT = WGM * WGN * 64;
    constexpr int BMR = 128 / (int)sizeof(T);
    constexpr size_t lds = 2 * (size_t)BMR * (BN + BC) * sizeof(T);
#ifndef MTN_EMU
    static MtnLdsAttr attr;
    mtn_set_max_lds(attr, &wgrad_dma_kernel<T, BN, BC, WGM, WGN>, lds);
#endif
    const int tiles = (p.N / BN) * (p.K / BC);
#ifdef MTN_EXPERIMENTS       // separate LDS stages (debug key 20): measured slower in every mode, DESIGN.md section 7d
    if constexpr (std::is_same<T, f32x3_t>::value) {
        if (BN == 128 && BC == 128 && g_x3_wgrad_sep) {
            MTN_LAUNCH((wgrad_dma2_kernel<T, BN, BC, WGM, WGN, true>), dim3(tiles * p.splits), dim3(NT), 0, stream, p);
            return mtn_check_launch("wgrad_dma2_sep");
        }
    }
    if (BN == 128 && BC == 128 && WGM * WGN == 8 && g_x3_wgrad_sep == 2) {
        MTN_LAUNCH((wgrad_dma2_kernel<T, BN, BC, WGM, WGN, true>), dim3(tiles * p.splits), dim3(NT), 0, stream, p);
        return mtn_check_launch("wgrad_dma2_sep");
    }
#endif
    if (BN == 128 && BC == 128 && WGM * WGN == 8) {
#ifndef MTN_EMU
        static MtnLdsAttr attr2;
        mtn_set_max_lds(attr2, &wgrad_dma2_kernel<T, BN, BC, WGM, WGN>, lds);
#endif
        MTN_LAUNCH((wgrad_dma2_kernel<T, BN, BC, WGM, WGN>), dim3(tiles * p.splits), dim3(NT), lds, stream, p);
        return mtn_check_launch("wgrad_dma2");
    }
    MTN_LAUNCH((wgrad_dma_kernel<T, BN, BC, WGM, WGN>), dim3(tiles * p.splits), dim3(NT), lds, stream, p);
    return mtn_check_launch("wgrad_dma");
}

template <class T>
int launch_wgrad_row3(const WgradArgs& p, hipStream_t stream) {
    constexpr int BMR = 128 / (int)sizeof(T);
    constexpr size_t lds = 2 * (size_t)BMR * (128 + 3 * 64) * sizeof(T);
#ifndef MTN_EMU
    static MtnLdsAttr attr;
    mtn_set_max_lds(attr, &wgrad_row3_kernel<T>, lds);
#endif
    const int tiles = (p.N / 128) * p.KH * (p.Cin / 64);
    MTN_LAUNCH(wgrad_row3_kernel<T>, dim3(tiles * p.splits), dim3(256), lds, stream, p);
    return mtn_check_launch("wgrad_row3");
}

static int g_wgrad3_ks256 = 1;   // debug key 29: 0 = the narrow step for every map (default: 256 slots for bf16 maps up to 48 wide, 128 for split storage up to 24)
static int mtn_wgrad3_ks(int dtype, int W) {
    if (dtype == MTN_BF16) return (g_wgrad3_ks256 && W <= 48) ? 256 : 128;
    return (dtype == MTN_F32X3 && g_wgrad3_ks256 && W <= 24) ? 128 : 64;    // split storage: 2 x 78 KB of stages for maps up to 24 wide
}
template <class T>
int launch_wgrad3(const WgradArgs& p, hipStream_t stream) {
    constexpr int KS = Wg3<T>::KS, RPB = Wg3<T>::RPB;
    const int halo = p.W + 2;
    const int wr = (KS + 2 * halo + RPB - 1) / RPB * RPB;
    MTN_REQUIRE((KS + wr) * 64 <= Wg3<T>::STAGE, "wgrad3: window of a %d-wide map does not fit", p.W);
    const int tiles = (p.N / 64) * (p.Cin / 64);
    if constexpr (sizeof(T) == 2) {
        if (mtn_wgrad3_ks(MTN_BF16, p.W) == 256) {
            MTN_REQUIRE(p.rows_per_split % 256 == 0, "wgrad3: %d slots per split for 256-slot steps", p.rows_per_split);
            MTN_LAUNCH((wgrad3_kernel<T, 256, 48>), dim3(tiles * p.splits), dim3(512), 0, stream, p);
            return mtn_check_launch("wgrad3");
        }
    }
    if constexpr (std::is_same<T, f32x3_t>::value) {
        if (mtn_wgrad3_ks(MTN_F32X3, p.W) == 128) {
            MTN_REQUIRE(p.rows_per_split % 128 == 0, "wgrad3: %d slots per split for 128-slot steps", p.rows_per_split);
            MTN_LAUNCH((wgrad3_kernel<T, 128, 24>), dim3(tiles * p.splits), dim3(512), 0, stream, p);
            return mtn_check_launch("wgrad3");
        }
    }
    MTN_LAUNCH(wgrad3_kernel<T>, dim3(tiles * p.splits), dim3(512), 0, stream, p);
    return mtn_check_launch("wgrad3");
}

template <class T>
int launch_wgrad3_group(const WgradGroupArgs& g, int blocks, int W, hipStream_t stream) {
    if constexpr (sizeof(T) == 2) {
        if (mtn_wgrad3_ks(MTN_BF16, W) == 256) {
            MTN_LAUNCH((wgrad3_group_kernel<T, 256, 48>), dim3(blocks), dim3(512), 0, stream, g);
            return mtn_check_launch("wgrad3_group");
        }
    }
    if constexpr (std::is_same<T, f32x3_t>::value) {
        if (mtn_wgrad3_ks(MTN_F32X3, W) == 128) {
            MTN_LAUNCH((wgrad3_group_kernel<T, 128, 24>), dim3(blocks), dim3(512), 0, stream, g);
            return mtn_check_launch("wgrad3_group");
        }
    }
    MTN_LAUNCH((wgrad3_group_kernel<T, Wg3<T>::KS, Wg3<T>::MAXW>), dim3(blocks), dim3(512), 0, stream, g);
    return mtn_check_launch("wgrad3_group");
}

}  // namespace

static int g_use_tr16 = 1;
static int g_wgrad3 = 1;         // debug key 13: 0 = per-tap tiles for 3x3 stride-1 layers too
void mtn_wgrad_set_win(int on) { g_wgrad3 = on; }
void mtn_wgrad_set_ks256(int on) { g_wgrad3_ks256 = on ? 1 : 0; }

// the window kernel serves 3x3 / stride 1 / pad 1 layers whose two window stages fit the LDS of a CU
int mtn_wgrad3_ok(int dtype, int H, int W, int ksize, int stride) {
    if (!g_wgrad3 || ksize != 3 || stride != 1) return 0;
    (void)H;
    return W <= (dtype == MTN_BF16 ? 96 : 48);                    // Wg3<T>::MAXW
}

// split of the padded slot axis: one workgroup per CU
int mtn_wgrad3_splits(int B, int H, int W, int N, int Cin, int dtype, int* slots_per_split) {
    const int ks = mtn_wgrad3_ks(dtype, W);
    const long S = (long)B * (H + 1) * (W + 1);
    const long tiles = (long)(N / 64) * (Cin / 64);
    long splits = (g_wgrad_slots > 0 ? g_wgrad_slots : 256) / tiles;
    if (splits < 1) splits = 1;
    const long max_splits = (S + ks - 1) / ks;
    if (splits > max_splits) splits = max_splits;
    long sps = (S + splits - 1) / splits;
    sps = (sps + ks - 1) / ks * ks;
    *slots_per_split = (int)sps;
    return (int)((S + sps - 1) / sps);
}

// the kernel-row form serves 3x3 layers whose input has ONE 64-channel slab per tap tile anyway (Cin = 64:
// wgrad_tile picks 128 x 64 there) and 128-wide outputs -- conv2 of the model
static int g_wgrad_row3 = 2;     // debug key 18 (0 off, 1: 64-channel inputs only, 2: every stride-2 3x3 layer too)
void mtn_wgrad_set_row3(int on) { g_wgrad_row3 = on; }
int mtn_wgrad_row3_ok(int dtype, int N, int Cin, int ksize, int stride) {
    (void)dtype;
    // (round 5: also the wider stride-2 layers -- down1 / down2 -- as kernel-row tiles of 64-channel slabs: one dY tile for the three
    //  taps of a kernel row instead of one per tap; down1 150 -> 134 us, down2 138 -> 135, step -0.03 ms.  Debug key 18 = 1: Cin = 64 only)
    return g_wgrad_row3 && g_wgrad_dma && ksize == 3 && (Cin == 64 || (g_wgrad_row3 == 2 && stride == 2 && Cin % 64 == 0)) && N % 128 == 0 &&
           !(stride == 1 && g_wgrad3);
}
int mtn_wgrad_row3_splits(int M, int N, int Cin, int dtype, int* rows_per_split) {
    const int bmr = dtype == MTN_BF16 ? 64 : 32;
    const long tiles = (long)(N / 128) * 3 * (Cin / 64);
    long splits = (g_wgrad_slots > 0 ? g_wgrad_slots : 512) / tiles;      // two workgroups per CU
    if (splits < 1) splits = 1;
    if (splits >= 8 && tiles <= 24) splits &= ~7L;
    const long max_splits = (M + bmr - 1) / bmr;
    if (splits > max_splits) splits = max_splits;
    long rps = (M + splits - 1) / splits;
    rps = (rps + bmr - 1) / bmr * bmr;
    *rows_per_split = (int)rps;
    long n = (M + rps - 1) / rps;
    if (n >= 8 && tiles <= 24) n = (n + 7) / 8 * 8;
    return (int)n;
}
#ifdef MTN_EXPERIMENTS
void mtn_wgrad_set_dma(int on) { g_wgrad_dma = on; }
void mtn_wgrad_set_tile(int t) { g_wgrad_tile = t; }
#else
void mtn_wgrad_set_dma(int on) { (void)on; g_wgrad_dma = 1; }
void mtn_wgrad_set_tile(int t) { g_wgrad_tile = t == 64 ? 64 : 0; }      // (64 x 64 tiles are product kernels: the 64-channel layers use them)
#endif
void mtn_wgrad_set_slots(int n) { g_wgrad_slots = n > 0 ? n : 0; }
#ifdef MTN_EXPERIMENTS
static int g_wgrad_skip_reduce = 0;
void mtn_wgrad_set_skip_reduce(int on) { g_wgrad_skip_reduce = on; }
#endif
#ifdef MTN_EXPERIMENTS
void mtn_wgrad_set_tr16(int on) { g_use_tr16 = on; }
void mtn_wgrad_set_x3_sep(int on) { g_x3_wgrad_sep = (on == 1 || on == 2) ? on : 0; }
#else
void mtn_wgrad_set_tr16(int on) { (void)on; }
void mtn_wgrad_set_x3_sep(int on) { (void)on; }
#endif

// tile of dW owned by one workgroup: BN output channels x BC input channels (of one tap)
static void wgrad_tile(int N, int Cin, int* bn, int* bc) {
    *bn = N % 128 == 0 ? 128 : 64;
    *bc = Cin % 128 == 0 ? 128 : 64;
    if (g_wgrad_tile == 64) { *bn = 64; *bc = 64; }
    if (*bn == 64 && *bc == 128) *bc = 64;          // (64,128) is not instantiated
}

// Split of the pixel axis: the launch should fill the chip's workgroup slots exactly once
// (slots = 256 CUs x workgroups per CU at this tile's LDS footprint): one block more than a full
// wave of workgroups costs a whole extra round, and every split adds a partial slab to reduce.
int mtn_wgrad_splits(int M, int N, int K, int Cin, int dtype, int* rows_per_split) {
    const int bmr = dtype == MTN_BF16 ? 64 : 32;
    int bn, bc;
    wgrad_tile(N, Cin, &bn, &bc);
    const long tiles = (long)(N / bn) * (K / bc);
    long slots = 1024;
    if (g_wgrad_dma) slots = (bn == 128 && bc == 128) ? 512 : (bn == 128 ? 768 : 1024);
    if (g_wgrad_slots > 0) slots = g_wgrad_slots;
    long splits = slots / tiles;
    if (splits < 1) splits = 1;
    if (splits >= 8 && tiles <= 24) splits &= ~7L;        // whole splits per XCD group (see the kernel)
    const long max_splits = (M + bmr - 1) / bmr;
    if (splits > max_splits) splits = max_splits;
    long rps = (M + splits - 1) / splits;
    rps = (rps + bmr - 1) / bmr * bmr;
    *rows_per_split = (int)rps;
    long n = (M + rps - 1) / rps;
    if (n >= 8 && tiles <= 24) n = (n + 7) / 8 * 8;        // empty splits add zero slabs
    return (int)n;
}

template <class T, bool TR>
static int wgrad_dispatch(const WgradArgs& p, hipStream_t stream) {
    int bn, bc;
    wgrad_tile(p.N, p.Cin, &bn, &bc);
    constexpr bool x3_ = std::is_same<T, f32x3_t>::value;
    if ((g_wgrad_dma || x3_) && (TR || sizeof(T) == 4)) {     // (split storage has no register-staged form)
        // 8 waves (4 DMA pieces per wave per step instead of 8: the issue of an LDS-DMA instruction
        // costs its wave ~100 cycles); debug key 7 = 4 selects the 4-wave layout (+5..9 % slower)
        // (split storage: the fragments need no conversion since round 4, so it takes the 8-wave layout like bf16; until then
        //  4 waves of 64 x 64 -- 8 fragment conversions per 48 MFMAs instead of 6 per 24 -- were 2 % faster)
        constexpr bool x3 = std::is_same<T, f32x3_t>::value;
        (void)x3;
#ifdef MTN_EXPERIMENTS
        if (bn == 128 && bc == 128 && g_wgrad_tile == 4) return launch_wgrad_dma<T, 128, 128, 2, 2>(p, stream);      // (debug key 7 = 4: 2x2 waves)
#endif
        if (bn == 128 && bc == 128) return launch_wgrad_dma<T, 128, 128, 4, 2>(p, stream);
        if (bn == 128) return launch_wgrad_dma<T, 128, 64, 4, 1>(p, stream);
        return launch_wgrad_dma<T, 64, 64, 2, 2>(p, stream);
    }
#ifdef MTN_EXPERIMENTS       // the register-staged pipeline (debug keys 5 = 0 / 1 = 0) the LDS-DMA kernels replaced in round 1
    if constexpr (!x3_) {
        if (bn == 128 && bc == 128) return launch_wgrad<T, 128, 128, 2, 2, TR>(p, stream);
        if (bn == 128) return launch_wgrad<T, 128, 64, 4, 1, TR>(p, stream);
        return launch_wgrad<T, 64, 64, 2, 2, TR>(p, stream);
    }
#endif
    return MTN_ERR_UNSUPPORTED;
}

// bf16 mode keeps the split-K partial tiles in bf16 (round 4: -0.11 ms per step, 1.4 GB of slab traffic halved).  Each partial is
// an f32 MFMA accumulation rounded once (2^-9 relative); the splits are summed in f32.  The reference's own bf16 path rounds the
// whole gradient to bf16 once at the end (autocast: the convolution's weight gradient has the dtype of its bf16 weight copy), so
// this is the same class of error; measured against the float64 oracle the bf16 column of the per-tensor gradient table does
// not move (5e-2 .. 8e-2, profiles/r04_grad_errors_f64_b4.txt).  fp32 / bf16x3 slabs stay f32.  Debug key 28 = 0: f32 slabs.
static int g_wgrad_slab_bf16 = 1;
void mtn_wgrad_set_slab_bf16(int on) { g_wgrad_slab_bf16 = on ? 1 : 0; }

int mtn_wgrad_run(const WgradArgs& p0, int dtype, hipStream_t stream) {
    WgradArgs p = p0;
    p.slab_bf16 = (dtype == MTN_BF16 && g_wgrad_slab_bf16) ? 1 : 0;
    const int bmr = dtype == MTN_BF16 ? 64 : 32;
    MTN_REQUIRE(dtype == MTN_BF16 || dtype == MTN_F32 || dtype == MTN_F32X3, "wgrad: bad dtype %d", dtype);
    MTN_REQUIRE(p.N % 64 == 0 && p.Cin % 64 == 0, "wgrad: N=%d and Cin=%d must be multiples of 64", p.N, p.Cin);
    MTN_REQUIRE(p.K == p.KH * p.KW * p.Cin, "wgrad: K mismatch");
    MTN_REQUIRE(p.M == p.B * p.OH * p.OW, "wgrad: M mismatch");
    const bool w3 = p.win == 1, row3 = p.win == 2;       // window kernel (padded slot axis) / kernel-row tiles
    MTN_REQUIRE(w3 || (p.splits >= 1 && p.rows_per_split % bmr == 0 && (long)p.splits * p.rows_per_split >= p.M),
                "wgrad: bad split (%d x %d rows for M=%d)", p.splits, p.rows_per_split, p.M);
    MTN_REQUIRE(!w3 || (p.splits >= 1 && (long)p.splits * p.rows_per_split >= (long)p.B * (p.H + 1) * (p.W + 1)),
                "wgrad3: bad split (%d x %d slots)", p.splits, p.rows_per_split);
    MTN_REQUIRE(!row3 || (p.KH == 3 && p.KW == 3 && p.N % 128 == 0), "wgrad_row3: 3x3 layers with 128-wide outputs only");
    MTN_REQUIRE(((uintptr_t)p.dY | (uintptr_t)p.X) % 16 == 0, "wgrad: pointers must be 16-byte aligned");
    MTN_REQUIRE(dtype != MTN_F32X3 || ((((uintptr_t)p.dY | (uintptr_t)p.X) % 16 == 0) && p.lddy % 32 == 0 && p.ldx % 32 == 0),
                "wgrad: split storage (MTN_DTYPE_F32X3) needs 16-byte aligned operands and row pitches that are multiples of 32");
    MTN_REQUIRE((p.KH * p.KW == 1 && p.stride == 1 && p.pad == 0) ||
                    (((long)p.M + 128) * p.OW < (1L << 32) && ((long)p.M / p.OW + 128) * p.OH < (1L << 32)),
                "wgrad: M * OW too large for the multiply-high pixel walk");
    MTN_REQUIRE((long)p.M * p.lddy < (1L << 30) && (long)p.B * p.H * p.W * p.ldx < (1L << 30), "wgrad: tensor too large for 32-bit byte offsets");
    {
        const size_t esz = dtype == MTN_BF16 ? 2 : 4;
        p.magic_ow = p.OW > 1 ? (unsigned)(0x100000000ull / (unsigned)p.OW) + 1u : 0u;
        p.magic_oh = p.OH > 1 ? (unsigned)(0x100000000ull / (unsigned)p.OH) + 1u : 0u;
        p.y_bytes = (unsigned)((((size_t)p.M - 1) * p.lddy + p.N) * esz);
        p.x_bytes = (unsigned)((((size_t)p.B * p.H * p.W - 1) * p.ldx + p.Cin) * esz);
    }
    int rc;
    if (row3) rc = dtype == MTN_BF16 ? launch_wgrad_row3<bf16_t>(p, stream)
                                     : (dtype == MTN_F32X3 ? launch_wgrad_row3<f32x3_t>(p, stream) : launch_wgrad_row3<float>(p, stream));
    else if (w3) {
        const unsigned w1 = (unsigned)p.W + 1u, h1 = (unsigned)p.H + 1u;
        MTN_REQUIRE(p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.OH == p.H && p.OW == p.W, "wgrad3: 3x3 stride-1 layers only");
        MTN_REQUIRE(((long)p.B * h1 * w1 + 4096) * w1 < (1L << 32), "wgrad3: too many slots for the multiply-high decode");
        p.magic_ow = (unsigned)(0x100000000ull / w1) + 1u;        // slot / (W+1)
        p.magic_oh = (unsigned)(0x100000000ull / h1) + 1u;        // row / (H+1)
        rc = dtype == MTN_BF16 ? launch_wgrad3<bf16_t>(p, stream)
                               : (dtype == MTN_F32X3 ? launch_wgrad3<f32x3_t>(p, stream) : launch_wgrad3<float>(p, stream));
    } else if (dtype == MTN_BF16) {
#ifdef MTN_EXPERIMENTS
        rc = g_use_tr16 ? wgrad_dispatch<bf16_t, true>(p, stream) : wgrad_dispatch<bf16_t, false>(p, stream);
#else
        rc = wgrad_dispatch<bf16_t, true>(p, stream);
#endif
    }
    else if (dtype == MTN_F32X3) rc = wgrad_dispatch<f32x3_t, false>(p, stream);
    else rc = wgrad_dispatch<float, false>(p, stream);
    if (rc) return rc;
#ifdef MTN_EXPERIMENTS
    if (g_wgrad_skip_reduce) return MTN_OK;              // (timing ablation, debug key 27: the step's result is WRONG)
#endif
    const long total = (long)p.N * p.K;
    const int blocks = (int)((total / 4 + 31) / 32);
    if (p.slab_bf16)
        MTN_LAUNCH(wgrad_reduce_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const bf16_t*>(p.partial), p.dW, p.splits,
                   p.N, p.K, p.Cin, p.KH * p.KW);
    else
        MTN_LAUNCH(wgrad_reduce_kernel<float>, dim3(blocks), dim3(256), 0, stream, (const float*)p.partial, p.dW, p.splits, p.N, p.K,
                   p.Cin, p.KH * p.KW);
    return mtn_check_launch("wgrad_reduce");
}


// ---- grouped pointwise weight gradients (wgrad_dma2_group_kernel) ----------------------------------------------------------
static int g_wgrad_group = 1;    // debug key 30: 0 = mtn_linear_wgrad_group runs its problems one launch each
void mtn_wgrad_set_group(int on) { g_wgrad_group = on ? 1 : 0; }

// Split of the row axis shared by the `count` problems [M][N_i] x [M][K_i] of one grouped launch (0: not groupable --
// more than MTN_WGRAD_GROUP_MAX problems, or a side that is no multiple of the 128 x 128 tile): all tiles of all problems
// x splits fill the 512 workgroup slots of the 8-wave tile once
int mtn_wgrad_group_plan(int dtype, int M, const int* N, const int* K, int count, int* rows_per_split) {
    if (!g_wgrad_group || count < 2 || count > MTN_WGRAD_GROUP_MAX || M <= 0) return 0;
    long tiles = 0;
    for (int i = 0; i < count; ++i) {
        if (N[i] <= 0 || K[i] <= 0 || N[i] % 128 || K[i] % 128) return 0;
        tiles += (long)(N[i] / 128) * (K[i] / 128);
    }
    const int bmr = dtype == MTN_BF16 ? 64 : 32;
    const long max_splits = ((long)M + bmr - 1) / bmr;
    long splits;
    if (g_wgrad_slots > 0) {                                  // (debug key 10: a slot budget instead of the search below)
        splits = g_wgrad_slots / tiles;
        if (splits < 1) splits = 1;
    } else {
        // The launch runs in rounds of 512 workgroups (two per CU); every problem's range is padded to a multiple of 8, and a
        // workgroup costs its K-steps plus ~16 steps' worth of prologue, ring fill and partial-tile store.  Pick the split count
        // with the least rounds x (steps + 16): at 37,120 rows and 96 tiles that is 10 splits (960 workgroups, two full rounds:
        // 209 us) -- 5 splits pad to 544 workgroups, a second round of 32 (318 us); tools/probes/linear_wgrad_group_probe.py
        long best = 0;
        splits = 1;
        for (long sp = 1; sp <= 64 && sp <= max_splits; ++sp) {
            long blocks = 0;
            for (int i = 0; i < count; ++i) blocks += ((long)(N[i] / 128) * (K[i] / 128) * sp + 7) / 8 * 8;
            const long rows = (((long)M + sp - 1) / sp + bmr - 1) / bmr * bmr;
            const long cost = ((blocks + 511) / 512) * (rows / bmr + 16);
            if (best == 0 || cost < best) { best = cost; splits = sp; }
        }
    }
    if (splits > max_splits) splits = max_splits;
    long rps = (M + splits - 1) / splits;
    rps = (rps + bmr - 1) / bmr * bmr;
    *rows_per_split = (int)rps;
    return (int)((M + rps - 1) / rps);
}

// g.item[i]: dY / lddy / X / ldx / dW / N / K (= Cin) filled by the caller; everything else here
int mtn_wgrad_group_run(WgradGroupArgs& g, int dtype, int M, void* workspace, size_t ws_bytes, hipStream_t stream) {
    MTN_REQUIRE(dtype == MTN_BF16 || dtype == MTN_F32 || dtype == MTN_F32X3, "wgrad group: bad dtype %d", dtype);
    MTN_REQUIRE(g.count >= 2 && g.count <= MTN_WGRAD_GROUP_MAX, "wgrad group: %d problems", g.count);
    int Ns[MTN_WGRAD_GROUP_MAX], Ks[MTN_WGRAD_GROUP_MAX], rps = 0;
    for (int i = 0; i < g.count; ++i) { Ns[i] = g.item[i].N; Ks[i] = g.item[i].K; }
    const int splits = mtn_wgrad_group_plan(dtype, M, Ns, Ks, g.count, &rps);
    MTN_REQUIRE(splits > 0, "wgrad group: the problems do not share the 128 x 128 tile");
    const size_t esz = dtype == MTN_BF16 ? 2 : 4;
    const int slab_bf16 = (dtype == MTN_BF16 && g_wgrad_slab_bf16) ? 1 : 0;
    size_t off = 0;
    int blocks = 0, rblocks = 0;
    for (int i = 0; i < g.count; ++i) {
        WgradArgs& p = g.item[i];
        MTN_REQUIRE(p.dY && p.X && p.dW, "wgrad group: null operand in problem %d", i);
        MTN_REQUIRE((((uintptr_t)p.dY | (uintptr_t)p.X) % 16) == 0 && p.lddy % 8 == 0 && p.ldx % 8 == 0, "wgrad group: operands must be 16-byte aligned");
        MTN_REQUIRE(dtype != MTN_F32X3 || (p.lddy % 32 == 0 && p.ldx % 32 == 0), "wgrad group: split storage needs row pitches that are multiples of 32");
        MTN_REQUIRE(p.lddy >= p.N && p.ldx >= p.K, "wgrad group: row pitch below the row length in problem %d", i);
        MTN_REQUIRE((long)M * p.lddy < (1L << 30) && (long)M * p.ldx < (1L << 30), "wgrad group: tensor too large for 32-bit byte offsets");
        p.B = 1; p.H = 1; p.W = M; p.OH = 1; p.OW = M; p.Cin = p.K; p.KH = p.KW = 1; p.stride = 1; p.pad = 0;
        p.M = M; p.splits = splits; p.rows_per_split = rps; p.win = 0;
        p.magic_ow = 0; p.magic_oh = 0;
        p.y_bytes = (unsigned)((((size_t)M - 1) * p.lddy + p.N) * esz);
        p.x_bytes = (unsigned)((((size_t)M - 1) * p.ldx + p.K) * esz);
        p.slab_bf16 = slab_bf16;
        p.partial = reinterpret_cast<float*>(static_cast<char*>(workspace) + off);
        off += ((size_t)splits * p.N * p.K * (slab_bf16 ? 2 : 4) + 255) / 256 * 256;
        g.first[i] = blocks;
        blocks += ((p.N / 128) * (p.K / 128) * splits + 7) / 8 * 8;
        g.rfirst[i] = rblocks;
        rblocks += (int)(((long)p.N * p.K / 4 + 31) / 32);
    }
    g.first[g.count] = blocks;
    g.rfirst[g.count] = rblocks;
    MTN_REQUIRE(off <= ws_bytes, "wgrad group: workspace too small (%zu bytes, %zu needed)", ws_bytes, off);
    constexpr size_t lds = 2 * 128 * (128 + 128);             // two stages of [BMR][128 + 128] T: BMR * sizeof(T) = 128
#ifndef MTN_EMU
    static MtnLdsAttr attr_b, attr_x, attr_f;
#endif
    if (dtype == MTN_BF16) {
#ifndef MTN_EMU
        mtn_set_max_lds(attr_b, &wgrad_dma2_group_kernel<bf16_t>, lds);
#endif
        MTN_LAUNCH(wgrad_dma2_group_kernel<bf16_t>, dim3(blocks), dim3(512), lds, stream, g);
    } else if (dtype == MTN_F32X3) {
#ifndef MTN_EMU
        mtn_set_max_lds(attr_x, &wgrad_dma2_group_kernel<f32x3_t>, lds);
#endif
        MTN_LAUNCH(wgrad_dma2_group_kernel<f32x3_t>, dim3(blocks), dim3(512), lds, stream, g);
    } else {
#ifndef MTN_EMU
        mtn_set_max_lds(attr_f, &wgrad_dma2_group_kernel<float>, lds);
#endif
        MTN_LAUNCH(wgrad_dma2_group_kernel<float>, dim3(blocks), dim3(512), lds, stream, g);
    }
    int rc = mtn_check_launch("wgrad_dma2_group");
    if (rc) return rc;
    if (slab_bf16) MTN_LAUNCH(wgrad_reduce_group_kernel<bf16_t>, dim3(rblocks), dim3(256), 0, stream, g);
    else MTN_LAUNCH(wgrad_reduce_group_kernel<float>, dim3(rblocks), dim3(256), 0, stream, g);
    return mtn_check_launch("wgrad_reduce_group");
}

// bytes of the slabs of a grouped launch (0: not groupable)
size_t mtn_wgrad_group_bytes(int dtype, int M, const int* N, const int* K, int count) {
    int rps = 0;
    const int splits = mtn_wgrad_group_plan(dtype, M, N, K, count, &rps);
    if (!splits) return 0;
    size_t off = 0;
    for (int i = 0; i < count; ++i) off += ((size_t)splits * N[i] * K[i] * 4 + 255) / 256 * 256;
    return off;
}

// ---- grouped 3x3 stride-1 weight gradients (wgrad3_group_kernel): `count` problems of ONE geometry ------------------------------
// split of the padded slot axis of each problem: all tiles of all problems x splits = one workgroup per CU (0: not groupable)
int mtn_wgrad3_group_splits(int B, int H, int W, int N, int Cin, int dtype, int count, int* slots_per_split) {
    if (!g_wgrad_group || count < 2 || count > MTN_WGRAD_GROUP_MAX || !mtn_wgrad3_ok(dtype, H, W, 3, 1) || N % 64 || Cin % 64) return 0;
    const int ks = mtn_wgrad3_ks(dtype, W);
    const long S = (long)B * (H + 1) * (W + 1);
    const long tiles = (long)(N / 64) * (Cin / 64) * count;
    long splits = (g_wgrad_slots > 0 ? g_wgrad_slots : 256) / tiles;
    if (splits < 1) splits = 1;
    const long max_splits = (S + ks - 1) / ks;
    if (splits > max_splits) splits = max_splits;
    long sps = (S + splits - 1) / splits;
    sps = (sps + ks - 1) / ks * ks;
    *slots_per_split = (int)sps;
    return (int)((S + sps - 1) / sps);
}

size_t mtn_wgrad3_group_bytes(int B, int H, int W, int N, int Cin, int dtype, int count) {
    int sps = 0;
    const int splits = mtn_wgrad3_group_splits(B, H, W, N, Cin, dtype, count, &sps);
    return splits ? (size_t)count * (((size_t)splits * N * 9 * Cin * 4 + 255) / 256 * 256) : 0;
}

// g.item[i]: dY / lddy / X / ldx / dW filled by the caller; the shared geometry and everything else here
int mtn_wgrad3_group_run(WgradGroupArgs& g, int dtype, int B, int H, int W, int Cin, int N, void* workspace, size_t ws_bytes, hipStream_t stream) {
    MTN_REQUIRE(dtype == MTN_BF16 || dtype == MTN_F32 || dtype == MTN_F32X3, "wgrad3 group: bad dtype %d", dtype);
    int sps = 0;
    const int splits = mtn_wgrad3_group_splits(B, H, W, N, Cin, dtype, g.count, &sps);
    MTN_REQUIRE(splits > 0, "wgrad3 group: the geometry is not served by the window kernel");
    const unsigned w1 = (unsigned)W + 1u, h1 = (unsigned)H + 1u;
    MTN_REQUIRE(((long)B * h1 * w1 + 4096) * w1 < (1L << 32), "wgrad3 group: too many slots for the multiply-high decode");
    const size_t esz = dtype == MTN_BF16 ? 2 : 4;
    const int slab_bf16 = (dtype == MTN_BF16 && g_wgrad_slab_bf16) ? 1 : 0;
    const long M = (long)B * H * W;
    size_t off = 0;
    int blocks = 0, rblocks = 0;
    for (int i = 0; i < g.count; ++i) {
        WgradArgs& p = g.item[i];
        MTN_REQUIRE(p.dY && p.X && p.dW, "wgrad3 group: null operand in problem %d", i);
        MTN_REQUIRE((((uintptr_t)p.dY | (uintptr_t)p.X) % 16) == 0 && p.lddy >= N && p.ldx >= Cin && p.lddy % 8 == 0 && p.ldx % 8 == 0,
                    "wgrad3 group: operands must be 16-byte aligned rows of at least the channel count (problem %d)", i);
        MTN_REQUIRE(dtype != MTN_F32X3 || (p.lddy % 32 == 0 && p.ldx % 32 == 0), "wgrad3 group: split storage needs row pitches that are multiples of 32");
        MTN_REQUIRE(M * p.lddy < (1L << 30) && M * p.ldx < (1L << 30), "wgrad3 group: tensor too large for 32-bit byte offsets");
        p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.OH = H; p.OW = W; p.KH = p.KW = 3; p.stride = 1; p.pad = 1;
        p.M = (int)M; p.N = N; p.K = 9 * Cin;
        p.splits = splits; p.rows_per_split = sps; p.win = 1;
        p.magic_ow = (unsigned)(0x100000000ull / w1) + 1u;
        p.magic_oh = (unsigned)(0x100000000ull / h1) + 1u;
        p.y_bytes = (unsigned)((((size_t)M - 1) * p.lddy + N) * esz);
        p.x_bytes = (unsigned)((((size_t)M - 1) * p.ldx + Cin) * esz);
        p.slab_bf16 = slab_bf16;
        p.partial = reinterpret_cast<float*>(static_cast<char*>(workspace) + off);
        off += ((size_t)splits * N * p.K * (slab_bf16 ? 2 : 4) + 255) / 256 * 256;
        g.first[i] = blocks;
        blocks += ((N / 64) * (Cin / 64) * splits + 7) / 8 * 8;
        g.rfirst[i] = rblocks;
        rblocks += (int)(((long)N * p.K / 4 + 31) / 32);
    }
    g.first[g.count] = blocks;
    g.rfirst[g.count] = rblocks;
    MTN_REQUIRE(off <= ws_bytes, "wgrad3 group: workspace too small (%zu bytes, %zu needed)", ws_bytes, off);
    int rc = dtype == MTN_BF16 ? launch_wgrad3_group<bf16_t>(g, blocks, W, stream)
                               : (dtype == MTN_F32X3 ? launch_wgrad3_group<f32x3_t>(g, blocks, W, stream) : launch_wgrad3_group<float>(g, blocks, W, stream));
    if (rc) return rc;
    if (slab_bf16) MTN_LAUNCH(wgrad_reduce_group_kernel<bf16_t>, dim3(rblocks), dim3(256), 0, stream, g);
    else MTN_LAUNCH(wgrad_reduce_group_kernel<float>, dim3(rblocks), dim3(256), 0, stream, g);
    return mtn_check_launch("wgrad_reduce_group");
}
