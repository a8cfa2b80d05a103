// igemm.hip -- NHWC implicit-GEMM on MFMA: conv forward, conv data-gradient, 1x1 conv, Linear.
//
//   Out[m][n] = epilogue( sum_k A(m,k) * Wp[n][k] ),   m = (b, oy, ox),  k = (kh, kw, ci)
//
// One kernel serves (reference sites in parentheses):
//   * Conv 3x3 / 1x1 forward            (model/gelan.py:37-45, 56)            gather: iy = oy*s - p + kh
//   * Conv data-gradient (dgrad)        (autograd of the above)               gather: oy = (iy + p - kh)/s
//   * proj 1x1                          (model/multitasknet.py:13,26)
//   * nn.Linear qkv / out / MLP + their dgrad (model/transformer.py:34-37,56-58)
// through the affine tap map  t = o*mul + kh*sgn + off,  valid iff t % div == 0 and 0 <= t/div < H.
//
// Tiling: BM x BN output tile per workgroup, 4 waves (64-wide) in a WGM x WGN grid, each wave an
// (BM/WGM) x (BN/WGN) sub-tile of 16x16 MFMA fragments; K step = 128 bytes of the reduction
// axis per row (64 bf16 / 32 f32 -- a whole number of channels of ONE tap since Cin % BK == 0).
// A and W tiles are staged global -> registers -> LDS (144-byte padded rows: ds_read_b128
// fragments 2-way conflicts at worst), double buffered, one barrier per K step: the loads of
// step k+1 are in flight while the MFMAs of step k run.  The accumulator tile is staged through
// LDS as f32 so the epilogue (bias, GELU, residual, BN partial statistics) is applied row-wise
// and the stores are 16-byte coalesced along the channel axis.
//
// BN statistics (model/gelan.py:46, train mode): every tile writes its per-channel sum and
// sum-of-squares over the tile's rows to stats[tile_m][2][N]; bn_finalize reduces the slab in
// tile order -> deterministic, no float atomics.
#include <type_traits>
#include "mtn_types.h"
#include "mtn_kernels.h"

namespace {

constexpr int ROW_BYTES = 128;          // reduction bytes per row per K step
constexpr int PITCH_BYTES = 144;        // padded LDS row (register-staged pipelines)


// In-kernel phase stamps (diagnostic builds only: make hip EXTRA=-DMTN_STAMPS; the shipped library
// has none).  Wave 0 of the first 256 workgroups sums the cycles of each phase of its K loop.
#ifdef MTN_STAMPS
__device__ unsigned long long g_stamps[256 * 16];
#define STAMP(var) unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define STAMP(var)
#endif

// One wave-instruction of LDS-DMA: lane l copies 16 bytes from its own global address to
// lds_block + 16*l (the destination is wave-uniform base + lane*16 -- cdna_hip_programming.md
// section 5, "Async global->LDS copy"); no VGPR staging, no ds_write.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_block) {
#ifdef MTN_EMU
    std::memcpy(static_cast<char*>(lds_block) + 16 * (threadIdx.x & 63), gsrc, 16);
#else
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_block, 16, 0, 0);
#endif
}

// STAGES = 2: double-buffered LDS, one barrier per K step, f32 C staging (2 workgroups / CU).
// STAGES = 1: single LDS buffer + register prefetch, two barriers per K step, C staged in T:
//             half the LDS -> 4 workgroups / CU, whose main loops / epilogues overlap each other.
// STAGES = 3: LDS-DMA (global_load_lds 16 B) into two unpadded, XOR-swizzled LDS buffers: the
//             swizzle is applied to the per-lane SOURCE chunk and to the fragment read, the LDS
//             image itself is lane-linear (rule 21).  One barrier per K step, no staging VGPRs.
// STAGES = 5: "window" mode for stride-1 3x3 taps (forward and data gradient).  In the flattened
//             (b, y, x) pixel order tap (kh, kw) of output row m reads input row m + dy*W + dx, so
//             the BM rows of a tile need only the CONTIGUOUS input rows [m0 - W - 1, m0 + BM + W + 1).
//             That window (up to 128 channels at a time) is DMA-ed into LDS once per tile and every
//             tap reads its A fragments from it at a row offset; rows whose tap falls into the zero
//             padding read a zero row instead (the neighbouring image row / image the shift would
//             hit is exactly the padded case).  Only the weight tiles stream per K step: the
//             activation operand crosses the L2 -> LDS path ~1.2-1.8x instead of 9x per tile.
// LDS allows two 128x128 workgroups (three of the smaller tiles) per CU: telling the compiler so
// lets its scheduler spend registers on keeping the fragment reads of a K step in flight instead
// of minimising pressure for an occupancy the LDS footprint rules out anyway.
#ifdef MTN_EMU
#define MTN_WAVES_PER_SIMD(lo, hi)
#else
#define MTN_WAVES_PER_SIMD(lo, hi) __attribute__((amdgpu_waves_per_eu(lo, hi)))
#endif
// (4-byte storage: a split-storage fragment is 8 registers, and left to itself hipcc spends 130-137 / 176 registers on the
//  8-wave 128x128 / the 4-wave tiles -- one over the step that keeps two / three workgroups per CU resident; the lower bound
//  holds it to 128 / 168)
template <class T, int BM, int BN, int NT> struct IgemmOcc {
    static constexpr bool big = BM * BN >= 128 * 128;
    static constexpr bool img = BM == 144;               // image tile: one workgroup per CU, all the registers it wants
    static constexpr int lo = img ? 1 : (sizeof(T) == 4 ? (big ? (NT == 512 ? 4 : 1) : 3) : 1);
    static constexpr int hi = img ? 2 : (big ? (sizeof(T) == 4 && NT == 512 ? 4 : 2) : 3);
};
template <class T, int BM, int BN, int WGM, int WGN, int STAGES, int MODE = 0>
__global__ __launch_bounds__(WGM* WGN * 64) MTN_WAVES_PER_SIMD((IgemmOcc<T, BM, BN, WGM * WGN * 64>::lo), (IgemmOcc<T, BM, BN, WGM * WGN * 64>::hi))
void igemm_kernel(IgemmArgs p) {
    constexpr int NT = WGM * WGN * 64;
    constexpr int BK = ROW_BYTES / (int)sizeof(T);
    constexpr int PITCH = PITCH_BYTES / (int)sizeof(T);
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int FM = WM / 16, FN = WN / 16;
    constexpr int KI = Mma<T>::KI;
    constexpr int RA = BM * 8 / NT;      // A chunks per thread per K step
    constexpr int RB = BN * 8 / NT;      // W chunks per thread per K step
    constexpr int RSTEP = NT / 8;        // rows between a thread's chunks
    typedef typename std::conditional<STAGES != 2, typename StageT<T>::type, float>::type CT;     // staged C element
    constexpr int EV = Ev<T>::N;         // channels per thread in the epilogue (split storage: a hi and a lo chunk of 8)
    constexpr bool DMA = STAGES >= 3;
    constexpr bool WIN = STAGES == 5 || STAGES == 6;
    constexpr bool IMG = STAGES == 6;      // window mode, whole reduction axis resident, weights through registers
    constexpr int WOFF = IMG ? 0 : 2 * BN; // LDS rows in front of the window (the two W tiles)
    constexpr int NBUF = STAGES == 1 ? 1 : ((STAGES == 4 || STAGES == 7) ? 3 : 2);
    constexpr int PITCH_E = DMA ? BK : PITCH;                              // LDS row pitch in elements
    constexpr int CP = BN + 16 / (int)sizeof(CT);                          // pitch of the staged C tile
    static_assert((STAGES >= 5 || BM * 8 % NT == 0) && BN * 8 % NT == 0, "loader mapping");

    MTN_DYN_SMEM(smem);
    // MODE fixes the tap set at compile time (0: runtime): 1 = 1x1 taps at stride 1 in natural row
    // order (every 1x1 conv and Linear: the tap machinery, the pixel decode and the parity-major
    // handling fold away -- 86 instead of 780 instructions before the first load of these short-K
    // tiles), 2 = 3x3 taps in natural row order, 3 = 3x3 taps in parity-major order.
    // 4 = mode 1 whose epilogue multiplies by gelu'(saved pre-activation) (MTN_ACT_GELU_GRAD_): its own
    // instantiation, because the erf / exp of that epilogue cost every other one ~16 VGPRs (the 8-wave
    // 128x128 tile then no longer fits two workgroups per CU).
    constexpr bool PW = MODE == 1 || MODE == 4 || MODE == 5;      // (5 = mode 1 with the token-assembly epilogue)
    const int pm_ = (PW || MODE == 2) ? 0 : (MODE == 3 ? 1 : p.pm);
    const int KH_ = PW ? 1 : (MODE >= 2 ? 3 : p.KH), KW_ = PW ? 1 : (MODE >= 2 ? 3 : p.KW);
    T* const sA = reinterpret_cast<T*>(smem);                       // [NBUF][BM][PITCH_E]  (WIN: unused)
    T* const sB = WIN ? sA : sA + NBUF * BM * PITCH_E;              // [NBUF][BN][PITCH_E]
    // WIN: window rows per channel slab (multiple of 8), resident 128-byte channel slabs, their
    // number of passes over Cin, and the LDS regions [CS][WR8][BK] window + one zero row
    const int win_halo = WIN ? p.W + 1 : 0;
    const int WR8 = (BM + 2 * win_halo + 7) & ~7;
    const int CS = WIN ? p.win_cs : 1;
    T* const sWin = sB + WOFF * BK;
    T* const sZero = sWin + CS * WR8 * BK;
    CT* const sC = reinterpret_cast<CT*>(smem);                     // [BM][CP] (after the K loop)

    STAMP(st_entry);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WGN, wn = wid % WGN;

    // XCD-aware tile order: workgroups that share an XCD (same bid % 8) get consecutive tiles,
    // n fastest, so the re-reads of one A row-panel by its N/BN tiles hit that XCD's L2.
    const int tiles_n = p.N / BN;
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    // (multiply-high divisions throughout the prologue: a runtime integer division is ~25
    // instructions and the prologue of a short-K tile is a fifth of its lifetime)
    int tile_m = tiles_n == 1 ? bid : (int)__umulhi((unsigned)bid, p.magic_tn);
    const int tile_n = bid - tile_m * tiles_n;
    if (pm_) {
        // parity-major rows: the four parity classes of one image region read the same rows of the
        // gradient; run them back to back (same XCD, adjacent in time) instead of one class sweep
        // after the other, so those rows are fetched from HBM once instead of once per class
        const int tm = nwg / tiles_n;
        if ((tm & 3) == 0 && p.M == tm * BM) tile_m = (tile_m & 3) * (tm >> 2) + (tile_m >> 2);
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const T* __restrict__ gA = reinterpret_cast<const T*>(p.A);
    const T* __restrict__ gW = reinterpret_cast<const T*>(p.Wp);

    // ---- per-thread loader state: RA rows of A (pixel decode), RB rows of W -------------------
    // Row order: natural (b, oy, ox), or -- for stride-2 data gradients (pm_) -- parity-major
    // (oy&1, ox&1, b, oy/2, ox/2): a tile then holds pixels of ONE parity class, for which only
    // 1, 2 or 4 of the 9 taps can hit a stored gradient, and the K loop skips the others
    // (2.25 taps on average instead of 9).  Output rows are always written at the natural index.
    // The first row of the tile is decoded once with real divisions (m0 is workgroup-uniform); a
    // row r of the tile is then (x0 + r) carried into y and b with multiply-high "magic" divisions
    // (exact for operands < 2^16, host-checked), which removes ~6 integer divisions per row from
    // both the loader prologue and the epilogue.
    const int dW_ = pm_ ? (p.OW >> 1) : p.OW, dH_ = pm_ ? (p.OH >> 1) : p.OH;
    int q0 = 0, b0, y0, x0;
    {
        int r = m0;
        if (pm_) {
            const int per = p.B * dH_ * dW_;
            q0 = (r >= per) + (r >= 2 * per) + (r >= 3 * per);
            r -= q0 * per;
        }
        const unsigned yt = dW_ == 1 ? (unsigned)r : __umulhi((unsigned)r, p.magic_w);        // r / dW_
        x0 = r - (int)yt * dW_;
        b0 = dH_ == 1 ? (int)yt : (int)__umulhi(yt, p.magic_h);                                // yt / dH_
        y0 = (int)yt - b0 * dH_;
    }
    // 1x1 taps at stride 1 read input pixel m itself: no decode, no padding
    const bool pointwise = PW || (KH_ * KW_ == 1 && p.mul == 1 && p.off == 0 && p.div == 1 && !pm_);
    auto decode = [&](int m, int& b, int& oy, int& ox) {
        const unsigned xt = (unsigned)(x0 + (m - m0));
        const unsigned yi = dW_ == 1 ? xt : __umulhi(xt, p.magic_w);       // xt / dW_
        const unsigned yt = (unsigned)y0 + yi;
        const unsigned bi = dH_ == 1 ? yt : __umulhi(yt, p.magic_h);       // yt / dH_
        int x = (int)(xt - yi * (unsigned)dW_), y = (int)(yt - bi * (unsigned)dH_);
        b = b0 + (int)bi;
        if (pm_) {
            int q = q0;
            while (b >= p.B) { b -= p.B; ++q; }        // a tile may straddle parity classes
            oy = 2 * y + (q >> 1);
            ox = 2 * x + (q & 1);
        } else {
            oy = y;
            ox = x;
        }
    };
    // row/chunk owned by this thread in each of its RA (RB) loads.  Register staging: rows
    // tid/8 + 32 i, chunk tid%8.  LDS-DMA: wave w, instruction i fills the 1 KiB block of rows
    // 8*(w*RA+i).., lane l lands at (row l/8, slot l%8) and FETCHES chunk slot ^ (row & 7).
    const int chunk = DMA ? ((lane & 7) ^ ((lane >> 3) & 7)) : (tid & 7);
    const int row0 = DMA ? (wid * RA * 8 + (lane >> 3)) : (tid >> 3);
    const int rstep_a = DMA ? 8 : RSTEP;
    const int brow0 = DMA ? (wid * RB * 8 + (lane >> 3)) : (tid >> 3);
    const int rstep_b = DMA ? 8 : RSTEP;
    const int dshift = p.div >> 1, dmask = p.div - 1;     // div in {1,2}
    int a_y[RA], a_x[RA], a_base[RA];
    unsigned a_vy[RA], a_vx[RA];                           // per row: bit kh / kw set iff that tap row / column reads a stored pixel
    unsigned my_taps = 0;
    MTN_UNROLL for (int i = 0; i < RA; ++i) {
        const int m = m0 + row0 + i * rstep_a;
        if (!WIN && pointwise && m < p.M) {
            a_y[i] = 0; a_x[i] = 0; a_base[i] = m; a_vy[i] = 1u; a_vx[i] = 1u;
        } else if (!WIN && m < p.M) {
            int b, oy, ox;
            decode(m, b, oy, ox);
            a_y[i] = oy * p.mul + p.off;
            a_x[i] = ox * p.mul + p.off;
            a_base[i] = b * p.H * p.W;
            // validity is separable: tap (kh, kw) is readable iff its row kh and its column kw are
            unsigned vy = 0, vx = 0;
            MTN_UNROLL for (int t = 0; t < 3; ++t) {
                const int ty = a_y[i] + t * p.sgn, tx = a_x[i] + t * p.sgn;
                if (t < KH_ && ty >= 0 && (ty & dmask) == 0 && (ty >> dshift) < p.H) vy |= 1u << t;
                if (t < KW_ && tx >= 0 && (tx & dmask) == 0 && (tx >> dshift) < p.W) vx |= 1u << t;
            }
            a_vy[i] = vy; a_vx[i] = vx;
            if (pm_)
                MTN_UNROLL for (int t = 0; t < 9; ++t)
                    if (t < KH_ * KW_ && (((vy >> (t / 3)) & (vx >> (t % 3))) & 1u)) my_taps |= 1u << t;
        } else {
            a_y[i] = -(1 << 28); a_x[i] = -(1 << 28); a_base[i] = 0; a_vy[i] = 0; a_vx[i] = 0;
        }
    }
    // taps that touch at least one row of this tile (workgroup-wide OR through LDS); only the
    // parity-major data gradient can skip taps, everything else runs all of them
    unsigned tile_taps = (1u << (KH_ * KW_)) - 1u;
    const bool one_class = pm_ && (p.B * dH_ * dW_) % BM == 0;   // tiles never straddle parity classes
    if (one_class) {
        // the taps a parity class can hit follow from the class alone (rows at the image border
        // only lose some of them, which their per-row masks already say): no workgroup-wide OR
        unsigned vy = 0, vx = 0;
        MTN_UNROLL for (int t = 0; t < 3; ++t) {
            if ((((q0 >> 1) * p.mul + t * p.sgn + p.off) & dmask) == 0) vy |= 1u << t;
            if ((((q0 & 1) * p.mul + t * p.sgn + p.off) & dmask) == 0) vx |= 1u << t;
        }
        tile_taps = 0;
        MTN_UNROLL for (int t = 0; t < 9; ++t)
            if (((vy >> (t / 3)) & (vx >> (t % 3))) & 1u) tile_taps |= 1u << t;
    } else if (pm_) {
        int* s_taps = reinterpret_cast<int*>(smem);
        if (tid == 0) *s_taps = 0;
        __syncthreads();
        if (my_taps) atomicOr(s_taps, (int)my_taps);
        __syncthreads();
        tile_taps = (unsigned)*s_taps;
        __syncthreads();
    }
    unsigned long long tap_list = 0;                       // 4 bits per listed tap (no indexed array)
    int ntaps = 0;
    for (int t = 0; t < 9; ++t)
        if (tile_taps & (1u << t)) { tap_list |= (unsigned long long)t << (4 * ntaps); ++ntaps; }
    const int spt = p.Cin / BK;                            // K steps per tap

    chunk16 ra[RA], rb[RB];
    auto load_tiles = [&](int kt) {
        const int ti = kt / spt, c0 = (kt - ti * spt) * BK;
        const int tap = (int)((tap_list >> (4 * ti)) & 15ull);
        const int k0 = tap * p.Cin + c0;
        const int kh = tap / KW_, kw = tap - kh * KW_;
        MTN_UNROLL for (int i = 0; i < RA; ++i) {
            const int ty = a_y[i] + kh * p.sgn, tx = a_x[i] + kw * p.sgn;
            const int iy = ty >> dshift, ix = tx >> dshift;
            const bool ok = (ty >= 0) && (tx >= 0) && ((ty & dmask) == 0) && ((tx & dmask) == 0) &&
                            (iy < p.H) && (ix < p.W);
            if (ok) {
                const long off = (long)(a_base[i] + iy * p.W + ix) * p.lda + c0 + chunk * VEC;
                ra[i] = *reinterpret_cast<const chunk16*>(gA + off);
            } else {
                ra[i] = zero_chunk();
            }
        }
        MTN_UNROLL for (int i = 0; i < RB; ++i) {
            const long off = (long)(n0 + brow0 + i * rstep_b) * p.K + k0 + chunk * VEC;
            rb[i] = *reinterpret_cast<const chunk16*>(gW + off);
        }
    };
    // LDS-DMA issue of K step kt into buffer `stage` (no registers, no LDS stores)
    const BufDesc bufA = make_buf(p.A, p.a_bytes), bufW = make_buf(p.Wp, p.w_bytes);
    // (K steps are issued in order: the tap index / channel offset of the next one are carried,
    // not divided out of kt every step)
    int ld_ti = 0, ld_c0 = 0;
    auto dma_tiles = [&](int kt, int stage) {
        const int ti = ld_ti, c0 = ld_c0;
        ld_c0 += BK;
        if (ld_c0 == p.Cin) { ld_c0 = 0; ++ld_ti; }
        const int tap = (int)((tap_list >> (4 * ti)) & 15ull);
        const int k0 = tap * p.Cin + c0;
        const int kh = tap / KW_, kw = tap - kh * KW_;
        char* a = reinterpret_cast<char*>(sA + stage * BM * PITCH_E);
        char* b = reinterpret_cast<char*>(sB + stage * BN * PITCH_E);
        const int cc = c0 + chunk * VEC;
        MTN_UNROLL for (int i = 0; i < RA; ++i) {
            const int iy = (a_y[i] + kh * p.sgn) >> dshift, ix = (a_x[i] + kw * p.sgn) >> dshift;
            const unsigned e = (unsigned)((a_base[i] + iy * p.W + ix) * p.lda + cc) * (unsigned)sizeof(T);
            buf_glds16(bufA, (((a_vy[i] >> kh) & (a_vx[i] >> kw)) & 1u) ? e : 0xFFFFFFF0u, a + (wid * RA + i) * 1024);
        }
        if constexpr (STAGES != 8) {
            MTN_UNROLL for (int i = 0; i < RB; ++i)
                buf_glds16(bufW, (unsigned)((n0 + brow0 + i * rstep_b) * p.K + k0 + chunk * VEC) * (unsigned)sizeof(T),
                           b + (wid * RB + i) * 1024);
        }
    };
    constexpr bool X3 = std::is_same<T, f32x3_t>::value;
    auto store_tiles = [&](int stage) {
        T* a = sA + stage * BM * PITCH;
        T* b = sB + stage * BN * PITCH;
        // (split storage: the 128 bytes of a K step are one block [32 hi][32 lo] as they come from memory -- copied like any
        //  other row; `compute` reads the hi fragment at 16 * g and the lo fragment 64 bytes further)
        MTN_UNROLL for (int i = 0; i < RA; ++i)
            *reinterpret_cast<chunk16*>(a + (row0 + i * RSTEP) * PITCH + chunk * VEC) = ra[i];
        MTN_UNROLL for (int i = 0; i < RB; ++i)
            *reinterpret_cast<chunk16*>(b + (row0 + i * RSTEP) * PITCH + chunk * VEC) = rb[i];
    };

    f32x4_t acc[FM][FN];
    MTN_UNROLL for (int i = 0; i < FM; ++i)
        MTN_UNROLL for (int j = 0; j < FN; ++j) acc[i][j] = zero4();

    const int KT = ntaps * spt;
    const int frow = lane & 15;
    const int fg = lane >> 4;              // k group of this lane inside a fragment (RowFrag, mtn_types.h)
    const int fk = fg * (KI / 4);

    // One K step of MFMAs with the fragment reads software-pipelined by hand: W fragments of a
    // k-substep, then A fragments two ahead of the row of MFMAs that consumes them.  The scheduler
    // fences pin that order -- left alone, hipcc reads one A fragment, waits for it, issues its
    // row of MFMAs, reads the next (in-kernel stamps: 640-1100 cycles per step for 512 of MFMA).
    auto mma_step = [&](auto&& loadA, auto&& loadB) {
        constexpr int KS = BK / KI, NA = KS * FM;
        typename Mma<T>::Frag fb[KS][FN], fa[NA];
        MTN_UNROLL for (int j = 0; j < FN; ++j) fb[0][j] = loadB(0, j);
        fa[0] = loadA(0, 0);
        if (NA > 1) fa[1] = loadA(1 / FM, 1 % FM);
        MTN_SCHED_FENCE();
        MTN_UNROLL for (int k = 0; k < NA; ++k) {
            if (k % FM == 0 && k / FM + 1 < KS) {
                MTN_UNROLL for (int j = 0; j < FN; ++j) fb[k / FM + 1][j] = loadB(k / FM + 1, j);
            }
            if (k + 2 < NA) fa[k + 2] = loadA((k + 2) / FM, (k + 2) % FM);
            MTN_UNROLL for (int j = 0; j < FN; ++j) acc[k % FM][j] = Mma<T>::mma(fb[k / FM][j], fa[k], acc[k % FM][j]);
            MTN_SCHED_FENCE();
        }
    };
    // WIN: per fragment row of this lane, which tap rows / columns read a stored pixel
    unsigned f_vy[FM], f_vx[FM];
    if (WIN) {
        MTN_UNROLL for (int i = 0; i < FM; ++i) {
            const int m = m0 + wm * WM + i * 16 + frow;
            unsigned vy = 0, vx = 0;
            if (m < p.M) {
                int b, oy, ox;
                decode(m, b, oy, ox);
                MTN_UNROLL for (int t = 0; t < 3; ++t) {
                    const int ty = oy + t * p.sgn + p.off, tx = ox + t * p.sgn + p.off;
                    if (ty >= 0 && ty < p.H) vy |= 1u << t;
                    if (tx >= 0 && tx < p.W) vx |= 1u << t;
                }
            }
            f_vy[i] = vy; f_vx[i] = vx;
        }
        if (tid < 128 / 4) reinterpret_cast<unsigned*>(sZero)[tid] = 0u;
    }
    const int win_total = p.B * p.H * p.W;                 // rows of the gather source
    // DMA of channel slab `slab` (CS x BK channels) of the window rows into sWin
    auto dma_window = [&](int slab) {
        const int ppc = WR8 >> 3;                          // 1 KiB pieces per BK-channel slab
        for (int q = wid; q < CS * ppc; q += NT / 64) {
            const int cs = q / ppc, pr = q - cs * ppc;
            const int j = pr * 8 + (lane >> 3);
            const int g = m0 - win_halo + j;
            const int ch = (slab * CS + cs) * BK + (((lane & 7) ^ (lane >> 3)) * VEC);
            const unsigned e = (unsigned)(g * p.lda + ch) * (unsigned)sizeof(T);
            buf_glds16(bufA, (g >= 0 && g < win_total) ? e : 0xFFFFFFF0u,
                       reinterpret_cast<char*>(sWin + (cs * WR8 + pr * 8) * BK));
        }
    };
    // W tile of window-mode K step kk of slab `slab` (order: tap outer, resident channel slab inner)
    auto dma_w = [&](int slab, int kk, int stage) {
        const int tap = kk / CS, cs = kk - tap * CS;
        const int k0 = tap * p.Cin + (slab * CS + cs) * BK;
        char* b = reinterpret_cast<char*>(sB + stage * BN * BK);
        MTN_UNROLL for (int i = 0; i < RB; ++i)
            buf_glds16(bufW, (unsigned)((n0 + brow0 + i * rstep_b) * p.K + k0 + chunk * VEC) * (unsigned)sizeof(T),
                       b + (wid * RB + i) * 1024);
    };
    auto compute_win = [&](int kk, int stage) {
        const int tap = kk / CS, cs = kk - tap * CS;
        const int kh = tap / 3, kw = tap - kh * 3;
        const int delta = (kh * p.sgn + p.off) * p.W + (kw * p.sgn + p.off) + win_halo;
        const T* b = sB + stage * BN * BK + (wn * WN + frow) * BK;
        const int swb = frow & 7;
        // A fragment positions as element offsets from the LDS base (window row or the zero row)
        int aoff[FM];
        int asw[FM];
        MTN_UNROLL for (int i = 0; i < FM; ++i) {
            const int j = wm * WM + i * 16 + frow + delta;
            const bool ok = ((f_vy[i] >> kh) & (f_vx[i] >> kw)) & 1u;
            aoff[i] = ok ? (WOFF + cs * WR8 + j) * BK : (WOFF + CS * WR8) * BK;
            asw[i] = j & 7;
        }
        mma_step([&](int ks, int i) { return RowFrag<T>::load(sB + aoff[i], ks, fg, asw[i]); },
                 [&](int ks, int j) { return RowFragW<T>::load(b + j * 16 * BK, ks, fg, swb); });
    };

    auto compute = [&](int stage) {
        if (DMA) {
            // fragment element offset ks*KI+fk -> 16-byte slot (ks*KI+fk)/VEC, swizzled by row & 7
            const T* a = sA + stage * BM * BK + (wm * WM + frow) * BK;
            const T* b = sB + stage * BN * BK + (wn * WN + frow) * BK;
            const int sw = frow & 7;                       // (row & 7): tile bases are multiples of 16
            mma_step([&](int ks, int i) { return RowFrag<T>::load(a + i * 16 * BK, ks, fg, sw); },
                     [&](int ks, int j) { return RowFragW<T>::load(b + j * 16 * BK, ks, fg, sw); });
            return;
        }
        if constexpr (X3) {
            // split storage: fragments are plain bf16x8 reads of the hi and the lo half of a row
            const char* a = reinterpret_cast<const char*>(sA + stage * BM * PITCH + (wm * WM + frow) * PITCH) + 16 * fg;
            const char* b = reinterpret_cast<const char*>(sB + stage * BN * PITCH + (wn * WN + frow) * PITCH) + 16 * fg;
            Frag3 fa[FM], fb[FN];
            MTN_UNROLL for (int i = 0; i < FM; ++i) {
                fa[i].hi = load_bf16x8(a + i * 16 * PITCH_BYTES);
                fa[i].lo = load_bf16x8(a + i * 16 * PITCH_BYTES + 64);
            }
            MTN_UNROLL for (int j = 0; j < FN; ++j) {
                fb[j].hi = load_bf16x8(b + j * 16 * PITCH_BYTES);
                fb[j].lo = load_bf16x8(b + j * 16 * PITCH_BYTES + 64);
            }
            MTN_UNROLL for (int i = 0; i < FM; ++i)
                MTN_UNROLL for (int j = 0; j < FN; ++j) acc[i][j] = Mma<T>::mma(fb[j], fa[i], acc[i][j]);
            return;
        }
        const T* a = sA + stage * BM * PITCH + (wm * WM + frow) * PITCH + fk;
        const T* b = sB + stage * BN * PITCH + (wn * WN + frow) * PITCH + fk;
        MTN_UNROLL for (int ks = 0; ks < BK / KI; ++ks) {
            typename Mma<T>::Frag fa[FM], fb[FN];
            MTN_UNROLL for (int i = 0; i < FM; ++i) fa[i] = Mma<T>::load(a + i * 16 * PITCH + ks * KI);
            MTN_UNROLL for (int j = 0; j < FN; ++j) fb[j] = Mma<T>::load(b + j * 16 * PITCH + ks * KI);
            MTN_UNROLL for (int i = 0; i < FM; ++i)
                MTN_UNROLL for (int j = 0; j < FN; ++j) acc[i][j] = Mma<T>::mma(fb[j], fa[i], acc[i][j]);
        }
    };
#ifdef MTN_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long st_begin = __builtin_amdgcn_s_memtime();
#endif
    if constexpr (IMG) {
        // STAGES = 6: the window holds ALL Cin channels of the tile's rows, loaded once; a wave's weight
        // fragments (its own WN columns) come straight from global memory into registers, D K-steps ahead (the
        // compiler counts vmcnt for register loads), and the K loop has NO barrier and no LDS writes -- the
        // waves of the workgroup drift apart and overlap each other's load and MFMA phases, which a workgroup
        // that meets at a barrier every K step cannot.
        // (split storage at Cin = 256 is eight 32-channel blocks = 174 KB of window: the channels come in two passes of four
        //  blocks -- the activations still cross L2 -> LDS once per tile, the accumulators carry over)
        constexpr int D = 3, KS = BK / KI;
        const int steps = 9 * CS;                          // CS in {1, 2, 4}: a multiple of D... 9, 18, 36
        const int cs_shift = CS == 4 ? 2 : (CS == 2 ? 1 : 0);
        const int passes = p.Cin / (CS * BK);
        typename Mma<T>::Frag wq[D][KS][FN];
        int pass = 0;
        auto load_w = [&](int kk, int slot) {
            const int tap = kk >> cs_shift, cs = kk - (tap << cs_shift);
            // (lane group g: the 16-byte slot g of the K step -- 8 bf16 channels, or the hi half of 8 split-storage channels,
            //  whose lo half Mma<f32x3_t>::load finds 64 bytes on)
            const T* w = gW + (long)(n0 + wn * WN + frow) * p.K + tap * p.Cin + (pass * CS + cs) * BK + fg * VEC;
            MTN_UNROLL for (int ks = 0; ks < KS; ++ks)
                MTN_UNROLL for (int j = 0; j < FN; ++j) wq[slot][ks][j] = Mma<T>::load(w + (long)j * 16 * p.K + ks * KI);
        };
        auto compute_img = [&](int kk, int slot) {
            const int tap = kk >> cs_shift, cs = kk - (tap << cs_shift);
            const int kh = tap / 3, kw = tap - kh * 3;
            const int delta = (kh * p.sgn + p.off) * p.W + (kw * p.sgn + p.off) + win_halo;
            int aoff[FM], asw[FM];
            MTN_UNROLL for (int i = 0; i < FM; ++i) {
                const int j = wm * WM + i * 16 + frow + delta;
                const bool ok = ((f_vy[i] >> kh) & (f_vx[i] >> kw)) & 1u;
                aoff[i] = ok ? (cs * WR8 + j) * BK : CS * WR8 * BK;
                asw[i] = j & 7;
            }
            mma_step([&](int ks, int i) { return RowFrag<T>::load(sWin + aoff[i], ks, fg, asw[i]); },
                     [&](int ks, int j) { return wq[slot][ks][j]; });
        };
        for (; pass < passes; ++pass) {
            if (pass) __syncthreads();                     // every wave is done with the previous pass's window
            dma_window(pass);
            __syncthreads();
            MTN_UNROLL for (int d = 0; d + 1 < D; ++d) load_w(d, d);
            for (int kk = 0; kk < steps; kk += D) {
                MTN_UNROLL for (int d = 0; d < D; ++d) {
                    if (kk + d + D - 1 < steps) load_w(kk + d + D - 1, (d + D - 1) % D);
                    compute_img(kk + d, d);
                }
            }
        }
        __syncthreads();                                   // every wave is done with the window: the C tile takes its place
    } else if (WIN) {
        const int slabs = p.Cin / (CS * BK), steps = 9 * CS;
        for (int slab = 0; slab < slabs; ++slab) {
            __syncthreads();                               // the previous window / W tiles are consumed
            STAMP(tw0);
            dma_window(slab);
            dma_w(slab, 0, 0);
            __syncthreads();
            STAMP(tw1);
#ifdef MTN_STAMPS
            st_acc[0] += tw1 - tw0;
#endif
            for (int kk = 0; kk < steps; ++kk) {
                const int cur = kk & 1;
                STAMP(t0);
                if (kk + 1 < steps) dma_w(slab, kk + 1, cur ^ 1);
                STAMP(t1);
                compute_win(kk, cur);
                STAMP(t2);
                __syncthreads();
                STAMP(t3);
#ifdef MTN_STAMPS
                st_acc[1] += t1 - t0; st_acc[2] += t2 - t1; st_acc[3] += t3 - t2; st_acc[4] += 1;
#endif
            }
        }
    } else if (STAGES == 4) {
        // three buffers, loads TWO tiles ahead: a counted vmcnt leaves the youngest tile in flight
        // across a raw s_barrier (a __syncthreads() here would drain it: its fence waits vmcnt(0))
        if (KT > 0) dma_tiles(0, 0);
        if (KT > 1) dma_tiles(1, 1);
#ifndef MTN_EMU
        if (KT > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RA + RB) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#else
        __syncthreads();
#endif
        int cur = 0;
        for (int kt = 0; kt < KT; ++kt) {
            int nxt2 = cur + 2;
            if (nxt2 >= 3) nxt2 -= 3;
            if (kt + 2 < KT) dma_tiles(kt + 2, nxt2);
            compute(cur);
#ifndef MTN_EMU
            if (kt + 2 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RA + RB) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#else
            __syncthreads();
#endif
            cur = cur + 1 == 3 ? 0 : cur + 1;
        }
        __syncthreads();
    } else if (STAGES == 8) {
        // experiment (debug key 3 = 8): only the A tile goes through LDS (DMA double buffer); a wave's weight fragments
        // come straight from global memory into registers one K step ahead, as in the image-tile kernel -- half the
        // LDS-DMA bytes per step (the per-CU DMA intake is what the K-heavy layers track).  Taps in
        // natural order only (MODE 1, 2, 4): K step kt starts at weight column kt * BK.
        constexpr int KS = BK / KI;
        typename Mma<T>::Frag wq0[KS][FN], wq1[KS][FN];
        const T* wbase = gW + (long)(n0 + wn * WN + frow) * p.K + fk;
        auto load_w = [&](int kt, typename Mma<T>::Frag (&wq)[KS][FN]) {
            const T* w = wbase + kt * BK;
            MTN_UNROLL for (int ks = 0; ks < KS; ++ks)
                MTN_UNROLL for (int j = 0; j < FN; ++j) wq[ks][j] = Mma<T>::load(w + (long)j * 16 * p.K + ks * KI);
        };
        auto compute_w = [&](int stage, typename Mma<T>::Frag (&wq)[KS][FN]) {
            const T* a = sA + stage * BM * BK + (wm * WM + frow) * BK;
            const int sw = frow & 7;
            mma_step([&](int ks, int i) { return RowFrag<T>::load(a + i * 16 * BK, ks, fg, sw); },
                     [&](int ks, int j) { return wq[ks][j]; });
        };
        if (KT > 0) { dma_tiles(0, 0); load_w(0, wq0); }
        __syncthreads();
        for (int kt = 0; kt < KT; kt += 2) {
            if (kt + 1 < KT) { dma_tiles(kt + 1, 1); load_w(kt + 1, wq1); }
            compute_w(0, wq0);
            __syncthreads();
            if (kt + 1 < KT) {
                if (kt + 2 < KT) { dma_tiles(kt + 2, 0); load_w(kt + 2, wq0); }
                compute_w(1, wq1);
                __syncthreads();
            }
        }
    } else if (STAGES == 7) {
        // experiment (debug key 3 = 7): intra-workgroup ping-pong.  Three buffers; every K step has two phases
        // separated by raw barriers: the lower half of the waves multiplies tile kt while the upper half issues its
        // share of the DMA of tile kt+2, then the roles swap -- each SIMD's matrix pipe always has one wave whose
        // program is MFMAs only, instead of every wave of the workgroup paying the DMA issue cycles at the same time.
        constexpr int HALF = WGM * WGN / 2;
        const bool lower = wid < HALF;
        if (KT > 0) dma_tiles(0, 0);
        if (KT > 1) dma_tiles(1, 1);
        auto meet = [&](bool newest_in_flight) {
#ifndef MTN_EMU
            if (newest_in_flight) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RA + RB) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#else
            __syncthreads();
#endif
        };
        meet(KT > 1);
        int cur = 0;
        for (int kt = 0; kt < KT; ++kt) {
            int nxt2 = cur + 2;
            if (nxt2 >= 3) nxt2 -= 3;
            const bool more = kt + 2 < KT;
            if (lower) compute(cur);
            else if (more) dma_tiles(kt + 2, nxt2);
            // a wave's youngest batch is tile kt+2 (upper half, just issued) or tile kt+1 (lower half, issued one
            // phase ago): leaving it in flight, everything older -- tile kt+1 for the upper half -- has landed
            meet(lower ? kt + 1 < KT : more);
            if (lower) { if (more) dma_tiles(kt + 2, nxt2); }
            else compute(cur);
            meet(lower ? more : kt + 1 < KT);
            cur = cur + 1 == 3 ? 0 : cur + 1;
        }
        __syncthreads();
    } else if (DMA) {
        // tile kt+1 streams into the other buffer while tile kt is multiplied; the barrier's
        // vmcnt(0) is exactly the wait the next step needs (one tile in flight)
        if (KT > 0) dma_tiles(0, 0);
        __syncthreads();
        for (int kt = 0; kt < KT; ++kt) {
            const int cur = kt & 1;
            STAMP(t0);
            if (kt + 1 < KT) dma_tiles(kt + 1, cur ^ 1);
#ifndef MTN_EMU
            // experiment (debug key 8 = 5): drain the DMA just issued before this step's fragment reads, i.e. give up the
            // overlap of a workgroup's own DMA with its own MFMAs the way wgrad's one-array pipeline does by accident
            if (p.dbg == 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            STAMP(t1);
            compute(cur);
            STAMP(t2);
            __syncthreads();
            STAMP(t3);
#ifdef MTN_STAMPS
            st_acc[1] += t1 - t0; st_acc[2] += t2 - t1; st_acc[3] += t3 - t2; st_acc[4] += 1;
#endif
        }
    } else
    if (STAGES == 2) {
        if (KT > 0) {
            load_tiles(0);
            store_tiles(0);
        }
        __syncthreads();
        for (int kt = 0; kt < KT; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < KT) load_tiles(kt + 1);
            compute(cur);
            if (kt + 1 < KT) store_tiles(cur ^ 1);
            __syncthreads();
        }
    } else {
        if (KT > 0) load_tiles(0);
        for (int kt = 0; kt < KT; ++kt) {
            store_tiles(0);
            __syncthreads();
            if (kt + 1 < KT) load_tiles(kt + 1);      // in flight while this step computes
            compute(0);
            __syncthreads();
        }
    }

#ifdef MTN_STAMPS
    const unsigned long long st_loop_end = __builtin_amdgcn_s_memtime();
#endif
    // ---- epilogue: accumulators -> LDS (f32) -> row-wise fused math -> 16-byte stores ----------
    T* __restrict__ gO = reinterpret_cast<T*>(p.Out);
    T* __restrict__ gAux = reinterpret_cast<T*>(p.aux_out);
    const T* __restrict__ gR = reinterpret_cast<const T*>(p.res);
    constexpr int CPR = BN / EV;      // channel groups per row
    constexpr int TR = NT / CPR;      // thread rows: a thread keeps ONE column group for all its rows
    constexpr int RPT = BM / TR;      // rows per thread
    static_assert(NT % CPR == 0 && BM % TR == 0, "store mapping");
    const int ccol = (tid % CPR) * EV, trow = tid / CPR;
    // output row of each of this thread's rows (-1: beyond M)
    int orow[RPT];
    MTN_UNROLL for (int i = 0; i < RPT; ++i) {
        int m = m0 + trow + i * TR;
        if (m >= p.M) m = -1;
        else if (pm_) {
            int b, oy, ox;
            decode(m, b, oy, ox);
            m = (b * p.OH + oy) * p.OW + ox;
        }
        orow[i] = m;
    }
    // Data gradient feeding a BatchNorm unit's backward (IgemmArgs::bn_y): that unit's per-channel constants
    // for this thread's channels, and its raw-output values of this thread's rows -- requested HERE,
    // before the accumulators are staged, so their latency hides behind the staging and its barrier.
    const T* __restrict__ gBy = reinterpret_cast<const T*>(p.bn_y);
    const T* __restrict__ gBr = reinterpret_cast<const T*>(p.bn_res);
    float bn_sc[EV], bn_sh[EV], bn_mu[EV], bn_rs[EV];
    RawVec<T> bn_yv[RPT];
    if (gBy) {
        MTN_UNROLL for (int e = 0; e < EV; ++e) {
            bn_sc[e] = p.bn_ss[n0 + ccol + e]; bn_sh[e] = p.bn_ss[p.N + n0 + ccol + e];
            bn_mu[e] = p.bn_mr[n0 + ccol + e]; bn_rs[e] = p.bn_mr[p.N + n0 + ccol + e];
        }
        MTN_UNROLL for (int i = 0; i < RPT; ++i) {
            const long m = orow[i] < 0 ? 0 : orow[i];
            bn_yv[i] = ld_raw<T>(gBy + m * p.bn_ldy + n0, ccol);
        }
    }
    // The MFMAs take the W fragment as their row operand, so a lane's four accumulator registers
    // are four CONSECUTIVE channels of one pixel: one 8- / 16-byte LDS store per fragment instead
    // of four scattered element stores.
    MTN_UNROLL for (int i = 0; i < FM; ++i)
        MTN_UNROLL for (int j = 0; j < FN; ++j)
            store4<CT>(sC + (wm * WM + i * 16 + (lane & 15)) * CP + wn * WN + j * 16 + 4 * (lane >> 4), acc[i][j]);
    __syncthreads();
    STAMP(st_e1);

    float st_s[EV], st_q[EV];
    MTN_UNROLL for (int e = 0; e < EV; ++e) { st_s[e] = 0.f; st_q[e] = 0.f; }
    MTN_UNROLL for (int i = 0; i < RPT; ++i) {
        const int row = trow + i * TR;
        const int m = orow[i];
        if (m < 0) continue;
        float v[EV];
        MTN_UNROLL for (int e = 0; e < EV; ++e) v[e] = to_f<CT>(sC[row * CP + ccol + e]);
        if (p.bias) {
            MTN_UNROLL for (int e = 0; e < EV; ++e) v[e] += p.bias[n0 + ccol + e];
        }
        if (p.act == MTN_ACT_GELU) {
            if (gAux) st_vec<T>(gAux + (long)m * p.ld_aux + n0, ccol, v);
            MTN_UNROLL for (int e = 0; e < EV; ++e) v[e] = geluf_(v[e]);
        } else if (MODE == 4) {                        // GELU backward on a data gradient: aux holds the saved pre-activation
            float av[EV];
            ld_vec<T>(gAux + (long)m * p.ld_aux + n0, ccol, av);
            MTN_UNROLL for (int e = 0; e < EV; ++e) v[e] *= gelu_grad_(av[e]);
        }
        if (gR) {
            float rr[EV];
            ld_vec<T>(gR + (long)m * p.ldr + n0, ccol, rr);
            MTN_UNROLL for (int e = 0; e < EV; ++e) v[e] += rr[e];
        }
        if (p.act == MTN_ACT_SILU) {                   // inference with folded BatchNorm: SiLU(conv + shift [+ res])
            MTN_UNROLL for (int e = 0; e < EV; ++e) v[e] = siluf_(v[e]);
        }
        if constexpr (MODE == 5) {
            // projection -> tokens: + pos[p], token-major row b*(P+1) + 1 + p; the image's first patch row also writes the class token
            const unsigned bi = p.tok_P == 1 ? (unsigned)m : __umulhi((unsigned)m, p.tok_magic), pp = (unsigned)m - bi * (unsigned)p.tok_P;
            MTN_UNROLL for (int e = 0; e < EV; ++e) v[e] += p.tok_pos[(long)pp * p.N + n0 + ccol + e];
            const long trow_ = (long)m + bi + 1;
            st_vec<T>(gO + trow_ * p.ldc + n0, ccol, v);
            if (pp == 0) {
                float c[EV];
                MTN_UNROLL for (int e = 0; e < EV; ++e) c[e] = p.tok_cls[n0 + ccol + e];
                st_vec<T>(gO + (trow_ - 1) * p.ldc + n0, ccol, c);
            }
            continue;
        }
        const RawVec<T> o = pack_raw<T>(v);
        st_raw<T>(gO + (long)m * p.ldc + n0, ccol, o);
        if (p.stats) {          // statistics of the values exactly as stored
            float w[EV];
            unpack_raw<T>(o, w);
            if (gBy) {          // sum dz, sum dz*xhat of the BatchNorm unit whose output gradient this row is
                float yv[EV], rr[EV], dz[EV], xh[EV];
                unpack_raw<T>(bn_yv[i], yv);
                if (gBr) ld_vec<T>(gBr + (long)m * p.bn_ldr + n0, ccol, rr);
                bn_dz_math<EV>(w, yv, rr, gBr != nullptr, bn_sc, bn_sh, bn_mu, bn_rs, p.bn_act == MTN_ACT_SILU, dz, xh);
                MTN_UNROLL for (int e = 0; e < EV; ++e) { st_s[e] += dz[e]; st_q[e] += dz[e] * xh[e]; }
            } else {
                MTN_UNROLL for (int e = 0; e < EV; ++e) { st_s[e] += w[e]; st_q[e] += w[e] * w[e]; }
            }
        }
    }
    STAMP(st_e2);
    if (p.stats) {
        __syncthreads();                                   // everyone is done reading the C tile
        float* red = reinterpret_cast<float*>(smem);       // [2][TR][BN]
        MTN_UNROLL for (int e = 0; e < EV; ++e) {
            red[trow * BN + ccol + e] = st_s[e];
            red[(TR + trow) * BN + ccol + e] = st_q[e];
        }
        __syncthreads();
        for (int i = tid; i < 2 * BN; i += NT) {
            const int which = i / BN, c = i - which * BN;
            float t = 0.f;
            MTN_UNROLL for (int q = 0; q < TR; ++q) t += red[(which * TR + q) * BN + c];
            p.stats[((long)tile_m * 2 + which) * p.N + n0 + c] = t;
        }
    }
#ifdef MTN_STAMPS
    // p.dbg == 4: stamp the LAST 256 workgroups (steady state) instead of the first (cold start)
    const int st_first = p.dbg == 4 ? (int)gridDim.x - 256 : 0;
    if (tid == 0 && (int)blockIdx.x >= st_first && (int)blockIdx.x < st_first + 256) {
        const unsigned long long st_end = __builtin_amdgcn_s_memtime();
        unsigned long long* o = g_stamps + ((int)blockIdx.x - st_first) * 16;
        o[0] = st_acc[0]; o[1] = st_acc[1]; o[2] = st_acc[2]; o[3] = st_acc[3]; o[4] = st_acc[4];
        o[5] = st_loop_end - st_begin; o[6] = st_end - st_loop_end; o[7] = st_begin - st_entry;
        o[8] = st_e1 - st_loop_end; o[9] = st_e2 - st_e1; o[10] = st_end - st_e2;
    }
#endif
}

// LDS bytes of the window mode: two W tiles + the window rows of the resident channel slabs + a zero row
static size_t igemm_window_bytes(const IgemmArgs& p, int bm, int bn, int cs) {      // bn = 0: no W tiles in LDS (STAGES = 6)
    const size_t wr8 = (size_t)((bm + 2 * (p.W + 1) + 7) & ~7);
    return (2 * (size_t)bn + cs * wr8 + 1) * ROW_BYTES;
}
static int g_window_limit_kb = 53;
// resident channel slabs of the window: two where they fit the LDS limit, else one (fp32 storage, 64-channel layers of
// stage 1: two 32-channel slabs are 59 KB of window, one is 30 KB and keeps three workgroups per CU)
static int igemm_window_cs(const IgemmArgs& p, int esz, int bm, int bn) {
    if (p.Cin / (ROW_BYTES / esz) < 2) return 1;
    return igemm_window_bytes(p, bm, bn, 2) <= (size_t)g_window_limit_kb * 1024 ? 2 : 1;
}
static bool igemm_window_shape(const IgemmArgs& p) {
    return p.KH == 3 && p.KW == 3 && p.mul == 1 && p.div == 1 && !p.pm && p.OH == p.H && p.OW == p.W;
}

// window mode serves stride-1 3x3 taps whose LDS footprint still lets three workgroups share a CU
// (53 KB): measured per layer it wins 25 % there (64-channel stage-1
// layers) and loses to the plain LDS-DMA pipeline once the window costs a resident workgroup.
// Debug key 9 raises the limit (tests run every shape through the window path).
static int g_tile_n64 = 128;     // rows per tile for 64-wide outputs (debug key 4)
static int g_igemm_stages = 5;   // window mode where it pays, LDS-DMA double buffer elsewhere
static bool igemm_window_ok(const IgemmArgs& p, int bm, int bn, int esz) {
    const int bk = ROW_BYTES / esz;
    const int cs = igemm_window_cs(p, esz, bm, bn);
    return igemm_window_shape(p) && p.Cin % (cs * bk) == 0 && igemm_window_bytes(p, bm, bn, cs) <= (size_t)g_window_limit_kb * 1024;
}

// "Image tiles": window mode with the WHOLE reduction axis resident -- a 144-row x 256-column tile (one 12x12
// image of the stage-3 maps per workgroup, 8 waves side by side over the columns, one workgroup per CU) keeps
// its 144 + 2(W+1) window rows of all Cin <= 256 channels in LDS (<= 90 KB) beside two 256 x 128-byte weight
// tiles: the activations cross L2 -> LDS once per tile instead of once per tap and column tile, and only
// weights stream (2.25x the FLOPs per byte through the DMA path of the 128x128 double buffer).
static int g_image_tiles = 1;      // debug key 16: 0 = off, 1 = on, 3 = on and ignore the tile-count floor (tests)
constexpr int IMG_BM = 144, IMG_BN = 256;
// (Measured and dropped: 288 x 128 with waves 2 x 4 for the 24 x 24 stage-2 maps, +10 % slower than the 128 x 128
//  double buffer; 256 x 64 with waves 4 x 2 for stage 1, +25 % slower than the 53 KB window mode -- DESIGN.md 7c.)
// channel blocks resident per pass (0: the image tile does not apply).  bf16: all of Cin in one pass; split storage (esz 4):
// the 256-channel layers take two passes of four 32-channel blocks (round 4)
static int igemm_image_tile_cs(const IgemmArgs& p, int esz) {
    if (!g_image_tiles || (g_igemm_stages != 5 && g_igemm_stages < 7) || !igemm_window_shape(p) || p.N % IMG_BN) return 0;
    const int bk = ROW_BYTES / esz, blocks = p.Cin / bk;
    if (p.Cin % bk) return 0;
    int cs = blocks;
    if (esz == 4 && blocks == 8) cs = 4;
    if (cs != 1 && cs != 2 && cs != 4) return 0;
    if (igemm_window_bytes(p, IMG_BM, 0, cs) > 158 * 1024) return 0;
    const long tiles = (long)mtn_cdiv(p.M, IMG_BM) * (p.N / IMG_BN);
    return (tiles >= 192 || (g_image_tiles & 2)) ? cs : 0;     // one workgroup per CU: fewer tiles leave CUs idle
}
static bool igemm_image_tile_ok(const IgemmArgs& p, int dtype) {
    if (dtype != MTN_BF16 && dtype != MTN_F32X3) return false;
    return igemm_image_tile_cs(p, dtype == MTN_BF16 ? 2 : 4) != 0;
}

template <class T, int BM, int BN, int WGM, int WGN, int STAGES, int MODE = 0>
int launch_igemm(const IgemmArgs& p, hipStream_t stream) {
    constexpr int NT = WGM * WGN * 64;
    constexpr size_t lds_ab = STAGES == 8 ? 2 * (size_t)BM * ROW_BYTES : STAGES >= 3 ? (STAGES == 7 ? 3 : STAGES - 1) * (size_t)(BM + BN) * ROW_BYTES : STAGES * (size_t)(BM + BN) * PITCH_BYTES;
    constexpr size_t csz = STAGES != 2 ? sizeof(T) : 4;
    constexpr size_t lds_c = (size_t)BM * (BN * csz + 16);
    constexpr size_t lds_r = (size_t)NT * Ev<T>::N * 2 * 4;    // stats reduction scratch [2][TR][BN], TR = NT / (BN / Ev<T>::N)
    constexpr size_t lds0 = lds_ab > lds_c ? lds_ab : lds_c;
    size_t lds = lds0 > lds_r ? lds0 : lds_r;
    int win_cs = 0;
    if (STAGES == 5 || STAGES == 6) {
        win_cs = STAGES == 6 ? igemm_image_tile_cs(p, (int)sizeof(T)) : ((BM == 128 && BN == 128) ? 2 : igemm_window_cs(p, (int)sizeof(T), BM, BN));
        const size_t win = igemm_window_bytes(p, BM, STAGES == 6 ? 0 : BN, win_cs);
        const size_t rest = lds_c > lds_r ? lds_c : lds_r;
        lds = win > rest ? win : rest;
    }
#ifndef MTN_EMU
    static MtnLdsAttr attr;
    mtn_set_max_lds(attr, &igemm_kernel<T, BM, BN, WGM, WGN, STAGES, MODE>, (STAGES == 5 || STAGES == 6) ? 160 * 1024 : lds);
#endif
    const int tiles = mtn_cdiv(p.M, BM) * (p.N / BN);
    MTN_REQUIRE(!p.stats || p.stats_rows == mtn_cdiv(p.M, BM),
                "igemm: the statistics slab was sized for %d row tiles, this launch writes %d (tile choice changed after "
                "mtn_conv_stats_tiles was queried?)", p.stats_rows, mtn_cdiv(p.M, BM));
    IgemmArgs q = p;
    q.win_cs = win_cs;
    q.magic_tn = p.N / BN > 1 ? (unsigned)(0x100000000ull / (unsigned)(p.N / BN)) + 1u : 0u;
    MTN_REQUIRE((long)tiles * (p.N / BN) < (1L << 32), "igemm: too many tiles");
    MTN_LAUNCH((igemm_kernel<T, BM, BN, WGM, WGN, STAGES, MODE>), dim3(tiles), dim3(NT), lds, stream, q);
    return mtn_check_launch("igemm");
}


// 1x1 taps at stride 1 in natural row order: every 1x1 convolution (forward and data gradient) and Linear
static bool igemm_pointwise(const IgemmArgs& p) {
    return p.KH * p.KW == 1 && p.mul == 1 && p.off == 0 && p.div == 1 && !p.pm;
}

// the default LDS-DMA double-buffer pipeline, specialised on the tap set
template <class T, int BM, int BN, int WGM, int WGN, int STG = 3>
int launch_igemm_dma2(const IgemmArgs& p, hipStream_t stream) {
    if (igemm_pointwise(p) && p.act == MTN_ACT_GELU_GRAD_) return launch_igemm<T, BM, BN, WGM, WGN, STG, 4>(p, stream);
    if constexpr (STG == 3 && (BM * BN == 128 * 128 || (BM == 64 && BN == 128))) {       // (the two tiles a projection can take)
        if (igemm_pointwise(p) && p.tok_pos) return launch_igemm<T, BM, BN, WGM, WGN, STG, 5>(p, stream);
    }
    if (igemm_pointwise(p)) return launch_igemm<T, BM, BN, WGM, WGN, STG, 1>(p, stream);
    if (p.KH == 3 && p.KW == 3) return p.pm ? launch_igemm<T, BM, BN, WGM, WGN, STG, 3>(p, stream) : launch_igemm<T, BM, BN, WGM, WGN, STG, 2>(p, stream);
    return launch_igemm<T, BM, BN, WGM, WGN, STG>(p, stream);
}

template <class T, int BM, int BN, int WGM, int WGN>
int launch_igemm_mode(const IgemmArgs& p, hipStream_t stream) {
    if (g_igemm_stages == 5 || g_igemm_stages >= 7) {
        if (igemm_window_ok(p, BM, BN, (int)sizeof(T))) return launch_igemm<T, BM, BN, WGM, WGN, 5, 2>(p, stream);
        return launch_igemm_dma2<T, BM, BN, WGM, WGN>(p, stream);
    }
#ifdef MTN_EXPERIMENTS       // the pipelines the default replaced (register staging with one / two LDS buffers, three DMA buffers)
    if (g_igemm_stages == 4) return launch_igemm<T, BM, BN, WGM, WGN, 4>(p, stream);
    if (g_igemm_stages == 3) return launch_igemm<T, BM, BN, WGM, WGN, 3>(p, stream);
    if (g_igemm_stages == 2) return launch_igemm<T, BM, BN, WGM, WGN, 2>(p, stream);
    return launch_igemm<T, BM, BN, WGM, WGN, 1>(p, stream);
#else
    return MTN_ERR_UNSUPPORTED;
#endif
}

static int g_igemm_dbg = 0;
static int g_igemm_tile = 0;     // debug key 6: 0 = heuristic, 1 = 128x128, 2 = 64x128, 3 = 128x64 (N % 128 == 0 only)

// 0: 128x128, 1: 64x128, 2: 256x64, 3: 128x64
// 128-wide outputs choose between the 128x128 tile (8 waves, 2 workgroups / CU = 512 slots) and the
// 64x128 tile (4 waves, 3 / CU = 768 slots).  With 8 waves the larger tile wins on every layer of the
// model (short-K ones included: +9..17 %) except where its last round of workgroups is nearly
// empty -- 576..580 tiles on 512 slots (stage-3 3x3, proj, the 256-wide Linears): there the smaller
// tile is 5..10 % faster (per-layer table in DESIGN.md section 7b).
// (K >= 2048 -- the stage-3 3x3 layers where the image tile does not apply, i.e. the fp32-storage modes: the larger tile
// still wins at 0.56 of a round, 176 -> 161 us in the split-bf16 mode; at K = 512 (proj) it does not)
static int igemm_config(int M, int N, int K) {
    if (N % 128 == 0) {
        if (g_igemm_tile == 1 || g_igemm_tile >= 4) return 0;
        if (g_igemm_tile == 2) return 1;
        if (g_igemm_tile == 3) return 3;
        if (M < 128 * 64) return 1;
        const double x128 = (double)mtn_cdiv(M, 128) * (N / 128) / 512.0;
        const double e128 = x128 / (double)(long)(x128 + 0.999999);
        return e128 < (K >= 2048 ? 0.5 : 0.6) ? 1 : 0;
    }
#ifdef MTN_EXPERIMENTS
    if (g_tile_n64 == 256 && M >= 256 * 256) return 2;
#endif
    return 3;
}

// bf16, 64 -> 64 channels: the 256x64 window tile on eight waves, where it fills at least two rounds of its 512 workgroup slots
// (debug key 8 = 7: whatever the size -- tests); the 128x64 tile otherwise
static bool igemm_window256_ok(const IgemmArgs& p) {
    return g_igemm_stages == 5 && g_igemm_dbg != 8 && igemm_window_shape(p) && p.Cin == 64 && p.N == 64 &&
           (p.M >= 256 * 1024 || g_igemm_dbg == 7) && igemm_window_bytes(p, 256, 64, 1) <= 79 * 1024;      // (debug key 8 = 8: off)
}

// bf16, 128x128 tile: window mode with both 64-channel slabs of a 128-channel layer resident, while two workgroups still fit a CU
static bool igemm_window128_ok(const IgemmArgs& p) {
    return g_igemm_stages == 5 && igemm_window_shape(p) && p.Cin == 128 && igemm_window_bytes(p, 128, 128, 2) <= 79 * 1024;
}

template <class T>
int dispatch_igemm(const IgemmArgs& p, hipStream_t stream) {
    if constexpr (sizeof(T) == 2 || std::is_same<T, f32x3_t>::value) {
        if (igemm_image_tile_ok(p, sizeof(T) == 2 ? MTN_BF16 : MTN_F32X3)) return launch_igemm<T, IMG_BM, IMG_BN, 1, 8, 6, 2>(p, stream);
    }
    const int cfg = igemm_config(p.M, p.N, p.K);
    // 128x128 runs with 8 waves (4x2, 32x64 per wave) in the default pipelines: an LDS-DMA
    // instruction costs its wave ~100 cycles of issue, so 4 pieces per wave instead of 8 shorten the
    // load phase of every K step (+5..7 % on these layers; the smaller tiles measured -2..4 %)
#ifdef MTN_EXPERIMENTS       // measured and rejected: kept for their tests, not in the product library
    if constexpr (sizeof(T) == 2) {
        // (debug key 6 = 5 / 6): 256x128 tile on 16 waves (4x4, 64x32 per wave), one workgroup per CU, with two
        // and with three LDS buffers -- a quarter less L2->LDS traffic per flop than the 128x128 tile
        if (cfg == 0 && g_igemm_stages == 5 && g_igemm_tile == 5) return launch_igemm_dma2<T, 256, 128, 4, 4>(p, stream);
        if (cfg == 0 && g_igemm_stages == 5 && g_igemm_tile == 6) return launch_igemm_dma2<T, 256, 128, 4, 4, 4>(p, stream);
        // (debug key 3 = 8): weights through registers, 2x4 waves (64x32 per wave)
        if (cfg == 0 && g_igemm_stages == 8 && !p.pm && (igemm_pointwise(p) || (p.KH == 3 && p.KW == 3))) {
            if (igemm_pointwise(p) && p.act == MTN_ACT_GELU_GRAD_) return launch_igemm<T, 128, 128, 2, 4, 8, 4>(p, stream);
            if (igemm_pointwise(p)) return launch_igemm<T, 128, 128, 2, 4, 8, 1>(p, stream);
            return launch_igemm<T, 128, 128, 2, 4, 8, 2>(p, stream);
        }
        // (debug key 3 = 7): ping-pong halves; debug key 6 = 4 picks the 256x128 tile for it
        if (cfg == 0 && g_igemm_stages == 7)
            return g_igemm_tile == 4 ? launch_igemm_dma2<T, 256, 128, 4, 2, 7>(p, stream) : launch_igemm_dma2<T, 128, 128, 4, 2, 7>(p, stream);
    }
#endif
    // the window mode on the 8-wave 128x128 tile (round 5): the stage-2 3x3 layers -- 128 channels at 24x24: 45 KB of window + two
    // 16 KB weight tiles = 78 KB, two workgroups per CU as before.  Round 2 measured the window at 128x128 with FOUR waves (+4 %),
    // the 8-wave tile came later (+5..9 %); together: forward 65.2 -> 64.0 us, step -0.035 ms (profiles/r05_window128_ab.txt)
    if constexpr (sizeof(T) == 2 || std::is_same<T, f32x3_t>::value) {
        // (a 256x128 window tile with SIXTEEN waves, one workgroup per CU, measured slower: forward 64 -> 76 us, profiles/r05_window256_ab.txt)
        // (split storage: two of the four 32-channel blocks resident, two passes over the window; debug key 8 = 10: bf16 only)
        if (cfg == 0 && igemm_window128_ok(p) && (sizeof(T) == 2 || g_igemm_dbg != 10)) return launch_igemm<T, 128, 128, 4, 2, 5, 2>(p, stream);
    }
    if (cfg == 0 && (g_igemm_stages == 5 || g_igemm_stages >= 7)) return launch_igemm_dma2<T, 128, 128, 4, 2>(p, stream);
#ifdef MTN_EXPERIMENTS
    if (cfg == 0 && g_igemm_stages == 3) return launch_igemm<T, 128, 128, 4, 2, 3>(p, stream);
    if (cfg == 0) return launch_igemm_mode<T, 128, 128, 2, 2>(p, stream);
    if (cfg == 2) return launch_igemm_mode<T, 256, 64, 4, 1>(p, stream);
#endif
    if (cfg == 1) return launch_igemm_mode<T, 64, 128, 1, 4>(p, stream);
    // 64 -> 64 channel 3x3 stride-1 layers (stage 1) on a 256x64 window tile with EIGHT waves (8 x 1: 32 x 64 per wave), two workgroups
    // per CU (round 5): halo 1.38x instead of 1.77x of the tile rows, the weight tiles streamed once per 256 rows, prologue and
    // epilogue once per 256 rows.  Round 1 measured 256x64 with four waves and without the window (-5..8 %), round 2 with the window
    // (-30 %); with eight waves: forward 78.3 -> 72.4 us, data gradient 87.5 -> 75.7, step -0.045 ms (profiles/r05_window256_ab.txt)
    if constexpr (sizeof(T) == 2 || std::is_same<T, f32x3_t>::value) {      // (split storage: one resident 32-channel slab, two passes)
        if (igemm_window256_ok(p)) return launch_igemm<T, 256, 64, 8, 1, 5, 2>(p, stream);
    }
    return launch_igemm_mode<T, 128, 64, 4, 1>(p, stream);
}
}  // namespace

#ifdef MTN_EXPERIMENTS
void mtn_igemm_set_stages(int n) { g_igemm_stages = ((n >= 1 && n <= 5) || n == 7 || n == 8) ? n : 5; }
int mtn_igemm_has_experiments(void) { return 1; }
#else
void mtn_igemm_set_stages(int n) { (void)n; g_igemm_stages = 5; }
int mtn_igemm_has_experiments(void) { return 0; }
#endif
void mtn_igemm_set_window_limit(int kb) { g_window_limit_kb = (kb > 0 && kb <= 150) ? kb : 53; }

void mtn_igemm_set_image_tiles(int v) { g_image_tiles = v & 3; }

void mtn_igemm_set_tile_n64(int rows) { g_tile_n64 = rows == 256 ? 256 : 128; }

void mtn_igemm_set_tile(int t) { g_igemm_tile = t; }
int mtn_igemm_read_stamps(void* dst, size_t bytes) {
#if defined(MTN_STAMPS) && !defined(MTN_EMU)
    if (bytes > sizeof(unsigned long long) * 256 * 16) bytes = sizeof(unsigned long long) * 256 * 16;
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
#else
    (void)dst; (void)bytes;
    return 1;
#endif
}
void mtn_igemm_set_dbg(int v) { g_igemm_dbg = v; }
void mtn_igemm_set_x3_waves(int v) { (void)v; }     // (debug key 19 -- wave layouts of the in-register split -- retired with it in round 4)

int mtn_igemm_tiles_m(int M, int N, int K) {
    static const int bm[4] = {128, 64, 256, 128};
    return mtn_cdiv(M, bm[igemm_config(M, N, K)]);
}

int mtn_igemm_stats_tiles(const IgemmArgs& p, int dtype) {
    if (mtn_dgrad2_ok(p, dtype)) return mtn_dgrad2_tiles_m(p);
    if (mtn_pw_ok(p, dtype)) return mtn_pw_rows(p);
    if (mtn_win3_ok(p, dtype)) return mtn_win3_rows(p);
    if (igemm_image_tile_ok(p, dtype)) return mtn_cdiv(p.M, IMG_BM);
    if ((dtype == MTN_BF16 || dtype == MTN_F32X3) && igemm_window256_ok(p)) return mtn_cdiv(p.M, 256);
    if (dtype == MTN_BF16 && ((g_igemm_stages == 7 && g_igemm_tile == 4) || (g_igemm_stages == 5 && g_igemm_tile >= 5)) && igemm_config(p.M, p.N, p.K) == 0)
        return mtn_cdiv(p.M, 256);
    return mtn_igemm_tiles_m(p.M, p.N, p.K);
}

// the kernel mtn_igemm_run launches for these arguments (same rules, no launch): MTN_KERNEL_* of include/mtn.h
int mtn_igemm_family(const IgemmArgs& p, int dtype) {
    if (mtn_dgrad2_ok(p, dtype)) return MTN_KERNEL_DGRAD2;
    if (mtn_pw_ok(p, dtype)) return MTN_KERNEL_PW;
    if (mtn_win3_ok(p, dtype)) return MTN_KERNEL_WIN3;
    if (igemm_image_tile_ok(p, dtype)) return MTN_KERNEL_IGEMM_IMAGE;
    const int esz = dtype == MTN_BF16 ? 2 : 4;
    if (g_igemm_stages == 5 && !p.pm && igemm_window_shape(p)) {
        static const int bm[4] = {128, 64, 256, 128}, bn[4] = {128, 128, 64, 64};
        if ((dtype == MTN_BF16 || dtype == MTN_F32X3) && igemm_window256_ok(p)) return MTN_KERNEL_IGEMM_WINDOW;
        const int cfg = igemm_config(p.M, p.N, p.K);      // (the 8-wave 128x128 tile: bf16 layers of 128 channels only, dispatch_igemm)
        if (cfg == 0 ? ((dtype == MTN_BF16 || (dtype == MTN_F32X3 && g_igemm_dbg != 10)) && igemm_window128_ok(p)) : igemm_window_ok(p, bm[cfg], bn[cfg], esz))
            return MTN_KERNEL_IGEMM_WINDOW;
    }
    return MTN_KERNEL_IGEMM;
}

int mtn_igemm_run(const IgemmArgs& p0, int dtype, hipStream_t stream) {
    IgemmArgs p = p0;
    p.dbg = g_igemm_dbg;
    {   // magic = floor(2^32 / d) + 1: __umulhi(x, magic) == x / d for x * d < 2^32
        const unsigned dw = (unsigned)(p.pm ? p.OW / 2 : p.OW), dh = (unsigned)(p.pm ? p.OH / 2 : p.OH);
        const size_t esz = dtype == MTN_BF16 ? 2 : 4;
        p.a_bytes = (unsigned)((((size_t)p.B * p.H * p.W - 1) * p.lda + p.Cin) * esz);     // extent of the gather source
        p.w_bytes = (unsigned)((size_t)p.N * p.K * esz);
        p.magic_w = dw > 1 ? (unsigned)(0x100000000ull / dw) + 1u : 0u;
        p.magic_h = dh > 1 ? (unsigned)(0x100000000ull / dh) + 1u : 0u;
    }
    const int bk = dtype == MTN_BF16 ? 64 : 32;
    MTN_REQUIRE(dtype == MTN_BF16 || dtype == MTN_F32 || dtype == MTN_F32X3, "igemm: bad dtype %d", dtype);
    MTN_REQUIRE(p.act != MTN_ACT_GELU_GRAD_ || (igemm_pointwise(p) && (g_igemm_stages == 5 || g_igemm_stages >= 7) && p.aux_out),
                "igemm: the GELU-backward epilogue exists for 1x1 taps in the default pipeline only");
    MTN_REQUIRE(p.M > 0 && p.N > 0 && p.N % 64 == 0, "igemm: N=%d must be a positive multiple of 64 (M=%d)", p.N, p.M);
    MTN_REQUIRE(p.Cin % bk == 0, "igemm: Cin=%d must be a multiple of %d", p.Cin, bk);
    MTN_REQUIRE(p.K == p.KH * p.KW * p.Cin, "igemm: K=%d != KH*KW*Cin", p.K);
    MTN_REQUIRE(p.div == 1 || p.div == 2, "igemm: div=%d", p.div);
    MTN_REQUIRE(p.KH * p.KW <= 9, "igemm: at most 9 taps");
    MTN_REQUIRE(p.OH < 32768 && p.OW < 32768, "igemm: spatial size too large for the magic row decode");
    MTN_REQUIRE(!p.pm || (p.OH % 2 == 0 && p.OW % 2 == 0 && (!p.stats || p.bn_y) && p.KH == 3 && p.KW == 3),
                "igemm: parity-major rows need a 3x3 tap set and even OH/OW");
    MTN_REQUIRE(!p.bn_y || (p.stats && p.bn_ss && p.bn_mr && p.bn_ldy % (bk / 8) == 0 && (!p.bn_res || p.bn_ldr % (bk / 8) == 0) &&
                            ((uintptr_t)p.bn_y | (uintptr_t)p.bn_res) % 16 == 0 &&
                            (p.bn_act == MTN_ACT_SILU || (p.bn_act == MTN_ACT_NONE && !p.bn_res))),
                "igemm: BatchNorm backward statistics need a slab, scale/shift, mean/rstd and aligned rows");
    MTN_REQUIRE((long)p.M * (p.pm ? p.OW / 2 : p.OW) < (1L << 32), "igemm: M * OW too large for the magic row decode");
    MTN_REQUIRE(p.lda % (bk / 8) == 0 && p.ldc % (bk / 8) == 0, "igemm: lda/ldc must keep 16-byte alignment");
    MTN_REQUIRE(p.M == p.B * p.OH * p.OW, "igemm: M=%d != B*OH*OW", p.M);
    MTN_REQUIRE((long)p.B * p.H * p.W * p.lda < (1L << 30) && (long)p.M * p.ldc < (1L << 31) && (long)p.N * p.K < (1L << 30),
                "igemm: tensor too large for 32-bit byte offsets");
    MTN_REQUIRE(((uintptr_t)p.A | (uintptr_t)p.Wp | (uintptr_t)p.Out | (uintptr_t)p.res | (uintptr_t)p.aux_out) % 16 == 0, "igemm: pointers must be 16-byte aligned");
    // split storage: rows are sequences of whole 32-channel blocks [32 hi][32 lo] -- every operand starts on one and
    // every row pitch is a whole number of them
    MTN_REQUIRE(dtype != MTN_F32X3 ||
                    (((uintptr_t)p.A | (uintptr_t)p.Wp | (uintptr_t)p.Out | (uintptr_t)p.res | (uintptr_t)p.aux_out | (uintptr_t)p.bn_y | (uintptr_t)p.bn_res) % 16 == 0 &&
                     p.lda % 32 == 0 && p.ldc % 32 == 0 && (!p.res || p.ldr % 32 == 0) && (!p.aux_out || p.ld_aux % 32 == 0) &&
                     (!p.bn_y || p.bn_ldy % 32 == 0) && (!p.bn_res || p.bn_ldr % 32 == 0)),
                "igemm: split storage (MTN_DTYPE_F32X3) needs 16-byte aligned operands and row pitches that are multiples of 32");
    if (p.tok_pos) {
        MTN_REQUIRE(igemm_pointwise(p) && p.tok_cls && p.tok_P > 0 && p.M % p.tok_P == 0 && !p.stats && !p.res && !p.bias && !p.aux_out &&
                        p.act == MTN_ACT_NONE && p.N % 128 == 0 && (g_igemm_stages == 5 || g_igemm_stages >= 7),
                    "igemm: the token epilogue serves a plain pointwise product whose rows are whole images of %d patches", p.tok_P);
        MTN_REQUIRE((long)p.M * p.tok_P < (1L << 32), "igemm: M * P too large for the multiply-high image index");
        p.tok_magic = p.tok_P > 1 ? (unsigned)(0x100000000ull / (unsigned)p.tok_P) + 1u : 0u;
    }
    if (mtn_dgrad2_ok(p, dtype)) return mtn_dgrad2_run(p, dtype, stream);     // 3x3 stride-2 data gradient: dgrad2.hip
    if (mtn_pw_ok(p, dtype)) return mtn_pw_run(p, dtype, stream);             // short-K pointwise products: pw.hip
    if (mtn_win3_ok(p, dtype)) return mtn_win3_run(p, dtype, stream);         // 64 -> 64 channel 3x3 stride-1 layers: win3.hip
    if (dtype == MTN_F32X3) return dispatch_igemm<f32x3_t>(p, stream);
    return dtype == MTN_BF16 ? dispatch_igemm<bf16_t>(p, stream) : dispatch_igemm<float>(p, stream);
}
