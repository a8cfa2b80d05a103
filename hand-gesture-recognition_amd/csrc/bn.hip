// bn.hip -- train-mode BatchNorm2d + SiLU (+ residual) around the conv kernels, forward and backward.
//
// Reference: Conv.forward = act(bn(conv(x))) (model/gelan.py:46,56), ResBasicBlock.forward =
// SiLU(x + bn(conv(...))) (model/gelan.py:78-87).  nn.BatchNorm2d defaults: eps 1e-5, momentum
// 0.1, batch mean / biased variance for normalisation, unbiased variance into running_var.
//
// Forward:  conv epilogue -> stats[tile][2][C] partial sums           (igemm.hip / conv1.hip)
//           rowblock_sum  -> [R][2][C]   (only when there are many tiles)
//           bn_finalize   -> scale = gamma*rstd, shift = beta - mean*scale, running stats
//           bn_act_fwd    -> out = act(y*scale + shift [+ res])       (HBM-bound, 16-byte lanes)
// Backward: bn_bwd_reduce -> partial sums of dz and dz*xhat, dz = dout * act'(.)
//           (rowblock_sum) + bn_bwd_finalize -> dgamma, dbeta, mean coefficients
//           bn_bwd_apply  -> dy = scale*(dz - mean(dz) - xhat*mean(dz*xhat)), residual gradient
// All reductions are slab + ordered sum: deterministic, no float atomics.
#include <atomic>
#include "mtn_types.h"
#include "mtn_kernels.h"

static int g_bn_fwd_descending = 1;     // debug key 24
void mtn_bn_set_fwd_order(int descending) { g_bn_fwd_descending = descending ? 1 : 0; }
#ifdef MTN_EXPERIMENTS
static int g_bn_ticket = 0;             // debug key 26 (experiments builds): 1 = rowblock_sum + finalize kernel as ONE launch
void mtn_bn_set_ticket(int on) { g_bn_ticket = on ? 1 : 0; }

// Tickets of the fused fold + finalize kernels below: one counter per (launch slot, 64-channel group), zero when no such
// kernel is running (the block that takes the last ticket resets it).  The host hands out slots round-robin, so two launches
// share a slot only if kTicketSlots launches lie between them -- more than any queue or captured plan of this library holds.
constexpr int kTicketSlots = 1024, kTicketGroups = 8;
__device__ unsigned g_fold_tickets[kTicketSlots][kTicketGroups];
#endif

namespace {

constexpr int kLanes = 16;            // row lanes of the 1024-thread reduction blocks (x 64 columns)

// out[rb][c] = sum of in[r][c] over the rb-th block of rows_per_block rows (in: row pitch ld >= W, out: dense)
__global__ __launch_bounds__(1024) void rowblock_sum_kernel(const float* __restrict__ in, int rows, int W, int ld,
                                                            int rows_per_block, float* __restrict__ out) {
    __shared__ float red[kLanes][64];
    const int tl = threadIdx.x >> 6, cl = threadIdx.x & 63;
    const int c = blockIdx.x * 64 + cl;
    const int r0 = blockIdx.y * rows_per_block;
    int r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    float s = 0.f;
    if (c < W)
        for (int r = r0 + tl; r < r1; r += kLanes) s += in[(long)r * ld + c];
    red[tl][cl] = s;
    __syncthreads();
    if (tl == 0 && c < W) {
        float t = 0.f;
        MTN_UNROLL for (int q = 0; q < kLanes; ++q) t += red[q][cl];
        out[(long)blockIdx.y * W + c] = t;
    }
}

// The same for a TABLE of slabs in one launch (grid.z = job): the parameter-gradient slabs of the ViT backward -- LayerNorm
// weight / bias, Linear biases -- are leaves of the dependency graph, so the plan leaves them unreduced and folds all of
// them with two launches of this kernel at the end of the bucket (level 1: 128 rows per block into the slab's scratch tail,
// level 2: the rest into the gradient) instead of two launches per slab.  job = {in, out, rows, W, ld, rows_per_block}.
struct ColJob { const float* in; float* out; int rows, W, ld, rows_per_block; };
__global__ __launch_bounds__(1024) void rowblock_sum_grouped_kernel(const ColJob* __restrict__ jobs) {
    __shared__ float red[kLanes][64];
    const ColJob j = jobs[blockIdx.z];
    const int tl = threadIdx.x >> 6, cl = threadIdx.x & 63;
    const int c = blockIdx.x * 64 + cl;
    const int r0 = blockIdx.y * j.rows_per_block;
    if (blockIdx.x * 64 >= j.W || r0 >= j.rows) return;           // (workgroup-uniform: the grid is the largest job's)
    int r1 = r0 + j.rows_per_block;
    if (r1 > j.rows) r1 = j.rows;
    float s = 0.f;
    if (c < j.W)
        for (int r = r0 + tl; r < r1; r += kLanes) s += j.in[(long)r * j.ld + c];
    red[tl][cl] = s;
    __syncthreads();
    if (tl == 0 && c < j.W) {
        float t = 0.f;
        MTN_UNROLL for (int q = 0; q < kLanes; ++q) t += red[q][cl];
        j.out[(long)blockIdx.y * j.W + c] = t;
    }
}

// block = 16 row-lanes x 64 channels; returns (sum over R rows of part[r][which][c]) in double
__device__ __forceinline__ void sum2_rows(const float* part, int R, int C, int c, bool valid,
                                          double& s0, double& s1) {
    __shared__ double red[2][kLanes][64];
    const int tl = threadIdx.x >> 6, cl = threadIdx.x & 63;
    double a = 0.0, b = 0.0;
    if (valid)
        for (int r = tl; r < R; r += kLanes) {
            a += (double)part[((long)r * 2 + 0) * C + c];
            b += (double)part[((long)r * 2 + 1) * C + c];
        }
    red[0][tl][cl] = a;
    red[1][tl][cl] = b;
    __syncthreads();
    s0 = 0.0; s1 = 0.0;
    MTN_UNROLL for (int q = 0; q < kLanes; ++q) { s0 += red[0][q][cl]; s1 += red[1][q][cl]; }
}

__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ part, int R, int C, float count,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ rmean, float* __restrict__ rvar,
                                                          long long* __restrict__ nbt, float momentum, float eps,
                                                          float* __restrict__ scale_shift, float* __restrict__ mean_rstd) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const bool valid = c < C;
    double s, ss;
    sum2_rows(part, R, C, c, valid, s, ss);
    if ((threadIdx.x >> 6) == 0 && valid) {
        const double mean = s / count;
        double var = ss / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gamma[c] * rstd;
        scale_shift[c] = sc;
        scale_shift[C + c] = beta[c] - (float)mean * sc;
        mean_rstd[c] = (float)mean;
        mean_rstd[C + c] = rstd;
        if (rmean) {
            const double unbiased = count > 1.f ? var * (double)count / ((double)count - 1.0) : var;
            rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
        }
    }
    if (nbt && blockIdx.x == 0 && threadIdx.x == 0) nbt[0] += 1;
}

#ifdef MTN_EXPERIMENTS
// ---- fold + finalize in ONE launch (round 4; MEASURED AND NOT ADOPTED: profiles/r04_tail_ab.txt) -----
// A dependent launch costs ~5 us of the stream's time however little it does (every rowblock_sum / bn_finalize /
// bn_bwd_finalize of the round-3 step ran 4.9-5.9 us for ~1 us of work).  The fused form below turned out to cost the same:
// it is a chain of dependent round trips to the coherent level (store acknowledge, ticket, reload).  Kept for its test
// and for the fence finding recorded at coherent_store.  Block (x, y) folds row block y of the
// 64-channel group x exactly as rowblock_sum does, publishes its row (device-scope fence) and takes a ticket; the block
// that takes the LAST ticket of its group then reads all gridDim.y folded rows and finalizes as bn_finalize /
// bn_bwd_finalize do -- the same sums in the same order as the two-launch path, so the results are bit-identical to it
// and independent of which block came last.  No block ever waits for another: nothing here can hang.
// Memory ordering WITHOUT device-scope fences: __threadfence() compiles to buffer_wbl2 + buffer_inv, a write-back and an
// invalidate of the XCD's WHOLE L2 per block -- measured +0.33 ms per step (the producer's output, which the next
// elementwise pass re-reads, sat dirty in those caches).  Instead the few folded rows themselves travel as agent-scope
// relaxed atomics (sc1 stores / loads: coherent across the XCDs' L2s by themselves), a block waits for its stores to be
// acknowledged (s_waitcnt vmcnt(0): stores count in vmcnt on gfx9) before thread 0 takes the ticket, and the last block's
// reads are control-dependent on the ticket it received.
__device__ __forceinline__ void coherent_store(float* p, float v) {
#ifdef MTN_EMU
    *p = v;
#else
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ float coherent_load(const float* p) {
#ifdef MTN_EMU
    return *p;
#else
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
struct FoldTicket {
    __device__ __forceinline__ static bool last(int slot) {
        __shared__ int s_last;
#ifndef MTN_EMU
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's part of the folded row has reached the coherent level
#endif
        __syncthreads();                                   // ... every thread's, before the ticket is taken
        if (threadIdx.x == 0) {
#ifdef MTN_EMU
            const unsigned t = g_fold_tickets[slot][blockIdx.x]++;
#else
            const unsigned t = __hip_atomic_fetch_add(&g_fold_tickets[slot][blockIdx.x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
            s_last = t == gridDim.y - 1;
            if (s_last) g_fold_tickets[slot][blockIdx.x] = 0u;          // (the next use is a later launch)
        }
        __syncthreads();
        return s_last != 0;
    }
};

// sum2_rows over rows other blocks of THIS launch wrote (coherent loads)
__device__ __forceinline__ void sum2_rows_coherent(const float* part, int R, int C, int c, bool valid, double& s0, double& s1) {
    __shared__ double red[2][kLanes][64];
    const int tl = threadIdx.x >> 6, cl = threadIdx.x & 63;
    // all of a lane's rows (R <= kDirectRows = 16 per lane) are requested before the first is used: these loads miss every
    // cache by design, one round trip instead of R / 16.  Rows past R add an exact 0.0: the sums are those of sum2_rows.
    double a = 0.0, b = 0.0;
    float va[16], vb[16];
    MTN_UNROLL for (int u = 0; u < 16; ++u) {
        const int r = tl + kLanes * u;
        const bool ok = valid && r < R;
        va[u] = ok ? coherent_load(part + ((long)r * 2 + 0) * C + c) : 0.f;
        vb[u] = ok ? coherent_load(part + ((long)r * 2 + 1) * C + c) : 0.f;
    }
    MTN_UNROLL for (int u = 0; u < 16; ++u) { a += (double)va[u]; b += (double)vb[u]; }
    red[0][tl][cl] = a;
    red[1][tl][cl] = b;
    __syncthreads();
    s0 = 0.0; s1 = 0.0;
    MTN_UNROLL for (int q = 0; q < kLanes; ++q) { s0 += red[0][q][cl]; s1 += red[1][q][cl]; }
}

// block = 16 row lanes x 64 channels; folds both statistics columns (c, C + c) of rows [y rpb, (y+1) rpb) into out[y]
__device__ __forceinline__ void fold2_block(const float* in, int rows, int C, int rpb, float* out) {
    __shared__ float red[2][kLanes][64];
    const int tl = threadIdx.x >> 6, cl = threadIdx.x & 63;
    const int c = blockIdx.x * 64 + cl;
    const int r0 = blockIdx.y * rpb;
    int r1 = r0 + rpb;
    if (r1 > rows) r1 = rows;
    float a = 0.f, b = 0.f;
    if (c < C)
        for (int r = r0 + tl; r < r1; r += kLanes) { a += in[((long)r * 2 + 0) * C + c]; b += in[((long)r * 2 + 1) * C + c]; }
    red[0][tl][cl] = a;
    red[1][tl][cl] = b;
    __syncthreads();
    if (tl < 2 && c < C) {
        float t = 0.f;
        MTN_UNROLL for (int q = 0; q < kLanes; ++q) t += red[tl][q][cl];
        coherent_store(out + ((long)blockIdx.y * 2 + tl) * C + c, t);
    }
}

__global__ __launch_bounds__(1024) void bn_fold_finalize_kernel(const float* in, int rows, int C, int rpb, float* out, int slot,
                                                               float count, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ rmean,
                                                               float* __restrict__ rvar, long long* __restrict__ nbt,
                                                               float momentum, float eps, float* __restrict__ scale_shift,
                                                               float* __restrict__ mean_rstd) {
    fold2_block(in, rows, C, rpb, out);
    if (!FoldTicket::last(slot)) return;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const bool valid = c < C;
    double s, ss;
    sum2_rows_coherent(out, (int)gridDim.y, C, c, valid, s, ss);
    if ((threadIdx.x >> 6) == 0 && valid) {
        const double mean = s / count;
        double var = ss / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gamma[c] * rstd;
        scale_shift[c] = sc;
        scale_shift[C + c] = beta[c] - (float)mean * sc;
        mean_rstd[c] = (float)mean;
        mean_rstd[C + c] = rstd;
        if (rmean) {
            const double unbiased = count > 1.f ? var * (double)count / ((double)count - 1.0) : var;
            rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
        }
    }
    if (nbt && blockIdx.x == 0 && threadIdx.x == 0) nbt[0] += 1;
}

__global__ __launch_bounds__(1024) void bn_bwd_fold_finalize_kernel(const float* in, int rows, int C, int rpb, float* out,
                                                                   int slot, float count, float* __restrict__ dgamma,
                                                                   float* __restrict__ dbeta, float* __restrict__ coef) {
    fold2_block(in, rows, C, rpb, out);
    if (!FoldTicket::last(slot)) return;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const bool valid = c < C;
    double s0, s1;
    sum2_rows_coherent(out, (int)gridDim.y, C, c, valid, s0, s1);
    if ((threadIdx.x >> 6) == 0 && valid) {
        dbeta[c] = (float)s0;
        dgamma[c] = (float)s1;
        coef[c] = (float)(s0 / count);
        coef[C + c] = (float)(s1 / count);
    }
}
#endif   // MTN_EXPERIMENTS

// eval mode (model.eval()): normalise with the running statistics
__global__ __launch_bounds__(256) void bn_eval_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ rmean, const float* __restrict__ rvar,
                                                      float eps, float* __restrict__ scale_shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const float sc = gamma[c] * rsqrtf(rvar[c] + eps);
        scale_shift[c] = sc;
        scale_shift[C + c] = beta[c] - rmean[c] * sc;
    }
}

// Elementwise passes: a thread keeps ONE 16-byte channel chunk for all its rows (the grid stride is
// a multiple of the chunks per row), so the per-channel constants are loaded once into registers
// and the loop body is loads -> math -> store with nothing else on its dependency chain.
template <int VEC> struct ChanConst {
    float a[VEC], b[VEC];
    __device__ __forceinline__ void load(const float* __restrict__ p, int C, int c) {
        MTN_UNROLL for (int e = 0; e < VEC; ++e) { a[e] = p[c + e]; b[e] = p[C + c + e]; }
    }
};

template <class T>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const T* __restrict__ y, int ldy, const float* __restrict__ ss,
                                                         const T* __restrict__ res, int ldr, T* __restrict__ out, int ldo,
                                                         int M, int C, int act, int descending) {
    constexpr int VEC = Ev<T>::N;
    const int cpr = C / VEC;
    const int gtid = blockIdx.x * 256 + threadIdx.x;
    const int c = (gtid % cpr) * VEC;
    const int rstep = gridDim.x * 256 / cpr;           // 256 % cpr == 0 (host-checked)
    ChanConst<VEC> k;
    k.load(ss, C, c);
    // rows in DESCENDING order (debug key 24 = 0: ascending): the pass then finishes on the first rows of `out`, which are the rows
    // the consumer -- the next convolution, walking its tiles upwards -- reads first, while they are still in the memory-side cache
    for (int q = gtid / cpr; q < M; q += rstep) {
        const int m = descending ? M - 1 - q : q;
        float v[VEC];
        ld_vec<T>(y + (long)m * ldy, c, v);
        MTN_UNROLL for (int e = 0; e < VEC; ++e) v[e] = v[e] * k.a[e] + k.b[e];
        if (res) {
            float r[VEC];
            ld_vec<T>(res + (long)m * ldr, c, r);
            MTN_UNROLL for (int e = 0; e < VEC; ++e) v[e] += r[e];
        }
        if (act == MTN_ACT_SILU) {
            MTN_UNROLL for (int e = 0; e < VEC; ++e) v[e] = siluf_(v[e]);
        }
        st_vec<T>(out + (long)m * ldo, c, v);
    }
}

// dz for one 16-byte chunk; xh = xhat.  pre-activation = y*scale+shift (+res)
template <class T, int VEC>
__device__ __forceinline__ void bn_dz(const T* __restrict__ dout, int lddo, const T* __restrict__ y, int ldy,
                                      const T* __restrict__ res, int ldr, const ChanConst<VEC>& ss,
                                      const ChanConst<VEC>& mr, int act, int m, int c, float* dz, float* xh) {
    float g[VEC], v[VEC], r[VEC];
    ld_vec<T>(dout + (long)m * lddo, c, g);
    ld_vec<T>(y + (long)m * ldy, c, v);
    const bool with_res = act == MTN_ACT_SILU && res;
    if (with_res) ld_vec<T>(res + (long)m * ldr, c, r);
    bn_dz_math<VEC>(g, v, r, with_res, ss.a, ss.b, mr.a, mr.b, act == MTN_ACT_SILU, dz, xh);
}

// partial[blockIdx.x][2][C]: sum dz, sum dz*xhat over this block's rows
template <class T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dout, int lddo, const T* __restrict__ y,
                                                            int ldy, const T* __restrict__ res, int ldr,
                                                            const float* __restrict__ ss, const float* __restrict__ mr,
                                                            int M, int C, int act, int rows_per_block,
                                                            float* __restrict__ partial) {
    constexpr int VEC = Ev<T>::N;
    MTN_DYN_SMEM(smem);
    float* red = reinterpret_cast<float*>(smem);        // [rl][2][C]
    const int cpr = C / VEC;
    const int rl_n = 256 / cpr;                         // row lanes
    const int rl = threadIdx.x / cpr, c = (threadIdx.x % cpr) * VEC;
    const int r0 = blockIdx.x * rows_per_block;
    int r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    float s0[VEC], s1[VEC];
    MTN_UNROLL for (int e = 0; e < VEC; ++e) { s0[e] = 0.f; s1[e] = 0.f; }
    ChanConst<VEC> kss, kmr;
    if (rl < rl_n) { kss.load(ss, C, c); kmr.load(mr, C, c); }
    if (rl < rl_n)
#pragma unroll 2
        for (int m = r0 + rl; m < r1; m += rl_n) {
            float dz[VEC], xh[VEC];
            bn_dz<T, VEC>(dout, lddo, y, ldy, res, ldr, kss, kmr, act, m, c, dz, xh);
            MTN_UNROLL for (int e = 0; e < VEC; ++e) { s0[e] += dz[e]; s1[e] += dz[e] * xh[e]; }
        }
    if (rl < rl_n) {
        MTN_UNROLL for (int e = 0; e < VEC; ++e) {
            red[(rl * 2 + 0) * C + c + e] = s0[e];
            red[(rl * 2 + 1) * C + c + e] = s1[e];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        float t = 0.f;
        for (int q = 0; q < rl_n; ++q) t += red[q * 2 * C + i];
        partial[(long)blockIdx.x * 2 * C + i] = t;
    }
}

// dgamma = sum dz*xhat, dbeta = sum dz; coef[0][c] = mean(dz), coef[1][c] = mean(dz*xhat)
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ part, int R, int C, float count,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* __restrict__ coef) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const bool valid = c < C;
    double s0, s1;
    sum2_rows(part, R, C, c, valid, s0, s1);
    if ((threadIdx.x >> 6) == 0 && valid) {
        dbeta[c] = (float)s0;
        dgamma[c] = (float)s1;
        coef[c] = (float)(s0 / count);
        coef[C + c] = (float)(s1 / count);
    }
}

template <class T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dout, int lddo, const T* __restrict__ y,
                                                           int ldy, const T* __restrict__ res, int ldr,
                                                           const float* __restrict__ ss, const float* __restrict__ mr,
                                                           const float* __restrict__ coef, T* __restrict__ dy, int lddy,
                                                           T* __restrict__ dres, int lddres, const T* __restrict__ extra,
                                                           int ldextra, int M, int C, int act) {
    constexpr int VEC = Ev<T>::N;
    const int cpr = C / VEC;
    const int gtid = blockIdx.x * 256 + threadIdx.x;
    const int c = (gtid % cpr) * VEC;
    const int rstep = gridDim.x * 256 / cpr;
    ChanConst<VEC> kss, kmr, kco;
    kss.load(ss, C, c); kmr.load(mr, C, c); kco.load(coef, C, c);
    // rows in DESCENDING order: the reduce pass that just ran left the tail of dout / y in the
    // memory-side cache, so the re-read starts where it still hits
    for (int m = M - 1 - gtid / cpr; m >= 0; m -= rstep) {
        float dz[VEC], xh[VEC], o[VEC];
        bn_dz<T, VEC>(dout, lddo, y, ldy, res, ldr, kss, kmr, act, m, c, dz, xh);
        MTN_UNROLL for (int e = 0; e < VEC; ++e) o[e] = kss.a[e] * (dz[e] - kco.a[e] - xh[e] * kco.b[e]);
        st_vec<T>(dy + (long)m * lddy, c, o);
        if (dres) {          // gradient of the residual input (= dz, the pre-activation gradient)
            if (extra) {
                float x[VEC];
                ld_vec<T>(extra + (long)m * ldextra, c, x);
                MTN_UNROLL for (int e = 0; e < VEC; ++e) dz[e] += x[e];
            }
            st_vec<T>(dres + (long)m * lddres, c, dz);
        }
    }
}

inline int ew_grid(long total) {
    long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

// ---- host-side launchers -----------------------------------------------------------------------
static const int kRowsPerBlock = 128;     // rows folded by one rowblock_sum block
static const int kDirectRows = 256;       // slabs up to this many rows go straight to the finalize kernels (<= 16 per row lane)
static const int kBwdBlocks = 2048;       // bn_bwd_reduce grid cap: 256 CUs x 8 resident workgroups, one round

// rows of scratch a slab with `rows` producer rows needs for the folding passes (appended after it)
int mtn_reduce_rows_needed(int rows) {
    int total = rows;
    while (rows > kDirectRows) {
        rows = mtn_cdiv(rows, kRowsPerBlock);
        total += rows;
    }
    return total;
}

// folds part[rows][W] 128 rows at a time into the scratch rows that follow it until a finalize
// kernel takes the rest (16 row lanes x 64 columns per block); returns
// pointer / row count of the slab the finalize kernel reads
static int fold_rows(const float*& part, int& rows, int W, hipStream_t stream, int* ld_io = nullptr) {
    int ld = ld_io ? *ld_io : W;                 // pitch of the producer's rows (>= W); folded rows are dense
    while (rows > kDirectRows) {
        const int R = mtn_cdiv(rows, kRowsPerBlock);
        float* out = const_cast<float*>(part) + (long)rows * ld;
        MTN_LAUNCH(rowblock_sum_kernel, dim3(mtn_cdiv(W, 64), R), dim3(1024), 0, stream, part, rows, W, ld, kRowsPerBlock, out);
        int rc = mtn_check_launch("rowblock_sum");
        if (rc) return rc;
        part = out;
        rows = R;
        ld = W;
    }
    if (ld_io) *ld_io = ld;
    return MTN_OK;
}

#ifdef MTN_EXPERIMENTS
// the fused path applies when ONE fold level remains (every shape of the model: <= 32768 producer rows) and the channel
// groups fit the ticket table
static bool fold_ticket_ok(int rows, int C) {
    return g_bn_ticket && rows > kDirectRows && mtn_cdiv(rows, kRowsPerBlock) <= kDirectRows && mtn_cdiv(C, 64) <= kTicketGroups;
}
static int next_ticket_slot() {
    static std::atomic<unsigned> next{0};
    return (int)(next.fetch_add(1u) % (unsigned)kTicketSlots);
}
#endif

int mtn_bn_finalize_run(float* stats, int tiles, int C, long count, const float* gamma, const float* beta,
                        float* rmean, float* rvar, long long* nbt, float momentum, float eps, float* scale_shift,
                        float* mean_rstd, hipStream_t stream) {
    MTN_REQUIRE(tiles > 0 && C > 0 && count > 0, "bn_finalize: bad sizes");
#ifdef MTN_EXPERIMENTS
    if (fold_ticket_ok(tiles, C)) {
        MTN_LAUNCH(bn_fold_finalize_kernel, dim3(mtn_cdiv(C, 64), mtn_cdiv(tiles, kRowsPerBlock)), dim3(1024), 0, stream, stats, tiles, C,
                   kRowsPerBlock, stats + (long)tiles * 2 * C, next_ticket_slot(), (float)count, gamma, beta, rmean, rvar, nbt,
                   momentum, eps, scale_shift, mean_rstd);
        return mtn_check_launch("bn_fold_finalize");
    }
#endif
    const float* part = stats;
    int rows = tiles;
    int rc = fold_rows(part, rows, 2 * C, stream);
    if (rc) return rc;
    MTN_LAUNCH(bn_finalize_kernel, dim3(mtn_cdiv(C, 64)), dim3(1024), 0, stream, part, rows, C, (float)count, gamma, beta,
               rmean, rvar, nbt, momentum, eps, scale_shift, mean_rstd);
    return mtn_check_launch("bn_finalize");
}

int mtn_bn_eval_run(const float* gamma, const float* beta, const float* rmean, const float* rvar, float eps,
                    float* scale_shift, int C, hipStream_t stream) {
    MTN_REQUIRE(C > 0, "bn_eval: C=%d", C);
    MTN_LAUNCH(bn_eval_kernel, dim3(mtn_cdiv(C, 256)), dim3(256), 0, stream, gamma, beta, rmean, rvar, eps, scale_shift, C);
    return mtn_check_launch("bn_eval");
}

int mtn_bn_act_fwd_run(const void* y, int ldy, const float* ss, const void* res, int ldr, void* out, int ldo, int dtype,
                       int M, int C, int act, hipStream_t stream) {
    const int vec = dtype == MTN_F32 ? 4 : 8;
    MTN_REQUIRE(C % vec == 0 && ldy % vec == 0 && ldo % vec == 0 && (!res || ldr % vec == 0), "bn_act_fwd: alignment");
    MTN_REQUIRE(dtype != MTN_F32X3 || (C % 32 == 0 && ldy % 32 == 0 && ldo % 32 == 0 && (!res || ldr % 32 == 0) &&
                                       ((uintptr_t)y | (uintptr_t)out | (uintptr_t)res) % 16 == 0),
                "bn_act_fwd: split storage needs whole 32-channel blocks (C, row pitches) and 16-byte aligned operands");
    MTN_REQUIRE(act == MTN_ACT_NONE || act == MTN_ACT_SILU, "bn_act_fwd: act=%d", act);
    MTN_REQUIRE(C / vec <= 256 && 256 % (C / vec) == 0, "bn_act_fwd: C=%d unsupported (16-byte chunks per row must divide 256)", C);
    const long total = (long)M * (C / vec);
    if (dtype == MTN_BF16)
        MTN_LAUNCH(bn_act_fwd_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, stream, (const bf16_t*)y, ldy, ss,
                   (const bf16_t*)res, ldr, (bf16_t*)out, ldo, M, C, act, g_bn_fwd_descending);
    else if (dtype == MTN_F32X3)
        MTN_LAUNCH(bn_act_fwd_kernel<f32x3_t>, dim3(ew_grid(total)), dim3(256), 0, stream, (const f32x3_t*)y, ldy, ss,
                   (const f32x3_t*)res, ldr, (f32x3_t*)out, ldo, M, C, act, g_bn_fwd_descending);
    else
        MTN_LAUNCH(bn_act_fwd_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, stream, (const float*)y, ldy, ss,
                   (const float*)res, ldr, (float*)out, ldo, M, C, act, g_bn_fwd_descending);
    return mtn_check_launch("bn_act_fwd");
}

// one round of resident workgroups at most: a second, partly filled round costs a whole round
int mtn_bn_bwd_blocks(int M) {
    const int b = mtn_cdiv(M, 64);
    return b < kBwdBlocks ? b : kBwdBlocks;
}

int mtn_bn_bwd_run(const BnBwdArgs& a, int dtype, hipStream_t stream) {
    const int vec = dtype == MTN_F32 ? 4 : 8;
    const int cpr = a.C / vec;
    MTN_REQUIRE(a.C % vec == 0 && cpr <= 256 && 256 % cpr == 0, "bn_bwd: C=%d unsupported", a.C);
    MTN_REQUIRE(dtype != MTN_F32X3 || (a.C % 32 == 0 && a.lddo % 32 == 0 && a.ldy % 32 == 0 && a.lddy % 32 == 0 && (!a.res || a.ldr % 32 == 0) &&
                                       (!a.dres || a.lddres % 32 == 0) && (!a.extra || a.ldextra % 32 == 0) &&
                                       ((uintptr_t)a.dout | (uintptr_t)a.y | (uintptr_t)a.res | (uintptr_t)a.dy | (uintptr_t)a.dres | (uintptr_t)a.extra) % 16 == 0),
                "bn_bwd: split storage needs whole 32-channel blocks (C, row pitches) and 16-byte aligned operands");
    MTN_REQUIRE(a.act == MTN_ACT_NONE || a.act == MTN_ACT_SILU, "bn_bwd: act=%d", a.act);
    MTN_REQUIRE(a.act == MTN_ACT_SILU || !a.res, "bn_bwd: residual needs SiLU");
    MTN_REQUIRE(a.partial_rows >= 0, "bn_bwd: partial_rows=%d", a.partial_rows);
    const int blocks = a.partial_rows > 0 ? a.partial_rows : mtn_bn_bwd_blocks(a.M);
    const int rows_per_block = mtn_cdiv(a.M, blocks);
    const size_t lds = (size_t)(256 / cpr) * 2 * a.C * sizeof(float);
    if (a.partial_rows > 0) {
        // the data gradient that wrote dout left the sums in partial (igemm.hip / dgrad2.hip epilogue)
    } else if (dtype == MTN_BF16)
        MTN_LAUNCH(bn_bwd_reduce_kernel<bf16_t>, dim3(blocks), dim3(256), lds, stream, (const bf16_t*)a.dout, a.lddo,
                   (const bf16_t*)a.y, a.ldy, (const bf16_t*)a.res, a.ldr, a.scale_shift, a.mean_rstd, a.M, a.C, a.act,
                   rows_per_block, a.partial);
    else if (dtype == MTN_F32X3)
        MTN_LAUNCH(bn_bwd_reduce_kernel<f32x3_t>, dim3(blocks), dim3(256), lds, stream, (const f32x3_t*)a.dout, a.lddo,
                   (const f32x3_t*)a.y, a.ldy, (const f32x3_t*)a.res, a.ldr, a.scale_shift, a.mean_rstd, a.M, a.C, a.act,
                   rows_per_block, a.partial);
    else
        MTN_LAUNCH(bn_bwd_reduce_kernel<float>, dim3(blocks), dim3(256), lds, stream, (const float*)a.dout, a.lddo,
                   (const float*)a.y, a.ldy, (const float*)a.res, a.ldr, a.scale_shift, a.mean_rstd, a.M, a.C, a.act,
                   rows_per_block, a.partial);
    int rc = mtn_check_launch("bn_bwd_reduce");
    if (rc) return rc;
#ifdef MTN_EXPERIMENTS
    const bool ticket = fold_ticket_ok(blocks, a.C);
    if (ticket) {
        MTN_LAUNCH(bn_bwd_fold_finalize_kernel, dim3(mtn_cdiv(a.C, 64), mtn_cdiv(blocks, kRowsPerBlock)), dim3(1024), 0, stream, a.partial,
                   blocks, a.C, kRowsPerBlock, a.partial + (long)blocks * 2 * a.C, next_ticket_slot(), (float)a.M, a.dgamma, a.dbeta,
                   a.coef);
        rc = mtn_check_launch("bn_bwd_fold_finalize");
    }
#else
    const bool ticket = false;
#endif
    if (!ticket) {
        const float* part = a.partial;
        int rows = blocks;
        rc = fold_rows(part, rows, 2 * a.C, stream);
        if (rc) return rc;
        MTN_LAUNCH(bn_bwd_finalize_kernel, dim3(mtn_cdiv(a.C, 64)), dim3(1024), 0, stream, part, rows, a.C, (float)a.M,
                   a.dgamma, a.dbeta, a.coef);
        rc = mtn_check_launch("bn_bwd_finalize");
    }
    if (rc) return rc;
    const long total = (long)a.M * cpr;
    if (dtype == MTN_BF16)
        MTN_LAUNCH(bn_bwd_apply_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, stream, (const bf16_t*)a.dout, a.lddo,
                   (const bf16_t*)a.y, a.ldy, (const bf16_t*)a.res, a.ldr, a.scale_shift, a.mean_rstd, a.coef,
                   (bf16_t*)a.dy, a.lddy, (bf16_t*)a.dres, a.lddres, (const bf16_t*)a.extra, a.ldextra, a.M, a.C, a.act);
    else if (dtype == MTN_F32X3)
        MTN_LAUNCH(bn_bwd_apply_kernel<f32x3_t>, dim3(ew_grid(total)), dim3(256), 0, stream, (const f32x3_t*)a.dout, a.lddo,
                   (const f32x3_t*)a.y, a.ldy, (const f32x3_t*)a.res, a.ldr, a.scale_shift, a.mean_rstd, a.coef,
                   (f32x3_t*)a.dy, a.lddy, (f32x3_t*)a.dres, a.lddres, (const f32x3_t*)a.extra, a.ldextra, a.M, a.C, a.act);
    else
        MTN_LAUNCH(bn_bwd_apply_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, stream, (const float*)a.dout, a.lddo,
                   (const float*)a.y, a.ldy, (const float*)a.res, a.ldr, a.scale_shift, a.mean_rstd, a.coef,
                   (float*)a.dy, a.lddy, (float*)a.dres, a.lddres, (const float*)a.extra, a.ldextra, a.M, a.C, a.act);
    return mtn_check_launch("bn_bwd_apply");
}

// Grouped form of mtn_colsum_ld_run: jobs int64 [njobs][6] on the device = {slab pointer, out pointer, rows, W, ld, 0}.  The
// host side gives the (largest) extents; the table for the second level is derived by the first launch's geometry: level 1
// writes job i's folded rows behind its slab (as fold_rows does), so the caller passes BOTH tables (see ops.colsum_grouped).
int mtn_colsum_grouped_run(const void* jobs_l1, int n1, int max_w1, int max_rowblocks1, const void* jobs_l2, int n2, int max_w2,
                           hipStream_t stream) {
    MTN_REQUIRE(n1 >= 0 && n2 > 0 && jobs_l2, "colsum_grouped: bad tables");
    if (n1 > 0) {
        MTN_REQUIRE(jobs_l1 && max_w1 > 0 && max_rowblocks1 > 0, "colsum_grouped: bad level-1 table");
        MTN_LAUNCH(rowblock_sum_grouped_kernel, dim3(mtn_cdiv(max_w1, 64), max_rowblocks1, n1), dim3(1024), 0, stream,
                   reinterpret_cast<const ColJob*>(jobs_l1));
        int rc = mtn_check_launch("colsum_grouped_l1");
        if (rc) return rc;
    }
    MTN_LAUNCH(rowblock_sum_grouped_kernel, dim3(mtn_cdiv(max_w2, 64), 1, n2), dim3(1024), 0, stream,
               reinterpret_cast<const ColJob*>(jobs_l2));
    return mtn_check_launch("colsum_grouped_l2");
}
int mtn_fold_rows_per_block(void) { return kRowsPerBlock; }
int mtn_fold_direct_rows(void) { return kDirectRows; }

// generic ordered column sum of an f32 slab: out[c] = sum_r in[r][c]  (bias / LayerNorm affine grads)
int mtn_colsum_run(float* slab, int rows, int W, float* out, hipStream_t stream) {
    return mtn_colsum_ld_run(slab, rows, W, W, out, stream);
}

// the same over the first W columns of rows with pitch ld (the scratch tail follows the ld-pitched rows)
int mtn_colsum_ld_run(float* slab, int rows, int W, int ld, float* out, hipStream_t stream) {
    MTN_REQUIRE(rows > 0 && W > 0 && ld >= W, "colsum: bad sizes (%d x %d, pitch %d)", rows, W, ld);
    const float* part = slab;
    int rc = fold_rows(part, rows, W, stream, &ld);
    if (rc) return rc;
    MTN_LAUNCH(rowblock_sum_kernel, dim3(mtn_cdiv(W, 64), 1), dim3(1024), 0, stream, part, rows, W, ld, rows, out);
    return mtn_check_launch("colsum");
}
