// vit.hip -- the row-wise / elementwise kernels of the ViT encoder and its heads (dim = 256):
//   token assembly        (model/transformer.py:132-139)  rearrange + pos-emb add + cls concat
//   LayerNorm fwd / bwd   (model/transformer.py:33,54,114; eps 1e-5, biased variance)
//   GELU backward         (model/transformer.py:35, exact erf form)
//   classification head   (model/transformer.py:113-116,142-144)  LN + Linear(256 -> classes)
// All are HBM-bound: one wave per 256-wide row, 8/16-byte lanes, fp32 math, ordered slab sums
// for the parameter gradients.
#include "mtn_types.h"
#include "mtn_kernels.h"

namespace {

constexpr int DIM = 256;
constexpr int EPL = DIM / 64;        // elements per lane in a wave-per-row kernel

// a lane's four channels lane*EPL .. +3 of a row: ld4 / st4 (mtn_types.h; split storage: 8 bytes of the hi half and 8 of the lo half)
// ---- token assembly ------------------------------------------------------------------------------
// tokens[b][0] = cls ; tokens[b][1+p] = feat[b*P + p] + pos[p]
template <class T>
__global__ __launch_bounds__(256) void tokens_fwd_kernel(const T* __restrict__ feat, const float* __restrict__ pos,
                                                         const float* __restrict__ cls, T* __restrict__ tok, int B, int P) {
    const int lane = threadIdx.x & 63;
    const long rows = (long)B * (P + 1);
    for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (long)gridDim.x * 4) {
        const int b = (int)(r / (P + 1)), t = (int)(r - (long)b * (P + 1));
        float v[EPL];
        if (t == 0) {
            MTN_UNROLL for (int e = 0; e < EPL; ++e) v[e] = cls[lane * EPL + e];
        } else {
            ld4<T>(feat + ((long)b * P + t - 1) * DIM, lane * EPL, v);
            MTN_UNROLL for (int e = 0; e < EPL; ++e) v[e] += pos[(t - 1) * DIM + lane * EPL + e];
        }
        st4<T>(tok + r * DIM, lane * EPL, v);
    }
}
// dfeat[b*P+p] = dtok[b][1+p]
template <class T>
__global__ __launch_bounds__(256) void tokens_bwd_kernel(const T* __restrict__ dtok, T* __restrict__ dfeat, int B, int P) {
    const int lane = threadIdx.x & 63;
    const long rows = (long)B * P;
    for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (long)gridDim.x * 4) {
        const int b = (int)(r / P), p = (int)(r - (long)b * P);
        float v[EPL];
        ld4<T>(dtok + ((long)b * (P + 1) + 1 + p) * DIM, lane * EPL, v);
        st4<T>(dfeat + r * DIM, lane * EPL, v);
    }
}
// dcls[c] = sum_b dtok[b][0][c]   (one 1024-thread block: 16 batch lanes x 64 channel quads; the rows of a lane are read
// eight at a time -- with one row per trip the block paid a memory round trip per sample: 19 us for 128 KB)
template <class T>
__global__ __launch_bounds__(1024) void cls_grad_kernel(const T* __restrict__ dtok, float* __restrict__ dcls, int B, int P) {
    __shared__ float red[16][DIM];
    const int cq = threadIdx.x & 63, bl = threadIdx.x >> 6;
    const long stride = (long)(P + 1) * DIM;
    float s[EPL] = {0.f, 0.f, 0.f, 0.f};
    for (int b0 = bl; b0 < B; b0 += 16 * 8) {
        float v[8][EPL];
        MTN_UNROLL for (int u = 0; u < 8; ++u) {
            const int b = b0 + 16 * u;
            if (b < B) ld4<T>(dtok + b * stride, cq * EPL, v[u]);
            else { MTN_UNROLL for (int e = 0; e < EPL; ++e) v[u][e] = 0.f; }
        }
        MTN_UNROLL for (int u = 0; u < 8; ++u)
            MTN_UNROLL for (int e = 0; e < EPL; ++e) s[e] += v[u][e];
    }
    MTN_UNROLL for (int e = 0; e < EPL; ++e) red[bl][cq * EPL + e] = s[e];
    __syncthreads();
    if (threadIdx.x < DIM) {
        float t = 0.f;
        MTN_UNROLL for (int q = 0; q < 16; ++q) t += red[q][threadIdx.x];
        dcls[threadIdx.x] = t;
    }
}

// ---- LayerNorm -----------------------------------------------------------------------------------
// A wave owns LN_R CONSECUTIVE rows and requests all of them before it touches the first (round 4): with one row per loop
// trip a wave paid a full memory round trip per row (loads at the top of the trip, two dependent wave reductions, store)
// and the forward needed four rounds of resident waves -- 11.6 / 19.3 us for passes whose bytes move in 6 / 13.
constexpr int LN_R = 4;
template <class T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y, int ldy,
                                                     float* __restrict__ mean_rstd, int M, float eps) {
    const int lane = threadIdx.x & 63;
    float gm[EPL], bt[EPL];
    MTN_UNROLL for (int e = 0; e < EPL; ++e) { gm[e] = gamma[lane * EPL + e]; bt[e] = beta[lane * EPL + e]; }
    const long r0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * LN_R;
    float v[LN_R][EPL];
    MTN_UNROLL for (int u = 0; u < LN_R; ++u)
        if (r0 + u < M) ld4<T>(x + (r0 + u) * ldx, lane * EPL, v[u]);
    MTN_UNROLL for (int u = 0; u < LN_R; ++u) {
        const long r = r0 + u;
        if (r >= M) break;                                        // (wave-uniform)
        const float mean = wave_sum(v[u][0] + v[u][1] + v[u][2] + v[u][3]) * (1.0f / DIM);
        float d = 0.f;
        MTN_UNROLL for (int e = 0; e < EPL; ++e) { v[u][e] -= mean; d += v[u][e] * v[u][e]; }
        const float rstd = rsqrtf(wave_sum(d) * (1.0f / DIM) + eps);
        MTN_UNROLL for (int e = 0; e < EPL; ++e) v[u][e] = v[u][e] * rstd * gm[e] + bt[e];
        st4<T>(y + r * ldy, lane * EPL, v[u]);
        if (mean_rstd && lane == 0) { mean_rstd[2 * r] = mean; mean_rstd[2 * r + 1] = rstd; }
    }
}

// dx = rstd*(g - mean(g) - xhat*mean(g*xhat)) [+ dres], g = dy*gamma;
// partial[block][0][c] = sum dy*xhat (dgamma), partial[block][1][c] = sum dy (dbeta); block = 4 waves x LN_R rows
template <class T>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ x, int ldx,
                                                     const float* __restrict__ mean_rstd, const float* __restrict__ gamma,
                                                     const T* __restrict__ dres, int lddres, T* __restrict__ dx, int lddx,
                                                     float* __restrict__ partial, int M) {
    __shared__ float red[4][2][DIM];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float gm[EPL], sg[EPL], sb[EPL];
    MTN_UNROLL for (int e = 0; e < EPL; ++e) { gm[e] = gamma[lane * EPL + e]; sg[e] = 0.f; sb[e] = 0.f; }
    const long r0 = ((long)blockIdx.x * 4 + w) * LN_R;
    float g[LN_R][EPL], xv[LN_R][EPL], rr[LN_R][EPL], mr[LN_R][2];
    MTN_UNROLL for (int u = 0; u < LN_R; ++u) {
        const long r = r0 + u;
        if (r < M) {
            ld4<T>(dy + r * lddy, lane * EPL, g[u]);
            ld4<T>(x + r * ldx, lane * EPL, xv[u]);
            mr[u][0] = mean_rstd[2 * r]; mr[u][1] = mean_rstd[2 * r + 1];
            if (dres) ld4<T>(dres + r * lddres, lane * EPL, rr[u]);
        }
    }
    MTN_UNROLL for (int u = 0; u < LN_R; ++u) {
        const long r = r0 + u;
        if (r >= M) break;                                        // (wave-uniform)
        const float mean = mr[u][0], rstd = mr[u][1];
        float a = 0.f, b = 0.f;
        MTN_UNROLL for (int e = 0; e < EPL; ++e) {
            xv[u][e] = (xv[u][e] - mean) * rstd;
            sg[e] += g[u][e] * xv[u][e];
            sb[e] += g[u][e];
            g[u][e] *= gm[e];
            a += g[u][e];
            b += g[u][e] * xv[u][e];
        }
        a = wave_sum(a) * (1.0f / DIM);
        b = wave_sum(b) * (1.0f / DIM);
        MTN_UNROLL for (int e = 0; e < EPL; ++e) g[u][e] = rstd * (g[u][e] - a - xv[u][e] * b);
        if (dres) { MTN_UNROLL for (int e = 0; e < EPL; ++e) g[u][e] += rr[u][e]; }
        st4<T>(dx + r * lddx, lane * EPL, g[u]);
    }
    MTN_UNROLL for (int e = 0; e < EPL; ++e) { red[w][0][lane * EPL + e] = sg[e]; red[w][1][lane * EPL + e] = sb[e]; }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * DIM; i += 256) {
        const int which = i / DIM, c = i - which * DIM;
        partial[(long)blockIdx.x * 2 * DIM + i] = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
    }
}

// ---- GELU backward: da = dh * gelu'(a) --------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const T* __restrict__ dh, const T* __restrict__ a, T* __restrict__ da,
                                                       long nchunks) {
    constexpr int VEC = Ev<T>::N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += (long)gridDim.x * blockDim.x) {
        const long e0 = i * VEC, blk = e0 & ~31L;             // (whole 32-channel blocks: valid for split storage too)
        const int c = (int)(e0 & 31);
        float g[VEC], v[VEC];
        ld_vec<T>(dh + blk, c, g);
        ld_vec<T>(a + blk, c, v);
        MTN_UNROLL for (int e = 0; e < VEC; ++e) g[e] *= gelu_grad_(v[e]);
        st_vec<T>(da + blk, c, g);
    }
}

// column sums of a T tensor: partial[block][c] = sum over the block's rows of x[r][c]  (bias grads)
template <class T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, int ldx, float* __restrict__ partial,
                                                             int M, int rows_per_block) {
    __shared__ float red[4][DIM];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float s[EPL];
    MTN_UNROLL for (int e = 0; e < EPL; ++e) s[e] = 0.f;
    const long r0 = (long)blockIdx.x * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    for (long r = r0 + w; r < r1; r += 4) {
        float v[EPL];
        ld4<T>(x + r * ldx, lane * EPL, v);
        MTN_UNROLL for (int e = 0; e < EPL; ++e) s[e] += v[e];
    }
    MTN_UNROLL for (int e = 0; e < EPL; ++e) red[w][lane * EPL + e] = s[e];
    __syncthreads();
    const int c = threadIdx.x;
    partial[(long)blockIdx.x * DIM + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
}

// ---- classification head -------------------------------------------------------------------------
// one wave per sample: xn = LN(tok[b][0]); logits = xn W^T + bias.  xn and (mean, rstd) are saved.
template <class T>
__global__ __launch_bounds__(64) void cls_head_fwd_kernel(const T* __restrict__ tok, long tok_stride, const float* __restrict__ lg,
                                                          const float* __restrict__ lb, const float* __restrict__ W,
                                                          const float* __restrict__ bias, float* __restrict__ logits,
                                                          float* __restrict__ xn_out, float* __restrict__ mean_rstd, int C,
                                                          float eps) {
    const int lane = threadIdx.x, b = blockIdx.x;
    float v[EPL];
    ld4<T>(tok + (long)b * tok_stride, lane * EPL, v);
    const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.0f / DIM);
    float d = 0.f;
    MTN_UNROLL for (int e = 0; e < EPL; ++e) { v[e] -= mean; d += v[e] * v[e]; }
    const float rstd = rsqrtf(wave_sum(d) * (1.0f / DIM) + eps);
    float xh[EPL];
    MTN_UNROLL for (int e = 0; e < EPL; ++e) {
        xh[e] = v[e] * rstd;
        v[e] = xh[e] * lg[lane * EPL + e] + lb[lane * EPL + e];
        xn_out[(long)b * DIM + lane * EPL + e] = xh[e];          // xhat (pre-affine) for the backward
    }
    if (lane == 0) { mean_rstd[2 * b] = mean; mean_rstd[2 * b + 1] = rstd; }
    // eight classes per trip: their wave reductions are independent chains (one class per trip serialised 19 of them)
    for (int c0 = 0; c0 < C; c0 += 8) {
        float s[8];
        MTN_UNROLL for (int j = 0; j < 8; ++j) {
            const int c = c0 + j < C ? c0 + j : C - 1;
            s[j] = 0.f;
            MTN_UNROLL for (int e = 0; e < EPL; ++e) s[j] += v[e] * W[c * DIM + lane * EPL + e];
        }
        MTN_UNROLL for (int j = 0; j < 8; ++j) s[j] = wave_sum(s[j]);
        MTN_UNROLL for (int j = 0; j < 8; ++j)
            if (lane == 0 && c0 + j < C) logits[(long)b * C + c0 + j] = s[j] + bias[c0 + j];
    }
}

// per sample: dxn = dlogits W; LN backward -> dtok[b][0] (written, not accumulated);
// partial[b][0][k] = dxn*xhat, partial[b][1][k] = dxn  (LayerNorm affine gradients, summed over b later)
template <class T>
__global__ __launch_bounds__(64) void cls_head_bwd_x_kernel(const float* __restrict__ dlogits, const float* __restrict__ W,
                                                            const float* __restrict__ lg, const float* __restrict__ xhat,
                                                            const float* __restrict__ mean_rstd, T* __restrict__ dtok,
                                                            long tok_stride, float* __restrict__ partial, int C) {
    const int lane = threadIdx.x, b = blockIdx.x;
    float g[EPL], xh[EPL];
    MTN_UNROLL for (int e = 0; e < EPL; ++e) { g[e] = 0.f; xh[e] = xhat[(long)b * DIM + lane * EPL + e]; }
    for (int c = 0; c < C; ++c) {
        const float dl = dlogits[(long)b * C + c];
        MTN_UNROLL for (int e = 0; e < EPL; ++e) g[e] += dl * W[c * DIM + lane * EPL + e];
    }
    float a = 0.f, bb = 0.f;
    MTN_UNROLL for (int e = 0; e < EPL; ++e) {
        partial[((long)b * 2 + 0) * DIM + lane * EPL + e] = g[e] * xh[e];
        partial[((long)b * 2 + 1) * DIM + lane * EPL + e] = g[e];
        g[e] *= lg[lane * EPL + e]; a += g[e]; bb += g[e] * xh[e];
    }
    a = wave_sum(a) * (1.0f / DIM);
    bb = wave_sum(bb) * (1.0f / DIM);
    const float rstd = mean_rstd[2 * b + 1];
    MTN_UNROLL for (int e = 0; e < EPL; ++e) g[e] = rstd * (g[e] - a - xh[e] * bb);
    st4<T>(dtok + (long)b * tok_stride, lane * EPL, g);
}

// block c (1024 threads = 4 batch lanes x 256 features): dW[c][k] = sum_b dl[b][c]*xn[b][k], dbias[c] = sum_b dl[b][c]
__global__ __launch_bounds__(1024) void cls_head_bwd_w_kernel(const float* __restrict__ dlogits, const float* __restrict__ lg,
                                                              const float* __restrict__ lb, const float* __restrict__ xhat,
                                                              float* __restrict__ dW, float* __restrict__ dbias, int B, int C) {
    __shared__ float red[2][4][DIM];
    const int k = threadIdx.x & 255, bq = threadIdx.x >> 8, c = blockIdx.x;
    const float gk = lg[k], bk = lb[k];
    float s = 0.f, sb = 0.f;
    for (int b0 = bq; b0 < B; b0 += 4 * 8) {                    // eight samples' loads in flight per trip
        float dl[8], xh[8];
        MTN_UNROLL for (int u = 0; u < 8; ++u) {
            const int b = b0 + 4 * u;
            dl[u] = b < B ? dlogits[(long)b * C + c] : 0.f;
            xh[u] = b < B ? xhat[(long)b * DIM + k] : 0.f;
        }
        MTN_UNROLL for (int u = 0; u < 8; ++u) {
            s += dl[u] * (xh[u] * gk + bk);
            sb += dl[u];
        }
    }
    red[0][bq][k] = s;
    red[1][bq][k] = sb;
    __syncthreads();
    if (bq == 0) {
        dW[c * DIM + k] = red[0][0][k] + red[0][1][k] + red[0][2][k] + red[0][3][k];
        if (k == 0) dbias[c] = red[1][0][0] + red[1][1][0] + red[1][2][0] + red[1][3][0];
    }
}

// ---- split storage <-> plain f32 (module boundary: backbone features handed to / gradients taken from the caller; the pose
// head, whose kernels keep plain f32 operands) ------------------------------------------------------------------------------
template <bool SPLIT>
__global__ __launch_bounds__(256) void x3_convert_kernel(const void* __restrict__ src, int lds_, void* __restrict__ dst, int ldd,
                                                         long M, int C) {
    const int gpr = C / 8;                                      // 8-channel groups per row
    const long total = M * gpr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / gpr;
        const int c = (int)(i - m * gpr) * 8;
        float v[8];
        if (SPLIT) {
            const float* s = reinterpret_cast<const float*>(src) + m * lds_ + c;
            unpack<float>(*reinterpret_cast<const chunk16*>(s), v);
            unpack<float>(*reinterpret_cast<const chunk16*>(s + 4), v + 4);
            st_vec<f32x3_t>(reinterpret_cast<f32x3_t*>(dst) + m * ldd, c, v);
        } else {
            ld_vec<f32x3_t>(reinterpret_cast<const f32x3_t*>(src) + m * lds_, c, v);
            float* d = reinterpret_cast<float*>(dst) + m * ldd + c;
            *reinterpret_cast<chunk16*>(d) = pack<float>(v);
            *reinterpret_cast<chunk16*>(d + 4) = pack<float>(v + 4);
        }
    }
}

inline int row_grid(long rows) {
    long b = (rows + 3) / 4;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

#define DISPATCH_T(dtype, KERN, ...)                                                   \
    do {                                                                               \
        if ((dtype) == MTN_BF16) { MTN_LAUNCH(KERN<bf16_t>, __VA_ARGS__); }            \
        else { MTN_LAUNCH(KERN<float>, __VA_ARGS__); }                                 \
    } while (0)

int mtn_tokens_fwd_run(const void* feat, const float* pos, const float* cls, void* tok, int dtype, int B, int P,
                       hipStream_t stream) {
    MTN_REQUIRE(B > 0 && P > 0, "tokens_fwd: bad sizes");
    if (dtype == MTN_BF16)
        MTN_LAUNCH(tokens_fwd_kernel<bf16_t>, dim3(row_grid((long)B * (P + 1))), dim3(256), 0, stream, (const bf16_t*)feat, pos, cls, (bf16_t*)tok, B, P);
    else if (dtype == MTN_F32X3)
        MTN_LAUNCH(tokens_fwd_kernel<f32x3_t>, dim3(row_grid((long)B * (P + 1))), dim3(256), 0, stream, (const f32x3_t*)feat, pos, cls, (f32x3_t*)tok, B, P);
    else
        MTN_LAUNCH(tokens_fwd_kernel<float>, dim3(row_grid((long)B * (P + 1))), dim3(256), 0, stream, (const float*)feat, pos, cls, (float*)tok, B, P);
    return mtn_check_launch("tokens_fwd");
}

int mtn_tokens_bwd_run(const void* dtok, void* dfeat, float* dcls, int dtype, int B, int P, hipStream_t stream) {
    MTN_REQUIRE(B > 0 && P > 0, "tokens_bwd: bad sizes");
    if (dtype == MTN_BF16) {
        MTN_LAUNCH(tokens_bwd_kernel<bf16_t>, dim3(row_grid((long)B * P)), dim3(256), 0, stream, (const bf16_t*)dtok, (bf16_t*)dfeat, B, P);
        MTN_LAUNCH(cls_grad_kernel<bf16_t>, dim3(1), dim3(1024), 0, stream, (const bf16_t*)dtok, dcls, B, P);
    } else if (dtype == MTN_F32X3) {
        MTN_LAUNCH(tokens_bwd_kernel<f32x3_t>, dim3(row_grid((long)B * P)), dim3(256), 0, stream, (const f32x3_t*)dtok, (f32x3_t*)dfeat, B, P);
        MTN_LAUNCH(cls_grad_kernel<f32x3_t>, dim3(1), dim3(1024), 0, stream, (const f32x3_t*)dtok, dcls, B, P);
    } else {
        MTN_LAUNCH(tokens_bwd_kernel<float>, dim3(row_grid((long)B * P)), dim3(256), 0, stream, (const float*)dtok, (float*)dfeat, B, P);
        MTN_LAUNCH(cls_grad_kernel<float>, dim3(1), dim3(1024), 0, stream, (const float*)dtok, dcls, B, P);
    }
    return mtn_check_launch("tokens_bwd");
}

int mtn_ln_fwd_run(const void* x, int ldx, const float* gamma, const float* beta, void* y, int ldy, float* mean_rstd,
                   int dtype, int M, float eps, hipStream_t stream) {
    MTN_REQUIRE(M > 0 && ldx % 4 == 0 && ldy % 4 == 0, "ln_fwd: bad sizes");
    if (dtype == MTN_BF16)
        MTN_LAUNCH(ln_fwd_kernel<bf16_t>, dim3(mtn_cdiv(M, 4 * LN_R)), dim3(256), 0, stream, (const bf16_t*)x, ldx, gamma, beta, (bf16_t*)y, ldy, mean_rstd, M, eps);
    else if (dtype == MTN_F32X3)
        MTN_LAUNCH(ln_fwd_kernel<f32x3_t>, dim3(mtn_cdiv(M, 4 * LN_R)), dim3(256), 0, stream, (const f32x3_t*)x, ldx, gamma, beta, (f32x3_t*)y, ldy, mean_rstd, M, eps);
    else
        MTN_LAUNCH(ln_fwd_kernel<float>, dim3(mtn_cdiv(M, 4 * LN_R)), dim3(256), 0, stream, (const float*)x, ldx, gamma, beta, (float*)y, ldy, mean_rstd, M, eps);
    return mtn_check_launch("ln_fwd");
}

static const int kLnRows = 4 * LN_R;     // rows per workgroup (LN_R consecutive rows per wave): 2320 workgroups at B*N = 37120
int mtn_rowslab_blocks(int M) { return mtn_cdiv(M, kLnRows); }

int mtn_ln_bwd_run(const void* dy, int lddy, const void* x, int ldx, const float* mean_rstd, const float* gamma,
                   const void* dres, int lddres, void* dx, int lddx, float* partial, float* dgamma_dbeta, int dtype,
                   int M, hipStream_t stream) {
    MTN_REQUIRE(M > 0, "ln_bwd: bad sizes");
    const int blocks = mtn_rowslab_blocks(M);
    if (dtype == MTN_BF16)
        MTN_LAUNCH(ln_bwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)dy, lddy, (const bf16_t*)x, ldx, mean_rstd,
                   gamma, (const bf16_t*)dres, lddres, (bf16_t*)dx, lddx, partial, M);
    else if (dtype == MTN_F32X3)
        MTN_LAUNCH(ln_bwd_kernel<f32x3_t>, dim3(blocks), dim3(256), 0, stream, (const f32x3_t*)dy, lddy, (const f32x3_t*)x, ldx, mean_rstd,
                   gamma, (const f32x3_t*)dres, lddres, (f32x3_t*)dx, lddx, partial, M);
    else
        MTN_LAUNCH(ln_bwd_kernel<float>, dim3(blocks), dim3(256), 0, stream, (const float*)dy, lddy, (const float*)x, ldx, mean_rstd,
                   gamma, (const float*)dres, lddres, (float*)dx, lddx, partial, M);
    int rc = mtn_check_launch("ln_bwd");
    if (rc) return rc;
    if (!dgamma_dbeta) return MTN_OK;                     // the caller folds the slab later (mtn_colsum_grouped)
    return mtn_colsum_run(partial, blocks, 2 * DIM, dgamma_dbeta, stream);   // [dgamma | dbeta]
}

int mtn_gelu_bwd_run(const void* dh, const void* a, void* da, int dtype, long n, hipStream_t stream) {
    const int vec = dtype == MTN_F32 ? 4 : (dtype == MTN_F32X3 ? 32 : 8);
    MTN_REQUIRE(n > 0 && n % vec == 0, "gelu_bwd: n=%ld must be a multiple of %d", n, vec);
    const long chunks = n / (dtype == MTN_F32 ? 4 : 8);
    long blocks = (chunks + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (dtype == MTN_BF16)
        MTN_LAUNCH(gelu_bwd_kernel<bf16_t>, dim3((int)blocks), dim3(256), 0, stream, (const bf16_t*)dh, (const bf16_t*)a, (bf16_t*)da, chunks);
    else if (dtype == MTN_F32X3)
        MTN_LAUNCH(gelu_bwd_kernel<f32x3_t>, dim3((int)blocks), dim3(256), 0, stream, (const f32x3_t*)dh, (const f32x3_t*)a, (f32x3_t*)da, chunks);
    else
        MTN_LAUNCH(gelu_bwd_kernel<float>, dim3((int)blocks), dim3(256), 0, stream, (const float*)dh, (const float*)a, (float*)da, chunks);
    return mtn_check_launch("gelu_bwd");
}

int mtn_bias_grad_run(const void* dy, int lddy, float* partial, float* dbias, int dtype, int M, hipStream_t stream) {
    MTN_REQUIRE(M > 0, "bias_grad: bad sizes");
    const int blocks = mtn_rowslab_blocks(M);
    if (dtype == MTN_BF16)
        MTN_LAUNCH(colsum_partial_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)dy, lddy, partial, M, kLnRows);
    else if (dtype == MTN_F32X3)
        MTN_LAUNCH(colsum_partial_kernel<f32x3_t>, dim3(blocks), dim3(256), 0, stream, (const f32x3_t*)dy, lddy, partial, M, kLnRows);
    else
        MTN_LAUNCH(colsum_partial_kernel<float>, dim3(blocks), dim3(256), 0, stream, (const float*)dy, lddy, partial, M, kLnRows);
    int rc = mtn_check_launch("bias_grad");
    if (rc) return rc;
    if (!dbias) return MTN_OK;                            // the caller folds the slab later (mtn_colsum_grouped)
    return mtn_colsum_run(partial, blocks, DIM, dbias, stream);
}

int mtn_cls_head_fwd_run(const void* tok, long tok_stride, const float* lg, const float* lb, const float* W,
                         const float* bias, float* logits, float* xhat, float* mean_rstd, int dtype, int B, int C,
                         float eps, hipStream_t stream) {
    MTN_REQUIRE(B > 0 && C > 0, "cls_head_fwd: bad sizes");
    if (dtype == MTN_BF16)
        MTN_LAUNCH(cls_head_fwd_kernel<bf16_t>, dim3(B), dim3(64), 0, stream, (const bf16_t*)tok, tok_stride, lg, lb, W, bias, logits, xhat, mean_rstd, C, eps);
    else if (dtype == MTN_F32X3)
        MTN_LAUNCH(cls_head_fwd_kernel<f32x3_t>, dim3(B), dim3(64), 0, stream, (const f32x3_t*)tok, tok_stride, lg, lb, W, bias, logits, xhat, mean_rstd, C, eps);
    else
        MTN_LAUNCH(cls_head_fwd_kernel<float>, dim3(B), dim3(64), 0, stream, (const float*)tok, tok_stride, lg, lb, W, bias, logits, xhat, mean_rstd, C, eps);
    return mtn_check_launch("cls_head_fwd");
}

int mtn_cls_head_bwd_run(const float* dlogits, const float* W, const float* lg, const float* lb, const float* xhat,
                         const float* mean_rstd, void* dtok, long tok_stride, float* dW, float* dbias,
                         float* dlg_dlb, float* partial, int dtype, int B, int C, hipStream_t stream) {
    MTN_REQUIRE(B > 0 && C > 0, "cls_head_bwd: bad sizes");
    if (dtype == MTN_BF16)
        MTN_LAUNCH(cls_head_bwd_x_kernel<bf16_t>, dim3(B), dim3(64), 0, stream, dlogits, W, lg, xhat, mean_rstd, (bf16_t*)dtok, tok_stride, partial, C);
    else if (dtype == MTN_F32X3)
        MTN_LAUNCH(cls_head_bwd_x_kernel<f32x3_t>, dim3(B), dim3(64), 0, stream, dlogits, W, lg, xhat, mean_rstd, (f32x3_t*)dtok, tok_stride, partial, C);
    else
        MTN_LAUNCH(cls_head_bwd_x_kernel<float>, dim3(B), dim3(64), 0, stream, dlogits, W, lg, xhat, mean_rstd, (float*)dtok, tok_stride, partial, C);
    MTN_LAUNCH(cls_head_bwd_w_kernel, dim3(C), dim3(1024), 0, stream, dlogits, lg, lb, xhat, dW, dbias, B, C);
    int rc = mtn_check_launch("cls_head_bwd");
    if (rc) return rc;
    return mtn_colsum_run(partial, B, 2 * DIM, dlg_dlb, stream);
}

int mtn_x3_convert_run(const void* src, int ld_src, void* dst, int ld_dst, int to_split, long M, int C, hipStream_t stream) {
    MTN_REQUIRE(M > 0 && C > 0 && C % 32 == 0 && ld_src % 32 == 0 && ld_dst % 32 == 0 && ld_src >= C && ld_dst >= C,
                "x3_convert: rows of whole 32-channel blocks only (C=%d, pitches %d / %d)", C, ld_src, ld_dst);
    MTN_REQUIRE(((uintptr_t)src | (uintptr_t)dst) % 16 == 0, "x3_convert: operands must be 16-byte aligned");
    const long total = M * (C / 8);
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (to_split) MTN_LAUNCH(x3_convert_kernel<true>, dim3((int)blocks), dim3(256), 0, stream, src, ld_src, dst, ld_dst, M, C);
    else MTN_LAUNCH(x3_convert_kernel<false>, dim3((int)blocks), dim3(256), 0, stream, src, ld_src, dst, ld_dst, M, C);
    return mtn_check_launch("x3_convert");
}
