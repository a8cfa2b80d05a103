// attention.hip -- multi-head self-attention core of the ViT encoder on MFMA, forward + backward.
//
// Reference: Attention.forward (model/transformer.py:62-77), heads = 8, head_dim = 32,
//   dots = q k^T * 32^-0.5 ; attn = softmax(dots) ; out = attn v ; merge heads.
// Inputs are read straight out of the fused qkv projection [B*N][768] = (q | k | v) x (h d)
// and the output is written head-merged [B*N][256], so none of the reference's four
// rearranges (transformer.py:66-67,74) exists as a pass.  One workgroup = one (batch, head);
// N = 145 (192^2) or 577 (384^2) tokens: K and V of the head live in LDS for the whole kernel.
//
// Orientation (cdna_hip_programming.md section 3, "An accumulator tile as the next MFMA's operand"):
// the score tile is computed TRANSPOSED, S^T[key][query] = K Q^T, so a lane owns ONE query
// column: softmax needs 2 shuffles per row, and the probability accumulators are directly the
// B operand of the next product that sums over keys (O^T = V^T P^T forward, dQ^T = K^T dS^T
// backward) -- P never goes through LDS or HBM.  The products that sum over queries (dV, dK)
// use the untransposed S = Q K^T in a second kernel whose LDS holds Q and dO.  Operands that
// must be read down the reduction axis of a row-major LDS tile use ds_read_b64_tr_b16.
// head_dim = 32 is exactly one bf16 MFMA K step.  Backward recomputes P from the saved
// log-sum-exp (flash style); only the last layer's probabilities are written (they are the
// module's attnmap output, transformer.py:96,152).
#include <type_traits>
#include "mtn_types.h"
#include "mtn_kernels.h"

namespace {

constexpr int HD = 32;                       // head_dim (model/multitasknet.py:21)
// LDS row pitch: bf16 rows are padded by 16 bytes (conflict-free fragment reads); the fp32
// validation mode keeps rows unpadded so that K and V of a 577-token head (384x384) still fit.
template <class T> struct Lay { static constexpr int PITCH = sizeof(T) == 2 ? HD + 8 : HD; };
// split storage: a head's row is ONE 32-channel block [32 hi][32 lo] = 128 bytes, kept unpadded (577-token heads fit) with its
// eight 16-byte slots XOR-swizzled by row & 7 on the way into LDS: the row fragments (slots g and g + 4) and the transposing
// column reads of acc_mma are then conflict-free like the padded bf16 rows
template <> struct Lay<f32x3_t> { static constexpr int PITCH = HD; };

// ---- operand over the head_dim axis of one row (A: [row][k], B: [k][col]) ----------------------
template <class T> struct RowOp;
template <> struct RowOp<bf16_t> {
    bf16x8_t f;
    __device__ __forceinline__ void load_lds(const bf16_t* tile, int r, int g) { load(tile + r * Lay<bf16_t>::PITCH, true, g); }
    __device__ __forceinline__ void load(const bf16_t* row, bool valid, int g) {
        chunk16 c = valid ? *reinterpret_cast<const chunk16*>(row + 8 * g) : zero_chunk();
#ifdef MTN_EMU
        std::memcpy(&f, &c, 16);
#else
        f = __builtin_bit_cast(bf16x8_t, c);
#endif
    }
};
template <> struct RowOp<float> {
    float f[8];
    __device__ __forceinline__ void load_lds(const float* tile, int r, int g) { load(tile + r * Lay<float>::PITCH, true, g); }
    __device__ __forceinline__ void load(const float* row, bool valid, int g) {
        MTN_UNROLL for (int ks = 0; ks < 8; ++ks) f[ks] = valid ? row[ks * 4 + g] : 0.f;
    }
};
// split storage (Mma<f32x3_t>, mtn_types.h): lane group g holds the hi and lo halves of k = 8g .. 8g+7 of its row, as stored
template <> struct RowOp<f32x3_t> {
    Frag3 f;
    // a row in global memory: one unswizzled block [32 hi][32 lo]
    __device__ __forceinline__ void load(const f32x3_t* row, bool valid, int g) {
        const chunk16 h = valid ? *reinterpret_cast<const chunk16*>(row + 4 * g) : zero_chunk();
        const chunk16 l = valid ? *reinterpret_cast<const chunk16*>(row + 16 + 4 * g) : zero_chunk();
#ifdef MTN_EMU
        std::memcpy(&f.hi, &h, 16); std::memcpy(&f.lo, &l, 16);
#else
        f.hi = __builtin_bit_cast(bf16x8_t, h); f.lo = __builtin_bit_cast(bf16x8_t, l);
#endif
    }
    // row r of a staged tile: slots swizzled by r & 7 (stage_heads)
    __device__ __forceinline__ void load_lds(const f32x3_t* tile, int r, int g) { f = RowFrag<f32x3_t>::load(tile + r * HD, 0, g, r & 7); }
};
__device__ __forceinline__ f32x4_t mma32(const RowOp<f32x3_t>& a, const RowOp<f32x3_t>& b, f32x4_t c) {
    return Mma<f32x3_t>::mma(a.f, b.f, c);
}
__device__ __forceinline__ f32x4_t mma32(const RowOp<bf16_t>& a, const RowOp<bf16_t>& b, f32x4_t c) {
    return Mma<bf16_t>::mma(a.f, b.f, c);
}
__device__ __forceinline__ f32x4_t mma32(const RowOp<float>& a, const RowOp<float>& b, f32x4_t c) {
    MTN_UNROLL for (int ks = 0; ks < 8; ++ks) c = Mma<float>::mma(a.f[ks], b.f[ks], c);
    return c;
}
template <class T> __device__ __forceinline__ float dot_rowops(const RowOp<T>& a, const RowOp<T>& b);
template <> __device__ __forceinline__ float dot_rowops<float>(const RowOp<float>& a, const RowOp<float>& b) {
    float s = 0.f;
    MTN_UNROLL for (int i = 0; i < 8; ++i) s += a.f[i] * b.f[i];
    return s;
}
template <> __device__ __forceinline__ float dot_rowops<f32x3_t>(const RowOp<f32x3_t>& a, const RowOp<f32x3_t>& b) {
    float s = 0.f;
    MTN_UNROLL for (int i = 0; i < 8; ++i) {
#ifdef MTN_EMU
        s += (emu::bf2f(a.f.hi[i]) + emu::bf2f(a.f.lo[i])) * (emu::bf2f(b.f.hi[i]) + emu::bf2f(b.f.lo[i]));
#else
        s += ((float)a.f.hi[i] + (float)a.f.lo[i]) * ((float)b.f.hi[i] + (float)b.f.lo[i]);
#endif
    }
    return s;
}
template <> __device__ __forceinline__ float dot_rowops<bf16_t>(const RowOp<bf16_t>& a, const RowOp<bf16_t>& b) {
    float s = 0.f;
    MTN_UNROLL for (int i = 0; i < 8; ++i) {
#ifdef MTN_EMU
        s += emu::bf2f(a.f[i]) * emu::bf2f(b.f[i]);
#else
        s += (float)a.f[i] * (float)b.f[i];
#endif
    }
    return s;
}

// out[c][n] += sum over the 32 rows r of col[r][c0 + c] * X[r][n], where X (32 x 16) is given as
// two accumulator tiles t0 (rows 0..15) and t1 (rows 16..31) in MFMA C/D layout (row 4g+j, col
// lane&15) and `col` points at [row0][c0] of a row-major LDS tile.  The k order inside the step
// is the accumulators' own: slot (g, j) <-> row 4g+j (j<4) / 16+4g+(j-4).
__device__ __forceinline__ f32x4_t acc_mma(const bf16_t* rows, int c0, int pitch, f32x4_t t0, f32x4_t t1, f32x4_t out, int lane) {
    const int g = lane >> 4, i = lane & 15;
    const bf16_t* col = rows + c0;
    const bf16_t* p0 = col + (4 * g + (i >> 2)) * pitch + 4 * (i & 3);
    const bf16_t* p1 = p0 + 16 * pitch;
    bf16x8_t a, b;
#ifdef MTN_EMU
    emu::u16x4 u = emu::ds_read_tr16_b64(p0), v = emu::ds_read_tr16_b64(p1);
    MTN_UNROLL for (int j = 0; j < 4; ++j) { a[j] = u[j]; a[4 + j] = v[j]; b[j] = f2bf(t0[j]).v; b[4 + j] = f2bf(t1[j]).v; }
#else
    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) bf16x4_t* lds_ptr;
    const bf16x4_t u = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(p0));
    const bf16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(p1));
    a = __builtin_shufflevector(u, v, 0, 1, 2, 3, 4, 5, 6, 7);
    MTN_UNROLL for (int j = 0; j < 4; ++j) { b[j] = (__bf16)t0[j]; b[4 + j] = (__bf16)t1[j]; }
#endif
    return Mma<bf16_t>::mma(a, b, out);
}
__device__ __forceinline__ f32x4_t acc_mma(const float* rows, int c0, int pitch, f32x4_t t0, f32x4_t t1, f32x4_t out, int lane) {
    const int g = lane >> 4, i = lane & 15;
    const float* col = rows + c0;
    MTN_UNROLL for (int j = 0; j < 4; ++j) out = Mma<float>::mma(col[(4 * g + j) * pitch + i], t0[j], out);
    MTN_UNROLL for (int j = 0; j < 4; ++j) out = Mma<float>::mma(col[(16 + 4 * g + j) * pitch + i], t1[j], out);
    return out;
}

// split storage: the column operand is read as stored -- transposing reads down the hi and the lo halves of the (swizzled)
// rows -- and only the accumulator operand, a value computed in registers, is split here
__device__ __forceinline__ f32x4_t acc_mma(const f32x3_t* rows, int c0, int pitch, f32x4_t t0, f32x4_t t1, f32x4_t out, int lane) {
    (void)pitch;                                                   // rows of HD elements = 128 bytes
    const int g = lane >> 4, i = lane & 15;
    const int r = 4 * g + (i >> 2), ch = c0 + 4 * (i & 3);         // rows r and r + 16 (the same r & 7), channels ch .. ch+3
    const char* base = reinterpret_cast<const char*>(rows) + r * 128 + ((ch & 7) << 1);
    const int sh = (((ch >> 3) ^ (r & 7)) << 4), sl = ((((ch >> 3) + 4) ^ (r & 7)) << 4);
    Frag3 a;
    float b[8];
    MTN_UNROLL for (int j = 0; j < 4; ++j) { b[j] = t0[j]; b[4 + j] = t1[j]; }
#ifdef MTN_EMU
    emu::u16x4 u = emu::ds_read_tr16_b64(reinterpret_cast<const bf16_t*>(base + sh)), v = emu::ds_read_tr16_b64(reinterpret_cast<const bf16_t*>(base + sh + 16 * 128));
    emu::u16x4 ul = emu::ds_read_tr16_b64(reinterpret_cast<const bf16_t*>(base + sl)), vl = emu::ds_read_tr16_b64(reinterpret_cast<const bf16_t*>(base + sl + 16 * 128));
    MTN_UNROLL for (int j = 0; j < 4; ++j) { a.hi[j] = u[j]; a.hi[4 + j] = v[j]; a.lo[j] = ul[j]; a.lo[4 + j] = vl[j]; }
#else
    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) bf16x4_t* lds_ptr;
    const bf16x4_t u = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(base + sh));
    const bf16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(base + sh + 16 * 128));
    const bf16x4_t ul = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(base + sl));
    const bf16x4_t vl = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(base + sl + 16 * 128));
    a.hi = __builtin_shufflevector(u, v, 0, 1, 2, 3, 4, 5, 6, 7);
    a.lo = __builtin_shufflevector(ul, vl, 0, 1, 2, 3, 4, 5, 6, 7);
#endif
    return Mma<f32x3_t>::mma(a, Mma<f32x3_t>::split(b), out);
}

// an accumulator quad -> four consecutive channels c .. c+3 of an output row
template <class T> __device__ __forceinline__ void st4v(T* row, int c, const f32x4_t& q) {
    float v[4];
    MTN_UNROLL for (int e = 0; e < 4; ++e) v[e] = q[e];
    st4<T>(row, c, v);
}

// stage rows [0, NP) x 32 of one head (column offset `coff` of a [B*N][ld] tensor) into LDS, zero padded
// TWO such tiles at once, with all global loads of a batch issued before the first LDS store: staged one 16-byte
// chunk per loop trip the workgroup paid one full memory round trip per trip (five of them for a 145-token head,
// ~7 us of a 33 us kernel that moves its bytes in 12); here a batch keeps 2 x 4 loads per thread in flight.
template <class T>
__device__ __forceinline__ void stage_heads(T* dst0, const T* __restrict__ src0, int ld0, int coff0, T* dst1,
                                            const T* __restrict__ src1, int ld1, int coff1, long row0, int N, int NP) {
    constexpr int VEC = 16 / (int)sizeof(T), CPR = HD / VEC, PITCH = Lay<T>::PITCH, NB = 4;
    const int total = NP * CPR, nt = (int)blockDim.x;
    for (int base = threadIdx.x; base < total; base += NB * nt) {
        chunk16 a[NB], b[NB];
        MTN_UNROLL for (int j = 0; j < NB; ++j) {
            const int i = base + j * nt, r = i / CPR, c = (i - r * CPR) * VEC;
            const bool ok = i < total && r < N;
            a[j] = ok ? *reinterpret_cast<const chunk16*>(src0 + (row0 + r) * ld0 + coff0 + c) : zero_chunk();
            b[j] = ok ? *reinterpret_cast<const chunk16*>(src1 + (row0 + r) * ld1 + coff1 + c) : zero_chunk();
        }
        MTN_UNROLL for (int j = 0; j < NB; ++j) {
            const int i = base + j * nt, r = i / CPR, c = (i - r * CPR) * VEC;
            if (i < total) {
                // (split storage: the 16-byte slot index is swizzled by the row, see Lay<f32x3_t>)
                const int cd = std::is_same<T, f32x3_t>::value ? (((c / VEC) ^ (r & 7)) * VEC) : c;
                *reinterpret_cast<chunk16*>(dst0 + r * PITCH + cd) = a[j];
                *reinterpret_cast<chunk16*>(dst1 + r * PITCH + cd) = b[j];
            }
        }
    }
}

// XCD-aware (batch, head) order: the H heads of one image read interleaved 64-byte slices of the
// same qkv rows (two heads per 128-byte line); give all heads of an image to one XCD group
// (workgroup id % 8) so those lines are fetched into one L2 once.
__device__ __forceinline__ int attn_bh(int H) {
    const int L = blockIdx.x, B = gridDim.x / H;
    if (B % 8 != 0) return L;
    const int b = (L / (8 * H)) * 8 + (L & 7), h = (L >> 3) % H;
    return b * H + h;
}

// ---- forward ------------------------------------------------------------------------------------
// FULL: the padded sequence is exactly MAXT key tiles (145 tokens -> 10, 577 -> 38: the model's two sequence lengths).
// With a runtime tile count every `kt < nkt` guard of the unrolled loops is a real branch and hipcc serialises each tile
// as LDS read -> s_waitcnt 0 -> MFMA -> s_nop -> dependent selects (seen in the ISA); with the count a compile-time
// constant the MFMAs of a query tile issue back to back and only the last two tiles carry a padded-key mask.
template <class T, int MAXT, bool FULL>
__global__ __launch_bounds__(512) void attn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out, float* __restrict__ lse,
                                                       float* __restrict__ attn, int N, int H, float scale) {
    constexpr int PITCH = Lay<T>::PITCH;
    MTN_DYN_SMEM(smem);
    const int NP = (N + 31) & ~31, nkt = NP / 16, nqt = (N + 15) / 16;
    T* sK = reinterpret_cast<T*>(smem);
    T* sV = sK + NP * PITCH;
    const int bh = attn_bh(H), b = bh / H, h = bh - b * H;
    const int ld = 3 * H * HD;
    const long row0 = (long)b * N;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, g = lane >> 4, li = lane & 15;
    const int nw = (int)(blockDim.x >> 6);
    // the query fragment of a tile comes straight from global memory: requested one tile ahead (the first one
    // before the K / V staging), so its latency is not on the tile's critical path
    RowOp<T> qnext;
    {
        const int q = wid * 16 + li;
        qnext.load(qkv + (row0 + (q < N ? q : 0)) * ld + h * HD, wid < nqt && q < N, g);
    }
    stage_heads<T>(sK, qkv, ld, H * HD + h * HD, sV, qkv, ld, 2 * H * HD + h * HD, row0, N, NP);
    __syncthreads();
    for (int qt = wid; qt < nqt; qt += nw) {
        const int q = qt * 16 + li;
        const bool qok = q < N;
        const RowOp<T> qf = qnext;
        {
            const int qn = (qt + nw) * 16 + li;
            qnext.load(qkv + (row0 + (qn < N ? qn : 0)) * ld + h * HD, qt + nw < nqt && qn < N, g);
        }
        // The softmax is the VALU-bound part of this kernel (40 scores per lane and tile against 20 MFMAs), so it is
        // kept to max / fma / exp2 / add per score: the maximum is taken over the RAW scores (scale > 0), scale and
        // log2(e) are one factor inside the exponent's fma, only the tiles that hold padded keys are masked (their
        // -3e38 turns into an exact 0 in exp2, no select), and the 1 / sum normalisation is applied to the 8 output
        // values of the lane's query instead of its scores (unless the probabilities themselves are returned).
        f32x4_t s[MAXT];
        float mx = -3.0e38f;
        MTN_UNROLL for (int kt = 0; kt < MAXT; ++kt) {
            if (FULL || kt < nkt) {
                RowOp<T> kf;
                kf.load_lds(sK, kt * 16 + li, g);
                s[kt] = mma32(kf, qf, zero4());
            }
        }
        MTN_UNROLL for (int kt = 0; kt < MAXT; ++kt) {
            if (FULL || kt < nkt) {
                if (FULL ? (kt >= MAXT - 2) : (kt * 16 + 16 > N)) {      // only these tiles can hold padded keys
                    MTN_UNROLL for (int r = 0; r < 4; ++r)
                        if (kt * 16 + 4 * g + r >= N) s[kt][r] = -3.0e38f;
                }
                MTN_UNROLL for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float c2 = scale * 1.4426950408889634f, mc = mx * c2;
        float sum = 0.f;
        MTN_UNROLL for (int kt = 0; kt < MAXT; ++kt) {
            if (FULL || kt < nkt) {
                MTN_UNROLL for (int r = 0; r < 4; ++r) {
                    const float e = exp2f_(fmaf(s[kt][r], c2, -mc));
                    s[kt][r] = e;
                    sum += e;
                }
            }
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        if (qok && g == 0) lse[(long)bh * N + q] = mx * scale + __logf(sum);
        if (attn && qok) {
            MTN_UNROLL for (int kt = 0; kt < MAXT; ++kt) {
                if (FULL || kt < nkt) {
                    float* ap = attn + ((long)bh * N + q) * N + kt * 16 + 4 * g;
                    if (FULL ? (kt < MAXT - 2) : (kt * 16 + 16 <= N)) {
                        // a whole key tile: ONE 16-byte store per lane (rows of N floats are only 4-byte aligned) --
                        // 64-byte runs per query row instead of four scattered dword stores per lane
                        f32x4u_t v;
                        MTN_UNROLL for (int r = 0; r < 4; ++r) v[r] = s[kt][r] * inv;
                        *reinterpret_cast<f32x4u_t*>(ap) = v;
                        MTN_SCHED_FENCE();             // (one tile's four products at a time: batched, 38 tiles of them spill)
                    } else {
                        MTN_UNROLL for (int r = 0; r < 4; ++r)
                            if (kt * 16 + 4 * g + r < N) ap[r] = s[kt][r] * inv;
                    }
                }
            }
        }
        f32x4_t o0 = zero4(), o1 = zero4();
        MTN_UNROLL for (int pr = 0; pr < MAXT / 2; ++pr) {
            if (FULL || 2 * pr < nkt) {
                o0 = acc_mma(sV + pr * 32 * PITCH, 0, PITCH, s[2 * pr], s[2 * pr + 1], o0, lane);
                o1 = acc_mma(sV + pr * 32 * PITCH, 16, PITCH, s[2 * pr], s[2 * pr + 1], o1, lane);
            }
        }
        MTN_UNROLL for (int r = 0; r < 4; ++r) { o0[r] *= inv; o1[r] *= inv; }
        if (qok) {      // O^T tiles: rows d = dt*16 + 4g + r, column q
            T* op = out + (row0 + q) * (H * HD) + h * HD;
            st4v<T>(op, 4 * g, o0);
            st4v<T>(op, 16 + 4 * g, o1);
        }
    }
}

// ---- backward 1: dQ (sums over keys) and delta = rowsum(dO * O) ----------------------------------
template <class T, int MAXT, bool FULL>
__global__ __launch_bounds__(512) void attn_bwd_dq_kernel(const T* __restrict__ qkv, const T* __restrict__ o,
                                                          const T* __restrict__ dout, const float* __restrict__ lse,
                                                          float* __restrict__ delta, T* __restrict__ dqkv, int N, int H,
                                                          float scale) {
    constexpr int PITCH = Lay<T>::PITCH;
    MTN_DYN_SMEM(smem);
    const int NP = (N + 31) & ~31, nkt = NP / 16, nqt = (N + 15) / 16;
    T* sK = reinterpret_cast<T*>(smem);
    T* sV = sK + NP * PITCH;
    const int bh = attn_bh(H), b = bh / H, h = bh - b * H;
    const int ld = 3 * H * HD, ldo = H * HD;
    const long row0 = (long)b * N;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, g = lane >> 4, li = lane & 15;
    const int nw = (int)(blockDim.x >> 6);
    RowOp<T> qn_, don_, on_;                              // row operands of the NEXT query tile (requested one tile ahead)
    auto fetch = [&](int qt) {
        const int q = qt * 16 + li;
        const bool ok = qt < nqt && q < N;
        const long qr = row0 + (ok ? q : 0);
        qn_.load(qkv + qr * ld + h * HD, ok, g);
        don_.load(dout + qr * ldo + h * HD, ok, g);
        on_.load(o + qr * ldo + h * HD, ok, g);
    };
    fetch(wid);
    stage_heads<T>(sK, qkv, ld, H * HD + h * HD, sV, qkv, ld, 2 * H * HD + h * HD, row0, N, NP);
    __syncthreads();
    for (int qt = wid; qt < nqt; qt += nw) {
        const int q = qt * 16 + li;
        const bool qok = q < N;
        const RowOp<T> qf = qn_, dof = don_, of = on_;
        fetch(qt + nw);
        float dl = dot_rowops<T>(dof, of);
        dl += __shfl_xor(dl, 16, 64);
        dl += __shfl_xor(dl, 32, 64);
        // only the tiles that hold padded keys are masked (below); padded query lanes compute a column that is never stored
        const float c2 = scale * 1.4426950408889634f;
        const float lq2 = (qok ? lse[(long)bh * N + q] : 0.f) * 1.4426950408889634f;
        if (qok && g == 0) delta[(long)bh * N + q] = dl;
        // dQ^T = K^T dS^T accumulates over the keys, so the score tiles need not all be alive at once (unlike the forward,
        // whose softmax needs the whole row): groups of GK key tiles -- 2 GK independent MFMAs in flight, their dS, then the
        // GK / 2 pair products into the two accumulators.  Registers stay at GK tiles whatever the sequence length, which
        // is what lets the 38-tile (577-token) form run with a compile-time tile count too.
        constexpr int GK = 8;
        f32x4_t d0 = zero4(), d1 = zero4();
        MTN_UNROLL for (int k0 = 0; k0 < MAXT; k0 += GK) {
            f32x4_t ds[GK];
            MTN_UNROLL for (int u = 0; u < GK; ++u) {
                const int kt = k0 + u;
                ds[u] = zero4();
                if (kt < MAXT && (FULL || kt < nkt)) {
                    RowOp<T> kf, vf;
                    kf.load_lds(sK, kt * 16 + li, g);
                    vf.load_lds(sV, kt * 16 + li, g);
                    const f32x4_t s = mma32(kf, qf, zero4());
                    const f32x4_t dp = mma32(vf, dof, zero4());
                    MTN_UNROLL for (int r = 0; r < 4; ++r) {
                        // (the softmax scale is applied once to the 8 accumulated values of the lane's query below, not to every
                        //  score: the kernel is VALU-bound -- fma / exp2 / sub / mul per score instead of five)
                        const float p = exp2f_(fmaf(s[r], c2, -lq2));
                        ds[u][r] = p * (dp[r] - dl);
                    }
                    if (FULL ? (kt >= MAXT - 2) : (kt * 16 + 16 > N)) {   // tiles that can hold padded keys: exp(0 - lse) may overflow
                        MTN_UNROLL for (int r = 0; r < 4; ++r)
                            if (kt * 16 + 4 * g + r >= N) ds[u][r] = 0.f;
                    }
                }
            }
            MTN_UNROLL for (int u = 0; u < GK; u += 2) {
                const int pr = (k0 + u) / 2;                       // (the padded tile count is even: pairs are complete)
                if (k0 + u < MAXT && (FULL || 2 * pr < nkt)) {
                    d0 = acc_mma(sK + pr * 32 * PITCH, 0, PITCH, ds[u], ds[u + 1], d0, lane);
                    d1 = acc_mma(sK + pr * 32 * PITCH, 16, PITCH, ds[u], ds[u + 1], d1, lane);
                }
            }
            if (FULL) MTN_SCHED_FENCE();
        }
        MTN_UNROLL for (int r = 0; r < 4; ++r) { d0[r] *= scale; d1[r] *= scale; }
        if (qok) {
            T* dp = dqkv + (row0 + q) * ld + h * HD;
            st4v<T>(dp, 4 * g, d0);
            st4v<T>(dp, 16 + 4 * g, d1);
        }
    }
}

// ---- backward 2: dK, dV (sum over queries); LDS holds Q and dO of the head ------------------------
// lse / delta of the head's queries sit in LDS behind the two operand tiles (read once per (query, key tile) in the loop) in every
// mode: the longest sequence the kernels accept (608 rows) leaves 3.3 KB of the 160 KB for them in the 4-byte modes
template <class T>
__global__ __launch_bounds__(512) void attn_bwd_dkv_kernel(const T* __restrict__ qkv, const T* __restrict__ dout,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           T* __restrict__ dqkv, int N, int H, float scale) {
    constexpr int PITCH = Lay<T>::PITCH;
    MTN_DYN_SMEM(smem);
    const int NP = (N + 31) & ~31, nkt = (N + 15) / 16, nqp = NP / 32;
    T* sQ = reinterpret_cast<T*>(smem);
    T* sD = sQ + NP * PITCH;
    float* sL = reinterpret_cast<float*>(sD + NP * PITCH);      // lse[NP]
    float* sDl = sL + NP;                                       // delta[NP]
    const int bh = attn_bh(H), b = bh / H, h = bh - b * H;
    const int ld = 3 * H * HD, ldo = H * HD;
    const long row0 = (long)b * N;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, g = lane >> 4, li = lane & 15;
    const int nw = (int)(blockDim.x >> 6);
    RowOp<T> kn_, vn_;                                    // K^T / V^T operands of the NEXT key tile
    auto fetch = [&](int kt) {
        const int key = kt * 16 + li;
        const bool ok = kt < nkt && key < N;
        const long kr = row0 + (ok ? key : 0);
        kn_.load(qkv + kr * ld + H * HD + h * HD, ok, g);
        vn_.load(qkv + kr * ld + 2 * H * HD + h * HD, ok, g);
    };
    fetch(wid);
    stage_heads<T>(sQ, qkv, ld, h * HD, sD, dout, ldo, h * HD, row0, N, NP);
    for (int i = threadIdx.x; i < NP; i += blockDim.x) {
        sL[i] = i < N ? lse[(long)bh * N + i] * 1.4426950408889634f : 0.f;          // (in log2 units: one multiply per row, not per score)
        sDl[i] = i < N ? delta[(long)bh * N + i] : 0.f;
    }
    __syncthreads();
    const float c2 = scale * 1.4426950408889634f;
    for (int kt = wid; kt < nkt; kt += nw) {
        const int key = kt * 16 + li;
        const bool kok = key < N;
        const RowOp<T> kf = kn_, vf = vn_;                // B operands K^T, V^T: [k = d][col = key]
        fetch(kt + nw);
        f32x4_t dk0 = zero4(), dk1 = zero4(), dv0 = zero4(), dv1 = zero4();
        for (int qp = 0; qp < nqp; ++qp) {
            f32x4_t p[2], ds[2];
            MTN_UNROLL for (int hf = 0; hf < 2; ++hf) {
                const int qb = qp * 32 + hf * 16;
                RowOp<T> qa, da;                          // A operands: rows = queries
                qa.load_lds(sQ, qb + li, g);
                da.load_lds(sD, qb + li, g);
                const f32x4_t s = mma32(qa, kf, zero4());       // S[q = qb+4g+r][key]
                const f32x4_t dp = mma32(da, vf, zero4());      // dP[q][key]
                // (padded queries need no mask: their Q and dO rows are zero in LDS, so they add nothing to
                //  dV = P^T dO and dK = dS^T Q; padded key lanes compute columns that are never stored)
                MTN_UNROLL for (int r = 0; r < 4; ++r) {
                    const int q = qb + 4 * g + r;
                    const float lq = sL[q], dq = sDl[q];
                    const float pv = exp2f_(fmaf(s[r], c2, -lq));
                    p[hf][r] = pv;
                    ds[hf][r] = pv * (dp[r] - dq);           // (scale: once on dK below)
                }
            }
            dv0 = acc_mma(sD + qp * 32 * PITCH, 0, PITCH, p[0], p[1], dv0, lane);          // dV^T += dO^T P
            dv1 = acc_mma(sD + qp * 32 * PITCH, 16, PITCH, p[0], p[1], dv1, lane);
            dk0 = acc_mma(sQ + qp * 32 * PITCH, 0, PITCH, ds[0], ds[1], dk0, lane);        // dK^T += Q^T dS
            dk1 = acc_mma(sQ + qp * 32 * PITCH, 16, PITCH, ds[0], ds[1], dk1, lane);
        }
        MTN_UNROLL for (int r = 0; r < 4; ++r) { dk0[r] *= scale; dk1[r] *= scale; }
        if (kok) {
            T* dkp = dqkv + (row0 + key) * ld + H * HD + h * HD;
            T* dvp = dqkv + (row0 + key) * ld + 2 * H * HD + h * HD;
            st4v<T>(dkp, 4 * g, dk0); st4v<T>(dkp, 16 + 4 * g, dk1);
            st4v<T>(dvp, 4 * g, dv0); st4v<T>(dvp, 16 + 4 * g, dv1);
        }
    }
}

template <class K>
void set_max_lds(K kern, size_t lds) {
#ifndef MTN_EMU
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
#endif
}

template <class T>
int attn_fwd_t(const AttnArgs& a, hipStream_t stream) {
    const int NP = (a.N + 31) & ~31;
    const size_t lds = 2 * (size_t)NP * Lay<T>::PITCH * sizeof(T);
    const dim3 grid(a.B * a.H), block(NP / 16 <= 10 ? 256 : 512);   // long sequences: 1 WG/CU -> 8 waves
#define MTN_ATTN_FWD(MT, FL)                                                                                                     \
    do {                                                                                                                         \
        set_max_lds(attn_fwd_kernel<T, MT, FL>, lds);                                                                            \
        MTN_LAUNCH((attn_fwd_kernel<T, MT, FL>), grid, block, lds, stream, (const T*)a.qkv, (T*)a.out, a.lse, a.attn, a.N, a.H, \
                   a.scale);                                                                                                     \
    } while (0)
    if (NP / 16 == 10) MTN_ATTN_FWD(10, true);
    else if (NP / 16 < 10) MTN_ATTN_FWD(10, false);
    else if (NP / 16 == 38 && !std::is_same<T, f32x3_t>::value) MTN_ATTN_FWD(38, true);    // (split-bf16 at 38 tiles spills either way)
    else MTN_ATTN_FWD(38, false);
#undef MTN_ATTN_FWD
    return mtn_check_launch("attn_fwd");
}

template <class T>
int attn_bwd_t(const AttnArgs& a, hipStream_t stream) {
    const int NP = (a.N + 31) & ~31;
    const size_t lds = 2 * (size_t)NP * Lay<T>::PITCH * sizeof(T);
    const dim3 grid(a.B * a.H), block(NP / 16 <= 10 ? 256 : 512);
#define MTN_ATTN_DQ(MT, FL)                                                                                                      \
    do {                                                                                                                         \
        set_max_lds(attn_bwd_dq_kernel<T, MT, FL>, lds);                                                                         \
        MTN_LAUNCH((attn_bwd_dq_kernel<T, MT, FL>), grid, block, lds, stream, (const T*)a.qkv, (const T*)a.out,                  \
                   (const T*)a.dout, a.lse, a.delta, (T*)a.dqkv, a.N, a.H, a.scale);                                             \
    } while (0)
    if (NP / 16 == 10) MTN_ATTN_DQ(10, true);
    else if (NP / 16 < 10) MTN_ATTN_DQ(10, false);
    else if (NP / 16 == 38) MTN_ATTN_DQ(38, true);
    else MTN_ATTN_DQ(38, false);
#undef MTN_ATTN_DQ
    int rc = mtn_check_launch("attn_bwd_dq");
    if (rc) return rc;
    // (round 5: the 4-byte modes stage lse / delta too -- the split-storage kernel spent 123 us per layer at 145 tokens against
    //  33 us in bf16, most of it waiting on those eight global loads per query pair: bf16x3 step -0.20 ms, -0.48 ms at 577 tokens)
    const size_t lds2 = lds + 2 * (size_t)NP * sizeof(float);
    MTN_REQUIRE(lds2 <= 160 * 1024, "attention backward: %d rows do not fit the LDS", NP);
    set_max_lds(attn_bwd_dkv_kernel<T>, lds2);
    MTN_LAUNCH((attn_bwd_dkv_kernel<T>), grid, block, lds2, stream, (const T*)a.qkv, (const T*)a.dout, a.lse, a.delta,
               (T*)a.dqkv, a.N, a.H, a.scale);
    return mtn_check_launch("attn_bwd_dkv");
}

// split storage: unpadded 128-byte rows like the exact-f32 kernels, so every sequence those accept fits
bool attn_x3_fits(int N) {
    const int NP = (N + 31) & ~31;
    return 2 * (size_t)NP * Lay<f32x3_t>::PITCH * sizeof(float) <= 160 * 1024;
}

int check_attn(const AttnArgs& a, int dtype) {
    MTN_REQUIRE(dtype == MTN_BF16 || dtype == MTN_F32, "attention: bad dtype");
    MTN_REQUIRE(a.B > 0 && a.H > 0 && a.N > 0 && a.D == HD, "attention: head_dim must be %d (got %d)", HD, a.D);
    // (the softmax takes its row maximum over the RAW scores and folds the scale into the exponent: a negative scale would
    //  pick the wrong extreme, a zero scale would un-mask the padded keys)
    MTN_REQUIRE(a.scale > 0.f && a.scale < 3.0e38f, "attention: scale must be positive and finite (got %g)", (double)a.scale);
    const int NP = (a.N + 31) & ~31;
    MTN_REQUIRE(NP / 16 <= 38, "attention: N=%d exceeds the register-resident limit 608", a.N);
    const size_t lds = dtype == MTN_BF16 ? 2 * (size_t)NP * (HD * 2 + 16) + 2 * (size_t)NP * 4 : 2 * (size_t)NP * HD * 4;
    MTN_REQUIRE(lds <= 160 * 1024, "attention: N=%d does not fit LDS in this dtype (%zu bytes)", a.N, lds);
    return MTN_OK;
}

}  // namespace

int mtn_attn_fwd_run(const AttnArgs& a, int dtype, hipStream_t stream) {
    const bool x3 = dtype == MTN_F32X3 && attn_x3_fits(a.N);
    dtype = mtn_storage_dtype(dtype);
    int rc = check_attn(a, dtype);
    if (rc) return rc;
    if (x3) return attn_fwd_t<f32x3_t>(a, stream);
    return dtype == MTN_BF16 ? attn_fwd_t<bf16_t>(a, stream) : attn_fwd_t<float>(a, stream);
}

int mtn_attn_bwd_run(const AttnArgs& a, int dtype, hipStream_t stream) {
    const bool x3 = dtype == MTN_F32X3 && attn_x3_fits(a.N);
    dtype = mtn_storage_dtype(dtype);
    int rc = check_attn(a, dtype);
    if (rc) return rc;
    if (x3) return attn_bwd_t<f32x3_t>(a, stream);
    return dtype == MTN_BF16 ? attn_bwd_t<bf16_t>(a, stream) : attn_bwd_t<float>(a, stream);
}
