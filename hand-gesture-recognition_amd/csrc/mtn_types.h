// mtn_types.h -- element types, 16-byte vector access and MFMA wrappers shared by all kernels.
//
// Two storage precisions run through the SAME kernels (template parameter T):
//   bf16_t : throughput mode -- bf16 activations/weights, MFMA bf16, fp32 accumulate
//   float  : validation mode -- fp32 storage, exact-f32 MFMA (v_mfma_f32_16x16x4_f32),
//            the mode the 1e-3 parity gate against the fp32 reference is run in.
//   f32x3_t: SPLIT storage, 4 bytes per element like `float`: every 32-channel block of a row (128 bytes) holds
//            [32 bf16 hi][32 bf16 lo] with hi = bf16_rne(x), lo = bf16_rne(x - hi), channels in natural order.
//            The PRODUCER of a tensor splits each element once (st_vec / st4 below); the GEMM-shaped kernels
//            (igemm / dgrad2 / wgrad / attention) read their MFMA fragments as plain bf16 -- two 16-byte rows
//            reads or ds_read_b64_tr_b16 column reads, no conversion -- and multiply with three bf16 MFMAs per
//            product (hi*hi + hi*lo + lo*hi, fp32 accumulate): ~2^-17 relative error per product at 3/16 of
//            the exact-f32 MFMA time; elementwise consumers rebuild x = hi + lo (one add).  Until round 4
//            the mode kept plain fp32 in memory and split every fragment in registers, per wave, tile and
//            tap (24 VALU per fragment: DESIGN.md section 7e).
#pragma once
#include "mtn_rt.h"

struct bf16_t { unsigned short v; };
struct f32x3_t { float v; };
// plain f32 in memory and in LDS, but MFMA operands are split into bf16 hi + lo IN REGISTERS when a fragment is loaded and
// multiplied with three bf16 MFMAs (the split-bf16 product without split storage): the small kernels that stage their own
// operands in LDS (pose head) under MTN_DTYPE_F32X3
struct f32s_t { float v; };

#ifdef MTN_EMU
typedef emu::bf16x8 bf16x8_t;
typedef emu::f32x4 f32x4_t;
typedef emu::f32x16 f32x16_t;
struct f32x4u_t { float v[4]; float& operator[](int i) { return v[i]; } };      // 16-byte store at 4-byte alignment
__device__ inline float bf2f(bf16_t h) { return emu::bf2f(h.v); }
__device__ inline bf16_t f2bf(float f) {      // round-to-nearest-even, NaN stays NaN
    uint32_t u; std::memcpy(&u, &f, 4);
    bf16_t r;
    if ((u & 0x7fffffffu) > 0x7f800000u) { r.v = (unsigned short)((u >> 16) | 0x40); return r; }
    u += 0x7fffu + ((u >> 16) & 1u);
    r.v = (unsigned short)(u >> 16);
    return r;
}
#else
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4u_t __attribute__((ext_vector_type(4), aligned(4)));              // 16-byte store at 4-byte alignment
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float((unsigned)h.v << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {   // v_cvt_pk_bf16_f32: RNE, keeps NaN
    bf16_t r; r.v = __builtin_bit_cast(unsigned short, (__bf16)f); return r;
}
#endif

template <class T> __device__ __forceinline__ float to_f(T x);
template <> __device__ __forceinline__ float to_f<float>(float x) { return x; }
template <> __device__ __forceinline__ float to_f<bf16_t>(bf16_t x) { return bf2f(x); }
template <> __device__ __forceinline__ float to_f<f32s_t>(f32s_t x) { return x.v; }
template <class T> __device__ __forceinline__ T from_f(float x);
template <> __device__ __forceinline__ float from_f<float>(float x) { return x; }
template <> __device__ __forceinline__ f32s_t from_f<f32s_t>(float x) { f32s_t r; r.v = x; return r; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float x) { return f2bf(x); }
// (f32x3_t -- split storage -- has no per-element to_f / from_f: its accessors are ld_vec / st_vec / ld4 / st4 / ld1 below)
// value as it will read back from storage of type T
template <class T> __device__ __forceinline__ float round_to(float x) { return to_f<T>(from_f<T>(x)); }
template <> __device__ __forceinline__ float round_to<f32x3_t>(float x);

// 16-byte chunk: the unit of every global<->LDS move (8 bf16 or 4 f32)
#ifdef MTN_EMU
struct alignas(16) chunk16 {
    unsigned w[4];
    unsigned& operator[](int i) { return w[i]; }
    const unsigned& operator[](int i) const { return w[i]; }
};
#else
typedef unsigned chunk16 __attribute__((ext_vector_type(4)));     // one VGPR quad, never an array in scratch
#endif
template <class T> struct Vec { static constexpr int N = 16 / sizeof(T); };

template <class T> __device__ __forceinline__ void unpack(const chunk16& c, float* f);
template <> __device__ __forceinline__ void unpack<float>(const chunk16& c, float* f) {
#ifdef MTN_EMU
    std::memcpy(f, c.w, 16);
#else
    MTN_UNROLL for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(c[i]);
#endif
}
template <> __device__ __forceinline__ void unpack<bf16_t>(const chunk16& c, float* f) {
    MTN_UNROLL for (int i = 0; i < 4; ++i) {
        bf16_t lo, hi; lo.v = (unsigned short)(c[i] & 0xffffu); hi.v = (unsigned short)(c[i] >> 16);
        f[2 * i] = bf2f(lo); f[2 * i + 1] = bf2f(hi);
    }
}
template <> __device__ __forceinline__ void unpack<f32s_t>(const chunk16& c, float* f) { unpack<float>(c, f); }
template <class T> __device__ __forceinline__ chunk16 pack(const float* f);
template <> __device__ __forceinline__ chunk16 pack<float>(const float* f) {
    chunk16 c;
#ifdef MTN_EMU
    std::memcpy(c.w, f, 16);
#else
    MTN_UNROLL for (int i = 0; i < 4; ++i) c[i] = __float_as_uint(f[i]);
#endif
    return c;
}
template <> __device__ __forceinline__ chunk16 pack<bf16_t>(const float* f) {
    chunk16 c;
    MTN_UNROLL for (int i = 0; i < 4; ++i)
        c[i] = (unsigned)f2bf(f[2 * i]).v | ((unsigned)f2bf(f[2 * i + 1]).v << 16);
    return c;
}
template <> __device__ __forceinline__ chunk16 pack<f32s_t>(const float* f) { return pack<float>(f); }
// four consecutive elements from one accumulator quad (dst 8- / 16-byte aligned)
template <class T> __device__ __forceinline__ void store4(T* dst, const f32x4_t& v);
template <> __device__ __forceinline__ void store4<float>(float* dst, const f32x4_t& v) {
#ifdef MTN_EMU
    MTN_UNROLL for (int r = 0; r < 4; ++r) dst[r] = v[r];
#else
    *reinterpret_cast<f32x4_t*>(dst) = v;
#endif
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* dst, const f32x4_t& v) {
#ifdef MTN_EMU
    MTN_UNROLL for (int r = 0; r < 4; ++r) dst[r] = f2bf(v[r]);
#else
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    u32x2_t o;
    o[0] = (unsigned)f2bf(v[0]).v | ((unsigned)f2bf(v[1]).v << 16);
    o[1] = (unsigned)f2bf(v[2]).v | ((unsigned)f2bf(v[3]).v << 16);
    *reinterpret_cast<u32x2_t*>(dst) = o;
#endif
}
template <> __device__ __forceinline__ void store4<f32s_t>(f32s_t* dst, const f32x4_t& v) { store4<float>(reinterpret_cast<float*>(dst), v); }
__device__ __forceinline__ chunk16 zero_chunk() { chunk16 c; c[0] = 0u; c[1] = 0u; c[2] = 0u; c[3] = 0u; return c; }

// ---- MFMA, 16x16 output tile -------------------------------------------------------------
// Mma<T>::KI = K depth of one instruction; Frag = one lane's A or B operand.
// Lane map (cdna_hip_programming.md section 3): bf16: A[l&15][8*(l>>4)+j], B[8*(l>>4)+j][l&15];
// f32: A[l&15][l>>4], B[l>>4][l&15];  D: col = l&15, row = 4*(l>>4)+reg.
// scheduler fence: no instruction is moved across it (hand-placed software pipelines)
#ifdef MTN_EMU
#define MTN_SCHED_FENCE() ((void)0)
#else
#define MTN_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

template <class T> struct Mma;
template <> struct Mma<bf16_t> {
    static constexpr int KI = 32;
    typedef bf16x8_t Frag;
    // fragment = 16 contiguous bytes at row-major [row][k0 + 8*(lane>>4)]
    static __device__ __forceinline__ Frag load(const bf16_t* p) {
#ifdef MTN_EMU
        Frag f; std::memcpy(&f, p, 16); return f;
#else
        return *reinterpret_cast<const Frag*>(p);
#endif
    }
    static __device__ __forceinline__ f32x4_t mma(const Frag& a, const Frag& b, f32x4_t c) {
#ifdef MTN_EMU
        return emu::mfma_16x16x32_bf16(a, b, c);
#else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
    }
};
template <> struct Mma<float> {
    static constexpr int KI = 4;
    typedef float Frag;
    static __device__ __forceinline__ Frag load(const float* p) { return *p; }
    static __device__ __forceinline__ f32x4_t mma(const Frag& a, const Frag& b, f32x4_t c) {
#ifdef MTN_EMU
        return emu::mfma_16x16x4_f32(a, b, c);
#else
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#endif
    }
};

// Split storage, split-bf16 multiply.  One "instruction" = three 16x16x32 bf16 MFMAs over 32 k: lane group
// g = lane >> 4 holds the hi and the lo halves of EIGHT k-values of its row,
//   hi = bf16_rne(x),  lo = bf16_rne(x - hi)        (x - hi is exact in fp32)
// and the product is hi*hi + hi*lo + lo*hi (lo*lo ~ 2^-18 |ab| is dropped).  WHICH eight k-values a lane group
// holds is free as long as both operands use the same assignment (an MFMA sums over k): the row-major LDS
// tiles of igemm / dgrad2 / attention give group g the channels 8g .. 8g+7 of the 32-channel block (RowFrag
// below), the pixel-major tiles of wgrad the pixels 8g .. 8g+7 of the step (ds_read_b64_tr_b16 down the hi and
// the lo columns).  Both halves come out of memory as stored; `split` serves values that are COMPUTED in
// registers (the attention probabilities).
struct Frag3 { bf16x8_t hi, lo; };
template <> struct Mma<f32x3_t> {
    static constexpr int KI = 32;
    typedef Frag3 Frag;
    static __device__ __forceinline__ Frag split(const float* x) {       // 8 values
        Frag f;
#ifdef MTN_EMU
        MTN_UNROLL for (int j = 0; j < 8; ++j) {
            const bf16_t h = f2bf(x[j]);
            f.hi[j] = h.v;
            f.lo[j] = f2bf(x[j] - bf2f(h)).v;
        }
#else
        MTN_UNROLL for (int j = 0; j < 8; ++j) {
            const __bf16 h = (__bf16)x[j];                               // v_cvt_pk_bf16_f32 (RNE)
            f.hi[j] = h;
            f.lo[j] = (__bf16)(x[j] - (float)h);
        }
#endif
        return f;
    }
    static __device__ __forceinline__ Frag split2(const chunk16& c0, const chunk16& c1) {
        float x[8];
        unpack<float>(c0, x);
        unpack<float>(c1, x + 4);
        return split(x);
    }
    // lane group g of an UNSWIZZLED split row: p = row + 4 g (the 16-byte hi slot of k = 8g .. 8g+7), lo 64 bytes further
    static __device__ __forceinline__ Frag load(const f32x3_t* p) {
        Frag f;
#ifdef MTN_EMU
        std::memcpy(&f.hi, p, 16); std::memcpy(&f.lo, p + 16, 16);
#else
        f.hi = *reinterpret_cast<const bf16x8_t*>(p);
        f.lo = *reinterpret_cast<const bf16x8_t*>(p + 16);
#endif
        return f;
    }
    static __device__ __forceinline__ f32x4_t mma(const Frag& a, const Frag& b, f32x4_t c) {
#ifdef MTN_EMU
        c = emu::mfma_16x16x32_bf16(a.lo, b.hi, c);
        c = emu::mfma_16x16x32_bf16(a.hi, b.lo, c);
        return emu::mfma_16x16x32_bf16(a.hi, b.hi, c);
#else
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.lo, b.hi, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.lo, c, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.hi, c, 0, 0, 0);
#endif
    }
};

template <> struct Mma<f32s_t> {
    static constexpr int KI = 32;
    typedef Frag3 Frag;
    // lane group g: 8 consecutive f32 k-values at p (p = row + k0 + 8 g), split in registers
    static __device__ __forceinline__ Frag load(const f32s_t* p) {
        const chunk16* q = reinterpret_cast<const chunk16*>(p);
        return Mma<f32x3_t>::split2(q[0], q[1]);
    }
    static __device__ __forceinline__ f32x4_t mma(const Frag& a, const Frag& b, f32x4_t c) { return Mma<f32x3_t>::mma(a, b, c); }
};

__device__ __forceinline__ bf16x8_t load_bf16x8(const void* p) {
#ifdef MTN_EMU
    bf16x8_t f; std::memcpy(&f, p, 16); return f;
#else
    return *reinterpret_cast<const bf16x8_t*>(p);
#endif
}

// ---- channel access to stored tensors --------------------------------------------------------------
// Ev<T>::N = channels one thread moves per access of an elementwise pass / epilogue: one 16-byte chunk for bf16 (8) and
// float (4); for split storage the 16-byte hi chunk and the 16-byte lo chunk (64 bytes further) of 8 channels.
// `row` is a pointer to channel 0 of a row -- or to any channel that is a multiple of 32 (slices of the concat buffers
// start at multiples of 64) -- and c a multiple of Ev<T>::N; pointer arithmetic in units of T stays valid for whole
// 32-channel blocks because a block is 32 x 4 bytes either way.
template <class T> struct Ev { static constexpr int N = 16 / (int)sizeof(T); };
template <> struct Ev<f32x3_t> { static constexpr int N = 8; };
// staging type of an accumulator tile in LDS (split storage stages plain f32 and splits on the way out)
template <class T> struct StageT { typedef T type; };
template <> struct StageT<f32x3_t> { typedef float type; };

// bf16 index of channel c's hi half inside a split row (lo: + 32)
__device__ __forceinline__ int x3_pos(int c) { return ((c & ~31) << 1) + (c & 31); }

__device__ __forceinline__ void x3_split8(const float* v, chunk16& h, chunk16& l) {
    MTN_UNROLL for (int i = 0; i < 4; ++i) {
        const bf16_t h0 = f2bf(v[2 * i]), h1 = f2bf(v[2 * i + 1]);
        const bf16_t l0 = f2bf(v[2 * i] - bf2f(h0)), l1 = f2bf(v[2 * i + 1] - bf2f(h1));
        h[i] = (unsigned)h0.v | ((unsigned)h1.v << 16);
        l[i] = (unsigned)l0.v | ((unsigned)l1.v << 16);
    }
}
__device__ __forceinline__ float bits_f(unsigned u) {
#ifdef MTN_EMU
    float f; std::memcpy(&f, &u, 4); return f;
#else
    return __uint_as_float(u);
#endif
}
__device__ __forceinline__ void x3_join8(const chunk16& h, const chunk16& l, float* v) {
    MTN_UNROLL for (int i = 0; i < 4; ++i) {
        v[2 * i] = bits_f(h[i] << 16) + bits_f(l[i] << 16);
        v[2 * i + 1] = bits_f(h[i] & 0xffff0000u) + bits_f(l[i] & 0xffff0000u);
    }
}

// raw bits of Ev<T>::N channels (a load that is issued early and unpacked later keeps this, not floats)
template <class T> struct RawVec { chunk16 c; };
template <> struct RawVec<f32x3_t> { chunk16 h, l; };
template <class T> __device__ __forceinline__ RawVec<T> ld_raw(const T* row, int c) {
    RawVec<T> r; r.c = *reinterpret_cast<const chunk16*>(row + c); return r;
}
template <> __device__ __forceinline__ RawVec<f32x3_t> ld_raw<f32x3_t>(const f32x3_t* row, int c) {
    const bf16_t* q = reinterpret_cast<const bf16_t*>(row) + x3_pos(c);
    RawVec<f32x3_t> r;
    r.h = *reinterpret_cast<const chunk16*>(q);
    r.l = *reinterpret_cast<const chunk16*>(q + 32);
    return r;
}
template <class T> __device__ __forceinline__ void st_raw(T* row, int c, const RawVec<T>& r) {
    *reinterpret_cast<chunk16*>(row + c) = r.c;
}
template <> __device__ __forceinline__ void st_raw<f32x3_t>(f32x3_t* row, int c, const RawVec<f32x3_t>& r) {
    bf16_t* q = reinterpret_cast<bf16_t*>(row) + x3_pos(c);
    *reinterpret_cast<chunk16*>(q) = r.h;
    *reinterpret_cast<chunk16*>(q + 32) = r.l;
}
template <class T> __device__ __forceinline__ void unpack_raw(const RawVec<T>& r, float* v) { unpack<T>(r.c, v); }
template <> __device__ __forceinline__ void unpack_raw<f32x3_t>(const RawVec<f32x3_t>& r, float* v) { x3_join8(r.h, r.l, v); }
template <class T> __device__ __forceinline__ RawVec<T> pack_raw(const float* v) { RawVec<T> r; r.c = pack<T>(v); return r; }
template <> __device__ __forceinline__ RawVec<f32x3_t> pack_raw<f32x3_t>(const float* v) {
    RawVec<f32x3_t> r; x3_split8(v, r.h, r.l); return r;
}
template <class T> __device__ __forceinline__ RawVec<T> zero_raw() { RawVec<T> r; r.c = zero_chunk(); return r; }
template <> __device__ __forceinline__ RawVec<f32x3_t> zero_raw<f32x3_t>() { RawVec<f32x3_t> r; r.h = zero_chunk(); r.l = zero_chunk(); return r; }
template <class T> __device__ __forceinline__ void ld_vec(const T* row, int c, float* v) { unpack_raw<T>(ld_raw<T>(row, c), v); }
template <class T> __device__ __forceinline__ void st_vec(T* row, int c, const float* v) { st_raw<T>(row, c, pack_raw<T>(v)); }

// four consecutive channels c .. c+3 (c % 4 == 0) of a row: the row-wise ViT kernels (one wave per 256-wide row) and
// the attention epilogues (an accumulator quad)
template <class T> __device__ __forceinline__ void ld4(const T* row, int c, float* v);
template <> __device__ __forceinline__ void ld4<float>(const float* row, int c, float* v) {
    unpack<float>(*reinterpret_cast<const chunk16*>(row + c), v);
}
template <> __device__ __forceinline__ void ld4<bf16_t>(const bf16_t* row, int c, float* v) {
    const unsigned* q = reinterpret_cast<const unsigned*>(row + c);
    const unsigned a = q[0], b = q[1];
    v[0] = bits_f(a << 16); v[1] = bits_f(a & 0xffff0000u); v[2] = bits_f(b << 16); v[3] = bits_f(b & 0xffff0000u);
}
template <> __device__ __forceinline__ void ld4<f32x3_t>(const f32x3_t* row, int c, float* v) {
    const bf16_t* p = reinterpret_cast<const bf16_t*>(row) + x3_pos(c);
    const unsigned* q = reinterpret_cast<const unsigned*>(p);
    const unsigned* ql = reinterpret_cast<const unsigned*>(p + 32);
    const unsigned a = q[0], b = q[1], la = ql[0], lb = ql[1];
    v[0] = bits_f(a << 16) + bits_f(la << 16); v[1] = bits_f(a & 0xffff0000u) + bits_f(la & 0xffff0000u);
    v[2] = bits_f(b << 16) + bits_f(lb << 16); v[3] = bits_f(b & 0xffff0000u) + bits_f(lb & 0xffff0000u);
}
template <class T> __device__ __forceinline__ void st4(T* row, int c, const float* v);
template <> __device__ __forceinline__ void st4<float>(float* row, int c, const float* v) {
    *reinterpret_cast<chunk16*>(row + c) = pack<float>(v);
}
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* row, int c, const float* v) {
    unsigned* q = reinterpret_cast<unsigned*>(row + c);
    q[0] = (unsigned)f2bf(v[0]).v | ((unsigned)f2bf(v[1]).v << 16);
    q[1] = (unsigned)f2bf(v[2]).v | ((unsigned)f2bf(v[3]).v << 16);
}
template <> __device__ __forceinline__ void st4<f32x3_t>(f32x3_t* row, int c, const float* v) {
    bf16_t* p = reinterpret_cast<bf16_t*>(row) + x3_pos(c);
    unsigned* q = reinterpret_cast<unsigned*>(p);
    unsigned* ql = reinterpret_cast<unsigned*>(p + 32);
    bf16_t h[4], l[4];
    MTN_UNROLL for (int e = 0; e < 4; ++e) { h[e] = f2bf(v[e]); l[e] = f2bf(v[e] - bf2f(h[e])); }
    q[0] = (unsigned)h[0].v | ((unsigned)h[1].v << 16); q[1] = (unsigned)h[2].v | ((unsigned)h[3].v << 16);
    ql[0] = (unsigned)l[0].v | ((unsigned)l[1].v << 16); ql[1] = (unsigned)l[2].v | ((unsigned)l[3].v << 16);
}
// one channel (reductions that walk a column)
template <class T> __device__ __forceinline__ float ld1(const T* row, int c) { return to_f<T>(row[c]); }
template <> __device__ __forceinline__ float ld1<f32x3_t>(const f32x3_t* row, int c) {
    const bf16_t* p = reinterpret_cast<const bf16_t*>(row) + x3_pos(c);
    return bf2f(p[0]) + bf2f(p[32]);
}
// value as it will read back from storage of type T (statistics are taken of what the consumers will see)
template <> __device__ __forceinline__ float round_to<f32x3_t>(float x) {
    const bf16_t h = f2bf(x);
    return bf2f(h) + bf2f(f2bf(x - bf2f(h)));
}

// Operand fragment of K-sub-step ks from a ROW of an XOR-swizzled row-major LDS tile (128-byte rows of
// 16-byte slots, slot s of row r stored at slot s ^ (r & 7); `sw` = r & 7, g = lane >> 4).
template <class T> struct RowFrag {
    static constexpr int VEC = 16 / (int)sizeof(T), KI = Mma<T>::KI;
    static __device__ __forceinline__ typename Mma<T>::Frag load(const T* row, int ks, int g, int sw) {
        const int e = ks * KI + g * (KI / 4);
        return Mma<T>::load(row + (((e / VEC) ^ sw) * VEC) + (e % VEC));
    }
};
// Split storage: the 128-byte row IS one 32-channel block [32 hi][32 lo]: lane group g takes slot g (hi of k = 8g .. 8g+7)
// and slot g + 4 (their lo halves) -- two plain 16-byte reads, conflict-free under the XOR swizzle (two adjacent slots
// are 2-way conflicted), NO conversion.  Activations arrive in this form from their producers, weights from the pack
// launch (optim.hip).
template <> struct RowFrag<f32x3_t> {
    static __device__ __forceinline__ Frag3 load(const f32x3_t* row, int ks, int g, int sw) {
        (void)ks;                                                        // KI = 32 = the whole 128-byte row
        Frag3 f;
        f.hi = load_bf16x8(row + ((g ^ sw) * 4));
        f.lo = load_bf16x8(row + (((g + 4) ^ sw) * 4));
        return f;
    }
};
template <class T> struct RowFragW {         // (weights and activations share one layout since round 4)
    static __device__ __forceinline__ typename Mma<T>::Frag load(const T* row, int ks, int g, int sw) { return RowFrag<T>::load(row, ks, g, sw); }
};

// operand fragment read down the reduction (row) axis of a [rows][pitch] LDS tile;
// base points at [k0][col0]; lane gets rows k0 + koff.., column col0 + (lane & 15).
template <class T, bool TR> struct ColFrag;
template <bool TR> struct ColFrag<float, TR> {
    static __device__ __forceinline__ float load(const float* base, int pitch, int lane) {
        return base[(lane >> 4) * pitch + (lane & 15)];
    }
};
template <bool TR> struct ColFrag<f32s_t, TR> {     // rows k0 + 8 g + j, j = 0..7: the k order of Mma<f32s_t>::load
    static __device__ __forceinline__ Frag3 load(const f32s_t* base, int pitch, int lane) {
        const float* p = reinterpret_cast<const float*>(base) + 8 * (lane >> 4) * pitch + (lane & 15);
        float x[8];
        MTN_UNROLL for (int j = 0; j < 8; ++j) x[j] = p[j * pitch];
        return Mma<f32x3_t>::split(x);
    }
};
template <> struct ColFrag<bf16_t, false> {
    static __device__ __forceinline__ bf16x8_t load(const bf16_t* base, int pitch, int lane) {
        bf16x8_t f;
        const unsigned short* p = reinterpret_cast<const unsigned short*>(base) + (8 * (lane >> 4)) * pitch + (lane & 15);
        unsigned short v[8];
        MTN_UNROLL for (int j = 0; j < 8; ++j) v[j] = p[j * pitch];
#ifdef MTN_EMU
        MTN_UNROLL for (int j = 0; j < 8; ++j) f[j] = v[j];
#else
        MTN_UNROLL for (int j = 0; j < 8; ++j) f[j] = __builtin_bit_cast(__bf16, v[j]);
#endif
        return f;
    }
};
template <> struct ColFrag<bf16_t, true> {
    // two transposed 4x16 block reads: rows 8g..8g+3 and 8g+4..8g+7 (g = lane>>4); lane 4q+p of
    // the group addresses row q, columns 4p..4p+3 and receives column (lane&15) of the 4 rows.
    static __device__ __forceinline__ bf16x8_t load(const bf16_t* base, int pitch, int lane) {
        const int g = lane >> 4, i = lane & 15;
        const bf16_t* p0 = base + (8 * g + (i >> 2)) * pitch + 4 * (i & 3);
        const bf16_t* p1 = p0 + 4 * pitch;
        bf16x8_t f;
#ifdef MTN_EMU
        emu::u16x4 a = emu::ds_read_tr16_b64(p0), b = emu::ds_read_tr16_b64(p1);
        MTN_UNROLL for (int j = 0; j < 4; ++j) { f[j] = a[j]; f[4 + j] = b[j]; }
#else
        // (the v4i16 form + per-element bit_cast is mis-lowered by hipcc 7.2: use v4bf16 + shuffle)
        typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) bf16x4_t* lds_ptr;
        const bf16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(p0));
        const bf16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(p1));
        f = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
#endif
        return f;
    }
};

__device__ __forceinline__ f32x4_t zero4() {
    f32x4_t z; z[0] = 0.f; z[1] = 0.f; z[2] = 0.f; z[3] = 0.f; return z;
}

// LDS-DMA through a buffer descriptor: lane l copies 16 bytes from base + voff (32-bit byte offset)
// to lds_block + 16*l; an offset beyond num_bytes is out of range and DELIVERS ZEROS, which is how
// the zero padding of the convolution (and the tap-invalid rows of a strided data gradient) is
// produced without a branch or 64-bit address arithmetic.
struct BufDesc {
#ifdef MTN_EMU
    const char* base; unsigned bytes;
#else
    __amdgpu_buffer_rsrc_t rsrc;
#endif
};
__device__ __forceinline__ BufDesc make_buf(const void* base, unsigned bytes) {
    BufDesc d;
#ifdef MTN_EMU
    d.base = static_cast<const char*>(base); d.bytes = bytes;
#else
    d.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
#endif
    return d;
}
__device__ __forceinline__ void buf_glds16(const BufDesc& d, unsigned voff, void* lds_block) {
#ifdef MTN_EMU
    char* dst = static_cast<char*>(lds_block) + 16 * (threadIdx.x & 63);
    if ((unsigned long long)voff + 16 <= d.bytes) std::memcpy(dst, d.base + voff, 16);
    else std::memset(dst, 0, 16);
#else
    __builtin_amdgcn_raw_ptr_buffer_load_lds(d.rsrc, (__attribute__((address_space(3))) void*)lds_block, 16, (int)voff, 0, 0, 0);
#endif
}

// ---- small math -----------------------------------------------------------------------------
// 1/x by v_rcp_f32 (1 ulp).  A plain `1.0f / x` compiles to the correctly rounded division sequence
// (v_div_scale x2, v_rcp, 4-5 v_fma, v_div_fmas, v_div_fixup: ~12 instructions), which made the
// BatchNorm / SiLU elementwise kernels VALU-bound below the HBM rate.
__device__ __forceinline__ float rcpf_(float x) {
#ifdef MTN_EMU
    return 1.0f / x;
#else
    return __builtin_amdgcn_rcpf(x);
#endif
}
// 2^x by v_exp_f32 (softmax: exp(s) = 2^(s log2 e), the factor folded into the score scale)
__device__ __forceinline__ float exp2f_(float x) {
#ifdef MTN_EMU
    return exp2f(x);
#else
    return __builtin_amdgcn_exp2f(x);
#endif
}
__device__ __forceinline__ float sigmoidf_(float x) { return rcpf_(1.0f + __expf(-x)); }
__device__ __forceinline__ float siluf_(float x) { return x * sigmoidf_(x); }
__device__ __forceinline__ float silu_grad_(float x) {   // d/dx x*sigmoid(x)
    const float s = sigmoidf_(x);
    return s * (1.0f + x * (1.0f - s));
}
__device__ __forceinline__ float geluf_(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// BatchNorm + activation backward, one 16-byte chunk: g = gradient of act(y*scale + shift [+ res]) as stored,
// v = raw conv output y (overwritten), r = residual or null  ->  dz (pre-activation gradient), xh (xhat).
// Shared by the BN backward passes (bn.hip) and the data-gradient epilogues that pre-reduce for them.
template <int VEC>
__device__ __forceinline__ void bn_dz_math(const float* g, float* v, const float* r, bool with_r, const float* scale, const float* shift,
                                           const float* mean, const float* rstd, bool silu, float* dz, float* xh) {
    MTN_UNROLL for (int e = 0; e < VEC; ++e) xh[e] = (v[e] - mean[e]) * rstd[e];
    if (silu) {
        MTN_UNROLL for (int e = 0; e < VEC; ++e) v[e] = v[e] * scale[e] + shift[e];
        if (with_r) {
            MTN_UNROLL for (int e = 0; e < VEC; ++e) v[e] += r[e];
        }
        MTN_UNROLL for (int e = 0; e < VEC; ++e) dz[e] = g[e] * silu_grad_(v[e]);
    } else {
        MTN_UNROLL for (int e = 0; e < VEC; ++e) dz[e] = g[e];
    }
}

__device__ __forceinline__ float wave_sum(float v) {
    MTN_UNROLL for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    MTN_UNROLL for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}
