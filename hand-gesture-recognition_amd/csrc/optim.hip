// optim.hip -- flat fused AdamW over all 7.4 M parameters and the per-step weight repack.
//
// Reference: torch.optim.AdamW(model.parameters(), lr) with defaults betas (0.9, 0.999), eps 1e-8,
// weight_decay 0.01 on EVERY tensor (train.py:50-51).  Parameters, gradients and both moments
// live in four flat fp32 buffers (the nn.Parameters are views), so one launch replaces the
// reference's 114 per-tensor updates.  Step-dependent scalars (step count, bias corrections,
// lr, gradient scale 1/world_size) are read from a small device-resident state block so the
// whole step can be replayed from a hipGraph.
//
// state f32[16]: 0 lr, 1 beta1, 2 beta2, 3 eps, 4 weight_decay, 5 step, 6 bias_corr1,
//                7 bias_corr2, 8 grad_scale
#include <type_traits>
#include "mtn_types.h"
#include "mtn_kernels.h"

namespace {

__global__ void adamw_tick_kernel(float* __restrict__ st) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float t = st[5] + 1.0f;
        st[5] = t;
        st[6] = 1.0f - powf(st[1], t);
        st[7] = 1.0f - powf(st[2], t);
    }
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long n4, const float* __restrict__ st) {
    const float lr = st[0], b1 = st[1], b2 = st[2], eps = st[3], wd = st[4], bc1 = st[6], bc2 = st[7], gs = st[8];
    const float step_size = lr / bc1, rs2 = rsqrtf(bc2), decay = 1.0f - lr * wd;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 pp = reinterpret_cast<float4*>(p)[i], gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        float* pa = &pp.x; float* ga = &gg.x; float* ma = &mm.x; float* va = &vv.x;
        MTN_UNROLL for (int e = 0; e < 4; ++e) {
            const float gr = ga[e] * gs;
            pa[e] *= decay;
            ma[e] = b1 * ma[e] + (1.0f - b1) * gr;
            va[e] = b2 * va[e] + (1.0f - b2) * gr * gr;
            const float denom = sqrtf(va[e]) * rs2 + eps;
            pa[e] -= step_size * (ma[e] / denom);
        }
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
}

// desc[t] = {src_off, dst_off, O, I, taps, mode}; mode 0: dst[o][tap][i], mode 1: dst[i][tap][o]; src is OIHW [o][i][tap]
template <class T>
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ src, const float* __restrict__ scale,
                                                   T* __restrict__ dst, const int* __restrict__ desc) {
    const int* d = desc + blockIdx.y * 7;
    const int so = d[0], dof = d[1], O = d[2], I = d[3], taps = d[4], mode = d[5] & 1;
    // split storage (T = f32x3_t): a packed row of K values is stored as K / 32 blocks of [32 hi bf16][32 lo bf16], k in
    // natural order -- the layout of the activations (mtn_types.h) -- unless the tensor asks for plain fp32 (d[5] & 2: the
    // stem, whose kernels multiply in exact f32).  Same bytes, same offsets: element (row, k) of the fp32 layout becomes
    // two bf16 at (row, block k / 32, k % 32) of the hi and the lo half.
    constexpr bool X3 = std::is_same<T, f32x3_t>::value;
    const bool presplit = X3 && !(d[5] & 2);
    auto put = [&](long row_len, long row, int k, float v) {
        if (presplit) {
            bf16_t* base = reinterpret_cast<bf16_t*>(dst + dof + row * row_len + (k & ~31));
            const bf16_t h = f2bf(v);
            const int pos = k & 31;
            base[pos] = h;
            base[32 + pos] = f2bf(v - bf2f(h));
        } else {
            if constexpr (X3) reinterpret_cast<float*>(dst)[dof + row * row_len + k] = v;
            else dst[dof + row * row_len + k] = from_f<T>(v);
        }
    };
    // d[6] >= 0: fold a per-output-channel factor scale[d[6] + o] into the weights (inference: the BatchNorm's
    // gamma / sqrt(running_var + eps), export.py:41-45 runs the model in eval()); forward layout only
    const float* sc = (d[6] >= 0 && scale != nullptr && mode == 0) ? scale + d[6] : nullptr;
    const int n = O * I * taps;
    if (mode == 1) {
        // dst[i][tap][o] is the TRANSPOSE of the source matrix [O][I*taps]: 64 x 64 tiles through LDS, so
        // that both the fp32 reads and the T writes run along contiguous rows (an element-wise gather
        // touches a different 128-byte line per lane: 32x the bytes through the memory-side cache)
        __shared__ float tile[64][65];
        const int R = O, Cn = I * taps;
        const int tc = (Cn + 63) / 64, nt = ((R + 63) / 64) * tc;
        const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
        for (int t = blockIdx.x; t < nt; t += gridDim.x) {
            const int r0 = (t / tc) * 64, c0 = (t % tc) * 64;
            MTN_UNROLL for (int k = 0; k < 16; ++k) {
                const int r = r0 + ty + 4 * k, c = c0 + tx;
                tile[ty + 4 * k][tx] = (r < R && c < Cn) ? src[so + r * Cn + c] : 0.f;
            }
            __syncthreads();
            MTN_UNROLL for (int k = 0; k < 16; ++k) {
                const int c = c0 + ty + 4 * k, r = r0 + tx;
                // destination row = input channel i = c / taps, its K index = (c % taps) * R + r
                if (c < Cn && r < R) put((long)taps * R, c / taps, (c % taps) * R + r, tile[tx][ty + 4 * k]);
            }
            __syncthreads();
        }
        return;
    }
    if (mode == 0 && taps == 9) {
        // dst[o][tap][i] of a 3x3 kernel: one thread per (o, i) reads its nine taps -- 36 contiguous bytes, the wave 2.3 KB
        // in a row -- and writes one value per tap, each tap's 64 lanes side by side (the element-wise form below walks the
        // source with a 36-byte stride nine separate times: 4.4x the bytes, profiles/r03_per_kernel_hbm_v3.txt)
        const int pairs = O * I;
        for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < pairs; idx += gridDim.x * blockDim.x) {
            const int o = (int)((unsigned)idx / (unsigned)I), i = idx - o * I;
            const float* sp = src + so + (long)idx * 9;
            const float f = sc ? sc[o] : 1.0f;
            float v[9];
            MTN_UNROLL for (int t = 0; t < 9; ++t) v[t] = sp[t];
            MTN_UNROLL for (int t = 0; t < 9; ++t) put((long)9 * I, o, t * I + i, v[t] * f);
        }
        return;
    }
    // inner = the destination's fastest axis (I for [o][tap][i], O for [i][tap][o]); one runtime division
    // per element (by inner), the tap split is by a literal 1 or 9 (the model's kernels), generic otherwise
    const int inner = mode == 0 ? I : O;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
        const int q = (int)((unsigned)idx / (unsigned)inner);
        const int f = idx - q * inner;                                           // idx % inner
        int outer, tap;
        if (taps == 1) { outer = q; tap = 0; }
        else if (taps == 9) { outer = q / 9; tap = q - outer * 9; }
        else { outer = q / taps; tap = q - outer * taps; }
        const int o = mode == 0 ? outer : f, i = mode == 0 ? f : outer;
        const float wv = src[so + (o * I + i) * taps + tap];
        // (idx = outer * (taps * inner) + tap * inner + f: row = outer, K index = tap * inner + f)
        put((long)taps * inner, outer, tap * inner + f, sc ? wv * sc[o] : wv);
    }
}

}  // namespace

int mtn_adamw_run(float* p, const float* g, float* m, float* v, long n, float* state, hipStream_t stream) {
    MTN_REQUIRE(n > 0 && n % 4 == 0, "adamw: n=%ld must be a positive multiple of 4", n);
    MTN_REQUIRE(((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16 == 0, "adamw: buffers must be 16-byte aligned");
    MTN_LAUNCH(adamw_tick_kernel, dim3(1), dim3(64), 0, stream, state);
    const long n4 = n / 4;
    long blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    MTN_LAUNCH(adamw_kernel, dim3((int)blocks), dim3(256), 0, stream, p, g, m, v, n4, (const float*)state);
    return mtn_check_launch("adamw");
}

int mtn_pack_run(const float* src, const float* scale, void* dst, const int* desc, int ntensors, int max_elems, int dtype, hipStream_t stream) {
    MTN_REQUIRE(ntensors > 0 && max_elems > 0, "pack: bad sizes");
    int bx = (max_elems + 255) / 256;
    if (bx > 256) bx = 256;
    if (dtype == MTN_BF16) MTN_LAUNCH(pack_kernel<bf16_t>, dim3(bx, ntensors), dim3(256), 0, stream, src, scale, (bf16_t*)dst, desc);
    else if (dtype == MTN_F32X3) MTN_LAUNCH(pack_kernel<f32x3_t>, dim3(bx, ntensors), dim3(256), 0, stream, src, scale, (f32x3_t*)dst, desc);
    else MTN_LAUNCH(pack_kernel<float>, dim3(bx, ntensors), dim3(256), 0, stream, src, scale, (float*)dst, desc);
    return mtn_check_launch("pack");
}
