// Launch-cost probe for the slab-folding kernels: N back-to-back dependent launches on one stream, timed with events.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/small tools/micro/small_kernels.hip && /tmp/small
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void empty_kernel(float* p) { if (p == nullptr) __builtin_trap(); }

constexpr int kLanes = 16;
__global__ __launch_bounds__(1024) void fold1024(const float* __restrict__ in, int rows, int W, int rpb, float* __restrict__ out) {
    __shared__ float red[kLanes][64];
    const int tl = threadIdx.x >> 6, cl = threadIdx.x & 63;
    const int c = blockIdx.x * 64 + cl;
    const int r0 = blockIdx.y * rpb;
    int r1 = r0 + rpb; if (r1 > rows) r1 = rows;
    float s = 0.f;
    if (c < W) for (int r = r0 + tl; r < r1; r += kLanes) s += in[(long)r * W + c];
    red[tl][cl] = s;
    __syncthreads();
    if (tl == 0 && c < W) { float t = 0.f; for (int q = 0; q < kLanes; ++q) t += red[q][cl]; out[(long)blockIdx.y * W + c] = t; }
}

// 256 threads: 4 row lanes x 64 float4 columns (W <= 256 per block.x), rows per block rpb
__global__ __launch_bounds__(256) void fold256(const float* __restrict__ in, int rows, int W, int rpb, float* __restrict__ out) {
    __shared__ float4 red[4][64];
    const int tl = threadIdx.x >> 6, cl = threadIdx.x & 63;
    const int c = (blockIdx.x * 64 + cl) * 4;
    const int r0 = blockIdx.y * rpb;
    int r1 = r0 + rpb; if (r1 > rows) r1 = rows;
    float4 s = {0, 0, 0, 0};
    if (c < W)
#pragma unroll 8
        for (int r = r0 + tl; r < r1; r += 4) { const float4 v = *reinterpret_cast<const float4*>(in + (long)r * W + c); s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    red[tl][cl] = s;
    __syncthreads();
    if (tl == 0 && c < W) {
        float4 t = red[0][cl];
        for (int q = 1; q < 4; ++q) { t.x += red[q][cl].x; t.y += red[q][cl].y; t.z += red[q][cl].z; t.w += red[q][cl].w; }
        *reinterpret_cast<float4*>(out + (long)blockIdx.y * W + c) = t;
    }
}

template <class F> float timeit(F f, int n, hipStream_t s) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; ++i) f();
    hipEventRecord(a, s);
    for (int i = 0; i < n; ++i) f();
    hipEventRecord(b, s); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms * 1000.f / n;
}

int main() {
    hipStream_t s; hipStreamCreate(&s);
    const int rows = 4608, W = 256;
    float *in, *out; hipMalloc(&in, (size_t)rows * W * 4 * 2); hipMalloc(&out, (size_t)rows * W * 4);
    hipMemset(in, 0, (size_t)rows * W * 4 * 2);
    const int n = 2000;
    printf("empty              %6.2f us\n", timeit([&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, in); }, n, s));
    printf("empty 144x1024     %6.2f us\n", timeit([&] { hipLaunchKernelGGL(empty_kernel, dim3(4, 36), dim3(1024), 0, s, in); }, n, s));
    printf("fold1024 4608->36  %6.2f us\n", timeit([&] { hipLaunchKernelGGL(fold1024, dim3(4, 36), dim3(1024), 0, s, in, rows, W, 128, out); }, n, s));
    printf("fold1024 36->1     %6.2f us\n", timeit([&] { hipLaunchKernelGGL(fold1024, dim3(4, 1), dim3(1024), 0, s, in, 36, W, 36, out); }, n, s));
    printf("fold256  4608->36  %6.2f us\n", timeit([&] { hipLaunchKernelGGL(fold256, dim3(1, 36), dim3(256), 0, s, in, rows, W, 128, out); }, n, s));
    printf("fold256  4608->144 %6.2f us\n", timeit([&] { hipLaunchKernelGGL(fold256, dim3(1, 144), dim3(256), 0, s, in, rows, W, 32, out); }, n, s));
    printf("fold256  4608->1   %6.2f us (one block)\n", timeit([&] { hipLaunchKernelGGL(fold256, dim3(1, 1), dim3(256), 0, s, in, rows, W, rows, out); }, 200, s));
    printf("fold256  144->1    %6.2f us\n", timeit([&] { hipLaunchKernelGGL(fold256, dim3(1, 1), dim3(256), 0, s, in, 144, W, 144, out); }, n, s));
    printf("fold256  36->1     %6.2f us\n", timeit([&] { hipLaunchKernelGGL(fold256, dim3(1, 1), dim3(256), 0, s, in, 36, W, 36, out); }, n, s));
    // pairs as the step runs them: fold + final
    printf("fold1024 pair      %6.2f us\n", timeit([&] { hipLaunchKernelGGL(fold1024, dim3(4, 36), dim3(1024), 0, s, in, rows, W, 128, out);
                                                          hipLaunchKernelGGL(fold1024, dim3(4, 1), dim3(1024), 0, s, out, 36, W, 36, in + rows * W); }, n, s));
    printf("fold256 pair       %6.2f us\n", timeit([&] { hipLaunchKernelGGL(fold256, dim3(1, 144), dim3(256), 0, s, in, rows, W, 32, out);
                                                          hipLaunchKernelGGL(fold256, dim3(1, 1), dim3(256), 0, s, out, 144, W, 144, in + rows * W); }, n, s));
    return 0;
}
