// mtn_rt.h -- the one place that knows how kernels are compiled and launched.
//
// Product build (hipcc --offload-arch=gfx950): real HIP.  Test build (-DMTN_EMU, g++): the
// fiber executor in tests/emu/hipemu.h runs the same kernel sources on the CPU so their logic
// can be checked without a GPU.  The product library is never built with MTN_EMU.
#pragma once
#include <cstdint>
#include <cstddef>

#ifdef MTN_EMU
#include "hipemu.h"
#define MTN_LAUNCH(kern, grid, block, smem, stream, ...) \
    MTN_EMU_LAUNCH(kern, grid, block, smem, stream, __VA_ARGS__)
#define MTN_DYN_SMEM(name) char* name = emu::g_dyn_smem
#define MTN_UNROLL
#else
#include <hip/hip_runtime.h>
#define MTN_LAUNCH(kern, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kern, grid, block, smem, stream, __VA_ARGS__)
#define MTN_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
#define MTN_UNROLL _Pragma("unroll")
#endif

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute: a launch site keeps one
// MtnLdsAttr (bit d = the attribute is set on device d), so a process that moves to another GPU
// (model.to('cuda:1')) sets it there too instead of failing every launch that needs > 64 KB of LDS.
#ifndef MTN_EMU
struct MtnLdsAttr { unsigned long long done = 0; };
template <class K>
static inline void mtn_set_max_lds(MtnLdsAttr& st, K kern, size_t bytes) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && ((st.done >> dev) & 1ull)) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (dev >= 0 && dev < 64) st.done |= 1ull << dev;     // (idempotent: a race between two threads only repeats the call)
}
#endif

// Workgroup barrier that orders LDS traffic only.  hipcc's __syncthreads() is a fence + barrier and
// waits vmcnt(0) first, which drains a register prefetch of global loads at every barrier of a
// loop (the prefetch then hides nothing); this form waits for the wave's LDS operations and leaves
// global loads in flight (cdna_hip_programming.md section 5, "Pipelining across barriers").  Use it
// only where the data handed over at the barrier lives in LDS.
#ifdef MTN_EMU
#define MTN_BARRIER_LDS() __syncthreads()
#else
#define MTN_BARRIER_LDS()                                         \
    do {                                                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        \
        __builtin_amdgcn_s_barrier();                             \
        asm volatile("" ::: "memory");                            \
    } while (0)
#endif

// status codes of the C ABI (include/mtn.h)
#define MTN_OK 0
#define MTN_ERR_BAD_ARG 1
#define MTN_ERR_LAUNCH 2
#define MTN_ERR_UNSUPPORTED 3

void mtn_set_error(const char* fmt, ...);
int mtn_check_launch(const char* what);

#define MTN_REQUIRE(cond, ...)                     \
    do {                                           \
        if (!(cond)) {                             \
            mtn_set_error(__VA_ARGS__);            \
            return MTN_ERR_BAD_ARG;                \
        }                                          \
    } while (0)

static inline int mtn_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
