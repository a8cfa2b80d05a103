// hipemu.h -- TEST INFRASTRUCTURE ONLY.
//
// A single-OS-thread, fiber-based executor for the HIP kernels under
// hand-gesture-recognition_amd/csrc, so that their *logic* (indexing, LDS layouts, MFMA fragment
// maps, barriers) can be exercised on the CPU-only build container at tiny sizes.  Every GPU
// thread is a fiber (own stack + a ~10 ns register switch); blocks run one after another; __syncthreads(), wave shuffles and
// MFMA are rendezvous points between the fibers of a block / a 64-lane wave.  Fibers are resumed
// round-robin, so a missing barrier between an LDS write and a read shows up as wrong data.
//
// The product never loads a library built with this header (see lib.py: only libmtn_hip.so);
// tests/ build libmtn_emu.so from the same kernel sources with -DMTN_EMU.
//
// MFMA lane maps follow /opt/skills/guides/cdna_hip_programming.md section 3; they are
// re-validated against the hardware by tests/test_mfma_layout.py (-m gpu).
#pragma once
#if defined(__SANITIZE_ADDRESS__)
#define MTN_EMU_ASAN 1
#include <sanitizer/asan_interface.h>
#endif
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
inline hipError_t hipGetLastError() { return 0; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }

struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
inline float4 make_float4(float a, float b, float c, float d) { return {a, b, c, d}; }
inline float2 make_float2(float a, float b) { return {a, b}; }
inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return {a, b, c, d}; }
inline uint2 make_uint2(unsigned a, unsigned b) { return {a, b}; }

inline dim3 threadIdx, blockIdx, blockDim, gridDim;

// Minimal x86-64 context switch (callee-saved registers + stack pointer): ~10 ns, no syscalls
// (swapcontext costs two rt_sigprocmask calls per switch and dominated the emulation time).
extern "C" void mtn_emu_switch(void** save_sp, void* load_sp);
asm(".text\n"
    ".weak mtn_emu_switch\n"
    ".type mtn_emu_switch,@function\n"
    "mtn_emu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n");

namespace emu {

constexpr int kStack = 96 * 1024;
constexpr int kWave = 64;

struct Fiber {
    void* sp;
    bool done;
};

struct Wave {
    int nlanes = 0;     // lanes that exist
    int alive = 0;      // lanes that have not returned
    int count = 0;      // arrivals at the current collective
    unsigned gen = 0;
    alignas(16) unsigned char slot[2][kWave][64];
};

inline void* g_main_sp = nullptr;
inline std::vector<Fiber> g_fibers;
inline std::vector<Wave> g_waves;
inline std::vector<char> g_stacks;
inline std::function<void()> g_body;
inline int g_cur = 0, g_n = 0, g_alive = 0;
inline int g_bar_count = 0;
inline unsigned g_bar_gen = 0;
inline unsigned long g_progress = 0;
inline char* g_dyn_smem = nullptr;

inline void yield() { mtn_emu_switch(&g_fibers[g_cur].sp, g_main_sp); }
inline int lane() { return g_cur & (kWave - 1); }
inline Wave& wave() { return g_waves[g_cur / kWave]; }

inline void barrier_release_if_complete() {
    if (g_bar_count > 0 && g_bar_count >= g_alive) {
        g_bar_count = 0;
        ++g_bar_gen;
        ++g_progress;
    }
}
inline void wave_release_if_complete(Wave& w) {
    if (w.count > 0 && w.count >= w.alive) {
        w.count = 0;
        ++w.gen;
        ++g_progress;
    }
}

inline void trampoline() {
    g_body();
    g_fibers[g_cur].done = true;
    --g_alive;
    Wave& w = wave();
    --w.alive;
    ++g_progress;
    barrier_release_if_complete();   // threads that returned no longer take part
    wave_release_if_complete(w);
    mtn_emu_switch(&g_fibers[g_cur].sp, g_main_sp);
    std::abort();                    // a finished fiber is never resumed
}

// Rendezvous of the live lanes of the calling fiber's wave.  Returns the slot parity used.
inline int wave_rendezvous(const void* deposit, int bytes) {
    Wave& w = wave();
    const unsigned gen = w.gen;
    const int p = gen & 1;
    if (bytes) std::memcpy(w.slot[p][lane()], deposit, bytes);
    ++w.count;
    wave_release_if_complete(w);
    while (w.gen == gen) yield();
    return p;
}

template <class F>
inline void launch(dim3 grid, dim3 block, size_t smem, F body) {
    const int n = block.x * block.y * block.z;
    if (n <= 0 || n > 1024) { std::fprintf(stderr, "hipemu: bad block size %d\n", n); std::abort(); }
    g_n = n;
    if ((int)g_fibers.size() < n) g_fibers.resize(n);
    if (g_stacks.size() < (size_t)n * kStack) g_stacks.resize((size_t)n * kStack);
    const int nw = (n + kWave - 1) / kWave;
    if ((int)g_waves.size() < nw) g_waves.resize(nw);
    std::vector<char> dyn(smem + 64);
    blockDim = block;
    gridDim = grid;
    g_body = body;
    for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
    for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx = dim3(bx, by, bz);
#ifdef MTN_EMU_ASAN
        __asan_unpoison_memory_region(dyn.data(), dyn.size());
#endif
        std::memset(dyn.data(), 0xFF, dyn.size());   // LDS is uninitialised: poison with NaNs
        g_dyn_smem = (char*)(((uintptr_t)dyn.data() + 63) & ~(uintptr_t)63);
#ifdef MTN_EMU_ASAN
        // sanitizer build (tools/asan_emu.sh): the alignment slack behind the requested bytes is poisoned, so an LDS
        // access one byte past the launch's dynamic shared memory is reported like any other heap overflow
        __asan_poison_memory_region(g_dyn_smem + smem, (size_t)(dyn.data() + dyn.size() - (g_dyn_smem + smem)));
#endif
        g_alive = n;
        g_bar_count = 0;
        for (int w = 0; w < nw; ++w) {
            g_waves[w].nlanes = g_waves[w].alive = (w == nw - 1) ? n - w * kWave : kWave;
            g_waves[w].count = 0;
        }
        for (int i = 0; i < n; ++i) {
            Fiber& f = g_fibers[i];
            f.done = false;
            // initial frame: six zeroed callee-saved registers, then the entry point as return address;
            // 16-byte aligned so that the entry sees the ABI's (rsp % 16 == 8) after the `ret`
            uintptr_t top = ((uintptr_t)(g_stacks.data() + (size_t)(i + 1) * kStack) & ~(uintptr_t)15) - 64;
            void** sp = reinterpret_cast<void**>(top);
            for (int q = 0; q < 6; ++q) sp[q] = nullptr;
            sp[6] = reinterpret_cast<void*>(&trampoline);
            sp[7] = nullptr;
            f.sp = sp;
        }
        unsigned long stalled_rounds = 0;
        while (g_alive > 0) {
            const unsigned long before = g_progress;
            for (int i = 0; i < n; ++i) {
                if (g_fibers[i].done) continue;
                g_cur = i;
                threadIdx = dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
                mtn_emu_switch(&g_main_sp, g_fibers[i].sp);
            }
            if (g_progress == before) {
                if (++stalled_rounds > 4) {
                    std::fprintf(stderr, "hipemu: deadlock in block (%u,%u,%u): %d threads alive, "
                                 "%d at barrier\n", bx, by, bz, g_alive, g_bar_count);
                    std::abort();
                }
            } else {
                stalled_rounds = 0;
            }
        }
    }
#ifdef MTN_EMU_ASAN
    __asan_unpoison_memory_region(dyn.data(), dyn.size());
#endif
}

// ---- shuffles ---------------------------------------------------------------------------
template <class T>
inline T shfl_from(T v, int src) {
    static_assert(sizeof(T) <= 64, "");
    const int p = wave_rendezvous(&v, sizeof(T));
    T r;
    Wave& w = wave();
    if (src < 0 || src >= w.nlanes) src = lane();
    std::memcpy(&r, w.slot[p][src], sizeof(T));
    return r;
}

// ---- MFMA -------------------------------------------------------------------------------
struct bf16x8 { unsigned short v[8]; unsigned short& operator[](int i) { return v[i]; }
                const unsigned short& operator[](int i) const { return v[i]; } };
struct f32x4 { float v[4]; float& operator[](int i) { return v[i]; }
               const float& operator[](int i) const { return v[i]; } };
struct f32x16 { float v[16]; float& operator[](int i) { return v[i]; }
                const float& operator[](int i) const { return v[i]; } };

inline float bf2f(unsigned short h) { uint32_t u = (uint32_t)h << 16; float f; std::memcpy(&f, &u, 4); return f; }

struct AB16 { bf16x8 a, b; };
struct ABf { float a, b; };

// D = A*B + C, 16x16x32 bf16.  lane l holds A[l&15][8*(l>>4)+j], B[8*(l>>4)+j][l&15];
// D: col = l&15, row = 4*(l>>4)+reg.
inline f32x4 mfma_16x16x32_bf16(const bf16x8& a, const bf16x8& b, f32x4 c) {
    AB16 d{a, b};
    const int p = wave_rendezvous(&d, sizeof(d));
    Wave& w = wave();
    const int l = lane(), col = l & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int k = 0; k < 32; ++k) {
            const AB16* sa = (const AB16*)w.slot[p][row + 16 * (k >> 3)];
            const AB16* sb = (const AB16*)w.slot[p][col + 16 * (k >> 3)];
            acc = std::fmaf(bf2f(sa->a[k & 7]), bf2f(sb->b[k & 7]), acc);
        }
        c[r] = acc;
    }
    return c;
}

// 32x32x16 bf16.  lane l (r=l&31,h=l>>5) holds A[r][8h+j], B[8h+j][r];
// D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
inline f32x16 mfma_32x32x16_bf16(const bf16x8& a, const bf16x8& b, f32x16 c) {
    AB16 d{a, b};
    const int p = wave_rendezvous(&d, sizeof(d));
    Wave& w = wave();
    const int l = lane(), col = l & 31;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 16; ++k) {
            const AB16* sa = (const AB16*)w.slot[p][row + 32 * (k >> 3)];
            const AB16* sb = (const AB16*)w.slot[p][col + 32 * (k >> 3)];
            acc = std::fmaf(bf2f(sa->a[k & 7]), bf2f(sb->b[k & 7]), acc);
        }
        c[r] = acc;
    }
    return c;
}

// 16x16x4 f32: lane l holds A[l&15][l>>4], B[l>>4][l&15]; exact k-ordered fmaf chain.
inline f32x4 mfma_16x16x4_f32(float a, float b, f32x4 c) {
    ABf d{a, b};
    const int p = wave_rendezvous(&d, sizeof(d));
    Wave& w = wave();
    const int l = lane(), col = l & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            const ABf* sa = (const ABf*)w.slot[p][row + 16 * k];
            const ABf* sb = (const ABf*)w.slot[p][col + 16 * k];
            acc = std::fmaf(sa->a, sb->b, acc);
        }
        c[r] = acc;
    }
    return c;
}

// 32x32x2 f32: lane l holds A[l&31][l>>5], B[l>>5][l&31].
inline f32x16 mfma_32x32x2_f32(float a, float b, f32x16 c) {
    ABf d{a, b};
    const int p = wave_rendezvous(&d, sizeof(d));
    Wave& w = wave();
    const int l = lane(), col = l & 31;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            const ABf* sa = (const ABf*)w.slot[p][row + 32 * k];
            const ABf* sb = (const ABf*)w.slot[p][col + 32 * k];
            acc = std::fmaf(sa->a, sb->b, acc);
        }
        c[r] = acc;
    }
    return c;
}

// ds_read_b64_tr_b16 (cdna_hip_programming.md T10): per 16-lane group, lane 4q+p supplies the
// address of row q, columns 4p..4p+3 of a 4x16 block of 16-bit elements; lane i of the group
// receives column i of the 4 rows (row q in element q).
struct u16x4 { unsigned short v[4]; unsigned short& operator[](int i) { return v[i]; }
               const unsigned short& operator[](int i) const { return v[i]; } };
inline u16x4 ds_read_tr16_b64(const void* addr) {
    const int p = wave_rendezvous(&addr, sizeof(addr));
    Wave& w = wave();
    const int l = lane(), g = l & ~15, i = l & 15;
    u16x4 r;
    for (int q = 0; q < 4; ++q) {
        const unsigned short* src;
        std::memcpy(&src, w.slot[p][g + 4 * q + (i >> 2)], sizeof(src));
        r[q] = src[i & 3];
    }
    return r;
}

}  // namespace emu

inline void __syncthreads() {
    using namespace emu;
    const unsigned gen = g_bar_gen;
    ++g_bar_count;
    barrier_release_if_complete();
    while (g_bar_gen == gen) yield();
}

template <class T> inline T __shfl_xor(T v, int mask, int width = 64) {
    (void)width; return emu::shfl_from(v, emu::lane() ^ mask);
}
template <class T> inline T __shfl_down(T v, unsigned d, int width = 64) {
    const int l = emu::lane(); const int src = ((l % width) + (int)d < width) ? l + (int)d : l;
    return emu::shfl_from(v, src);
}
template <class T> inline T __shfl_up(T v, unsigned d, int width = 64) {
    const int l = emu::lane(); const int src = ((l % width) >= (int)d) ? l - (int)d : l;
    return emu::shfl_from(v, src);
}
template <class T> inline T __shfl(T v, int src, int width = 64) {
    const int l = emu::lane(); return emu::shfl_from(v, (l / width) * width + (src % width));
}

inline void __threadfence() {}      // (blocks run one after another on one host thread)
template <class T> inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
inline int atomicOr(int* p, int v) { int o = *p; *p = o | v; return o; }
inline float atomicMax(float* p, float v) { float o = *p; if (v > o) *p = v; return o; }
inline int atomicMax(int* p, int v) { int o = *p; if (v > o) *p = v; return o; }

inline float __expf(float x) { return std::exp(x); }
inline float __logf(float x) { return std::log(x); }
inline float __fdividef(float a, float b) { return a / b; }
inline float rsqrtf(float x) { return 1.0f / std::sqrt(x); }
inline float __frcp_rn(float x) { return 1.0f / x; }
#include <math.h>

#define MTN_EMU_LAUNCH(kern, grid, block, smem, stream, ...) \
    emu::launch((grid), (block), (smem), [&]() { kern(__VA_ARGS__); })
