// hipemu.h -- TEST INFRASTRUCTURE: a minimal host-side executor for the HIP kernels in this directory.
//
// There is no GPU in the build container, so kernel *logic* (index maps, LDS layouts, barrier
// placement, MFMA fragment bookkeeping) is debugged by compiling the very same .hip sources for
// x86 with -DCFFM_EMU: every thread of a workgroup runs as a ucontext fiber on one OS thread,
// __syncthreads()/wave exchanges are cooperative yields, workgroups are spread over OS threads.
// The resulting libcffm_emu.so is loaded ONLY by tests/ (tests/emu.py); the product loader in
// vss_cffm_amd/_lib.py never looks for it and raises when libcffm_hip.so / a GPU is missing.
// Nothing here is a CPU fallback of the product path.
#pragma once
#ifndef CFFM_EMU
#error "hipemu.h is only for the -DCFFM_EMU host build"
#endif
#include <ucontext.h>

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static thread_local

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
#define hipMemcpyDeviceToDevice 3

namespace emu {

constexpr int kStack = 96 * 1024;
constexpr int kMaxWaves = 16;

struct Lane {
    ucontext_t ctx;
    dim3 tid;
    bool done;
};

struct WaveX {
    int count = 0;
    int gen = 0;
    alignas(16) unsigned char slot[64][96];
};

struct Block {
    dim3 bid, bdim, gdim;
    int nthreads = 0;
    int cur = 0;
    int live = 0;
    std::vector<Lane> lanes;
    std::vector<char> stacks;
    ucontext_t home;
    char* dynsmem = nullptr;
    int arrive = 0, gen = 0;
    WaveX waves[kMaxWaves];
    long spins = 0;
    const std::function<void()>* body = nullptr;
};

extern thread_local Block* g_blk;

inline void switch_next() {
    Block* b = g_blk;
    int from = b->cur;
    int nxt = from;
    for (int k = 0; k < b->nthreads; ++k) {
        nxt = (nxt + 1) % b->nthreads;
        if (!b->lanes[nxt].done) break;
    }
    if (b->lanes[nxt].done) {  // everybody finished
        swapcontext(&b->lanes[from].ctx, &b->home);
        return;
    }
    if (nxt == from) {
        if (++b->spins > 100000000L) {
            fprintf(stderr, "[hipemu] deadlock: lane %d spins alone (divergent barrier?)\n", from);
            abort();
        }
        return;
    }
    b->cur = nxt;
    swapcontext(&b->lanes[from].ctx, &b->lanes[nxt].ctx);
}

inline void lane_entry() {
    Block* b = g_blk;
    (*b->body)();
    b->lanes[b->cur].done = true;
    b->live--;
    if (b->live > 0 && b->arrive == b->live) {  // the others were only waiting for this lane
        b->arrive = 0;
        b->gen++;
    }
    switch_next();
}

inline void block_barrier() {
    Block* b = g_blk;
    int g = b->gen;
    if (++b->arrive == b->live) {
        b->arrive = 0;
        b->gen++;
        b->spins = 0;
    } else {
        long guard = 0;
        while (b->gen == g) {
            switch_next();
            if (++guard > 50000000L) { fprintf(stderr, "[hipemu] __syncthreads deadlock\n"); abort(); }
        }
    }
}

inline int lane_linear() {
    Block* b = g_blk;
    const dim3& t = b->lanes[b->cur].tid;
    return t.x + b->bdim.x * (t.y + b->bdim.y * t.z);
}

inline void wave_barrier() {
    Block* b = g_blk;
    int lin = lane_linear();
    WaveX& w = b->waves[lin >> 6];
    int width = b->nthreads - (lin & ~63);
    if (width > 64) width = 64;
    int g = w.gen;
    if (++w.count == width) {
        w.count = 0;
        w.gen++;
    } else {
        long guard = 0;
        while (w.gen == g) {
            switch_next();
            if (++guard > 50000000L) { fprintf(stderr, "[hipemu] wave exchange deadlock (divergent shuffle/mfma?)\n"); abort(); }
        }
    }
}

// every lane deposits `n` bytes; after the call slot[l] of all lanes of this wave is readable
// until release().
inline unsigned char (*deposit(const void* src, int n))[96] {
    Block* b = g_blk;
    int lin = lane_linear();
    WaveX& w = b->waves[lin >> 6];
    memcpy(w.slot[lin & 63], src, n);
    wave_barrier();
    return w.slot;
}
inline void release() { wave_barrier(); }

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);

}  // namespace emu

#define threadIdx (emu::g_blk->lanes[emu::g_blk->cur].tid)
#define blockIdx (emu::g_blk->bid)
#define blockDim (emu::g_blk->bdim)
#define gridDim (emu::g_blk->gdim)

static inline void __syncthreads() { emu::block_barrier(); }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

template <class T>
static inline T __shfl(T v, int src, int width = 64) {
    int lane = emu::lane_linear() & 63;
    int base = lane & ~(width - 1);
    auto s = emu::deposit(&v, sizeof(T));
    T r;
    memcpy(&r, s[base + (src & (width - 1))], sizeof(T));
    emu::release();
    return r;
}
template <class T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
    int lane = emu::lane_linear() & 63;
    return __shfl(v, (lane ^ mask) & (width - 1), width);
}
template <class T>
static inline T __shfl_down(T v, int d, int width = 64) {
    int lane = emu::lane_linear() & 63;
    int src = (lane & (width - 1)) + d;
    if (src >= width) src = lane & (width - 1);
    return __shfl(v, src, width);
}

static inline float atomicAdd(float* p, float v) {
    uint32_t* u = reinterpret_cast<uint32_t*>(p);
    uint32_t old = __atomic_load_n(u, __ATOMIC_RELAXED), neu;
    float f;
    do {
        memcpy(&f, &old, 4);
        f += v;
        memcpy(&neu, &f, 4);
    } while (!__atomic_compare_exchange_n(u, &old, neu, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    memcpy(&f, &old, 4);
    return f;
}
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned int atomicAdd(unsigned int* p, unsigned int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }

