// TEST INFRASTRUCTURE ONLY — a minimal host stand-in for <hip/hip_runtime.h> so that the kernels of
// cramtools_amd/csrc can be compiled with g++ and run under AddressSanitizer (one std::thread per lane, lock-step
// only at cross-lane operations).  It exists to find out-of-bounds accesses on the CPU before a kernel reaches a
// GPU; it is never built into libcramgpu.so and nothing in the package loads it.
#pragma once
#include <pthread.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#define HIP_SIM 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __constant__ static const
#define __launch_bounds__(...)
#define __align__(n) alignas(n)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct uint2 { uint32_t x, y; };
struct alignas(16) uint4 { uint32_t x, y, z, w; };
static inline uint2 make_uint2(uint32_t a, uint32_t b) { return uint2{a, b}; }
static inline uint4 make_uint4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return uint4{a, b, c, d}; }
struct alignas(16) int4 { int32_t x, y, z, w; };
static inline int4 make_int4(int32_t a, int32_t b, int32_t c, int32_t d) { return int4{a, b, c, d}; }

namespace hipsim {
struct WaveX {
    pthread_barrier_t bar;
    uint64_t slot[64];
};
struct BlockCtx {
    pthread_barrier_t bar;
    std::vector<WaveX> waves;
    std::vector<uint8_t> lds;
};
inline thread_local BlockCtx* g_blk = nullptr;
inline uint8_t* dyn_lds() { return g_blk->lds.data(); }
}  // namespace hipsim

inline thread_local dim3 threadIdx, blockIdx;
inline dim3 blockDim, gridDim;

namespace hipsim {
inline WaveX& wave() { return g_blk->waves[threadIdx.x >> 6]; }
inline int lane() { return (int)(threadIdx.x & 63); }
inline int wave_lanes() {
    int w = threadIdx.x >> 6;
    int n = (int)blockDim.x - w * 64;
    return n > 64 ? 64 : n;
}
// every lane publishes v, then reads lane src (src < 0: nothing)
inline uint64_t exchange(uint64_t v, int src) {
    WaveX& w = wave();
    w.slot[lane()] = v;
    pthread_barrier_wait(&w.bar);
    uint64_t r = (src >= 0 && src < wave_lanes()) ? w.slot[src] : v;
    pthread_barrier_wait(&w.bar);
    return r;
}
// One launch at a time, whatever host thread it comes from: blockDim / gridDim and the kernels' static "LDS" are global
// here (the device pool and the coalescing queues launch from one dispatcher thread per device).
inline std::mutex& launch_mutex() {
    static std::mutex m;
    return m;
}
template <class F>
void launch(dim3 grid, dim3 block, size_t lds, F f) {
    std::lock_guard<std::mutex> serial(launch_mutex());
    blockDim = block;
    gridDim = grid;
    const int nt = (int)block.x;
    const int nw = (nt + 63) / 64;
    for (unsigned by = 0; by < grid.y; by++)
        for (unsigned bx = 0; bx < grid.x; bx++) {
            BlockCtx ctx;
            ctx.waves.resize(nw);
            ctx.lds.assign(lds + 16, 0xCD);
            pthread_barrier_init(&ctx.bar, nullptr, nt);
            for (int w = 0; w < nw; w++) pthread_barrier_init(&ctx.waves[w].bar, nullptr, std::min(64, nt - 64 * w));
            std::vector<std::thread> th;
            th.reserve(nt);
            for (int t = 0; t < nt; t++)
                th.emplace_back([&, t]() {
                    g_blk = &ctx;
                    threadIdx = dim3(t, 0, 0);
                    blockIdx = dim3(bx, by, 0);
                    f();
                });
            for (auto& t : th) t.join();
            for (int w = 0; w < nw; w++) pthread_barrier_destroy(&ctx.waves[w].bar);
            pthread_barrier_destroy(&ctx.bar);
        }
}
}  // namespace hipsim

#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
    hipsim::launch((grid), (block), (lds), [&]() { kernel(__VA_ARGS__); })

// ---- cross-lane and misc device intrinsics ----
inline void __syncthreads() { pthread_barrier_wait(&hipsim::g_blk->bar); }
inline void __threadfence() { std::atomic_thread_fence(std::memory_order_seq_cst); }
inline void __threadfence_block() { std::atomic_thread_fence(std::memory_order_seq_cst); }
inline void __builtin_amdgcn_wave_barrier() { hipsim::exchange(0, -1); }
template <class T>
inline T __shfl(T v, int src) {
    uint64_t u = 0;
    memcpy(&u, &v, sizeof(T));
    u = hipsim::exchange(u, src & 63);
    T r;
    memcpy(&r, &u, sizeof(T));
    return r;
}
template <class T>
inline T __shfl_xor(T v, int m) { return __shfl(v, hipsim::lane() ^ m); }
template <class T>
inline T __shfl_up(T v, int d) {
    int s = hipsim::lane() - d;
    uint64_t u = 0;
    memcpy(&u, &v, sizeof(T));
    u = hipsim::exchange(u, s);  // s < 0 returns own value
    T r;
    memcpy(&r, &u, sizeof(T));
    return r;
}
inline unsigned long long __ballot(bool p) {
    hipsim::WaveX& w = hipsim::wave();
    w.slot[hipsim::lane()] = p ? 1 : 0;
    pthread_barrier_wait(&w.bar);
    unsigned long long m = 0;
    for (int i = 0; i < hipsim::wave_lanes(); i++) m |= (unsigned long long)(w.slot[i] & 1) << i;
    pthread_barrier_wait(&w.bar);
    return m;
}
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline unsigned __brev(unsigned v) {
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    return __builtin_bswap32(v);
}
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __clz(int v) { return v == 0 ? 32 : __builtin_clz((unsigned)v); }
inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
inline uint32_t __umul24(uint32_t a, uint32_t b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
inline uint32_t __builtin_amdgcn_mbcnt_lo(uint32_t m, uint32_t base) {
    int l = hipsim::lane();
    uint32_t lt = l >= 32 ? 0xFFFFFFFFu : ((1u << l) - 1);
    return base + (uint32_t)__builtin_popcount(m & lt);
}
inline uint32_t __builtin_amdgcn_mbcnt_hi(uint32_t m, uint32_t base) {
    int l = hipsim::lane();
    uint32_t lt = l <= 32 ? 0u : ((1u << (l - 32)) - 1);
    return base + (uint32_t)__builtin_popcount(m & lt);
}
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool) {
    int l = hipsim::lane(), s;
    if (ctrl <= 0xFF)
        s = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);
    else if (ctrl == 0x141)
        s = (l & ~7) | (7 - (l & 7));
    else if (ctrl == 0x140)
        s = (l & ~15) | (15 - (l & 15));
    else if (ctrl >= 0x111 && ctrl <= 0x11F)  // row_shr:n, bound_ctrl off: lanes without a source keep `old`
        s = (l & 15) >= (ctrl - 0x110) ? l - (ctrl - 0x110) : -1;
    else if (ctrl >= 0x121 && ctrl <= 0x12F)  // row_ror:n
        s = (l & ~15) | ((l - (ctrl - 0x120)) & 15);
    else if (ctrl == 0x142)  // row_bcast:15: lane 15 of a row to the next row
        s = l >= 16 ? (l & ~15) - 1 : -1;
    else if (ctrl == 0x143)  // row_bcast:31: lane 31 to rows 2 and 3
        s = l >= 32 ? 31 : -1;
    else
        abort();
    const bool enabled = ((row_mask >> (l >> 4)) & 1) && ((bank_mask >> ((l >> 2) & 3)) & 1);
    int r = (int)(uint32_t)hipsim::exchange((uint32_t)src, s);
    return (s < 0 || !enabled) ? old : r;
}
inline uint32_t atomicAdd(uint32_t* p, uint32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline uint32_t atomicOr(uint32_t* p, uint32_t v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
inline uint32_t atomicMin(uint32_t* p, uint32_t v) {
    uint32_t o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v < o && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {
    }
    return o;
}
inline int atomicMin(int* p, int v) {
    int o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v < o && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
    }
    return o;
}
inline int atomicCAS(int* p, int expect, int v) {
    __atomic_compare_exchange_n(p, &expect, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
    return expect;
}
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) {
    return __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
}
using std::max;
using std::min;

// ---- runtime API ----
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorNotReady = 600, hipErrorUnknown = 999 };
typedef void* hipStream_t;
typedef void* hipEvent_t;
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipHostMallocPortable = 1 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
struct hipDeviceProp_t { int multiProcessorCount; };
inline const char* hipGetErrorString(hipError_t) { return "sim"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) {  // HIPSIM_DEVICES: the device pool's routing is tested over several "devices"
    const char* e = getenv("HIPSIM_DEVICES");
    *n = e && atoi(e) > 0 ? atoi(e) : 1;
    return hipSuccess;
}
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { p->multiProcessorCount = 1; return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n); return *p ? hipSuccess : hipErrorUnknown; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n); return *p ? hipSuccess : hipErrorUnknown; }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
#define HIP_SYMBOL(x) (&(x))
template <class T>
inline hipError_t hipGetSymbolAddress(void** p, T* sym) { *p = (void*)sym; return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (void*)1; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (void*)1; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline long long clock64() { return 0; }
