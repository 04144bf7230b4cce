// Packs the right-aligned streams the encoder leaves in its slots into one dense arena (HBM-bound copy).
// No reference counterpart beyond ExternalCompression#toByteArray L146-150 (the per-block exact-size copy).
#include "internal.h"
#ifndef HIP_SIM
#include <hipcub/hipcub.hpp>
#else
#include <algorithm>
#endif

namespace {

// single workgroup: exclusive scan of align_up(max(len,0)) over n pieces -> dst_off[n+1]
__global__ __launch_bounds__(1024) void gather_scan_kernel(const int32_t* len, int32_t n, int32_t align, int64_t* dst_off,
                                                           int64_t dst_cap, int32_t* err) {
    __shared__ int64_t part[1024];
    const int t = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int lo = t * per, hi = min(n, lo + per);
    int64_t s = 0;
    for (int i = lo; i < hi; i++) {
        int64_t L = len[i] > 0 ? len[i] : 0;
        s += (L + align - 1) & ~(int64_t)(align - 1);
    }
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        int64_t v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int64_t base = part[t] - s;
    for (int i = lo; i < hi; i++) {
        dst_off[i] = base;
        int64_t L = len[i] > 0 ? len[i] : 0;
        base += (L + align - 1) & ~(int64_t)(align - 1);
    }
    if (t == 1023) {
        dst_off[n] = part[1023];
        *err = part[1023] > dst_cap ? CRAMGPU_E_INSUFFICIENT : 0;
    }
}

// one workgroup per (piece, 64 KiB chunk)
__global__ __launch_bounds__(256) void gather_copy_kernel(const uint8_t* src, const int64_t* pos, const int32_t* len,
                                                          const int64_t* dst_off, uint8_t* dst, const int32_t* err) {
    if (*err) return;
    const int b = blockIdx.x;
    const int32_t L = len[b];
    const int64_t c0 = (int64_t)blockIdx.y * 65536;
    if (L <= 0 || c0 >= L) return;
    const uint8_t* s = src + pos[b] + c0;
    uint8_t* d = dst + dst_off[b] + c0;
    const int32_t m = (int32_t)min((int64_t)65536, (int64_t)L - c0);
    // head until the destination is 16-byte aligned, then 16-byte stores fed by unaligned 16-byte loads
    const int head = (int)((16 - ((uintptr_t)d & 15)) & 15);
    const int h = head < m ? head : m;
    if ((int)threadIdx.x < h) d[threadIdx.x] = s[threadIdx.x];
    const int body = (m - h) & ~15;
    for (int i = threadIdx.x * 16; i < body; i += 256 * 16) {
        uint4 v;
        __builtin_memcpy(&v, s + h + i, 16);
        *reinterpret_cast<uint4*>(d + h + i) = v;
    }
    const int tail0 = h + body;
    if (tail0 + (int)threadIdx.x < m) d[tail0 + threadIdx.x] = s[tail0 + threadIdx.x];
}

}  // namespace

hipError_t launch_gather(const uint8_t* src, const int64_t* pos, const int32_t* len, int32_t n, int32_t align,
                         uint8_t* dst, int64_t dst_cap, int64_t* dst_off, int32_t* err, int32_t max_len,
                         hipStream_t s) {
    hipLaunchKernelGGL(gather_scan_kernel, dim3(1), dim3(1024), 0, s, len, n, align, dst_off, dst_cap, err);
    const int chunks = max(1, (max_len + 65535) / 65536);
    hipLaunchKernelGGL(gather_copy_kernel, dim3(n, chunks), dim3(256), 0, s, src, pos, len, dst_off, dst, err);
    return hipGetLastError();
}

// ---- the block table of a launch made on the device (cramgpu_rans_encode_series_dev) ----
// The encoder straight behind the data-series writer: block c * n_ids + k is series ids[k] of container c.  Its input lies
// where W put the stream, its output slot is RANS#allocateIfNeeded's bound (cramgpu_rans_bound), the slots one after the
// other on 16-byte boundaries (an exclusive scan by one workgroup, as gather_scan_kernel).
namespace {

__global__ __launch_bounds__(1024) void series_blocks_kernel(const int64_t* series_off, const int32_t* series_len, int32_t n_cont,
                                                             SeriesPick pick, int64_t in_shift, int64_t out_cap,
                                                             const int32_t* w_status, cramgpu_block* blocks, int32_t* status) {
    __shared__ int64_t part[1024];
    __shared__ int32_t bad;
    const int t = threadIdx.x;
    const int n = n_cont * pick.n_ids;
    const int per = (n + 1023) / 1024;
    const int lo = min(n, t * per), hi = min(n, lo + per);
    if (t == 0) bad = w_status != nullptr && *w_status != 0 ? *w_status : 0;
    __syncthreads();
    auto slot_of = [&](int i, int32_t& len, int32_t& cap) -> int64_t {
        const int c = i / pick.n_ids, k = i - c * pick.n_ids;
        len = series_len[(size_t)c * CRAMGPU_N_SERIES + pick.ids[k]];
        if (len < 0) len = 0;
        cap = (int32_t)(1.05 * (double)len + 257.0 * 257.0 * 3.0 + 4.0);  // cramgpu_rans_bound
        return ((int64_t)cap + 15) & ~(int64_t)15;
    };
    int64_t sum = 0;
    for (int i = lo; i < hi; i++) {
        int32_t len, cap;
        sum += slot_of(i, len, cap);
    }
    part[t] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int64_t v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    const bool over = part[1023] + 16 > out_cap;
    const bool empty = over || bad != 0;  // nothing is coded then: every block is an empty one
    int64_t base = part[t] - sum;
    for (int i = lo; i < hi; i++) {
        int32_t len, cap;
        const int64_t slot = slot_of(i, len, cap);
        const int c = i / pick.n_ids, k = i - c * pick.n_ids;
        cramgpu_block b;
        b.in_off = empty ? 0 : series_off[(size_t)c * CRAMGPU_N_SERIES + pick.ids[k]] + in_shift;
        b.out_off = empty ? 0 : base;
        b.in_len = empty ? 0 : len;
        b.out_cap = empty ? 0 : cap;
        b.order = pick.orders[k];
        b.reserved = 0;
        blocks[i] = b;
        base += slot;
    }
    if (t == 0) *status = bad != 0 ? bad : over ? CRAMGPU_E_INSUFFICIENT : 0;
}

__global__ __launch_bounds__(256) void lpt_keys_kernel(const cramgpu_block* desc, int32_t n, uint32_t* key, int32_t* val) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int32_t len = desc[i].in_len;
        key[i] = 0x7FFFFFFFu - (uint32_t)(len > 0 ? len : 0);  // ascending keys = longest first; the sort is stable
        val[i] = i;
    }
}

constexpr size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

hipError_t launch_series_blocks(const int64_t* d_series_off, const int32_t* d_series_len, int32_t n_cont, const SeriesPick& pick,
                                int64_t in_shift, int64_t out_cap, const int32_t* d_w_status, cramgpu_block* d_blocks,
                                int32_t* d_status, hipStream_t s) {
    hipLaunchKernelGGL(series_blocks_kernel, dim3(1), dim3(1024), 0, s, d_series_off, d_series_len, n_cont, pick, in_shift, out_cap,
                       d_w_status, d_blocks, d_status);
    return hipGetLastError();
}

// workspace of launch_lpt_sort: keys in / out, indices in, the radix sort's own
size_t lpt_sort_bytes(int32_t n) {
    size_t tmp = 0;
#ifndef HIP_SIM
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const int32_t*)nullptr,
                                       (int32_t*)nullptr, n, 0, 32, (hipStream_t)0);
#endif
    return 3 * align256((size_t)n * 4) + align256(tmp) + 256;
}

// longest-processing-time order of any number of blocks whose descriptors lie on the device (the host path sorts where it
// sees them, lpt_perm_kernel does up to 4096 in LDS): a stable radix sort of (0x7FFFFFFF - length, index)
hipError_t launch_lpt_sort(const cramgpu_block* d_desc, int32_t n, int32_t* d_perm, void* ws, size_t ws_bytes, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (ws_bytes < lpt_sort_bytes(n)) return hipErrorInvalidValue;
#ifdef HIP_SIM
    (void)ws;
    for (int32_t i = 0; i < n; i++) d_perm[i] = i;
    std::stable_sort(d_perm, d_perm + n, [&](int32_t x, int32_t y) {
        return std::max(d_desc[x].in_len, 0) > std::max(d_desc[y].in_len, 0);
    });
    return hipSuccess;
#else
    uint8_t* p = reinterpret_cast<uint8_t*>(ws);
    const size_t seg = align256((size_t)n * 4);
    uint32_t* key_in = reinterpret_cast<uint32_t*>(p);
    uint32_t* key_out = reinterpret_cast<uint32_t*>(p + seg);
    int32_t* val_in = reinterpret_cast<int32_t*>(p + 2 * seg);
    void* tmp = p + 3 * seg;
    size_t tmp_bytes = ws_bytes - 3 * seg;
    hipLaunchKernelGGL(lpt_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_desc, n, key_in, val_in);
    return hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, key_in, key_out, val_in, d_perm, n, 0, 32, s);
#endif
}
