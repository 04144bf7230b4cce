// Packs the right-aligned streams the encoder leaves in its slots into one dense arena (HBM-bound copy).
// No reference counterpart beyond ExternalCompression#toByteArray L146-150 (the per-block exact-size copy).
#include "internal.h"

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
