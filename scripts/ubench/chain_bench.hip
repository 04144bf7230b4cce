// Micro-benchmark of the rANS encoder state chain on gfx950: cycles per coding step of one wave, for several
// formulations of Encoding#RansEncPutSymbol.  Not part of the library; used to choose the kernel's inner loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

struct Sym { uint32_t xmax, xmax8, rcp, bias, cmpl, sh; double rcpd; };

constexpr int kSteps = 16;

// variant 0: the library's current chain (two compares, selects, mulhi, mul24)
template <int V>
__global__ __launch_bounds__(64) void chain_kernel(const Sym* syms, int nsym, int nbatch, uint32_t* out, long long* cyc) {
    __shared__ uint4 sA[256];
    __shared__ uint2 sB[256];
    __shared__ double sD[256];
    const int l = threadIdx.x;
    for (int i = l; i < 256; i += 64) {
        Sym s = syms[i % nsym];
        sA[i] = make_uint4(s.xmax, s.xmax8, s.rcp, s.bias);
        sB[i] = make_uint2(s.cmpl, s.sh);
        sD[i] = s.rcpd;
    }
    __syncthreads();
    uint32_t x = 1u << 23;
    uint32_t acc = 0;
    const int sl = l & 3;
    long long t0 = clock64();
    for (int b = 0; b < nbatch; b++) {
        const int base = (b * 64) & 255;
        uint4 sa[kSteps];
        uint2 sb[kSteps];
        double sd[kSteps];
#pragma unroll
        for (int tt = 0; tt < kSteps; tt++) {
            sa[tt] = sA[(base + tt * 4 + sl) & 255];
            sb[tt] = sB[(base + tt * 4 + sl) & 255];
            if (V == 3) sd[tt] = sD[(base + tt * 4 + sl) & 255];
        }
#pragma unroll
        for (int tt = 0; tt < kSteps; tt++) {
            acc ^= x;
            if (V == 0) {
                const bool r1 = x >= sa[tt].x, r2 = x >= sa[tt].y;
                const uint32_t sh = r2 ? 16u : (r1 ? 8u : 0u);
                const uint32_t xr = x >> sh;
                const uint32_t qq = __umulhi(xr, sa[tt].z) >> sb[tt].y;
                x = __umul24(qq, sb[tt].x) + (xr + sa[tt].w);
            } else if (V == 1) {  // mulhi on the unshifted state, in parallel with the compares
                const uint32_t m = __umulhi(x, sa[tt].z);
                const bool r1 = x >= sa[tt].x, r2 = x >= sa[tt].y;
                const uint32_t s = r2 ? 16u : (r1 ? 8u : 0u);
                const uint32_t xr = x >> s;
                const uint32_t qq = m >> (sb[tt].y + s);
                x = __umul24(qq, sb[tt].x) + (xr + sa[tt].w);
            } else if (V == 2) {  // single compare (all freq >= 16 in the batch)
                const bool r1 = x >= sa[tt].x;
                const uint32_t xr = r1 ? (x >> 8) : x;
                const uint32_t qq = __umulhi(xr, sa[tt].z) >> sb[tt].y;
                x = __umul24(qq, sb[tt].x) + (xr + sa[tt].w);
            } else if (V == 3) {  // double reciprocal, single compare
                const bool r1 = x >= sa[tt].x;
                const uint32_t xr = r1 ? (x >> 8) : x;
                const uint32_t qq = (uint32_t)((double)xr * sd[tt]);
                x = __umul24(qq, sb[tt].x) + (xr + sa[tt].w);
            } else if (V == 4) {  // single compare + mulhi on unshifted state
                const uint32_t m = __umulhi(x, sa[tt].z);
                const bool r1 = x >= sa[tt].x;
                const uint32_t xr = r1 ? (x >> 8) : x;
                const uint32_t qq = m >> (r1 ? sb[tt].y + 8 : sb[tt].y);
                x = __umul24(qq, sb[tt].x) + (xr + sa[tt].w);
            } else if (V == 5) {  // float32 estimate + correction, single compare
                const bool r1 = x >= sa[tt].x;
                const uint32_t xr = r1 ? (x >> 8) : x;
                uint32_t qq = (uint32_t)((float)xr * __uint_as_float(sa[tt].z));
                x = __umul24(qq, sb[tt].x) + (xr + sa[tt].w);
            }
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * 64 + l] = x ^ acc;
    if (l == 0) cyc[blockIdx.x] = t1 - t0;
}


#define STEP_W(K) \
    "v_cmp_lt_u32 vcc, %[x], %[a" #K "]\n\t" \
    "v_lshrrev_b32 %[t], 8, %[x]\n\t" \
    "ds_write_b32 %[adr], %[x] offset:%c[o" #K "]\n\t" \
    "v_cndmask_b32 %[x], %[t], %[x], vcc\n\t" \
    "v_mul_hi_u32 %[m], %[x], %[b" #K "]\n\t" \
    "v_lshrrev_b32_sdwa %[m], %[d" #K "], %[m] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t" \
    "v_mul_u32_u24 %[m], %[m], %[d" #K "]\n\t" \
    "v_add3_u32 %[x], %[x], %[c" #K "], %[m]\n\t"
#define STEP_N(K) \
    "v_cmp_lt_u32 vcc, %[x], %[a" #K "]\n\t" \
    "v_lshrrev_b32 %[t], 8, %[x]\n\t" \
    "s_nop 0\n\t" \
    "v_cndmask_b32 %[x], %[t], %[x], vcc\n\t" \
    "v_mul_hi_u32 %[m], %[x], %[b" #K "]\n\t" \
    "v_lshrrev_b32_sdwa %[m], %[d" #K "], %[m] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t" \
    "v_mul_u32_u24 %[m], %[m], %[d" #K "]\n\t" \
    "v_add3_u32 %[x], %[x], %[c" #K "], %[m]\n\t"
#define STEP_X(K) \
    "v_cmpx_le_u32 %[a" #K "], %[x]\n\t" \
    "v_lshrrev_b32 %[x], 8, %[x]\n\t" \
    "s_mov_b64 exec, %[ex]\n\t" \
    "v_mul_hi_u32 %[m], %[x], %[b" #K "]\n\t" \
    "v_lshrrev_b32_sdwa %[m], %[d" #K "], %[m] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t" \
    "v_mul_u32_u24 %[m], %[m], %[d" #K "]\n\t" \
    "v_add3_u32 %[x], %[x], %[c" #K "], %[m]\n\t"

// asm variants: params as uint4 {xmax, rcp, bias, cmpl | sh << 24}.  MODE 0: with log write, 1: s_nop filler, 2: cmpx
// LANES4: run under lane < 4.  PARTNER: a second wave of the workgroup spins on VALU work meanwhile.
template <int MODE, bool LANES4, int PARTNER>
__global__ __launch_bounds__(128) void chain_asm_kernel(const Sym* syms, int nsym, int nbatch, uint32_t* out, long long* cyc) {
    __shared__ uint4 sP[256];
    __shared__ uint32_t slog[64 * 16];
    __shared__ int done;
    const int l = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x == 0) done = 0;
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        Sym s = syms[i % nsym];
        sP[i] = make_uint4(s.xmax, s.rcp, s.bias, s.cmpl | (s.sh << 24));
    }
    __syncthreads();
    if (w == 1) {
        if (PARTNER == 0) return;
        uint32_t a = l, b = 3;
        if (PARTNER == 1) {  // VALU spinner
            while (__builtin_amdgcn_readfirstlane(*(volatile int*)&done) == 0) {
                asm volatile(REP16("v_add_u32 %0, %0, %1\n") : "+v"(a) : "v"(b));
            }
        } else {  // LDS + VALU mix
            while (__builtin_amdgcn_readfirstlane(*(volatile int*)&done) == 0) {
                asm volatile(REP16("v_add_u32 %0, %0, %1\n ds_write_b32 %2, %0 offset:8192\n") : "+v"(a) : "v"(b), "v"(l * 4) : "memory");
            }
        }
        out[4096 + l] = a;
        return;
    }
    uint32_t x = 1u << 23;
    const int sl = l & 3;
    long long t0 = clock64();
    if (!LANES4 || l < 4) {
        const unsigned long long ex = __builtin_amdgcn_read_exec();
        for (int b = 0; b < nbatch; b++) {
            const int base = (b * 64) & 255;
            uint4 P[16];
#pragma unroll
            for (int tt = 0; tt < 16; tt++) P[tt] = sP[(base + tt * 4 + sl) & 255];
            uint32_t adr = (uint32_t)(uintptr_t)&slog[l * 16];
#pragma unroll
            for (int g = 0; g < 16; g += 4) {
                uint32_t t, m;
                if (MODE == 0)
                    asm volatile(STEP_W(0) STEP_W(1) STEP_W(2) STEP_W(3)
                                 : [x] "+v"(x), [t] "=&v"(t), [m] "=&v"(m)
                                 : [adr] "v"(adr), [o0] "n"(0), [o1] "n"(4), [o2] "n"(8), [o3] "n"(12),
                                   [a0] "v"(P[g].x), [b0] "v"(P[g].y), [c0] "v"(P[g].z), [d0] "v"(P[g].w),
                                   [a1] "v"(P[g + 1].x), [b1] "v"(P[g + 1].y), [c1] "v"(P[g + 1].z), [d1] "v"(P[g + 1].w),
                                   [a2] "v"(P[g + 2].x), [b2] "v"(P[g + 2].y), [c2] "v"(P[g + 2].z), [d2] "v"(P[g + 2].w),
                                   [a3] "v"(P[g + 3].x), [b3] "v"(P[g + 3].y), [c3] "v"(P[g + 3].z), [d3] "v"(P[g + 3].w)
                                 : "vcc", "memory");
                else if (MODE == 1)
                    asm volatile(STEP_N(0) STEP_N(1) STEP_N(2) STEP_N(3)
                                 : [x] "+v"(x), [t] "=&v"(t), [m] "=&v"(m)
                                 : [a0] "v"(P[g].x), [b0] "v"(P[g].y), [c0] "v"(P[g].z), [d0] "v"(P[g].w),
                                   [a1] "v"(P[g + 1].x), [b1] "v"(P[g + 1].y), [c1] "v"(P[g + 1].z), [d1] "v"(P[g + 1].w),
                                   [a2] "v"(P[g + 2].x), [b2] "v"(P[g + 2].y), [c2] "v"(P[g + 2].z), [d2] "v"(P[g + 2].w),
                                   [a3] "v"(P[g + 3].x), [b3] "v"(P[g + 3].y), [c3] "v"(P[g + 3].z), [d3] "v"(P[g + 3].w)
                                 : "vcc", "memory");
                else
                    asm volatile(STEP_X(0) STEP_X(1) STEP_X(2) STEP_X(3)
                                 : [x] "+v"(x), [m] "=&v"(m)
                                 : [ex] "s"(ex),
                                   [a0] "v"(P[g].x), [b0] "v"(P[g].y), [c0] "v"(P[g].z), [d0] "v"(P[g].w),
                                   [a1] "v"(P[g + 1].x), [b1] "v"(P[g + 1].y), [c1] "v"(P[g + 1].z), [d1] "v"(P[g + 1].w),
                                   [a2] "v"(P[g + 2].x), [b2] "v"(P[g + 2].y), [c2] "v"(P[g + 2].z), [d2] "v"(P[g + 2].w),
                                   [a3] "v"(P[g + 3].x), [b3] "v"(P[g + 3].y), [c3] "v"(P[g + 3].z), [d3] "v"(P[g + 3].w)
                                 : "vcc", "memory");
                adr += 16;
            }
        }
    }
    long long t1 = clock64();
    if (l == 0) *(volatile int*)&done = 1;
    out[blockIdx.x * 64 + l] = x;
    if (l == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, bool LANES4, int PARTNER>
void run_asm(const char* name, const Sym* d_syms, int nsym, int grid, int nbatch, uint32_t* d_out, long long* d_cyc) {
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL((chain_asm_kernel<MODE, LANES4, PARTNER>), dim3(grid), dim3(128), 0, 0, d_syms, nsym, nbatch, d_out, d_cyc);
        CHECK(hipDeviceSynchronize());
    }
    std::vector<long long> cyc(grid);
    CHECK(hipMemcpy(cyc.data(), d_cyc, grid * sizeof(long long), hipMemcpyDeviceToHost));
    double avg = 0;
    for (auto c : cyc) avg += (double)c;
    avg /= grid;
    printf("%-44s grid %5d  clock64/step %7.2f\n", name, grid, avg / ((double)nbatch * 16));
}

template <int V>
void run(const char* name, const Sym* d_syms, int nsym, int grid, int nbatch, uint32_t* d_out, long long* d_cyc) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(chain_kernel<V>, dim3(grid), dim3(64), 0, 0, d_syms, nsym, nbatch, d_out, d_cyc);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(chain_kernel<V>, dim3(grid), dim3(64), 0, 0, d_syms, nsym, nbatch, d_out, d_cyc);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> cyc(grid);
    CHECK(hipMemcpy(cyc.data(), d_cyc, grid * sizeof(long long), hipMemcpyDeviceToHost));
    double avg = 0;
    for (auto c : cyc) avg += (double)c;
    avg /= grid;
    const double steps = (double)nbatch * kSteps;
    printf("%-28s grid %5d  %8.3f ms  clock64/step %7.2f  ns/step %7.2f\n", name, grid, ms, avg / steps, ms * 1e6 / steps);
}

int main() {
    std::vector<Sym> syms;
    srand(1);
    for (int i = 0; i < 64; i++) {
        uint32_t f = 16 + (rand() % 2000);
        Sym s;
        s.xmax = f << 19;
        s.xmax8 = 0xFFFFFFFFu;
        uint32_t shift = 0;
        while ((unsigned long long)f > (1ull << shift)) shift++;
        s.rcp = (uint32_t)(((1ull << (shift + 31)) + f - 1) / f);
        s.sh = shift - 1;
        s.bias = rand() % (4096 - f);
        s.cmpl = 4096 - f;
        s.rcpd = 1.0 / f * (1 + 1e-15);
        syms.push_back(s);
    }
    Sym* d_syms;
    CHECK(hipMalloc(&d_syms, syms.size() * sizeof(Sym)));
    CHECK(hipMemcpy(d_syms, syms.data(), syms.size() * sizeof(Sym), hipMemcpyHostToDevice));
    uint32_t* d_out;
    long long* d_cyc;
    CHECK(hipMalloc(&d_out, 8192 * 64 * 4));
    CHECK(hipMalloc(&d_cyc, 8192 * 8));
    const int nbatch = 20000;
    for (int grid : {1, 1024}) {
        run_asm<0, false, 0>("asm log-write, 64 lanes", d_syms, 64, grid, nbatch, d_out, d_cyc);
        run_asm<0, true, 0>("asm log-write, 4 lanes", d_syms, 64, grid, nbatch, d_out, d_cyc);
        run_asm<1, true, 0>("asm s_nop filler, 4 lanes", d_syms, 64, grid, nbatch, d_out, d_cyc);
        run_asm<2, true, 0>("asm cmpx, 4 lanes", d_syms, 64, grid, nbatch, d_out, d_cyc);
        run_asm<0, true, 1>("asm log-write, 4 lanes + VALU partner", d_syms, 64, grid, nbatch, d_out, d_cyc);
        run_asm<0, true, 2>("asm log-write, 4 lanes + LDS partner", d_syms, 64, grid, nbatch, d_out, d_cyc);
    }
    for (int grid : {1}) {
        run<0>("v0 current", d_syms, 64, grid, nbatch, d_out, d_cyc);
        run<1>("v1 mulhi unshifted", d_syms, 64, grid, nbatch, d_out, d_cyc);
        run<2>("v2 one compare", d_syms, 64, grid, nbatch, d_out, d_cyc);
        run<3>("v3 f64 rcp one compare", d_syms, 64, grid, nbatch, d_out, d_cyc);
        run<4>("v4 one cmp unshifted mulhi", d_syms, 64, grid, nbatch, d_out, d_cyc);
        run<5>("v5 f32 estimate (inexact)", d_syms, 64, grid, nbatch, d_out, d_cyc);
    }
    return 0;
}
