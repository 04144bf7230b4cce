// Does a vector instruction cost less when few lanes are enabled?  Dependent v_add / v_mul_hi chains under EXEC masks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

template <int V>
__global__ __launch_bounds__(64) void k(uint32_t* out, long long* cyc, int iters, int lanes, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1, c = 5, d = 7;
    long long t0 = 0, t1 = 0;
    if ((int)threadIdx.x < lanes) {
        t0 = clock64();
        for (int i = 0; i < iters; i++) {
            if (V == 0) asm volatile(REP64("v_add_u32 %0, %0, %1\n") : "+v"(a) : "v"(b));
            if (V == 1) asm volatile(REP64("v_mul_hi_u32 %0, %0, %1\n") : "+v"(a) : "v"(b));
            if (V == 2) asm volatile(REP16("v_add_u32 %0, %0, %3\n v_add_u32 %1, %1, %3\n v_add_u32 %2, %2, %3\n v_add_u32 %0, %0, %1\n") : "+v"(a), "+v"(c), "+v"(d) : "v"(b));
        }
        t1 = clock64();
    }
    out[threadIdx.x] = a ^ c ^ d;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int V>
void run(const char* name, uint32_t* d_out, long long* d_cyc) {
    const int iters = 2000;
    const int lanes[] = {64, 33, 32, 17, 16, 4, 1};
    printf("%-40s", name);
    for (int l : lanes) {
        for (int r = 0; r < 2; r++) {
            hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, d_out, d_cyc, iters, l, 12345u);
            CHECK(hipDeviceSynchronize());
        }
        long long c;
        CHECK(hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost));
        printf("  %2d lanes %5.2f", l, (double)c / iters / 64.0);
    }
    printf("   cycles per instruction\n");
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    uint32_t* d_out;
    long long* d_cyc;
    CHECK(hipMalloc(&d_out, 4096));
    CHECK(hipMalloc(&d_cyc, 64));
    run<0>("dependent v_add_u32", d_out, d_cyc);
    run<1>("dependent v_mul_hi_u32", d_out, d_cyc);
    run<2>("3 independent + 1 dependent v_add", d_out, d_cyc);
    return 0;
}
