// Instruction timing probes for one wave on gfx950 (dependent vs independent issue), used to model the rANS chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

template <int V>
__global__ __launch_bounds__(64) void k(uint32_t* out, long long* cyc, int iters, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1, c = seed | 5, d = seed + 7, e = 9, f = 11, g = 13, h = 17;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        if (V == 0) asm volatile(REP64("v_add_u32 %0, %0, %1\n") : "+v"(a) : "v"(b));                       // dependent add
        if (V == 1) asm volatile(REP16("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n") : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));  // 4 independent chains
        if (V == 2) asm volatile(REP64("v_mul_hi_u32 %0, %0, %1\n") : "+v"(a) : "v"(b));                    // dependent mulhi
        if (V == 3) asm volatile(REP16("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4\n") : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));
        if (V == 4) asm volatile(REP64("v_mul_u32_u24 %0, %0, %1\n") : "+v"(a) : "v"(b));                   // dependent mul24
        if (V == 5) asm volatile(REP64("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %2, %0, vcc\n") : "+v"(a) : "v"(b), "v"(c) : "vcc");  // cmp -> cndmask dependent (assembler does not add nops)
        if (V == 6) asm volatile(REP64("v_cmp_lt_u32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %2, %0, vcc\n") : "+v"(a) : "v"(b), "v"(c) : "vcc");
        if (V == 7) asm volatile(REP64("v_lshrrev_b32 %0, 1, %0\n") : "+v"(a));                              // dependent shift
        if (V == 8) asm volatile(REP64("v_add3_u32 %0, %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c));
        if (V == 9) asm volatile(REP64("v_mul_hi_u32 %0, %0, %1\n v_add_u32 %2, %2, %1\n v_add_u32 %3, %3, %1\n v_add_u32 %4, %4, %1\n") : "+v"(a) : "v"(b), "v"(c), "v"(d), "v"(e));  // mulhi + 3 independent adds
        if (V == 10) asm volatile(REP64("v_cvt_f64_u32 %1, %0\n v_mul_f64 %1, %1, %2\n v_cvt_u32_f64 %0, %1\n") : "+v"(a) : "v"((double)1.0), "v"((double)0.999));
        if (V == 11) asm volatile(REP64("v_cvt_f32_u32 %1, %0\n v_mul_f32 %1, %1, %2\n v_cvt_u32_f32 %0, %1\n") : "+v"(a) : "v"(1.0f), "v"(0.999f));
        if (V == 12) asm volatile(REP64("s_nop 0\n"));
        if (V == 13) asm volatile(REP64("v_sub_u32 %0, %0, %1 clamp\n") : "+v"(a) : "v"(b));
        if (V == 14) asm volatile(REP64("v_mad_u32_u24 %0, %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c));
        if (V == 15) asm volatile(REP64("v_add_u32 %0, %0, %1\n s_nop 0\n") : "+v"(a) : "v"(b));             // dependent add + nop filler
        if (V == 16) asm volatile(REP64("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2\n") : "+v"(a), "+v"(c) : "v"(b));  // 2 chains
        if (V == 17) asm volatile(REP64("v_mad_u64_u32 %0, vcc, %1, %2, %0\n") : "+v"(*(unsigned long long*)&g) : "v"(a), "v"(b) : "vcc");
        if (V == 18) asm volatile(REP64("v_min_u32 %0, %0, %1\n") : "+v"(a) : "v"(b));
        if (V == 19) asm volatile(REP64("v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n") : "+v"(a) : "v"(b));
    }
    long long t1 = clock64();
    out[threadIdx.x] = a ^ c ^ d ^ e ^ f ^ g ^ h;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int V>
void run(const char* name, int per_iter, uint32_t* d_out, long long* d_cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, d_out, d_cyc, iters, 12345u);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, d_out, d_cyc, iters, 12345u);
    CHECK(hipDeviceSynchronize());
    long long c;
    CHECK(hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost));
    printf("%-44s %7.2f cycles per group of %d instr\n", name, (double)c / iters / 64.0, per_iter);
}

int main() {
    uint32_t* d_out;
    long long* d_cyc;
    CHECK(hipMalloc(&d_out, 4096));
    CHECK(hipMalloc(&d_cyc, 64));
    run<0>("dependent v_add_u32", 1, d_out, d_cyc);
    run<1>("4 independent v_add_u32 chains (per 4)", 4, d_out, d_cyc);
    run<16>("2 independent v_add_u32 chains (per 2)", 2, d_out, d_cyc);
    run<15>("dependent v_add_u32 + s_nop 0", 2, d_out, d_cyc);
    run<2>("dependent v_mul_hi_u32", 1, d_out, d_cyc);
    run<3>("4 independent v_mul_hi_u32 (per 4)", 4, d_out, d_cyc);
    run<9>("v_mul_hi_u32 dep + 3 indep adds (per 4)", 4, d_out, d_cyc);
    run<4>("dependent v_mul_u32_u24", 1, d_out, d_cyc);
    run<14>("dependent v_mad_u32_u24", 1, d_out, d_cyc);
    run<17>("dependent v_mad_u64_u32", 1, d_out, d_cyc);
    run<5>("v_cmp -> v_cndmask dependent, no nop (per 2)", 2, d_out, d_cyc);
    run<6>("v_cmp -> s_nop 1 -> v_cndmask (per 3)", 3, d_out, d_cyc);
    run<7>("dependent v_lshrrev_b32", 1, d_out, d_cyc);
    run<8>("dependent v_add3_u32", 1, d_out, d_cyc);
    run<13>("dependent v_sub_u32 clamp", 1, d_out, d_cyc);
    run<18>("dependent v_min_u32", 1, d_out, d_cyc);
    run<19>("dependent v_add_u32_dpp row_shr:1", 1, d_out, d_cyc);
    run<10>("cvt_f64_u32 -> mul_f64 -> cvt_u32_f64 (per 3)", 3, d_out, d_cyc);
    run<11>("cvt_f32_u32 -> mul_f32 -> cvt_u32_f32 (per 3)", 3, d_out, d_cyc);
    run<12>("s_nop 0", 1, d_out, d_cyc);
    return 0;
}
