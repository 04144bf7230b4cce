// Timing of dependent instruction SEQUENCES of the rANS step on one wave (gfx950): which instruction pairs stall.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

#define SEQ(ID, N, TEXT) if (V == ID) { asm volatile(REP16(TEXT) : "+v"(x), "+v"(t), "+v"(m), "+v"(q4), "+v"(q2) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(adr), "s"(ex) : "vcc", "memory"); }

template <int V>
__global__ __launch_bounds__(64) void k(uint32_t* out, long long* cyc, int iters, uint32_t seed) {
    __shared__ uint32_t slog[1024];
    uint32_t x = (1u << 23) + threadIdx.x, t = 0, m = 0;
    const uint32_t a = 900u << 19, b = 0x91A2B3C5u, c = 77, d = (4096 - 900) | (9u << 24);
    const uint32_t adr = (uint32_t)(uintptr_t)&slog[threadIdx.x];
    const unsigned long long ex = __builtin_amdgcn_read_exec();
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    v4u q4 = {seed, 1, 2, 3};
    v2u q2 = {seed, 5};
    const bool few = seed == 777u;  // run the loop on 4 lanes only
    (void)seed;
    if (few && threadIdx.x >= 4) return;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        // %0 x %1 t %2 m %3 q4 %4 q2 %5 a(xmax) %6 b(rcp) %7 c(bias) %8 d(cmpl|sh<<24) %9 adr %10 exec
        SEQ(0, 4, "v_mul_hi_u32 %2, %0, %6\n v_lshrrev_b32_sdwa %2, %8, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n v_mul_u32_u24 %2, %2, %8\n v_add3_u32 %0, %0, %7, %2\n")
        SEQ(1, 4, "v_mul_hi_u32 %2, %0, %6\n v_lshrrev_b32 %2, 9, %2\n v_mul_u32_u24 %2, %2, %8\n v_add3_u32 %0, %0, %7, %2\n")
        SEQ(2, 5, "v_mul_hi_u32 %2, %0, %6\n v_lshrrev_b32 %2, 9, %2\n v_mul_u32_u24 %2, %2, %8\n v_add_u32 %0, %0, %7\n v_add_u32 %0, %0, %2\n")
        SEQ(3, 4, "v_cmp_lt_u32 vcc, %0, %5\n v_lshrrev_b32 %1, 8, %0\n s_nop 0\n v_cndmask_b32 %0, %1, %0, vcc\n")
        SEQ(4, 3, "v_cmpx_le_u32 %5, %0\n v_lshrrev_b32 %0, 8, %0\n s_mov_b64 exec, %10\n")
        SEQ(5, 2, "v_mul_hi_u32 %2, %0, %6\n v_add_u32 %0, %0, %2\n")
        SEQ(6, 2, "v_mul_u32_u24 %2, %0, %8\n v_add_u32 %0, %0, %2\n")
        SEQ(7, 2, "v_lshrrev_b32_sdwa %2, %8, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n v_add_u32 %0, %0, %2\n")
        SEQ(8, 2, "v_add3_u32 %0, %0, %7, %2\n v_add_u32 %0, %0, %2\n")
        SEQ(9, 3, "v_mul_hi_u32 %2, %0, %6\n v_lshrrev_b32 %2, 9, %2\n v_mad_u32_u24 %0, %2, %8, %0\n")
        SEQ(10, 4, "v_cmp_lt_u32 vcc, %0, %5\n v_lshrrev_b32 %1, 8, %0\n ds_write_b32 %9, %0\n v_cndmask_b32 %0, %1, %0, vcc\n")
        SEQ(11, 3, "v_sub_u32 %1, %0, %5 clamp\n v_min_u32 %1, 8, %1\n v_lshrrev_b32 %0, %1, %0\n")
        SEQ(12, 2, "v_mul_hi_u32 %2, %0, %6\n v_mul_hi_u32 %0, %2, %6\n")
        SEQ(13, 2, "v_lshrrev_b32 %2, 9, %0\n v_lshrrev_b32 %0, 1, %2\n")
        SEQ(14, 3, "v_mul_hi_u32 %2, %0, %6\n v_lshrrev_b32 %2, 9, %2\n v_add_u32 %0, %0, %2\n")
        SEQ(15, 3, "v_cmp_lt_u32 vcc, %0, %5\n s_nop 1\n v_cndmask_b32 %0, %1, %0, vcc\n")
        SEQ(17, 5, "ds_read_b128 %3, %9\n v_add_u32 %0, %0, %5\n v_add_u32 %0, %0, %5\n v_add_u32 %0, %0, %5\n v_add_u32 %0, %0, %5\n")
        SEQ(18, 5, "ds_read_b32 %1, %9\n v_add_u32 %0, %0, %5\n v_add_u32 %0, %0, %5\n v_add_u32 %0, %0, %5\n v_add_u32 %0, %0, %5\n")
        SEQ(19, 5, "ds_write_b32 %9, %0\n v_add_u32 %0, %0, %5\n v_add_u32 %0, %0, %5\n v_add_u32 %0, %0, %5\n v_add_u32 %0, %0, %5\n")
        SEQ(20, 5, "ds_write_b128 %9, %3\n v_add_u32 %0, %0, %5\n v_add_u32 %0, %0, %5\n v_add_u32 %0, %0, %5\n v_add_u32 %0, %0, %5\n")
        SEQ(21, 1, "ds_read_b128 %3, %9\n")
        SEQ(22, 9, "ds_read_b64 %4, %9\n v_add_u32 %0, %0, %5\n v_add_u32 %0, %0, %5\n v_add_u32 %0, %0, %5\n v_add_u32 %0, %0, %5\n v_add_u32 %0, %0, %5\n v_add_u32 %0, %0, %5\n v_add_u32 %0, %0, %5\n v_add_u32 %0, %0, %5\n")
        SEQ(23, 3, "v_sub_u32 %1, %5, %0\n v_lshrrev_b32 %1, 31, %1\n v_alignbyte_b32 %0, 0, %0, %1\n")
        SEQ(24, 7, "v_sub_u32 %1, %5, %0\n v_lshrrev_b32 %1, 31, %1\n v_alignbyte_b32 %0, 0, %0, %1\n v_mul_hi_u32 %2, %0, %6\n v_lshrrev_b32_sdwa %2, %8, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n v_mul_u32_u24 %2, %2, %8\n v_add3_u32 %0, %0, %7, %2\n")
        SEQ(16, 4, "v_cmp_lt_u32 vcc, %0, %5\n v_mul_hi_u32 %2, %0, %6\n v_lshrrev_b32 %1, 8, %0\n v_cndmask_b32 %0, %1, %0, vcc\n")
    }
    long long t1 = clock64();
    out[threadIdx.x] = x ^ t ^ m ^ q4.x ^ q2.y;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int V>
void run(const char* name, int n, uint32_t* d_out, long long* d_cyc, uint32_t seed = 12345u) {
    const int iters = 2000;
    for (int r = 0; r < 2; r++) {
        hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, d_out, d_cyc, iters, seed);
        CHECK(hipDeviceSynchronize());
    }
    long long c;
    CHECK(hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost));
    printf("%-58s %7.2f cycles per sequence of %d  (%.2f / instr)\n", name, (double)c / iters / 16.0, n, (double)c / iters / 16.0 / n);
}

int main() {
    uint32_t* d_out;
    long long* d_cyc;
    CHECK(hipMalloc(&d_out, 4096));
    CHECK(hipMalloc(&d_cyc, 64));
    run<0>("mulhi, lshr_sdwa, mul24, add3", 4, d_out, d_cyc);
    run<1>("mulhi, lshr(imm), mul24, add3", 4, d_out, d_cyc);
    run<2>("mulhi, lshr(imm), mul24, add, add", 5, d_out, d_cyc);
    run<9>("mulhi, lshr(imm), mad24", 3, d_out, d_cyc);
    run<14>("mulhi, lshr(imm), add", 3, d_out, d_cyc);
    run<3>("cmp, lshr, s_nop 0, cndmask", 4, d_out, d_cyc);
    run<10>("cmp, lshr, ds_write, cndmask", 4, d_out, d_cyc);
    run<15>("cmp, s_nop 1, cndmask", 3, d_out, d_cyc);
    run<16>("cmp, mulhi(indep), lshr, cndmask", 4, d_out, d_cyc);
    run<4>("cmpx, lshr(masked), s_mov exec", 3, d_out, d_cyc);
    run<11>("sub clamp, min, lshr(var)", 3, d_out, d_cyc);
    run<5>("mulhi, add", 2, d_out, d_cyc);
    run<12>("mulhi, mulhi", 2, d_out, d_cyc);
    run<6>("mul24, add", 2, d_out, d_cyc);
    run<7>("lshr_sdwa, add", 2, d_out, d_cyc);
    run<8>("add3, add", 2, d_out, d_cyc);
    run<13>("lshr, lshr", 2, d_out, d_cyc);
    run<23>("sub, lshr31, alignbyte", 3, d_out, d_cyc);
    run<24>("full step: sub, lshr31, alignbyte, mulhi, sdwa, mul24, add3", 7, d_out, d_cyc);
    run<17>("ds_read_b128 + 4 add (64 lanes)", 5, d_out, d_cyc);
    run<17>("ds_read_b128 + 4 add (4 lanes)", 5, d_out, d_cyc, 777u);
    run<18>("ds_read_b32 + 4 add (4 lanes)", 5, d_out, d_cyc, 777u);
    run<22>("ds_read_b64 + 8 add (4 lanes)", 9, d_out, d_cyc, 777u);
    run<19>("ds_write_b32 + 4 add (4 lanes)", 5, d_out, d_cyc, 777u);
    run<20>("ds_write_b128 + 4 add (4 lanes)", 5, d_out, d_cyc, 777u);
    run<21>("ds_read_b128 back to back (4 lanes)", 1, d_out, d_cyc, 777u);
    run<21>("ds_read_b128 back to back (64 lanes)", 1, d_out, d_cyc);
    return 0;
}
