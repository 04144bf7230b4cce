// Latency probes for the rANS decoder's lookup chain on gfx950: dependent LDS reads, ds_bpermute, readlane round trips,
// dependent global (L2) loads.  One wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

template <int V>
__global__ __launch_bounds__(64) void k(uint32_t* out, long long* cyc, const uint32_t* g, int iters) {
    __shared__ uint32_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = ((i * 97 + 13) & 4095) * 4;  // byte offsets, a permutation walk
    __syncthreads();
    uint32_t a = threadIdx.x * 4, b = 0;
    unsigned long long w = threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        if (V == 0) asm volatile(REP16("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n") : "+v"(a));
        if (V == 1) {
#pragma unroll
            for (int u = 0; u < 16; u++) { unsigned long long t = *reinterpret_cast<volatile unsigned long long*>(reinterpret_cast<char*>(lds) + (a & 0x3ff8)); a = (uint32_t)t; b ^= (uint32_t)(t >> 32); }
        }
        if (V == 2) asm volatile(REP16("ds_read_u8 %0, %0\n s_waitcnt lgkmcnt(0)\n v_lshlrev_b32 %0, 2, %0\n") : "+v"(a));
        if (V == 3) asm volatile(REP16("ds_bpermute_b32 %0, %0, %0\n s_waitcnt lgkmcnt(0)\n") : "+v"(a));
        if (V == 4) asm volatile(REP16("v_readlane_b32 s20, %0, 3\n v_add_u32 %0, s20, %0\n") : "+v"(a) : : "s20");
        if (V == 5) asm volatile(REP16("global_load_dword %0, %0, %1\n s_waitcnt vmcnt(0)\n") : "+v"(a) : "s"(g));
        if (V == 6) asm volatile(REP16("v_cmp_lt_u32 vcc, %0, %1\n s_bcnt1_i32_b64 s20, vcc\n v_add_u32 %0, s20, %0\n") : "+v"(a) : "v"(b) : "s20", "vcc", "scc");
        if (V == 7) asm volatile(REP16("v_readfirstlane_b32 s20, %0\n s_add_u32 s20, s20, 4\n v_mov_b32 %0, s20\n") : "+v"(a) : : "s20", "scc");
        if (V == 8) asm volatile(REP16("ds_read_b32 %0, %0\n ds_read_b32 %1, %2\n s_waitcnt lgkmcnt(1)\n") : "+v"(a), "+v"(b) : "v"(threadIdx.x * 4));
        if (V == 9) asm volatile(REP16("v_or_b32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n") : "+v"(a));
        if (V == 10) asm volatile(REP16("v_lshlrev_b64 %0, 8, %0\n") : "+v"(w));
        if (V == 11) asm volatile(REP16("v_perm_b32 %0, %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(0x03020100u));
        if (V == 12) asm volatile(REP16("v_mbcnt_lo_u32_b32 %0, %1, %0\n") : "+v"(a) : "s"(0x55u));
        if (V == 13) asm volatile(REP16("s_add_u32 s20, s20, 4\n") : : : "s20", "scc");
        if (V == 14) asm volatile(REP16("v_cmp_lt_u32 s[20:21], %0, %1\n v_cndmask_b32 %0, %0, %1, s[20:21]\n") : "+v"(a) : "v"(b) : "s20", "s21");
        if (V == 15) asm volatile(REP16("v_mov_b32_dpp %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n") : "+v"(a));
    }
    long long t1 = clock64();
    out[threadIdx.x] = a ^ b ^ (uint32_t)w;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int V>
void run(const char* name, uint32_t* d_out, long long* d_cyc, const uint32_t* g) {
    const int iters = 2000;
    printf("%-60s ", name);
    for (int r = 0; r < 2; r++) {
        hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, d_out, d_cyc, g, iters);
        CHECK(hipDeviceSynchronize());
    }
    long long c;
    CHECK(hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost));
    printf("%7.2f cycles per link\n", (double)c / iters / 16.0);
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    uint32_t *d_out, *g;
    long long* d_cyc;
    CHECK(hipMalloc(&d_out, 4096));
    CHECK(hipMalloc(&d_cyc, 64));
    CHECK(hipMalloc(&g, 16384));
    uint32_t h[4096];
    for (int i = 0; i < 4096; i++) h[i] = ((i * 97 + 13) & 4095) * 4;
    CHECK(hipMemcpy(g, h, 16384, hipMemcpyHostToDevice));
    if (only < 0 || only == 0) run<0>("dependent ds_read_b32", d_out, d_cyc, g);
    if (only < 0 || only == 1) run<1>("dependent ds_read_b64 + mov", d_out, d_cyc, g);
    if (only < 0 || only == 2) run<2>("dependent ds_read_u8 + shift", d_out, d_cyc, g);
    if (only < 0 || only == 8) run<8>("dependent ds_read_b32 with a second read in flight", d_out, d_cyc, g);
    if (only < 0 || only == 3) run<3>("dependent ds_bpermute_b32", d_out, d_cyc, g);
    if (only < 0 || only == 4) run<4>("v_readlane -> v_add (sgpr round trip)", d_out, d_cyc, g);
    if (only < 0 || only == 7) run<7>("v_readfirstlane -> s_add -> v_mov", d_out, d_cyc, g);
    if (only < 0 || only == 6) run<6>("v_cmp -> s_bcnt1 -> v_add", d_out, d_cyc, g);
    if (only < 0 || only == 14) run<14>("v_cmp sgpr pair -> v_cndmask", d_out, d_cyc, g);
    if (only < 0 || only == 5) run<5>("dependent global_load_dword (L2 resident)", d_out, d_cyc, g);
    if (only < 0 || only == 9) run<9>("dependent v_or_b32_dpp quad_perm", d_out, d_cyc, g);
    if (only < 0 || only == 15) run<15>("dependent v_mov_b32_dpp row_bcast:15", d_out, d_cyc, g);
    if (only < 0 || only == 10) run<10>("dependent v_lshlrev_b64", d_out, d_cyc, g);
    if (only < 0 || only == 11) run<11>("dependent v_perm_b32", d_out, d_cyc, g);
    if (only < 0 || only == 12) run<12>("dependent v_mbcnt_lo", d_out, d_cyc, g);
    if (only < 0 || only == 13) run<13>("dependent s_add_u32", d_out, d_cyc, g);
    return 0;
}
