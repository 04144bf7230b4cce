// Micro-benchmark of the narrow order-1 decode step (rans_dec.hip, narrow_block) on gfx950: what each part of the step
// costs a lone wave.  Two deterministic workloads: a context whose one symbol has F = 4096 (x never changes: no step
// renormalises) and one whose symbol has F = 16 over a stream of zero bytes (x = 2^23 -> 2^15 -> one byte -> 2^23: every
// state takes one byte every step).  Prints shader clocks per step.  Not part of the library.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

#define MIN_(DST, CTRL) "v_min_u32_dpp %[" DST "], %[v], %[v] " CTRL " row_mask:0xf bank_mask:0xf\n\t"
#define P_TEST "s_waitcnt lgkmcnt(0)\n\t" "v_lshl_add_u32 %[v], %[x], 20, %[e]\n\t"
#define P_TEST_NOWAIT "v_lshl_add_u32 %[v], %[x], 20, %[e]\n\t"
#define P_BFLY(KW, KO)                                                                      \
    "v_lshrrev_b32 %[xs], 12, %[x]\n\t"                                                     \
    MIN_("v", "quad_perm:[1,0,3,2]")                                                        \
    "v_mov_b32_dpp %[" KO "], %[" KW "] row_shr:1 row_mask:0xf bank_mask:0xf\n\t"           \
    MIN_("v", "quad_perm:[2,3,0,1]")                                                        \
    "s_nop 0\n\t"                                                                           \
    MIN_(KW, "row_half_mirror")                                                             \
    "v_and_or_b32 %[t], %[" KW "], 28, %[lb]\n\t"
#define P_NOBFLY(KW, KO)                                                                    \
    "v_lshrrev_b32 %[xs], 12, %[x]\n\t"                                                     \
    "v_mov_b32 %[" KW "], %[v]\n\t"                                                         \
    "v_and_or_b32 %[t], %[" KW "], 28, %[lb]\n\t"
#define P_LDS "ds_read_b32 %[e], %[t]\n\t"
#define P_XUPD(KW)                                                                          \
    "v_lshrrev_b32 %[d], 20, %[" KW "]\n\t"                                                 \
    "v_bfe_u32 %[f], %[" KW "], 7, 13\n\t"                                                  \
    "v_mad_u32_u24 %[x], %[f], %[xs], %[d]\n\t"
#define P_REN(N)                                                                            \
    "v_cndmask_b32_e32 %[sel], %[selno], %[selone], vcc\n\t"                                \
    "s_and_b64 vcc, vcc, %[ktop]\n\t"                                                       \
    "v_mbcnt_lo_u32_b32 %[a], vcc_lo, 0\n\t"                                                \
    "v_mbcnt_hi_u32_b32 %[a], vcc_hi, %[a]\n\t"                                             \
    "s_bcnt1_i32_b64 %[stot], vcc\n\t"                                                      \
    "v_cmp_gt_u32_e32 vcc, 0x8000, %[x]\n\t"                                                \
    "s_cbranch_vccnz .Lend_%=\n\t"                                                          \
    "s_waitcnt vmcnt(0)\n\t"                                                                \
    "v_alignbyte_b32 v102, v101, v100, %[a]\n\t"                                            \
    "v_perm_b32 %[x], %[x], v102, %[sel]\n\t"                                               \
    "v_add_u32 %[curv], %[stot], %[curv]\n\t"                                               \
    "v_and_b32 %[curv], 0xfff, %[curv]\n\t"                                                 \
    "global_load_dwordx2 v[100:101], %[curv], %[sb]\n\t"
// the product's step: the renormalisation falls through, a step without one branches over it
#define STEP_INLINE(N, KW, KO)                                                              \
    P_TEST P_BFLY(KW, KO) P_LDS P_XUPD(KW)                                                  \
    "v_cmp_gt_u32_e32 vcc, 0x800000, %[x]\n\t"                                              \
    "s_cbranch_vccz .Ldone" #N "_%=\n\t"                                                    \
    P_REN(N)                                                                                \
    ".Ldone" #N "_%=:\n\t"
// the renormalisation out of line: a step without one takes no branch, one with it takes two
#define STEP_OOL(N, KW, KO)                                                                 \
    P_TEST P_BFLY(KW, KO) P_LDS P_XUPD(KW)                                                  \
    "v_cmp_gt_u32_e32 vcc, 0x800000, %[x]\n\t"                                              \
    "s_cbranch_vccnz .Lren" #N "_%=\n\t"                                                    \
    ".Ldone" #N "_%=:\n\t"
#define OOL_REN(N, KW, KO) ".Lren" #N "_%=:\n\t" P_REN(N) "s_branch .Ldone" #N "_%=\n\t"
#define STEP_NOLDS(N, KW, KO)                                                               \
    P_TEST_NOWAIT P_BFLY(KW, KO) P_XUPD(KW)                                                 \
    "v_cmp_gt_u32_e32 vcc, 0x800000, %[x]\n\t"                                              \
    "s_cbranch_vccnz .Lend_%=\n\t"
#define STEP_NOBFLY(N, KW, KO)                                                              \
    P_TEST P_NOBFLY(KW, KO) P_LDS P_XUPD(KW)                                                \
    "v_cmp_gt_u32_e32 vcc, 0x800000, %[x]\n\t"                                              \
    "s_cbranch_vccnz .Lend_%=\n\t"
#define STEP_NOXUPD(N, KW, KO)                                                              \
    P_TEST P_BFLY(KW, KO) P_LDS                                                             \
    "v_cmp_gt_u32_e32 vcc, 0x800000, %[x]\n\t"                                              \
    "s_cbranch_vccnz .Lend_%=\n\t"
#define STEP_NOCHK(N, KW, KO) P_TEST P_BFLY(KW, KO) P_LDS P_XUPD(KW)
#define STEP_VALU(N, KW, KO) "v_add_u32 %[x], %[x], %[e]\n\t"
#define STEP_BR_TAKEN(N, KW, KO)                                                            \
    "v_cmp_gt_u32_e32 vcc, 0x800000, %[x]\n\t"                                              \
    "s_cbranch_vccz .Ldone" #N "_%=\n\t"                                                    \
    "v_add_u32 %[x], %[x], %[e]\n\t"                                                        \
    ".Ldone" #N "_%=:\n\t"
#define STEP_BR_NOT(N, KW, KO)                                                              \
    "v_cmp_gt_u32_e32 vcc, 0x800000, %[x]\n\t"                                              \
    "s_cbranch_vccnz .Lend_%=\n\t"
#define STEP_LDSLAT(N, KW, KO)                                                              \
    "v_and_or_b32 %[t], %[e], 28, %[lb]\n\t"                                                \
    "ds_read_b32 %[e], %[t]\n\t"                                                            \
    "s_waitcnt lgkmcnt(0)\n\t"
#define STEP_DPP(N, KW, KO) MIN_("v", "quad_perm:[1,0,3,2]") "s_nop 0\n\t"
#define STEP_DPP_FILL(N, KW, KO) MIN_("v", "quad_perm:[1,0,3,2]") "v_add_u32 %[d], %[d], %[e]\n\t"
#define STEP_WAITCNT(N, KW, KO) "s_waitcnt lgkmcnt(0)\n\t" "v_add_u32 %[x], %[x], %[e]\n\t"
#define X16(M)                                                                              \
    M(0, "k0", "k1") M(1, "k1", "k0") M(2, "k0", "k1") M(3, "k1", "k0") M(4, "k0", "k1") M(5, "k1", "k0") M(6, "k0", "k1")     \
    M(7, "k1", "k0") M(8, "k0", "k1") M(9, "k1", "k0") M(10, "k0", "k1") M(11, "k1", "k0") M(12, "k0", "k1") M(13, "k1", "k0") \
    M(14, "k0", "k1") M(15, "k1", "k0")
#define BODY(STEPS, TAIL)                                                                   \
    "v_mov_b32 v100, 0\n\t v_mov_b32 v101, 0\n\t"                                           \
    ".Lloop_%=:\n\t" STEPS                                                                  \
    "s_sub_u32 %[cnt], %[cnt], 1\n\t"                                                       \
    "s_cmp_lg_u32 %[cnt], 0\n\t"                                                            \
    "s_cbranch_scc1 .Lloop_%=\n\t"                                                          \
    "s_branch .Lend_%=\n\t" TAIL ".Lend_%=:\n\t"                                            \
    "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"

template <int V>
__global__ __launch_bounds__(64) void dec_kernel(uint32_t entry, uint32_t x0, int iters, const uint8_t* stream, uint32_t* out,
                                                 long long* cyc) {
    __shared__ uint32_t ent[64];
    const int l = threadIdx.x;
    ent[l] = entry;
    __syncthreads();
    typedef __attribute__((address_space(3))) const void* LdsPtr;
    const uint32_t lb = (uint32_t)(uintptr_t)(LdsPtr)ent + (uint32_t)(l & 7) * 32u;
    uint32_t x = x0, e = entry, curv = 0, k0 = 0, k1 = 0, v = 0, xs = 0, t = 0, d = 0, f = 0, a = 0, sel = 0, stot = 0;
    uint32_t cnt = (uint32_t)iters;
    const uint32_t selno = 0x07060504u, selone = 0x06050400u;
    const unsigned long long ktop = 0x8000800080008000ull;
    const long long t0 = clock64();
#define OPERANDS                                                                                                              \
    : [x] "+v"(x), [e] "+v"(e), [curv] "+v"(curv), [k0] "+v"(k0), [k1] "+v"(k1), [v] "+v"(v), [xs] "+v"(xs), [t] "+v"(t),      \
      [d] "+v"(d), [f] "+v"(f), [a] "+v"(a), [sel] "+v"(sel), [cnt] "+s"(cnt), [stot] "+s"(stot)                              \
    : [lb] "v"(lb), [sb] "s"(stream), [ktop] "s"(ktop), [selno] "v"(selno), [selone] "v"(selone)                              \
    : "vcc", "scc", "memory", "v100", "v101", "v102", "v103"
    if (V == 0) asm volatile(BODY(X16(STEP_INLINE), "") OPERANDS);
    if (V == 1) asm volatile(BODY(X16(STEP_OOL), X16(OOL_REN)) OPERANDS);
    if (V == 2) asm volatile(BODY(X16(STEP_NOLDS), "") OPERANDS);
    if (V == 3) asm volatile(BODY(X16(STEP_NOBFLY), "") OPERANDS);
    if (V == 4) asm volatile(BODY(X16(STEP_NOXUPD), "") OPERANDS);
    if (V == 5) asm volatile(BODY(X16(STEP_NOCHK), "") OPERANDS);
    if (V == 6) asm volatile(BODY(X16(STEP_VALU), "") OPERANDS);
    if (V == 7) asm volatile(BODY(X16(STEP_BR_TAKEN), "") OPERANDS);
    if (V == 8) asm volatile(BODY(X16(STEP_BR_NOT), "") OPERANDS);
    if (V == 9) asm volatile(BODY(X16(STEP_LDSLAT), "") OPERANDS);
    if (V == 10) asm volatile(BODY(X16(STEP_DPP), "") OPERANDS);
    if (V == 11) asm volatile(BODY(X16(STEP_DPP_FILL), "") OPERANDS);
    if (V == 12) asm volatile(BODY(X16(STEP_WAITCNT), "") OPERANDS);
    const long long t1 = clock64();
    out[blockIdx.x * 64 + l] = x + e + v + d + k0 + k1;
    if (l == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V>
void run(const char* name, uint32_t entry, uint32_t x0, const uint8_t* d_stream, uint32_t* d_out, long long* d_cyc) {
    const int iters = 20000;
    for (int grid : {1, 1024}) {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(dec_kernel<V>, dim3(grid), dim3(64), 0, 0, entry, x0, iters, d_stream, d_out, d_cyc);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(dec_kernel<V>, dim3(grid), dim3(64), 0, 0, entry, x0, iters, d_stream, d_out, d_cyc);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<long long> cyc(grid);
        CHECK(hipMemcpy(cyc.data(), d_cyc, grid * sizeof(long long), hipMemcpyDeviceToHost));
        uint32_t o = 0;
        CHECK(hipMemcpy(&o, d_out, 4, hipMemcpyDeviceToHost));
        double avg = 0;
        for (auto c : cyc) avg += (double)c;
        avg /= grid;
        const double steps = (double)iters * 16;
        printf("%-58s grid %5d  %7.3f ms  clock64/step %7.2f  ns/step %7.2f  (out %08x)\n", name, grid, ms, avg / steps,
               ms * 1e6 / steps, o);
    }
}

int main() {
    uint8_t* d_stream;
    uint32_t* d_out;
    long long* d_cyc;
    CHECK(hipMalloc(&d_stream, 8192));
    CHECK(hipMemset(d_stream, 0, 8192));
    CHECK(hipMalloc(&d_out, 1024 * 64 * 4));
    CHECK(hipMalloc(&d_cyc, 1024 * 8));
    const uint32_t never = 4096u << 7;  // C = 0, F = 4096, symbol 0: x' = x
    const uint32_t always = 16u << 7;   // C = 0, F = 16 over a zero stream: one byte per state per step
    const uint32_t X = 1u << 23;
    run<0>("step, renorm inline, none taken (branch over it)", never, X | 0x123, d_stream, d_out, d_cyc);
    run<1>("step, renorm out of line, none taken (no branch)", never, X | 0x123, d_stream, d_out, d_cyc);
    run<2>("  ... without the LDS read and its wait", never, X | 0x123, d_stream, d_out, d_cyc);
    run<3>("  ... without the butterfly", never, X | 0x123, d_stream, d_out, d_cyc);
    run<4>("  ... without the state update", never, X | 0x123, d_stream, d_out, d_cyc);
    run<5>("  ... without compare and branch", never, X | 0x123, d_stream, d_out, d_cyc);
    run<0>("step, renorm inline, every step takes one byte", always, X, d_stream, d_out, d_cyc);
    run<1>("step, renorm out of line, every step (two branches)", always, X, d_stream, d_out, d_cyc);
    run<6>("one dependent v_add", never, X, d_stream, d_out, d_cyc);
    run<7>("v_cmp + branch taken over one v_add", never, X, d_stream, d_out, d_cyc);
    run<8>("v_cmp + branch not taken", never, X, d_stream, d_out, d_cyc);
    run<9>("dependent LDS read (and_or, ds_read, wait)", 0, X, d_stream, d_out, d_cyc);
    run<10>("dpp min + s_nop 0", never, X, d_stream, d_out, d_cyc);
    run<11>("dpp min + independent v_add", never, X, d_stream, d_out, d_cyc);
    run<12>("s_waitcnt (nothing outstanding) + v_add", never, X, d_stream, d_out, d_cyc);
    return 0;
}
