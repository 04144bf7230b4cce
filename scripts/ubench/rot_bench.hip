// Micro-benchmark of the "rotating" rANS encoder chain on gfx950 (round 2): the four states of a block live in the
// four 16-lane DPP rows of ONE wave, step u of state j runs in lane j*16 + (u & 15) on that lane's own record, and
// the state moves to the next lane with a row_ror:1 DPP move.  Every lane executes every step (15 of 16 compute
// garbage), but the wave needs no LDS read per step, no 64 record registers and no per-step log write: the
// bank-masked DPP move leaves "state before my step" in one of four registers per lane.
// Checked against a scalar host model; prints cycles per step.  Not part of the library.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

constexpr int kRing = 8;  // half batches of records kept in LDS (the records repeat with this period)

// one step: XP = previous output register, LG = log register (state before the step), BK = bank mask, NOP = "" or s_nop
#define ROT_STEP(S, XP, LG, BK, NOP)                                                                          \
    NOP "v_mov_b32_dpp %[" #LG #S "], %[" #XP #S "] row_ror:1 row_mask:0xf bank_mask:" #BK "\n\t"            \
    "v_sub_u32 %[t" #S "], %[a" #S "], %[" #LG #S "]\n\t"                                                     \
    "v_lshrrev_b32 %[t" #S "], 31, %[t" #S "]\n\t"                                                            \
    "v_alignbyte_b32 %[r" #S "], 0, %[" #LG #S "], %[t" #S "]\n\t"                                            \
    "v_mul_hi_u32 %[m" #S "], %[r" #S "], %[b" #S "]\n\t"                                                     \
    "v_lshrrev_b32_sdwa %[m" #S "], %[d" #S "], %[m" #S "] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t" \
    "v_mul_u32_u24 %[m" #S "], %[m" #S "], %[d" #S "]\n\t"                                                    \
    "v_add3_u32 %[" #XP #S "], %[r" #S "], %[c" #S "], %[m" #S "]\n\t"

#define ROT4(S, BK, NOP) ROT_STEP(S, x, l0, BK, NOP) ROT_STEP(S, x, l1, BK, NOP) ROT_STEP(S, x, l2, BK, NOP) ROT_STEP(S, x, l3, BK, NOP)
#define ROT16(S, NOP) ROT4(S, 0x1, NOP) ROT4(S, 0x2, NOP) ROT4(S, 0x4, NOP) ROT4(S, 0x8, NOP)

// the same with two blocks A and B interleaved instruction by instruction (no instruction depends on its predecessor)
#define ROT_STEP2(LG, BK)                                                                                     \
    "v_mov_b32_dpp %[" #LG "A], %[xA] row_ror:1 row_mask:0xf bank_mask:" #BK "\n\t"                           \
    "v_mov_b32_dpp %[" #LG "B], %[xB] row_ror:1 row_mask:0xf bank_mask:" #BK "\n\t"                           \
    "v_sub_u32 %[tA], %[aA], %[" #LG "A]\n\t"                                                                 \
    "v_sub_u32 %[tB], %[aB], %[" #LG "B]\n\t"                                                                 \
    "v_lshrrev_b32 %[tA], 31, %[tA]\n\t"                                                                      \
    "v_lshrrev_b32 %[tB], 31, %[tB]\n\t"                                                                      \
    "v_alignbyte_b32 %[rA], 0, %[" #LG "A], %[tA]\n\t"                                                        \
    "v_alignbyte_b32 %[rB], 0, %[" #LG "B], %[tB]\n\t"                                                        \
    "v_mul_hi_u32 %[mA], %[rA], %[bA]\n\t"                                                                    \
    "v_mul_hi_u32 %[mB], %[rB], %[bB]\n\t"                                                                    \
    "v_lshrrev_b32_sdwa %[mA], %[dA], %[mA] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t" \
    "v_lshrrev_b32_sdwa %[mB], %[dB], %[mB] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t" \
    "v_mul_u32_u24 %[mA], %[mA], %[dA]\n\t"                                                                   \
    "v_mul_u32_u24 %[mB], %[mB], %[dB]\n\t"                                                                   \
    "v_add3_u32 %[xA], %[rA], %[cA], %[mA]\n\t"                                                               \
    "v_add3_u32 %[xB], %[rB], %[cB], %[mB]\n\t"
#define ROT4_2(BK) ROT_STEP2(l0, BK) ROT_STEP2(l1, BK) ROT_STEP2(l2, BK) ROT_STEP2(l3, BK)
#define ROT16_2 ROT4_2(0x1) ROT4_2(0x2) ROT4_2(0x4) ROT4_2(0x8)

__device__ __forceinline__ uint32_t sel4(uint32_t l0, uint32_t l1, uint32_t l2, uint32_t l3, int lane) {
    const int k = lane & 3;
    return k == 0 ? l0 : k == 1 ? l1 : k == 2 ? l2 : l3;
}

// MODE 0: one block, no nop; 1: one block, s_nop 1 in front of the DPP move; 2: two blocks interleaved
template <int MODE>
__global__ __launch_bounds__(64) void rot_kernel(const uint4* recs, int nhalf, uint32_t* out, long long* cyc) {
    __shared__ uint4 sP[2][kRing][64];
    __shared__ uint32_t slog[2][2][64];
    const int l = threadIdx.x;
    for (int i = l; i < kRing * 64; i += 64) {
        sP[0][i >> 6][i & 63] = recs[i];
        sP[1][i >> 6][i & 63] = recs[kRing * 64 + i];
    }
    __syncthreads();
    uint32_t xA = 1u << 23, xB = 1u << 23, accA = 0, accB = 0;
    uint32_t l0A = 0, l1A = 0, l2A = 0, l3A = 0, l0B = 0, l1B = 0, l2B = 0, l3B = 0;
    uint4 PA = sP[0][0][l], PB = sP[1][0][l];
    const long long t0 = clock64();
    for (int h = 0; h < nhalf; h++) {
        const uint4 NA = sP[0][(h + 1) & (kRing - 1)][l];
        uint4 NB = NA;
        if (MODE == 2) NB = sP[1][(h + 1) & (kRing - 1)][l];
        uint32_t tA, rA, mA, tB, rB, mB;
        if (MODE == 0)
            asm volatile(ROT16(A, "")
                         : [xA] "+v"(xA), [l0A] "+v"(l0A), [l1A] "+v"(l1A), [l2A] "+v"(l2A), [l3A] "+v"(l3A), [tA] "=&v"(tA),
                           [rA] "=&v"(rA), [mA] "=&v"(mA)
                         : [aA] "v"(PA.x), [bA] "v"(PA.y), [cA] "v"(PA.z), [dA] "v"(PA.w));
        else if (MODE == 1)
            asm volatile(ROT16(A, "s_nop 1\n\t")
                         : [xA] "+v"(xA), [l0A] "+v"(l0A), [l1A] "+v"(l1A), [l2A] "+v"(l2A), [l3A] "+v"(l3A), [tA] "=&v"(tA),
                           [rA] "=&v"(rA), [mA] "=&v"(mA)
                         : [aA] "v"(PA.x), [bA] "v"(PA.y), [cA] "v"(PA.z), [dA] "v"(PA.w));
        else if (MODE == 3)
            asm volatile(ROT16(A, "s_nop 0\n\t")
                         : [xA] "+v"(xA), [l0A] "+v"(l0A), [l1A] "+v"(l1A), [l2A] "+v"(l2A), [l3A] "+v"(l3A), [tA] "=&v"(tA),
                           [rA] "=&v"(rA), [mA] "=&v"(mA)
                         : [aA] "v"(PA.x), [bA] "v"(PA.y), [cA] "v"(PA.z), [dA] "v"(PA.w));
        else if (MODE == 4)
            asm volatile(ROT16(A, "v_add_u32 %[f0], %[f0], %[f1]\n\tv_add_u32 %[f1], %[f1], %[f0]\n\t")
                         : [xA] "+v"(xA), [l0A] "+v"(l0A), [l1A] "+v"(l1A), [l2A] "+v"(l2A), [l3A] "+v"(l3A), [tA] "=&v"(tA),
                           [rA] "=&v"(rA), [mA] "=&v"(mA), [f0] "+v"(accB), [f1] "+v"(xB)
                         : [aA] "v"(PA.x), [bA] "v"(PA.y), [cA] "v"(PA.z), [dA] "v"(PA.w));
        else if (MODE == 5)
            asm volatile(ROT16(A, "v_add_u32 %[f0], %[f0], %[f1]\n\t")
                         : [xA] "+v"(xA), [l0A] "+v"(l0A), [l1A] "+v"(l1A), [l2A] "+v"(l2A), [l3A] "+v"(l3A), [tA] "=&v"(tA),
                           [rA] "=&v"(rA), [mA] "=&v"(mA), [f0] "+v"(accB), [f1] "+v"(xB)
                         : [aA] "v"(PA.x), [bA] "v"(PA.y), [cA] "v"(PA.z), [dA] "v"(PA.w));
        else
            asm volatile(ROT16_2
                         : [xA] "+v"(xA), [l0A] "+v"(l0A), [l1A] "+v"(l1A), [l2A] "+v"(l2A), [l3A] "+v"(l3A), [tA] "=&v"(tA),
                           [rA] "=&v"(rA), [mA] "=&v"(mA), [xB] "+v"(xB), [l0B] "+v"(l0B), [l1B] "+v"(l1B), [l2B] "+v"(l2B),
                           [l3B] "+v"(l3B), [tB] "=&v"(tB), [rB] "=&v"(rB), [mB] "=&v"(mB)
                         : [aA] "v"(PA.x), [bA] "v"(PA.y), [cA] "v"(PA.z), [dA] "v"(PA.w), [aB] "v"(PB.x), [bB] "v"(PB.y),
                           [cB] "v"(PB.z), [dB] "v"(PB.w));
        const uint32_t lgA = sel4(l0A, l1A, l2A, l3A, l);
        slog[0][h & 1][l] = lgA;
        accA = ((accA << 1) | (accA >> 31)) ^ lgA;
        if (MODE == 2) {
            const uint32_t lgB = sel4(l0B, l1B, l2B, l3B, l);
            slog[1][h & 1][l] = lgB;
            accB = ((accB << 1) | (accB >> 31)) ^ lgB;
        }
        PA = NA;
        PB = NB;
    }
    const long long t1 = clock64();
    out[blockIdx.x * 256 + l] = xA;
    out[blockIdx.x * 256 + 64 + l] = accA;
    out[blockIdx.x * 256 + 128 + l] = xB;
    out[blockIdx.x * 256 + 192 + l] = accB;
    if (l == 0) cyc[blockIdx.x] = t1 - t0 + (slog[0][0][1] & 0);
}

struct Rec { uint32_t a, b, c, d; };

static uint32_t step(uint32_t x, const Rec& r) {
    if (x > r.a) x >>= 8;
    const uint32_t q = (uint32_t)(((unsigned long long)x * r.b) >> 32) >> ((r.d >> 24) & 31);
    return x + r.c + q * (r.d & 0xFFFFFF);
}

template <int MODE>
void run(const char* name, const uint4* d_recs, const std::vector<Rec>& recs, int grid, int nhalf, uint32_t* d_out, long long* d_cyc) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rot_kernel<MODE>, dim3(grid), dim3(64), 0, 0, d_recs, nhalf, d_out, d_cyc);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(rot_kernel<MODE>, dim3(grid), dim3(64), 0, 0, d_recs, nhalf, d_out, d_cyc);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> cyc(grid);
    CHECK(hipMemcpy(cyc.data(), d_cyc, grid * sizeof(long long), hipMemcpyDeviceToHost));
    std::vector<uint32_t> out(256);
    CHECK(hipMemcpy(out.data(), d_out, 256 * 4, hipMemcpyDeviceToHost));
    // host model
    int bad = 0;
    for (int blk = 0; blk < (MODE == 2 ? 2 : 1); blk++) {
        uint32_t x[4] = {1u << 23, 1u << 23, 1u << 23, 1u << 23};
        std::vector<uint32_t> acc(64, 0);
        for (int h = 0; h < nhalf; h++)
            for (int tt = 0; tt < 16; tt++)
                for (int j = 0; j < 4; j++) {
                    const int lane = j * 16 + tt;
                    const Rec& r = recs[(size_t)blk * kRing * 64 + (size_t)(h & (kRing - 1)) * 64 + lane];
                    acc[lane] = ((acc[lane] << 1) | (acc[lane] >> 31)) ^ x[j];
                    x[j] = step(x[j], r);
                }
        for (int j = 0; j < 4; j++)
            if (out[blk * 128 + j * 16 + 15] != x[j]) bad++;
        for (int lane = 0; lane < 64; lane++)
            if (out[blk * 128 + 64 + lane] != acc[lane]) bad++;
    }
    double avg = 0;
    for (auto c : cyc) avg += (double)c;
    avg /= grid;
    const double steps = (double)nhalf * 16;
    printf("%-36s grid %5d  %8.3f ms  clock64/step %7.2f  ns/step %7.2f  %s\n", name, grid, ms, avg / steps, ms * 1e6 / steps,
           bad ? "MISMATCH vs host model" : "ok");
}

int main() {
    std::vector<Rec> recs(2 * kRing * 64);
    srand(1);
    for (auto& r : recs) {
        const uint32_t f = 16 + (rand() % 2000);
        uint32_t shift = 0;
        while ((unsigned long long)f > (1ull << shift)) shift++;
        r.a = (f << 19) - 1;
        r.b = (uint32_t)(((1ull << (shift + 31)) + f - 1) / f);
        r.c = rand() % (4096 - f);
        r.d = (4096 - f) | ((shift - 1) << 24);
    }
    uint4* d_recs;
    CHECK(hipMalloc(&d_recs, recs.size() * sizeof(Rec)));
    CHECK(hipMemcpy(d_recs, recs.data(), recs.size() * sizeof(Rec), hipMemcpyHostToDevice));
    uint32_t* d_out;
    long long* d_cyc;
    CHECK(hipMalloc(&d_out, 8192 * 256 * 4));
    CHECK(hipMalloc(&d_cyc, 8192 * 8));
    const int nhalf = 23438;  // 375 000 steps = one 1.5 MB quality block
    for (int grid : {1, 1024, 2048}) {
        run<0>("rotating, one block, no nop", d_recs, recs, grid, nhalf, d_out, d_cyc);
        run<1>("rotating, one block, s_nop 1", d_recs, recs, grid, nhalf, d_out, d_cyc);
        run<3>("rotating, one block, s_nop 0", d_recs, recs, grid, nhalf, d_out, d_cyc);
        run<4>("rotating, one block, 2 VALU fillers", d_recs, recs, grid, nhalf, d_out, d_cyc);
        run<5>("rotating, one block, 1 VALU filler", d_recs, recs, grid, nhalf, d_out, d_cyc);
        run<2>("rotating, two blocks interleaved", d_recs, recs, grid, nhalf, d_out, d_cyc);
    }
    return 0;
}
