// K1 — CRAM 3.0 rANS 4x8 encoder (order 0 and order 1) for gfx950, byte-identical to htsjdk 2.4.1.
//
// Reference (htsjdk!encoding/rans/*, original source lines):
//   RANS#compress L60-73, #encode_order0_way4 L90-106, #encode_order1_way4 L111-127, #finalizeCompressed L132-142
//   Frequencies#calcFrequencies_o0 L109-145, #calcFrequencies_o1 L149-193, #writeFrequencies_o0/o1 L215-244/269-328,
//   #buildSyms_o0/o1 L197-211/248-265, Encoding#RansEncSymbolInit L21-43, #RansEncPutSymbol L46-71,
//   E04#compress L12-57, E14#compress L13-86, Utils#reverse L7-33
//
// Execution model.  A CRAM block is one rANS stream whose four interleaved states form four strictly serial
// dependency chains; byte identity forbids re-blocking, so the only parallelism inside a block is (a) the four
// states and (b) everything that does not depend on the states.  One wave issues a vector instruction every 4.4-5.4
// cycles whether it depends on its predecessor or not (scripts/ubench/step_bench.hip), so the time of a block is (vector
// instructions the chain's wave issues per step) x ~5 cycles x steps: everything that is not the chain is moved OFF the
// chain's wave, and since a vector instruction costs the same whatever its lanes do, ONE chain wave serves the chains of
// four blocks at once (a block per 16-lane DPP row).  The kernel is a resident grid of nine-wave workgroups, each with
// four "slots" that pull blocks (longest first) from a ticket counter independently of each other:
//   waves 2s, 2s+1     the two waves of slot s.  Both histogram the slot's block (LDS counters, lane-replicated for
//   ("aux", "helper")  small alphabets); the helper wave then normalises, serialises the frequency table, builds packed
//                      encoder symbols (a row per lane) and, per batch of 64 coding steps (four groups of 16), fetches the
//                      symbol/context bytes, looks the symbols up, expands them into a 16-byte record per (step, state)
//                      in an LDS staging buffer, and turns the states the chain logged for an earlier batch into output
//                      bytes, placed with DPP prefix sums in the reference's emission order (state 3,2,1,0 per step) and
//                      written from the end of the slot downwards, which realises Utils#reverse for free.
//   wave 8 ("chain")   row s of the wave carries the four states of slot s's block: lane 4q + t of the row holds the
//                      records of steps 4q .. 4q+3 (of a group of 16) of state 3 - t, the states move four lanes per
//                      run (row_ror:4 DPP move) and are coded four steps in a row by the lane that holds those records;
//                      the bank-masked move leaves "state before my run" in a log register that leaves as ONE 4-byte LDS
//                      write per lane and group.  Per step: compare, conditional byte shift, mul_hi, shift, 24-bit
//                      multiply, add - Encoding#RansEncPutSymbol with everything state-independent removed.
// Slots and the chain wave meet through two counters per slot in LDS (batches staged / batches coded) instead of a
// workgroup barrier: a slot whose block is being histogrammed, or has none left, simply feeds identity records.  The
// staging area is four batches deep.  Order-1 tables live in LDS over the compacted alphabet (present symbols + 0) when
// n*(n|1)*8 B fits the table region, else (same dense layout) in a per-slot global scratch.
#include <climits>

#include "internal.h"

namespace {

constexpr int kRun = 4;      // consecutive steps a chain lane codes on its own four records
constexpr int kGroup = 16;   // steps per group: four runs, one per DPP bank of a row
constexpr int kGroups = 4;   // groups between two barriers (amortises the barrier and both waves' fixed costs)
constexpr int kBatch = kGroup * kGroups;
constexpr int kStageBufs = 4;  // batch i-1 is being turned into bytes, i runs, i+1 is prefetched, i+2 is being staged

constexpr int kWgThreads = 128;   // the two waves of a slot
constexpr int kSlots = 4;         // blocks in flight per workgroup, one per row of the chain wave
// Twelve waves are launched and nine work.  The hardware places wave w of a workgroup on SIMD (w + k) mod 4 (measured:
// scripts/ubench/hwid_bench.hip), so waves 0, 4 and 8 share a SIMD: wave 0 is the chain wave and waves 4 and 8 end at
// once, which leaves the chain a SIMD of its own; the eight slot waves are 1-3, 5-7, 9 and 10 (wave 11 ends too).
constexpr int kBlockThreads = 12 * 64;

constexpr int kRingBytes = 4096;  // input ring of the helper wave: 4 x 1 KiB (order 1) or 1 x 4 KiB (order 0)

// LDS of one slot, statically sized so that the four slots of a workgroup fill the 160 KiB of a CU.
struct EncP2 {              // phase 2 (tables)
    uint32_t rowT[256];     // order-1 row totals (0 = context absent)
    uint32_t rowoff[256];   // serialised length of each row
    uint32_t rowhdr[256];   // context-level header of each row: count | byte0 << 8 | byte1 << 16
};
struct alignas(16) EncP3 {  // phase 3 (coding loop)
    // [buffer][helper lane * 4 + k]: the record of step 4 * (lane >> 2) + k of the batch, state 3 - (lane & 3):
    // x_max-1, rcp_freq, bias, cmpl_freq | shift<<24 | small<<31.  A chain lane reads the four records of its run as 64 contiguous bytes.
    uint4 stage[kStageBufs][kBatch * 4];
    uint32_t xs[kStageBufs][64];  // [buffer][helper lane]: state before the first step of that lane's run
    uint8_t ring[kRingBytes];  // input bytes (dense symbol indices when the order-1 tables are in LDS)
    uint32_t gflag[kStageBufs];  // bit G: group G of the batch holds a symbol with freq < 16
    uint32_t fin[4];        // states after the last batch the chain ran
    unsigned long long cbusy, cbusy2, cgen;  // profiling runs only: cycles the chain wave spent outside barriers
};
struct alignas(16) EncFixed {
    uint8_t dmap[256];    // byte value -> dense index
    uint8_t syml[256];    // dense index -> byte value
    // block hand-over between the waves (written by the helper wave, read by both after a barrier)
    int32_t mode, stride, ent_off, tab_len, status;
    uint32_t ticket;
    uint32_t sync_cnt;    // arrivals at the slot's two-wave barrier
    // hand-over between the slot's helper wave and the chain wave: absolute batch numbers of this slot since the launch
    uint32_t staged;      // batches below this number are staged (written by the helper wave)
    uint32_t done;        // batches below this number are coded and logged (written by the chain wave)
    uint32_t emitted;     // batches below this number are turned into bytes: their staging buffers are free (helper wave)
    uint32_t base;        // number of the current block's batch 0
    uint32_t exitf;       // the slot has run out of blocks
    uint32_t pad[4];
    uint4 ident[4];       // identity records: what the chain wave's row reads while the slot has nothing staged
    uint32_t hwid_chain;  // profiling build: HW_ID of the chain wave, batch-times this row stood still beside working rows
    uint32_t idle_chain;
    uint32_t pad2[2];
    uint32_t sink[16];    // ... and where it writes meanwhile (p3 below is the slot's histogram space during phases 1 and 2)
    // The union comes last, directly in front of the table region: the part of the coding loop's buffers that phase 2
    // does not use (everything behind p2) is free during phases 1 and 2 and forms, together with the table region, one
    // contiguous "big" area for what only those phases need (pair counters, histogram words, staged rows).
    union {
        EncP2 p2;
        EncP3 p3;
    };
};
static_assert(sizeof(EncP3) % 16 == 0 && sizeof(EncP2) % 16 == 0, "the spare area starts and ends on 16-byte boundaries");
constexpr int kTabBytes = ((160 * 1024 / 4 - (int)sizeof(EncFixed)) & ~255) - 256;  // table region (symbols, counts, replicas)
constexpr int kSpareBytes = (int)sizeof(EncP3) - (int)sizeof(EncP2);               // free in phases 1 and 2
constexpr int kBigBytes = kSpareBytes + kTabBytes;
struct alignas(16) EncLds : EncFixed {
    alignas(16) uint8_t tab[kTabBytes];
};
static_assert(sizeof(EncLds) == sizeof(EncFixed) + kTabBytes, "table region directly behind the coding-loop buffers");
typedef EncLds EncFixedLds;  // phases 1 and 2 only touch the fixed part
#ifdef HIP_SIM
static EncLds lds4[kSlots];
#else
__shared__ EncLds lds4[kSlots];
#endif
static_assert(sizeof(EncLds) * kSlots <= 160 * 1024, "four slots per CU");
#define lds (lds4[slot_idx()])  // the slot of the calling wave (waves 2s and 2s+1)

// A value that is the same in every lane of the wave, moved to a scalar register so that branches on it are scalar
// branches: a wave then never walks a path of the other wave's role with an empty EXEC mask (where a barrier would
// still count).
#ifdef HIP_SIM
#define UNIFORM(x) (x)
#else
#define UNIFORM(x) __builtin_amdgcn_readfirstlane((int)(x))
#endif

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ int phys_wave() { return UNIFORM(threadIdx.x >> 6); }
__device__ __forceinline__ int worker_id() { return phys_wave() - 1 - (phys_wave() >> 2); }  // 1-3, 5-7, 9, 10 -> 0..7
__device__ __forceinline__ int wave_id() { return worker_id() & 1; }   // within the slot: 0 aux, 1 helper
__device__ __forceinline__ int slot_idx() { return (worker_id() >> 1) & 3; }
__device__ __forceinline__ int stid() { return ((worker_id() & 1) << 6) | (int)(threadIdx.x & 63u); }  // thread of the slot
// start of the big area (kBigBytes): behind p2, through the rest of p3 and the whole table region
__device__ __forceinline__ uint8_t* lds_big() { return reinterpret_cast<uint8_t*>(&lds.p3) + sizeof(EncP2); }

// Hand-over words in LDS (read by one wave while another writes them).  On the GPU a plain LDS access between two
// compiler barriers: the barrier in front makes the compiler load the word again, the one behind keeps what the word
// announces from being read ahead of it (acquire) or written behind it (release); the LDS executes a wave's operations
// in order and one 4-byte access is indivisible, so the hardware needs nothing.  (A volatile access would do, but it
// compiles to a system-scope FLAT access with a full wait right behind it: a round trip on the chain wave's path.)
// Atomics in the host harness.
__device__ __forceinline__ uint32_t ld_flag(const uint32_t* p) {
#ifdef HIP_SIM
    return __atomic_load_n(p, __ATOMIC_ACQUIRE);
#else
    asm volatile("" ::: "memory");
    const uint32_t v = *p;
    asm volatile("" ::: "memory");
    return v;
#endif
}
__device__ __forceinline__ void st_flag(uint32_t* p, uint32_t v) {
#ifdef HIP_SIM
    __atomic_store_n(p, v, __ATOMIC_RELEASE);
#else
    asm volatile("" ::: "memory");
    *p = v;
    asm volatile("" ::: "memory");
#endif
}
// the calling wave sleeps until *p has reached goal (a counter that only grows)
__device__ __forceinline__ void wait_flag_ge(const uint32_t* p, uint32_t goal) {
#ifdef HIP_SIM
    if (lane_id() == 0)
        while ((int32_t)(ld_flag(p) - goal) < 0) std::this_thread::yield();
    __builtin_amdgcn_wave_barrier();
#else
    while ((int32_t)((uint32_t)UNIFORM(ld_flag(p)) - goal) < 0) __builtin_amdgcn_s_sleep(2);
#endif
}
// Barrier between the two waves of a slot (the other slots and the chain wave go their own way, so the workgroup
// barrier is not available): every wave adds one to a counter that only grows and waits for the next even value.
// Memory operations issued before it have completed when it returns (the fences wait for both counters).
__device__ __forceinline__ void slot_sync() {
    uint32_t* cnt = &lds.sync_cnt;
#ifdef HIP_SIM
    __builtin_amdgcn_wave_barrier();
    if (lane_id() == 0) {
        const uint32_t t = __atomic_add_fetch(cnt, 1u, __ATOMIC_ACQ_REL);
        const uint32_t goal = (t + 1u) & ~1u;
        while ((int32_t)(ld_flag(cnt) - goal) < 0) std::this_thread::yield();
    }
    __builtin_amdgcn_wave_barrier();
#else
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    uint32_t t = 0;
    if (lane_id() == 0) t = atomicAdd(cnt, 1u) + 1u;
    const uint32_t goal = ((uint32_t)UNIFORM(t) + 1u) & ~1u;
    for (int k = 0; (int32_t)((uint32_t)UNIFORM(ld_flag(cnt)) - goal) < 0; k++) {
        if (k < 16)
            __builtin_amdgcn_s_sleep(1);
        else
            __builtin_amdgcn_s_sleep(24);  // the aux wave waits out the whole coding loop here
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
#endif
}

// Synchronisation among the lanes of ONE wave (phase 2 runs in the helper wave alone): LDS operations of a wave execute in order, so only the compiler has to be held.
__device__ __forceinline__ void wave_sync() {
#ifdef HIP_SIM
    __builtin_amdgcn_wave_barrier();
#else
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
#endif
}
#define WSYNC() wave_sync()

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ unsigned long long wave_max64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long w = __shfl_xor(v, o);
        v = w > v ? w : v;
    }
    return v;
}
// exclusive prefix sum over the 64 lanes; total = sum of all
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t& total) {
    uint32_t x = v;
    const int l = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t y = __shfl_up(x, o);
        if (l >= o) x += y;
    }
    total = __shfl(x, 63);
    return x - v;
}
__device__ __forceinline__ uint32_t mbcnt(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Encoding#RansEncSymbolInit L21-43 packed in 8 bytes: x = rcp_freq, y = freq | bias<<12 | (rcp_shift-32)<<25
__device__ __forceinline__ uint2 pack_symbol(int32_t freq, uint32_t start, const uint32_t* rcp_tab) {
    uint32_t f = (uint32_t)freq & 4095u;
    uint32_t rcp, bias, sh;
    if (f < 2) {
        rcp = ~0u;
        sh = 0;
        bias = start + 4095u;
    } else {
        uint32_t shift = 32u - (uint32_t)__clz((int)(f - 1));  // smallest shift with (1<<shift) >= f
        rcp = rcp_tab[f];
        sh = shift - 1;
        bias = start;
    }
    return make_uint2(rcp, f | ((bias & 8191u) << 12) | (sh << 25));
}

// expand a packed symbol into the 16-byte record the chain wave consumes
__device__ __forceinline__ uint4 expand_symbol(uint2 e) {
    const uint32_t f = e.y & 4095u;
    uint4 A;
    A.x = (f << 19) - 1u;          // x_max - 1 with x_max = ((L >> 12) << 8) * freq: renormalise iff x > x_max - 1
    A.y = e.x;                     // rcp_freq
    A.z = (e.y >> 12) & 8191u;     // bias
    // cmpl_freq in the low 24 bits (all a 24-bit multiply reads), rcp_shift-32 in bits 24..28 (all a shift reads of
    // its count), bit 31: freq < 16, i.e. x_max < L and the first renormalisation byte is certain
    A.w = (4096u - f) | ((e.y >> 25) << 24) | (f < 16u ? 0x80000000u : 0u);
    return A;
}

// ---------------------------------------------------------------------------------------------------------
// phase 1: histograms
// ---------------------------------------------------------------------------------------------------------
// Both waves of the workgroup count (the chain wave has nothing else to do until the tables exist): tid = 0..127.
// The histogram passes read the block in 16-byte units, unit k of a sweep at byte (sweep * 128 + tid) * 16.  A sweep is
// ~200-400 cycles of LDS atomics against ~1-2 us of load latency, so four sweeps are kept in flight in four fixed
// registers (a rotating register would make the compiler wait at the copy).  F(v, i, prev): unit v at byte i, prev =
// the byte before it (0 at i == 0; only loaded when WITH_PREV).
template <bool WITH_PREV, class F>
__device__ __forceinline__ void for_each_unit(const uint8_t* src, int32_t n16, int l, F f) {
    constexpr int32_t kSweep = kWgThreads * 16;
    uint4 v[4];
    uint32_t pv[4];
    int32_t i = l * 16;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int32_t p = i + k * kSweep;
        v[k] = make_uint4(0, 0, 0, 0);
        pv[k] = 0;
        if (p < n16) {
            v[k] = *reinterpret_cast<const uint4*>(src + p);
            if (WITH_PREV && p > 0) pv[k] = src[p - 1];
        }
    }
    for (; i < n16; i += 4 * kSweep) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int32_t p = i + k * kSweep;
            if (p < n16) f(v[k], p, pv[k]);
            const int32_t pn = p + 4 * kSweep;
            if (pn < n16) {
                v[k] = *reinterpret_cast<const uint4*>(src + pn);
                if (WITH_PREV) pv[k] = src[pn - 1];
            }
        }
    }
}

// order-0 counts of src[0..n) into cnt[256] using kRep0 lane-replicated LDS histograms at rep.  The replicas are 257
// words apart: at 256 every replica would put a symbol in the same LDS bank and the lanes that see the same symbol
// (most of them, in a quality stream) would serialise exactly as without replicas.
constexpr int kRep0 = 16, kRep0Stride = 257;
static_assert(1024 + 2048 <= kTabBytes && kRep0 * kRep0Stride * 4 <= kBigBytes, "order-0 counts, symbols and replicas must fit");
__device__ void hist_o0(const uint8_t* src, int32_t n, uint32_t* cnt, uint32_t* rep) {
    const int l = stid();  // both waves of the slot
    constexpr int R = kRep0;
    for (int i = l; i < R * kRep0Stride; i += kWgThreads) rep[i] = 0;
    slot_sync();
    uint32_t* my = rep + (l & (R - 1)) * kRep0Stride;
    const int32_t n16 = n & ~15;
    for_each_unit<false>(src, n16, l, [&](const uint4& v, int32_t, uint32_t) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            atomicAdd(&my[w[k] & 255u], 1u);
            atomicAdd(&my[(w[k] >> 8) & 255u], 1u);
            atomicAdd(&my[(w[k] >> 16) & 255u], 1u);
            atomicAdd(&my[w[k] >> 24], 1u);
        }
    });
    if (n16 + l < n) atomicAdd(&my[src[n16 + l]], 1u);
    slot_sync();
    for (int j = l; j < 256; j += kWgThreads) {
        uint32_t s = 0;
        for (int r = 0; r < R; r++) s += rep[r * kRep0Stride + j];
        cnt[j] = s;
    }
    slot_sync();
}

// order-1 pair counts: cnt[row(prev)*stride + col(cur)] with prev = 0 before the first byte
// (Frequencies#calcFrequencies_o1 L149-168); indices go through dmap (identity for the 256 x 256 global table).
// Deliberately ONE non-inlined body over generic pointers for the LDS and the global tables, like tables_o1: the
// machine code the wide-alphabet path runs is then the code every small-alphabet block has already exercised.
__device__ __attribute__((noinline)) void hist_o1(const uint8_t* src, int32_t n, uint32_t* cnt, int stride, const uint8_t* dmap, uint32_t* rep,
                        int R, int rstride, int cells) {
    const int l = stid();  // both waves of the slot; cnt has been zeroed (and that made visible) by the caller
    uint32_t* my = cnt;
    if (R > 1) {
        for (int i = l; i < R * rstride; i += kWgThreads) rep[i] = 0;
        my = rep + (l & (R - 1)) * rstride;
    }
    slot_sync();
    const int32_t n16 = n & ~15;
    for_each_unit<true>(src, n16, l, [&](const uint4& v, int32_t, uint32_t prev) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        // all 17 index look-ups first: interleaved with the atomics every look-up would wait out an LDS round trip
        uint32_t col[17];
        col[0] = dmap[prev];
#pragma unroll
        for (int k = 0; k < 16; k++) col[k + 1] = dmap[(w[k >> 2] >> (8 * (k & 3))) & 255u];
#pragma unroll
        for (int k = 0; k < 16; k++) atomicAdd(&my[col[k] * (uint32_t)stride + col[k + 1]], 1u);
    });
    if (n16 + l < n) {
        int32_t i = n16 + l;
        uint32_t prev = i ? src[i - 1] : 0u, c = src[i];
        uint32_t row = (uint32_t)dmap[prev], col = (uint32_t)dmap[c];
        atomicAdd(&my[row * stride + col], 1u);
    }
    if (l == 0) {  // L170-173: the three quarter-start symbols are also counted in context 0
        const int32_t q = n >> 2;
        for (int k = 1; k < 4; k++) {
            uint32_t c = src[k * q];
            atomicAdd(&my[(uint32_t)dmap[c]], 1u);
        }
    }
    slot_sync();
    if (R > 1) {
        for (int j = l; j < cells; j += kWgThreads) {
            uint32_t s = 0;
            for (int r = 0; r < R; r++) s += rep[r * rstride + j];
            cnt[j] += s;
        }
        slot_sync();
    }
}

// Order-1 pair counts of a SHORT block (n < 65536) over an alphabet too wide for 32-bit LDS counters: two 16-bit
// counters per LDS word.  Global-memory atomics on the few hot cells of such a stream (one L2 round trip each, all
// lanes on the same address) cost ~100 cycles per byte; LDS atomics a few.  The counts are then unpacked into the
// global table cnt[cells].
__device__ __attribute__((noinline)) void hist_o1_packed16(const uint8_t* src, int32_t n, uint32_t* cnt, int stride, const uint8_t* dmap,
                                                           uint32_t* w, int cells) {
    const int l = stid();  // both waves of the slot
    const int words = (cells + 1) >> 1;
    for (int i = l; i < words; i += kWgThreads) w[i] = 0;
    slot_sync();
    auto bump = [&](uint32_t idx) { atomicAdd(&w[idx >> 1], 1u << (16u * (idx & 1u))); };
    const int32_t n16 = n & ~15;
    for_each_unit<true>(src, n16, l, [&](const uint4& v, int32_t, uint32_t prev) {
        const uint32_t x[4] = {v.x, v.y, v.z, v.w};
        uint32_t col[17];
        col[0] = dmap[prev];
#pragma unroll
        for (int k = 0; k < 16; k++) col[k + 1] = dmap[(x[k >> 2] >> (8 * (k & 3))) & 255u];
#pragma unroll
        for (int k = 0; k < 16; k++) bump(col[k] * (uint32_t)stride + col[k + 1]);
    });
    if (n16 + l < n) {
        const int32_t i = n16 + l;
        const uint32_t prev = i ? src[i - 1] : 0u, c = src[i];
        bump((uint32_t)dmap[prev] * (uint32_t)stride + dmap[c]);
    }
    if (l == 0) {  // L170-173: the three quarter-start symbols are also counted in context 0
        const int32_t q = n >> 2;
        for (int k = 1; k < 4; k++) bump((uint32_t)dmap[src[k * q]]);
    }
    slot_sync();
    for (int i = l; i < cells; i += kWgThreads) cnt[i] = (w[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
    slot_sync();
}

// ---------------------------------------------------------------------------------------------------------
// phase 2: normalisation, table serialisation, encoder symbols (one table row per lane)
// ---------------------------------------------------------------------------------------------------------
// Frequencies#writeFrequencies_o0 L215-244 over one row in dense form (syml: dense index -> byte value).
template <bool WRITE>
__device__ int row_body(const uint32_t* c, int n, const uint8_t* syml, uint8_t* dst) {
    int p = 0, rle = 0;
#define PUT(v)                          \
    do {                                \
        if (WRITE) dst[p] = (uint8_t)(v); \
        p++;                            \
    } while (0)
    for (int d = 0; d < n; d++) {
        int32_t f = (int32_t)c[d];
        if (f == 0) continue;
        int v = syml[d];
        if (rle != 0) {
            rle--;
        } else {
            PUT(v);
            if (v > 0 && d > 0 && syml[d - 1] == v - 1 && c[d - 1] != 0) {
                int k = d + 1;
                while (k < n && syml[k] == syml[k - 1] + 1 && c[k] != 0) k++;
                rle = k - (d + 1);
                PUT(rle);
            }
        }
        if (f < 128) {
            PUT(f);
        } else {
            PUT(128 | (f >> 8));
            PUT(f & 0xff);
        }
    }
    PUT(0);
#undef PUT
    return p;
}

// order-1: normalise rows (L175-193), serialise (writeFrequencies_o1 L269-328), build symbols (buildSyms_o1 L248-265).
// Returns the table length; table bytes go to tabdst.
__device__ __attribute__((noinline)) int32_t tables_o1(uint32_t* cnt, uint2* ent, int nrows, int n, int stride, EncFixedLds* L, uint8_t* tabdst,
                             const uint32_t* rcp_tab, int32_t tab_cap) {
    const int l = lane_id();
    const int rpl = (nrows + 63) >> 6;  // rows per lane, contiguous so a lane scan gives row order
    // normalise
#pragma unroll 1
    for (int k = 0; k < rpl; k++) {
        int row = l * rpl + k;
        if (row >= nrows) break;
        uint32_t* c = cnt + row * stride;
        uint32_t T = 0;
        for (int j = 0; j < n; j++) T += c[j];
        L->p2.rowT[row] = T;
        if (T == 0) continue;
        const double p = 4096.0 / (double)(int32_t)T;
        int32_t m = 0, M = 0, t2 = 0;
        for (int j = 0; j < n; j++) {
            int32_t v = (int32_t)c[j];
            if (v == 0) continue;
            if (m < v) {
                m = v;
                M = j;
            }
            int32_t f = (int32_t)((double)v * p);
            if (f == 0) f = 1;
            t2 += f;
            c[j] = (uint32_t)f;
        }
        t2++;
        c[M] = (uint32_t)((int32_t)c[M] + (RANS_TOTFREQ - t2));
    }
    WSYNC();
    // serialised length of each present row: context header (index byte, run byte) + body.  The header is kept in
    // LDS (count | byte0 << 8 | byte1 << 16, in the high half of rowoff) so that the write pass only replays it.
    uint32_t mylen = 0;
#pragma unroll 1
    for (int k = 0; k < rpl; k++) {
        const int row = l * rpl + k;
        if (row >= nrows) break;
        uint32_t len = 0, hdr = 0;
        if (L->p2.rowT[row] != 0) {
            const int v = L->syml[row];
            const bool link = v > 0 && row > 0 && L->syml[row - 1] == v - 1 && L->p2.rowT[row - 1] != 0;
            if (!link) {
                hdr = 1u | ((uint32_t)v << 8);
            } else {
                const int pv = v - 1;
                const bool plink = pv > 0 && row > 1 && L->syml[row - 2] == pv - 1 && L->p2.rowT[row - 2] != 0;
                if (!plink) {
                    int kk = row + 1;
                    while (kk < nrows && L->syml[kk] == L->syml[kk - 1] + 1 && L->p2.rowT[kk] != 0) kk++;
                    hdr = 2u | ((uint32_t)v << 8) | ((uint32_t)(kk - (row + 1)) << 16);
                }
            }
            len = (hdr & 3u) + (uint32_t)row_body<false>(cnt + row * stride, n, L->syml, nullptr);
        }
        L->p2.rowoff[row] = len;
        L->p2.rowhdr[row] = hdr;
        mylen += len;
    }
    uint32_t total;
    uint32_t base = wave_excl_scan(mylen, total);
    WSYNC();
    // RANS#allocateIfNeeded L79-85 budgets 257*257*3 bytes for the table; nothing larger can be a valid table, and
    // every row is written strictly inside [0, total): a stray length can therefore never leave the block's slot
    if (total > 257u * 257u * 3u || (int64_t)total + 1 > (int64_t)tab_cap) return CRAMGPU_E_INSUFFICIENT;
#pragma unroll 1
    for (int k = 0; k < rpl; k++) {
        const int row = l * rpl + k;
        if (row >= nrows) break;
        const uint32_t len = L->p2.rowoff[row];
        if (len != 0 && base + len <= total) {
            const uint32_t hdr = L->p2.rowhdr[row];
            uint8_t* d = tabdst + base;
            if ((hdr & 3u) >= 1) d[0] = (uint8_t)(hdr >> 8);
            if ((hdr & 3u) == 2) d[1] = (uint8_t)(hdr >> 16);
            row_body<true>(cnt + row * stride, n, L->syml, d + (hdr & 3u));
        }
        base += len;
    }
    if (l == 0) tabdst[total] = 0;
    WSYNC();
    // encoder symbols
#pragma unroll 1
    for (int k = 0; k < rpl; k++) {
        int row = l * rpl + k;
        if (row >= nrows) break;
        if (L->p2.rowT[row] == 0) continue;
        const uint32_t* c = cnt + row * stride;
        uint2* e = ent + row * stride;
        uint32_t C = 0;
        for (int j = 0; j < n; j++) {
            int32_t f = (int32_t)c[j];
            if (f == 0) continue;
            e[j] = pack_symbol(f, C, rcp_tab);
            C += (uint32_t)f;
        }
    }
    WSYNC();
    return (int32_t)total + 1;
}

// tables_o1 for counts that live in GLOBAL memory (alphabets too wide for the LDS table region).  Walking a row of
// global memory element by element costs a memory round trip per element, so the rows are staged through the LDS
// table region G at a time (coalesced copies), one row per lane, in two sweeps: normalise + measure, then (after the
// scan of the row lengths) serialise + build the symbols.  Same arithmetic, same row_body, as tables_o1.
__device__ __attribute__((noinline)) int32_t tables_o1_staged(uint32_t* cnt, uint2* ent, int n, int stride, EncFixedLds* L,
                                                              uint32_t* stg, int G, uint8_t* tabdst, const uint32_t* rcp_tab,
                                                              int32_t tab_cap) {
    const int l = lane_id();
    // sweep 1: normalise (L175-193), body length of every row
#pragma unroll 1
    for (int g0 = 0; g0 < n; g0 += G) {
        const int rows = min(G, n - g0), words = rows * stride;
        for (int i = l; i < words; i += 64) stg[i] = cnt[g0 * stride + i];
        WSYNC();
        if (l < rows) {
            const int row = g0 + l;
            uint32_t* c = stg + l * stride;
            uint32_t T = 0;
            for (int j = 0; j < n; j++) T += c[j];
            L->p2.rowT[row] = T;
            uint32_t blen = 0;
            if (T != 0) {
                const double p = 4096.0 / (double)(int32_t)T;
                int32_t m = 0, M = 0, t2 = 0;
                for (int j = 0; j < n; j++) {
                    int32_t v = (int32_t)c[j];
                    if (v == 0) continue;
                    if (m < v) {
                        m = v;
                        M = j;
                    }
                    int32_t f = (int32_t)((double)v * p);
                    if (f == 0) f = 1;
                    t2 += f;
                    c[j] = (uint32_t)f;
                }
                t2++;
                c[M] = (uint32_t)((int32_t)c[M] + (RANS_TOTFREQ - t2));
                blen = (uint32_t)row_body<false>(c, n, L->syml, nullptr);
            }
            L->p2.rowoff[row] = blen;
        }
        WSYNC();
        for (int i = l; i < words; i += 64) cnt[g0 * stride + i] = stg[i];
        WSYNC();
    }
    __threadfence();
    WSYNC();
    // context headers (writeFrequencies_o1 L269-328) and offsets: 4 consecutive rows per lane, n <= 256
    uint32_t len4[4], mylen = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int row = l * 4 + k;
        uint32_t len = 0, hdr = 0;
        if (row < n && L->p2.rowT[row] != 0) {
            const int v = L->syml[row];
            const bool link = v > 0 && row > 0 && L->syml[row - 1] == v - 1 && L->p2.rowT[row - 1] != 0;
            if (!link) {
                hdr = 1u | ((uint32_t)v << 8);
            } else {
                const int pv = v - 1;
                const bool plink = pv > 0 && row > 1 && L->syml[row - 2] == pv - 1 && L->p2.rowT[row - 2] != 0;
                if (!plink) {
                    int kk = row + 1;
                    while (kk < n && L->syml[kk] == L->syml[kk - 1] + 1 && L->p2.rowT[kk] != 0) kk++;
                    hdr = 2u | ((uint32_t)v << 8) | ((uint32_t)(kk - (row + 1)) << 16);
                }
            }
            len = (hdr & 3u) + L->p2.rowoff[row];
        }
        len4[k] = len;
        if (row < n) L->p2.rowhdr[row] = hdr;
        mylen += len;
    }
    uint32_t total;
    uint32_t base = wave_excl_scan(mylen, total);
    WSYNC();
    if (total > 257u * 257u * 3u || (int64_t)total + 1 > (int64_t)tab_cap) return CRAMGPU_E_INSUFFICIENT;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int row = l * 4 + k;
        if (row < n) L->p2.rowoff[row] = base | (len4[k] << 20);  // offset < 2^18, length < 2^10
        base += len4[k];
    }
    WSYNC();
    // sweep 2: serialise and build the encoder symbols (buildSyms_o1 L248-265)
#pragma unroll 1
    for (int g0 = 0; g0 < n; g0 += G) {
        const int rows = min(G, n - g0), words = rows * stride;
        for (int i = l; i < words; i += 64) stg[i] = cnt[g0 * stride + i];
        WSYNC();
        if (l < rows) {
            const int row = g0 + l;
            const uint32_t* c = stg + l * stride;
            const uint32_t ol = L->p2.rowoff[row], off = ol & 0xFFFFFu, len = ol >> 20;
            if (len != 0 && off + len <= total) {
                const uint32_t hdr = L->p2.rowhdr[row];
                uint8_t* d = tabdst + off;
                if ((hdr & 3u) >= 1) d[0] = (uint8_t)(hdr >> 8);
                if ((hdr & 3u) == 2) d[1] = (uint8_t)(hdr >> 16);
                row_body<true>(c, n, L->syml, d + (hdr & 3u));
            }
            if (L->p2.rowT[row] != 0) {
                uint2* e = ent + row * stride;
                uint32_t C = 0;
                for (int j = 0; j < n; j++) {
                    int32_t f = (int32_t)c[j];
                    if (f == 0) continue;
                    e[j] = pack_symbol(f, C, rcp_tab);
                    C += (uint32_t)f;
                }
            }
        }
        WSYNC();
    }
    if (l == 0) tabdst[total] = 0;
    WSYNC();
    return (int32_t)total + 1;
}

// order-0: Frequencies#calcFrequencies_o0 L109-145, #writeFrequencies_o0 L215-244, #buildSyms_o0 L197-211.
// cnt[256] raw counts in, normalised out; ent[256].
__device__ int32_t tables_o0(uint32_t* cnt, uint2* ent, int32_t n_in, EncFixedLds* L, uint8_t* tabdst,
                             const uint32_t* rcp_tab, int32_t tab_cap) {
    const int l = lane_id();
    const unsigned long long tr =
        ((unsigned long long)RANS_TOTFREQ << 31) / (unsigned long long)n_in + (unsigned long long)((1 << 30) / n_in);
    uint32_t c4[4];
    unsigned long long key = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        int j = l * 4 + k;
        c4[k] = cnt[j];
        unsigned long long kk = ((unsigned long long)c4[k] << 9) | (unsigned long long)(511 - j);
        if (c4[k] != 0 && kk > key) key = kk;
    }
    key = wave_max64(key);
    const int M = 511 - (int)(key & 511ull);
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (c4[k] != 0) {
            int32_t f = (int32_t)(((unsigned long long)c4[k] * tr) >> 31);
            if (f == 0) f = 1;
            c4[k] = (uint32_t)f;
            s += (uint32_t)f;
        }
    }
    const int32_t fsum = (int32_t)wave_sum(s) + 1;
    uint32_t tot, pre;
    {
        uint32_t mine = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int j = l * 4 + k;
            if (j == M) c4[k] = (uint32_t)((int32_t)c4[k] + (RANS_TOTFREQ - fsum));
            cnt[j] = c4[k];
            mine += c4[k];
        }
        pre = wave_excl_scan(mine, tot);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        int j = l * 4 + k;
        if (c4[k] != 0) ent[j] = pack_symbol((int32_t)c4[k], pre, rcp_tab);
        pre += c4[k];
    }
    WSYNC();
    int32_t len = 0;
    if (l == 0) {
        len = row_body<false>(cnt, 256, L->syml, nullptr);
        if (len <= tab_cap) row_body<true>(cnt, 256, L->syml, tabdst);
    }
    len = __shfl(len, 0);
    WSYNC();
    return len;
}

// ---------------------------------------------------------------------------------------------------------
// phase 3: the coding loop (both waves)
// ---------------------------------------------------------------------------------------------------------
enum { MODE_O0 = 0, MODE_O1_LDS = 1, MODE_O1_GLOBAL = 2 };
constexpr int kGsEntOff = 264 * 1024;  // global scratch: packed symbols behind 256 x 257 counts
static_assert(kGsEntOff >= 4 * 256 * 257 && kGsEntOff + 8 * 256 * 257 <= (int)ENC_GSCRATCH_BYTES, "global scratch layout");

// Input positions.  Step u of state j codes the byte at pos(u, j); positions fall by one per step inside a state's
// stream (order 1: four quarters; order 0: one interleaved stream, four bytes per step).
//   order 0 (E04 L12-57):  pos = (n & ~3) - 4u + j, step 0 is the n&3 tail (only states j < (n&3))
//   order 1 (E14 L13-86):  pos = top_j - u with top_3 = 3q + U - 1 and top_j = jq + q - 1 + rem for j < 3; states
//                          0-2 are idle during the first rem steps (state 3 codes the remainder alone, L27-31); the
//                          first byte of a quarter is coded in context 0.
struct Geo {
    int32_t n, q, rem, U;
};
template <int MODE>
__device__ __forceinline__ int32_t pos_of(const Geo& g, int32_t u, int j) {
    if (MODE == MODE_O0) return (g.n & ~3) - 4 * u + j;
    return (j == 3 ? 3 * g.q + g.U - 1 : j * g.q + g.q - 1 + g.rem) - u;
}
template <int MODE>
__device__ __forceinline__ bool active_of(const Geo& g, int32_t u, int j, int32_t pos) {
    if (u >= g.U) return false;
    if (MODE == MODE_O0) return pos < g.n;
    return j == 3 || u >= g.rem;
}
// ring address of input position pos of state j: the ring holds dense symbol indices (order 1) or bytes (order 0)
template <int MODE>
__device__ __forceinline__ int ring_addr(int32_t pos, int j) {
    if (MODE == MODE_O0) return pos & (kRingBytes - 1);
    return j * (kRingBytes / 4) + (pos & (kRingBytes / 4 - 1));
}

constexpr int kChunkSteps = 256;  // steps per input chunk = 8 batches
constexpr int kChunkBatches = kChunkSteps / kBatch;

// One input chunk = the positions steps [256c, 256c + 256) touch, as 16-byte units: order 0 one run of <= 65 units,
// order 1 four runs (one per quarter) of <= 18 units.  A lane owns up to two units.
struct ChunkRegs {
    uint4 v[2];
    int32_t pos[2];  // first position of the unit, -1 = none
    int ring[2];
};
template <int MODE>
__device__ __forceinline__ void chunk_load(const uint8_t* src, const Geo& g, int32_t c, int l, ChunkRegs& R) {
    const int32_t u0 = c * kChunkSteps, u1 = u0 + kChunkSteps - 1;
    int j;
    int32_t hi, lo;
    int first;
    if (MODE == MODE_O0) {
        j = 0;
        hi = pos_of<MODE>(g, u0, 3);
        lo = pos_of<MODE>(g, u1, 0);
        first = l;
    } else {
        j = l >> 4;
        hi = pos_of<MODE>(g, u0, j);
        lo = pos_of<MODE>(g, u1, j) - 1;  // context byte of the last step
        first = l & 15;
    }
    if (lo < 0) lo = 0;
    if (hi > g.n - 1) hi = g.n - 1;
    const int32_t alo = lo & ~15;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int32_t p = alo + 16 * (first + (MODE == MODE_O0 ? 64 : 16) * k);
        R.pos[k] = -1;
        R.v[k] = make_uint4(0, 0, 0, 0);
        if (hi >= 0 && p <= hi) {  // the arena is padded to 16 bytes, a unit may be read whole
            R.pos[k] = p;
            R.ring[k] = ring_addr<MODE>(p, j);
            R.v[k] = *reinterpret_cast<const uint4*>(src + p);
        }
    }
}
__device__ __forceinline__ uint32_t dense4(uint32_t w, const uint8_t* dmap) {
    return (uint32_t)dmap[w & 255u] | ((uint32_t)dmap[(w >> 8) & 255u] << 8) | ((uint32_t)dmap[(w >> 16) & 255u] << 16) |
           ((uint32_t)dmap[w >> 24] << 24);
}
template <int MODE>
__device__ __forceinline__ void chunk_store(const ChunkRegs& R, uint8_t* ring, const uint8_t* dmap) {
#pragma unroll
    for (int k = 0; k < 2; k++) {
        if (R.pos[k] >= 0) {
            uint4 v = R.v[k];
            if (MODE != MODE_O0) v = make_uint4(dense4(v.x, dmap), dense4(v.y, dmap), dense4(v.z, dmap), dense4(v.w, dmap));
            *reinterpret_cast<uint4*>(ring + R.ring[k]) = v;
        }
    }
}

// Helper lane l owns the run g = l >> 2 (steps 4g .. 4g+3 of the batch) of state 3 - (l & 3): lane order is then the
// reference's emission order (state 3,2,1,0 per step) within a step, runs in lane order, and the four steps of a run are
// consecutive input positions of one state.  The lane's input position falls by a constant from batch to batch and its
// steps are active for a contiguous range, so the per-batch bookkeeping is a handful of adds and compares.
struct Cursor {
    int32_t pos;      // input position of the first step of the lane's run in the batch being looked up
    uint32_t u;       // that step
    uint32_t ulo, ucnt;  // step u is active iff u - ulo < ucnt
    uint32_t ufirst;  // order 1: step at which the state sits on the first byte of its quarter (context 0)
    int ringbase;
};
template <int MODE>
__device__ __forceinline__ Cursor cursor_init(const Geo& g, int t, int j) {  // t = first step of the run within the batch
    Cursor c;
    c.u = (uint32_t)t;
    c.pos = pos_of<MODE>(g, t, j);
    if (MODE == MODE_O0) {
        c.ulo = j < (g.n & 3) ? 0u : 1u;  // step 0 is the n&3 tail
        c.ufirst = 0xFFFFFFFFu;
        c.ringbase = 0;
    } else {
        c.ulo = j == 3 ? 0u : (uint32_t)g.rem;
        c.ufirst = (uint32_t)(pos_of<MODE>(g, 0, j) - j * g.q);
        c.ringbase = j * (kRingBytes / 4);
    }
    c.ucnt = (uint32_t)g.U > c.ulo ? (uint32_t)g.U - c.ulo : 0u;
    return c;
}
template <int MODE>
__device__ __forceinline__ void cursor_next(Cursor& c) {
    c.u += kBatch;
    c.pos -= MODE == MODE_O0 ? 4 * kBatch : kBatch;
}
// ring -> packed symbols of the four steps of the cursor's run (reads are unconditional: a stale ring byte is a valid
// table index).  Order 1: the context of step k is the symbol of step k+1 (the byte before it), five ring bytes in all.
template <int MODE>
__device__ __forceinline__ void lookup_run(const Cursor& c, const uint8_t* ring, const uint2* ent, int stride, uint2 (&e)[kRun],
                                           bool (&act)[kRun]) {
    constexpr int mask = MODE == MODE_O0 ? kRingBytes - 1 : kRingBytes / 4 - 1;
    if (MODE == MODE_O0) {
        uint32_t sym[kRun];
#pragma unroll
        for (int k = 0; k < kRun; k++) sym[k] = ring[(c.pos - 4 * k) & mask];
#pragma unroll
        for (int k = 0; k < kRun; k++) {
            act[k] = c.u + (uint32_t)k - c.ulo < c.ucnt;
            e[k] = ent[sym[k]];
        }
    } else {
        uint32_t B[kRun + 1];
#pragma unroll
        for (int k = 0; k <= kRun; k++) B[k] = ring[c.ringbase + ((c.pos - k) & mask)];
#pragma unroll
        for (int k = 0; k < kRun; k++) {
            act[k] = c.u + (uint32_t)k - c.ulo < c.ucnt;
            const uint32_t ctx = c.u + (uint32_t)k == c.ufirst ? 0u : B[k + 1];
            const uint32_t idx = ctx * (uint32_t)stride + B[k];
            e[k] = ent[act[k] ? idx : 0u];  // an inactive step may hold a stale ring byte: no stray table reads
        }
    }
}

__device__ __forceinline__ uint32_t shr_bytes(uint32_t x, uint32_t nbytes) {
#ifdef HIP_SIM
    return x >> (8u * (nbytes & 3u));
#else
    return __builtin_amdgcn_alignbyte(0u, x, nbytes);
#endif
}
// Encoding#RansEncPutSymbol L46-71 on one expanded record (helper wave: the states between the logged ones)
__device__ __forceinline__ uint32_t put_symbol(uint32_t x, const uint4 P, uint32_t& nbytes) {
    const uint32_t small = P.w >> 31;             // freq < 16: the first renormalisation byte is certain (x >= L > x_max)
    const uint32_t x1 = shr_bytes(x, small);
    const uint32_t ren = (P.x - x1) >> 31;        // x1 > x_max - 1
    nbytes = small + ren;
    const uint32_t r = shr_bytes(x1, ren);
    const uint32_t q = __umulhi(r, P.y) >> ((P.w >> 24) & 31u);  // q < 2^19, cmpl_freq < 2^12
    return r + P.z + __umul24(q, P.w);
}

// DPP data movement of the helper wave's prefix sums (lanes without a source, or outside the row mask, read 0)
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ uint32_t dpp0(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    v += dpp0<0x111>(v);        // row_shr:1
    v += dpp0<0x112>(v);        // row_shr:2
    v += dpp0<0x114>(v);        // row_shr:4
    v += dpp0<0x118>(v);        // row_shr:8
    v += dpp0<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v += dpp0<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ uint32_t lane63(uint32_t v) {
#ifdef HIP_SIM
    return __shfl(v, 63);
#else
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
#endif
}

// Encoding#RansEncPutSymbol L46-71 for the 16 steps of a group, all four states (chain wave).  Lane 4q + s of a row
// (q = 0..3: the row's four DPP banks) holds the records of steps 4q .. 4q+3 of state 3 - s; the states sit in bank
// q - 1 before run q, move on with ONE row_ror:4 DPP move per four steps and are coded four steps in a row by the lane
// that holds those records: 7 dependent vector instructions per step plus the move and its hazard slot once per four
// (round 2 moved the state after every step: 9 issue slots per step instead of 7.5; scripts/ubench/step_bench.hip: 50.5 ->
// 42.0 cycles per step).  The move of run q writes only bank q of lg, so after the group lg holds, in every lane, the
// state before that lane's run: the log of a group leaves as one 4-byte LDS write per lane and the helper wave recomputes
// the three states inside the run from it (put_symbol, 64 runs per instruction).  Every lane executes every run (three
// of four compute garbage that is never read): a vector instruction costs the wave the same whatever its EXEC mask.
// The renormalisation test is arithmetic - the sign of (x_max - 1) - x selects a byte shift (v_alignbyte) - because a
// compare + select through VCC is no faster and ties up VCC.
// General variant: the group holds symbols with freq < 16, whose first renormalisation byte is certain.
#ifdef HIP_SIM
template <bool GENERAL>
__device__ __forceinline__ uint32_t group_steps(uint32_t x, const uint4 (&P)[kRun], uint32_t& lg) {
    const int l = lane_id();
    for (int q = 0; q < 4; q++) {
        const uint32_t xin = __shfl(x, (l & 48) | ((l - 4) & 15));  // row_ror:4
        if (((l >> 2) & 3) == q) lg = xin;
        uint32_t y = lg, nb;
        for (int k = 0; k < kRun; k++) {
            uint4 R = P[k];
            if (!GENERAL) R.w &= 0x7FFFFFFFu;
            y = put_symbol(y, R, nb);
        }
        x = y;
    }
    return x;
}
#else
// A DPP operand read straight behind the VALU write of the same register returns the OLD value on gfx950
// (scripts/ubench/rot_bench.hip), hence the s_nop in front of every move.  The shift count of the quotient is byte 3 of
// the record's last word (SDWA; the shift reads its low five bits), the 24-bit multiply ignores that byte.
#define QSTEP(IN, K)                                                                                                 \
    "v_sub_u32 %[t], %[a" #K "], %[" #IN "]\n\t"                                                                     \
    "v_lshrrev_b32 %[t], 31, %[t]\n\t"                                                                               \
    "v_alignbyte_b32 %[r], 0, %[" #IN "], %[t]\n\t"                                                                  \
    "v_mul_hi_u32 %[m], %[r], %[b" #K "]\n\t"                                                                        \
    "v_lshrrev_b32_sdwa %[m], %[d" #K "], %[m] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t" \
    "v_mul_u32_u24 %[m], %[m], %[d" #K "]\n\t"                                                                       \
    "v_add3_u32 %[x], %[r], %[c" #K "], %[m]\n\t"
#define QSTEP_G(IN, K)                             \
    "v_lshrrev_b32 %[t], 31, %[d" #K "]\n\t"       \
    "v_alignbyte_b32 %[p], 0, %[" #IN "], %[t]\n\t" QSTEP(p, K)
#define QRUN(STEP, BK)                                                               \
    "s_nop 0\n\t"                                                                    \
    "v_mov_b32_dpp %[lg], %[x] row_ror:4 row_mask:0xf bank_mask:" #BK "\n\t" STEP(lg, 0) STEP(x, 1) STEP(x, 2) STEP(x, 3)
#define QGROUP(STEP) QRUN(STEP, 0x1) QRUN(STEP, 0x2) QRUN(STEP, 0x4) QRUN(STEP, 0x8)
// Both variants in ONE asm statement, the choice made by a scalar branch inside it (two statements under a C++ branch
// leave the compiler free to keep the state in different registers on the two paths and to move registers at every join).
// gs = the ballot of the lanes whose batch flags the group as general: a scalar register pair by construction.
__device__ __forceinline__ uint32_t group_steps_both(uint32_t x, const uint4 (&P)[kRun], unsigned long long gs, uint32_t& lg) {
    uint32_t t, r, m, p;
    asm volatile("s_cmp_eq_u64 %[gs], 0\n\t"
                 "s_cbranch_scc1 .LgrpF_%=\n\t" QGROUP(QSTEP_G) "s_branch .LgrpD_%=\n"
                 ".LgrpF_%=:\n\t" QGROUP(QSTEP) ".LgrpD_%=:\n\t"
                 : [x] "+v"(x), [lg] "+v"(lg), [t] "=&v"(t), [r] "=&v"(r), [m] "=&v"(m), [p] "=&v"(p)
                 : [a0] "v"(P[0].x), [b0] "v"(P[0].y), [c0] "v"(P[0].z), [d0] "v"(P[0].w), [a1] "v"(P[1].x), [b1] "v"(P[1].y),
                   [c1] "v"(P[1].z), [d1] "v"(P[1].w), [a2] "v"(P[2].x), [b2] "v"(P[2].y), [c2] "v"(P[2].z), [d2] "v"(P[2].w),
                   [a3] "v"(P[3].x), [b3] "v"(P[3].y), [c3] "v"(P[3].z), [d3] "v"(P[3].w), [gs] "s"(gs)
                 : "scc");
    return x;
}
#endif

// Phase 3 of a slot, both of its waves.  The aux wave STAGES: it fetches the symbol / context bytes (through an LDS
// ring refilled with 16-byte loads a whole chunk of 4 batches before it is needed, so that no batch waits for global
// memory), looks the symbols up, expands them into the staging area and publishes the batch to the chain wave.  The
// helper wave EMITS: once the chain wave has coded a batch it turns the logged states into bytes (re-reading the batch's
// records from the staging area), written downwards from pend, and releases the batch's staging buffer.  Batches carry
// absolute numbers (base + b) that index the four staging / log buffers, so that the chain wave needs no notion of
// blocks: it codes whatever is staged.  Returns (helper wave) the payload length including the 16 state bytes, or a
// negative code; the aux wave returns 0.
// In-kernel cycle stamps exist only in the profiling build (make prof -> libcramgpu_prof.so): s_memtime shares the
// LDS counter and, being out of order, makes the compiler wait for ALL outstanding LDS reads at every use.
#ifdef CRAMGPU_PROF_BUILD
#define PROF_ON(p) ((p) != nullptr)
#else
#define PROF_ON(p) false
#endif
template <int MODE>
__device__ int32_t encode_stream(const uint8_t* src, int32_t n_, const uint2* ent, int stride_, uint8_t* pend, int32_t room_,
                                 unsigned long long* prof_) {
    unsigned long long* prof = PROF_ON(prof_) ? prof_ : nullptr;
    EncLds* L = &lds;
    const int32_t n = UNIFORM(n_), stride = UNIFORM(stride_), room = UNIFORM(room_);  // arrive in vector registers
    const int l = lane_id();
    const int js = l & 3, j = 3 - js;        // lane -> state: lane order = emission order within a step
    Geo g;
    g.n = n;
    g.q = n >> 2;
    g.rem = MODE == MODE_O0 ? 0 : n - 4 * g.q;
    g.U = MODE == MODE_O0 ? g.q + 1 : g.q + g.rem;
    const int32_t nbatch = (g.U + kBatch - 1) / kBatch;
    const uint32_t base = (uint32_t)UNIFORM(L->base);
    if (wave_id() == 0) {
        // ---------------- the stager
        const uint4 ident = make_uint4(0x7FFFFFFFu, 0u, 0u, 0u);  // identity step for inactive (step, state) pairs (x < 2^31)
        uint2 e2[kRun];     // looked-up symbols of the next batch
        bool act2[kRun];
        Cursor cur = cursor_init<MODE>(g, kRun * (l >> 2), j);
        ChunkRegs CR;
        // expand the looked-up run into the staging buffer of batch b (relative number), general flags of its four
        // groups (bit 4: the chain wave starts a new block with this batch), and let the chain wave have the batch
        auto stage_run = [&](int32_t b) {
            const int sb = (int)((base + (uint32_t)b) & (kStageBufs - 1));
            uint32_t any = 0;
#pragma unroll
            for (int k = 0; k < kRun; k++) {
                const uint4 A = act2[k] ? expand_symbol(e2[k]) : ident;
                L->p3.stage[sb][l * kRun + k] = A;
                any |= A.w;
            }
            const unsigned long long sm = __ballot((int32_t)any < 0);
            if (l == 0)
                L->p3.gflag[sb] = ((sm & 0xFFFFull) ? 1u : 0u) | ((sm & 0xFFFF0000ull) ? 2u : 0u) | ((sm & 0xFFFF00000000ull) ? 4u : 0u) |
                                  ((sm >> 48) ? 8u : 0u) | (b == 0 ? 16u : 0u);
            WSYNC();
            if (l == 0) st_flag(&L->staged, base + (uint32_t)b + 1u);
        };
        chunk_load<MODE>(src, g, 0, l, CR);
        chunk_store<MODE>(CR, L->p3.ring, L->dmap);
        chunk_load<MODE>(src, g, 1, l, CR);
        WSYNC();
        lookup_run<MODE>(cur, L->p3.ring, ent, stride, e2, act2);
        cursor_next<MODE>(cur);
        for (int32_t b = 0; b < nbatch; b++) {
            // the buffer held batch b - 4: coded, and turned into bytes by the helper wave
            wait_flag_ge(&L->emitted, base + (uint32_t)b - (uint32_t)(kStageBufs - 1));
            stage_run(b);
            // input ring: chunk c (kChunkBatches batches) is needed by the lookup of its first batch, which happens one
            // batch earlier; it is written another two batches earlier from the registers that were loaded a chunk of
            // batches before that, and the next chunk's loads are issued
            if (((b - 2) & (kChunkBatches - 1)) == 0 && b >= 2) {
                chunk_store<MODE>(CR, L->p3.ring, L->dmap);
                chunk_load<MODE>(src, g, (b - 2) / kChunkBatches + 2, l, CR);
                WSYNC();
            }
            lookup_run<MODE>(cur, L->p3.ring, ent, stride, e2, act2);  // batch b + 1 (the table reads land while this wave waits)
            cursor_next<MODE>(cur);
        }
        return 0;
    }
    // ---------------- the emitter
    const int32_t limit = room - 16;
    const uint32_t m1 = js >= 1 ? ~0u : 0u, m2 = js >= 2 ? ~0u : 0u;
    int32_t run = 0;
    bool failed = false;
    unsigned long long busy = 0;
    for (int32_t b = 0; b < nbatch; b++) {
        // bytes of batch b in the reference's emission order: step, then state 3,2,1,0 = run, step of the run, lane & 3
        wait_flag_ge(&L->done, base + (uint32_t)b + 1u);  // the chain wave has logged the batch
        const long long tb = prof ? clock64() : 0;
        const int eb = (int)((base + (uint32_t)b) & (kStageBufs - 1));
        uint32_t xv = L->p3.xs[eb][l];
        uint4 R[kRun];
#pragma unroll
        for (int k = 0; k < kRun; k++) R[k] = L->p3.stage[eb][l * kRun + k];
        uint32_t lo16[kRun], nbp = 0;  // byte counts of the four steps, one per byte of nbp
#pragma unroll
        for (int k = 0; k < kRun; k++) {
            uint32_t nb;
            lo16[k] = xv;
            xv = put_symbol(xv, R[k], nb);
            nbp |= nb << (8 * k);
        }
        WSYNC();
        if (l == 0) st_flag(&L->emitted, base + (uint32_t)b + 1u);  // log and records are in registers: the buffer is free
        // positions: runs in lane order; inside a run step by step, inside a step the quad's lanes in order
        uint32_t S = nbp;  // inclusive scan over the quad, all four steps at once (a count is at most 2)
        S += dpp0<0x90>(S) & m1;   // quad_perm:[0,0,1,2]
        S += dpp0<0x44>(S) & m2;   // quad_perm:[0,1,0,1]
        const uint32_t Qp = dpp0<0xFF>(S);  // quad totals per step
        const uint32_t Bp = S - nbp;        // bytes of the lower lanes of the quad per step
        const uint32_t I = wave_incl_scan((nbp * 0x01010101u) >> 24);
        const int32_t tot = (int32_t)lane63(I);
        uint32_t acc = (uint32_t)run + dpp0<0xFF>(I) - ((Qp * 0x01010101u) >> 24);  // bytes in front of the quad's run
        if (run + tot > limit) failed = true;
        if (!failed) {
#pragma unroll
            for (int k = 0; k < kRun; k++) {
                const uint32_t nb = (nbp >> (8 * k)) & 0xFFu;
                const int32_t idx = (int32_t)(acc + ((Bp >> (8 * k)) & 0xFFu));
                if (nb >= 1) pend[-1 - idx] = (uint8_t)lo16[k];
                if (nb == 2) pend[-2 - idx] = (uint8_t)(lo16[k] >> 8);
                acc += (Qp >> (8 * k)) & 0xFFu;
            }
            run += tot;
        }
        if (prof) busy += (unsigned long long)(clock64() - tb);
    }
    // the chain wave has coded all nbatch batches: the next block's batches follow on
    if (l == 0) L->base = base + (uint32_t)nbatch;
    if (prof && l == 0) {
        prof[2] = busy;
        prof[4] = (unsigned long long)nbatch;
#ifdef CRAMGPU_PROF_BUILD
        prof[3] = ((unsigned long long)L->hwid_chain << 32) | __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
        prof[5] = L->idle_chain;
        L->idle_chain = 0;
#endif
    }
    if (failed) return CRAMGPU_E_INSUFFICIENT;
    // E04 L49-53 / E14 L78-82: big-endian rans3..rans0, then Utils#reverse => stream starts LE rans0..rans3
    if (l < 4) {
        const uint32_t xf = L->p3.fin[l];
        uint8_t* p = pend - run - 16 + 4 * l;
        p[0] = (uint8_t)xf;
        p[1] = (uint8_t)(xf >> 8);
        p[2] = (uint8_t)(xf >> 16);
        p[3] = (uint8_t)(xf >> 24);
    }
    WSYNC();
    return run + 16;
}

// The chain wave: row s (lanes 16s .. 16s+15) codes whatever slot s has staged, batch by batch, and never waits for a
// slot: a row whose slot has nothing staged reads identity records (its states stand still) while the other rows go on.
// It ends when every slot has run out of blocks.
__device__ void chain_loop() {
    const int l = lane_id();
    const int cl = l & 15;
    EncLds* S = &lds4[l >> 4];
    uint32_t x = RANS_BYTE_L, lg = 0;
    uint32_t c = 0;  // batches of my slot coded so far
#ifndef HIP_SIM
    __builtin_amdgcn_s_setprio(3);  // the chain is the critical path of all four blocks
#endif
#ifdef CRAMGPU_PROF_BUILD
    if (cl == 0) S->hwid_chain = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
#endif
    // what the next batch-time needs is fetched during the current one: the slot's staged counter during group 1, the
    // flags and the first group's records during group 2 (addresses depend on the counter), so that no batch begins by
    // waiting for LDS.  A row whose slot had nothing staged at that moment stands still for one batch-time.
    auto row_view = [&](uint32_t st) {
#ifdef HIP_SIM
        st = __shfl(st, l & 48);  // the harness's lanes read at different times: the row takes its first lane's view
#endif
        return st;
    };
    uint32_t st = row_view(ld_flag(&S->staged));
    bool act = (int32_t)(st - c) > 0, was = false;
    int buf = (int)(c & (kStageBufs - 1));
    uint32_t f = act ? S->p3.gflag[buf] : 0u;
    const uint4* rec = act ? &S->p3.stage[buf][cl * kRun] : &S->ident[0];
    uint4 Pc[kRun];
#pragma unroll
    for (int k = 0; k < kRun; k++) Pc[k] = rec[k];
    for (;;) {
        if (__ballot(act) == 0) {  // nothing staged anywhere: poll
            const bool gone = ld_flag(&S->exitf) != 0;
            st = row_view(ld_flag(&S->staged));
            act = (int32_t)(st - c) > 0;
            if (__ballot(!gone || act) == 0) break;  // every slot has run out of blocks
            if (__ballot(act) == 0) {
#ifdef HIP_SIM
                std::this_thread::yield();
#else
                __builtin_amdgcn_s_sleep(4);
#endif
                continue;
            }
            f = act ? S->p3.gflag[buf] : 0u;
            rec = act ? &S->p3.stage[buf][cl * kRun] : &S->ident[0];
#pragma unroll
            for (int k = 0; k < kRun; k++) Pc[k] = rec[k];
        }
        // a row that has just run dry looks once more (its helper may have been a moment late: without this the row would
        // stand still for a whole batch-time); rows that are dry for longer (histogram phases) do not hold the others up
        if (__ballot(!act && was) != 0) {
            st = row_view(ld_flag(&S->staged));
            const bool now = !act && was && (int32_t)(st - c) > 0;
            if (now) {
                act = true;
                f = S->p3.gflag[buf];
                rec = &S->p3.stage[buf][cl * kRun];
#pragma unroll
                for (int k = 0; k < kRun; k++) Pc[k] = rec[k];
            }
        }
        was = act;
#ifdef CRAMGPU_PROF_BUILD
        if (!act && cl == 0) S->idle_chain = S->idle_chain + 1;
#endif
        if (f & 16u) x = RANS_BYTE_L;  // the batch opens a block
        // a group takes the general variant when any row's batch flags it: the ballot is the scalar the branch tests
        unsigned long long gs[kGroups];
#pragma unroll
        for (int G = 0; G < kGroups; G++) gs[G] = __ballot(((f >> G) & 1u) != 0);
        // the run's four records: 64 contiguous bytes per lane, groups 16 lanes apart; an idle row reads the identity
        // and writes into the sink (p3 is its slot's histogram space during phases 1 and 2)
        const int gstep = act ? kGroup * kRun : 0;
        uint32_t* logp = act ? &S->p3.xs[buf][cl] : &S->sink[cl];
        const int lstep = act ? kGroup : 0;
        uint32_t st2 = 0, f2 = 0;
        bool act2 = false;
        int buf2 = 0;
        const uint4* rec2 = rec;
#pragma unroll
        for (int G = 0; G < kGroups; G++) {
            uint4 Pn[kRun];  // records of the next group land while this one runs
            if (G + 1 < kGroups) {
#pragma unroll
                for (int k = 0; k < kRun; k++) Pn[k] = rec[gstep * (G + 1) + k];
            } else {
#pragma unroll
                for (int k = 0; k < kRun; k++) Pn[k] = rec2[k];
            }
#ifdef HIP_SIM
            x = gs[G] ? group_steps<true>(x, Pc, lg) : group_steps<false>(x, Pc, lg);
#else
            x = group_steps_both(x, Pc, gs[G], lg);
#endif
            logp[lstep * G] = lg;
            if (G == 1) st2 = row_view(ld_flag(&S->staged));
            if (G == 2) {
                const uint32_t c2 = c + (act ? 1u : 0u);
                act2 = (int32_t)(st2 - c2) > 0;
                buf2 = (int)(c2 & (kStageBufs - 1));
                f2 = act2 ? S->p3.gflag[buf2] : 0u;
                rec2 = act2 ? &S->p3.stage[buf2][cl * kRun] : &S->ident[0];
            }
#pragma unroll
            for (int k = 0; k < kRun; k++) Pc[k] = Pn[k];
        }
        if (cl >= 12) *(act ? &S->p3.fin[3 - (cl & 3)] : &S->sink[cl]) = x;  // bank 3 holds the states after a group
        c += act ? 1u : 0u;
#ifdef HIP_SIM
        __builtin_amdgcn_wave_barrier();
#endif
        st_flag(&S->done, c);
        st = st2, act = act2, buf = buf2, f = f2, rec = rec2;
    }
#ifndef HIP_SIM
    __builtin_amdgcn_s_setprio(0);
#endif
}

// Each mode's coding loop is a non-inlined function so that it gets its own register allocation.
struct Blk {
    const uint8_t* src;
    uint8_t* slot;
    uint8_t* gs;
    const uint32_t* rcp_tab;
    unsigned long long* prof;  // 8 words of this block in RansArgs::prof, or NULL
    int32_t n, cap;
};
struct Prep {
    int32_t mode, stride, ent_off, tab_len, status;  // status 0 = tables ready, else a negative code
};

__device__ __attribute__((noinline)) int32_t stream_o0(Blk B, int32_t tab_len) {
    const uint2* ent = reinterpret_cast<const uint2*>(lds.tab + 1024);
    return encode_stream<MODE_O0>(B.src, B.n, ent, 256, B.slot + B.cap, B.cap - 9 - tab_len, B.prof);
}
__device__ __attribute__((noinline)) int32_t stream_o1_lds(Blk B, int32_t tab_len, int32_t stride, int32_t ent_off) {
    const uint2* ent = reinterpret_cast<const uint2*>(lds.tab + ent_off);
    return encode_stream<MODE_O1_LDS>(B.src, B.n, ent, stride, B.slot + B.cap, B.cap - 9 - tab_len, B.prof);
}
__device__ __attribute__((noinline)) int32_t stream_o1_global(Blk B, int32_t tab_len, int32_t stride) {
    const uint2* ent = reinterpret_cast<const uint2*>(B.gs + kGsEntOff);
    return encode_stream<MODE_O1_GLOBAL>(B.src, B.n, ent, stride, B.slot + B.cap, B.cap - 9 - tab_len, B.prof);
}

// ---- phases 1 and 2: both waves count, the helper wave alone builds the tables (its result is the one that counts) ----
__device__ __attribute__((noinline)) Prep prep_o0(Blk B) {
    const int l = lane_id();
    const bool helper = wave_id() == 1;
    EncFixedLds* L = &lds;
    uint8_t* tab = lds.tab;
    uint32_t* cnt0 = reinterpret_cast<uint32_t*>(tab);    // 256 words
    uint2* ent0 = reinterpret_cast<uint2*>(tab + 1024);  // 256 x 8 B
    uint32_t* rep0 = reinterpret_cast<uint32_t*>(lds_big());  // the coding loop's buffers are free until phase 3
    hist_o0(B.src, B.n, cnt0, rep0);
    Prep r{MODE_O0, 256, 1024, 0, 0};
    if (!helper) return r;
    for (int i = l; i < 256; i += 64) L->syml[i] = (uint8_t)i;
    WSYNC();
    r.tab_len = tables_o0(cnt0, ent0, B.n, L, B.slot + 9, B.rcp_tab, B.cap - 9 - 16);
    if (9 + r.tab_len + 16 > B.cap) r.status = CRAMGPU_E_INSUFFICIENT;
    return r;
}

// Order 1: packed symbols over the dense alphabet in the LDS table region and pair counts in the spare area when they
// fit, else both in the workgroup's global scratch (same layout, same code: hist_o1 / tables_o1 take generic pointers).
// The alphabet of a long block is taken from its first kSampleBytes: the pair counts only need to know which symbols
// occur (the order-0 counts themselves are not part of an order-1 stream), so the order-0 pass over the whole block -
// a third of the input traffic and 40 % of phase 1 - shrinks to a sample.  Symbols the sample did not see count into an
// overflow row / column of the pair table; if anything landed there the block is counted again with its full alphabet.
constexpr int32_t kSampleBytes = 64 * 1024;
__device__ __attribute__((noinline)) Prep prep_o1(Blk B) {
    const int l = lane_id();
    const bool helper = wave_id() == 1;
    EncFixedLds* L = &lds;
    uint8_t* tab = lds.tab;
    uint32_t* cnt0 = reinterpret_cast<uint32_t*>(tab);
    uint32_t* rep0 = reinterpret_cast<uint32_t*>(lds_big());
    uint8_t* big = lds_big();
    Prep r{MODE_O1_LDS, 0, 0, 0, 0};
    uint32_t* cnt;
    uint2* ent;
    int ns, stride;
    bool in_lds;
    for (int attempt = B.n > kSampleBytes ? 0 : 1;; attempt++) {
        hist_o0(B.src, attempt == 0 ? kSampleBytes : B.n, cnt0, rep0);
        // dense alphabet = present symbols plus 0 (context 0 always exists); both waves derive its size, one writes the maps
        uint32_t present = 0, mine = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int jj = l * 4 + k;
            bool p = cnt0[jj] != 0 || jj == 0;
            present |= (p ? 1u : 0u) << k;
            mine += p ? 1u : 0u;
        }
        uint32_t nsym;
        uint32_t di = wave_excl_scan(mine, nsym);
        const bool ovf = attempt == 0 && nsym < 256;  // a sample that holds all 256 values has nothing left to miss
        if (helper) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int jj = l * 4 + k;
                if (present & (1u << k)) {
                    L->dmap[jj] = (uint8_t)di;
                    L->syml[di] = (uint8_t)jj;
                    di++;
                } else {
                    L->dmap[jj] = ovf ? (uint8_t)nsym : 0;
                }
            }
            if (ovf && l == 0) L->syml[nsym] = 0;  // never serialised: its counts are zero or the block is counted again
        }
        slot_sync();  // maps written, cnt0 read by both waves: the table region can be reused
        ns = (int)nsym + (ovf ? 1 : 0);
        stride = ns | 1;
        const int cells = ns * stride;
        // in LDS: packed symbols (8 B per cell, needed by the coding loop) in the table region, pair counts (4 B per cell,
        // dead once the symbols exist) in the spare area
        in_lds = 8 * cells + 16 <= kTabBytes && 4 * cells + 16 <= kSpareBytes;
        r.mode = in_lds ? MODE_O1_LDS : MODE_O1_GLOBAL;
        r.stride = stride;
        uint32_t* rep = nullptr;
        int R = 1;
        const int rstride = cells | 1;
        if (in_lds) {
            cnt = reinterpret_cast<uint32_t*>(big);
            ent = reinterpret_cast<uint2*>(tab);
            const int rep_off = (8 * cells + 15) & ~15;
            while (R < 32 && rep_off + 2 * R * rstride * 4 <= kTabBytes) R <<= 1;
            rep = reinterpret_cast<uint32_t*>(tab + rep_off);
            r.ent_off = 0;
        } else {
            cnt = reinterpret_cast<uint32_t*>(B.gs);
            ent = reinterpret_cast<uint2*>(B.gs + kGsEntOff);
        }
        const bool packed = !in_lds && B.n < 65536 && 2 * cells + 8 <= kBigBytes;
        if (packed) {
            hist_o1_packed16(B.src, B.n, cnt, stride, L->dmap, reinterpret_cast<uint32_t*>(big), cells);
        } else {
            for (int i = stid(); i < cells; i += kWgThreads) cnt[i] = 0;
            if (!in_lds) __threadfence();  // the scratch was another block's a moment ago: write back / invalidate around each phase
            hist_o1(B.src, B.n, cnt, stride, L->dmap, rep, R, rstride, cells);  // begins with a slot barrier
        }
        if (!in_lds) __threadfence();
        slot_sync();
        if (!ovf) break;
        // did a symbol outside the sample's alphabet turn up?  (its row and its column of the pair table)
        if (helper) {
            uint32_t stray = 0;
            for (int k = l; k < ns; k += 64) stray += cnt[(ns - 1) * stride + k] + cnt[k * stride + (ns - 1)];
            stray = wave_sum(stray);
            if (l == 0) L->status = (int32_t)(stray != 0);
        }
        slot_sync();
        const bool again = UNIFORM(L->status) != 0;
        slot_sync();  // both waves have read the verdict before anything overwrites it
        if (!again) break;
    }
    if (!helper) return r;
    if (in_lds) {
        r.tab_len = tables_o1(cnt, ent, ns, ns, stride, L, B.slot + 9, B.rcp_tab, B.cap - 9 - 16);
    } else {
        const int G = min(64, kBigBytes / (stride * 4));  // >= 32 rows of the widest alphabet
        r.tab_len = tables_o1_staged(cnt, ent, ns, stride, L, reinterpret_cast<uint32_t*>(big), G, B.slot + 9, B.rcp_tab,
                                     B.cap - 9 - 16);
    }
    if (!in_lds) __threadfence();
    WSYNC();
    if (r.tab_len < 0)
        r.status = r.tab_len;
    else
        r.status = 9 + r.tab_len + 16 > B.cap ? CRAMGPU_E_INSUFFICIENT : 0;
    return r;
}

__device__ void encode_block(const RansArgs& a, int b, uint8_t* gs) {
    EncFixedLds* L = &lds;
    const int l = lane_id();
    const bool helper = wave_id() == 1;
    const cramgpu_block d = a.desc[b];
    const int32_t n = UNIFORM(d.in_len);
    int order = UNIFORM(d.order);
    uint8_t* slot = a.out + d.out_off;
    const int32_t cap = UNIFORM(d.out_cap);
    int32_t result;
    int64_t pos = d.out_off;
    if (order != 0 && order != 1) {
        result = CRAMGPU_E_ORDER;
    } else if (d.in_off < 0 || d.out_off < 0 || (d.in_off & (CRAMGPU_ARENA_ALIGN - 1)) || (d.out_off & (CRAMGPU_ARENA_ALIGN - 1))) {
        result = CRAMGPU_E_ARG;  // the host checks this where it sees the descriptors; device-built tables are checked here
    } else if (n == 0) {
        result = 0;  // RANS#compress L61-62: EMPTY_BUFFER
    } else if (n < 0 || cap < 9 + 2 + 16) {
        result = n < 0 ? CRAMGPU_E_ARG : CRAMGPU_E_INSUFFICIENT;
    } else {
        if (n < 4) order = 0;  // RANS#compress L63-64
        uint8_t* tabdst = slot + 9;
        uint8_t* pend = slot + cap;
        Blk B{a.in + d.in_off, slot, gs, a.rcp_tab, a.prof ? a.prof + 8 * (size_t)b : nullptr, n, cap};
        const long long t0 = PROF_ON(a.prof) ? clock64() : 0;
        {  // phases 1 and 2: both waves of the slot count, the helper wave builds the tables
            Prep p;
            if (order == 0) {
                p = prep_o0(B);
            } else {
                p = prep_o1(B);
            }
            if (helper && l == 0) {
                L->mode = p.mode;
                L->stride = p.stride;
                L->ent_off = p.ent_off;
                L->tab_len = p.tab_len;
                L->status = p.status;
            }
        }
        slot_sync();
        const int32_t mode = UNIFORM(L->mode), stride = UNIFORM(L->stride), ent_off = UNIFORM(L->ent_off),
                      tab_len = UNIFORM(L->tab_len);
        int32_t pay_len = UNIFORM(L->status);
        const long long t1 = PROF_ON(a.prof) ? clock64() : 0;
#ifdef CRAMGPU_PROF_BUILD
        const long long w1 = PROF_ON(a.prof) ? (long long)__builtin_amdgcn_s_memrealtime() : 0;  // constant 100 MHz
#endif
        if (pay_len == 0) {
            if (mode == MODE_O0)
                pay_len = stream_o0(B, tab_len);
            else if (mode == MODE_O1_LDS)
                pay_len = stream_o1_lds(B, tab_len, stride, ent_off);
            else
                pay_len = stream_o1_global(B, tab_len, stride);
        }
        if (!helper) return;  // the epilogue belongs to the helper wave (it holds the payload length)
        if (PROF_ON(a.prof) && l == 0) {
            B.prof[0] = (unsigned long long)(t1 - t0);
            B.prof[1] = (unsigned long long)(clock64() - t1);
#ifdef CRAMGPU_PROF_BUILD
            B.prof[6] = (unsigned long long)((long long)__builtin_amdgcn_s_memrealtime() - w1);  // phase 3 in 10 ns units
            B.prof[7] = (unsigned long long)__builtin_amdgcn_s_memrealtime();                    // end of the block
#endif
        }
        if (pay_len < 0) {
            result = pay_len;
        } else {
            // move the table in front of the payload (dst >= src: copy from the top down), then the 9-byte header
            WSYNC();
            uint8_t* dst = pend - pay_len - tab_len;
            if (dst != tabdst) {
                for (int32_t hi = tab_len; hi > 0; hi -= 64) {
                    int32_t k = hi - 1 - l;
                    uint8_t v = 0;
                    if (k >= 0) v = tabdst[k];
                    __threadfence_block();
                    WSYNC();
                    if (k >= 0) dst[k] = v;
                    __threadfence_block();
                    WSYNC();
                }
            }
            uint8_t* h = dst - 9;
            const uint32_t comp = (uint32_t)(tab_len + pay_len);
            if (l == 0) h[0] = (uint8_t)order;  // RANS#finalizeCompressed L132-142
            if (l >= 1 && l < 5) h[l] = (uint8_t)(comp >> (8 * (l - 1)));
            if (l >= 5 && l < 9) h[l] = (uint8_t)((uint32_t)n >> (8 * (l - 5)));
            result = 9 + tab_len + pay_len;
            pos = (int64_t)(h - a.out);
        }
    }
    if (helper && l == 0) {
        a.out_len[b] = result;
        a.out_pos[b] = pos;
    }
}

__global__ __launch_bounds__(kBlockThreads) void rans_encode_kernel(RansArgs a) {
    const int pw = phys_wave();
    const bool chain = pw == 0, spare = pw == 4 || pw == 8 || pw == 11;
    if (!chain && !spare) {  // the slot's hand-over words and identity records, before anybody looks at them
        EncFixedLds* L = &lds;
        const int t = stid();
        if (t == 64) {
            L->sync_cnt = 0;
            L->staged = 0;
            L->done = 0;
            L->emitted = 0;
            L->base = 0;
            L->exitf = 0;
            L->idle_chain = 0;
        }
        if (t >= 64 && t < 68) L->ident[t - 64] = make_uint4(0x7FFFFFFFu, 0u, 0u, 0u);
    }
    __syncthreads();
    if (spare) return;
    if (chain) {
        chain_loop();
        return;
    }
    EncFixedLds* L = &lds;
    uint8_t* gs = a.gscratch + ((size_t)blockIdx.x * kSlots + (size_t)slot_idx()) * ENC_GSCRATCH_BYTES;
    for (;;) {
        if (stid() == 64) L->ticket = atomicAdd(a.ticket, 1u);
        slot_sync();
        const uint32_t t = (uint32_t)UNIFORM(L->ticket);
        slot_sync();  // both waves hold the ticket before the helper can draw the next one
        if (t >= (uint32_t)a.n_blocks) break;  // every slot reaches this: the queue only grows
        encode_block(a, a.perm ? a.perm[t] : (int)t, gs);  // no permutation: blocks in table order
    }
    if (stid() == 64) st_flag(&L->exitf, 1u);
}

// Encoding#RansEncSymbolInit L31-36: rcp_freq = (int)(((1L << (shift + 31)) + freq - 1) / freq) for freq >= 2
__global__ void rcp_init_kernel(uint32_t* tab) {
    uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= 4096) return;
    uint32_t v = ~0u;
    if (f >= 2) {
        uint32_t shift = 0;
        while ((unsigned long long)f > (1ull << shift)) shift++;
        v = (uint32_t)(((1ull << (shift + 31)) + f - 1) / f);
    }
    tab[f] = v;
}

}  // namespace

size_t rans_encode_fixed_lds() { return 0; }  // the encoder's LDS is static

hipError_t launch_rans_encode(const RansArgs& a, int n_groups, size_t lds_bytes, hipStream_t s) {
    (void)lds_bytes;
    // n_groups = slots (blocks in flight) the caller sized the global scratch for, four to a workgroup
    hipLaunchKernelGGL(rans_encode_kernel, dim3((n_groups + kSlots - 1) / kSlots), dim3(kBlockThreads), 0, s, a);
    return hipGetLastError();
}

// Longest-processing-time order of up to kLptMax blocks on the device (the host does this with std::stable_sort where it
// sees the descriptors): bitonic sort of (0x7FFFFFFF - length, index) keys in LDS by one workgroup.
namespace {
constexpr int kLptMax = 4096;
__global__ __launch_bounds__(1024) void lpt_perm_kernel(const cramgpu_block* desc, int32_t n, int32_t* perm) {
    __shared__ unsigned long long key[kLptMax];
    for (int i = threadIdx.x; i < kLptMax; i += blockDim.x) {
        unsigned long long k = ~0ull;  // padding sorts last
        if (i < n) {
            const int32_t len = desc[i].in_len;
            k = ((unsigned long long)(0x7FFFFFFFu - (uint32_t)(len > 0 ? len : 0)) << 32) | (uint32_t)i;
        }
        key[i] = k;
    }
    __syncthreads();
    for (int size = 2; size <= kLptMax; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < kLptMax / 2; t += blockDim.x) {
                const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                const bool up = (lo & size) == 0;
                const unsigned long long a = key[lo], b = key[hi];
                if ((a > b) == up) {
                    key[lo] = b;
                    key[hi] = a;
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < n; i += blockDim.x) perm[i] = (int32_t)(uint32_t)key[i];
}
}  // namespace

hipError_t launch_lpt_perm(const cramgpu_block* d_desc, int32_t n, int32_t* d_perm, hipStream_t s) {
    if (n > kLptMax) return hipErrorInvalidValue;
    hipLaunchKernelGGL(lpt_perm_kernel, dim3(1), dim3(1024), 0, s, d_desc, n, d_perm);
    return hipGetLastError();
}

hipError_t launch_rcp_init(uint32_t* d_rcp_tab, hipStream_t s) {
    hipLaunchKernelGGL(rcp_init_kernel, dim3(16), dim3(256), 0, s, d_rcp_tab);
    return hipGetLastError();
}
