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
// states and (b) everything that does not depend on the states.  The kernel is a resident grid of single-wave
// workgroups that pull blocks (longest first) from a ticket counter.  For its block a wave
//   1. histograms the input with all 64 lanes (LDS counters, lane-replicated for small alphabets),
//   2. normalises, serialises the frequency table and builds packed encoder symbols, one table row per lane,
//   3. encodes in batches of 16 steps: all 64 lanes fetch symbol/context bytes and expand the 64 encoder symbols
//      of the batch into an LDS staging area; lanes 0-3 run the four state chains reading that area at fixed
//      offsets and log the pre-renormalisation states; then all 64 lanes turn the logged states into output
//      bytes, placed with a ballot/popcount prefix in the reference's emission order (state 3,2,1,0 per step).
//      Bytes are written from the end of the block's slot downwards, which realises Utils#reverse for free.
// Order-1 tables live in LDS over the compacted alphabet (present symbols + 0) when n*(n|1)*12 B fits the table
// region, else in a per-wave global scratch over the full 256x256 alphabet.
#include <climits>

#include "internal.h"

namespace {

constexpr int kSteps = 16;  // steps per batch: 16 steps x 4 states = 64 lanes

struct EncFixedLds {
    union {
        struct {                    // phase 2 (tables)
            uint32_t rowT[256];     // order-1 row totals (0 = context absent)
            uint32_t rowoff[256];   // serialised length of each row
            uint32_t rowhdr[256];   // context-level header of each row: count | byte0 << 8 | byte1 << 16
        } p2;
        struct {                    // phase 3 (coding loop)
            uint4 stageA[2][64];    // x_max, x_max<<8 (saturated), rcp_freq, bias     (double buffered)
            uint2 stageB[2][64];    // cmpl_freq, rcp_shift-32
            uint32_t xs[kSteps][64];  // state of every lane before each step of the current batch
        } p3;
    };
    uint8_t dmap[256];    // byte value -> dense index
    uint8_t syml[256];    // dense index -> byte value
};

__device__ __forceinline__ int lane_id() { return (int)threadIdx.x; }

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

// expand a packed symbol into the staged form the state lanes consume
__device__ __forceinline__ void expand_symbol(uint2 e, uint4& A, uint2& B) {
    uint32_t f = e.y & 4095u;
    A.x = f << 19;                              // x_max = ((L >> 12) << 8) * freq
    A.y = f < 16 ? (f << 27) : 0xFFFFFFFFu;     // second renormalisation byte iff x >= x_max << 8 (x < 2^31)
    A.z = e.x;
    A.w = (e.y >> 12) & 8191u;
    B.x = 4096u - f;
    B.y = e.y >> 25;
}

// ---------------------------------------------------------------------------------------------------------
// phase 1: histograms
// ---------------------------------------------------------------------------------------------------------
// order-0 counts of src[0..n) into cnt[256] using R lane-replicated LDS histograms at rep (R*256 words)
__device__ void hist_o0(const uint8_t* src, int32_t n, uint32_t* cnt, uint32_t* rep, int R) {
    const int l = lane_id();
    for (int i = l; i < R * 256; i += 64) rep[i] = 0;
    __syncthreads();
    uint32_t* my = rep + (l & (R - 1)) * 256;
    const int32_t n16 = n & ~15;
    for (int32_t i = l * 16; i < n16; i += 64 * 16) {
        uint4 v = *reinterpret_cast<const uint4*>(src + i);
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            atomicAdd(&my[w[k] & 255u], 1u);
            atomicAdd(&my[(w[k] >> 8) & 255u], 1u);
            atomicAdd(&my[(w[k] >> 16) & 255u], 1u);
            atomicAdd(&my[w[k] >> 24], 1u);
        }
    }
    if (n16 + l < n) atomicAdd(&my[src[n16 + l]], 1u);
    __syncthreads();
    for (int j = l; j < 256; j += 64) {
        uint32_t s = 0;
        for (int r = 0; r < R; r++) s += rep[r * 256 + j];
        cnt[j] = s;
    }
    __syncthreads();
}

// order-1 pair counts: cnt[row(prev)*stride + col(cur)] with prev = 0 before the first byte
// (Frequencies#calcFrequencies_o1 L149-168); indices go through dmap (identity for the 256 x 256 global table).
// Deliberately ONE non-inlined body over generic pointers for the LDS and the global tables, like tables_o1: the
// machine code the wide-alphabet path runs is then the code every small-alphabet block has already exercised.
__device__ __attribute__((noinline)) void hist_o1(const uint8_t* src, int32_t n, uint32_t* cnt, int stride, const uint8_t* dmap, uint32_t* rep,
                        int R, int rstride, int cells) {
    const int l = lane_id();
    uint32_t* my = cnt;
    if (R > 1) {
        for (int i = l; i < R * rstride; i += 64) rep[i] = 0;
        my = rep + (l & (R - 1)) * rstride;
    }
    __syncthreads();
    const int32_t n16 = n & ~15;
    for (int32_t i = l * 16; i < n16; i += 64 * 16) {
        uint4 v = *reinterpret_cast<const uint4*>(src + i);
        uint32_t prev = i ? src[i - 1] : 0u;
        uint32_t row = (uint32_t)dmap[prev] * (uint32_t)stride;
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
#pragma unroll
            for (int bshift = 0; bshift < 32; bshift += 8) {
                uint32_t c = (w[k] >> bshift) & 255u;
                uint32_t col = (uint32_t)dmap[c];
                atomicAdd(&my[row + col], 1u);
                row = col * (uint32_t)stride;
            }
        }
    }
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
    __syncthreads();
    if (R > 1) {
        for (int j = l; j < cells; j += 64) {
            uint32_t s = 0;
            for (int r = 0; r < R; r++) s += rep[r * rstride + j];
            cnt[j] += s;
        }
        __syncthreads();
    }
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
                             const uint32_t* rcp_tab, int32_t tab_cap, uint32_t* dbg = nullptr, int stop = 0) {
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
    __syncthreads();
    if (stop == 3) return -53;
    if (lane_id() == 0) CRAMGPU_MARK(dbg, 1, 50);
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
    __syncthreads();
    if (lane_id() == 0) {
        CRAMGPU_MARK(dbg, 1, 51);
        CRAMGPU_MARK(dbg, 2, total);
    }
    if (stop == 4) return -54;
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
    if (l == 0) CRAMGPU_MARK(dbg, 1, 52);
    __syncthreads();
    if (stop == 5) return -55;
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
    __syncthreads();
    if (stop == 6) return -56;
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
    __syncthreads();
    int32_t len = 0;
    if (l == 0) {
        len = row_body<false>(cnt, 256, L->syml, nullptr);
        if (len <= tab_cap) row_body<true>(cnt, 256, L->syml, tabdst);
    }
    len = __shfl(len, 0);
    __syncthreads();
    return len;
}

// ---------------------------------------------------------------------------------------------------------
// phase 3: the coding loop
// ---------------------------------------------------------------------------------------------------------
enum { MODE_O0 = 0, MODE_O1_LDS = 1, MODE_O1_GLOBAL = 2 };

// The bytes one lane prepares for a batch: lane (t, j) owns step ub + t of state j.
struct Fetch {
    uint32_t sym, ctx;
    bool act;
};

template <int MODE>
__device__ __forceinline__ Fetch fetch_batch(const uint8_t* src, int32_t n, int32_t q, int32_t rem, int32_t U, int32_t ub,
                                             int t, int j) {
    Fetch f{0u, 0u, false};
    const int32_t u = ub + t;
    bool act = u < U;
    int32_t pos = 0;
    if (MODE == MODE_O0) {
        pos = (n & ~3) - 4 * u + j;
        act = act && pos < n;  // u == 0: only states j < (n&3)
    } else {
        int32_t k;
        if (j == 3) {
            k = U - 1 - u;
            pos = 3 * q + k;
        } else {
            k = q - 1 - (u - rem);
            pos = j * q + k;
            act = act && u >= rem;
        }
        if (act && k > 0) f.ctx = src[pos - 1];  // k == 0: the quarter's first symbol is coded in context 0
    }
    if (act) f.sym = src[pos];
    f.act = act;
    return f;
}

template <int MODE>
__device__ __forceinline__ void stage_batch(const Fetch& f, const uint2* ent, int stride, EncFixedLds* L, int buf, int slot,
                                            uint4& A) {
    A = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u);  // identity step for inactive (step, state) pairs
    uint2 B = make_uint2(0u, 0u);
    if (f.act) {
        uint2 e;
        if (MODE == MODE_O0)
            e = ent[f.sym];
        else if (MODE == MODE_O1_LDS)
            e = ent[(uint32_t)L->dmap[f.ctx] * (uint32_t)stride + L->dmap[f.sym]];
        else
            e = ent[f.ctx * 256u + f.sym];
        expand_symbol(e, A, B);
    }
    L->p3.stageA[buf][slot] = A;
    L->p3.stageB[buf][slot] = B;
}

// Returns payload length (including the 16 state bytes) or a negative code.  Payload is written downwards from pend.
// Software pipeline over batches of 16 steps, all in one loop body so that the scheduler can fill the latency gaps of
// the serial chain: iteration i  (a) stages batch i+1 (symbol lookup + expansion, from bytes fetched one iteration
// earlier) and fetches the bytes of batch i+2,  (b) runs the 16 chain steps of batch i,  (c) turns the logged states
// of batch i-1 into output bytes.  The chain is executed by ALL lanes (lane l replicates state l&3): that costs
// nothing on a SIMD and keeps EXEC constant, which is what lets (a) and (c) interleave with (b).
template <int MODE>
__device__ int32_t encode_stream(const uint8_t* src, int32_t n, const uint2* ent, int stride, EncFixedLds* L,
                                 uint8_t* pend, int32_t room, uint32_t* dbg = nullptr) {
    (void)dbg;
    const int l = lane_id();
    const int t = l >> 2, j = 3 - (l & 3), sl = l & 3;
    const int slot = t * 4 + j;
    const int32_t q = n >> 2;
    int32_t U, rem = 0;
    if (MODE == MODE_O0) {
        U = q + 1;  // step 0 is the n&3 tail (E04 L17-33), then q full steps
    } else {
        rem = n - 4 * q;  // state 3 codes the remainder alone first (E14 L27-31)
        U = q + rem;
    }
    const int32_t nbatch = (U + kSteps - 1) / kSteps;
    uint32_t x = RANS_BYTE_L;
    int32_t run = 0;
    const int32_t limit = room - 16;
    uint4 A, prevA = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u);
    Fetch f = fetch_batch<MODE>(src, n, q, rem, U, 0, t, j);
    stage_batch<MODE>(f, ent, stride, L, 0, slot, A);
    f = fetch_batch<MODE>(src, n, q, rem, U, kSteps, t, j);
    __syncthreads();
    for (int32_t i = 0; i < nbatch; i++) {
        const int buf = i & 1;
        // (c) states logged by the previous batch (read before this batch overwrites the log)
        const uint32_t xv = L->p3.xs[t][3 - (l & 3)];
        CRAMGPU_LOCKSTEP();
        // (a) stage batch i+1, fetch batch i+2
        uint4 An;
        stage_batch<MODE>(f, ent, stride, L, buf ^ 1, slot, An);
        f = fetch_batch<MODE>(src, n, q, rem, U, (i + 2) * kSteps, t, j);
        // (b) Encoding#RansEncPutSymbol L46-71, 16 steps
        // all 16 staged symbols go to registers first so that no LDS latency sits between two chain steps
        uint4 sa[kSteps];
        uint2 sb[kSteps];
#pragma unroll
        for (int tt = 0; tt < kSteps; tt++) {
            sa[tt] = L->p3.stageA[buf][tt * 4 + sl];
            sb[tt] = L->p3.stageB[buf][tt * 4 + sl];
        }
#pragma unroll
        for (int tt = 0; tt < kSteps; tt++) {
            L->p3.xs[tt][l] = x;
            const bool r1 = x >= sa[tt].x, r2 = x >= sa[tt].y;  // two independent compares, then two selects
            const uint32_t sh = r2 ? 16u : (r1 ? 8u : 0u);
            const uint32_t xr = x >> sh;
            const uint32_t qq = __umulhi(xr, sa[tt].z) >> sb[tt].y;  // qq < 2^19, cmpl_freq < 2^12: 24-bit multiply is exact
            x = __umul24(qq, sb[tt].x) + (xr + sa[tt].w);
        }
        // (c) bytes of batch i-1 in the reference's emission order (state 3,2,1,0 per step = lane order)
        if (i > 0) {
            const int nb = (xv >= prevA.x ? 1 : 0) + (xv >= prevA.y ? 1 : 0);
            const unsigned long long m1 = __ballot(nb >= 1), m2 = __ballot(nb == 2);
            const int32_t tot = __popcll(m1) + __popcll(m2);
            if (run + tot > limit) return CRAMGPU_E_INSUFFICIENT;
            const int32_t idx = run + (int32_t)(mbcnt(m1) + mbcnt(m2));
            if (nb >= 1) pend[-1 - idx] = (uint8_t)xv;
            if (nb == 2) pend[-2 - idx] = (uint8_t)(xv >> 8);
            run += tot;
        }
        prevA = A;
        A = An;
        __syncthreads();
    }
    if (nbatch > 0) {
        const uint32_t xv = L->p3.xs[t][3 - (l & 3)];
        const int nb = (xv >= prevA.x ? 1 : 0) + (xv >= prevA.y ? 1 : 0);
        const unsigned long long m1 = __ballot(nb >= 1), m2 = __ballot(nb == 2);
        const int32_t tot = __popcll(m1) + __popcll(m2);
        if (run + tot > limit) return CRAMGPU_E_INSUFFICIENT;
        const int32_t idx = run + (int32_t)(mbcnt(m1) + mbcnt(m2));
        if (nb >= 1) pend[-1 - idx] = (uint8_t)xv;
        if (nb == 2) pend[-2 - idx] = (uint8_t)(xv >> 8);
        run += tot;
    }
    // E04 L49-53 / E14 L78-82: big-endian rans3..rans0, then Utils#reverse => stream starts LE rans0..rans3
    if (l < 4) {
        uint8_t* p = pend - run - 16 + 4 * l;
        p[0] = (uint8_t)x;
        p[1] = (uint8_t)(x >> 8);
        p[2] = (uint8_t)(x >> 16);
        p[3] = (uint8_t)(x >> 24);
    }
    return run + 16;
}

// One block's coding work is split into three non-inlined functions (order 0, order 1 with LDS tables, order 1 with
// global tables) so that each gets its own register allocation: inlined into one body the kernel needed ~50 SGPR
// spills, and the wide-alphabet path misbehaved on hardware in exactly that build.
struct Blk {
    const uint8_t* src;
    uint8_t* slot;
    uint8_t* gs;
    const uint32_t* rcp_tab;
    uint32_t* dbg;
    int32_t n, cap, tab_bytes, stop;
};
struct Res {
    int32_t tab_len, pay_len;
};

__device__ __attribute__((noinline)) Res run_o0(Blk B) {
    CRAMGPU_DYN_LDS(smem);
    const int l = lane_id();
    EncFixedLds* L = reinterpret_cast<EncFixedLds*>(smem);
    uint8_t* tab = smem + sizeof(EncFixedLds);
    uint32_t* cnt0 = reinterpret_cast<uint32_t*>(tab);    // 256 words
    uint2* ent0 = reinterpret_cast<uint2*>(tab + 1024);  // 256 x 8 B
    uint32_t* rep0 = reinterpret_cast<uint32_t*>(tab + 1024 + 2048);
    int R0 = 1;
    while (R0 < 16 && (1024 + 2048 + 2 * R0 * 1024) <= B.tab_bytes) R0 <<= 1;
    hist_o0(B.src, B.n, cnt0, rep0, R0);
    for (int i = l; i < 256; i += 64) L->syml[i] = (uint8_t)i;
    __syncthreads();
    Res r;
    r.tab_len = tables_o0(cnt0, ent0, B.n, L, B.slot + 9, B.rcp_tab, B.cap - 9 - 16);
    if (9 + r.tab_len + 16 > B.cap)
        r.pay_len = CRAMGPU_E_INSUFFICIENT;
    else
        r.pay_len = encode_stream<MODE_O0>(B.src, B.n, ent0, 256, L, B.slot + B.cap, B.cap - 9 - r.tab_len);
    return r;
}

__device__ __attribute__((noinline)) Res run_o1_global(Blk B) {
    CRAMGPU_DYN_LDS(smem);
    const int l = lane_id();
    EncFixedLds* L = reinterpret_cast<EncFixedLds*>(smem);
    uint32_t* cnt = reinterpret_cast<uint32_t*>(B.gs);
    uint2* ent = reinterpret_cast<uint2*>(B.gs + 256 * 1024);
    uint4* z = reinterpret_cast<uint4*>(B.gs);
    for (int i = l; i < 256 * 1024 / 16; i += 64) z[i] = make_uint4(0, 0, 0, 0);
    for (int i = l; i < 256; i += 64) {
        L->syml[i] = (uint8_t)i;
        L->dmap[i] = (uint8_t)i;
    }
    __threadfence();
    __syncthreads();
    Res r{0, 0};
    const int stop = B.stop;  // debugging aid: leave the path after phase `stop`
    if (stop == 1) r.pay_len = -51;
    if (r.pay_len == 0) {
        hist_o1(B.src, B.n, cnt, 256, L->dmap, nullptr, 1, 0, 0);
        __threadfence();
        __syncthreads();
        if (stop == 2) r.pay_len = -52;
    }
    if (r.pay_len == 0) {
        r.tab_len = tables_o1(cnt, ent, 256, 256, 256, L, B.slot + 9, B.rcp_tab, B.cap - 9 - 16, B.dbg, stop);
        __threadfence();
        __syncthreads();
        if (r.tab_len < 0) r.pay_len = r.tab_len;
    }
    if (r.pay_len == 0) {
        if (9 + r.tab_len + 16 > B.cap)
            r.pay_len = CRAMGPU_E_INSUFFICIENT;
        else
            r.pay_len = encode_stream<MODE_O1_GLOBAL>(B.src, B.n, ent, 256, L, B.slot + B.cap, B.cap - 9 - r.tab_len, B.dbg);
        if (stop == 7 && r.pay_len >= 0) r.pay_len = -57;
    }
    return r;
}

// returns pay_len == INT32_MIN when the alphabet does not fit the LDS table region
__device__ __attribute__((noinline)) Res run_o1_lds(Blk B) {
    CRAMGPU_DYN_LDS(smem);
    const int l = lane_id();
    EncFixedLds* L = reinterpret_cast<EncFixedLds*>(smem);
    uint8_t* tab = smem + sizeof(EncFixedLds);
    uint32_t* cnt0 = reinterpret_cast<uint32_t*>(tab);
    uint32_t* rep0 = reinterpret_cast<uint32_t*>(tab + 1024 + 2048);
    int R0 = 1;
    while (R0 < 16 && (1024 + 2048 + 2 * R0 * 1024) <= B.tab_bytes) R0 <<= 1;
    hist_o0(B.src, B.n, cnt0, rep0, R0);
    // dense alphabet = present symbols plus 0 (context 0 always exists)
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
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        int jj = l * 4 + k;
        if (present & (1u << k)) {
            L->dmap[jj] = (uint8_t)di;
            L->syml[di] = (uint8_t)jj;
            di++;
        } else {
            L->dmap[jj] = 0;
        }
    }
    __syncthreads();
    const int ns = (int)nsym;
    const int stride = ns | 1;
    const int cells = ns * stride;
    Res r{0, INT32_MIN};
    if (12 * cells + 16 > B.tab_bytes) return r;
    uint32_t* cnt = reinterpret_cast<uint32_t*>(tab);
    const int ent_off = (4 * cells + 15) & ~15;
    uint2* ent = reinterpret_cast<uint2*>(tab + ent_off);
    const int rep_off = ent_off + 8 * cells;
    const int rstride = cells | 1;
    int R = 1;
    while (R < 32 && rep_off + 2 * R * rstride * 4 <= B.tab_bytes) R <<= 1;
    uint32_t* rep = reinterpret_cast<uint32_t*>(tab + rep_off);
    for (int i = l; i < cells; i += 64) cnt[i] = 0;
    __syncthreads();
    hist_o1(B.src, B.n, cnt, stride, L->dmap, rep, R, rstride, cells);
    r.tab_len = tables_o1(cnt, ent, ns, ns, stride, L, B.slot + 9, B.rcp_tab, B.cap - 9 - 16);
    if (9 + r.tab_len + 16 > B.cap)
        r.pay_len = CRAMGPU_E_INSUFFICIENT;
    else
        r.pay_len = encode_stream<MODE_O1_LDS>(B.src, B.n, ent, stride, L, B.slot + B.cap, B.cap - 9 - r.tab_len);
    return r;
}

__device__ void encode_block(const RansArgs& a, int b, uint8_t* gs) {
    const int l = lane_id();
    const cramgpu_block d = a.desc[b];
    const int32_t n = d.in_len;
    int order = d.order;
    uint8_t* slot = a.out + d.out_off;
    const int32_t cap = d.out_cap;
    int32_t result;
    int64_t pos = d.out_off;
    if (order != 0 && order != 1) {
        result = CRAMGPU_E_ORDER;
    } else if (n == 0) {
        result = 0;  // RANS#compress L61-62: EMPTY_BUFFER
    } else if (n < 0 || cap < 9 + 2 + 16) {
        result = n < 0 ? CRAMGPU_E_ARG : CRAMGPU_E_INSUFFICIENT;
    } else {
        if (n < 4) order = 0;  // RANS#compress L63-64
        uint8_t* tabdst = slot + 9;
        uint8_t* pend = slot + cap;
        Blk B{a.in + d.in_off, slot, gs, a.rcp_tab, a.dbg, n, cap, a.tab_bytes, a.stop_phase};
        Res r;
        if (order == 0) {
            r = run_o0(B);
        } else {
            r = run_o1_lds(B);
            if (r.pay_len == INT32_MIN) r = run_o1_global(B);
        }
        const int32_t tab_len = r.tab_len, pay_len = r.pay_len;
        if (l == 0) {
            CRAMGPU_MARK(a.dbg, 0, 8);
            CRAMGPU_MARK(a.dbg, 4, tab_len);
            CRAMGPU_MARK(a.dbg, 5, pay_len);
        }
        if (pay_len < 0) {
            result = pay_len;
        } else {
            // move the table in front of the payload (dst >= src: copy from the top down), then the 9-byte header
            __threadfence_block();
            __syncthreads();
            uint8_t* dst = pend - pay_len - tab_len;
            if (dst != tabdst) {
                for (int32_t hi = tab_len; hi > 0; hi -= 64) {
                    int32_t k = hi - 1 - l;
                    uint8_t v = 0;
                    if (k >= 0) v = tabdst[k];
                    __threadfence_block();
                    __syncthreads();
                    if (k >= 0) dst[k] = v;
                    __threadfence_block();
                    __syncthreads();
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
    if (l == 0) {
        CRAMGPU_MARK(a.dbg, 0, 9);
        a.out_len[b] = result;
        a.out_pos[b] = pos;
        CRAMGPU_MARK(a.dbg, 0, 10);
    }
}

__global__ __launch_bounds__(64) void rans_encode_kernel(RansArgs a) {
    uint8_t* gs = a.gscratch + (size_t)blockIdx.x * ENC_GSCRATCH_BYTES;
    for (;;) {
        uint32_t t = 0;
        if (threadIdx.x == 0) t = atomicAdd(a.ticket, 1u);
        t = __shfl(t, 0);
        if (t >= (uint32_t)a.n_blocks) break;  // every wave reaches this: the queue only grows
        encode_block(a, a.perm[t], gs);
        __syncthreads();
    }
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

size_t rans_encode_fixed_lds() { return sizeof(EncFixedLds); }

hipError_t launch_rans_encode(const RansArgs& a, int n_waves, size_t lds_bytes, hipStream_t s) {
    hipLaunchKernelGGL(rans_encode_kernel, dim3(n_waves), dim3(64), lds_bytes, s, a);
    return hipGetLastError();
}

hipError_t launch_rcp_init(uint32_t* d_rcp_tab, hipStream_t s) {
    hipLaunchKernelGGL(rcp_init_kernel, dim3(16), dim3(256), 0, s, d_rcp_tab);
    return hipGetLastError();
}
