// K2 — CRAM 3.0 rANS 4x8 decoder (order 0 and order 1) for gfx950.
//
// Reference (htsjdk!encoding/rans/*):
//   RANS#uncompress L29-56, Frequencies#readStats_o0 L13-46, #readStats_o1 L50-106, Decoding L24-90,
//   D04#uncompress L9-85, D14#uncompress L9-70
//
// A resident grid of single-wave workgroups pulling blocks from a ticket counter.  A block is one serial chain (the
// four states share one byte cursor, renormalisation order rans0..rans3 per step), and a lone wave issues about one
// instruction per 5 cycles whatever it is, so the step is built to be short and to hold ONE LDS round trip:
//   * each state lives in a 16-lane DPP row (same value in all 16 lanes);
//   * order 0: slot -> (symbol, freq-1, slot-cum) from one 4096 x u32 LDS table;
//   * order 1: 256 contexts x 4096 slots do not fit LDS, so a context is a row of packed (cum, freq, symbol) words
//     over the block's dense alphabet: lane s tests the slot against symbols s, s+16, ..., a 4-step DPP OR butterfly
//     hands the winner's word to the row, and the next context's row is fetched while the state renormalises;
//   * renormalisation waits for no memory: the 8 stream bytes at the cursor sit in a register pair that is reloaded
//     (one unaligned global load, on the vector-memory counter so that a wait for a table row never waits for it)
//     only after a step that consumed bytes; a step in which no state needs a byte (most steps of a quality block)
//     skips it with one uniform branch, and the usual case of at most one byte per state has a short path;
//   * decoded symbols stay in registers (lane j of a row keeps step j of a 16-step chunk) and leave as one byte store
//     per lane per chunk.
#include "internal.h"

namespace {

constexpr int kRing = 4096;  // LDS ring of compressed bytes (frequency-table parse)
constexpr int kChunk = 16;   // decode steps between ring checks / output stores
constexpr unsigned long long kTop = 0x8000800080008000ull;  // lane 15 of each 16-lane row: one representative per state

struct DecFixedLds {
    uint8_t ring[kRing];
    uint8_t dmap[256];
    uint8_t syml[256];
    uint8_t present[256];
    uint16_t nsym[256];      // symbols listed by context (byte) i
    uint32_t rowinfo[256];   // ragged rows: entry offset | 16-entry units << 24, by dense context
};

__device__ __forceinline__ int lane_id() { return (int)threadIdx.x; }
__device__ __forceinline__ uint32_t mbcnt2(unsigned long long m1, unsigned long long m2) {
    uint32_t a = __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u);
    a = __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), a);
    a = __builtin_amdgcn_mbcnt_lo((uint32_t)m2, a);
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m2 >> 32), a);
}

// OR over the 16 lanes of a DPP row (butterfly: quad xor 1, quad xor 2, half mirror, row mirror)
__device__ __forceinline__ uint32_t row_or(uint32_t v) {
#ifdef HIP_SIM
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);  // row_half_mirror
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false);  // row_mirror
#else
    // a DPP source written by the preceding vector instruction needs two wait states (the compiler cannot see into
    // the asm, so the first s_nop is ours as well)
    asm("s_nop 1\n\t"
        "v_or_b32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_or_b32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_or_b32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_or_b32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf"
        : "+v"(v));
#endif
    return v;
}

struct __attribute__((packed)) PackedU64 { unsigned long long v; };

// reader of the frequency tables: bytes [fill-kRing, fill) of the stream are resident in the ring; the next 1 KiB
// piece is already on its way in registers (`pre`) when a refill writes the previous one
struct Stream {
    const uint8_t* src;   // 16-byte aligned start of the stream
    int32_t len;          // stream bytes
    int32_t fill;         // multiple of 1024
    uint8_t* ring;
    uint4 pre;
    __device__ __forceinline__ void fetch() {
        const int32_t i = fill + lane_id() * 16;
        if (i < len) pre = *reinterpret_cast<const uint4*>(src + i);  // arena padded to 16 bytes: a partial last unit is read whole
    }
    __device__ __forceinline__ void reset() {
        __syncthreads();
        fill = 0;
        pre = make_uint4(0, 0, 0, 0);
        fetch();
    }
    __device__ __forceinline__ void refill() {
        const int32_t pos = (fill & (kRing - 1)) + lane_id() * 16;
        *reinterpret_cast<uint4*>(ring + pos) = pre;
        fill += 1024;
        fetch();
        __syncthreads();
    }
    // byte at stream offset p (uniform p), sequential access
    __device__ __forceinline__ uint32_t at(int32_t p) {
        while (p >= fill) refill();
        return ring[p & (kRing - 1)];
    }
};

#ifdef HIP_SIM
#define UNIFORM(x) (x)
#else
#define UNIFORM(x) __builtin_amdgcn_readfirstlane((int)(x))
#endif

// The payload is read straight from global memory (vector memory counter), the tables from LDS (LDS counter): a wait
// for a table row then never waits for a window that a renormalisation requested after it.
struct Payload {
    const uint8_t* src;  // uniform
    int32_t len;         // uniform
    // the 8 stream bytes at offset p, little endian; clamped so that a corrupt stream cannot walk out of the arena
    // (the arena has 16 bytes of padding behind the last stream)
    __device__ __forceinline__ unsigned long long window_v(uint32_t p) const {  // p in a vector register
        const uint32_t c = p < (uint32_t)len ? p : (uint32_t)len;
#ifdef HIP_SIM
        return reinterpret_cast<const PackedU64*>(src + c)->v;
#else
        typedef const __attribute__((address_space(1))) PackedU64* GPtr;
        return ((GPtr)((unsigned long long)src + c))->v;
#endif
    }
    __device__ __forceinline__ unsigned long long window(int32_t p) const {
        const int32_t c = p < len ? p : len;
#ifdef HIP_SIM
        return reinterpret_cast<const PackedU64*>(src + c)->v;
#else
        // an explicit global pointer: a flat load would count on the LDS counter as well
        typedef const __attribute__((address_space(1))) PackedU64* GPtr;
        return ((GPtr)(unsigned long long)(src + c))->v;
#endif
    }
};
__device__ __forceinline__ const uint8_t* uniform_ptr(const uint8_t* p) {
    const unsigned long long v = (unsigned long long)p;
    return (const uint8_t*)(((unsigned long long)(uint32_t)UNIFORM((uint32_t)(v >> 32)) << 32) | (uint32_t)UNIFORM((uint32_t)v));
}

// Decoding#RansDecRenorm for the four states of a step, in the reference's order rans0..rans3 from one shared cursor
// (D04#uncompress L60-63, D14#uncompress L50-53).  A state lives in a 16-lane row (same value in every lane);
// `w` holds the 8 stream bytes at `cur`: a step consumes at most 2 bytes per state.
// byte c of the result = byte sel[c] (0..7) of {hi:lo}
__device__ __forceinline__ uint32_t perm8(uint32_t hi, uint32_t lo, uint32_t sel) {
#ifdef HIP_SIM
    const unsigned long long pool = ((unsigned long long)hi << 32) | lo;
    uint32_t r = 0;
    for (int c = 0; c < 4; c++) r |= (uint32_t)((pool >> (8 * ((sel >> (8 * c)) & 7))) & 255u) << (8 * c);
    return r;
#else
    return __builtin_amdgcn_perm(hi, lo, sel);
#endif
}
// a value the compiler must keep in a vector register and may not assume uniform
__device__ __forceinline__ uint32_t in_vgpr(uint32_t v) {
#ifndef HIP_SIM
    asm volatile("" : "+v"(v));
#endif
    return v;
}
// `cur` is the cursor for the (uniform) bounds checks, `curv` its copy in a vector register: the window address is
// then a 32-bit vector offset on the scalar stream base (two vector instructions instead of five scalar ones).
__device__ __forceinline__ void renorm(uint32_t& x, int32_t& cur, uint32_t& curv, unsigned long long& w, const Payload& P,
                                       bool live) {
    const bool n1 = live && x < RANS_BYTE_L;
    const unsigned long long b1 = __ballot(n1);
    if (b1 != 0) {  // uniform; most steps of a well-compressed block need no byte at all
        const bool n2 = live && x < (1u << 15);
        const unsigned long long b2 = __ballot(n2);
        const unsigned long long m1 = b1 & kTop;
        uint32_t a = __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u));
        int32_t tot = __popcll(m1);
        // x <- x with 0, 1 or 2 stream bytes shifted in: a byte permutation of {x : window word}
        uint32_t sel = n1 ? 0x06050400u : 0x07060504u;
        if (b2 != 0) {  // some state needs a second byte (that takes a symbol of frequency < 16)
            const unsigned long long m2 = b2 & kTop;
            a = __builtin_amdgcn_mbcnt_hi((uint32_t)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m2, a));
            tot += __popcll(m2);
            sel = n2 ? 0x05040001u : sel;
        }
        const uint32_t t = (uint32_t)(w >> (8u * a));  // this state's bytes, first one lowest
        x = perm8(x, t, sel);
        cur += tot;
        curv += (uint32_t)tot;
        w = P.window_v(curv);
    }
}

// lane s of a row takes the value of lane s-1, lane 0 takes v: after 16 calls lane s holds the v of 15-s calls ago
__device__ __forceinline__ uint32_t row_push(uint32_t keep, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)keep, 0x111, 0xF, 0xF, false);  // row_shr:1
}

// Frequencies#readStats_o0 L13-46: one row.  V::sym(j, F, C) is called per listed symbol.  Returns 0 or a code.
template <bool O1, class V>
__device__ __forceinline__ int read_row(Stream& S, int32_t& p, V& v) {
    int rle = 0, x = 0;
    if (p >= S.len) return CRAMGPU_E_CORRUPT;
    int j = (int)S.at(p++);
    do {
        if (p >= S.len) return CRAMGPU_E_CORRUPT;
        int F = (int)S.at(p++);
        if (F >= 128) {
            if (p >= S.len) return CRAMGPU_E_CORRUPT;
            F = ((F & 127) << 8) | (int)S.at(p++);
        }
        if (O1 && F == 0) F = RANS_TOTFREQ;
        if (x + F > RANS_TOTFREQ) return CRAMGPU_E_CORRUPT;
        v.sym(j, F, x);
        x += F;
        if (p >= S.len) return CRAMGPU_E_CORRUPT;
        if (rle == 0 && j + 1 == (int)S.at(p)) {
            j = (int)S.at(p++);
            if (p >= S.len) return CRAMGPU_E_CORRUPT;
            rle = (int)S.at(p++);
        } else if (rle != 0) {
            rle--;
            j++;
            if (j > 255) return CRAMGPU_E_CORRUPT;
        } else {
            j = (int)S.at(p++);
        }
    } while (j != 0);
    return 0;
}

// Frequencies#readStats_o1 L50-106: V::ctx(i) then the row's symbols.
template <class V>
__device__ __forceinline__ int read_rows_o1(Stream& S, int32_t& p, V& v) {
    int rle = 0;
    if (p >= S.len) return CRAMGPU_E_CORRUPT;
    int i = (int)S.at(p++);
    do {
        v.ctx(i);
        int rc = read_row<true>(S, p, v);
        if (rc) return rc;
        if (p >= S.len) return CRAMGPU_E_CORRUPT;
        if (rle == 0 && i + 1 == (int)S.at(p)) {
            i = (int)S.at(p++);
            if (p >= S.len) return CRAMGPU_E_CORRUPT;
            rle = (int)S.at(p++);
        } else if (rle != 0) {
            rle--;
            i++;
            if (i > 255) return CRAMGPU_E_CORRUPT;
        } else {
            i = (int)S.at(p++);
        }
    } while (i != 0);
    return 0;
}

struct FillO0 {
    uint32_t* R;  // [4096] sym | (F-1)<<8 | (slot-C)<<20
    __device__ __forceinline__ void sym(int j, int F, int C) {
        const uint32_t base = (uint32_t)j | ((uint32_t)(F - 1) << 8);
        for (int k = lane_id(); k < F; k += 64) R[C + k] = base | ((uint32_t)k << 20);
    }
};
struct MarkO1 {
    uint8_t* present;
    uint16_t* nsym;
    int cur;
    __device__ __forceinline__ void ctx(int i) {
        present[i] = 1;
        cur = i;
    }
    __device__ __forceinline__ void sym(int j, int, int) {
        present[j] = 1;
        if (lane_id() == 0 && nsym[cur] < 256) nsym[cur]++;
    }
};
// Ragged rows for wide alphabets: a context's row lists only the symbols it has, padded to a multiple of 16 with
// copies of its first entry (a 200-symbol ITF8 series has ~3 symbols in most contexts: 13 tests per step become 1-2,
// and the rows fit LDS).
struct FillRagged {
    uint32_t* rows;
    const uint8_t* dmap;
    const uint32_t* rowinfo;
    uint32_t cur, cap, fill, first;
    __device__ __forceinline__ void finish() {
        if (fill > 0)
            for (uint32_t q = fill + (uint32_t)lane_id(); q < cap; q += 64) rows[cur + q] = first;
        fill = 0;
    }
    __device__ __forceinline__ void ctx(int i) {
        finish();
        const uint32_t info = rowinfo[dmap[i]];
        cur = info & 0xFFFFFFu;
        cap = 16u * (info >> 24);
    }
    __device__ __forceinline__ void sym(int j, int F, int C) {
        const uint32_t e = ((4096u - (uint32_t)C) & 4095u) | ((uint32_t)dmap[j] << 12) | ((uint32_t)(F - 1) << 20);
        if (fill == 0) first = e;
        if (lane_id() == 0 && fill < cap) rows[cur + fill] = e;
        fill++;
    }
};
struct FillO1 {
    uint32_t* rows;  // [nd][stride]: (4096 - C) & 4095 | k << 12 | (F-1) << 20, see decode_o1
    const uint8_t* dmap;
    int stride;
    int cur;
    __device__ __forceinline__ void ctx(int i) { cur = dmap[i] * stride; }
    __device__ __forceinline__ void sym(int j, int F, int C) {
        if (lane_id() == 0)
            rows[cur + dmap[j]] = ((4096u - (uint32_t)C) & 4095u) | ((uint32_t)dmap[j] << 12) | ((uint32_t)(F - 1) << 20);
    }
};
constexpr uint32_t kAbsent = 0xFFFFFFFFu;  // not a valid entry: it would be C = 1 with F = 4096

// rows[nd][stride] <- kAbsent
__device__ __forceinline__ void rows_clear(uint32_t* rows, int total) {
    for (int i = lane_id(); i < total; i += 64) rows[i] = kAbsent;
}
// every absent entry of a row becomes a copy of the row's first present entry (0 for a context the table never
// defined: such a row is only reachable from a corrupt stream and decodes to garbage without leaving the tables)
__device__ __forceinline__ void rows_fix(uint32_t* rows, int nd, int stride) {
    const int l = lane_id();
    for (int r = 0; r < nd; r++) {
        uint32_t* row = rows + r * stride;
        uint32_t first = 0;
        bool found = false;
        for (int c0 = 0; c0 < stride && !found; c0 += 64) {
            const uint32_t v = c0 + l < stride ? row[c0 + l] : kAbsent;
            const unsigned long long m = __ballot(v != kAbsent);
            if (m != 0) {
                first = __shfl(v, __ffsll((long long)m) - 1);
                found = true;
            }
        }
        for (int c = l; c < stride; c += 64)
            if (row[c] == kAbsent) row[c] = first;
    }
}

// D04#uncompress L9-85.  R[4096]: sym | (F-1)<<8 | (slot-C)<<20.  State g = lanes 16g..16g+15.
__device__ int decode_o0(Stream& S, int32_t p, const uint32_t* R, uint8_t* out, int32_t n) {
    const int l = lane_id();
    const int g = l >> 4, s = l & 15;
    if (p + 16 > S.len) return CRAMGPU_E_CORRUPT;
    const Payload P{uniform_ptr(S.src), UNIFORM(S.len)};
    uint32_t x = (uint32_t)P.window(p + 4 * g);
    int32_t cur = UNIFORM(p) + 16;
    uint32_t curv = in_vgpr((uint32_t)cur);
    unsigned long long w = P.window(cur);
    const int32_t nsteps = n >> 2;
    int32_t i = 0;
    for (; i + kChunk <= nsteps; i += kChunk) {
        if (cur > S.len) return CRAMGPU_E_CORRUPT;
        uint32_t keep = 0;
#pragma unroll
        for (int j = 0; j < kChunk; j++) {
            const uint32_t e = R[x & 4095u];
            keep = row_push(keep, e);
            x = (((e >> 8) & 4095u) + 1u) * (x >> 12) + (e >> 20);
            renorm(x, cur, curv, w, P, true);
        }
        out[4 * (i + 15 - s) + g] = (uint8_t)keep;
    }
    const int32_t rem = nsteps - i;
    if (rem > 0) {
        uint32_t keep = 0;
        for (int j = 0; j < rem; j++) {
            const uint32_t e = R[x & 4095u];
            keep = s == j ? e : keep;
            x = (((e >> 8) & 4095u) + 1u) * (x >> 12) + (e >> 20);
            renorm(x, cur, curv, w, P, true);
        }
        if (s < rem) out[4 * (i + s) + g] = (uint8_t)keep;
    }
    if (cur > S.len) return CRAMGPU_E_CORRUPT;
    if (s == 0 && g < (n & 3)) out[4 * nsteps + g] = (uint8_t)R[x & 4095u];
    return 0;
}

// D14#uncompress L9-70.  rows[nd][16*M] over the dense alphabet (index 0 = byte 0); an entry is
//   (4096 - C) & 4095 | k << 12 | (F-1) << 20
// so that with xa = x + entry the slot offset inside the symbol is xa & 4095 and the slot belongs to the symbol iff
// (xa << 20) <= entry (C + F <= 4096 keeps a slot below C from aliasing).  Lane s of a state's row tests the symbols
// s, s+16, ...; symbols a context does not have carry a copy of one it has, so every hit ORs to the same word.
// MS > 0: square rows[nd][16*MS], MS known at compile time (the next row is fetched while the state renormalises).
// MS == 0: ragged rows (FillRagged): rowinfo[k] = entry offset | 16-entry units << 24 of context k's row.
template <int MS>
__device__ int decode_o1(Stream& S, int32_t p, const uint32_t* rows, const uint32_t* rowinfo, const uint8_t* syml,
                         uint8_t* out, int32_t n) {
    const int l = lane_id();
    const int g = l >> 4, s = l & 15;
    const uint32_t stride = 16u * (uint32_t)(MS > 0 ? MS : 1);
    if (p + 16 > S.len) return CRAMGPU_E_CORRUPT;
    const Payload P{uniform_ptr(S.src), UNIFORM(S.len)};
    uint32_t x = (uint32_t)P.window(p + 4 * g);
    int32_t cur = UNIFORM(p) + 16;
    uint32_t curv = in_vgpr((uint32_t)cur);
    unsigned long long w = P.window(cur);
    const int32_t q = n >> 2;
    uint32_t e[MS > 0 ? MS : 1];
    uint32_t roff = (uint32_t)s;  // context 0
    uint32_t Mc = 1;               // ragged: 16-entry units of this state's current row
    if (MS > 0) {
#pragma unroll
        for (int m = 0; m < MS; m++) e[m] = rows[roff + 16 * m];
    } else {
        const uint32_t info = rowinfo[0];
        roff = (info & 0xFFFFFFu) + (uint32_t)s;
        Mc = info >> 24;
    }
    auto step = [&](bool live) -> uint32_t {
        uint32_t pk = 0;
        if (MS > 0) {
#pragma unroll
            for (int m = 0; m < MS; m++) {
                const uint32_t xa = x + e[m];
                pk |= (xa << 20) <= e[m] ? ((e[m] & ~4095u) | (xa & 4095u)) : 0u;
            }
        } else {
            // up to the longest row of the four states, four entries per round trip; an index past a row's end
            // repeats its last entry, which ORs to the same word
            for (uint32_t m = 0; __ballot(m < Mc) != 0; m += 4) {
                uint32_t em[4];
#pragma unroll
                for (uint32_t c = 0; c < 4; c++) em[c] = rows[roff + 16u * (m + c < Mc ? m + c : Mc - 1u)];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const uint32_t xa = x + em[c];
                    pk |= (xa << 20) <= em[c] ? ((em[c] & ~4095u) | (xa & 4095u)) : 0u;
                }
            }
        }
        pk = row_or(pk);
        const uint32_t k = (pk >> 12) & 255u;
        if (live) {
            if (MS > 0) {
                roff = k * stride + (uint32_t)s;
#pragma unroll
                for (int m = 0; m < MS; m++) e[m] = rows[roff + 16 * m];
            } else {
                const uint32_t info = rowinfo[k];
                roff = (info & 0xFFFFFFu) + (uint32_t)s;
                Mc = info >> 24;
            }
            x = ((pk >> 20) + 1u) * (x >> 12) + (pk & 4095u);
        }
        renorm(x, cur, curv, w, P, live);
        return k;
    };
    int32_t i = 0;
    for (; i + kChunk <= q; i += kChunk) {
        if (cur > S.len) return CRAMGPU_E_CORRUPT;
        uint32_t keep = 0;
#pragma unroll
        for (int j = 0; j < kChunk; j++) {
            keep = row_push(keep, step(true));
        }
        out[g * q + i + 15 - s] = syml[keep];
    }
    // the last q % 16 common steps, then the n - 4q bytes that only state 3 decodes
    const int32_t total = q + (n - 4 * q);
    while (i < total) {
        if (cur > S.len) return CRAMGPU_E_CORRUPT;
        const int32_t cnt = total - i < kChunk ? total - i : kChunk;
        uint32_t keep = 0;
        for (int j = 0; j < cnt; j++) {
            const bool live = i + j < q || g == 3;
            const uint32_t k = step(live);
            keep = s == j ? k : keep;
        }
        if (s < cnt && (i + s < q || g == 3)) out[g * q + i + s] = syml[keep];
        i += cnt;
    }
    if (cur > S.len) return CRAMGPU_E_CORRUPT;
    return 0;
}

__device__ void decode_block(const RansArgs& a, int b, uint8_t* smem, uint8_t* gs) {
    const int l = lane_id();
    DecFixedLds* L = reinterpret_cast<DecFixedLds*>(smem);
    uint8_t* tab = smem + sizeof(DecFixedLds);
    const cramgpu_block d = a.desc[b];
    const int32_t len = d.in_len;
    const uint8_t* src = a.in + d.in_off;
    uint8_t* out = a.out + d.out_off;
    int32_t result;
    // RANS#uncompress L29-56
    if (len == 0) {
        result = 0;
    } else if (len < 9) {
        result = CRAMGPU_E_INPUT_LENGTH;
    } else {
        const int order = src[0];
        const uint32_t comp = (uint32_t)src[1] | ((uint32_t)src[2] << 8) | ((uint32_t)src[3] << 16) | ((uint32_t)src[4] << 24);
        const int32_t raw = (int32_t)((uint32_t)src[5] | ((uint32_t)src[6] << 8) | ((uint32_t)src[7] << 16) | ((uint32_t)src[8] << 24));
        if (order != 0 && order != 1) {
            result = CRAMGPU_E_ORDER;
        } else if ((int64_t)comp != (int64_t)len - 9) {
            result = CRAMGPU_E_INPUT_LENGTH;
        } else if (raw < 0) {
            result = CRAMGPU_E_CORRUPT;
        } else if (raw > d.out_cap) {
            result = CRAMGPU_E_OUT_TOO_SMALL;
        } else if (raw == 0) {
            result = 0;
        } else {
#ifndef HIP_SIM
            // the launch lasts as long as its longest chains: a long block's wave issues ahead of the waves of
            // short blocks that share its SIMD
            if (raw >= (1 << 18))
                __builtin_amdgcn_s_setprio(3);
            else
                __builtin_amdgcn_s_setprio(0);
#endif
            Stream S{src, len, 0, L->ring, make_uint4(0, 0, 0, 0)};
            S.reset();
            int32_t p = 9;
            int rc;
            if (order == 0) {
                uint32_t* R = reinterpret_cast<uint32_t*>(tab);
                for (int i = l; i < 4096; i += 64) R[i] = 0;
                __syncthreads();
                FillO0 v{R};
                rc = read_row<false>(S, p, v);
                __syncthreads();
                if (!rc) rc = decode_o0(S, p, R, out, raw);
            } else {
                for (int i = l; i < 256; i += 64) {
                    L->present[i] = i == 0;
                    L->nsym[i] = 0;
                }
                __syncthreads();
                MarkO1 mk{L->present, L->nsym, 0};
                int32_t p1 = p;
                rc = read_rows_o1(S, p1, mk);
                __syncthreads();
                if (!rc) {
                    // dense alphabet in byte order, byte 0 first; ragged row geometry alongside
                    int nd = 0, total = 0;
                    if (l == 0) {
                        for (int j = 0; j < 256; j++) {
                            if (L->present[j]) {
                                const int units = L->nsym[j] ? (L->nsym[j] + 15) >> 4 : 1;
                                L->rowinfo[nd] = (uint32_t)total | ((uint32_t)units << 24);
                                total += 16 * units;
                                L->dmap[j] = (uint8_t)nd;
                                L->syml[nd] = (uint8_t)j;
                                nd++;
                            } else {
                                L->dmap[j] = 0;
                            }
                        }
                    }
                    nd = __shfl(nd, 0);
                    total = __shfl(total, 0);
                    __syncthreads();
                    S.reset();
                    if (nd <= 48) {  // square rows, 16 * M entries each, always in LDS (<= 9 KiB)
                        const int M = (nd + 15) >> 4, stride = 16 * M;
                        uint32_t* r = reinterpret_cast<uint32_t*>(tab);
                        rows_clear(r, nd * stride);
                        __syncthreads();
                        FillO1 v{r, L->dmap, stride, 0};
                        rc = read_rows_o1(S, p, v);
                        __syncthreads();
                        if (!rc) {
                            rows_fix(r, nd, stride);
                            __syncthreads();
                            if (M == 1)
                                rc = decode_o1<1>(S, p, r, nullptr, L->syml, out, raw);
                            else if (M == 2)
                                rc = decode_o1<2>(S, p, r, nullptr, L->syml, out, raw);
                            else
                                rc = decode_o1<3>(S, p, r, nullptr, L->syml, out, raw);
                        }
                    } else {  // ragged rows, in LDS when they fit (total <= 256 * 256 entries = the global scratch)
                        const bool in_lds = total * 4 <= a.tab_bytes;
                        if (in_lds) {
                            uint32_t* r = reinterpret_cast<uint32_t*>(tab);
                            for (int i = l; i < total; i += 64) r[i] = 0;
                            __syncthreads();
                            FillRagged v{r, L->dmap, L->rowinfo, 0, 0, 0, 0};
                            rc = read_rows_o1(S, p, v);
                            v.finish();
                            __syncthreads();
                            if (!rc) rc = decode_o1<0>(S, p, r, L->rowinfo, L->syml, out, raw);
                        } else {
                            uint32_t* r = reinterpret_cast<uint32_t*>(gs);
                            for (int i = l; i < total; i += 64) r[i] = 0;
                            __threadfence();
                            __syncthreads();
                            FillRagged v{r, L->dmap, L->rowinfo, 0, 0, 0, 0};
                            rc = read_rows_o1(S, p, v);
                            v.finish();
                            __threadfence();
                            __syncthreads();
                            if (!rc) rc = decode_o1<0>(S, p, r, L->rowinfo, L->syml, out, raw);
                        }
                    }
                }
            }
            result = rc ? rc : raw;
        }
    }
    if (l == 0) a.out_len[b] = result;
}

__global__ __launch_bounds__(64) void rans_decode_kernel(RansArgs a) {
    CRAMGPU_DYN_LDS(smem);
    uint8_t* gs = a.gscratch + (size_t)blockIdx.x * DEC_GSCRATCH_BYTES;
    for (;;) {
        uint32_t t = 0;
        if (threadIdx.x == 0) t = atomicAdd(a.ticket, 1u);
        t = __shfl(t, 0);
        if (t >= (uint32_t)a.n_blocks) break;
        decode_block(a, a.perm[t], smem, gs);
        __syncthreads();
    }
}

}  // namespace

size_t rans_decode_fixed_lds() { return sizeof(DecFixedLds); }

hipError_t launch_rans_decode(const RansArgs& a, int n_waves, size_t lds_bytes, hipStream_t s) {
    hipLaunchKernelGGL(rans_decode_kernel, dim3(n_waves), dim3(64), lds_bytes, s, a);
    return hipGetLastError();
}
