// K2 — CRAM 3.0 rANS 4x8 decoder (order 0 and order 1) for gfx950.
//
// Reference (htsjdk!encoding/rans/*):
//   RANS#uncompress L29-56, Frequencies#readStats_o0 L13-46, #readStats_o1 L50-106, Decoding L24-90,
//   D04#uncompress L9-85, D14#uncompress L9-70
//
// A resident grid of single-wave workgroups pulling blocks from a two-ended work queue (the first wave of the launch on a
// SIMD from the front of the longest-first order, later waves from its back: rans_decode_kernel).  A block is one serial chain (the
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

constexpr int kChunk = 16;   // decode steps between ring checks / output stores
constexpr unsigned long long kTop = 0x8000800080008000ull;  // lane 15 of each 16-lane row: one representative per state

struct DecFixedLds {
    uint8_t dmap[256];
    uint8_t syml[256];
    uint8_t present[256];
    uint32_t nsym[256];      // symbols listed by context (byte) i
    uint32_t rowinfo[256];   // ragged rows: entry offset | 16-entry units << 24, by context
    uint8_t stage[1024];     // narrow order-1 path: dense symbols of 256 steps x 4 states, flushed 16 bytes per lane
};
static_assert(sizeof(DecFixedLds) % 64 == 0, "tables behind the fixed part start on a 64-byte boundary");

__device__ __forceinline__ int lane_id() { return (int)threadIdx.x; }
__device__ __forceinline__ uint32_t mbcnt2(unsigned long long m1, unsigned long long m2) {
    uint32_t a = __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u);
    a = __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), a);
    a = __builtin_amdgcn_mbcnt_lo((uint32_t)m2, a);
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m2 >> 32), a);
}

struct __attribute__((packed)) PackedU64 { unsigned long long v; };

// Reader of the frequency tables.  The parse is one serial chain of byte-sized decisions, the same in every lane: it runs
// on the scalar unit.  16 stream bytes at a 4-byte boundary sit in two scalar register pairs, fetched with scalar loads
// (read only; the stream was written before the launch) and re-fetched once the cursor has moved 12 bytes or more:
// a symbol of a table costs about 30 scalar instructions and a third of a load, where bytes fetched one at a time
// from an LDS copy (values in vector registers, every branch an exec-mask update) cost some 600 cycles.
struct Stream {
    const uint8_t* src;   // 16-byte aligned start of the stream (uniform)
    int32_t len;          // stream bytes (uniform)
    int32_t base;         // stream offset of lo's first byte (multiple of 4)
    unsigned long long lo, hi;
    __device__ __forceinline__ void load(int32_t p) {
        const int32_t c = p < len ? p : len;  // the arena has 16 bytes of padding behind its last stream
        base = c & ~3;
#ifdef HIP_SIM
        lo = reinterpret_cast<const PackedU64*>(src + base)->v;
        hi = reinterpret_cast<const PackedU64*>(src + base + 8)->v;
#else
        typedef const __attribute__((address_space(4))) uint32_t* CPtr;
        CPtr q = (CPtr)((unsigned long long)src + (unsigned long long)(uint32_t)base);
        const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3];
        lo = ((unsigned long long)d1 << 32) | d0;
        hi = ((unsigned long long)d3 << 32) | d2;
#endif
    }
    // at least 5 stream bytes starting at offset p (first one lowest); p < len
    __device__ __forceinline__ unsigned long long at5(int32_t p) {
        uint32_t off = (uint32_t)(p - base);
        if (off > 11u) {
            load(p);
            off = (uint32_t)(p - base);
        }
        const uint32_t sh = 8u * (off & 7u);
        const unsigned long long a = off < 8u ? lo : hi;
        const unsigned long long b = off < 8u ? hi : 0ull;
        return sh != 0 ? (a >> sh) | (b << (64u - sh)) : a;
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
// A listed symbol takes at most 4 bytes (a two-byte frequency, the next symbol, a run length).  The reference reads 16
// bytes of states behind the tables, so a table that comes within 4 bytes of the stream's end is a failure there too.
template <bool O1, class V>
__device__ __forceinline__ int read_row(Stream& S, int32_t& p, V& v) {
    int rle = 0, x = 0;
    if (p >= S.len) return CRAMGPU_E_CORRUPT;
    int j = (int)(S.at5(p) & 255u);
    p++;
    do {
        if (p + 4 > S.len) return CRAMGPU_E_CORRUPT;
        unsigned long long w = S.at5(p);
        int F = (int)(w & 255u);
        if (F >= 128) {
            F = ((F & 127) << 8) | (int)((w >> 8) & 255u);
            w >>= 16;
            p += 2;
        } else {
            w >>= 8;
            p += 1;
        }
        if (O1 && F == 0) F = RANS_TOTFREQ;
        if (x + F > RANS_TOTFREQ) return CRAMGPU_E_CORRUPT;
        v.sym(j, F, x);
        x += F;
        const int nx = (int)(w & 255u);
        if (rle == 0 && j + 1 == nx) {
            j = nx;
            rle = (int)((w >> 8) & 255u);
            p += 2;
        } else if (rle != 0) {
            rle--;
            j++;
            if (j > 255) return CRAMGPU_E_CORRUPT;
        } else {
            j = nx;
            p++;
        }
    } while (j != 0);
    return 0;
}

// Frequencies#readStats_o1 L50-106: V::ctx(i) then the row's symbols.
template <class V>
__device__ __forceinline__ int read_rows_o1(Stream& S, int32_t& p, V& v) {
    int rle = 0;
    if (p >= S.len) return CRAMGPU_E_CORRUPT;
    int i = (int)(S.at5(p) & 255u);
    p++;
    do {
        v.ctx(i);
        int rc = read_row<true>(S, p, v);
        if (rc) return rc;
        if (p + 2 > S.len) return CRAMGPU_E_CORRUPT;
        const unsigned long long w = S.at5(p);
        const int nx = (int)(w & 255u);
        if (rle == 0 && i + 1 == nx) {
            i = nx;
            rle = (int)((w >> 8) & 255u);
            p += 2;
        } else if (rle != 0) {
            rle--;
            i++;
            if (i > 255) return CRAMGPU_E_CORRUPT;
        } else {
            i = nx;
            p++;
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
    uint32_t* nsym;
    int cur;
    uint32_t cnt;  // symbols of the current row so far (uniform)
    __device__ __forceinline__ void finish() {
        if (cnt != 0 && lane_id() == 0) atomicAdd(&nsym[cur], cnt);  // a context listed twice (corrupt) adds up
        cnt = 0;
    }
    __device__ __forceinline__ void ctx(int i) {
        finish();
        present[i] = 1;
        cur = i;
    }
    __device__ __forceinline__ void sym(int j, int, int) {
        present[j] = 1;
        cnt++;
    }
};
// Ragged rows for wide alphabets: a context's row lists only the symbols it has, padded to a multiple of 16 with
// copies of its first entry (a 200-symbol ITF8 series has ~3 symbols in most contexts: 13 tests per step become 1-2,
// and the rows fit LDS).  Symbols and contexts are byte values here (rowinfo is indexed by the byte, the symbol table
// is the identity): a listed symbol's entry is made of scalars alone and leaves as one store.
struct FillRagged {
    uint32_t* rows;
    const uint32_t* rowinfo;
    uint32_t cur, cap, fill, first;  // uniform
    __device__ __forceinline__ void finish() {
        if (fill > 0)
            for (uint32_t q = fill + (uint32_t)lane_id(); q < cap; q += 64) rows[cur + q] = first;
        fill = 0;
    }
    __device__ __forceinline__ void ctx(int i) {
        finish();
        const uint32_t info = (uint32_t)UNIFORM(rowinfo[i]);
        cur = info & 0xFFFFFFu;
        cap = 16u * (info >> 24);
    }
    __device__ __forceinline__ void sym(int j, int F, int C) {
        const uint32_t e = (((4096u - (uint32_t)C) & 4095u) << 20) | ((uint32_t)(F - 1) << 8) | (uint32_t)j;
        if (fill == 0) first = e;
        if (lane_id() == 0 && fill < cap) rows[cur + fill] = e;
        fill++;
    }
};
struct FillO1 {
    uint32_t* rows;  // [nd][stride]: (4096 - C) & 4095 << 20 | (F-1) << 8 | k, see decode_o1
    const uint8_t* dmap;
    int stride;
    int cur;
    __device__ __forceinline__ void ctx(int i) { cur = dmap[i] * stride; }
    __device__ __forceinline__ void sym(int j, int F, int C) {
        if (lane_id() == 0)
            rows[cur + dmap[j]] = (((4096u - (uint32_t)C) & 4095u) << 20) | ((uint32_t)(F - 1) << 8) | (uint32_t)dmap[j];
    }
};
constexpr uint32_t kAbsent = 0xFFFFFFFFu;  // not a valid entry: it would be C = 1 with F = 4096 (and loses every minimum)

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
__device__ __forceinline__ int decode_o0(Stream& S, int32_t p, const uint32_t* R, uint8_t* out, int32_t n) {
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

// unsigned minimum over a 16-lane row (SMALL: lanes 8..15 mirror lanes 0..7, three stages)
template <bool SMALL>
__device__ __forceinline__ uint32_t row_min(uint32_t v) {
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
    v = t < v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);  // quad_perm [2,3,0,1]
    v = t < v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);  // row_half_mirror
    v = t < v ? t : v;
    if (!SMALL) {
        t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false);  // row_mirror
        v = t < v ? t : v;
    }
    return v;
}

// D14#uncompress L9-70.  rows[nd][16*M] over the dense alphabet (index 0 = byte 0); an entry is
//   (4096 - C) & 4095 << 20 | (F-1) << 8 | k
// so that v = (x << 20) + entry carries (slot - C) mod 4096 in its top 12 bits.  The slot belongs to the symbol with the
// largest C <= slot, whose slot - C is the smallest of the row (a symbol with C > slot wraps to 4096 - (C - slot), which
// is larger): the unsigned minimum of the row's v is the winner's word, slot - C on top, F-1 and the symbol below.  Lane
// s of a state's row takes the minimum over the symbols s, s+16, ... (two instructions per entry), an unsigned-min
// butterfly over the row does the rest.  Symbols a context does not have carry a copy of one it has.
// MS > 0: square rows[nd][16*MS], MS known at compile time (the next row is fetched while the state renormalises).
// MS == 0: ragged rows (FillRagged): rowinfo[k] = entry offset | 16-entry units << 24 of context k's row.
template <int MS>
__device__ __forceinline__ int decode_o1(Stream& S, int32_t p, const uint32_t* rows, const uint32_t* rowinfo, const uint8_t* syml,
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
    uint32_t Ml = 0;               // ragged: index of the last 16-entry unit of this state's current row
    if (MS > 0) {
#pragma unroll
        for (int m = 0; m < MS; m++) e[m] = rows[roff + 16 * m];
    } else {
        const uint32_t info = rowinfo[0];
        roff = (info & 0xFFFFFFu) + (uint32_t)s;
        Ml = (info >> 24) - 1u;
    }
    auto step = [&](bool live) -> uint32_t {
        const uint32_t xs = x << 20;
        uint32_t v;
        if (MS > 0) {
            v = xs + e[0];
#pragma unroll
            for (int m = 1; m < MS; m++) {
                const uint32_t t = xs + e[m];
                v = t < v ? t : v;
            }
        } else {
            // up to the longest row of the four states, four entries per round trip; an index past a row's end
            // repeats its last unit, which leaves the minimum as it is
            v = xs + rows[roff];
            if (__ballot(Ml != 0u) != 0) {
                for (uint32_t m = 1; __ballot(m <= Ml) != 0; m += 4) {
                    uint32_t em[4];
#pragma unroll
                    for (uint32_t c = 0; c < 4; c++) em[c] = rows[roff + 16u * (m + c < Ml ? m + c : Ml)];
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const uint32_t t = xs + em[c];
                        v = t < v ? t : v;
                    }
                }
            }
        }
        v = row_min<false>(v);
        const uint32_t k = v & 255u;
        if (live) {
            if (MS > 0) {
                roff = k * stride + (uint32_t)s;
#pragma unroll
                for (int m = 0; m < MS; m++) e[m] = rows[roff + 16 * m];
            } else {
                const uint32_t info = rowinfo[k];
                roff = (info & 0xFFFFFFu) + (uint32_t)s;
                Ml = (info >> 24) - 1u;
            }
            x = (((v >> 8) & 4095u) + 1u) * (x >> 12) + (v >> 20);
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


// ---- order 1 over a dense alphabet of at most 16 symbols (quality series): the narrow path ----
// An entry is  (4096 - C) & 4095 in bits 20..31 | F << 7 | symbol << 2  (F = 1..4096 in 13 bits), so that
//   v = (x << 20) + entry
// carries (slot - C) mod 4096 in its top 12 bits.  The symbol the slot belongs to is the one with the largest C <= slot:
// its (slot - C) is the smallest of the row (a symbol with C > slot wraps to 4096 - (C - slot), which is larger), so an
// unsigned MIN butterfly over the row delivers the winner's word: slot - C on top, F and the symbol below — one
// instruction per lane instead of a five-instruction range test, and no select.  Symbols a context does not have carry a
// copy of one it has.
//   SMALL (alphabet <= 8): ent[symbol][context] (8 x 8 words); lanes 8..15 of a row mirror lanes 0..7, three butterfly
//     stages, and the next entry's address is one v_and_or (symbol << 2 selects the word in the lane's own 32 bytes);
//   otherwise (<= 16):     ent[context][symbol] (16 x 16 words), four stages.
// The main loop is hand-scheduled (narrow_block): 256 steps per call, decoded symbols pushed through a DPP shift
// register and staged in LDS, flushed as one 16-byte store per lane (a byte store per 16 steps kept the vector-memory
// counter busy: the next renormalisation's wait for its window also waited for that store).
struct FillNarrow {
    uint32_t* ent;
    const uint8_t* dmap;
    int ctx_stride, sym_stride;  // words
    int cur;
    __device__ __forceinline__ void ctx(int i) { cur = dmap[i] * ctx_stride; }
    __device__ __forceinline__ void sym(int j, int F, int C) {
        if (lane_id() == 0)
            ent[cur + dmap[j] * sym_stride] =
                (((4096u - (uint32_t)C) & 4095u) << 20) | ((uint32_t)F << 7) | ((uint32_t)dmap[j] << 2);
    }
};
// every absent entry of a context becomes a copy of the context's first present entry (0 for a context the table
// never defined, reachable from a corrupt stream only: it decodes to garbage without leaving the tables)
__device__ __forceinline__ void narrow_fix(uint32_t* ent, int nsl, int ctx_stride, int sym_stride) {
    const int c = lane_id();
    if (c < nsl) {
        uint32_t first = 0;
        for (int j = nsl - 1; j >= 0; j--) {
            const uint32_t v = ent[c * ctx_stride + j * sym_stride];
            if (v != kAbsent) first = v;
        }
        for (int j = 0; j < nsl; j++)
            if (ent[c * ctx_stride + j * sym_stride] == kAbsent) ent[c * ctx_stride + j * sym_stride] = first;
    }
}

struct __attribute__((packed)) PackedU128 { uint32_t a, b, c, d; };

// stage[g][256] (dense symbols of steps i0 .. i0+255 of state g) -> out[g * q + i0 ...], 16 bytes per lane
__device__ __forceinline__ void flush_stage(const uint8_t* stage, const uint8_t* syml, uint8_t* out, int32_t q, int32_t i0) {
    const int l = lane_id();
    const int g = l >> 4, part = l & 15;
    const uint4 raw = *reinterpret_cast<const uint4*>(stage + g * 256 + part * 16);
    const uint32_t r[4] = {raw.x, raw.y, raw.z, raw.w};
    uint32_t o[4];
#pragma unroll
    for (int c = 0; c < 4; c++)
        o[c] = (uint32_t)syml[r[c] & 255u] | ((uint32_t)syml[(r[c] >> 8) & 255u] << 8) |
               ((uint32_t)syml[(r[c] >> 16) & 255u] << 16) | ((uint32_t)syml[r[c] >> 24] << 24);
    *reinterpret_cast<PackedU128*>(out + (size_t)g * q + i0 + part * 16) = PackedU128{o[0], o[1], o[2], o[3]};
}

#ifndef HIP_SIM
// One decode step of the narrow path.  KW: the register this step's winner word is written to, KO: the other one of the
// pair (it holds the previous step's word in every lane; the row_shr:1 move turns it into the shift register of the
// chunk's words, lane j = j steps ago).  A DPP source may not be read by the instruction right after the one that wrote
// it: the two independent instructions of the step (x >> 12, the shift-register move) fill two of those slots.
#define NB_MIN(DST, CTRL) "v_min_u32_dpp %[" DST "], %[v], %[v] " CTRL " row_mask:0xf bank_mask:0xf\n\t"
#ifdef NB_LATE_FILL
#define NB_FILL1 "s_nop 0\n\t"
#define NB_FILL2(KW, KO) "s_nop 0\n\t"
#define NB_LATE(KW, KO)                                                                    \
    "v_mov_b32_dpp %[" KW "], %[" KO "] row_shr:1 row_mask:0xf bank_mask:0xf\n\t"           \
    "v_lshrrev_b32 %[xs], 12, %[x]\n\t"
#define NB_LAST_PUSH
#else
#define NB_FILL1 "v_lshrrev_b32 %[xs], 12, %[x]\n\t"
#define NB_FILL2(KW, KO) "v_mov_b32_dpp %[" KO "], %[" KW "] row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define NB_LATE(KW, KO)
#define NB_LAST_PUSH "v_mov_b32_dpp %[k1], %[k0] row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#endif
#define NB_BUTTERFLY_SMALL(KW, KO)                                                         \
    NB_FILL1                                                                                \
    NB_MIN("v", "quad_perm:[1,0,3,2]")                                                      \
    NB_FILL2(KW, KO)                                                                        \
    NB_MIN("v", "quad_perm:[2,3,0,1]")                                                      \
    "s_nop 0\n\t"                                                                           \
    NB_MIN(KW, "row_half_mirror")                                                           \
    "v_and_or_b32 %[t], %[" KW "], 28, %[lb]\n\t"
#define NB_BUTTERFLY_MID(KW, KO)                                                           \
    NB_FILL1                                                                                \
    NB_MIN("v", "quad_perm:[1,0,3,2]")                                                      \
    NB_FILL2(KW, KO)                                                                        \
    NB_MIN("v", "quad_perm:[2,3,0,1]")                                                      \
    "s_nop 0\n\t"                                                                           \
    NB_MIN("v", "row_half_mirror")                                                          \
    "s_nop 0\n\t"                                                                           \
    NB_MIN(KW, "row_mirror")                                                                \
    "v_and_b32 %[t], 60, %[" KW "]\n\t"                                                     \
    "v_lshl_add_u32 %[t], %[t], 4, %[lb]\n\t"
#define NB_STEP(N, BUTTERFLY, KW, KO)                                                      \
    "s_waitcnt lgkmcnt(0)\n\t"                                                              \
    "v_lshl_add_u32 %[v], %[x], 20, %[e]\n\t"                                               \
    BUTTERFLY(KW, KO)                                                                       \
    "ds_read_b32 %[e], %[t]\n\t"                                                            \
    "v_lshrrev_b32 %[d], 20, %[" KW "]\n\t"                                                 \
    "v_bfe_u32 %[f], %[" KW "], 7, 13\n\t"                                                  \
    NB_LATE(KW, KO)                                                                         \
    "v_mad_u32_u24 %[x], %[f], %[xs], %[d]\n\t"                                             \
    "v_cmp_gt_u32_e32 vcc, 0x800000, %[x]\n\t"                                              \
    "s_cbranch_vccz .Ldone" #N "_%=\n\t"                                                    \
    "v_cmp_gt_u32_e64 %[n2], %[c15], %[x]\n\t"                                              \
    "v_cndmask_b32_e32 %[sel], %[selno], %[selone], vcc\n\t"                                \
    "s_and_b64 vcc, vcc, %[ktop]\n\t"                                                       \
    "v_mbcnt_lo_u32_b32 %[a], vcc_lo, 0\n\t"                                                \
    "v_mbcnt_hi_u32_b32 %[a], vcc_hi, %[a]\n\t"                                             \
    "s_bcnt1_i32_b64 %[stot], vcc\n\t"                                                      \
    "s_waitcnt vmcnt(0)\n\t"                                                                \
    "v_alignbyte_b32 v102, v101, v100, %[a]\n\t"                                            \
    "s_cmp_lg_u64 %[n2], 0\n\t"                                                             \
    "s_cbranch_scc1 .Ltwo" #N "_%=\n\t"                                                     \
    ".Lapply" #N "_%=:\n\t"                                                                 \
    "v_perm_b32 %[x], %[x], v102, %[sel]\n\t"                                               \
    "v_add_u32 %[curv], %[stot], %[curv]\n\t"                                               \
    "global_load_dwordx2 v[100:101], %[curv], %[sb]\n\t"                                    \
    ".Ldone" #N "_%=:\n\t"
// the rare case: some state needs a second byte (it decoded a symbol of frequency < 16)
#define NB_TWO(N)                                                                          \
    ".Ltwo" #N "_%=:\n\t"                                                                   \
    "s_mov_b64 vcc, %[n2]\n\t"                                                              \
    "v_cndmask_b32_e32 %[sel], %[sel], %[seltwo], vcc\n\t"                                  \
    "s_and_b64 vcc, vcc, %[ktop]\n\t"                                                       \
    "v_mbcnt_lo_u32_b32 %[a], vcc_lo, %[a]\n\t"                                             \
    "v_mbcnt_hi_u32_b32 %[a], vcc_hi, %[a]\n\t"                                             \
    "s_bcnt1_i32_b64 %[s2], vcc\n\t"                                                        \
    "s_add_u32 %[stot], %[stot], %[s2]\n\t"                                                 \
    "v_lshlrev_b32 %[a], 3, %[a]\n\t"                                                       \
    "v_lshrrev_b64 v[102:103], %[a], v[100:101]\n\t"                                        \
    "s_branch .Lapply" #N "_%=\n\t"
#define NB_16(M, ...)                                                                      \
    M(0, __VA_ARGS__, "k0", "k1") M(1, __VA_ARGS__, "k1", "k0") M(2, __VA_ARGS__, "k0", "k1") M(3, __VA_ARGS__, "k1", "k0")   \
    M(4, __VA_ARGS__, "k0", "k1") M(5, __VA_ARGS__, "k1", "k0") M(6, __VA_ARGS__, "k0", "k1") M(7, __VA_ARGS__, "k1", "k0")   \
    M(8, __VA_ARGS__, "k0", "k1") M(9, __VA_ARGS__, "k1", "k0") M(10, __VA_ARGS__, "k0", "k1") M(11, __VA_ARGS__, "k1", "k0") \
    M(12, __VA_ARGS__, "k0", "k1") M(13, __VA_ARGS__, "k1", "k0") M(14, __VA_ARGS__, "k0", "k1") M(15, __VA_ARGS__, "k1", "k0")
#define NB_TWOS                                                                            \
    NB_TWO(0) NB_TWO(1) NB_TWO(2) NB_TWO(3) NB_TWO(4) NB_TWO(5) NB_TWO(6) NB_TWO(7)         \
    NB_TWO(8) NB_TWO(9) NB_TWO(10) NB_TWO(11) NB_TWO(12) NB_TWO(13) NB_TWO(14) NB_TWO(15)
#define NB_BODY(BUTTERFLY)                                                                 \
    "v_mov_b32 v100, %[wlo]\n\t"                                                            \
    "v_mov_b32 v101, %[whi]\n\t"                                                            \
    "s_movk_i32 %[cnt], 16\n\t"                                                             \
    ".Lchunk_%=:\n\t"                                                                       \
    NB_16(NB_STEP, BUTTERFLY)                                                               \
    NB_LAST_PUSH                                                                            \
    "s_sub_u32 %[cnt], %[cnt], 1\n\t"                                                       \
    "v_bfe_u32 %[t], %[k1], 2, 4\n\t"                                                       \
    "ds_write_b8 %[st], %[t]\n\t"                                                           \
    "v_add_u32 %[st], 16, %[st]\n\t"                                                        \
    "s_cmp_lg_u32 %[cnt], 0\n\t"                                                            \
    "s_cbranch_scc1 .Lchunk_%=\n\t"                                                         \
    "s_branch .Lend_%=\n\t"                                                                 \
    NB_TWOS                                                                                 \
    ".Lend_%=:\n\t"                                                                         \
    "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"                                                     \
    "v_mov_b32 %[wlo], v100\n\t"                                                            \
    "v_mov_b32 %[whi], v101\n\t"
#endif

// 256 steps of all four states.  x: state, e: the lane's entry of the state's context, curv: byte cursor, wlo/whi: the
// 8 stream bytes at the cursor.  The caller guarantees 256 * 8 + 8 readable bytes behind the cursor.
template <bool SMALL>
__device__ __forceinline__ void narrow_block(uint32_t& x, uint32_t& e, uint32_t& curv, uint32_t& wlo, uint32_t& whi,
                                             uint32_t lane_base, uint32_t stage_addr, const uint8_t* src) {
#ifndef HIP_SIM
    uint32_t v, xs, k0 = 0, k1 = 0, t, d, f, a, sel, cnt, stot, s2;
    unsigned long long n2;
    const uint32_t selno = 0x07060504u, selone = 0x06050400u, seltwo = 0x05040001u;
    if (SMALL) {
        asm volatile(NB_BODY(NB_BUTTERFLY_SMALL)
                     : [x] "+v"(x), [e] "+v"(e), [curv] "+v"(curv), [wlo] "+v"(wlo), [whi] "+v"(whi), [st] "+v"(stage_addr),
                       [k0] "+v"(k0), [k1] "+v"(k1), [v] "=&v"(v), [xs] "=&v"(xs), [t] "=&v"(t), [d] "=&v"(d), [f] "=&v"(f),
                       [a] "=&v"(a), [sel] "=&v"(sel), [cnt] "=&s"(cnt), [stot] "=&s"(stot), [s2] "=&s"(s2), [n2] "=&s"(n2)
                     : [lb] "v"(lane_base), [sb] "s"(src), [ktop] "s"(kTop), [selno] "v"(selno), [selone] "v"(selone),
                       [seltwo] "v"(seltwo), [c15] "s"(1u << 15)
                     : "vcc", "scc", "memory", "v100", "v101", "v102", "v103");
    } else {
        asm volatile(NB_BODY(NB_BUTTERFLY_MID)
                     : [x] "+v"(x), [e] "+v"(e), [curv] "+v"(curv), [wlo] "+v"(wlo), [whi] "+v"(whi), [st] "+v"(stage_addr),
                       [k0] "+v"(k0), [k1] "+v"(k1), [v] "=&v"(v), [xs] "=&v"(xs), [t] "=&v"(t), [d] "=&v"(d), [f] "=&v"(f),
                       [a] "=&v"(a), [sel] "=&v"(sel), [cnt] "=&s"(cnt), [stot] "=&s"(stot), [s2] "=&s"(s2), [n2] "=&s"(n2)
                     : [lb] "v"(lane_base), [sb] "s"(src), [ktop] "s"(kTop), [selno] "v"(selno), [selone] "v"(selone),
                       [seltwo] "v"(seltwo), [c15] "s"(1u << 15)
                     : "vcc", "scc", "memory", "v100", "v101", "v102", "v103");
    }
#endif
}

// D14#uncompress L9-70 over narrow tables.  stage: DecFixedLds::stage.
template <bool SMALL>
__device__ __forceinline__ int decode_o1_narrow(Stream& S, int32_t p, const uint32_t* ent, uint8_t* stage, const uint8_t* syml, uint8_t* out,
                                int32_t n) {
    const int l = lane_id();
    const int g = l >> 4, s = l & 15;
    if (p + 16 > S.len) return CRAMGPU_E_CORRUPT;
    const Payload P{uniform_ptr(S.src), UNIFORM(S.len)};
    uint32_t x = (uint32_t)P.window(p + 4 * g);
    int32_t cur = UNIFORM(p) + 16;
    uint32_t curv = in_vgpr((uint32_t)cur);
    unsigned long long w = P.window(cur);
    const int32_t q = n >> 2;
    const uint32_t* lane_ent = ent + (SMALL ? (s & 7) * 8 : s);  // word k (SMALL) or 16 k of it: this lane's entry in context k
    uint32_t e = lane_ent[0];
    auto step = [&](bool live) -> uint32_t {
        const uint32_t v = row_min<SMALL>((x << 20) + e);
        if (live) {
            const uint32_t k = (v >> 2) & 15u;
            e = lane_ent[SMALL ? k : 16u * k];
            x = ((v >> 7) & 8191u) * (x >> 12) + (v >> 20);
        }
        renorm(x, cur, curv, w, P, live);
        return (v >> 2) & 15u;
    };
    int32_t i = 0;
    // blocks of 256 steps while the stream has 8 bytes per step and a window left behind the cursor
    while (i + 256 <= q && cur + 256 * 8 + 8 <= S.len) {
#ifdef HIP_SIM
        for (int c = 0; c < 16; c++) {
            uint32_t keep = 0;
            for (int j = 0; j < kChunk; j++) keep = row_push(keep, step(true));
            stage[g * 256 + c * 16 + 15 - s] = (uint8_t)keep;
        }
        __syncthreads();
#else
        typedef __attribute__((address_space(3))) const void* LdsPtr;
        uint32_t wlo = (uint32_t)w, whi = (uint32_t)(w >> 32);
        narrow_block<SMALL>(x, e, curv, wlo, whi, (uint32_t)(uintptr_t)(LdsPtr)lane_ent,
                            (uint32_t)(uintptr_t)(LdsPtr)stage + (uint32_t)(g * 256 + 15 - s), P.src);
        w = ((unsigned long long)whi << 32) | wlo;
        cur = UNIFORM(curv);
#endif
        flush_stage(stage, syml, out, q, i);
#ifdef HIP_SIM
        __syncthreads();
#endif
        i += 256;
    }
    for (; i + kChunk <= q; i += kChunk) {
        if (cur > S.len) return CRAMGPU_E_CORRUPT;
        uint32_t keep = 0;
#pragma unroll
        for (int j = 0; j < kChunk; j++) keep = row_push(keep, step(true));
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

__device__ __forceinline__ void decode_block(const RansArgs& a, int b, uint8_t* smem, uint8_t* gs) {
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
            Stream S{uniform_ptr(src), UNIFORM(len), -64, 0ull, 0ull};
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
                MarkO1 mk{L->present, L->nsym, 0, 0};
                int32_t p1 = p;
                rc = read_rows_o1(S, p1, mk);
                mk.finish();
                __syncthreads();
                if (!rc) {
                    // dense alphabet in byte order, byte 0 first; ragged row geometry (by byte) alongside: 64 bytes a
                    // round, rank and row offset from a ballot and a prefix sum over the lanes
                    int nd = 0, total = 0;
                    for (int j0 = 0; j0 < 256; j0 += 64) {
                        const int j = j0 + l;
                        const bool pres = L->present[j] != 0;
                        const uint32_t ns = L->nsym[j] < 256u ? L->nsym[j] : 256u;
                        const int units = !pres ? 0 : ns ? (int)((ns + 15u) >> 4) : 1;
                        const unsigned long long m = __ballot(pres);
                        const int rank = nd + __popcll(m & ((1ull << l) - 1ull));
                        int incl = units;
                        for (int d = 1; d < 64; d <<= 1) {
                            const int t = __shfl_up(incl, d);
                            if (l >= d) incl += t;
                        }
                        L->rowinfo[j] = pres ? (uint32_t)(total + 16 * (incl - units)) | ((uint32_t)units << 24) : (1u << 24);
                        L->dmap[j] = pres ? (uint8_t)rank : (uint8_t)0;
                        if (pres) L->syml[rank] = (uint8_t)j;
                        nd += __popcll(m);
                        total += 16 * __shfl(incl, 63);
                    }
                    __syncthreads();
                    if (nd <= 16) {  // narrow tables: 8 x 8 or 16 x 16 words
                        const bool small = nd <= 8;
                        const int nsl = small ? 8 : 16;
                        uint32_t* r = reinterpret_cast<uint32_t*>(tab);
                        rows_clear(r, nsl * nsl);
                        __syncthreads();
                        FillNarrow v{r, L->dmap, small ? 1 : 16, small ? 8 : 1, 0};
                        rc = read_rows_o1(S, p, v);
                        __syncthreads();
                        if (!rc) {
                            narrow_fix(r, nsl, small ? 1 : 16, small ? 8 : 1);
                            __syncthreads();
                            rc = small ? decode_o1_narrow<true>(S, p, r, L->stage, L->syml, out, raw)
                                       : decode_o1_narrow<false>(S, p, r, L->stage, L->syml, out, raw);
                        }
                    } else if (nd <= 48) {  // square rows, 16 * M entries each, always in LDS (<= 9 KiB)
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
                            if (M == 2)
                                rc = decode_o1<2>(S, p, r, nullptr, L->syml, out, raw);
                            else
                                rc = decode_o1<3>(S, p, r, nullptr, L->syml, out, raw);
                        }
                    } else {  // ragged rows, in LDS when they fit (total <= 256 * 256 entries = the global scratch)
                        const bool in_lds = total * 4 <= a.tab_bytes;
                        for (int i = l; i < 256; i += 64) L->syml[i] = (uint8_t)i;  // ragged rows carry byte values
                        if (in_lds) {
                            uint32_t* r = reinterpret_cast<uint32_t*>(tab);
                            for (int i = l; i < total; i += 64) r[i] = 0;
                            __syncthreads();
                            FillRagged v{r, L->rowinfo, 0, 0, 0, 0};
                            rc = read_rows_o1(S, p, v);
                            v.finish();
                            __syncthreads();
                            if (!rc) rc = decode_o1<0>(S, p, r, L->rowinfo, L->syml, out, raw);
                        } else {
                            uint32_t* r = reinterpret_cast<uint32_t*>(gs);
                            for (int i = l; i < total; i += 64) r[i] = 0;
                            __threadfence();
                            __syncthreads();
                            FillRagged v{r, L->rowinfo, 0, 0, 0, 0};
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

// A launch mixes a few long chains (a quality block of a container: 25 ms) with many short ones, and a SIMD shared by
// two busy waves runs each of them slower.  The launch lasts as long as its slowest long chain, so the short blocks must
// not pile up beside some of the long ones: the first wave of the launch on a SIMD takes blocks from the front of the
// longest-first permutation, every later wave on that SIMD from its back, and with two waves per SIMD the short
// blocks spread over all SIMDs evenly.  One 64-bit counter holds both ends (blocks taken from the front in the low
// word, from the back in the high word): a ticket is good while their sum is below the number of blocks, so the two
// ends meet without handing a block out twice, and a wave whose end has run dry simply works from the other side.
__device__ __forceinline__ bool first_wave_on_simd(uint32_t* arrivals, uint32_t launch_no) {
#ifdef HIP_SIM
    (void)arrivals;
    (void)launch_no;
    return true;
#else
    if (arrivals == nullptr) return true;
    const uint32_t hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_REG_HW_ID: SIMD 5:4, CU/SH/SE 15:8
    const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));  // HW_REG_XCC_ID
    const uint32_t key = ((xcc & 7u) << 10) | (((hw >> 8) & 255u) << 2) | ((hw >> 4) & 3u);
    uint32_t seen = 0;  // the table keeps the number of the last launch whose wave came by: nothing to clear per launch
    if (threadIdx.x == 0) seen = atomicMax(&arrivals[key], launch_no);
    return (uint32_t)__shfl((int)seen, 0) < launch_no;
#endif
}

__global__ __launch_bounds__(64) void rans_decode_kernel(RansArgs a) {
    CRAMGPU_DYN_LDS(smem);
    uint8_t* gs = a.gscratch + (size_t)blockIdx.x * DEC_GSCRATCH_BYTES;
    unsigned long long* ends = reinterpret_cast<unsigned long long*>(a.ticket);
    const bool front = first_wave_on_simd(a.simd_arrivals, a.launch_no);
    const uint32_t n = (uint32_t)a.n_blocks;
    for (;;) {
        uint32_t head = 0, tail = 0;
        if (threadIdx.x == 0) {
            const unsigned long long prev = atomicAdd(ends, front ? 1ull : (1ull << 32));
            head = (uint32_t)prev;
            tail = (uint32_t)(prev >> 32);
        }
        head = __shfl(head, 0);
        tail = __shfl(tail, 0);
        if (head >= n || tail >= n || head + tail >= n) break;
        decode_block(a, a.perm[front ? head : n - 1u - tail], smem, gs);
        __syncthreads();
    }
}

}  // namespace

size_t rans_decode_fixed_lds() { return sizeof(DecFixedLds); }

hipError_t launch_rans_decode(const RansArgs& a, int n_waves, size_t lds_bytes, hipStream_t s) {
    hipLaunchKernelGGL(rans_decode_kernel, dim3(n_waves), dim3(64), lds_bytes, s, a);
    return hipGetLastError();
}
