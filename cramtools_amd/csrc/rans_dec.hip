// K2 — CRAM 3.0 rANS 4x8 decoder (order 0 and order 1) for gfx950.
//
// Reference (htsjdk!encoding/rans/*):
//   RANS#uncompress L29-56, Frequencies#readStats_o0 L13-46, #readStats_o1 L50-106, Decoding L24-90,
//   D04#uncompress L9-85, D14#uncompress L9-70
//
// Like the encoder this is a resident grid of single-wave workgroups pulling blocks from a ticket counter.
// The four states of a stream share one byte cursor (renormalisation order rans0..rans3 per step), so a step is
//   symbol lookup -> state update -> per-state byte need (0..2) -> ballot/popcount prefix over the four states
//   -> byte fetch from an LDS ring holding the compressed stream -> cursor advance.
// Order 0: lanes 0-3 own the states; slot -> (symbol, freq-1, slot-cum) comes from one 4096 x u32 LDS table.
// Order 1: 256 contexts x 4096 slots does not fit LDS, so the cumulative tables are kept as rows of (cum, freq)
//   over the block's dense alphabet and each state is given a 16-lane DPP row: every lane tests the slot against
//   its own symbols, the single winner computes the next state, and a 4-step DPP OR-butterfly hands it to the
//   row.  The next context row is addressed directly by the dense symbol index, and its fetch overlaps the
//   renormalisation of the current step.
#include "internal.h"

namespace {

constexpr int kRing = 4096;  // LDS ring of compressed bytes

struct DecFixedLds {
    uint8_t ring[kRing];
    uint8_t dmap[256];
    uint8_t syml[256];
    uint8_t present[256];
};

__device__ __forceinline__ int lane_id() { return (int)threadIdx.x; }
__device__ __forceinline__ uint32_t mbcnt(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// OR over the 16 lanes of a DPP row (butterfly: quad xor 1, quad xor 2, half mirror, row mirror)
__device__ __forceinline__ uint32_t row_or(uint32_t v) {
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);  // row_half_mirror
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false);  // row_mirror
    return v;
}

// compressed-stream reader: bytes [fill-kRing, fill) of the stream are resident in the ring
struct Stream {
    const uint8_t* src;   // 16-byte aligned start of the stream
    int32_t len;          // stream bytes
    int32_t fill;         // multiple of 1024
    uint8_t* ring;
    __device__ __forceinline__ void refill() {
        const int32_t i = fill + lane_id() * 16;
        if (i < len) {  // the arena is padded to 16 bytes, so a partial last chunk may be read whole
            uint4 v = *reinterpret_cast<const uint4*>(src + i);
            *reinterpret_cast<uint4*>(ring + (i & (kRing - 1))) = v;
        }
        fill += 1024;
        __syncthreads();
    }
    __device__ __forceinline__ void reset() {
        fill = 0;
        __syncthreads();
    }
    // byte at stream offset p (uniform p), sequential access
    __device__ __forceinline__ uint32_t at(int32_t p) {
        while (p >= fill) refill();
        return ring[p & (kRing - 1)];
    }
};

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
    __device__ __forceinline__ void ctx(int i) { present[i] = 1; }
    __device__ __forceinline__ void sym(int j, int, int) { present[j] = 1; }
};
struct FillO1 {
    uint32_t* rows;  // [n][n] C | F<<16
    const uint8_t* dmap;
    int n;
    int cur;
    __device__ __forceinline__ void ctx(int i) { cur = dmap[i] * n; }
    __device__ __forceinline__ void sym(int j, int F, int C) {
        if (lane_id() == 0) rows[cur + dmap[j]] = (uint32_t)C | ((uint32_t)F << 16);
    }
};

// D04#uncompress L9-85
__device__ int decode_o0(Stream& S, int32_t p, const uint32_t* R, uint8_t* out, int32_t n) {
    const int l = lane_id();
    if (p + 16 > S.len) return CRAMGPU_E_CORRUPT;
    uint32_t x = RANS_BYTE_L;
    S.at(p);
    while (S.fill < p + 16 && S.fill < ((S.len + 1023) & ~1023)) S.refill();
    if (l < 4) {
        const int32_t a = p + 4 * l;
        x = (uint32_t)S.ring[a & (kRing - 1)] | ((uint32_t)S.ring[(a + 1) & (kRing - 1)] << 8) |
            ((uint32_t)S.ring[(a + 2) & (kRing - 1)] << 16) | ((uint32_t)S.ring[(a + 3) & (kRing - 1)] << 24);
    }
    int32_t cur = p + 16;
    const int32_t out_end = n & ~3;
    for (int32_t i = 0; i < out_end; i += 4) {
        if (S.fill - cur < 2048 && S.fill < S.len) S.refill();
        const uint32_t e = R[x & 4095u];
        if (l < 4) {
            out[i + l] = (uint8_t)e;
            x = (((e >> 8) & 4095u) + 1u) * (x >> 12) + (e >> 20);
        }
        const bool n1 = x < RANS_BYTE_L, n2 = x < (1u << 15);
        const unsigned long long m1 = __ballot(n1), m2 = __ballot(n2);
        const int32_t tot = __popcll(m1) + __popcll(m2);
        if (cur + tot > S.len) return CRAMGPU_E_CORRUPT;
        const int32_t a = cur + (int32_t)(mbcnt(m1) + mbcnt(m2));
        const uint32_t b0 = S.ring[a & (kRing - 1)], b1 = S.ring[(a + 1) & (kRing - 1)];
        x = n2 ? ((x << 16) | (b0 << 8) | b1) : (n1 ? ((x << 8) | b0) : x);
        cur += tot;
    }
    if (l < (n & 3)) out[out_end + l] = (uint8_t)R[x & 4095u];
    return 0;
}

// D14#uncompress L9-70.  rows[n][n] over the dense alphabet (index 0 = byte 0), M = ceil(n/16).
__device__ int decode_o1(Stream& S, int32_t p, const uint32_t* rows, int nd, const uint8_t* syml, uint8_t* out,
                         int32_t n) {
    const int l = lane_id();
    const int g = l >> 4, s = l & 15;
    if (p + 16 > S.len) return CRAMGPU_E_CORRUPT;
    S.at(p);
    while (S.fill < p + 16 && S.fill < ((S.len + 1023) & ~1023)) S.refill();
    uint32_t x;
    {
        const int32_t a = p + 4 * g;
        x = (uint32_t)S.ring[a & (kRing - 1)] | ((uint32_t)S.ring[(a + 1) & (kRing - 1)] << 8) |
            ((uint32_t)S.ring[(a + 2) & (kRing - 1)] << 16) | ((uint32_t)S.ring[(a + 3) & (kRing - 1)] << 24);
    }
    int32_t cur = p + 16;
    const int32_t q = n >> 2;
    const int M = (nd + 15) >> 4;
    const unsigned long long top = 0x8000800080008000ull;  // one representative lane per state
    uint32_t rowbase = 0;                                   // context 0
    int32_t o = g * q;
    const int32_t steps = q + (n - 4 * q);                  // q common steps, then state 3 alone
    for (int32_t i = 0; i < steps; i++, o++) {
        if (S.fill - cur < 2048 && S.fill < S.len) S.refill();
        const bool live = i < q || g == 3;
        const uint32_t slot = x & 4095u, xh = x >> 12;
        uint32_t cand = 0, symd = 0;
        for (int m = 0; m < M; m++) {
            const int k = s + 16 * m;
            if (k < nd) {
                const uint32_t cf = rows[rowbase + k];
                const uint32_t off = slot - (cf & 0xFFFFu), F = cf >> 16;
                if (off < F) {
                    cand = F * xh + off;
                    symd = (uint32_t)k;
                }
            }
        }
        cand = row_or(cand);
        symd = row_or(symd);
        if (__ballot(live && cand == 0) != 0) return CRAMGPU_E_CORRUPT;  // context or slot absent from the tables
        if (live) {
            if (s == 0) out[o] = syml[symd];
            x = cand;
            rowbase = symd * (uint32_t)nd;
        }
        const bool n1 = live && x < RANS_BYTE_L, n2 = live && x < (1u << 15);
        const unsigned long long m1 = __ballot(n1) & top, m2 = __ballot(n2) & top;
        const int32_t tot = __popcll(m1) + __popcll(m2);
        if (cur + tot > S.len) return CRAMGPU_E_CORRUPT;
        const int32_t a = cur + (int32_t)(mbcnt(m1) + mbcnt(m2));
        const uint32_t b0 = S.ring[a & (kRing - 1)], b1 = S.ring[(a + 1) & (kRing - 1)];
        x = n2 ? ((x << 16) | (b0 << 8) | b1) : (n1 ? ((x << 8) | b0) : x);
        cur += tot;
    }
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
            Stream S{src, len, 0, L->ring};
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
                for (int i = l; i < 256; i += 64) L->present[i] = i == 0;
                __syncthreads();
                MarkO1 mk{L->present};
                int32_t p1 = p;
                rc = read_rows_o1(S, p1, mk);
                __syncthreads();
                if (!rc) {
                    // dense alphabet in byte order, byte 0 first
                    int nd = 0;
                    if (l == 0) {
                        for (int j = 0; j < 256; j++) {
                            if (L->present[j]) {
                                L->dmap[j] = (uint8_t)nd;
                                L->syml[nd] = (uint8_t)j;
                                nd++;
                            } else {
                                L->dmap[j] = 0;
                            }
                        }
                    }
                    nd = __shfl(nd, 0);
                    __syncthreads();
                    const bool in_lds = nd * nd * 4 <= a.tab_bytes;
                    uint32_t* rows = in_lds ? reinterpret_cast<uint32_t*>(tab) : reinterpret_cast<uint32_t*>(gs);
                    if (in_lds) {
                        uint32_t* r = reinterpret_cast<uint32_t*>(tab);
                        for (int i = l; i < nd * nd; i += 64) r[i] = 0;
                    } else {
                        uint32_t* r = reinterpret_cast<uint32_t*>(gs);
                        for (int i = l; i < nd * nd; i += 64) r[i] = 0;
                    }
                    __threadfence();
                    __syncthreads();
                    S.reset();
                    FillO1 v{rows, L->dmap, nd, 0};
                    rc = read_rows_o1(S, p, v);
                    __threadfence();
                    __syncthreads();
                    if (!rc) {
                        if (in_lds)
                            rc = decode_o1(S, p, reinterpret_cast<uint32_t*>(tab), nd, L->syml, out, raw);
                        else
                            rc = decode_o1(S, p, reinterpret_cast<uint32_t*>(gs), nd, L->syml, out, raw);
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
