// G — gunzip of gzip members on the device (the decode halves): the external blocks the
// reference stores with method GZIP (htsjdk!io/ExternalCompression#gunzip L41-48 = java.util.zip.GZIPInputStream, called
// from htsjdk!structure/Block#uncompress L228-262 for the 18 gzip series of a CRAM container) and the BGZF blocks of a BAM
// file (htsjdk!util/BlockCompressedInputStream in front of src/main/java/htsjdk/samtools/cram/paralell/Bam_OBA_Supplier.java).
// The formats are public: RFC 1952 (gzip member: header, DEFLATE data, CRC-32, ISIZE), RFC 1951 (DEFLATE), the SAM
// specification §4.1 (BGZF = gzip members of <= 64 KiB with a 'BC' extra field holding the member size).  Decompression is
// a function: any conforming inflater returns the same bytes, so parity is byte equality with zlib.
//
// Shape on the machine: DEFLATE is one serial bit stream per member — the parallelism is across members (a BAM file is
// tens of thousands of BGZF blocks, a batch of containers thousands of gzip series).  A lane per stream: 2 to 16 streams share
// one wave (few, so that the streams in flight spread over every SIMD and diverge little), each with its own canonical Huffman tables in LDS (counts per code length + symbols in code
// order — 0.7 KB per lane, interleaved by lane so that a wave's table reads spread over the banks — and in front of them a
// direct table over the next 10 / 7 bits (symbol and code length in one 16-bit entry: one LDS read per symbol instead of
// one per code bit; longer codes fall back to the bit-by-bit walk), a 64-bit bit buffer refilled with unaligned 4-byte
// loads, matches copied 4 bytes at a time when the distance allows.
// Then CRC-32 of what was written, as GZIPInputStream#readTrailer checks it.
#include "internal.h"

namespace {

struct __attribute__((packed)) W32u { uint32_t v; };
struct __attribute__((packed)) W16u { uint16_t v; };
struct __attribute__((packed)) W128u { uint32_t a, b, c, d; };

constexpr int kMaxLanes = 16;
constexpr int kLitSyms = 288, kDistSyms = 32, kLenCodes = 19;
// per-lane LDS (u16 entries, interleaved by lane): counts lit [16], counts dist [16], offsets [16], symbols lit [288], dist [32]
constexpr int kCntL = 0, kCntD = 16, kOffs = 32, kSymL = 48, kSymD = kSymL + kLitSyms, kTabU16 = kSymD + kDistSyms;
constexpr int kLensBytes = (kLitSyms + kDistSyms) / 2;  // code lengths, a nibble each
// direct tables: entry = symbol << 4 | code length (0 = a longer code or none: the canonical walk decides)
constexpr int kFastLBits = 10, kFastDBits = 7;
constexpr int kFastL = kTabU16, kFastD = kFastL + (1 << kFastLBits), kAllU16 = kFastD + (1 << kFastDBits);

template <int LANES>
struct InflateLds {
    uint16_t tab[kAllU16][LANES];
    uint8_t lens[kLensBytes][LANES];
    uint32_t crc[256];
};
static_assert(sizeof(InflateLds<kMaxLanes>) <= 64 * 1024, "static LDS of a workgroup");

struct Bits {
    const uint8_t* p;
    int32_t n, pos;
    unsigned long long buf;
    int cnt;
    bool err;
};
__device__ __forceinline__ void refill(Bits& b) {
    if (b.cnt <= 32) {
        if (b.pos + 4 <= b.n) {
            b.buf |= (unsigned long long)reinterpret_cast<const W32u*>(b.p + b.pos)->v << b.cnt;
            b.pos += 4;
            b.cnt += 32;
        } else {
            while (b.cnt <= 56 && b.pos < b.n) {
                b.buf |= (unsigned long long)b.p[b.pos++] << b.cnt;
                b.cnt += 8;
            }
        }
    }
}
__device__ __forceinline__ uint32_t getbits(Bits& b, int k) {  // k <= 16; sets b.err when the stream has fewer bits
    refill(b);
    const bool ok = b.cnt >= k;
    const uint32_t v = (uint32_t)b.buf & ((1u << k) - 1u);
    b.err |= !ok;
    const int take = ok ? k : 0;
    b.buf >>= take;
    b.cnt -= take;
    return ok ? v : 0u;
}

template <int LANES>
struct TablesT {
    InflateLds<LANES>* L;
    int lane;
    __device__ __forceinline__ uint16_t& t(int i) { return L->tab[i][lane]; }
    __device__ __forceinline__ int len_get(int i) { return (L->lens[i >> 1][lane] >> ((i & 1) * 4)) & 15; }
    __device__ __forceinline__ void len_set(int i, int v) {
        uint8_t& c = L->lens[i >> 1][lane];
        c = (uint8_t)((c & ~(15 << ((i & 1) * 4))) | (v << ((i & 1) * 4)));
    }
};

// Control flow in this file is kept to loops with ONE exit and sticky error codes instead of early returns: 4 to 16
// diverging streams share a wave, and every extra way out of a loop nest costs the wave execution-mask bookkeeping on
// every turn (the first version, written with returns, spent most of its time there).

// canonical code of the lengths lens[first .. first + n): counts per length at cnt0, symbols in code order at sym0, and
// the direct table over the next `bits` bits at fast0 (canonical code values in symbol order, RFC 1951 §3.2.2, bit-reversed
// because the stream delivers a code's first bit lowest; bits == 0: none).
// Returns false for an over-subscribed set, or an incomplete one with more than a single one-bit code (zlib's rule:
// inftrees.c accepts an incomplete literal / distance code only when its longest code has one bit).
template <class Tables>
__device__ bool construct(Tables& T, int first, int n, int cnt0, int sym0, bool strict, int fast0, int bits) {
    for (int l = 0; l < 16; l++) T.t(cnt0 + l) = 0;
    for (int s = 0; s < n; s++) T.t(cnt0 + T.len_get(first + s))++;
    int left = 1, max_len = 0;
    bool over = false;
    for (int l = 1; l < 16; l++) {
        const int c = (int)T.t(cnt0 + l);
        left = (left << 1) - c;
        over |= left < 0;
        left = left < 0 ? 0 : left;
        max_len = c ? l : max_len;
    }
    const bool good = !over && !(left > 0 && (strict ? max_len != 0 : max_len > 1));  // the code-length code must be complete
    // offsets of the symbols per length, then (for the direct table) the next code value per length, in the same 16 cells
    T.t(kOffs + 1) = 0;
    for (int l = 1; l < 15; l++) T.t(kOffs + l + 1) = (uint16_t)(T.t(kOffs + l) + T.t(cnt0 + l));
    for (int s = 0; s < n; s++) {
        const int l = T.len_get(first + s);
        if (l) {
            const int o = T.t(kOffs + l);
            T.t(sym0 + o) = (uint16_t)s;
            T.t(kOffs + l) = (uint16_t)(o + 1);
        }
    }
    if (bits > 0 && good) {
        for (int i = 0; i < (1 << bits); i++) T.t(fast0 + i) = 0;
        int code = 0;
        for (int l = 1; l < 16; l++) {  // next_code[l]
            code = (code + (l > 1 ? (int)T.t(cnt0 + l - 1) : 0)) << 1;
            T.t(kOffs + l) = (uint16_t)code;
        }
        for (int s = 0; s < n; s++) {
            const int l = T.len_get(first + s);
            if (l != 0) {
                const uint32_t c = T.t(kOffs + l);
                T.t(kOffs + l) = (uint16_t)(c + 1);
                if (l <= bits) {
                    const uint16_t e = (uint16_t)((s << 4) | l);
                    for (uint32_t i = __brev(c) >> (32 - l); i < (1u << bits); i += 1u << l) T.t(fast0 + (int)i) = e;
                }
            }
        }
    }
    return good;
}

// bit-by-bit walk of the canonical code; -1 when the bits run out or no code matches
template <class Tables>
__device__ __forceinline__ int decode_walk(Bits& b, Tables& T, int cnt0, int sym0) {
    refill(b);
    int code = 0, first = 0, index = 0, sym = -1;
    bool stop = false;
#pragma unroll 1
    for (int l = 1; l < 16 && !stop; l++) {
        if (b.cnt == 0) {
            b.err = true;
            stop = true;
        } else {
            code |= (int)(b.buf & 1ull);
            b.buf >>= 1;
            b.cnt--;
            const int count = (int)T.t(cnt0 + l);
            if (code - count < first) {
                sym = (int)T.t(sym0 + index + (code - first));
                stop = true;
            } else {
                index += count;
                first = (first + count) << 1;
                code <<= 1;
            }
        }
    }
    return sym;
}

template <class Tables>
__device__ __forceinline__ int decode_fast(Bits& b, Tables& T, int cnt0, int sym0, int fast0, int bits) {
    refill(b);
    const uint32_t e = T.t(fast0 + (int)((uint32_t)b.buf & ((1u << bits) - 1u)));
    const int l = (int)(e & 15u);
    int sym;
    if (l != 0 && l <= b.cnt) {
        b.buf >>= l;
        b.cnt -= l;
        sym = (int)(e >> 4);
    } else {
        sym = decode_walk(b, T, cnt0, sym0);  // a longer code, an unused pattern, or the last bits of the stream
    }
    return sym;
}

// One DEFLATE stream into out[o ..).  Returns 0 or a negative code; the member start m0 bounds the match distances.
template <class Tables>
__device__ int inflate_stream(Bits& b, Tables& T, uint8_t* out, int32_t cap, int32_t& o_io, int32_t m0) {
    static constexpr uint8_t kOrder[kLenCodes] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int rc = 0;
    int32_t o = o_io;
    bool last = false;
    while (rc == 0 && !last) {
        last = getbits(b, 1) != 0;
        const uint32_t type = getbits(b, 2);
        if (b.err || type == 3) {
            rc = CRAMGPU_E_CORRUPT;
        } else if (type == 0) {  // stored
            b.buf >>= (b.cnt & 7);
            b.cnt -= (b.cnt & 7);
            const uint32_t len = getbits(b, 16), nlen = getbits(b, 16);
            if (b.err || (len ^ 0xFFFFu) != nlen) {
                rc = CRAMGPU_E_CORRUPT;
            } else if ((int64_t)o + len > cap) {
                rc = CRAMGPU_E_OUT_TOO_SMALL;
            } else {
                for (uint32_t i = 0; i < len && !b.err; i++) out[o + (int32_t)i] = (uint8_t)getbits(b, 8);
                o += (int32_t)len;
                if (b.err) rc = CRAMGPU_E_CORRUPT;
            }
        } else {
            int nlen = 288, ndist = 32;
            if (type == 1) {  // fixed codes (RFC 1951 §3.2.6): complete sets; symbols past 285 / 29 are refused when they come up
                for (int s = 0; s < 288; s++) T.len_set(s, s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
                for (int s = 0; s < 32; s++) T.len_set(kLitSyms + s, 5);
            } else {  // dynamic codes (§3.2.7)
                nlen = (int)getbits(b, 5) + 257;
                ndist = (int)getbits(b, 5) + 1;
                const int ncode = (int)getbits(b, 4) + 4;
                for (int i = 0; i < kLenCodes; i++) T.len_set(i, 0);
                for (int i = 0; i < ncode; i++) T.len_set(kOrder[i], (int)getbits(b, 3));
                // the code-length code sits in the distance areas while the lengths are read (both are rebuilt below)
                if (b.err || nlen > 286 || ndist > 30 || !construct(T, 0, kLenCodes, kCntD, kSymD, true, 0, 0)) rc = CRAMGPU_E_CORRUPT;
                // the 19 lengths of the code-length code are dead once it is built: the symbol lengths overwrite them
                int idx = 0, prev = 0;
                const int total = nlen + ndist;
                while (rc == 0 && idx < total) {
                    const int sym = decode_walk(b, T, kCntD, kSymD);
                    int rep = 1, val = sym;
                    if (sym == 16) {
                        val = prev;
                        rep = 3 + (int)getbits(b, 2);
                        if (idx == 0) rc = CRAMGPU_E_CORRUPT;
                    } else if (sym == 17) {
                        val = 0;
                        rep = 3 + (int)getbits(b, 3);
                    } else if (sym == 18) {
                        val = 0;
                        rep = 11 + (int)getbits(b, 7);
                    }
                    if (sym < 0 || b.err || idx + rep > total) rc = CRAMGPU_E_CORRUPT;
                    if (rc == 0) {
                        for (int q = 0; q < rep; q++, idx++) T.len_set(idx < nlen ? idx : kLitSyms + (idx - nlen), val);
                        prev = val;
                    }
                }
                if (rc == 0 && T.len_get(256) == 0) rc = CRAMGPU_E_CORRUPT;  // no end-of-block code
            }
            if (rc == 0 && !construct(T, 0, nlen, kCntL, kSymL, false, kFastL, kFastLBits)) rc = CRAMGPU_E_CORRUPT;
            if (rc == 0 && !construct(T, kLitSyms, ndist, kCntD, kSymD, false, kFastD, kFastDBits)) rc = CRAMGPU_E_CORRUPT;
            bool eob = false;
            while (rc == 0 && !eob) {
                const int sym = decode_fast(b, T, kCntL, kSymL, kFastL, kFastLBits);
                if (sym < 0 || sym > 285) {
                    rc = CRAMGPU_E_CORRUPT;
                } else if (sym < 256) {
                    if (o >= cap) {
                        rc = CRAMGPU_E_OUT_TOO_SMALL;
                    } else {
                        out[o] = (uint8_t)sym;
                        o++;
                    }
                } else if (sym == 256) {
                    eob = true;
                } else {
                    int len = sym - 254;  // 257..264
                    if (sym == 285) {
                        len = 258;
                    } else if (sym >= 265) {
                        const int e = (sym - 261) >> 2;
                        len = ((4 + ((sym - 261) & 3)) << e) + 3 + (int)getbits(b, e);
                    }
                    const int ds = decode_fast(b, T, kCntD, kSymD, kFastD, kFastDBits);
                    int32_t dist = ds + 1;  // 0..3
                    if (ds >= 4 && ds <= 29) {
                        const int e = (ds >> 1) - 1;  // <= 13
                        dist = ((2 + (ds & 1)) << e) + 1 + (int32_t)getbits(b, e);
                    }
                    if (ds < 0 || ds > 29 || b.err || dist > o - m0) {
                        rc = CRAMGPU_E_CORRUPT;
                    } else if ((int64_t)o + len > cap) {
                        rc = CRAMGPU_E_OUT_TOO_SMALL;
                    } else {
                        uint8_t* d = out + o;
                        const uint8_t* s = d - dist;
                        int i = 0;
                        if (dist >= 16) {
                            // A stream waits for memory more than for anything else, and a copy in 4-byte and 1-byte turns
                            // is a round trip per turn (the next load may depend on the last store: 4 + 3 of them for 19
                            // bytes).  At distance >= 16 the source of a 16-byte piece lies wholly behind its destination:
                            // one load, then the stores.  The last piece reads up to 15 bytes past the match's source —
                            // bytes this stream wrote before d + i.
                            for (; i + 16 <= len; i += 16) *reinterpret_cast<W128u*>(d + i) = *reinterpret_cast<const W128u*>(s + i);
                            const int r = len - i;
                            if (r > 0) {
                                const W128u v = *reinterpret_cast<const W128u*>(s + i);
                                uint8_t* q = d + i;
                                uint32_t w = v.a;
                                if (r >= 4) {
                                    reinterpret_cast<W32u*>(q)->v = v.a;
                                    w = v.b;
                                }
                                if (r >= 8) {
                                    reinterpret_cast<W32u*>(q + 4)->v = v.b;
                                    w = v.c;
                                }
                                if (r >= 12) {
                                    reinterpret_cast<W32u*>(q + 8)->v = v.c;
                                    w = v.d;
                                }
                                q += r & ~3;
                                if (r & 2) {
                                    reinterpret_cast<W16u*>(q)->v = (uint16_t)w;
                                    w >>= 16;
                                    q += 2;
                                }
                                if (r & 1) *q = (uint8_t)w;
                            }
                        } else if (dist >= 4) {
                            for (; i + 4 <= len; i += 4) reinterpret_cast<W32u*>(d + i)->v = reinterpret_cast<const W32u*>(s + i)->v;
                            for (; i < len; i++) d[i] = s[i];
                        } else {
                            // a run of period 1..3 (constant series deflate to these): the period is read once and the run
                            // is stored from registers — a byte-by-byte copy would read back every byte it has just stored
                            uint32_t pat = s[0];
                            if (dist > 1) pat |= (uint32_t)s[1] << 8;
                            if (dist > 2) pat |= (uint32_t)s[2] << 16;
                            if (dist == 3) {
                                for (; i < len; i++) d[i] = (uint8_t)(pat >> (8 * (i % 3)));
                            } else {
                                const uint32_t w = dist == 1 ? pat * 0x01010101u : pat * 0x00010001u;
                                for (; i + 4 <= len; i += 4) reinterpret_cast<W32u*>(d + i)->v = w;
                                for (; i < len; i++) d[i] = (uint8_t)(w >> (8 * (i & 3)));
                            }
                        }
                        o += len;
                    }
                }
            }
        }
    }
    o_io = o;
    return rc;
}

// LANES streams per workgroup (one wave, LANES of its lanes at work): the streams in flight are spread over many waves so
// that every SIMD has some, and few streams diverge inside a wave
template <int LANES>
__global__ __launch_bounds__(LANES) void gunzip_kernel(const uint8_t* in, uint8_t* out, const cramgpu_block* desc, int32_t n,
                                                       int32_t* out_len) {
    __shared__ InflateLds<LANES> lds;
    for (int i = threadIdx.x; i < 256; i += LANES) {  // CRC-32 (IEEE 802.3, reflected 0xEDB88320) byte table
        uint32_t c = (uint32_t)i;
        for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        lds.crc[i] = c;
    }
    __syncthreads();
    const int32_t k = (int32_t)(blockIdx.x * LANES + threadIdx.x);
    if (k < n) {
        const cramgpu_block D = desc[k];
        TablesT<LANES> T{&lds, (int)threadIdx.x};
        Bits b{in + D.in_off, D.in_len, 0, 0ull, 0, false};
        uint8_t* o_base = out + D.out_off;
        const int32_t cap = D.out_cap;
        int32_t o = 0;
        int rc = 0;  // 0 go on, 1 done, < 0 failed
        bool any = false;
        while (rc == 0) {
            // member header (RFC 1952 §2.3); after the first member anything that is not a header ends the stream quietly,
            // as GZIPInputStream#readTrailer does
            const int32_t h = b.pos;
            const uint8_t* p = b.p + h;
            const int32_t left = b.n - h;
            const int stop = any ? 1 : CRAMGPU_E_CORRUPT;
            int32_t q = 10;
            if (left < 18 || p[0] != 0x1F || p[1] != 0x8B || p[2] != 8) {
                rc = stop;
            } else {
                const int flg = p[3];
                if (flg & 4) q += 2 + ((int32_t)p[10] | (int32_t)p[11] << 8);
                for (int f = 8; f <= 16; f <<= 1)
                    if (flg & f) {
                        while (q < left && p[q] != 0) q++;
                        q++;
                    }
                if ((flg & 2) && q + 2 <= left) {  // FHCRC: the low 16 bits of the CRC-32 of the header so far (GZIPInputStream#readHeader checks it)
                    uint32_t hc = 0xFFFFFFFFu;
                    for (int32_t i = 0; i < q; i++) hc = lds.crc[(hc ^ p[i]) & 255u] ^ (hc >> 8);
                    if (((hc ^ 0xFFFFFFFFu) & 0xFFFFu) != ((uint32_t)p[q] | (uint32_t)p[q + 1] << 8)) rc = stop;
                }
                if (flg & 2) q += 2;
                if (rc == 0 && q + 8 > left) rc = stop;
            }
            if (rc == 0) {
                b.pos = h + q;
                b.buf = 0ull, b.cnt = 0, b.err = false;
                const int32_t m0 = o;
                rc = inflate_stream(b, T, o_base, cap, o, m0);
                if (rc == 0) {
                    // trailer: the bytes still in the bit buffer go back
                    b.pos -= b.cnt >> 3;
                    b.buf = 0ull, b.cnt = 0;
                    if (b.pos + 8 > b.n) {
                        rc = CRAMGPU_E_CORRUPT;
                    } else {
                        const uint8_t* t = b.p + b.pos;
                        const uint32_t want_crc = (uint32_t)t[0] | (uint32_t)t[1] << 8 | (uint32_t)t[2] << 16 | (uint32_t)t[3] << 24;
                        const uint32_t want_len = (uint32_t)t[4] | (uint32_t)t[5] << 8 | (uint32_t)t[6] << 16 | (uint32_t)t[7] << 24;
                        b.pos += 8;
                        uint32_t c = 0xFFFFFFFFu;
                        int32_t i = m0;
                        for (; i + 4 <= o; i += 4) {
                            uint32_t w = reinterpret_cast<const W32u*>(o_base + i)->v;
#pragma unroll
                            for (int z = 0; z < 4; z++, w >>= 8) c = lds.crc[(c ^ w) & 255u] ^ (c >> 8);
                        }
                        for (; i < o; i++) c = lds.crc[(c ^ o_base[i]) & 255u] ^ (c >> 8);
                        if ((c ^ 0xFFFFFFFFu) != want_crc || (uint32_t)(o - m0) != want_len) rc = CRAMGPU_E_CORRUPT;
                        any = true;
                    }
                }
            }
        }
        out_len[k] = rc < 0 ? rc : o;
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// A wave per stream, the bit stream on the scalar unit (round 3).  The lane-per-stream kernel above is bound by vector
// issue: ~115 vector instructions per symbol for two useful lanes of 64 (SQ counters: 4.4e10 VALU instructions for the
// 41 828 BGZF members of the bench's stripe, 74 % of the SIMD cycles of its 110 ms).  Here a member has the whole wave:
//   * the bit position, the 16-byte window over the input (scalar loads: the input was written before the launch), the
//     table entry (an LDS read at a uniform address, brought over with v_readfirstlane), code lengths, extra bits and
//     every branch are scalar — a peek delivers >= 57 bits, enough for a length code, its extra bits, a distance code
//     and its extra bits, so a symbol is one window read however long it is;
//   * literals collect in a scalar register, eight to a store (lane j stores byte j);
//   * a match is copied by the wave, byte i by lane i mod 64 (source byte i mod distance: overlapping runs included),
//     behind a workgroup fence that makes the bytes other lanes stored visible;
//   * stored blocks are copied by the wave, CRC-32 is computed by the wave (a segment per lane, combined with
//     x^(8 n) mod P).
// Table construction is the code above, run redundantly by all lanes on one LDS copy (a few thousand vector instructions per
// DEFLATE block, against hundreds of thousands for its symbols before).
#ifdef HIP_SIM
#define G_UNI(x) (x)
#else
#define G_UNI(x) __builtin_amdgcn_readfirstlane((int)(x))
#endif

// The bit reader of a wave: everything in scalar registers.  `buf` holds the next `cnt` bits (first one lowest) and is
// topped up 32 bits at a time from `q`, eight input bytes fetched ahead by a scalar load from a 4-byte boundary (the load
// is issued when q runs empty and not needed before the refill after next).  Behind the stream's end zeros come in —
// never a byte beyond the dword that holds the stream's last byte — and `rem`, the stream's bits not yet consumed, turns
// negative: that is the "ran out of input" of the lane kernel's reader.
struct SBits {
    unsigned long long buf, q;
    unsigned long long nxt, end_al;  // address of the next 8 bytes to fetch (multiple of 4); end of the last dword
    long long rem;
    int32_t cnt, qn;
    __device__ __forceinline__ void fetch() {
        q = 0;
#ifdef HIP_SIM
        uint8_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int j = 0; j < 8; j++)
            if (nxt + j >= sim_lo && nxt + j < sim_hi) t[j] = *(const uint8_t*)(nxt + j);
        memcpy(&q, t, 8);
#else
        typedef const __attribute__((address_space(4))) uint32_t* CPtr;
        CPtr p4 = (CPtr)nxt;
        if (nxt + 8ull <= end_al) {
            const uint32_t d0 = p4[0], d1 = p4[1];
            q = ((unsigned long long)d1 << 32) | d0;
        } else if (nxt + 4ull <= end_al) {
            q = p4[0];
        }
#endif
        nxt += 8ull;
        qn = 2;
    }
    __device__ __forceinline__ void refill() {  // afterwards cnt >= 33
        if (cnt <= 32) {
            buf |= (q & 0xFFFFFFFFull) << cnt;
            cnt += 32;
            q >>= 32;
            if (--qn == 0) fetch();
        }
    }
    // the reader at byte `at` of the stream p[0, n)
    __device__ __forceinline__ void open(const uint8_t* p, int32_t n, int32_t at) {
        const unsigned long long a = (unsigned long long)p + (unsigned long long)at;
        end_al = ((unsigned long long)p + (unsigned long long)n + 3ull) & ~3ull;
#ifdef HIP_SIM
        sim_lo = (unsigned long long)p;
        sim_hi = sim_lo + (unsigned long long)n;
#endif
        nxt = a & ~3ull;
        rem = 8ll * (n - at);
        buf = 0;
        cnt = 0;
        fetch();
        refill();
        const int skip = 8 * (int)(a & 3ull);  // the bytes in front of the start in its dword
        buf >>= skip;
        cnt -= skip;
        refill();
    }
    __device__ __forceinline__ void drop(int k) {  // k <= cnt
        buf >>= k;
        cnt -= k;
        rem -= k;
    }
#ifdef HIP_SIM
    unsigned long long sim_lo, sim_hi;
#endif
};

// The shared tables are written by every lane with the same values (lock-step lanes); on the thread-per-lane host build
// (tests/cpu_sim) that would be a race, so there one thread writes them — with a barrier behind every such section.
#ifdef HIP_SIM
#define G_SOLO(lane) ((lane) == 0)
#else
#define G_SOLO(lane) true
#endif

struct WaveTab {
    InflateLds<1>* L;
    __device__ __forceinline__ uint32_t u(int i) const { return (uint32_t)G_UNI(L->tab[i][0]); }
};

// bit-by-bit walk of the canonical code over peeked bits; -1 when the bits run out or no code matches
__device__ __forceinline__ int wave_walk(unsigned long long w, const WaveTab& W, int cnt0, int sym0, int& used) {
    int code = 0, first = 0, index = 0, sym = -1;
    used = 0;
    bool stop = false;
#pragma unroll 1
    for (int l = 1; l < 16 && !stop; l++) {
        {
            code |= (int)((w >> (l - 1)) & 1ull);
            const int count = (int)W.u(cnt0 + l);
            if (code - count < first) {
                sym = (int)W.u(sym0 + index + (code - first));
                used = l;
                stop = true;
            } else {
                index += count;
                first = (first + count) << 1;
                code <<= 1;
            }
        }
    }
    return sym;
}

__device__ __forceinline__ uint32_t g_mulmodp(uint32_t a, uint32_t b) {  // a * b mod P, x^0 = bit 31
    uint32_t p = 0;
    for (int i = 0; i < 32; i++) {
        p ^= (a & (0x80000000u >> i)) ? b : 0u;
        b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
    }
    return p;
}
// CRC-32 of p[0, n) by the whole wave
__device__ uint32_t g_wave_crc(const uint8_t* p, int32_t n, const uint32_t* T, int lane) {
    if (n <= 0) return 0u;
    const int32_t seg = (n + 63) / 64, pad = 64 * seg - n;
    const int32_t lo = lane * seg - pad, hi = lo + seg;
    uint32_t c = 0;
    if (hi > 0) {
        if (lo <= 0) c = 0xFFFFFFFFu;
        int32_t i = lo > 0 ? lo : 0;
        for (; i + 4 <= hi; i += 4) {
            const uint32_t w = reinterpret_cast<const W32u*>(p + i)->v;
            c = T[(c ^ w) & 255u] ^ (c >> 8);
            c = T[(c ^ (w >> 8)) & 255u] ^ (c >> 8);
            c = T[(c ^ (w >> 16)) & 255u] ^ (c >> 8);
            c = T[(c ^ (w >> 24)) & 255u] ^ (c >> 8);
        }
        for (; i < hi; i++) c = T[(c ^ p[i]) & 255u] ^ (c >> 8);
    }
    uint32_t xd = 0x80000000u, base = 0x00800000u;  // x^(8 seg): square and multiply from x^8
    for (int32_t e = seg; e != 0; e >>= 1) {
        if (e & 1) xd = g_mulmodp(xd, base);
        base = g_mulmodp(base, base);
    }
    for (int d = 1; d < 64; d <<= 1) {  // lane l (a multiple of 2 d) <- [l, l + d) followed by [l + d, l + 2 d)
        const uint32_t other = (uint32_t)__shfl((int)c, (lane + d) & 63);
        c = g_mulmodp(xd, c) ^ other;
        xd = g_mulmodp(xd, xd);
    }
    return (uint32_t)__shfl((int)c, 0) ^ 0xFFFFFFFFu;
}

// literals waiting for their store: up to eight, first one lowest
struct Lits {
    unsigned long long v;
    int32_t k;
    __device__ __forceinline__ void flush(uint8_t* out, int32_t& o, int lane) {
        if (k > 0) {
            if (lane < k) out[o + lane] = (uint8_t)(v >> (8 * lane));
            o += k;
            k = 0;
            v = 0;
        }
    }
};

// One DEFLATE stream into out[o ..) by the wave, from byte `at` of src[0, n).  Returns 0 or a negative code; `at` becomes
// the byte behind the stream; the member start m0 bounds the match distances.
__device__ int inflate_stream_wave(const uint8_t* src, int32_t n, int32_t& at, InflateLds<1>* L, uint8_t* out, int32_t cap,
                                   int32_t& o_io, int32_t m0, int lane) {
    static constexpr uint8_t kOrder[kLenCodes] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    TablesT<1> T{L, 0};
    const WaveTab W{L};
    SBits B;
    B.open(src, n, at);
    const long long bits0 = B.rem;
    int rc = 0;
    int32_t o = o_io;
    bool last = false;
    auto getbits = [&](int k) -> uint32_t {  // k <= 16
        B.refill();
        const uint32_t v = (uint32_t)B.buf & ((1u << k) - 1u);
        B.drop(k);
        return v;
    };
    while (rc == 0 && !last) {
        last = getbits(1) != 0;
        const uint32_t type = getbits(2);
        if (B.rem < 0 || type == 3) {
            rc = CRAMGPU_E_CORRUPT;
        } else if (type == 0) {  // stored
            const int pad = (int)((bits0 - B.rem) & 7);
            if (pad) getbits(8 - pad);
            const uint32_t len = getbits(16), nlen = getbits(16);
            if (B.rem < 0 || (len ^ 0xFFFFu) != nlen || B.rem < 8ll * len) {
                rc = CRAMGPU_E_CORRUPT;
            } else if ((int64_t)o + len > cap) {
                rc = CRAMGPU_E_OUT_TOO_SMALL;
            } else {
                const int32_t from = at + (int32_t)((bits0 - B.rem) >> 3);
                for (uint32_t i = (uint32_t)lane; i < len; i += 64u) out[o + (int32_t)i] = src[from + (int32_t)i];
                o += (int32_t)len;
                const long long rem_after = B.rem - 8ll * len;
                B.open(src, n, from + (int32_t)len);  // the reader again, behind the stored bytes
                B.rem = rem_after;                    // (bits0 stays the count at the stream's start)
            }
        } else {
            int nlen = 288, ndist = 32;
            if (type == 1) {  // fixed codes (RFC 1951 §3.2.6)
                if (G_SOLO(lane)) {
                    for (int s = 0; s < 288; s++) T.len_set(s, s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
                    for (int s = 0; s < 32; s++) T.len_set(kLitSyms + s, 5);
                }
            } else {  // dynamic codes (§3.2.7)
                nlen = (int)getbits(5) + 257;
                ndist = (int)getbits(5) + 1;
                const int ncode = (int)getbits(4) + 4;
                if (G_SOLO(lane))
                    for (int i = 0; i < kLenCodes; i++) T.len_set(i, 0);
                for (int i = 0; i < ncode; i++) {
                    const int v = (int)getbits(3);
                    if (G_SOLO(lane)) T.len_set(kOrder[i], v);
                }
                bool good = true;
                if (G_SOLO(lane)) good = construct(T, 0, kLenCodes, kCntD, kSymD, true, 0, 0);
                __syncthreads();
                good = __shfl((int)good, 0) != 0;
                if (B.rem < 0 || nlen > 286 || ndist > 30 || !good) rc = CRAMGPU_E_CORRUPT;
                rc = G_UNI(rc);
                int idx = 0, prev = 0;
                const int total = nlen + ndist;
                while (rc == 0 && idx < total) {
                    B.refill();
                    int used = 0;
                    const int sym = wave_walk(B.buf, W, kCntD, kSymD, used);
                    B.drop(used);
                    int rep = 1, val = sym;
                    if (sym == 16) {
                        val = prev;
                        rep = 3 + (int)getbits(2);
                        if (idx == 0) rc = CRAMGPU_E_CORRUPT;
                    } else if (sym == 17) {
                        val = 0;
                        rep = 3 + (int)getbits(3);
                    } else if (sym == 18) {
                        val = 0;
                        rep = 11 + (int)getbits(7);
                    }
                    if (sym < 0 || B.rem < 0 || idx + rep > total) rc = CRAMGPU_E_CORRUPT;
                    if (rc == 0) {
                        // the lengths overwrite the 19 of the code-length code, which the walk no longer reads (it reads
                        // the counts and symbols construct() made of them)
                        for (int q = 0; q < rep; q++, idx++)
                            if (G_SOLO(lane)) T.len_set(idx < nlen ? idx : kLitSyms + (idx - nlen), val);
                        prev = val;
                    }
                }
                __syncthreads();
                if (rc == 0 && T.len_get(256) == 0) rc = CRAMGPU_E_CORRUPT;  // no end-of-block code
            }
            __syncthreads();
            rc = G_UNI(rc);
            if (rc == 0) {
                bool good = true;
                if (G_SOLO(lane))
                    good = construct(T, 0, nlen, kCntL, kSymL, false, kFastL, kFastLBits) &&
                           construct(T, kLitSyms, ndist, kCntD, kSymD, false, kFastD, kFastDBits);
                __syncthreads();
                if (__shfl((int)good, 0) == 0) rc = CRAMGPU_E_CORRUPT;
            }
            rc = G_UNI(rc);
            bool eob = false;
            Lits lits{0ull, 0};
            while (rc == 0 && !eob) {
                B.refill();  // >= 33 bits: a literal / length code and a length's extra bits
                const uint32_t e = W.u(kFastL + (int)((uint32_t)B.buf & ((1u << kFastLBits) - 1u)));
                int l = (int)(e & 15u), sym = (int)(e >> 4);
                if (l == 0) sym = wave_walk(B.buf, W, kCntL, kSymL, l);
                B.drop(l);
                if (sym < 0 || sym > 285 || B.rem < 0) {
                    rc = CRAMGPU_E_CORRUPT;
                } else if (sym < 256) {
                    if (o + lits.k >= cap) {
                        rc = CRAMGPU_E_OUT_TOO_SMALL;
                    } else {
                        lits.v |= (unsigned long long)sym << (8 * lits.k);
                        lits.k++;
                        if (lits.k == 8) lits.flush(out, o, lane);
                    }
                } else if (sym == 256) {
                    eob = true;
                } else {
                    int len = sym - 254;  // 257..264
                    if (sym == 285) {
                        len = 258;
                    } else if (sym >= 265) {
                        const int eb = (sym - 261) >> 2;
                        len = ((4 + ((sym - 261) & 3)) << eb) + 3 + (int)((uint32_t)B.buf & ((1u << eb) - 1u));
                        B.drop(eb);
                    }
                    B.refill();  // >= 33 bits: a distance code and its extra bits
                    const uint32_t ed = W.u(kFastD + (int)((uint32_t)B.buf & ((1u << kFastDBits) - 1u)));
                    int ld = (int)(ed & 15u), ds = (int)(ed >> 4);
                    if (ld == 0) ds = wave_walk(B.buf, W, kCntD, kSymD, ld);
                    B.drop(ld);
                    int32_t dist = ds + 1;  // 0..3
                    if (ds >= 4 && ds <= 29) {
                        const int eb = (ds >> 1) - 1;  // <= 13
                        dist = ((2 + (ds & 1)) << eb) + 1 + (int32_t)((uint32_t)B.buf & ((1u << eb) - 1u));
                        B.drop(eb);
                    }
                    lits.flush(out, o, lane);
                    if (ds < 0 || ds > 29 || B.rem < 0 || dist > o - m0) {
                        rc = CRAMGPU_E_CORRUPT;
                    } else if ((int64_t)o + len > cap) {
                        rc = CRAMGPU_E_OUT_TOO_SMALL;
                    } else {
                        __syncthreads();  // the bytes other lanes stored are visible to this lane's loads
                        uint8_t* d = out + o;
                        const uint8_t* sp = d - dist;
                        if (dist >= len) {
                            for (int i = lane; i < len; i += 64) d[i] = sp[i];
                        } else {  // an overlapping run repeats its first `dist` bytes
                            for (int i = lane; i < len; i += 64) d[i] = sp[i % dist];
                        }
                        o += len;
                    }
                }
            }
            lits.flush(out, o, lane);
        }
    }
    at += (int32_t)((bits0 - B.rem + 7) >> 3);
    o_io = o;
    return rc;
}

__global__ __launch_bounds__(64) void gunzip_wave_kernel(const uint8_t* in, uint8_t* out, const cramgpu_block* desc, int32_t n,
                                                         int32_t* out_len) {
    __shared__ InflateLds<1> lds;
    const int lane = (int)threadIdx.x;
    for (int i = lane; i < 256; i += 64) {  // CRC-32 (IEEE 802.3, reflected 0xEDB88320) byte table
        uint32_t c = (uint32_t)i;
        for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        lds.crc[i] = c;
    }
    __syncthreads();
    const int32_t k = (int32_t)blockIdx.x;
    if (k >= n) return;
    const cramgpu_block D = desc[k];
    const uint8_t* src = in + D.in_off;
    const int32_t len_in = D.in_len;
    uint8_t* o_base = out + D.out_off;
    const int32_t cap = D.out_cap;
    int32_t pos = 0;  // byte position of the next member header
    int32_t o = 0;
    int rc = 0;  // 0 go on, 1 done, < 0 failed
    bool any = false;
    auto byte_at = [&](int32_t i) -> uint32_t { return (uint32_t)G_UNI(src[i]); };
    while (rc == 0) {
        // member header (RFC 1952 §2.3); after the first member anything that is not a header ends the stream quietly,
        // as GZIPInputStream#readTrailer does
        const int32_t h = pos;
        const int32_t left = len_in - h;
        const int stop = any ? 1 : CRAMGPU_E_CORRUPT;
        int32_t q = 10;
        if (left < 18 || byte_at(h) != 0x1Fu || byte_at(h + 1) != 0x8Bu || byte_at(h + 2) != 8u) {
            rc = stop;
        } else {
            const uint32_t flg = byte_at(h + 3);
            if (flg & 4u) q += 2 + (int32_t)(byte_at(h + 10) | byte_at(h + 11) << 8);
            for (uint32_t f = 8; f <= 16; f <<= 1)
                if (flg & f) {
                    while (q < left && byte_at(h + q) != 0u) q++;
                    q++;
                }
            if ((flg & 2u) && q + 2 <= left) {  // FHCRC: the low 16 bits of the CRC-32 of the header so far
                uint32_t hc = 0xFFFFFFFFu;
                for (int32_t i = 0; i < q; i++) hc = (uint32_t)G_UNI(lds.crc[(hc ^ byte_at(h + i)) & 255u]) ^ (hc >> 8);
                if (((hc ^ 0xFFFFFFFFu) & 0xFFFFu) != (byte_at(h + q) | byte_at(h + q + 1) << 8)) rc = stop;
            }
            if (flg & 2u) q += 2;
            if (rc == 0 && (q < 0 || q + 8 > left)) rc = stop;
        }
        if (rc == 0) {
            const int32_t m0 = o;
            pos = h + q;
            rc = inflate_stream_wave(src, len_in, pos, &lds, o_base, cap, o, m0, lane);
            if (rc == 0) {
                if (pos + 8 > len_in) {
                    rc = CRAMGPU_E_CORRUPT;
                } else {
                    const uint32_t want_crc = byte_at(pos) | byte_at(pos + 1) << 8 | byte_at(pos + 2) << 16 | byte_at(pos + 3) << 24;
                    const uint32_t want_len = byte_at(pos + 4) | byte_at(pos + 5) << 8 | byte_at(pos + 6) << 16 | byte_at(pos + 7) << 24;
                    pos += 8;
                    __syncthreads();  // every lane's stores before any lane's loads
                    const uint32_t c = g_wave_crc(o_base + m0, o - m0, lds.crc, lane);
                    if (c != want_crc || (uint32_t)(o - m0) != want_len) rc = CRAMGPU_E_CORRUPT;
                    any = true;
                }
            }
        }
    }
    if (lane == 0) out_len[k] = rc < 0 ? rc : o;
}

}  // namespace

hipError_t launch_gunzip(const uint8_t* in, uint8_t* out, const cramgpu_block* d_desc, int32_t n, int32_t* out_len, hipStream_t s) {
    if (n == 0) return hipSuccess;
    // CRAMGPU_GUNZIP_LANES (read at every launch: tests switch it): 64 = a wave per stream, 2 / 4 / 8 / 16 = that many streams
    // per wave, a lane each.  Unset, the launch's shape decides.  The lane kernel has the better throughput (both are
    // bound by instruction issue, ~50 per symbol on the scalar unit here against ~115 vector ones for two lanes there:
    // 20 000 BGZF blocks 82 ms by lanes, 91 ms by waves), the wave kernel the shorter chain: a launch of up to two
    // rounds of waves (64 per CU) lasts as long as its longest stream, and a stream alone in its wave runs 1.7 times
    // faster — the 10 000 GZIP series of the bench beside the rANS decoder are done after 15.5 ms instead of 26.8.
    // Lane kernel, rounds 2 and 3: 2 per wave 81 ms, 1 per wave 88 ms (2.4 rounds), 4: 101, 8: 127, 16: 138; the 41 828 BGZF
    // members of the bench's stripe 131 ms at 2, 158 at 4, 190 at 8.
    const char* env = getenv("CRAMGPU_GUNZIP_LANES");
    int forced = env ? atoi(env) : 0;
    if (forced != 2 && forced != 4 && forced != 8 && forced != 16 && forced != 64) {
#ifdef HIP_SIM
        forced = 2;  // the thread-per-lane host build pays a barrier of 64 threads per match in the wave kernel: only on request
#else
        int dev = 0, n_cu = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
        forced = n <= 64 * n_cu ? 64 : 2;
#endif
    }
    if (forced == 64) {
        hipLaunchKernelGGL(gunzip_wave_kernel, dim3((unsigned)n), dim3(64), 0, s, in, out, d_desc, n, out_len);
        return hipGetLastError();
    }
    const int lanes = forced;
#define GUNZIP_LAUNCH(L) \
    hipLaunchKernelGGL(gunzip_kernel<L>, dim3((unsigned)((n + L - 1) / L)), dim3(L), 0, s, in, out, d_desc, n, out_len)
    switch (lanes) {
        case 16: GUNZIP_LAUNCH(16); break;
        case 8: GUNZIP_LAUNCH(8); break;

        case 4: GUNZIP_LAUNCH(4); break;
        default: GUNZIP_LAUNCH(2); break;
    }
#undef GUNZIP_LAUNCH
    return hipGetLastError();
}
