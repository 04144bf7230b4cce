// G — gunzip of gzip members on the device (the decode halves): the external blocks the
// reference stores with method GZIP (htsjdk!io/ExternalCompression#gunzip L41-48 = java.util.zip.GZIPInputStream, called
// from htsjdk!structure/Block#uncompress L228-262 for the 18 gzip series of a CRAM container) and the BGZF blocks of a BAM
// file (htsjdk!util/BlockCompressedInputStream in front of src/main/java/htsjdk/samtools/cram/paralell/Bam_OBA_Supplier.java).
// The formats are public: RFC 1952 (gzip member: header, DEFLATE data, CRC-32, ISIZE), RFC 1951 (DEFLATE), the SAM
// specification §4.1 (BGZF = gzip members of <= 64 KiB with a 'BC' extra field holding the member size).  Decompression is
// a function: any conforming inflater returns the same bytes, so parity is byte equality with zlib.
//
// Shape on the machine: DEFLATE is one serial bit stream per member — the parallelism is across members (a BAM file is
// tens of thousands of BGZF blocks, a batch of containers thousands of gzip series).  A lane per stream: 64 streams share
// one instruction stream, each with its own canonical Huffman tables in LDS (counts per code length + symbols in code
// order, decoded bit by bit — 0.7 KB per lane, interleaved by lane so that a wave's table reads spread over the banks), a
// 64-bit bit buffer refilled with unaligned 4-byte loads, matches copied 4 bytes at a time when the distance allows.
// Then CRC-32 of what was written, as GZIPInputStream#readTrailer checks it.
#include "internal.h"

namespace {

struct __attribute__((packed)) W32u { uint32_t v; };

constexpr int kMaxLanes = 64;
constexpr int kLitSyms = 288, kDistSyms = 32, kLenCodes = 19;
// per-lane LDS (u16 entries, interleaved by lane): counts lit [16], counts dist [16], offsets [16], symbols lit [288], dist [32]
constexpr int kCntL = 0, kCntD = 16, kOffs = 32, kSymL = 48, kSymD = kSymL + kLitSyms, kTabU16 = kSymD + kDistSyms;
constexpr int kLensBytes = (kLitSyms + kDistSyms) / 2;  // code lengths, a nibble each

template <int LANES>
struct InflateLds {
    uint16_t tab[kTabU16][LANES];
    uint8_t lens[kLensBytes][LANES];
    uint32_t crc[256];
};
static_assert(sizeof(InflateLds<kMaxLanes>) <= 64 * 1024, "static LDS of a workgroup");

struct Bits {
    const uint8_t* p;
    int64_t n, pos;
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
__device__ __forceinline__ uint32_t getbits(Bits& b, int k) {  // k <= 16
    refill(b);
    if (b.cnt < k) {
        b.err = true;
        return 0;
    }
    const uint32_t v = (uint32_t)b.buf & ((1u << k) - 1u);
    b.buf >>= k;
    b.cnt -= k;
    return v;
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

// canonical code of the lengths lens[first .. first + n): counts per length at cnt0, symbols in code order at sym0.
// Returns false for an over-subscribed set, or an incomplete one with more than a single one-bit code (zlib's rule:
// inftrees.c accepts an incomplete literal / distance code only when its longest code has one bit).
template <class Tables>
__device__ bool construct(Tables& T, int first, int n, int cnt0, int sym0, bool strict) {
    for (int l = 0; l < 16; l++) T.t(cnt0 + l) = 0;
    for (int s = 0; s < n; s++) T.t(cnt0 + T.len_get(first + s))++;
    int left = 1, max_len = 0;
    for (int l = 1; l < 16; l++) {
        left <<= 1;
        left -= (int)T.t(cnt0 + l);
        if (left < 0) return false;
        if (T.t(cnt0 + l)) max_len = l;
    }
    if (left > 0 && (strict ? max_len != 0 : max_len > 1)) return false;  // the code-length code must be complete
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
    return true;
}

template <class Tables>
__device__ int decode_sym(Bits& b, Tables& T, int cnt0, int sym0) {
    refill(b);
    int code = 0, first = 0, index = 0;
    for (int l = 1; l < 16; l++) {
        if (b.cnt == 0) {
            b.err = true;
            return -1;
        }
        code |= (int)(b.buf & 1ull);
        b.buf >>= 1;
        b.cnt--;
        const int count = (int)T.t(cnt0 + l);
        if (code - count < first) return (int)T.t(sym0 + index + (code - first));
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return -1;  // ran out of codes
}

// One DEFLATE stream into out[o ..).  Returns 0 or a negative code; o / member start m0 bound the match distances.
template <class Tables>
__device__ int inflate_stream(Bits& b, Tables& T, uint8_t* out, int64_t cap, int64_t& o, int64_t m0) {
    static constexpr uint8_t kOrder[kLenCodes] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    for (;;) {
        const uint32_t last = getbits(b, 1), type = getbits(b, 2);
        if (b.err) return CRAMGPU_E_CORRUPT;
        if (type == 0) {  // stored
            b.buf >>= (b.cnt & 7);
            b.cnt -= (b.cnt & 7);
            const uint32_t len = getbits(b, 16), nlen = getbits(b, 16);
            if (b.err || (len ^ 0xFFFFu) != nlen) return CRAMGPU_E_CORRUPT;
            if (o + len > cap) return CRAMGPU_E_OUT_TOO_SMALL;
            for (uint32_t i = 0; i < len; i++) {
                const uint32_t c = getbits(b, 8);
                if (b.err) return CRAMGPU_E_CORRUPT;
                out[o++] = (uint8_t)c;
            }
        } else if (type == 3) {
            return CRAMGPU_E_CORRUPT;
        } else {
            int nlen, ndist;
            if (type == 1) {  // fixed codes (RFC 1951 §3.2.6)
                nlen = 288, ndist = 32;  // complete sets; the symbols past 285 / 29 are refused when they come up
                for (int s = 0; s < 288; s++) T.len_set(s, s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
                for (int s = 0; s < 32; s++) T.len_set(kLitSyms + s, 5);
            } else {  // dynamic codes (§3.2.7)
                nlen = (int)getbits(b, 5) + 257;
                ndist = (int)getbits(b, 5) + 1;
                const int ncode = (int)getbits(b, 4) + 4;
                if (b.err || nlen > 286 || ndist > 30) return CRAMGPU_E_CORRUPT;
                for (int i = 0; i < kLenCodes; i++) T.len_set(i, 0);
                for (int i = 0; i < ncode; i++) {
                    const int v = (int)getbits(b, 3);
                    T.len_set(kOrder[i], v);
                }
                if (b.err) return CRAMGPU_E_CORRUPT;
                // the code-length code sits in the distance areas while the lengths are read (both are rebuilt below)
                if (!construct(T, 0, kLenCodes, kCntD, kSymD, true)) return CRAMGPU_E_CORRUPT;
                // the 19 lengths of the code-length code are dead once it is built: the symbol lengths overwrite them
                int idx = 0;
                int prev = 0;
                while (idx < nlen + ndist) {
                    int sym = decode_sym(b, T, kCntD, kSymD);
                    if (sym < 0 || b.err) return CRAMGPU_E_CORRUPT;
                    if (sym < 16) {
                        T.len_set(idx < nlen ? idx : kLitSyms + (idx - nlen), sym);
                        prev = sym;
                        idx++;
                    } else {
                        int rep, val = 0;
                        if (sym == 16) {
                            if (idx == 0) return CRAMGPU_E_CORRUPT;
                            val = prev;
                            rep = 3 + (int)getbits(b, 2);
                        } else if (sym == 17) {
                            rep = 3 + (int)getbits(b, 3);
                        } else {
                            rep = 11 + (int)getbits(b, 7);
                        }
                        if (b.err || idx + rep > nlen + ndist) return CRAMGPU_E_CORRUPT;
                        for (int q = 0; q < rep; q++, idx++) T.len_set(idx < nlen ? idx : kLitSyms + (idx - nlen), val);
                        prev = val;
                    }
                }
                if (T.len_get(256) == 0) return CRAMGPU_E_CORRUPT;  // no end-of-block code
            }
            if (!construct(T, 0, nlen, kCntL, kSymL, false)) return CRAMGPU_E_CORRUPT;
            if (!construct(T, kLitSyms, ndist, kCntD, kSymD, false)) return CRAMGPU_E_CORRUPT;
            for (;;) {
                int sym = decode_sym(b, T, kCntL, kSymL);
                if (sym < 0 || b.err) return CRAMGPU_E_CORRUPT;
                if (sym < 256) {
                    if (o >= cap) return CRAMGPU_E_OUT_TOO_SMALL;
                    out[o++] = (uint8_t)sym;
                    continue;
                }
                if (sym == 256) break;
                if (sym > 285) return CRAMGPU_E_CORRUPT;
                int len;
                if (sym < 265) {
                    len = sym - 254;
                } else if (sym == 285) {
                    len = 258;
                } else {
                    const int e = (sym - 261) >> 2;
                    len = ((4 + ((sym - 261) & 3)) << e) + 3 + (int)getbits(b, e);
                }
                const int ds = decode_sym(b, T, kCntD, kSymD);
                if (ds < 0 || ds > 29 || b.err) return CRAMGPU_E_CORRUPT;
                int64_t dist;
                if (ds < 4) {
                    dist = ds + 1;
                } else {
                    const int e = (ds >> 1) - 1;
                    uint32_t x = e > 16 ? 0u : getbits(b, e);  // e <= 13
                    dist = ((int64_t)(2 + (ds & 1)) << e) + 1 + x;
                }
                if (b.err || dist > o - m0) return CRAMGPU_E_CORRUPT;
                if (o + len > cap) return CRAMGPU_E_OUT_TOO_SMALL;
                uint8_t* d = out + o;
                const uint8_t* s = d - dist;
                int i = 0;
                if (dist >= 4)
                    for (; i + 4 <= len; i += 4) reinterpret_cast<W32u*>(d + i)->v = reinterpret_cast<const W32u*>(s + i)->v;
                for (; i < len; i++) d[i] = s[i];
                o += len;
            }
        }
        if (last) return 0;
    }
}

// LANES streams per workgroup (one wave, LANES of its lanes at work): few streams are spread over many waves so that every
// SIMD has some — a lane's chain of table reads and loads is latency, and 64 diverging streams in one wave serialise
template <int LANES>
__global__ __launch_bounds__(LANES) void gunzip_kernel(const uint8_t* in, uint8_t* out, const cramgpu_block* desc, int32_t n,
                                                       int32_t* out_len, int exp) {
    __shared__ InflateLds<LANES> lds;
    for (int i = threadIdx.x; i < 256; i += LANES) {  // CRC-32 (IEEE 802.3, reflected 0xEDB88320) byte table
        uint32_t c = (uint32_t)i;
        for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        lds.crc[i] = c;
    }
    __syncthreads();
    const int32_t k = (int32_t)(blockIdx.x * LANES + threadIdx.x);
    if (k >= n) return;
    const cramgpu_block D = desc[k];
    TablesT<LANES> T{&lds, (int)threadIdx.x};
    Bits b{in + D.in_off, D.in_len, 0, 0ull, 0, false};
    uint8_t* o_base = out + D.out_off;
    const int64_t cap = D.out_cap;
    int64_t o = 0;
    int rc = 0;
    bool any = false;
    while (rc == 0) {
        // member header (RFC 1952 §2.3); after the first member anything that is not a header ends the stream quietly,
        // as GZIPInputStream#readTrailer does
        const int64_t h = b.pos;
        const uint8_t* p = b.p + h;
        const int64_t left = b.n - h;
        if (left < 18 || p[0] != 0x1F || p[1] != 0x8B || p[2] != 8) {
            if (!any) rc = CRAMGPU_E_CORRUPT;
            break;
        }
        const int flg = p[3];
        int64_t q = 10;
        if (flg & 4) {
            if (q + 2 > left) { rc = any ? 1 : CRAMGPU_E_CORRUPT; break; }
            q += 2 + ((int64_t)p[q] | (int64_t)p[q + 1] << 8);
        }
        for (int f = 8; f <= 16 && q <= left; f <<= 1)
            if (flg & f) {
                while (q < left && p[q] != 0) q++;
                q++;
            }
        if (flg & 2) q += 2;
        if (q + 8 > left) { rc = any ? 1 : CRAMGPU_E_CORRUPT; break; }
        b.pos = h + q;
        b.buf = 0ull, b.cnt = 0, b.err = false;
        const int64_t m0 = o;
        rc = inflate_stream(b, T, o_base, cap, o, m0);
        if (rc != 0) break;
        // trailer: the bytes still in the bit buffer go back
        b.pos -= b.cnt >> 3;
        b.buf = 0ull, b.cnt = 0;
        if (b.pos + 8 > b.n) { rc = CRAMGPU_E_CORRUPT; break; }
        const uint8_t* t = b.p + b.pos;
        const uint32_t want_crc = (uint32_t)t[0] | (uint32_t)t[1] << 8 | (uint32_t)t[2] << 16 | (uint32_t)t[3] << 24;
        const uint32_t want_len = (uint32_t)t[4] | (uint32_t)t[5] << 8 | (uint32_t)t[6] << 16 | (uint32_t)t[7] << 24;
        b.pos += 8;
        uint32_t c = 0xFFFFFFFFu;
        if (!(exp & 1)) for (int64_t i = m0; i < o; i++) c = lds.crc[(c ^ o_base[i]) & 255u] ^ (c >> 8);
        if ((!(exp & 1) && (c ^ 0xFFFFFFFFu) != want_crc) || (uint32_t)(o - m0) != want_len) { rc = CRAMGPU_E_CORRUPT; break; }
        any = true;
    }
    out_len[k] = rc < 0 ? rc : (int32_t)o;
}

}  // namespace

hipError_t launch_gunzip(const uint8_t* in, uint8_t* out, const cramgpu_block* d_desc, int32_t n, int32_t* out_len, hipStream_t s) {
    if (n == 0) return hipSuccess;
    static const int exp = [] {
        const char* e = getenv("CRAMGPU_GUNZIP_EXP");
        return e ? atoi(e) : 0;
    }();
    static const int forced = [] {
        const char* e = getenv("CRAMGPU_GUNZIP_LANES");
        return e ? atoi(e) : 0;
    }();
    // about four waves per SIMD (4096 waves) before a wave takes more streams
    const int lanes = forced ? forced : n >= 64 * 4096 ? 64 : n >= 32 * 4096 ? 32 : n >= 16 * 4096 ? 16 : n >= 8 * 4096 ? 8 : 4;
#define GUNZIP_LAUNCH(L) \
    hipLaunchKernelGGL(gunzip_kernel<L>, dim3((unsigned)((n + L - 1) / L)), dim3(L), 0, s, in, out, d_desc, n, out_len, exp)
    switch (lanes) {
        case 64: GUNZIP_LAUNCH(64); break;
        case 32: GUNZIP_LAUNCH(32); break;
        case 16: GUNZIP_LAUNCH(16); break;
        case 8: GUNZIP_LAUNCH(8); break;
        default: GUNZIP_LAUNCH(4); break;
    }
#undef GUNZIP_LAUNCH
    return hipGetLastError();
}
