// Z — gzip members written on the device (the encode halves): the external blocks the
// reference stores with method GZIP (htsjdk!io/ExternalCompression#gzip L30-39 = java.util.zip.GZIPOutputStream at the
// default level, called from htsjdk!structure/Block for the 18 gzip series of a container) and the BGZF blocks behind
// src/main/java/htsjdk/samtools/cram/paralell/CramToBam_OBA_Function.java:68-75 (htsjdk!util/BlockCompressedOutputStream).
// The formats are public: RFC 1952 (member: header, DEFLATE data, CRC-32, ISIZE), RFC 1951 (DEFLATE), SAM specification
// §4.1 (BGZF: FEXTRA subfield 'BC' = member size - 1).  DEFLATE leaves the choice of matches and codes to the writer, so
// byte identity with zlib is NOT part of the contract: the contract is that every conforming
// inflater (zlib, GZIPInputStream, kernel G) returns the input, and the size is reported beside zlib's.
//
// Shape on the machine: as in G, one serial bit stream per member and the parallelism across members — a lane per stream,
// 2 to 8 streams per wave.  Per DEFLATE block of at most 32 KiB of input (positions and counts then fit 16 bits): pass 1
// parses greedily with a single-probe hash table over 3-byte prefixes (LDS, 1.5 KiB per lane; emptied per block, so every
// match lies inside the block and inside the window) and counts literal / length / distance symbols; the lane then builds
// the two length-limited Huffman codes (counting sort of the used symbols, Moffat & Katajainen's in-place
// minimum-redundancy lengths, the Kraft repair that zlib-compatible writers use for the 15- / 7-bit limits), run-length
// codes the lengths (symbols 16 / 17 / 18) and writes the dynamic block header; pass 2 repeats the same parse — it is
// deterministic — and emits the codes.  A block whose dynamic form would not beat the stored form is stored.
#include "internal.h"

namespace {

struct __attribute__((packed)) W32u { uint32_t v; };
__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { return reinterpret_cast<const W32u*>(p)->v; }

constexpr int kMaxLanes = 8;
// 512 entries: LDS per stream decides how many streams a CU holds, and throughput is streams in flight / latency — 2048
// entries: 286 ms for 20 000 BGZF blocks, 1024: 225 ms, 512: 165 ms, with members 0.7 % SMALLER (far short matches forgotten)
constexpr int kHashBits = 9, kHash = 1 << kHashBits;
constexpr int kBlock = 32768;
constexpr int kNL = 286, kND = 30, kNC = 19, kAll = kNL + kND;

template <int LANES>
struct DeflateLds {  // per-lane arrays interleaved by lane
    uint16_t hash[kHash + 1][LANES];  // + a spare slot for the positions that have no 3 bytes (and one to compare) behind them
    uint8_t tag[kHash + 1][LANES];    // 8 more bits of the position's hash
    uint16_t freq[kAll][LANES];   // symbol counts of the block, then the (bit-reversed) codes
    uint16_t work[kNL][LANES];    // sorted counts / the in-place tree
    uint16_t syms[kAll][LANES];   // symbols in sorted order, then the run-length tokens (symbol | extra << 5)
    uint8_t len[kAll][LANES];     // code lengths: literal / length symbols, then distance symbols
    uint16_t clfreq[kNC][LANES];  // code-length code: counts, then codes
    uint8_t cllen[kNC][LANES];
    uint16_t num[16][LANES];      // codes per length / next code per length
    uint32_t crc[256];
};
static_assert(sizeof(DeflateLds<kMaxLanes>) <= 80 * 1024, "two workgroups per CU");
static_assert(kHash >= 256, "the match table doubles as the 256 counters of the counting sort");

struct Col16 {
    uint16_t* p;
    int st;
    __device__ __forceinline__ uint16_t& operator[](int i) const { return p[i * st]; }
};
struct Col8 {
    uint8_t* p;
    int st;
    __device__ __forceinline__ uint8_t& operator[](int i) const { return p[i * st]; }
};

struct BitW {
    uint8_t* o;
    int32_t pos, cap;  // pos keeps counting past cap (nothing is stored there): the caller sees the overflow at the end
    unsigned long long acc;
    int cnt;
};
__device__ __forceinline__ void put(BitW& w, uint32_t v, int k, bool store = true) {  // k <= 32, v < 2^k
    w.acc |= (unsigned long long)v << w.cnt;
    w.cnt += k;
    if (w.cnt >= 32) {
        if (store && w.pos + 4 <= w.cap) reinterpret_cast<W32u*>(w.o + w.pos)->v = (uint32_t)w.acc;
        w.pos += 4;
        w.acc >>= 32;
        w.cnt -= 32;
    }
}
__device__ __forceinline__ void flush_bytes(BitW& w, bool store = true) {  // to a byte boundary, everything out
    w.cnt = (w.cnt + 7) & ~7;
    while (w.cnt > 0) {
        if (store && w.pos < w.cap) w.o[w.pos] = (uint8_t)w.acc;
        w.pos++;
        w.acc >>= 8;
        w.cnt -= 8;
    }
}

__device__ void lengths_from_sorted(int m, int max_len, Col8 len, Col16 A, Col16 S, Col16 num, bool assign);

// Code lengths (<= max_len) of the symbols with a non-zero count; returns how many there are.  cnt: 256 cells of scratch
// for the counting sort of large alphabets (none: insertion sort only).
__device__ int build_lengths(Col16 freq, int n, int max_len, Col8 len, Col16 A, Col16 S, Col16 num, Col16 cnt) {
    int m = 0;
    for (int s = 0; s < n; s++) {  // the used symbols, in symbol order, parked in A
        len[s] = 0;
        if (freq[s] != 0) A[m++] = (uint16_t)s;
    }
    // (A, S) = (count, symbol) ascending by count, equal counts in symbol order
    if (cnt.p != nullptr && m > 48) {
        // two stable counting passes over the count's bytes: ~4 m + 512 LDS round trips instead of the ~m * m / 4 of the
        // insertion sort (a block of BAM-like bytes uses ~280 symbols)
        for (int pass = 0; pass < 2; pass++) {
            const Col16 src = pass == 0 ? A : S, dst = pass == 0 ? S : A;
            const int sh = 8 * pass;
            for (int i = 0; i < 256; i++) cnt[i] = 0;
            for (int i = 0; i < m; i++) {
                const int b = (freq[src[i]] >> sh) & 255;
                cnt[b] = (uint16_t)(cnt[b] + 1);
            }
            int run = 0;
            for (int i = 0; i < 256; i++) {
                const int c = cnt[i];
                cnt[i] = (uint16_t)run;
                run += c;
            }
            for (int i = 0; i < m; i++) {
                const uint16_t sym = src[i];
                const int b = (freq[sym] >> sh) & 255;
                const int at = cnt[b];
                cnt[b] = (uint16_t)(at + 1);
                dst[at] = sym;
            }
        }
        for (int i = 0; i < m; i++) {
            const uint16_t sym = A[i];
            S[i] = sym;
            A[i] = freq[sym];
        }
    } else {
        for (int k = 0; k < m; k++) {  // insertion sort in place: slot k is read before anything is shifted into it
            const uint16_t sym = A[k], f = freq[sym];
            int j = k;
            while (j > 0 && A[j - 1] > f) {
                A[j] = A[j - 1];
                S[j] = S[j - 1];
                j--;
            }
            A[j] = f;
            S[j] = sym;
        }
    }
    lengths_from_sorted(m, max_len, len, A, S, num, true);
    return m;
}

// (A, S) = (count, symbol) of the m used symbols, ascending by count: the length-limited code lengths.  assign == false
// leaves the last step — lengths to symbols, from num[] and the sorted order — to the caller (the wave kernel spreads it
// over its lanes).
__device__ void lengths_from_sorted(int m, int max_len, Col8 len, Col16 A, Col16 S, Col16 num, bool assign) {
    if (m == 1) {
        for (int l = 0; l < 16; l++) num[l] = 0;
        num[1] = 1;
        if (assign) len[S[0]] = 1;
    }
    if (m >= 2) {
        // Moffat & Katajainen, "In-place calculation of minimum-redundancy codes" (1995): A holds counts, then parent
        // indices of the internal nodes, then depths
        A[0] = (uint16_t)(A[0] + A[1]);
        int root = 0, leaf = 2;
        for (int next = 1; next < m - 1; next++) {
            if (leaf >= m || A[root] < A[leaf]) {
                A[next] = A[root];
                A[root++] = (uint16_t)next;
            } else {
                A[next] = A[leaf++];
            }
            if (leaf >= m || (root < next && A[root] < A[leaf])) {
                A[next] = (uint16_t)(A[next] + A[root]);
                A[root++] = (uint16_t)next;
            } else {
                A[next] = (uint16_t)(A[next] + A[leaf++]);
            }
        }
        A[m - 2] = 0;
        for (int next = m - 3; next >= 0; next--) A[next] = (uint16_t)(A[A[next]] + 1);
        int avbl = 1, used = 0, dpth = 0;
        root = m - 2;
        int next = m - 1;
        while (avbl > 0) {
            while (root >= 0 && (int)A[root] == dpth) {
                used++;
                root--;
            }
            while (avbl > used) {
                A[next--] = (uint16_t)dpth;
                avbl--;
            }
            avbl = 2 * used;
            dpth++;
            used = 0;
        }
        // the limit: depths beyond it are cut to it, then codes are lengthened one at a time until the Kraft sum is 1 again
        for (int l = 0; l < 16; l++) num[l] = 0;
        for (int i = 0; i < m; i++) {
            const int d = (int)A[i] < max_len ? (int)A[i] : max_len;
            num[d] = (uint16_t)(num[d] + 1);
        }
        uint32_t total = 0;
        for (int l = max_len; l >= 1; l--) total += (uint32_t)num[l] << (max_len - l);
        while (total > (1u << max_len)) {
            num[max_len] = (uint16_t)(num[max_len] - 1);
            bool moved = false;
            for (int l = max_len - 1; l >= 1; l--) {
                if (!moved && num[l] != 0) {
                    num[l] = (uint16_t)(num[l] - 1);
                    num[l + 1] = (uint16_t)(num[l + 1] + 2);
                    moved = true;
                }
            }
            total--;
        }
        if (assign) {
            int k = 0;  // sorted order: rarest first, so the longest codes go first
            for (int l = max_len; l >= 1; l--)
                for (int c = (int)num[l]; c > 0; c--) len[S[k++]] = (uint8_t)l;
        }
    }
}

// canonical codes of the lengths (RFC 1951 §3.2.2), bit-reversed for a writer that sends a code's first bit lowest
__device__ void assign_codes(Col8 len, int n, Col16 code, Col16 num) {
    for (int l = 0; l < 16; l++) num[l] = 0;
    for (int s = 0; s < n; s++) num[len[s]] = (uint16_t)(num[len[s]] + 1);
    int c = 0, prev = 0;
    for (int l = 1; l < 16; l++) {  // num[l] becomes next_code[l]
        c = (c + prev) << 1;
        prev = (int)num[l];
        num[l] = (uint16_t)c;
    }
    for (int s = 0; s < n; s++) {
        const int l = len[s];
        uint32_t v = 0;
        if (l != 0) {
            v = __brev((uint32_t)num[l]) >> (32 - l);
            num[l] = (uint16_t)(num[l] + 1);
        }
        code[s] = (uint16_t)v;
    }
}

struct Lane {
    Col16 hash, freq, work, syms, clfreq, num;
    Col8 len, cllen, tag;
    const uint32_t* crc;  // CRC-32 byte table (shared by the workgroup)
};

// 8 input bytes at offset q of p[0, n): zero beyond the end
__device__ __forceinline__ unsigned long long ld64s(const uint8_t* p, int32_t q, int32_t n) {
    unsigned long long v = 0;
    if (q + 8 <= n) {
        v = (unsigned long long)ld32(p + q) | (unsigned long long)ld32(p + q + 4) << 32;
    } else {
        for (int k = 0; q + k < n; k++) v |= (unsigned long long)p[q + k] << (8 * k);
    }
    return v;
}

// One greedy parse of p[0, n).  EMIT == false: counts symbols, returns the number of extra bits; EMIT == true: writes.
// Every turn of the loop is one input position for every lane, whatever the lane is doing there (inside a match it only
// enters the position into the table, as zlib does): several streams share the wave, and turns of uneven length would make all
// of them wait for the slowest.  The input comes through a 16-byte register window refilled 8 bytes at a time, one
// refill ahead of its use, so a literal costs no global load; a table entry carries 8 more bits of the hash, so the
// candidate's bytes are only fetched when they are almost surely a match.
template <bool EMIT>
__device__ uint32_t parse(const uint8_t* p, int32_t n, const Lane& L, BitW& w, uint32_t& crc, long long* prof) {
    for (int i = 0; i <= kHash; i++) L.hash[i] = 0xFFFF;
    uint32_t extra = 0;
    unsigned long long lo = ld64s(p, 0, n), hi = ld64s(p, 8, n), nx = ld64s(p, 16, n);
    int32_t skip = 0;
    for (int32_t i = 0; i < n; i++) {
        const int o = i & 7;
        if (o == 0 && i != 0) {
            lo = hi;
            hi = nx;
            nx = ld64s(p, i + 16, n);
        }
        const unsigned long long w8 = o == 0 ? lo : (lo >> (8 * o)) | (hi << (64 - 8 * o));  // input bytes i .. i + 7
        const uint32_t w4 = (uint32_t)w8;
        // every LDS read the turn may need is issued here, before the first wait: table entry, tag, and the literal's
        // count (pass 1) or code (pass 2) — one LDS latency per turn instead of three in a row.  The turn of a literal
        // has no branch of its own (a divergent branch costs the wave ~40 cycles of mask bookkeeping, and the first
        // version had a dozen per turn): the last three positions of a block use a spare table slot instead of a test,
        // the count is written back with or without its increment, the code is put with a length of zero bits when
        // there is none.  Only a table hit leaves the straight line.
        const int lit = (int)(w4 & 255u);
        const uint32_t hh = (w4 & 0xFFFFFFu) * 2654435761u;
        const uint8_t tag = (uint8_t)(hh >> (24 - kHashBits));
        const bool can = i + 4 <= n;
        const uint32_t h = can ? hh >> (32 - kHashBits) : (uint32_t)kHash;
        const uint16_t f_lit = L.freq[lit];
        const uint8_t l_lit = EMIT ? L.len[lit] : (uint8_t)0;
        // the member's CRC-32 rides along in pass 1 (every byte passes here exactly once): its table read is issued with
        // the others instead of being a 90-cycle round trip per byte in a loop of its own
        const uint32_t c_tab = EMIT ? 0u : L.crc[(crc ^ (uint32_t)lit) & 255u];
        const int32_t cand = (int32_t)L.hash[h];
        const uint8_t tv = L.tag[h];
        L.hash[h] = (uint16_t)i;
        L.tag[h] = tag;
        const bool active = skip == 0;
        skip -= active ? 0 : 1;
        int32_t mlen = 0;
        if (active && can && cand != 0xFFFF && tv == tag) {
#ifdef CRAMGPU_PROF_BUILD
            const long long tb = clock64();
            prof[5]++;
#endif
            // the candidate's first 8 bytes against the window: most matches in these data are shorter than that, and
            // are then verified and measured behind ONE global load
            const int32_t maxl = n - i < 258 ? n - i : 258;
            bool go = false;
            if (i + 8 <= n) {
                const unsigned long long x = ((unsigned long long)ld32(p + cand) | (unsigned long long)ld32(p + cand + 4) << 32) ^ w8;
                if ((x & 0xFFFFFFull) == 0) {
                    go = x == 0;
                    mlen = go ? 8 : (__ffsll((long long)x) - 1) >> 3;
                }
            } else if (((ld32(p + cand) ^ w4) & 0xFFFFFFu) == 0) {
                mlen = 3;
                go = true;
            }
            if (mlen != 0) {
                while (go && mlen + 4 <= maxl) {
                    const uint32_t x = ld32(p + i + mlen) ^ ld32(p + cand + mlen);
                    if (x != 0) {
                        mlen += (__ffs((int)x) - 1) >> 3;
                        go = false;
                    } else {
                        mlen += 4;
                    }
                }
                while (go && mlen < maxl) {
                    if (p[i + mlen] == p[cand + mlen])
                        mlen++;
                    else
                        go = false;
                }
                // length 3..258 -> symbol 257..285 + extra bits, distance 1..32768 -> symbol 0..29 + extra bits (RFC 1951 §3.2.5)
                const int lm = mlen - 3;
                int ls = 257 + lm, leb = 0;
                if (lm >= 8) {
                    const int k = 31 - __clz(lm);
                    leb = k - 2;
                    ls = 261 + 4 * leb + ((lm >> leb) & 3);
                }
                if (mlen == 258) ls = 285, leb = 0;
                const int dm = i - cand - 1;
                int ds = dm, deb = 0;
                if (dm >= 4) {
                    const int k = 31 - __clz(dm);
                    deb = k - 1;
                    ds = 2 * k + ((dm >> deb) & 1);
                }
                if (EMIT) {
                    put(w, (uint32_t)L.freq[ls] | (uint32_t)(lm & ((1 << leb) - 1)) << L.len[ls], L.len[ls] + leb);
                    put(w, (uint32_t)L.freq[kNL + ds] | (uint32_t)(dm & ((1 << deb) - 1)) << L.len[kNL + ds], L.len[kNL + ds] + deb);
                } else {
                    L.freq[ls] = (uint16_t)(L.freq[ls] + 1);
                    L.freq[kNL + ds] = (uint16_t)(L.freq[kNL + ds] + 1);
                    extra += (uint32_t)(leb + deb);
                }
                skip = mlen - 1;
#ifdef CRAMGPU_PROF_BUILD
                prof[6]++;
#endif
            }
#ifdef CRAMGPU_PROF_BUILD
            prof[7] += clock64() - tb;
#endif
        }
        const bool literal = active && mlen == 0;
        if (EMIT)
            put(w, literal ? (uint32_t)f_lit : 0u, literal ? (int)l_lit : 0);
        else
            L.freq[lit] = (uint16_t)(f_lit + (literal ? 1 : 0));
        if (!EMIT) crc = c_tab ^ (crc >> 8);
    }
    return extra;
}

// order in which the header carries the code-length code's own lengths (RFC 1951 §3.2.7):
// 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15 — the last sixteen as nibbles of one constant
__device__ __forceinline__ int cl_order(int i) { return i < 3 ? 16 + i : (int)((0xF1E2D3C4B5A69780ull >> (4 * (i - 3))) & 15u); }

// code lengths of the block as the sequence the header carries: literal / length symbols [0, hlit), then distances
__device__ __forceinline__ int seq_len(const Lane& L, int hlit, int i) { return i < hlit ? L.len[i] : L.len[kNL + i - hlit]; }

#ifdef CRAMGPU_PROF_BUILD
#define GZ_STAMP(k) prof[k] += clock64() - t0, t0 = clock64()
#else
#define GZ_STAMP(k)
#endif
__device__ void deflate_block(const uint8_t* p, int32_t n, bool last, const Lane& L, BitW& w, uint32_t& crc, long long* prof) {
#ifdef CRAMGPU_PROF_BUILD
    long long t0 = clock64();
#endif
    for (int s = 0; s < kAll; s++) L.freq[s] = 0;
    L.freq[256] = 1;  // end of block
    const uint32_t extra = parse<false>(p, n, L, w, crc, prof);
    GZ_STAMP(0);
    build_lengths(L.freq, kNL, 15, L.len, L.work, L.syms, L.num, L.hash);
    const Col16 dfreq{&L.freq[kNL], L.freq.st};
    const Col8 dlen{&L.len[kNL], L.len.st};
    build_lengths(dfreq, kND, 15, dlen, L.work, L.syms, L.num, Col16{nullptr, 0});
    uint32_t body = extra;
    for (int s = 0; s < kAll; s++) body += (uint32_t)L.freq[s] * L.len[s];
    int hlit = kNL, hdist = kND;
    while (hlit > 257 && L.len[hlit - 1] == 0) hlit--;
    while (hdist > 1 && dlen[hdist - 1] == 0) hdist--;
    // run-length tokens of the hlit + hdist lengths (RFC 1951 §3.2.7: one sequence across both alphabets)
    const int total = hlit + hdist;
    for (int s = 0; s < kNC; s++) L.clfreq[s] = 0;
    int nt = 0;
    uint32_t hdr = 0;
    for (int i = 0; i < total;) {
        const int v = seq_len(L, hlit, i);
        int run = 1;
        while (i + run < total && seq_len(L, hlit, i + run) == v) run++;
        i += run;
        if (v != 0) {
            L.syms[nt++] = (uint16_t)v;
            L.clfreq[v] = (uint16_t)(L.clfreq[v] + 1);
            run--;
        }
        while (run > 0) {
            int sym, ext = 0, take = 1;
            if (v == 0 && run >= 11) {
                take = run < 138 ? run : 138;
                sym = 18, ext = take - 11, hdr += 7;
            } else if (v == 0 && run >= 3) {
                take = run;
                sym = 17, ext = take - 3, hdr += 3;
            } else if (v != 0 && run >= 3) {
                take = run < 6 ? run : 6;
                sym = 16, ext = take - 3, hdr += 2;
            } else {
                sym = v;
            }
            L.syms[nt++] = (uint16_t)(sym | ext << 5);
            L.clfreq[sym] = (uint16_t)(L.clfreq[sym] + 1);
            run -= take;
        }
    }
    const int mcl = build_lengths(L.clfreq, kNC, 7, L.cllen, L.work, Col16{&L.work[kNC], L.work.st}, L.num, Col16{nullptr, 0});
    if (mcl == 1) {  // a code-length code of one symbol is incomplete, which inflaters refuse: a second one-bit code completes it
        const int other = L.cllen[0] != 0 ? 1 : 0;
        L.cllen[other] = 1;
    }
    for (int s = 0; s < kNC; s++) hdr += (uint32_t)L.clfreq[s] * L.cllen[s];
    int hclen = kNC;
    while (hclen > 4 && L.cllen[cl_order(hclen - 1)] == 0) hclen--;
    hdr += 3 + 14 + 3 * (uint32_t)hclen;
    GZ_STAMP(1);
    const uint32_t stored_bits = 3 + 7 + 32 + 8 * (uint32_t)n;
    if (hdr + body >= stored_bits) {
        // stored block (RFC 1951 §3.2.4): header bits, pad to a byte, LEN, ~LEN, the bytes (n <= 32768 < 65536)
        put(w, last ? 1u : 0u, 3);
        flush_bytes(w);
        put(w, (uint32_t)n | (~(uint32_t)n << 16), 32);
        int32_t i = 0;
        for (; i + 4 <= n; i += 4) put(w, ld32(p + i), 32);
        for (; i < n; i++) put(w, p[i], 8);
    } else {
        assign_codes(L.len, kNL, L.freq, L.num);
        assign_codes(dlen, kND, dfreq, L.num);
        assign_codes(L.cllen, kNC, L.clfreq, L.num);
        put(w, (last ? 1u : 0u) | 2u << 1, 3);
        put(w, (uint32_t)(hlit - 257) | (uint32_t)(hdist - 1) << 5 | (uint32_t)(hclen - 4) << 10, 14);
        for (int i = 0; i < hclen; i++)
            put(w, L.cllen[cl_order(i)], 3);
        for (int t = 0; t < nt; t++) {
            const int tok = L.syms[t], sym = tok & 31, ext = tok >> 5;
            put(w, L.clfreq[sym], L.cllen[sym]);
            if (sym == 16) put(w, (uint32_t)ext, 2);
            if (sym == 17) put(w, (uint32_t)ext, 3);
            if (sym == 18) put(w, (uint32_t)ext, 7);
        }
        GZ_STAMP(2);
        parse<true>(p, n, L, w, crc, prof);
        put(w, L.freq[256], L.len[256]);
        GZ_STAMP(3);
    }
}

template <int LANES>
__global__ __launch_bounds__(LANES) void gzip_kernel(const uint8_t* in, uint8_t* out, const cramgpu_block* desc, int32_t n,
                                                     int32_t* out_len, int32_t bgzf) {
    __shared__ DeflateLds<LANES> lds;
    for (int i = threadIdx.x; i < 256; i += LANES) {  // CRC-32 (IEEE 802.3, reflected 0xEDB88320) byte table
        uint32_t c = (uint32_t)i;
        for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        lds.crc[i] = c;
    }
    __syncthreads();
    const int32_t k = (int32_t)(blockIdx.x * LANES + threadIdx.x);
    if (k < n) {
        const int t = (int)threadIdx.x;
        const cramgpu_block D = desc[k];
        const Lane L{Col16{&lds.hash[0][t], LANES}, Col16{&lds.freq[0][t], LANES}, Col16{&lds.work[0][t], LANES},
                     Col16{&lds.syms[0][t], LANES}, Col16{&lds.clfreq[0][t], LANES}, Col16{&lds.num[0][t], LANES},
                     Col8{&lds.len[0][t], LANES}, Col8{&lds.cllen[0][t], LANES}, Col8{&lds.tag[0][t], LANES}, lds.crc};
        const uint8_t* p = in + D.in_off;
        const int32_t len = D.in_len;
        BitW w{out + D.out_off, 0, D.out_cap, 0ull, 0};
        long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        (void)prof;
        // member header (RFC 1952 §2.3): no name, no time, OS unknown; BGZF adds the 6-byte extra field whose size
        // slot is filled in at the end
        put(w, 0x00088B1Fu | (bgzf ? 4u << 24 : 0u), 32);
        put(w, 0u, 32);
        put(w, 0xFF00u, 16);
        if (bgzf) {
            put(w, 6u | (uint32_t)'B' << 16 | (uint32_t)'C' << 24, 32);
            put(w, 2u, 32);  // SLEN = 2, BSIZE = 0 for now
        }
        uint32_t c = 0xFFFFFFFFu;
        if (len == 0) {
            put(w, 3u, 10);  // one empty block of the fixed code: BFINAL, BTYPE 01, end of block (7 zero bits)
        } else {
            for (int32_t at = 0; at < len; at += kBlock) {
                const int32_t m = len - at < kBlock ? len - at : kBlock;
                deflate_block(p + at, m, at + m >= len, L, w, c, prof);
            }
        }
        flush_bytes(w);
#ifdef CRAMGPU_PROF_BUILD
        long long t0 = clock64();
#endif
        GZ_STAMP(4);
#ifdef CRAMGPU_PROF_BUILD
        if (k == 0 || k == n / 2) printf("gzip stream %d (%d bytes): parse1 %lld build %lld header %lld parse2 %lld crc %lld cycles; both passes: %lld table hits, %lld matches, %lld cycles behind a hit\n", k, len, prof[0], prof[1], prof[2], prof[3], prof[4], prof[5], prof[6], prof[7]);
#endif
        put(w, c ^ 0xFFFFFFFFu, 32);
        put(w, (uint32_t)len, 32);
        flush_bytes(w);
        int32_t rc = w.pos;
        if (w.pos > w.cap) rc = CRAMGPU_E_OUT_TOO_SMALL;
        if (len < 0 || D.out_cap < 0) rc = CRAMGPU_E_ARG;  // nothing was read or written: the loops above ran zero turns and cap < 0 admits no store
        if (bgzf && rc > 0) {
            if (rc > 65536) {
                rc = CRAMGPU_E_OUT_TOO_SMALL;  // a BGZF member holds at most 64 KiB: the caller cuts smaller blocks
            } else {
                w.o[16] = (uint8_t)(rc - 1);
                w.o[17] = (uint8_t)((rc - 1) >> 8);
            }
        }
        out_len[k] = rc;
    }
}


// =====================================================================================================================
// A WAVE per member (round 3).  The lane-per-stream kernel above walks one input position per turn through a chain of
// ~55 dependent instructions; here the 64 lanes take 64 consecutive positions at once:
//   A  every lane hashes the three bytes at its position, reads the table's candidate (the latest earlier position with
//      that hash, from the chunks before) and enters its own; it measures the match against that candidate and against
//      the position right before it (runs of one byte, which a table that only knows earlier chunks would miss); a
//      scalar walk over the lanes' lengths makes the greedy choice (jumping from match to match with a ballot: a chunk
//      of literals costs one turn), the lanes a token starts at write it to a token buffer in global memory and count
//      its symbols with LDS atomics;
//   B  lane 0 builds the length-limited codes and the header's run-length tokens (the routines of the kernel above);
//   C  64 tokens at a time: code + extra bits of a token (up to 48 bits), a prefix sum of the bit counts across the
//      wave, the bits ORed into an LDS staging buffer, whole words out with one store per lane.
// The member's CRC-32 no longer rides along a serial parse: every lane runs the byte table over its own 64th of the
// input (front-padded with zeros, which a zero state ignores; the lane holding the first byte starts from 0xFFFFFFFF) and
// the 64 states are combined in six steps with crc(A || B) = crc(A) * x^(8 |B|) mod P xor crc(B) (zlib's crc32_combine).
// Every conforming inflater returns the input, as before; the bytes differ from the lane kernel's (other candidates).
// =====================================================================================================================
constexpr int kWHashBits = 11, kWHash = 1 << kWHashBits;
constexpr int kNear = 8;  // at most 8: the near candidates come out of a 16-byte register window
constexpr int kStageWords = 104;  // 31 carried bits + 64 tokens of at most 48 bits = 3103 bits = 97 words, + the spill words

struct WaveLds {
    // Phases reuse each other's space (LDS per wave decides how many waves a CU holds, and the kernel lives on waves in
    // flight: 52 / 38 / 30 ms for 20 000 BGZF blocks at 8 / 12 / 16 waves per CU):
    //   hash: A — position + 1 of the latest occurrence in the block, 0 = none; B — the counting sort's 256 counters;
    //         C — the staging words of the bit writer (kStageWords of them)
    //   cnt:  A — symbol counts (LDS atomics), copied to freq in front of B;  work / syms: B — the code builder's arrays,
    //         syms keeps the header's run-length tokens until they are written
    union {
        uint16_t hash[kWHash];
        uint32_t stage[kStageWords];
    };
    union {
        uint32_t cnt[kAll];
        struct {
            uint16_t work[kNL];
            uint16_t syms[kAll];
        };
    };
    uint16_t freq[kAll];    // the counts as build_lengths wants them, then the (bit-reversed) codes
    uint8_t len[kAll];
    uint16_t clfreq[kNC];
    uint8_t cllen[kNC];
    uint16_t num[16];
    uint32_t crc[256];
    int32_t misc[8];  // hlit, hdist, hclen, nt, header bits
};
static_assert(sizeof(uint16_t) * kWHash >= sizeof(uint32_t) * kStageWords, "the staging words fit the match table");
static_assert(kWHash >= 256, "the match table doubles as the 256 counters of the counting sort");

#ifdef HIP_SIM
#define GZ_UNI(x) (x)
#else
#define GZ_UNI(x) __builtin_amdgcn_readfirstlane((int)(x))
#endif

__device__ __forceinline__ uint32_t mulmodp(uint32_t a, uint32_t b) {  // a * b mod P, x^0 = bit 31 (zlib's multmodp)
    uint32_t p = 0;
    for (int i = 0; i < 32; i++) {
        p ^= (a & (0x80000000u >> i)) ? b : 0u;
        b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
    }
    return p;
}

// CRC-32 of p[0, n) by the whole wave (n > 0)
__device__ uint32_t wave_crc(const uint8_t* p, int32_t n, const uint32_t* T, int lane) {
    const int32_t seg = (n + 63) / 64, pad = 64 * seg - n;
    const int32_t lo = lane * seg - pad, hi = lo + seg;
    uint32_t c = 0;
    if (hi > 0) {
        if (lo <= 0) c = 0xFFFFFFFFu;
        int32_t i = lo > 0 ? lo : 0;
        for (; i + 4 <= hi; i += 4) {
            const uint32_t w = ld32(p + i);
            c = T[(c ^ w) & 255u] ^ (c >> 8);
            c = T[(c ^ (w >> 8)) & 255u] ^ (c >> 8);
            c = T[(c ^ (w >> 16)) & 255u] ^ (c >> 8);
            c = T[(c ^ (w >> 24)) & 255u] ^ (c >> 8);
        }
        for (; i < hi; i++) c = T[(c ^ p[i]) & 255u] ^ (c >> 8);
    }
    uint32_t xd = 0x80000000u, base = 0x00800000u;  // x^(8 seg): square and multiply from x^8
    for (int32_t e = seg; e != 0; e >>= 1) {
        if (e & 1) xd = mulmodp(xd, base);
        base = mulmodp(base, base);
    }
    for (int d = 1; d < 64; d <<= 1) {  // lane l (a multiple of 2 d) <- [l, l + d) followed by [l + d, l + 2 d)
        const uint32_t other = (uint32_t)__shfl((int)c, (lane + d) & 63);
        c = mulmodp(xd, c) ^ other;
        xd = mulmodp(xd, xd);
    }
    return (uint32_t)__shfl((int)c, 0) ^ 0xFFFFFFFFu;
}

__device__ __forceinline__ int match_len(const uint8_t* p, int32_t c, int32_t i, int32_t maxl, int l = 0) {  // l bytes are known to agree
    bool go = true;
    while (go && l + 4 <= maxl) {
        const uint32_t x = ld32(p + i + l) ^ ld32(p + c + l);
        if (x != 0) {
            l += (__ffs((int)x) - 1) >> 3;
            go = false;
        } else {
            l += 4;
        }
    }
    while (go && l < maxl) {
        if (p[i + l] == p[c + l])
            l++;
        else
            go = false;
    }
    return l;
}

// length 3..258 -> symbol 257..285 + extra bits, distance 1..32768 -> symbol 0..29 + extra bits (RFC 1951 §3.2.5)
__device__ __forceinline__ void len_symbol(int lm, int& ls, int& leb) {  // lm = length - 3
    ls = 257 + lm, leb = 0;
    if (lm >= 8) {
        const int k = 31 - __clz(lm);
        leb = k - 2;
        ls = 261 + 4 * leb + ((lm >> leb) & 3);
    }
    if (lm == 255) ls = 285, leb = 0;
}
__device__ __forceinline__ void dist_symbol(int dm, int& ds, int& deb) {  // dm = distance - 1
    ds = dm, deb = 0;
    if (dm >= 4) {
        const int k = 31 - __clz(dm);
        deb = k - 1;
        ds = 2 * k + ((dm >> deb) & 1);
    }
}
__device__ __forceinline__ int extra_bits_of(int s) {  // of literal / length symbol s < kNL, or distance symbol s - kNL
    if (s < kNL) return s >= 265 && s < 285 ? (s - 261) >> 2 : 0;
    const int d = s - kNL;
    return d >= 4 ? (d >> 1) - 1 : 0;
}

// Phase B by the wave.  The used symbols of freq[0, n) sorted ascending by (count, symbol) into (A, S): every lane ranks
// its own symbols (lane, lane + 64, ...) against all n by broadcast reads — n * n / 64 compares per lane instead of lane 0's
// counting sort (~1700 dependent LDS round trips for 286 symbols).  Then lane 0 runs the in-place tree (serial by
// nature: m - 1 merges), and the lengths go to the symbols from num[] in parallel.  Returns the number of used symbols.
__device__ int wave_build_lengths(const uint16_t* freq, int n, int max_len, uint8_t* len, uint16_t* A, uint16_t* Ssym, uint16_t* num,
                                  int lane) {
    int m = 0;
    for (int s0 = 0; s0 < n; s0 += 64) {
        const int s = s0 + lane;
        const uint32_t f = s < n ? freq[s] : 0u;
        m += __popcll(__ballot(f != 0));
        if (s < n) len[s] = 0;
        if (f != 0) {
            const uint32_t key = f << 9 | (uint32_t)s;
            int rank = 0;
            for (int t = 0; t < n; t++) {
                const uint32_t g = freq[t];
                rank += (g != 0 && (g << 9 | (uint32_t)t) < key) ? 1 : 0;
            }
            A[rank] = (uint16_t)f;
            Ssym[rank] = (uint16_t)s;
        }
    }
    if (lane < 16) num[lane] = 0;
    __syncthreads();
    if (lane == 0) lengths_from_sorted(m, max_len, Col8{len, 1}, Col16{A, 1}, Col16{Ssym, 1}, Col16{num, 1}, false);
    __syncthreads();
    // sorted order: rarest first, so the longest codes go first — the k-th symbol gets the length whose running count
    // (from max_len down) first exceeds k
    for (int k = lane; k < m; k += 64) {
        int run = 0, l = max_len;
        for (; l >= 1; l--) {
            run += (int)num[l];
            if (k < run) break;
        }
        len[Ssym[k]] = (uint8_t)l;
    }
    __syncthreads();
    return m;
}

// canonical codes (RFC 1951 §3.2.2, bit-reversed) by the wave: per length the symbols that have it are counted with
// ballots, a symbol's code is the length's first code + the number of lower symbols of that length
__device__ void wave_assign_codes(const uint8_t* len, int n, uint16_t* code, uint16_t* num, int lane) {
    uint32_t mine[5] = {0, 0, 0, 0, 0};  // the codes of symbols lane, lane + 64, ...  (n <= 320)
    int c = 0, prev = 0;
    for (int l = 1; l < 16; l++) {
        c = (c + prev) << 1;  // first code of length l
        int run = 0;
#pragma unroll
        for (int r = 0; r < 5; r++) {
            const int s = 64 * r + lane;
            const bool has = s < n && len[s] == l;
            const unsigned long long mk = __ballot(has);
            if (has) mine[r] = __brev((uint32_t)(c + run + __popcll(mk & ((1ull << lane) - 1)))) >> (32 - l);
            run += __popcll(mk);
        }
        prev = run;
    }
    (void)num;
#pragma unroll
    for (int r = 0; r < 5; r++) {
        const int s = 64 * r + lane;
        if (s < n) code[s] = (uint16_t)mine[r];
    }
    __syncthreads();
}

__device__ void wave_block(const uint8_t* p, int32_t n, bool last, WaveLds& S, uint32_t* tok, BitW& w, int lane) {
    const bool l0 = lane == 0;
    for (int i = lane; i < kAll; i += 64) S.cnt[i] = 0;
    for (int i = lane; i < kWHash; i += 64) S.hash[i] = 0;
    __syncthreads();
    // ---- A: matches and tokens ----
    int32_t ntok = 0, carry = 0;
    // the 16 input bytes around a lane's position (eight before it, eight at it), loaded a chunk AHEAD of their use: the
    // kernel waits on memory 60 % of its wave cycles (SQ counters), and these loads depend on nothing
    struct Win {
        uint32_t w4, whi;
        unsigned long long prev8;
    };
    auto load_win = [&](int32_t i) -> Win {
        Win q{0u, 0u, 0ull};
        if (i + 4 <= n)
            q.w4 = ld32(p + i);
        else if (i < n)
            q.w4 = (uint32_t)p[i];
        if (i + 8 <= n) q.whi = ld32(p + i + 4);
        if (i + 4 <= n) {
            if (i >= 8) {
                q.prev8 = (unsigned long long)ld32(p + i - 8) | (unsigned long long)ld32(p + i - 4) << 32;
            } else {
                for (int k = 0; k < i; k++) q.prev8 |= (unsigned long long)p[k] << (8 * (8 - i + k));
            }
        }
        return q;
    };
    Win nxt = load_win(lane);
    for (int32_t base = 0; base < n; base += 64) {
        const int32_t i = base + lane;
        const bool valid = i < n, can = i + 4 <= n;
        (void)valid;
        const Win cur = nxt;
        if (base + 64 < n) nxt = load_win(i + 64);
        const uint32_t w4 = cur.w4;
        const int lit = (int)(w4 & 255u);
        const uint32_t h = ((w4 & 0xFFFFFFu) * 2654435761u) >> (32 - kWHashBits);
        const int32_t cand = can ? (int32_t)S.hash[h] : 0;
        __syncthreads();  // every lane has read the table of the chunks before
        // lanes with one hash: the LAST position stays, whatever order the hardware serves the lanes of one write in
        // (the bytes of a member must not depend on it): a lane whose entry lost to an earlier position writes again
        if (can) S.hash[h] = (uint16_t)(i + 1);
        __syncthreads();
        for (;;) {
            const bool again = can && (int32_t)S.hash[h] < i + 1;
            if (__ballot(again) == 0) break;
            __syncthreads();
            if (again) S.hash[h] = (uint16_t)(i + 1);
            __syncthreads();
        }
        // Candidates: the positions right before this one, which the table cannot know (they are this chunk's) — distances
        // 1..kNear, nearest first: runs of one byte, values of a fixed width — and the table's.  Each is measured against the
        // eight bytes at the position with one unaligned load, no loop; only the winner is extended beyond eight bytes.
        int mlen = 0, dist = 0;
        const int32_t maxl = n - i < 258 ? n - i : 258;
        if (can) {
            const bool can8 = i + 8 <= n;
            const unsigned long long w8 = can8 ? (unsigned long long)w4 | (unsigned long long)cur.whi << 32 : 0ull;
            auto measure = [&](unsigned long long c8, int d) {  // c8: the eight bytes at i - d
                int l;
                if (can8) {
                    const unsigned long long x = c8 ^ w8;
                    l = x != 0 ? (__ffsll((long long)x) - 1) >> 3 : 8;
                } else {
                    l = (((uint32_t)c8 ^ w4) & 0xFFFFFFu) == 0 ? 3 : 0;
                }
                if (l >= 3 && l > mlen) mlen = l, dist = d;
            };
            // the eight bytes before the position and the eight at it are a 16-byte window that holds every near candidate
            const unsigned long long prev8 = cur.prev8;
            const unsigned long long cur8 = can8 ? w8 : (unsigned long long)w4;
#pragma unroll
            for (int d = 1; d <= kNear; d++)
                if (i >= d) measure(d == 8 ? prev8 : (prev8 >> (8 * (8 - d))) | (cur8 << (8 * d)), d);
            if (cand != 0) {
                const int32_t c = cand - 1;
                measure(can8 ? (unsigned long long)ld32(p + c) | (unsigned long long)ld32(p + c + 4) << 32 : (unsigned long long)ld32(p + c), i - c);
            }
            if (mlen != 0 && (mlen == 8 || !can8)) mlen = match_len(p, i - dist, i, maxl, mlen);
            if (mlen == 3 && dist > 4096) mlen = 0, dist = 0;  // zlib's TOO_FAR: such a match costs more bits than three literals
        }
        // the greedy choice: from `carry` (positions the last chunk's final match still covers) literal by literal up to
        // the next lane that holds a match, then over that match
        const unsigned long long M = __ballot(mlen > 0);
        const int nvalid = n - base < 64 ? n - base : 64;
        unsigned long long starts = 0;
        int pos = carry;
        while (pos < nvalid) {
            const unsigned long long rest = M >> pos;
            const int t = rest != 0 ? __ffsll((long long)rest) - 1 : 64;
            const int lit_end = pos + t < nvalid ? pos + t : nvalid;
            const int k = lit_end - pos;
            if (k > 0) starts |= (k >= 64 ? ~0ull : ((1ull << k) - 1)) << pos;
            pos = lit_end;
            if (pos < nvalid) {
                starts |= 1ull << pos;
                pos = GZ_UNI(pos + __shfl(mlen, pos));
            }
        }
        carry = pos - 64;
        if ((starts >> lane) & 1ull) {
            const int at = ntok + __popcll(starts & ((1ull << lane) - 1));
            if (mlen > 0) {
                int ls, leb, ds, deb;
                len_symbol(mlen - 3, ls, leb);
                dist_symbol(dist - 1, ds, deb);
                tok[at] = 0x80000000u | (uint32_t)(mlen - 3) << 16 | (uint32_t)(dist - 1);
                atomicAdd(&S.cnt[ls], 1u);
                atomicAdd(&S.cnt[kNL + ds], 1u);
            } else {
                tok[at] = (uint32_t)lit;
                atomicAdd(&S.cnt[lit], 1u);
            }
        }
        ntok += __popcll(starts);
    }
    __syncthreads();
    if (l0) S.cnt[256] = 1;  // end of block
    __syncthreads();
    for (int i = lane; i < kAll; i += 64) S.freq[i] = (uint16_t)S.cnt[i];
    __syncthreads();
    // ---- B: codes and header (lane 0; the arrays are the wave's own) ----
    const Col16 work{S.work, 1}, clfreq{S.clfreq, 1}, num{S.num, 1};
    const Col8 dlen{S.len + kNL, 1}, cllen{S.cllen, 1};
    wave_build_lengths(S.freq, kNL, 15, S.len, S.work, S.syms, S.num, lane);
    wave_build_lengths(S.freq + kNL, kND, 15, S.len + kNL, S.work, S.syms, S.num, lane);
    if (l0) {
        int hlit = kNL, hdist = kND;
        while (hlit > 257 && S.len[hlit - 1] == 0) hlit--;
        while (hdist > 1 && dlen[hdist - 1] == 0) hdist--;
        const int total = hlit + hdist;
        for (int s = 0; s < kNC; s++) S.clfreq[s] = 0;
        int nt = 0;
        uint32_t hdr = 0;
        auto seq = [&](int i) -> int { return i < hlit ? S.len[i] : S.len[kNL + i - hlit]; };
        for (int i = 0; i < total;) {
            const int v = seq(i);
            int run = 1;
            while (i + run < total && seq(i + run) == v) run++;
            i += run;
            if (v != 0) {
                S.syms[nt++] = (uint16_t)v;
                S.clfreq[v] = (uint16_t)(S.clfreq[v] + 1);
                run--;
            }
            while (run > 0) {
                int sym, ext = 0, take = 1;
                if (v == 0 && run >= 11) {
                    take = run < 138 ? run : 138;
                    sym = 18, ext = take - 11, hdr += 7;
                } else if (v == 0 && run >= 3) {
                    take = run;
                    sym = 17, ext = take - 3, hdr += 3;
                } else if (v != 0 && run >= 3) {
                    take = run < 6 ? run : 6;
                    sym = 16, ext = take - 3, hdr += 2;
                } else {
                    sym = v;
                }
                S.syms[nt++] = (uint16_t)(sym | ext << 5);
                S.clfreq[sym] = (uint16_t)(S.clfreq[sym] + 1);
                run -= take;
            }
        }
        const int mcl = build_lengths(clfreq, kNC, 7, cllen, work, Col16{S.work + kNC, 1}, num, Col16{nullptr, 0});
        if (mcl == 1) {  // see deflate_block
            const int other = S.cllen[0] != 0 ? 1 : 0;
            S.cllen[other] = 1;
        }
        for (int s = 0; s < kNC; s++) hdr += (uint32_t)S.clfreq[s] * S.cllen[s];
        int hclen = kNC;
        while (hclen > 4 && S.cllen[cl_order(hclen - 1)] == 0) hclen--;
        hdr += 3 + 14 + 3 * (uint32_t)hclen;
        S.misc[0] = hlit, S.misc[1] = hdist, S.misc[2] = hclen, S.misc[3] = nt, S.misc[4] = (int32_t)hdr;
    }
    __syncthreads();
    uint32_t body = 0;
    for (int s = lane; s < kAll; s += 64) body += (uint32_t)S.freq[s] * (uint32_t)(S.len[s] + extra_bits_of(s));  // freq: still the counts
    for (int d = 32; d > 0; d >>= 1) body += (uint32_t)__shfl_xor((int)body, d);
    const int hlit = S.misc[0], hdist = S.misc[1], hclen = S.misc[2], nt = S.misc[3];
    const uint32_t hdr = (uint32_t)S.misc[4];
    const uint32_t stored_bits = 3 + 7 + 32 + 8 * (uint32_t)n;
    if (hdr + body >= stored_bits) {
        // stored block (RFC 1951 §3.2.4): header bits, pad to a byte, LEN, ~LEN, then the bytes, copied by the wave
        put(w, last ? 1u : 0u, 3, l0);
        flush_bytes(w, l0);
        put(w, (uint32_t)n | (~(uint32_t)n << 16), 32, l0);  // the accumulator is empty behind it
        for (int32_t i = lane; i < n; i += 64)
            if (w.pos + i < w.cap) w.o[w.pos + i] = p[i];
        w.pos += n;
        __syncthreads();
        return;
    }
    wave_assign_codes(S.len, kNL, S.freq, S.num, lane);
    wave_assign_codes(S.len + kNL, kND, S.freq + kNL, S.num, lane);
    if (l0) assign_codes(cllen, kNC, clfreq, num);
    __syncthreads();
    put(w, (last ? 1u : 0u) | 2u << 1, 3, l0);
    put(w, (uint32_t)(hlit - 257) | (uint32_t)(hdist - 1) << 5 | (uint32_t)(hclen - 4) << 10, 14, l0);
    for (int i = 0; i < hclen; i++) put(w, S.cllen[cl_order(i)], 3, l0);
    for (int t = 0; t < nt; t++) {
        const int tk = S.syms[t], sym = tk & 31, ext = tk >> 5;
        put(w, S.clfreq[sym], S.cllen[sym], l0);
        if (sym == 16) put(w, (uint32_t)ext, 2, l0);
        if (sym == 17) put(w, (uint32_t)ext, 3, l0);
        if (sym == 18) put(w, (uint32_t)ext, 7, l0);
    }
    // ---- C: the codes of 64 tokens at a time ----
    for (int32_t tb = 0; tb < ntok; tb += 64) {
        const int32_t t = tb + lane;
        unsigned long long v = 0;
        int nb = 0;
        if (t < ntok) {
            const uint32_t tk = tok[t];
            if (tk >> 31) {
                const int lm = (int)((tk >> 16) & 255u), dm = (int)(tk & 0x7FFFu);
                int ls, leb, ds, deb;
                len_symbol(lm, ls, leb);
                dist_symbol(dm, ds, deb);
                const int ll = S.len[ls], dl = S.len[kNL + ds];
                v = (unsigned long long)S.freq[ls] | (unsigned long long)(lm & ((1 << leb) - 1)) << ll |
                    (unsigned long long)S.freq[kNL + ds] << (ll + leb) | (unsigned long long)(dm & ((1 << deb) - 1)) << (ll + leb + dl);
                nb = ll + leb + dl + deb;
            } else {
                v = S.freq[tk];
                nb = S.len[tk];
            }
        }
        int incl = nb;
        for (int d = 1; d < 64; d <<= 1) {
            const int x = __shfl_up(incl, d);
            if (lane >= d) incl += x;
        }
        const int total = __shfl(incl, 63);
        const int off = w.cnt + incl - nb;
        const int full = (w.cnt + total) >> 5;
        for (int j = lane; j < full + 3 && j < kStageWords; j += 64) S.stage[j] = 0;
        __syncthreads();
        if (l0 && w.cnt != 0) atomicOr(&S.stage[0], (uint32_t)w.acc);
        if (nb != 0) {
            const int wi = off >> 5, sh = off & 31;
            const unsigned long long lo = v << sh;
            atomicOr(&S.stage[wi], (uint32_t)lo);
            if ((lo >> 32) != 0) atomicOr(&S.stage[wi + 1], (uint32_t)(lo >> 32));
            if (sh != 0 && (v >> (64 - sh)) != 0) atomicOr(&S.stage[wi + 2], (uint32_t)(v >> (64 - sh)));
        }
        __syncthreads();
        for (int j = lane; j < full; j += 64)
            if (w.pos + 4 * j + 4 <= w.cap) reinterpret_cast<W32u*>(w.o + w.pos + 4 * j)->v = S.stage[j];
        w.acc = S.stage[full];
        w.cnt = (w.cnt + total) & 31;
        w.pos += 4 * full;
        __syncthreads();
    }
    put(w, S.freq[256], S.len[256], l0);
}

#ifndef GZ_WPE
#define GZ_WPE 5  // 96 VGPRs (20 bytes of spill), five waves per SIMD: 30.2 ms at 16 waves per CU, 27.2 at 20 (six: 27.0 at 21)
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GZ_WPE, GZ_WPE))) void gzip_wave_kernel(const uint8_t* in, uint8_t* out, const cramgpu_block* desc, int32_t n,
                                                       int32_t* out_len, int32_t bgzf, uint32_t* ws) {
    __shared__ WaveLds S;
    const int lane = (int)threadIdx.x;
    const bool l0 = lane == 0;
    for (int i = lane; i < 256; i += 64) {  // CRC-32 (IEEE 802.3, reflected 0xEDB88320) byte table
        uint32_t c = (uint32_t)i;
        for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        S.crc[i] = c;
    }
    __syncthreads();
    uint32_t* tok = ws + (size_t)blockIdx.x * kBlock;
    for (int32_t k = (int32_t)blockIdx.x; k < n; k += (int32_t)gridDim.x) {
        const cramgpu_block D = desc[k];
        const int32_t len = D.in_len;
        if (len < 0 || D.out_cap < 0) {
            if (l0) out_len[k] = CRAMGPU_E_ARG;
            continue;
        }
        const uint8_t* p = in + D.in_off;
        BitW w{out + D.out_off, 0, D.out_cap, 0ull, 0};
        put(w, 0x00088B1Fu | (bgzf ? 4u << 24 : 0u), 32, l0);  // member header: see gzip_kernel
        put(w, 0u, 32, l0);
        put(w, 0xFF00u, 16, l0);
        if (bgzf) {
            put(w, 6u | (uint32_t)'B' << 16 | (uint32_t)'C' << 24, 32, l0);
            put(w, 2u, 32, l0);
        }
        uint32_t crc = 0;
        if (len == 0) {
            put(w, 3u, 10, l0);
        } else {
            crc = wave_crc(p, len, S.crc, lane);
            for (int32_t at = 0; at < len; at += kBlock) {
                const int32_t m = len - at < kBlock ? len - at : kBlock;
                wave_block(p + at, m, at + m >= len, S, tok, w, lane);
            }
        }
        flush_bytes(w, l0);
        put(w, crc, 32, l0);
        put(w, (uint32_t)len, 32, l0);
        flush_bytes(w, l0);
        int32_t rc = w.pos;
        if (w.pos > w.cap) rc = CRAMGPU_E_OUT_TOO_SMALL;
        if (bgzf && rc > 0) {
            if (rc > 65536) {
                rc = CRAMGPU_E_OUT_TOO_SMALL;
            } else if (l0) {
                w.o[16] = (uint8_t)(rc - 1);
                w.o[17] = (uint8_t)((rc - 1) >> 8);
            }
        }
        if (l0) out_len[k] = rc;
        __syncthreads();  // the member's last stores (lane 0) before the next member's tables are cleared
    }
}

}  // namespace

size_t gzip_ws_bytes(int n_waves) { return (size_t)n_waves * kBlock * sizeof(uint32_t); }

hipError_t launch_gzip(const uint8_t* in, uint8_t* out, const cramgpu_block* d_desc, int32_t n, int32_t* out_len, int32_t bgzf,
                       uint32_t* ws, int n_waves, hipStream_t s) {
    if (n == 0) return hipSuccess;
    static const int forced = [] {
        const char* e = getenv("CRAMGPU_GZIP_LANES");
        return e ? atoi(e) : 0;
    }();
    if (ws != nullptr && n_waves > 0 && !(forced == 2 || forced == 4 || forced == 8)) {
        // a wave per member, resident waves striding over the members (the caller sorts them longest first)
        hipLaunchKernelGGL(gzip_wave_kernel, dim3((unsigned)(n < n_waves ? n : n_waves)), dim3(64), 0, s, in, out, d_desc, n, out_len,
                           bgzf, ws);
        return hipGetLastError();
    }
    // the lane-per-stream kernel of round 2 (CRAMGPU_GZIP_LANES = 2 / 4 / 8 streams per wave)
    const int lanes = forced == 2 || forced == 4 || forced == 8 ? forced : 2;
#define GZIP_LAUNCH(L) \
    hipLaunchKernelGGL(gzip_kernel<L>, dim3((unsigned)((n + L - 1) / L)), dim3(L), 0, s, in, out, d_desc, n, out_len, bgzf)
    switch (lanes) {
        case 8: GZIP_LAUNCH(8); break;
        case 4: GZIP_LAUNCH(4); break;
        default: GZIP_LAUNCH(2); break;
    }
#undef GZIP_LAUNCH
    return hipGetLastError();
}
