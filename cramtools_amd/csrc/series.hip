// W — CRAM data series from flat records and back for gfx950.
//
// Reference: htsjdk!encoding/writer/Writer#write L139-261, htsjdk!encoding/reader/CramRecordReader#read L59-211,
// htsjdk!io/ITF8#writeUnsignedITF8 L53-84 / #readUnsignedITF8 (bytecode only); the same field order in source:
// src/main/java/htsjdk/samtools/cram/encoding/reader/ReaderToBAM.java:64-176, ReadFeatureBuffer.java:41-100.
//
// The reference appends every value of a record to one growing byte stream per series: a position in a stream is a
// prefix sum over all earlier records.  Here a workgroup owns a container and walks its records in tiles of 256 (a
// thread per record): per tile the byte counts of the 30 series are scanned in LDS (a wave per series, four values per
// lane), every thread then writes its record's values at its own offsets, and the quality / base arrays — nearly all
// the bytes — leave as one coalesced copy per tile when the tile's records lie back to back (every lossless container)
// or as one wave-wide copy per record otherwise.  Three launches: count (stream lengths per container), layout
// (16-byte aligned offsets, capacity check), write.  HBM-bound: algorithmic bytes = the record columns read once +
// the streams written once.
#include "internal.h"

namespace {

constexpr int NS = CRAMGPU_N_SERIES;
constexpr int kTile = 256;

__device__ __forceinline__ uint32_t itf8_len(uint32_t v) {
    return 1u + (v >= (1u << 7)) + (v >= (1u << 14)) + (v >= (1u << 21)) + (v >= (1u << 28));
}
// ITF8#writeUnsignedITF8 L53-84 (widths as SURVEY.md §8(d) lists them)
__device__ __forceinline__ uint8_t* itf8_put(uint8_t* p, uint32_t v) {
    if (v < (1u << 7)) {
        p[0] = (uint8_t)v;
        return p + 1;
    }
    if (v < (1u << 14)) {
        p[0] = (uint8_t)(0x80u | (v >> 8));
        p[1] = (uint8_t)v;
        return p + 2;
    }
    if (v < (1u << 21)) {
        p[0] = (uint8_t)(0xC0u | (v >> 16));
        p[1] = (uint8_t)(v >> 8);
        p[2] = (uint8_t)v;
        return p + 3;
    }
    if (v < (1u << 28)) {
        p[0] = (uint8_t)(0xE0u | (v >> 24));
        p[1] = (uint8_t)(v >> 16);
        p[2] = (uint8_t)(v >> 8);
        p[3] = (uint8_t)v;
        return p + 4;
    }
    p[0] = (uint8_t)(0xF0u | (v >> 28));
    p[1] = (uint8_t)(v >> 20);
    p[2] = (uint8_t)(v >> 12);
    p[3] = (uint8_t)(v >> 4);
    p[4] = (uint8_t)(v & 0x0Fu);
    return p + 5;
}

// lanes below this one that are set in m
__device__ __forceinline__ uint32_t mbcnt64(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

struct WArgs {
    cramgpu_reads r;
    cramgpu_record_cols c;
    cramgpu_features f;
    uint8_t* arena;
    int64_t cap;
    int64_t* off;
    int32_t* len;
    int32_t* status;
    uint8_t* alias;  // [n_cont] workspace: QS of the container is not copied (CRAMGPU_WS_ALIAS_QS)
};

// first record of container c: the records of a container are contiguous and read_cont ascends
__device__ __forceinline__ int32_t cont_first(const cramgpu_reads& r, int32_t c) {
    if (r.read_cont == nullptr) return c <= 0 ? 0 : r.n_reads;
    int32_t lo = 0, hi = r.n_reads;
    while (lo < hi) {
        const int32_t mid = lo + ((hi - lo) >> 1);
        if (r.read_cont[mid] < c)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

struct Sizes {
    uint32_t v[NS];
};

// bytes record r adds to every series (Writer#write's branches); bad != 0: an operator byte the reference would throw on
__device__ __forceinline__ void record_sizes(const WArgs& A, int32_t r, bool multiref, int32_t prev_start, Sizes& z, int& bad) {
#pragma unroll
    for (int k = 0; k < NS; k++) z.v[k] = 0;
    const uint32_t cf = (uint32_t)A.c.comp_flags[r], bf = (uint32_t)A.c.bam_flags[r];
    const uint32_t L = (uint32_t)A.r.read_len[r];
    z.v[CRAMGPU_DS_BF] = itf8_len(bf);
    z.v[CRAMGPU_DS_CF] = itf8_len(cf);
    if (multiref) z.v[CRAMGPU_DS_RI] = itf8_len((uint32_t)A.c.ref_id[r]);
    z.v[CRAMGPU_DS_RL] = itf8_len(L);
    z.v[CRAMGPU_DS_AP] = itf8_len((uint32_t)(A.c.ap_delta ? A.r.aln_start[r] - prev_start : A.r.aln_start[r]));
    z.v[CRAMGPU_DS_RG] = itf8_len((uint32_t)A.c.read_group[r]);
    // Writer#write: `if (captureReadNames) RN` in front of the mate block, `if (!captureReadNames) RN` inside it
    if (A.c.names != nullptr && (A.c.capture_names || (cf & CRAMGPU_CF_DETACHED)))
        z.v[CRAMGPU_DS_RN] = (uint32_t)(A.c.name_len[r] > 0 ? A.c.name_len[r] : 0) + 1u;
    if (cf & CRAMGPU_CF_DETACHED) {
        z.v[CRAMGPU_DS_MF] = itf8_len((uint32_t)A.c.mate_flags[r]);
        z.v[CRAMGPU_DS_NS] = itf8_len((uint32_t)A.c.mate_ref[r]);
        z.v[CRAMGPU_DS_NP] = itf8_len((uint32_t)A.c.mate_start[r]);
        z.v[CRAMGPU_DS_TS] = itf8_len((uint32_t)A.c.tlen[r]);
    } else if (cf & CRAMGPU_CF_HAS_MATE_DOWNSTREAM) {
        z.v[CRAMGPU_DS_NF] = itf8_len((uint32_t)A.c.next_frag[r]);
    }
    z.v[CRAMGPU_DS_TL] = itf8_len((uint32_t)A.c.tag_list[r]);
    if (!(bf & CRAMGPU_BF_SEGMENT_UNMAPPED)) {
        const int64_t f0 = A.f.off[r], f1 = A.f.off[r + 1];
        z.v[CRAMGPU_DS_FN] = itf8_len((uint32_t)(f1 - f0));
        z.v[CRAMGPU_DS_FC] = (uint32_t)(f1 - f0);
        uint32_t fp = 0, bs = 0, dl = 0, rs = 0, hc = 0, pd = 0, ba = 0, qs = 0, in = 0, sc = 0;
        int32_t prev = 0;
        for (int64_t f = f0; f < f1; f++) {
            const int32_t p = A.f.pos[f];
            fp += itf8_len((uint32_t)(p - prev));
            prev = p;
            const uint32_t a = (uint32_t)A.f.a[f];
            switch (A.f.op[f]) {
                case 'B': ba++, qs++; break;
                case 'X': bs++; break;
                case 'I': in += (uint32_t)A.f.b[f] + 1u; break;
                case 'S': sc += (uint32_t)A.f.b[f] + 1u; break;
                case 'H': hc += itf8_len(a); break;
                case 'P': pd += itf8_len(a); break;
                case 'D': dl += itf8_len(a); break;
                case 'N': rs += itf8_len(a); break;
                case 'i': ba++; break;
                case 'Q': qs++; break;
                default: bad = 1;
            }
        }
        z.v[CRAMGPU_DS_FP] = fp;
        z.v[CRAMGPU_DS_BS] = bs;
        z.v[CRAMGPU_DS_DL] = dl;
        z.v[CRAMGPU_DS_RS] = rs;
        z.v[CRAMGPU_DS_HC] = hc;
        z.v[CRAMGPU_DS_PD] = pd;
        z.v[CRAMGPU_DS_BA] = ba;
        z.v[CRAMGPU_DS_IN] = in;
        z.v[CRAMGPU_DS_SC] = sc;
        z.v[CRAMGPU_DS_MQ] = itf8_len((uint32_t)A.r.mapq[r]);
        z.v[CRAMGPU_DS_QS] = qs + ((cf & CRAMGPU_CF_FORCE_PRESERVE_QS) ? L : 0u);
    } else {
        z.v[CRAMGPU_DS_BA] = (cf & CRAMGPU_CF_UNKNOWN_BASES) ? 0u : L;
        z.v[CRAMGPU_DS_QS] = (cf & CRAMGPU_CF_FORCE_PRESERVE_QS) ? L : 0u;
    }
}

__device__ __forceinline__ int32_t start_before(const WArgs& A, int32_t r, int32_t first, int32_t c) {
    if (r > first) return A.r.aln_start[r - 1];
    return A.c.cont_ap_base ? A.c.cont_ap_base[c] : A.r.aln_start[first];
}

__global__ __launch_bounds__(kTile) void series_count_kernel(WArgs A) {
    __shared__ uint32_t tot[NS];
    __shared__ int noalias;
    const int32_t c = (int32_t)blockIdx.x;
    const int tid = (int)threadIdx.x;
    if (tid < NS) tot[tid] = 0;
    if (tid == 0) noalias = 0;
    __syncthreads();
    const int32_t first = cont_first(A.r, c), last = cont_first(A.r, c + 1);
    const bool multiref = A.c.cont_multiref != nullptr && A.c.cont_multiref[c] != 0;
    Sizes acc;
#pragma unroll
    for (int k = 0; k < NS; k++) acc.v[k] = 0;
    int bad = 0;
    bool alias_ok = true;
    for (int32_t r = first + tid; r < last; r += kTile) {
        Sizes z;
        record_sizes(A, r, multiref, start_before(A, r, first, c), z, bad);
#pragma unroll
        for (int k = 0; k < NS; k++) acc.v[k] += z.v[k];
        // the record adds exactly its array to QS and the next record's array follows it in reads->quals
        if (z.v[CRAMGPU_DS_QS] != (uint32_t)A.r.read_len[r] || !(A.c.comp_flags[r] & CRAMGPU_CF_FORCE_PRESERVE_QS) ||
            (r + 1 < last && A.r.base_off[r + 1] != A.r.base_off[r] + A.r.read_len[r]))
            alias_ok = false;
    }
#pragma unroll
    for (int k = 0; k < NS; k++)
        if (acc.v[k]) atomicAdd(&tot[k], acc.v[k]);
    if (bad) *A.status = CRAMGPU_E_ARG;
    if (!alias_ok) noalias = 1;
    __syncthreads();
    if (tid < NS) A.len[(int64_t)c * NS + tid] = (int32_t)tot[tid];
    // CRAMGPU_WS_ALIAS_QS: the container's QS stream is its records' quality arrays where they lie
    if (tid == 0)
        A.alias[c] = (A.c.flags & CRAMGPU_WS_ALIAS_QS) && !noalias && last > first &&
                     (((uintptr_t)(A.r.quals + A.r.base_off[first])) & 15) == 0;
}

// offsets: the streams of container 0 in content-id order, then container 1, ...; each on a 16-byte boundary
__global__ __launch_bounds__(1024) void series_layout_kernel(WArgs A, int32_t n_cont) {
    __shared__ int64_t part[1024];
    __shared__ int64_t carry;
    const int tid = (int)threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int32_t c0 = 0; c0 < n_cont; c0 += 1024) {
        const int32_t c = c0 + tid;
        int64_t t = 0;
        if (c < n_cont)
            for (int k = 0; k < NS; k++)
                if (!(k == CRAMGPU_DS_QS && A.alias[c])) t += ((int64_t)A.len[(int64_t)c * NS + k] + 15) & ~(int64_t)15;
        part[tid] = t;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {  // inclusive scan
            const int64_t v = tid >= d ? part[tid - d] : 0;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        if (c < n_cont) {
            int64_t o = carry + part[tid] - t;
            for (int k = 0; k < NS; k++) {
                if (k == CRAMGPU_DS_QS && A.alias[c]) {
                    A.off[(int64_t)c * NS + k] = (A.r.quals + A.r.base_off[cont_first(A.r, c)]) - A.arena;
                    continue;
                }
                A.off[(int64_t)c * NS + k] = o;
                o += ((int64_t)A.len[(int64_t)c * NS + k] + 15) & ~(int64_t)15;
            }
        }
        __syncthreads();
        if (tid == 0) carry += part[1023];
        __syncthreads();
    }
    if (tid == 0 && carry + 16 > A.cap && *A.status == 0) *A.status = CRAMGPU_E_INSUFFICIENT;
}

struct __attribute__((packed)) PackedU128 { uint32_t a, b, c, d; };

// n bytes src -> dst by the whole workgroup: 16-byte stores on dst's alignment, unaligned 16-byte loads
__device__ __forceinline__ void wg_copy(uint8_t* dst, const uint8_t* src, int64_t n, int tid) {
    int64_t head = (16 - ((uintptr_t)dst & 15)) & 15;
    if (head > n) head = n;
    if (tid < head) dst[tid] = src[tid];
    const int64_t n16 = (n - head) >> 4;
    for (int64_t i = tid; i < n16; i += kTile) {
        const PackedU128 v = *reinterpret_cast<const PackedU128*>(src + head + 16 * i);
        *reinterpret_cast<uint4*>(dst + head + 16 * i) = make_uint4(v.a, v.b, v.c, v.d);
    }
    const int64_t done = head + 16 * n16;
    if (tid < n - done) dst[done + tid] = src[done + tid];
}

__global__ __launch_bounds__(kTile) void series_write_kernel(WArgs A) {
    __shared__ uint32_t sz[NS][kTile];  // sizes, then exclusive prefixes within the tile
    __shared__ uint32_t tile_tot[NS];
    __shared__ int64_t base[NS];        // arena offset of the next byte of every series
    __shared__ int64_t job_src[kTile], job_dst[kTile];
    __shared__ uint32_t job_len[kTile];
    __shared__ int contig;
    if (*A.status != 0) return;
    const int32_t c = (int32_t)blockIdx.x;
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid < NS) base[tid] = A.off[(int64_t)c * NS + tid];
    const int32_t first = cont_first(A.r, c), last = cont_first(A.r, c + 1);
    const bool multiref = A.c.cont_multiref != nullptr && A.c.cont_multiref[c] != 0;
    const bool aliased = A.alias[c] != 0;
    for (int32_t t0 = first; t0 < last; t0 += kTile) {
        const int32_t r = t0 + tid;
        const bool live = r < last;
        Sizes z;
        int bad = 0;
        int32_t prev_start = 0;
        if (live) {
            prev_start = start_before(A, r, first, c);
            record_sizes(A, r, multiref, prev_start, z, bad);
        } else {
#pragma unroll
            for (int k = 0; k < NS; k++) z.v[k] = 0;
        }
#pragma unroll
        for (int k = 0; k < NS; k++) sz[k][tid] = z.v[k];
        if (tid == 0) contig = 1;
        __syncthreads();
        // a wave per series: four values per lane, wave-wide exclusive scan of the lane sums
        for (int k = wave; k < NS; k += kTile / 64) {
            const uint32_t v0 = sz[k][4 * lane], v1 = sz[k][4 * lane + 1], v2 = sz[k][4 * lane + 2], v3 = sz[k][4 * lane + 3];
            const uint32_t s = v0 + v1 + v2 + v3;
            uint32_t inc = s;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(inc, d);
                if (lane >= d) inc += o;
            }
            const uint32_t ex = inc - s;
            sz[k][4 * lane] = ex;
            sz[k][4 * lane + 1] = ex + v0;
            sz[k][4 * lane + 2] = ex + v0 + v1;
            sz[k][4 * lane + 3] = ex + v0 + v1 + v2;
            if (lane == 63) tile_tot[k] = inc;
        }
        __syncthreads();
        uint32_t qlen = 0;
        if (live) {
            const uint32_t cf = (uint32_t)A.c.comp_flags[r], bf = (uint32_t)A.c.bam_flags[r];
            const uint32_t L = (uint32_t)A.r.read_len[r];
#define SP(id) (A.arena + base[id] + sz[id][tid])
            itf8_put(SP(CRAMGPU_DS_BF), bf);
            itf8_put(SP(CRAMGPU_DS_CF), cf);
            if (multiref) itf8_put(SP(CRAMGPU_DS_RI), (uint32_t)A.c.ref_id[r]);
            itf8_put(SP(CRAMGPU_DS_RL), L);
            itf8_put(SP(CRAMGPU_DS_AP), (uint32_t)(A.c.ap_delta ? A.r.aln_start[r] - prev_start : A.r.aln_start[r]));
            itf8_put(SP(CRAMGPU_DS_RG), (uint32_t)A.c.read_group[r]);
            if (A.c.names != nullptr && (A.c.capture_names || (cf & CRAMGPU_CF_DETACHED))) {
                uint8_t* rn = SP(CRAMGPU_DS_RN);
                const uint8_t* src = A.c.names + A.c.name_off[r];
                const int32_t nl = A.c.name_len[r] > 0 ? A.c.name_len[r] : 0;
                for (int32_t i = 0; i < nl; i++) rn[i] = src[i];
                rn[nl] = '\t';
            }
            if (cf & CRAMGPU_CF_DETACHED) {
                itf8_put(SP(CRAMGPU_DS_MF), (uint32_t)A.c.mate_flags[r]);
                itf8_put(SP(CRAMGPU_DS_NS), (uint32_t)A.c.mate_ref[r]);
                itf8_put(SP(CRAMGPU_DS_NP), (uint32_t)A.c.mate_start[r]);
                itf8_put(SP(CRAMGPU_DS_TS), (uint32_t)A.c.tlen[r]);
            } else if (cf & CRAMGPU_CF_HAS_MATE_DOWNSTREAM) {
                itf8_put(SP(CRAMGPU_DS_NF), (uint32_t)A.c.next_frag[r]);
            }
            itf8_put(SP(CRAMGPU_DS_TL), (uint32_t)A.c.tag_list[r]);
            const uint8_t* rb = A.r.bases + A.r.base_off[r];
            uint8_t* qs = SP(CRAMGPU_DS_QS);
            if (!(bf & CRAMGPU_BF_SEGMENT_UNMAPPED)) {
                const int64_t f0 = A.f.off[r], f1 = A.f.off[r + 1];
                itf8_put(SP(CRAMGPU_DS_FN), (uint32_t)(f1 - f0));
                if (f1 > f0) {
                    uint8_t *fc = SP(CRAMGPU_DS_FC), *fp = SP(CRAMGPU_DS_FP), *bs = SP(CRAMGPU_DS_BS), *dl = SP(CRAMGPU_DS_DL),
                            *rs = SP(CRAMGPU_DS_RS), *hc = SP(CRAMGPU_DS_HC), *pd = SP(CRAMGPU_DS_PD), *ba = SP(CRAMGPU_DS_BA),
                            *in = SP(CRAMGPU_DS_IN), *sc = SP(CRAMGPU_DS_SC);
                    int32_t prev = 0;
                    for (int64_t f = f0; f < f1; f++) {
                        const uint8_t o = A.f.op[f];
                        const int32_t p = A.f.pos[f];
                        const uint32_t a = (uint32_t)A.f.a[f];
                        *fc++ = o;
                        fp = itf8_put(fp, (uint32_t)(p - prev));
                        prev = p;
                        switch (o) {
                            case 'B':
                                *ba++ = (uint8_t)a;
                                *qs++ = (uint8_t)A.f.b[f];
                                break;
                            case 'X': *bs++ = (uint8_t)(a >> 16); break;
                            case 'I': {
                                const int32_t n = A.f.b[f];
                                for (int32_t i = 0; i < n; i++) in[i] = rb[a + i];
                                in[n] = '\t';
                                in += n + 1;
                                break;
                            }
                            case 'S': {
                                const int32_t n = A.f.b[f];
                                for (int32_t i = 0; i < n; i++) sc[i] = rb[a + i];
                                sc[n] = '\t';
                                sc += n + 1;
                                break;
                            }
                            case 'H': hc = itf8_put(hc, a); break;
                            case 'P': pd = itf8_put(pd, a); break;
                            case 'D': dl = itf8_put(dl, a); break;
                            case 'N': rs = itf8_put(rs, a); break;
                            case 'i': *ba++ = (uint8_t)a; break;
                            case 'Q': *qs++ = (uint8_t)a; break;
                            default: break;
                        }
                    }
                }
                itf8_put(SP(CRAMGPU_DS_MQ), (uint32_t)A.r.mapq[r]);
            } else if (!(cf & CRAMGPU_CF_UNKNOWN_BASES)) {
                uint8_t* ba = SP(CRAMGPU_DS_BA);
                for (uint32_t i = 0; i < L; i++) ba[i] = rb[i];
            }
#undef SP
            if ((cf & CRAMGPU_CF_FORCE_PRESERVE_QS) && !aliased) {  // the quality array: copied by the group below
                qlen = L;
                job_src[tid] = A.r.base_off[r];
                job_dst[tid] = (int64_t)(qs - A.arena);
            }
        }
        job_len[tid] = qlen;
        __syncthreads();
        if (live && tid > 0 &&
            (qlen == 0 || job_len[tid - 1] == 0 || job_src[tid] != job_src[tid - 1] + job_len[tid - 1] ||
             job_dst[tid] != job_dst[tid - 1] + job_len[tid - 1]))
            contig = 0;
        if (tid == 0 && qlen == 0) contig = 0;
        __syncthreads();
        const int32_t cnt = last - t0 < kTile ? last - t0 : kTile;
        if (aliased) {
        } else if (contig) {
            const int64_t n = job_src[cnt - 1] + job_len[cnt - 1] - job_src[0];
            wg_copy(A.arena + job_dst[0], A.r.quals + job_src[0], n, tid);
        } else {
            for (int j = wave; j < cnt; j += kTile / 64) {
                const uint32_t n = job_len[j];
                const uint8_t* s = A.r.quals + job_src[j];
                uint8_t* d = A.arena + job_dst[j];
                for (uint32_t i = lane; i < n; i += 64) d[i] = s[i];
            }
        }
        __syncthreads();
        if (tid < NS) base[tid] += tile_tot[tid];
        __syncthreads();
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Reader: CramRecordReader#read (htsjdk, L59-211; field order as ReaderToBAM.java:64-176) over whole containers.
//   R0 layout of the value workspace   R1 a thread per (container, series) turns the ITF8 streams into int arrays and
//   lists the tab-closed items of IN / SC   R2 per-container bases / features / sequence bytes -> offsets of the batch
//   R3 a workgroup per container, tiles of 256 records: ranks of the records that carry the optional values, prefix
//   sums of read lengths / start deltas / feature counts, per-record cursors into the byte series (LDS scans as in the
//   writer), then every thread assembles its record; quality arrays leave as one coalesced copy per tile when the
//   tile's records all carry one.
// The ITF8 parse of R1 is sequential per stream (one 8-byte load per value): fine for Illumina-sized containers, slow
// for a container of 10 kbp reads whose FP stream holds millions of values.
struct RArgs {
    const uint8_t* arena;
    const int64_t* off;
    const int32_t* len;
    int32_t n_cont, n_reads;
    const int32_t* cont_first;  // [n_cont + 1]
    cramgpu_record_cols c;      // output columns; cont_multiref / cont_ap_base / ap_delta are inputs
    cramgpu_read_out o;
    cramgpu_features f;
    int32_t* vals;       // workspace: parsed values of the integer series, item starts of IN / SC
    int64_t* vals_off;   // [n_cont][NS]
    int32_t* vals_cnt;   // [n_cont][NS]
    int64_t* cont_base;  // [n_cont + 1][3]: bases, features, sequence bytes in front of container c
    int32_t* status;
    int64_t vals_cap;    // ints behind vals
};

__device__ __forceinline__ bool is_int_series(int k) {
    return k == CRAMGPU_DS_BF || k == CRAMGPU_DS_CF || k == CRAMGPU_DS_RI || k == CRAMGPU_DS_RL || k == CRAMGPU_DS_AP ||
           k == CRAMGPU_DS_RG || k == CRAMGPU_DS_MF || k == CRAMGPU_DS_NS || k == CRAMGPU_DS_NP || k == CRAMGPU_DS_TS ||
           k == CRAMGPU_DS_NF || k == CRAMGPU_DS_TL || k == CRAMGPU_DS_FN || k == CRAMGPU_DS_FP || k == CRAMGPU_DS_DL ||
           k == CRAMGPU_DS_RS || k == CRAMGPU_DS_HC || k == CRAMGPU_DS_PD || k == CRAMGPU_DS_MQ;
}
__device__ __forceinline__ bool is_item_series(int k) { return k == CRAMGPU_DS_IN || k == CRAMGPU_DS_SC || k == CRAMGPU_DS_RN; }

// R0: workspace slots: a series of n bytes holds at most n values (n + 1 item starts)
__global__ __launch_bounds__(1024) void series_ws_layout_kernel(RArgs A) {
    __shared__ int64_t part[1024];
    __shared__ int64_t carry;
    const int tid = (int)threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int32_t c0 = 0; c0 < A.n_cont; c0 += 1024) {
        const int32_t c = c0 + tid;
        int64_t t = 0;
        if (c < A.n_cont)
            for (int k = 0; k < NS; k++) {
                // the lengths come from block headers: a negative one would make the slots overlap
                const int32_t len = A.len[(int64_t)c * NS + k];
                if (len < 0) *A.status = CRAMGPU_E_CORRUPT;
                if (is_int_series(k) || is_item_series(k)) t += ((int64_t)(len < 0 ? 0 : len) + 4) & ~(int64_t)3;
            }
        part[tid] = t;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const int64_t v = tid >= d ? part[tid - d] : 0;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        if (c < A.n_cont) {
            int64_t o = carry + part[tid] - t;
            for (int k = 0; k < NS; k++) {
                A.vals_off[(int64_t)c * NS + k] = o;
                const int32_t len = A.len[(int64_t)c * NS + k];
                if (is_int_series(k) || is_item_series(k)) o += ((int64_t)(len < 0 ? 0 : len) + 4) & ~(int64_t)3;
            }
        }
        __syncthreads();
        if (tid == 0) carry += part[1023];
        __syncthreads();
    }
    // the workspace was sized from the caller's host-side byte count: the device lengths must fit it (the parse
    // launch writes a value per byte at most) or nothing is parsed
    if (tid == 0 && carry > A.vals_cap && *A.status == 0) *A.status = CRAMGPU_E_INSUFFICIENT;
}

// R1: a wave per (container, series).  A value's position depends on every value before it, but only through its FIRST
// byte: per 64-byte chunk (a byte per lane) four ballots give the masks of the bytes that would open a value of 1..5
// bytes, and a scalar walk marks the true starts — a whole run of equal-length values per turn (the run's starts are a
// periodic bit pattern ANDed with the mask of that length: one ctz finds where it breaks), so a stream of same-width
// values (RL, RG, BF, NS ...) costs one turn per chunk.  The start lanes then decode in parallel (branch-free) and
// store at cnt + rank.  IN / SC: a tab closes an item.  The stream is staged through LDS 1 KiB at a time (16-byte
// loads a block ahead, so the walk never waits for memory).
constexpr int kBlk = 1024;
__device__ __forceinline__ unsigned long long itf8_pattern(int m) {  // bit j set iff j % m == 0
    return m == 1 ? ~0ull : m == 2 ? 0x5555555555555555ull : m == 3 ? 0x9249249249249249ull : m == 4 ? 0x1111111111111111ull
                                                                                                     : 0x1084210842108421ull;
}
__global__ __launch_bounds__(64) void series_parse_kernel(RArgs A) {
    __shared__ __align__(16) uint8_t buf[2][kBlk];
    const int32_t c = (int32_t)blockIdx.y;
    const int k = (int)blockIdx.x;
    const int l = (int)threadIdx.x;
    const int64_t slot = (int64_t)c * NS + k;
    const bool item = is_item_series(k);
    // set by the layout launch (lengths that do not fit the workspace): one wave, one load, the same value in every lane.
    // (Other waves of THIS launch may set it to CRAMGPU_E_CORRUPT meanwhile; whoever sees that early just stops early.)
    if (__shfl(*A.status, 0) != 0) {
        if (l == 0) A.vals_cnt[slot] = 0;
        return;
    }
    if (!item && !is_int_series(k)) {
        if (l == 0) A.vals_cnt[slot] = 0;
        return;
    }
    const uint8_t* p = A.arena + A.off[slot];
    const int32_t n = A.len[slot];
    int32_t* out = A.vals + A.vals_off[slot];
    const bool aligned = ((uintptr_t)p & 15) == 0;
    auto fetch = [&](int32_t blk) -> uint4 {  // bytes [1024 blk + 16 l, +16) of the stream, zeros behind its end
        const int32_t o = blk * kBlk + 16 * l;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (o < n) {
            if (aligned) {
                v = *reinterpret_cast<const uint4*>(p + o);  // <= 15 bytes of the arena's padding read
            } else {
                uint32_t w[4] = {0, 0, 0, 0};
                for (int j = 0; j < 16; j++)
                    if (o + j < n) w[j >> 2] |= (uint32_t)p[o + j] << (8 * (j & 3));
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
            // bytes behind the end read as zero (a value must not borrow them: checked below)
            if (o + 16 > n) {
                uint32_t w[4] = {v.x, v.y, v.z, v.w};
                for (int j = 0; j < 16; j++)
                    if (o + j >= n) w[j >> 2] &= ~(255u << (8 * (j & 3)));
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
        return v;
    };
    const int32_t n_blk = (n + kBlk - 1) / kBlk;
    *reinterpret_cast<uint4*>(&buf[0][16 * l]) = fetch(0);
    *reinterpret_cast<uint4*>(&buf[1][16 * l]) = fetch(1);
    __syncthreads();
    int32_t cnt = 0, e = 0;
    long long acc = 0;
    bool bad = false;
    if (item && l == 0) out[0] = 0;
    for (int32_t blk = 0; blk < n_blk; blk++) {
        const uint4 ahead = fetch(blk + 2);  // lands in the buffer of this block once the block is done
        const int32_t c_end = (blk + 1) * kBlk < n ? (blk + 1) * kBlk : n;
        for (int32_t c0 = blk * kBlk; c0 < c_end; c0 += 64) {
            const int32_t q = c0 + l;
            const bool valid = q < n;
            uint32_t b[5];
#pragma unroll
            for (int j = 0; j < 5; j++) b[j] = buf[((q + j) >> 10) & 1][(q + j) & (kBlk - 1)];
            if (item) {
                const unsigned long long tabs = __ballot(valid && b[0] == '\t');
                if (valid && b[0] == '\t') out[cnt + 1 + (int32_t)mbcnt64(tabs)] = q + 1;
                cnt += __popcll(tabs);
                if (valid && q == n - 1 && b[0] != '\t') bad = true;  // the last item is not closed
                continue;
            }
            const unsigned long long V = __ballot(valid);
            const unsigned long long G2 = __ballot(valid && b[0] >= 0x80u), G3 = __ballot(valid && b[0] >= 0xC0u),
                                     G4 = __ballot(valid && b[0] >= 0xE0u), G5 = __ballot(valid && b[0] >= 0xF0u);
            const int lead = __clz((int)~(b[0] << 24));
            const int m = (lead < 4 ? lead : 4) + 1;
            // the true starts of the chunk.  First a run of equal-length values (the run's starts are a periodic pattern:
            // one turn covers the whole chunk of a same-width stream) ...
            unsigned long long S = 0;
            int32_t pp = e;
            if (pp < 64 && ((V >> pp) & 1ull)) {
                const int mr = 1 + (int)((G2 >> pp) & 1ull) + (int)((G3 >> pp) & 1ull) + (int)((G4 >> pp) & 1ull) +
                               (int)((G5 >> pp) & 1ull);
                const unsigned long long Lm = mr == 1 ? V & ~G2 : mr == 2 ? G2 & ~G3 : mr == 3 ? G3 & ~G4 : mr == 4 ? G4 & ~G5 : G5;
                const unsigned long long pat = itf8_pattern(mr);
                const unsigned long long brk = ~(Lm >> pp) & pat;  // first pattern position that does not open such a value
                const int t = brk ? __ffsll((long long)brk) - 1 : 64;
                const unsigned long long take = pat & (t >= 64 ? ~0ull : ((1ull << t) - 1ull));
                S |= take << pp;
                pp += mr * __popcll(take);
            }
            // ... and where widths mix (template sizes, mate positions: a turn per value cost ~4000 cycles a chunk), no walk
            // at all: "bytes of the current value still to come" behind a byte is a function of the same count in front
            // of it (0 -> width - 1, k -> k - 1), five 3-bit entries in a word; a prefix scan composes the functions over
            // the lanes, and a byte opens a value where the count in front of it is 0.
            if (pp < 64 && ((V >> pp) & 1ull)) {
                uint32_t P = (uint32_t)(m - 1) | 0x3440u;  // f(0) = m - 1, f(1..4) = 0..3
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t T = __shfl_up(P, d);  // the lanes l - 2d + 1 .. l - d: applied first
                    uint32_t r = 0;
#pragma unroll
                    for (int q5 = 0; q5 < 5; q5++) r |= ((P >> (3u * ((T >> (3 * q5)) & 7u))) & 7u) << (3 * q5);
                    P = l >= d ? r : P;
                }
                const uint32_t prev = __shfl_up(P, 1);
                const uint32_t before = l == 0 ? (uint32_t)e : (prev >> (3 * e)) & 7u;
                S = __ballot(valid && before == 0u);
                pp = 64 + (int32_t)((__shfl(P, 63) >> (3 * e)) & 7u);
            }
            e = pp >= 64 ? pp - 64 : 0;
            const bool start = (S >> l) & 1ull;
            if (start && q + m > n) bad = true;  // the stream ends inside a value
            const unsigned long long W = ((unsigned long long)b[0] << 32) | (b[1] << 24) | (b[2] << 16) | (b[3] << 8) | b[4];
            const uint32_t v4 = (uint32_t)(W >> (8 * (5 - m))) & ((1u << (m == 5 ? 0 : 7 * m)) - 1u);
            const uint32_t v5 = ((uint32_t)(W >> 8) << 4) | (b[4] & 0x0Fu);
            const uint32_t v = m == 5 ? v5 : v4;
            if (start) {
                out[cnt + (int32_t)mbcnt64(S)] = (int32_t)v;
                acc += (int32_t)v;
            }
            cnt += __popcll(S);
        }
        __syncthreads();
        *reinterpret_cast<uint4*>(&buf[blk & 1][16 * l]) = ahead;
        __syncthreads();
    }
    if (__ballot(bad) != 0 && l == 0) *A.status = CRAMGPU_E_CORRUPT;
    if (k == CRAMGPU_DS_RL || k == CRAMGPU_DS_FN) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d);
        if (l == 0) A.cont_base[(int64_t)c * 3 + (k == CRAMGPU_DS_RL ? 0 : 1)] = acc;
    }
    if (l == 0) A.vals_cnt[slot] = cnt;
}

// R2: offsets of the batch.  cont_base[c] = (bases, features, sequence bytes) in front of container c
__global__ __launch_bounds__(1024) void series_out_layout_kernel(RArgs A) {
    __shared__ int64_t part[3][1024];
    __shared__ int64_t carry[3];
    const int tid = (int)threadIdx.x;
    if (tid < 3) carry[tid] = 0;
    __syncthreads();
    for (int32_t c0 = 0; c0 < A.n_cont; c0 += 1024) {
        const int32_t c = c0 + tid;
        int64_t t[3] = {0, 0, 0};
        if (c < A.n_cont) {
            t[0] = A.cont_base[(int64_t)c * 3 + 0];
            t[1] = A.cont_base[(int64_t)c * 3 + 1];
            const int64_t si = (int64_t)c * NS + CRAMGPU_DS_IN, ss = (int64_t)c * NS + CRAMGPU_DS_SC;
            t[2] = (int64_t)A.len[si] - A.vals_cnt[si] + (int64_t)A.len[ss] - A.vals_cnt[ss];
            if (t[0] < 0 || t[1] < 0) {
                *A.status = CRAMGPU_E_CORRUPT;
                t[0] = t[1] = 0;
            }
        }
        for (int q = 0; q < 3; q++) part[q][tid] = t[q];
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            int64_t v[3];
            for (int q = 0; q < 3; q++) v[q] = tid >= d ? part[q][tid - d] : 0;
            __syncthreads();
            for (int q = 0; q < 3; q++) part[q][tid] += v[q];
            __syncthreads();
        }
        if (c < A.n_cont)
            for (int q = 0; q < 3; q++) A.cont_base[(int64_t)c * 3 + q] = carry[q] + part[q][tid] - t[q];
        __syncthreads();
        if (tid < 3) carry[tid] += part[tid][1023];
        __syncthreads();
    }
    if (tid == 0) {
        for (int q = 0; q < 3; q++) A.cont_base[(int64_t)A.n_cont * 3 + q] = carry[q];
        A.o.base_off[A.n_reads] = carry[0];
        A.f.off[A.n_reads] = carry[1];
        A.o.seq_off[A.n_reads] = carry[2];
        if ((carry[0] > A.o.bases_cap || carry[1] > A.f.cap || carry[2] > A.o.seq_cap) && *A.status == 0)
            *A.status = CRAMGPU_E_INSUFFICIENT;
    }
}

// exclusive scan of NSC rows of 256 values in LDS, a wave per row; totals to tot[]
template <int NSC>
__device__ __forceinline__ void tile_scan(uint32_t (*v)[kTile], uint32_t* tot, int wave, int lane) {
    for (int k = wave; k < NSC; k += kTile / 64) {
        const uint32_t v0 = v[k][4 * lane], v1 = v[k][4 * lane + 1], v2 = v[k][4 * lane + 2], v3 = v[k][4 * lane + 3];
        const uint32_t s = v0 + v1 + v2 + v3;
        uint32_t inc = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(inc, d);
            if (lane >= d) inc += o;
        }
        const uint32_t ex = inc - s;
        v[k][4 * lane] = ex;
        v[k][4 * lane + 1] = ex + v0;
        v[k][4 * lane + 2] = ex + v0 + v1;
        v[k][4 * lane + 3] = ex + v0 + v1 + v2;
        if (lane == 63) tot[k] = inc;
    }
}

// rows of the three scan rounds of R3
enum { S_DET = 0, S_DOWN, S_MAP, S_AP, S_RL, S1_N };                                              // round 1
enum { T_X = 0, T_D, T_N, T_H, T_P, T_I, T_S, T_BA, T_QS, T_SEQ, T_FEAT, S2_N };                  // round 2

__global__ __launch_bounds__(kTile) void series_read_kernel(RArgs A) {
    __shared__ uint32_t sc[S2_N][kTile];
    __shared__ uint32_t tot[S2_N];
    __shared__ long long run1[S1_N];   // running totals of round 1 (AP may be negative: 64-bit signed)
    __shared__ long long run2[S2_N];
    __shared__ int64_t job_src[kTile], job_dst[kTile];
    __shared__ uint32_t job_len[kTile];
    __shared__ uint8_t job_fill[kTile];  // no array in the stream: the scores are filled with 0xFF (offsets may be negative)
    __shared__ int contig;
    __shared__ int corrupt;
    __shared__ int entry_status;
    const int32_t c = (int32_t)blockIdx.x;
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // other workgroups of this launch write the status while this one reads it: one thread reads, all branch on its copy
    // (waves that saw different values would part ways in front of the barriers below)
    if (tid == 0) entry_status = *A.status;
    __syncthreads();
    if (entry_status != 0) return;
    const int32_t first = A.cont_first[c], last = A.cont_first[c + 1];
    const bool multiref = A.c.cont_multiref != nullptr && A.c.cont_multiref[c] != 0;
    const int64_t* voff = A.vals_off + (int64_t)c * NS;
    const int32_t* vcnt = A.vals_cnt + (int64_t)c * NS;
    const int64_t* soff = A.off + (int64_t)c * NS;
    const int32_t* slen = A.len + (int64_t)c * NS;
    if (tid < S1_N) run1[tid] = 0;
    if (tid < S2_N) run2[tid] = 0;
    if (tid == 0) corrupt = 0;
    __syncthreads();
    const int32_t n_rec = last - first;
    // every record has BF, CF, RL, AP, RG, TL (and RI in a multi-reference slice)
    if (vcnt[CRAMGPU_DS_BF] != n_rec || vcnt[CRAMGPU_DS_CF] != n_rec || vcnt[CRAMGPU_DS_RL] != n_rec ||
        vcnt[CRAMGPU_DS_AP] != n_rec || vcnt[CRAMGPU_DS_RG] != n_rec || vcnt[CRAMGPU_DS_TL] != n_rec ||
        (multiref && vcnt[CRAMGPU_DS_RI] != n_rec)) {
        if (tid == 0) *A.status = CRAMGPU_E_CORRUPT;
        return;
    }
    const int64_t base_b = A.cont_base[(int64_t)c * 3 + 0], base_f = A.cont_base[(int64_t)c * 3 + 1],
                  base_s = A.cont_base[(int64_t)c * 3 + 2];
    const int32_t ap_base = A.c.cont_ap_base ? A.c.cont_ap_base[c] : 0;
#define VAL(id, i) (A.vals[voff[id] + (i)])
    for (int32_t t0 = first; t0 < last; t0 += kTile) {
        const int32_t r = t0 + tid, k = r - first;
        const bool live = r < last;
        uint32_t bf = 0, cf = 0, L = 0;
        int32_t ap = 0;
        if (live) {
            bf = (uint32_t)VAL(CRAMGPU_DS_BF, k);
            cf = (uint32_t)VAL(CRAMGPU_DS_CF, k);
            L = (uint32_t)VAL(CRAMGPU_DS_RL, k);
            ap = VAL(CRAMGPU_DS_AP, k);
        }
        const bool det = live && (cf & CRAMGPU_CF_DETACHED);
        const bool down = live && !det && (cf & CRAMGPU_CF_HAS_MATE_DOWNSTREAM);
        const bool mapped = live && !(bf & CRAMGPU_BF_SEGMENT_UNMAPPED);
        // round 1 (AP deltas are summed as two's complement 32-bit words: the carry-free low word is all that is used)
        sc[S_DET][tid] = det;
        sc[S_DOWN][tid] = down;
        sc[S_MAP][tid] = mapped;
        sc[S_AP][tid] = (uint32_t)ap;
        sc[S_RL][tid] = L;
        if (tid == 0) contig = 1;
        __syncthreads();
        tile_scan<S1_N>(sc, tot, wave, lane);
        __syncthreads();
        const int64_t i_det = run1[S_DET] + sc[S_DET][tid], i_down = run1[S_DOWN] + sc[S_DOWN][tid],
                      i_map = run1[S_MAP] + sc[S_MAP][tid];
        const int32_t start = A.c.ap_delta ? (int32_t)((uint32_t)ap_base + (uint32_t)run1[S_AP] + sc[S_AP][tid] + (uint32_t)ap) : ap;
        const int64_t boff = base_b + run1[S_RL] + sc[S_RL][tid];
        __syncthreads();
        if (tid < S1_N) run1[tid] += tot[tid];
        // counts of the record's features by the series their payload comes from
        uint32_t nfeat = 0, nx = 0, nd = 0, nn = 0, nh = 0, np = 0, ni = 0, ns = 0, nba = 0, nqs = 0;
        bool bad = false;
        int64_t f_loc = 0;  // container-local index of the record's first feature: needs the scan of nfeat first
        if (mapped) {
            if (i_map < vcnt[CRAMGPU_DS_FN] && i_map < vcnt[CRAMGPU_DS_MQ])
                nfeat = (uint32_t)VAL(CRAMGPU_DS_FN, i_map);
            else
                bad = true;
        }
        sc[T_FEAT][tid] = nfeat;
        __syncthreads();
        tile_scan<1>(sc + T_FEAT, tot + T_FEAT, wave, lane);
        __syncthreads();
        f_loc = run2[T_FEAT] + sc[T_FEAT][tid];
        const uint8_t* fc = A.arena + soff[CRAMGPU_DS_FC];
        if (mapped && !bad) {
            if (f_loc + nfeat > slen[CRAMGPU_DS_FC] || f_loc + nfeat > vcnt[CRAMGPU_DS_FP]) {
                bad = true;
                nfeat = 0;
            }
            for (uint32_t j = 0; j < nfeat; j++) {
                switch (fc[f_loc + j]) {
                    case 'B': nba++, nqs++; break;
                    case 'X': nx++; break;
                    case 'I': ni++; break;
                    case 'S': ns++; break;
                    case 'H': nh++; break;
                    case 'P': np++; break;
                    case 'D': nd++; break;
                    case 'N': nn++; break;
                    case 'i': nba++; break;
                    case 'Q': nqs++; break;
                    default: bad = true;
                }
            }
        }
        const uint32_t ba_arr = (live && !mapped && !(cf & CRAMGPU_CF_UNKNOWN_BASES)) ? L : 0u;
        const uint32_t qs_arr = (live && (cf & CRAMGPU_CF_FORCE_PRESERVE_QS)) ? L : 0u;
        __syncthreads();
        if (tid == 0) run2[T_FEAT] += tot[T_FEAT];
        sc[T_X][tid] = nx;
        sc[T_D][tid] = nd;
        sc[T_N][tid] = nn;
        sc[T_H][tid] = nh;
        sc[T_P][tid] = np;
        sc[T_I][tid] = ni;
        sc[T_S][tid] = ns;
        sc[T_BA][tid] = nba + ba_arr;
        sc[T_QS][tid] = nqs + qs_arr;
        __syncthreads();
        tile_scan<T_SEQ>(sc, tot, wave, lane);   // rows T_X .. T_QS
        __syncthreads();
        int64_t cx = run2[T_X] + sc[T_X][tid], cd = run2[T_D] + sc[T_D][tid], cn = run2[T_N] + sc[T_N][tid],
                ch = run2[T_H] + sc[T_H][tid], cp = run2[T_P] + sc[T_P][tid], ci = run2[T_I] + sc[T_I][tid],
                cs = run2[T_S] + sc[T_S][tid], cba = run2[T_BA] + sc[T_BA][tid], cqs = run2[T_QS] + sc[T_QS][tid];
        const int64_t cqs0 = cqs;
        // cursors must stay inside what the streams hold
        if (live && (cx + nx > slen[CRAMGPU_DS_BS] || cd + nd > vcnt[CRAMGPU_DS_DL] || cn + nn > vcnt[CRAMGPU_DS_RS] ||
                     ch + nh > vcnt[CRAMGPU_DS_HC] || cp + np > vcnt[CRAMGPU_DS_PD] || ci + ni > vcnt[CRAMGPU_DS_IN] ||
                     cs + ns > vcnt[CRAMGPU_DS_SC] || cba + nba + ba_arr > slen[CRAMGPU_DS_BA] ||
                     cqs + nqs + qs_arr > slen[CRAMGPU_DS_QS] || (det && i_det >= vcnt[CRAMGPU_DS_MF]) ||
                     (det && (i_det >= vcnt[CRAMGPU_DS_NS] || i_det >= vcnt[CRAMGPU_DS_NP] || i_det >= vcnt[CRAMGPU_DS_TS])) ||
                     (down && i_down >= vcnt[CRAMGPU_DS_NF])))
            bad = true;
        // the record's name (ReaderToBAM.java:86-93): every record's when names are captured, a detached record's otherwise;
        // a container written without names has an empty RN stream and none of its records carries one
        const bool rn_any = A.c.capture_names || vcnt[CRAMGPU_DS_RN] != 0;
        const bool named = live && rn_any && (A.c.capture_names || det);
        const int64_t i_name = A.c.capture_names ? (int64_t)k : i_det;
        if (named && i_name >= vcnt[CRAMGPU_DS_RN]) bad = true;
        // sequence bytes of the record's I / S features
        uint32_t nseq = 0;
        if (live && !bad) {
            for (uint32_t j = 0; j < ni; j++) nseq += (uint32_t)(VAL(CRAMGPU_DS_IN, ci + j + 1) - VAL(CRAMGPU_DS_IN, ci + j) - 1);
            for (uint32_t j = 0; j < ns; j++) nseq += (uint32_t)(VAL(CRAMGPU_DS_SC, cs + j + 1) - VAL(CRAMGPU_DS_SC, cs + j) - 1);
        }
        __syncthreads();
        if (tid < T_SEQ) run2[tid] += tot[tid];
        sc[T_SEQ][tid] = nseq;
        __syncthreads();
        tile_scan<1>(sc + T_SEQ, tot + T_SEQ, wave, lane);
        __syncthreads();
        const int64_t seq0 = base_s + run2[T_SEQ] + sc[T_SEQ][tid];
        if (bad) corrupt = 1;
        uint32_t qlen = 0;
        if (live) {
            A.c.bam_flags[r] = (int32_t)bf;
            A.c.comp_flags[r] = (int32_t)cf;
            if (A.c.ref_id) A.c.ref_id[r] = multiref ? VAL(CRAMGPU_DS_RI, k) : 0;
            A.o.read_len[r] = (int32_t)L;
            A.o.aln_start[r] = start;
            A.c.read_group[r] = VAL(CRAMGPU_DS_RG, k);
            A.c.tag_list[r] = VAL(CRAMGPU_DS_TL, k);
            const bool ok = !bad;
            A.c.mate_flags[r] = det && ok ? VAL(CRAMGPU_DS_MF, i_det) : 0;
            A.c.mate_ref[r] = det && ok ? VAL(CRAMGPU_DS_NS, i_det) : 0;
            A.c.mate_start[r] = det && ok ? VAL(CRAMGPU_DS_NP, i_det) : 0;
            A.c.tlen[r] = det && ok ? VAL(CRAMGPU_DS_TS, i_det) : 0;
            A.c.next_frag[r] = down && ok ? VAL(CRAMGPU_DS_NF, i_down) : 0;
            A.o.mapq[r] = mapped && ok ? (uint8_t)VAL(CRAMGPU_DS_MQ, i_map) : (uint8_t)0;  // NO_MAPPING_QUALITY
            A.o.base_off[r] = boff;
            if (A.o.rflags)
                A.o.rflags[r] = (uint8_t)(((bf & CRAMGPU_BF_SEGMENT_UNMAPPED) ? CRAMGPU_RF_UNMAPPED : 0) |
                                          ((cf & CRAMGPU_CF_UNKNOWN_BASES) ? CRAMGPU_RF_UNKNOWN_BASES : 0) |
                                          ((cf & CRAMGPU_CF_FORCE_PRESERVE_QS) ? CRAMGPU_RF_FORCE_QS : 0));
            A.o.seq_off[r] = seq0;
            if (A.o.name_off != nullptr && A.o.name_len != nullptr) {
                const bool has = named && ok;
                const int32_t s0 = has ? VAL(CRAMGPU_DS_RN, i_name) : 0, s1 = has ? VAL(CRAMGPU_DS_RN, i_name + 1) : 0;
                A.o.name_off[r] = has ? soff[CRAMGPU_DS_RN] + s0 : 0;
                A.o.name_len[r] = has ? s1 - s0 - 1 : -1;
            }
            A.f.off[r] = base_f + f_loc;
            if (ok && boff + L <= A.o.bases_cap) {
                if (mapped && base_f + f_loc + nfeat <= A.f.cap && seq0 + nseq <= A.o.seq_cap) {
                    const uint8_t* bs = A.arena + soff[CRAMGPU_DS_BS];
                    const uint8_t* ba = A.arena + soff[CRAMGPU_DS_BA];
                    const uint8_t* qs = A.arena + soff[CRAMGPU_DS_QS];
                    int32_t pos = 0;
                    int64_t sq = seq0;
                    for (uint32_t j = 0; j < nfeat; j++) {
                        const int64_t g = base_f + f_loc + j;
                        const uint8_t o = fc[f_loc + j];
                        pos += VAL(CRAMGPU_DS_FP, f_loc + j);
                        int32_t a = 0, b = 0;
                        switch (o) {
                            case 'B':
                                a = ba[cba++];
                                b = qs[cqs++];
                                break;
                            case 'X': a = (int32_t)bs[cx++] << 16; break;
                            case 'I':
                            case 'S': {
                                const int id = o == 'I' ? CRAMGPU_DS_IN : CRAMGPU_DS_SC;
                                int64_t& cur = o == 'I' ? ci : cs;
                                const int32_t s0 = VAL(id, cur), s1 = VAL(id, cur + 1);
                                cur++;
                                const uint8_t* src = A.arena + soff[id] + s0;
                                b = s1 - s0 - 1;
                                a = (int32_t)(sq - seq0);
                                for (int32_t i = 0; i < b; i++) A.o.seq[sq + i] = src[i];
                                sq += b;
                                break;
                            }
                            case 'H': a = VAL(CRAMGPU_DS_HC, ch++); break;
                            case 'P': a = VAL(CRAMGPU_DS_PD, cp++); break;
                            case 'D': a = VAL(CRAMGPU_DS_DL, cd++); break;
                            case 'N': a = VAL(CRAMGPU_DS_RS, cn++); break;
                            case 'i': a = ba[cba++]; break;
                            case 'Q': a = qs[cqs++]; break;
                            default: break;
                        }
                        A.f.op[g] = o;
                        A.f.pos[g] = pos;
                        A.f.a[g] = a;
                        A.f.b[g] = b;
                    }
                } else if (ba_arr) {
                    const uint8_t* src = A.arena + soff[CRAMGPU_DS_BA] + cba;
                    for (uint32_t i = 0; i < L; i++) A.o.bases[boff + i] = src[i];
                }
                qlen = L;  // the quality array of the record: copied (force-preserved) or filled with 0xFF by the group
                job_src[tid] = soff[CRAMGPU_DS_QS] + cqs0 + nqs;  // behind the scores of the record's B / Q features
                job_fill[tid] = qs_arr ? 0 : 1;
                job_dst[tid] = boff;
            }
        }
        job_len[tid] = qlen;
        __syncthreads();
        if (tid < 1) run2[T_SEQ] += tot[T_SEQ];
        if (live && (qlen == 0 || job_fill[tid] ||
                     (tid > 0 && (job_len[tid - 1] == 0 || job_src[tid] != job_src[tid - 1] + job_len[tid - 1] ||
                                  job_dst[tid] != job_dst[tid - 1] + job_len[tid - 1]))))
            contig = 0;
        __syncthreads();
        const int32_t cnt = last - t0 < kTile ? last - t0 : kTile;
        if (contig) {
            wg_copy(A.o.quals + job_dst[0], A.arena + job_src[0], job_src[cnt - 1] + job_len[cnt - 1] - job_src[0], tid);
        } else {
            for (int j = wave; j < cnt; j += kTile / 64) {
                const uint32_t n = job_len[j];
                uint8_t* d = A.o.quals + job_dst[j];
                if (!job_fill[j]) {
                    const uint8_t* s = A.arena + job_src[j];
                    for (uint32_t i = lane; i < n; i += 64) d[i] = s[i];
                } else {
                    for (uint32_t i = lane; i < n; i += 64) d[i] = 0xFF;
                }
            }
        }
        __syncthreads();
    }
#undef VAL
    if (tid == 0) {
        // every optional value must have found its record
        if (corrupt || run1[S_DET] != vcnt[CRAMGPU_DS_MF] || run1[S_DOWN] != vcnt[CRAMGPU_DS_NF] ||
            run1[S_DET] != vcnt[CRAMGPU_DS_NS] || run1[S_DET] != vcnt[CRAMGPU_DS_NP] || run1[S_DET] != vcnt[CRAMGPU_DS_TS] ||
            run1[S_MAP] != vcnt[CRAMGPU_DS_FN] || run1[S_MAP] != vcnt[CRAMGPU_DS_MQ] || run2[T_FEAT] != slen[CRAMGPU_DS_FC] ||
            run2[T_FEAT] != vcnt[CRAMGPU_DS_FP] || run2[T_BA] != slen[CRAMGPU_DS_BA] || run2[T_QS] != slen[CRAMGPU_DS_QS] ||
            run2[T_X] != slen[CRAMGPU_DS_BS] || run2[T_D] != vcnt[CRAMGPU_DS_DL] || run2[T_N] != vcnt[CRAMGPU_DS_RS] ||
            run2[T_H] != vcnt[CRAMGPU_DS_HC] || run2[T_P] != vcnt[CRAMGPU_DS_PD] || run2[T_I] != vcnt[CRAMGPU_DS_IN] ||
            run2[T_S] != vcnt[CRAMGPU_DS_SC] ||
            ((A.c.capture_names || vcnt[CRAMGPU_DS_RN] != 0) && (A.c.capture_names ? (long long)n_rec : run1[S_DET]) != vcnt[CRAMGPU_DS_RN]))
            *A.status = CRAMGPU_E_CORRUPT;
    }
}

}  // namespace

hipError_t launch_write_series(const cramgpu_reads& r, const cramgpu_record_cols& c, const cramgpu_features& f, uint8_t* arena,
                               int64_t cap, int64_t* d_off, int32_t* d_len, int32_t* d_status, uint8_t* d_alias, hipStream_t s) {
    WArgs A{r, c, f, arena, cap, d_off, d_len, d_status, d_alias};
    hipError_t e = hipMemsetAsync(d_status, 0, sizeof(int32_t), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(series_count_kernel, dim3(r.n_cont), dim3(kTile), 0, s, A);
    hipLaunchKernelGGL(series_layout_kernel, dim3(1), dim3(1024), 0, s, A, r.n_cont);
    hipLaunchKernelGGL(series_write_kernel, dim3(r.n_cont), dim3(kTile), 0, s, A);
    return hipGetLastError();
}

size_t read_series_ws_ints(int64_t series_bytes, int32_t n_cont) { return (size_t)(series_bytes + (int64_t)n_cont * CRAMGPU_N_SERIES * 4 + 64); }

hipError_t launch_read_series(const uint8_t* arena, const int64_t* d_off, const int32_t* d_len, int32_t n_cont, int32_t n_reads,
                              const int32_t* d_cont_first, const cramgpu_record_cols& c, const cramgpu_read_out& o,
                              const cramgpu_features& f, int32_t* ws_vals, int64_t ws_vals_cap, int64_t* ws_off, int32_t* ws_cnt,
                              int64_t* ws_base, int32_t* d_status, hipStream_t s) {
    RArgs A{arena, d_off, d_len, n_cont, n_reads, d_cont_first, c, o, f, ws_vals, ws_off, ws_cnt, ws_base, d_status, ws_vals_cap};
    hipError_t e = hipMemsetAsync(d_status, 0, sizeof(int32_t), s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(ws_base, 0, (size_t)(n_cont + 1) * 3 * sizeof(int64_t), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(series_ws_layout_kernel, dim3(1), dim3(1024), 0, s, A);
    hipLaunchKernelGGL(series_parse_kernel, dim3(NS, n_cont), dim3(64), 0, s, A);
    hipLaunchKernelGGL(series_out_layout_kernel, dim3(1), dim3(1024), 0, s, A);
    hipLaunchKernelGGL(series_read_kernel, dim3(n_cont), dim3(kTile), 0, s, A);
    return hipGetLastError();
}
