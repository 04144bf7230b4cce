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

struct WArgs {
    cramgpu_reads r;
    cramgpu_record_cols c;
    cramgpu_features f;
    uint8_t* arena;
    int64_t cap;
    int64_t* off;
    int32_t* len;
    int32_t* status;
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
    const int32_t c = (int32_t)blockIdx.x;
    const int tid = (int)threadIdx.x;
    if (tid < NS) tot[tid] = 0;
    __syncthreads();
    const int32_t first = cont_first(A.r, c), last = cont_first(A.r, c + 1);
    const bool multiref = A.c.cont_multiref != nullptr && A.c.cont_multiref[c] != 0;
    Sizes acc;
#pragma unroll
    for (int k = 0; k < NS; k++) acc.v[k] = 0;
    int bad = 0;
    for (int32_t r = first + tid; r < last; r += kTile) {
        Sizes z;
        record_sizes(A, r, multiref, start_before(A, r, first, c), z, bad);
#pragma unroll
        for (int k = 0; k < NS; k++) acc.v[k] += z.v[k];
    }
#pragma unroll
    for (int k = 0; k < NS; k++)
        if (acc.v[k]) atomicAdd(&tot[k], acc.v[k]);
    if (bad) *A.status = CRAMGPU_E_ARG;
    __syncthreads();
    if (tid < NS) A.len[(int64_t)c * NS + tid] = (int32_t)tot[tid];
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
            for (int k = 0; k < NS; k++) t += ((int64_t)A.len[(int64_t)c * NS + k] + 15) & ~(int64_t)15;
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
            if (cf & CRAMGPU_CF_FORCE_PRESERVE_QS) {  // the quality array: copied by the group below
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
        if (contig) {
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

}  // namespace

hipError_t launch_write_series(const cramgpu_reads& r, const cramgpu_record_cols& c, const cramgpu_features& f, uint8_t* arena,
                               int64_t cap, int64_t* d_off, int32_t* d_len, int32_t* d_status, hipStream_t s) {
    WArgs A{r, c, f, arena, cap, d_off, d_len, d_status};
    hipError_t e = hipMemsetAsync(d_status, 0, sizeof(int32_t), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(series_count_kernel, dim3(r.n_cont), dim3(kTile), 0, s, A);
    hipLaunchKernelGGL(series_layout_kernel, dim3(1), dim3(1024), 0, s, A, r.n_cont);
    hipLaunchKernelGGL(series_write_kernel, dim3(r.n_cont), dim3(kTile), 0, s, A);
    return hipGetLastError();
}
