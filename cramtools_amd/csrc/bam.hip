// A — container assembly and BAM records for gfx950.
//
// Reference: CIGAR from read features — htsjdk!build/Cram2SamRecordFactory#getCigar2 L119-225 (bytecode only), the same
// algorithm in source in src/main/java/htsjdk/samtools/cram/encoding/reader/ReadFeatureBuffer.java:180-294; mate
// restoration — src/main/java/htsjdk/samtools/cram/build/CramNormalizer.java:82-105, :255-285, :455-470; the BAM record
// layout — src/main/java/htsjdk/samtools/cram/encoding/reader/BAMRecordView.java:23-35, :106-119, :150-185, :229-241,
// :360-415, driven as in ReaderToBAM.java:64-176 with the values of htsjdk!build/Cram2SamRecordFactory#create; stripes
// of records — src/main/java/htsjdk/samtools/cram/paralell/Bam_OBA_Supplier.java:45-82, CramToBam_OBA_Function.java:64-75.
//
// Shape on the machine: the per-record decisions (CIGAR walk, mate chains, fixed fields) are a thread per record; the
// bytes of a record (4-bit bases, scores, names, tags — a BAM record is ~280 bytes for a 150-base read) move with 16
// lanes per record, 16 bases per lane and step, so that a wave writes four records' contiguous runs.  Offsets into the
// stripe are prefix sums over record sizes (the tiled scan of the feature kernels).  HBM-bound byte shuffling:
// algorithmic bytes = columns + bases + scores read once, records written once (and the reverse for the decoder).
#include "internal.h"

namespace {

enum { OP_M = 0, OP_I = 1, OP_D = 2, OP_N = 3, OP_S = 4, OP_H = 5, OP_P = 6 };
enum { BF_PAIRED = 1, BF_UNMAPPED = 4, BF_MATE_UNMAPPED = 8, BF_REVERSE = 16, BF_MATE_REVERSE = 32 };
enum { MF_NEG_STRAND = 1, MF_UNMAPPED = 2 };

struct __attribute__((packed)) U32u { uint32_t v; };
struct __attribute__((packed)) U32x2u { uint32_t v[2]; };
struct __attribute__((packed)) U32x4u { uint32_t v[4]; };

// ---------------------------------------------------------------------------------------------------------------
// CIGAR from features (ReadFeatureBuffer.java:180-294), a thread per record; WRITE=false counts.
// ---------------------------------------------------------------------------------------------------------------
struct CigarOut {
    uint32_t* out;  // nullptr: count only
    int32_t n;
    __device__ __forceinline__ void emit(int32_t len, int op) {
        if (out) out[n] = ((uint32_t)len << 4) | (uint32_t)op;
        n++;
    }
};

__device__ __forceinline__ int feature_cigar_op(uint8_t op, int32_t a, int32_t b, int32_t& len) {  // :213-257
    switch (op) {
        case 'I': len = b; return OP_I;
        case 'S': len = b; return OP_S;
        case 'H': len = a; return OP_H;
        case 'i': len = 1; return OP_I;
        case 'D': len = a; return OP_D;
        case 'N': len = a; return OP_N;
        case 'P': len = a; return OP_P;
        case 'X':
        case 'B': len = 1; return OP_M;
        default: return -1;  // Q and the array operators carry no CIGAR element
    }
}

__device__ int32_t cigar_walk(int32_t read_len, const cramgpu_features& F, int64_t f0, int64_t f1, uint32_t* out) {
    CigarOut E{out, 0};
    if (f1 == f0) {  // :182-185
        E.emit(read_len, OP_M);
        return E.n;
    }
    int last_op = OP_M;
    int32_t last_len = 0, last_pos = 1;
    for (int64_t k = f0; k < f1; k++) {
        const int32_t rf_pos = F.pos[k];
        const int32_t gap = rf_pos - (last_pos + last_len);  // :199-211
        if (gap > 0) {
            if (last_op != OP_M) {
                E.emit(last_len, last_op);
                last_pos += last_len;
                last_len = gap;
            } else {
                last_len += gap;
            }
            last_op = OP_M;
        }
        int32_t rf_len = 0;
        const int co = feature_cigar_op(F.op[k], F.a[k], F.b[k], rf_len);
        if (co < 0) continue;
        if (last_op != co) {  // :259-269
            if (last_len > 0) E.emit(last_len, last_op);
            last_op = co;
            last_len = rf_len;
            last_pos = rf_pos;
        } else {
            last_len += rf_len;
        }
        if (!(co == OP_M || co == OP_I || co == OP_S)) last_pos -= rf_len;  // :271-272
    }
    if (last_op != OP_M) {  // :275-286
        E.emit(last_len, last_op);
        if (read_len >= last_pos + last_len) E.emit(read_len - (last_len + last_pos) + 1, OP_M);
    } else if (read_len > last_pos - 1) {
        E.emit(read_len - last_pos + 1, OP_M);
    }
    if (E.n == 0) E.emit(read_len, OP_M);  // :288-291
    return E.n;
}

template <bool WRITE>
__global__ __launch_bounds__(256) void cigar_kernel(int32_t n, const int32_t* read_len, const int32_t* aln_start,
                                                    const int32_t* bam_flags, cramgpu_features F, int32_t* n_cigar,
                                                    const int64_t* cigar_off, uint32_t* cigar, int64_t cap, int32_t* aln_end,
                                                    int32_t* status) {
    const int32_t r = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    if (WRITE) {
        const bool fits = cigar_off[n] <= cap;
        if (r == 0) status[0] = fits ? 0 : CRAMGPU_E_INSUFFICIENT;
        if (!fits) return;
    }
    if (r >= n) return;
    const int64_t f0 = F.off[r], f1 = F.off[r + 1];
    if (bam_flags[r] & BF_UNMAPPED) {  // SAMRecord.NO_ALIGNMENT_CIGAR
        if (!WRITE) n_cigar[r] = 0, aln_end[r] = 0;
        return;
    }
    if (WRITE) {
        cigar_walk(read_len[r], F, f0, f1, cigar + cigar_off[r]);
        return;
    }
    n_cigar[r] = cigar_walk(read_len[r], F, f0, f1, nullptr);
    int32_t span = read_len[r];  // CramCompressionRecord#getAlignmentEnd
    for (int64_t k = f0; k < f1; k++) {
        const uint8_t o = F.op[k];
        if (o == 'i')
            span--;
        else if (o == 'I' || o == 'S')
            span -= F.b[k];
        else if (o == 'D')
            span += F.a[k];
    }
    aln_end[r] = aln_start[r] + span - 1;
}

// ---------------------------------------------------------------------------------------------------------------
// Mate restoration (CramNormalizer.java:82-105, 255-285, 455-470)
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int32_t cont_of(const int32_t* cont_first, int32_t n_cont, int32_t r) {  // largest c: first[c] <= r
    int32_t lo = 0, hi = n_cont - 1;
    while (lo < hi) {
        const int32_t mid = (lo + hi + 1) >> 1;
        if (cont_first[mid] <= r)
            lo = mid;
        else
            hi = mid - 1;
    }
    return lo;
}

// link[r] = record the NF distance of r points at, or -1; nprev[j] = records that point at j; serial[c] = a record of
// container c has more than one (the reference then lets the LAST chain that reaches it win: done by one thread)
__global__ __launch_bounds__(256) void mates_link_kernel(int32_t n, int32_t n_cont, const int32_t* cont_first,
                                                         cramgpu_record_cols C, const int32_t* cont_ref_id, int32_t* link,
                                                         uint32_t* nprev, uint8_t* serial, int32_t* status) {
    const int32_t r = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    if (r >= n) return;
    const int32_t c = cont_of(cont_first, n_cont, r);
    if (!(C.cont_multiref && C.cont_multiref[c])) C.ref_id[r] = cont_ref_id[c];
    const int32_t bf = C.bam_flags[r], cf = C.comp_flags[r];
    int32_t to = -1;
    if (!(bf & BF_PAIRED) || (cf & CRAMGPU_CF_DETACHED)) {
        C.next_frag[r] = -1;  // :85
    } else if (cf & CRAMGPU_CF_HAS_MATE_DOWNSTREAM) {
        const int64_t j = (int64_t)r + 1 + C.next_frag[r];
        if (C.next_frag[r] < 0 || j >= cont_first[c + 1]) {
            status[c] = CRAMGPU_E_JAVA_THROWS;  // records.get(...) throws; a negative distance links a record to itself or
                                                // backwards, where restoreMateInfo's walk does not end: refused
        } else {
            to = (int32_t)j;
            if (atomicAdd(&nprev[j], 1u) != 0u) serial[c] = 1;
        }
    }
    link[r] = to;
}

struct MateCols {
    const int32_t* bam_flags;
    const int32_t* ref_id;
    const int32_t* aln_start;
    const int32_t* aln_end;
    int32_t* mate_flags;
    int32_t* mate_ref;
    int32_t* mate_start;
    int32_t* tlen;
};
__device__ __forceinline__ void set_next_mate(const MateCols& M, int32_t r, int32_t nx) {  // :278-285
    const int32_t bf = M.bam_flags[nx];
    int32_t ms = M.aln_start[nx];
    const int32_t mr = M.ref_id[nx];
    if (mr == -1) ms = 0;
    M.mate_start[r] = ms;
    M.mate_ref[r] = mr;
    M.mate_flags[r] = (M.mate_flags[r] & ~(MF_UNMAPPED | MF_NEG_STRAND)) | ((bf & BF_UNMAPPED) ? MF_UNMAPPED : 0) |
                      ((bf & BF_REVERSE) ? MF_NEG_STRAND : 0);
}
__device__ void restore_chain(const MateCols& M, const int32_t* link, int32_t head) {  // :255-276
    int32_t cur = head;
    while (link[cur] >= 0) {
        set_next_mate(M, cur, link[cur]);
        cur = link[cur];
    }
    set_next_mate(M, cur, head);
    int32_t t = 0;  // :455-470
    const int32_t b1 = M.bam_flags[head], b2 = M.bam_flags[cur];
    if (!(b1 & BF_UNMAPPED) && !(b2 & BF_UNMAPPED) && M.ref_id[head] == M.ref_id[cur]) {
        const int32_t p1 = (b1 & BF_REVERSE) ? M.aln_end[head] : M.aln_start[head];
        const int32_t p2 = (b2 & BF_REVERSE) ? M.aln_end[cur] : M.aln_start[cur];
        t = p2 - p1 + (p2 >= p1 ? 1 : -1);
    }
    M.tlen[head] = t;
    M.tlen[cur] = -t;
}
__device__ __forceinline__ bool linked(const cramgpu_record_cols& C, int32_t r) {
    return (C.bam_flags[r] & BF_PAIRED) && !(C.comp_flags[r] & CRAMGPU_CF_DETACHED);
}

__global__ __launch_bounds__(256) void mates_walk_kernel(int32_t n, int32_t n_cont, const int32_t* cont_first,
                                                         cramgpu_record_cols C, const int32_t* aln_start, const int32_t* aln_end,
                                                         const int32_t* link, const uint32_t* nprev, const uint8_t* serial,
                                                         const int32_t* status) {
    const int32_t r = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    if (r >= n) return;
    const int32_t c = cont_of(cont_first, n_cont, r);
    if (status[c] != 0) return;
    const MateCols M{C.bam_flags, C.ref_id, aln_start, aln_end, C.mate_flags, C.mate_ref, C.mate_start, C.tlen};
    if (serial[c]) {  // chains that share records: in record order, as the reference's loop
        if (r != cont_first[c]) return;
        for (int32_t i = cont_first[c]; i < cont_first[c + 1]; i++)
            if (link[i] >= 0 && !(nprev[i] != 0u && linked(C, i))) restore_chain(M, link, i);
        return;
    }
    // head of a chain: points at a record and is pointed at by none (a record that is not multi-fragment or is
    // detached forgets who pointed at it: :84-89 runs after the pointer was set)
    if (link[r] >= 0 && !(nprev[r] != 0u && linked(C, r))) restore_chain(M, link, r);
}

// ---------------------------------------------------------------------------------------------------------------
// BAM records out
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int32_t reg2bin(int64_t beg, int64_t end) {  // SAM specification §5.3
    --end;
    if (beg >> 14 == end >> 14) return (int32_t)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int32_t)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int32_t)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int32_t)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int32_t)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}
__device__ __forceinline__ int32_t dec_digits(long long v) {
    int32_t k = 1;
    while (v >= 10) v /= 10, k++;
    return k;
}

__global__ __launch_bounds__(256) void bam_size_kernel(int32_t n, cramgpu_bam_cols B, int32_t* sizes) {
    const int32_t r = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    if (r >= n) return;
    const int64_t ln = B.names ? B.name_off[r + 1] - B.name_off[r] : dec_digits(B.name_base + r);
    const int64_t lt = B.tags ? B.tag_off[r + 1] - B.tag_off[r] : 0;
    // a negative length (a corrupt RL value upstream) is laid out as 0 and reported by the write pass
    const int64_t L = B.read_len[r] > 0 ? B.read_len[r] : 0, nc = B.n_cigar[r] > 0 ? B.n_cigar[r] : 0;
    sizes[r] = (int32_t)(36 + (ln > 0 ? ln : 0) + 1 + 4 * nc + (L + 1) / 2 + L + (lt > 0 ? lt : 0));
}

// BAMRecordView.java:360-415: code of a base, 0xFF = IllegalArgumentException
__device__ __forceinline__ uint8_t nybble_of(uint32_t c) {
    switch (c) {
        case '=': return 0;
        case 'A': case 'a': return 1;
        case 'C': case 'c': return 2;
        case 'M': case 'm': return 3;
        case 'G': case 'g': return 4;
        case 'R': case 'r': return 5;
        case 'S': case 's': return 6;
        case 'V': case 'v': return 7;
        case 'T': case 't': return 8;
        case 'W': case 'w': return 9;
        case 'Y': case 'y': return 10;
        case 'H': case 'h': return 11;
        case 'K': case 'k': return 12;
        case 'D': case 'd': return 13;
        case 'B': case 'b': return 14;
        case 'N': case 'n': case '.': return 15;
        default: return 0xFF;
    }
}

constexpr int kSub = 16;  // lanes per record

__global__ __launch_bounds__(256) void bam_write_kernel(int32_t n, cramgpu_bam_cols B, uint8_t* bam, int64_t cap,
                                                        const int64_t* rec_off, int32_t* status) {
    __shared__ uint8_t lut[256];
    lut[threadIdx.x] = nybble_of(threadIdx.x);
    __syncthreads();
    const int32_t r = (int32_t)(blockIdx.x * (blockDim.x / kSub) + threadIdx.x / kSub);
    const int sl = (int)(threadIdx.x % kSub);
    if (r >= n) return;
    if (rec_off[n] > cap) {
        if (sl == 0) status[r] = CRAMGPU_E_INSUFFICIENT;
        return;
    }
    uint8_t* p = bam + rec_off[r];
    const int32_t size = (int32_t)(rec_off[r + 1] - rec_off[r]);
    const int32_t ln_raw = B.names ? (int32_t)(B.name_off[r + 1] - B.name_off[r]) : dec_digits(B.name_base + r);
    const int32_t lt_raw = B.tags ? (int32_t)(B.tag_off[r + 1] - B.tag_off[r]) : 0;
    const int32_t ln = ln_raw > 0 ? ln_raw : 0, lt = lt_raw > 0 ? lt_raw : 0;
    bool bad = B.read_len[r] < 0 || B.n_cigar[r] < 0 || ln_raw < 0 || lt_raw < 0;  // laid out as empty (bam_size_kernel), reported
    const int32_t L = B.read_len[r] > 0 ? B.read_len[r] : 0, nc = B.n_cigar[r] > 0 ? B.n_cigar[r] : 0;
    bad = bad || ln + 1 > 255 || nc > 65535;
    if (sl == 0) {  // the fixed part: BAMRecordView.java:23-35
        const int32_t bf = B.bam_flags[r];
        const int32_t rid = B.ref_id[r];
        const bool placed = rid != -1;
        const int32_t start = placed ? B.aln_start[r] : 0;
        int32_t flags = bf & ~(BF_MATE_UNMAPPED | BF_MATE_REVERSE);
        int32_t mref = -1, mstart = 0;
        if (bf & BF_PAIRED) {
            mref = B.mate_ref[r];
            mstart = B.mate_start[r] > 0 ? B.mate_start[r] : 0;
            const int32_t mf = B.mate_flags[r];
            if (mf & MF_NEG_STRAND) flags |= BF_MATE_REVERSE;
            if (mf & MF_UNMAPPED) flags |= BF_MATE_UNMAPPED;
        }
        int32_t bin = 0;
        if (placed) {
            const int64_t beg = (int64_t)start - 1;
            int64_t end = (bf & BF_UNMAPPED) ? 0 : B.aln_end[r];
            if (end <= 0 || end <= beg) end = beg + 1;
            bin = reg2bin(beg, end);
        }
        U32u* w = reinterpret_cast<U32u*>(p);
        w[0].v = (uint32_t)(size - 4);
        w[1].v = (uint32_t)rid;
        w[2].v = (uint32_t)(start - 1);
        w[3].v = (uint32_t)((ln + 1) & 255) | (uint32_t)(placed ? B.mapq[r] : 0) << 8 | (uint32_t)(bin & 0xFFFF) << 16;
        w[4].v = (uint32_t)(nc & 0xFFFF) | (uint32_t)(flags & 0xFFFF) << 16;
        w[5].v = (uint32_t)L;
        w[6].v = (uint32_t)mref;
        w[7].v = (uint32_t)(mstart - 1);
        w[8].v = (uint32_t)B.tlen[r];
        p[36 + ln] = 0;
    }
    if (B.names) {
        const uint8_t* nm = B.names + B.name_off[r];
        for (int32_t i = sl; i < ln; i += kSub) p[36 + i] = nm[i];
    } else {
        long long v = B.name_base + r;  // digit sl from the right
        for (int32_t i = 0; i < sl && i < ln; i++) v /= 10;
        if (sl < ln) p[36 + ln - 1 - sl] = (uint8_t)('0' + v % 10);
        for (int32_t i = sl + kSub; i < ln; i += kSub) {  // more than 16 digits
            for (int32_t q = 0; q < kSub; q++) v /= 10;
            p[36 + ln - 1 - i] = (uint8_t)('0' + v % 10);
        }
    }
    uint8_t* q = p + 36 + ln + 1;
    {
        const uint32_t* cg = B.cigar + B.cigar_off[r];
        U32u* w = reinterpret_cast<U32u*>(q);
        for (int32_t k = sl; k < nc; k += kSub) w[k].v = cg[k];
    }
    q += 4 * (int64_t)nc;
    const int64_t bo = B.base_off[r];
    const int64_t n_bases = B.base_off[n];
    const uint8_t* bs = B.bases + bo;
    for (int32_t u = sl * 16; u < L; u += kSub * 16) {  // BAMRecordView.java:150-166
        uint32_t w[4];
        if (bo + u + 16 <= n_bases) {
            const U32x4u t = *reinterpret_cast<const U32x4u*>(bs + u);
            w[0] = t.v[0], w[1] = t.v[1], w[2] = t.v[2], w[3] = t.v[3];
        } else {
            w[0] = w[1] = w[2] = w[3] = 0;
            for (int32_t j = 0; u + j < L; j++) w[j >> 2] |= (uint32_t)bs[u + j] << (8 * (j & 3));
        }
        const int32_t nb = L - u < 16 ? L - u : 16;
        uint32_t o[2] = {0u, 0u};
#pragma unroll
        for (int j = 0; j < 16; j++) {
            uint32_t c = lut[(w[j >> 2] >> (8 * (j & 3))) & 255u];
            if (j >= nb)
                c = 0;
            else if (c == 0xFFu)
                bad = true, c = 15;
            o[j >> 3] |= c << (8 * ((j & 7) >> 1) + ((j & 1) ? 0 : 4));
        }
        uint8_t* d = q + (u >> 1);
        if (nb == 16) {
            U32x2u t;
            t.v[0] = o[0], t.v[1] = o[1];
            *reinterpret_cast<U32x2u*>(d) = t;
        } else {
            for (int32_t j = 0; j < (nb + 1) / 2; j++) d[j] = (uint8_t)(o[j >> 2] >> (8 * (j & 3)));
        }
    }
    q += (L + 1) / 2;
    const bool has_q = !(B.qual_status && B.qual_status[r] == 1);
    const uint8_t* qs = B.quals + bo;
    for (int32_t u = sl * 16; u < L; u += kSub * 16) {  // :172-185
        if (u + 16 <= L && (!has_q || bo + u + 16 <= n_bases)) {
            U32x4u t;
            if (has_q)
                t = *reinterpret_cast<const U32x4u*>(qs + u);
            else
                t.v[0] = t.v[1] = t.v[2] = t.v[3] = 0xFFFFFFFFu;
            *reinterpret_cast<U32x4u*>(q + u) = t;
        } else {
            for (int32_t j = u; j < L && j < u + 16; j++) q[j] = has_q ? qs[j] : (uint8_t)0xFF;
        }
    }
    q += L;
    if (lt) {
        const uint8_t* tg = B.tags + B.tag_off[r];
        for (int32_t i = sl; i < lt; i += kSub) q[i] = tg[i];
    }
    if (bad) status[r] = CRAMGPU_E_JAVA_THROWS;  // status was zeroed before the launch; every lane stores the same code
}

// ---------------------------------------------------------------------------------------------------------------
// BAM records in
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int32_t ld32(const uint8_t* p) { return (int32_t)reinterpret_cast<const U32u*>(p)->v; }

__global__ __launch_bounds__(256) void bam_head_kernel(int32_t n, const uint8_t* bam, const int64_t* rec_off, cramgpu_bam_out O,
                                                       int32_t* status) {
    const int32_t r = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    if (r >= n) return;
    const uint8_t* p = bam + rec_off[r];
    const int64_t size = rec_off[r + 1] - rec_off[r];
    int32_t ln = 0, nc = 0, L = 0;
    int64_t need = 0;
    bool ok = size >= 36;
    if (ok) {
        ln = p[12], nc = (int32_t)p[16] | (int32_t)p[17] << 8, L = ld32(p + 20);
        need = 36 + (int64_t)ln + 4 * (int64_t)nc + ((int64_t)L + 1) / 2 + L;
        ok = L >= 0 && need <= size && (int64_t)ld32(p) + 4 == size;
    }
    if (!ok) {
        status[r] = CRAMGPU_E_CORRUPT;
        O.ref_id[r] = -1, O.aln_start[r] = 0, O.mapq[r] = 0, O.bam_flags[r] = 0, O.mate_ref[r] = -1, O.mate_start[r] = 0;
        O.tlen[r] = 0, O.read_len[r] = 0, O.n_cigar[r] = 0, O.aln_span[r] = 0, O.rflags[r] = CRAMGPU_RF_SKIP;
        O.name_off[r] = 0, O.name_len[r] = 0, O.tag_off[r] = 0, O.tag_len[r] = 0;
        return;
    }
    status[r] = 0;
    const int32_t flags = (int32_t)p[18] | (int32_t)p[19] << 8;
    O.ref_id[r] = ld32(p + 4);
    O.aln_start[r] = ld32(p + 8) + 1;
    O.mapq[r] = p[13];
    O.bam_flags[r] = flags;
    O.read_len[r] = L;
    O.mate_ref[r] = ld32(p + 24);
    O.mate_start[r] = ld32(p + 28) + 1;
    O.tlen[r] = ld32(p + 32);
    O.n_cigar[r] = nc;
    O.name_off[r] = rec_off[r] + 36;
    O.name_len[r] = ln > 0 ? ln - 1 : 0;
    O.tag_off[r] = rec_off[r] + need;
    O.tag_len[r] = (int32_t)(size - need);
    int32_t span = 0;
    const uint8_t* q = p + 36 + ln;
    for (int32_t k = 0; k < nc; k++) {  // BAMRecordView.java:589-621
        const uint32_t e = (uint32_t)ld32(q + 4 * k);
        const uint32_t o = e & 15u;
        if (o == 0 || o == 2 || o == 3 || o == 7 || o == 8) span += (int32_t)(e >> 4);
    }
    O.aln_span[r] = span;
    const uint8_t* sc = q + 4 * (int64_t)nc + (L + 1) / 2;
    O.rflags[r] = (uint8_t)(((flags & BF_UNMAPPED) ? CRAMGPU_RF_UNMAPPED : 0) | ((L == 0 || sc[0] == 0xFF) ? CRAMGPU_RF_NULL_QUALS : 0) |
                            (L == 0 ? CRAMGPU_RF_NO_BASES : 0));
}

__global__ __launch_bounds__(256) void bam_body_kernel(int32_t n, const uint8_t* bam, const int64_t* rec_off, cramgpu_bam_out O,
                                                       int32_t* status) {
    __shared__ uint8_t dec[16];
    if (threadIdx.x < 16) dec[threadIdx.x] = (uint8_t)"=ACMGRSVTWYHKDBN"[threadIdx.x];
    __syncthreads();
    const int32_t r = (int32_t)(blockIdx.x * (blockDim.x / kSub) + threadIdx.x / kSub);
    const int sl = (int)(threadIdx.x % kSub);
    if (r >= n) return;
    if (O.base_off[n] > O.bases_cap || O.cigar_off[n] > O.cigar_cap) {
        if (sl == 0) status[r] = CRAMGPU_E_INSUFFICIENT;
        return;
    }
    if (status[r] != 0) return;
    const uint8_t* p = bam + rec_off[r];
    const int32_t ln = p[12], nc = O.n_cigar[r], L = O.read_len[r];
    const uint8_t* q = p + 36 + ln;
    uint32_t* cg = O.cigar + O.cigar_off[r];
    for (int32_t k = sl; k < nc; k += kSub) cg[k] = (uint32_t)ld32(q + 4 * k);
    q += 4 * (int64_t)nc;
    const int64_t bo = O.base_off[r];
    const int64_t size_left = rec_off[r + 1] - (rec_off[r] + 36 + ln + 4 * (int64_t)nc);  // bytes from q to the record's end
    uint8_t* bd = O.bases + bo;
    for (int32_t u = sl * 16; u < L; u += kSub * 16) {
        const int32_t nb = L - u < 16 ? L - u : 16;
        uint32_t w[2] = {0u, 0u};
        if ((u >> 1) + 8 <= size_left) {
            const U32x2u t = *reinterpret_cast<const U32x2u*>(q + (u >> 1));
            w[0] = t.v[0], w[1] = t.v[1];
        } else {
            for (int32_t j = 0; j < (nb + 1) / 2; j++) w[j >> 2] |= (uint32_t)q[(u >> 1) + j] << (8 * (j & 3));
        }
        uint32_t o[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t byte = (w[j >> 3] >> (8 * ((j & 7) >> 1))) & 255u;
            const uint32_t c = dec[(j & 1) ? (byte & 15u) : (byte >> 4)];
            o[j >> 2] |= c << (8 * (j & 3));
        }
        if (nb == 16 && bo + u + 16 <= O.bases_cap) {
            U32x4u t;
            t.v[0] = o[0], t.v[1] = o[1], t.v[2] = o[2], t.v[3] = o[3];
            *reinterpret_cast<U32x4u*>(bd + u) = t;
        } else {
            for (int32_t j = 0; j < nb; j++) bd[u + j] = (uint8_t)(o[j >> 2] >> (8 * (j & 3)));
        }
    }
    q += (L + 1) / 2;
    uint8_t* qd = O.quals + bo;
    for (int32_t u = sl * 16; u < L; u += kSub * 16) {
        if (u + 16 <= L) {
            *reinterpret_cast<U32x4u*>(qd + u) = *reinterpret_cast<const U32x4u*>(q + u);
        } else {
            for (int32_t j = u; j < L; j++) qd[j] = q[j];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Slice reference MD5 — htsjdk!structure/Slice#setRefMD5 (L145-177) on encode (CramSerilization.java:184-197), the check of
// src/main/java/net/sf/cram/Cram2Bam.java:212-234 through src/main/java/net/sf/cram/ref/ReferenceRegion.java:116-124 on
// decode: MD5 (RFC 1321) of the reference bases the slice spans, the span clipped at the end of the reference.
// One chain of 64 dependent rounds per 64-byte block and no parallelism inside a message: a thread per slice, the
// slices of a batch side by side (a 64 KB span is ~1000 blocks, ~2 M cycles).
// ---------------------------------------------------------------------------------------------------------------
__constant__ uint32_t kMd5K[64] = {
    0xd76aa478u, 0xe8c7b756u, 0x242070dbu, 0xc1bdceeeu, 0xf57c0fafu, 0x4787c62au, 0xa8304613u, 0xfd469501u, 0x698098d8u, 0x8b44f7afu,
    0xffff5bb1u, 0x895cd7beu, 0x6b901122u, 0xfd987193u, 0xa679438eu, 0x49b40821u, 0xf61e2562u, 0xc040b340u, 0x265e5a51u, 0xe9b6c7aau,
    0xd62f105du, 0x02441453u, 0xd8a1e681u, 0xe7d3fbc8u, 0x21e1cde6u, 0xc33707d6u, 0xf4d50d87u, 0x455a14edu, 0xa9e3e905u, 0xfcefa3f8u,
    0x676f02d9u, 0x8d2a4c8au, 0xfffa3942u, 0x8771f681u, 0x6d9d6122u, 0xfde5380cu, 0xa4beea44u, 0x4bdecfa9u, 0xf6bb4b60u, 0xbebfbc70u,
    0x289b7ec6u, 0xeaa127fau, 0xd4ef3085u, 0x04881d05u, 0xd9d4d039u, 0xe6db99e5u, 0x1fa27cf8u, 0xc4ac5665u, 0xf4292244u, 0x432aff97u,
    0xab9423a7u, 0xfc93a039u, 0x655b59c3u, 0x8f0ccc92u, 0xffeff47du, 0x85845dd1u, 0x6fa87e4fu, 0xfe2ce6e0u, 0xa3014314u, 0x4e0811a1u,
    0xf7537e82u, 0xbd3af235u, 0x2ad7d2bbu, 0xeb86d391u};

struct Md5 {
    uint32_t a, b, c, d;
};
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
__device__ void md5_block(Md5& h, const uint32_t* m) {
    constexpr int S[4][4] = {{7, 12, 17, 22}, {5, 9, 14, 20}, {4, 11, 16, 23}, {6, 10, 15, 21}};
    uint32_t a = h.a, b = h.b, c = h.c, d = h.d;
#pragma unroll
    for (int i = 0; i < 64; i++) {
        uint32_t f;
        int g;
        if (i < 16)
            f = (b & c) | (~b & d), g = i;
        else if (i < 32)
            f = (d & b) | (~d & c), g = (5 * i + 1) & 15;
        else if (i < 48)
            f = b ^ c ^ d, g = (3 * i + 5) & 15;
        else
            f = c ^ (b | ~d), g = (7 * i) & 15;
        const uint32_t t = d;
        d = c;
        c = b;
        b = b + rotl32(a + f + kMd5K[i] + m[g], S[i >> 4][i & 3]);
        a = t;
    }
    h.a += a, h.b += b, h.c += c, h.d += d;
}

__global__ __launch_bounds__(64) void slice_md5_kernel(int32_t n, const uint8_t* ref, const cramgpu_refwin* win,
                                                       const int32_t* start, const int32_t* span, uint8_t* md5) {
    const int32_t sidx = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    if (sidx >= n) return;
    U32x4u* out = reinterpret_cast<U32x4u*>(md5 + 16 * (int64_t)sidx);
    U32x4u o;
    if (start[sidx] < 1) {  // a slice without alignment start: 16 zero bytes (Slice#setRefMD5's empty case)
        o.v[0] = o.v[1] = o.v[2] = o.v[3] = 0u;
        *out = o;
        return;
    }
    const cramgpu_refwin W = win[sidx];
    const int64_t from = (int64_t)start[sidx] - 1 - W.ref_start0;  // ReferenceRegion.java:117-123
    int64_t len = 0;
    if (from >= 0 && from < W.ref_len) {
        len = span[sidx] < 0 ? 0 : span[sidx];
        if (len > W.ref_len - from) len = W.ref_len - from;
    }
    const uint8_t* p = ref + W.ref_off + (from > 0 ? from : 0);
    Md5 h{0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u};
    uint32_t m[16];
    int64_t at = 0;
    for (; at + 64 <= len; at += 64) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const U32x4u t = *reinterpret_cast<const U32x4u*>(p + at + 16 * q);
            m[4 * q] = t.v[0], m[4 * q + 1] = t.v[1], m[4 * q + 2] = t.v[2], m[4 * q + 3] = t.v[3];
        }
        md5_block(h, m);
    }
    // the tail: remaining bytes, 0x80, zeros, the bit length (one or two blocks)
    const int rem = (int)(len - at);
#pragma unroll
    for (int q = 0; q < 16; q++) m[q] = 0u;
    for (int j = 0; j < rem; j++) m[j >> 2] |= (uint32_t)p[at + j] << (8 * (j & 3));
    m[rem >> 2] |= 0x80u << (8 * (rem & 3));
    if (rem >= 56) {
        md5_block(h, m);
#pragma unroll
        for (int q = 0; q < 16; q++) m[q] = 0u;
    }
    const unsigned long long bits = (unsigned long long)len * 8ull;
    m[14] = (uint32_t)bits, m[15] = (uint32_t)(bits >> 32);
    md5_block(h, m);
    o.v[0] = h.a, o.v[1] = h.b, o.v[2] = h.c, o.v[3] = h.d;
    *out = o;
}

// ---------------------------------------------------------------------------------------------------------------
// setMateInfo — src/main/java/htsjdk/samtools/cram/CramSerilization.java:274-373 (and #detach :265-272): which records of a
// container leave as a (mate downstream, NF) pair and which as detached records.  The reference walks the records in
// order through two TreeMap<readName, record> (primary / secondary alignments) and links the second record of a name to
// the first; a third record of the name detaches all of them (:294-298; the links stay, every later check can only
// detach again); a pair survives the checks of :320-346 or both ends are detached; a name seen once is detached
// (:350-370).  Per (container, secondary?, name) that is a function of the group's size and of its two lowest record
// indices, which is what is computed here: an open-addressing table over the batch (exact comparison of the name
// bytes on a hit), a count and the lowest index per group, then one thread per pair evaluates the checks.
// ---------------------------------------------------------------------------------------------------------------
struct MateInfo {
    const int32_t *bam_flags, *ref_id, *aln_start, *aln_end, *mate_flags, *mate_ref, *mate_start, *tlen;
    int32_t *comp_flags, *next_frag;
    const uint8_t* names;
    const int64_t* name_off;
    const int32_t* name_len;
};
constexpr int32_t BF_FIRST_SEGMENT = 0x40, BF_LAST_SEGMENT = 0x80, BF_SECONDARY = 0x100;

__device__ __forceinline__ void mate_set_lone(const MateInfo& M, int32_t r) {  // :279-282, :353-356, #detach
    M.comp_flags[r] = (M.comp_flags[r] | CRAMGPU_CF_DETACHED) & ~CRAMGPU_CF_HAS_MATE_DOWNSTREAM;
    M.next_frag[r] = -1;
}

__global__ __launch_bounds__(256) void mate_group_kernel(int32_t n, int32_t n_cont, const int32_t* cont_first, MateInfo M,
                                                         int32_t* table, uint32_t mask, uint32_t* gfirst, uint32_t* gcount,
                                                         int32_t* slot_of) {
    const int32_t r = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    if (r >= n) return;
    const int32_t bf = M.bam_flags[r];
    if (!(bf & BF_PAIRED)) {  // :278-284
        mate_set_lone(M, r);
        slot_of[r] = -1;
        return;
    }
    const int32_t c = cont_of(cont_first, n_cont, r);
    const int32_t lo = cont_first[c], hi = cont_first[c + 1];
    const uint32_t sec = (bf & BF_SECONDARY) ? 1u : 0u;
    const uint8_t* nm = M.names + M.name_off[r];
    const int32_t ln = M.name_len[r];
    uint32_t h = 2166136261u ^ ((uint32_t)c * 0x9E3779B1u + sec);
    for (int32_t i = 0; i < ln; i++) h = (h ^ nm[i]) * 16777619u;
    h ^= h >> 15;
    uint32_t at = h & mask;
    for (;; at = (at + 1) & mask) {
        int32_t j = atomicCAS(&table[at], -1, r);
        if (j == -1) break;  // a new (container, map, name)
        if (j < lo || j >= hi || M.name_len[j] != ln || (((uint32_t)M.bam_flags[j] & BF_SECONDARY) ? 1u : 0u) != sec) continue;
        const uint8_t* o = M.names + M.name_off[j];
        int32_t i = 0;
        while (i < ln && o[i] == nm[i]) i++;
        if (i == ln) break;
    }
    slot_of[r] = (int32_t)at;
    atomicMin(&gfirst[at], (uint32_t)r);
    atomicAdd(&gcount[at], 1u);
}

__global__ __launch_bounds__(256) void mate_decide_kernel(int32_t n, MateInfo M, const uint32_t* gfirst, const uint32_t* gcount,
                                                          const int32_t* slot_of) {
    const int32_t z = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    if (z >= n) return;
    const int32_t at = slot_of[z];
    if (at < 0) return;
    if (gcount[at] != 2u) {  // a name seen once (:350-370) or more than twice (:294-298)
        mate_set_lone(M, z);
        return;
    }
    const int32_t r = (int32_t)gfirst[at];
    if (r == z) return;  // the pair is decided by its second record
    const int32_t bfr = M.bam_flags[r], bfz = M.bam_flags[z];
    bool ok = ((bfr & BF_FIRST_SEGMENT) && (bfz & BF_LAST_SEGMENT)) || ((bfz & BF_FIRST_SEGMENT) && (bfr & BF_LAST_SEGMENT));  // :320
    int32_t t = 0;  // CramNormalizer.java:455-470
    if (!(bfr & BF_UNMAPPED) && !(bfz & BF_UNMAPPED) && M.ref_id[r] == M.ref_id[z]) {
        const int32_t p1 = (bfr & BF_REVERSE) ? M.aln_end[r] : M.aln_start[r];
        const int32_t p2 = (bfz & BF_REVERSE) ? M.aln_end[z] : M.aln_start[z];
        t = p2 - p1 + (p2 >= p1 ? 1 : -1);
    }
    ok = ok && M.tlen[r] == t && M.tlen[z] == -t;  // :323-336
    ok = ok && M.mate_ref[r] == M.ref_id[z] && M.ref_id[r] == M.mate_ref[z] &&  // :341-346
         ((M.mate_flags[r] & MF_NEG_STRAND) != 0) == ((bfz & BF_REVERSE) != 0) &&
         ((M.mate_flags[z] & MF_NEG_STRAND) != 0) == ((bfr & BF_REVERSE) != 0) && M.mate_start[r] == M.aln_start[z] &&
         M.mate_start[z] == M.aln_start[r];
    if (!ok) {
        mate_set_lone(M, r);
        mate_set_lone(M, z);
        return;
    }
    M.next_frag[r] = z - r - 1;  // :301-307; the second record's recordsToNextFragment is left as it is
    M.comp_flags[r] = (M.comp_flags[r] | CRAMGPU_CF_HAS_MATE_DOWNSTREAM) & ~CRAMGPU_CF_DETACHED;
    M.comp_flags[z] = M.comp_flags[z] & ~(CRAMGPU_CF_HAS_MATE_DOWNSTREAM | CRAMGPU_CF_DETACHED);
}

// ---------------------------------------------------------------------------------------------------------------
// Slice content digests — htsjdk!cram/digest/ContentDigests as htsjdk!build/ContainerFactory#buildSlice drives it
// (ContentDigests.create(ALL), add(record) per record, getAsTags() -> Slice.sliceTags).  BD / SD: java.util.zip.CRC32 of
// every record's readBases / qualityScores (Crc32Hasher), the per-record values added as 32-bit integers
// (IntegerSumCombine).  A thread per (record, series): CRC-32 (polynomial 0xEDB88320, the zlib one), sixteen bytes per
// load and four bytes per step through four 256-entry tables in LDS (slicing by four: the four reads of a step do not
// depend on each other), the tail byte by byte; then one atomic add per record into the slice's sum.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void content_crc_kernel(int32_t n, int32_t n_cont, const int32_t* cont_first,
                                                          const int64_t* base_off, const int32_t* read_len, const uint8_t* bases,
                                                          const uint8_t* quals, uint32_t* crc_sum) {
    __shared__ uint32_t T[4][256];
    {
        uint32_t v = threadIdx.x;
#pragma unroll
        for (int k = 0; k < 8; k++) v = (v >> 1) ^ (0xEDB88320u & (0u - (v & 1u)));
        T[0][threadIdx.x] = v;
    }
    __syncthreads();
    {   // T[k][b] = CRC of byte b followed by k zero bytes
        uint32_t v = T[0][threadIdx.x];
#pragma unroll
        for (int k = 1; k < 4; k++) {
            v = T[0][v & 0xFFu] ^ (v >> 8);
            T[k][threadIdx.x] = v;
        }
    }
    __syncthreads();
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t r = (int32_t)(id >> 1), which = (int32_t)(id & 1);
    uint32_t v = 0;  // the record's CRC (nothing for the lanes behind the last record)
    int32_t cont = -1;
    if (r < n) {
        const uint8_t* p = (which ? quals : bases) + base_off[r];
        const int32_t L = read_len[r];
        uint32_t c = 0xFFFFFFFFu;
        int32_t i = 0;
        for (; i + 16 <= L; i += 16) {
            const U32x4u w = *reinterpret_cast<const U32x4u*>(p + i);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t x = c ^ w.v[k];
                c = T[3][x & 0xFFu] ^ T[2][(x >> 8) & 0xFFu] ^ T[1][(x >> 16) & 0xFFu] ^ T[0][x >> 24];
            }
        }
        for (; i < L; i++) c = T[0][(c ^ p[i]) & 0xFFu] ^ (c >> 8);
        v = ~c;
        cont = cont_of(cont_first, n_cont, r);
    }
    // A wave holds 32 consecutive records, nearly always of one container: its 64 values are summed across the lanes of a
    // parity first (bases in the even lanes, scores in the odd ones) and leave as TWO atomic adds.  One add per record made
    // every wave of the launch queue up on the two words of the container in hand: 13.6 ms for configs[1], 12 of them waiting.
    const int32_t c0 = __shfl(cont, 0);
    const bool same = __ballot(cont != c0 && cont >= 0) == 0 && c0 >= 0;
    if (same) {
        for (int d = 2; d < 64; d <<= 1) v += (uint32_t)__shfl_xor((int)v, d);
        if ((threadIdx.x & 63) < 2) atomicAdd(&crc_sum[2 * (int64_t)c0 + which], v);
    } else if (cont >= 0) {
        atomicAdd(&crc_sum[2 * (int64_t)cont + which], v);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Record boundaries of a BAM stripe on the device (cramgpu_bam_index_dev).  off[k + 1] = off[k] + 4 + block_size(off[k])
// is one chain over the whole stripe (src/main/java/htsjdk/samtools/cram/paralell/Bam_OBA_Supplier.java:49-62 walks it on
// the host, and so does cramgpu_bam_index); walked by one thread it would take a memory round trip per record.
//   1. GUESS: the stripe is cut into segments of 4 KiB; a wave per segment looks for the first position that reads like the
//      head of a record AND whose successor does (sizes that add up, reference ids in range, the read name's NUL where
//      l_read_name says): almost surely the first record that starts in the segment.
//   2. WALK: a thread per segment follows the chain from its guess to the first record start beyond the segment.
//   3. VERIFY: one workgroup goes through the segments in order; the position the TRUE chain enters segment k at is the exit
//      of the last segment before it that had a start (a max-scan, 1024 segments at a time).  Where the guess differs, it is
//      replaced by the chain's position and the segment is walked again, and the scan repeated.  By induction every segment
//      then starts from the true chain: the result does not depend on how good the guesses were, only the time does.
//   4. counts -> offsets (scan), and a second walk writes the record offsets.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kIdxSeg = 4096;
constexpr int64_t kIdxNone = 0x7FFFFFFFFFFFFFFFll;

__device__ __forceinline__ int32_t idx_i32(const uint8_t* p) { return (int32_t)reinterpret_cast<const U32u*>(p)->v; }

// does bam[o ...] read like the head of a record?  next: where its successor starts
__device__ __forceinline__ bool idx_plausible(const uint8_t* bam, int64_t len, int64_t o, int32_t n_ref, int64_t& next) {
    if (o + 36 > len) return false;
    const int32_t bs = idx_i32(bam + o);
    if (bs < 32 || o + 4 + (int64_t)bs > len) return false;
    const int32_t ref = idx_i32(bam + o + 4), pos = idx_i32(bam + o + 8), l_seq = idx_i32(bam + o + 20);
    const int32_t mref = idx_i32(bam + o + 24), mpos = idx_i32(bam + o + 28);
    const int32_t l_name = bam[o + 12], n_cigar = (int32_t)bam[o + 16] | (int32_t)bam[o + 17] << 8;
    if (ref < -1 || ref >= n_ref || mref < -1 || mref >= n_ref || pos < -1 || mpos < -1 || l_seq < 0 || l_name < 1) return false;
    const int64_t need = 32 + (int64_t)l_name + 4 * (int64_t)n_cigar + ((int64_t)l_seq + 1) / 2 + (int64_t)l_seq;
    if (need > bs) return false;
    if (bam[o + 36 + l_name - 1] != 0) return false;
    next = o + 4 + (int64_t)bs;
    return true;
}

__global__ __launch_bounds__(64) void bam_idx_guess_kernel(const uint8_t* bam, int64_t len, int32_t n_ref, int64_t n_seg,
                                                           int64_t* entry) {
    const int64_t k = (int64_t)blockIdx.x;
    const int lane = (int)threadIdx.x;
    if (k == 0) {  // the stripe starts with a record
        if (lane == 0) entry[0] = len > 0 ? 0 : kIdxNone;
        return;
    }
    const int64_t lo = k * kIdxSeg, hi = lo + kIdxSeg < len ? lo + kIdxSeg : len;
    int64_t found = kIdxNone;
    for (int64_t at = lo; at < hi; at += 64) {
        const int64_t o = at + lane;
        int64_t nx = 0, nx2 = 0;
        bool ok = o < hi && idx_plausible(bam, len, o, n_ref, nx);
        if (ok && nx < len) ok = idx_plausible(bam, len, nx, n_ref, nx2);
        const unsigned long long m = __ballot(ok);
        if (m != 0) {
            found = at + (__ffsll((long long)m) - 1);
            break;
        }
    }
    if (lane == 0) entry[k] = found;
}

// the chain from e to the first record start at or beyond seg_end; returns the number of records that start in between
__device__ __forceinline__ int32_t idx_walk(const uint8_t* bam, int64_t len, int64_t e, int64_t seg_end, int64_t* exit_at,
                                            int64_t* out, int64_t out_room, int32_t* status) {
    int64_t o = e;
    int32_t n = 0;
    while (o < seg_end) {
        bool bad = o + 4 > len;
        int32_t bs = 0;
        if (!bad) {
            bs = idx_i32(bam + o);
            bad = bs < 32 || o + 4 + (int64_t)bs > len;
        }
        if (bad) {  // Bam_OBA_Supplier: SAMFormatException "Invalid record length" — the chain ends here
            *status = CRAMGPU_E_CORRUPT;
            o = len;
            break;
        }
        if (out != nullptr && n < out_room) out[n] = o;
        n++;
        o += 4 + (int64_t)bs;
    }
    *exit_at = o;
    return n;
}

__global__ __launch_bounds__(256) void bam_idx_walk_kernel(const uint8_t* bam, int64_t len, int64_t n_seg, const int64_t* entry,
                                                           int64_t* exit_at, int32_t* count) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_seg) return;
    const int64_t e = entry[k];
    int32_t scratch = 0;  // a guess may run into garbage: the last pass, over the verified chain, reports
    if (e == kIdxNone) {
        exit_at[k] = 0;
        count[k] = 0;
        return;
    }
    const int64_t hi = (k + 1) * kIdxSeg < len ? (k + 1) * kIdxSeg : len;
    count[k] = idx_walk(bam, len, e, hi, &exit_at[k], nullptr, 0, &scratch);
}

__global__ __launch_bounds__(1024) void bam_idx_verify_kernel(const uint8_t* bam, int64_t len, int64_t n_seg, int64_t* entry,
                                                              int64_t* exit_at, int32_t* count) {
    __shared__ long long sc[1024];
    __shared__ long long carry;   // where the chain stands in front of the chunk
    __shared__ int first_bad;
    const int tid = (int)threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int64_t k0 = 0; k0 < n_seg; k0 += 1024) {
        for (;;) {
            const int64_t k = k0 + tid;
            const bool live = k < n_seg;
            int64_t e = live ? entry[k] : kIdxNone;
            if (live && carry >= len && e != kIdxNone) {  // the verified chain has ended (or broken off) in front of this chunk
                entry[k] = e = kIdxNone;
                exit_at[k] = 0;
                count[k] = 0;
            }
            sc[tid] = e != kIdxNone ? exit_at[k] : 0;
            if (tid == 0) first_bad = 1024;
            __syncthreads();
            for (int d = 1; d < 1024; d <<= 1) {  // inclusive max-scan
                const long long v = tid >= d ? sc[tid - d] : 0;
                __syncthreads();
                if (v > sc[tid]) sc[tid] = v;
                __syncthreads();
            }
            long long c = carry;
            if (tid > 0 && sc[tid - 1] > c) c = sc[tid - 1];
            const int64_t lo = k * kIdxSeg, hi = lo + kIdxSeg < len ? lo + kIdxSeg : len;
            const int64_t want = (c >= lo && c < hi) ? (int64_t)c : kIdxNone;
            if (live && e != want) atomicMin(&first_bad, tid);
            __syncthreads();
            const int fb = first_bad;
            if (fb == 1024) {
                __syncthreads();
                if (tid == 1023 && sc[1023] > carry) carry = sc[1023];
                __syncthreads();
                break;
            }
            if (tid == fb) {  // every segment before this one is on the true chain, so `want` is where the chain enters it
                entry[k] = want;
                if (want == kIdxNone) {
                    exit_at[k] = 0;
                    count[k] = 0;
                } else {
                    int32_t scratch = 0;
                    count[k] = idx_walk(bam, len, want, hi, &exit_at[k], nullptr, 0, &scratch);
                }
            }
            __threadfence_block();
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(256) void bam_idx_write_kernel(const uint8_t* bam, int64_t len, int64_t n_seg, const int64_t* entry,
                                                            const int64_t* base, int64_t* rec_off, int64_t cap, int64_t* n_out,
                                                            int32_t* status) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_seg) return;
    const int64_t total = base[n_seg];
    if (k == 0) {
        *n_out = total;
        if (total < cap) rec_off[total] = len;
    }
    const int64_t e = entry[k];
    if (e == kIdxNone) return;
    const int64_t hi = (k + 1) * kIdxSeg < len ? (k + 1) * kIdxSeg : len, b = base[k];
    int64_t ex;
    idx_walk(bam, len, e, hi, &ex, rec_off + (b < cap ? b : cap), cap > b ? cap - b : 0, status);  // the verified chain: reports
}

inline dim3 grid1(int64_t n, int per_block) { return dim3((unsigned)((n + per_block - 1) / per_block > 0 ? (n + per_block - 1) / per_block : 1)); }

}  // namespace

hipError_t launch_cigars(int32_t n, const int32_t* read_len, const int32_t* aln_start, const int32_t* bam_flags,
                         const cramgpu_features& f, int32_t* n_cigar, int64_t* cigar_off, uint32_t* cigar, int64_t cap,
                         int32_t* aln_end, int32_t* status, int64_t* d_total, int64_t* d_scan_ws, hipStream_t s) {
    hipLaunchKernelGGL(cigar_kernel<false>, grid1(n, 256), dim3(256), 0, s, n, read_len, aln_start, bam_flags, f, n_cigar,
                       (const int64_t*)nullptr, (uint32_t*)nullptr, cap, aln_end, status);
    hipError_t e = launch_feature_scan(n, n_cigar, cigar_off, d_total, d_scan_ws, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(cigar_kernel<true>, grid1(n, 256), dim3(256), 0, s, n, read_len, aln_start, bam_flags, f, n_cigar,
                       (const int64_t*)cigar_off, cigar, cap, aln_end, status);
    return hipGetLastError();
}

hipError_t launch_restore_mates(int32_t n, int32_t n_cont, const int32_t* cont_first, const cramgpu_record_cols& c,
                                const int32_t* cont_ref_id, const int32_t* aln_start, const int32_t* aln_end, int32_t* link,
                                uint32_t* nprev, uint8_t* serial, int32_t* status, hipStream_t s) {
    hipError_t e = hipMemsetAsync(nprev, 0, (size_t)(n > 0 ? n : 1) * sizeof(uint32_t), s);
    if (e == hipSuccess) e = hipMemsetAsync(serial, 0, (size_t)n_cont, s);
    if (e == hipSuccess) e = hipMemsetAsync(status, 0, (size_t)n_cont * sizeof(int32_t), s);
    if (e != hipSuccess || n == 0) return e;
    hipLaunchKernelGGL(mates_link_kernel, grid1(n, 256), dim3(256), 0, s, n, n_cont, cont_first, c, cont_ref_id, link, nprev,
                       serial, status);
    hipLaunchKernelGGL(mates_walk_kernel, grid1(n, 256), dim3(256), 0, s, n, n_cont, cont_first, c, aln_start, aln_end,
                       (const int32_t*)link, (const uint32_t*)nprev, (const uint8_t*)serial, (const int32_t*)status);
    return hipGetLastError();
}

// ws: table[cap] | gfirst[cap] | gcount[cap] | slot_of[n], cap = mate_table_slots(n)
size_t mate_table_slots(int32_t n) {
    size_t cap = 1024;
    while (cap < 2 * (size_t)(n > 0 ? n : 1)) cap <<= 1;
    return cap;
}
hipError_t launch_set_mate_info(int32_t n, int32_t n_cont, const int32_t* cont_first, const cramgpu_record_cols& c,
                                const int32_t* aln_start, const int32_t* aln_end, const uint8_t* names, const int64_t* name_off,
                                const int32_t* name_len, int32_t* ws, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const size_t cap = mate_table_slots(n);
    int32_t* table = ws;
    uint32_t* gfirst = reinterpret_cast<uint32_t*>(ws + cap);
    uint32_t* gcount = reinterpret_cast<uint32_t*>(ws + 2 * cap);
    int32_t* slot_of = ws + 3 * cap;
    hipError_t e = hipMemsetAsync(table, 0xFF, 2 * cap * sizeof(int32_t), s);  // table = -1, gfirst = 0xFFFFFFFF
    if (e == hipSuccess) e = hipMemsetAsync(gcount, 0, cap * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    const MateInfo M{c.bam_flags, c.ref_id,    aln_start, aln_end, c.mate_flags, c.mate_ref, c.mate_start,
                     c.tlen,      c.comp_flags, c.next_frag, names, name_off,     name_len};
    hipLaunchKernelGGL(mate_group_kernel, grid1(n, 256), dim3(256), 0, s, n, n_cont, cont_first, M, table, (uint32_t)(cap - 1),
                       gfirst, gcount, slot_of);
    hipLaunchKernelGGL(mate_decide_kernel, grid1(n, 256), dim3(256), 0, s, n, M, (const uint32_t*)gfirst, (const uint32_t*)gcount,
                       (const int32_t*)slot_of);
    return hipGetLastError();
}

hipError_t launch_content_crc(int32_t n, int32_t n_cont, const int32_t* cont_first, const int64_t* base_off, const int32_t* read_len,
                              const uint8_t* bases, const uint8_t* quals, uint32_t* crc_sum, hipStream_t s) {
    hipError_t e = hipMemsetAsync(crc_sum, 0, 2 * (size_t)n_cont * sizeof(uint32_t), s);
    if (e != hipSuccess || n == 0) return e;
    hipLaunchKernelGGL(content_crc_kernel, grid1(2 * (int64_t)n, 256), dim3(256), 0, s, n, n_cont, cont_first, base_off, read_len,
                       bases, quals, crc_sum);
    return hipGetLastError();
}

hipError_t launch_bam_encode(int32_t n, const cramgpu_bam_cols& b, uint8_t* bam, int64_t cap, int64_t* rec_off, int32_t* status,
                             int32_t* d_sizes, int64_t* d_total, int64_t* d_scan_ws, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(bam_size_kernel, grid1(n, 256), dim3(256), 0, s, n, b, d_sizes);
    hipError_t e = launch_feature_scan(n, d_sizes, rec_off, d_total, d_scan_ws, s);
    if (e == hipSuccess && n > 0) e = hipMemsetAsync(status, 0, (size_t)n * sizeof(int32_t), s);
    if (e != hipSuccess || n == 0) return e;
    hipLaunchKernelGGL(bam_write_kernel, grid1(n, 256 / kSub), dim3(256), 0, s, n, b, bam, cap, (const int64_t*)rec_off, status);
    return hipGetLastError();
}

hipError_t launch_bam_decode(int32_t n, const uint8_t* bam, const int64_t* rec_off, const cramgpu_bam_out& o, int32_t* status,
                             int64_t* d_total, int64_t* d_scan_ws, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(bam_head_kernel, grid1(n, 256), dim3(256), 0, s, n, bam, rec_off, o, status);
    hipError_t e = launch_feature_scan(n, o.read_len, o.base_off, d_total, d_scan_ws, s);
    if (e == hipSuccess) e = launch_feature_scan(n, o.n_cigar, o.cigar_off, d_total, d_scan_ws, s);
    if (e != hipSuccess || n == 0) return e;
    hipLaunchKernelGGL(bam_body_kernel, grid1(n, 256 / kSub), dim3(256), 0, s, n, bam, rec_off, o, status);
    return hipGetLastError();
}

hipError_t launch_slice_md5(int32_t n, const uint8_t* ref, const cramgpu_refwin* win, const int32_t* start, const int32_t* span,
                            uint8_t* md5, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(slice_md5_kernel, grid1(n, 64), dim3(64), 0, s, n, ref, win, start, span, md5);
    return hipGetLastError();
}

size_t bam_index_segments(int64_t len) { return (size_t)((len + kIdxSeg - 1) / kIdxSeg); }

// ws_entry / ws_exit: int64[n_seg]; ws_count: int32[n_seg]; ws_base: int64[n_seg + 1]
hipError_t launch_bam_index(const uint8_t* bam, int64_t len, int32_t n_ref, int64_t* rec_off, int64_t cap, int64_t* d_count,
                            int32_t* d_status, int64_t* ws_entry, int64_t* ws_exit, int32_t* ws_count, int64_t* ws_base,
                            int64_t* d_total, int64_t* d_scan_ws, hipStream_t s) {
    hipError_t e = hipMemsetAsync(d_status, 0, sizeof(int32_t), s);
    if (e != hipSuccess) return e;
    const int64_t n_seg = (int64_t)bam_index_segments(len);
    if (n_seg == 0) {  // an empty stripe: no record, rec_off[0] = 0
        e = hipMemsetAsync(d_count, 0, sizeof(int64_t), s);
        if (e == hipSuccess && cap > 0) e = hipMemsetAsync(rec_off, 0, sizeof(int64_t), s);
        return e;
    }
    hipLaunchKernelGGL(bam_idx_guess_kernel, dim3((unsigned)n_seg), dim3(64), 0, s, bam, len, n_ref, n_seg, ws_entry);
    hipLaunchKernelGGL(bam_idx_walk_kernel, grid1(n_seg, 256), dim3(256), 0, s, bam, len, n_seg, (const int64_t*)ws_entry, ws_exit,
                       ws_count);
    hipLaunchKernelGGL(bam_idx_verify_kernel, dim3(1), dim3(1024), 0, s, bam, len, n_seg, ws_entry, ws_exit, ws_count);
    e = launch_feature_scan((int32_t)n_seg, ws_count, ws_base, d_total, d_scan_ws, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(bam_idx_write_kernel, grid1(n_seg, 256), dim3(256), 0, s, bam, len, n_seg, (const int64_t*)ws_entry,
                       (const int64_t*)ws_base, rec_off, cap, d_count, d_status);
    return hipGetLastError();
}
