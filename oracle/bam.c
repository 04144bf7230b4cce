/*
 * TEST INFRASTRUCTURE — CPU oracle of container assembly and the BAM record codec.
 * Never linked into the product library; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use it.
 *
 * What is restated, and from where:
 *   orc_cigars         CIGAR of a mapped record from its read features: htsjdk 2.4.1 build/Cram2SamRecordFactory#getCigar2
 *                      (L119-225, bytecode only in lib/htsjdk-2.4.1.jar).  The reference tree states the same algorithm
 *                      in source: src/main/java/htsjdk/samtools/cram/encoding/reader/ReadFeatureBuffer.java:180-294
 *                      (#getCigar).  Followed statement by statement, over feature columns instead of that class's
 *                      private byte buffer (whose cursor :249-250 advances two bytes for a one-byte `X` payload, :59 —
 *                      a property of that buffer, not of the algorithm).
 *                      Alignment end: htsjdk structure/CramCompressionRecord#getAlignmentEnd (bytecode only): the span
 *                      is readLength - |I| - |i| - |S| + |D|; restated from the published 2.4.1 source, PARITY UNPINNED
 *                      (no fixture of the reference covers it).
 *   orc_restore_mates  src/main/java/htsjdk/samtools/cram/build/CramNormalizer.java:82-105 (pairing), :255-285
 *                      (restoreMateInfo, setNextMate), :455-470 (computeInsertSize).  Source in the reference tree.
 *   orc_bam_encode     one BAM alignment record per CRAM record: the byte layout of
 *                      src/main/java/htsjdk/samtools/cram/encoding/reader/BAMRecordView.java:23-35 (field offsets),
 *                      :78-80,:142-144 (0-based starts), :106-119 (CIGAR words), :150-166 (4-bit bases, table :360-415),
 *                      :172-185 (scores; 0xFF x length when absent), :229-241 (block size), as driven by
 *                      ReaderToBAM.java:64-176; the values are those htsjdk build/Cram2SamRecordFactory#create hands to
 *                      BAMRecordCodec#encode in CramToBam_OBA_Function.java:71-74 (unplaced record: start 0, mapping
 *                      quality 0; unpaired record: no mate reference / start; mate flag bits from the mate flags).
 *                      Index bin: SAM specification §4.2.1 / §5.3 reg2bin(pos, end) with end = pos + 1 for a record
 *                      without alignment span, 0 for a record without reference (BAMRecordCodec#encode; bytecode only).
 *   orc_bam_decode     the inverse field walk (BAMRecordCodec#decode as called from BamToCram_OBA_Function over the
 *                      stripes of Bam_OBA_Supplier.java:45-82): columns of cramgpu_reads / cramgpu_record_cols.
 * No fixture of the reference holds BAM records made by these classes and the bytecode cannot run here (no JVM):
 * PARITY UNPINNED at the reference level; the known-answer tests of tests/test_oracle_bam.py are worked by hand from
 * the sources cited above and from the SAM specification's record layout.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { OP_M = 0, OP_I = 1, OP_D = 2, OP_N = 3, OP_S = 4, OP_H = 5, OP_P = 6 };
enum { BF_PAIRED = 1, BF_UNMAPPED = 4, BF_MATE_UNMAPPED = 8, BF_REVERSE = 16, BF_MATE_REVERSE = 32 };
enum { CF_DETACHED = 2, CF_MATE_DOWNSTREAM = 4 };
enum { MF_NEG_STRAND = 1, MF_UNMAPPED = 2 };

/* operator and length of a feature for the CIGAR walk; -1 = not a CIGAR feature (Q, q, b: ReadFeatureBuffer.java:252-256) */
static int cigar_op_of(uint8_t op, int32_t a, int32_t b, int32_t* len) {
    switch (op) {
        case 'I': *len = b; return OP_I;
        case 'S': *len = b; return OP_S;
        case 'H': *len = a; return OP_H;
        case 'i': *len = 1; return OP_I;
        case 'D': *len = a; return OP_D;
        case 'N': *len = a; return OP_N;
        case 'P': *len = a; return OP_P;
        case 'X':
        case 'B': *len = 1; return OP_M;
        default: return -1;
    }
}
static int consumes_read(int op) { return op == OP_M || op == OP_I || op == OP_S; }

/* ReadFeatureBuffer.java:180-294.  Writes at most cap elements to out (may be NULL to count); returns the count. */
static int32_t cigar_of(int32_t read_len, const uint8_t* op, const int32_t* pos, const int32_t* a, const int32_t* b,
                        int64_t n_feat, uint32_t* out) {
    int32_t n = 0;
#define EMIT(len, o)                                                  \
    do {                                                              \
        if (out) out[n] = ((uint32_t)(len) << 4) | (uint32_t)(o);     \
        n++;                                                          \
    } while (0)
    if (n_feat == 0) { /* :182-185 */
        EMIT(read_len, OP_M);
        return n;
    }
    int last_op = OP_M, last_len = 0, last_pos = 1;
    for (int64_t k = 0; k < n_feat; k++) {
        const int32_t rf_pos = pos[k];
        const int32_t gap = rf_pos - (last_pos + last_len); /* :199-211 */
        if (gap > 0) {
            if (last_op != OP_M) {
                EMIT(last_len, last_op);
                last_pos += last_len;
                last_len = gap;
            } else {
                last_len += gap;
            }
            last_op = OP_M;
        }
        int32_t rf_len = 0;
        const int co = cigar_op_of(op[k], a[k], b[k], &rf_len);
        if (co < 0) continue;
        if (last_op != co) { /* :259-269 */
            if (last_len > 0) EMIT(last_len, last_op);
            last_op = co;
            last_len = rf_len;
            last_pos = rf_pos;
        } else {
            last_len += rf_len;
        }
        if (!consumes_read(co)) last_pos -= rf_len; /* :271-272 */
    }
    if (last_op != OP_M) { /* :275-286 */
        EMIT(last_len, last_op);
        if (read_len >= last_pos + last_len) EMIT(read_len - (last_len + last_pos) + 1, OP_M);
    } else if (read_len > last_pos - 1) {
        EMIT(read_len - last_pos + 1, OP_M);
    }
    if (n == 0) EMIT(read_len, OP_M); /* :288-291 */
#undef EMIT
    return n;
}

/* CIGARs and alignment ends of n records.  Unmapped records (BF & 4): no CIGAR ("*"), end 0.  n_cigar[r], cigar_off[n+1]
 * always; cigar written when the total fits cigar_cap.  Returns the total. */
int64_t orc_cigars(int32_t n, const int32_t* read_len, const int32_t* aln_start, const int32_t* bam_flags, const uint8_t* op,
                   const int32_t* pos, const int32_t* a, const int32_t* b, const int64_t* feat_off, int32_t* n_cigar,
                   int64_t* cigar_off, uint32_t* cigar, int64_t cigar_cap, int32_t* aln_end) {
    int64_t tot = 0;
    for (int32_t r = 0; r < n; r++) {
        const int64_t f0 = feat_off[r], f1 = feat_off[r + 1];
        cigar_off[r] = tot;
        if (bam_flags[r] & BF_UNMAPPED) {
            n_cigar[r] = 0;
            aln_end[r] = 0;
            continue;
        }
        n_cigar[r] = cigar_of(read_len[r], op + f0, pos + f0, a + f0, b + f0, f1 - f0, NULL);
        tot += n_cigar[r];
        int32_t span = read_len[r];
        for (int64_t k = f0; k < f1; k++) switch (op[k]) {
                case 'i': span--; break;
                case 'I':
                case 'S': span -= b[k]; break;
                case 'D': span += a[k]; break;
                default: break;
            }
        aln_end[r] = aln_start[r] + span - 1;
    }
    cigar_off[n] = tot;
    if (tot > cigar_cap) return tot;
    for (int32_t r = 0; r < n; r++)
        if (n_cigar[r]) {
            const int64_t f0 = feat_off[r], f1 = feat_off[r + 1];
            cigar_of(read_len[r], op + f0, pos + f0, a + f0, b + f0, f1 - f0, cigar + cigar_off[r]);
        }
    return tot;
}

/* CramNormalizer.java:82-105, 255-285, 455-470 for the records [first, last) of one container.
 * ref_id[r] = sequence id of record r.  Returns 0, or -9 where Java's records.get(...) throws (or, for a negative
 * distance, would link a record to itself or backwards). */
int32_t orc_restore_mates(int32_t first, int32_t last, const int32_t* bam_flags, const int32_t* comp_flags, const int32_t* ref_id,
                          const int32_t* aln_start, const int32_t* aln_end, int32_t* next_frag, int32_t* mate_flags,
                          int32_t* mate_ref, int32_t* mate_start, int32_t* tlen) {
    const int32_t n = last - first;
    int32_t* next = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    int32_t* prev = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    for (int32_t i = 0; i < n; i++) next[i] = prev[i] = -1;
    for (int32_t i = 0; i < n; i++) { /* :83-97 */
        const int32_t r = first + i;
        if (!(bam_flags[r] & BF_PAIRED) || (comp_flags[r] & CF_DETACHED)) {
            next_frag[r] = -1;
            next[i] = prev[i] = -1;
            continue;
        }
        if (comp_flags[r] & CF_MATE_DOWNSTREAM) {
            const int64_t j = (int64_t)i + 1 + next_frag[r];
            if (next_frag[r] < 0 || j >= n) { /* negative: a self / backward link, the walk of :262-265 would not end */
                free(next);
                free(prev);
                return -9;
            }
            next[i] = (int32_t)j;
            prev[j] = i;
        }
    }
#define SET_NEXT_MATE(ri, ni) /* :278-285 */                                                                      \
    do {                                                                                                          \
        mate_start[ri] = aln_start[ni];                                                                           \
        mate_flags[ri] = (mate_flags[ri] & ~(MF_UNMAPPED | MF_NEG_STRAND)) | ((bam_flags[ni] & BF_UNMAPPED) ? MF_UNMAPPED : 0) | \
                         ((bam_flags[ni] & BF_REVERSE) ? MF_NEG_STRAND : 0);                                      \
        mate_ref[ri] = ref_id[ni];                                                                                \
        if (mate_ref[ri] == -1) mate_start[ri] = 0;                                                               \
    } while (0)
    for (int32_t i = 0; i < n; i++) { /* :98-104 */
        if (prev[i] >= 0 || next[i] < 0) continue;
        int32_t cur = i;
        while (next[cur] >= 0) {
            SET_NEXT_MATE(first + cur, first + next[cur]);
            cur = next[cur];
        }
        const int32_t a = first + i, z = first + cur;
        SET_NEXT_MATE(z, a);
        int32_t t = 0; /* :455-470 */
        if (!(bam_flags[a] & BF_UNMAPPED) && !(bam_flags[z] & BF_UNMAPPED) && ref_id[a] == ref_id[z]) {
            const int32_t p1 = (bam_flags[a] & BF_REVERSE) ? aln_end[a] : aln_start[a];
            const int32_t p2 = (bam_flags[z] & BF_REVERSE) ? aln_end[z] : aln_start[z];
            t = p2 - p1 + (p2 >= p1 ? 1 : -1);
        }
        tlen[a] = t;
        tlen[z] = -t;
    }
#undef SET_NEXT_MATE
    free(next);
    free(prev);
    return 0;
}

/* SAM specification §5.3: bin of the 0-based half-open interval [beg, end) */
static int32_t reg2bin(int64_t beg, int64_t end) {
    --end;
    if (beg >> 14 == end >> 14) return (int32_t)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int32_t)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int32_t)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int32_t)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int32_t)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

/* BAMRecordView.java:360-415 (either nybble); -1 = IllegalArgumentException */
static int base_nybble(uint8_t c) {
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
        default: return -1;
    }
}
static void put32(uint8_t* p, int32_t v) {
    p[0] = (uint8_t)v, p[1] = (uint8_t)(v >> 8), p[2] = (uint8_t)(v >> 16), p[3] = (uint8_t)(v >> 24);
}
static int32_t get32(const uint8_t* p) {
    return (int32_t)((uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24);
}

/* n records -> BAM alignment records back to back.  names / tags may be NULL (then the name is the decimal record
 * index + name_base, as CramNormalizer.java:108-117 assigns them, and there are no tags).  rec_off[n + 1] always; bytes
 * written when they fit cap.  status[r]: 0, or -9 (a base the 4-bit table rejects, a name longer than 254 bytes, more
 * than 65535 CIGAR elements).  quals: 0xFF x length when has_quals[r] == 0.  Returns the total size. */
int64_t orc_bam_encode(int32_t n, const int32_t* ref_id, const int32_t* aln_start, const int32_t* aln_end, const uint8_t* mapq,
                       const int32_t* bam_flags, const int32_t* mate_flags, const int32_t* mate_ref, const int32_t* mate_start,
                       const int32_t* tlen, const int32_t* read_len, const int64_t* base_off, const uint8_t* bases,
                       const uint8_t* quals, const uint8_t* has_quals, const int32_t* n_cigar, const int64_t* cigar_off,
                       const uint32_t* cigar, const uint8_t* names, const int64_t* name_off, int64_t name_base,
                       const uint8_t* tags, const int64_t* tag_off, uint8_t* out, int64_t cap, int64_t* rec_off,
                       int32_t* status) {
    int64_t tot = 0;
    char tmp[24];
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 1 && tot > cap) return tot;
        int64_t at = 0;
        for (int32_t r = 0; r < n; r++) {
            int32_t ln;
            const uint8_t* nm;
            if (names) {
                ln = (int32_t)(name_off[r + 1] - name_off[r]);
                nm = names + name_off[r];
            } else {
                long long v = (long long)name_base + r;
                int k = 0;
                char rev[24];
                do rev[k++] = (char)('0' + v % 10); while ((v /= 10) > 0);
                for (int i = 0; i < k; i++) tmp[i] = rev[k - 1 - i];
                ln = k;
                nm = (const uint8_t*)tmp;
            }
            const int32_t lt = tags ? (int32_t)(tag_off[r + 1] - tag_off[r]) : 0;
            const int32_t L = read_len[r], nc = n_cigar[r];
            const int64_t size = 36 + (ln + 1) + 4 * (int64_t)nc + (L + 1) / 2 + L + lt;
            if (pass == 0) {
                rec_off[r] = tot;
                tot += size;
                continue;
            }
            uint8_t* p = out + at;
            int32_t st = (ln + 1 > 255 || nc > 65535) ? -9 : 0;
            const int32_t bf = bam_flags[r];
            const int placed = ref_id[r] != -1;
            const int32_t start = placed ? aln_start[r] : 0;                    /* Cram2SamRecordFactory#create */
            int32_t flags = bf & ~(BF_MATE_UNMAPPED | BF_MATE_REVERSE);
            int32_t mref = -1, mstart = 0;
            if (bf & BF_PAIRED) {
                mref = mate_ref[r];
                mstart = mate_start[r] > 0 ? mate_start[r] : 0;
                if (mate_flags[r] & MF_NEG_STRAND) flags |= BF_MATE_REVERSE;
                if (mate_flags[r] & MF_UNMAPPED) flags |= BF_MATE_UNMAPPED;
            }
            int32_t bin = 0;
            if (placed) {
                const int64_t beg = (int64_t)start - 1;
                int64_t end = (bf & BF_UNMAPPED) ? 0 : aln_end[r];
                if (end <= 0 || end <= beg) end = beg + 1;
                bin = reg2bin(beg, end);
            }
            put32(p, (int32_t)(size - 4));
            put32(p + 4, ref_id[r]);
            put32(p + 8, start - 1);
            p[12] = (uint8_t)(ln + 1);
            p[13] = placed ? mapq[r] : 0;
            p[14] = (uint8_t)bin, p[15] = (uint8_t)(bin >> 8);
            p[16] = (uint8_t)nc, p[17] = (uint8_t)(nc >> 8);
            p[18] = (uint8_t)flags, p[19] = (uint8_t)(flags >> 8);
            put32(p + 20, L);
            put32(p + 24, mref);
            put32(p + 28, mstart - 1);
            put32(p + 32, tlen[r]);
            memcpy(p + 36, nm, (size_t)ln);
            p[36 + ln] = 0;
            uint8_t* q = p + 36 + ln + 1;
            for (int32_t k = 0; k < nc; k++) put32(q + 4 * k, (int32_t)cigar[cigar_off[r] + k]);
            q += 4 * (int64_t)nc;
            const uint8_t* bs = bases + base_off[r];
            for (int32_t i = 0; i < L; i += 2) {
                const int hi = base_nybble(bs[i]), lo = i + 1 < L ? base_nybble(bs[i + 1]) : 0;
                if (hi < 0 || lo < 0) st = -9;
                q[i >> 1] = (uint8_t)(((hi < 0 ? 15 : hi) << 4) | (lo < 0 ? 15 : lo));
            }
            q += (L + 1) / 2;
            if (has_quals[r])
                memcpy(q, quals + base_off[r], (size_t)L);
            else
                memset(q, 0xFF, (size_t)L);
            q += L;
            if (lt) memcpy(q, tags + tag_off[r], (size_t)lt);
            status[r] = st;
            at += size;
        }
    }
    rec_off[n] = tot;
    return tot;
}

/* Record offsets of a stripe (the walk of Bam_OBA_Supplier.java:45-82: block size, then the record).  Returns the number
 * of records, or -5 when a block size is below 32 (:59-62) or runs past the stripe. */
int64_t orc_bam_index(const uint8_t* bam, int64_t len, int64_t* rec_off, int64_t cap) {
    int64_t at = 0, n = 0;
    while (at < len) {
        if (at + 4 > len) return -5;
        const int32_t bs = get32(bam + at);
        if (bs < 32 || at + 4 + bs > len) return -5;
        if (n < cap) rec_off[n] = at;
        n++;
        at += 4 + (int64_t)bs;
    }
    if (n < cap) rec_off[n] = at;
    return n;
}

/* BAM records -> columns.  rec_off[n + 1].  base_off / cigar_off [n + 1] are prefix sums of l_seq / n_cigar_op.
 * name_off / name_len / tag_off / tag_len: where the strings lie in the stripe (the host keeps them).
 * rflags: 1 unmapped (flag 4), 2 NULL_QUALS (first score 0xFF or no scores), 4 no bases (l_seq == 0).
 * status[r]: 0 or -5 (fields that overrun the record).  Returns the number of bases, or a negative code when the
 * capacities do not hold the batch (-3). */
int64_t orc_bam_decode(int32_t n, const uint8_t* bam, const int64_t* rec_off, int32_t* ref_id, int32_t* aln_start, uint8_t* mapq,
                       int32_t* bam_flags, int32_t* mate_ref, int32_t* mate_start, int32_t* tlen, int32_t* read_len,
                       int64_t* base_off, uint8_t* bases, uint8_t* quals, int64_t bases_cap, int32_t* n_cigar, int64_t* cigar_off,
                       uint32_t* cigar, int64_t cigar_cap, int32_t* aln_span, uint8_t* rflags, int64_t* name_off,
                       int32_t* name_len, int64_t* tag_off, int32_t* tag_len, int32_t* status) {
    static const char dec[] = "=ACMGRSVTWYHKDBN";
    int64_t nb = 0, ncg = 0;
    for (int32_t r = 0; r < n; r++) {
        const uint8_t* p = bam + rec_off[r];
        const int64_t size = rec_off[r + 1] - rec_off[r];
        base_off[r] = nb;
        cigar_off[r] = ncg;
        status[r] = 0;
        const int32_t ln = p[12], nc = p[16] | p[17] << 8, L = get32(p + 20);
        const int64_t need = 36 + (int64_t)ln + 4 * (int64_t)nc + ((int64_t)L + 1) / 2 + L;
        if (size < 36 || L < 0 || need > size || get32(p) + 4 != size) {
            status[r] = -5;
            ref_id[r] = -1, aln_start[r] = 0, mapq[r] = 0, bam_flags[r] = 0, mate_ref[r] = -1, mate_start[r] = 0, tlen[r] = 0;
            read_len[r] = 0, n_cigar[r] = 0, aln_span[r] = 0, rflags[r] = 32, name_off[r] = 0, name_len[r] = 0;
            tag_off[r] = 0, tag_len[r] = 0;
            continue;
        }
        ref_id[r] = get32(p + 4);
        aln_start[r] = get32(p + 8) + 1;
        mapq[r] = p[13];
        bam_flags[r] = p[18] | p[19] << 8;
        read_len[r] = L;
        mate_ref[r] = get32(p + 24);
        mate_start[r] = get32(p + 28) + 1;
        tlen[r] = get32(p + 32);
        n_cigar[r] = nc;
        name_off[r] = rec_off[r] + 36;
        name_len[r] = ln > 0 ? ln - 1 : 0;
        tag_off[r] = rec_off[r] + need;
        tag_len[r] = (int32_t)(size - need);
        nb += L;
        ncg += nc;
        int32_t span = 0;
        const uint8_t* q = p + 36 + ln;
        for (int32_t k = 0; k < nc; k++) {
            const uint32_t e = (uint32_t)get32(q + 4 * k);
            const int o = (int)(e & 15u);
            if (o == 0 || o == 2 || o == 3 || o == 7 || o == 8) span += (int32_t)(e >> 4); /* BAMRecordView.java:602-609 */
        }
        aln_span[r] = span;
        const uint8_t* sc = q + 4 * (int64_t)nc + (L + 1) / 2;
        rflags[r] = (uint8_t)(((bam_flags[r] & 4) ? 1 : 0) | ((L == 0 || sc[0] == 0xFF) ? 2 : 0) | (L == 0 ? 4 : 0));
    }
    base_off[n] = nb;
    cigar_off[n] = ncg;
    if (nb > bases_cap || ncg > cigar_cap) return -3;
    for (int32_t r = 0; r < n; r++) {
        if (status[r]) continue;
        const uint8_t* p = bam + rec_off[r];
        const int32_t ln = p[12], nc = n_cigar[r], L = read_len[r];
        const uint8_t* q = p + 36 + ln;
        for (int32_t k = 0; k < nc; k++) cigar[cigar_off[r] + k] = (uint32_t)get32(q + 4 * k);
        q += 4 * (int64_t)nc;
        for (int32_t i = 0; i < L; i++) bases[base_off[r] + i] = (uint8_t)dec[(q[i >> 1] >> ((i & 1) ? 0 : 4)) & 15];
        q += (L + 1) / 2;
        memcpy(quals + base_off[r], q, (size_t)L);
    }
    return nb;
}

/* ------------------------------------------------------------------------------------------------------------------
 * orc_set_mate_info: src/main/java/htsjdk/samtools/cram/CramSerilization.java:265-373 (detach, setMateInfo) for the records
 * [first, last) of one container, statement by statement; the two TreeMap<String, record> of :275-276 are one
 * open-addressing table keyed by (secondary?, name bytes) here (only get / put / values() are used: order-free).
 * computeInsertSize: CramNormalizer.java:455-470.  comp_flags: bits 2 (detached) and 4 (mate downstream) are rewritten,
 * the others kept.  next_frag: recordsToNextFragment (-1 where the reference sets -1).
 * PARITY UNPINNED at the reference level (no fixture of the reference holds records after setMateInfo); the
 * known-answer cases of tests/test_oracle_bam.py are worked by hand from the source.
 * ------------------------------------------------------------------------------------------------------------------ */
enum { BF_FIRST = 0x40, BF_LAST = 0x80, BF_SECONDARY = 0x100 };

typedef struct {
    const int32_t *bam_flags, *mate_flags, *ref_id, *aln_start, *aln_end, *mate_ref, *mate_start, *tlen;
    int32_t *comp_flags, *next_frag, *next, *prev;
    int32_t first;
} MateCtx;

static void mi_detach(const MateCtx* m, int32_t i) { /* :265-272 */
    do {
        m->comp_flags[m->first + i] = (m->comp_flags[m->first + i] | CF_DETACHED) & ~CF_MATE_DOWNSTREAM;
        m->next_frag[m->first + i] = -1;
    } while ((i = m->next[i]) >= 0);
}
static int32_t mi_insert_size(const MateCtx* m, int32_t a, int32_t z) { /* CramNormalizer.java:455-470 (a, z absolute) */
    if ((m->bam_flags[a] & BF_UNMAPPED) || (m->bam_flags[z] & BF_UNMAPPED)) return 0;
    if (m->ref_id[a] != m->ref_id[z]) return 0;
    const int32_t p1 = (m->bam_flags[a] & BF_REVERSE) ? m->aln_end[a] : m->aln_start[a];
    const int32_t p2 = (m->bam_flags[z] & BF_REVERSE) ? m->aln_end[z] : m->aln_start[z];
    return p2 - p1 + (p2 >= p1 ? 1 : -1);
}
static uint64_t mi_hash(const uint8_t* s, int64_t n, int sec) {
    uint64_t h = 1469598103934665603ull ^ (uint64_t)sec;
    for (int64_t i = 0; i < n; i++) h = (h ^ s[i]) * 1099511628211ull;
    return h;
}

int32_t orc_set_mate_info(int32_t first, int32_t last, const int32_t* bam_flags, const int32_t* mate_flags, const int32_t* ref_id,
                          const int32_t* aln_start, const int32_t* aln_end, const int32_t* mate_ref, const int32_t* mate_start,
                          const int32_t* tlen, const uint8_t* names, const int64_t* name_off, int32_t* comp_flags,
                          int32_t* next_frag) {
    const int32_t n = last - first;
    if (n <= 0) return 0;
    size_t cap = 16;
    while (cap < 2 * (size_t)n) cap <<= 1;
    int32_t* table = (int32_t*)malloc(sizeof(int32_t) * cap); /* index of the record a (map, name) was put with, or -1 */
    int32_t* next = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    int32_t* prev = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    for (size_t i = 0; i < cap; i++) table[i] = -1;
    for (int32_t i = 0; i < n; i++) next[i] = prev[i] = -1;
    MateCtx m = {bam_flags, mate_flags, ref_id, aln_start, aln_end, mate_ref, mate_start, tlen, comp_flags, next_frag, next, prev, first};
#define SET_LONE(i) /* :279-284, :353-358 */                                                           \
    do {                                                                                               \
        comp_flags[first + (i)] = (comp_flags[first + (i)] | CF_DETACHED) & ~CF_MATE_DOWNSTREAM;       \
        next_frag[first + (i)] = -1;                                                                   \
        next[i] = prev[i] = -1;                                                                        \
    } while (0)
    for (int32_t i = 0; i < n; i++) { /* :277-310 */
        const int32_t r = first + i;
        if (!(bam_flags[r] & BF_PAIRED)) {
            SET_LONE(i);
            continue;
        }
        const int sec = (bam_flags[r] & BF_SECONDARY) ? 1 : 0;
        const uint8_t* nm = names + name_off[r];
        const int64_t ln = name_off[r + 1] - name_off[r];
        size_t at = (size_t)mi_hash(nm, ln, sec) & (cap - 1);
        int32_t mate = -1;
        for (;; at = (at + 1) & (cap - 1)) {
            const int32_t j = table[at];
            if (j < 0) break;
            const int32_t q = first + j;
            if (((bam_flags[q] & BF_SECONDARY) ? 1 : 0) == sec && name_off[q + 1] - name_off[q] == ln &&
                memcmp(names + name_off[q], nm, (size_t)ln) == 0) {
                mate = j;
                break;
            }
        }
        if (mate < 0) {
            table[at] = i; /* :291 */
            continue;
        }
        int32_t p = mate;
        if (next[p] >= 0 || prev[p] >= 0) { /* :294-298 */
            mi_detach(&m, i);
            mi_detach(&m, p);
            continue;
        }
        while (next[p] >= 0) p = next[p];
        next_frag[first + p] = i - p - 1; /* :301 (index difference) */
        next[p] = i;
        prev[i] = p;
        comp_flags[first + p] = (comp_flags[first + p] | CF_MATE_DOWNSTREAM) & ~CF_DETACHED;
        comp_flags[r] = comp_flags[r] & ~(CF_MATE_DOWNSTREAM | CF_DETACHED);
    }
    for (int32_t i = 0; i < n; i++) { /* :313-348 */
        if (next[i] < 0 || prev[i] >= 0) continue;
        const int32_t r = first + i;
        int32_t l = i;
        while (next[l] >= 0) l = next[l];
        int32_t z = first + l;
        if (((bam_flags[r] & BF_FIRST) && (bam_flags[z] & BF_LAST)) || ((bam_flags[z] & BF_FIRST) && (bam_flags[r] & BF_LAST))) {
            const int32_t t = mi_insert_size(&m, r, z);
            if (tlen[r] == t) {
                l = next[i];
                while (next[l] >= 0) {
                    if (tlen[first + l] != -t) break;
                    l = next[l];
                }
                if (tlen[first + l] != -t) mi_detach(&m, i);
            } else {
                mi_detach(&m, i);
            }
        } else {
            mi_detach(&m, i);
        }
        z = first + l;
        if (mate_ref[r] != ref_id[z] || ref_id[r] != mate_ref[z] ||
            ((mate_flags[r] & MF_NEG_STRAND) != 0) != ((bam_flags[z] & BF_REVERSE) != 0) ||
            ((mate_flags[z] & MF_NEG_STRAND) != 0) != ((bam_flags[r] & BF_REVERSE) != 0) || mate_start[r] != aln_start[z] ||
            mate_start[z] != aln_start[r])
            mi_detach(&m, i);
    }
    for (size_t s = 0; s < cap; s++) { /* :350-370: the values of both maps */
        const int32_t i = table[s];
        if (i < 0 || next[i] >= 0) continue;
        SET_LONE(i);
    }
#undef SET_LONE
    free(table);
    free(next);
    free(prev);
    return 0;
}
