/*
 * TEST INFRASTRUCTURE — CPU oracle of the data-series writer / reader.  Never linked into the
 * product library; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use it.
 *
 * Restates htsjdk 2.4.1 `encoding/writer/Writer#write` (L139-261) and `encoding/reader/CramRecordReader#read`
 * (L59-211) for the integer / byte / byte-array-stop series of a slice, and `io/ITF8#writeUnsignedITF8` (L53-84) /
 * `#readUnsignedITF8`.  Those classes ship only as bytecode (lib/htsjdk-2.4.1.jar), so the statements here follow
 *   - SURVEY.md Appendix A (record field order, operator bytes, compression-flag bits) and §8(d) (ITF8 width rule,
 *     series-to-codec map, content id = EncodingKey ordinal), and
 *   - the reference's OWN readers of the same series, which are source:
 *     src/main/java/htsjdk/samtools/cram/encoding/reader/ReaderToBAM.java:64-176 (record field order: BF, CF, [RI],
 *     RL, AP (+= delta), RG, [RN], detached { MF, [RN], NS, NP, TS } else mate-downstream { NF }, TL, tags, mapped
 *     { features, MQ, [QS x readLength] } else { BA x readLength, [QS x readLength] }) and
 *     src/main/java/htsjdk/samtools/cram/encoding/reader/ReadFeatureBuffer.java:41-100 (FN, then per feature FC,
 *     FP (delta from the previous feature), payload: B -> BA, QS; X -> BS; I -> IN; S -> SC; H -> HC; P -> PD; D -> DL;
 *     N -> RS; i -> BA; Q -> QS).
 * PARITY UNPINNED at the reference level for the exact bytes: no fixture of the reference holds a Writer output and
 * the bytecode cannot run here.  Read names (RN) and tag values (TN/TC and the per-tag series) are strings the host
 * keeps: not written here.
 *
 * Series index = content id = htsjdk EncodingKey ordinal:
 *   BF 0, AP 1, FP 2, FC 3, QS 4, DL 5, BA 6, TN 7, NF 8, RL 9, RG 10, MQ 11, RN 12, NP 13, TS 14, FN 15, BS 16, IN 17,
 *   TC 18, MF 19, NS 20, CF 21, TV 22, TM 23, TL 24, RI 25, RS 26, SC 27, HC 28, PD 29
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { BF = 0, AP = 1, FP = 2, FC = 3, QS = 4, DL = 5, BA = 6, NF = 8, RL = 9, RG = 10, MQ = 11, NP = 13, TS = 14, FN = 15,
       BS = 16, IN = 17, MF = 19, NS = 20, CF = 21, TL = 24, RI = 25, RS = 26, SC = 27, HC = 28, PD = 29, N_SERIES = 30 };
enum { CF_FORCE_QS = 1, CF_DETACHED = 2, CF_MATE_DOWNSTREAM = 4, CF_UNKNOWN_BASES = 8 };
enum { BAM_UNMAPPED = 4 };

/* ITF8#writeUnsignedITF8 L53-84 (width rule of SURVEY.md §8(d)) */
int32_t orc_itf8_put(uint8_t* p, int32_t value) {
    const uint32_t v = (uint32_t)value;
    if (v < (1u << 7)) {
        p[0] = (uint8_t)v;
        return 1;
    }
    if (v < (1u << 14)) {
        p[0] = (uint8_t)(0x80 | (v >> 8));
        p[1] = (uint8_t)v;
        return 2;
    }
    if (v < (1u << 21)) {
        p[0] = (uint8_t)(0xC0 | (v >> 16));
        p[1] = (uint8_t)(v >> 8);
        p[2] = (uint8_t)v;
        return 3;
    }
    if (v < (1u << 28)) {
        p[0] = (uint8_t)(0xE0 | (v >> 24));
        p[1] = (uint8_t)(v >> 16);
        p[2] = (uint8_t)(v >> 8);
        p[3] = (uint8_t)v;
        return 4;
    }
    p[0] = (uint8_t)(0xF0 | (v >> 28));
    p[1] = (uint8_t)(v >> 20);
    p[2] = (uint8_t)(v >> 12);
    p[3] = (uint8_t)(v >> 4);
    p[4] = (uint8_t)(v & 0x0F);
    return 5;
}

/* ITF8#readUnsignedITF8: returns bytes consumed (0 when the value runs past `end`) */
int32_t orc_itf8_get(const uint8_t* p, const uint8_t* end, int32_t* value) {
    if (p >= end) return 0;
    const uint32_t b1 = p[0];
    int n = b1 < 0x80 ? 1 : b1 < 0xC0 ? 2 : b1 < 0xE0 ? 3 : b1 < 0xF0 ? 4 : 5;
    if (p + n > end) return 0;
    uint32_t v;
    switch (n) {
        case 1: v = b1; break;
        case 2: v = ((b1 & 0x7F) << 8) | p[1]; break;
        case 3: v = ((b1 & 0x3F) << 16) | ((uint32_t)p[1] << 8) | p[2]; break;
        case 4: v = ((b1 & 0x1F) << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; break;
        default: v = ((b1 & 0x0F) << 28) | ((uint32_t)p[1] << 20) | ((uint32_t)p[2] << 12) | ((uint32_t)p[3] << 4) | (p[4] & 0x0F);
    }
    *value = (int32_t)v;
    return n;
}

typedef struct {
    uint8_t* p;
    int64_t len, cap;
} buf_t;

static int put_i(buf_t* b, int32_t v) {
    if (b->len + 5 > b->cap) return -1;
    b->len += orc_itf8_put(b->p + b->len, v);
    return 0;
}
static int put_b(buf_t* b, uint8_t v) {
    if (b->len + 1 > b->cap) return -1;
    b->p[b->len++] = v;
    return 0;
}
static int put_bytes(buf_t* b, const uint8_t* s, int64_t n) {
    if (b->len + n > b->cap) return -1;
    memcpy(b->p + b->len, s, (size_t)n);
    b->len += n;
    return 0;
}
#define TRY(e) do { if ((e) != 0) return -1; } while (0)

/*
 * Writer#write for the n_reads records of one slice, in order.  Feature columns as cramgpu_features on the encode side
 * (X: a = base | ref << 8 | code << 16; S, I: a = offset of the sequence in the read's bases, b = length; D N H P:
 * a = length; i: a = base; Q: a = score; B: a = base, b = score).  out = N_SERIES buffers of `cap` bytes each, back to
 * back; out_len[N_SERIES].  multiref: the slice's sequence id is MULTI_REFERENCE (RI written).  ap_base: the slice
 * alignment start the first record's delta is taken from.  Returns 0, or -1 when a buffer is too small.
 */
int32_t orc_write_series(int32_t n_reads, const int32_t* read_len, const int32_t* aln_start, const int32_t* bam_flags,
                         const int32_t* comp_flags, const int32_t* ref_id, const int32_t* read_group, const int32_t* mate_flags,
                         const int32_t* mate_ref, const int32_t* mate_start, const int32_t* tlen, const int32_t* next_frag,
                         const int32_t* tag_list, const uint8_t* mapq, const uint8_t* bases, const uint8_t* quals,
                         const int64_t* base_off, const int64_t* feat_off, const uint8_t* op, const int32_t* pos,
                         const int32_t* fa, const int32_t* fb, int32_t multiref, int32_t ap_base, int32_t ap_delta,
                         uint8_t* out, int64_t cap, int64_t* out_len) {
    buf_t s[N_SERIES];
    for (int k = 0; k < N_SERIES; k++) {
        s[k].p = out + (int64_t)k * cap;
        s[k].len = 0;
        s[k].cap = cap;
    }
    int32_t prev_start = ap_base;
    for (int32_t r = 0; r < n_reads; r++) {
        const int32_t cf = comp_flags[r];
        TRY(put_i(&s[BF], bam_flags[r]));
        TRY(put_i(&s[CF], cf));
        if (multiref) TRY(put_i(&s[RI], ref_id[r]));
        TRY(put_i(&s[RL], read_len[r]));
        if (ap_delta) {
            TRY(put_i(&s[AP], aln_start[r] - prev_start));
            prev_start = aln_start[r];
        } else {
            TRY(put_i(&s[AP], aln_start[r]));
        }
        TRY(put_i(&s[RG], read_group[r]));
        if (cf & CF_DETACHED) {
            TRY(put_i(&s[MF], mate_flags[r]));
            TRY(put_i(&s[NS], mate_ref[r]));
            TRY(put_i(&s[NP], mate_start[r]));
            TRY(put_i(&s[TS], tlen[r]));
        } else if (cf & CF_MATE_DOWNSTREAM) {
            TRY(put_i(&s[NF], next_frag[r]));
        }
        TRY(put_i(&s[TL], tag_list[r]));
        const uint8_t* rb = bases + base_off[r];
        const uint8_t* rq = quals + base_off[r];
        if (!(bam_flags[r] & BAM_UNMAPPED)) {
            const int64_t f0 = feat_off[r], f1 = feat_off[r + 1];
            TRY(put_i(&s[FN], (int32_t)(f1 - f0)));
            int32_t prev = 0;
            for (int64_t f = f0; f < f1; f++) {
                TRY(put_b(&s[FC], op[f]));
                TRY(put_i(&s[FP], pos[f] - prev));
                prev = pos[f];
                switch (op[f]) {
                    case 'B':
                        TRY(put_b(&s[BA], (uint8_t)fa[f]));
                        TRY(put_b(&s[QS], (uint8_t)fb[f]));
                        break;
                    case 'X': TRY(put_b(&s[BS], (uint8_t)(fa[f] >> 16))); break;
                    case 'I':
                        TRY(put_bytes(&s[IN], rb + fa[f], fb[f]));
                        TRY(put_b(&s[IN], '\t'));
                        break;
                    case 'S':
                        TRY(put_bytes(&s[SC], rb + fa[f], fb[f]));
                        TRY(put_b(&s[SC], '\t'));
                        break;
                    case 'H': TRY(put_i(&s[HC], fa[f])); break;
                    case 'P': TRY(put_i(&s[PD], fa[f])); break;
                    case 'D': TRY(put_i(&s[DL], fa[f])); break;
                    case 'N': TRY(put_i(&s[RS], fa[f])); break;
                    case 'i': TRY(put_b(&s[BA], (uint8_t)fa[f])); break;
                    case 'Q': TRY(put_b(&s[QS], (uint8_t)fa[f])); break;
                    default: return -2;
                }
            }
            TRY(put_i(&s[MQ], mapq[r]));
            if (cf & CF_FORCE_QS) TRY(put_bytes(&s[QS], rq, read_len[r]));
        } else {
            if (!(cf & CF_UNKNOWN_BASES)) TRY(put_bytes(&s[BA], rb, read_len[r]));
            if (cf & CF_FORCE_QS) TRY(put_bytes(&s[QS], rq, read_len[r]));
        }
    }
    for (int k = 0; k < N_SERIES; k++) out_len[k] = s[k].len;
    return 0;
}

typedef struct {
    const uint8_t* p;
    const uint8_t* end;
} cur_t;
static int get_i(cur_t* c, int32_t* v) {
    const int n = orc_itf8_get(c->p, c->end, v);
    if (n == 0) return -1;
    c->p += n;
    return 0;
}
static int get_b(cur_t* c, uint8_t* v) {
    if (c->p >= c->end) return -1;
    *v = *c->p++;
    return 0;
}

/*
 * CramRecordReader#read for n_reads records: the inverse of orc_write_series.  in = the N_SERIES streams back to back
 * with stride `cap`, in_len[N_SERIES].  Columns not present for a record are left at the value the reference's
 * CramCompressionRecord has before read(): 0 (mate columns, next_frag), -1 never written here.  Decode-side feature
 * columns: S, I: a = offset of the sequence in `seq` relative to seq_off[r], b = length (the bytes of IN / SC are
 * appended to seq in record order); X: a = code << 16.  Unmapped records' bases go to bases_out at base_off[r], QS
 * arrays to quals_out at base_off[r] (0xFF-filled where a record has no array).  Returns the feature count, or -1 on a
 * truncated stream, -2 on an unknown operator, -3 when a capacity is too small.
 */
int64_t orc_read_series(int32_t n_reads, const uint8_t* in, int64_t cap, const int64_t* in_len, int32_t multiref, int32_t ap_base,
                        int32_t ap_delta, int32_t* read_len, int32_t* aln_start, int32_t* bam_flags, int32_t* comp_flags,
                        int32_t* ref_id, int32_t* read_group, int32_t* mate_flags, int32_t* mate_ref, int32_t* mate_start,
                        int32_t* tlen, int32_t* next_frag, int32_t* tag_list, uint8_t* mapq, int64_t* base_off,
                        uint8_t* bases_out, uint8_t* quals_out, int64_t bases_cap, int64_t* feat_off, uint8_t* op, int32_t* pos,
                        int32_t* fa, int32_t* fb, int64_t feat_cap, uint8_t* seq, int64_t* seq_off, int64_t seq_cap) {
    cur_t s[N_SERIES];
    for (int k = 0; k < N_SERIES; k++) {
        s[k].p = in + (int64_t)k * cap;
        s[k].end = s[k].p + in_len[k];
    }
    int32_t prev_start = ap_base;
    int64_t nf = 0, nb = 0, nseq = 0;
    for (int32_t r = 0; r < n_reads; r++) {
        int32_t cf;
        TRY(get_i(&s[BF], &bam_flags[r]));
        TRY(get_i(&s[CF], &cf));
        comp_flags[r] = cf;
        ref_id[r] = 0;
        if (multiref) TRY(get_i(&s[RI], &ref_id[r]));
        TRY(get_i(&s[RL], &read_len[r]));
        int32_t v;
        TRY(get_i(&s[AP], &v));
        if (ap_delta) {
            prev_start += v;
            aln_start[r] = prev_start;
        } else {
            aln_start[r] = v;
        }
        TRY(get_i(&s[RG], &read_group[r]));
        mate_flags[r] = mate_ref[r] = mate_start[r] = tlen[r] = next_frag[r] = 0;
        if (cf & CF_DETACHED) {
            TRY(get_i(&s[MF], &mate_flags[r]));
            TRY(get_i(&s[NS], &mate_ref[r]));
            TRY(get_i(&s[NP], &mate_start[r]));
            TRY(get_i(&s[TS], &tlen[r]));
        } else if (cf & CF_MATE_DOWNSTREAM) {
            TRY(get_i(&s[NF], &next_frag[r]));
        }
        TRY(get_i(&s[TL], &tag_list[r]));
        const int32_t L = read_len[r];
        if (L < 0 || nb + L > bases_cap) return -3;
        base_off[r] = nb;
        seq_off[r] = nseq;
        feat_off[r] = nf;
        memset(quals_out + nb, 0xFF, (size_t)L);
        mapq[r] = 0; /* SAMRecord.NO_MAPPING_QUALITY (ReaderToBAM.java:157); a record without MQ keeps the field's 0 */
        if (!(bam_flags[r] & BAM_UNMAPPED)) {
            int32_t n;
            TRY(get_i(&s[FN], &n));
            if (n < 0 || nf + n > feat_cap) return -3;
            int32_t prev = 0;
            for (int32_t k = 0; k < n; k++, nf++) {
                uint8_t o, b;
                TRY(get_b(&s[FC], &o));
                TRY(get_i(&s[FP], &v));
                prev += v;
                op[nf] = o;
                pos[nf] = prev;
                fa[nf] = fb[nf] = 0;
                switch (o) {
                    case 'B':
                        TRY(get_b(&s[BA], &b));
                        fa[nf] = b;
                        TRY(get_b(&s[QS], &b));
                        fb[nf] = b;
                        break;
                    case 'X':
                        TRY(get_b(&s[BS], &b));
                        fa[nf] = (int32_t)b << 16;
                        break;
                    case 'I':
                    case 'S': {
                        cur_t* c = &s[o == 'I' ? IN : SC];
                        const uint8_t* start = c->p;
                        while (c->p < c->end && *c->p != '\t') c->p++;
                        if (c->p >= c->end) return -1;
                        const int64_t len = c->p - start;
                        c->p++;
                        if (nseq + len > seq_cap) return -3;
                        memcpy(seq + nseq, start, (size_t)len);
                        fa[nf] = (int32_t)(nseq - seq_off[r]);
                        fb[nf] = (int32_t)len;
                        nseq += len;
                        break;
                    }
                    case 'H': TRY(get_i(&s[HC], &fa[nf])); break;
                    case 'P': TRY(get_i(&s[PD], &fa[nf])); break;
                    case 'D': TRY(get_i(&s[DL], &fa[nf])); break;
                    case 'N': TRY(get_i(&s[RS], &fa[nf])); break;
                    case 'i':
                        TRY(get_b(&s[BA], &b));
                        fa[nf] = b;
                        break;
                    case 'Q':
                        TRY(get_b(&s[QS], &b));
                        fa[nf] = b;
                        break;
                    default: return -2;
                }
            }
            TRY(get_i(&s[MQ], &v));
            mapq[r] = (uint8_t)v;
            if (cf & CF_FORCE_QS) {
                if (s[QS].p + L > s[QS].end) return -1;
                memcpy(quals_out + nb, s[QS].p, (size_t)L);
                s[QS].p += L;
            }
        } else {
            if (!(cf & CF_UNKNOWN_BASES)) {
                if (s[BA].p + L > s[BA].end) return -1;
                memcpy(bases_out + nb, s[BA].p, (size_t)L);
                s[BA].p += L;
            }
            if (cf & CF_FORCE_QS) {
                if (s[QS].p + L > s[QS].end) return -1;
                memcpy(quals_out + nb, s[QS].p, (size_t)L);
                s[QS].p += L;
            }
        }
        nb += L;
    }
    feat_off[n_reads] = nf;
    seq_off[n_reads] = nseq;
    base_off[n_reads] = nb;
    return nf;
}
