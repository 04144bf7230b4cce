/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * Plain-C CPU restatement of the record-level kernels of the cramtools hot path, over the same
 * flat (structure-of-arrays) layout the HIP kernels use (include/cramgpu.h):
 *
 *   F1/F2  htsjdk!build/Sam2CramRecordFactory#createVariations L205-264,
 *          #addSubstitutionsAndMaskedBases L297-335, #addInsertion L282-292, #addSoftClip L268-272,
 *          htsjdk!build/Utils#normalizeBase L31-49
 *   F3     htsjdk!build/CompressionHeaderFactory#buildFrequencies L196-209,
 *          htsjdk!structure/SubstitutionMatrix#<init>([[J) L38-57, #rank L141-169
 *   N1     src/main/java/htsjdk/samtools/cram/build/CramNormalizer.java:335-441 (restoreReadBases)
 *   N2     CramNormalizer.java:287-333 (restoreQualityScores)
 *   Q2     src/main/java/htsjdk/samtools/cram/lossy/QualityScorePreservation.java:258-301,332-489,
 *          lossy/Binning.java:63-71, ref/ReferenceTracks.java:127-155
 *
 * Parity pin: SURVEY.md §8(c) KAT-2, KAT-3, KAT-4 (outputs of the reference's own bytecode), held in
 * tests/golden/records_kat.json and checked by tests/test_oracle_records.py.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* BAM cigar op codes */
enum { OP_M = 0, OP_I = 1, OP_D = 2, OP_N = 3, OP_S = 4, OP_H = 5, OP_P = 6, OP_EQ = 7, OP_X = 8 };

/* per-read flag bits shared with the product ABI (cramgpu.h CRAMGPU_RF_*) */
#define RF_UNMAPPED 1
#define RF_NULL_QUALS 2
#define RF_NO_BASES 4      /* encode: SAMRecord has no bases ("*") */
#define RF_UNKNOWN_BASES 8 /* decode: CF_UNKNOWN_BASES */
#define RF_FORCE_QS 16     /* decode: CF_FORCE_PRESERVE_QS */
#define RF_SKIP 32

static inline uint8_t normalize_base(uint8_t b) {
    switch (b) {
        case 'a': case 'A': return 'A';
        case 'c': case 'C': return 'C';
        case 'g': case 'G': return 'G';
        case 't': case 'T': return 'T';
        default: return 'N';
    }
}

static inline int acgtn_index(uint8_t b) {
    switch (b) {
        case 'A': return 0;
        case 'C': return 1;
        case 'G': return 2;
        case 'T': return 3;
        case 'N': return 4;
        default: return -1;
    }
}
static const uint8_t ACGTN[5] = {'A', 'C', 'G', 'T', 'N'};

static inline int consumes_read(int op) { return op == OP_M || op == OP_I || op == OP_S || op == OP_EQ || op == OP_X; }
static inline int consumes_ref(int op) { return op == OP_M || op == OP_D || op == OP_N || op == OP_EQ || op == OP_X; }

/*
 * F1/F2 over one container (one reference contig).
 * Output features (SoA): f_op, f_pos (1-based in-read), f_a, f_b:
 *   X: a = base | refBase<<8 | 0xFF<<16 (code unset)   D/N/H/P: a = length
 *   S: a = 0-based offset in the read of the clipped bases, b = length
 *   i: a = base
 * feat_off has n_reads+1 entries.  sub_hist[5][5] += counts of (refBase, base) over ACGTN x ACGTN.
 * Returns total feature count, or -1 if feat_cap is too small.
 */
int64_t orc_extract_features(int32_t n_reads, const uint8_t* bases, const int64_t* base_off, const int32_t* read_len,
                             const int32_t* aln_start, const uint8_t* rflags, const uint32_t* cigar,
                             const int64_t* cigar_off, const int32_t* n_cigar, const uint8_t* ref, int64_t ref_len,
                             int64_t feat_cap, uint8_t* f_op, int32_t* f_pos, int32_t* f_a, int32_t* f_b,
                             int64_t* feat_off, int64_t* sub_hist) {
    int64_t nf = 0;
#define EMIT(op, pos, a, b)                 \
    do {                                    \
        if (nf >= feat_cap) return -1;      \
        f_op[nf] = (uint8_t)(op);           \
        f_pos[nf] = (int32_t)(pos);         \
        f_a[nf] = (int32_t)(a);             \
        f_b[nf] = (int32_t)(b);             \
        nf++;                               \
    } while (0)
    for (int32_t r = 0; r < n_reads; r++) {
        feat_off[r] = nf;
        if ((rflags[r] & (RF_UNMAPPED | RF_SKIP)) || aln_start[r] == 0) continue;
        const uint8_t* rb = bases + base_off[r];
        int nobases = (rflags[r] & RF_NO_BASES) != 0;
        int pos = 0;     /* zeroBasedPositionInRead */
        int refoff = 0;  /* alignmentStartOffset */
        for (int k = 0; k < n_cigar[r]; k++) {
            uint32_t ce = cigar[cigar_off[r] + k];
            int op = (int)(ce & 15);
            int len = (int)(ce >> 4);
            switch (op) {
                case OP_D: EMIT('D', pos + 1, len, 0); break;
                case OP_N: EMIT('N', pos + 1, len, 0); break;
                case OP_P: EMIT('P', pos + 1, len, 0); break;
                case OP_H: EMIT('H', pos + 1, len, 0); break;
                case OP_S: EMIT('S', pos + 1, pos, len); break;
                case OP_I:
                    for (int i = 0; i < len; i++) EMIT('i', pos + 1 + i, nobases ? 'N' : rb[pos + i], 0);
                    break;
                case OP_M: case OP_EQ: case OP_X:
                    for (int i = 0; i < len; i++) {
                        int64_t rc = (int64_t)aln_start[r] + i + refoff - 1;
                        uint8_t refb = rc >= ref_len ? 'N' : ref[rc];
                        refb = normalize_base(refb);
                        uint8_t b = nobases ? 'N' : rb[pos + i];
                        if (b != refb) {
                            EMIT('X', pos + i + 1, (int32_t)b | ((int32_t)refb << 8) | (0xFF << 16), 0);
                            int bi = acgtn_index(b);
                            if (bi >= 0) sub_hist[acgtn_index(refb) * 5 + bi]++;
                        }
                    }
                    break;
                default: break;
            }
            if (consumes_read(op)) pos += len;
            if (consumes_ref(op)) refoff += len;
        }
    }
    feat_off[n_reads] = nf;
#undef EMIT
    return nf;
}

/*
 * F3: SubstitutionMatrix#<init>([[J).  hist is [ref][base] over ACGTN x ACGTN.  For every ref, the four other
 * bases are ranked by descending count, ties in ACGTN order; the header byte lists the 2-bit ranks in
 * ACGTN-minus-ref order, first base in the top two bits.
 */
void orc_substitution_matrix(const int64_t* hist, uint8_t* matrix5) {
    for (int r = 0; r < 5; r++) {
        int others[4], k = 0;
        for (int b = 0; b < 5; b++)
            if (b != r) others[k++] = b;
        int order[4] = {0, 1, 2, 3};
        /* stable insertion sort by descending count */
        for (int i = 1; i < 4; i++) {
            int v = order[i], j = i - 1;
            while (j >= 0 && hist[r * 5 + others[order[j]]] < hist[r * 5 + others[v]]) {
                order[j + 1] = order[j];
                j--;
            }
            order[j + 1] = v;
        }
        int rank[4];
        for (int i = 0; i < 4; i++) rank[order[i]] = i;
        matrix5[r] = (uint8_t)((rank[0] << 6) | (rank[1] << 4) | (rank[2] << 2) | rank[3]);
    }
}

/* codes[ref][base]: SubstitutionMatrix#code — 0 for any base outside ACGTN (and for base == ref) */
static void matrix_tables(const uint8_t* matrix5, uint8_t codes[5][256], uint8_t basesof[5][4]) {
    memset(codes, 0, 5 * 256);
    for (int r = 0; r < 5; r++) {
        int k = 0;
        for (int b = 0; b < 5; b++) {
            if (b == r) continue;
            int code = (matrix5[r] >> (6 - 2 * k)) & 3;
            codes[r][ACGTN[b]] = (uint8_t)code;
            basesof[r][code] = ACGTN[b];
            k++;
        }
    }
}

/* Writer#write L139-261 picks matrix.code(ref, base) for every Substitution whose code is unset */
void orc_assign_codes(int64_t n_feat, const uint8_t* f_op, int32_t* f_a, const uint8_t* matrix5) {
    uint8_t codes[5][256], basesof[5][4];
    matrix_tables(matrix5, codes, basesof);
    for (int64_t i = 0; i < n_feat; i++) {
        if (f_op[i] != 'X') continue;
        uint8_t b = (uint8_t)(f_a[i] & 0xff), rf = (uint8_t)((f_a[i] >> 8) & 0xff);
        int ri = acgtn_index(rf);
        int code = ri >= 0 ? codes[ri][b] : 0;
        f_a[i] = (f_a[i] & 0xffff) | (code << 16);
    }
}

/*
 * N1: CramNormalizer.restoreReadBases over one reference window.
 * status[r]: 0 = bases written, 1 = NULL_SEQUENCE, 2 = skipped, -1 = Java would throw (index out of range).
 * X features carry the code in bits 16..23 of f_a; S/I: a = offset into seq + seq_off[r], b = length;
 * i: a = base; B: a = base.
 */
void orc_restore_bases(int32_t n_reads, const int32_t* read_len, const int32_t* aln_start, const uint8_t* rflags,
                       const int64_t* feat_off, const uint8_t* f_op, const int32_t* f_pos, const int32_t* f_a,
                       const int32_t* f_b, const uint8_t* seq, const int64_t* seq_off, const uint8_t* win,
                       int64_t win_len, int64_t win_start0, const uint8_t* matrix5, uint8_t* out,
                       const int64_t* out_off, int32_t* status) {
    uint8_t codes[5][256], basesof[5][4];
    matrix_tables(matrix5, codes, basesof);
    for (int32_t r = 0; r < n_reads; r++) {
        if (rflags[r] & RF_SKIP) { status[r] = 2; continue; }
        if ((rflags[r] & RF_UNKNOWN_BASES) || read_len[r] == 0) { status[r] = 1; continue; }
        int L = read_len[r];
        uint8_t* b = out + out_off[r];
        int64_t al0 = (int64_t)aln_start[r] - 1;
        int64_t nf = feat_off[r + 1] - feat_off[r];
        status[r] = 0;
        if (nf == 0) {
            if (win_len + win_start0 < al0 + L) {
                memset(b, 'N', (size_t)L);
                int64_t off = al0 - win_start0, avail = win_len - off;
                if (off > 0 && avail > 0) memcpy(b, win + off, (size_t)(L < avail ? L : avail));
            } else {
                if (al0 - win_start0 < 0) { status[r] = -1; continue; }
                memcpy(b, win + (al0 - win_start0), (size_t)L);
            }
            continue;
        }
        memset(b, 0, (size_t)L);
        int pir = 1;        /* posInRead */
        int64_t pis = 0;    /* posInSeq */
        int bad = 0;
        for (int64_t f = feat_off[r]; f < feat_off[r + 1] && !bad; f++) {
            for (; pir < f_pos[f]; pir++) {
                int64_t rp = al0 + pis++ - win_start0;
                if (rp < 0 || pir - 1 >= L) { bad = 1; break; }
                b[pir - 1] = rp >= win_len ? 'N' : win[rp];
            }
            if (bad) break;
            switch (f_op[f]) {
                case 'X': {
                    int64_t rp = al0 + pis - win_start0;
                    if (rp < 0 || pir - 1 >= L) { bad = 1; break; }
                    uint8_t rb = normalize_base(rp >= win_len ? 'N' : win[rp]);
                    b[pir++ - 1] = basesof[acgtn_index(rb)][(f_a[f] >> 16) & 3];
                    pis++;
                    break;
                }
                case 'I': case 'S':
                    for (int i = 0; i < f_b[f]; i++) {
                        if (pir - 1 >= L) { bad = 1; break; }
                        b[pir++ - 1] = seq[seq_off[r] + f_a[f] + i];
                    }
                    break;
                case 'D': case 'N': pis += f_a[f]; break;
                case 'i':
                    if (pir - 1 >= L) { bad = 1; break; }
                    b[pir++ - 1] = (uint8_t)f_a[f];
                    break;
                default: break;
            }
        }
        if (bad) { status[r] = -1; continue; }
        for (; pir <= L && al0 + pis - win_start0 < win_len; pir++, pis++) {
            if (al0 + pis - win_start0 < 0) { bad = 1; break; }
            b[pir - 1] = win[al0 + pis - win_start0];
        }
        if (bad) { status[r] = -1; continue; }
        for (int64_t f = feat_off[r]; f < feat_off[r + 1]; f++)
            if (f_op[f] == 'B') {
                if (f_pos[f] < 1 || f_pos[f] > L) { bad = 1; break; }
                b[f_pos[f] - 1] = (uint8_t)f_a[f];
            }
        if (bad) { status[r] = -1; continue; }
        for (int i = 0; i < L; i++) b[i] = normalize_base(b[i]);
    }
}

/*
 * N2: CramNormalizer.restoreQualityScores.  scores (in/out) at out_off[r]; for force-preserved reads the
 * array already holds the decoded scores (0xFF = missing).  status 0 = scores valid, 1 = NULL_QUALS.
 */
void orc_restore_quals(int32_t n_reads, const int32_t* read_len, const uint8_t* rflags, const int64_t* feat_off,
                       const uint8_t* f_op, const int32_t* f_pos, const int32_t* f_a, const int32_t* f_b,
                       uint8_t default_qs, uint8_t* scores, const int64_t* out_off, int32_t* status) {
    for (int32_t r = 0; r < n_reads; r++) {
        if (rflags[r] & RF_SKIP) { status[r] = 2; continue; }
        int L = read_len[r];
        uint8_t* s = scores + out_off[r];
        if (!(rflags[r] & RF_FORCE_QS)) {
            int star = 1;
            memset(s, default_qs, (size_t)L);
            for (int64_t f = feat_off[r]; f < feat_off[r + 1]; f++) {
                if (f_op[f] == 'Q') { s[f_pos[f] - 1] = (uint8_t)f_a[f]; star = 0; }
                else if (f_op[f] == 'B') { s[f_pos[f] - 1] = (uint8_t)f_b[f]; star = 0; }
            }
            status[r] = star ? 1 : 0;
        } else {
            int missing = 0;
            for (int i = 0; i < L; i++)
                if (s[i] == 0xFF) { s[i] = default_qs; missing++; }
            status[r] = (missing == L) ? 1 : 0;
        }
    }
}

/* ---- Q2 ---- */
typedef struct {
    int32_t read_cat;   /* 0 none, 1 UNPLACED, 2 HIGHER_MAPPING_SCORE, 3 LOWER_MAPPING_SCORE, 4 ALL */
    int32_t read_param;
    int32_t treatment;  /* 0 PRESERVE, 1 BIN, 2 DROP */
    int32_t n_base_cat;
    int32_t base_cat[8]; /* 1 MATCH, 2 MISMATCH, 3 FLANKING_DELETION, 4 PILEUP, 5 LOWER_COVERAGE, 6 INSERTION */
    int32_t base_param[8];
} orc_policy_t;

static const uint8_t ILLUMINA8[110] = {
    0,  1,  6,  6,  6,  6,  6,  6,  6,  6,  15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 22, 22, 22, 22, 22, 27, 27, 27,
    27, 27, 33, 33, 33, 33, 33, 37, 37, 37, 37, 37, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40,
    40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40,
    40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40};

static int read_gate(const orc_policy_t* p, int unmapped, int mapq) {
    switch (p->read_cat) {
        case 0: return 1;
        case 4: return 1;
        case 1: return unmapped;
        case 3: return mapq < p->read_param;
        case 2: return mapq > p->read_param;
        default: return 0;
    }
}

/*
 * ReferenceTracks window emulation (ReferenceTracks.java:87-133): the 1-based window position seen by each
 * read when addQualityScores runs over the reads in order.  wpos[r] = 0 when ensureRange is never reached.
 * Returns 0, or -(r+1) if Java would throw at read r.
 */
int32_t orc_tracks_windows(int32_t n_reads, const int32_t* aln_start, const int32_t* aln_span, const uint8_t* rflags,
                           const uint8_t* mapq, const orc_policy_t* pol, int32_t n_pol, int64_t ref_len,
                           int32_t window, int32_t* wpos) {
    int64_t W = window < ref_len ? window : ref_len;
    int64_t position = 1;
    for (int32_t r = 0; r < n_reads; r++) {
        wpos[r] = 0;
        if (rflags[r] & (RF_NULL_QUALS | RF_SKIP)) continue;
        int need = 0;
        for (int k = 0; k < n_pol; k++)
            if (pol[k].n_base_cat > 0 && read_gate(&pol[k], (rflags[r] & RF_UNMAPPED) != 0, mapq[r])) need = 1;
        if (!need || aln_start[r] == 0) continue;
        int64_t start = aln_start[r];
        if (start < position) return -(r + 1);
        if (start + aln_span[r] > position + W) {
            int64_t np = start;
            if (np - 1 >= ref_len) return -(r + 1);
            if (np > ref_len - W + 1) np = ref_len - W + 1;
            position = np;
        }
        wpos[r] = (int32_t)position;
    }
    return 0;
}

/*
 * Q2: QualityScorePreservation.addQualityScores over one container.
 * scores_out[base_off[r] .. +L) receives the score array (0xFF = -1 hole); force[r] = force-preserve flag;
 * status[r]: 0 ok, 1 NULL_QUALS in -> NULL_QUALS out, 2 skipped, -1 Java would throw.
 * The coverage / mismatch pile-up tracks are out of scope: LOWER_COVERAGE and PILEUP
 * are evaluated against all-zero tracks, i.e. a freshly constructed ReferenceTracks (the state KAT-3 was
 * minted with): coverageAt() == 0 and mismatchesAt() == 0 everywhere.
 */
int32_t orc_lossy_quals(int32_t n_reads, const uint8_t* bases, const uint8_t* quals, const int64_t* base_off,
                        const int32_t* read_len, const int32_t* aln_start, const uint8_t* rflags,
                        const uint8_t* mapq, const uint32_t* cigar, const int64_t* cigar_off, const int32_t* n_cigar,
                        const uint8_t* ref, int64_t ref_len, const int32_t* wpos, int32_t window,
                        const orc_policy_t* pol, int32_t n_pol, uint8_t* scores_out, uint8_t* force,
                        int32_t* status) {
    int64_t W = window < ref_len ? window : ref_len;
    for (int32_t r = 0; r < n_reads; r++) {
        force[r] = 0;
        if (rflags[r] & RF_SKIP) { status[r] = 2; continue; }
        if (rflags[r] & RF_NULL_QUALS) { status[r] = 1; continue; }
        int L = read_len[r];
        const uint8_t* qs = quals + base_off[r];
        const uint8_t* rb = bases + base_off[r];
        uint8_t* sc = scores_out + base_off[r];
        memset(sc, 0xFF, (size_t)L);
        uint8_t* mask = (uint8_t*)malloc((size_t)L + 1);
        int bad = 0;
        int frc = 0;
        for (int k = 0; k < n_pol && !bad; k++) {
            const orc_policy_t* p = &pol[k];
            if (!read_gate(p, (rflags[r] & RF_UNMAPPED) != 0, mapq[r])) continue;
            if (p->n_base_cat == 0) {
                if (p->treatment == 1) {
                    for (int i = 0; i < L; i++) {
                        if (qs[i] >= 110) { bad = 1; break; }
                        sc[i] = ILLUMINA8[qs[i]];
                    }
                    frc = 1;
                } else if (p->treatment == 0) {
                    memcpy(sc, qs, (size_t)L);
                    frc = 1;
                } else {
                    frc = 0;
                }
                continue;
            }
            if (aln_start[r] == 0) continue;
            memset(mask, 0, (size_t)L + 1);
            for (int c = 0; c < p->n_base_cat && !bad; c++) {
                int type = p->base_cat[c];
                int pos = 0;
                int64_t refpos = aln_start[r];
                for (int e = 0; e < n_cigar[r] && !bad; e++) {
                    uint32_t ce = cigar[cigar_off[r] + e];
                    int op = (int)(ce & 15), len = (int)(ce >> 4);
                    if (type == 3) {
                        if (op == OP_D) {
                            if (pos >= L) { bad = 1; break; }
                            mask[pos] = 1;
                            if (pos + 1 < L) mask[pos + 1] = 1;
                        }
                    } else if (type == 1 || type == 2) {
                        if (consumes_read(op))
                            for (int i = 0; i < len; i++) {
                                if (pos + i >= L) { bad = 1; break; }
                                int64_t rp = refpos + i;
                                if (rp - wpos[r] < 0) { bad = 1; break; }
                                uint8_t tb = (rp - wpos[r] < W) ? ref[rp - 1] : 'N';
                                int match = rb[pos + i] == tb;
                                mask[pos + i] = (type == 1 && match) || (type == 2 && !match);
                            }
                    } else if (type == 6) {
                        if (op == OP_I)
                            for (int i = 0; i < len; i++) {
                                if (pos + i >= L) { bad = 1; break; }
                                mask[pos + i] = 1;
                            }
                    } else if (type == 5) {
                        if (op == OP_M || op == OP_EQ || op == OP_X)
                            for (int i = 0; i < len; i++) {
                                if (pos + i >= L) { bad = 1; break; }
                                mask[pos + i] = 0 < p->base_param[c]; /* coverageAt() == 0 */
                            }
                    }
                    if (consumes_read(op)) pos += len;
                    if (consumes_ref(op)) refpos += len;
                }
                if (type == 4 && 0 > p->base_param[c]) /* mismatchesAt() == 0 */
                    for (int i = 0; i < L; i++) mask[i] = 1;
            }
            if (bad) break;
            int masked = 0;
            for (int i = 0; i < L; i++)
                if (mask[i]) {
                    if (p->treatment == 1) {
                        if (qs[i] >= 110) { bad = 1; break; }
                        sc[i] = ILLUMINA8[qs[i]];
                    } else if (p->treatment == 2) {
                        sc[i] = 0xFF;
                    } else {
                        sc[i] = qs[i];
                    }
                    masked++;
                }
            if (masked > L / 2) frc = 1;
        }
        free(mask);
        force[r] = (uint8_t)frc;
        status[r] = bad ? -1 : 0;
    }
    return 0;
}

/*
 * Tail of addQualityScores (QualityScorePreservation.java:288-299): for reads that are not force-preserved,
 * append BaseQualityScore(i+1, scores[i]) for every scores[i] > -1 and stable-sort by position (original
 * features first at equal positions).  Writes merged features; m_off has n_reads+1 entries.
 * Returns total merged count or -1 when cap is too small.
 */
int64_t orc_merge_quality_features(int32_t n_reads, const int32_t* read_len, const int64_t* base_off,
                                   const uint8_t* scores, const uint8_t* force, const int32_t* qstatus,
                                   const int64_t* feat_off, const uint8_t* f_op, const int32_t* f_pos,
                                   const int32_t* f_a, const int32_t* f_b, int64_t cap, uint8_t* m_op,
                                   int32_t* m_pos, int32_t* m_a, int32_t* m_b, int64_t* m_off) {
    int64_t n = 0;
    for (int32_t r = 0; r < n_reads; r++) {
        m_off[r] = n;
        int64_t f = feat_off[r], fe = feat_off[r + 1];
        int addq = (qstatus[r] == 0) && !force[r];
        int L = read_len[r];
        const uint8_t* sc = scores + base_off[r];
        for (int i = 0; addq && i < L; i++) {
            if (sc[i] == 0xFF || sc[i] >= 128) continue; /* scores[i] > -1 as a signed byte */
            while (f < fe && f_pos[f] <= i + 1) {
                if (n >= cap) return -1;
                m_op[n] = f_op[f]; m_pos[n] = f_pos[f]; m_a[n] = f_a[f]; m_b[n] = f_b[f];
                n++; f++;
            }
            if (n >= cap) return -1;
            m_op[n] = 'Q'; m_pos[n] = i + 1; m_a[n] = sc[i]; m_b[n] = 0;
            n++;
        }
        while (f < fe) {
            if (n >= cap) return -1;
            m_op[n] = f_op[f]; m_pos[n] = f_pos[f]; m_a[n] = f_a[f]; m_b[n] = f_b[f];
            n++; f++;
        }
    }
    m_off[n_reads] = n;
    return n;
}
