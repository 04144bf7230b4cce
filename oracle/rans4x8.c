/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * Plain-C CPU restatement of the CRAM 3.0 "rANS 4x8" codec exactly as htsjdk 2.4.1
 * (com.github.samtools:htsjdk:2.4.1, vendored as bytecode in the reference's
 * lib/htsjdk-2.4.1.jar, pinned by build.gradle:33) implements it.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Parity pin: every vector of SURVEY.md §8(c) KAT-1 (true outputs of
 * ExternalCompression.rans from the htsjdk-2.4.1 bytecode) is checked in
 * tests/test_oracle_rans.py against tests/golden/rans_kat1.json.
 *
 * Reference locations (htsjdk!pkg/Class#method, original source line numbers):
 *   RANS#compress L60-73, #uncompress L29-56, #encode_order0_way4 L90-106,
 *   #encode_order1_way4 L111-127, #finalizeCompressed L132-142
 *   Frequencies#calcFrequencies_o0 L109-145, #calcFrequencies_o1 L149-193,
 *   #writeFrequencies_o0 L215-244, #writeFrequencies_o1 L269-328,
 *   #readStats_o0 L13-46, #readStats_o1 L50-106, #buildSyms_o0/o1 L197-211/248-265
 *   Encoding#RansEncSymbolInit L21-43, #RansEncPutSymbol L46-71
 *   E04#compress L12-57, E14#compress L13-86, D04#uncompress L9-85, D14#uncompress L9-70
 *   Utils#reverse L7-33
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TF_SHIFT 12
#define TOTFREQ 4096
#define RANS_BYTE_L (1u << 23)

#define ORC_E_INPUT_LENGTH (-1)   /* "Incorrect input length." */
#define ORC_E_OUT_TOO_SMALL (-2)  /* "Output buffer too small to fit N bytes." */
#define ORC_E_INSUFFICIENT (-3)   /* "Insufficient buffer size." */
#define ORC_E_ORDER (-4)          /* "Unknown rANS order: X" */
#define ORC_E_CORRUPT (-5)        /* Java would raise an index/underflow exception */

typedef struct {
    uint32_t x_max, rcp_freq, bias, cmpl_freq, rcp_shift;
} enc_sym_t;

/* Encoding#RansEncSymbolInit L21-43 */
static void enc_sym_init(enc_sym_t* s, uint32_t start, uint32_t freq) {
    s->x_max = ((RANS_BYTE_L >> TF_SHIFT) << 8) * freq;
    s->cmpl_freq = (1u << TF_SHIFT) - freq;
    if (freq < 2) {
        s->rcp_freq = ~0u;
        s->rcp_shift = 0;
        s->bias = start + (1u << TF_SHIFT) - 1;
    } else {
        uint32_t shift = 0;
        while ((uint64_t)freq > (1ull << shift)) shift++;
        s->rcp_freq = (uint32_t)(((1ull << (shift + 31)) + freq - 1) / freq);
        s->rcp_shift = shift - 1;
        s->bias = start;
    }
    s->rcp_shift += 32;
}

/* Encoding#RansEncPutSymbol L46-71: bytes go forward into buf, low byte first */
static inline uint32_t enc_put(uint32_t r, uint8_t* buf, int64_t* pos, const enc_sym_t* s) {
    uint32_t x = r;
    if (x >= s->x_max) {
        buf[(*pos)++] = (uint8_t)(x & 0xff);
        x >>= 8;
        if (x >= s->x_max) {
            buf[(*pos)++] = (uint8_t)(x & 0xff);
            x >>= 8;
        }
    }
    uint32_t q = (uint32_t)(((uint64_t)x * (uint64_t)s->rcp_freq) >> s->rcp_shift);
    return x + s->bias + q * s->cmpl_freq;
}

static void put_be32(uint8_t* buf, int64_t* pos, uint32_t v) {
    buf[(*pos)++] = (uint8_t)(v >> 24);
    buf[(*pos)++] = (uint8_t)(v >> 16);
    buf[(*pos)++] = (uint8_t)(v >> 8);
    buf[(*pos)++] = (uint8_t)v;
}

static void reverse_bytes(uint8_t* b, int64_t n) {
    for (int64_t i = 0, j = n - 1; i < j; i++, j--) {
        uint8_t t = b[i];
        b[i] = b[j];
        b[j] = t;
    }
}

/* Frequencies#calcFrequencies_o0 L109-145 */
static void calc_freq_o0(const uint8_t* in, int n, int* F) {
    memset(F, 0, 256 * sizeof(int));
    for (int i = 0; i < n; i++) F[in[i]]++;
    int T = n;
    uint64_t tr = ((uint64_t)TOTFREQ << 31) / (uint64_t)T + (uint64_t)((1 << 30) / T);
    int m = 0, M = 0;
    for (int j = 0; j < 256; j++)
        if (m < F[j]) {
            m = F[j];
            M = j;
        }
    int fsum = 0;
    for (int j = 0; j < 256; j++) {
        if (F[j] == 0) continue;
        F[j] = (int)(((uint64_t)F[j] * tr) >> 31);
        if (F[j] == 0) F[j] = 1;
        fsum += F[j];
    }
    fsum++;
    if (fsum < TOTFREQ)
        F[M] += TOTFREQ - fsum;
    else
        F[M] -= fsum - TOTFREQ;
}

/* Frequencies#calcFrequencies_o1 L149-193; F is int[256][256], T int[256] */
static void calc_freq_o1(const uint8_t* in, int n, int (*F)[256], int* T) {
    memset(F, 0, 256 * 256 * sizeof(int));
    memset(T, 0, 256 * sizeof(int));
    int last = 0;
    for (int i = 0; i < n; i++) {
        int c = in[i];
        F[last][c]++;
        T[last]++;
        last = c;
    }
    int q = n >> 2;
    F[0][in[1 * q]]++;
    F[0][in[2 * q]]++;
    F[0][in[3 * q]]++;
    T[0] += 3;
    for (int i = 0; i < 256; i++) {
        if (T[i] == 0) continue;
        double p = ((double)TOTFREQ) / (double)T[i];
        int t2 = 0, m = 0, M = 0;
        for (int j = 0; j < 256; j++) {
            if (F[i][j] == 0) continue;
            if (m < F[i][j]) {
                m = F[i][j];
                M = j;
            }
            F[i][j] = (int)((double)F[i][j] * p);
            if (F[i][j] == 0) F[i][j] = 1;
            t2 += F[i][j];
        }
        t2++;
        if (t2 < TOTFREQ)
            F[i][M] += TOTFREQ - t2;
        else
            F[i][M] -= t2 - TOTFREQ;
    }
}

/* Frequencies#writeFrequencies_o0 L215-244 (also the per-context row writer of o1) */
static int64_t write_freq_row(uint8_t* cp, const int* F) {
    int64_t p = 0;
    int rle = 0;
    for (int j = 0; j < 256; j++) {
        if (F[j] == 0) continue;
        if (rle != 0) {
            rle--;
        } else {
            cp[p++] = (uint8_t)j;
            if (j > 0 && F[j - 1] != 0) {
                for (rle = j + 1; rle < 256 && F[rle] != 0; rle++)
                    ;
                rle -= j + 1;
                cp[p++] = (uint8_t)rle;
            }
        }
        if (F[j] < 128) {
            cp[p++] = (uint8_t)F[j];
        } else {
            cp[p++] = (uint8_t)(128 | (F[j] >> 8));
            cp[p++] = (uint8_t)(F[j] & 0xff);
        }
    }
    cp[p++] = 0;
    return p;
}

/* Frequencies#writeFrequencies_o1 L269-328 */
static int64_t write_freq_o1(uint8_t* cp, int (*F)[256]) {
    int T[256];
    for (int i = 0; i < 256; i++) {
        int s = 0;
        for (int j = 0; j < 256; j++) s += F[i][j];
        T[i] = s;
    }
    int64_t p = 0;
    int rle_i = 0;
    for (int i = 0; i < 256; i++) {
        if (T[i] == 0) continue;
        if (rle_i != 0) {
            rle_i--;
        } else {
            cp[p++] = (uint8_t)i;
            if (i > 0 && T[i - 1] != 0) {
                for (rle_i = i + 1; rle_i < 256 && T[rle_i] != 0; rle_i++)
                    ;
                rle_i -= i + 1;
                cp[p++] = (uint8_t)rle_i;
            }
        }
        p += write_freq_row(cp + p, F[i]);
    }
    cp[p++] = 0;
    return p;
}

/* RANS#allocateIfNeeded L79-85: (int)(1.05*n + 257*257*3 + 4) */
int32_t orc_rans_bound(int32_t n) { return (int32_t)(1.05 * (double)n + 257.0 * 257.0 * 3.0 + 4.0); }

/* E04#compress L12-57; payload written at cp, returns payload length */
static int64_t e04(const uint8_t* in, int n, const enc_sym_t* syms, uint8_t* cp) {
    uint32_t r0 = RANS_BYTE_L, r1 = RANS_BYTE_L, r2 = RANS_BYTE_L, r3 = RANS_BYTE_L;
    int64_t p = 0;
    switch (n & 3) {
        case 3:
            r2 = enc_put(r2, cp, &p, &syms[in[n - 1]]);
            r1 = enc_put(r1, cp, &p, &syms[in[n - 2]]);
            r0 = enc_put(r0, cp, &p, &syms[in[n - 3]]);
            break;
        case 2:
            r1 = enc_put(r1, cp, &p, &syms[in[n - 1]]);
            r0 = enc_put(r0, cp, &p, &syms[in[n - 2]]);
            break;
        case 1:
            r0 = enc_put(r0, cp, &p, &syms[in[n - 1]]);
            break;
        default:
            break;
    }
    for (int i = (n & ~3); i > 0; i -= 4) {
        r3 = enc_put(r3, cp, &p, &syms[in[i - 1]]);
        r2 = enc_put(r2, cp, &p, &syms[in[i - 2]]);
        r1 = enc_put(r1, cp, &p, &syms[in[i - 3]]);
        r0 = enc_put(r0, cp, &p, &syms[in[i - 4]]);
    }
    put_be32(cp, &p, r3);
    put_be32(cp, &p, r2);
    put_be32(cp, &p, r1);
    put_be32(cp, &p, r0);
    reverse_bytes(cp, p);
    return p;
}

/* E14#compress L13-86; syms[ctx][sym], ctx = preceding symbol */
static int64_t e14(const uint8_t* in, int n, enc_sym_t (*syms)[256], uint8_t* cp) {
    uint32_t r0 = RANS_BYTE_L, r1 = RANS_BYTE_L, r2 = RANS_BYTE_L, r3 = RANS_BYTE_L;
    int64_t p = 0;
    int q = n >> 2;
    int i0 = 1 * q - 2, i1 = 2 * q - 2, i2 = 3 * q - 2, i3;
    int l0 = (i0 + 1 >= 0) ? in[i0 + 1] : 0;
    int l1 = (i1 + 1 >= 0) ? in[i1 + 1] : 0;
    int l2 = (i2 + 1 >= 0) ? in[i2 + 1] : 0;
    int l3 = in[n - 1];
    for (i3 = n - 2; i3 > 4 * q - 2 && i3 >= 0; i3--) {
        int c3 = in[i3];
        r3 = enc_put(r3, cp, &p, &syms[c3][l3]);
        l3 = c3;
    }
    for (; i0 >= 0; i0--, i1--, i2--, i3--) {
        int c0 = in[i0], c1 = in[i1], c2 = in[i2], c3 = in[i3];
        r3 = enc_put(r3, cp, &p, &syms[c3][l3]);
        r2 = enc_put(r2, cp, &p, &syms[c2][l2]);
        r1 = enc_put(r1, cp, &p, &syms[c1][l1]);
        r0 = enc_put(r0, cp, &p, &syms[c0][l0]);
        l0 = c0;
        l1 = c1;
        l2 = c2;
        l3 = c3;
    }
    r3 = enc_put(r3, cp, &p, &syms[0][l3]);
    r2 = enc_put(r2, cp, &p, &syms[0][l2]);
    r1 = enc_put(r1, cp, &p, &syms[0][l1]);
    r0 = enc_put(r0, cp, &p, &syms[0][l0]);
    put_be32(cp, &p, r3);
    put_be32(cp, &p, r2);
    put_be32(cp, &p, r1);
    put_be32(cp, &p, r0);
    reverse_bytes(cp, p);
    return p;
}

static void put_le32(uint8_t* p, uint32_t v) {
    p[0] = (uint8_t)v;
    p[1] = (uint8_t)(v >> 8);
    p[2] = (uint8_t)(v >> 16);
    p[3] = (uint8_t)(v >> 24);
}
static uint32_t get_le32(const uint8_t* p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

/* Per-thread workspace of the multi-block drivers: the order-1 tables (0.75 MB) and the staging buffer are allocated
 * once per thread, not once per block (bench.py cpu_baseline). */
typedef struct {
    int (*F)[256];
    enc_sym_t (*syms)[256];
    uint8_t* scratch;
    int64_t scratch_cap;
    void* D[256]; /* dec_tab_t*, kept across blocks */
} orc_ws;

static void ws_free(orc_ws* ws) {
    free(ws->F);
    free(ws->syms);
    free(ws->scratch);
    for (int k = 0; k < 256; k++) free(ws->D[k]);
    memset(ws, 0, sizeof(*ws));
}

/*
 * RANS#compress L60-73.  Returns the number of bytes written to out (0 for n == 0),
 * or a negative ORC_E_* code.  out must hold orc_rans_bound(n) bytes; a payload that
 * would overflow it (Java: BufferOverflowException) gives ORC_E_INSUFFICIENT.
 */
static int32_t rans_compress_ws(orc_ws* ws, const uint8_t* in, int32_t n, int32_t order, uint8_t* out, int32_t cap) {
    if (order != 0 && order != 1) return ORC_E_ORDER;
    if (n == 0) return 0;
    if (n < 4) order = 0;
    int32_t bound = orc_rans_bound(n);
    /* worst case the encoder can physically emit: 2 bytes/symbol + tables + states */
    int64_t scratch_sz = (int64_t)n * 2 + 257 * 257 * 3 + 64;
    if (ws->scratch_cap < scratch_sz) {
        free(ws->scratch);
        ws->scratch = (uint8_t*)malloc((size_t)scratch_sz);
        ws->scratch_cap = ws->scratch ? scratch_sz : 0;
    }
    uint8_t* scratch = ws->scratch;
    if (!scratch) return ORC_E_INSUFFICIENT;
    int64_t tab_len, pay_len;
    if (order == 0) {
        int F[256];
        enc_sym_t syms[256];
        calc_freq_o0(in, n, F);
        int C = 0;
        for (int j = 0; j < 256; j++) {
            if (F[j] != 0) {
                enc_sym_init(&syms[j], (uint32_t)C, (uint32_t)F[j]);
                C += F[j];
            }
        }
        tab_len = write_freq_row(scratch + 9, F);
        pay_len = e04(in, n, syms, scratch + 9 + tab_len);
    } else {
        if (!ws->F) ws->F = (int(*)[256])malloc(256 * 256 * sizeof(int));
        if (!ws->syms) ws->syms = (enc_sym_t(*)[256])malloc(256 * 256 * sizeof(enc_sym_t));
        int(*F)[256] = ws->F;
        enc_sym_t(*syms)[256] = ws->syms;
        if (!F || !syms) return ORC_E_INSUFFICIENT;
        int T[256];
        calc_freq_o1(in, n, F, T);
        for (int i = 0; i < 256; i++) {
            int C = 0;
            for (int j = 0; j < 256; j++) {
                if (F[i][j] != 0) {
                    enc_sym_init(&syms[i][j], (uint32_t)C, (uint32_t)F[i][j]);
                    C += F[i][j];
                }
            }
        }
        tab_len = write_freq_o1(scratch + 9, F);
        pay_len = e14(in, n, syms, scratch + 9 + tab_len);
    }
    int64_t total = 9 + tab_len + pay_len;
    if (total > bound || total > cap) return ORC_E_INSUFFICIENT;
    scratch[0] = (uint8_t)order;
    put_le32(scratch + 1, (uint32_t)(tab_len + pay_len));
    put_le32(scratch + 5, (uint32_t)n);
    memcpy(out, scratch, (size_t)total);
    return (int32_t)total;
}

int32_t orc_rans_compress(const uint8_t* in, int32_t n, int32_t order, uint8_t* out, int32_t cap) {
    orc_ws ws;
    memset(&ws, 0, sizeof(ws));
    int32_t r = rans_compress_ws(&ws, in, n, order, out, cap);
    ws_free(&ws);
    return r;
}

/* ---- decode ---- */
typedef struct {
    uint16_t F[256], C[256];
    uint8_t R[TOTFREQ];
} dec_tab_t;

/* Frequencies#readStats_o0 L13-46 (row reader shared with o1; o1 maps F==0 -> 4096) */
static int read_freq_row(const uint8_t* cp, int64_t* pp, int64_t end, dec_tab_t* D, int o1) {
    int64_t p = *pp;
    int rle = 0, x = 0;
    if (p >= end) return -1;
    int j = cp[p++];
    do {
        if (p >= end) return -1;
        int F = cp[p++];
        if (F >= 128) {
            if (p >= end) return -1;
            F = ((F & 127) << 8) | cp[p++];
        }
        if (o1 && F == 0) F = TOTFREQ;
        if (x + F > TOTFREQ) return -1;
        D->F[j] = (uint16_t)F;
        D->C[j] = (uint16_t)x;
        memset(D->R + x, j, (size_t)F);
        x += F;
        if (p >= end) return -1;
        if (rle == 0 && j + 1 == cp[p]) {
            j = cp[p++];
            if (p >= end) return -1;
            rle = cp[p++];
        } else if (rle != 0) {
            rle--;
            j++;
            if (j > 255) return -1;
        } else {
            j = cp[p++];
        }
    } while (j != 0);
    *pp = p;
    return 0;
}

#define RENORM(x)                                   \
    while ((x) < RANS_BYTE_L) {                     \
        if (p >= end) return ORC_E_CORRUPT;         \
        (x) = ((x) << 8) | cp[p++];                 \
    }

/* D04#uncompress L9-85 */
static int d04(const uint8_t* cp, int64_t p, int64_t end, const dec_tab_t* D, uint8_t* out, int n) {
    if (p + 16 > end) return ORC_E_CORRUPT;
    uint32_t r0 = get_le32(cp + p), r1 = get_le32(cp + p + 4), r2 = get_le32(cp + p + 8), r3 = get_le32(cp + p + 12);
    p += 16;
    int out_end = n & ~3;
    for (int i = 0; i < out_end; i += 4) {
        uint8_t c0 = D->R[r0 & 4095], c1 = D->R[r1 & 4095], c2 = D->R[r2 & 4095], c3 = D->R[r3 & 4095];
        out[i] = c0;
        out[i + 1] = c1;
        out[i + 2] = c2;
        out[i + 3] = c3;
        r0 = D->F[c0] * (r0 >> 12) + (r0 & 4095) - D->C[c0];
        r1 = D->F[c1] * (r1 >> 12) + (r1 & 4095) - D->C[c1];
        r2 = D->F[c2] * (r2 >> 12) + (r2 & 4095) - D->C[c2];
        r3 = D->F[c3] * (r3 >> 12) + (r3 & 4095) - D->C[c3];
        RENORM(r0);
        RENORM(r1);
        RENORM(r2);
        RENORM(r3);
    }
    switch (n & 3) {
        case 3:
            out[out_end + 2] = D->R[r2 & 4095]; /* fallthrough */
        case 2:
            out[out_end + 1] = D->R[r1 & 4095]; /* fallthrough */
        case 1:
            out[out_end] = D->R[r0 & 4095];
        default:
            break;
    }
    return 0;
}

/* D14#uncompress L9-70 */
static int d14(const uint8_t* cp, int64_t p, int64_t end, dec_tab_t** D, uint8_t* out, int n) {
    if (p + 16 > end) return ORC_E_CORRUPT;
    uint32_t r0 = get_le32(cp + p), r1 = get_le32(cp + p + 4), r2 = get_le32(cp + p + 8), r3 = get_le32(cp + p + 12);
    p += 16;
    int q = n >> 2;
    int i0 = 0, i1 = q, i2 = 2 * q, i3 = 3 * q;
    int l0 = 0, l1 = 0, l2 = 0, l3 = 0;
#define STEP(r, l, c)                                                      \
    {                                                                      \
        if (!D[l]) return ORC_E_CORRUPT;                                   \
        c = D[l]->R[r & 4095];                                             \
        r = D[l]->F[c] * (r >> 12) + (r & 4095) - D[l]->C[c];              \
    }
    for (; i0 < q; i0++, i1++, i2++, i3++) {
        int c0, c1, c2, c3;
        STEP(r0, l0, c0);
        STEP(r1, l1, c1);
        STEP(r2, l2, c2);
        STEP(r3, l3, c3);
        out[i0] = (uint8_t)c0;
        out[i1] = (uint8_t)c1;
        out[i2] = (uint8_t)c2;
        out[i3] = (uint8_t)c3;
        RENORM(r0);
        RENORM(r1);
        RENORM(r2);
        RENORM(r3);
        l0 = c0;
        l1 = c1;
        l2 = c2;
        l3 = c3;
    }
    for (; i3 < n; i3++) {
        int c3;
        STEP(r3, l3, c3);
        out[i3] = (uint8_t)c3;
        RENORM(r3);
        l3 = c3;
    }
#undef STEP
    return 0;
}

/* LE32 raw size at offset 5 (RANS#uncompress L29-56 header layout); 0 for an empty stream */
int32_t orc_rans_raw_size(const uint8_t* in, int32_t n) {
    if (n == 0) return 0;
    if (n < 9) return ORC_E_INPUT_LENGTH;
    return (int32_t)get_le32(in + 5);
}

/* RANS#uncompress L29-56. Returns raw size, or negative ORC_E_*. */
static int32_t rans_uncompress_ws(orc_ws* ws, const uint8_t* in, int32_t n, uint8_t* out, int32_t cap) {
    if (n == 0) return 0;
    if (n < 9) return ORC_E_INPUT_LENGTH;
    int order = in[0];
    if (order != 0 && order != 1) return ORC_E_ORDER;
    uint32_t comp = get_le32(in + 1);
    if ((int64_t)comp != (int64_t)n - 1 - 4 - 4) return ORC_E_INPUT_LENGTH;
    int32_t raw = (int32_t)get_le32(in + 5);
    if (raw < 0) return ORC_E_CORRUPT;
    if (raw > cap) return ORC_E_OUT_TOO_SMALL;
    if (raw == 0) return 0;
    int64_t p = 9, end = n;
    int rc;
    if (order == 0) {
        if (!ws->D[0]) ws->D[0] = calloc(1, sizeof(dec_tab_t));
        dec_tab_t* D = (dec_tab_t*)ws->D[0];
        memset(D, 0, sizeof(*D));
        if (read_freq_row(in, &p, end, D, 0) != 0) return ORC_E_CORRUPT;
        rc = d04(in, p, end, D, out, raw);
    } else {
        dec_tab_t* D[256]; /* the contexts this block defines (a workspace table is cleared when a block takes it) */
        memset(D, 0, sizeof(D));
        rc = 0;
        /* Frequencies#readStats_o1 L50-106 */
        int rle_i = 0;
        if (p >= end) rc = ORC_E_CORRUPT;
        int i = rc ? 0 : in[p++];
        if (!rc) do {
                if (!D[i]) {
                    if (!ws->D[i]) ws->D[i] = calloc(1, sizeof(dec_tab_t));
                    D[i] = (dec_tab_t*)ws->D[i];
                    memset(D[i], 0, sizeof(dec_tab_t));
                }
                if (read_freq_row(in, &p, end, D[i], 1) != 0 || p >= end) {
                    rc = ORC_E_CORRUPT;
                    break;
                }
                if (rle_i == 0 && i + 1 == in[p]) {
                    i = in[p++];
                    if (p >= end) {
                        rc = ORC_E_CORRUPT;
                        break;
                    }
                    rle_i = in[p++];
                } else if (rle_i != 0) {
                    rle_i--;
                    i++;
                    if (i > 255) {
                        rc = ORC_E_CORRUPT;
                        break;
                    }
                } else {
                    i = in[p++];
                }
            } while (i != 0);
        if (!rc) rc = d14(in, p, end, D, out, raw);
    }
    return rc ? rc : raw;
}

int32_t orc_rans_uncompress(const uint8_t* in, int32_t n, uint8_t* out, int32_t cap) {
    orc_ws ws;
    memset(&ws, 0, sizeof(ws));
    int32_t r = rans_uncompress_ws(&ws, in, n, out, cap);
    ws_free(&ws);
    return r;
}

/* Multi-block drivers of bench.py's cpu_baseline legs: one call = one thread's share of the blocks, one workspace. */
int64_t orc_rans_compress_many(const uint8_t* in, const int64_t* in_off, const int32_t* in_len, const int32_t* order,
                               int32_t nblocks, uint8_t* out, const int64_t* out_off, int32_t* out_len) {
    orc_ws ws;
    memset(&ws, 0, sizeof(ws));
    int64_t total = 0;
    for (int b = 0; b < nblocks; b++) {
        int32_t cap = orc_rans_bound(in_len[b]);
        int32_t r = rans_compress_ws(&ws, in + in_off[b], in_len[b], order[b], out + out_off[b], cap);
        out_len[b] = r;
        if (r < 0) {
            total = r;
            break;
        }
        total += r;
    }
    ws_free(&ws);
    return total;
}

int64_t orc_rans_uncompress_many(const uint8_t* in, const int64_t* in_off, const int32_t* in_len, int32_t nblocks, uint8_t* out,
                                 const int64_t* out_off, const int32_t* out_cap, int32_t* out_len) {
    orc_ws ws;
    memset(&ws, 0, sizeof(ws));
    int64_t total = 0;
    for (int b = 0; b < nblocks; b++) {
        int32_t r = rans_uncompress_ws(&ws, in + in_off[b], in_len[b], out + out_off[b], out_cap[b]);
        out_len[b] = r;
        if (r < 0) {
            total = r;
            break;
        }
        total += r;
    }
    ws_free(&ws);
    return total;
}
