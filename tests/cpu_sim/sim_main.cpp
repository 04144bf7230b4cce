// TEST INFRASTRUCTURE ONLY — runs the kernels of libcramgpu on the CPU (tests/cpu_sim/hip/hip_runtime.h) under
// AddressSanitizer against the CPU oracle.  Usage: sim_main [rans|records|gunzip|bam|all]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <random>
#include <string>
#include <vector>

#include <zlib.h>

#include "../../include/cramgpu.h"

extern "C" {
int32_t orc_rans_bound(int32_t n);
int32_t orc_rans_compress(const uint8_t* in, int32_t n, int32_t order, uint8_t* out, int32_t cap);
int32_t orc_rans_uncompress(const uint8_t* in, int32_t n, uint8_t* out, int32_t cap);
}

typedef std::vector<uint8_t> Bytes;

static Bytes orc(const Bytes& d, int order) {
    Bytes o((size_t)orc_rans_bound((int)d.size()));
    int r = orc_rans_compress(d.data(), (int)d.size(), order, o.data(), (int)o.size());
    if (r < 0) { printf("oracle error %d\n", r); exit(2); }
    o.resize((size_t)r);
    return o;
}

static int check_rans(cramgpu_ctx* c, const std::vector<Bytes>& datas, int order) {
    const int n = (int)datas.size();
    std::vector<cramgpu_block> blocks(n);
    int64_t io = 0, oo = 0;
    for (int i = 0; i < n; i++) {
        int len = (int)datas[i].size(), cap = cramgpu_rans_bound(len);
        blocks[i] = cramgpu_block{io, oo, len, cap, order, 0};
        io += (len + 15) & ~15;
        oo += (cap + 15) & ~15;
    }
    Bytes in((size_t)io + 16), out((size_t)oo + 16);
    for (int i = 0; i < n; i++) memcpy(in.data() + blocks[i].in_off, datas[i].data(), datas[i].size());
    std::vector<int32_t> lens(n);
    int rc = cramgpu_rans_encode_batch(c, in.data(), (int64_t)in.size(), out.data(), (int64_t)out.size(), blocks.data(), n, lens.data());
    if (rc) { printf("encode_batch rc=%d %s\n", rc, cramgpu_last_error()); return 1; }
    std::vector<Bytes> streams(n);
    for (int i = 0; i < n; i++) {
        Bytes want = orc(datas[i], order);
        if (lens[i] != (int)want.size() || memcmp(want.data(), out.data() + blocks[i].out_off, want.size())) {
            printf("MISMATCH encode order %d block %d n=%zu got len %d want %zu\n", order, i, datas[i].size(), lens[i], want.size());
            return 1;
        }
        streams[i] = want;
    }
    // decode
    io = oo = 0;
    for (int i = 0; i < n; i++) {
        blocks[i] = cramgpu_block{io, oo, (int)streams[i].size(), (int)datas[i].size(), 0, 0};
        io += (streams[i].size() + 15) & ~15;
        oo += (datas[i].size() + 15) & ~15;
    }
    Bytes din((size_t)io + 16), dout((size_t)oo + 16);
    for (int i = 0; i < n; i++) memcpy(din.data() + blocks[i].in_off, streams[i].data(), streams[i].size());
    rc = cramgpu_rans_decode_batch(c, din.data(), (int64_t)din.size(), dout.data(), (int64_t)dout.size(), blocks.data(), n, lens.data());
    if (rc) { printf("decode_batch rc=%d %s\n", rc, cramgpu_last_error()); return 1; }
    for (int i = 0; i < n; i++) {
        if (lens[i] != (int)datas[i].size() || memcmp(datas[i].data(), dout.data() + blocks[i].out_off, datas[i].size())) {
            printf("MISMATCH decode order %d block %d n=%zu got %d\n", order, i, datas[i].size(), lens[i]);
            return 1;
        }
    }
    return 0;
}

int run_records(cramgpu_ctx* c);

// ---- gunzip under the sanitizers: every stream and every output slot is a heap block of its exact size, so a read or a
// write one byte outside is caught; valid streams at every level / strategy, then mutated ones (bit flips, truncation,
// garbage) must give zlib's verdict: the same bytes, or an error where zlib reports one
static Bytes gz_make(const Bytes& d, int level, int strategy) {
    z_stream z{};
    deflateInit2(&z, level, Z_DEFLATED, 31, 8, strategy);
    Bytes o(deflateBound(&z, (uLong)d.size()) + 64);
    z.next_in = const_cast<Bytef*>(d.data());
    z.avail_in = (uInt)d.size();
    z.next_out = o.data();
    z.avail_out = (uInt)o.size();
    deflate(&z, Z_FINISH);
    o.resize(z.total_out);
    deflateEnd(&z);
    return o;
}
static int gz_ref(const Bytes& s0, size_t cap, Bytes& out) {  // zlib's own inflate of one member: 0 ok, -1 error, -2 too small
    Bytes s = s0;
    if (s.size() > 3) s[3] &= 0x1F;  // reserved flag bits: zlib refuses them, java.util.zip.GZIPInputStream#readHeader ignores them
    z_stream z{};
    inflateInit2(&z, 31);
    out.assign(cap + 1, 0);
    z.next_in = const_cast<Bytef*>(s.data());
    z.avail_in = (uInt)s.size();
    z.next_out = out.data();
    z.avail_out = (uInt)out.size();
    const int r = inflate(&z, Z_FINISH);
    const size_t got = z.total_out;
    inflateEnd(&z);
    if (r == Z_STREAM_END) {
        if (got > cap) return -2;
        out.resize(got);
        return 0;
    }
    if ((r == Z_BUF_ERROR || r == Z_OK) && z.avail_out == 0) return -2;
    return -1;
}
static int run_gunzip(cramgpu_ctx* c) {
    std::mt19937 rng(11);
    std::vector<Bytes> plain;
    for (int n : {0, 1, 5, 300, 5000, 40000}) {
        Bytes a((size_t)n), b((size_t)n), r((size_t)n);
        for (int i = 0; i < n; i++) a[i] = "ACGT"[rng() & 3], b[i] = (uint8_t)((i / 97) % 3 + 60), r[i] = (uint8_t)rng();
        plain.push_back(a), plain.push_back(b), plain.push_back(r);
    }
    struct Case { Bytes* in; uint8_t* out; size_t cap; int want; Bytes ref; };
    std::vector<Case> cases;
    auto add = [&](const Bytes& s, size_t cap) {
        Case k;
        k.in = new Bytes(s);
        k.in->shrink_to_fit();
        k.cap = cap;
        k.out = new uint8_t[cap ? cap : 1];
        k.want = gz_ref(s, cap, k.ref);
        cases.push_back(std::move(k));
    };
    for (const Bytes& d : plain)
        for (int level : {0, 1, 6, 9})
            for (int strat : {Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY, Z_RLE}) {
                const Bytes s = gz_make(d, level, strat);
                add(s, d.size());
                if (d.size() > 4) add(s, d.size() - 1);  // a slot one byte short
                for (int m = 0; m < 6; m++) {             // mutations
                    Bytes t = s;
                    const int kind = (int)(rng() % 4);
                    if (kind == 0 && t.size() > 12) t[10 + rng() % (t.size() - 10)] ^= (uint8_t)(1u << (rng() & 7));
                    else if (kind == 1) t.resize(rng() % (t.size() + 1));
                    else if (kind == 2 && t.size() > 20) for (int q = 0; q < 8; q++) t[10 + rng() % (t.size() - 10)] = (uint8_t)rng();
                    else if (!t.empty()) t[rng() % t.size()] = (uint8_t)rng();
                    add(t, d.size() + (rng() % 3) * 7);
                }
            }
    const int n = (int)cases.size();
    // offsets relative to the first stream / slot; the kernel only adds them to the base pointers
    std::vector<cramgpu_block> blocks(n);
    const uint8_t* in0 = cases[0].in->data();
    uint8_t* out0 = cases[0].out;
    for (int i = 0; i < n; i++)
        blocks[i] = cramgpu_block{(int64_t)(cases[i].in->data() - in0), (int64_t)(cases[i].out - out0), (int32_t)cases[i].in->size(),
                                  (int32_t)cases[i].cap, 0, 0};
    std::vector<int32_t> lens(n, 12345);
    if (cramgpu_gunzip_dev(c, in0, out0, blocks.data(), n, lens.data(), nullptr) != 0) { printf("gunzip rc %s\n", cramgpu_last_error()); return 1; }
    int bad = 0, ok = 0, refused = 0;
    for (int i = 0; i < n; i++) {
        const Case& k = cases[i];
        if (k.want == 0) {
            if (lens[i] != (int)k.ref.size() || memcmp(k.out, k.ref.data(), k.ref.size())) bad++, printf("gunzip case %d: got %d want %zu bytes\n", i, lens[i], k.ref.size());
            ok++;
        } else {
            // zlib refuses: so must the kernel (too small a slot may surface as either code when the data is also corrupt)
            if (lens[i] >= 0) {
                bad++;
                printf("gunzip case %d: zlib %d, kernel accepted %d bytes; stream:", i, k.want, lens[i]);
                for (size_t q = 0; q < k.in->size() && q < 40; q++) printf(" %02x", (*k.in)[q]);
                printf("\n");
            }
            refused++;
        }
    }
    for (Case& k : cases) {
        delete k.in;
        delete[] k.out;
    }
    printf("gunzip: %s (%d streams: %d valid, %d refused)\n", bad ? "FAILED" : "ok", n, ok, refused);
    return bad != 0;
}

// ---- gzip written by the kernel, under the sanitizers: inputs and output slots are heap blocks of their exact sizes;
// zlib's inflate and the gunzip kernel must both return the input; slots one byte short must be refused
static int run_gzip(cramgpu_ctx* c) {
    std::mt19937 rng(17);
    std::vector<Bytes> plain;
    for (int n : {0, 1, 2, 3, 4, 5, 9, 300, 5000, 32768, 32769, 70001}) {
        Bytes a((size_t)n), b((size_t)n), r((size_t)n), t((size_t)n);
        for (int i = 0; i < n; i++) {
            a[i] = "ACGT"[rng() & 3], b[i] = (uint8_t)((i / 97) % 3 + 60), r[i] = (uint8_t)rng();
            t[i] = (uint8_t)("the quick brown fox jumps over the lazy dog "[(i + (rng() % 50 == 0)) % 44]);
        }
        plain.push_back(a), plain.push_back(b), plain.push_back(r), plain.push_back(t);
    }
    {   // Fibonacci counts: an optimal code deeper than 15 bits, so the length limit has to act (literals, and the
        // code-length code's 7 bits through the spread of lengths this produces)
        Bytes f;
        uint32_t x = 1, y = 1;
        for (int sym = 0; sym < 21; sym++) {
            for (uint32_t k = 0; k < x; k++) f.push_back((uint8_t)(33 + sym * 7));
            const uint32_t z = x + y;
            x = y, y = z;
        }
        std::shuffle(f.begin(), f.end(), rng);
        plain.push_back(f);
        Bytes zeros(66000, 0);
        plain.push_back(zeros);
    }
    int bad = 0, total = 0;
    size_t raw = 0, comp = 0, zl = 0;
    for (int bgzf = 0; bgzf < 2; bgzf++) {
        struct Case { Bytes* in; uint8_t* out; size_t cap; bool tight; };
        std::vector<Case> cases;
        for (const Bytes& d : plain) {
            if (bgzf && d.size() > 65000) continue;
            for (int tight = 0; tight < 2; tight++) {
                // (a slot one byte short: the small inputs and one size behind a block boundary — the thread-per-lane
                // build under the sanitizers takes a second per large member)
                if (tight && d.size() > 9000 && d.size() != 32769) continue;
                Case k;
                k.in = new Bytes(d);
                k.in->shrink_to_fit();
                k.cap = (size_t)cramgpu_gzip_bound((int64_t)d.size());
                k.tight = tight != 0;
                cases.push_back(k);
            }
        }
        const int n = (int)cases.size();
        // first pass with bound-sized slots gives the sizes; the tight cases then get a slot one byte short of theirs
        std::vector<cramgpu_block> blocks(n);
        std::vector<int32_t> lens(n, 12345);
        for (int pass = 0; pass < 2; pass++) {
            for (int i = 0; i < n; i++) {
                if (pass == 1 && cases[i].tight) cases[i].cap = (size_t)lens[i] - 1;
                if (pass == 1 && !cases[i].tight) cases[i].cap = (size_t)lens[i];
                cases[i].out = new uint8_t[cases[i].cap ? cases[i].cap : 1];
            }
            const uint8_t* in0 = cases[n - 1].in->data();  // an empty vector has no storage: its offset is never used
            uint8_t* out0 = cases[0].out;
            for (int i = 0; i < n; i++)
                blocks[i] = cramgpu_block{cases[i].in->empty() ? 0 : (int64_t)(cases[i].in->data() - in0), (int64_t)(cases[i].out - out0),
                                          (int32_t)cases[i].in->size(), (int32_t)cases[i].cap, 0, 0};
            std::vector<int32_t> got(n, 12345);
            if (cramgpu_gzip_dev(c, in0, out0, blocks.data(), n, got.data(), bgzf, nullptr) != 0) { printf("gzip rc %s\n", cramgpu_last_error()); return 1; }
            for (int i = 0; i < n; i++) {
                const Case& k = cases[i];
                total++;
                if (pass == 1 && k.tight) {
                    if (got[i] != CRAMGPU_E_OUT_TOO_SMALL) bad++, printf("gzip case %d: a slot one byte short gave %d\n", i, got[i]);
                } else {
                    Bytes s(k.out, k.out + (got[i] > 0 ? got[i] : 0)), back;
                    const int r = got[i] > 0 ? gz_ref(s, k.in->size(), back) : -9;
                    if (r != 0 || back != *k.in) bad++, printf("gzip case %d (n=%zu bgzf=%d): kernel %d, zlib verdict %d\n", i, k.in->size(), bgzf, got[i], r);
                    if (pass == 1 && got[i] != lens[i]) bad++, printf("gzip case %d: size changed between runs (%d, %d)\n", i, lens[i], got[i]);
                    if (bgzf && got[i] > 18 && (s[3] != 4 || s[12] != 'B' || s[13] != 'C' || (int)(s[16] | s[17] << 8) != got[i] - 1))
                        bad++, printf("gzip case %d: BGZF header\n", i);
                    if (pass == 0 && got[i] > 0) {
                        // the gunzip kernel reads it back too
                        Bytes o(k.in->size() + 1);
                        cramgpu_block b1{0, 0, got[i], (int32_t)k.in->size(), 0, 0};
                        int32_t l1 = 0;
                        cramgpu_gunzip_dev(c, s.data(), o.data(), &b1, 1, &l1, nullptr);
                        if (l1 != (int)k.in->size() || (!k.in->empty() && memcmp(o.data(), k.in->data(), k.in->size()))) bad++, printf("gzip case %d: gunzip kernel gave %d\n", i, l1);
                        if (!bgzf && !k.tight) raw += k.in->size(), comp += (size_t)got[i], zl += gz_make(*k.in, 6, Z_DEFAULT_STRATEGY).size();
                    }
                }
                if (pass == 0) lens[i] = got[i];
            }
            for (Case& k : cases) delete[] k.out;
        }
        for (Case& k : cases) delete k.in;
    }
    printf("gzip: %s (%d members; %zu raw bytes -> %zu, zlib level 6: %zu)\n", bad ? "FAILED" : "ok", total, raw, comp, zl);
    return bad != 0;
}

// ---- BAM stripes with damaged fields: the decoder must flag the record, never read outside the stripe (a heap block of its
// exact size) or write outside its outputs
static void put32(Bytes& b, int32_t v) { for (int k = 0; k < 4; k++) b.push_back((uint8_t)(v >> (8 * k))); }
static int run_bam_fuzz(cramgpu_ctx* c) {
    std::mt19937 rng(3);
    int bad = 0;
    for (int round = 0; round < 60; round++) {
        const int n = 1 + (int)(rng() % 40);
        Bytes stripe;
        std::vector<int64_t> rec_off;
        for (int r = 0; r < n; r++) {
            const int L = (int)(rng() % 70), nc = (int)(rng() % 4), ln = 1 + (int)(rng() % 12), lt = (int)(rng() % 9);
            rec_off.push_back((int64_t)stripe.size());
            put32(stripe, 32 + ln + 4 * nc + (L + 1) / 2 + L + lt);
            put32(stripe, (int32_t)(rng() % 25) - 1);
            put32(stripe, (int32_t)(rng() % 100000));
            stripe.push_back((uint8_t)ln), stripe.push_back((uint8_t)(rng() % 61));
            stripe.push_back(0), stripe.push_back(0);
            stripe.push_back((uint8_t)nc), stripe.push_back(0);
            stripe.push_back((uint8_t)rng()), stripe.push_back((uint8_t)(rng() & 15));
            put32(stripe, L), put32(stripe, -1), put32(stripe, -1), put32(stripe, 0);
            for (int i = 0; i < ln - 1; i++) stripe.push_back((uint8_t)('a' + rng() % 26));
            stripe.push_back(0);
            for (int k = 0; k < nc; k++) put32(stripe, (int32_t)(((rng() % 50) << 4) | (rng() % 9)));
            for (int i = 0; i < (L + 1) / 2 + L + lt; i++) stripe.push_back((uint8_t)rng());
        }
        rec_off.push_back((int64_t)stripe.size());
        for (int m = 0, nm = (int)(rng() % 6); m < nm; m++) {  // damage: any byte but the block sizes
            const int r = (int)(rng() % n);
            const int64_t a = rec_off[r] + 4, b = rec_off[r + 1];
            stripe[(size_t)(a + (int64_t)(rng() % (uint64_t)(b - a)))] = (uint8_t)rng();
        }
        uint8_t* heap = new uint8_t[stripe.size()];
        memcpy(heap, stripe.data(), stripe.size());
        const int64_t cap_b = (int64_t)stripe.size() * 2 + 16, cap_c = (int64_t)stripe.size() / 4 + 1;
        std::vector<int32_t> i32[9];
        for (auto& v : i32) v.assign((size_t)n, 0);
        std::vector<int64_t> boff((size_t)n + 1), coff((size_t)n + 1), noff((size_t)n), toff((size_t)n);
        Bytes mapq((size_t)n), rfl((size_t)n), bases((size_t)cap_b), quals((size_t)cap_b);
        std::vector<uint32_t> cig((size_t)cap_c);
        std::vector<int32_t> nlen((size_t)n), tlen2((size_t)n), status((size_t)n, 77);
        cramgpu_bam_out o{};
        o.ref_id = i32[0].data(), o.aln_start = i32[1].data(), o.mapq = mapq.data(), o.bam_flags = i32[2].data();
        o.mate_ref = i32[3].data(), o.mate_start = i32[4].data(), o.tlen = i32[5].data(), o.read_len = i32[6].data();
        o.base_off = boff.data(), o.bases = bases.data(), o.quals = quals.data(), o.bases_cap = cap_b;
        o.n_cigar = i32[7].data(), o.cigar_off = coff.data(), o.cigar = cig.data(), o.cigar_cap = cap_c;
        o.aln_span = i32[8].data(), o.rflags = rfl.data(), o.name_off = noff.data(), o.name_len = nlen.data();
        o.tag_off = toff.data(), o.tag_len = tlen2.data();
        if (cramgpu_bam_decode_dev(c, n, heap, rec_off.data(), &o, status.data(), nullptr) != 0) bad++;
        for (int r = 0; r < n; r++)
            if (status[r] != 0 && status[r] != CRAMGPU_E_CORRUPT && status[r] != CRAMGPU_E_INSUFFICIENT) bad++;
        delete[] heap;
    }
    printf("bam fuzz: %s\n", bad ? "FAILED" : "ok");
    return bad != 0;
}

// ---- BAM records out: every output (stripe, offsets, status) a heap block of its exact size
static int run_bam_encode(cramgpu_ctx* c) {
    std::mt19937 rng(8);
    int bad = 0;
    for (int round = 0; round < 40; round++) {
        const int n = 1 + (int)(rng() % 70);
        std::vector<int32_t> ref_id(n), st(n), en(n), bf(n), mf(n), mr(n), ms(n), tl(n), rl(n), nc(n), qst(n);
        std::vector<int64_t> boff(n + 1), coff(n + 1), noff(n + 1), toff(n + 1);
        Bytes mapq(n);
        int64_t nb = 0, ncg = 0, nn = 0, nt = 0, total = 0;
        const bool with_names = round & 1, with_tags = round & 2;
        for (int r = 0; r < n; r++) {
            rl[r] = (int)(rng() % 5 == 0 ? rng() % 3 : rng() % 200);
            nc[r] = (int)(rng() % 5);
            ref_id[r] = (int)(rng() % 4) - 1, st[r] = (int)(rng() % 100000), en[r] = st[r] + rl[r], bf[r] = (int)(rng() % 4096);
            mf[r] = (int)(rng() % 4), mr[r] = (int)(rng() % 4) - 1, ms[r] = (int)(rng() % 1000) - 3, tl[r] = (int)(rng() % 999) - 500;
            mapq[r] = (uint8_t)rng(), qst[r] = (int)(rng() % 3 == 0);
            boff[r] = nb, coff[r] = ncg, noff[r] = nn, toff[r] = nt;
            nb += rl[r], ncg += nc[r];
            const int ln = with_names ? 1 + (int)(rng() % 20) : 0, lt = with_tags ? (int)(rng() % 30) : 0;
            nn += ln, nt += lt;
            int digits = 1;
            for (long long v = 990 + r; v >= 10; v /= 10) digits++;
            total += 36 + (with_names ? ln : digits) + 1 + 4 * nc[r] + (rl[r] + 1) / 2 + rl[r] + lt;
        }
        boff[n] = nb, coff[n] = ncg, noff[n] = nn, toff[n] = nt;
        // inputs of exact size too (the kernel reads 16 bases at a time where the arena allows)
        uint8_t* bases = new uint8_t[nb ? nb : 1];
        uint8_t* quals = new uint8_t[nb ? nb : 1];
        for (int64_t i = 0; i < nb; i++) bases[i] = (uint8_t)"ACGTNacgtn=MRSVWYHKDB.@"[rng() % 23], quals[i] = (uint8_t)rng();
        uint32_t* cig = new uint32_t[ncg ? ncg : 1];
        for (int64_t i = 0; i < ncg; i++) cig[i] = (uint32_t)rng();
        uint8_t* names = new uint8_t[nn ? nn : 1];
        uint8_t* tags = new uint8_t[nt ? nt : 1];
        for (int64_t i = 0; i < nn; i++) names[i] = (uint8_t)('a' + rng() % 26);
        for (int64_t i = 0; i < nt; i++) tags[i] = (uint8_t)rng();
        cramgpu_bam_cols b{};
        b.ref_id = ref_id.data(), b.aln_start = st.data(), b.aln_end = en.data(), b.mapq = mapq.data(), b.bam_flags = bf.data();
        b.mate_flags = mf.data(), b.mate_ref = mr.data(), b.mate_start = ms.data(), b.tlen = tl.data(), b.read_len = rl.data();
        b.base_off = boff.data(), b.bases = bases, b.quals = quals, b.qual_status = qst.data();
        b.n_cigar = nc.data(), b.cigar_off = coff.data(), b.cigar = cig;
        if (with_names) b.names = names, b.name_off = noff.data();
        b.name_base = 990;
        if (with_tags) b.tags = tags, b.tag_off = toff.data();
        uint8_t* out = new uint8_t[total];
        int64_t* rec_off = new int64_t[n + 1];
        int32_t* status = new int32_t[n];
        if (cramgpu_bam_encode_dev(c, n, &b, out, total, rec_off, status, nullptr) != 0 || rec_off[n] != total) {
            printf("bam encode: rc / size mismatch (%lld vs %lld)\n", (long long)rec_off[n], (long long)total);
            bad++;
        }
        int64_t at = 0;  // the stripe walks: block sizes chain to the end
        for (int r = 0; r < n && !bad; r++) {
            int32_t bs;
            memcpy(&bs, out + at, 4);
            if (rec_off[r] != at || bs < 32) bad++;
            at += 4 + bs;
        }
        if (!bad && at != total) bad++;
        // one byte short: nothing may be written
        if (total > 1 && cramgpu_bam_encode_dev(c, n, &b, out, total - 1, rec_off, status, nullptr) != 0) bad++;
        for (int r = 0; r < n && total > 1; r++)
            if (status[r] != CRAMGPU_E_INSUFFICIENT) bad++;
        delete[] bases; delete[] quals; delete[] cig; delete[] names; delete[] tags; delete[] out; delete[] rec_off; delete[] status;
    }
    printf("bam encode: %s\n", bad ? "FAILED" : "ok");
    return bad != 0;
}

int main(int argc, char** argv) {
    std::string what = argc > 1 ? argv[1] : "all";
    cramgpu_ctx* c = cramgpu_open(0);
    if (!c) { printf("open failed\n"); return 1; }
    int bad = 0;
    if (what == "file") {
        FILE* f = fopen(argv[2], "rb");
        Bytes d(1 << 24);
        d.resize(fread(d.data(), 1, d.size(), f));
        fclose(f);
        std::vector<Bytes> datas{d};
        bad |= check_rans(c, datas, atoi(argv[3]));
        printf("file: %s\n", bad ? "FAILED" : "ok");
    }
    if (what == "rans" || what == "all") {
        std::mt19937 rng(5);
        std::vector<Bytes> datas;
        const bool quick = argc > 2 && std::string(argv[2]) == "quick";
        const std::vector<int> sizes = quick ? std::vector<int>{1, 3, 4, 5, 17, 63, 100, 700}
                                             : std::vector<int>{1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 63, 64, 65, 127, 255, 1000, 4099, 20000};
        const std::vector<int> alph = quick ? std::vector<int>{1, 4, 13, 41, 60, 256} : std::vector<int>{1, 2, 4, 8, 13, 16, 17, 41, 57, 90, 256};
        for (int n : sizes)
            for (int a : alph) {
                Bytes d((size_t)n);
                for (auto& b : d) b = (uint8_t)((rng() % a) * (a <= 8 ? 31 : 1) + (a == 1 ? 65 : 0));
                datas.push_back(d);
            }
        {   // ITF8-like series: a wide alphabet whose contexts have few successors each (ragged order-1 rows in LDS)
            Bytes itf;
            for (int i = 0; i < (quick ? 400 : 6000); i++) {
                int v = 250 + (int)(rng() % 101);
                if (rng() & 1) {
                    itf.push_back((uint8_t)(0x80 | (v >> 8)));
                    itf.push_back((uint8_t)v);
                } else {
                    uint32_t u = (uint32_t)(-v);
                    itf.push_back((uint8_t)(0xF0 | (u >> 28)));
                    itf.push_back((uint8_t)(u >> 20));
                    itf.push_back((uint8_t)(u >> 12));
                    itf.push_back((uint8_t)(u >> 4));
                    itf.push_back((uint8_t)(u & 0x0F));
                }
            }
            datas.push_back(itf);
        }
        Bytes runs(3000);
        for (size_t i = 0; i < runs.size(); i++) runs[i] = (uint8_t)((i / 37) % 5 + 40);
        datas.push_back(runs);
        if (!quick) {  // blocks beyond the alphabet sample (64 KiB): alphabet complete in the sample, a late symbol (counted
                       // again with the full alphabet), and the same over an alphabet too wide for the LDS tables
            for (int wide = 0; wide < 2; wide++)
                for (int late = 0; late < 2; late++) {
                    Bytes d(70000);
                    for (auto& b : d) b = (uint8_t)(wide ? rng() % 200 : 30 + rng() % 9);
                    if (late) d[69990] = 250, d[66000] = 7;
                    datas.push_back(d);
                }
        }
        for (int order = 0; order < 2; order++) bad |= check_rans(c, datas, order);
        printf("rans: %s (%zu blocks x 2 orders)\n", bad ? "FAILED" : "ok", datas.size());
    }
    if (!bad && (what == "records" || what == "all")) bad |= run_records(c);
    if (!bad && (what == "gunzip" || what == "all")) bad |= run_gunzip(c);
    if (!bad && (what == "gzip" || what == "all")) bad |= run_gzip(c);
    if (!bad && (what == "bam" || what == "all")) bad |= run_bam_fuzz(c);
    if (!bad && (what == "bam" || what == "all")) bad |= run_bam_encode(c);
    cramgpu_close(c);
    return bad;
}
