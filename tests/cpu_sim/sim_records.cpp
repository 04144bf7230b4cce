// TEST INFRASTRUCTURE ONLY — record-kernel half of the host sanitizer harness: random containers through the
// kernels of cramtools_amd/csrc/records.hip (compiled for the CPU, see hip/hip_runtime.h) against oracle/records.c.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../include/cramgpu.h"

extern "C" {
typedef struct {
    int32_t read_cat, read_param, treatment, n_base_cat;
    int32_t base_cat[8], base_param[8];
} orc_policy_t;
int64_t orc_extract_features(int32_t n_reads, const uint8_t* bases, const int64_t* base_off, const int32_t* read_len,
                             const int32_t* aln_start, const uint8_t* rflags, const uint32_t* cigar,
                             const int64_t* cigar_off, const int32_t* n_cigar, const uint8_t* ref, int64_t ref_len,
                             int64_t feat_cap, uint8_t* f_op, int32_t* f_pos, int32_t* f_a, int32_t* f_b,
                             int64_t* feat_off, int64_t* sub_hist);
void orc_substitution_matrix(const int64_t* hist, uint8_t* matrix5);
void orc_assign_codes(int64_t n_feat, const uint8_t* f_op, int32_t* f_a, const uint8_t* matrix5);
void orc_restore_bases(int32_t n_reads, const int32_t* read_len, const int32_t* aln_start, const uint8_t* rflags,
                       const int64_t* feat_off, const uint8_t* f_op, const int32_t* f_pos, const int32_t* f_a,
                       const int32_t* f_b, const uint8_t* seq, const int64_t* seq_off, const uint8_t* win,
                       int64_t win_len, int64_t win_start0, const uint8_t* matrix5, uint8_t* out,
                       const int64_t* out_off, int32_t* status);
void orc_restore_quals(int32_t n_reads, const int32_t* read_len, const uint8_t* rflags, const int64_t* feat_off,
                       const uint8_t* f_op, const int32_t* f_pos, const int32_t* f_a, const int32_t* f_b,
                       uint8_t default_qs, uint8_t* scores, const int64_t* out_off, int32_t* status);
int32_t orc_tracks_windows(int32_t n_reads, const int32_t* aln_start, const int32_t* aln_span, const uint8_t* rflags,
                           const uint8_t* mapq, const orc_policy_t* pol, int32_t n_pol, int64_t ref_len,
                           int32_t window, int32_t* wpos);
int32_t orc_lossy_quals(int32_t n_reads, const uint8_t* bases, const uint8_t* quals, const int64_t* base_off,
                        const int32_t* read_len, const int32_t* aln_start, const uint8_t* rflags,
                        const uint8_t* mapq, const uint32_t* cigar, const int64_t* cigar_off, const int32_t* n_cigar,
                        const uint8_t* ref, int64_t ref_len, const int32_t* wpos, int32_t window,
                        const orc_policy_t* pol, int32_t n_pol, uint8_t* scores_out, uint8_t* force,
                        int32_t* status);
int64_t orc_merge_quality_features(int32_t n_reads, const int32_t* read_len, const int64_t* base_off,
                                   const uint8_t* scores, const uint8_t* force, const int32_t* qstatus,
                                   const int64_t* feat_off, const uint8_t* f_op, const int32_t* f_pos,
                                   const int32_t* f_a, const int32_t* f_b, int64_t cap, uint8_t* m_op,
                                   int32_t* m_pos, int32_t* m_a, int32_t* m_b, int64_t* m_off);
}

namespace {

struct Container {
    std::vector<uint8_t> ref, bases, quals, rflags, mapq;
    std::vector<int64_t> base_off, cigar_off;
    std::vector<int32_t> read_len, aln_start, n_cigar, aln_span;
    std::vector<uint32_t> cigar;
    int n = 0;
};

// mode 0: Illumina-like (mostly one M), 1: many short elements, 2: long reads
Container make(std::mt19937& rng, int n_reads, int ref_len, int mode, bool sorted) {
    Container c;
    const char* AL = "ACGT";
    c.ref.resize(ref_len + 16);
    for (int i = 0; i < ref_len; i++) {
        int r = rng() % 100;
        c.ref[i] = r < 94 ? AL[rng() % 4] : r < 97 ? 'N' : r < 99 ? "acgt"[rng() % 4] : 'R';
    }
    std::vector<int> starts(n_reads);
    for (auto& s : starts) s = 1 + rng() % (ref_len + 20);  // some start past the end
    if (sorted) std::sort(starts.begin(), starts.end());
    for (int r = 0; r < n_reads; r++) {
        std::vector<std::pair<int, int>> ops;  // (op, len)
        int n_el = mode == 0 ? (rng() % 10 == 0 ? 1 + rng() % 5 : 1) : mode == 1 ? 1 + rng() % 12 : 20 + rng() % 60;
        for (int e = 0; e < n_el; e++) {
            int op;
            if (mode == 0 && n_el == 1) op = 0;
            else {
                static const int pick[] = {0, 0, 0, 0, 1, 2, 4, 3, 5, 6, 7, 8, 1, 2};
                op = pick[rng() % 14];
            }
            int len = mode == 0 && n_el == 1 ? 150 : mode == 2 ? 1 + rng() % 200 : 1 + rng() % 70;
            ops.push_back({op, len});
        }
        if (rng() % 9 == 0) ops.push_back({2, 1 + (int)(rng() % 4)});  // deletion as the last element
        int L = 0, span = 0;
        for (auto& o : ops) {
            if (o.first == 0 || o.first == 1 || o.first == 4 || o.first == 7 || o.first == 8) L += o.second;
            if (o.first == 0 || o.first == 2 || o.first == 3 || o.first == 7 || o.first == 8) span += o.second;
        }
        uint8_t fl = 0;
        int start = starts[r];
        int sel = rng() % 40;
        if (sel == 0) fl |= CRAMGPU_RF_UNMAPPED;
        if (sel == 1) start = 0;
        if (sel == 2) fl |= CRAMGPU_RF_NULL_QUALS;
        if (sel == 3) fl |= CRAMGPU_RF_SKIP;
        c.base_off.push_back((int64_t)c.bases.size());
        c.cigar_off.push_back((int64_t)c.cigar.size());
        c.read_len.push_back(L);
        c.aln_start.push_back(start);
        c.aln_span.push_back(span - 1);
        c.n_cigar.push_back((int)ops.size());
        c.rflags.push_back(fl);
        c.mapq.push_back(rng() % 5 == 0 ? 0 : (uint8_t)(rng() % 61));
        for (auto& o : ops) c.cigar.push_back(((uint32_t)o.second << 4) | (uint32_t)o.first);
        // bases: copy of the reference along the alignment with errors
        int rp = start > 0 ? start - 1 : 0;
        for (auto& o : ops) {
            bool cr = o.first == 0 || o.first == 1 || o.first == 4 || o.first == 7 || o.first == 8;
            bool cf = o.first == 0 || o.first == 2 || o.first == 3 || o.first == 7 || o.first == 8;
            if (cr)
                for (int i = 0; i < o.second; i++) {
                    uint8_t b = (cf && rp + i < ref_len) ? c.ref[rp + i] : AL[rng() % 4];
                    int e = rng() % 100;
                    if (e < 5) b = AL[rng() % 4];
                    if (e == 5) b = 'N';
                    if (e == 6) b = "acgtn"[rng() % 5];
                    c.bases.push_back(b);
                    c.quals.push_back((uint8_t)(rng() % 20 == 0 ? rng() % 120 : rng() % 42));
                }
            if (cf) rp += o.second;
        }
    }
    c.n = n_reads;
    c.bases.resize(c.bases.size() + 16);
    c.quals.resize(c.quals.size() + 16);
    c.cigar.push_back(0);
    return c;
}

#define CHECK(cond, ...)                 \
    do {                                 \
        if (!(cond)) {                   \
            printf("records: " __VA_ARGS__); \
            printf("\n");                \
            return 1;                    \
        }                                \
    } while (0)

int run_container(cramgpu_ctx* ctx, Container& c, int ref_len, int tag) {
    const int n = c.n;
    cramgpu_refwin win{0, ref_len, 0};
    cramgpu_reads R{};
    R.n_reads = n;
    R.n_cont = 1;
    R.n_bases = (int64_t)c.bases.size();
    R.max_read_len = 0;
    for (int r = 0; r < n; r++) R.max_read_len = std::max(R.max_read_len, c.read_len[r]);
    R.bases = c.bases.data();
    R.quals = c.quals.data();
    R.base_off = c.base_off.data();
    R.read_len = c.read_len.data();
    R.aln_start = c.aln_start.data();
    R.rflags = c.rflags.data();
    R.mapq = c.mapq.data();
    R.cigar = c.cigar.data();
    R.cigar_off = c.cigar_off.data();
    R.n_cigar = c.n_cigar.data();
    R.read_cont = nullptr;
    R.win = &win;
    // ---- K3 ----
    const int64_t cap = (int64_t)c.bases.size() + 4 * (int64_t)c.cigar.size() + 16;
    std::vector<uint8_t> op(cap), oop(cap);
    std::vector<int32_t> pos(cap), a(cap), b(cap), opos(cap), oa(cap), ob(cap);
    std::vector<int64_t> off(n + 1), ooff(n + 1), hist(25, 0), ohist(25, 0);
    cramgpu_features F{op.data(), pos.data(), a.data(), b.data(), off.data(), cap};
    int64_t nf = cramgpu_extract_features_dev(ctx, &R, c.ref.data(), &F, hist.data(), nullptr);
    int64_t onf = orc_extract_features(n, c.bases.data(), c.base_off.data(), c.read_len.data(), c.aln_start.data(),
                                       c.rflags.data(), c.cigar.data(), c.cigar_off.data(), c.n_cigar.data(),
                                       c.ref.data(), ref_len, cap, oop.data(), opos.data(), oa.data(), ob.data(),
                                       ooff.data(), ohist.data());
    CHECK(nf == onf, "[%d] feature count %lld vs oracle %lld", tag, (long long)nf, (long long)onf);
    CHECK(!memcmp(off.data(), ooff.data(), (n + 1) * 8), "[%d] feature offsets differ", tag);
    for (int64_t i = 0; i < nf; i++)
        CHECK(op[i] == oop[i] && pos[i] == opos[i] && a[i] == oa[i] && b[i] == ob[i], "[%d] feature %lld differs: %c@%d vs %c@%d",
              tag, (long long)i, op[i], pos[i], oop[i], opos[i]);
    CHECK(hist == ohist, "[%d] substitution histogram differs", tag);
    // ---- F3 + codes ----
    uint8_t m5[5], om5[5];
    cramgpu_substitution_matrix(hist.data(), m5);
    orc_substitution_matrix(ohist.data(), om5);
    CHECK(!memcmp(m5, om5, 5), "[%d] substitution matrix differs", tag);
    CHECK(cramgpu_assign_codes_dev(ctx, n, nullptr, 1, &F, m5, nullptr) == 0, "assign_codes failed");
    orc_assign_codes(onf, oop.data(), oa.data(), om5);
    for (int64_t i = 0; i < nf; i++) CHECK(a[i] == oa[i], "[%d] code of feature %lld differs", tag, (long long)i);
    // ---- K4 over a window that cuts the contig short on both sides ----
    for (int wv = 0; wv < 2; wv++) {
        int64_t lo = wv ? 37 : 0, hi = wv ? ref_len - 91 : ref_len;
        cramgpu_refwin w2{lo, hi - lo, lo};
        std::vector<int64_t> out_off(n);
        int64_t tot = 0;
        for (int r = 0; r < n; r++) {
            out_off[r] = tot;
            tot += c.read_len[r];
        }
        std::vector<uint8_t> out(tot + 16, 0xEE), oout(tot + 16, 0xEE);
        std::vector<int32_t> st(n), ost(n);
        // S features index the read's own bases (encode side); on decode the sequence arena plays that part
        CHECK(cramgpu_restore_bases_dev(ctx, n, c.read_len.data(), c.aln_start.data(), c.rflags.data(), nullptr, 1, &w2, &F,
                                        c.bases.data(), c.base_off.data(), (int64_t)c.bases.size(), c.ref.data(), m5, out.data(),
                                        out_off.data(),
                                        st.data(), nullptr) == 0, "restore_bases failed");
        orc_restore_bases(n, c.read_len.data(), c.aln_start.data(), c.rflags.data(), ooff.data(), oop.data(), opos.data(),
                          oa.data(), ob.data(), c.bases.data(), c.base_off.data(), c.ref.data() + lo, hi - lo, lo, om5,
                          oout.data(), out_off.data(), ost.data());
        for (int r = 0; r < n; r++) {
            int want = ost[r] == -1 ? CRAMGPU_E_JAVA_THROWS : ost[r];
            CHECK(st[r] == want, "[%d] restore status of read %d: %d vs %d (window %d)", tag, r, st[r], want, wv);
            if (st[r] == 0)
                CHECK(!memcmp(out.data() + out_off[r], oout.data() + out_off[r], c.read_len[r]),
                      "[%d] restored bases of read %d differ (window %d)", tag, r, wv);
        }
    }
    // ---- K5 + merge + N2 for several policies ----
    // terms that depend on the coverage / mismatch tracks are refused: only the no-ops X0 / P9999 run
    for (const char* spec : {"R40-X5", "U8-P3"}) {
        cramgpu_policy pol[16];
        int np = cramgpu_parse_policies(spec, pol, 16);
        CHECK(np >= 0, "parse %s failed", spec);
        std::vector<int32_t> wpos(n + 1);
        CHECK(cramgpu_tracks_windows(n, c.aln_start.data(), c.aln_span.data(), c.rflags.data(), c.mapq.data(), pol, np, ref_len,
                                     1000000, wpos.data(), nullptr) == CRAMGPU_E_UNSUPPORTED, "tracks_windows accepted %s", spec);
        std::vector<uint8_t> sc(c.bases.size() + 1), fr(n + 1);
        std::vector<int32_t> st(n + 1);
        CHECK(cramgpu_lossy_quals_dev(ctx, &R, c.ref.data(), wpos.data(), 1000000, pol, np, sc.data(), fr.data(), st.data(),
                                      nullptr) == CRAMGPU_E_UNSUPPORTED, "lossy_quals accepted %s", spec);
    }
    const char* specs[] = {"N40-D8", "R8-N40-D8", "*8", "N40", "m5", "I40", "R8", "R40-X0", "U8-P9999", "M10_0-N8"};
    for (const char* spec : specs) {
        cramgpu_policy pol[16];
        int np = cramgpu_parse_policies(spec, pol, 16);
        CHECK(np >= 0, "parse %s failed", spec);
        static_assert(sizeof(cramgpu_policy) == sizeof(orc_policy_t), "policy layout");
        std::vector<int32_t> wpos(n), owpos(n);
        for (int window : {1000000, 700}) {
            int32_t at = -1;
            int rc = cramgpu_tracks_windows(n, c.aln_start.data(), c.aln_span.data(), c.rflags.data(), c.mapq.data(), pol, np,
                                            ref_len, window, wpos.data(), &at);
            if (rc == CRAMGPU_E_JAVA_THROWS) rc = -(at + 1);  // the oracle's convention
            int orc = orc_tracks_windows(n, c.aln_start.data(), c.aln_span.data(), c.rflags.data(), c.mapq.data(),
                                         (const orc_policy_t*)pol, np, ref_len, window, owpos.data());
            CHECK(rc == orc, "[%d] tracks_windows rc %d vs %d (%s)", tag, rc, orc, spec);
            if (rc != 0) continue;  // unsorted containers make ReferenceTracks throw: nothing more to compare
            CHECK(wpos == owpos, "[%d] tracks windows differ (%s)", tag, spec);
            std::vector<uint8_t> sc(c.bases.size(), 0x77), osc(c.bases.size(), 0x77), fr(n), ofr(n);
            std::vector<int32_t> st(n), ost(n);
            CHECK(cramgpu_lossy_quals_dev(ctx, &R, c.ref.data(), wpos.data(), window, pol, np, sc.data(), fr.data(), st.data(),
                                          nullptr) == 0, "lossy_quals failed");
            orc_lossy_quals(n, c.bases.data(), c.quals.data(), c.base_off.data(), c.read_len.data(), c.aln_start.data(),
                            c.rflags.data(), c.mapq.data(), c.cigar.data(), c.cigar_off.data(), c.n_cigar.data(), c.ref.data(),
                            ref_len, owpos.data(), window, (const orc_policy_t*)pol, np, osc.data(), ofr.data(), ost.data());
            for (int r = 0; r < n; r++) {
                int want = ost[r] == -1 ? CRAMGPU_E_JAVA_THROWS : ost[r];
                CHECK(st[r] == want, "[%d] lossy status read %d: %d vs %d (%s)", tag, r, st[r], want, spec);
                if (st[r] != 0) continue;
                CHECK(fr[r] == ofr[r], "[%d] force flag read %d (%s)", tag, r, spec);
                CHECK(!memcmp(sc.data() + c.base_off[r], osc.data() + c.base_off[r], c.read_len[r]),
                      "[%d] lossy scores read %d differ (%s)", tag, r, spec);
            }
            // merge (only statuses 0/1/2 are meaningful downstream; map throws to skipped on both sides)
            for (int r = 0; r < n; r++)
                if (st[r] < 0) st[r] = ost[r] = 2;
            const int64_t mcap = nf + (int64_t)c.bases.size() + 16;
            std::vector<uint8_t> mop(mcap), omop(mcap);
            std::vector<int32_t> mpos(mcap), ma(mcap), mb(mcap), ompos(mcap), oma(mcap), omb(mcap);
            std::vector<int64_t> moff(n + 1), omoff(n + 1);
            cramgpu_features M{mop.data(), mpos.data(), ma.data(), mb.data(), moff.data(), mcap};
            int64_t nm = cramgpu_merge_quality_features_dev(ctx, n, c.read_len.data(), c.base_off.data(), sc.data(), fr.data(),
                                                            st.data(), &F, &M, nullptr);
            int64_t onm = orc_merge_quality_features(n, c.read_len.data(), c.base_off.data(), osc.data(), ofr.data(), ost.data(),
                                                     ooff.data(), oop.data(), opos.data(), oa.data(), ob.data(), mcap,
                                                     omop.data(), ompos.data(), oma.data(), omb.data(), omoff.data());
            CHECK(nm == onm, "[%d] merged count %lld vs %lld (%s)", tag, (long long)nm, (long long)onm, spec);
            CHECK(!memcmp(moff.data(), omoff.data(), (n + 1) * 8), "[%d] merged offsets (%s)", tag, spec);
            for (int64_t i = 0; i < nm; i++)
                CHECK(mop[i] == omop[i] && mpos[i] == ompos[i] && ma[i] == oma[i] && mb[i] == omb[i],
                      "[%d] merged feature %lld (%s)", tag, (long long)i, spec);
            // N2 on the merged features
            std::vector<uint8_t> rfl(n);
            std::vector<int64_t> qoff(n);
            int64_t tot = 0;
            for (int r = 0; r < n; r++) {
                rfl[r] = (uint8_t)((c.rflags[r] & CRAMGPU_RF_SKIP) | (fr[r] ? CRAMGPU_RF_FORCE_QS : 0));
                qoff[r] = tot;
                tot += c.read_len[r];
            }
            std::vector<uint8_t> q(tot + 16), oq(tot + 16);
            for (int r = 0; r < n; r++) {
                memcpy(q.data() + qoff[r], sc.data() + c.base_off[r], c.read_len[r]);
                memcpy(oq.data() + qoff[r], sc.data() + c.base_off[r], c.read_len[r]);
            }
            std::vector<int32_t> qs(n), oqs(n);
            CHECK(cramgpu_restore_quals_dev(ctx, n, c.read_len.data(), rfl.data(), &M, 30, q.data(), qoff.data(), qs.data(),
                                            nullptr) == 0, "restore_quals failed");
            orc_restore_quals(n, c.read_len.data(), rfl.data(), omoff.data(), omop.data(), ompos.data(), oma.data(), omb.data(),
                              30, oq.data(), qoff.data(), oqs.data());
            CHECK(qs == oqs, "[%d] restore_quals status (%s)", tag, spec);
            CHECK(q == oq, "[%d] restored qualities differ (%s)", tag, spec);
        }
    }
    return 0;
}

}  // namespace

int run_records(cramgpu_ctx* ctx) {
    std::mt19937 rng(7);
    int bad = 0, tag = 0;
    struct Cfg { int n, ref_len, mode; bool sorted; } cfgs[] = {
        {40, 3000, 0, true}, {30, 2000, 1, true}, {24, 1500, 1, false}, {6, 30000, 2, true}, {1, 400, 1, true}, {0, 100, 0, true}};
    for (auto& cf : cfgs) {
        Container c = make(rng, cf.n, cf.ref_len, cf.mode, cf.sorted);
        bad |= run_container(ctx, c, cf.ref_len, tag++);
        if (bad) break;
    }
    printf("records: %s (%d containers)\n", bad ? "FAILED" : "ok", tag);
    return bad;
}
