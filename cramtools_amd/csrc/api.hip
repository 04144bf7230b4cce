// Host side of libcramgpu.so: contexts, staging, launches.  C ABI declared in include/cramgpu.h.
// There is deliberately no CPU implementation of any kernel here: without a gfx950 device every compute entry
// point fails with CRAMGPU_E_NO_DEVICE / CRAMGPU_E_HIP.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "internal.h"

// ---- scratch of cramgpu_rans_encode_static: device memory the library never allocates (include/cramgpu.h) ----
// Layout: [0, 256) work-queue ticket | [256, 256 + 16 KiB) longest-first permutation of up to 4096 blocks |
// kStaticGroups per-workgroup table areas (order-1 tables of alphabets too wide for LDS).  Every GPU the process uses
// holds its own copy of this array until the process ends; in host code the name is only a handle for
// hipGetSymbolAddress (cramgpu_open), kernels and memsets get the per-device address kept in the context.
constexpr int kStaticGroups = 32;
constexpr int kStaticMaxBlocks = 4096;
constexpr size_t kStaticPermOff = 256;
constexpr size_t kStaticGsOff = kStaticPermOff + (size_t)kStaticMaxBlocks * sizeof(int32_t);
constexpr size_t kStaticBytes = kStaticGsOff + (size_t)kStaticGroups * ENC_GSCRATCH_BYTES;
static __device__ __align__(256) uint8_t scratch[kStaticBytes];

namespace {

// Hardware queues of the process: HIP spreads streams over GPU_MAX_HW_QUEUES queues (its default: 4) round robin and
// launches of streams that share a queue serialise; the job slots and the coalescing queue overlap launches of several
// streams (each lasts one chain latency however small it is).  Asked for when the library is loaded, which in a JNI
// host (System.loadLibrary) is before the first HIP call initialises the runtime; an explicit setting wins.
__attribute__((constructor)) void process_defaults() { (void)setenv("GPU_MAX_HW_QUEUES", "8", 0); }

thread_local std::string g_err;

int fail(int code, const char* what, hipError_t e = hipSuccess) {
    char buf[512];
    if (e != hipSuccess)
        snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    else
        snprintf(buf, sizeof buf, "%s", what);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                              \
    do {                                                          \
        hipError_t _e = (expr);                                   \
        if (_e != hipSuccess) return fail(CRAMGPU_E_HIP, #expr, _e); \
    } while (0)

int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

// hipFree / hipHostFree wait for the whole device.  A thread that must not stall while other launches are in flight (the
// coalescing queue's dispatcher) points this at a list: buffers it outgrows are parked there instead of freed, and grow
// geometrically so that the list stays short; the owner of the list frees it when nothing is in flight.
struct Graveyard {
    std::vector<void*> dev, pin;
};
thread_local Graveyard* tl_graveyard = nullptr;

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;  // elements
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p && tl_graveyard) tl_graveyard->dev.push_back(p);
        else if (p) (void)hipFree(p);
        p = nullptr;
        size_t want = std::max(n, (size_t)1024);
        if (tl_graveyard) want = std::max(want, 2 * cap);
        cap = 0;
        hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};
template <class T>
struct PinBuf {
    T* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p && tl_graveyard) tl_graveyard->pin.push_back(p);
        else if (p) (void)hipHostFree(p);
        p = nullptr;
        size_t want = std::max(n, (size_t)1024);
        if (tl_graveyard) want = std::max(want, 2 * cap);
        cap = 0;
        hipError_t e = hipHostMalloc((void**)&p, want * sizeof(T), hipHostMallocDefault);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

}  // namespace

// What one rANS launch needs besides the arenas: descriptor / permutation staging, a ticket word, the per-workgroup
// global scratch.  The context has one set (synchronous calls, stream-ordered); every asynchronous job slot has its
// own set and its own stream, so that the (small, latency-bound) launches of several jobs overlap on the device.
struct LaunchRes {
    hipEvent_t staged = nullptr;  // completion of the last descriptor upload (pinned staging reuse)
    bool staged_pending = false;
    PinBuf<cramgpu_block> h_desc;
    PinBuf<int32_t> h_perm;
    DevBuf<cramgpu_block> d_desc;
    DevBuf<int32_t> d_perm;
    DevBuf<uint32_t> tickets;
    DevBuf<uint32_t> simd_arrivals;  // decoder: waves of the launch per SIMD (rans_dec.hip, first_wave_on_simd)
    DevBuf<uint8_t> sort_ws;         // cramgpu_rans_encode_series_dev: workspace of the longest-first sort on the device
    uint32_t ticket_next = 0;
    uint32_t launch_no = 0;          // decoder launches so far (the arrival table's stamp)
    DevBuf<uint8_t> enc_gs, dec_gs;
    void release() {
        h_desc.release(); h_perm.release(); d_desc.release(); d_perm.release(); tickets.release(); simd_arrivals.release(); sort_ws.release();
        enc_gs.release(); dec_gs.release();
        if (staged) (void)hipEventDestroy(staged);
        staged = nullptr;
    }
};

// What the second half of a record job (once the feature count is on the host) still has to copy back.
struct CopyBack {
    void* dst;
    const void* src;
    size_t bytes;     // fixed size, or per-feature size when per_feature
    bool per_feature;
};

struct cramgpu_job {
    LaunchRes res;
    hipStream_t stream = nullptr;
    cramgpu_ctx* ctx = nullptr;
    bool busy = false, decode = false, fetched = false;
    int kind = 0;                 // 0 rANS blocks, 1 cramgpu_submit_reads, 2 cramgpu_submit_restore
    DevBuf<uint8_t> arena;        // record jobs: every device buffer of the job, carved out of one allocation
    PinBuf<uint8_t> pin;          // record jobs: small host-built tables on their way up, the feature count on its way down
    std::vector<CopyBack> back;   // record jobs: results to copy once the count is known
    int64_t* h_total = nullptr;   // in pin
    int64_t feat_cap = 0;
    int64_t* h_feat_count = nullptr;  // caller's feats.off[n_reads]
    int32_t early_rc = 0;         // verdict known at submit time (reported by wait)
    int32_t n = 0;
    uint8_t* out_arena = nullptr;
    int32_t* out_len = nullptr;
    std::vector<cramgpu_block> blocks;
    DevBuf<uint8_t> d_in, d_out;
    DevBuf<int32_t> d_len;
    DevBuf<int64_t> d_pos;
    PinBuf<int32_t> h_len;
    PinBuf<int64_t> h_pos;       // encode: start of every stream in the packed arena (n + 1 entries)
    DevBuf<uint8_t> d_pack;      // encode: the streams packed back to back (what crosses PCIe)
    DevBuf<int64_t> d_off;
    DevBuf<int32_t> d_err;
    int64_t out_bytes = 0, span_lo = 0, span_hi = 0;
    int64_t* out_pos = nullptr;  // cramgpu_rans_submit_packed: the streams stay packed, their starts go here
    hipEvent_t coded = nullptr, done = nullptr;  // lengths on the host / streams in the caller's arena
    hipEvent_t uploaded = nullptr;               // input arena on the device (recorded on the context's upload stream)
};

// Small host tables (code tables, policies) on their way to the device: a ring of pinned slots, each with its own
// device copy and an event that says when the kernels reading that copy have been queued behind it.  A *_dev call
// therefore never waits for the device; only a ring that wraps onto a slot whose upload has not run yet waits.
constexpr int kSmallSlots = 8;
struct SmallSlot {
    PinBuf<uint8_t> h;
    DevBuf<uint8_t> d;
    hipEvent_t ev = nullptr;
    bool pending = false;
};

struct cramgpu_ctx {
    cramgpu_job jobs[CRAMGPU_MAX_JOBS];
    SmallSlot small[kSmallSlots];
    int small_next = 0;
    // Calls on one context share scratch (descriptor staging, LPT permutation, per-workgroup tables, counters).  The
    // mutex orders them on the host; this event orders them on the DEVICE when consecutive calls use different
    // streams: every call records it behind its last launch, the next call's stream waits for it first.
    hipEvent_t last_use = nullptr;
    hipStream_t last_stream = nullptr;
    bool last_valid = false;
    uint8_t* static_scratch = nullptr;  // this device's address of `scratch` (cramgpu_rans_encode_static)
    // Input arenas of asynchronous jobs go up on ONE stream, in submission order: copies queued on the jobs' own streams
    // share the PCIe link, all finish together (four 425 MB uploads: all at 29 ms) and no kernel starts before that;
    // first come, first served lets job k's kernels run under job k+1's upload.
    hipStream_t upload_stream = nullptr;
    // reference contigs of the host-form level-2 calls (cramgpu_ref_upload) and their ReferenceTracks window positions
    struct Contig {
        DevBuf<uint8_t> d;
        int64_t len = -1;
        int64_t tracks_pos = 1;
    };
    std::vector<Contig> contigs;
    int device = 0;
    int n_cu = 0;
    hipStream_t stream = nullptr;
    LaunchRes res;
    std::mutex mu;
    // rANS launch geometry
    int tab_bytes = 0, dec_tab_bytes = 0, enc_wpc = 0, dec_wpc = 0, n_waves_enc = 0, n_waves_dec = 0;
    bool dec_roles = true;
    size_t enc_lds = 0, dec_lds = 0;
    DevBuf<uint32_t> rcp_tab;
    // descriptor staging
    // host-batch / level-0 staging
    DevBuf<uint8_t> d_in, d_out, d_pack;
    DevBuf<int32_t> d_len;
    DevBuf<int64_t> d_pos, d_off;
    DevBuf<int32_t> d_err;
    PinBuf<uint8_t> h_stage;
    PinBuf<int32_t> h_len;
    PinBuf<int64_t> h_off;
    // record kernels
    DevBuf<int32_t> d_counts;
    DevBuf<int32_t> d_list;
    DevBuf<int64_t> d_total, d_scan_ws;
    DevBuf<int64_t> d_idx_ws;                      // cramgpu_bam_index_dev: per-segment entry / exit / base
    DevBuf<uint8_t> d_series_alias;                // cramgpu_write_series_dev: containers whose QS is not copied
    DevBuf<uint32_t> d_mate_prev;                  // cramgpu_restore_mates_dev: records pointing at a record
    DevBuf<uint8_t> d_mate_serial;                 //   containers whose chains share records
    DevBuf<int32_t> d_series_vals, d_series_cnt;   // cramgpu_read_series_dev: parsed values, counts
    DevBuf<int64_t> d_series_off, d_series_base;  // ... their offsets, per-container output bases
    PinBuf<int64_t> h_total;
    DevBuf<uint8_t> d_maskws, d_codes25;
    DevBuf<uint8_t> d_gz_ws;            // cramgpu_gzip_dev: token space of the resident waves
    hipEvent_t gz_use = nullptr;        //   ... and the order among the calls that share it
    hipStream_t gz_stream = nullptr;
    bool gz_valid = false;
    DevBuf<unsigned long long> d_prof;  // CRAMGPU_PROF=1
    bool prof_on = false;
};

namespace {

constexpr int kTicketSlots = 256;

// NULL is HIP's own null stream (what torch calls the default stream), NOT the context's private stream: work the
// caller queued on the null stream before the call is ordered before ours.
hipStream_t pick(cramgpu_ctx*, void* stream) { return (hipStream_t)stream; }

// Device-side ordering of consecutive calls on one context (see cramgpu_ctx::last_use).  enter() before the first
// touch of shared scratch, leave() behind the last launch of the call; both under the context mutex.
int enter(cramgpu_ctx* c, hipStream_t s) {
    HIPCHK(hipSetDevice(c->device));
    if (c->last_valid && c->last_stream != s) HIPCHK(hipStreamWaitEvent(s, c->last_use, 0));
    return 0;
}
int leave(cramgpu_ctx* c, hipStream_t s) {
    HIPCHK(hipEventRecord(c->last_use, s));  // created by cramgpu_open
    c->last_stream = s;
    c->last_valid = true;
    return 0;
}
struct CallScope {  // leave() on every exit path of an entry point
    cramgpu_ctx* c;
    hipStream_t s;
    ~CallScope() { (void)leave(c, s); }
};

// Copies a small host table into the next ring slot and queues its upload on s; *dptr is valid in stream order.
// Call release_small() after the kernels that read it have been launched.
int stage_small(cramgpu_ctx* c, const void* src, size_t bytes, hipStream_t s, uint8_t** dptr, SmallSlot** slot) {
    SmallSlot& sl = c->small[c->small_next];
    c->small_next = (c->small_next + 1) % kSmallSlots;
    if (!sl.ev) HIPCHK(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
    if (sl.pending) {
        HIPCHK(hipEventSynchronize(sl.ev));
        sl.pending = false;
    }
    HIPCHK(sl.h.reserve(bytes));
    HIPCHK(sl.d.reserve(bytes));
    memcpy(sl.h.p, src, bytes);
    HIPCHK(hipMemcpyAsync(sl.d.p, sl.h.p, bytes, hipMemcpyHostToDevice, s));
    *dptr = sl.d.p;
    *slot = &sl;
    return 0;
}
int release_small(SmallSlot* sl, hipStream_t s) {
    HIPCHK(hipEventRecord(sl->ev, s));
    sl->pending = true;
    return 0;
}

// uploads descriptors + LPT permutation, returns device pointers valid in stream order
int stage_blocks(LaunchRes& r, const cramgpu_block* blocks, int32_t n, hipStream_t s, bool encode) {
    if (!r.staged) HIPCHK(hipEventCreateWithFlags(&r.staged, hipEventDisableTiming));
    if (r.staged_pending) {
        HIPCHK(hipEventSynchronize(r.staged));
        r.staged_pending = false;
    }
    HIPCHK(r.h_desc.reserve(n));
    HIPCHK(r.h_perm.reserve(n));
    HIPCHK(r.d_desc.reserve(n));
    HIPCHK(r.d_perm.reserve(n));
    for (int32_t i = 0; i < n; i++) {
        const cramgpu_block& b = blocks[i];
        if (b.in_off < 0 || b.out_off < 0 || (b.in_off % CRAMGPU_ARENA_ALIGN) || (encode && (b.out_off % CRAMGPU_ARENA_ALIGN)))
            return fail(CRAMGPU_E_ARG, "block offsets must be non-negative multiples of CRAMGPU_ARENA_ALIGN");
        if (b.in_len < 0 || b.out_cap < 0) return fail(CRAMGPU_E_ARG, "negative block size");
    }
    memcpy(r.h_desc.p, blocks, (size_t)n * sizeof(cramgpu_block));
    std::iota(r.h_perm.p, r.h_perm.p + n, 0);
    std::stable_sort(r.h_perm.p, r.h_perm.p + n,
                     [&](int32_t x, int32_t y) { return blocks[x].in_len > blocks[y].in_len; });
    HIPCHK(hipMemcpyAsync(r.d_desc.p, r.h_desc.p, (size_t)n * sizeof(cramgpu_block), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(r.d_perm.p, r.h_perm.p, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIPCHK(hipEventRecord(r.staged, s));
    r.staged_pending = true;
    return 0;
}

// The work-queue word(s) of a launch: 8 bytes of a ring that is cleared when it is made; every launch clears its slot
// behind itself (stream order), so no fill runs in front of a kernel.
int next_ticket(LaunchRes& r, hipStream_t s, uint32_t** out) {
    if (!r.tickets.p) {
        HIPCHK(r.tickets.reserve(kTicketSlots));
        HIPCHK(hipMemsetAsync(r.tickets.p, 0, r.tickets.cap * sizeof(uint32_t), s));
    }
    *out = r.tickets.p + 2 * (r.ticket_next++ % (kTicketSlots / 2));
    return 0;
}

int rans_run(cramgpu_ctx* c, LaunchRes& r, bool encode, const uint8_t* d_in, uint8_t* d_out, const cramgpu_block* blocks,
             int32_t n, int32_t* d_out_len, int64_t* d_out_pos, hipStream_t s) {
    if (n == 0) return 0;
    HIPCHK(hipSetDevice(c->device));
    int rc = stage_blocks(r, blocks, n, s, encode);
    if (rc) return rc;
    // resident workgroups of this launch and their global scratch (grown on demand: a job of a few containers needs
    // a few workgroups' worth, the synchronous calls of a whole read set all of them)
    // Resident workgroups of this launch.  A block is one serial chain, so the launch cannot end before its longest block
    // does: more workgroups than (total work / longest block) only hold LDS and wave slots that the launches of other
    // job slots could use (a job of 256 containers has 256 long blocks among 2560: 1024 workgroups of it kept the next
    // job's kernel waiting for a whole chain latency).  Work of a block ~ its bytes + a fixed cost of ~40 KB worth.
    int groups = std::min(n, encode ? c->n_waves_enc : c->n_waves_dec);
    {
        const int64_t fixed = 40 * 1024;
        int64_t total = 0, longest = 1;
        for (int32_t i = 0; i < n; i++) {
            const int64_t w = (int64_t)blocks[i].in_len * (encode ? 1 : 4) + fixed;  // a stream decodes to ~4-8x its size
            total += w;
            longest = std::max(longest, w);
        }
        const int64_t need = (total + longest - 1) / longest;
        groups = (int)std::min<int64_t>(groups, need + need / 8 + 8);
    }
    if (encode)
        HIPCHK(r.enc_gs.reserve((size_t)((groups + 3) & ~3) * ENC_GSCRATCH_BYTES));  // four slots to a workgroup
    else
        HIPCHK(r.dec_gs.reserve((size_t)groups * DEC_GSCRATCH_BYTES));
    uint32_t* ticket = nullptr;
    if (int trc = next_ticket(r, s, &ticket)) return trc;
    RansArgs a{};
    a.in = d_in;
    a.out = d_out;
    a.desc = r.d_desc.p;
    a.perm = r.d_perm.p;
    a.n_blocks = n;
    a.out_len = d_out_len;
    a.out_pos = d_out_pos;
    a.ticket = ticket;
    a.rcp_tab = c->rcp_tab.p;
    a.tab_bytes = encode ? c->tab_bytes : c->dec_tab_bytes;
    a.prof = nullptr;
    a.simd_arrivals = nullptr;
    if (!encode && c->dec_roles) {
        // the arrival table holds the number of the last launch whose wave came by (atomicMax): cleared once, when it is
        // made — no fill in front of the kernel, which would let a kernel launched beside it on another stream (the gunzip
        // of the same batch) reach the CUs first and keep the decoder's resident waves out for the length of its streams
        if (!r.simd_arrivals.p) {
            HIPCHK(r.simd_arrivals.reserve(DEC_SIMD_KEYS));
            HIPCHK(hipMemsetAsync(r.simd_arrivals.p, 0, r.simd_arrivals.cap * sizeof(uint32_t), s));
        }
        a.simd_arrivals = r.simd_arrivals.p;
        a.launch_no = ++r.launch_no;
    }
    if (encode && c->prof_on) {  // profiling aid: per-block cycle counters, printed after a synchronising copy
        HIPCHK(c->d_prof.reserve((size_t)n * 8));
        HIPCHK(hipMemsetAsync(c->d_prof.p, 0, (size_t)n * 8 * sizeof(unsigned long long), s));
        a.prof = c->d_prof.p;
    }
    if (encode) {
        a.gscratch = r.enc_gs.p;
        HIPCHK(launch_rans_encode(a, groups, c->enc_lds, s));
        if (a.prof) {
            std::vector<unsigned long long> h((size_t)n * 8);
            HIPCHK(hipMemcpyAsync(h.data(), a.prof, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            int32_t big = 0;
            for (int32_t i = 1; i < n; i++)
                if (blocks[i].in_len > blocks[big].in_len) big = i;
            const unsigned long long* r = h.data() + (size_t)big * 8;
            {   // when the blocks began and ended (100 MHz stamps), how long their phases took
                unsigned long long t_first = ~0ull, t_last = 0, pmax = 0, smax = 0, smin = ~0ull;
                double psum = 0, ssum = 0;
                int cnt = 0, late = 0;
                for (int32_t i = 0; i < n; i++) {
                    const unsigned long long* q = h.data() + (size_t)i * 8;
                    if (blocks[i].in_len < blocks[big].in_len / 2 || !q[7]) continue;
                    const unsigned long long dur = (unsigned long long)((double)(q[0] + q[1]) * (double)q[6] / (double)(q[1] ? q[1] : 1));
                    t_first = std::min(t_first, q[7] - dur);
                    t_last = std::max(t_last, q[7]);
                    pmax = std::max(pmax, q[0]);
                    smax = std::max(smax, q[1]);
                    smin = std::min(smin, q[1]);
                    psum += (double)q[0];
                    ssum += (double)q[1];
                    cnt++;
                }
                for (int32_t i = 0; i < n; i++) {
                    const unsigned long long* q = h.data() + (size_t)i * 8;
                    if (blocks[i].in_len < blocks[big].in_len / 2 || !q[7]) continue;
                    const unsigned long long dur = (unsigned long long)((double)(q[0] + q[1]) * (double)q[6] / (double)(q[1] ? q[1] : 1));
                    if (q[7] - dur > t_first + 50000) late++;  // began more than 0.5 ms after the first
                }
                if (cnt)
                    fprintf(stderr, "[cramgpu prof] %d long blocks: first begins to last ends %.3f ms; prep mean %.0f max %llu cyc; stream mean %.0f "
                                    "min %llu max %llu cyc; %d began > 0.5 ms after the first\n",
                            cnt, (double)(t_last - t_first) * 1e-5, psum / cnt, pmax, ssum / cnt, smin, smax, late);
            }
            // HW_ID: wave 3:0, SIMD 5:4, CU 11:8, SH 12, SE 15:13
            fprintf(stderr, "[cramgpu prof] largest block %d B: prep %llu cyc, stream %llu cyc, helper busy %llu, batches %llu "
                            "(per step: stream %.1f, helper %.1f); chain wave on SIMD %llu, helper on SIMD %llu (CU %llu / %llu); "
                            "batch-times the chain row stood still during this block: %llu; shader clock over phase 3: %.0f MHz\n",
                    blocks[big].in_len, r[0], r[1], r[2], r[4], r[4] ? (double)r[1] / (64.0 * r[4]) : 0.0,
                    r[4] ? (double)r[2] / (64.0 * r[4]) : 0.0, (r[3] >> 36) & 3, (r[3] >> 4) & 3, (r[3] >> 40) & 15, (r[3] >> 8) & 15, r[5], r[6] ? 100.0 * (double)r[1] / (double)r[6] : 0.0);
        }
    } else {
        a.gscratch = r.dec_gs.p;
        HIPCHK(launch_rans_decode(a, groups, c->dec_lds, s));
    }
    HIPCHK(hipMemsetAsync(ticket, 0, 2 * sizeof(uint32_t), s));  // behind the launch: the slot is clean when its turn comes again
    return 0;
}

// a free job slot of the context, its stream and events created on first use
cramgpu_job* grab_job(cramgpu_ctx* c, int kind) {  // under c->mu
    cramgpu_job* j = nullptr;
    for (auto& cand : c->jobs)
        if (!cand.busy) {
            j = &cand;
            break;
        }
    if (!j) {
        fail(CRAMGPU_E_ARG, "submit: CRAMGPU_MAX_JOBS jobs outstanding, wait for one first");
        return nullptr;
    }
    auto ok = [&](hipError_t e) { return e == hipSuccess || (fail(CRAMGPU_E_HIP, hipGetErrorString(e)), false); };
    if (!ok(hipSetDevice(c->device))) return nullptr;
    if (!j->coded && !(ok(hipEventCreateWithFlags(&j->coded, hipEventDisableTiming)) &&
                       ok(hipEventCreateWithFlags(&j->done, hipEventDisableTiming)) &&
                       ok(hipEventCreateWithFlags(&j->uploaded, hipEventDisableTiming)) &&
                       ok(hipStreamCreateWithFlags(&j->stream, hipStreamNonBlocking))))
        return nullptr;
    if (!c->upload_stream && !ok(hipStreamCreateWithFlags(&c->upload_stream, hipStreamNonBlocking))) return nullptr;
    j->ctx = c;
    j->kind = kind;
    j->fetched = false;
    j->early_rc = 0;
    j->back.clear();
    j->h_total = nullptr;
    j->h_feat_count = nullptr;
    j->feat_cap = 0;
    return j;
}

std::mutex g_default_mu;
cramgpu_ctx* g_default = nullptr;

cramgpu_ctx* default_ctx() {
    std::lock_guard<std::mutex> lk(g_default_mu);
    if (!g_default) g_default = cramgpu_open(0);
    return g_default;
}

int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) & ~(a - 1); }

}  // namespace

extern "C" {

const char* cramgpu_last_error(void) { return g_err.c_str(); }

int32_t cramgpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

cramgpu_ctx* cramgpu_open(int32_t device_index) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_index < 0 || device_index >= n) {
        fail(CRAMGPU_E_NO_DEVICE, "cramgpu_open: no such HIP device (libcramgpu has no CPU path)");
        return nullptr;
    }
    if (hipSetDevice(device_index) != hipSuccess) {
        fail(CRAMGPU_E_HIP, "hipSetDevice failed");
        return nullptr;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_index) != hipSuccess) {
        fail(CRAMGPU_E_HIP, "hipGetDeviceProperties failed");
        return nullptr;
    }
    cramgpu_ctx* c = new cramgpu_ctx();
    c->device = device_index;
    c->n_cu = prop.multiProcessorCount;
    records_set_cu_count(c->n_cu);
    const int lds_per_cu = 160 * 1024;
    // LDS table region of a resident rANS workgroup: 4 encoder groups (2 waves each) per CU fill the 160 KiB
    c->tab_bytes = env_int("CRAMGPU_TAB_BYTES", 35 * 1024) & ~15;
    if (c->tab_bytes < 20 * 1024) c->tab_bytes = 20 * 1024;  // order-0 decode needs a 16 KiB slot table
    c->enc_lds = 0;  // static LDS, sized for 4 workgroups per CU (rans_enc.hip)
    // decoder: 16 KiB = the order-0 slot table; order-1 rows of alphabets up to 64 symbols fit the same region
    c->dec_tab_bytes = std::max(16 * 1024, env_int("CRAMGPU_DEC_TAB_BYTES", 16 * 1024) & ~15);
    c->dec_lds = rans_decode_fixed_lds() + c->dec_tab_bytes;
    c->enc_wpc = std::max(1, std::min(env_int("CRAMGPU_ENC_GROUPS_PER_CU", 4), 4));
    c->dec_wpc = std::max(1, std::min(env_int("CRAMGPU_DEC_WAVES_PER_CU", 5), (int)(lds_per_cu / c->dec_lds)));
    c->dec_roles = env_int("CRAMGPU_DEC_ROLES", 1) != 0;
    c->prof_on = env_int("CRAMGPU_PROF", 0) != 0;
    // ask for a full share of the CU's LDS so that the dispatcher cannot stack more than dec_wpc decoder waves on one
    // CU while others hold fewer: the launch lasts as long as its slowest (most crowded) chain
    if (env_int("CRAMGPU_DEC_LDS_SHARE", 1)) {
        // ... minus a reserve for kernels that run beside the decoder on other streams (the gunzip of the same batch's GZIP
        // series: 7.4 KiB per two streams) — as much as still keeps dec_wpc + 1 shares from fitting a CU
        size_t share = (size_t)(lds_per_cu / c->dec_wpc - 1024) & ~(size_t)255;
        const size_t floor_share = ((size_t)lds_per_cu / (c->dec_wpc + 1) + 512) & ~(size_t)255;
        const size_t reserve = (size_t)std::max(0, env_int("CRAMGPU_DEC_LDS_RESERVE", 24 * 1024));
        const size_t lean = (((size_t)lds_per_cu - std::min(reserve, (size_t)lds_per_cu / 2)) / c->dec_wpc) & ~(size_t)255;
        share = std::min(share, std::max(lean, floor_share));
        c->dec_lds = std::max(c->dec_lds, share);
        // the share is there anyway: the table region takes all of it (ragged order-1 rows of wide alphabets)
        c->dec_tab_bytes = (int)((c->dec_lds - rans_decode_fixed_lds()) & ~(size_t)15);
    }
    c->n_waves_enc = c->n_cu * c->enc_wpc;
    c->n_waves_dec = c->n_cu * c->dec_wpc;
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess &&
              c->rcp_tab.reserve(4096) == hipSuccess &&
              c->d_err.reserve(16) == hipSuccess && c->d_total.reserve(16) == hipSuccess &&
              c->h_total.reserve(16) == hipSuccess;
    if (ok) ok = launch_rcp_init(c->rcp_tab.p, c->stream) == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess;
    // the event that orders calls on different streams, and this device's address of the static scratch, up front: the
    // allocation-free entry point creates and looks up nothing
    if (ok) ok = hipEventCreateWithFlags(&c->last_use, hipEventDisableTiming) == hipSuccess;
    if (ok) {
        void* p = nullptr;
        ok = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) == hipSuccess && p != nullptr;
        c->static_scratch = static_cast<uint8_t*>(p);
    }
    if (!ok) {
        fail(CRAMGPU_E_HIP, "cramgpu_open: allocation or init kernel failed", hipGetLastError());
        cramgpu_close(c);
        return nullptr;
    }
    return c;
}

void cramgpu_close(cramgpu_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    c->res.release(); c->rcp_tab.release();
    c->d_in.release(); c->d_out.release(); c->d_pack.release(); c->d_len.release(); c->d_pos.release();
    c->d_off.release(); c->d_err.release(); c->h_stage.release(); c->h_len.release(); c->h_off.release();
    c->d_mate_prev.release(); c->d_mate_serial.release();
    c->d_series_alias.release(); c->d_series_vals.release(); c->d_series_cnt.release(); c->d_series_off.release(); c->d_series_base.release();
    c->d_counts.release(); c->d_list.release(); c->d_total.release(); c->d_scan_ws.release(); c->d_idx_ws.release(); c->h_total.release(); c->d_maskws.release(); c->d_codes25.release(); c->d_prof.release();
    for (auto& sl : c->small) {
        sl.h.release();
        sl.d.release();
        if (sl.ev) (void)hipEventDestroy(sl.ev);
    }
    if (c->last_use) (void)hipEventDestroy(c->last_use);
    if (c->gz_use) (void)hipEventDestroy(c->gz_use);
    c->d_gz_ws.release();
    for (auto& g : c->contigs) g.d.release();
    for (auto& j : c->jobs) {
        j.arena.release();
        j.pin.release();
        j.d_in.release(); j.d_out.release(); j.d_len.release(); j.d_pos.release(); j.h_len.release(); j.h_pos.release();
        j.d_pack.release(); j.d_off.release(); j.d_err.release();
        j.res.release();
        if (j.stream) (void)hipStreamDestroy(j.stream);
        if (j.uploaded) (void)hipEventDestroy(j.uploaded);
        if (j.coded) (void)hipEventDestroy(j.coded);
        if (j.done) (void)hipEventDestroy(j.done);
    }
    if (c->upload_stream) (void)hipStreamDestroy(c->upload_stream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int32_t cramgpu_info(cramgpu_ctx* c, int32_t* n_cu, int32_t* n_waves) {
    if (!c) return fail(CRAMGPU_E_ARG, "null context");
    if (n_cu) *n_cu = c->n_cu;
    if (n_waves) *n_waves = c->n_waves_enc;
    return 0;
}

int32_t cramgpu_rans_bound(int32_t n) { return (int32_t)(1.05 * (double)n + 257.0 * 257.0 * 3.0 + 4.0); }

int32_t cramgpu_rans_raw_size(const uint8_t* in, int32_t n) {
    if (n == 0) return 0;
    if (!in || n < 9) return CRAMGPU_E_INPUT_LENGTH;
    return (int32_t)((uint32_t)in[5] | ((uint32_t)in[6] << 8) | ((uint32_t)in[7] << 16) | ((uint32_t)in[8] << 24));
}

int32_t cramgpu_rans_encode_dev(cramgpu_ctx* c, const uint8_t* d_in, uint8_t* d_out, const cramgpu_block* blocks,
                                int32_t n, int32_t* d_out_len, int64_t* d_out_pos, void* stream) {
    if (!c || n < 0 || (n > 0 && (!d_in || !d_out || !blocks || !d_out_len || !d_out_pos)))
        return fail(CRAMGPU_E_ARG, "cramgpu_rans_encode_dev: bad argument");
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t s = pick(c, stream);
    if (int rc = enter(c, s)) return rc;
    CallScope scope{c, s};
    return rans_run(c, c->res, true, d_in, d_out, blocks, n, d_out_len, d_out_pos, s);
}

int32_t cramgpu_stream_wait_staged(cramgpu_ctx* c, void* stream) {
    if (!c) return fail(CRAMGPU_E_ARG, "cramgpu_stream_wait_staged: bad argument");
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->res.staged && c->res.staged_pending) {
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipStreamWaitEvent((hipStream_t)stream, c->res.staged, 0));
    }
    return 0;
}

int32_t cramgpu_rans_encode_series_dev(cramgpu_ctx* c, const uint8_t* d_base, int64_t in_shift, const int64_t* d_series_off,
                                       const int32_t* d_series_len, const int32_t* d_w_status, int32_t n_cont,
                                       const int32_t* ids, const int32_t* orders, int32_t n_ids, uint8_t* d_out,
                                       int64_t out_cap, cramgpu_block* d_blocks, int32_t* d_out_len, int64_t* d_out_pos,
                                       int32_t* d_status, void* stream) {
    if (!c || n_cont < 0 || n_ids < 0 || n_ids > CRAMGPU_N_SERIES || (int64_t)n_cont * n_ids > INT32_MAX ||
        (n_cont > 0 && n_ids > 0 && (!d_base || !d_series_off || !d_series_len || !ids || !orders || !d_out || !d_blocks ||
                                     !d_out_len || !d_out_pos || !d_status)))
        return fail(CRAMGPU_E_ARG, "cramgpu_rans_encode_series_dev: bad argument");
    if ((in_shift % CRAMGPU_ARENA_ALIGN) != 0) return fail(CRAMGPU_E_ARG, "cramgpu_rans_encode_series_dev: in_shift must be a multiple of CRAMGPU_ARENA_ALIGN");
    SeriesPick sel{};
    sel.n_ids = n_ids;
    for (int32_t k = 0; k < n_ids; k++) {
        if (ids[k] < 0 || ids[k] >= CRAMGPU_N_SERIES) return fail(CRAMGPU_E_ARG, "cramgpu_rans_encode_series_dev: unknown series id");
        if (orders[k] != 0 && orders[k] != 1) return fail(CRAMGPU_E_ORDER, "Unknown rANS order");
        sel.ids[k] = ids[k];
        sel.orders[k] = orders[k];
    }
    const int32_t n = n_cont * n_ids;
    if (n == 0) return 0;
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t s = pick(c, stream);
    if (int rc = enter(c, s)) return rc;
    CallScope scope{c, s};
    HIPCHK(hipSetDevice(c->device));
    LaunchRes& r = c->res;
    const int groups = std::min(n, c->n_waves_enc);
    const size_t ws_bytes = lpt_sort_bytes(n);
    HIPCHK(r.d_perm.reserve(n));
    HIPCHK(r.sort_ws.reserve(ws_bytes));
    HIPCHK(r.enc_gs.reserve((size_t)((groups + 3) & ~3) * ENC_GSCRATCH_BYTES));
    // the table, its longest-first order, the encoder: all in stream order, nothing comes back to the host in between
    HIPCHK(launch_series_blocks(d_series_off, d_series_len, n_cont, sel, in_shift, out_cap, d_w_status, d_blocks, d_status, s));
    HIPCHK(launch_lpt_sort(d_blocks, n, r.d_perm.p, r.sort_ws.p, ws_bytes, s));
    uint32_t* ticket = nullptr;
    if (int trc = next_ticket(r, s, &ticket)) return trc;
    RansArgs a{};
    a.in = d_base;
    a.out = d_out;
    a.desc = d_blocks;
    a.perm = r.d_perm.p;
    a.n_blocks = n;
    a.out_len = d_out_len;
    a.out_pos = d_out_pos;
    a.ticket = ticket;
    a.rcp_tab = c->rcp_tab.p;
    a.tab_bytes = c->tab_bytes;
    a.prof = nullptr;
    a.gscratch = r.enc_gs.p;
    HIPCHK(launch_rans_encode(a, groups, c->enc_lds, s));
    HIPCHK(hipMemsetAsync(ticket, 0, 2 * sizeof(uint32_t), s));
    return 0;
}

int32_t cramgpu_rans_encode_static(cramgpu_ctx* c, const uint8_t* d_in, uint8_t* d_out, const cramgpu_block* d_blocks,
                                   int32_t n, int32_t* d_out_len, int64_t* d_out_pos, void* stream) {
    if (!c || n < 0 || (n > 0 && (!d_in || !d_out || !d_blocks || !d_out_len || !d_out_pos)))
        return fail(CRAMGPU_E_ARG, "cramgpu_rans_encode_static: bad argument");
    if (n > kStaticMaxBlocks)
        return fail(CRAMGPU_E_ARG, "cramgpu_rans_encode_static: at most 4096 blocks per call (the static scratch holds no more)");
    if (n == 0) return 0;
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t s = pick(c, stream);
    if (int rc = enter(c, s)) return rc;
    CallScope scope{c, s};
    // everything below works on this device's copy of the static array, through the address taken at cramgpu_open
    uint8_t* base = c->static_scratch;
    uint32_t* ticket = reinterpret_cast<uint32_t*>(base);
    int32_t* perm = reinterpret_cast<int32_t*>(base + kStaticPermOff);
    HIPCHK(hipMemsetAsync(base, 0, kStaticPermOff, s));  // the ticket must start at zero; the rest is rebuilt per call
    HIPCHK(launch_lpt_perm(d_blocks, n, perm, s));
    RansArgs a{};
    a.in = d_in;
    a.out = d_out;
    a.desc = d_blocks;
    a.perm = perm;
    a.n_blocks = n;
    a.out_len = d_out_len;
    a.out_pos = d_out_pos;
    a.ticket = ticket;
    a.rcp_tab = c->rcp_tab.p;
    a.tab_bytes = c->tab_bytes;
    a.prof = nullptr;
    a.gscratch = base + kStaticGsOff;
    HIPCHK(launch_rans_encode(a, std::min(n, kStaticGroups), c->enc_lds, s));
    return 0;
}

int32_t cramgpu_rans_decode_dev(cramgpu_ctx* c, const uint8_t* d_in, uint8_t* d_out, const cramgpu_block* blocks,
                                int32_t n, int32_t* d_out_len, void* stream) {
    if (!c || n < 0 || (n > 0 && (!d_in || !d_out || !blocks || !d_out_len)))
        return fail(CRAMGPU_E_ARG, "cramgpu_rans_decode_dev: bad argument");
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t s = pick(c, stream);
    if (int rc = enter(c, s)) return rc;
    CallScope scope{c, s};
    return rans_run(c, c->res, false, d_in, d_out, blocks, n, d_out_len, nullptr, s);
}

int32_t cramgpu_gather_dev(cramgpu_ctx* c, const uint8_t* d_src, const int64_t* d_pos, const int32_t* d_len, int32_t n,
                           int32_t max_len, int32_t align, uint8_t* d_dst, int64_t dst_cap, int64_t* d_dst_off,
                           void* stream) {
    if (!c || n < 0 || max_len < 0 || align < 1 || (align & (align - 1)) || (n > 0 && (!d_src || !d_pos || !d_len || !d_dst || !d_dst_off)))
        return fail(CRAMGPU_E_ARG, "cramgpu_gather_dev: bad argument");
    if (n == 0) return 0;
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t s = pick(c, stream);
    if (int rc = enter(c, s)) return rc;
    CallScope scope{c, s};
    HIPCHK(launch_gather(d_src, d_pos, d_len, n, align, d_dst, dst_cap, d_dst_off, c->d_err.p, max_len, s));
    return 0;
}

// ---- host-arena batches ----
static int32_t host_batch(cramgpu_ctx* c, bool encode, const uint8_t* in_arena, int64_t in_bytes, uint8_t* out_arena,
                          int64_t out_bytes, const cramgpu_block* blocks, int32_t n, int32_t* out_len) {
    if (!c || n < 0 || in_bytes < 0 || out_bytes < 0 || (n > 0 && (!in_arena || !out_arena || !blocks || !out_len)))
        return fail(CRAMGPU_E_ARG, "batch: bad argument");
    if (n == 0) return 0;
    int64_t max_cap = 0;
    for (int32_t i = 0; i < n; i++) {
        if (blocks[i].in_off + blocks[i].in_len > in_bytes || blocks[i].out_off + blocks[i].out_cap > out_bytes)
            return fail(CRAMGPU_E_ARG, "batch: block outside its arena");
        max_cap = std::max<int64_t>(max_cap, blocks[i].out_cap);
    }
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t s = c->stream;
    if (int rc = enter(c, s)) return rc;
    CallScope scope{c, s};
    HIPCHK(c->d_in.reserve((size_t)align_up(in_bytes, 16) + 16));
    HIPCHK(c->d_out.reserve((size_t)align_up(out_bytes, 16) + 16));
    HIPCHK(c->d_len.reserve(n));
    HIPCHK(c->h_len.reserve(n));
    HIPCHK(hipMemcpyAsync(c->d_in.p, in_arena, (size_t)in_bytes, hipMemcpyHostToDevice, s));
    if (!encode) {
        int rc = rans_run(c, c->res, false, c->d_in.p, c->d_out.p, blocks, n, c->d_len.p, nullptr, s);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(c->h_len.p, c->d_len.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        // copy back each decoded block (ranges may be sparse inside the arena)
        for (int32_t i = 0; i < n; i++) {
            out_len[i] = c->h_len.p[i];
            if (out_len[i] > 0)
                HIPCHK(hipMemcpyAsync(out_arena + blocks[i].out_off, c->d_out.p + blocks[i].out_off, (size_t)out_len[i],
                                      hipMemcpyDeviceToHost, s));
        }
        HIPCHK(hipStreamSynchronize(s));
        return 0;
    }
    HIPCHK(c->d_pos.reserve(n));
    HIPCHK(c->d_off.reserve((size_t)n + 1));
    HIPCHK(c->h_off.reserve((size_t)n + 1));
    int rc = rans_run(c, c->res, true, c->d_in.p, c->d_out.p, blocks, n, c->d_len.p, c->d_pos.p, s);
    if (rc) return rc;
    // pack on the device so that only compressed bytes cross PCIe
    HIPCHK(c->d_pack.reserve((size_t)align_up(out_bytes, 16) + (size_t)n * 16 + 16));
    HIPCHK(launch_gather(c->d_out.p, c->d_pos.p, c->d_len.p, n, 16, c->d_pack.p, (int64_t)c->d_pack.cap, c->d_off.p,
                         c->d_err.p, (int32_t)std::min<int64_t>(max_cap, 0x7fffffff), s));
    HIPCHK(hipMemcpyAsync(c->h_len.p, c->d_len.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(c->h_off.p, c->d_off.p, ((size_t)n + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const int64_t packed = c->h_off.p[n];
    // gather_scan_kernel skips every copy when the packed size exceeds the destination (overlapping slots can make
    // it larger than out_bytes + 16 n): the bytes of d_pack would be stale
    if (packed > (int64_t)c->d_pack.cap) return fail(CRAMGPU_E_INSUFFICIENT, "batch: packed streams exceed the staging buffer (overlapping output slots?)");
    HIPCHK(c->h_stage.reserve((size_t)packed + 16));
    HIPCHK(hipMemcpyAsync(c->h_stage.p, c->d_pack.p, (size_t)packed, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (int32_t i = 0; i < n; i++) {
        out_len[i] = c->h_len.p[i];
        if (out_len[i] > 0) memcpy(out_arena + blocks[i].out_off, c->h_stage.p + c->h_off.p[i], (size_t)out_len[i]);
    }
    return 0;
}

int32_t cramgpu_rans_encode_batch(cramgpu_ctx* c, const uint8_t* in_arena, int64_t in_bytes, uint8_t* out_arena,
                                  int64_t out_bytes, const cramgpu_block* blocks, int32_t n, int32_t* out_len) {
    return host_batch(c, true, in_arena, in_bytes, out_arena, out_bytes, blocks, n, out_len);
}
int32_t cramgpu_rans_decode_batch(cramgpu_ctx* c, const uint8_t* in_arena, int64_t in_bytes, uint8_t* out_arena,
                                  int64_t out_bytes, const cramgpu_block* blocks, int32_t n, int32_t* out_len) {
    return host_batch(c, false, in_arena, in_bytes, out_arena, out_bytes, blocks, n, out_len);
}

// ---- asynchronous level 1 ----
void* cramgpu_host_alloc(int64_t bytes) {
    void* p = nullptr;
    if (bytes <= 0 || hipHostMalloc(&p, (size_t)bytes, hipHostMallocPortable) != hipSuccess) return nullptr;  // pinned for every GPU (device pool)
    return p;
}
void cramgpu_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

static cramgpu_job* rans_submit(cramgpu_ctx* c, int32_t decode, const uint8_t* in_arena, int64_t in_bytes, uint8_t* out_arena,
                                int64_t out_bytes, const cramgpu_block* blocks, int32_t n, int32_t* out_len, int64_t* out_pos) {
    if (!c || n <= 0 || in_bytes < 0 || out_bytes < 0 || !in_arena || !out_arena || !blocks || !out_len) {
        fail(CRAMGPU_E_ARG, "submit: bad argument");
        return nullptr;
    }
    int64_t max_cap = 0, lo = out_bytes, hi = 0, prev_end = 0;
    for (int32_t i = 0; i < n; i++) {
        if (blocks[i].in_off + blocks[i].in_len > in_bytes || blocks[i].out_off + blocks[i].out_cap > out_bytes) {
            fail(CRAMGPU_E_ARG, "submit: block outside its arena");
            return nullptr;
        }
        // the streams come back packed and are spread to their slots from the last one down, in place: the slots have
        // to lie in ascending order without overlap (Engine.plan_encode / any back-to-back layout)
        if (!decode && !out_pos && blocks[i].out_off < prev_end) {
            fail(CRAMGPU_E_ARG, "submit: output slots must be ascending and must not overlap");
            return nullptr;
        }
        prev_end = blocks[i].out_off + blocks[i].out_cap;
        max_cap = std::max<int64_t>(max_cap, blocks[i].out_cap);
        lo = std::min<int64_t>(lo, blocks[i].out_off);
        hi = std::max<int64_t>(hi, blocks[i].out_off + blocks[i].out_cap);
    }
    std::lock_guard<std::mutex> lk(c->mu);
    cramgpu_job* j = grab_job(c, 0);
    if (!j) return nullptr;
    auto ok = [&](hipError_t e) { return e == hipSuccess || (fail(CRAMGPU_E_HIP, hipGetErrorString(e)), false); };
    hipStream_t s = j->stream;  // every job slot has its own stream: the launches of outstanding jobs overlap
    if (!(ok(j->d_in.reserve((size_t)align_up(in_bytes, 16) + 16)) && ok(j->d_out.reserve((size_t)align_up(out_bytes, 16) + 16)) &&
          ok(j->d_len.reserve(n)) && ok(j->d_pos.reserve(n)) && ok(j->h_len.reserve(n)) && ok(j->h_pos.reserve((size_t)n + 1))))
        return nullptr;
    j->decode = decode != 0;
    j->n = n;
    j->out_arena = out_arena;
    j->out_len = out_len;
    j->out_bytes = out_bytes;
    j->out_pos = out_pos;
    j->span_lo = lo;
    j->span_hi = hi;
    j->blocks.assign(blocks, blocks + n);
    if (!(ok(hipMemcpyAsync(j->d_in.p, in_arena, (size_t)in_bytes, hipMemcpyHostToDevice, c->upload_stream)) &&
          ok(hipEventRecord(j->uploaded, c->upload_stream)) && ok(hipStreamWaitEvent(s, j->uploaded, 0))))
        return nullptr;
    if (rans_run(c, j->res, !j->decode, j->d_in.p, j->d_out.p, blocks, n, j->d_len.p, j->decode ? nullptr : j->d_pos.p, s))
        return nullptr;
    if (!ok(hipMemcpyAsync(j->h_len.p, j->d_len.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, s))) return nullptr;
    if (!j->decode) {
        // pack on the device: one copy instead of one per block brings the streams back (2560 small copies of a
        // 256-container job cost more host time than the job's kernels took)
        if (!(ok(j->d_pack.reserve((size_t)align_up(out_bytes, 16) + 16)) && ok(j->d_off.reserve((size_t)n + 1)) &&
              ok(j->d_err.reserve(16))))
            return nullptr;
        if (!ok(launch_gather(j->d_out.p, j->d_pos.p, j->d_len.p, n, 16, j->d_pack.p, out_bytes, j->d_off.p, j->d_err.p,
                              (int32_t)std::min<int64_t>(max_cap, 0x7fffffff), s)))
            return nullptr;
        if (!ok(hipMemcpyAsync(j->h_pos.p, j->d_off.p, ((size_t)n + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, s))) return nullptr;
    }
    if (!ok(hipEventRecord(j->coded, s))) return nullptr;
    j->busy = true;
    return j;
}

cramgpu_job* cramgpu_rans_submit(cramgpu_ctx* c, int32_t decode, const uint8_t* in_arena, int64_t in_bytes, uint8_t* out_arena,
                                 int64_t out_bytes, const cramgpu_block* blocks, int32_t n, int32_t* out_len) {
    return rans_submit(c, decode, in_arena, in_bytes, out_arena, out_bytes, blocks, n, out_len, nullptr);
}

cramgpu_job* cramgpu_rans_submit_packed(cramgpu_ctx* c, const uint8_t* in_arena, int64_t in_bytes, uint8_t* out_arena,
                                        int64_t out_bytes, const cramgpu_block* blocks, int32_t n, int32_t* out_len,
                                        int64_t* out_pos) {
    if (!out_pos) {
        fail(CRAMGPU_E_ARG, "submit_packed: out_pos missing");
        return nullptr;
    }
    return rans_submit(c, 0, in_arena, in_bytes, out_arena, out_bytes, blocks, n, out_len, out_pos);
}

// second half of a job: once the lengths are on the host, ONE copy brings the results into the caller's arena —
// the packed streams (encode; spread to their slots by job_wait) or the span of the raw blocks (decode)
static int job_fetch(cramgpu_job* j) {
    if (j->fetched) return 0;
    cramgpu_ctx* c = j->ctx;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipEventSynchronize(j->coded));
    if (j->kind != 0) {  // record job: the feature count is on the host now, the results follow
        const int64_t total = j->h_total ? *j->h_total : 0;
        if (j->h_feat_count) *j->h_feat_count = total;
        const bool fits = total <= j->feat_cap;
        if (!fits && !j->early_rc) j->early_rc = fail(CRAMGPU_E_INSUFFICIENT, "record job: feature capacity too small (feats.off[n_reads] = capacity needed)");
        for (const CopyBack& b : j->back) {
            if (b.per_feature && !fits) continue;
            const size_t bytes = b.per_feature ? b.bytes * (size_t)total : b.bytes;
            if (bytes) HIPCHK(hipMemcpyAsync(b.dst, b.src, bytes, hipMemcpyDeviceToHost, j->stream));
        }
        HIPCHK(hipEventRecord(j->done, j->stream));
        j->fetched = true;
        return 0;
    }
    for (int32_t i = 0; i < j->n; i++) j->out_len[i] = j->h_len.p[i];
    if (j->decode) {
        if (j->span_hi > j->span_lo)
            HIPCHK(hipMemcpyAsync(j->out_arena + j->span_lo, j->d_out.p + j->span_lo, (size_t)(j->span_hi - j->span_lo),
                                  hipMemcpyDeviceToHost, j->stream));
    } else {
        const int64_t packed = j->h_pos.p[j->n];
        if (packed > j->out_bytes) return fail(CRAMGPU_E_INSUFFICIENT, "job: packed streams exceed the output arena");
        if (packed > 0) HIPCHK(hipMemcpyAsync(j->out_arena, j->d_pack.p, (size_t)packed, hipMemcpyDeviceToHost, j->stream));
    }
    HIPCHK(hipEventRecord(j->done, j->stream));
    j->fetched = true;
    return 0;
}

int32_t cramgpu_job_poll(cramgpu_job* j) {
    if (!j || !j->busy) return fail(CRAMGPU_E_ARG, "poll: no such job");
    std::lock_guard<std::mutex> lk(j->ctx->mu);
    if (!j->fetched) {
        const hipError_t q = hipEventQuery(j->coded);
        if (q == hipErrorNotReady) return 0;
        if (q != hipSuccess) return fail(CRAMGPU_E_HIP, hipGetErrorString(q));
        int rc = job_fetch(j);
        if (rc) return rc;
    }
    const hipError_t q = hipEventQuery(j->done);
    if (q == hipErrorNotReady) return 0;
    return q == hipSuccess ? 1 : fail(CRAMGPU_E_HIP, hipGetErrorString(q));
}

int32_t cramgpu_job_wait(cramgpu_job* j) {
    if (!j || !j->busy) return fail(CRAMGPU_E_ARG, "wait: no such job");
    std::lock_guard<std::mutex> lk(j->ctx->mu);
    int rc = job_fetch(j);
    if (!rc) {
        const hipError_t e = hipEventSynchronize(j->done);
        if (e != hipSuccess) rc = fail(CRAMGPU_E_HIP, hipGetErrorString(e));
    }
    if (!rc && j->kind != 0) rc = j->early_rc;
    if (!rc && j->kind == 0 && !j->decode && j->out_pos) {
        memcpy(j->out_pos, j->h_pos.p, ((size_t)j->n + 1) * sizeof(int64_t));
    } else if (!rc && j->kind == 0 && !j->decode) {
        // spread the packed streams to the starts of their slots, last one first: stream i sits at or below its slot and
        // everything that still has to move lies below it
        for (int32_t i = j->n - 1; i >= 0; i--) {
            const int32_t len = j->h_len.p[i];
            if (len > 0 && j->blocks[i].out_off != j->h_pos.p[i])
                memmove(j->out_arena + j->blocks[i].out_off, j->out_arena + j->h_pos.p[i], (size_t)len);
        }
    }
    j->busy = false;  // the slot (and its device buffers) can take the next submission
    return rc;
}

}  // extern "C"

// ---- coalescing queue: one container per call, many calls at once ----
// The reference hands ONE container to a worker thread (paralell/TransformerJob.java:29-41 runs a
// BamToCram_OBA_Function / CramToBam_OBA_Function per stripe; CRAMContainerAsynchWriter.java:139-149 a Batch per
// container).  A launch per container costs one chain latency (the container's longest block, ~8 ms for 1.5 MB of
// scores) whatever else the GPU could do meanwhile, so the library commits in groups: submissions queue up, a dispatcher
// thread turns EVERYTHING that is waiting into one launch the moment a batch slot is free, a completer thread brings
// the results back and wakes the callers.  Under load a batch holds all the containers that arrived during the previous
// batch's latency; alone, a submission starts at once.
struct CoSub {
    int64_t id = 0;
    bool decode = false, packed = false;
    const uint8_t* in = nullptr;
    uint8_t* out = nullptr;
    int64_t in_bytes = 0, out_bytes = 0;
    std::vector<cramgpu_block> blocks;
    int32_t* out_len = nullptr;
    int64_t* out_pos = nullptr;
    int state = 0;  // 0 queued, 1 in a batch, 2 finished
    std::chrono::steady_clock::time_point arrived;
    int32_t rc = 0;
    std::string err;
    int64_t in_base = 0, out_base = 0, lo = 0, hi = 0, max_cap = 0;
    int32_t first = 0;
};

struct CoBatch {
    LaunchRes res;
    hipStream_t stream = nullptr;
    hipEvent_t uploaded = nullptr, coded = nullptr, done = nullptr;
    DevBuf<uint8_t> d_in, d_out, d_pack;
    DevBuf<int32_t> d_len, d_err;
    DevBuf<int64_t> d_pos, d_off;
    PinBuf<int32_t> h_len;
    PinBuf<int64_t> h_pos;
    std::vector<CoSub*> subs;
    std::vector<cramgpu_block> blocks;
    bool decode = false;
    int phase = 0;  // 1 launched (waiting for `coded`), 2 results on their way to the callers' arenas (waiting for `done`)
    double t_pick = 0, t_issued = 0, t_coded = 0;  // CRAMGPU_COALESCE_TRACE=1
};

struct cramgpu_coalescer {
    cramgpu_ctx* ctx = nullptr;
    std::mutex mu;
    std::condition_variable cv_work, cv_flight, cv_done;
    std::deque<CoSub*> queue;
    std::unordered_map<int64_t, CoSub*> tickets;
    std::vector<CoBatch> batches;
    std::vector<CoBatch*> free_batches;
    std::deque<CoBatch*> inflight;
    hipStream_t upload_stream = nullptr;
    std::thread dispatcher, completer;
    int64_t next_id = 1;
    bool stop = false;
    // what one batch may hold: bounds the device arenas of a batch slot
    int64_t max_in_bytes = 0;
    int32_t max_subs = 0;
    // While other batches are in flight a lone arrival is held back this long for company: callers woken by one batch
    // come back within a fraction of a millisecond of each other, and a launch of one container ties up a stream (the
    // hardware runs few launches at once) for as long as a launch of fifty.
    int32_t hold_us = 0;
    Graveyard graveyard;
    // counters (cramgpu_coalesce_stats)
    int64_t n_batches = 0, n_subs = 0, largest = 0;
    bool trace = false;  // CRAMGPU_COALESCE_TRACE=1: one line per batch on stderr (host clock, ms since the queue opened)
    std::chrono::steady_clock::time_point t0;
    double now() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

namespace {

size_t co_round(size_t n) { return std::max<size_t>(n, 1 << 20); }

int co_launch(cramgpu_coalescer* q, CoBatch* b) {
    cramgpu_ctx* c = q->ctx;
    HIPCHK(hipSetDevice(c->device));
    if (!b->stream) {
        HIPCHK(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&b->uploaded, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&b->coded, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&b->done, hipEventDisableTiming));
    }
    // arenas of the batch: the submissions' arenas back to back, their descriptors rebased
    int64_t in_run = 0, out_run = 0, max_cap = 0;
    b->blocks.clear();
    for (CoSub* s : b->subs) {
        s->in_base = in_run;
        s->out_base = out_run;
        s->first = (int32_t)b->blocks.size();
        for (const cramgpu_block& blk : s->blocks) {
            cramgpu_block r = blk;
            r.in_off += in_run;
            r.out_off += out_run;
            b->blocks.push_back(r);
        }
        max_cap = std::max(max_cap, s->max_cap);
        in_run += align_up(s->in_bytes, 16);
        out_run += align_up(s->out_bytes, 16);
    }
    const int32_t n = (int32_t)b->blocks.size();
    hipStream_t s = b->stream;
    HIPCHK(b->d_in.reserve(co_round((size_t)in_run + 16)));
    HIPCHK(b->d_out.reserve(co_round((size_t)out_run + 16)));
    HIPCHK(b->d_len.reserve(co_round(n) / 4));
    HIPCHK(b->d_pos.reserve(co_round(n) / 4));
    HIPCHK(b->h_len.reserve(co_round(n) / 4));
    HIPCHK(b->h_pos.reserve(co_round((size_t)n + 1) / 4));
    // uploads first come first served on one stream (see cramgpu_ctx::upload_stream)
    for (CoSub* u : b->subs)
        if (u->in_bytes > 0)
            HIPCHK(hipMemcpyAsync(b->d_in.p + u->in_base, u->in, (size_t)u->in_bytes, hipMemcpyHostToDevice, q->upload_stream));
    HIPCHK(hipEventRecord(b->uploaded, q->upload_stream));
    HIPCHK(hipStreamWaitEvent(s, b->uploaded, 0));
    if (int rc = rans_run(c, b->res, !b->decode, b->d_in.p, b->d_out.p, b->blocks.data(), n, b->d_len.p,
                          b->decode ? nullptr : b->d_pos.p, s))
        return rc;
    HIPCHK(hipMemcpyAsync(b->h_len.p, b->d_len.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (!b->decode) {
        HIPCHK(b->d_pack.reserve(co_round((size_t)out_run + 16)));
        HIPCHK(b->d_off.reserve(co_round((size_t)n + 1) / 4));
        HIPCHK(b->d_err.reserve(16));
        HIPCHK(launch_gather(b->d_out.p, b->d_pos.p, b->d_len.p, n, 16, b->d_pack.p, out_run, b->d_off.p, b->d_err.p,
                             (int32_t)std::min<int64_t>(max_cap, 0x7fffffff), s));
        HIPCHK(hipMemcpyAsync(b->h_pos.p, b->d_off.p, ((size_t)n + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    }
    HIPCHK(hipEventRecord(b->coded, s));
    return 0;
}

// phase 1 -> 2: lengths and positions are on the host, queue the copies of every submission's results into its own arena
int co_fetch(cramgpu_coalescer* q, CoBatch* b) {
    if (q->trace) b->t_coded = q->now();
    for (CoSub* u : b->subs) {
        const int32_t nb = (int32_t)u->blocks.size();
        if (b->decode) {
            if (u->hi > u->lo)
                HIPCHK(hipMemcpyAsync(u->out + u->lo, b->d_out.p + u->out_base + u->lo, (size_t)(u->hi - u->lo),
                                      hipMemcpyDeviceToHost, b->stream));
            continue;
        }
        // the streams of one submission are neighbours in the packed arena
        const int64_t p0 = b->h_pos.p[u->first], p1 = b->h_pos.p[u->first + nb];
        if (p1 - p0 > u->out_bytes) {
            u->rc = CRAMGPU_E_INSUFFICIENT;
            u->err = "coalesced job: packed streams exceed the output arena";
            continue;
        }
        if (p1 > p0) HIPCHK(hipMemcpyAsync(u->out, b->d_pack.p + p0, (size_t)(p1 - p0), hipMemcpyDeviceToHost, b->stream));
    }
    HIPCHK(hipEventRecord(b->done, b->stream));
    return 0;
}

// phase 2 -> home: the copies have landed
void co_deliver(CoBatch* b) {
    for (CoSub* u : b->subs) {
        const int32_t nb = (int32_t)u->blocks.size();
        for (int32_t i = 0; i < nb; i++) u->out_len[i] = b->h_len.p[u->first + i];
        if (b->decode || u->rc) continue;
        const int64_t p0 = b->h_pos.p[u->first];
        if (u->packed) {
            for (int32_t i = 0; i <= nb; i++) u->out_pos[i] = b->h_pos.p[u->first + i] - p0;
        } else {  // spread to the slots, last stream first (cramgpu_job_wait)
            for (int32_t i = nb - 1; i >= 0; i--) {
                const int32_t len = u->out_len[i];
                const int64_t at = b->h_pos.p[u->first + i] - p0;
                if (len > 0 && u->blocks[i].out_off != at) memmove(u->out + u->blocks[i].out_off, u->out + at, (size_t)len);
            }
        }
    }
}

void co_finish(cramgpu_coalescer* q, CoBatch* b, int rc) {  // under q->mu
    for (CoSub* u : b->subs) {
        if (rc && !u->rc) {
            u->rc = rc;
            u->err = g_err;
        }
        u->state = 2;
    }
    b->subs.clear();
    q->free_batches.push_back(b);
}

void co_dispatch_loop(cramgpu_coalescer* q) {
    tl_graveyard = &q->graveyard;
    std::unique_lock<std::mutex> lk(q->mu);
    for (;;) {
        q->cv_work.wait(lk, [&] { return q->stop || (!q->queue.empty() && !q->free_batches.empty()); });
        if (q->stop) return;
        if (q->hold_us > 0 && !q->inflight.empty() && (int32_t)q->queue.size() < q->max_subs) {
            const auto until = q->queue.front()->arrived + std::chrono::microseconds(q->hold_us);
            if (std::chrono::steady_clock::now() < until) {
                q->cv_work.wait_until(lk, until, [&] { return q->stop || (int32_t)q->queue.size() >= q->max_subs; });
                continue;  // look again: the queue has grown, or the hold has run out
            }
        }
        CoBatch* b = q->free_batches.back();
        q->free_batches.pop_back();
        b->decode = q->queue.front()->decode;
        b->subs.clear();
        int64_t in_run = 0;
        while (!q->queue.empty() && (int32_t)b->subs.size() < q->max_subs) {
            CoSub* u = q->queue.front();
            if (u->decode != b->decode) break;  // one direction per launch; the other one goes into the next batch
            if (!b->subs.empty() && in_run + u->in_bytes > q->max_in_bytes) break;
            in_run += align_up(u->in_bytes, 16);
            u->state = 1;
            b->subs.push_back(u);
            q->queue.pop_front();
        }
        q->n_batches++;
        q->n_subs += (int64_t)b->subs.size();
        q->largest = std::max<int64_t>(q->largest, (int64_t)b->subs.size());
        lk.unlock();
        if (q->trace) b->t_pick = q->now();
        const int rc = co_launch(q, b);
        if (q->trace) b->t_issued = q->now();
        lk.lock();
        if (rc) {
            co_finish(q, b, rc);
            q->cv_done.notify_all();
        } else {
            b->phase = 1;
            q->inflight.push_back(b);
            q->cv_flight.notify_one();
        }
    }
}

// Brings batches home in the order they FINISH (a short batch launched behind a long one is not kept waiting): polls
// the events of everything in flight, sleeping a few tens of microseconds between rounds.
void co_complete_loop(cramgpu_coalescer* q) {
    (void)hipSetDevice(q->ctx->device);
    std::unique_lock<std::mutex> lk(q->mu);
    std::vector<CoBatch*> fl;
    for (;;) {
        q->cv_flight.wait(lk, [&] { return !q->inflight.empty() || q->stop; });
        if (q->inflight.empty()) return;  // stop: batches in flight are still brought home first
        fl.assign(q->inflight.begin(), q->inflight.end());
        lk.unlock();
        bool progress = false;
        for (CoBatch* b : fl) {
            int rc = 0;
            bool home = false;
            hipError_t e = hipEventQuery(b->phase == 1 ? b->coded : b->done);
            if (e == hipErrorNotReady) continue;
            progress = true;
            if (e != hipSuccess) {
                rc = fail(CRAMGPU_E_HIP, "coalesced job", e);
                home = true;
            } else if (b->phase == 1) {
                rc = co_fetch(q, b);
                b->phase = 2;
                home = rc != 0;
            } else {
                co_deliver(b);
                home = true;
            }
            if (!home) continue;
            if (q->trace)
                fprintf(stderr, "[cramgpu coalesce] batch of %3zu (%s): picked %9.3f, issued +%.3f, coded +%.3f, home +%.3f ms\n", b->subs.size(),
                        b->decode ? "decode" : "encode", b->t_pick, b->t_issued - b->t_pick, b->t_coded - b->t_pick, q->now() - b->t_pick);
            lk.lock();
            q->inflight.erase(std::find(q->inflight.begin(), q->inflight.end(), b));
            co_finish(q, b, rc);
            lk.unlock();
            q->cv_done.notify_all();
            q->cv_work.notify_one();
        }
        if (!progress) std::this_thread::sleep_for(std::chrono::microseconds(20));
        lk.lock();
    }
}

}  // namespace

extern "C" {

cramgpu_coalescer* cramgpu_coalescer_open(cramgpu_ctx* c, int32_t max_batches) {
    if (!c) {
        fail(CRAMGPU_E_ARG, "coalescer: null context");
        return nullptr;
    }
    if (max_batches <= 0) max_batches = env_int("CRAMGPU_COALESCE_BATCHES", 6);
    max_batches = std::min(max_batches, 16);
    if (hipSetDevice(c->device) != hipSuccess) {
        fail(CRAMGPU_E_HIP, "hipSetDevice failed");
        return nullptr;
    }
    cramgpu_coalescer* q = new cramgpu_coalescer();
    q->ctx = c;
    q->batches.resize((size_t)max_batches);
    for (auto& b : q->batches) q->free_batches.push_back(&b);
    q->max_in_bytes = (int64_t)env_int("CRAMGPU_COALESCE_MAX_MB", 1024) << 20;
    q->max_subs = env_int("CRAMGPU_COALESCE_MAX_SUBS", 1024);
    q->hold_us = env_int("CRAMGPU_COALESCE_HOLD_US", 300);
    q->trace = env_int("CRAMGPU_COALESCE_TRACE", 0) != 0;
    q->t0 = std::chrono::steady_clock::now();
    if (hipStreamCreateWithFlags(&q->upload_stream, hipStreamNonBlocking) != hipSuccess) {
        fail(CRAMGPU_E_HIP, "coalescer: stream creation failed");
        delete q;
        return nullptr;
    }
    q->dispatcher = std::thread(co_dispatch_loop, q);
    q->completer = std::thread(co_complete_loop, q);
    return q;
}

void cramgpu_coalescer_close(cramgpu_coalescer* q) {
    if (!q) return;
    {
        std::unique_lock<std::mutex> lk(q->mu);
        q->stop = true;
        // submissions nobody launched yet are refused, batches in flight come home
        for (CoSub* u : q->queue) {
            u->rc = CRAMGPU_E_ARG;
            u->err = "coalescer closed";
            u->state = 2;
        }
        q->queue.clear();
    }
    q->cv_work.notify_all();
    q->cv_flight.notify_all();
    q->cv_done.notify_all();
    if (q->dispatcher.joinable()) q->dispatcher.join();
    q->cv_flight.notify_all();
    if (q->completer.joinable()) q->completer.join();
    (void)hipSetDevice(q->ctx->device);
    for (auto& b : q->batches) {
        if (b.stream) (void)hipStreamSynchronize(b.stream);
        b.res.release();
        b.d_in.release(); b.d_out.release(); b.d_pack.release(); b.d_len.release(); b.d_err.release();
        b.d_pos.release(); b.d_off.release(); b.h_len.release(); b.h_pos.release();
        if (b.stream) (void)hipStreamDestroy(b.stream);
        if (b.uploaded) (void)hipEventDestroy(b.uploaded);
        if (b.coded) (void)hipEventDestroy(b.coded);
        if (b.done) (void)hipEventDestroy(b.done);
    }
    if (q->upload_stream) (void)hipStreamDestroy(q->upload_stream);
    for (void* p : q->graveyard.dev) (void)hipFree(p);
    for (void* p : q->graveyard.pin) (void)hipHostFree(p);
    for (auto& kv : q->tickets) delete kv.second;
    delete q;
}

int64_t cramgpu_rans_enqueue(cramgpu_coalescer* q, int32_t decode, const uint8_t* in_arena, int64_t in_bytes, uint8_t* out_arena,
                             int64_t out_bytes, const cramgpu_block* blocks, int32_t n, int32_t* out_len, int64_t* out_pos) {
    if (!q || n <= 0 || in_bytes < 0 || out_bytes < 0 || !in_arena || !out_arena || !blocks || !out_len)
        return fail(CRAMGPU_E_ARG, "enqueue: bad argument");
    int64_t max_cap = 0, lo = out_bytes, hi = 0, prev_end = 0;
    for (int32_t i = 0; i < n; i++) {
        const cramgpu_block& b = blocks[i];
        if (b.in_off < 0 || b.out_off < 0 || (b.in_off % CRAMGPU_ARENA_ALIGN) || (!decode && (b.out_off % CRAMGPU_ARENA_ALIGN)))
            return fail(CRAMGPU_E_ARG, "block offsets must be non-negative multiples of CRAMGPU_ARENA_ALIGN");
        if (b.in_len < 0 || b.out_cap < 0) return fail(CRAMGPU_E_ARG, "negative block size");
        if (b.in_off + b.in_len > in_bytes || b.out_off + b.out_cap > out_bytes)
            return fail(CRAMGPU_E_ARG, "enqueue: block outside its arena");
        if (!decode && !out_pos && b.out_off < prev_end)
            return fail(CRAMGPU_E_ARG, "enqueue: output slots must be ascending and must not overlap");
        prev_end = b.out_off + b.out_cap;
        max_cap = std::max<int64_t>(max_cap, b.out_cap);
        lo = std::min<int64_t>(lo, b.out_off);
        hi = std::max<int64_t>(hi, b.out_off + b.out_cap);
    }
    CoSub* u = new CoSub();
    u->decode = decode != 0;
    u->packed = !decode && out_pos != nullptr;
    u->in = in_arena;
    u->out = out_arena;
    u->in_bytes = in_bytes;
    u->out_bytes = out_bytes;
    u->blocks.assign(blocks, blocks + n);
    u->out_len = out_len;
    u->out_pos = out_pos;
    u->lo = lo;
    u->hi = hi;
    u->max_cap = max_cap;
    int64_t id;
    {
        std::lock_guard<std::mutex> lk(q->mu);
        if (q->stop) {
            delete u;
            return fail(CRAMGPU_E_ARG, "enqueue: coalescer closed");
        }
        id = u->id = q->next_id++;
        u->arrived = std::chrono::steady_clock::now();
        q->tickets[id] = u;
        q->queue.push_back(u);
    }
    q->cv_work.notify_one();
    return id;
}

int32_t cramgpu_ticket_poll(cramgpu_coalescer* q, int64_t ticket) {
    if (!q) return fail(CRAMGPU_E_ARG, "poll: null coalescer");
    std::lock_guard<std::mutex> lk(q->mu);
    auto it = q->tickets.find(ticket);
    if (it == q->tickets.end()) return fail(CRAMGPU_E_ARG, "poll: no such ticket");
    return it->second->state == 2 ? 1 : 0;
}

int32_t cramgpu_ticket_wait(cramgpu_coalescer* q, int64_t ticket) {
    if (!q) return fail(CRAMGPU_E_ARG, "wait: null coalescer");
    std::unique_lock<std::mutex> lk(q->mu);
    auto it = q->tickets.find(ticket);
    if (it == q->tickets.end()) return fail(CRAMGPU_E_ARG, "wait: no such ticket");
    CoSub* u = it->second;
    q->cv_done.wait(lk, [&] { return u->state == 2; });
    q->tickets.erase(it);
    lk.unlock();
    const int32_t rc = u->rc;
    if (rc) g_err = u->err;
    delete u;
    return rc;
}

int32_t cramgpu_rans_coalesced(cramgpu_coalescer* q, int32_t decode, const uint8_t* in_arena, int64_t in_bytes, uint8_t* out_arena,
                               int64_t out_bytes, const cramgpu_block* blocks, int32_t n, int32_t* out_len, int64_t* out_pos) {
    const int64_t t = cramgpu_rans_enqueue(q, decode, in_arena, in_bytes, out_arena, out_bytes, blocks, n, out_len, out_pos);
    if (t < 0) return (int32_t)t;
    return cramgpu_ticket_wait(q, t);
}

int32_t cramgpu_coalesce_stats(cramgpu_coalescer* q, int64_t* n_batches, int64_t* n_submissions, int64_t* largest_batch) {
    if (!q) return fail(CRAMGPU_E_ARG, "stats: null coalescer");
    std::lock_guard<std::mutex> lk(q->mu);
    if (n_batches) *n_batches = q->n_batches;
    if (n_submissions) *n_submissions = q->n_subs;
    if (largest_batch) *largest_batch = q->largest;
    return 0;
}

// ---- device pool: container k -> GPU k mod G, results in ordinal order (OrderingConveyer.java:24-38) ----
}  // extern "C"

struct cramgpu_pool {
    std::vector<cramgpu_ctx*> ctx;
    std::vector<cramgpu_coalescer*> queue;
    std::mutex mu;
    std::condition_variable cv;
    std::unordered_map<int64_t, int64_t> ticket;  // ordinal -> ticket of queue[ordinal mod G]
    int64_t next = 0;
    bool closing = false;
};

extern "C" {

cramgpu_pool* cramgpu_pool_open(int32_t n_devices, int32_t max_batches) {
    const int32_t have = cramgpu_device_count();
    if (n_devices <= 0) n_devices = have;
    if (have <= 0 || n_devices > have) {
        fail(CRAMGPU_E_NO_DEVICE, "cramgpu_pool_open: fewer HIP devices than asked for (libcramgpu has no CPU path)");
        return nullptr;
    }
    cramgpu_pool* p = new cramgpu_pool();
    for (int32_t d = 0; d < n_devices; d++) {
        cramgpu_ctx* c = cramgpu_open(d);
        cramgpu_coalescer* q = c ? cramgpu_coalescer_open(c, max_batches) : nullptr;
        if (!q) {
            if (c) cramgpu_close(c);
            cramgpu_pool_close(p);
            return nullptr;
        }
        p->ctx.push_back(c);
        p->queue.push_back(q);
    }
    return p;
}

void cramgpu_pool_close(cramgpu_pool* p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->closing = true;
    }
    p->cv.notify_all();
    for (auto* q : p->queue) cramgpu_coalescer_close(q);
    for (auto* c : p->ctx) cramgpu_close(c);
    delete p;
}

int32_t cramgpu_pool_devices(cramgpu_pool* p) { return p ? (int32_t)p->ctx.size() : 0; }

int32_t cramgpu_pool_submit(cramgpu_pool* p, int64_t ordinal, int32_t decode, const uint8_t* in_arena, int64_t in_bytes,
                            uint8_t* out_arena, int64_t out_bytes, const cramgpu_block* blocks, int32_t n, int32_t* out_len,
                            int64_t* out_pos) {
    if (!p || ordinal < 0) return fail(CRAMGPU_E_ARG, "pool_submit: bad argument");
    const size_t g = (size_t)(ordinal % (int64_t)p->queue.size());
    {
        std::lock_guard<std::mutex> lk(p->mu);
        if (ordinal < p->next || p->ticket.count(ordinal)) return fail(CRAMGPU_E_ARG, "pool_submit: ordinal used twice");
    }
    const int64_t t = cramgpu_rans_enqueue(p->queue[g], decode, in_arena, in_bytes, out_arena, out_bytes, blocks, n, out_len, out_pos);
    if (t < 0) return (int32_t)t;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->ticket[ordinal] = t;
    }
    p->cv.notify_all();
    return 0;
}

int32_t cramgpu_pool_next(cramgpu_pool* p, int64_t* ordinal) {
    if (!p || !ordinal) return fail(CRAMGPU_E_ARG, "pool_next: bad argument");
    int64_t k, t;
    {
        std::unique_lock<std::mutex> lk(p->mu);
        k = p->next;
        p->cv.wait(lk, [&] { return p->closing || p->ticket.count(k) != 0; });
        if (!p->ticket.count(k)) return fail(CRAMGPU_E_ARG, "pool_next: pool closed");
        t = p->ticket[k];
        p->ticket.erase(k);
        p->next = k + 1;
    }
    *ordinal = k;
    return cramgpu_ticket_wait(p->queue[(size_t)(k % (int64_t)p->queue.size())], t);
}

// ---- level 0 ----
int32_t cramgpu_rans_compress(const uint8_t* in, int32_t n, int32_t order, uint8_t* out, int32_t cap) {
    if (order != 0 && order != 1) return fail(CRAMGPU_E_ORDER, "Unknown rANS order");
    if (n == 0) return 0;
    if (!in || !out || n < 0 || cap < 0) return fail(CRAMGPU_E_ARG, "cramgpu_rans_compress: bad argument");
    cramgpu_ctx* c = default_ctx();
    if (!c) return CRAMGPU_E_NO_DEVICE;
    // RANS#allocateIfNeeded: the reference codes into its own bound-sized buffer and hands back an exact-size array
    const int32_t bound = cramgpu_rans_bound(n);
    cramgpu_block b{0, 0, n, bound, order, 0};
    std::vector<uint8_t> tmp((size_t)bound);
    int32_t len = 0;
    int rc = host_batch(c, true, in, n, tmp.data(), bound, &b, 1, &len);
    if (rc) return rc;
    if (len < 0) return fail(len, "rans compress failed");
    if (len > cap) return fail(CRAMGPU_E_INSUFFICIENT, "Insufficient buffer size.");
    memcpy(out, tmp.data(), (size_t)len);
    return len;
}

int32_t cramgpu_rans_uncompress(const uint8_t* in, int32_t n, uint8_t* out, int32_t cap) {
    if (n == 0) return 0;
    if (!in || n < 0 || cap < 0) return fail(CRAMGPU_E_ARG, "cramgpu_rans_uncompress: bad argument");
    if (n < 9) return fail(CRAMGPU_E_INPUT_LENGTH, "Incorrect input length.");
    cramgpu_ctx* c = default_ctx();
    if (!c) return CRAMGPU_E_NO_DEVICE;
    cramgpu_block b{0, 0, n, cap, 0, 0};
    int32_t len = 0;
    std::vector<uint8_t> dummy(1);
    int rc = host_batch(c, false, in, n, cap > 0 && out ? out : dummy.data(), cap, &b, 1, &len);
    if (rc) return rc;
    if (len < 0) return fail(len, "rans uncompress failed");
    return len;
}

// =====================================================================================================
// Level 2 — record kernels
// =====================================================================================================

// SubstitutionMatrix#<init>([[J) L38-57: rank the four other bases by descending count, ties in ACGTN order
void cramgpu_substitution_matrix(const int64_t* hist, uint8_t* matrix5) {
    for (int r = 0; r < 5; r++) {
        int others[4], k = 0;
        for (int b = 0; b < 5; b++)
            if (b != r) others[k++] = b;
        int order[4] = {0, 1, 2, 3};
        std::stable_sort(order, order + 4, [&](int x, int y) { return hist[r * 5 + others[x]] > hist[r * 5 + others[y]]; });
        int rank[4];
        for (int i = 0; i < 4; i++) rank[order[i]] = i;
        matrix5[r] = (uint8_t)((rank[0] << 6) | (rank[1] << 4) | (rank[2] << 2) | rank[3]);
    }
}

// SubstitutionMatrix#code / #base as tables: codes[5][256] (0 for bases outside ACGTN) and basesof[5][4]
static void matrix_tables(const uint8_t* m5, uint8_t* codes, uint8_t* basesof) {
    static const uint8_t ACGTN[5] = {'A', 'C', 'G', 'T', 'N'};
    memset(codes, 0, 5 * 256);
    for (int r = 0; r < 5; r++) {
        int k = 0;
        for (int b = 0; b < 5; b++) {
            if (b == r) continue;
            const int code = (m5[r] >> (6 - 2 * k)) & 3;
            codes[r * 256 + ACGTN[b]] = (uint8_t)code;
            basesof[r * 4 + code] = ACGTN[b];
            k++;
        }
    }
}

int64_t cramgpu_extract_features_dev(cramgpu_ctx* c, const cramgpu_reads* reads, const uint8_t* d_ref,
                                     cramgpu_features* feats, int64_t* d_sub_hist, void* stream) {
    if (!c || !reads || !feats || reads->n_reads < 0 || !feats->off) return fail(CRAMGPU_E_ARG, "extract_features: bad argument");
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t s = pick(c, stream);
    if (int rc = enter(c, s)) return rc;
    CallScope scope{c, s};
    const int32_t n = reads->n_reads;
    HIPCHK(c->d_counts.reserve((size_t)n + 1));
    HIPCHK(c->d_list.reserve((size_t)n + 1));  // reads that have features + (last word) their number
    int32_t* list = c->d_list.p;
    uint32_t* list_n = reinterpret_cast<uint32_t*>(c->d_list.p + n);
    if (n > 0) HIPCHK(launch_extract_features(*reads, d_ref, *feats, nullptr, c->d_counts.p, false, list, list_n, nullptr, s));
    HIPCHK(c->d_scan_ws.reserve(feature_scan_ws_elems(n)));
    HIPCHK(launch_feature_scan(n, c->d_counts.p, feats->off, c->d_total.p, c->d_scan_ws.p, s));
    HIPCHK(hipMemcpyAsync(c->h_total.p, c->d_total.p, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const int64_t total = c->h_total.p[0];
    if (total > feats->cap) return fail(CRAMGPU_E_INSUFFICIENT, "extract_features: feature capacity too small");
    if (n > 0 && total > 0)
        HIPCHK(launch_extract_features(*reads, d_ref, *feats, d_sub_hist, c->d_counts.p, true, list, list_n, nullptr, s));
    return total;
}

int32_t cramgpu_write_series_dev(cramgpu_ctx* c, const cramgpu_reads* reads, const cramgpu_record_cols* cols,
                                 const cramgpu_features* feats, uint8_t* d_arena, int64_t arena_cap, int64_t* d_series_off,
                                 int32_t* d_series_len, int32_t* d_status, void* stream) {
    if (!c || !reads || !cols || !feats || !d_arena || !d_series_off || !d_series_len || !d_status || reads->n_reads < 0 ||
        reads->n_cont < 1 || arena_cap < 16)
        return fail(CRAMGPU_E_ARG, "write_series: bad argument");
    if (!cols->bam_flags || !cols->comp_flags || !cols->read_group || !cols->mate_flags || !cols->mate_ref || !cols->mate_start ||
        !cols->tlen || !cols->next_frag || !cols->tag_list || (cols->cont_multiref && !cols->ref_id) || !reads->mapq ||
        !reads->quals || !feats->off || (reads->n_cont > 1 && !reads->read_cont))
        return fail(CRAMGPU_E_ARG, "write_series: a column is missing");
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t s = pick(c, stream);
    if (int rc = enter(c, s)) return rc;
    CallScope scope{c, s};
    HIPCHK(c->d_series_alias.reserve((size_t)reads->n_cont));
    HIPCHK(launch_write_series(*reads, *cols, *feats, d_arena, arena_cap, d_series_off, d_series_len, d_status,
                               c->d_series_alias.p, s));
    return 0;
}

int32_t cramgpu_read_series_dev(cramgpu_ctx* c, int32_t n_reads, int32_t n_cont, const int32_t* d_cont_first,
                                const uint8_t* d_arena, const int64_t* d_series_off, const int32_t* d_series_len,
                                int64_t series_bytes, const cramgpu_record_cols* cols, const cramgpu_read_out* out,
                                const cramgpu_features* feats, int32_t* d_status, void* stream) {
    if (!c || n_reads < 0 || n_cont < 1 || !d_cont_first || !d_arena || !d_series_off || !d_series_len || series_bytes < 0 ||
        !cols || !out || !feats || !d_status)
        return fail(CRAMGPU_E_ARG, "read_series: bad argument");
    if (!cols->bam_flags || !cols->comp_flags || !cols->read_group || !cols->mate_flags || !cols->mate_ref || !cols->mate_start ||
        !cols->tlen || !cols->next_frag || !cols->tag_list || (cols->cont_multiref && !cols->ref_id) || !out->read_len ||
        !out->aln_start || !out->mapq || !out->base_off || !out->bases || !out->quals || !out->seq || !out->seq_off ||
        !feats->op || !feats->pos || !feats->a || !feats->b || !feats->off)
        return fail(CRAMGPU_E_ARG, "read_series: an output column is missing");
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t s = pick(c, stream);
    if (int rc = enter(c, s)) return rc;
    CallScope scope{c, s};
    const size_t ws_ints = read_series_ws_ints(series_bytes, n_cont);
    HIPCHK(c->d_series_vals.reserve(ws_ints));
    HIPCHK(c->d_series_cnt.reserve((size_t)n_cont * CRAMGPU_N_SERIES));
    HIPCHK(c->d_series_off.reserve((size_t)n_cont * CRAMGPU_N_SERIES));
    HIPCHK(c->d_series_base.reserve((size_t)(n_cont + 1) * 3));
    HIPCHK(launch_read_series(d_arena, d_series_off, d_series_len, n_cont, n_reads, d_cont_first, *cols, *out, *feats,
                              c->d_series_vals.p, (int64_t)ws_ints, c->d_series_off.p, c->d_series_cnt.p, c->d_series_base.p, d_status, s));
    return 0;
}

int32_t cramgpu_cigars_dev(cramgpu_ctx* c, int32_t n_reads, const int32_t* d_read_len, const int32_t* d_aln_start,
                           const int32_t* d_bam_flags, const cramgpu_features* feats, int32_t* d_n_cigar, int64_t* d_cigar_off,
                           uint32_t* d_cigar, int64_t cigar_cap, int32_t* d_aln_end, int32_t* d_status, void* stream) {
    if (!c || n_reads < 0 || !d_read_len || !d_aln_start || !d_bam_flags || !feats || !feats->off || !d_n_cigar || !d_cigar_off ||
        !d_cigar || cigar_cap < 0 || !d_aln_end || !d_status)
        return fail(CRAMGPU_E_ARG, "cigars: bad argument");
    if (n_reads > 0 && feats->cap > 0 && (!feats->op || !feats->pos || !feats->a || !feats->b))
        return fail(CRAMGPU_E_ARG, "cigars: a feature column is missing");
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t s = pick(c, stream);
    if (int rc = enter(c, s)) return rc;
    CallScope scope{c, s};
    HIPCHK(c->d_scan_ws.reserve(feature_scan_ws_elems(n_reads)));
    HIPCHK(launch_cigars(n_reads, d_read_len, d_aln_start, d_bam_flags, *feats, d_n_cigar, d_cigar_off, d_cigar, cigar_cap,
                         d_aln_end, d_status, c->d_total.p, c->d_scan_ws.p, s));
    return 0;
}

int32_t cramgpu_restore_mates_dev(cramgpu_ctx* c, int32_t n_reads, int32_t n_cont, const int32_t* d_cont_first,
                                  const cramgpu_record_cols* cols, const int32_t* d_cont_ref_id, const int32_t* d_aln_start,
                                  const int32_t* d_aln_end, int32_t* d_status, void* stream) {
    if (!c || n_reads < 0 || n_cont < 1 || !d_cont_first || !cols || !d_cont_ref_id || !d_aln_start || !d_aln_end || !d_status)
        return fail(CRAMGPU_E_ARG, "restore_mates: bad argument");
    if (!cols->bam_flags || !cols->comp_flags || !cols->ref_id || !cols->mate_flags || !cols->mate_ref || !cols->mate_start ||
        !cols->tlen || !cols->next_frag)
        return fail(CRAMGPU_E_ARG, "restore_mates: a column is missing");
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t s = pick(c, stream);
    if (int rc = enter(c, s)) return rc;
    CallScope scope{c, s};
    HIPCHK(c->d_list.reserve((size_t)n_reads + 1));
    HIPCHK(c->d_mate_prev.reserve((size_t)n_reads + 1));
    HIPCHK(c->d_mate_serial.reserve((size_t)n_cont));
    HIPCHK(launch_restore_mates(n_reads, n_cont, d_cont_first, *cols, d_cont_ref_id, d_aln_start, d_aln_end, c->d_list.p,
                                c->d_mate_prev.p, c->d_mate_serial.p, d_status, s));
    return 0;
}

int32_t cramgpu_set_mate_info_dev(cramgpu_ctx* c, int32_t n_reads, int32_t n_cont, const int32_t* d_cont_first,
                                  const cramgpu_record_cols* cols, const int32_t* d_aln_start, const int32_t* d_aln_end,
                                  const uint8_t* d_names, const int64_t* d_name_off, const int32_t* d_name_len, void* stream) {
    if (!c || n_reads < 0 || n_cont < 1 || !d_cont_first || !cols || !d_aln_start || !d_aln_end || !d_names || !d_name_off ||
        !d_name_len)
        return fail(CRAMGPU_E_ARG, "set_mate_info: bad argument");
    if (!cols->bam_flags || !cols->comp_flags || !cols->ref_id || !cols->mate_flags || !cols->mate_ref || !cols->mate_start ||
        !cols->tlen || !cols->next_frag)
        return fail(CRAMGPU_E_ARG, "set_mate_info: a column is missing");
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t s = pick(c, stream);
    if (int rc = enter(c, s)) return rc;
    CallScope scope{c, s};
    HIPCHK(c->d_list.reserve(3 * mate_table_slots(n_reads) + (size_t)n_reads + 1));
    HIPCHK(launch_set_mate_info(n_reads, n_cont, d_cont_first, *cols, d_aln_start, d_aln_end, d_names, d_name_off, d_name_len,
                                c->d_list.p, s));
    return 0;
}

int32_t cramgpu_content_digests_dev(cramgpu_ctx* c, int32_t n_reads, int32_t n_cont, const int32_t* d_cont_first,
                                    const int64_t* d_base_off, const int32_t* d_read_len, const uint8_t* d_bases,
                                    const uint8_t* d_quals, uint32_t* d_crc, void* stream) {
    if (!c || n_reads < 0 || n_cont < 1 || !d_cont_first || !d_base_off || !d_read_len || !d_bases || !d_quals || !d_crc)
        return fail(CRAMGPU_E_ARG, "content_digests: bad argument");
    // no scratch of the context is touched: not ordered against the context's other calls, so that a caller may run it on
    // a stream of its own beside them (bench.py encode_full: beside the rANS encoder)
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = pick(c, stream);
    HIPCHK(launch_content_crc(n_reads, n_cont, d_cont_first, d_base_off, d_read_len, d_bases, d_quals, d_crc, s));
    return 0;
}

int32_t cramgpu_bam_encode_dev(cramgpu_ctx* c, int32_t n_reads, const cramgpu_bam_cols* b, uint8_t* d_bam, int64_t bam_cap,
                               int64_t* d_rec_off, int32_t* d_status, void* stream) {
    if (!c || n_reads < 0 || !b || !d_bam || bam_cap < 0 || !d_rec_off || !d_status)
        return fail(CRAMGPU_E_ARG, "bam_encode: bad argument");
    if (!b->ref_id || !b->aln_start || !b->aln_end || !b->mapq || !b->bam_flags || !b->mate_flags || !b->mate_ref ||
        !b->mate_start || !b->tlen || !b->read_len || !b->base_off || !b->bases || !b->quals || !b->n_cigar || !b->cigar_off ||
        !b->cigar || (b->names && !b->name_off) || (b->tags && !b->tag_off) || (!b->names && b->name_base < 0))
        return fail(CRAMGPU_E_ARG, "bam_encode: a column is missing");
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t s = pick(c, stream);
    if (int rc = enter(c, s)) return rc;
    CallScope scope{c, s};
    HIPCHK(c->d_counts.reserve((size_t)n_reads + 1));
    HIPCHK(c->d_scan_ws.reserve(feature_scan_ws_elems(n_reads)));
    HIPCHK(launch_bam_encode(n_reads, *b, d_bam, bam_cap, d_rec_off, d_status, c->d_counts.p, c->d_total.p, c->d_scan_ws.p, s));
    return 0;
}

int32_t cramgpu_bam_decode_dev(cramgpu_ctx* c, int32_t n_reads, const uint8_t* d_bam, const int64_t* d_rec_off,
                               const cramgpu_bam_out* o, int32_t* d_status, void* stream) {
    if (!c || n_reads < 0 || !d_bam || !d_rec_off || !o || !d_status) return fail(CRAMGPU_E_ARG, "bam_decode: bad argument");
    if (!o->ref_id || !o->aln_start || !o->mapq || !o->bam_flags || !o->mate_ref || !o->mate_start || !o->tlen || !o->read_len ||
        !o->base_off || !o->bases || !o->quals || o->bases_cap < 0 || !o->n_cigar || !o->cigar_off || !o->cigar ||
        o->cigar_cap < 0 || !o->aln_span || !o->rflags || !o->name_off || !o->name_len || !o->tag_off || !o->tag_len)
        return fail(CRAMGPU_E_ARG, "bam_decode: an output column is missing");
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t s = pick(c, stream);
    if (int rc = enter(c, s)) return rc;
    CallScope scope{c, s};
    HIPCHK(c->d_scan_ws.reserve(feature_scan_ws_elems(n_reads)));
    HIPCHK(launch_bam_decode(n_reads, d_bam, d_rec_off, *o, d_status, c->d_total.p, c->d_scan_ws.p, s));
    return 0;
}

int32_t cramgpu_slice_md5_dev(cramgpu_ctx* c, int32_t n_slices, const uint8_t* d_ref, const cramgpu_refwin* d_wins,
                              const int32_t* d_start, const int32_t* d_span, uint8_t* d_md5, void* stream) {
    if (!c || n_slices < 0 || !d_ref || !d_wins || !d_start || !d_span || !d_md5) return fail(CRAMGPU_E_ARG, "slice_md5: bad argument");
    // like the content digests: no scratch, not ordered against the context's other calls
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = pick(c, stream);
    HIPCHK(launch_slice_md5(n_slices, d_ref, d_wins, d_start, d_span, d_md5, s));
    return 0;
}

int32_t cramgpu_gunzip_dev(cramgpu_ctx* c, const uint8_t* d_in, uint8_t* d_out, const cramgpu_block* d_blocks, int32_t n,
                           int32_t* d_out_len, void* stream) {
    if (!c || n < 0 || (n > 0 && (!d_in || !d_out || !d_blocks || !d_out_len))) return fail(CRAMGPU_E_ARG, "gunzip: bad argument");
    // no scratch of the context is touched: the launch is not ordered against the context's other calls, so a caller may
    // run it on a stream of its own beside them (e.g. beside the rANS decode of the same container batch)
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(launch_gunzip(d_in, d_out, d_blocks, n, d_out_len, pick(c, stream)));
    return 0;
}

int32_t cramgpu_gzip_dev(cramgpu_ctx* c, const uint8_t* d_in, uint8_t* d_out, const cramgpu_block* d_blocks, int32_t n,
                         int32_t* d_out_len, int32_t bgzf, void* stream) {
    if (!c || n < 0 || (n > 0 && (!d_in || !d_out || !d_blocks || !d_out_len))) return fail(CRAMGPU_E_ARG, "gzip: bad argument");
    // like gunzip, not ordered against the context's other calls (it runs beside the rANS encoder of the same batch on
    // a stream of its own); its token workspace is ordered among the gzip calls themselves
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    if (n == 0) return 0;
    hipStream_t s = pick(c, stream);
    const int waves = std::min(n, c->n_cu * std::max(1, std::min(env_int("CRAMGPU_GZIP_WAVES_PER_CU", 20), 32)));
    if (gzip_ws_bytes(waves) > c->d_gz_ws.cap) {
        if (c->gz_valid) HIPCHK(hipEventSynchronize(c->gz_use));  // the buffer about to be replaced may still be in use
        HIPCHK(c->d_gz_ws.reserve(gzip_ws_bytes(waves)));
    }
    if (!c->gz_use) HIPCHK(hipEventCreateWithFlags(&c->gz_use, hipEventDisableTiming));
    if (c->gz_valid && c->gz_stream != s) HIPCHK(hipStreamWaitEvent(s, c->gz_use, 0));
    HIPCHK(launch_gzip(d_in, d_out, d_blocks, n, d_out_len, bgzf, reinterpret_cast<uint32_t*>(c->d_gz_ws.p), waves, s));
    HIPCHK(hipEventRecord(c->gz_use, s));
    c->gz_stream = s;
    c->gz_valid = true;
    return 0;
}

int64_t cramgpu_gzip_bound(int64_t n) {
    if (n < 0) return fail(CRAMGPU_E_ARG, "gzip_bound: negative length");
    // worst case = stored blocks: 5 bytes per 32 KiB block (+ up to 4 of the writer's word granularity), 18 of header, 8 of trailer
    return n + 5 * ((n + 32767) / 32768 + 1) + 18 + 8 + 8;
}

int64_t cramgpu_bgzf_index(const uint8_t* data, int64_t len, int64_t* off, int32_t* usize, int64_t cap) {
    if (!data || len < 0 || (cap > 0 && (!off || !usize))) return fail(CRAMGPU_E_ARG, "bgzf_index: bad argument");
    int64_t at = 0, n = 0;
    while (at < len) {  // SAM specification 4.1: gzip member with FEXTRA, subfield 'B' 'C' of 2 bytes = BSIZE (size - 1)
        if (at + 18 > len || data[at] != 0x1F || data[at + 1] != 0x8B || data[at + 2] != 8 || !(data[at + 3] & 4))
            return fail(CRAMGPU_E_CORRUPT, "bgzf_index: not a BGZF block header");
        const int64_t xlen = (int64_t)data[at + 10] | (int64_t)data[at + 11] << 8;
        if (at + 12 + xlen > len) return fail(CRAMGPU_E_CORRUPT, "bgzf_index: extra field runs past the end");
        int64_t bsize = -1;
        for (int64_t q = at + 12; q + 4 <= at + 12 + xlen;) {
            const int64_t slen = (int64_t)data[q + 2] | (int64_t)data[q + 3] << 8;
            if (data[q] == 'B' && data[q + 1] == 'C' && slen == 2 && q + 6 <= at + 12 + xlen)
                bsize = ((int64_t)data[q + 4] | (int64_t)data[q + 5] << 8) + 1;
            q += 4 + slen;
        }
        if (bsize < 12 + xlen + 8 || at + bsize > len) return fail(CRAMGPU_E_CORRUPT, "bgzf_index: bad block size");
        if (n < cap) {
            off[n] = at;
            const uint8_t* t = data + at + bsize - 4;
            usize[n] = (int32_t)((uint32_t)t[0] | (uint32_t)t[1] << 8 | (uint32_t)t[2] << 16 | (uint32_t)t[3] << 24);
        }
        n++;
        at += bsize;
    }
    if (n < cap) off[n] = at;
    return n;
}

int32_t cramgpu_bam_index_dev(cramgpu_ctx* c, const uint8_t* d_bam, int64_t len, int32_t n_ref, int64_t* d_rec_off, int64_t cap,
                              int64_t* d_count, int32_t* d_status, void* stream) {
    if (!c || !d_bam || len < 0 || n_ref < 0 || cap < 0 || (cap > 0 && !d_rec_off) || !d_count || !d_status)
        return fail(CRAMGPU_E_ARG, "bam_index_dev: bad argument");
    if (len >= ((int64_t)1 << 31) * 4096) return fail(CRAMGPU_E_ARG, "bam_index_dev: stripe too long");
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t s = pick(c, stream);
    if (int rc = enter(c, s)) return rc;
    CallScope scope{c, s};
    const size_t n_seg = bam_index_segments(len);
    HIPCHK(c->d_idx_ws.reserve(3 * n_seg + 2));  // entry | exit | base (int64); counts share d_counts
    HIPCHK(c->d_counts.reserve(n_seg + 1));
    HIPCHK(c->d_scan_ws.reserve(feature_scan_ws_elems((int32_t)n_seg)));
    int64_t* ws = c->d_idx_ws.p;
    HIPCHK(launch_bam_index(d_bam, len, n_ref, d_rec_off, cap, d_count, d_status, ws, ws + n_seg, c->d_counts.p, ws + 2 * n_seg,
                            c->d_total.p, c->d_scan_ws.p, s));
    return 0;
}

int64_t cramgpu_bam_index(const uint8_t* bam, int64_t len, int64_t* rec_off, int64_t cap) {
    if (!bam || len < 0 || (cap > 0 && !rec_off)) return fail(CRAMGPU_E_ARG, "bam_index: bad argument");
    int64_t at = 0, n = 0;
    while (at < len) {  // Bam_OBA_Supplier.java:49-62
        if (at + 4 > len) return fail(CRAMGPU_E_CORRUPT, "bam_index: the stripe ends inside a block size");
        const int32_t bs = (int32_t)((uint32_t)bam[at] | (uint32_t)bam[at + 1] << 8 | (uint32_t)bam[at + 2] << 16 | (uint32_t)bam[at + 3] << 24);
        if (bs < 32 || at + 4 + (int64_t)bs > len) return fail(CRAMGPU_E_CORRUPT, "bam_index: invalid record length");
        if (n < cap) rec_off[n] = at;
        n++;
        at += 4 + (int64_t)bs;
    }
    if (n < cap) rec_off[n] = at;
    return n;
}

int32_t cramgpu_assign_codes_dev(cramgpu_ctx* c, int32_t n_reads, const int32_t* d_read_cont, int32_t n_cont,
                                 const cramgpu_features* feats, const uint8_t* matrices, void* stream) {
    if (!c || !feats || !matrices || n_reads < 0 || n_cont < 1) return fail(CRAMGPU_E_ARG, "assign_codes: bad argument");
    if (n_reads == 0) return 0;
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t s = pick(c, stream);
    if (int rc = enter(c, s)) return rc;
    CallScope scope{c, s};
    std::vector<uint8_t> codes((size_t)n_cont * 5 * 256), tmp(20);
    for (int i = 0; i < n_cont; i++) matrix_tables(matrices + 5 * i, codes.data() + (size_t)i * 5 * 256, tmp.data());
    uint8_t* d_codes;
    SmallSlot* slot;
    if (int rc = stage_small(c, codes.data(), codes.size(), s, &d_codes, &slot)) return rc;
    HIPCHK(launch_assign_codes(n_reads, d_read_cont, feats->off, feats->op, feats->a, d_codes, s));
    return release_small(slot, s);
}

int32_t cramgpu_matrices_assign_codes_dev(cramgpu_ctx* c, int32_t n_reads, const int32_t* d_read_cont, int32_t n_cont,
                                          const cramgpu_features* feats, const int64_t* d_sub_hist, uint8_t* d_matrices,
                                          void* stream) {
    if (!c || !feats || !d_sub_hist || !d_matrices || n_reads < 0 || n_cont < 1)
        return fail(CRAMGPU_E_ARG, "matrices_assign_codes: bad argument");
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t s = pick(c, stream);
    if (int rc = enter(c, s)) return rc;
    CallScope scope{c, s};
    HIPCHK(c->d_codes25.reserve((size_t)n_cont * 25));
    HIPCHK(launch_matrices_and_codes(d_sub_hist, n_cont, d_matrices, c->d_codes25.p, n_reads, d_read_cont, feats->off, feats->op,
                                     feats->a, nullptr, 0, s));
    return 0;
}

int32_t cramgpu_restore_bases_dev(cramgpu_ctx* c, int32_t n_reads, const int32_t* d_read_len, const int32_t* d_aln_start,
                                  const uint8_t* d_rflags, const int32_t* d_read_cont, int32_t n_cont,
                                  const cramgpu_refwin* d_wins, const cramgpu_features* feats, const uint8_t* d_seq,
                                  const int64_t* d_seq_off, int64_t seq_bytes, const uint8_t* d_ref, const uint8_t* matrices,
                                  uint8_t* d_out, const int64_t* d_out_off, int32_t* d_status, void* stream) {
    if (!c || !feats || !matrices || n_reads < 0 || n_cont < 1 || seq_bytes < 0)
        return fail(CRAMGPU_E_ARG, "restore_bases: bad argument");
    if (n_reads == 0) return 0;
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t s = pick(c, stream);
    if (int rc = enter(c, s)) return rc;
    CallScope scope{c, s};
    std::vector<uint8_t> basesof((size_t)n_cont * 20), tmp(5 * 256);
    for (int i = 0; i < n_cont; i++) matrix_tables(matrices + 5 * i, tmp.data(), basesof.data() + (size_t)i * 20);
    uint8_t* d_basesof;
    SmallSlot* slot;
    if (int rc = stage_small(c, basesof.data(), basesof.size(), s, &d_basesof, &slot)) return rc;
    HIPCHK(launch_restore_bases(n_reads, d_read_len, d_aln_start, d_rflags, d_read_cont, d_wins, *feats, d_seq, d_seq_off,
                                seq_bytes, d_ref, d_basesof, d_out, d_out_off, d_status, s));
    return release_small(slot, s);
}

int32_t cramgpu_restore_quals_dev(cramgpu_ctx* c, int32_t n_reads, const int32_t* d_read_len, const uint8_t* d_rflags,
                                  const cramgpu_features* feats, uint8_t default_qs, uint8_t* d_scores,
                                  const int64_t* d_out_off, int32_t* d_status, void* stream) {
    if (!c || !feats || n_reads < 0) return fail(CRAMGPU_E_ARG, "restore_quals: bad argument");
    if (n_reads == 0) return 0;
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t s = pick(c, stream);
    if (int rc = enter(c, s)) return rc;
    CallScope scope{c, s};
    HIPCHK(launch_restore_quals(n_reads, d_read_len, d_rflags, *feats, default_qs, d_scores, d_out_off, d_status, s));
    return 0;
}

// QualityScorePreservation.parsePolicies :167-191, parseSinglePolicy :193-256, readParam :139-146, readTreatment :148-165
int32_t cramgpu_parse_policies(const char* spec, cramgpu_policy* out, int32_t max_pol) {
    if (!spec || !out || max_pol < 0) return fail(CRAMGPU_E_ARG, "parse_policies: bad argument");
    std::vector<cramgpu_policy> pols;
    const char* p = spec;
    while (true) {
        const char* e = strchr(p, '-');
        std::string tok = e ? std::string(p, e) : std::string(p);
        if (!tok.empty()) {  // String.split drops nothing in the middle, but an empty token parses to a no-op policy
            cramgpu_policy pol{};
            pol.read_param = -1;
            int treatment = -1;
            size_t i = 0;
            auto read_param = [&]() {
                int v = 0;
                while (i < tok.size() && tok[i] >= '0' && tok[i] <= '9') v = v * 10 + (tok[i++] - '0');
                return v;
            };
            auto read_treatment = [&]() {
                const int v = read_param();
                return v == 0 ? 2 : v == 40 ? 0 : 1;  // 0 -> DROP, 40 -> PRESERVE, anything else -> BIN
            };
            auto add_cat = [&](int cat, int param) {
                if (pol.n_base_cat < 8) {
                    pol.base_cat[pol.n_base_cat] = cat;
                    pol.base_param[pol.n_base_cat] = param;
                    pol.n_base_cat++;
                }
            };
            while (i < tok.size()) {
                const char code = tok[i++];
                switch (code) {
                    case 'R': add_cat(1, -1); treatment = read_treatment(); break;
                    case 'N': add_cat(2, -1); treatment = read_treatment(); break;
                    case 'X': add_cat(5, read_param()); break;
                    case 'D': add_cat(3, -1); treatment = read_treatment(); break;
                    case 'M': pol.read_cat = 2; pol.read_param = read_param(); break;
                    case 'm': pol.read_cat = 3; pol.read_param = read_param(); break;
                    case 'U': pol.read_cat = 1; treatment = read_treatment(); break;
                    case 'P': add_cat(4, read_param()); treatment = read_treatment(); break;
                    case 'I': add_cat(6, -1); treatment = read_treatment(); break;
                    case '_': treatment = read_treatment(); break;
                    case '*': pol.read_cat = 4; treatment = read_treatment(); break;
                    default: return fail(CRAMGPU_E_ARG, "Unknown read or base category");
                }
                if (treatment < 0) treatment = 0;  // :249-250 default PRESERVE
            }
            pol.treatment = treatment < 0 ? 0 : treatment;
            pols.push_back(pol);
        }
        if (!e) break;
        p = e + 1;
    }
    // :178-188 Collections.sort (stable) by treatment ordinal, descending: DROP(2), BIN(1), PRESERVE(0)
    std::stable_sort(pols.begin(), pols.end(), [](const cramgpu_policy& a, const cramgpu_policy& b) { return a.treatment > b.treatment; });
    if ((int32_t)pols.size() > max_pol) return fail(CRAMGPU_E_INSUFFICIENT, "parse_policies: too many policies");
    for (size_t k = 0; k < pols.size(); k++) out[k] = pols[k];
    return (int32_t)pols.size();
}

// The pile-up selectors X<n> (LOWER_COVERAGE) and P<n> (PILEUP) read the coverage / mismatch tracks that
// CramSerilization.updateTracks (src/main/java/htsjdk/samtools/cram/CramSerilization.java:114-128, 225-257) fills before
// the per-record loop.  This library does not build those tracks (out of scope), so it refuses
// every term whose result could depend on them instead of silently answering with all-zero tracks.  Accepted are the
// terms that select nothing whatever the tracks hold: X<n> with n <= 0 (coverage < n is never true) and P<n> with
// n >= 9999 (more than n mismatching reads on one position; P9999 is the no-op the survey recommends to make the Java
// CLI create its ReferenceTracks).
static int policies_supported(const cramgpu_policy* pol, int32_t n_pol) {
    for (int32_t k = 0; k < n_pol; k++)
        for (int32_t i = 0; i < pol[k].n_base_cat && i < 8; i++) {
            const int cat = pol[k].base_cat[i], prm = pol[k].base_param[i];
            if ((cat == 5 && prm > 0) || (cat == 4 && prm < 9999))
                return fail(CRAMGPU_E_UNSUPPORTED,
                            "lossy model: X<n> / P<n> terms need the coverage and mismatch tracks of CramSerilization.updateTracks, "
                            "which this library does not build (only the no-ops X0 and P>=9999 are accepted)");
        }
    return 0;
}

static bool host_read_gate(const cramgpu_policy& p, bool unmapped, int mapq) {
    switch (p.read_cat) {
        case 0: case 4: return true;
        case 1: return unmapped;
        case 3: return mapq < p.read_param;
        case 2: return mapq > p.read_param;
        default: return false;
    }
}

// ReferenceTracks.ensureRange :127-133 / moveForwardTo :87-119 replayed in read order, from window position `position`
// (1 for a fresh ReferenceTracks); *pos_out = the position afterwards.
static int32_t tracks_windows_from(int32_t n_reads, const int32_t* aln_start, const int32_t* aln_span, const uint8_t* rflags,
                                   const uint8_t* mapq, const cramgpu_policy* pol, int32_t n_pol, int64_t ref_len,
                                   int32_t window, int64_t position, int32_t* wpos, int32_t* throws_at, int64_t* pos_out) {
    if (throws_at) *throws_at = -1;
    if (int rc = policies_supported(pol, n_pol)) return rc;
    const int64_t W = window < ref_len ? window : ref_len;
    for (int32_t r = 0; r < n_reads; r++) {
        wpos[r] = 0;
        if (rflags[r] & (CRAMGPU_RF_NULL_QUALS | CRAMGPU_RF_SKIP)) continue;
        bool need = false;
        for (int k = 0; k < n_pol; k++)
            if (pol[k].n_base_cat > 0 && host_read_gate(pol[k], (rflags[r] & CRAMGPU_RF_UNMAPPED) != 0, mapq[r])) need = true;
        if (!need || aln_start[r] == 0) continue;
        const int64_t start = aln_start[r];
        auto throws = [&]() {
            if (throws_at) *throws_at = r;
            return fail(CRAMGPU_E_JAVA_THROWS, "ReferenceTracks would throw (reads not sorted by start, or a start beyond the contig)");
        };
        if (start < position) return throws();
        if (start + aln_span[r] > position + W) {
            int64_t np = start;
            if (np - 1 >= ref_len) return throws();
            if (np > ref_len - W + 1) np = ref_len - W + 1;
            position = np;
        }
        wpos[r] = (int32_t)position;
    }
    if (pos_out) *pos_out = position;
    return 0;
}

int32_t cramgpu_tracks_windows(int32_t n_reads, const int32_t* aln_start, const int32_t* aln_span, const uint8_t* rflags,
                               const uint8_t* mapq, const cramgpu_policy* pol, int32_t n_pol, int64_t ref_len,
                               int32_t window, int32_t* wpos, int32_t* throws_at) {
    return tracks_windows_from(n_reads, aln_start, aln_span, rflags, mapq, pol, n_pol, ref_len, window, 1, wpos, throws_at,
                               nullptr);
}

int32_t cramgpu_lossy_quals_dev(cramgpu_ctx* c, const cramgpu_reads* reads, const uint8_t* d_ref, const int32_t* d_wpos,
                                int32_t window, const cramgpu_policy* pol, int32_t n_pol, uint8_t* d_scores,
                                uint8_t* d_force, int32_t* d_status, void* stream) {
    if (!c || !reads || n_pol < 0 || (n_pol > 0 && !pol)) return fail(CRAMGPU_E_ARG, "lossy_quals: bad argument");
    if (int rc = policies_supported(pol, n_pol)) return rc;
    if (reads->n_reads == 0) return 0;
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t s = pick(c, stream);
    if (int rc = enter(c, s)) return rc;
    CallScope scope{c, s};
    cramgpu_policy dummy{};
    uint8_t* d_pol;
    SmallSlot* slot;
    if (int rc = stage_small(c, n_pol ? (const void*)pol : (const void*)&dummy, sizeof(cramgpu_policy) * (size_t)std::max(n_pol, 1), s,
                             &d_pol, &slot))
        return rc;
    // global mask scratch for reads longer than the LDS mask (sized by the caller-visible arena: not known here, so
    // it is sized from the largest base offset + length, read back once)
    if (reads->max_read_len < 0 || reads->n_bases < 0) return fail(CRAMGPU_E_ARG, "lossy_quals: n_bases / max_read_len unset");
    HIPCHK(c->d_maskws.reserve(reads->max_read_len + 1 > 16384 ? (size_t)reads->n_bases + (size_t)reads->n_reads + 64 : 64));
    HIPCHK(c->d_list.reserve((size_t)reads->n_reads + 1));  // reads the lane-per-read kernel leaves to the wave kernel
    HIPCHK(launch_lossy_quals(*reads, d_ref, d_wpos, window, (const cramgpu_policy*)d_pol, n_pol, d_scores, d_force,
                              d_status, c->d_maskws.p, c->d_list.p, reinterpret_cast<uint32_t*>(c->d_list.p + reads->n_reads),
                              s));
    return release_small(slot, s);
}

int64_t cramgpu_merge_quality_features_dev(cramgpu_ctx* c, int32_t n_reads, const int32_t* d_read_len,
                                           const int64_t* d_base_off, const uint8_t* d_scores, const uint8_t* d_force,
                                           const int32_t* d_qstatus, const cramgpu_features* in, cramgpu_features* out,
                                           void* stream) {
    if (!c || !in || !out || n_reads < 0 || !out->off) return fail(CRAMGPU_E_ARG, "merge_quality_features: bad argument");
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t s = pick(c, stream);
    if (int rc = enter(c, s)) return rc;
    CallScope scope{c, s};
    HIPCHK(c->d_counts.reserve((size_t)n_reads + 1));
    if (n_reads > 0)
        HIPCHK(launch_merge_quality(n_reads, d_read_len, d_base_off, d_scores, d_force, d_qstatus, *in, *out, c->d_counts.p,
                                    false, nullptr, nullptr, s));
    HIPCHK(c->d_scan_ws.reserve(feature_scan_ws_elems(n_reads)));
    HIPCHK(launch_feature_scan(n_reads, c->d_counts.p, out->off, c->d_total.p, c->d_scan_ws.p, s));
    HIPCHK(hipMemcpyAsync(c->h_total.p, c->d_total.p, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const int64_t total = c->h_total.p[0];
    if (total > out->cap) return fail(CRAMGPU_E_INSUFFICIENT, "merge_quality_features: capacity too small");
    if (n_reads > 0 && total > 0)
        HIPCHK(launch_merge_quality(n_reads, d_read_len, d_base_off, d_scores, d_force, d_qstatus, *in, *out, c->d_counts.p,
                                    true, nullptr, nullptr, s));
    return total;
}

}  // extern "C"

// =====================================================================================================
// Level 2, host form: record jobs over host stripes
// =====================================================================================================
namespace {

// Device memory of a record job, carved out of one allocation: first pass measures, second pass hands out.
struct Carver {
    uint8_t* base;
    size_t off = 0;
    template <class T>
    T* take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
};

// windows of the containers of a stripe: the whole contig, or the caller's region clipped to it.  The reference arena
// argument of the record kernels is the null pointer and ref_off the contig's device ADDRESS: contigs are separate
// allocations (uploaded at different times), and "arena + offset" is all the kernels do with the pair.
int build_windows(cramgpu_ctx* c, int32_t n_cont, const int32_t* cont_contig, const int64_t* start0, const int64_t* len,
                  cramgpu_refwin* out) {
    for (int32_t k = 0; k < n_cont; k++) {
        const int32_t id = cont_contig[k];
        if (id < 0 || id >= (int32_t)c->contigs.size() || c->contigs[id].len < 0)
            return fail(CRAMGPU_E_ARG, "record job: a container names a contig that was not uploaded (cramgpu_ref_upload)");
        const auto& g = c->contigs[id];
        int64_t s0 = 0, ln = g.len;
        if (start0 && len) {
            s0 = std::max<int64_t>(0, start0[k]);
            ln = std::max<int64_t>(0, std::min<int64_t>(len[k], g.len - s0));
        }
        out[k] = cramgpu_refwin{(int64_t)(uintptr_t)(g.d.p + s0), ln, s0};
    }
    return 0;
}

}  // namespace

extern "C" {

int32_t cramgpu_ref_upload(cramgpu_ctx* c, int32_t contig, const uint8_t* bases, int64_t len) {
    if (!c || contig < 0 || contig >= CRAMGPU_MAX_CONTIGS || len < 0 || (len > 0 && !bases))
        return fail(CRAMGPU_E_ARG, "cramgpu_ref_upload: bad argument");
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    for (auto& j : c->jobs)  // a job in flight may be reading the contig this call replaces
        if (j.busy && j.kind != 0) HIPCHK(hipStreamSynchronize(j.stream));
    if ((size_t)contig >= c->contigs.size()) c->contigs.resize((size_t)contig + 1);
    auto& g = c->contigs[contig];
    HIPCHK(g.d.reserve((size_t)len + 32));  // kernels read whole 16-byte units
    if (len) HIPCHK(hipMemcpy(g.d.p, bases, (size_t)len, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(g.d.p + len, 0, 32));
    g.len = len;
    g.tracks_pos = 1;
    return 0;
}

int32_t cramgpu_ref_drop(cramgpu_ctx* c, int32_t contig) {
    if (!c || contig < 0) return fail(CRAMGPU_E_ARG, "cramgpu_ref_drop: bad argument");
    std::lock_guard<std::mutex> lk(c->mu);
    if ((size_t)contig >= c->contigs.size() || c->contigs[contig].len < 0) return 0;
    HIPCHK(hipSetDevice(c->device));
    for (auto& j : c->jobs)
        if (j.busy && j.kind != 0) HIPCHK(hipStreamSynchronize(j.stream));
    c->contigs[contig].d.release();
    c->contigs[contig].len = -1;
    return 0;
}

cramgpu_job* cramgpu_submit_reads(cramgpu_ctx* c, const cramgpu_host_reads* R, const char* lossy_spec, int32_t window,
                                  cramgpu_host_read_results* out) {
    if (!c || !R || !out || R->n_reads < 0 || R->n_cont < 1 || R->n_bases < 0 || R->n_cigar_ops < 0 || !R->cont_reads ||
        !R->cont_contig || !out->feats.off || out->feats.cap < 0 || !out->matrices ||
        (R->n_reads > 0 && (!R->bases || !R->read_len || !R->aln_start || !R->rflags || !R->cigar || !R->n_cigar)) ||
        (out->feats.cap > 0 && (!out->feats.op || !out->feats.pos || !out->feats.a || !out->feats.b))) {
        fail(CRAMGPU_E_ARG, "cramgpu_submit_reads: bad argument");
        return nullptr;
    }
    const bool lossy = lossy_spec != nullptr;
    if (lossy && (!R->quals || !R->mapq || !out->scores || !out->force || !out->status)) {
        fail(CRAMGPU_E_ARG, "cramgpu_submit_reads: the lossy model needs quals, mapq and the scores / force / status outputs");
        return nullptr;
    }
    const int32_t n = R->n_reads, nc = R->n_cont;
    const int64_t nb = R->n_bases, ncg = R->n_cigar_ops, cap = out->feats.cap;
    // host-side bookkeeping: container table, totals, longest read
    std::vector<int32_t> cont_first((size_t)nc + 1, 0);
    for (int32_t k = 0; k < nc; k++) {
        if (R->cont_reads[k] < 0) {
            fail(CRAMGPU_E_ARG, "cramgpu_submit_reads: negative container size");
            return nullptr;
        }
        cont_first[k + 1] = cont_first[k] + R->cont_reads[k];
    }
    int64_t sum_len = 0, sum_cg = 0;
    int32_t max_len = 0;
    for (int32_t r = 0; r < n; r++) {
        if (R->read_len[r] < 0 || R->n_cigar[r] < 0) {
            fail(CRAMGPU_E_ARG, "cramgpu_submit_reads: negative read or CIGAR length");
            return nullptr;
        }
        sum_len += R->read_len[r];
        sum_cg += R->n_cigar[r];
        max_len = std::max(max_len, R->read_len[r]);
    }
    if (cont_first[nc] != n || sum_len != nb || sum_cg != ncg) {
        fail(CRAMGPU_E_ARG, "cramgpu_submit_reads: cont_reads / n_bases / n_cigar_ops do not add up");
        return nullptr;
    }
    cramgpu_policy pol[64];
    int32_t n_pol = 0;
    if (lossy) {
        n_pol = cramgpu_parse_policies(lossy_spec, pol, 64);
        if (n_pol < 0) return nullptr;
        if (policies_supported(pol, n_pol)) return nullptr;
    }
    if (window <= 0) window = 1000000;  // ReferenceTracks.java:23

    std::lock_guard<std::mutex> lk(c->mu);
    cramgpu_job* j = grab_job(c, 1);
    if (!j) return nullptr;
    hipStream_t s = j->stream;
    auto ok = [&](hipError_t e) { return e == hipSuccess || (fail(CRAMGPU_E_HIP, hipGetErrorString(e)), false); };

    // ---- pinned staging: windows, container table, policies, window positions, the count on its way back ----
    const size_t pin_bytes = 256 + (size_t)nc * sizeof(cramgpu_refwin) + 256 + ((size_t)nc + 1) * 4 + 256 +
                             sizeof(pol) + 256 + (lossy ? (size_t)n * 4 : 0) + 256;
    if (!ok(j->pin.reserve(pin_bytes))) return nullptr;
    Carver hp{j->pin.p};
    j->h_total = hp.take<int64_t>(2);
    j->h_total[0] = 0;
    cramgpu_refwin* h_win = hp.take<cramgpu_refwin>(nc);
    int32_t* h_first = hp.take<int32_t>((size_t)nc + 1);
    cramgpu_policy* h_pol = hp.take<cramgpu_policy>(64);
    int32_t* h_wpos = lossy ? hp.take<int32_t>(n) : nullptr;
    if (build_windows(c, nc, R->cont_contig, nullptr, nullptr, h_win)) return nullptr;
    memcpy(h_first, cont_first.data(), ((size_t)nc + 1) * 4);
    if (lossy) {
        memcpy(h_pol, pol, sizeof(pol));
        // ReferenceTracks.ensureRange replayed container by container; a contig's window position carries over from
        // one container (and one submission) to the next, as the writer's per-reference ReferenceTracks does
        // (CRAMContainerStreamWriter.java:209-227)
        std::vector<int32_t> span;
        int64_t co = 0;  // CIGAR offset of the read at hand
        for (int32_t k = 0; k < nc; k++) {
            const int32_t r0 = cont_first[k], m = R->cont_reads[k];
            span.assign((size_t)m, 0);
            for (int32_t r = 0; r < m; r++) {
                int64_t ref_len = 0;
                for (int32_t e = 0; e < R->n_cigar[r0 + r]; e++) {
                    const uint32_t ce = R->cigar[co + e];
                    const uint32_t op = ce & 15u;
                    if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_len += ce >> 4;
                }
                co += R->n_cigar[r0 + r];
                span[r] = (int32_t)(ref_len - 1);
            }
            auto& g = c->contigs[R->cont_contig[k]];
            int32_t at = -1;
            int64_t pos_out = g.tracks_pos;
            const int rc = tracks_windows_from(m, R->aln_start + r0, span.data(), R->rflags + r0, R->mapq + r0, pol, n_pol, g.len,
                                               window, g.tracks_pos, h_wpos + r0, &at, &pos_out);
            if (rc) {
                j->early_rc = rc;  // reported by wait; nothing is launched for this job
                break;
            }
            g.tracks_pos = pos_out;
        }
    }

    // ---- device buffers ----
    struct Dev {
        uint8_t *bases, *quals, *rflags, *mapq, *scores, *force, *mask_ws, *mats, *codes25;
        int32_t *read_len, *aln_start, *n_cigar, *read_cont, *first, *counts, *list, *wpos, *qstatus;
        uint32_t* cigar;
        int64_t *base_off, *cigar_off, *scan_ws, *total, *hist;
        cramgpu_refwin* win;
        cramgpu_policy* pol;
        cramgpu_features F1, F2;
    } D{};
    auto carve_feats = [&](Carver& cv, cramgpu_features& F) {
        F.op = cv.take<uint8_t>((size_t)cap + 16);
        F.pos = cv.take<int32_t>((size_t)cap + 4);
        F.a = cv.take<int32_t>((size_t)cap + 4);
        F.b = cv.take<int32_t>((size_t)cap + 4);
        F.off = cv.take<int64_t>((size_t)n + 1);
        F.cap = cap;
    };
    auto layout = [&](Carver& cv) {
        D.bases = cv.take<uint8_t>((size_t)nb + 32);
        D.quals = lossy ? cv.take<uint8_t>((size_t)nb + 32) : nullptr;
        D.read_len = cv.take<int32_t>((size_t)n + 4);
        D.aln_start = cv.take<int32_t>((size_t)n + 4);
        D.n_cigar = cv.take<int32_t>((size_t)n + 4);
        D.rflags = cv.take<uint8_t>((size_t)n + 16);
        D.mapq = cv.take<uint8_t>((size_t)n + 16);
        D.cigar = cv.take<uint32_t>((size_t)ncg + 4);
        D.base_off = cv.take<int64_t>((size_t)n + 1);
        D.cigar_off = cv.take<int64_t>((size_t)n + 1);
        D.read_cont = cv.take<int32_t>((size_t)n + 4);
        D.first = cv.take<int32_t>((size_t)nc + 1);
        D.win = cv.take<cramgpu_refwin>(nc);
        D.counts = cv.take<int32_t>((size_t)n + 4);
        D.list = cv.take<int32_t>((size_t)n + 4);
        D.scan_ws = cv.take<int64_t>(feature_scan_ws_elems(n));
        D.total = cv.take<int64_t>(4);
        D.hist = cv.take<int64_t>((size_t)nc * 25);
        D.mats = cv.take<uint8_t>((size_t)nc * 5 + 16);
        D.codes25 = cv.take<uint8_t>((size_t)nc * 25 + 16);
        carve_feats(cv, D.F1);
        if (lossy) {
            D.pol = cv.take<cramgpu_policy>(64);
            D.wpos = cv.take<int32_t>((size_t)n + 4);
            D.scores = cv.take<uint8_t>((size_t)nb + 32);
            D.force = cv.take<uint8_t>((size_t)n + 16);
            D.qstatus = cv.take<int32_t>((size_t)n + 4);
            D.mask_ws = cv.take<uint8_t>(max_len + 1 > 16384 ? (size_t)nb + (size_t)n + 64 : 64);
            carve_feats(cv, D.F2);
        }
    };
    Carver measure{nullptr};
    layout(measure);
    if (!ok(j->arena.reserve(measure.off + 256))) return nullptr;
    Carver cv{j->arena.p};
    layout(cv);
    j->feat_cap = cap;
    j->h_feat_count = out->feats.off + n;

    const bool run = j->early_rc == 0;
    bool good = true;
    auto up = [&](void* d, const void* h, size_t bytes) {
        if (good && bytes) good = ok(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s));
    };
    auto step = [&](hipError_t e) {
        if (good) good = ok(e);
    };
    if (run) {
        up(D.bases, R->bases, (size_t)nb);
        if (lossy) up(D.quals, R->quals, (size_t)nb);
        up(D.read_len, R->read_len, (size_t)n * 4);
        up(D.aln_start, R->aln_start, (size_t)n * 4);
        up(D.n_cigar, R->n_cigar, (size_t)n * 4);
        up(D.rflags, R->rflags, (size_t)n);
        if (R->mapq) up(D.mapq, R->mapq, (size_t)n);
        up(D.cigar, R->cigar, (size_t)ncg * 4);
        up(D.win, h_win, (size_t)nc * sizeof(cramgpu_refwin));
        up(D.first, h_first, ((size_t)nc + 1) * 4);
        step(hipMemsetAsync(D.hist, 0, (size_t)nc * 25 * sizeof(int64_t), s));
        // offsets and container indices on the device (no 16 bytes per read across PCIe for what a scan gives)
        step(launch_feature_scan(n, D.read_len, D.base_off, D.total, D.scan_ws, s));
        step(launch_feature_scan(n, D.n_cigar, D.cigar_off, D.total, D.scan_ws, s));
        step(launch_fill_read_cont(n, D.first, nc, D.read_cont, s));
        cramgpu_reads dr{};
        dr.n_reads = n, dr.n_cont = nc, dr.n_bases = nb, dr.max_read_len = max_len;
        dr.bases = D.bases, dr.quals = D.quals, dr.base_off = D.base_off, dr.read_len = D.read_len;
        dr.aln_start = D.aln_start, dr.rflags = D.rflags, dr.mapq = D.mapq, dr.cigar = D.cigar;
        dr.cigar_off = D.cigar_off, dr.n_cigar = D.n_cigar, dr.read_cont = D.read_cont, dr.win = D.win;
        const uint8_t* ref0 = nullptr;  // windows carry device addresses (build_windows)
        uint32_t* list_n = reinterpret_cast<uint32_t*>(D.list + n);
        // K3: count, scan, write (the write pass checks the capacity on the device: nothing here waits for the count)
        if (n > 0) step(launch_extract_features(dr, ref0, D.F1, nullptr, D.counts, false, D.list, list_n, nullptr, s));
        step(launch_feature_scan(n, D.counts, D.F1.off, D.total, D.scan_ws, s));
        if (n > 0) step(launch_extract_features(dr, ref0, D.F1, D.hist, D.counts, true, D.list, list_n, D.total, s));
        // F3 + codes
        // (the guards: the first set's offsets are the full scan even when the set did not fit and was not written)
        step(launch_matrices_and_codes(D.hist, nc, D.mats, D.codes25, n, D.read_cont, D.F1.off, D.F1.op, D.F1.a, D.total, D.F1.cap, s));
        const cramgpu_features* Fout = &D.F1;
        if (lossy) {
            up(D.pol, h_pol, sizeof(pol));
            up(D.wpos, h_wpos, (size_t)n * 4);
            if (n > 0) {
                // K3's list is free again by now on this stream
                step(launch_lossy_quals(dr, ref0, D.wpos, window, D.pol, n_pol, D.scores, D.force, D.qstatus, D.mask_ws, D.list,
                                        list_n, s));
                step(launch_merge_quality(n, D.read_len, D.base_off, D.scores, D.force, D.qstatus, D.F1, D.F2, D.counts, false,
                                          nullptr, D.total, s));
            }
            step(launch_feature_scan(n, D.counts, D.F2.off, D.total + 1, D.scan_ws, s));
            // the merged set only exists if the first one fitted; its own count decides its write pass
            if (n > 0)
                step(launch_merge_quality(n, D.read_len, D.base_off, D.scores, D.force, D.qstatus, D.F1, D.F2, D.counts, true,
                                          D.total + 1, D.total, s));
            Fout = &D.F2;
        }
        // phase 1 of the way back: offsets and the count
        const int64_t* d_count = lossy ? D.total + 1 : D.total;
        if (good) good = ok(hipMemcpyAsync(out->feats.off, Fout->off, ((size_t)n + 1) * 8, hipMemcpyDeviceToHost, s));
        if (good) good = ok(hipMemcpyAsync(j->h_total, d_count, 8, hipMemcpyDeviceToHost, s));
        j->back.push_back({out->feats.op, Fout->op, 1, true});
        j->back.push_back({out->feats.pos, Fout->pos, 4, true});
        j->back.push_back({out->feats.a, Fout->a, 4, true});
        j->back.push_back({out->feats.b, Fout->b, 4, true});
        j->back.push_back({out->matrices, D.mats, (size_t)nc * 5, false});
        if (out->sub_hist) j->back.push_back({out->sub_hist, D.hist, (size_t)nc * 25 * 8, false});
        if (lossy) {
            j->back.push_back({out->scores, D.scores, (size_t)nb, false});
            j->back.push_back({out->force, D.force, (size_t)n, false});
            j->back.push_back({out->status, D.qstatus, (size_t)n * 4, false});
        }
    } else {
        j->h_feat_count = nullptr;
    }
    if (!good || !ok(hipEventRecord(j->coded, s))) return nullptr;
    j->busy = true;
    return j;
}

cramgpu_job* cramgpu_submit_restore(cramgpu_ctx* c, const cramgpu_host_records* R, cramgpu_host_restore_results* out) {
    if (!c || !R || !out || R->n_reads < 0 || R->n_cont < 1 || R->n_seq < 0 || !R->cont_reads || !R->cont_contig ||
        !R->matrices || !R->feats.off || !out->bases || !out->status_bases ||
        (R->n_reads > 0 && (!R->read_len || !R->aln_start || !R->rflags)) || (out->scores && !out->status_quals)) {
        fail(CRAMGPU_E_ARG, "cramgpu_submit_restore: bad argument");
        return nullptr;
    }
    const int32_t n = R->n_reads, nc = R->n_cont;
    const int64_t nf = R->feats.off[n];
    if (nf < 0 || (nf > 0 && (!R->feats.op || !R->feats.pos || !R->feats.a || !R->feats.b))) {
        fail(CRAMGPU_E_ARG, "cramgpu_submit_restore: feature arrays missing");
        return nullptr;
    }
    std::vector<int32_t> cont_first((size_t)nc + 1, 0);
    for (int32_t k = 0; k < nc; k++) cont_first[k + 1] = cont_first[k] + std::max(0, R->cont_reads[k]);
    int64_t nb = 0;
    for (int32_t r = 0; r < n; r++) {
        if (R->read_len[r] < 0) {
            fail(CRAMGPU_E_ARG, "cramgpu_submit_restore: negative read length");
            return nullptr;
        }
        nb += R->read_len[r];
    }
    if (cont_first[nc] != n) {
        fail(CRAMGPU_E_ARG, "cramgpu_submit_restore: cont_reads does not add up to n_reads");
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    cramgpu_job* j = grab_job(c, 2);
    if (!j) return nullptr;
    hipStream_t s = j->stream;
    auto ok = [&](hipError_t e) { return e == hipSuccess || (fail(CRAMGPU_E_HIP, hipGetErrorString(e)), false); };
    const size_t pin_bytes = 256 + (size_t)nc * sizeof(cramgpu_refwin) + 256 + ((size_t)nc + 1) * 4 + 256 + (size_t)nc * 20 + 256;
    if (!ok(j->pin.reserve(pin_bytes))) return nullptr;
    Carver hp{j->pin.p};
    j->h_total = hp.take<int64_t>(2);
    j->h_total[0] = 0;
    cramgpu_refwin* h_win = hp.take<cramgpu_refwin>(nc);
    int32_t* h_first = hp.take<int32_t>((size_t)nc + 1);
    uint8_t* h_basesof = hp.take<uint8_t>((size_t)nc * 20);
    if (build_windows(c, nc, R->cont_contig, R->cont_win_start0, R->cont_win_len, h_win)) return nullptr;
    memcpy(h_first, cont_first.data(), ((size_t)nc + 1) * 4);
    {
        uint8_t tmp[5 * 256];
        for (int32_t k = 0; k < nc; k++) matrix_tables(R->matrices + 5 * k, tmp, h_basesof + (size_t)k * 20);
    }
    struct Dev {
        uint8_t *rflags, *seq, *basesof, *bases, *scores;
        int32_t *read_len, *aln_start, *read_cont, *first, *st_b, *st_q;
        int64_t *out_off, *seq_off, *scan_ws, *total;
        cramgpu_refwin* win;
        cramgpu_features F;
    } D{};
    auto layout = [&](Carver& cv) {
        D.read_len = cv.take<int32_t>((size_t)n + 4);
        D.aln_start = cv.take<int32_t>((size_t)n + 4);
        D.rflags = cv.take<uint8_t>((size_t)n + 16);
        D.read_cont = cv.take<int32_t>((size_t)n + 4);
        D.first = cv.take<int32_t>((size_t)nc + 1);
        D.win = cv.take<cramgpu_refwin>(nc);
        D.basesof = cv.take<uint8_t>((size_t)nc * 20 + 16);
        D.out_off = cv.take<int64_t>((size_t)n + 1);
        D.seq_off = cv.take<int64_t>((size_t)n + 1);
        D.scan_ws = cv.take<int64_t>(feature_scan_ws_elems(n));
        D.total = cv.take<int64_t>(4);
        D.seq = cv.take<uint8_t>((size_t)R->n_seq + 32);
        D.bases = cv.take<uint8_t>((size_t)nb + 32);
        D.scores = cv.take<uint8_t>((size_t)nb + 32);
        D.st_b = cv.take<int32_t>((size_t)n + 4);
        D.st_q = cv.take<int32_t>((size_t)n + 4);
        D.F.op = cv.take<uint8_t>((size_t)nf + 16);
        D.F.pos = cv.take<int32_t>((size_t)nf + 4);
        D.F.a = cv.take<int32_t>((size_t)nf + 4);
        D.F.b = cv.take<int32_t>((size_t)nf + 4);
        D.F.off = cv.take<int64_t>((size_t)n + 1);
        D.F.cap = nf;
    };
    Carver measure{nullptr};
    layout(measure);
    if (!ok(j->arena.reserve(measure.off + 256))) return nullptr;
    Carver cv{j->arena.p};
    layout(cv);
    bool good = true;
    auto up = [&](void* d, const void* h, size_t bytes) {
        if (good && bytes) good = ok(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s));
    };
    auto step = [&](hipError_t e) {
        if (good) good = ok(e);
    };
    up(D.read_len, R->read_len, (size_t)n * 4);
    up(D.aln_start, R->aln_start, (size_t)n * 4);
    up(D.rflags, R->rflags, (size_t)n);
    up(D.first, h_first, ((size_t)nc + 1) * 4);
    up(D.win, h_win, (size_t)nc * sizeof(cramgpu_refwin));
    up(D.basesof, h_basesof, (size_t)nc * 20);
    up(D.F.op, R->feats.op, (size_t)nf);
    up(D.F.pos, R->feats.pos, (size_t)nf * 4);
    up(D.F.a, R->feats.a, (size_t)nf * 4);
    up(D.F.b, R->feats.b, (size_t)nf * 4);
    up(D.F.off, R->feats.off, ((size_t)n + 1) * 8);
    if (R->seq && R->n_seq) up(D.seq, R->seq, (size_t)R->n_seq);
    if (R->seq_off)
        up(D.seq_off, R->seq_off, (size_t)n * 8);
    else
        step(hipMemsetAsync(D.seq_off, 0, ((size_t)n + 1) * 8, s));
    if (out->scores) up(D.scores, out->scores, (size_t)nb);
    step(launch_feature_scan(n, D.read_len, D.out_off, D.total, D.scan_ws, s));
    step(launch_fill_read_cont(n, D.first, nc, D.read_cont, s));
    if (n > 0) {
        step(launch_restore_bases(n, D.read_len, D.aln_start, D.rflags, D.read_cont, D.win, D.F, D.seq, D.seq_off, R->n_seq, nullptr,
                                  D.basesof, D.bases, D.out_off, D.st_b, s));
        if (out->scores)
            step(launch_restore_quals(n, D.read_len, D.rflags, D.F, (uint8_t)R->default_qs, D.scores, D.out_off, D.st_q, s));
    }
    j->feat_cap = 0;
    j->back.push_back({out->bases, D.bases, (size_t)nb, false});
    j->back.push_back({out->status_bases, D.st_b, (size_t)n * 4, false});
    if (out->scores) {
        j->back.push_back({out->scores, D.scores, (size_t)nb, false});
        j->back.push_back({out->status_quals, D.st_q, (size_t)n * 4, false});
    }
    if (!good || !ok(hipEventRecord(j->coded, s))) return nullptr;
    j->busy = true;
    return j;
}

}  // extern "C"
