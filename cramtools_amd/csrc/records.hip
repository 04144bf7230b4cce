// K3 / K4 / K5 — record-level kernels of the cramtools hot path over flat (structure-of-arrays) containers.
//
// Reference:
//   F1/F2  htsjdk!build/Sam2CramRecordFactory#createVariations L205-264, #addSubstitutionsAndMaskedBases L297-335,
//          #addInsertion L282-292, #addSoftClip L268-272, htsjdk!build/Utils#normalizeBase L31-49
//   F3     htsjdk!build/CompressionHeaderFactory#buildFrequencies L196-209 (the 5x5 substitution histogram)
//   N1     src/main/java/htsjdk/samtools/cram/build/CramNormalizer.java:335-441 (restoreReadBases)
//   N2     CramNormalizer.java:287-333 (restoreQualityScores)
//   Q2     src/main/java/htsjdk/samtools/cram/lossy/QualityScorePreservation.java:258-301,332-489,
//          lossy/Binning.java:63-71, ref/ReferenceTracks.java:127-155
//
// Mapping: one wavefront per read; the CIGAR / feature list of the read is walked by the whole wave (uniform
// control flow) and the 64 lanes sweep the bases of each element with coalesced byte loads of the read and of the
// reference window.  Mismatches are compacted with a wave ballot + mbcnt prefix so that features leave in the
// reference's order (CIGAR order, ascending position).  Feature extraction runs twice (count, exclusive scan,
// write) because the feature arrays are dense across reads.
#include <cstdlib>

#include "internal.h"

namespace {

enum { OP_M = 0, OP_I = 1, OP_D = 2, OP_N = 3, OP_S = 4, OP_H = 5, OP_P = 6, OP_EQ = 7, OP_X = 8 };

__device__ __forceinline__ uint32_t mbcnt(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
// htsjdk!build/Utils#normalizeBase L31-49: acgtACGT -> upper case, everything else -> 'N'
__device__ __forceinline__ uint8_t normalize_base(uint8_t b) {
    const uint8_t lo = b | 0x20;
    return (lo == 'a' || lo == 'c' || lo == 'g' || lo == 't') ? (uint8_t)(b & 0xDF) : (uint8_t)'N';
}
__device__ __forceinline__ int acgtn_index(uint8_t b) {
    return b == 'A' ? 0 : b == 'C' ? 1 : b == 'G' ? 2 : b == 'T' ? 3 : b == 'N' ? 4 : -1;
}
__device__ __forceinline__ bool consumes_read(int op) {
    return op == OP_M || op == OP_I || op == OP_S || op == OP_EQ || op == OP_X;
}
__device__ __forceinline__ bool consumes_ref(int op) {
    return op == OP_M || op == OP_D || op == OP_N || op == OP_EQ || op == OP_X;
}
__device__ __forceinline__ int wave_index() { return (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6); }
__device__ __forceinline__ int wave_count() { return (int)((gridDim.x * blockDim.x) >> 6); }

// ------------------------------------------------------------------------------------------------------
// K3: feature extraction.  WRITE=false: counts[r] = number of features of read r.  WRITE=true: emit at off[r].
// A wave takes tiles of 64 consecutive reads: every lane loads the metadata of one read of the tile (coalesced, one
// memory round trip for the tile), the reads are then walked one by one with their metadata broadcast from the
// owning lane, and a match run is swept 256 bases at a time with all its loads issued before the first use.
// ------------------------------------------------------------------------------------------------------
#ifdef HIP_SIM
template <class T>
__device__ __forceinline__ T lane_bcast(T v, int k) { return __shfl(v, k); }
#else
__device__ __forceinline__ int32_t lane_bcast(int32_t v, int k) { return __builtin_amdgcn_readlane(v, k); }
__device__ __forceinline__ uint32_t lane_bcast(uint32_t v, int k) { return (uint32_t)__builtin_amdgcn_readlane((int)v, k); }
__device__ __forceinline__ int64_t lane_bcast(int64_t v, int k) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, k);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), k);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
#endif

template <bool WRITE>
__global__ __launch_bounds__(256) void extract_kernel(cramgpu_reads R, const uint8_t* ref, cramgpu_features F,
                                                      int64_t* sub_hist, int32_t* counts, int tile_reads,
                                                      const int64_t* total) {
    // write pass of a stream-ordered caller: the host has not seen the count, so the capacity check happens here
    if (WRITE && total && *total > F.cap) return;
    // tile_reads = 64: the tiled walk described above; tile_reads = 1: a wave per read (long reads: a tile of 64
    // ten-kilobase reads would keep one wave busy for 640 000 bases while most of the GPU idles)
    const int lane = threadIdx.x & 63;
    const int n_waves = wave_count();
    const int n_tiles = (R.n_reads + tile_reads - 1) / tile_reads;
    for (int tile = wave_index(); tile < n_tiles; tile += n_waves) {
        const int r0 = tile * tile_reads;
        const int rm = r0 + lane;
        const bool valid = lane < tile_reads && rm < R.n_reads;
        // one read's metadata per lane
        int32_t m_fl = CRAMGPU_RF_SKIP, m_start = 0, m_nc = 0, m_cont = 0;
        int64_t m_boff = 0, m_coff = 0, m_off = 0;
        uint32_t m_ce0 = 0;
        if (valid) {
            m_fl = R.rflags[rm];
            m_start = R.aln_start[rm];
            m_nc = R.n_cigar[rm];
            m_boff = R.base_off[rm];
            m_coff = R.cigar_off[rm];
            if (R.read_cont) m_cont = R.read_cont[rm];
            if (WRITE) m_off = F.off[rm];
            if (m_nc > 0) m_ce0 = R.cigar[m_coff];
        }
        int32_t nf_mine = 0;
        const int in_tile = min(tile_reads, R.n_reads - r0);
        cramgpu_refwin W{0, 0, 0};
        int wcont = -1;
        for (int k = 0; k < in_tile; k++) {
            const int32_t fl = lane_bcast(m_fl, k);
            const int32_t start = lane_bcast(m_start, k);
            const bool mapped = !(fl & (CRAMGPU_RF_UNMAPPED | CRAMGPU_RF_SKIP)) && start != 0;
            const int cont = lane_bcast(m_cont, k);
            if (mapped && cont != wcont) {
                W = R.win[cont];
                wcont = cont;
            }
            int64_t nf = 0;
            uint32_t hcnt = 0;  // substitution counts of this read, pair k in lane k: one atomic per pair per READ
                                // (per sweep, 6 M atomics on the 50 counters of two long-read containers took 20 ms)
            if (mapped) {
                const int nc = lane_bcast(m_nc, k);
                const int64_t o = WRITE ? lane_bcast(m_off, k) : 0;
                const uint8_t* wref = ref + W.ref_off;
                const uint8_t* rb = R.bases + lane_bcast(m_boff, k);
                const bool nobases = (fl & CRAMGPU_RF_NO_BASES) != 0;
                const uint32_t* cg = R.cigar + lane_bcast(m_coff, k);
                const uint32_t ce0 = lane_bcast(m_ce0, k);
                int pos = 0, refoff = 0;
                for (int e = 0; e < nc; e++) {
                    const uint32_t ce = e == 0 ? ce0 : cg[e];
                    const int op = (int)(ce & 15u), len = (int)(ce >> 4);
                    if (op == OP_D || op == OP_N || op == OP_P || op == OP_H || op == OP_S) {
                        if (WRITE && lane == 0) {
                            const int64_t i = o + nf;
                            F.op[i] = op == OP_D ? 'D' : op == OP_N ? 'N' : op == OP_P ? 'P' : op == OP_H ? 'H' : 'S';
                            F.pos[i] = pos + 1;
                            F.a[i] = op == OP_S ? pos : len;
                            F.b[i] = op == OP_S ? len : 0;
                        }
                        nf += 1;
                    } else if (op == OP_I) {
                        if (WRITE) {
                            for (int i = lane; i < len; i += 64) {
                                const int64_t x = o + nf + i;
                                F.op[x] = 'i';
                                F.pos[x] = pos + 1 + i;
                                F.a[x] = nobases ? 'N' : rb[pos + i];
                                F.b[x] = 0;
                            }
                        }
                        nf += len;
                    } else if (op == OP_M || op == OP_EQ || op == OP_X) {
                        const int64_t rc0 = (int64_t)start + refoff - 1 - W.ref_start0;  // window index of the run's base 0
                        for (int i0 = 0; i0 < len; i0 += 256) {
                            uint8_t bb[4], rr[4];
#pragma unroll
                            for (int c = 0; c < 4; c++) {  // all loads of the sweep before the first use
                                const int i = i0 + 64 * c + lane;
                                bb[c] = 'N';
                                rr[c] = 'N';
                                if (i < len) {
                                    const int64_t rc = rc0 + i;
                                    if (rc >= 0 && rc < W.ref_len) rr[c] = wref[rc];
                                    if (!nobases) bb[c] = rb[pos + i];
                                }
                            }
#pragma unroll
                            for (int c = 0; c < 4; c++) {
                                const int i = i0 + 64 * c + lane;
                                if (i0 + 64 * c >= len) break;
                                const uint8_t refb = normalize_base(rr[c]);
                                const uint8_t b = bb[c];
                                const bool mis = i < len && b != refb;
                                const unsigned long long m = __ballot(mis);
                                if (WRITE && m) {
                                    int key = -1;
                                    if (mis) {
                                        const int64_t x = o + nf + mbcnt(m);
                                        F.op[x] = 'X';
                                        F.pos[x] = pos + i + 1;
                                        F.a[x] = (int32_t)b | ((int32_t)refb << 8) | (0xFF << 16);
                                        F.b[x] = 0;
                                        const int bi = acgtn_index(b);
                                        if (bi >= 0) key = acgtn_index(refb) * 5 + bi;
                                    }
                                    if (sub_hist) {
                                        // pair k is tallied in lane k (flushed once per read, below)
                                        unsigned long long todo = __ballot(key >= 0);
                                        while (todo) {
                                            const int leader = __ffsll((long long)todo) - 1;
                                            const int k0 = __shfl(key, leader);
                                            const unsigned long long same = __ballot(key == k0);
                                            if (lane == k0) hcnt += (uint32_t)__popcll(same);  // lane k counts pair k
                                            todo &= ~same;
                                        }
                                    }
                                }
                                nf += __popcll(m);
                            }
                        }
                    }
                    if (consumes_read(op)) pos += len;
                    if (consumes_ref(op)) refoff += len;
                }
            }
            if (WRITE && sub_hist && hcnt)
                atomicAdd((unsigned long long*)&sub_hist[(int64_t)cont * 25 + lane], (unsigned long long)hcnt);
            if (lane == k) nf_mine = (int32_t)nf;
        }
        if (!WRITE && valid) counts[rm] = nf_mine;
    }
}

// ------------------------------------------------------------------------------------------------------
// K3 for short reads (Illumina-like, max_read_len <= kLaneReadMax): one LANE per read.  A match run is compared 16
// bases at a time (one unaligned 16-byte load of the read, one of the reference window); build/Utils#normalizeBase
// runs on four bases at once: h = (base >> 1) & 7 sends A C G T N (either case) to distinct slots of an 8-entry byte
// table held in two registers (v_perm_b32), and a base whose table entry differs from it in more than the case bit
// becomes 'N'.  A lane emits its own read's features in order, so nothing has to be compacted across lanes; the
// wave-per-tile kernel above spends ~100 wave instructions per read, this one about 8.
// ------------------------------------------------------------------------------------------------------
constexpr int kLaneReadMax = 1024;

struct __attribute__((packed)) PackedU32x4 { uint32_t v[4]; };

// byte c of the result = byte sel[c] (0..7) of {hi:lo}
__device__ __forceinline__ uint32_t perm8(uint32_t hi, uint32_t lo, uint32_t sel) {
#ifdef HIP_SIM
    const unsigned long long pool = ((unsigned long long)hi << 32) | lo;
    uint32_t r = 0;
    for (int c = 0; c < 4; c++) r |= (uint32_t)((pool >> (8 * ((sel >> (8 * c)) & 7))) & 255u) << (8 * c);
    return r;
#else
    return __builtin_amdgcn_perm(hi, lo, sel);
#endif
}
// normalizeBase on four bases
__device__ __forceinline__ uint32_t normalize4(uint32_t x) {
    const uint32_t y = perm8(0x4E000000u, 0x47544341u, (x >> 1) & 0x07070707u);  // slots: A C T G - - - N
    const uint32_t d = (y ^ x) & 0xDFDFDFDFu;                                     // nonzero byte: not that letter
    const uint32_t nz = (((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d) & 0x80808080u;
    const uint32_t m = (nz >> 7) * 255u;
    return (y & ~m) | (0x4E4E4E4Eu & m);
}

// the features of read r (WRITE: emitted at F.off[r]); returns their number
template <bool WRITE>
__device__ __forceinline__ int64_t lane_read_features(const cramgpu_reads& R, const uint8_t* ref, const cramgpu_features& F,
                                                      int64_t* sub_hist, int r) {
    {
        const int32_t fl = R.rflags[r];
        const int32_t start = R.aln_start[r];
        const bool mapped = !(fl & (CRAMGPU_RF_UNMAPPED | CRAMGPU_RF_SKIP)) && start != 0;
        int64_t nf = 0;
        if (mapped) {
            const int cont = R.read_cont ? R.read_cont[r] : 0;
            const cramgpu_refwin W = R.win[cont];
            const int64_t o = WRITE ? F.off[r] : 0;
            const uint8_t* wref = ref + W.ref_off;
            const int64_t boff = R.base_off[r];
            const uint8_t* rb = R.bases + boff;
            const bool nobases = (fl & CRAMGPU_RF_NO_BASES) != 0;
            const uint32_t* cg = R.cigar + R.cigar_off[r];
            const int nc = R.n_cigar[r];
            int pos = 0, refoff = 0;
            for (int e = 0; e < nc; e++) {
                const uint32_t ce = cg[e];
                const int op = (int)(ce & 15u), len = (int)(ce >> 4);
                if (op == OP_D || op == OP_N || op == OP_P || op == OP_H || op == OP_S) {
                    if (WRITE) {
                        const int64_t i = o + nf;
                        F.op[i] = op == OP_D ? 'D' : op == OP_N ? 'N' : op == OP_P ? 'P' : op == OP_H ? 'H' : 'S';
                        F.pos[i] = pos + 1;
                        F.a[i] = op == OP_S ? pos : len;
                        F.b[i] = op == OP_S ? len : 0;
                    }
                    nf += 1;
                } else if (op == OP_I) {
                    if (WRITE) {
                        for (int i = 0; i < len; i++) {
                            const int64_t x = o + nf + i;
                            F.op[x] = 'i';
                            F.pos[x] = pos + 1 + i;
                            F.a[x] = nobases ? 'N' : rb[pos + i];
                            F.b[x] = 0;
                        }
                    }
                    nf += len;
                } else if (op == OP_M || op == OP_EQ || op == OP_X) {
                    const int64_t rc0 = (int64_t)start + refoff - 1 - W.ref_start0;  // window index of the run's base 0
                    for (int i0 = 0; i0 < len; i0 += 16) {
                        const int n = len - i0 < 16 ? len - i0 : 16;
                        const int64_t rc = rc0 + i0;
                        uint32_t b[4], rr[4];
                        if (nobases) {
                            b[0] = b[1] = b[2] = b[3] = 0x4E4E4E4Eu;
                        } else if (boff + pos + i0 + 16 <= R.n_bases) {  // a whole unit lies inside the base arena
                            const PackedU32x4 t = *reinterpret_cast<const PackedU32x4*>(rb + pos + i0);
                            b[0] = t.v[0], b[1] = t.v[1], b[2] = t.v[2], b[3] = t.v[3];
                        } else {
                            b[0] = b[1] = b[2] = b[3] = 0;
                            for (int j = 0; j < n; j++) b[j >> 2] |= (uint32_t)rb[pos + i0 + j] << (8 * (j & 3));
                        }
                        if (rc >= 0 && rc + 16 <= W.ref_len) {
                            const PackedU32x4 t = *reinterpret_cast<const PackedU32x4*>(wref + rc);
                            rr[0] = t.v[0], rr[1] = t.v[1], rr[2] = t.v[2], rr[3] = t.v[3];
                        } else {  // the run leaves the window: bases outside it read as 'N'
                            rr[0] = rr[1] = rr[2] = rr[3] = 0;
                            for (int j = 0; j < 16; j++) {
                                const int64_t q = rc + j;
                                const uint32_t v = (j < n && q >= 0 && q < W.ref_len) ? wref[q] : (uint32_t)'N';
                                rr[j >> 2] |= v << (8 * (j & 3));
                            }
                        }
                        uint32_t any = 0, diff[4];
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            rr[c] = normalize4(rr[c]);
                            diff[c] = b[c] ^ rr[c];
                            const int left = n - 4 * c;  // bases of this word inside the run
                            if (left < 4) diff[c] &= left <= 0 ? 0u : (1u << (8 * left)) - 1u;
                            any |= diff[c];
                        }
                        if (any == 0) continue;
                        for (int j = 0; j < n; j++) {
                            if (((diff[j >> 2] >> (8 * (j & 3))) & 255u) == 0) continue;
                            const uint8_t bj = (uint8_t)(b[j >> 2] >> (8 * (j & 3)));
                            const uint8_t refb = (uint8_t)(rr[j >> 2] >> (8 * (j & 3)));
                            if (WRITE) {
                                const int64_t x = o + nf;
                                F.op[x] = 'X';
                                F.pos[x] = pos + i0 + j + 1;
                                F.a[x] = (int32_t)bj | ((int32_t)refb << 8) | (0xFF << 16);
                                F.b[x] = 0;
                                const int bi = acgtn_index(bj);
                                if (sub_hist && bi >= 0)
                                    atomicAdd((unsigned long long*)&sub_hist[(int64_t)cont * 25 + acgtn_index(refb) * 5 + bi], 1ull);
                            }
                            nf += 1;
                        }
                    }
                }
                if (consumes_read(op)) pos += len;
                if (consumes_ref(op)) refoff += len;
            }
        }
        return nf;
    }
}

// Count pass: counts[r] for every read, and the reads that have features collected in `list` (any order: every read
// writes at its own offset), so that the write pass spends its lanes only on those (14 % of Illumina-like reads).
__global__ __launch_bounds__(256) void extract_lane_count_kernel(cramgpu_reads R, const uint8_t* ref, cramgpu_features F,
                                                                 int32_t* counts, int32_t* list, uint32_t* list_n) {
    // the workgroup collects its reads in LDS and claims list space with one global atomic per flush (156 000 wave
    // atomics on one address cost 0.7 ms: the L2 serialises them)
    constexpr uint32_t kCap = 2048;
    __shared__ int32_t sl[kCap];
    __shared__ uint32_t sn, sbase;
    const int stride = (int)(gridDim.x * blockDim.x);
    const int lane = threadIdx.x & 63;
    if (threadIdx.x == 0) sn = 0;
    __syncthreads();
    auto flush = [&]() {  // called by the whole workgroup
        if (threadIdx.x == 0) sbase = sn ? atomicAdd(list_n, sn) : 0u;
        __syncthreads();
        const uint32_t cnt = sn, base = sbase;
        for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) list[base + i] = sl[i];
        __syncthreads();
        if (threadIdx.x == 0) sn = 0;
        __syncthreads();
    };
    for (int r0 = (int)(blockIdx.x * blockDim.x); r0 < R.n_reads; r0 += stride) {  // uniform per workgroup
        const int r = r0 + (int)threadIdx.x;
        int64_t nf = 0;
        if (r < R.n_reads) {
            nf = lane_read_features<false>(R, ref, F, nullptr, r);
            counts[r] = (int32_t)nf;
        }
        const unsigned long long m = __ballot(nf > 0);
        if (m != 0) {
            const int leader = __ffsll((long long)m) - 1;
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(&sn, (uint32_t)__popcll(m));
            base = __shfl(base, leader);
            if (nf > 0) sl[base + mbcnt(m)] = r;
        }
        __syncthreads();
        if (sn > kCap - 256) flush();
    }
    flush();
}

__global__ __launch_bounds__(256) void extract_lane_write_kernel(cramgpu_reads R, const uint8_t* ref, cramgpu_features F,
                                                                 int64_t* sub_hist, const int32_t* list,
                                                                 const uint32_t* list_n, const int64_t* total) {
    if (total && *total > F.cap) return;  // see extract_kernel
    const uint32_t n = *list_n;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        lane_read_features<true>(R, ref, F, sub_hist, list[i]);
}

// Exclusive scan of int32 counts into int64 offsets (off[n] = total) in three launches: per-tile sums, scan of the
// tile sums by one workgroup, per-tile scan with the tile's base.  A tile is 256 threads x 16 consecutive counts.
constexpr int kScanThreads = 256, kScanPer = 16, kScanTile = kScanThreads * kScanPer;

__device__ __forceinline__ int64_t block_excl_scan_256(int64_t v, int64_t* sh, int64_t& total) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    int64_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int64_t y = __shfl_up(x, o);
        if (lane >= o) x += y;
    }
    if (lane == 63) sh[w] = x;
    __syncthreads();
    int64_t base = 0;
    for (int k = 0; k < w; k++) base += sh[k];
    total = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return base + x - v;
}

__global__ __launch_bounds__(kScanThreads) void scan_tile_sums_kernel(const int32_t* counts, int32_t n, int64_t* tile_sum) {
    __shared__ int64_t sh[4];
    const int64_t lo = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanPer;
    int64_t s = 0;
    if (lo + kScanPer <= n) {
        const uint4* p = reinterpret_cast<const uint4*>(counts + lo);  // counts are >= 0
#pragma unroll
        for (int k = 0; k < kScanPer / 4; k++) {
            const uint4 v = p[k];
            s += (int64_t)v.x + v.y + v.z + v.w;
        }
    } else {
        for (int64_t i = lo; i < n && i < lo + kScanPer; i++) s += counts[i];
    }
    int64_t total;
    (void)block_excl_scan_256(s, sh, total);
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = total;
}

// in-place exclusive scan of the tile sums by one workgroup; tile_sum[n_tiles] = grand total
__global__ __launch_bounds__(kScanThreads) void scan_tiles_kernel(int64_t* tile_sum, int32_t n_tiles, int64_t* total_out) {
    __shared__ int64_t sh[4];
    int64_t carry = 0;
    for (int32_t b0 = 0; b0 < n_tiles; b0 += kScanThreads) {
        const int32_t i = b0 + (int32_t)threadIdx.x;
        const int64_t v = i < n_tiles ? tile_sum[i] : 0;
        int64_t total;
        const int64_t e = block_excl_scan_256(v, sh, total);
        if (i < n_tiles) tile_sum[i] = carry + e;
        carry += total;
    }
    if (threadIdx.x == 0) {
        tile_sum[n_tiles] = carry;
        if (total_out) *total_out = carry;
    }
}

__global__ __launch_bounds__(kScanThreads) void scan_apply_kernel(const int32_t* counts, int32_t n, const int64_t* tile_base,
                                                                  int32_t n_tiles, int64_t* off) {
    __shared__ int64_t sh[4];
    const int64_t lo = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanPer;
    int32_t c[kScanPer];
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < kScanPer; k++) {
        c[k] = lo + k < n ? counts[lo + k] : 0;
        s += c[k];
    }
    int64_t total;
    int64_t base = tile_base[blockIdx.x] + block_excl_scan_256(s, sh, total);
#pragma unroll
    for (int k = 0; k < kScanPer; k++) {
        if (lo + k < n) off[lo + k] = base;
        base += c[k];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) off[n] = tile_base[n_tiles];
}

// htsjdk!encoding/writer/Writer#write L139-261 -> SubstitutionMatrix#code: codes[cont][ref 5][base 256]
__global__ void assign_codes_kernel(int32_t n_reads, const int32_t* read_cont, const int64_t* off, const uint8_t* op,
                                    int32_t* a, const uint8_t* codes) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const uint8_t* cd = codes + (size_t)(read_cont ? read_cont[r] : 0) * 5 * 256;
    for (int64_t f = off[r]; f < off[r + 1]; f++) {
        if (op[f] != 'X') continue;
        const int32_t v = a[f];
        const int ri = acgtn_index((uint8_t)(v >> 8));
        const int code = ri >= 0 ? cd[ri * 256 + (v & 0xff)] : 0;
        a[f] = (v & 0xffff) | (code << 16);
    }
}

// F3 on the device: SubstitutionMatrix#<init>([[J) L38-57 / #rank L141-169 for every container of a batch, a thread
// per container.  mat5 = the 5 header bytes, codes25[cont][ref 5][base 5] = SubstitutionMatrix#code over ACGTN.
__global__ void matrices_kernel(const int64_t* hist, int32_t n_cont, uint8_t* mat5, uint8_t* codes25) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cont) return;
    const int64_t* h = hist + (int64_t)c * 25;
    for (int r = 0; r < 5; r++) {
        int others[4], k = 0;
        for (int b = 0; b < 5; b++)
            if (b != r) others[k++] = b;
        // rank = number of bases that sort before: larger count, or equal count and earlier in ACGTN order (stable)
        uint32_t byte = 0;
        for (int i = 0; i < 4; i++) {
            int rank = 0;
            for (int j2 = 0; j2 < 4; j2++) {
                const int64_t a = h[r * 5 + others[j2]], b2 = h[r * 5 + others[i]];
                if (a > b2 || (a == b2 && j2 < i)) rank++;
            }
            byte |= (uint32_t)rank << (6 - 2 * i);
            codes25[(int64_t)c * 25 + r * 5 + others[i]] = (uint8_t)rank;
        }
        codes25[(int64_t)c * 25 + r * 5 + r] = 0;
        mat5[(int64_t)c * 5 + r] = (uint8_t)byte;
    }
}

// Writer#write L139-261 -> SubstitutionMatrix#code with the device tables of matrices_kernel
__global__ void assign_codes25_kernel(int32_t n_reads, const int32_t* read_cont, const int64_t* off, const uint8_t* op,
                                      int32_t* a, const uint8_t* codes25, const int64_t* total, int64_t cap) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    if (total && *total > cap) return;  // the features did not fit and were never written (see extract_kernel): off[] is the full scan
    const uint8_t* cd = codes25 + (size_t)(read_cont ? read_cont[r] : 0) * 25;
    for (int64_t f = off[r]; f < off[r + 1]; f++) {
        if (op[f] != 'X') continue;
        const int32_t v = a[f];
        const int ri = acgtn_index((uint8_t)(v >> 8)), bi = acgtn_index((uint8_t)v);
        const int code = (ri >= 0 && bi >= 0) ? cd[ri * 5 + bi] : 0;
        a[f] = (v & 0xffff) | (code << 16);
    }
}

// ------------------------------------------------------------------------------------------------------
// K4 / N1: CramNormalizer.restoreReadBases.  basesof[cont][ref 5][code 4].
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void restore_bases_kernel(int32_t n_reads, const int32_t* read_len,
                                                            const int32_t* aln_start, const uint8_t* rflags,
                                                            const int32_t* read_cont, const cramgpu_refwin* wins,
                                                            cramgpu_features F, const uint8_t* seq, const int64_t* seq_off,
                                                            int64_t seq_bytes, const uint8_t* ref, const uint8_t* basesof, uint8_t* out,
                                                            const int64_t* out_off, int32_t* status) {
    const int lane = threadIdx.x & 63;
    const int n_waves = wave_count();
    const int n_tiles = (n_reads + 63) >> 6;
    for (int tile = wave_index(); tile < n_tiles; tile += n_waves) {
      // short reads belong to the lane-per-read kernel; this one walks the long reads of the tile
      const int rm = (tile << 6) + lane;
      unsigned long long todo = __ballot(rm < n_reads && read_len[rm] > kLaneReadMax);
      for (; todo; todo &= todo - 1) {
        const int r = (tile << 6) + __ffsll((long long)todo) - 1;
        const uint8_t fl = rflags[r];
        const int L = read_len[r];
        int st = 0;
        if (fl & CRAMGPU_RF_SKIP) {
            st = 2;
        } else if ((fl & CRAMGPU_RF_UNKNOWN_BASES) || L == 0) {
            st = 1;  // :336-337 NULL_SEQUENCE
        } else {
            const int cont = read_cont ? read_cont[r] : 0;
            const cramgpu_refwin W = wins[cont];
            const uint8_t* win = ref + W.ref_off;
            const int64_t wlen = W.ref_len, ws0 = W.ref_start0;
            const uint8_t* bo = basesof + (size_t)cont * 20;
            uint8_t* b = out + out_off[r];
            const int64_t al0 = (int64_t)aln_start[r] - 1;
            const int64_t fb = F.off[r], fe = F.off[r + 1];
            if (fb == fe) {
                // :343-364 no features: plain copy, 'N' beyond the window, no normalisation
                if (wlen + ws0 < al0 + L) {
                    const int64_t off = al0 - ws0, avail = wlen - off;
                    const int64_t ncopy = (off > 0 && avail > 0) ? (L < avail ? L : avail) : 0;
                    for (int i = lane; i < L; i += 64) b[i] = i < ncopy ? win[off + i] : (uint8_t)'N';
                } else if (al0 - ws0 < 0) {
                    st = CRAMGPU_E_JAVA_THROWS;
                } else {
                    const uint8_t* s = win + (al0 - ws0);
                    for (int i = lane; i < L; i += 64) b[i] = s[i];
                }
            } else {
                int pir = 1;      // posInRead (1-based)
                int64_t pis = 0;  // posInSeq
                bool bad = false;
                for (int64_t f = fb; f < fe && !bad; f++) {
                    const int fpos = F.pos[f];
                    const int opc = F.op[f];
                    const int32_t fa = F.a[f], fbv = F.b[f];
                    const int gap = fpos - pir;
                    if (gap > 0) {  // :368-370 copy reference up to the feature
                        if (al0 + pis - ws0 < 0 || pir - 1 + gap > L) {
                            bad = true;
                            break;
                        }
                        for (int i = lane; i < gap; i += 64) {
                            const int64_t rp = al0 + pis + i - ws0;
                            b[pir - 1 + i] = normalize_base(rp >= wlen ? (uint8_t)'N' : win[rp]);
                        }
                        pir += gap;
                        pis += gap;
                    }
                    if (opc == 'X') {
                        const int64_t rp = al0 + pis - ws0;
                        if (rp < 0 || pir - 1 >= L) {
                            bad = true;
                            break;
                        }
                        if (lane == 0) {
                            const uint8_t rbase = normalize_base(rp >= wlen ? (uint8_t)'N' : win[rp]);
                            b[pir - 1] = normalize_base(bo[acgtn_index(rbase) * 4 + ((fa >> 16) & 3)]);
                        }
                        pir++;
                        pis++;
                    } else if (opc == 'I' || opc == 'S') {
                        // the sequence bytes must lie inside the seq arena (Java: the feature owns its byte[])
                        const int64_t so = seq_off[r] + fa;
                        if (fbv < 0 || pir - 1 + fbv > L || fa < 0 || so < 0 || so + fbv > seq_bytes) {
                            bad = true;
                            break;
                        }
                        const uint8_t* s = seq + so;
                        for (int i = lane; i < fbv; i += 64) b[pir - 1 + i] = normalize_base(s[i]);
                        pir += fbv;
                    } else if (opc == 'D' || opc == 'N') {
                        pis += fa;
                    } else if (opc == 'i') {
                        if (pir - 1 >= L) {
                            bad = true;
                            break;
                        }
                        if (lane == 0) b[pir - 1] = normalize_base((uint8_t)fa);
                        pir++;
                    }
                }
                if (!bad) {
                    // :412-415 tail from the reference while it lasts, then the never-written rest reads as 0 -> 'N'
                    const int64_t rp0 = al0 + pis - ws0;
                    int64_t ncopy = (int64_t)L - (pir - 1);
                    if (wlen - rp0 < ncopy) ncopy = wlen - rp0;
                    if (ncopy < 0) ncopy = 0;
                    if (ncopy > 0 && rp0 < 0) {
                        bad = true;
                    } else {
                        for (int i = lane; i < L - (pir - 1); i += 64)
                            b[pir - 1 + i] = i < ncopy ? normalize_base(win[rp0 + i]) : (uint8_t)'N';
                        // :418-427 ReadBase features overwrite
                        __threadfence_block();
                        __builtin_amdgcn_wave_barrier();
                        for (int64_t f = fb + lane; f < fe; f += 64) {
                            if (F.op[f] == 'B') {
                                const int p = F.pos[f];
                                if (p < 1 || p > L)
                                    bad = true;
                                else
                                    b[p - 1] = normalize_base((uint8_t)F.a[f]);
                            }
                        }
                        bad = __ballot(bad) != 0;
                    }
                }
                if (bad) st = CRAMGPU_E_JAVA_THROWS;
            }
        }
        if (lane == 0) status[r] = st;
      }
    }
}

// N2: CramNormalizer.restoreQualityScores :287-333
__global__ __launch_bounds__(256) void restore_quals_kernel(int32_t n_reads, const int32_t* read_len, const uint8_t* rflags,
                                                            cramgpu_features F, uint8_t default_qs, uint8_t* scores,
                                                            const int64_t* out_off, int32_t* status) {
    const int lane = threadIdx.x & 63;
    const int n_waves = wave_count();
    const int n_tiles = (n_reads + 63) >> 6;
    for (int tile = wave_index(); tile < n_tiles; tile += n_waves) {
      // short reads belong to the lane-per-read kernel; this one walks the long reads of the tile
      const int rm = (tile << 6) + lane;
      unsigned long long todo = __ballot(rm < n_reads && read_len[rm] > kLaneReadMax);
      for (; todo; todo &= todo - 1) {
        const int r = (tile << 6) + __ffsll((long long)todo) - 1;
        const uint8_t fl = rflags[r];
        const int L = read_len[r];
        int st;
        if (fl & CRAMGPU_RF_SKIP) {
            st = 2;
        } else {
            uint8_t* s = scores + out_off[r];
            if (!(fl & CRAMGPU_RF_FORCE_QS)) {
                for (int i = lane; i < L; i += 64) s[i] = default_qs;
                __threadfence_block();
                __builtin_amdgcn_wave_barrier();
                bool any = false, bad = false;
                for (int64_t f = F.off[r] + lane; f < F.off[r + 1]; f += 64) {
                    const int opc = F.op[f];
                    if (opc == 'Q' || opc == 'B') {
                        const int p = F.pos[f];
                        if (p < 1 || p > L) {  // :302 / :307 scores[pos - 1] throws ArrayIndexOutOfBounds
                            bad = true;
                        } else {
                            s[p - 1] = (uint8_t)(opc == 'Q' ? F.a[f] : F.b[f]);
                            any = true;
                        }
                    }
                }
                st = __ballot(bad) ? CRAMGPU_E_JAVA_THROWS : __ballot(any) ? 0 : 1;
            } else {
                int missing = 0;
                for (int i = lane; i < L; i += 64) {
                    if (s[i] == 0xFF) {
                        s[i] = default_qs;
                        missing++;
                    }
                }
                for (int o = 32; o > 0; o >>= 1) missing += __shfl_xor(missing, o);
                st = missing == L ? 1 : 0;
            }
        }
        if (lane == 0) status[r] = st;
      }
    }
}

// ------------------------------------------------------------------------------------------------------
// N1 / N2 for short reads (max read length <= kLaneReadMax): one LANE per read, 16 bases per step, the same
// statements as restore_bases_kernel / restore_quals_kernel above (which stay in charge of long reads).
// ------------------------------------------------------------------------------------------------------
// dst[0..cnt) = normalizeBase(window[rp + i]), 'N' at and beyond the window's end (getByteOrDefault :436-441).
// room = bytes of the read from dst on: a whole 16-byte unit is stored only inside the read.
__device__ __forceinline__ void copy_ref_norm(uint8_t* dst, const uint8_t* win, int64_t rp, int cnt, int64_t wlen, int room) {
    int i = 0;
    for (; i + 16 <= cnt && rp + i + 16 <= wlen; i += 16) {
        PackedU32x4 t = *reinterpret_cast<const PackedU32x4*>(win + rp + i);
#pragma unroll
        for (int c = 0; c < 4; c++) t.v[c] = normalize4(t.v[c]);
        *reinterpret_cast<PackedU32x4*>(dst + i) = t;
    }
    (void)room;
    for (; i < cnt; i++) dst[i] = normalize_base(rp + i >= wlen ? (uint8_t)'N' : win[rp + i]);
}

__global__ __launch_bounds__(256) void restore_bases_lane_kernel(int32_t n_reads, const int32_t* read_len,
                                                                 const int32_t* aln_start, const uint8_t* rflags,
                                                                 const int32_t* read_cont, const cramgpu_refwin* wins,
                                                                 cramgpu_features F, const uint8_t* seq,
                                                                 const int64_t* seq_off, int64_t seq_bytes,
                                                                 const uint8_t* ref, const uint8_t* basesof, uint8_t* out,
                                                                 const int64_t* out_off, int32_t* status) {
    const int stride = (int)(gridDim.x * blockDim.x);
    for (int r = (int)(blockIdx.x * blockDim.x + threadIdx.x); r < n_reads; r += stride) {
        const uint8_t fl = rflags[r];
        const int L = read_len[r];
        if (L > kLaneReadMax) continue;  // restore_bases_kernel
        int st = 0;
        if (fl & CRAMGPU_RF_SKIP) {
            st = 2;
        } else if ((fl & CRAMGPU_RF_UNKNOWN_BASES) || L == 0) {
            st = 1;  // :336-337 NULL_SEQUENCE
        } else {
            const int cont = read_cont ? read_cont[r] : 0;
            const cramgpu_refwin W = wins[cont];
            const uint8_t* win = ref + W.ref_off;
            const int64_t wlen = W.ref_len, ws0 = W.ref_start0;
            const uint8_t* bo = basesof + (size_t)cont * 20;
            uint8_t* b = out + out_off[r];
            const int64_t al0 = (int64_t)aln_start[r] - 1;
            const int64_t fb = F.off[r], fe = F.off[r + 1];
            if (fb == fe) {
                // :343-364 no features: plain copy, 'N' beyond the window, no normalisation
                if (wlen + ws0 < al0 + L) {
                    const int64_t off = al0 - ws0, avail = wlen - off;
                    const int64_t ncopy = (off > 0 && avail > 0) ? (L < avail ? L : avail) : 0;
                    for (int i = 0; i < L; i++) b[i] = i < ncopy ? win[off + i] : (uint8_t)'N';
                } else if (al0 - ws0 < 0) {
                    st = CRAMGPU_E_JAVA_THROWS;
                } else {
                    const uint8_t* s = win + (al0 - ws0);
                    int i = 0;
                    for (; i + 16 <= L; i += 16)
                        *reinterpret_cast<PackedU32x4*>(b + i) = *reinterpret_cast<const PackedU32x4*>(s + i);
                    for (; i < L; i++) b[i] = s[i];
                }
            } else {
                int pir = 1;      // posInRead (1-based)
                int64_t pis = 0;  // posInSeq
                bool bad = false;
                for (int64_t f = fb; f < fe && !bad; f++) {
                    const int fpos = F.pos[f];
                    const int opc = F.op[f];
                    const int32_t fa = F.a[f], fbv = F.b[f];
                    const int gap = fpos - pir;
                    if (gap > 0) {  // :368-370 copy reference up to the feature
                        if (al0 + pis - ws0 < 0 || pir - 1 + gap > L) {
                            bad = true;
                            break;
                        }
                        copy_ref_norm(b + pir - 1, win, al0 + pis - ws0, gap, wlen, L - (pir - 1));
                        pir += gap;
                        pis += gap;
                    }
                    if (opc == 'X') {
                        const int64_t rp = al0 + pis - ws0;
                        if (rp < 0 || pir - 1 >= L) {
                            bad = true;
                            break;
                        }
                        const uint8_t rbase = normalize_base(rp >= wlen ? (uint8_t)'N' : win[rp]);
                        b[pir - 1] = normalize_base(bo[acgtn_index(rbase) * 4 + ((fa >> 16) & 3)]);
                        pir++;
                        pis++;
                    } else if (opc == 'I' || opc == 'S') {
                        // the sequence bytes must lie inside the seq arena (Java: the feature owns its byte[])
                        const int64_t so = seq_off[r] + fa;
                        if (fbv < 0 || pir - 1 + fbv > L || fa < 0 || so < 0 || so + fbv > seq_bytes) {
                            bad = true;
                            break;
                        }
                        const uint8_t* s = seq + so;
                        for (int i = 0; i < fbv; i++) b[pir - 1 + i] = normalize_base(s[i]);
                        pir += fbv;
                    } else if (opc == 'D' || opc == 'N') {
                        pis += fa;
                    } else if (opc == 'i') {
                        if (pir - 1 >= L) {
                            bad = true;
                            break;
                        }
                        b[pir - 1] = normalize_base((uint8_t)fa);
                        pir++;
                    }
                }
                if (!bad) {
                    // :412-415 tail from the reference while it lasts, then the never-written rest reads as 0 -> 'N'
                    const int64_t rp0 = al0 + pis - ws0;
                    const int rest = L - (pir - 1);
                    int64_t ncopy = rest;
                    if (wlen - rp0 < ncopy) ncopy = wlen - rp0;
                    if (ncopy < 0) ncopy = 0;
                    if (ncopy > 0 && rp0 < 0) {
                        bad = true;
                    } else {
                        if (ncopy > 0) copy_ref_norm(b + pir - 1, win, rp0, (int)ncopy, wlen, rest);
                        for (int i = (int)ncopy; i < rest; i++) b[pir - 1 + i] = (uint8_t)'N';
                        // :418-427 ReadBase features overwrite
                        for (int64_t f = fb; f < fe; f++) {
                            if (F.op[f] == 'B') {
                                const int p = F.pos[f];
                                if (p < 1 || p > L)
                                    bad = true;
                                else
                                    b[p - 1] = normalize_base((uint8_t)F.a[f]);
                            }
                        }
                    }
                }
                if (bad) st = CRAMGPU_E_JAVA_THROWS;
            }
        }
        status[r] = st;
    }
}

__global__ __launch_bounds__(256) void restore_quals_lane_kernel(int32_t n_reads, const int32_t* read_len,
                                                                 const uint8_t* rflags, cramgpu_features F,
                                                                 uint8_t default_qs, uint8_t* scores,
                                                                 const int64_t* out_off, int32_t* status) {
    const int stride = (int)(gridDim.x * blockDim.x);
    const uint32_t dq4 = default_qs * 0x01010101u;
    const int lane = threadIdx.x & 63;
    for (int r0 = (int)(blockIdx.x * blockDim.x + (threadIdx.x & ~63u)); r0 < n_reads; r0 += stride) {  // uniform per wave
        const int r = r0 + lane;
        const bool valid = r < n_reads;
        const uint8_t fl = valid ? rflags[r] : (uint8_t)CRAMGPU_RF_SKIP;
        const int L = valid ? read_len[r] : 0;
        const int64_t off = valid ? out_off[r] : 0;
        // Usual case of a lossless container: 64 force-preserved reads whose score arrays lie back to back.  The wave
        // then scans the whole span with coalesced 16-byte loads (a lane walking its own read at a 150-byte stride
        // fetches every line four times) and falls back to the per-read code only if it meets a 0xFF hole.
        {
            const int64_t nxt = __shfl(off, (lane + 1) & 63);
            const bool plain = valid && (fl & (CRAMGPU_RF_SKIP | CRAMGPU_RF_FORCE_QS)) == CRAMGPU_RF_FORCE_QS && L > 0 &&
                               L <= kLaneReadMax && (lane == 63 || nxt == off + L);
            if (__ballot(plain) == ~0ull) {
                const int64_t lo = __shfl(off, 0), hi = __shfl(off + L, 63);
                bool hole = false;
                int64_t p = lo + lane * 16;
                for (; p + 16 <= hi; p += 1024) {
                    const PackedU32x4 t = *reinterpret_cast<const PackedU32x4*>(scores + p);
                    uint32_t all = 0x80808080u;
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const uint32_t inv = ~t.v[c];
                        all &= ((inv & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | inv;
                    }
                    hole |= (all & 0x80808080u) != 0x80808080u;
                }
                for (; p < hi; p++) hole |= scores[p] == 0xFF;  // the lane that owns the last, partial unit
                if (__ballot(hole) == 0) {
                    status[r] = 0;
                    continue;
                }
            }
        }
        if (!valid || L > kLaneReadMax) continue;  // long reads: restore_quals_kernel
        int st;
        if (fl & CRAMGPU_RF_SKIP) {
            st = 2;
        } else {
            uint8_t* s = scores + off;
            if (!(fl & CRAMGPU_RF_FORCE_QS)) {
                int i = 0;
                for (; i + 16 <= L; i += 16) *reinterpret_cast<PackedU32x4*>(s + i) = PackedU32x4{{dq4, dq4, dq4, dq4}};
                for (; i < L; i++) s[i] = default_qs;
                bool any = false, bad = false;
                for (int64_t f = F.off[r]; f < F.off[r + 1]; f++) {
                    const int opc = F.op[f];
                    if (opc == 'Q' || opc == 'B') {
                        const int p = F.pos[f];
                        if (p < 1 || p > L) {  // :302 / :307 scores[pos - 1] throws ArrayIndexOutOfBounds
                            bad = true;
                        } else {
                            s[p - 1] = (uint8_t)(opc == 'Q' ? F.a[f] : F.b[f]);
                            any = true;
                        }
                    }
                }
                st = bad ? CRAMGPU_E_JAVA_THROWS : any ? 0 : 1;
            } else {
                int missing = 0, i = 0;
                for (; i + 16 <= L; i += 16) {
                    const PackedU32x4 t = *reinterpret_cast<const PackedU32x4*>(s + i);
                    uint32_t all = 0x80808080u;
#pragma unroll
                    for (int c = 0; c < 4; c++) {  // high bit of a byte stays set iff the byte is not 0xFF
                        const uint32_t inv = ~t.v[c];
                        all &= ((inv & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | inv;
                    }
                    if ((all & 0x80808080u) == 0x80808080u) continue;
                    for (int j = 0; j < 16; j++)
                        if (s[i + j] == 0xFF) {
                            s[i + j] = default_qs;
                            missing++;
                        }
                }
                for (; i < L; i++)
                    if (s[i] == 0xFF) {
                        s[i] = default_qs;
                        missing++;
                    }
                st = missing == L ? 1 : 0;
            }
        }
        status[r] = st;
    }
}

// ------------------------------------------------------------------------------------------------------
// K5 / Q2: QualityScorePreservation.addQualityScores (one wave per read; mask in LDS, global scratch for long reads)
// ------------------------------------------------------------------------------------------------------
__constant__ uint8_t kIllumina8[110] = {  // lossy/Binning.java:63-71
    0,  1,  6,  6,  6,  6,  6,  6,  6,  6,  15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 22, 22, 22, 22, 22, 27, 27, 27,
    27, 27, 33, 33, 33, 33, 33, 37, 37, 37, 37, 37, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40,
    40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40,
    40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40};

constexpr int kMaskLds = 16384;

__device__ __forceinline__ bool read_gate(const cramgpu_policy& p, bool unmapped, int mapq) {
    switch (p.read_cat) {
        case 0: case 4: return true;
        case 1: return unmapped;
        case 3: return mapq < p.read_param;
        case 2: return mapq > p.read_param;
        default: return false;
    }
}

__global__ __launch_bounds__(64) void lossy_quals_kernel(cramgpu_reads R, const uint8_t* ref, const int32_t* wpos,
                                                         int32_t window, const cramgpu_policy* pol, int32_t n_pol,
                                                         uint8_t* scores, uint8_t* force, int32_t* status,
                                                         uint8_t* mask_ws, int32_t mask_lds, const int32_t* list,
                                                         const uint32_t* list_n) {
    // the mask of a read lives in LDS when it fits the launch's dynamic allocation (sized by the batch's longest read, at
    // most kMaskLds: short reads leave the LDS to many resident waves — a wave per read is bound by the chain of
    // dependent loads per read, so reads in flight are what counts)
    CRAMGPU_DYN_LDS(lmask);
    // The workgroup is ONE wave: its LDS accesses complete in program order, so phases that hand mask bytes from lane to
    // lane need no barrier — and above all no fence: __threadfence_block() + __syncthreads() also wait for every global
    // store of the scores (a memory round trip, ten times per read: 82 ms for the 10 M reads of configs[1]).  Only a
    // mask in global scratch (reads longer than the LDS mask) keeps the fence.
#define K5_SYNC()                     \
    do {                              \
        if (mask_in_lds) {            \
            CRAMGPU_LOCKSTEP();       \
        } else {                      \
            __threadfence_block();    \
            __syncthreads();          \
        }                             \
    } while (0)
    const int lane = threadIdx.x;
    // list != nullptr: only the reads the lane-per-read kernel left to this one
    const int n_items = list ? (int)*list_n : R.n_reads;
    for (int it = blockIdx.x; it < n_items; it += gridDim.x) {
        const int r = list ? list[it] : it;
        const uint8_t fl = R.rflags[r];
        int st = 0;
        int frc = 0;
        if (fl & CRAMGPU_RF_SKIP) {
            st = 2;
        } else if (fl & CRAMGPU_RF_NULL_QUALS) {
            st = 1;  // :277-281
        } else {
            const int L = R.read_len[r];
            const int64_t bo = R.base_off[r];
            const uint8_t* qs = R.quals + bo;
            const uint8_t* rb = R.bases + bo;
            uint8_t* sc = scores + bo;
            const bool mask_in_lds = L + 1 <= mask_lds;
            uint8_t* mask = mask_in_lds ? lmask : (mask_ws + bo + r);
            const int cont = R.read_cont ? R.read_cont[r] : 0;
            const cramgpu_refwin W = R.win[cont];
            const uint8_t* wref = ref + W.ref_off;
            const int64_t contig_len = W.ref_start0 + W.ref_len;
            const int64_t Wn = window < contig_len ? window : contig_len;
            const int32_t start = R.aln_start[r];
            const bool unmapped = (fl & CRAMGPU_RF_UNMAPPED) != 0;
            const int mapq = R.mapq[r];
            const int nc = R.n_cigar[r];
            const uint32_t* cg = R.cigar + R.cigar_off[r];
            const int64_t wp = wpos[r];
            bool bad = false;
            for (int i = lane; i < L; i += 64) sc[i] = 0xFF;
            K5_SYNC();
            for (int k = 0; k < n_pol && !bad; k++) {
                const cramgpu_policy p = pol[k];
                if (!read_gate(p, unmapped, mapq)) continue;
                if (p.n_base_cat == 0) {  // :362-388 whole-read treatment
                    if (p.treatment == 1) {
                        for (int i = lane; i < L; i += 64) {
                            const uint8_t q = qs[i];
                            if (q >= 110)
                                bad = true;
                            else
                                sc[i] = kIllumina8[q];
                        }
                        frc = 1;
                    } else if (p.treatment == 0) {
                        for (int i = lane; i < L; i += 64) sc[i] = qs[i];
                        frc = 1;
                    } else {
                        frc = 0;
                    }
                    bad = __ballot(bad) != 0;
                    continue;
                }
                if (start == 0) continue;  // :394-396
                for (int i = lane; i < L + 1; i += 64) mask[i] = 0;
                K5_SYNC();
                for (int c = 0; c < p.n_base_cat && !bad; c++) {
                    const int type = p.base_cat[c];
                    int pos = 0;
                    int64_t refpos = start;
                    if (type == 4) {  // PILEUP against all-zero tracks: mismatchesAt() == 0
                        if (0 > p.base_param[c])
                            for (int i = lane; i < L; i += 64) mask[i] = 1;
                    } else {
                        for (int e = 0; e < nc && !bad; e++) {
                            const uint32_t ce = cg[e];
                            const int op = (int)(ce & 15u), len = (int)(ce >> 4);
                            if (type == 3) {
                                if (op == OP_D) {
                                    if (pos >= L) {
                                        bad = true;
                                    } else if (lane == 0) {
                                        mask[pos] = 1;
                                        if (pos + 1 < L) mask[pos + 1] = 1;
                                    }
                                }
                            } else if (type == 1 || type == 2) {
                                if (consumes_read(op)) {
                                    if (pos + len > L || refpos - wp < 0) {
                                        bad = len > 0;
                                    } else {
                                        for (int i = lane; i < len; i += 64) {
                                            const int64_t rp = refpos + i;
                                            const int64_t gi = rp - 1 - W.ref_start0;
                                            const uint8_t tb = (rp - wp < Wn && gi >= 0 && gi < W.ref_len) ? wref[gi] : (uint8_t)'N';
                                            const bool match = rb[pos + i] == tb;
                                            mask[pos + i] = (type == 1 && match) || (type == 2 && !match);
                                        }
                                    }
                                }
                            } else if (type == 6) {
                                if (op == OP_I) {
                                    if (pos + len > L)
                                        bad = len > 0;
                                    else
                                        for (int i = lane; i < len; i += 64) mask[pos + i] = 1;
                                }
                            } else if (type == 5) {  // LOWER_COVERAGE against all-zero tracks: coverageAt() == 0
                                if (op == OP_M || op == OP_EQ || op == OP_X) {
                                    if (pos + len > L)
                                        bad = len > 0;
                                    else
                                        for (int i = lane; i < len; i += 64) mask[pos + i] = 0 < p.base_param[c];
                                }
                            }
                            if (consumes_read(op)) pos += len;
                            if (consumes_ref(op)) refpos += len;
                            if (type == 3) {  // lane 0's two writes must land before a later category overwrites them
                                K5_SYNC();
                            }
                        }
                    }
                    K5_SYNC();
                }
                if (bad) break;
                int masked = 0;
                for (int i = lane; i < L; i += 64) {
                    if (mask[i]) {
                        const uint8_t q = qs[i];
                        if (p.treatment == 1) {
                            if (q >= 110)
                                bad = true;
                            else
                                sc[i] = kIllumina8[q];
                        } else if (p.treatment == 2) {
                            sc[i] = 0xFF;
                        } else {
                            sc[i] = q;
                        }
                        masked++;
                    }
                }
                for (int o = 32; o > 0; o >>= 1) masked += __shfl_xor(masked, o);
                bad = __ballot(bad) != 0;
                if (masked > L / 2) frc = 1;  // :486-488 safety latch
                K5_SYNC();
            }
            if (bad) st = CRAMGPU_E_JAVA_THROWS;
        }
        if (lane == 0) {
            force[r] = (uint8_t)frc;
            status[r] = st;
        }
        CRAMGPU_LOCKSTEP();
    }
#undef K5_SYNC
}

// ------------------------------------------------------------------------------------------------------
// K5 for short reads: a lane per read.  A wave per 150-base read spends ~3000 wave instructions on it (strided loops
// of three turns, a CIGAR walk per category, lock-steps): 56 ms for the 10 M reads of configs[1].  Here a lane walks
// its read in units of 16 bases: the scores of a unit stay in four registers through every policy (whole-read
// treatments, then per category a 16-bit mask built from the CIGAR elements that overlap the unit; MATCH / MISMATCH
// compare 16 bases with one unaligned 16-byte load of the reference), the Illumina-8 table sits in LDS, and the unit
// leaves as one 16-byte store.  Reads the reference would throw on (a score >= 110 under binning, CIGARs that overrun
// the read, a window behind the read) and reads longer than kLaneReadMax are handed to the wave kernel through a list,
// which redoes them from scratch with the exact partial effects.
// ------------------------------------------------------------------------------------------------------
constexpr int kLanePolMax = 8;

__device__ __forceinline__ uint32_t eq_mask4(uint32_t a, uint32_t b) {  // bit j = byte j of a and b equal
    const uint32_t x = a ^ b;
    const uint32_t nz = (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
    const uint32_t z = ~nz & 0x80808080u;
    return ((z >> 7) & 1u) | ((z >> 14) & 2u) | ((z >> 21) & 4u) | ((z >> 28) & 8u);
}

__global__ __launch_bounds__(256) void lossy_quals_lane_kernel(cramgpu_reads R, const uint8_t* ref, const int32_t* wpos,
                                                               int32_t window, const cramgpu_policy* pol, int32_t n_pol,
                                                               uint8_t* scores, uint8_t* force, int32_t* status,
                                                               int32_t* defer, uint32_t* defer_n) {
    __shared__ cramgpu_policy sp[kLanePolMax];
    __shared__ uint8_t lut[128];
    __shared__ uint16_t s_masked[kLanePolMax][256];
    for (int i = threadIdx.x; i < n_pol * (int)(sizeof(cramgpu_policy) / 4); i += blockDim.x)
        reinterpret_cast<uint32_t*>(sp)[i] = reinterpret_cast<const uint32_t*>(pol)[i];
    if (threadIdx.x < 128) lut[threadIdx.x] = threadIdx.x < 110 ? kIllumina8[threadIdx.x] : (uint8_t)0;
    __syncthreads();
    const int tid = (int)threadIdx.x;
    const int stride = (int)(gridDim.x * blockDim.x);
    for (int r = (int)(blockIdx.x * blockDim.x) + tid; r < R.n_reads; r += stride) {
        const uint8_t fl = R.rflags[r];
        if (fl & CRAMGPU_RF_SKIP) {
            force[r] = 0;
            status[r] = 2;
            continue;
        }
        if (fl & CRAMGPU_RF_NULL_QUALS) {  // :277-281
            force[r] = 0;
            status[r] = 1;
            continue;
        }
        const int L = R.read_len[r];
        const int64_t bo = R.base_off[r];
        const int32_t start = R.aln_start[r];
        const int nc = R.n_cigar[r];
        const uint32_t* cg = R.cigar + R.cigar_off[r];
        const int64_t wp = wpos[r];
        // reads this kernel does not decide: too long, or any condition under which the reference throws half way
        bool leave = L > kLaneReadMax || (start != 0 && (int64_t)start - wp < 0);
        {
            int pos = 0;
            for (int e = 0; e < nc && !leave; e++) {
                const uint32_t ce = cg[e];
                const int op = (int)(ce & 15u), len = (int)(ce >> 4);
                if (op == OP_D && pos >= L) leave = true;
                if (consumes_read(op)) {
                    if (pos + len > L) leave = true;
                    pos += len;
                }
            }
        }
        const uint8_t* qs = R.quals + bo;
        const uint8_t* rb = R.bases + bo;
        uint8_t* sc = scores + bo;
        const int cont = R.read_cont ? R.read_cont[r] : 0;
        const cramgpu_refwin W = R.win[cont];
        const uint8_t* wref = ref + W.ref_off;
        const int64_t contig_len = W.ref_start0 + W.ref_len;
        const int64_t Wn = window < contig_len ? window : contig_len;
        const bool unmapped = (fl & CRAMGPU_RF_UNMAPPED) != 0;
        const int mapq = R.mapq[r];
        for (int k = 0; k < n_pol; k++) s_masked[k][tid] = 0;
        for (int u0 = 0; u0 < L && !leave; u0 += 16) {
            const int nb = L - u0 < 16 ? L - u0 : 16;
            const uint32_t valid = nb == 16 ? 0xFFFFu : (1u << nb) - 1u;
            uint32_t q[4], b[4], o[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            if (bo + u0 + 16 <= R.n_bases) {
                const PackedU32x4 t = *reinterpret_cast<const PackedU32x4*>(qs + u0);
                q[0] = t.v[0], q[1] = t.v[1], q[2] = t.v[2], q[3] = t.v[3];
                const PackedU32x4 tb = *reinterpret_cast<const PackedU32x4*>(rb + u0);
                b[0] = tb.v[0], b[1] = tb.v[1], b[2] = tb.v[2], b[3] = tb.v[3];
            } else {
                q[0] = q[1] = q[2] = q[3] = b[0] = b[1] = b[2] = b[3] = 0;
                for (int j = 0; j < nb; j++) {
                    q[j >> 2] |= (uint32_t)qs[u0 + j] << (8 * (j & 3));
                    b[j >> 2] |= (uint32_t)rb[u0 + j] << (8 * (j & 3));
                }
            }
            // a score >= 110 throws under binning (Binning.java:63-71): such reads go to the wave kernel
            uint32_t big = 0;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const uint32_t h = (((q[c] & 0x7F7F7F7Fu) + 0x12121212u) | q[c]) & 0x80808080u;
                const int left = nb - 4 * c;
                big |= left >= 4 ? h : left <= 0 ? 0u : h & ((1u << (8 * left)) - 1u);
            }
            if (big) {
                leave = true;
                break;
            }
            for (int k = 0; k < n_pol; k++) {
                const cramgpu_policy& p = sp[k];
                if (!read_gate(p, unmapped, mapq)) continue;
                uint32_t m16;
                if (p.n_base_cat == 0) {  // :362-388 whole-read treatment
                    if (p.treatment == 2) continue;
                    m16 = valid;
                } else {
                    if (start == 0) continue;  // :394-396
                    m16 = 0;
                    for (int c = 0; c < p.n_base_cat; c++) {
                        const int type = p.base_cat[c];
                        if (type == 4) {  // PILEUP against all-zero tracks
                            if (0 > p.base_param[c]) m16 = valid;
                            continue;
                        }
                        int pos = 0;
                        int64_t refpos = start;
                        for (int e = 0; e < nc; e++) {
                            const uint32_t ce = cg[e];
                            const int op = (int)(ce & 15u), len = (int)(ce >> 4);
                            const bool cr = consumes_read(op);
                            // positions of the element inside the unit: bits [lo, hi)
                            int lo = pos - u0, hi = pos + (cr ? len : 0) - u0;
                            lo = lo < 0 ? 0 : lo;
                            hi = hi > nb ? nb : hi;
                            const uint32_t span = hi > lo ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
                            if (type == 3) {
                                if (op == OP_D) {  // pos < L was checked above
                                    if (pos >= u0 && pos < u0 + nb) m16 |= 1u << (pos - u0);
                                    if (pos + 1 < L && pos + 1 >= u0 && pos + 1 < u0 + nb) m16 |= 1u << (pos + 1 - u0);
                                }
                            } else if (type == 1 || type == 2) {
                                if (cr && span) {
                                    // byte j of the unit faces reference position refpos + (u0 + j - pos)
                                    const int64_t rp0 = refpos + (u0 - pos);           // 1-based position facing byte 0
                                    const int64_t gi0 = rp0 - 1 - W.ref_start0;        // its window index
                                    uint32_t eq;
                                    if (gi0 + lo >= 0 && gi0 + hi <= W.ref_len && rp0 + hi - 1 - wp < Wn && gi0 >= 0 &&
                                        gi0 + 16 <= W.ref_len) {
                                        const PackedU32x4 t = *reinterpret_cast<const PackedU32x4*>(wref + gi0);
                                        eq = eq_mask4(b[0], t.v[0]) | (eq_mask4(b[1], t.v[1]) << 4) | (eq_mask4(b[2], t.v[2]) << 8) |
                                             (eq_mask4(b[3], t.v[3]) << 12);
                                    } else {
                                        eq = 0;
                                        for (int j = lo; j < hi; j++) {
                                            const int64_t rp = rp0 + j, gi = gi0 + j;
                                            const uint8_t tb = (rp - wp < Wn && gi >= 0 && gi < W.ref_len) ? wref[gi] : (uint8_t)'N';
                                            if ((uint8_t)(b[j >> 2] >> (8 * (j & 3))) == tb) eq |= 1u << j;
                                        }
                                    }
                                    const uint32_t val = type == 1 ? eq : ~eq;
                                    m16 = (m16 & ~span) | (val & span);
                                }
                            } else if (type == 6) {
                                if (op == OP_I) m16 |= span;
                            } else if (type == 5) {  // LOWER_COVERAGE against all-zero tracks
                                if (op == OP_M || op == OP_EQ || op == OP_X) m16 = 0 < p.base_param[c] ? (m16 | span) : (m16 & ~span);
                            }
                            if (cr) pos += len;
                            if (consumes_ref(op)) refpos += len;
                        }
                    }
                    s_masked[k][tid] += (uint16_t)__popc(m16 & valid);
                }
                m16 &= valid;
                if (m16 == 0) continue;
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    if (!((m16 >> j) & 1u)) continue;
                    const uint32_t qb = (q[j >> 2] >> (8 * (j & 3))) & 255u;
                    const uint32_t v = p.treatment == 1 ? lut[qb] : p.treatment == 2 ? 0xFFu : qb;
                    o[j >> 2] = (o[j >> 2] & ~(255u << (8 * (j & 3)))) | (v << (8 * (j & 3)));
                }
            }
            if (nb == 16) {
                PackedU32x4 t;
                t.v[0] = o[0], t.v[1] = o[1], t.v[2] = o[2], t.v[3] = o[3];
                *reinterpret_cast<PackedU32x4*>(sc + u0) = t;
            } else {
                for (int j = 0; j < nb; j++) sc[u0 + j] = (uint8_t)(o[j >> 2] >> (8 * (j & 3)));
            }
        }
        if (leave) {
            defer[atomicAdd(defer_n, 1u)] = r;
            continue;
        }
        int frc = 0;
        for (int k = 0; k < n_pol; k++) {
            const cramgpu_policy& p = sp[k];
            if (!read_gate(p, unmapped, mapq)) continue;
            if (p.n_base_cat == 0) {
                frc = p.treatment == 2 ? 0 : 1;
            } else if (start != 0 && (int)s_masked[k][tid] > L / 2) {
                frc = 1;  // :486-488 safety latch
            }
        }
        force[r] = (uint8_t)frc;
        status[r] = 0;
    }
}

// Tail of addQualityScores :288-299.  WRITE=false: counts[r] = merged feature count.  One thread per read.
template <bool WRITE>
__global__ void merge_quality_kernel(int32_t n_reads, const int32_t* read_len, const int64_t* base_off,
                                     const uint8_t* scores, const uint8_t* force, const int32_t* qstatus,
                                     cramgpu_features in, cramgpu_features out, int32_t* counts, const int64_t* total,
                                     const int64_t* in_total) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    if (in_total && *in_total > in.cap) {
        // the input set did not fit and was never written: in.off[] runs past its columns.  Count an upper bound (every
        // input feature plus a score per base where scores are added) without touching them: the total then exceeds the
        // capacity as well, the job reports CRAMGPU_E_INSUFFICIENT and a capacity of off[n] is enough for the retry
        if (!WRITE) counts[r] = (int32_t)(in.off[r + 1] - in.off[r]) + ((qstatus[r] == 0 && !force[r]) ? read_len[r] : 0);
        return;
    }
    if (WRITE && total && *total > out.cap) return;  // see extract_kernel
    int64_t f = in.off[r];
    const int64_t fe = in.off[r + 1];
    const bool addq = qstatus[r] == 0 && !force[r];
    const int L = read_len[r];
    const uint8_t* sc = scores + base_off[r];
    int64_t n = WRITE ? out.off[r] : 0;
    const int64_t n0 = n;
#define COPYF()                     \
    do {                            \
        if (WRITE) {                \
            out.op[n] = in.op[f];   \
            out.pos[n] = in.pos[f]; \
            out.a[n] = in.a[f];     \
            out.b[n] = in.b[f];     \
        }                           \
        n++;                        \
        f++;                        \
    } while (0)
    if (addq) {
        for (int i = 0; i < L; i++) {
            const uint8_t s = sc[i];
            if (s >= 128) continue;  // scores[i] > -1 as a signed byte
            while (f < fe && in.pos[f] <= i + 1) COPYF();
            if (WRITE) {
                out.op[n] = 'Q';
                out.pos[n] = i + 1;
                out.a[n] = s;
                out.b[n] = 0;
            }
            n++;
        }
    }
    while (f < fe) COPYF();
#undef COPYF
    if (!WRITE) counts[r] = (int32_t)(n - n0);
}

}  // namespace

// CU count of the device the launches go to (set by cramgpu_open from hipDeviceProp_t; every MI355X has 256)
static int g_n_cu = 256;
void records_set_cu_count(int n_cu) {
    if (n_cu > 0) g_n_cu = n_cu;
}

static int lane_grid(int64_t n_reads) {  // a lane per read, grid-stride beyond the resident workgroups
    static const int per_cu = [] {
        const char* e = getenv("CRAMGPU_LANE_WGS_PER_CU");
        const int v = e ? atoi(e) : 0;
        return v > 0 && v <= 64 ? v : 32;
    }();
    const int64_t blocks = (n_reads + 255) / 256, cap = g_n_cu * (int64_t)per_cu;
    return (int)(blocks < 1 ? 1 : blocks > cap ? cap : blocks);
}
static int grid_for_waves(int64_t n_items, int waves_per_block) {
    int64_t blocks = (n_items + waves_per_block - 1) / waves_per_block;
    if (blocks < 1) blocks = 1;
    if (blocks > g_n_cu * 16) blocks = g_n_cu * 16;
    return (int)blocks;
}

hipError_t launch_extract_features(const cramgpu_reads& r, const uint8_t* ref, cramgpu_features f, int64_t* sub_hist,
                                   int32_t* d_counts, bool write_pass, int32_t* list, uint32_t* list_n,
                                   const int64_t* d_total_guard, hipStream_t s) {
    if (r.max_read_len <= kLaneReadMax && list && list_n) {  // short reads: a lane per read
        if (write_pass) {
            // the list holds the reads with features: at most n_reads, usually a small fraction
            hipLaunchKernelGGL(extract_lane_write_kernel, dim3(lane_grid(((int64_t)r.n_reads + 3) / 4)), dim3(256), 0, s, r,
                               ref, f, sub_hist, list, list_n, d_total_guard);
        } else {
            hipError_t e = hipMemsetAsync(list_n, 0, sizeof(uint32_t), s);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(extract_lane_count_kernel, dim3(lane_grid(r.n_reads)), dim3(256), 0, s, r, ref, f, d_counts,
                               list, list_n);
        }
        return hipGetLastError();
    }
    const int tile_reads = r.max_read_len > kLaneReadMax ? 1 : 64;  // long reads: a wave each
    const int grid = grid_for_waves(((int64_t)r.n_reads + tile_reads - 1) / tile_reads, 4);
    if (write_pass)
        hipLaunchKernelGGL(extract_kernel<true>, dim3(grid), dim3(256), 0, s, r, ref, f, sub_hist, d_counts, tile_reads,
                           d_total_guard);
    else
        hipLaunchKernelGGL(extract_kernel<false>, dim3(grid), dim3(256), 0, s, r, ref, f, sub_hist, d_counts, tile_reads,
                           d_total_guard);
    return hipGetLastError();
}

namespace {
// read_cont[r] = container of read r from the containers' first-read table cont_first[n_cont + 1] (binary search)
__global__ void fill_read_cont_kernel(int32_t n_reads, const int32_t* cont_first, int32_t n_cont, int32_t* read_cont) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    int lo = 0, hi = n_cont - 1;  // largest c with cont_first[c] <= r
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (cont_first[mid] <= r)
            lo = mid;
        else
            hi = mid - 1;
    }
    read_cont[r] = lo;
}
}  // namespace

hipError_t launch_fill_read_cont(int32_t n_reads, const int32_t* d_cont_first, int32_t n_cont, int32_t* d_read_cont,
                                 hipStream_t s) {
    if (n_reads > 0)
        hipLaunchKernelGGL(fill_read_cont_kernel, dim3((n_reads + 255) / 256), dim3(256), 0, s, n_reads, d_cont_first, n_cont,
                           d_read_cont);
    return hipGetLastError();
}

size_t feature_scan_ws_elems(int32_t n_reads) { return (size_t)((n_reads + kScanTile - 1) / kScanTile) + 2; }

hipError_t launch_feature_scan(int32_t n_reads, const int32_t* d_counts, int64_t* d_off, int64_t* d_total, int64_t* d_ws,
                               hipStream_t s) {
    const int32_t n_tiles = (n_reads + kScanTile - 1) / kScanTile;  // 0 tiles for n_reads == 0: off[0] = total = 0
    if (n_tiles > 0)
        hipLaunchKernelGGL(scan_tile_sums_kernel, dim3(n_tiles), dim3(kScanThreads), 0, s, d_counts, n_reads, d_ws);
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(1), dim3(kScanThreads), 0, s, d_ws, n_tiles, d_total);
    if (n_tiles > 0)
        hipLaunchKernelGGL(scan_apply_kernel, dim3(n_tiles), dim3(kScanThreads), 0, s, d_counts, n_reads, d_ws, n_tiles, d_off);
    else
        return hipMemcpyAsync(d_off, d_ws, sizeof(int64_t), hipMemcpyDeviceToDevice, s);
    return hipGetLastError();
}

hipError_t launch_assign_codes(int32_t n_reads, const int32_t* read_cont, const int64_t* off, const uint8_t* op,
                               int32_t* a, const uint8_t* d_codes, hipStream_t s) {
    hipLaunchKernelGGL(assign_codes_kernel, dim3((n_reads + 255) / 256), dim3(256), 0, s, n_reads, read_cont, off, op, a,
                       d_codes);
    return hipGetLastError();
}

hipError_t launch_matrices_and_codes(const int64_t* d_hist, int32_t n_cont, uint8_t* d_mat5, uint8_t* d_codes25, int32_t n_reads,
                                     const int32_t* read_cont, const int64_t* off, const uint8_t* op, int32_t* a,
                                     const int64_t* d_total_guard, int64_t cap, hipStream_t s) {
    hipLaunchKernelGGL(matrices_kernel, dim3((n_cont + 63) / 64), dim3(64), 0, s, d_hist, n_cont, d_mat5, d_codes25);
    if (n_reads > 0)
        hipLaunchKernelGGL(assign_codes25_kernel, dim3((n_reads + 255) / 256), dim3(256), 0, s, n_reads, read_cont, off, op, a,
                           d_codes25, d_total_guard, cap);
    return hipGetLastError();
}

hipError_t launch_restore_bases(int32_t n_reads, const int32_t* read_len, const int32_t* aln_start, const uint8_t* rflags,
                                const int32_t* read_cont, const cramgpu_refwin* wins, cramgpu_features f,
                                const uint8_t* seq, const int64_t* seq_off, int64_t seq_bytes, const uint8_t* ref,
                                const uint8_t* d_basesof, uint8_t* out, const int64_t* out_off, int32_t* status, hipStream_t s) {
    // reads up to kLaneReadMax bases: a lane each; longer ones: a wave each (found 64 reads at a time)
    hipLaunchKernelGGL(restore_bases_lane_kernel, dim3(lane_grid(n_reads)), dim3(256), 0, s, n_reads, read_len, aln_start,
                       rflags, read_cont, wins, f, seq, seq_off, seq_bytes, ref, d_basesof, out, out_off, status);
    hipLaunchKernelGGL(restore_bases_kernel, dim3(grid_for_waves(((int64_t)n_reads + 63) / 64, 4)), dim3(256), 0, s, n_reads, read_len,
                       aln_start, rflags, read_cont, wins, f, seq, seq_off, seq_bytes, ref, d_basesof, out, out_off, status);
    return hipGetLastError();
}

hipError_t launch_restore_quals(int32_t n_reads, const int32_t* read_len, const uint8_t* rflags, cramgpu_features f,
                                uint8_t default_qs, uint8_t* scores, const int64_t* out_off, int32_t* status,
                                hipStream_t s) {
    hipLaunchKernelGGL(restore_quals_lane_kernel, dim3(lane_grid(n_reads)), dim3(256), 0, s, n_reads, read_len, rflags,
                       f, default_qs, scores, out_off, status);
    hipLaunchKernelGGL(restore_quals_kernel, dim3(grid_for_waves(((int64_t)n_reads + 63) / 64, 4)), dim3(256), 0, s, n_reads, read_len, rflags,
                       f, default_qs, scores, out_off, status);
    return hipGetLastError();
}

hipError_t launch_lossy_quals(const cramgpu_reads& r, const uint8_t* ref, const int32_t* wpos, int32_t window,
                              const cramgpu_policy* d_pol, int32_t n_pol, uint8_t* scores, uint8_t* force,
                              int32_t* status, uint8_t* d_mask_ws, int32_t* d_list, uint32_t* d_list_n, hipStream_t s) {
    const int mask_lds = r.max_read_len + 1 <= kMaskLds ? ((r.max_read_len + 1 + 255) & ~255) : kMaskLds;
    const int per_cu = mask_lds <= 4096 ? 32 : 8;  // single-wave workgroups resident per CU
    int grid = r.n_reads < g_n_cu * per_cu ? (r.n_reads > 0 ? r.n_reads : 1) : g_n_cu * per_cu;
    static const bool lane_ok = [] {
        const char* e = getenv("CRAMGPU_K5_LANE");
        return !(e && atoi(e) == 0);
    }();
    if (lane_ok && r.max_read_len <= kLaneReadMax && n_pol <= kLanePolMax && d_list && d_list_n) {
        // short reads: a lane per read; what it leaves (reads the reference throws on) goes to the wave kernel
        hipError_t e = hipMemsetAsync(d_list_n, 0, sizeof(uint32_t), s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(lossy_quals_lane_kernel, dim3(lane_grid(r.n_reads)), dim3(256), 0, s, r, ref, wpos, window, d_pol, n_pol,
                           scores, force, status, d_list, d_list_n);
        hipLaunchKernelGGL(lossy_quals_kernel, dim3(grid < 256 ? grid : 256), dim3(64), (size_t)mask_lds, s, r, ref, wpos, window,
                           d_pol, n_pol, scores, force, status, d_mask_ws, mask_lds, d_list, d_list_n);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(lossy_quals_kernel, dim3(grid), dim3(64), (size_t)mask_lds, s, r, ref, wpos, window, d_pol, n_pol, scores,
                       force, status, d_mask_ws, mask_lds, (const int32_t*)nullptr, (const uint32_t*)nullptr);
    return hipGetLastError();
}

hipError_t launch_merge_quality(int32_t n_reads, const int32_t* read_len, const int64_t* base_off, const uint8_t* scores,
                                const uint8_t* force, const int32_t* qstatus, cramgpu_features in, cramgpu_features out,
                                int32_t* d_counts, bool write_pass, const int64_t* d_total_guard, const int64_t* d_in_total_guard,
                                hipStream_t s) {
    const int grid = (n_reads + 255) / 256;
    if (write_pass)
        hipLaunchKernelGGL(merge_quality_kernel<true>, dim3(grid), dim3(256), 0, s, n_reads, read_len, base_off, scores,
                           force, qstatus, in, out, d_counts, d_total_guard, d_in_total_guard);
    else
        hipLaunchKernelGGL(merge_quality_kernel<false>, dim3(grid), dim3(256), 0, s, n_reads, read_len, base_off, scores,
                           force, qstatus, in, out, d_counts, d_total_guard, d_in_total_guard);
    return hipGetLastError();
}
