// Internal declarations shared by the HIP translation units of libcramgpu (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cramgpu.h"

// dynamic LDS of a kernel (the host-side sanitizer harness under tests/cpu_sim substitutes a heap buffer)
#ifdef HIP_SIM
#define CRAMGPU_DYN_LDS(name) uint8_t* name = hipsim::dyn_lds()
#else
#define CRAMGPU_DYN_LDS(name) extern __shared__ __align__(16) uint8_t name[]
#endif

// Lanes of a wave execute in lock step on the GPU, so "lane A reads LDS, then (later in program order) lane B
// overwrites it" needs nothing there; the host harness runs lanes as free threads and needs a rendezvous.
#ifdef HIP_SIM
#define CRAMGPU_LOCKSTEP() __builtin_amdgcn_wave_barrier()
#else
#define CRAMGPU_LOCKSTEP() ((void)0)
#endif

#define RANS_TF_SHIFT 12
#define RANS_TOTFREQ 4096
#define RANS_BYTE_L (1u << 23)

// global scratch of one resident rANS workgroup: order-1 tables that do not fit LDS
//   encode: counts u32[n][n|1] (<= 257 KiB, at 0) + packed symbols uint2[n][n|1] (<= 514 KiB, at 264 KiB), n <= 256
//   decode: order-1 rows u32[256][256] (256 KiB)
#define ENC_GSCRATCH_BYTES (784u * 1024u)
#define DEC_GSCRATCH_BYTES (256u * 1024u)
#define DEC_SIMD_KEYS 8192u  // XCD (3 bits) | shader engine, array, CU (8 bits of HW_ID) | SIMD (2 bits)

struct RansArgs {
    const uint8_t* in;
    uint8_t* out;
    const cramgpu_block* desc;  // device copy of the descriptors
    const int32_t* perm;        // blocks sorted by descending length (longest-processing-time first)
    int32_t n_blocks;
    int32_t* out_len;
    int64_t* out_pos;           // encode only
    uint32_t* ticket;           // work-queue head, zero at the launch (decode: 8 bytes, 8-byte aligned: the
                                // blocks taken from the front and from the back of perm)
    uint32_t* simd_arrivals;    // decode: [DEC_SIMD_KEYS] number of the last launch whose wave came by, per SIMD; or NULL
    uint32_t launch_no;         // decode: this launch's number (> every earlier one on this table)
    uint8_t* gscratch;          // n_groups x {ENC,DEC}_GSCRATCH_BYTES
    const uint32_t* rcp_tab;    // [4096] Encoding#RansEncSymbolInit rcp_freq per freq
    int32_t tab_bytes;          // bytes of the LDS table region
    unsigned long long* prof;   // CRAMGPU_PROF=1: 8 cycle counters per block (encode), else NULL
};

// launchers (defined in the .hip files)
hipError_t launch_rans_encode(const RansArgs& a, int n_groups, size_t lds_bytes, hipStream_t s);  // 2-wave groups
hipError_t launch_rans_decode(const RansArgs& a, int n_waves, size_t lds_bytes, hipStream_t s);
hipError_t launch_rcp_init(uint32_t* d_rcp_tab, hipStream_t s);
hipError_t launch_lpt_perm(const cramgpu_block* d_desc, int32_t n, int32_t* d_perm, hipStream_t s);  // n <= 4096
// cramgpu_rans_encode_series_dev: the block table and its longest-first order made on the device (gather.hip)
struct SeriesPick {
    int32_t n_ids;
    int32_t ids[CRAMGPU_N_SERIES];     // content ids of the series to code, in block order
    int32_t orders[CRAMGPU_N_SERIES];  // their rANS orders
};
hipError_t launch_series_blocks(const int64_t* d_series_off, const int32_t* d_series_len, int32_t n_cont, const SeriesPick& pick,
                                int64_t in_shift, int64_t out_cap, const int32_t* d_w_status, cramgpu_block* d_blocks,
                                int32_t* d_status, hipStream_t s);
size_t lpt_sort_bytes(int32_t n);
hipError_t launch_lpt_sort(const cramgpu_block* d_desc, int32_t n, int32_t* d_perm, void* ws, size_t ws_bytes, hipStream_t s);
size_t rans_encode_fixed_lds();
size_t rans_decode_fixed_lds();
hipError_t launch_gather(const uint8_t* src, const int64_t* pos, const int32_t* len, int32_t n, int32_t align,
                         uint8_t* dst, int64_t dst_cap, int64_t* dst_off, int32_t* err, int32_t max_len,
                         hipStream_t s);

// record kernels
hipError_t launch_write_series(const cramgpu_reads& r, const cramgpu_record_cols& c, const cramgpu_features& f, uint8_t* arena,
                               int64_t cap, int64_t* d_off, int32_t* d_len, int32_t* d_status, uint8_t* d_alias, hipStream_t s);
size_t read_series_ws_ints(int64_t series_bytes, int32_t n_cont);
hipError_t launch_read_series(const uint8_t* arena, const int64_t* d_off, const int32_t* d_len, int32_t n_cont, int32_t n_reads,
                              const int32_t* d_cont_first, const cramgpu_record_cols& c, const cramgpu_read_out& o,
                              const cramgpu_features& f, int32_t* ws_vals, int64_t ws_vals_cap, int64_t* ws_off, int32_t* ws_cnt,
                              int64_t* ws_base, int32_t* d_status, hipStream_t s);  // ws_vals_cap: ints behind ws_vals
void records_set_cu_count(int n_cu);  // grid sizing of the record kernels
hipError_t launch_extract_features(const cramgpu_reads& r, const uint8_t* ref, cramgpu_features f, int64_t* sub_hist,
                                   int32_t* d_counts, bool write_pass, int32_t* list, uint32_t* list_n,
                                   const int64_t* d_total_guard, hipStream_t s);  // guard: write nothing if *guard > f.cap
hipError_t launch_fill_read_cont(int32_t n_reads, const int32_t* d_cont_first, int32_t n_cont, int32_t* d_read_cont,
                                 hipStream_t s);
size_t feature_scan_ws_elems(int32_t n_reads);  // int64 elements of workspace launch_feature_scan needs
hipError_t launch_feature_scan(int32_t n_reads, const int32_t* d_counts, int64_t* d_off, int64_t* d_total, int64_t* d_ws,
                               hipStream_t s);
hipError_t launch_assign_codes(int32_t n_reads, const int32_t* read_cont, const int64_t* off, const uint8_t* op,
                               int32_t* a, const uint8_t* d_codes, hipStream_t s);
hipError_t launch_matrices_and_codes(const int64_t* d_hist, int32_t n_cont, uint8_t* d_mat5, uint8_t* d_codes25, int32_t n_reads,
                                     const int32_t* read_cont, const int64_t* off, const uint8_t* op, int32_t* a,
                                     const int64_t* d_total_guard, int64_t cap, hipStream_t s);  // guard: touch no feature if *guard > cap
hipError_t launch_restore_bases(int32_t n_reads, const int32_t* read_len, const int32_t* aln_start, const uint8_t* rflags,
                                const int32_t* read_cont, const cramgpu_refwin* wins, cramgpu_features f,
                                const uint8_t* seq, const int64_t* seq_off, int64_t seq_bytes, const uint8_t* ref,
                                const uint8_t* d_basesof, uint8_t* out, const int64_t* out_off, int32_t* status, hipStream_t s);
hipError_t launch_restore_quals(int32_t n_reads, const int32_t* read_len, const uint8_t* rflags, cramgpu_features f,
                                uint8_t default_qs, uint8_t* scores, const int64_t* out_off, int32_t* status,
                                hipStream_t s);
hipError_t launch_lossy_quals(const cramgpu_reads& r, const uint8_t* ref, const int32_t* wpos, int32_t window,
                              const cramgpu_policy* d_pol, int32_t n_pol, uint8_t* scores, uint8_t* force, int32_t* status,
                              uint8_t* d_mask_ws, int32_t* d_list, uint32_t* d_list_n, hipStream_t s);
hipError_t launch_merge_quality(int32_t n_reads, const int32_t* read_len, const int64_t* base_off, const uint8_t* scores,
                                const uint8_t* force, const int32_t* qstatus, cramgpu_features in, cramgpu_features out,
                                int32_t* d_counts, bool write_pass, const int64_t* d_total_guard, const int64_t* d_in_total_guard,
                                hipStream_t s);  // in-guard: the input set overflowed its capacity and does not exist

// container assembly / BAM records (bam.hip)
hipError_t launch_cigars(int32_t n, const int32_t* read_len, const int32_t* aln_start, const int32_t* bam_flags,
                         const cramgpu_features& f, int32_t* n_cigar, int64_t* cigar_off, uint32_t* cigar, int64_t cap,
                         int32_t* aln_end, int32_t* status, int64_t* d_total, int64_t* d_scan_ws, hipStream_t s);
hipError_t launch_restore_mates(int32_t n, int32_t n_cont, const int32_t* cont_first, const cramgpu_record_cols& c,
                                const int32_t* cont_ref_id, const int32_t* aln_start, const int32_t* aln_end, int32_t* link,
                                uint32_t* nprev, uint8_t* serial, int32_t* status, hipStream_t s);
size_t mate_table_slots(int32_t n);  // set_mate_info workspace: 3 * slots + n ints
hipError_t launch_set_mate_info(int32_t n, int32_t n_cont, const int32_t* cont_first, const cramgpu_record_cols& c,
                                const int32_t* aln_start, const int32_t* aln_end, const uint8_t* names, const int64_t* name_off,
                                const int32_t* name_len, int32_t* ws, hipStream_t s);
hipError_t launch_content_crc(int32_t n, int32_t n_cont, const int32_t* cont_first, const int64_t* base_off, const int32_t* read_len,
                              const uint8_t* bases, const uint8_t* quals, uint32_t* crc_sum, hipStream_t s);
hipError_t launch_bam_encode(int32_t n, const cramgpu_bam_cols& b, uint8_t* bam, int64_t cap, int64_t* rec_off, int32_t* status,
                             int32_t* d_sizes, int64_t* d_total, int64_t* d_scan_ws, hipStream_t s);
hipError_t launch_bam_decode(int32_t n, const uint8_t* bam, const int64_t* rec_off, const cramgpu_bam_out& o, int32_t* status,
                             int64_t* d_total, int64_t* d_scan_ws, hipStream_t s);
size_t bam_index_segments(int64_t len);
hipError_t launch_bam_index(const uint8_t* bam, int64_t len, int32_t n_ref, int64_t* rec_off, int64_t cap, int64_t* d_count,
                            int32_t* d_status, int64_t* ws_entry, int64_t* ws_exit, int32_t* ws_count, int64_t* ws_base,
                            int64_t* d_total, int64_t* d_scan_ws, hipStream_t s);
hipError_t launch_slice_md5(int32_t n, const uint8_t* ref, const cramgpu_refwin* win, const int32_t* start, const int32_t* span,
                            uint8_t* md5, hipStream_t s);
hipError_t launch_gunzip(const uint8_t* in, uint8_t* out, const cramgpu_block* d_desc, int32_t n, int32_t* out_len, hipStream_t s);
hipError_t launch_gzip(const uint8_t* in, uint8_t* out, const cramgpu_block* d_desc, int32_t n, int32_t* out_len, int32_t bgzf,
                       uint32_t* ws, int n_waves, hipStream_t s);  // ws: gzip_ws_bytes(n_waves) of token space (nullptr: lane-per-stream kernel)
size_t gzip_ws_bytes(int n_waves);
