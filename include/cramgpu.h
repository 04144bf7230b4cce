/*
 * cramgpu.h — C ABI of libcramgpu.so, the MI355X (gfx950) engine for the cramtools CRAM hot path.
 *
 * Every entry point names the reference interface it stands in for.  Citation forms:
 *   src/...java:N                        a line of /root/reference/src/...
 *   htsjdk!pkg/Class#method (L a-b)      a method of htsjdk 2.4.1 (lib/htsjdk-2.4.1.jar, bytecode only),
 *                                        original source line numbers from its LineNumberTable
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types.
 *   - "host" pointers are ordinary process memory; "dev" pointers are HIP device memory on the context's GPU.
 *   - stream arguments are hipStream_t passed as void* (NULL = HIP's null stream, i.e. torch's default stream).
 *   - functions return >= 0 on success and a negative CRAMGPU_E_* code on failure; per-block results of
 *     batch calls carry the same codes in out_len[].
 *   - all functions are thread-safe on distinct contexts; one context serialises its calls internally: on the host
 *     by a mutex, on the device by an event (consecutive calls on DIFFERENT streams of one context are ordered one
 *     behind the other, because they share the context's scratch; calls on one stream are stream-ordered anyway).
 *     Use one context per stream that should run concurrently.
 *   - the level-0 functions share ONE process-wide context on device 0: concurrent callers (TransformerJob threads)
 *     serialise on it.  They are the day-one shim; a converter thread that wants overlap uses level 1.
 */
#ifndef CRAMGPU_H
#define CRAMGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes: the unchecked exceptions of the reference mapped to integers ---- */
#define CRAMGPU_OK 0
#define CRAMGPU_E_INPUT_LENGTH (-1)   /* RANS#uncompress L29-56: "Incorrect input length." */
#define CRAMGPU_E_OUT_TOO_SMALL (-2)  /* RANS#uncompress: "Output buffer too small to fit N bytes." */
#define CRAMGPU_E_INSUFFICIENT (-3)   /* RANS#allocateIfNeeded L79-85 / BufferOverflowException: "Insufficient buffer size." */
#define CRAMGPU_E_ORDER (-4)          /* RANS.ORDER.fromInt: "Unknown rANS order: X" */
#define CRAMGPU_E_CORRUPT (-5)        /* Java raises an index / buffer-underflow exception on a malformed stream */
#define CRAMGPU_E_ARG (-6)            /* NULL pointer, misaligned offset, negative size */
#define CRAMGPU_E_HIP (-7)            /* a HIP runtime call failed; cramgpu_last_error() has the text */
#define CRAMGPU_E_NO_DEVICE (-8)      /* no gfx950 device visible: the library never falls back to a CPU path */
#define CRAMGPU_E_JAVA_THROWS (-9)    /* record kernels: the reference would throw (index out of range, NPE) */
#define CRAMGPU_E_UNSUPPORTED (-10)   /* a lossy-model term that needs ReferenceTracks' coverage / mismatch tracks */

#define CRAMGPU_ARENA_ALIGN 16        /* every block / stream offset inside a device arena is a multiple of this */

typedef struct cramgpu_ctx cramgpu_ctx;

/* ---- context ---- */
/* One context per GPU (containers shard over GPUs by index, one context each).
 * Has no counterpart in the reference (which is a single JVM); it owns the HIP stream, the pinned staging
 * arenas, the device workspace of the persistent rANS kernels and the per-device rcp table. */
cramgpu_ctx* cramgpu_open(int32_t device_index);
void cramgpu_close(cramgpu_ctx* ctx);
const char* cramgpu_last_error(void);
int32_t cramgpu_device_count(void);
/* CUs of the context's device and the number of resident rANS waves it launches. */
int32_t cramgpu_info(cramgpu_ctx* ctx, int32_t* n_cu, int32_t* n_waves);

/* =====================================================================================================
 * Level 0 — block codec, synchronous, host buffers.
 * Replaces htsjdk!io/ExternalCompression#rans (L89-90, L101-102) and #unrans (L112-113), i.e.
 * htsjdk!encoding/rans/RANS#compress (L60-73) / #uncompress (L29-56), as called from
 * htsjdk!build/ContainerFactory#buildSlice (L129-205), htsjdk!build/CompressionHeaderFactory
 * #getBestExternalCompressor (L307-323) and htsjdk!structure/Block#uncompress (L228-262).
 * These four use a process-wide default context on device 0 (created on first use).
 * ===================================================================================================== */

/* RANS#allocateIfNeeded L79-85: (int)(1.05*n + 257*257*3 + 4). */
int32_t cramgpu_rans_bound(int32_t n);

/* RANS#compress: out receives [order:1][compSize:LE32][rawSize:LE32][freq table][payload].
 * n == 0 -> returns 0 (EMPTY_BUFFER); n < 4 -> order 0 whatever was asked.  Returns bytes written. */
int32_t cramgpu_rans_compress(const uint8_t* in, int32_t n, int32_t order, uint8_t* out, int32_t cap);

/* RANS#uncompress: returns the raw size, out must hold cramgpu_rans_raw_size() bytes. */
int32_t cramgpu_rans_uncompress(const uint8_t* in, int32_t n, uint8_t* out, int32_t cap);

/* LE32 at offset 5 of a stream (RANS#uncompress L29-56 header layout); 0 for an empty stream. */
int32_t cramgpu_rans_raw_size(const uint8_t* in, int32_t n);

/* =====================================================================================================
 * Level 1 — batch of blocks (one call = all external blocks of one or many containers).
 * The batch seam of the reference is the per-container compress loop of ContainerFactory#buildSlice L198-201,
 * the eager uncompress of Block#readFromInputStream L78-100, and one level up the
 * Function<OrderedByteArray,OrderedByteArray> of src/main/java/htsjdk/samtools/cram/paralell/
 * BamToCram_OBA_Function.java:51-94 / CramToBam_OBA_Function.java:53-83.
 * ===================================================================================================== */
typedef struct {
    int64_t in_off;   /* offset of the block's input in the input arena (multiple of CRAMGPU_ARENA_ALIGN) */
    int64_t out_off;  /* offset of the block's output slot in the output arena (multiple of CRAMGPU_ARENA_ALIGN) */
    int32_t in_len;   /* raw bytes (encode) / stream bytes (decode) */
    int32_t out_cap;  /* slot size: >= cramgpu_rans_bound(in_len) for encode, >= raw size for decode */
    int32_t order;    /* encode: 0 or 1 (RANS.ORDER); decode: ignored */
    int32_t reserved;
} cramgpu_block;

/* Device-resident batch encode.  d_in / d_out are device arenas, blocks[] is a HOST array.
 * After the call (stream-ordered) block b's stream occupies d_out[d_out_pos[b] .. + d_out_len[b]) — it is
 * right-aligned inside its slot — or d_out_len[b] is a negative CRAMGPU_E_* code.
 * d_out_len (int32[n]) and d_out_pos (int64[n]) are DEVICE arrays. */
int32_t cramgpu_rans_encode_dev(cramgpu_ctx* ctx, const uint8_t* d_in, uint8_t* d_out, const cramgpu_block* blocks,
                                int32_t n_blocks, int32_t* d_out_len, int64_t* d_out_pos, void* stream);

/* cramgpu_rans_encode_dev for callers that cannot allocate device memory: code running under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc.  It has no counterpart in the reference (a JVM allocates freely).
 * Same arenas, same results and same d_out_len / d_out_pos as cramgpu_rans_encode_dev, with two differences:
 *   - d_blocks is a DEVICE array of descriptors (the host table would need a staged upload);
 *   - the kernels' scratch (work-queue ticket, longest-first permutation, per-workgroup order-1 tables of wide
 *     alphabets) lives in a static device array of the library (about 24.6 MiB per GPU, held until the process ends)
 *     instead of buffers grown with hipMalloc: the call allocates nothing, frees nothing and creates nothing.
 * The descriptors never pass through the host, so the checks cramgpu_rans_encode_dev makes there (offsets non-negative
 * multiples of CRAMGPU_ARENA_ALIGN, sizes >= 0) are made by the kernel, block by block: such a block gets CRAMGPU_E_ARG (a
 * slot too small CRAMGPU_E_INSUFFICIENT) in d_out_len and nothing of it is read or written.
 * Limit: n_blocks <= 4096 per call (CRAMGPU_E_ARG beyond); at most 32 workgroups run the batch, so it suits one or a
 * few containers per call, not a whole read set.  All calls on one GPU share the one array: calls that overlap on a
 * GPU (different contexts or processes' streams on the same device) race; calls on one context are ordered by the
 * library as every other call is.  Stream-ordered, does not synchronise. */
int32_t cramgpu_rans_encode_static(cramgpu_ctx* ctx, const uint8_t* d_in, uint8_t* d_out, const cramgpu_block* d_blocks,
                                   int32_t n_blocks, int32_t* d_out_len, int64_t* d_out_pos, void* stream);

/* For a caller that launches another kernel BESIDE a rANS launch on a second stream (the gunzip of a batch's GZIP series
 * beside its rANS decode): makes `stream` wait until the block table of this context's last cramgpu_rans_{encode,decode}_dev
 * call has reached the device, i.e. until that call's kernel is the next thing in its own stream.  Without it the second
 * stream's kernel can reach the CUs first when the GPU is backlogged — the rANS kernels are resident grids, and a grid
 * that finds the wave slots taken starts its long blocks late (decode: 50 ms instead of 26).  No counterpart in the
 * reference.  Stream-ordered, does not synchronise. */
int32_t cramgpu_stream_wait_staged(cramgpu_ctx* ctx, void* stream);

/* Device-resident batch decode: block b's raw bytes land at d_out[blocks[b].out_off ..); d_out_len[b] = raw size
 * or a negative code. */
int32_t cramgpu_rans_decode_dev(cramgpu_ctx* ctx, const uint8_t* d_in, uint8_t* d_out, const cramgpu_block* blocks,
                                int32_t n_blocks, int32_t* d_out_len, void* stream);

/* Packs n variable-length pieces d_src[d_pos[b] .. + max(d_len[b],0)) back to back (each start rounded up to
 * `align` bytes, a power of two >= 1) into d_dst, writes the start offsets to d_dst_off (int64[n+1], device).
 * max_len = an upper bound of every d_len[b] (e.g. the largest out_cap).
 * Used to shrink the encode output before the D2H copy.  No reference counterpart (Java's toByteArray L146-150
 * is the per-block analogue).  d_dst_off is always complete; when the packed size d_dst_off[n] exceeds dst_cap NOTHING
 * is copied (the call is stream-ordered and cannot report it): size d_dst for the sum of the rounded-up lengths, or
 * check d_dst_off[n] <= dst_cap before reading d_dst. */
int32_t cramgpu_gather_dev(cramgpu_ctx* ctx, const uint8_t* d_src, const int64_t* d_pos, const int32_t* d_len,
                           int32_t n, int32_t max_len, int32_t align, uint8_t* d_dst, int64_t dst_cap,
                           int64_t* d_dst_off, void* stream);

/* Host-buffer batch (pinned staging + hipMemcpyAsync inside): in_arena/out_arena are host memory laid out by
 * blocks[] exactly like the device arenas; on return out_len[b] holds the stream length (encode: the stream
 * has been moved to the START of its slot, out_arena + blocks[b].out_off) or a negative code. */
int32_t cramgpu_rans_encode_batch(cramgpu_ctx* ctx, const uint8_t* in_arena, int64_t in_bytes, uint8_t* out_arena,
                                  int64_t out_bytes, const cramgpu_block* blocks, int32_t n_blocks,
                                  int32_t* out_len);
int32_t cramgpu_rans_decode_batch(cramgpu_ctx* ctx, const uint8_t* in_arena, int64_t in_bytes, uint8_t* out_arena,
                                  int64_t out_bytes, const cramgpu_block* blocks, int32_t n_blocks,
                                  int32_t* out_len);

/* Asynchronous form of the host-buffer batch: what TransformerJob threads (paralell/TransformerJob.java:16-41) and
 * CRAMContainerAsynchWriter.Batch.run (CRAMContainerAsynchWriter.java:139-149) do with a thread per container, the
 * library does with a queue per context: cramgpu_rans_submit copies the input arena to the GPU, codes the blocks,
 * packs the streams on the device and fetches lengths with hipMemcpyAsync on the job slot's own stream and returns at
 * once; the Java thread marshals the next container meanwhile.  cramgpu_job_wait brings the streams into out_arena (as
 * cramgpu_rans_*_batch leaves them: ONE copy of the packed streams, spread to their slots on the host; encode slots
 * must therefore be ascending and non-overlapping) and out_len, and releases the job; cramgpu_job_poll says whether
 * wait would block (1 = finished).
 * (cramgpu_job_poll / cramgpu_job_wait also serve the jobs of cramgpu_submit_reads / cramgpu_submit_restore.)
 * Arenas and out_len must stay valid and untouched from submit to wait; for copies that overlap host work they
 * should come from cramgpu_host_alloc (pinned memory, exposed to Java as direct ByteBuffers).  At most
 * CRAMGPU_MAX_JOBS may be outstanding per context (submit fails with CRAMGPU_E_ARG beyond); every slot has its own
 * stream and launch resources, so the copies and kernels of outstanding jobs overlap.  One context per GPU. */
#define CRAMGPU_MAX_JOBS 16
typedef struct cramgpu_job cramgpu_job;
void* cramgpu_host_alloc(int64_t bytes);
void cramgpu_host_free(void* p);
cramgpu_job* cramgpu_rans_submit(cramgpu_ctx* ctx, int32_t decode, const uint8_t* in_arena, int64_t in_bytes,
                                 uint8_t* out_arena, int64_t out_bytes, const cramgpu_block* blocks, int32_t n_blocks,
                                 int32_t* out_len);
/* The encode form a JNI host wants: the streams STAY packed back to back (16-byte aligned starts) from the beginning
 * of out_arena and out_pos (int64[n_blocks + 1]) receives their starts after the wait; Block.setContent copies each
 * stream into its own byte[] anyway, so nothing is gained by spreading them to slots on the host first.  blocks[b]
 * .out_off is ignored beyond sizing (out_cap is the slot the kernels code into on the device). */
cramgpu_job* cramgpu_rans_submit_packed(cramgpu_ctx* ctx, const uint8_t* in_arena, int64_t in_bytes, uint8_t* out_arena,
                                        int64_t out_bytes, const cramgpu_block* blocks, int32_t n_blocks, int32_t* out_len,
                                        int64_t* out_pos);
int32_t cramgpu_job_poll(cramgpu_job* job);
int32_t cramgpu_job_wait(cramgpu_job* job);

/* Coalescing queue: the form for a host that keeps the reference's thread model, ONE container per worker thread
 * (paralell/TransformerJob.java:29-41 runs one BamToCram_OBA_Function / CramToBam_OBA_Function per stripe,
 * BamToCram.java:146-155 feeds them, CRAMContainerAsynchWriter.java:139-149 runs a Batch per container).  A launch
 * per container lasts one chain latency (the container's longest block) whatever its size, so the library commits in
 * groups: every worker calls cramgpu_rans_coalesced (blocking; or cramgpu_rans_enqueue + cramgpu_ticket_wait) with its
 * own container; a dispatcher thread of the queue turns ALL submissions waiting at that moment into one launch as soon
 * as one of `max_batches` batch slots (own stream and launch resources each; 0 = 6) is free, and a completer thread
 * copies every submission's streams into its own out_arena and wakes its caller.  Arguments and results are those of
 * cramgpu_rans_submit (out_pos == NULL: streams at the starts of their slots, decode: raw blocks at out_off) or
 * cramgpu_rans_submit_packed (out_pos != NULL).  Any number of threads may call at once; tickets are positive, a
 * negative return is a CRAMGPU_E_* code.  A ticket is released by its wait.  Arenas should be pinned
 * (cramgpu_host_alloc) and must stay untouched until the wait returns.  cramgpu_coalescer_close waits for the batches
 * in flight and refuses what was never launched (CRAMGPU_E_ARG at its wait); close it before its context. */
typedef struct cramgpu_coalescer cramgpu_coalescer;
cramgpu_coalescer* cramgpu_coalescer_open(cramgpu_ctx* ctx, int32_t max_batches);
void cramgpu_coalescer_close(cramgpu_coalescer* q);
int64_t cramgpu_rans_enqueue(cramgpu_coalescer* q, int32_t decode, const uint8_t* in_arena, int64_t in_bytes,
                             uint8_t* out_arena, int64_t out_bytes, const cramgpu_block* blocks, int32_t n_blocks,
                             int32_t* out_len, int64_t* out_pos);
int32_t cramgpu_ticket_poll(cramgpu_coalescer* q, int64_t ticket);
int32_t cramgpu_ticket_wait(cramgpu_coalescer* q, int64_t ticket);
int32_t cramgpu_rans_coalesced(cramgpu_coalescer* q, int32_t decode, const uint8_t* in_arena, int64_t in_bytes,
                               uint8_t* out_arena, int64_t out_bytes, const cramgpu_block* blocks, int32_t n_blocks,
                               int32_t* out_len, int64_t* out_pos);
/* launches made, submissions served and the most submissions one launch held, since the queue was opened */
int32_t cramgpu_coalesce_stats(cramgpu_coalescer* q, int64_t* n_batches, int64_t* n_submissions, int64_t* largest_batch);

/* Device pool: all GPUs of the node behind one handle, for a host that is ONE process (a JVM) — SURVEY.md §8(e).
 * Containers are independent, so there is no collective: the container with ordinal k goes to GPU k mod G (one context and
 * one coalescing queue per GPU) and results are handed back in ordinal order, which is what the reference's
 * src/main/java/htsjdk/samtools/cram/paralell/OrderingConveyer.java:24-38 does with the OrderedByteArrays its worker
 * threads return.  cramgpu_pool_submit may be called from any thread, in any order of ordinals (each ordinal once, counting
 * from 0); arguments are those of cramgpu_rans_enqueue and the arenas should be pinned (cramgpu_host_alloc memory is
 * pinned for every GPU).  cramgpu_pool_next blocks until the container with the NEXT ordinal (0, 1, 2, ...) is done, stores
 * that ordinal and returns its code; called from one thread (the writer).  n_devices = 0: every visible GPU. */
typedef struct cramgpu_pool cramgpu_pool;
cramgpu_pool* cramgpu_pool_open(int32_t n_devices, int32_t max_batches);
void cramgpu_pool_close(cramgpu_pool* p);
int32_t cramgpu_pool_devices(cramgpu_pool* p);
int32_t cramgpu_pool_submit(cramgpu_pool* p, int64_t ordinal, int32_t decode, const uint8_t* in_arena, int64_t in_bytes,
                            uint8_t* out_arena, int64_t out_bytes, const cramgpu_block* blocks, int32_t n_blocks,
                            int32_t* out_len, int64_t* out_pos);
int32_t cramgpu_pool_next(cramgpu_pool* p, int64_t* ordinal);

/* =====================================================================================================
 * Level 2 — record kernels over one container in flat (structure-of-arrays) form.
 * ===================================================================================================== */

/* per-read flag bits */
#define CRAMGPU_RF_UNMAPPED 1      /* SAMRecord.getReadUnmappedFlag() */
#define CRAMGPU_RF_NULL_QUALS 2    /* SAMRecord.NULL_QUALS on input */
#define CRAMGPU_RF_NO_BASES 4      /* encode: SAMRecord has no bases ("*"): treated as all 'N' */
#define CRAMGPU_RF_UNKNOWN_BASES 8 /* decode: CramCompressionRecord.CF_UNKNOWN_BASES */
#define CRAMGPU_RF_FORCE_QS 16     /* decode: CF_FORCE_PRESERVE_QS */
#define CRAMGPU_RF_SKIP 32         /* record not processed */

/* Reference bytes available to one container: d_ref[ref_off .. ref_off + ref_len) holds the 0-based reference
 * positions [ref_start0, ref_start0 + ref_len) of the container's contig.  Encode passes the whole contig
 * (ref_start0 = 0; Sam2CramRecordFactory#setRefBases, src/main/java/htsjdk/samtools/cram/CramSerilization.java:147-151),
 * decode passes the ReferenceRegion window of CramNormalizer.java:212-226.  Positions at or beyond
 * ref_start0 + ref_len read as past-the-end ('N'), exactly like the reference's length checks. */
typedef struct {
    int64_t ref_off;
    int64_t ref_len;
    int64_t ref_start0;
} cramgpu_refwin;

/* Reads of one or more containers, all arrays on the device, reads of a container contiguous.
 * CIGAR elements are BAM-packed (len<<4 | op, op in MIDNSHP=X). */
typedef struct {
    int32_t n_reads;
    int32_t n_cont;            /* number of containers (>= 1) */
    int64_t n_bases;           /* bytes in bases / quals */
    int32_t max_read_len;      /* largest read_len[r] */
    int32_t reserved;
    const uint8_t* bases;      /* concatenated read bases (ASCII) */
    const uint8_t* quals;      /* concatenated phred bytes, same offsets as bases (may be NULL where unused) */
    const int64_t* base_off;   /* [n_reads] offset of read r in bases/quals */
    const int32_t* read_len;   /* [n_reads] */
    const int32_t* aln_start;  /* [n_reads] 1-based alignment start, 0 = unplaced */
    const uint8_t* rflags;     /* [n_reads] CRAMGPU_RF_* */
    const uint8_t* mapq;       /* [n_reads] (lossy model only) */
    const uint32_t* cigar;     /* packed CIGAR elements */
    const int64_t* cigar_off;  /* [n_reads] */
    const int32_t* n_cigar;    /* [n_reads] */
    const int32_t* read_cont;  /* [n_reads] container index of read r, or NULL when n_cont == 1 */
    const cramgpu_refwin* win; /* [n_cont] */
} cramgpu_reads;

/* Flat read features (the layout of src/main/java/htsjdk/samtools/cram/encoding/reader/ReadFeatureBuffer.java:41-100
 * widened to fixed columns).  op = operator byte, pos = 1-based position in the read.
 *   X: a = base | refBase<<8 | code<<16 (code 0xFF = not yet assigned)   D N H P: a = length
 *   S, I: a = offset of the sequence in the read's bases (or in seq on decode), b = length
 *   i: a = base      Q: a = score      B: a = base, b = score */
typedef struct {
    uint8_t* op;
    int32_t* pos;
    int32_t* a;
    int32_t* b;
    int64_t* off;   /* [n_reads + 1] feature range of read r = [off[r], off[r+1]) */
    int64_t cap;    /* capacity of op/pos/a/b in features */
} cramgpu_features;

/* K3 — htsjdk!build/Sam2CramRecordFactory#createVariations (L205-264), #addSubstitutionsAndMaskedBases (L297-335),
 * #addInsertion (L282-292), #addSoftClip (L268-272), htsjdk!build/Utils#normalizeBase (L31-49); called per record
 * from src/main/java/htsjdk/samtools/cram/CramSerilization.java:154.  d_ref is the reference arena reads->win points
 * into.  Also accumulates d_sub_hist[n_cont][25] (int64, [ref][base] over ACGTN x ACGTN, must be zeroed by the
 * caller) for htsjdk!build/CompressionHeaderFactory#buildFrequencies (L196-209); may be NULL.
 * Returns the feature count or a negative code (CRAMGPU_E_INSUFFICIENT when feats->cap is too small; feats->off is
 * valid then and off[n_reads] is the capacity needed).  Synchronises the stream (the count comes back to the host). */
int64_t cramgpu_extract_features_dev(cramgpu_ctx* ctx, const cramgpu_reads* reads, const uint8_t* d_ref,
                                     cramgpu_features* feats, int64_t* d_sub_hist, void* stream);

/* F3 — htsjdk!structure/SubstitutionMatrix#<init>([[J) (L38-57), #rank (L141-169): 25 counts -> 5 header bytes.
 * Host function (25 numbers; the reference does the same per container). */
void cramgpu_substitution_matrix(const int64_t* sub_hist, uint8_t* matrix5);

/* htsjdk!encoding/writer/Writer#write (L139-261) picks matrix.code(ref, base) for each Substitution: fills bits 16..23
 * of a[] for every X feature.  matrices = HOST array of n_cont x 5 header bytes (copied before the call returns);
 * d_read_cont as in cramgpu_reads.  Stream-ordered, does not synchronise. */
int32_t cramgpu_assign_codes_dev(cramgpu_ctx* ctx, int32_t n_reads, const int32_t* d_read_cont, int32_t n_cont,
                                 const cramgpu_features* feats, const uint8_t* matrices, void* stream);

/* F3 + code assignment without a host round trip: htsjdk!structure/SubstitutionMatrix#<init>([[J) (L38-57) for every
 * container of the batch from d_sub_hist[n_cont][25] (as accumulated by cramgpu_extract_features_dev), the 5 header
 * bytes of container k written to d_matrices[5k..5k+5) (DEVICE), then Writer#write's matrix.code(ref, base) for every
 * X feature as in cramgpu_assign_codes_dev.  Stream-ordered, does not synchronise; the context keeps the code tables
 * until its next call of this function. */
int32_t cramgpu_matrices_assign_codes_dev(cramgpu_ctx* ctx, int32_t n_reads, const int32_t* d_read_cont, int32_t n_cont,
                                          const cramgpu_features* feats, const int64_t* d_sub_hist, uint8_t* d_matrices,
                                          void* stream);

/* K4 / N1 — src/main/java/htsjdk/samtools/cram/build/CramNormalizer.java:335-441 (restoreReadBases), called from
 * :137 and :239 with a ReferenceRegion window (d_wins[n_cont] into d_ref).  matrices = HOST n_cont x 5 bytes.
 * d_seq/d_seq_off: bytes of S / I features (feature a = offset relative to d_seq_off[r]); seq_bytes = size of d_seq:
 * a feature whose bytes would lie outside [0, seq_bytes) gives its read CRAMGPU_E_JAVA_THROWS.
 * d_status[r]: 0 bases written, 1 NULL_SEQUENCE, 2 skipped, CRAMGPU_E_JAVA_THROWS (a feature position outside the
 * read, a window the read starts in front of: the reference raises an index exception there).
 * Stream-ordered, does not synchronise (the matrices are copied before the call returns). */
int32_t cramgpu_restore_bases_dev(cramgpu_ctx* ctx, int32_t n_reads, const int32_t* d_read_len, const int32_t* d_aln_start,
                                  const uint8_t* d_rflags, const int32_t* d_read_cont, int32_t n_cont,
                                  const cramgpu_refwin* d_wins, const cramgpu_features* feats, const uint8_t* d_seq,
                                  const int64_t* d_seq_off, int64_t seq_bytes, const uint8_t* d_ref, const uint8_t* matrices,
                                  uint8_t* d_out, const int64_t* d_out_off, int32_t* d_status, void* stream);

/* N2 — CramNormalizer.java:287-333 (restoreQualityScores; default score 30 = '?'-'!', :141,:204).
 * d_scores is in/out at d_out_off[r]; d_status[r]: 0 valid, 1 NULL_QUALS, 2 skipped, CRAMGPU_E_JAVA_THROWS (a Q or B
 * feature whose position lies outside [1, read_len]: CramNormalizer.java:302,307 index scores[pos - 1]). */
int32_t cramgpu_restore_quals_dev(cramgpu_ctx* ctx, int32_t n_reads, const int32_t* d_read_len, const uint8_t* d_rflags,
                                  const cramgpu_features* feats, uint8_t default_qs, uint8_t* d_scores,
                                  const int64_t* d_out_off, int32_t* d_status, void* stream);

/* =====================================================================================================
 * Data series — htsjdk!encoding/writer/Writer#write (L139-261) and htsjdk!encoding/reader/CramRecordReader#read (L59-211)
 * over flat records, htsjdk!io/ITF8#writeUnsignedITF8 (L53-84) / #readUnsignedITF8.  Call sites in the reference:
 * htsjdk!build/ContainerFactory#buildSlice L129-205 (from src/main/java/htsjdk/samtools/cram/CramSerilization.java:182)
 * and htsjdk!build/ContainerParser#getRecords (from src/main/java/net/sf/cram/Cram2Bam.java:209); the same field
 * order is spelled out in source by the reference's own readers,
 * src/main/java/htsjdk/samtools/cram/encoding/reader/ReaderToBAM.java:64-176 and ReadFeatureBuffer.java:41-100.
 * A series is addressed by its content id = htsjdk EncodingKey ordinal.  Integer series are ITF8, FC / BS / BA / QS are
 * bytes, IN / SC / RN are byte arrays closed by '\t' (ByteArrayStopEncoding).  Read names (RN) travel when the caller
 * hands over the name bytes (cramgpu_record_cols.names): for every record when capture_names is set
 * (CompressionHeader.readNamesIncluded), otherwise for the detached records only — Writer#write's two RN lines, mirrored
 * in source by ReaderToBAM.java:86-93.  Tag values (TN, TC and the per-tag series, a series set that depends on the
 * file's tag dictionary) are the host's: their slots stay empty.
 * ===================================================================================================== */
#define CRAMGPU_N_SERIES 30
enum {
    CRAMGPU_DS_BF = 0, CRAMGPU_DS_AP = 1, CRAMGPU_DS_FP = 2, CRAMGPU_DS_FC = 3, CRAMGPU_DS_QS = 4, CRAMGPU_DS_DL = 5,
    CRAMGPU_DS_BA = 6, CRAMGPU_DS_TN = 7, CRAMGPU_DS_NF = 8, CRAMGPU_DS_RL = 9, CRAMGPU_DS_RG = 10, CRAMGPU_DS_MQ = 11,
    CRAMGPU_DS_RN = 12, CRAMGPU_DS_NP = 13, CRAMGPU_DS_TS = 14, CRAMGPU_DS_FN = 15, CRAMGPU_DS_BS = 16, CRAMGPU_DS_IN = 17,
    CRAMGPU_DS_TC = 18, CRAMGPU_DS_MF = 19, CRAMGPU_DS_NS = 20, CRAMGPU_DS_CF = 21, CRAMGPU_DS_TV = 22, CRAMGPU_DS_TM = 23,
    CRAMGPU_DS_TL = 24, CRAMGPU_DS_RI = 25, CRAMGPU_DS_RS = 26, CRAMGPU_DS_SC = 27, CRAMGPU_DS_HC = 28, CRAMGPU_DS_PD = 29
};
/* CramCompressionRecord compression-flag bits and the BAM flag the writer branches on */
#define CRAMGPU_CF_FORCE_PRESERVE_QS 1
#define CRAMGPU_CF_DETACHED 2
#define CRAMGPU_CF_HAS_MATE_DOWNSTREAM 4
#define CRAMGPU_CF_UNKNOWN_BASES 8
#define CRAMGPU_BF_SEGMENT_UNMAPPED 4

/* Writer option: a container whose QS stream is exactly its records' quality arrays back to back in reads->quals (every
 * record force-preserved, no B / Q feature, a 16-byte aligned start) gets NO copy: d_series_off points at the arrays
 * where they lie: an offset relative to d_arena that leaves the arena and may be negative.  A caller that hands the
 * streams to cramgpu_rans_encode_dev passes the lower of the two addresses (d_arena, reads->quals) as its input arena
 * and rebases the offsets on it. */
#define CRAMGPU_WS_ALIAS_QS 1

/* The fields of CramCompressionRecord that cramgpu_reads / cramgpu_features do not hold (device arrays, [n_reads]).
 * The writer reads them, the reader fills them (columns a record does not carry are set to 0). */
typedef struct {
    int32_t* bam_flags;   /* BF */
    int32_t* comp_flags;  /* CF */
    int32_t* ref_id;      /* RI: written / read for containers with cont_multiref[c] != 0 */
    int32_t* read_group;  /* RG */
    int32_t* mate_flags;  /* MF (detached records) */
    int32_t* mate_ref;    /* NS */
    int32_t* mate_start;  /* NP */
    int32_t* tlen;        /* TS */
    int32_t* next_frag;   /* NF (records with a mate downstream) */
    int32_t* tag_list;    /* TL */
    const uint8_t* cont_multiref; /* [n_cont] slice sequence id is MULTI_REFERENCE, or NULL = none */
    const int32_t* cont_ap_base;  /* [n_cont] slice alignment start the first AP delta is taken from, or NULL = the
                                     first record's own start (delta 0) */
    int32_t ap_delta;             /* CompressionHeader.AP_seriesDelta */
    int32_t flags;                /* writer options: CRAMGPU_WS_* */
    const uint8_t* names;         /* writer: bytes the read names lie in (e.g. the BAM stripe), or NULL = RN stays empty */
    const int64_t* name_off;      /*   [n_reads] offset of record r's name in names */
    const int32_t* name_len;      /*   [n_reads] its length (no terminator; the stream gets the '\t') */
    int32_t capture_names;        /* CompressionHeader.readNamesIncluded: every record carries its name (writer and reader) */
    int32_t reserved;
} cramgpu_record_cols;

/* Writer#write for every record of every container of the batch.  Needs reads->bases, quals, base_off, read_len,
 * aln_start, mapq, read_cont and the feature columns after code assignment (X: a >> 16 = code).  The streams of
 * container c are laid into d_arena one after the other in content-id order, each on a 16-byte boundary (the layout the
 * rANS encoder wants: cramgpu_block.in_off / in_len can be taken straight from the two tables):
 *   d_series_off[c * CRAMGPU_N_SERIES + id], d_series_len[...]   (device)
 * d_status[0] = 0, or CRAMGPU_E_INSUFFICIENT when the streams need more than arena_cap bytes (nothing is written then;
 * d_series_len still holds the lengths), or CRAMGPU_E_ARG for an unknown operator byte.
 * Stream-ordered, does not synchronise; the context keeps one byte per container (allocated on first use). */
int32_t cramgpu_write_series_dev(cramgpu_ctx* ctx, const cramgpu_reads* reads, const cramgpu_record_cols* cols,
                                 const cramgpu_features* feats, uint8_t* d_arena, int64_t arena_cap, int64_t* d_series_off,
                                 int32_t* d_series_len, int32_t* d_status, void* stream);

/* The rANS encoder straight behind cramgpu_write_series_dev, with no host round trip between the two: the per-container
 * compress loop of ContainerFactory#buildSlice L198-201 over the series ExternalCompressor codes with rANS
 * (CompressionHeaderFactory#build L82-124), for all containers of the batch in one launch.  cramgpu_rans_encode_dev wants
 * its block table on the host, which costs a device-to-host copy of the stream lengths, the table's making and its upload
 * while the GPU idles (about 1 ms of a 13 ms step at 1000 containers); here the table is made on the device:
 *   block b = c * n_ids + k  is series ids[k] (content id) of container c, coded with order orders[k] (0 or 1);
 *   its input is d_base[in_shift + d_series_off[c * CRAMGPU_N_SERIES + ids[k]] .. + d_series_len[...]) — in_shift rebases
 *   the writer's offsets (relative to its arena; streams aliased into the caller's quality array lie outside it) on a
 *   base address below all of them; in_shift and the offsets are multiples of CRAMGPU_ARENA_ALIGN;
 *   its output slot has cramgpu_rans_bound(length) bytes, the slots follow one another on 16-byte boundaries in d_out.
 * d_blocks[n_cont * n_ids] (DEVICE, written by the call) receives the table, for the host to read afterwards together
 * with d_out_len / d_out_pos (as cramgpu_rans_encode_dev fills them).  d_status[0] = 0; or the writer's status when
 * d_w_status (may be NULL) points at a non-zero one; or CRAMGPU_E_INSUFFICIENT when the slots need more than out_cap
 * bytes — then every block is left empty (d_out_len[b] = 0).  The longest-first order of the launch is a radix sort on
 * the device.  Stream-ordered, does not synchronise. */
int32_t cramgpu_rans_encode_series_dev(cramgpu_ctx* ctx, const uint8_t* d_base, int64_t in_shift, const int64_t* d_series_off,
                                       const int32_t* d_series_len, const int32_t* d_w_status, int32_t n_cont,
                                       const int32_t* ids, const int32_t* orders, int32_t n_ids, uint8_t* d_out,
                                       int64_t out_cap, cramgpu_block* d_blocks, int32_t* d_out_len, int64_t* d_out_pos,
                                       int32_t* d_status, void* stream);

/* What the reader produces besides cramgpu_record_cols and cramgpu_features (device arrays). */
typedef struct {
    int32_t* read_len;   /* [n_reads] RL */
    int32_t* aln_start;  /* [n_reads] AP (deltas summed from cont_ap_base) */
    uint8_t* mapq;       /* [n_reads] MQ (0 = SAMRecord.NO_MAPPING_QUALITY for unmapped records, ReaderToBAM.java:157) */
    int64_t* base_off;   /* [n_reads + 1] offset of record r in bases / quals: records back to back */
    uint8_t* bases;      /* bases of unmapped records (BA); mapped records are left for cramgpu_restore_bases_dev */
    uint8_t* quals;      /* the QS array of force-preserved records, 0xFF elsewhere (what cramgpu_restore_quals_dev expects) */
    int64_t bases_cap;   /* bytes in bases / quals */
    uint8_t* seq;        /* bytes of the I / S features (IN, SC) in record order */
    int64_t* seq_off;    /* [n_reads + 1]; an I / S feature has a = offset relative to seq_off[r], b = length */
    int64_t seq_cap;
    uint8_t* rflags;     /* [n_reads] or NULL: CRAMGPU_RF_UNMAPPED / UNKNOWN_BASES / FORCE_QS from BF / CF, the per-read flags
                            cramgpu_restore_bases_dev and cramgpu_restore_quals_dev take */
    int64_t* name_off;   /* [n_reads] or NULL: where record r's name lies in the RN stream, relative to d_arena (the names stay
                            where they are, as cramgpu_bam_decode_dev leaves them in the stripe) */
    int32_t* name_len;   /* [n_reads] or NULL: its length, -1 for a record that carries no name */
} cramgpu_read_out;

/* CramRecordReader#read for every record of every container of the batch: the inverse of cramgpu_write_series_dev.
 * d_cont_first[n_cont + 1] = first record of every container (the slice headers' record counts, summed);
 * d_series_off / d_series_len as the writer lays them out (any layout will do: offsets are relative to d_arena, every
 * stream needs 16 readable bytes behind its end); series_bytes = an upper bound of the sum of all stream lengths.
 * cols: output columns, with cont_multiref / cont_ap_base (slice alignment start; NULL = 0) / ap_delta as inputs.
 * feats: op / pos / a / b / off filled (X: a = code << 16), feats->off[n_reads] = feature count.
 * d_status[0] = 0, CRAMGPU_E_CORRUPT (a stream ends early or holds values no record claims, unknown operator) or
 * CRAMGPU_E_INSUFFICIENT (bases_cap, seq_cap or feats->cap too small; base_off / off / seq_off [n_reads] hold the needs).
 * Stream-ordered, does not synchronise; the context keeps a workspace of 4 * series_bytes bytes. */
int32_t cramgpu_read_series_dev(cramgpu_ctx* ctx, int32_t n_reads, int32_t n_cont, const int32_t* d_cont_first,
                                const uint8_t* d_arena, const int64_t* d_series_off, const int32_t* d_series_len,
                                int64_t series_bytes, const cramgpu_record_cols* cols, const cramgpu_read_out* out,
                                const cramgpu_features* feats, int32_t* d_status, void* stream);

/* =====================================================================================================
 * Container assembly and BAM records: what stands between the reader's columns and the
 * BAM stripe of src/main/java/htsjdk/samtools/cram/paralell/CramToBam_OBA_Function.java:64-75 (normalize, then per record
 * Cram2SamRecordFactory#create + BAMRecordCodec#encode), and between the BAM stripe of Bam_OBA_Supplier.java:45-82 and
 * the columns K3 takes.  The BGZF members either side are cramgpu_gunzip_dev / cramgpu_gzip_dev (below).
 * ===================================================================================================== */

/* CIGAR of every mapped record from its read features — htsjdk!build/Cram2SamRecordFactory#getCigar2 (L119-225); the
 * reference tree states the algorithm in source, src/main/java/htsjdk/samtools/cram/encoding/reader/ReadFeatureBuffer.java
 * :180-294 (#getCigar) — and the record's alignment end (htsjdk!structure/CramCompressionRecord#getAlignmentEnd:
 * start + readLength - |I| - |i| - |S| + |D| - 1).  Elements are BAM words (len << 4 | op).  Unmapped records
 * (d_bam_flags & 4) get no element ("*") and end 0.
 * d_n_cigar[n_reads], d_cigar_off[n_reads + 1], d_aln_end[n_reads] are always filled; d_cigar only when the total fits
 * cigar_cap.  d_status[0] = 0 or CRAMGPU_E_INSUFFICIENT (d_cigar_off[n_reads] = the capacity needed).
 * Stream-ordered, does not synchronise. */
int32_t cramgpu_cigars_dev(cramgpu_ctx* ctx, int32_t n_reads, const int32_t* d_read_len, const int32_t* d_aln_start,
                           const int32_t* d_bam_flags, const cramgpu_features* feats, int32_t* d_n_cigar, int64_t* d_cigar_off,
                           uint32_t* d_cigar, int64_t cigar_cap, int32_t* d_aln_end, int32_t* d_status, void* stream);

/* Mate restoration of CramNormalizer#normalize — src/main/java/htsjdk/samtools/cram/build/CramNormalizer.java:82-105
 * (pairing by NF), :255-285 (restoreMateInfo / setNextMate), :455-470 (computeInsertSize) — for every container of the
 * batch.  Reads cols->bam_flags, comp_flags, next_frag, d_aln_start, d_aln_end (cramgpu_cigars_dev); writes
 * cols->mate_flags (bits 1 = mate on the negative strand, 2 = mate unmapped), mate_ref, mate_start, tlen of the chained
 * records, next_frag = -1 for records that are not multi-fragment or are detached (:85), and cols->ref_id[r] =
 * d_cont_ref_id[c] for every record of a container that is not multi-reference (so that ref_id is the record's
 * sequence id afterwards in every container).  d_status[n_cont]: 0, or CRAMGPU_E_JAVA_THROWS for a container with a
 * next-fragment distance that points outside it (records.get throws) or is negative (a self / backward link, on which
 * restoreMateInfo's walk does not end); the mate columns of its records are left as they were, its next_frag column is
 * unspecified (the reference's call ends in the exception half way through the container).
 * Stream-ordered, does not synchronise. */
int32_t cramgpu_restore_mates_dev(cramgpu_ctx* ctx, int32_t n_reads, int32_t n_cont, const int32_t* d_cont_first,
                                  const cramgpu_record_cols* cols, const int32_t* d_cont_ref_id, const int32_t* d_aln_start,
                                  const int32_t* d_aln_end, int32_t* d_status, void* stream);

/* Mate linking before a container is written — src/main/java/htsjdk/samtools/cram/CramSerilization.java:274-373
 * (#setMateInfo, with #detach :265-272 and CramNormalizer.java:455-470 #computeInsertSize), called per container from
 * #convert (:103-223) — for every container of the batch.  Records that share a read name (d_names + d_name_off[r],
 * d_name_len[r] bytes: where cramgpu_bam_decode_dev found them in the stripe) within a container and within the
 * primary / secondary (flag 0x100) map: a name seen once -> detached; two records that pass the checks of :320-346
 * (first / last segment, template sizes +-computeInsertSize, each one's mate reference / start / strand equal to the
 * other's own) -> the first gets CRAMGPU_CF_HAS_MATE_DOWNSTREAM and next_frag = the records between them, both lose
 * CRAMGPU_CF_DETACHED; otherwise, and for every record of a name seen three times or more (:294-298), detached.  A record
 * that is not multi-fragment (flag 0x1 clear) is detached (:278-284).
 * Reads cols->bam_flags, ref_id (the record's sequence id), mate_flags, mate_ref, mate_start, tlen, d_aln_start, d_aln_end
 * (cramgpu_cigars_dev's: CramCompressionRecord#getAlignmentEnd); rewrites bits 2 and 4 of cols->comp_flags and writes
 * cols->next_frag (-1 for detached records; the second record of a pair keeps its value).
 * Stream-ordered, does not synchronise; the context keeps a table of about 7 ints per record. */
int32_t cramgpu_set_mate_info_dev(cramgpu_ctx* ctx, int32_t n_reads, int32_t n_cont, const int32_t* d_cont_first,
                                  const cramgpu_record_cols* cols, const int32_t* d_aln_start, const int32_t* d_aln_end,
                                  const uint8_t* d_names, const int64_t* d_name_off, const int32_t* d_name_len, void* stream);

/* Slice content digests — htsjdk!cram/digest/ContentDigests (bytecode only) as htsjdk!build/ContainerFactory#buildSlice
 * drives it: create(ContentDigests.ALL), add(record) for every record, getAsTags() -> Slice.sliceTags.  d_crc[2 * c] = BD,
 * d_crc[2 * c + 1] = SD of container c: the sum (32-bit, wrapping: IntegerSumCombine) over its records of
 * java.util.zip.CRC32 of the record's readBases / qualityScores (Crc32Hasher; the tag value is the sum's four bytes,
 * little-endian).  The arrays are read as the caller holds them when the slice is built (quality scores after lossy
 * treatment).  The other four tags of ALL (B5 / S5 "SHA-512", B1 / S1 "SHA-1") do not depend on the content in htsjdk
 * 2.4.1: MessageDigestHasher#asByteArray returns messageDigest.digest() of the just-reset digest, the digest of the
 * empty message — constants a host writes without the device (cramtools_amd.bam.slice_content_tags).
 * Stream-ordered, does not synchronise. */
int32_t cramgpu_content_digests_dev(cramgpu_ctx* ctx, int32_t n_reads, int32_t n_cont, const int32_t* d_cont_first,
                                    const int64_t* d_base_off, const int32_t* d_read_len, const uint8_t* d_bases,
                                    const uint8_t* d_quals, uint32_t* d_crc, void* stream);

/* Slice reference MD5 — htsjdk!structure/Slice#setRefMD5 (L145-177), called per slice from
 * src/main/java/htsjdk/samtools/cram/CramSerilization.java:184-197, and the value src/main/java/net/sf/cram/Cram2Bam.java
 * :212-234 checks through src/main/java/net/sf/cram/ref/ReferenceRegion.java:116-124: MD5 (RFC 1321) of the reference bases
 * at the 1-based positions [d_start[s], d_start[s] + d_span[s]), clipped at the end of the slice's window d_wins[s]
 * (MD5 of nothing when the start lies beyond it).  d_start[s] < 1 (a slice of unplaced records): 16 zero bytes.
 * d_md5[16 * n_slices].  Stream-ordered, does not synchronise. */
int32_t cramgpu_slice_md5_dev(cramgpu_ctx* ctx, int32_t n_slices, const uint8_t* d_ref, const cramgpu_refwin* d_wins,
                              const int32_t* d_start, const int32_t* d_span, uint8_t* d_md5, void* stream);

/* The columns of one batch of records as BAMRecordCodec#encode needs them (device arrays, [n_reads] unless noted). */
typedef struct {
    const int32_t* ref_id;      /* sequence id of the record (-1 = unplaced) */
    const int32_t* aln_start;   /* 1-based */
    const int32_t* aln_end;     /* 1-based inclusive (cramgpu_cigars_dev); index bin only */
    const uint8_t* mapq;
    const int32_t* bam_flags;   /* BF; the mate-unmapped / mate-strand bits are taken from mate_flags */
    const int32_t* mate_flags;
    const int32_t* mate_ref;
    const int32_t* mate_start;
    const int32_t* tlen;
    const int32_t* read_len;
    const int64_t* base_off;    /* [n_reads + 1] */
    const uint8_t* bases;       /* ASCII, restored */
    const uint8_t* quals;       /* restored scores */
    const int32_t* qual_status; /* d_status of cramgpu_restore_quals_dev (1 = no scores: 0xFF x length), or NULL */
    const int32_t* n_cigar;
    const int64_t* cigar_off;
    const uint32_t* cigar;
    const uint8_t* names;       /* read names without terminator, or NULL: the decimal of name_base + r, the name */
    const int64_t* name_off;    /*   CramNormalizer.java:108-117 gives a record that carries none ([n_reads + 1]) */
    int64_t name_base;
    const uint8_t* tags;        /* BAM-encoded tag bytes per record, or NULL */
    const int64_t* tag_off;     /* [n_reads + 1] */
} cramgpu_bam_cols;

/* One BAM alignment record per record, back to back (block_size first) — the layout of
 * src/main/java/htsjdk/samtools/cram/encoding/reader/BAMRecordView.java:23-35, :106-119, :150-185, :229-241 with the
 * values htsjdk!build/Cram2SamRecordFactory#create sets (unplaced: start 0, mapping quality 0; unpaired: mate reference
 * -1, mate start 0) and the index bin of the SAM specification §5.3 (reg2bin; 0 for an unplaced record).
 * d_rec_off[n_reads + 1] always; bytes only when they fit bam_cap.  d_status[n_reads]: 0, CRAMGPU_E_JAVA_THROWS (a base
 * outside "=ACMGRSVTWYHKDBN." of either case: BAMRecordView.java:412 throws; a name above 254 bytes; more than 65535
 * CIGAR elements) or CRAMGPU_E_INSUFFICIENT for every record when bam_cap is too small (d_rec_off[n_reads] = need).
 * Stream-ordered, does not synchronise. */
int32_t cramgpu_bam_encode_dev(cramgpu_ctx* ctx, int32_t n_reads, const cramgpu_bam_cols* cols, uint8_t* d_bam, int64_t bam_cap,
                               int64_t* d_rec_off, int32_t* d_status, void* stream);

/* What cramgpu_bam_decode_dev fills (device arrays, [n_reads] unless noted): the columns of cramgpu_reads and
 * cramgpu_record_cols a BAM record carries. */
typedef struct {
    int32_t* ref_id;
    int32_t* aln_start;   /* 1-based (pos + 1) */
    uint8_t* mapq;
    int32_t* bam_flags;
    int32_t* mate_ref;
    int32_t* mate_start;  /* 1-based */
    int32_t* tlen;
    int32_t* read_len;
    int64_t* base_off;    /* [n_reads + 1] */
    uint8_t* bases;       /* ASCII from the 4-bit codes ("=ACMGRSVTWYHKDBN") */
    uint8_t* quals;
    int64_t bases_cap;
    int32_t* n_cigar;
    int64_t* cigar_off;   /* [n_reads + 1] */
    uint32_t* cigar;
    int64_t cigar_cap;
    int32_t* aln_span;    /* reference bases the CIGAR consumes (M D N = X; BAMRecordView.java:589-621) */
    uint8_t* rflags;      /* CRAMGPU_RF_UNMAPPED (flag 4), CRAMGPU_RF_NULL_QUALS (no scores: first byte 0xFF),
                             CRAMGPU_RF_NO_BASES (l_seq 0), CRAMGPU_RF_SKIP (malformed record) */
    int64_t* name_off;    /* where the read name lies in the stripe */
    int32_t* name_len;    /* without the terminator */
    int64_t* tag_off;     /* where the tag bytes lie in the stripe */
    int32_t* tag_len;
} cramgpu_bam_out;

/* BAMRecordCodec#decode over a stripe of BAM records (src/main/java/htsjdk/samtools/cram/paralell/Bam_OBA_Supplier.java
 * :45-82 builds it: block_size + record, up to 10 000 records of one reference).  d_rec_off[n_reads + 1] = record
 * offsets in d_bam (the supplier reads every block_size while it builds the stripe, :49-51; cramgpu_bam_index is the
 * same walk for a host that did not keep them).  d_status[n_reads]: 0, CRAMGPU_E_CORRUPT (fields overrun the record)
 * or CRAMGPU_E_INSUFFICIENT for every record when bases_cap / cigar_cap are too small (base_off / cigar_off [n_reads]
 * hold the needs).  Strings stay in the stripe: name_off / tag_off point at them.
 * Stream-ordered, does not synchronise. */
int32_t cramgpu_bam_decode_dev(cramgpu_ctx* ctx, int32_t n_reads, const uint8_t* d_bam, const int64_t* d_rec_off,
                               const cramgpu_bam_out* out, int32_t* d_status, void* stream);

/* Host: record offsets of a stripe (block sizes walked as Bam_OBA_Supplier.java:49-62 does).  rec_off[cap]; returns the
 * number of records (rec_off[n] = stripe length when n < cap) or CRAMGPU_E_CORRUPT (a block size below 32 or past the
 * end: SAMFormatException "Invalid record length"). */
int64_t cramgpu_bam_index(const uint8_t* bam, int64_t len, int64_t* rec_off, int64_t cap);

/* The same walk for a stripe that is in device memory (e.g. BGZF members inflated by cramgpu_gunzip_dev): the inflated bytes
 * need not visit the host to be cut into records.  The chain of block sizes is resolved in parallel — guessed record starts
 * per 4 KiB segment, a walk per segment, then a pass that checks every segment against the chain and repairs wrong guesses,
 * so the result is the walk's whatever the bytes look like; n_ref (reference sequences of the file; any upper bound) only
 * sharpens the guesses.  d_rec_off[cap] receives the offsets (and the stripe length behind the last one when it fits),
 * d_count[0] the number of records — compare with cap — and d_status[0] 0 or CRAMGPU_E_CORRUPT (a block size below 32 or
 * past the end; the records in front of it are listed).  Stream-ordered, does not synchronise. */
int32_t cramgpu_bam_index_dev(cramgpu_ctx* ctx, const uint8_t* d_bam, int64_t len, int32_t n_ref, int64_t* d_rec_off,
                              int64_t cap, int64_t* d_count, int32_t* d_status, void* stream);

/* =====================================================================================================
 * gunzip and gzip: htsjdk!io/ExternalCompression#gunzip (L41-48, a
 * java.util.zip.GZIPInputStream read to its end) as htsjdk!structure/Block#uncompress (L228-262) calls it for the GZIP
 * external blocks of a container, and the BGZF blocks of a BAM file (SAM specification §4.1; htsjdk!util/
 * BlockCompressedInputStream in front of Bam_OBA_Supplier).  RFC 1952 members (several in a row are read through, as
 * GZIPInputStream does; bytes behind the last member that are no header are ignored), RFC 1951 data, CRC-32 and ISIZE
 * checked.  The way back is cramgpu_gzip_dev below.
 * ===================================================================================================== */

/* Stream k = d_in[in_off, in_off + in_len) of d_blocks[k] (a DEVICE array; offsets need no alignment, `order` is ignored)
 * is inflated to d_out + out_off, at most out_cap bytes (the block header's raw size).  d_out_len[k] = bytes written, or
 * CRAMGPU_E_CORRUPT (no gzip header, invalid DEFLATE data, a stream that ends early, CRC-32 / ISIZE mismatch: the
 * ZipException / EOFException cases) or CRAMGPU_E_OUT_TOO_SMALL.  Stream-ordered, does not synchronise; uses no scratch of
 * the context, so unlike the other calls it is NOT ordered against calls on other streams: it may run beside them. */
int32_t cramgpu_gunzip_dev(cramgpu_ctx* ctx, const uint8_t* d_in, uint8_t* d_out, const cramgpu_block* d_blocks,
                           int32_t n_blocks, int32_t* d_out_len, void* stream);

/* The encode halves of the same two rows: htsjdk!io/ExternalCompression#gzip (L30-39, a java.util.zip.GZIPOutputStream)
 * for the GZIP external blocks, and the BGZF members htsjdk!util/BlockCompressedOutputStream writes behind
 * src/main/java/htsjdk/samtools/cram/paralell/CramToBam_OBA_Function.java:68-75.  Stream k = d_in[in_off, in_off + in_len)
 * of d_blocks[k] (a DEVICE array) becomes ONE gzip member at d_out + out_off, at most out_cap bytes (cramgpu_gzip_bound
 * always suffices); d_out_len[k] = its size, CRAMGPU_E_OUT_TOO_SMALL, or CRAMGPU_E_ARG for a negative in_len / out_cap.  bgzf != 0: the member carries the 'BC' extra
 * field with its own size (SAM specification 4.1) and must fit 64 KiB (else CRAMGPU_E_OUT_TOO_SMALL: cut the input into
 * smaller blocks; 65280 bytes always fit).  DEFLATE leaves matches and codes to the writer, so the bytes differ from
 * zlib's (SURVEY 8(f) row 2 puts byte identity outside the contract): the contract is that every conforming inflater —
 * zlib, GZIPInputStream, cramgpu_gunzip_dev — returns the input.  Dynamic Huffman blocks of at most 32 KiB of input over
 * a greedy single-probe LZ77 parse; a block that would not shrink is stored.  Stream-ordered, does not synchronise, uses
 * no scratch of the context (not ordered against calls on other streams). */
int32_t cramgpu_gzip_dev(cramgpu_ctx* ctx, const uint8_t* d_in, uint8_t* d_out, const cramgpu_block* d_blocks, int32_t n_blocks,
                         int32_t* d_out_len, int32_t bgzf, void* stream);

/* Host: an out_cap that cramgpu_gzip_dev never exceeds for n input bytes. */
int64_t cramgpu_gzip_bound(int64_t n);

/* Host: the BGZF blocks of a file image — off[cap] (off[n] = len when n < cap) and the uncompressed size of every block
 * (its ISIZE), from the 'BC' subfield of the headers; nothing is inflated.  Returns the number of blocks or
 * CRAMGPU_E_CORRUPT. */
int64_t cramgpu_bgzf_index(const uint8_t* data, int64_t len, int64_t* off, int32_t* usize, int64_t cap);

/* Q1 result — htsjdk!lossy/PreservationPolicy as a POD (parsed on the host by the caller, see
 * src/main/java/htsjdk/samtools/cram/lossy/QualityScorePreservation.java:167-256). */
typedef struct {
    int32_t read_cat;    /* 0 none, 1 UNPLACED, 2 HIGHER_MAPPING_SCORE, 3 LOWER_MAPPING_SCORE, 4 ALL */
    int32_t read_param;
    int32_t treatment;   /* 0 PRESERVE, 1 BIN, 2 DROP */
    int32_t n_base_cat;
    int32_t base_cat[8]; /* 1 MATCH, 2 MISMATCH, 3 FLANKING_DELETION, 4 PILEUP, 5 LOWER_COVERAGE, 6 INSERTION */
    int32_t base_param[8];
} cramgpu_policy;

/* Q1 — QualityScorePreservation.parsePolicies / parseSinglePolicy / readTreatment (:139-256): spec string ->
 * policies sorted like the reference (stable, by treatment ordinal descending).  Host function.
 * Returns the number of policies (<= max_pol) or CRAMGPU_E_ARG for an unknown category letter. */
int32_t cramgpu_parse_policies(const char* spec, cramgpu_policy* out, int32_t max_pol);

/* K5 / Q2 — QualityScorePreservation.addQualityScores :276-301, addQS :332-489, applyTreatment :263-274,
 * lossy/Binning.java:63-71, ref/ReferenceTracks.java:127-155 (baseAt semantics).  The coverage / mismatch tracks of
 * CramSerilization.updateTracks (CramSerilization.java:114-128, 225-257) are NOT built: a policy with an X<n> (n > 0)
 * or P<n> (n < 9999) term is refused with CRAMGPU_E_UNSUPPORTED, here and in cramgpu_tracks_windows; the no-op terms X0
 * and P9999 are accepted.  Stream-ordered, does not synchronise (the policies are copied first).
 * d_wpos[r] = 1-based ReferenceTracks window position in force when read r is processed (ReferenceTracks.java:87-133;
 * sequential host state, computed by cramgpu_tracks_windows); reads->win must cover the whole contig.  pol = HOST array.
 * Outputs: d_scores (0xFF = hole) at base_off[r],
 * d_force[r], d_status[r] (0 ok, 1 NULL_QUALS, 2 skipped, CRAMGPU_E_JAVA_THROWS). */
int32_t cramgpu_lossy_quals_dev(cramgpu_ctx* ctx, const cramgpu_reads* reads, const uint8_t* d_ref,
                                const int32_t* d_wpos, int32_t window, const cramgpu_policy* pol, int32_t n_pol,
                                uint8_t* d_scores, uint8_t* d_force, int32_t* d_status, void* stream);

/* ReferenceTracks.ensureRange / moveForwardTo (ReferenceTracks.java:87-133) replayed over the reads of a container in
 * order: HOST arrays in, wpos[] out.  Returns 0, CRAMGPU_E_JAVA_THROWS when the reference would throw (the read's
 * index goes to *throws_at, which may be NULL), or CRAMGPU_E_UNSUPPORTED (see cramgpu_lossy_quals_dev). */
int32_t cramgpu_tracks_windows(int32_t n_reads, const int32_t* aln_start, const int32_t* aln_span, const uint8_t* rflags,
                               const uint8_t* mapq, const cramgpu_policy* pol, int32_t n_pol, int64_t ref_len,
                               int32_t window, int32_t* wpos, int32_t* throws_at);

/* Tail of addQualityScores (:288-299): appends BaseQualityScore features for non-forced reads and stable-sorts by
 * position.  Returns the merged feature count (synchronises the stream) or a negative code. */
int64_t cramgpu_merge_quality_features_dev(cramgpu_ctx* ctx, int32_t n_reads, const int32_t* d_read_len,
                                           const int64_t* d_base_off, const uint8_t* d_scores, const uint8_t* d_force,
                                           const int32_t* d_qstatus, const cramgpu_features* in, cramgpu_features* out,
                                           void* stream);

/* =====================================================================================================
 * Level 2, host form — what a JNI shim calls: no HIP on the caller's side, asynchronous like cramgpu_rans_submit.
 * The stripe shape is the one src/main/java/htsjdk/samtools/cram/paralell/Bam_OBA_Supplier.java:45-82 cuts (at most
 * 10 000 records of one reference per container), marshalled to structure-of-arrays form by the shim inside
 * Sam2CramRecordFactory.createCramRecord / CramNormalizer.normalize (CramNormalizer.java:66,145) /
 * QualityScorePreservation.addQualityScores (QualityScorePreservation.java:276).  All arrays are HOST memory (best
 * from cramgpu_host_alloc, so that the copies overlap) and must stay valid and untouched from submit to
 * cramgpu_job_wait; jobs share the CRAMGPU_MAX_JOBS slots of the context with cramgpu_rans_submit.
 * ===================================================================================================== */
#define CRAMGPU_MAX_CONTIGS 4096

/* Reference bases of one contig (upper-cased by the host as ReferenceSource does, ReferenceSource.java:239), copied to
 * the device once and kept by the context: Sam2CramRecordFactory#setRefBases (CramSerilization.java:147-151) on
 * encode, ReferenceSource.getRegion (CramNormalizer.java:212-226) on decode.  Synchronous; replaces an earlier upload
 * of the same id and resets the contig's ReferenceTracks window to position 1. */
int32_t cramgpu_ref_upload(cramgpu_ctx* ctx, int32_t contig, const uint8_t* bases, int64_t len);
int32_t cramgpu_ref_drop(cramgpu_ctx* ctx, int32_t contig);

/* Reads of n_cont containers (encode side): the fields of cramgpu_reads a host has at hand; offsets, container
 * indices and reference windows are derived by the library. */
typedef struct {
    int32_t n_reads;
    int32_t n_cont;
    int64_t n_bases;             /* sum of read_len */
    int64_t n_cigar_ops;         /* sum of n_cigar */
    const uint8_t* bases;        /* [n_bases] read after read */
    const uint8_t* quals;        /* [n_bases] or NULL (lossless callers keep their qualities) */
    const int32_t* read_len;     /* [n_reads] */
    const int32_t* aln_start;    /* [n_reads] 1-based, 0 = unplaced */
    const uint8_t* rflags;       /* [n_reads] CRAMGPU_RF_* */
    const uint8_t* mapq;         /* [n_reads] or NULL (lossy model only) */
    const uint32_t* cigar;       /* [n_cigar_ops] BAM-packed */
    const int32_t* n_cigar;      /* [n_reads] */
    const int32_t* cont_reads;   /* [n_cont] reads per container, in order */
    const int32_t* cont_contig;  /* [n_cont] contig id (cramgpu_ref_upload) of each container */
} cramgpu_host_reads;

/* Where cramgpu_submit_reads leaves its results (host memory of the caller). */
typedef struct {
    cramgpu_features feats;      /* HOST arrays; after the wait feats.off[n_reads] = number of features */
    uint8_t* matrices;           /* [n_cont * 5] SubstitutionMatrix header bytes */
    int64_t* sub_hist;           /* [n_cont * 25] or NULL */
    uint8_t* scores;             /* lossy: [n_bases], 0xFF = hole (r.qualityScores); NULL for lossless */
    uint8_t* force;              /* lossy: [n_reads] setForcePreserveQualityScores */
    int32_t* status;             /* lossy: [n_reads] as cramgpu_lossy_quals_dev */
} cramgpu_host_read_results;

/* createCramRecord for every read of the stripe: K3 feature extraction, F3 matrices and substitution codes and, when
 * lossy_spec is not NULL, the lossy quality model (parsePolicies, the ReferenceTracks window replayed per contig
 * across submissions in submission order, addQualityScores, the merged and position-sorted BaseQualityScore
 * features).  window = ReferenceTracks window (0 = the reference's 1 000 000).  Returns a job or NULL
 * (cramgpu_last_error).  cramgpu_job_wait returns CRAMGPU_E_INSUFFICIENT when out->feats.cap is too small
 * (out->feats.off is complete then: off[n_reads] = capacity needed), CRAMGPU_E_JAVA_THROWS when ReferenceTracks would
 * throw (reads of a container not sorted by start), CRAMGPU_E_UNSUPPORTED as cramgpu_lossy_quals_dev. */
cramgpu_job* cramgpu_submit_reads(cramgpu_ctx* ctx, const cramgpu_host_reads* reads, const char* lossy_spec,
                                  int32_t window, cramgpu_host_read_results* out);

/* Records of n_cont containers on the decode side (what ContainerParser.getRecords leaves, flattened). */
typedef struct {
    int32_t n_reads;
    int32_t n_cont;
    int64_t n_seq;                   /* bytes in seq */
    const int32_t* read_len;         /* [n_reads] */
    const int32_t* aln_start;        /* [n_reads] */
    const uint8_t* rflags;           /* [n_reads] CRAMGPU_RF_UNKNOWN_BASES / FORCE_QS / SKIP */
    const int32_t* cont_reads;       /* [n_cont] */
    const int32_t* cont_contig;      /* [n_cont] */
    const int64_t* cont_win_start0;  /* [n_cont] 0-based start of the ReferenceRegion, or NULL = whole contigs */
    const int64_t* cont_win_len;     /* [n_cont] its length (clipped to the contig) */
    cramgpu_features feats;          /* HOST arrays, feats.off[n_reads] = number of features */
    const uint8_t* seq;              /* bytes of S / I features */
    const int64_t* seq_off;          /* [n_reads] */
    const uint8_t* matrices;         /* [n_cont * 5] */
    int32_t default_qs;              /* CramNormalizer.java:141,204: 30 */
    int32_t reserved;
} cramgpu_host_records;

typedef struct {
    uint8_t* bases;         /* [sum read_len] restored bases, read after read */
    uint8_t* scores;        /* in/out [sum read_len] (0xFF = missing score) or NULL to skip restoreQualityScores */
    int32_t* status_bases;  /* [n_reads] as cramgpu_restore_bases_dev */
    int32_t* status_quals;  /* [n_reads] as cramgpu_restore_quals_dev (untouched when scores is NULL) */
} cramgpu_host_restore_results;

/* CramNormalizer.restoreReadBases (:335-441) and restoreQualityScores (:287-333) for every record of the stripe. */
cramgpu_job* cramgpu_submit_restore(cramgpu_ctx* ctx, const cramgpu_host_records* recs, cramgpu_host_restore_results* out);

#ifdef __cplusplus
}
#endif
#endif /* CRAMGPU_H */
