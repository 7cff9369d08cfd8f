/* cudasw4_amd.h — C ABI of libcudasw4_amd.so: the MI355X (gfx950) Smith-Waterman DP hot path.
 *
 * Drop-in boundary for the kernel layer of CUDASW++4.0 (reference files cited per entry point;
 * paths relative to the reference's src/).  The reference boundary is a C++ template API
 * (kernels.cuh:31-164 — call_NW_local_affine_{single,multi}_pass_{half2,dpx_s16,dpx_s32,float} and
 * the two overflow launchers) plus setProgramWideBlosum (blosum.hpp:26).  This library exposes the
 * same operations with plain pointers and sizes:
 *
 *   - every pointer marked DEVICE is a device pointer on the context's device;
 *   - the caller owns every buffer, nothing is allocated inside the scan entry points
 *     (the context owns only the substitution matrix and the per-query profile);
 *   - all work is enqueued asynchronously on `stream` (a hipStream_t passed as void*);
 *   - return value: 0 on success, a negative SW_ERR_* otherwise; nothing throws across the ABI;
 *   - one host thread per context; different contexts (devices) may be driven from one thread
 *     in turn, exactly like the reference's cudaSetDevice loops (cudasw4.cuh:1509-1524).
 *
 * Scores are the reference's: affine-gap local alignment (H = max(0, diag+M, E, F)), score only,
 * written as float next to the subject's global id (util.cuh:159-192 BatchResultList).
 */
#ifndef CUDASW4_AMD_H
#define CUDASW4_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Arithmetic kinds == the reference's KernelType (types.hpp:11-16), same order. */
enum {
    SW_KIND_F16X2 = 0, /* Half2  : two subjects per lane group, packed fp16, exact below 2048  (half2_kernels.cuh)   */
    SW_KIND_I16X2 = 1, /* DPXs16 : two subjects per lane group, packed int16, exact below 25000 (dpx_s16_kernels.cuh) */
    SW_KIND_I32   = 2, /* DPXs32 : one subject per lane group, int32 results; computed in fp32 lanes (30 % faster on gfx950)
                          whenever min(query, subject) * max(matrix) + 2^22 < 2^24 proves that exact (dpx_s32_kernels.cuh) */
    SW_KIND_F32   = 3  /* Float  : one subject per lane group, fp32, exact below 2^24           (float_kernels.cuh)   */
};

enum {
    SW_OK = 0,
    SW_ERR_INVALID = -1,   /* bad argument (null pointer, unknown kind, positive gap score, ...) */
    SW_ERR_HIP = -2,       /* a HIP runtime call failed; sw_last_error() has the text */
    SW_ERR_NO_QUERY = -3,  /* scan before sw_set_query */
    SW_ERR_NO_MATRIX = -4, /* scan before sw_set_matrix */
    SW_ERR_TEMP = -5,      /* temp buffer too small for this query/partition (see sw_scan_temp_bytes) */
    SW_ERR_NO_DEVICE = -6  /* no usable HIP device: the library never falls back to a CPU path */
};

/* kernels.cuh:4-5 — MAX_ACC_HALF2 / MAX_ACC_SHORT: a packed score at or above these is an overflow. */
#define SW_MAX_ACC_F16 2048
#define SW_MAX_ACC_I16 25000
/* Longest subject a launch accepts.  max_subject_len / part_maxlen are the longest subject of the range (the DB is sorted by
 * length: lengths[last of the range]), NOT a partition's nominal boundary — the last partition's boundary is INT_MAX in the
 * reference (length_partitions.hpp:13-60) and is refused here with SW_ERR_INVALID. */
#define SW_MAX_SUBJECT_LEN (1 << 28)

/* length_partitions.hpp:75-113 */
#define SW_NUM_LENGTH_PARTITIONS 36

typedef struct sw_ctx sw_ctx;

/* Library/version probe that needs no GPU. */
const char* sw_version(void);
/* Text of the last error on the calling thread (never NULL). */
const char* sw_last_error(void);
/* hipGetDeviceCount; 0 when there is no GPU. */
int sw_device_count(void);

/* Per-device context.  Replaces the reference's per-GPU program-wide state
 * (__constant__ deviceBlosum, blosum.cu:9-11; d_query, cudasw4.cuh:305-306). */
int sw_ctx_create(int device, sw_ctx** out);
int sw_ctx_destroy(sw_ctx* ctx);

/* setProgramWideBlosum (blosum.hpp:26, blosum.cu:21-119): install a dim x dim substitution matrix (HOST pointer,
 * row-major int8, rows = query letters, columns = subject letters).
 *   dim 21: 20 amino acids + "other" (types.hpp:29-270).  The last row/column must be negative: padding is scored
 *           with it (half2_kernels.cuh:251-257, cudasw4.cuh:1298).
 *   dim 25: the full tables, letter order ARNDCQEGHILKMFPSTWYVBJZX* (types.hpp:205-396; the reference's
 *           CAN_USE_FULL_BLOSUM build, options.cpp:135-143).  QUERY codes are then 0..24.  SUBJECT codes stay the
 *           dbdata alphabet 0..20 — makedb encodes every DB with ConvertAA_20 (makedb.cpp:171,361), so B, J, Z, X and
 *           '*' of a subject are all code 20 — and code 20 is scored with the table's X column (every entry of which
 *           is negative, so it still neutralises padding; the column is checked).  The reference's own full build
 *           would index the table with the raw code 20, i.e. score unknown subject letters and all padding as 'B'. */
int sw_set_matrix(sw_ctx* ctx, const int8_t* matrix_host, int dim);

/* CudaSW4::setQuery (cudasw4.cuh:1280-1310): install the encoded query (HOST pointer, codes
 * 0..dim-1).  Builds the device-side query profile the kernels read (lazily, per kind).  The codes are staged in a
 * pinned buffer of the context: the caller's buffer is free again on return, the upload is enqueued on `stream` behind
 * the scans of the previous query, and the call does not wait for the GPU. */
int sw_set_query(sw_ctx* ctx, const int8_t* query_codes_host, int32_t qlen, void* stream);

/* Bytes of temp memory the sw_scan_partition call with the same (kind, part_id, n, max_subject_len) needs
 * for the CURRENT query (0 when the query fits one stripe).  sw_rescore_overflow: pass part_id = -1 and
 * n = max_count.  Replaces the reference's tempBytesPerBlockPerBuffer / tempBytesPerSubjectPerBuffer
 * sizing (cudasw4.cuh:1928-1938,2028-2033).  Launches that run concurrently on different streams need
 * separate temp buffers. */
size_t sw_scan_temp_bytes(sw_ctx* ctx, int kind, int part_id, int32_t n, int32_t max_subject_len);

/* One length partition of one batch: replaces call_NW_local_affine_{single,multi}_pass_*
 * (kernels.cuh:31-164; dispatch cudasw4.cuh:1764-1912,1920-2096).
 *
 *   kind            SW_KIND_*
 *   part_id         reference length-partition index 0..35.  Any subject length works with any id; ids 34/35
 *                   (the long partitions) select the wave-wide group shape when n is small, so that a few
 *                   giant subjects do not become the tail of the scan; for ids 0..33 the library chooses 8- or
 *                   16-lane groups from the query length and max_subject_len
 *   chars           DEVICE int8 codes, each subject padded to a multiple of 4 (dbdata layout)
 *   offsets         DEVICE uint64[>= first_pos+n+1]; offsets[i]-offsets[0] = byte offset of subject i
 *   lengths         DEVICE int32 true lengths
 *   first_pos, n    subjects first_pos .. first_pos+n-1 (batch-local positions, the reference's
 *                   counting PositionsIterator, kernels.cuh:28)
 *   max_subject_len upper bound of lengths[first_pos .. first_pos+n).  CONTRACT: it sizes the stripe-border scratch of
 *                   multi-stripe queries (sw_scan_temp_bytes) and the kernels never walk past that scratch, so an
 *                   under-reported bound is memory-safe but SILENTLY truncates longer subjects at the bound (their
 *                   scores are then those of the truncated subject).  Pass the true maximum (the reference passes the
 *                   partition's boundary, cudasw4.cuh:1767-1912); over-reporting only costs scratch.
 *                   Debugging a binding: with CUDASW4_AMD_CHECK_BOUNDS=1 in the environment of sw_ctx_create every scan
 *                   and re-score first finds the longest subject of its range on the device (one small kernel and a
 *                   stream synchronisation) and returns SW_ERR_INVALID when the bound under-reports it.
 *   gop, gex        gap open / extend scores, both <= 0 (reference: -11 / -1)
 *   scores, ids     DEVICE, indexed by position: scores[pos] = score, ids[pos] = id_offset + pos
 *   ovf_pos/count   DEVICE; when ovf_check != 0 a subject whose packed score reaches the kind's
 *                   limit is appended to ovf_pos[atomicAdd(ovf_count,1)] and its score is left
 *                   untouched (half2_kernels.cuh:1087-1109).  Ignored for the 32-bit kinds.
 *                   The list may hold a few subjects more than strictly overflowed: the kernels keep
 *                   column j's values raised by |gex|*(j+16) and flag a subject as soon as the bound
 *                   score + |gex|*(min(columns,K)+36) reaches the limit (never one scoring below limit-1536
 *                   for F16X2 / limit-2100 for I16X2 with gex = -1).  Re-scoring them all keeps every score exact.
 *   temp            DEVICE scratch of at least sw_scan_temp_bytes() (may be NULL when that is 0)
 */
int sw_scan_partition(sw_ctx* ctx, int kind, int part_id,
                      const int8_t* chars, const uint64_t* offsets, const int32_t* lengths,
                      int32_t first_pos, int32_t n, int32_t max_subject_len,
                      int gop, int gex,
                      float* scores, int32_t* ids, int64_t id_offset,
                      int32_t* ovf_pos, int32_t* ovf_count, int ovf_check,
                      void* temp, size_t temp_bytes, void* stream);

/* launch_process_overflow_alignments_kernel_NW_local_affine_multi_pass_{float,dpx_s32}
 * (float_kernels.cuh:1189-1318, dpx_s32_kernels.cuh:1182-1290; call site cudasw4.cuh:2134-2169):
 * re-score the subjects listed in ovf_pos[0 .. *ovf_count) with a 32-bit kind.  The count is read
 * on the device; no host round trip and no device-side launch. */
int sw_rescore_overflow(sw_ctx* ctx, int kind /* SW_KIND_I32 | SW_KIND_F32 */,
                        const int32_t* ovf_pos, const int32_t* ovf_count, int32_t max_count,
                        const int8_t* chars, const uint64_t* offsets, const int32_t* lengths,
                        int32_t max_subject_len, int gop, int gex,
                        float* scores, int32_t* ids, int64_t id_offset,
                        void* temp, size_t temp_bytes, void* stream);

/* The same, and additionally *true_overflow_count += the number of re-scored subjects whose exact score is >=
 * packed_limit (SW_MAX_ACC_F16 / SW_MAX_ACC_I16 of the kind that flagged them): the reference's "num overflows"
 * statistic (main.cu:243-245) counts exactly those, whereas the overflow list may hold a few subjects more (see
 * sw_scan_partition).  true_overflow_count: DEVICE int32, may be NULL. */
int sw_rescore_overflow_stat(sw_ctx* ctx, int kind,
                             const int32_t* ovf_pos, const int32_t* ovf_count, int32_t max_count,
                             const int8_t* chars, const uint64_t* offsets, const int32_t* lengths,
                             int32_t max_subject_len, int gop, int gex,
                             float* scores, int32_t* ids, int64_t id_offset,
                             void* temp, size_t temp_bytes,
                             int32_t packed_limit, int32_t* true_overflow_count, void* stream);

/* 1 if a kernel on stream_b can start while a kernel on stream_a is still running, 0 if the runtime serialises the two
 * streams (it multiplexes streams onto a few hardware queues; two kernels of one queue never overlap), < 0 on error.  A
 * ~5 ms probe for callers that are about to keep a POLLING kernel on one of the streams (sw_rescore_service): behind a
 * polling kernel on the same queue, the launch it waits for would never start.  Synchronises both streams. */
int sw_streams_run_concurrently(sw_ctx* ctx, void* stream_a, void* stream_b);

/* Measurement aid for the roofline of this path (bench.py): a ~millis ms micro-run of the instruction mix
 * that bounds a kind's inner loop on every SIMD of the device — mix 0: v_pk_maximum3_f16 (the VOP3P issue rate of the packed
 * kinds), 1: the fp32 kind's 4 : 3.5 v_add_f32 : v_max3_f32 mix (co-issue), 2: the int32 kind's 2.25 : 3.5 v_add_u32 :
 * v_max3_i32 mix, 3: the packed kernels' OWN mix (the instruction histogram of the dominant loop body: 55 % v_pk_maximum3_f16,
 * 16 % v_pk_fma_f16, 20 % v_pk_add_f16, 9 % DPP moves / v_perm_b32 / v_add_u32 — a peak the kernel cannot beat, unlike the pure
 * VOP3P stream of mix 0, which issues fewer lane-instructions per second than the kernel itself), 4: mix 3 with every
 * instruction's three sources in three different register banks (explicit registers: no operand-fetch conflict cycles — the
 * issue ceiling proper).  *lane_instr_per_s = lane-instructions per second the device issued (waves x 64 x instructions / HIP-event
 * time): the VALU peak a kernel of that kind is priced against, at the clock the chip actually holds under that load;
 * *shader_hz = shader-clock ticks per second seen by the waves themselves (s_memtime against the 100 MHz s_memrealtime; 0 if
 * the counters do not allow it).  Runs on the null stream and synchronises. */
int sw_measure_valu_rate(sw_ctx* ctx, int mix, int millis, double* lane_instr_per_s, double* shader_hz);

/* The building blocks sw_scan_batch (below) is made of — side-launch handshake, row pipelines for the giants, windows, the
 * re-score service and its claim protocol, the tail hand-over signals — are declared in cudasw4_amd_engine.h: exported, tested
 * one by one, but nothing a binding needs (round 5 had them all here: 45 exports for a boundary of 6 calls). */

/* ------------------------------------------------------------------------------------------------------------------
 * ONE call per batch: the whole body of the reference's runAlignmentKernels (cudasw4.cuh:1742-2103) and its overflow
 * block (cudasw4.cuh:2134-2172).  sw_scan_partition / sw_rescore_overflow above are the launchers the reference calls
 * one by one; a caller that walks the 36 partitions with them, one launch behind the other on one stream (INTEGRATION.md
 * section 2, first form), pays a launch tail per partition and leaves everything that makes real DBs fast to itself:
 * merging the partitions of one arithmetic kind into one persistent grid, running the few long subjects BESIDE that grid
 * (side streams + the start handshake), cutting the giants into pipelines of one-wave stages or into windows, re-scoring
 * the overflow lists while they are filled.  sw_scan_batch does all of that behind the boundary:
 *
 *   sw_batch_create(ctx, work_stream, &b)   once per context: side streams, signal memory, probes (bounded)
 *   sw_set_query(ctx, ...)                  per query, as before
 *   sw_scan_batch(b, &args)                 per batch of the DB (a resident DB is one batch): plans and enqueues every
 *                                           launch of the batch; returns without waiting for the device
 *   sw_batch_join(b, stream)                `stream` waits for everything the batches so far put on side streams
 *   sw_topk(...)
 *
 * The engine owns its side streams and scratch buffers (grown on demand, each at most max_temp_bytes); nothing in it is
 * shared between contexts.  Round 6 moved this orchestration out of the host driver (cudasw4_amd/csrc/host), which now is
 * a caller of sw_scan_batch like any other binding. */
typedef struct sw_batch sw_batch;

/* counters a batch leaves in device memory (sw_batch_args::counters, zeroed by the CALLER — one memset per query covers
 * the blocks of all its batches): */
#define SW_BATCH_COUNTERS 8
#define SW_BATCH_CNT_OVERFLOWS 0   /* subjects whose exact score reached their packed kind's limit (the reference's statistic) */
#define SW_BATCH_CNT_LIST0 1       /* .. LIST0 + 3: lengths of the batch's overflow lists = subjects re-scored in 32 bits */
#define SW_BATCH_CNT_FAILED 5      /* pipeline stages that gave up waiting (sw_scan_rows_pipelined): the scan failed if != 0 */
#define SW_BATCH_CNT_PIPE_OVER 6   /* pipelined subjects of packed partitions at or above the packed limit (scored in 32 bits) */
#define SW_BATCH_CNT_DIRTY 7       /* of the listed subjects: those re-scored only because the subject streamed through the lanes right before them scored
                                    * at or above the zero-level jump (csrc/sw_stream_kernel.hpp); short ones, cheap to re-score */

/* one timed launch (optional: sw_batch_args::records) */
typedef struct sw_launch_record {
    void* ev0; void* ev1;            /* hipEvent_t pair supplied by the caller, recorded around the launch on its stream */
    int32_t kind, part_id;           /* requested kind; largest partition of the run (-1: re-score) */
    int32_t eff_kind, rows, nstripes, lanes;   /* the instantiation the library chose (sw_plan_launch) */
    int32_t begin, end;              /* batch-local subject range of the run */
    int32_t rescore;
} sw_launch_record;

typedef struct sw_batch_args {
    int kinds[4];                    /* KernelTypeConfig (cudasw4.cuh:88-93): singlePass, manyPass_small, manyPass_large, overflow */
    const int8_t* chars;             /* DEVICE: the batch's dbdata arrays (as for sw_scan_partition) */
    const uint64_t* offsets;
    const int32_t* lengths;
    int32_t n;                       /* subjects of the batch, sorted by length (dbdata order) */
    const int32_t* part_begin;       /* HOST, SW_NUM_LENGTH_PARTITIONS + 1: first batch-local position of every length partition */
    const int32_t* part_maxlen;      /* HOST, SW_NUM_LENGTH_PARTITIONS: longest subject of every partition (empty: anything) */
    const int32_t* long_lengths;     /* HOST, optional: true lengths of the positions part_begin[34] .. n - 1 (partitions 34 / 35): lets
                                        the engine pick the subjects whose lone walk would outlast the bulk launch (pipelines, windows);
                                        NULL: whole partitions by their bounds */
    const uint64_t* long_offsets;    /* HOST, optional, with long_lengths: byte offsets of the same positions; minus long_offsets_bias they are
                                        relative to `chars` (windows of long subjects for short queries) */
    uint64_t long_offsets_bias;
    uint64_t batch_bytes;            /* padded subject bytes of the batch (planning estimates only) */
    int gop, gex;
    float* scores; int32_t* ids; int64_t id_offset;   /* DEVICE, indexed by batch-local position */
    int32_t* ovf_pos;                /* DEVICE, n entries: the overflow lists of the batch's packed launches live in slices of it */
    int32_t* counters;               /* DEVICE, SW_BATCH_COUNTERS ints, zeroed before the call (stream-ordered) ... */
    int zero_counters;               /* ... or by the call itself (1: one memset on `stream`; a caller with several batches per query zeroes all
                                        their blocks with one memset of its own and passes 0) */
    size_t max_temp_bytes;           /* cap of each of the engine's scratch buffers (0: 4 GiB) */
    void* stream;                    /* the work stream: bulk launch, its re-score; side streams fork from it */
    int work_slot;                   /* 0 / 1: which of the engine's two work-stream scratch buffers (batches that overlap on two work streams) */
    int allow_service;               /* re-score service beside the bulk launch allowed (resident chars, nothing else of the caller polls) */
    /* tail hand-over between two queries in flight (sw_set_dry_signal): the bulk launch arms arm_signal with arm_value,
     * and waits for wait_signal >= wait_value before it starts (NULL / 0: none) */
    uint32_t* arm_signal; uint32_t arm_value;
    uint32_t* wait_signal; uint32_t wait_value;
    int32_t grid_reserve_side;       /* workgroup slots the bulk grid leaves free when the batch has side work (two queries in flight) */
    int alt_side_stream;             /* start the side launches on the engine's second auxiliary stream (consecutive queries alternate) */
    sw_launch_record* records; int32_t records_cap; int32_t* records_used;   /* optional: event pairs + what ran between them */
    int record_mode;                 /* 0: none, 1: every launch, 2: work-stream launches only */
} sw_batch_args;

int sw_batch_create(sw_ctx* ctx, void* work_stream, sw_batch** out);
int sw_batch_destroy(sw_batch* b);
int sw_scan_batch(sw_batch* b, const sw_batch_args* args);
/* `stream` waits for the side streams' work of every batch since the last join */
int sw_batch_join(sw_batch* b, void* stream);
/* record `events[i]` (hipEvent_t, i = 0..2: the two auxiliary streams, the service stream) behind the last side launch of
 * the batch just enqueued on that stream; used[i] = 1 where the batch put work there (a staging buffer the batch read may
 * be overwritten only after those events) */
int sw_batch_side_events(sw_batch* b, void* const* events, int* used);
/* after a query's counters came back: how many subjects it re-scored because of their OWN score (the lists' lengths minus
 * SW_BATCH_CNT_DIRTY): sizes and arms the re-score service of later scans */
int sw_batch_feedback(sw_batch* b, int32_t rescored);
/* 1: the start handshake passed its probe (side launches run beside the bulk grid by construction); 0: plain stream order */
int sw_batch_handshake_active(const sw_batch* b);
/* statistics since creation: out[0] pipelined launches, [1] pipelined re-scores, [2] window launches, [3] windows scanned,
 * [4] service launches, [5] side launches */
int sw_batch_stats(const sw_batch* b, int64_t* out, int n);
/* the host-visible signal words of the engine (watchdog of a caller that polls with a deadline): values and targets;
 * sw_batch_open_gates releases every stream wait of the engine by hand */
int sw_batch_signal_state(const sw_batch* b, uint32_t* start_now, uint32_t* start_target, uint32_t* done_now, uint32_t* done_target);
int sw_batch_open_gates(sw_batch* b);
/* after a failed scan and a device synchronisation: nothing of the engine is in flight, its counts start afresh */
int sw_batch_reset(sw_batch* b);
/* test hook: the n-th side launch from now on is counted but never enqueued (0: off) */
int sw_batch_test_lose_side_launch(sw_batch* b, int nth);
int32_t sw_query_length(const sw_ctx* ctx);

/* Per-GPU top-K (cudasw4.cuh:1357-1401): the k best (score desc, id asc on ties) of n results.
 * out_scores/out_ids: DEVICE, k entries, padded with (-1, -1) when n < k.
 * temp: DEVICE scratch of sw_topk_temp_bytes(n, k). */
size_t sw_topk_temp_bytes(int64_t n, int k);
int sw_topk(sw_ctx* ctx, const float* scores, const int32_t* ids, int64_t n, int k,
            float* out_scores, int32_t* out_ids, void* temp, size_t temp_bytes, void* stream);

/* Guard for callers that upload subject chars they have not validated (a memory-mapped DB too large to read at load):
 * *bad_flag (DEVICE int32, zeroed by the caller) is set to 1 when any of chars[0 .. n) (DEVICE) is not a letter code
 * 0..20.  The scan kernels turn every letter byte into an LDS row offset (the reference indexes its shared-memory score
 * table the same way, half2_kernels.cuh:243-261), so a foreign or corrupt DB yields garbage scores: check, then refuse. */
int sw_check_letter_codes(sw_ctx* ctx, const int8_t* chars, size_t n, int32_t* bad_flag, void* stream);

/* Introspection for measurement: which kernel instantiation sw_scan_partition (part_id >= 0) or sw_rescore_overflow
 * (part_id = -1) would launch for the CURRENT query: the arithmetic kind actually computed in (SW_KIND_I32 may be served
 * in fp32 lanes, see SW_KIND_I32), rows per lane, query stripes and lanes per alignment group (4 | 8 | 16 | 64) — the
 * template arguments of swk::sw_scan_kernel as rocprofv3 prints them.  Any output pointer may be NULL. */
int sw_plan_launch(sw_ctx* ctx, int kind, int part_id, int32_t n, int32_t max_subject_len,
                   int32_t* effective_kind, int32_t* rows_per_lane, int32_t* nstripes, int32_t* lanes);

/* Introspection for tests / tuning: rows per lane and number of query stripes chosen for qlen. */
int sw_plan_query(int kind, int32_t qlen, int32_t* rows_per_lane, int32_t* nstripes);

#ifdef __cplusplus
}
#endif
#endif /* CUDASW4_AMD_H */
