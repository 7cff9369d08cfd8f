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

/* Re-scoring an overflow list WHILE it is filled.  The ordinary re-score launch runs behind the packed launch that fills
 * its list: on real data (the queries' own family in the DB) the few long subjects it then walks are pure tail — 9 ms for
 * one 5 500-residue subject against a 5 478-residue query, behind a 106 ms scan.  A SERVICE launch of a few workgroups,
 * started beside the packed launch (sw_set_start_signal), polls the list's length, takes entries as they appear and
 * leaves when *done_flag (a word the caller sets behind the packed launch, e.g. hipStreamWriteValue32) has reached
 * done_value; sw_rescore_overflow_claim then re-scores what the service has not taken.  Both take entries by
 * compare-and-swap, so the list must start as all -1 (hipMemsetAsync 0xFF over the packed launch's ovf_pos slice) and is
 * consumed (entries become -2).  `workgroups`: size of the service (each holds a workgroup slot for the packed launch's
 * whole duration); temp: sw_rescore_service_temp_bytes.  Otherwise as sw_rescore_overflow_stat. */
size_t sw_rescore_service_temp_bytes(sw_ctx* ctx, int kind, int32_t max_subject_len, int workgroups);
int sw_rescore_service(sw_ctx* ctx, int kind, int32_t* ovf_pos, const int32_t* ovf_count, int32_t max_count,
                       const int8_t* chars, const uint64_t* offsets, const int32_t* lengths, int32_t max_subject_len,
                       int gop, int gex, float* scores, int32_t* ids, int64_t id_offset, void* temp, size_t temp_bytes,
                       int32_t packed_limit, int32_t* true_overflow_count, const uint32_t* done_flag, uint32_t done_value,
                       int workgroups, void* stream);
int sw_rescore_overflow_claim(sw_ctx* ctx, int kind, int32_t* ovf_pos, const int32_t* ovf_count, int32_t max_count,
                              const int8_t* chars, const uint64_t* offsets, const int32_t* lengths, int32_t max_subject_len,
                              int gop, int gex, float* scores, int32_t* ids, int64_t id_offset, void* temp, size_t temp_bytes,
                              int32_t packed_limit, int32_t* true_overflow_count, void* stream);

/* 1 if a kernel on stream_b can start while a kernel on stream_a is still running, 0 if the runtime serialises the two
 * streams (it multiplexes streams onto a few hardware queues; two kernels of one queue never overlap), < 0 on error.  A
 * ~5 ms probe for callers that are about to keep a POLLING kernel on one of the streams (sw_rescore_service): behind a
 * polling kernel on the same queue, the launch it waits for would never start.  Synchronises both streams. */
int sw_streams_run_concurrently(sw_ctx* ctx, void* stream_a, void* stream_b);

/* Start handshake for launches that must run BESIDE a launch that fills the GPU (the reference gets that overlap from
 * its ten work streams, cudasw4.cuh:293,1745-1748; on this runtime a persistent grid that is dispatched first keeps every
 * workgroup slot until its end, and a small launch on another stream — the few giant subjects of partition 35 — then
 * runs BEHIND it instead of beside it, whichever stream was enqueued first).  One-shot: the NEXT sw_scan_partition /
 * sw_rescore_overflow launch of this context adds 1 to *signal (system scope) as soon as its workgroups are resident
 * (all of them up to 64; the first 64 of a larger launch).  `signal` must be signal memory
 * (hipExtMallocWithFlags(..., hipMallocSignalMemory)); the caller orders the big launch behind it with
 * hipStreamWaitValue32(stream, signal, expected, hipStreamWaitValueGte).  NULL cancels; so does a launch that fails or
 * has n == 0 (nothing is enqueued, nothing will fire: do not wait for it). */
int sw_set_start_signal(sw_ctx* ctx, uint32_t* signal);

/* Measurement aid for the roofline of this path (bench.py): a ~millis ms micro-run of the instruction mix
 * that bounds a kind's inner loop on every SIMD of the device — mix 0: v_pk_maximum3_f16 (the VOP3P issue rate of the packed
 * kinds), 1: the fp32 kind's 4 : 3.5 v_add_f32 : v_max3_f32 mix (co-issue), 2: the int32 kind's 2.25 : 3.5 v_add_u32 :
 * v_max3_i32 mix, 3: the packed kernels' OWN mix (the instruction histogram of the dominant loop body: 55 % v_pk_maximum3_f16,
 * 16 % v_pk_fma_f16, 20 % v_pk_add_f16, 9 % DPP moves / v_perm_b32 / v_add_u32 — a peak the kernel cannot beat, unlike the pure
 * VOP3P stream of mix 0, which issues fewer lane-instructions per second than the kernel itself).  *lane_instr_per_s = lane-instructions per second the device issued (waves x 64 x instructions / HIP-event
 * time): the VALU peak a kernel of that kind is priced against, at the clock the chip actually holds under that load;
 * *shader_hz = shader-clock ticks per second seen by the waves themselves (s_memtime against the 100 MHz s_memrealtime; 0 if
 * the counters do not allow it).  Runs on the null stream and synchronises. */
int sw_measure_valu_rate(sw_ctx* ctx, int mix, int millis, double* lane_instr_per_s, double* shader_hz);

/* The handshake in miniature, for callers that are about to rely on it: a one-thread kernel on side_stream
 * adds 1 to *signal and stays resident for at most 10 ms; gated_stream waits for that value
 * (hipStreamWaitValue32) and then runs a kernel the side kernel looks for.  1: the gated kernel started BESIDE the side
 * kernel — the pattern sw_set_start_signal, sw_rescore_service and sw_set_dry_signal build on works here; 0: it did not
 * (kernels serialised by a profiler or a debug setting, wait-value packets that are never released: after ~2 s the host
 * releases the wait by hand) — do not use them, a polling side kernel would hang the stream; < 0: error.  Synchronises
 * both streams; *signal is left as it was found. */
int sw_probe_handshake(sw_ctx* ctx, void* side_stream, void* gated_stream, uint32_t* signal);

/* The few VERY long subjects of a real DB (partition 35: more than 8000 residues, 35 000 in Swiss-Prot), row-parallel.
 * sw_scan_partition gives a subject to one alignment group — for these one wave —, which walks its columns one by one:
 * 35 000 dependent steps per stripe of the query, whatever else the GPU does (the reference has the same shape: one
 * thread group per subject, cudasw4.cuh:2026-2103).  Beside the bulk launch of a whole DB that is hidden; on a shard of a
 * DB (what each of N GPUs gets) it is the floor of every query, and for short queries it outlasts the bulk launch on one
 * GPU.  sw_scan_rows gives every subject of [first_pos, first_pos + n) a WORKGROUP of 1024 threads that walks the query
 * row by row, all columns of the subject at once, the horizontal gap as a max-plus prefix over the workgroup (exact for
 * gop <= gex; csrc/sw_rows_kernel.hpp).  int32 arithmetic; same scores (as floats) and ids as sw_scan_partition with a
 * 32-bit kind, no overflow list, no scratch.  max_subject_len must not exceed sw_scan_rows_max_subject() (40 960) and
 * must cover every subject of the range: the kernel never compares a length with it, columns beyond it would be dropped
 * (CUDASW4_AMD_CHECK_BOUNDS=1 verifies the contract on the device, as for sw_scan_partition).  Honours
 * sw_set_start_signal.  Errors (SW_ERR_INVALID): gop > gex, a subject bound above the limit, an armed sw_set_dry_signal
 * (these launches have no work counter that could run dry: the armed signal is cancelled and the call refused). */
int32_t sw_scan_rows_max_subject(void);
int sw_scan_rows(sw_ctx* ctx, const int8_t* chars, const uint64_t* offsets, const int32_t* lengths, int32_t first_pos,
                 int32_t n, int32_t max_subject_len, int gop, int gex, float* scores, int32_t* ids, int64_t id_offset,
                 void* stream);

/* The same subjects on MANY compute units at once (csrc/sw_rows_pipeline.hpp; round 5).  sw_scan_rows is bound by one
 * CU per subject: 4.1 us per query row for a 35 000-residue protein, whatever the GPU's size — the rank of a sharded
 * real DB that holds that protein ran at 0.60 of the full-DB rate.  Here every subject is cut into spans of 256 ... 1024
 * columns and every span is a stage of a pipeline — one wave that walks the query row by row, a few rows behind its left
 * neighbour, which hands it the row's prefix maximum and its last H as one 64-bit word through `temp` (agent-scope
 * atomics; a word is written once and read once).  ~0.3 us per query row for ANY subject length.  Same results, same
 * contract as sw_scan_rows (gop <= gex; max_subject_len covers every subject; honours sw_set_start_signal — ALL
 * workgroups count themselves in; refuses an armed dry signal), any subject length with max_subject_len * |gex| < 2^28.
 *   temp / temp_bytes  at least sw_scan_rows_pipelined_temp_bytes(ctx, n, max_subject_len) for the CURRENT query
 *                      (8 bytes x (query length + 1) x n x stages of the longest subject); overwritten by the launch.
 *   fail_count         optional device word (zeroed by the caller): += 1 for every stage that gave up waiting for its
 *                      neighbour.  Cannot happen on a healthy device (a stage only waits for a workgroup that started
 *                      before it), but every wait is bounded all the same (CUDASW4_AMD_PIPE_SPIN_LIMIT polls, default
 *                      2^20 ~ 2 s); the subject's score is then -2 and the caller must treat the scan as failed.
 *   over_limit_count / over_limit_count2 / packed_limit
 *                      optional device words: each += 1 per subject whose score is >= packed_limit — for subjects of a
 *                      partition that would otherwise run on a packed kind, the reference's overflow statistic
 *                      (half2_kernels.cuh:1087-1109; cf. sw_rescore_overflow_stat) and the caller's count of subjects
 *                      scored in 32 bits. */
/* The 32-bit re-score of the LONG subjects of an overflow list, pipelined (round 5).  A flagged subject is one alignment
 * group's walk in sw_rescore_overflow* — 16 ms for a 5 500-residue relative of a 5 478-residue query, behind the launch
 * that flagged it: on a shard of a real DB that is longer than the whole bulk launch.  This call moves the entries of the
 * list whose subject has at least min_subject_len residues (at most 64 of them) to a list of its own inside `temp`, marks
 * them taken in the original list (compare-and-swap, the protocol of sw_rescore_service / sw_rescore_overflow_claim) and
 * scores them with the stages of sw_scan_rows_pipelined (same contract: gop <= gex, max_subject_len covers the list's
 * subjects and max_subject_len * |gex| < 2^28).  Call it on the stream BEHIND the launch that filled the list and IN FRONT
 * of sw_rescore_overflow_claim, which then re-scores what is left.  true_overflow_count / packed_limit as in
 * sw_rescore_overflow_stat; fail_count as in sw_scan_rows_pipelined; temp_bytes at least
 * sw_rescore_overflow_pipelined_temp_bytes(ctx, max_subject_len) for the current query. */
size_t sw_rescore_overflow_pipelined_temp_bytes(sw_ctx* ctx, int32_t max_subject_len);
int sw_rescore_overflow_pipelined(sw_ctx* ctx, int32_t* ovf_pos, const int32_t* ovf_count, int32_t max_count, const int8_t* chars,
                                  const uint64_t* offsets, const int32_t* lengths, int32_t max_subject_len,
                                  int32_t min_subject_len, int gop, int gex, float* scores, int32_t* ids, int64_t id_offset,
                                  int32_t* fail_count, int32_t packed_limit, int32_t* true_overflow_count, void* temp,
                                  size_t temp_bytes, void* stream);

/* A pipelined launch that runs BESIDE a persistent scan launch must not leave holes behind: a SIMD's vector registers are
 * allocated as contiguous ranges, the scan launch's waves stay where they were placed to the end of the scan, and a hole
 * smaller than one of its waves at the start of the register file costs it a wave per SIMD for its whole duration
 * (measured: 8 ... 35 % of the bulk launch's rate for a pipelined launch of 2 ms).  sw_launch_vgpr_slot says how many
 * VGPRs a wave of the launch sw_scan_partition (part_id >= 0) / sw_rescore_overflow (part_id = -1) would make for the
 * current query may take (128, 168 or 256; 0: unknown); sw_set_rows_pipeline_slot (sticky) makes every stage of the
 * following sw_scan_rows_pipelined launches occupy exactly that many, so that a queued wave of the scan launch fits the
 * hole a stage leaves (0, the default: as few as the stage needs).  The stages use no LDS for the same reason. */
int sw_launch_vgpr_slot(sw_ctx* ctx, int kind, int part_id, int32_t n, int32_t max_subject_len);
int sw_set_rows_pipeline_slot(sw_ctx* ctx, int vgprs);
size_t sw_scan_rows_pipelined_temp_bytes(sw_ctx* ctx, int32_t n, int32_t max_subject_len);
int sw_scan_rows_pipelined(sw_ctx* ctx, const int8_t* chars, const uint64_t* offsets, const int32_t* lengths,
                           int32_t first_pos, int32_t n, int32_t max_subject_len, int gop, int gex, float* scores,
                           int32_t* ids, int64_t id_offset, int32_t* fail_count, int32_t* over_limit_count,
                           int32_t* over_limit_count2, int32_t packed_limit, void* temp, size_t temp_bytes, void* stream);

/* Sticky: partition 34 (1281 ... 8000 residues) runs on 16-lane groups from `subjects` subjects of a launch up and on
 * wave-wide groups below (default 512; < 0: back to the default).  Wave-wide groups finish ONE subject 3 x sooner at 60 %
 * of the throughput: a caller whose launch is short against its longest subject's walk on 16 lanes — a small shard of a
 * real DB — asks for them whatever the count (the host driver's latency mode). */
int sw_set_long16_min(sw_ctx* ctx, int32_t subjects);

/* Tail hand-over between consecutive queries.  The reference scans one query at a time (main.cu:217-260); on a small
 * shard (what each of N GPUs gets from a DB) the last, partly filled round of a query's persistent grid leaves most of
 * the GPU idle.  sw_set_dry_signal is one-shot like sw_set_start_signal: the NEXT sw_scan_partition launch of this
 * context stores `value` in *signal (system scope; signal memory) when its work counter runs dry — the first workgroup
 * finds nothing left to take while the others finish their last batches.  The caller orders the next query's bulk launch
 * (another context, stream, score array and scratch) behind that value with hipStreamWaitValue32(..., Gte), so that its
 * workgroups take the slots this launch frees one by one; launched without the gate, the two grids would share the CUs
 * for their whole duration.  Values must increase from launch to launch.  A launch that fails or has n == 0 never fires.
 * sw_set_grid_reserve (sticky): scan launches of this context leave `workgroups` of the slots the device has for their
 * kernel free (the grid is capped at resident - reserve), so that small launches of other streams find a slot while a
 * persistent grid holds the rest.  0: none (the host driver's setting: measured, the hand-over gains nothing from it). */
int sw_set_dry_signal(sw_ctx* ctx, uint32_t* signal, uint32_t value);
int sw_set_grid_reserve(sw_ctx* ctx, int32_t workgroups);

/* Long subjects against SHORT queries: exact windowing.  An alignment with a positive score of a query of Q residues spans
 * fewer than W = Q + Q * max(matrix) / min(|gop|, |gex|) + 1 subject columns (every gap column costs at least the
 * cheaper gap score, the aligned columns are worth at most Q * max(matrix)), so the DP value of any cell is already exact
 * when the recurrence starts W columns to its left with the local-alignment boundary.  A subject may therefore be cut into
 * overlapping WINDOWS — window k = columns [k * C - W, (k + 1) * C), starts on multiples of 4 — that are scanned like
 * independent subjects (sw_scan_partition on arrays of window offsets and lengths; a window is a valid subject as it is:
 * the kernels read whole 4-letter words and treat everything from the window's length on as padding), and the subject's
 * score is the maximum of its windows' scores: bit-identical to the unsplit scan, with len / C + 1 alignment groups
 * working on a 35 000-residue protein instead of one (the reference, and rounds 1-3 here, walk such a subject's 35 000
 * dependent steps with a single group: for a 48-residue query that one subject took longer than the rest of Swiss-Prot).
 *   sw_window_overlap  W for the CURRENT query and these gap scores (-1: no bound, e.g. gex == 0)
 *   sw_reduce_windows  scores[real_pos[i]] = max(win_scores[win_first[i] .. win_first[i + 1])), ids[real_pos[i]] =
 *                      id_offset + real_pos[i] for i < n_real; all pointers DEVICE. */
int32_t sw_window_overlap(sw_ctx* ctx, int gop, int gex);
int sw_reduce_windows(sw_ctx* ctx, const float* win_scores, const int32_t* win_first, const int32_t* real_pos, int32_t n_real,
                      float* scores, int32_t* ids, int64_t id_offset, void* stream);

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
