/* cudasw4_amd_engine.h — the building blocks of sw_scan_batch (cudasw4_amd.h), exported by libcudasw4_amd.so.
 *
 * A binding of the reference does not need any of these: sw_scan_batch plans and enqueues them itself.  They are declared
 * (and tested one by one: tests/test_gpu_rows*.py, test_gpu_driver.py, test_gpu_parity.py) because the engine is built from
 * them through this very interface (cudasw4_amd/csrc/sw_batch.hip uses nothing else), and for callers that want to build a
 * different orchestration.  Everything here follows the conventions of cudasw4_amd.h (device pointers, caller's stream,
 * int error codes). */
#ifndef CUDASW4_AMD_ENGINE_H
#define CUDASW4_AMD_ENGINE_H

#include "cudasw4_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

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

/* The handshake in miniature, for callers that are about to rely on it: a one-thread kernel on side_stream
 * adds 1 to *signal and stays resident for at most 10 ms; gated_stream waits for that value
 * (hipStreamWaitValue32) and then runs a kernel the side kernel looks for.  1: the gated kernel started BESIDE the side
 * kernel — the pattern sw_set_start_signal, sw_rescore_service and sw_set_dry_signal build on works here; 0: it did not
 * (kernels serialised by a profiler or a debug setting, wait-value packets that are never released: after ~2 s the host
 * releases the wait by hand) — do not use them, a polling side kernel would hang the stream; < 0: error.  Synchronises
 * both streams; *signal is left as it was found. */
int sw_probe_handshake(sw_ctx* ctx, void* side_stream, void* gated_stream, uint32_t* signal);

/* The few VERY long subjects of a real DB (partition 35: more than 8000 residues, 35 000 in Swiss-Prot), row-parallel on
 * MANY compute units at once (csrc/sw_rows_pipeline.hpp; round 5).  sw_scan_partition gives a subject to one alignment
 * group — for these one wave —, which walks its columns one by one: 35 000 dependent steps per stripe of the query,
 * whatever else the GPU does (the reference has the same shape: one thread group per subject, cudasw4.cuh:2026-2103).
 * Beside the bulk launch of a whole DB that is hidden; on a shard of a DB (what each of N GPUs gets) it is the floor of
 * every query, and for short queries it outlasts the bulk launch on one GPU.  Here every subject is cut into spans of
 * 256 ... 1024 columns and every span is a stage of a pipeline — one wave that walks the QUERY row by row, a few rows
 * behind its left neighbour, which hands it the row's prefix maximum (the horizontal gap as a max-plus prefix: exact for
 * gop <= gex) and its last H as one 64-bit word through `temp` (agent-scope atomics; a word is written once and read
 * once).  ~0.3 us per query row for ANY subject length.  int32 arithmetic; same scores (as floats) and ids as
 * sw_scan_partition with a 32-bit kind, no overflow list.  max_subject_len must cover every subject of the range (the
 * kernel never compares a length with it; CUDASW4_AMD_CHECK_BOUNDS=1 verifies the contract on the device) and
 * max_subject_len * |gex| < 2^28.  Honours sw_set_start_signal — ALL workgroups count themselves in.  Errors
 * (SW_ERR_INVALID): gop > gex, an armed sw_set_dry_signal (these launches have no work counter that could run dry: the
 * armed signal is cancelled and the call refused).  (Round 4's one-workgroup-per-subject form, sw_scan_rows, was removed
 * in round 6: the pipelines supersede it.)
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
/* Sticky: the streamed scan launches of this context (csrc/sw_stream_kernel.hpp) add to *counter (device memory) the subjects
 * they put on their overflow list only because the subject before them in the round scored at or above the zero-level jump
 * (exact scores need the 32-bit re-score, but it is a short one: callers that size a re-score service by the list's length
 * subtract them).  nullptr: not counted. */
int sw_set_dirty_counter(sw_ctx* ctx, int32_t* counter);
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

/* What sw_scan_batch WOULD launch for these arguments with the context's current query, as text (debugging a binding; the
 * CPU tests of the planner): "pipeline p35 [b,e) maxlen m; bulk kind k p33 [b,e) maxlen m list 0; side ...; service 0; split34 0".
 * No buffer of `a` is touched, nothing is enqueued. */
int sw_batch_describe_plan(sw_batch* b, const sw_batch_args* a, char* out, size_t cap);

#ifdef __cplusplus
}
#endif

#endif /* CUDASW4_AMD_ENGINE_H */
