/* cudasw4_amd_driver.h — C ABI of libcudasw4_host.so: the C++ host driver (DB sharding over the GPUs
 * of a node, residency / batch streaming, partition walk, overflow re-score, per-GPU top-K, host merge)
 * for embedding and tests.  It mirrors the public surface of the reference's class CudaSW4
 * (cudasw4.cuh:496-839) the way `align` uses it (main.cu:157-259):
 *
 *   swdrv_create       CudaSW4::CudaSW4(deviceIds, numTop, blosumType, KernelTypeConfig, MemoryConfig, verbose)
 *   swdrv_open_db      loadDB(prefix) + setDatabase            (dbdata.cpp:207-222, cudasw4.cuh:552-568)
 *   swdrv_pseudo_db    loadPseudoDB(num, length) + setDatabase (dbdata.hpp:222-272)
 *   swdrv_upload       prefetchDBToGpus                        (cudasw4.cuh:651-696)
 *   swdrv_scan         scan(query, length) -> ScanResult       (cudasw4.cuh:698-765)
 *
 * All calls return 0 on success, -1 on error (text from swdrv_last_error()).  Scores are computed only by
 * libcudasw4_amd.so; there is no CPU path.
 */
#ifndef CUDASW4_AMD_DRIVER_H
#define CUDASW4_AMD_DRIVER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct swdrv swdrv;

const char* swdrv_last_error(void);

/* devices/ndev: HIP device ids; ndev = 0 -> all visible devices.
 * matrix: 45 | 50 | 62 | 80 (the 21-letter tables) or 4525 | 5025 | 6225 | 8025 (the full 25-letter tables: the query
 * is then encoded with 25 letters, the DB stays the 21-code dbdata alphabet, see include/cudasw4_amd.h).  kinds: SW_KIND_* for single / many_small / many_large / overflow.
 * max_gpu_mem = 0 -> unlimited. */
int swdrv_create(const int* devices, int ndev, int num_top, int matrix,
                 int kind_single, int kind_many_small, int kind_many_large, int kind_overflow,
                 size_t max_gpu_mem, size_t max_batch_bytes, size_t max_batch_sequences, size_t max_temp_bytes,
                 int gop, int gex, int verbose, swdrv** out);
int swdrv_destroy(swdrv* d);

int swdrv_open_db(swdrv* d, const char* prefix, int prefetch);
int swdrv_pseudo_db(swdrv* d, size_t num, int32_t length);
/* DB from arrays already in dbdata layout (ascending length, every subject padded to a multiple of 4 with code 20);
 * the arrays are copied.  Subject i gets the header "S". */
int swdrv_db_from_arrays(swdrv* d, const int8_t* chars, size_t nchars, const uint64_t* offsets, const int32_t* lengths, size_t n);
/* One process per GPU: this driver scans shards rank*ngpu .. rank*ngpu+ngpu-1 of world*ngpu char-balanced shards of
 * every length partition (partitionDBAmongstGpus over all GPUs of the job, cudasw4.cuh:928-1004); reported ids stay
 * global (+ id_base).  Call before swdrv_open_db / swdrv_pseudo_db / swdrv_db_from_arrays. */
int swdrv_set_shard(swdrv* d, int rank, int world, int64_t id_base);
int swdrv_upload(swdrv* d);
int64_t swdrv_num_sequences(swdrv* d);
int swdrv_num_gpus(swdrv* d);
int swdrv_set_num_top(swdrv* d, int num_top);

/* query: residue letters (not encoded).  scores/ids: capacity `cap`; *nres = number of results written.
 * *num_overflows: subjects whose exact score reached the packed kind's limit (2048 / 25000) — the reference's "num
 * overflows" (main.cu:243-245); swdrv_last_rescored: how many subjects the last scan re-scored in 32 bits (>= that). */
int swdrv_scan(swdrv* d, const char* query, int32_t qlen, int32_t* scores, int64_t* ids, int cap,
               int* nres, int* num_overflows, double* seconds, double* gcups);
int swdrv_last_rescored(swdrv* d);

/* The same scan in two halves (SearchDriver::submit / collect): swdrv_scan_submit enqueues everything the query needs on
 * every GPU and returns without waiting; swdrv_scan_collect waits for the OLDEST submitted query and returns its merged
 * results.  At most two queries may be in flight, so a caller that knows its next query (align reads whole query files,
 * main.cu:157-259) submits it before collecting the current one and the GPUs never idle between queries.  Results are
 * the same as swdrv_scan's. */
int swdrv_scan_submit(swdrv* d, const char* query, int32_t qlen);
int swdrv_scan_collect(swdrv* d, int32_t* scores, int64_t* ids, int cap, int* nres, int* num_overflows, double* seconds,
                       double* gcups);
int swdrv_in_flight(swdrv* d);

/* ---- measurement / verification hooks (bench.py, tests) ----
 * Kernel events: HIP events around every DP launch on the stream it runs on.  take: 15 doubles per launch
 * (gpu index, kind, part_id, query length, subjects, cells, padded subject bytes, milliseconds, begin ms, end ms — these
 * two on the device clock since recording was switched on: launches on different streams overlap, the union of the
 * intervals is the time the DP kernels kept the GPU busy — and the kernel instantiation: kind computed in, rows per lane,
 * query stripes, lanes per group; last: 1 for an overflow re-score launch, whose cells / bytes are 0 — the length of
 * its list is only known on the device); returns the number of launches recorded since the last call
 * (may exceed cap), -1 on error. */
int swdrv_record_kernel_events(swdrv* d, int on);
int swdrv_take_kernel_events(swdrv* d, double* out, int cap);
int swdrv_shard_info(swdrv* d, int gpu, int64_t* num_local, int64_t* residues, int64_t* chars, int* resident);
/* hybrid residency (cudasw4.cuh:1044-1046,1087-1144): padded subject bytes of the GPU's shard that stay in device memory
 * (== chars when the shard is resident, 0 when all of it is streamed); -1 on error */
int64_t swdrv_cached_chars(swdrv* d, int gpu);
/* subject bytes copied host -> device by scans since swdrv_create (the one-time upload of resident / cached chars is
 * not counted): the bus traffic of streamed shards */
int64_t swdrv_streamed_bytes(swdrv* d);
/* every score of the last scan on GPU `gpu` (shard order) and the global id of every position (the
 * CUDASW_DEBUG_CHECK_CORRECTNESS view, cudasw4.cuh:728-756); num_local entries each */
int swdrv_last_scores(swdrv* d, int gpu, float* scores, int64_t* ids);
/* last streamed scan: (gpu index, begin ms, end ms) per batch, device clock relative to that GPU's scan start */
int swdrv_batch_intervals(swdrv* d, float* out, int cap);
/* last scan: (begin s, end s) per GPU on the host clock, relative to the scan's start */
int swdrv_gpu_spans(swdrv* d, double* out, int cap);
/* The launch planner (no GPU needed): runs of a length-sorted subject list, largest partition first;
 * 5 int64 per run (kind, part_id, begin, end, max length); returns the number of runs, -1 on error. */
int swdrv_plan_runs(const int32_t* sorted_lengths, size_t n, int kind_single, int kind_many_small, int kind_many_large,
                    int64_t* out, int cap);
/* the same; latency_mode != 0: the plan of the driver's latency mode (swdrv_latency_scans) — partition 34 keeps a launch
 * of its own whatever its size */
int swdrv_plan_runs_mode(const int32_t* sorted_lengths, size_t n, int kind_single, int kind_many_small, int kind_many_large,
                         int latency_mode, int64_t* out, int cap);

/* partitionDBAmongstGpus (cudasw4.cuh:928-1004) on raw arrays (no GPU needed): out[(rank*36 + partition)*2 + {0,1}] =
 * begin / end of the subject range of `rank` in `partition`. */
int swdrv_shard_ranges(const int32_t* sorted_lengths, const uint64_t* offsets, size_t n, int world, int64_t* out);

/* The residency decision of one GPU's shard (no GPU needed; GpuWorkingSet + assignBatchesToGpuMem + computeDbCopyPlan,
 * cudasw4.cuh:317-392,1087-1144,1177-1277): local_offsets[n + 1] = byte offsets of the shard's subjects in ascending
 * length, max_len its longest subject, free_mem what the device has free, the limits as in swdrv_create (0 = default).
 * -> *cache_begin: subjects [cache_begin, n) keep their chars in device memory (0: the shard is resident), *cache_bytes
 * their bytes, *batch_bytes the batch size of the streamed rest, batches[2 i], [2 i + 1] = begin / end of streamed batch
 * i; *temp_per_stream (may be NULL): what each of the 4 stripe-border scratch buffers of a GPU (work, second work, two
 * auxiliary streams) may grow to, so that cached chars + 3 staging buffers + 4 scratch buffers + 24 bytes per subject fit
 * the limit (from 4.3 GiB up; below that the 256 MiB floor per buffer wins); returns the number of batches (may exceed
 * cap), -1 on error.  allow_cache = 0: all-or-nothing residency. */
int swdrv_plan_residency(const uint64_t* local_offsets, size_t n, int32_t max_len, size_t max_gpu_mem, size_t max_batch_bytes,
                         size_t max_batch_sequences, size_t max_temp_bytes, size_t free_mem, int allow_cache,
                         int64_t* cache_begin, int64_t* cache_bytes, int64_t* batch_bytes, int64_t* batches, int cap,
                         int64_t* temp_per_stream);

/* long subjects scanned as overlapping windows (exact for short queries, include/cudasw4_amd.h: sw_window_overlap): side
 * launches that did so and windows scanned since swdrv_create */
int swdrv_window_stats(swdrv* d, int64_t* launches, int64_t* windows);
/* re-score service launches (include/cudasw4_amd.h: sw_rescore_service) since swdrv_create: the bulk launch's overflow list
 * re-scored while it was being filled.  CUDASW4_AMD_RESCORE_SERVICE=0|1 forces the service off / on; by default it runs
 * while recent scans re-scored anything. */
int64_t swdrv_service_launches(swdrv* d);
/* side launches of the longest subjects (partitions 34 / 35) that ran as pipelines of one-wave stages over many compute
 * units (sw_scan_rows_pipelined) since swdrv_create: taken when the one-wave-per-subject launch would be what the scan
 * waits for (shards of a real DB, short queries) */
int64_t swdrv_pipeline_launches(swdrv* d);
/* 1: the start handshake (sw_probe_handshake) holds on every GPU of the driver — side launches, re-score service and tail
 * hand-over are in use; 0: the driver fell back to plain stream order (a profiler that serialises kernels, ...) */
int swdrv_handshake_active(swdrv* d);
/* queries a caller with a query file should keep pending once a query of this length has been submitted: 1, or 2 where the tail
 * hand-over applies (up to SearchDriver::kMaxInFlight = 4 may be pending; more than two was measured and is slower) */
int swdrv_preferred_in_flight(swdrv* d, int32_t query_length);
/* scans planned in LATENCY MODE since swdrv_create: partition 34 (1281 ... 8000 residues) on wave-wide groups beside the
 * bulk launch instead of inside it, when the whole launch is short against the walk of its longest subject on 16 lanes
 * (small shards of real DBs).  CUDASW4_AMD_LATENCY_MODE=never|always overrides the estimate. */
int64_t swdrv_latency_scans(swdrv* d);
/* Tail hand-over between two queries in flight (include/cudasw4_amd.h: sw_set_dry_signal): a query submitted with
 * swdrv_submit while the one before is still pending runs on a second lane of the GPU (context, work stream, score arrays)
 * and its bulk launch starts when the earlier one's work counter runs dry, filling the slots its last round leaves idle —
 * on resident shards of a few rounds of workgroups (what each of N GPUs gets from a small DB) and for queries that are
 * scanned in a few milliseconds; results are those of swdrv_scan.  Returns the queries gated that way since swdrv_create.  CUDASW4_AMD_TAIL_OVERLAP=0 turns the hand-over
 * off, =1 lifts the shard-size rule. */
int64_t swdrv_tail_overlaps(swdrv* d);
/* 1 when the hand-over applies — the loaded DB's shards are small (at most 20 rounds of workgroups: 491 520 subjects on 256 CUs), or a query of
 * query_length residues (0: not considered) is scanned in a few milliseconds (<= 8 ms at 10 TCUPS: short queries, whatever
 * the DB) —: a caller with its next query at hand should then swdrv_submit it before it collects the current one (`align`
 * and bench.py do) */
int swdrv_prefers_two_in_flight(swdrv* d, int32_t query_length);

/* NUMA placement: the node of the gpu-th GPU's PCI function (-1: unknown) and its HIP device ordinal.  In-process
 * multi-GPU drivers run each GPU's worker thread on that node themselves; a one-process-per-GPU caller binds its own
 * thread with swdrv_bind_to_numa_node (0: bound; -1: unknown node or none of its CPUs allowed, affinity unchanged). */
int swdrv_numa_node(swdrv* d, int gpu);
int swdrv_device_of(swdrv* d, int gpu);
int swdrv_device_numa_node(int device);   /* the same for a HIP device ordinal, before any driver exists */
int swdrv_bind_to_numa_node(int node);

/* header / length of a subject by global id (getReferenceHeader / getReferenceLength) */
int32_t swdrv_reference_length(swdrv* d, int64_t id);
int swdrv_reference_header(swdrv* d, int64_t id, char* buf, int cap);

/* ---- input helpers (no GPU needed): what `align` does to its inputs before the scan ---- */

/* ConvertAA_20 (convert.cuh:6-34): letters -> codes 0..20 */
void swdrv_encode(const char* letters, int8_t* codes, size_t n);
/* the pseudo-DB subject (dbdata.hpp:222-272): `length` codes from std::mt19937(seed) + uniform_int_distribution<>(0,19) */
void swdrv_pseudo_sequence(int32_t length, int seed, int8_t* codes);
/* 21 x 21 substitution matrix (types.hpp:29-270); matrix = 45 | 50 | 62 | 80; returns 0 or -1 */
int swdrv_matrix(int matrix, int8_t* out441);
/* 25 x 25 table (types.hpp:205-396), letter order ARNDCQEGHILKMFPSTWYVBJZX*, and its query encoder (anything else -> X) */
int swdrv_matrix25(int matrix, int8_t* out625);
void swdrv_encode25(const char* letters, int8_t* codes, size_t n);
/* FASTA / FASTQ (.gz) reader (kseqpp/kseqpp.hpp:54-118): open -> next* -> close.  next returns 1 while there is a
 * record; the header / sequence pointers stay valid until the following call. */
typedef struct swdrv_reader swdrv_reader;
int swdrv_reader_open(const char* path, swdrv_reader** out);
int swdrv_reader_next(swdrv_reader* r, const char** header, size_t* header_len, const char** sequence, size_t* sequence_len);
void swdrv_reader_close(swdrv_reader* r);

#ifdef __cplusplus
}
#endif
#endif
