// search_driver.hpp — the C++ host driver: DB sharding over the GPUs of a node, residency, batch
// streaming, the partition walk, overflow re-score, per-GPU top-K and the host-side merge.
//
// Mirrors the public surface of the reference's class CudaSW4 (cudasw4.cuh:496-839):
//   SearchDriver(deviceIds, numTop, matrix, KernelTypeConfig, MemoryConfig, verbose)
//   setDatabase / prefetchDBToGpus / scan / totalTimerStart / totalTimerStop / printDBInfo ...
// but drives the GPUs only through the C ABI of include/cudasw4_amd.h.  With one GPU the calling thread
// issues everything; with several, one host worker thread per GPU runs that GPU's whole scan (query upload,
// batches, re-score, top-K, copy back), so streamed shards progress on all GPUs at once — the role of the
// reference's host-callback staging threads (cudasw4.cuh:1649-1658, dbbatching.cuh:255-276).  Results are merged
// on the host instead of peer copies to GPU 0 (cudasw4.cuh:1415-1463).
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <string_view>
#include <utility>
#include <vector>

#include "db_format.hpp"
#include "sequence_codec.hpp"

struct sw_ctx;

namespace swh {

// types.hpp:11-16, same numbering as SW_KIND_*
enum class KernelType { Half2 = 0, DPXs16 = 1, DPXs32 = 2, Float = 3 };
const char* to_string(KernelType t);
bool parse_kernel_type(const std::string& s, KernelType& out);

struct KernelTypeConfig {  // cudasw4.cuh:88-93
    KernelType singlePassType = KernelType::Half2;
    KernelType manyPassType_small = KernelType::Half2;
    KernelType manyPassType_large = KernelType::Float;
    KernelType overflowType = KernelType::Float;
    bool valid(std::string* why = nullptr) const;  // cudasw4.cuh:841-855
    KernelType for_partition(int part_id) const;
};

struct MemoryConfig {  // cudasw4.cuh:95-100, options.hpp:34-39
    size_t maxBatchBytes = size_t(128) << 20;
    size_t maxBatchSequences = 10'000'000;
    size_t maxTempBytes = size_t(4) << 30;
    size_t maxGpuMem = SIZE_MAX;
};

struct BenchmarkStats {  // cudasw4.cuh:75-80
    int numOverflows = 0;  // subjects whose exact score reached the packed kind's limit: what the reference counts and prints
    int numRescored = 0;   // subjects the packed kernels flagged and the 32-bit kind re-scored (a few more, include/cudasw4_amd.h)
    double seconds = 0;
    double gcups = 0;
};

struct ScanResult {  // cudasw4.cuh:82-86
    std::vector<int> scores;
    std::vector<int64_t> referenceIds;
    BenchmarkStats stats;
};

// One launch of the scan kernel: a run of adjacent length partitions that use the same arithmetic kind.
struct LaunchRun {
    KernelType kind;
    int part_id;        // largest partition of the run
    size_t begin, end;  // subject range (shard-local positions)
    int32_t maxlen;
};

// Partition 34 (1281..8000 residues) joins the bulk launch when it alone could not run on wave-wide groups anyway: from
// this many subjects up the library gives it the 16-lane shape (sw_api.hip: lanes_for_partition), i.e. the very kernel
// the partitions below it run on.
constexpr size_t kLongPartitionMergeMin = 512;
// sw_scan_rows_pipelined takes the longest subjects of partitions 34 / 35 — those whose lone walk on an alignment group
// would take more than kPipelineWalkShare of the bulk launch's estimated time — but at most this many (a handful of long
// subjects is a latency problem, thousands of them are throughput: the scan kernels)
constexpr int32_t kPipelineMaxSubjects = 256;
constexpr double kPipelineWalkShare = 0.3;
// ... and of a subject of partition 35 scored in a 32-bit kind (its walk is final: no re-score can follow it)
constexpr double kPipelineWalkShareFinal = 0.8;
// ... and the flagged subjects of an overflow list whose 32-bit re-score walk would take more than this share of it
constexpr double kPipelineRescoreShare = 0.1;

// Partition walk of runAlignmentKernels (cudasw4.cuh:1742-2103) over the positions [begin, end) of a subject list
// whose partition p occupies [partBegin[p], partBegin[p+1]): largest partition first, adjacent partitions of equal
// kind merged into one launch (the kernels take any subject length).  Partition 35 (> 8000) always keeps a launch of
// its own (few giant subjects: wave-wide groups, a side stream); partition 34 keeps one while it holds fewer than
// kLongPartitionMergeMin subjects of the range (wave-wide groups) and otherwise MERGES with the partitions below it when
// the kinds agree: one grid takes the long subjects first instead of two grids sharing the CUs, and one side stream
// less has to find a hardware queue of its own (round 3: which streams shared a queue depended on the order they were
// created in, 10.0 ... 11.15 TCUPS on the Swiss-Prot-like DB).  A merged run reports part_id 33.
// maxLenOf(pos) = true length of the subject at pos.
// The ONE planner: the Python mirror (cudasw4_amd/search.py) calls it through swdrv_plan_runs.
// mergeMin: partition 34 merges from this many subjects up (SIZE_MAX: never — the driver's latency mode, search_driver.cpp)
template <class MaxLenOf>
std::vector<LaunchRun> plan_launch_runs(const KernelTypeConfig& kt, const size_t* partBegin, size_t begin, size_t end,
                                        MaxLenOf&& maxLenOf, size_t mergeMin = kLongPartitionMergeMin) {
    std::vector<LaunchRun> runs;
    constexpr int kSmallLong = kNumLengthPartitions - 2, kLargeLong = kNumLengthPartitions - 1;
    auto shape_of = [&](int p, size_t count) {  // 2: giants, 1: few long subjects, 0: bulk
        if (p == kLargeLong) return 2;
        if (p == kSmallLong && count < mergeMin) return 1;
        return 0;
    };
    int lastShape = -1;
    for (int p = kNumLengthPartitions - 1; p >= 0; p--) {
        const size_t b = begin > partBegin[p] ? begin : partBegin[p];
        const size_t e = end < partBegin[p + 1] ? end : partBegin[p + 1];
        if (e <= b) continue;
        const KernelType kind = kt.for_partition(p);
        const int shape = shape_of(p, e - b);
        if (!runs.empty() && runs.back().kind == kind && runs.back().begin == e && shape == 0 && lastShape == 0) {
            runs.back().begin = b;
        } else {
            const int pid = (p == kSmallLong && shape == 0) ? kSmallLong - 1 : p;
            runs.push_back(LaunchRun{kind, pid, b, e, maxLenOf(e - 1)});
        }
        lastShape = shape;
    }
    return runs;
}

// Residency of one GPU's shard (GpuWorkingSet + assignBatchesToGpuMem + computeDbCopyPlan, cudasw4.cuh:317-392,1087-1144,
// 1177-1277): what stays in device memory and how the rest is cut into streamed batches.  Pure host logic (no GPU call);
// the driver calls it in setDatabase, the CPU tests through swdrv_plan_residency.
constexpr int kTempStreams = 5;  // streams of a GPU that can hold a stripe-border scratch at a time: work, second work, 2 auxiliary, re-score service
struct ResidencyPlan {
    size_t cacheBegin = 0;      // shard-local subjects [cacheBegin, n) keep their chars in device memory (0: resident)
    uint64_t cacheBytes = 0;
    size_t tempPerStream = 0;   // cap of each stream's scratch buffer (<= --maxTempBytes; kTempStreams of them fit the limit)
    uint64_t batchBytes = 0;    // size limit of a streamed batch (0 when nothing is streamed)
    std::vector<std::pair<size_t, size_t>> batches;  // [begin, end) of every streamed batch, ascending
};
// localOffsets[n + 1]: byte offsets of the shard's subjects (ascending length); maxLen: its longest subject;
// freeMem: what the device reports free after the per-subject metadata and result arrays (24 bytes per subject) are
// allocated; allowCache = false: all-or-nothing residency (CUDASW4_AMD_NO_HYBRID=1)
ResidencyPlan plan_residency(const std::vector<uint64_t>& localOffsets, int32_t maxLen, const MemoryConfig& memory,
                             size_t freeMem, int stagingSlots, bool allowCache);

// One timed launch (HIP events on the stream the kernel ran on), for bench.py's roofline
struct KernelEvent {
    int gpu, kind, part_id;
    int32_t qlen;
    int64_t subjects;
    double cells;       // qlen x true residues of the launch's subjects
    double chars;       // padded subject bytes read by the launch
    float ms;
    // begin / end on the device clock, relative to the moment recording was switched on: launches on different streams
    // overlap, so the time the DP kernels kept the GPU busy is the measure of the UNION of these intervals
    float t0_ms, t1_ms;
    // the kernel instantiation the library chose (sw_plan_launch): arithmetic kind actually computed in, rows per lane,
    // query stripes, lanes per alignment group
    int eff_kind, rows, nstripes, lanes;
    // 1: an overflow re-score launch (cudasw4.cuh:2134-2172) over the list of the run [subjects: the run's, an upper
    // bound of the list; cells / chars: 0 — the list's length is only known on the device; the query's totals are in
    // ScanResult::stats.numRescored]
    int rescore;
};

// CPUs of a NUMA node (/sys/devices/system/node/nodeN/cpulist; empty: unknown) and binding of the calling thread to them —
// what a one-process-per-GPU caller (bench.py ranks, `align` on one GPU) does with SearchDriver::numaNode()
int numa_node_of_device(int device);  // from the PCI function of HIP device `device` (-1: unknown)
std::vector<int> cpus_of_numa_node(int node);
bool bind_thread_to_numa_node(int node);

class SearchDriver {
public:
    SearchDriver(std::vector<int> deviceIds, int numTop, MatrixId matrix, KernelTypeConfig kernels, MemoryConfig memory,
                 bool verbose, int gop, int gex);
    ~SearchDriver();
    SearchDriver(const SearchDriver&) = delete;
    SearchDriver& operator=(const SearchDriver&) = delete;

    // One process per GPU (torch.distributed / bench.py): this driver takes shards rank*numGpus .. of world*numGpus
    // of every length partition (partitionDBAmongstGpus over all GPUs of the job, cudasw4.cuh:928-1004); ids stay
    // global.  Call before setDatabase.  idBase is added to every reported id (weak-scaling replicas).
    void setShard(int rank, int world, int64_t idBase = 0);
    void setDatabase(std::shared_ptr<Database> db);  // cudasw4.cuh:552-568
    void prefetchDBToGpus();                         // cudasw4.cuh:651-696 (--uploadFull)
    void setNumTop(int k) { numTop_ = k; }           // cudasw4.cuh:574-587

    // raw residue letters, as read from the query file (cudasw4.cuh:698-765)
    ScanResult scan(const char* query, int32_t queryLength);
    // The same in two halves, so that a caller with several queries at hand (align reads whole files, main.cu:157-259)
    // keeps the GPUs busy across query boundaries: submit() enqueues everything a query needs on every GPU and returns
    // without waiting for it; collect() waits for the OLDEST submitted query and merges its results.  At most
    // kMaxInFlight queries may be pending; scan() == submit() + collect().  Results are identical either way: the
    // queries of one driver still run one after the other on each GPU.
    static constexpr int kMaxInFlight = 4;
    void submit(const char* query, int32_t queryLength);
    ScanResult collect();
    int inFlight() const { return int(pendingCount_); }

    void totalTimerStart();          // cudasw4.cuh:818-824
    BenchmarkStats totalTimerStop(); // cudasw4.cuh:826-839

    std::string_view getReferenceHeader(int64_t id) const { return db_->header(size_t(id - idBase_)); }
    int32_t getReferenceLength(int64_t id) const { return db_->length(size_t(id - idBase_)); }
    std::string getReferenceSequence(int64_t id) const { return db_->sequence_letters(size_t(id - idBase_)); }
    void printDBInfo() const;              // cudasw4.cuh:799-807
    void printDBLengthPartitions() const;  // cudasw4.cuh:809-816
    int numGpus() const { return int(gpus_.size()); }

    // ---- measurement / verification hooks (bench.py, tests) ----
    // HIP events around the DP launches from now on: 0 off, 1 every launch, 2 only the launches on the work stream
    // (the bulk run of every batch; the few-subject launches on the auxiliary streams stay unobserved)
    void recordKernelEvents(int mode);
    std::vector<KernelEvent> takeKernelEvents();      // elapsed times of the launches recorded so far (synchronises)
    size_t numLocal(int gpu) const;                   // subjects of this GPU's shard
    int numaNode(int gpu) const;                      // NUMA node of the GPU's PCI function (-1: unknown); its worker thread runs there
    int deviceOf(int gpu) const;                      // HIP device ordinal of the driver's gpu-th GPU
    uint64_t localResidues(int gpu) const;            // true residues of this GPU's shard
    uint64_t localChars(int gpu) const;               // padded subject bytes of this GPU's shard
    bool isResident(int gpu) const;                   // the whole shard's chars are kept in device memory
    // hybrid residency (cudasw4.cuh:1044-1046,1087-1144): padded subject bytes of the shard that stay in device memory
    // (== localChars when resident, 0 when everything is streamed); the rest crosses the bus on every query
    uint64_t cachedChars(int gpu) const;
    // bytes of subject chars copied host -> device by scans since the driver was created (the one-time upload of
    // resident / cached chars is NOT counted): what a streamed shard costs per query on the bus
    uint64_t streamedBytesTotal() const;
    // side launches that scanned their long subjects as overlapping windows (short queries: sw_window_overlap), and the
    // windows they scanned, since the driver was created
    void windowStats(int64_t* launches, int64_t* windows) const;
    int64_t serviceLaunches() const;   // re-score service launches (sw_rescore_service) since the driver was created
    int64_t latencyScans() const;      // scans planned in latency mode (partition 34 on wave-wide groups beside the bulk launch: small shards of real DBs)
    bool handshakeActive() const;      // the start handshake (and with it the re-score service and the tail hand-over) passed its probe on every GPU
    int64_t pipelineLaunches() const;  // ... of them as pipelines of one-wave stages over many CUs (sw_scan_rows_pipelined)
    // queries whose bulk launch was gated on the dry signal of the query before it (tail hand-over between two queries in
    // flight: submit() while a query is pending, resident shards of a few rounds of workgroups), since the driver was created
    int64_t tailOverlaps() const;
    // true when a caller that has its next query at hand should submit() it before it collect()s the current one: some
    // GPU's shard qualifies for the tail hand-over (after setDatabase) — by its size, or because a query of queryLength
    // residues (0: not considered) is scanned in a few milliseconds
    bool prefersTwoInFlight(int32_t queryLength = 0) const;
    // how many queries a caller with a query file should keep pending once a query of this length has been submitted: 1
    // (one at a time, like the reference) or 2 where the tail hand-over applies (more than two in flight was measured for very
    // short scans and is slower: search_driver.cpp)
    int preferredInFlight(int32_t queryLength) const;
    // every score of the last scan on `gpu` (the CUDASW_DEBUG_CHECK_CORRECTNESS view, cudasw4.cuh:728-756) with
    // the global id of each position; both arrays hold numLocal(gpu) entries
    void lastScores(int gpu, float* scores, int64_t* ids);
    // per-batch scan intervals of the last streamed scan: (gpu, begin ms, end ms) relative to the scan's start
    struct BatchInterval { int gpu; float begin_ms, end_ms; };
    std::vector<BatchInterval> lastBatchIntervals();
    // host-clock span of every GPU's part of the last scan, in seconds since the scan started: concurrent GPUs overlap
    struct GpuSpan { int gpu; double begin_s, end_s; };
    std::vector<GpuSpan> lastGpuSpans() const;

private:
    struct Gpu;
    struct Worker;
    void uploadShard(Gpu& g);
    void scanStreamed(Gpu& g);
    void enqueueOnGpu(Gpu& g, int32_t queryLength, int k, int slot, bool inFlight);
    bool prepareLane(Gpu& g, int32_t queryLength);
    bool laneEligible(const Gpu& g, int32_t queryLength) const;
    void finishOnGpu(Gpu& g, int slot, int32_t qlen);
    void waitForScan(Gpu& g, void* doneEvent, int32_t qlen);
    void registerStreamedRanges();
    void unregisterRanges();
    template <class F> void forEachGpu(F&& fn);
    std::vector<std::unique_ptr<Gpu>> gpus_;
    std::vector<std::unique_ptr<Worker>> workers_;
    std::shared_ptr<Database> db_;
    int numTop_;
    const SubstitutionMatrix& matrix_;
    KernelTypeConfig kernels_;
    MemoryConfig memory_;
    bool verbose_;
    int gop_, gex_;
    int shardRank_ = 0, shardWorld_ = 1;
    int64_t idBase_ = 0;
    int recordEvents_ = 0;
    bool dbRegistered_ = false;  // the streamed ranges of the DB's chars mapping are registered with the runtime (direct DMA)
    std::vector<std::pair<const int8_t*, size_t>> registered_;  // what hipHostRegister was given
    std::vector<int8_t> encodedQuery_;
    double scanT0_ = 0;
    double watchdogSeconds_ = 60.0;   // CUDASW4_AMD_WATCHDOG_SECONDS: base of collect()'s deadline (0: none)
    // queries submitted and not yet collected, oldest first (a ring of kMaxInFlight result slots per GPU)
    struct PendingScan { int slot = 0; int32_t qlen = 0; int k = 0; double t0 = 0; };
    PendingScan pending_[kMaxInFlight];
    size_t pendingHead_ = 0, pendingCount_ = 0;
    int nextSlot_ = 0;
    double lastDone_ = 0;
    // total timer
    double totalSeconds_ = 0;
    double totalCells_ = 0;
    int totalOverflows_ = 0;
    bool totalRunning_ = false;
};

}  // namespace swh
