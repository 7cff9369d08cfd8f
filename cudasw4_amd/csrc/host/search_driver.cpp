#include "search_driver.hpp"
#include "parallel_blocks.hpp"
#include "trace_ranges.hpp"

#include <hip/hip_runtime_api.h>

#include <sched.h>
#include <time.h>

#include <cctype>
#include <cmath>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <mutex>
#include <numeric>
#include <stdexcept>
#include <thread>

#include "../../../include/cudasw4_amd.h"

namespace swh {

namespace {

void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
#define HIPCHECK(x) hip_check((x), #x)

void sw_check(int rc, const char* what) {
    if (rc != SW_OK) throw std::runtime_error(std::string(what) + ": " + sw_last_error());
}
#define SWCHECK(x) sw_check((x), #x)

double now_seconds() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

bool is_packed(KernelType t) { return t == KernelType::Half2 || t == KernelType::DPXs16; }

// drivers' GPU objects alive per device in this process (several shards of one device, or several drivers): the re-score
// service keeps a polling kernel on a stream, and with other objects' streams multiplexed onto the same few hardware
// queues a launch it waits for could be queued behind another object's polling kernel
std::atomic<int> g_liveOnDevice[64];

}  // namespace

// NUMA node of a HIP device: the numa_node attribute of its PCI function (-1: unknown, or the box has a single node)
int numa_node_of_device(int device) {
    char bus[64] = {};
    if (hipDeviceGetPCIBusId(bus, int(sizeof(bus)) - 1, device) != hipSuccess) { (void)hipGetLastError(); return -1; }
    std::string id(bus);
    for (char& c : id) c = char(std::tolower(static_cast<unsigned char>(c)));
    std::ifstream f("/sys/bus/pci/devices/" + id + "/numa_node");
    int node = -1;
    if (!(f >> node)) return -1;
    return node;
}

namespace {

// "0-63,128-191" -> cpu numbers
std::vector<int> parse_cpulist(const std::string& text) {
    std::vector<int> cpus;
    size_t i = 0;
    while (i < text.size()) {
        size_t j = text.find(',', i);
        if (j == std::string::npos) j = text.size();
        const std::string part = text.substr(i, j - i);
        const size_t dash = part.find('-');
        try {
            const int a = std::stoi(part.substr(0, dash));
            const int b = dash == std::string::npos ? a : std::stoi(part.substr(dash + 1));
            for (int c = a; c <= b && c < CPU_SETSIZE; c++) cpus.push_back(c);
        } catch (...) {}
        i = j + 1;
    }
    return cpus;
}

}  // namespace

std::vector<int> cpus_of_numa_node(int node) {
    if (node < 0) return {};
    std::ifstream f("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist");
    std::string text;
    if (!std::getline(f, text)) return {};
    return parse_cpulist(text);
}

// The calling thread (and the threads it creates from now on) runs on the CPUs of `node`, restricted to the CPUs the
// process may use at all (a container's cpuset); false: unknown node, or nothing of it is allowed — affinity unchanged.
bool bind_thread_to_numa_node(int node) {
    const std::vector<int> cpus = cpus_of_numa_node(node);
    if (cpus.empty()) return false;
    cpu_set_t allowed, want;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return false;
    CPU_ZERO(&want);
    int n = 0;
    for (int c : cpus)
        if (CPU_ISSET(c, &allowed)) { CPU_SET(c, &want); n++; }
    if (n == 0) return false;
    return sched_setaffinity(0, sizeof(want), &want) == 0;
}

const char* to_string(KernelType t) {
    switch (t) {
        case KernelType::Half2: return "Half2";
        case KernelType::DPXs16: return "DPXs16";
        case KernelType::DPXs32: return "DPXs32";
        case KernelType::Float: return "Float";
    }
    return "?";
}

bool parse_kernel_type(const std::string& s, KernelType& out) {
    if (s == "Half2") { out = KernelType::Half2; return true; }
    if (s == "DPXs16") { out = KernelType::DPXs16; return true; }
    if (s == "DPXs32") { out = KernelType::DPXs32; return true; }
    if (s == "Float") { out = KernelType::Float; return true; }
    return false;
}

bool KernelTypeConfig::valid(std::string* why) const {
    auto bad = [&](const char* msg) { if (why) *why = msg; return false; };
    if (!is_packed(manyPassType_small)) return bad("manyPassType_small must be Half2 or DPXs16");
    if (is_packed(manyPassType_large)) return bad("manyPassType_large must be Float or DPXs32");
    if (is_packed(overflowType)) return bad("overflowType must be Float or DPXs32");
    return true;
}

KernelType KernelTypeConfig::for_partition(int part_id) const {
    if (part_id < kNumLengthPartitions - 2) return singlePassType;
    return part_id == kNumLengthPartitions - 2 ? manyPassType_small : manyPassType_large;
}

// A batch of a streamed shard: the shard-local subjects [lbegin, lend), which may span several length
// partitions (each partition's slice is one contiguous piece of the DB's chars file).
struct Batch {
    size_t lbegin = 0, lend = 0;
    uint64_t bytes = 0;  // padded subject bytes == localOffsets[lend] - localOffsets[lbegin]
    int32_t maxLen = 0;
};

struct TimedLaunch {
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int kind = 0, part_id = 0;
    int32_t eff_kind = 0, rows = 0, nstripes = 0, lanes = 0;
    int32_t qlen = 0;
    size_t lbegin = 0, lend = 0;
    bool rescore = false;
};

struct SearchDriver::Gpu {
    // counted in g_liveOnDevice for as long as the object exists (also when a constructor of the driver throws half-way)
    struct LiveToken {
        int dev = -1;
        void hold(int d) { if (d >= 0 && d < 64) { dev = d; g_liveOnDevice[d]++; } }
        ~LiveToken() { if (dev >= 0) g_liveOnDevice[dev]--; }
    } live;
    int index = 0;   // position in gpus_
    int device = 0;
    int numaNode = -1;  // of the device's PCI function; the GPU's worker thread (and the staging copies it starts) run there
    sw_ctx* ctx = nullptr;
    // stream: the work stream (resident scans, every other batch of a streamed scan, top-K, copy-back); stream2: the
    // batches in between, so that the first workgroups of a batch fill the CUs the last round of the batch before
    // leaves idle
    hipStream_t stream = nullptr, stream2 = nullptr, copyStream = nullptr;
    std::array<ShardRange, kNumLengthPartitions> ranges{};
    std::array<size_t, kNumLengthPartitions + 1> localBegin{};
    size_t numLocal = 0;
    uint64_t localChars = 0, localResidues = 0;
    int32_t maxLen = 0;
    std::vector<uint64_t> localOffsets;   // [numLocal + 1] byte offsets into the shard's concatenated chars
    std::vector<uint64_t> resPrefix;      // lazily: prefix sums of true lengths (kernel-event statistics)
    // subject metadata is always resident (12 bytes per subject); only the chars are streamed when the shard
    // does not fit the memory limit
    uint64_t* d_offsets = nullptr;
    int32_t* d_lengths = nullptr;
    int8_t* d_chars = nullptr;
    // Hybrid residency (GpuWorkingSet / assignBatchesToGpuMem, cudasw4.cuh:317-392,1087-1144): the shard-local subjects
    // [cacheBegin, numLocal) — the longest ones, which every scan takes first — keep their chars in d_chars; the subjects
    // below cacheBegin are streamed in batches on every query.  cacheBegin == 0: the whole shard is resident;
    // cacheBegin == numLocal: everything is streamed.  cacheFilled: d_chars holds the data (--uploadFull, or the first scan).
    size_t cacheBegin = 0;
    uint64_t cacheBytes = 0;
    bool cacheFilled = false;
    uint64_t streamedBytes = 0;  // chars copied host -> device by scans so far
    // a DB whose letter codes were not validated at load (Database::codes_validated) is checked on the device as its
    // chars arrive: the cached part after its upload, every streamed batch the first time it is scanned
    std::vector<bool> batchChecked;
    bool badCodes = false;
    // streaming: three device staging buffers fed from the (registered) DB mapping, or through pinned host buffers
    std::vector<Batch> batches;
    // Two batches compute at a time (one per work stream) while a third is being copied in: three buffers
    static constexpr int kSlots = 3;
    int8_t* d_staging[kSlots] = {nullptr, nullptr, nullptr};
    size_t stagingCap = 0;
    int8_t* h_pinned[kSlots] = {nullptr, nullptr, nullptr};
    int8_t* h_pad = nullptr;  // pinned: 64 padding letters (the bytes behind a staged batch)
    size_t pinnedCap = 0;
    hipEvent_t copied[kSlots] = {nullptr, nullptr, nullptr}, scanned[kSlots] = {nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> batchEv;  // 2 per batch of the last streamed scan (intervals for tests / tuning)
    hipEvent_t scanStartEv = nullptr;
    hipEvent_t recordRefEv = nullptr; // recorded when kernel-event recording was switched on: origin of KernelEvent::t0_ms
    bool recordRefValid = false;
    float* d_scores = nullptr;
    int32_t* d_ids = nullptr;
    int32_t* d_ovfPos = nullptr;
    // one block of SW_BATCH_COUNTERS words per batch of a scan (block 0: the resident part; SW_BATCH_CNT_*: the reference's
    // overflow statistic, the lengths of the batch's overflow lists — every packed launch has its own —, pipeline stages
    // that gave up), and behind the blocks the bad-letter flag of streamed batches that are checked on the device
    int32_t* d_ovfCount = nullptr;
    size_t ovfCountCap = 0;
    // The launches of a batch — bulk grid, side launches of the long subjects beside it, pipelines, windows, overflow
    // re-score, re-score service — are the kernel library's business since round 6: one sw_scan_batch call per batch on
    // the engine below (include/cudasw4_amd.h; side streams, signal memory and scratch buffers live there).
    sw_batch* eng = nullptr;
    static constexpr int kSide = 3;                       // side streams of an engine: two auxiliary, the service's
    // the side launches of a streamed batch read the batch's staging buffer: recorded behind the last of them on each side
    // stream (sw_batch_side_events), waited for before the buffer is overwritten
    hipEvent_t auxDone[kSlots][kSide] = {};
    bool auxPending[kSlots][kSide] = {};
    hipEvent_t join2Event = nullptr;
    bool handshake = false;            // the engine's start handshake passed its probe (sw_batch_handshake_active)
    int sideReserve = 32;              // CUDASW4_AMD_SIDE_RESERVE: slots the bulk grid leaves free when two queries are in flight and the batch has side work
    int testLoseSideLaunch = 0;        // CUDASW4_AMD_TEST_LOSE_SIDE_LAUNCH=n (tests of the watchdog)
    bool stream2Used = false;
    bool firstBatchStaged = false;  // staging buffer slotBase already holds the first batch of the next streamed scan
    size_t slotBase = 0;            // staging buffer of the first batch of the next streamed scan
    bool slotUsed[kSlots] = {};     // the buffer has been scanned from since the DB was set: scanned[] is valid
    bool prefetchNext = true;       // (A/B measurements of round 3: switch in the source)
    bool twoWorkStreams = true;   // false: every batch of a streamed scan on the work stream (A/B measurements of round 3)
    // host copies of what the engine plans with: lengths and byte offsets of the shard's subjects of partitions 34 / 35
    std::vector<int32_t> longLengths;
    std::vector<uint64_t> longOffsets;
    // Tail hand-over between two queries in flight (include/cudasw4_amd.h: sw_set_dry_signal).  A query that is submitted
    // while the one before is still running goes to the OTHER lane: a second context (its own profile and work counters),
    // the second work stream with its scratch, and second score / id / overflow / top-K arrays (Lane below; the first set is
    // the fields above, LaneGuard exchanges the two for the duration of an enqueue).  Its bulk launch waits on its stream
    // for the value the bulk launch of the query before stores when its work counter runs dry, so its workgroups take the
    // slots that launch frees one by one.  The gate fixes the order: a persistent grid that is resident holds every slot,
    // so a later grid can only fill what it frees — but two grids submitted back to back (the first two queries of a
    // batch) would otherwise share the CUs half and half for their whole duration (round 3's two-lane experiment).
    // Resident shards of at most kLaneMaxRounds rounds of workgroups, or queries whose scan takes at most kLaneMaxSeconds:
    // on a large shard the last round is a small part of a long scan (10^6 x 512 peak DB: -0.2 %; 500 000: +0.4 %;
    // 250 000: +1.6 %; 125 000: +4.2 %; 62 500: +2.9 %).
    // CUDASW4_AMD_TAIL_OVERLAP=0 turns it off, =1 lifts the size rule.  Slots left free for the small launches around the
    // bulk grids (sw_set_grid_reserve) were measured and bring nothing (0 / 4 / 16: equal; 48: -0.8 %; 16 on the 567 ...
    // 1000-residue queries: -5 %): the reserve stays 0.
    struct Lane {
        sw_ctx* ctx = nullptr;
        sw_batch* eng = nullptr;
        float* d_scores = nullptr;
        int32_t* d_ids = nullptr;
        int32_t* d_ovfPos = nullptr;
        int32_t* d_ovfCount = nullptr;
        size_t ovfCountCap = 0;
        hipEvent_t scanStartEv = nullptr;
        void* d_topkTemp = nullptr;
        size_t topkTempBytes = 0;
        float* d_topS = nullptr;
        int32_t* d_topI = nullptr;
        int topCapacity = 0;
    } lane1;
    // With two queries in flight AND side work (pipelined subjects, side launches, their re-scores) a few workgroup slots
    // are worth keeping free beside the bulk grids (sideReserve above): whatever is enqueued while the other lane's
    // persistent grid holds every register of every SIMD — a re-score launch, the stages of the next query's long subjects,
    // even a one-workgroup helper kernel — is dispatched only as that grid drains, and the query it belongs to completes
    // that much later (1/4 and 1/8 Swiss-Prot-like shards: +3 ... +4 % with 32 of ~768 slots, profiles/r05_shard_proxy.txt).
    bool laneGate = true;              // false: second lane without the dry-signal gate (A/B measurements of round 4)
    static constexpr size_t kLaneMaxRounds = 20;
    static constexpr double kLaneMaxSeconds = 0.008;   // ... or scans of at most this long, at 10 TCUPS
    int laneForce = -1;                // CUDASW4_AMD_TAIL_OVERLAP
    bool lanesConcurrent = false;      // probed with the second work stream: the two work streams run beside each other
    bool lanesProbed = false;
    bool lanesFailed = false;          // the second set could not be allocated: stay on one lane
    bool lane1Ready = false;           // lane1's per-DB arrays exist
    bool laneSwapped = false;          // the fields above currently hold lane 1's set (inside an enqueue only)
    bool laneActive = false;           // the query being enqueued overlaps the one before
    int lastLane = 0;                  // lane of the query enqueued last
    int resultLane = 0;                // lane of the query collected last (lastScores)
    uint32_t* drySignal = nullptr;     // signal memory
    uint32_t drySeq = 0;               // value the bulk launch armed last stores
    uint32_t lastArmedSeq = 0;         // ... of the query enqueued last (0: it armed none)
    uint32_t waitDry = 0;              // the bulk launch being enqueued waits for this value (0: no wait)
    int64_t laneOverlaps = 0;          // queries whose bulk launch was gated on the one before
    void swapLane() {
        std::swap(ctx, lane1.ctx);
        std::swap(eng, lane1.eng);
        std::swap(stream, stream2);
        std::swap(d_scores, lane1.d_scores);
        std::swap(d_ids, lane1.d_ids);
        std::swap(d_ovfPos, lane1.d_ovfPos);
        std::swap(d_ovfCount, lane1.d_ovfCount);
        std::swap(ovfCountCap, lane1.ovfCountCap);
        std::swap(scanStartEv, lane1.scanStartEv);
        std::swap(d_topkTemp, lane1.d_topkTemp);
        std::swap(topkTempBytes, lane1.topkTempBytes);
        std::swap(d_topS, lane1.d_topS);
        std::swap(d_topI, lane1.d_topI);
        std::swap(topCapacity, lane1.topCapacity);
        laneSwapped = !laneSwapped;
    }
    struct LaneGuard {
        Gpu& g;
        bool swapped;
        LaneGuard(Gpu& gpu, int lane) : g(gpu), swapped(lane == 1) { if (swapped) g.swapLane(); }
        ~LaneGuard() { if (swapped) g.swapLane(); }
    };
    size_t tempCap = SIZE_MAX;  // plan_residency: what each of the engine's scratch buffers may grow to inside the memory limit
    void* d_topkTemp = nullptr;
    size_t topkTempBytes = 0;
    float* d_topS = nullptr;
    int32_t* d_topI = nullptr;
    int topCapacity = 0;
    // what a scan leaves for the host, per query in flight (SearchDriver::kMaxInFlight): pinned copies of the top-K and
    // of the overflow counters, and the event behind the last of those copies
    struct ResultSlot {
        hipEvent_t done = nullptr;
        float* h_topS = nullptr;
        int32_t* h_topI = nullptr;
        int topCap = 0;
        int32_t* h_ovf = nullptr;  // copy of d_ovfCount after the scan
        size_t ovfCap = 0;
        size_t ncounters = 0;
        int top = 0;
        bool used = false;         // this GPU took part in the scan (its shard is not empty)
        int lane = 0;              // whose score arrays the scan filled (Gpu::Lane)
    };
    ResultSlot res[SearchDriver::kMaxInFlight];
    int lastTop = 0;
    const float* lastTopS = nullptr;   // the finished slot's lists (valid until that slot is reused)
    const int32_t* lastTopI = nullptr;
    int lastOverflows = 0;      // subjects of the last query whose exact score reached the packed kind's limit (the reference's statistic)
    int lastRescored = 0;       // subjects the packed launches flagged and the 32-bit kind re-scored (>= lastOverflows)
    int32_t qlen = 0;
    double spanBegin = 0, spanEnd = 0;  // host clock, seconds since the scan started
    std::vector<TimedLaunch> timed;     // launches recorded since the last takeKernelEvents
    std::vector<TimedLaunch> freeTimed; // event pairs to reuse

    void use() const { HIPCHECK(hipSetDevice(device)); }

    // shard-local index -> subject index in the DB (HostGpuPartitionOffsets, cudasw4.cuh:103-213)
    int64_t toGlobal(int64_t local) const {
        const int p = int(std::upper_bound(localBegin.begin(), localBegin.end(), size_t(local)) - localBegin.begin()) - 1;
        return int64_t(ranges[p].begin + (size_t(local) - localBegin[p]));
    }
};

// One host thread per GPU (multi-GPU drivers only): runs the tasks posted to it on its own device.
struct SearchDriver::Worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<void()> task;
    bool pending = false, busy = false, stop = false;
    std::exception_ptr err;

    explicit Worker(int numaNode) {
        th = std::thread([this, numaNode] {
            // per-GPU host thread on the GPU's NUMA node: its pinned staging copies and its launches stay off the other
            // socket's memory and interconnect (CUDASW4_AMD_NO_NUMA_BIND=1: leave the affinity alone)
            const char* no = std::getenv("CUDASW4_AMD_NO_NUMA_BIND");
            if (!(no && no[0] == '1')) (void)bind_thread_to_numa_node(numaNode);
            std::unique_lock<std::mutex> lk(m);
            for (;;) {
                cv.wait(lk, [this] { return pending || stop; });
                if (stop) return;
                pending = false;
                auto fn = std::move(task);
                lk.unlock();
                std::exception_ptr e;
                try { fn(); } catch (...) { e = std::current_exception(); }
                lk.lock();
                err = e;
                busy = false;
                cv.notify_all();
            }
        });
    }
    ~Worker() {
        { std::lock_guard<std::mutex> lk(m); stop = true; }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
    void post(std::function<void()> fn) {
        std::lock_guard<std::mutex> lk(m);
        task = std::move(fn);
        pending = true;
        busy = true;
        cv.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [this] { return !busy; });
        if (err) { auto e = err; err = nullptr; std::rethrow_exception(e); }
    }
};

SearchDriver::SearchDriver(std::vector<int> deviceIds, int numTop, MatrixId matrix, KernelTypeConfig kernels,
                           MemoryConfig memory, bool verbose, int gop, int gex)
    : numTop_(numTop), matrix_(substitution_matrix(matrix)), kernels_(kernels), memory_(memory), verbose_(verbose),
      gop_(gop), gex_(gex) {
    std::string why;
    if (!kernels_.valid(&why)) throw std::runtime_error("Invalid kernel type configuration: " + why);
    if (deviceIds.empty()) throw std::runtime_error("No GPU found");
    if (const char* e = std::getenv("CUDASW4_AMD_WATCHDOG_SECONDS")) watchdogSeconds_ = std::max(0.0, std::atof(e));
    for (int dev : deviceIds) {
        auto g = std::make_unique<Gpu>();
        g->index = int(gpus_.size());
        g->device = dev;
        g->numaNode = numa_node_of_device(dev);
        g->use();
        SWCHECK(sw_ctx_create(dev, &g->ctx));
        SWCHECK(sw_set_matrix(g->ctx, matrix_.m.data(), matrix_.dim));
        HIPCHECK(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
        HIPCHECK(hipStreamCreateWithFlags(&g->copyStream, hipStreamNonBlocking));
        HIPCHECK(hipHostMalloc(&g->h_pad, 64));
        std::memset(g->h_pad, kOtherCode, 64);
        HIPCHECK(hipEventCreateWithFlags(&g->join2Event, hipEventDisableTiming));
        HIPCHECK(hipEventCreate(&g->scanStartEv));
        HIPCHECK(hipEventCreate(&g->recordRefEv));
        for (auto& r : g->res) HIPCHECK(hipEventCreateWithFlags(&r.done, hipEventDisableTiming));
        for (int i = 0; i < Gpu::kSlots; i++) {
            HIPCHECK(hipEventCreateWithFlags(&g->copied[i], hipEventDisableTiming));
            HIPCHECK(hipEventCreateWithFlags(&g->scanned[i], hipEventDisableTiming));
            for (int a = 0; a < Gpu::kSide; a++) HIPCHECK(hipEventCreateWithFlags(&g->auxDone[i][a], hipEventDisableTiming));
        }
        // the batch engine of this context: side streams (high priority: a pool of hardware queues of their own), start /
        // done signals, the probes of what the runtime really does (include/cudasw4_amd.h: sw_batch_create)
        SWCHECK(sw_batch_create(g->ctx, g->stream, &g->eng));
        g->handshake = sw_batch_handshake_active(g->eng) == 1;
        if (verbose_ && !g->handshake) std::cout << "GPU " << dev << ": start handshake, re-score service and tail hand-over off (plain stream order)\n";
        if (g->handshake) {
            // tail hand-over between two queries in flight: the dry signal of a bulk launch (Gpu::Lane)
            if (hipExtMallocWithFlags(reinterpret_cast<void**>(&g->drySignal), 8, hipMallocSignalMemory) != hipSuccess) {
                (void)hipGetLastError();
                g->drySignal = nullptr;
            } else {
                *g->drySignal = 0;
            }
            if (const char* e = std::getenv("CUDASW4_AMD_TAIL_OVERLAP")) g->laneForce = e[0] == '1' ? 1 : 0;
        }
        if (const char* e = std::getenv("CUDASW4_AMD_SIDE_RESERVE")) g->sideReserve = std::max(0, std::atoi(e));
        g->ovfCountCap = 2 * SW_BATCH_COUNTERS + 1;
        HIPCHECK(hipMalloc(&g->d_ovfCount, g->ovfCountCap * sizeof(int32_t)));
        g->live.hold(dev);
        gpus_.push_back(std::move(g));
    }
    if (gpus_.size() > 1)
        for (size_t i = 0; i < gpus_.size(); i++) workers_.push_back(std::make_unique<Worker>(gpus_[i]->numaNode));
}

SearchDriver::~SearchDriver() {
    workers_.clear();  // joins the worker threads
    for (auto& gp : gpus_) {
        Gpu& g = *gp;
        (void)hipSetDevice(g.device);
        (void)hipDeviceSynchronize();
        (void)hipFree(g.d_offsets); (void)hipFree(g.d_lengths); (void)hipFree(g.d_chars);
        for (int i = 0; i < Gpu::kSlots; i++) {
            (void)hipFree(g.d_staging[i]);
            (void)hipHostFree(g.h_pinned[i]);
            if (g.copied[i]) (void)hipEventDestroy(g.copied[i]);
            if (g.scanned[i]) (void)hipEventDestroy(g.scanned[i]);
            for (int a = 0; a < Gpu::kSide; a++)
                if (g.auxDone[i][a]) (void)hipEventDestroy(g.auxDone[i][a]);
        }
        (void)hipHostFree(g.h_pad);
        if (g.laneSwapped) g.swapLane();
        if (g.eng) sw_batch_destroy(g.eng);
        if (g.lane1.eng) sw_batch_destroy(g.lane1.eng);
        if (g.drySignal) (void)hipFree(g.drySignal);
        (void)hipFree(g.lane1.d_scores); (void)hipFree(g.lane1.d_ids); (void)hipFree(g.lane1.d_ovfPos); (void)hipFree(g.lane1.d_ovfCount);
        (void)hipFree(g.lane1.d_topkTemp); (void)hipFree(g.lane1.d_topS); (void)hipFree(g.lane1.d_topI);
        if (g.lane1.scanStartEv) (void)hipEventDestroy(g.lane1.scanStartEv);
        if (g.lane1.ctx) sw_ctx_destroy(g.lane1.ctx);
        for (hipEvent_t e : g.batchEv) (void)hipEventDestroy(e);
        if (g.scanStartEv) (void)hipEventDestroy(g.scanStartEv);
        if (g.recordRefEv) (void)hipEventDestroy(g.recordRefEv);
        for (auto& r : g.res) {
            if (r.done) (void)hipEventDestroy(r.done);
            (void)hipHostFree(r.h_topS); (void)hipHostFree(r.h_topI); (void)hipHostFree(r.h_ovf);
        }
        for (auto* v : {&g.timed, &g.freeTimed})
            for (TimedLaunch& t : *v) { (void)hipEventDestroy(t.ev0); (void)hipEventDestroy(t.ev1); }
        (void)hipFree(g.d_scores); (void)hipFree(g.d_ids); (void)hipFree(g.d_ovfPos); (void)hipFree(g.d_ovfCount);
        if (g.join2Event) (void)hipEventDestroy(g.join2Event);
        (void)hipFree(g.d_topkTemp); (void)hipFree(g.d_topS); (void)hipFree(g.d_topI);
        if (g.stream) (void)hipStreamDestroy(g.stream);
        if (g.copyStream) (void)hipStreamDestroy(g.copyStream);
        if (g.stream2) (void)hipStreamDestroy(g.stream2);
        if (g.ctx) sw_ctx_destroy(g.ctx);
    }
    unregisterRanges();
}

void SearchDriver::unregisterRanges() {
    for (auto& r : registered_) (void)hipHostUnregister(const_cast<int8_t*>(r.first));
    registered_.clear();
    dbRegistered_ = false;
}

// fn(gpu) for every GPU: on the GPUs' worker threads when there are several, else on the calling thread; the first
// exception is rethrown after all of them have finished
template <class F>
void SearchDriver::forEachGpu(F&& fn) {
    if (workers_.empty()) {
        for (auto& g : gpus_) fn(*g);
        return;
    }
    for (size_t i = 0; i < gpus_.size(); i++) {
        Gpu* g = gpus_[i].get();
        workers_[i]->post([&fn, g] { fn(*g); });
    }
    std::exception_ptr first;
    for (auto& w : workers_) {
        try { w->wait(); } catch (...) { if (!first) first = std::current_exception(); }
    }
    if (first) std::rethrow_exception(first);
}

void SearchDriver::setShard(int rank, int world, int64_t idBase) {
    if (world < 1 || rank < 0 || rank >= world) throw std::runtime_error("bad shard rank / world size");
    shardRank_ = rank;
    shardWorld_ = world;
    idBase_ = idBase;
}

// fn(src, bytes, dstOffset) for every contiguous piece of the DB's chars that the shard-local subjects [lbegin, lend)
// occupy: one slice of the chars file per length partition the range touches, dstOffset counted from the range's start
template <class GpuT, class F>
static void for_each_piece(const GpuT& g, const Database& db, size_t lbegin, size_t lend, F&& fn) {
    const uint64_t* off = db.offsets();
    uint64_t pos = 0;
    for (int p = 0; p < kNumLengthPartitions; p++) {
        const size_t lb = std::max(lbegin, g.localBegin[p]), le = std::min(lend, g.localBegin[p + 1]);
        if (le <= lb) continue;
        const size_t gb = g.ranges[p].begin + (lb - g.localBegin[p]), ge = gb + (le - lb);
        const uint64_t bytes = off[ge] - off[gb];
        fn(db.chars() + (off[gb] - off[0]), bytes, pos);
        pos += bytes;
    }
}

void SearchDriver::setDatabase(std::shared_ptr<Database> db) {
    if (pendingCount_) throw std::runtime_error("setDatabase with queries in flight: collect() them first");
    for (auto& gp : gpus_) {  // a prefetch of the old DB's first batch may still be in flight
        gp->use();
        (void)hipStreamSynchronize(gp->copyStream);
        gp->firstBatchStaged = false;
        (void)hipStreamSynchronize(gp->stream);
        if (gp->stream2) (void)hipStreamSynchronize(gp->stream2);
        for (bool& u : gp->slotUsed) u = false;
        gp->slotBase = 0;
    }
    unregisterRanges();
    db_ = std::move(db);
    if (db_->num_sequences() > size_t(INT32_MAX) - 1) throw std::runtime_error("Too many sequences in DB");
    const int ngpu = int(gpus_.size());
    const auto shards = shard_database(*db_, ngpu * shardWorld_);
    const char* noHybrid = std::getenv("CUDASW4_AMD_NO_HYBRID");  // 1: a shard that does not fit is streamed in full (A/B measurements)
    bool anyStreamed = false;
    for (int gi = 0; gi < ngpu; gi++) {
        Gpu& g = *gpus_[size_t(gi)];
        g.use();
        g.ranges = shards[size_t(shardRank_ * ngpu + gi)];
        g.localBegin[0] = 0;
        g.maxLen = 0;
        for (int p = 0; p < kNumLengthPartitions; p++) {
            g.localBegin[p + 1] = g.localBegin[p] + g.ranges[p].size();
            if (g.ranges[p].size()) g.maxLen = std::max(g.maxLen, db_->length(g.ranges[p].end - 1));
        }
        g.numLocal = g.localBegin[kNumLengthPartitions];
        g.cacheFilled = false;
        g.badCodes = false;
        g.resPrefix.clear();
        // shard-local byte offsets (the pieces of the partitions follow each other) and true residues
        g.localOffsets.assign(g.numLocal + 1, 0);
        const uint64_t* off = db_->offsets();
        uint64_t charPos = 0, residues = 0;
        for (int p = 0; p < kNumLengthPartitions; p++) {
            const ShardRange r = g.ranges[p];
            for (size_t i = 0; i < r.size(); i++) {
                g.localOffsets[g.localBegin[p] + i] = charPos + (off[r.begin + i] - off[r.begin]);
                residues += uint64_t(db_->length(r.begin + i));
            }
            if (r.size()) charPos += off[r.end] - off[r.begin];
        }
        g.localOffsets[g.numLocal] = charPos;
        g.localChars = charPos;
        g.localResidues = residues;
        // what the batch engine plans the long subjects with (pipelines, windows): lengths and byte offsets of the shard's
        // subjects of partitions 34 / 35 — a few per cent of a real DB's subjects
        {
            const size_t from = g.localBegin[kNumLengthPartitions - 2];
            g.longLengths.resize(g.numLocal - from);
            g.longOffsets.resize(g.numLocal - from);
            for (size_t i = from; i < g.numLocal; i++) {
                g.longLengths[i - from] = db_->length(size_t(g.toGlobal(int64_t(i))));
                g.longOffsets[i - from] = g.localOffsets[i];
            }
        }

        const size_t n = std::max<size_t>(g.numLocal, 1);
        (void)hipFree(g.d_scores); (void)hipFree(g.d_ids); (void)hipFree(g.d_ovfPos);
        (void)hipFree(g.lane1.d_scores); (void)hipFree(g.lane1.d_ids); (void)hipFree(g.lane1.d_ovfPos);
        g.lane1.d_scores = nullptr; g.lane1.d_ids = nullptr; g.lane1.d_ovfPos = nullptr;
        g.lane1Ready = false;
        g.lanesFailed = false;
        g.lastLane = g.resultLane = 0;
        (void)hipFree(g.d_offsets); (void)hipFree(g.d_lengths); (void)hipFree(g.d_chars);
        g.d_chars = nullptr;
        for (int i = 0; i < Gpu::kSlots; i++) { (void)hipFree(g.d_staging[i]); g.d_staging[i] = nullptr; }
        g.stagingCap = 0;
        HIPCHECK(hipMalloc(&g.d_scores, n * sizeof(float)));
        HIPCHECK(hipMalloc(&g.d_ids, n * sizeof(int32_t)));
        HIPCHECK(hipMalloc(&g.d_ovfPos, n * sizeof(int32_t)));
        HIPCHECK(hipMalloc(&g.d_offsets, (n + 1) * sizeof(uint64_t)));
        HIPCHECK(hipMalloc(&g.d_lengths, n * sizeof(int32_t)));
        HIPCHECK(hipMemcpyAsync(g.d_offsets, g.localOffsets.data(), (g.numLocal + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, g.stream));
        for (int p = 0; p < kNumLengthPartitions; p++) {
            const ShardRange r = g.ranges[p];
            if (!r.size()) continue;
            HIPCHECK(hipMemcpyAsync(g.d_lengths + g.localBegin[p], db_->lengths() + r.begin, r.size() * sizeof(int32_t), hipMemcpyHostToDevice, g.stream));
        }
        HIPCHECK(hipStreamSynchronize(g.stream));
        size_t freeMem = 0, totalMem = 0;
        HIPCHECK(hipMemGetInfo(&freeMem, &totalMem));
        const ResidencyPlan rp = plan_residency(g.localOffsets, g.maxLen, memory_, freeMem, Gpu::kSlots, !(noHybrid && noHybrid[0] == '1'));
        g.cacheBegin = rp.cacheBegin;
        g.cacheBytes = rp.cacheBytes;
        g.tempCap = rp.tempPerStream;
        g.batches.clear();
        for (const auto& be : rp.batches) {
            Batch bt;
            bt.lbegin = be.first; bt.lend = be.second;
            bt.bytes = g.localOffsets[be.second] - g.localOffsets[be.first];
            bt.maxLen = db_->length(size_t(g.toGlobal(int64_t(be.second - 1))));
            g.batches.push_back(bt);
        }
        g.batchChecked.assign(g.batches.size(), false);
        if (!g.batches.empty()) anyStreamed = true;
        if (verbose_) {
            std::cout << "gpu " << g.device << ": " << g.numLocal << " sequences, " << g.localChars << " chars, ";
            if (g.cacheBegin == 0) std::cout << "resident\n";
            else std::cout << g.cacheBytes << " chars cached in gpu memory, " << (g.localChars - g.cacheBytes) << " streamed in "
                           << g.batches.size() << " batches\n";
        }
    }
    if (anyStreamed) registerStreamedRanges();
}

// Streamed shards read the DB's chars straight from the host mapping when it can be registered with the runtime
// (hipHostRegister: the copies become asynchronous DMA from the page cache, no staging memcpy); otherwise they go through
// three pinned staging buffers per GPU.  CUDASW4_AMD_NO_HOSTREGISTER=1 forces the latter.  Only the byte ranges this
// driver streams are registered — with one process per GPU (setShard) that is this rank's overhang, not the whole file —
// and registering pins them (and faults a memory-mapped file in): above CUDASW4_AMD_HOSTREGISTER_MAX_GB (default 64) of
// streamed bytes the pinned staging buffers are used instead.
void SearchDriver::registerStreamedRanges() {
    const char* no = std::getenv("CUDASW4_AMD_NO_HOSTREGISTER");
    double maxGb = 64.0;
    if (const char* e = std::getenv("CUDASW4_AMD_HOSTREGISTER_MAX_GB")) maxGb = std::atof(e);
    std::vector<std::pair<const int8_t*, size_t>> ranges;
    for (auto& gp : gpus_)
        for_each_piece(*gp, *db_, 0, gp->cacheBegin, [&](const int8_t* src, uint64_t bytes, uint64_t) {
            if (bytes) ranges.emplace_back(src, size_t(bytes));
        });
    std::sort(ranges.begin(), ranges.end());
    // neighbours (the slices of adjacent shards of one partition) and near neighbours (closer than a huge page: their
    // pages would be pinned twice) become one registration
    std::vector<std::pair<const int8_t*, size_t>> merged;
    const size_t gap = size_t(2) << 20;
    for (const auto& r : ranges) {
        if (!merged.empty() && r.first <= merged.back().first + merged.back().second + gap)
            merged.back().second = std::max(merged.back().second, size_t(r.first - merged.back().first) + r.second);
        else merged.push_back(r);
    }
    uint64_t total = 0;
    for (const auto& r : merged) total += r.second;
    const bool small = double(total) <= maxGb * double(size_t(1) << 30);
    if (total && !(no && no[0] == '1') && small) {
        bool ok = true;
        for (const auto& r : merged) {
            if (hipHostRegister(const_cast<int8_t*>(r.first), r.second, hipHostRegisterPortable) != hipSuccess) {
                (void)hipGetLastError();
                ok = false;
                break;
            }
            registered_.push_back(r);
        }
        // HIP rejects a copy whose source starts inside a registered object and runs past its end: every piece the scans
        // will copy (streamed batches, and the cached part's one-time upload) must lie wholly inside ONE registration or
        // wholly outside all of them.  Holds by construction today (batches are cut from the streamed prefix, the merge
        // only closes gaps); checked so that a change to the piece or merge boundaries falls back to pinned staging
        // instead of failing uploads with hipErrorInvalidValue.
        auto consistent = [&](const int8_t* src, uint64_t bytes, bool mustBeInside) {
            bool inside = false;
            for (const auto& r : registered_) {
                const int8_t* rb = r.first;
                const int8_t* re = r.first + r.second;
                if (src >= rb && src + bytes <= re) { inside = true; continue; }
                if (src < re && src + bytes > rb) return false;  // straddles the edge of a registration
            }
            return inside || !mustBeInside;
        };
        for (auto& gp : gpus_) {
            if (!ok) break;
            for (const Batch& b : gp->batches)
                for_each_piece(*gp, *db_, b.lbegin, b.lend, [&](const int8_t* src, uint64_t bytes, uint64_t) { ok = ok && (!bytes || consistent(src, bytes, true)); });
            for_each_piece(*gp, *db_, gp->cacheBegin, gp->numLocal, [&](const int8_t* src, uint64_t bytes, uint64_t) { ok = ok && (!bytes || consistent(src, bytes, false)); });
        }
        if (!ok) unregisterRanges();
        dbRegistered_ = ok;
    }
    if (verbose_) {
        if (dbRegistered_) std::cout << "DB chars: " << total << " streamed bytes in " << merged.size() << " ranges registered for direct DMA\n";
        else std::cout << "DB chars staged through pinned buffers\n";
    }
}

// Residency decision (GpuWorkingSet, cudasw4.cuh:317-392,1020-1026).  The limit covers everything a GPU holds: the
// per-subject metadata and result arrays (24 bytes per subject), the scratch of multi-stripe queries, and the chars —
// resident when they fit, otherwise three staging buffers of one batch each plus as much of the shard as still fits next
// to them (the reference: "N out of M DB batches will be cached").  The streamed part is cut into batches like
// computeDbCopyPlan (cudasw4.cuh:1177-1277): consecutive subjects up to batchBytes / maxBatchSequences each; a batch may
// span adjacent length partitions (its scan is then several launches).
ResidencyPlan plan_residency(const std::vector<uint64_t>& localOffsets, int32_t maxLen, const MemoryConfig& memory,
                             size_t freeMem, int stagingSlots, bool allowCache) {
    ResidencyPlan rp;
    const size_t numLocal = localOffsets.empty() ? 0 : localOffsets.size() - 1;
    const uint64_t localChars = numLocal ? localOffsets[numLocal] : 0;
    const size_t n = std::max<size_t>(numLocal, 1);
    const size_t meta = n * 24 + 8;
    size_t limit = std::min(memory.maxGpuMem > meta ? memory.maxGpuMem - meta : 0, freeMem);
    const size_t safety = size_t(256) << 20;
    if (limit > safety) limit -= safety;  // cudasw4.cuh:1020-1026: a limit below the margin is taken as it is
    // scratch of multi-stripe queries: ONE budget for all streams that can hold a scratch buffer at the same time (work
    // stream, second work stream, auxiliary streams).  A resident shard with memory to spare keeps --maxTempBytes per
    // buffer; a shard under a binding limit splits the budget (a quarter of the limit, at most 1 GiB) evenly, so that
    // cached chars + staging + every scratch buffer stay inside the limit (a smaller scratch only lowers the number of
    // workgroups a multi-stripe launch keeps in flight, sw_api.hip: scan_common).
    const size_t allTemp = memory.maxTempBytes > SIZE_MAX / size_t(kTempStreams) ? SIZE_MAX : memory.maxTempBytes * size_t(kTempStreams);
    // (kTempStreams x the 256 MiB floor below = 1.25 GiB: with that cap the floor is INSIDE the budget whenever a quarter of
    // the limit (behind the safety margin) reaches it, i.e. from a 5.25 GiB limit up — ADVICE r4: with a 1 GiB cap the five buffers could outgrow the
    // budget by the whole safety margin at every limit)
    const size_t fixed = std::min({allTemp, (size_t(5) << 30) / 4, limit / 4});
    const size_t avail = limit - fixed;
    rp.cacheBegin = 0;
    rp.cacheBytes = localChars;
    rp.tempPerStream = memory.maxTempBytes;
    if (!numLocal || localChars + 64 <= avail) {
        // resident: the buffers may grow to --maxTempBytes each only while that still fits next to the chars
        const size_t spare = avail - size_t(localChars ? localChars + 64 : 0) + fixed;
        rp.tempPerStream = std::min(memory.maxTempBytes, std::max(spare / size_t(kTempStreams), std::min(memory.maxTempBytes, size_t(256) << 20)));
        return rp;
    }
    // never below 256 MiB (or --maxTempBytes) per buffer: under limits below 5.25 GiB (a quarter of which is less than five
    // such buffers) the limit may be exceeded by the difference — only when all five streams hold a multi-stripe scratch at
    // once; the buffers grow on demand — like the reference takes a limit below its safety margin as it is
    // (cudasw4.cuh:1020-1026).  The second lane's result arrays are outside this plan: prepareLane allocates them only with
    // a gibibyte of device memory to spare.
    rp.tempPerStream = std::max(fixed / size_t(kTempStreams), std::min(memory.maxTempBytes, size_t(256) << 20));
    // staging takes at most half of what is there, a batch is never smaller than the longest subject
    const uint64_t minBatch = std::max<uint64_t>(uint64_t(maxLen) + 4, std::min<uint64_t>(memory.maxBatchBytes, uint64_t(1) << 20));
    const uint64_t batchBytes = std::max<uint64_t>(std::min<uint64_t>(memory.maxBatchBytes, avail / (2 * uint64_t(stagingSlots))), minBatch);
    const uint64_t staging = uint64_t(stagingSlots) * (batchBytes + 64);
    const uint64_t budget = avail > staging + 64 ? avail - staging - 64 : 0;
    // the cached suffix: the longest subjects, as many as fit; a sliver below one batch is not worth a launch
    size_t cb = size_t(std::lower_bound(localOffsets.begin(), localOffsets.end(), localChars > budget ? localChars - budget : 0) - localOffsets.begin());
    if (!allowCache || localChars - localOffsets[cb] < batchBytes) cb = numLocal;
    rp.cacheBegin = cb;
    rp.cacheBytes = localChars - localOffsets[cb];
    rp.batchBytes = batchBytes;
    const size_t maxSeq = std::max<size_t>(1, memory.maxBatchSequences);
    const auto end = localOffsets.begin() + long(cb) + 1;
    size_t cur = 0;
    while (cur < cb) {
        // largest e with offsets[e] - offsets[cur] <= batchBytes and e - cur <= maxSeq
        const uint64_t target = localOffsets[cur] + batchBytes;
        size_t e = size_t(std::upper_bound(localOffsets.begin() + long(cur), end, target) - localOffsets.begin()) - 1;
        e = std::min(e, cur + maxSeq);
        e = std::min(std::max(e, cur + 1), cb);
        rp.batches.emplace_back(cur, e);
        cur = e;
    }
    return rp;
}

// the chars that stay in device memory: the whole shard, or its cached suffix
void SearchDriver::uploadShard(Gpu& g) {
    g.use();
    if (g.cacheBegin >= g.numLocal) { g.cacheFilled = true; return; }
    if (!g.d_chars) HIPCHECK(hipMalloc(&g.d_chars, g.cacheBytes + 64));
    for_each_piece(g, *db_, g.cacheBegin, g.numLocal, [&](const int8_t* src, uint64_t bytes, uint64_t pos) {
        HIPCHECK(hipMemcpyAsync(g.d_chars + pos, src, bytes, hipMemcpyHostToDevice, g.stream));
    });
    HIPCHECK(hipMemsetAsync(g.d_chars + g.cacheBytes, kOtherCode, 64, g.stream));
    if (!db_->codes_validated()) {
        // the flag borrows the first overflow counter (no scan is in flight on this GPU during an upload)
        int32_t bad = 0;
        HIPCHECK(hipMemsetAsync(g.d_ovfCount, 0, sizeof(int32_t), g.stream));
        SWCHECK(sw_check_letter_codes(g.ctx, g.d_chars, g.cacheBytes, g.d_ovfCount, g.stream));
        HIPCHECK(hipMemcpyAsync(&bad, g.d_ovfCount, sizeof(int32_t), hipMemcpyDeviceToHost, g.stream));
        HIPCHECK(hipStreamSynchronize(g.stream));
        if (bad) { g.badCodes = true; throw DbLoadError("DB chars hold letter codes outside 0..20 (not a cudasw4 DB, or corrupt)"); }
    }
    HIPCHECK(hipStreamSynchronize(g.stream));
    g.cacheFilled = true;
}

// --uploadFull (cudasw4.cuh:651-696): every GPU takes what it keeps — its whole shard, or the cached part of a shard
// that does not fit — before the first query, all GPUs at once (one worker thread per GPU)
void SearchDriver::prefetchDBToGpus() {
    if (!db_) throw std::runtime_error("setDatabase first");
    if (pendingCount_) throw std::runtime_error("prefetchDBToGpus with queries in flight");
    forEachGpu([this](Gpu& g) {
        if (!g.cacheFilled && g.numLocal) uploadShard(g);
    });
}

// Enqueue the scan of the shard-local subjects [lbegin, lend) whose chars start at `chars` (device) as batch `batch` of the
// current query, staged in buffer `slot` (-1: resident chars): ONE call of the kernel library (sw_scan_batch: the body of
// the reference's runAlignmentKernels and its overflow block, cudasw4.cuh:1742-2172).  What stays here is what only the
// driver knows: where the batch's partitions begin, how long its long subjects are, which work stream and counter block
// the batch gets, whether a second query is in flight (tail hand-over), and the bookkeeping of a staging buffer's readers.
// The side launches of a batch are NOT joined at its end — a giant subject keeps one wave busy for as long as the whole
// batch takes — but before the top-K (join_aux); a staging buffer is overwritten only after the side launches that read it.
template <class GpuT>
static void enqueue_batch(GpuT& g, const int8_t* chars, size_t lbegin, size_t lend, const Database& db,
                          const KernelTypeConfig& kt, const MemoryConfig& mem, int gop, int gex, int recordMode,
                          size_t batch, int slot, bool second) {
    TraceRange traceBatch("batch %d: subjects [%ld, %ld) %s", int(batch), long(lbegin), long(lend), slot < 0 ? "resident" : "streamed");
    if (lend <= lbegin) return;
    if (lend - lbegin > size_t(INT32_MAX)) throw std::runtime_error("batch too large for 32-bit positions");
    constexpr int kSmallLong = kNumLengthPartitions - 2;
    int32_t partBegin[kNumLengthPartitions + 1], partMax[kNumLengthPartitions];
    for (int p = 0; p <= kNumLengthPartitions; p++)
        partBegin[p] = int32_t(std::min(std::max(g.localBegin[p], lbegin), lend) - lbegin);
    for (int p = 0; p < kNumLengthPartitions; p++) {
        const size_t e = std::min(g.localBegin[p + 1], lend), b = std::max(g.localBegin[p], lbegin);
        partMax[p] = e > b ? int32_t(db.length(size_t(g.toGlobal(int64_t(e - 1))))) : 0;
    }
    sw_batch_args a{};
    a.kinds[0] = int(kt.singlePassType); a.kinds[1] = int(kt.manyPassType_small);
    a.kinds[2] = int(kt.manyPassType_large); a.kinds[3] = int(kt.overflowType);
    a.chars = chars;
    a.offsets = g.d_offsets + lbegin;
    a.lengths = g.d_lengths + lbegin;
    a.n = int32_t(lend - lbegin);
    a.part_begin = partBegin;
    a.part_maxlen = partMax;
    const size_t longFrom = std::max(lbegin, g.localBegin[kSmallLong]);   // first subject of partitions 34 / 35 inside the batch
    if (longFrom < lend && !g.longLengths.empty()) {
        a.long_lengths = g.longLengths.data() + (longFrom - g.localBegin[kSmallLong]);
        a.long_offsets = g.longOffsets.data() + (longFrom - g.localBegin[kSmallLong]);
        a.long_offsets_bias = g.localOffsets[lbegin];
    }
    a.batch_bytes = g.localOffsets[lend] - g.localOffsets[lbegin];
    a.gop = gop; a.gex = gex;
    a.scores = g.d_scores + lbegin;
    a.ids = g.d_ids + lbegin;
    a.id_offset = int64_t(lbegin);
    a.ovf_pos = g.d_ovfPos + lbegin;
    a.counters = g.d_ovfCount + batch * size_t(SW_BATCH_COUNTERS);   // zeroed at the start of the scan
    a.max_temp_bytes = std::min(mem.maxTempBytes, g.tempCap);
    a.stream = second ? g.stream2 : g.stream;
    a.work_slot = second ? 1 : 0;
    // the re-score service polls on a stream of its own: resident chars only (a staging buffer would have to wait for it),
    // and only while no other object's streams share this device's few hardware queues with it
    const bool aloneOnDevice = g.device < 0 || g.device >= 64 || g_liveOnDevice[g.device].load() == 1;
    a.allow_service = slot < 0 && !second && !g.laneActive && aloneOnDevice;
    // tail hand-over (Gpu::Lane): the bulk launch of a resident scan arms the dry signal for whoever is submitted while it
    // runs, and waits for the dry signal of the query before it when that one still runs on the other lane
    if (g.drySignal && g.handshake && slot < 0 && !second) {
        a.arm_signal = g.drySignal;
        a.arm_value = ++g.drySeq;
        g.lastArmedSeq = g.drySeq;
        if (g.waitDry) {
            a.wait_signal = g.drySignal;
            a.wait_value = g.waitDry;
            g.waitDry = 0;
            g.laneOverlaps++;
        }
    }
    a.grid_reserve_side = g.laneActive ? g.sideReserve : 0;
    // two queries in flight: the side launches of consecutive queries start on different side streams of their engines
    a.alt_side_stream = g.lastLane == 1 && slot < 0;
    constexpr int kMaxRecords = 24;
    sw_launch_record recs[kMaxRecords];
    int32_t used = 0;
    if (recordMode) {
        for (auto& r : recs) {
            TimedLaunch t;
            if (!g.freeTimed.empty()) { t = g.freeTimed.back(); g.freeTimed.pop_back(); }
            else { HIPCHECK(hipEventCreate(&t.ev0)); HIPCHECK(hipEventCreate(&t.ev1)); }
            r.ev0 = t.ev0; r.ev1 = t.ev1;
        }
        a.records = recs; a.records_cap = kMaxRecords; a.records_used = &used; a.record_mode = recordMode;
    }
    if (g.testLoseSideLaunch > 0) { SWCHECK(sw_batch_test_lose_side_launch(g.eng, g.testLoseSideLaunch)); g.testLoseSideLaunch = 0; }
    const int rc = sw_scan_batch(g.eng, &a);
    if (recordMode) {
        for (int i = 0; i < kMaxRecords; i++) {
            TimedLaunch t;
            t.ev0 = static_cast<hipEvent_t>(recs[i].ev0); t.ev1 = static_cast<hipEvent_t>(recs[i].ev1);
            if (i < used && rc == SW_OK) {
                t.kind = recs[i].kind; t.part_id = recs[i].part_id; t.qlen = g.qlen;
                t.lbegin = lbegin + size_t(recs[i].begin); t.lend = lbegin + size_t(recs[i].end);
                t.rescore = recs[i].rescore != 0;
                t.eff_kind = recs[i].eff_kind; t.rows = recs[i].rows; t.nstripes = recs[i].nstripes; t.lanes = recs[i].lanes;
                g.timed.push_back(t);
            } else {
                g.freeTimed.push_back(t);
            }
        }
    }
    SWCHECK(rc);
    if (slot >= 0) {
        void* evs[GpuT::kSide];
        int usedSide[GpuT::kSide] = {};
        for (int i = 0; i < GpuT::kSide; i++) evs[i] = g.auxDone[slot][i];
        SWCHECK(sw_batch_side_events(g.eng, evs, usedSide));
        for (int i = 0; i < GpuT::kSide; i++)
            if (usedSide[i]) g.auxPending[slot][i] = true;
    }
}

// the work stream continues only after everything the side streams were given in this scan
template <class GpuT>
static void join_aux(GpuT& g) {
    SWCHECK(sw_batch_join(g.eng, g.stream));
    // auxPending stays set: the host no longer waits for the end of a scan before it enqueues the next one, so the copy
    // that reuses a staging buffer must still wait for the side launches that read it (an event that has long completed
    // costs nothing)
    if (g.stream2Used) {
        HIPCHECK(hipEventRecord(g.join2Event, g.stream2));
        HIPCHECK(hipStreamWaitEvent(g.stream, g.join2Event, 0));
        g.stream2Used = false;
    }
}

static void ensure_ovf_slots(int32_t*& h, size_t& cap, size_t need) {
    if (need <= cap) return;
    (void)hipHostFree(h);
    h = nullptr; cap = 0;
    HIPCHECK(hipHostMalloc(&h, need * sizeof(int32_t)));
    cap = need;
}

// The part of a shard that does not fit the memory limit: its chars stream through three device staging buffers (copy
// stream -> the two work streams in turn: two batches compute while the third is copied in), cf. cudasw4.cuh:1560-1712;
// offsets and lengths are resident.  Batches run longest subjects first, so the tail of the query consists of short
// subjects.  With the DB mapping registered the whole scan is enqueued without blocking the host; the pinned fallback
// blocks only on its own host buffers.  Batch k of the scan order uses overflow counters k + 1 (0: the cached part).
void SearchDriver::scanStreamed(Gpu& g) {
    const size_t nb = g.batches.size();
    const bool cached = g.cacheBegin < g.numLocal;
    uint64_t maxBytes = 0;
    for (const Batch& b : g.batches) maxBytes = std::max(maxBytes, b.bytes);
    if (maxBytes + 64 > g.stagingCap) {
        for (int i = 0; i < Gpu::kSlots; i++) {
            (void)hipFree(g.d_staging[i]);
            g.d_staging[i] = nullptr;
            HIPCHECK(hipMalloc(&g.d_staging[i], maxBytes + 64));
        }
        g.stagingCap = maxBytes + 64;
    }
    if (!dbRegistered_ && maxBytes + 64 > g.pinnedCap) {
        for (int i = 0; i < Gpu::kSlots; i++) {
            (void)hipHostFree(g.h_pinned[i]);
            g.h_pinned[i] = nullptr;
            HIPCHECK(hipHostMalloc(&g.h_pinned[i], maxBytes + 64));
        }
        g.pinnedCap = maxBytes + 64;
    }
    while (g.batchEv.size() < 2 * nb) {
        hipEvent_t e;
        HIPCHECK(hipEventCreate(&e));
        g.batchEv.push_back(e);
    }
    // The buffers rotate across scans (slotBase): the first batch of the next scan then gets the buffer whose last user
    // finishes earliest, two batches before the end of this scan.
    bool* const slotUsed = g.slotUsed;
    auto slot_of = [&](size_t k) { return int((g.slotBase + k) % Gpu::kSlots); };
    // copy of batch k of the scan order (longest subjects first; k == nb: the first batch again, for the next scan) into
    // its staging buffer, on the copy stream
    auto copy_batch = [&](size_t k) {
        const Batch& b = g.batches[nb - 1 - (k % nb)];
        const int slot = slot_of(k);
        int8_t* dst = g.d_staging[slot];
        // the scan that last used this device buffer must have finished before the copy overwrites it
        if (slotUsed[slot]) HIPCHECK(hipStreamWaitEvent(g.copyStream, g.scanned[slot], 0));
        for (int a = 0; a < Gpu::kSide; a++) {
            if (!g.auxPending[slot][a]) continue;
            HIPCHECK(hipStreamWaitEvent(g.copyStream, g.auxDone[slot][a], 0));
            g.auxPending[slot][a] = false;
        }
        if (!dbRegistered_ && slotUsed[slot]) HIPCHECK(hipEventSynchronize(g.copied[slot]));  // pinned buffer free again
        // the batch's pieces: one contiguous slice of the chars file per length partition it touches
        for_each_piece(g, *db_, b.lbegin, b.lend, [&](const int8_t* src, uint64_t bytes, uint64_t pos) {
            if (dbRegistered_) {
                HIPCHECK(hipMemcpyAsync(dst + pos, src, bytes, hipMemcpyHostToDevice, g.copyStream));
            } else {
                // page cache / mmap -> pinned staging, in parallel chunks (one thread moves ~6 GB/s only); plain threads
                // that end with the copy, not OpenMP workers that would spin next to the scan (parallel_blocks.hpp)
                const size_t chunk = size_t(4) << 20;
                int8_t* hdst = g.h_pinned[slot] + pos;
                parallel_blocks((bytes + chunk - 1) / chunk, 8, [&](size_t c) {
                    const size_t o = c * chunk;
                    std::memcpy(hdst + o, src + o, std::min(chunk, size_t(bytes) - o));
                });
            }
        });
        if (!dbRegistered_) HIPCHECK(hipMemcpyAsync(dst, g.h_pinned[slot], b.bytes, hipMemcpyHostToDevice, g.copyStream));
        // the 64 padding bytes behind the batch come by DMA from a pinned block of padding letters, not from
        // hipMemsetAsync: a memset is a KERNEL, and on the copy stream it would wait for a free workgroup slot — which the
        // persistent grid of the batch that is computing only frees at its tail — with every later copy queued behind it
        HIPCHECK(hipMemcpyAsync(dst + b.bytes, g.h_pad, 64, hipMemcpyHostToDevice, g.copyStream));
        HIPCHECK(hipEventRecord(g.copied[slot], g.copyStream));
        g.streamedBytes += b.bytes;
    };
    for (size_t k = 0; k < nb; k++) {
        const Batch& b = g.batches[nb - 1 - k];
        const int slot = slot_of(k);
        int8_t* dst = g.d_staging[slot];
        // the first batch may already be there: the previous scan copied it in behind its own last batches (below)
        if (k == 0 && g.firstBatchStaged) g.firstBatchStaged = false;
        else copy_batch(k);
        // Batches alternate between the two work streams; with a cached part in front the first streamed batch takes
        // the second one, next to the cached part's launch.  stream2 is created with the first streamed scan that needs
        // it: a driver whose shards are resident keeps the set of streams it was tuned with (one more stream of the work
        // stream's priority changes which streams end up sharing a hardware queue: the giants' launch of a RESIDENT
        // Swiss-Prot-like DB went back in front of the bulk launch, 138 instead of 107 ms for the longest query, when
        // stream2 was created in the constructor)
        // Round 6: where the cached part is the larger part of the shard, EVERY streamed batch takes the second work stream:
        // a batch on the first one queues behind the cached part's launch (they share its stripe-border scratch), and a
        // Swiss-Prot-like shard with 65 % of its chars cached then ran three of its five batches only after that launch had
        // ended — alone, behind a GPU that had been half idle (8 575 GCUPS; tools/hybrid_diag.py).
        const bool besideCached = cached && 2 * uint64_t(g.cacheBytes) >= uint64_t(g.localChars);
        const bool second = (besideCached || ((k + (cached ? 1 : 0)) & 1)) && g.twoWorkStreams;
        if (second && !g.stream2) HIPCHECK(hipStreamCreateWithFlags(&g.stream2, hipStreamNonBlocking));
        const hipStream_t work = second ? g.stream2 : g.stream;
        if (second && !g.stream2Used) {
            // ordered after the query upload and the zeroed counters on the work stream
            HIPCHECK(hipStreamWaitEvent(g.stream2, g.scanStartEv, 0));
            g.stream2Used = true;
        }
        HIPCHECK(hipStreamWaitEvent(work, g.copied[slot], 0));
        if (!db_->codes_validated() && !g.batchChecked[nb - 1 - k]) {
            // the flag is the word behind this scan's overflow counters (enqueueOnGpu zeroes and copies it back)
            SWCHECK(sw_check_letter_codes(g.ctx, dst, b.bytes, g.d_ovfCount + (1 + nb) * size_t(SW_BATCH_COUNTERS), work));
            g.batchChecked[nb - 1 - k] = true;
        }
        HIPCHECK(hipEventRecord(g.batchEv[2 * k], work));
        enqueue_batch(g, dst, b.lbegin, b.lend, *db_, kernels_, memory_, gop_, gex_, recordEvents_, k + 1, slot, second);
        HIPCHECK(hipEventRecord(g.batchEv[2 * k + 1], work));
        HIPCHECK(hipEventRecord(g.scanned[slot], work));
        slotUsed[slot] = true;
    }
    // The DB does not depend on the query: the first batch of the NEXT scan goes into its buffer as soon as this scan's
    // last user of that buffer is done, behind this scan's last copies — otherwise every query starts with a copy that
    // nothing hides (2-4 ms per 128 MB batch; a 144-residue query scans such a batch in 1.7 ms).  Only with the DB mapping
    // registered (pure DMA; the pinned fallback would block the host here); one batch copy is wasted after the last query.
    if (dbRegistered_ && nb > 0 && g.prefetchNext) {
        copy_batch(nb);
        g.firstBatchStaged = true;
    }
    g.slotBase = (g.slotBase + nb) % Gpu::kSlots;
}

// Everything one GPU is GIVEN for one query; runs on the GPU's worker thread when there are several.  Nothing here waits
// for the GPU (the pinned-staging fallback of a streamed shard waits for its own host buffers): the results land in
// result slot `slot` and finishOnGpu picks them up.  The scans of consecutive queries are ordered by the streams alone:
// the query upload, the zeroed counters and the first launches of query i + 1 queue up behind the top-K and the copies of
// query i on the work stream, the auxiliary streams fork from it and join it again before the top-K.
// The second lane's resources (Gpu::Lane), created with the first query that can use them.  False: stay on one lane.
// queryLength: the shorter one of the two queries that would overlap (0: only the shard-size rule)
bool SearchDriver::laneEligible(const Gpu& g, int32_t queryLength) const {
    if (g.lanesFailed || !g.handshake || !g.drySignal || g.laneForce == 0 || g.numLocal == 0) return false;
    if (g.cacheBegin != 0 || !g.batches.empty()) return false;  // resident shards only
    if (g.laneForce == 1) return true;
    // a scan that is over in a few milliseconds is mostly ramp-up, last round and fixed work per query, whatever the
    // shard's size (streams of 48 ... 222-residue queries on the Swiss-Prot-like DB: +26 ... +8 %; 375 residues, 8 ms: +1 %)
    if (queryLength > 0 && double(queryLength) * double(g.localResidues) / 1e13 <= Gpu::kLaneMaxSeconds) return true;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, g.device) != hipSuccess || cus <= 0) return false;
    // a round: three workgroups per CU, 16 groups of two subjects each
    return g.numLocal <= Gpu::kLaneMaxRounds * size_t(cus) * 3 * 32;
}

bool SearchDriver::prefersTwoInFlight(int32_t queryLength) const {
    if (!db_) return false;
    for (auto& gp : gpus_)
        if (laneEligible(*gp, queryLength)) return true;
    return false;
}

int SearchDriver::preferredInFlight(int32_t queryLength) const {
    // Two where the tail hand-over applies.  More than two (kMaxInFlight allows four) was measured for scans of under 2 ms
    // and is SLOWER: a stream of 48-residue queries on the Swiss-Prot-like DB 7 672 -> 7 175 GCUPS, the 1/8 shard's 20
    // queries 9 571 -> 9 167 — the third query's side launches and helper kernels queue up behind two persistent grids
    // that hold every slot (profiles/r05_short_queries.txt).
    return prefersTwoInFlight(queryLength) ? 2 : 1;
}

bool SearchDriver::prepareLane(Gpu& g, int32_t queryLength) {
    if (!laneEligible(g, queryLength) || !g.cacheFilled) return false;
    try {
        if (!g.stream2) HIPCHECK(hipStreamCreateWithFlags(&g.stream2, hipStreamNonBlocking));
        if (!g.lanesProbed) {
            g.lanesConcurrent = sw_streams_run_concurrently(g.ctx, g.stream, g.stream2) == 1;
            g.lanesProbed = true;
        }
        if (!g.lanesConcurrent) { g.lanesFailed = true; return false; }
        if (!g.lane1.ctx) {
            SWCHECK(sw_ctx_create(g.device, &g.lane1.ctx));
            SWCHECK(sw_set_matrix(g.lane1.ctx, matrix_.m.data(), matrix_.dim));
        }
        if (!g.lane1.eng) {
            SWCHECK(sw_batch_create(g.lane1.ctx, g.stream2, &g.lane1.eng));
            if (sw_batch_handshake_active(g.lane1.eng) != 1) { g.lanesFailed = true; return false; }
        }
        if (!g.lane1.scanStartEv) HIPCHECK(hipEventCreate(&g.lane1.scanStartEv));
        if (!g.lane1.d_ovfCount) {
            g.lane1.ovfCountCap = SW_BATCH_COUNTERS + 1;
            HIPCHECK(hipMalloc(&g.lane1.d_ovfCount, g.lane1.ovfCountCap * sizeof(int32_t)));
        }
        if (!g.lane1Ready) {
            const size_t n = std::max<size_t>(g.numLocal, 1);
            // the second set of result arrays is not in the residency plan's budget: only with room to spare (a shard that
            // fills the device keeps one lane rather than risk the scratch buffers it still has to grow)
            size_t freeMem = 0, totalMem = 0;
            HIPCHECK(hipMemGetInfo(&freeMem, &totalMem));
            if (freeMem < 4 * n * 12 + (size_t(1) << 30)) { g.lanesFailed = true; return false; }
            HIPCHECK(hipMalloc(&g.lane1.d_scores, n * sizeof(float)));
            HIPCHECK(hipMalloc(&g.lane1.d_ids, n * sizeof(int32_t)));
            HIPCHECK(hipMalloc(&g.lane1.d_ovfPos, n * sizeof(int32_t)));
            g.lane1Ready = true;
        }
    } catch (const std::exception&) {
        (void)hipGetLastError();
        (void)hipFree(g.lane1.d_scores); (void)hipFree(g.lane1.d_ids); (void)hipFree(g.lane1.d_ovfPos);
        g.lane1.d_scores = nullptr; g.lane1.d_ids = nullptr; g.lane1.d_ovfPos = nullptr;
        g.lane1Ready = false;
        g.lanesFailed = true;
        return false;
    }
    return true;
}

void SearchDriver::enqueueOnGpu(Gpu& g, int32_t queryLength, int k, int slot, bool inFlight) {
    Gpu::ResultSlot& rs = g.res[slot];
    rs.used = false;
    rs.top = 0;
    rs.ncounters = 0;
    rs.lane = 0;
    g.spanBegin = g.spanEnd = now_seconds() - scanT0_;
    if (g.numLocal == 0) return;
    TraceRange traceGpu("enqueue on GPU %d: query of %d residues, %ld subjects", g.device, int(queryLength), long(g.numLocal));
    g.use();
    // a query submitted while the one before is running takes the other lane and is gated on that one's dry signal
    const bool overlap = inFlight && prepareLane(g, std::min(g.qlen, queryLength));  // (g.qlen: the query before)
    const int lane = overlap ? 1 - g.lastLane : 0;
    Gpu::LaneGuard laneGuard(g, lane);
    g.laneActive = overlap;
    g.waitDry = overlap && g.laneGate ? g.lastArmedSeq : 0;
    g.lastArmedSeq = 0;
    g.lastLane = lane;
    rs.lane = lane;
    try {
        if (g.badCodes) throw DbLoadError("DB chars hold letter codes outside 0..20 (not a cudasw4 DB, or corrupt)");
        g.qlen = queryLength;
        if (!g.cacheFilled) uploadShard(g);  // the first query pays the upload unless --uploadFull
        SWCHECK(sw_set_query(g.ctx, encodedQuery_.data(), queryLength, g.stream));
        // thrust::fill(scores, -1) (cudasw4.cuh:405-409) is not needed: every slot is written by a scan or a re-score
        // + 1: the bad-letter flag of streamed batches that are checked on the device (scanStreamed); + 1: pipeline stages
        // that gave up waiting (sw_scan_rows_pipelined: fail_count)
        // one counter block per batch (the resident part is batch 0) and the bad-letter flag of streamed batches that are
        // checked on the device (scanStreamed) behind them: zeroed by ONE memset, copied back by one copy
        const size_t ncounters = (1 + g.batches.size()) * size_t(SW_BATCH_COUNTERS);
        if (ncounters + 1 > g.ovfCountCap) {
            (void)hipFree(g.d_ovfCount);
            g.d_ovfCount = nullptr;
            g.ovfCountCap = 0;
            HIPCHECK(hipMalloc(&g.d_ovfCount, (ncounters + 1) * sizeof(int32_t)));
            g.ovfCountCap = ncounters + 1;
        }
        ensure_ovf_slots(rs.h_ovf, rs.ovfCap, ncounters + 1);
        HIPCHECK(hipMemsetAsync(g.d_ovfCount, 0, (ncounters + 1) * sizeof(int32_t), g.stream));
        HIPCHECK(hipEventRecord(g.scanStartEv, g.stream));
        // the cached part first (the longest subjects: one set of launches over everything that is resident), then the
        // streamed batches — whose first copies run while the cached part computes
        if (g.cacheBegin < g.numLocal) {
            enqueue_batch(g, g.d_chars, g.cacheBegin, g.numLocal, *db_, kernels_, memory_, gop_, gex_, recordEvents_, 0, -1, false);
            // (the first streamed batch, usually staged already by the previous scan, starts on the second work stream next
            // to this launch; making it wait for the cached part was measured and is slower: 10.49 against 10.71 TCUPS on the
            // Swiss-Prot-like DB with 65 % cached, 10.91 against 11.21 on the peak DB with 25 % cached)
        }
        if (!g.batches.empty()) scanStreamed(g);
        join_aux(g);
        const int kk = int(std::min<size_t>(size_t(std::max(k, 0)), g.numLocal));
        if (kk > 0) {
            if (kk > g.topCapacity) {
                (void)hipFree(g.d_topS); (void)hipFree(g.d_topI);
                g.d_topS = nullptr; g.d_topI = nullptr; g.topCapacity = 0;
                HIPCHECK(hipMalloc(&g.d_topS, kk * sizeof(float)));
                HIPCHECK(hipMalloc(&g.d_topI, kk * sizeof(int32_t)));
                g.topCapacity = kk;
            }
            if (kk > rs.topCap) {
                (void)hipHostFree(rs.h_topS); (void)hipHostFree(rs.h_topI);
                rs.h_topS = nullptr; rs.h_topI = nullptr; rs.topCap = 0;
                HIPCHECK(hipHostMalloc(&rs.h_topS, kk * sizeof(float)));
                HIPCHECK(hipHostMalloc(&rs.h_topI, kk * sizeof(int32_t)));
                rs.topCap = kk;
            }
            const size_t tb = sw_topk_temp_bytes(int64_t(g.numLocal), kk);
            if (tb > g.topkTempBytes) {
                (void)hipFree(g.d_topkTemp);
                g.d_topkTemp = nullptr;
                g.topkTempBytes = 0;
                HIPCHECK(hipMalloc(&g.d_topkTemp, tb));
                g.topkTempBytes = tb;
            }
            {
                TraceRange traceTop("top-%d of %ld scores", kk, long(g.numLocal));
                SWCHECK(sw_topk(g.ctx, g.d_scores, g.d_ids, int64_t(g.numLocal), kk, g.d_topS, g.d_topI, g.d_topkTemp,
                                g.topkTempBytes, g.stream));
            }
            HIPCHECK(hipMemcpyAsync(rs.h_topS, g.d_topS, kk * sizeof(float), hipMemcpyDeviceToHost, g.stream));
            HIPCHECK(hipMemcpyAsync(rs.h_topI, g.d_topI, kk * sizeof(int32_t), hipMemcpyDeviceToHost, g.stream));
            rs.top = kk;
        }
        // per-query totals (addKernel, cudasw4.cuh:46-49,2175): summed on the host after the copy
        HIPCHECK(hipMemcpyAsync(rs.h_ovf, g.d_ovfCount, (ncounters + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, g.stream));
        HIPCHECK(hipEventRecord(rs.done, g.stream));
        rs.ncounters = ncounters;
        rs.used = true;
    } catch (...) {
        // Work may still be queued on the auxiliary, second and copy streams, and the bookkeeping of who waits for
        // whom is half-updated: drain the device and forget it, so that a later scan starts from a clean state
        for (sw_batch* e : {g.eng, g.lane1.eng})
            if (e) (void)sw_batch_open_gates(e);       // a re-score service that is still polling, a bulk launch behind a side launch that never went in
        if (g.drySignal) *g.drySignal = g.drySeq;      // ... and a bulk launch that waits for a launch that never went in
        g.waitDry = 0;
        g.lastArmedSeq = 0;
        (void)hipDeviceSynchronize();
        (void)hipGetLastError();
        for (sw_batch* e : {g.eng, g.lane1.eng})
            if (e) (void)sw_batch_reset(e);            // nothing is in flight now: the engines' counts start afresh
        g.stream2Used = false;
        for (auto& sl : g.auxPending)
            for (bool& pnd : sl) pnd = false;
        for (bool& u : g.slotUsed) u = false;
        g.firstBatchStaged = false;
        g.slotBase = 0;
        // the letter-code checks this scan enqueued never reported back: check those batches again
        g.batchChecked.assign(g.batchChecked.size(), false);
        rs.used = false;
        throw;
    }
}

// The host-side watchdog.  A scan's work stream can wait in hipStreamWaitValue32 for values only
// device code raises — the start signal of its side launches, the dry signal of the query before it — and a re-score service
// polls until the work stream writes behind the bulk launch.  Device-side waits are bounded (spin limits); this bounds the
// host side: the done event is polled with a deadline of watchdogSeconds_ (CUDASW4_AMD_WATCHDOG_SECONDS, default 60; 0:
// wait without a limit) plus ten times what the scan should take at a tenth of the usual rate.  On expiry the gates are
// opened by hand — signal memory is host-visible — so that the streams drain, the device is given a bounded time to do
// so, and the scan FAILS with a message that names the signal that never arrived: the caller gets an exception, `align`
// exits non-zero; nothing is re-executed.
void SearchDriver::waitForScan(Gpu& g, void* doneEvent, int32_t qlen) {
    const hipEvent_t done = static_cast<hipEvent_t>(doneEvent);
    if (watchdogSeconds_ <= 0.0) { HIPCHECK(hipEventSynchronize(done)); return; }
    const double estimate = double(std::max(qlen, 1)) * double(g.localResidues) / 1e12 + double(g.localChars - g.cacheBytes) / 2e9;
    const double t0 = now_seconds(), deadline = watchdogSeconds_ + 10.0 * estimate;
    auto poll = [&](double limit) -> bool {   // true: the event completed
        for (uint64_t spins = 0;; spins++) {
            const hipError_t q = hipEventQuery(done);
            if (q == hipSuccess) return true;
            if (q != hipErrorNotReady) hip_check(q, "hipEventQuery(scan done)");
            const double waited = now_seconds() - t0;
            if (waited > limit) return false;
            // a scan of a millisecond is picked up by plain polling; a long one lets the core go
            if (waited > 0.02) { struct timespec ts = {0, 200000}; nanosleep(&ts, nullptr); }
        }
    };
    if (poll(deadline)) return;
    // which gate is closed?
    std::string what = "scan timed out after " + std::to_string(int(now_seconds() - t0)) + " s (deadline " + std::to_string(int(deadline)) + " s)";
    bool named = false;
    for (sw_batch* e : {g.eng, g.lane1.eng}) {
        if (!e || named) continue;
        uint32_t startNow = 0, startTarget = 0, doneNow = 0, doneTarget = 0;
        (void)sw_batch_signal_state(e, &startNow, &startTarget, &doneNow, &doneTarget);
        if (int32_t(startTarget - startNow) > 0) {
            what += ": " + std::to_string(startTarget - startNow) + " side launch(es) never reported their workgroups resident (start signal " +
                    std::to_string(startNow) + " of " + std::to_string(startTarget) + "): the bulk launch behind it never started";
            named = true;
        } else if (int32_t(doneTarget - doneNow) > 0) {
            what += ": the re-score service was never told that its producer had finished (done signal " + std::to_string(doneNow) + " of " +
                    std::to_string(doneTarget) + ")";
            named = true;
        }
    }
    const uint32_t dryNow = g.drySignal ? *reinterpret_cast<volatile uint32_t*>(g.drySignal) : 0;
    if (!named && g.drySignal && int32_t(g.drySeq - dryNow) > 0)
        what += ": the dry signal stands at " + std::to_string(dryNow) + " of " + std::to_string(g.drySeq) + ": a bulk launch gated on the query before it never started";
    else if (!named)
        what += ": no gate of the driver is closed — a kernel does not finish";
    for (sw_batch* e : {g.eng, g.lane1.eng})
        if (e) (void)sw_batch_open_gates(e);
    if (g.drySignal) *reinterpret_cast<volatile uint32_t*>(g.drySignal) = g.drySeq;
    const bool drained = poll(deadline + 10.0);
    what += drained ? "; the gates were opened by hand and the device drained, the scan's results are not valid" : "; the device did not drain within 10 s after the gates were opened";
    // forget who waits for whom: a later scan on this driver starts clean (as after any failed enqueue)
    g.waitDry = 0;
    g.lastArmedSeq = 0;
    throw std::runtime_error(what);
}

// wait for the results of the query that went into result slot `slot` and add up its counters
void SearchDriver::finishOnGpu(Gpu& g, int slot, int32_t qlen) {
    Gpu::ResultSlot& rs = g.res[slot];
    g.lastTop = 0;
    g.lastOverflows = 0;
    g.lastRescored = 0;
    g.lastTopS = rs.h_topS;
    g.lastTopI = rs.h_topI;
    g.resultLane = rs.lane;
    if (!rs.used) return;
    g.use();
    waitForScan(g, rs.done, qlen);
    rs.used = false;
    // A streamed batch is checked by the FIRST scan that uses it; with two queries in flight the second one was enqueued
    // before the first one's flag came back and carries no check of its own: a flag raised by any earlier query
    // (g.badCodes) condemns its scores as well.
    if (rs.h_ovf[rs.ncounters] || g.badCodes) {
        g.badCodes = true;
        throw DbLoadError("DB chars hold letter codes outside 0..20 (not a cudasw4 DB, or corrupt)");
    }
    int failed = 0;
    int64_t dirty = 0;
    for (size_t k = 0; k < rs.ncounters; k += size_t(SW_BATCH_COUNTERS)) {   // one block per batch of the scan
        const int32_t* c = rs.h_ovf + k;
        g.lastOverflows += c[SW_BATCH_CNT_OVERFLOWS];
        for (int i = 0; i < 4; i++) g.lastRescored += c[SW_BATCH_CNT_LIST0 + i];
        g.lastRescored += c[SW_BATCH_CNT_PIPE_OVER];   // pipelined subjects a packed launch would have flagged: scored in 32 bits as well
        dirty += c[SW_BATCH_CNT_DIRTY];
        failed += c[SW_BATCH_CNT_FAILED];
    }
    if (failed)
        throw std::runtime_error("scan failed: " + std::to_string(failed) +
                                 " pipeline stage(s) of a long subject gave up waiting for their neighbour (sw_scan_rows_pipelined)");
    g.lastTop = rs.top;
    // how much recent scans re-scored arms and sizes the re-score service of the lane's engine
    if (sw_batch* e = rs.lane == 1 ? (g.laneSwapped ? g.eng : g.lane1.eng) : (g.laneSwapped ? g.lane1.eng : g.eng)) SWCHECK(sw_batch_feedback(e, int32_t(std::max<int64_t>(0, int64_t(g.lastRescored) - dirty))));
    rs.used = false;
    g.spanEnd = now_seconds() - scanT0_;
}

void SearchDriver::submit(const char* query, int32_t queryLength) {
    if (!db_) throw std::runtime_error("setDatabase first");
    if (queryLength <= 0) throw std::runtime_error("empty query");
    if (queryLength > INT32_MAX - 132) throw std::runtime_error("query too long");  // cudasw4.cuh:1281-1285
    if (pendingCount_ >= size_t(kMaxInFlight)) throw std::runtime_error("too many queries in flight: collect() first");
    encodedQuery_.resize(size_t(queryLength));
    // 25-letter tables: the query keeps B, J, Z, X and '*' apart; the DB side stays the dbdata alphabet (include/cudasw4_amd.h)
    if (matrix_.dim == 25) for (int32_t i = 0; i < queryLength; i++) encodedQuery_[size_t(i)] = encode_residue25(query[i]);
    else for (int32_t i = 0; i < queryLength; i++) encodedQuery_[size_t(i)] = encode_residue(query[i]);

    TraceRange traceQuery("submit query of %d residues (%d in flight)", int(queryLength), int(pendingCount_));
    PendingScan ps;
    ps.slot = nextSlot_;
    ps.qlen = queryLength;
    ps.k = numTop_;
    ps.t0 = now_seconds();
    if (pendingCount_ == 0) scanT0_ = ps.t0;
    const int slot = ps.slot, k = ps.k;
    const bool inFlight = pendingCount_ > 0;
    forEachGpu([this, queryLength, k, slot, inFlight](Gpu& g) { enqueueOnGpu(g, queryLength, k, slot, inFlight); });
    nextSlot_ = (nextSlot_ + 1) % kMaxInFlight;
    pending_[(pendingHead_ + pendingCount_) % kMaxInFlight] = ps;
    pendingCount_++;
}

ScanResult SearchDriver::collect() {
    if (!pendingCount_) throw std::runtime_error("collect() without a submitted query");
    const PendingScan ps = pending_[pendingHead_];
    TraceRange traceCollect("collect query of %d residues", int(ps.qlen));
    pendingHead_ = (pendingHead_ + 1) % kMaxInFlight;
    pendingCount_--;
    std::exception_ptr first;
    for (auto& gp : gpus_) {  // plain waits: no need for the worker threads
        try { finishOnGpu(*gp, ps.slot, ps.qlen); } catch (...) { if (!first) first = std::current_exception(); }
    }
    if (first) std::rethrow_exception(first);

    ScanResult result;
    struct Hit { int score; int64_t id; };
    std::vector<Hit> hits;
    for (auto& gp : gpus_) {
        Gpu& g = *gp;
        result.stats.numOverflows += g.lastOverflows;
        result.stats.numRescored += g.lastRescored;
        for (int i = 0; i < g.lastTop; i++) hits.push_back(Hit{int(g.lastTopS[i]), idBase_ + g.toGlobal(g.lastTopI[i])});
    }
    // host merge of the per-GPU lists (replaces cudasw4.cuh:1415-1463): score desc, id asc
    std::sort(hits.begin(), hits.end(), [](const Hit& a, const Hit& b) { return a.score != b.score ? a.score > b.score : a.id < b.id; });
    if (int(hits.size()) > ps.k) hits.resize(size_t(std::max(ps.k, 0)));
    for (const Hit& h : hits) { result.scores.push_back(h.score); result.referenceIds.push_back(h.id); }

    // a query's time runs from its submission, or from the end of the query before it when it had to queue behind that
    const double t1 = now_seconds();
    result.stats.seconds = t1 - std::max(ps.t0, lastDone_);
    lastDone_ = t1;
    uint64_t residues = 0;
    for (auto& gp : gpus_) residues += gp->localResidues;
    const double cells = double(ps.qlen) * double(residues);
    result.stats.gcups = cells / 1e9 / result.stats.seconds;  // cudasw4.cuh:2264-2271
    if (totalRunning_) { totalCells_ += cells; totalOverflows_ += result.stats.numOverflows; }
    return result;
}

ScanResult SearchDriver::scan(const char* query, int32_t queryLength) {
    if (pendingCount_) throw std::runtime_error("scan() with queries in flight: collect() them first");
    submit(query, queryLength);
    return collect();
}

void SearchDriver::totalTimerStart() {
    for (auto& gp : gpus_) { gp->use(); HIPCHECK(hipDeviceSynchronize()); }
    totalSeconds_ = now_seconds();
    totalCells_ = 0;
    totalOverflows_ = 0;
    totalRunning_ = true;
}

BenchmarkStats SearchDriver::totalTimerStop() {
    for (auto& gp : gpus_) { gp->use(); HIPCHECK(hipDeviceSynchronize()); }
    BenchmarkStats s;
    s.seconds = now_seconds() - totalSeconds_;
    s.gcups = totalCells_ / 1e9 / s.seconds;
    s.numOverflows = totalOverflows_;
    totalRunning_ = false;
    return s;
}

// ---- measurement / verification hooks

void SearchDriver::recordKernelEvents(int mode) {
    if (mode && !recordEvents_) {
        for (auto& gp : gpus_) {  // origin of the launches' begin / end times
            gp->use();
            HIPCHECK(hipEventRecord(gp->recordRefEv, gp->stream));
            gp->recordRefValid = true;
        }
    }
    recordEvents_ = mode;
}

std::vector<KernelEvent> SearchDriver::takeKernelEvents() {
    std::vector<KernelEvent> out;
    for (auto& gp : gpus_) {
        Gpu& g = *gp;
        g.use();
        HIPCHECK(hipDeviceSynchronize());
        if (!g.timed.empty() && g.resPrefix.empty()) {
            g.resPrefix.assign(g.numLocal + 1, 0);
            for (size_t i = 0; i < g.numLocal; i++) g.resPrefix[i + 1] = g.resPrefix[i] + uint64_t(db_->length(size_t(g.toGlobal(int64_t(i)))));
        }
        for (TimedLaunch& t : g.timed) {
            KernelEvent e{};
            e.gpu = g.index; e.kind = t.kind; e.part_id = t.part_id; e.qlen = t.qlen;
            e.eff_kind = t.eff_kind; e.rows = t.rows; e.nstripes = t.nstripes; e.lanes = t.lanes;
            e.subjects = int64_t(t.lend - t.lbegin);
            e.rescore = t.rescore ? 1 : 0;
            e.cells = t.rescore ? 0.0 : double(t.qlen) * double(g.resPrefix[t.lend] - g.resPrefix[t.lbegin]);
            e.chars = t.rescore ? 0.0 : double(g.localOffsets[t.lend] - g.localOffsets[t.lbegin]);
            HIPCHECK(hipEventElapsedTime(&e.ms, t.ev0, t.ev1));
            e.t0_ms = e.t1_ms = 0.f;
            if (g.recordRefValid) {
                HIPCHECK(hipEventElapsedTime(&e.t0_ms, g.recordRefEv, t.ev0));
                HIPCHECK(hipEventElapsedTime(&e.t1_ms, g.recordRefEv, t.ev1));
            }
            out.push_back(e);
            g.freeTimed.push_back(t);
        }
        g.timed.clear();
    }
    return out;
}

size_t SearchDriver::numLocal(int gpu) const { return gpus_.at(size_t(gpu))->numLocal; }
int SearchDriver::numaNode(int gpu) const { return gpus_.at(size_t(gpu))->numaNode; }
int SearchDriver::deviceOf(int gpu) const { return gpus_.at(size_t(gpu))->device; }
uint64_t SearchDriver::localResidues(int gpu) const { return gpus_.at(size_t(gpu))->localResidues; }
uint64_t SearchDriver::localChars(int gpu) const { return gpus_.at(size_t(gpu))->localChars; }
bool SearchDriver::isResident(int gpu) const {
    const Gpu& g = *gpus_.at(size_t(gpu));
    return g.cacheBegin == 0 && g.cacheFilled;
}
uint64_t SearchDriver::cachedChars(int gpu) const { return gpus_.at(size_t(gpu))->cacheBytes; }
namespace {
int64_t engine_stat(const sw_batch* a, const sw_batch* b, int which) {
    int64_t v[6] = {}, w[6] = {};
    if (a) (void)sw_batch_stats(a, v, 6);
    if (b) (void)sw_batch_stats(b, w, 6);
    return v[which] + w[which];
}
}  // namespace

void SearchDriver::windowStats(int64_t* launches, int64_t* windows) const {
    int64_t l = 0, w = 0;
    for (auto& gp : gpus_) { l += engine_stat(gp->eng, gp->lane1.eng, 2); w += engine_stat(gp->eng, gp->lane1.eng, 3); }
    if (launches) *launches = l;
    if (windows) *windows = w;
}
int64_t SearchDriver::latencyScans() const { return 0; }   // (latency mode: measured, superseded by the walk-time cut, removed in round 6)

bool SearchDriver::handshakeActive() const {
    for (auto& gp : gpus_)
        if (!gp->handshake) return false;
    return !gpus_.empty();
}

int64_t SearchDriver::pipelineLaunches() const {
    int64_t n = 0;
    for (auto& gp : gpus_) n += engine_stat(gp->eng, gp->lane1.eng, 0);
    return n;
}


int64_t SearchDriver::tailOverlaps() const {
    int64_t n = 0;
    for (auto& gp : gpus_) n += gp->laneOverlaps;
    return n;
}

int64_t SearchDriver::serviceLaunches() const {
    int64_t n = 0;
    for (auto& gp : gpus_) n += engine_stat(gp->eng, gp->lane1.eng, 4);
    return n;
}
uint64_t SearchDriver::streamedBytesTotal() const {
    uint64_t t = 0;
    for (auto& gp : gpus_) t += gp->streamedBytes;
    return t;
}

void SearchDriver::lastScores(int gpu, float* scores, int64_t* ids) {
    Gpu& g = *gpus_.at(size_t(gpu));
    if (!g.numLocal) return;
    g.use();
    HIPCHECK(hipMemcpy(scores, g.resultLane == 1 ? g.lane1.d_scores : g.d_scores, g.numLocal * sizeof(float), hipMemcpyDeviceToHost));
    for (int p = 0; p < kNumLengthPartitions; p++)
        for (size_t i = 0; i < g.ranges[p].size(); i++) ids[g.localBegin[p] + i] = idBase_ + int64_t(g.ranges[p].begin + i);
}

std::vector<SearchDriver::BatchInterval> SearchDriver::lastBatchIntervals() {
    std::vector<BatchInterval> out;
    for (auto& gp : gpus_) {
        Gpu& g = *gp;
        if (g.batches.empty() || g.batchEv.size() < 2 * g.batches.size()) continue;
        g.use();
        HIPCHECK(hipDeviceSynchronize());
        for (size_t k = 0; k < g.batches.size(); k++) {
            BatchInterval bi{g.index, 0.f, 0.f};
            HIPCHECK(hipEventElapsedTime(&bi.begin_ms, g.scanStartEv, g.batchEv[2 * k]));
            HIPCHECK(hipEventElapsedTime(&bi.end_ms, g.scanStartEv, g.batchEv[2 * k + 1]));
            out.push_back(bi);
        }
    }
    return out;
}

std::vector<SearchDriver::GpuSpan> SearchDriver::lastGpuSpans() const {
    std::vector<GpuSpan> out;
    for (auto& gp : gpus_) out.push_back(GpuSpan{gp->index, gp->spanBegin, gp->spanEnd});
    return out;
}

void SearchDriver::printDBInfo() const {
    std::cout << db_->num_sequences() << " sequences, " << db_->num_chars() << " characters\n";
    if (db_->num_sequences()) {
        std::cout << "Min length " << db_->length(0) << ", max length " << db_->length(db_->num_sequences() - 1)
                  << ", avg length " << double(db_->total_residues()) / double(db_->num_sequences()) << "\n";
    }
}

void SearchDriver::printDBLengthPartitions() const {
    const auto& b = length_partition_bounds();
    for (int p = 0; p < kNumLengthPartitions; p++)
        std::cout << "<= " << b[p] << ": " << db_->partition_counts()[p] << "\n";
}

}  // namespace swh
