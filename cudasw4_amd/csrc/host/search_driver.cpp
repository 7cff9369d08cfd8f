#include "search_driver.hpp"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <iostream>
#include <numeric>
#include <stdexcept>

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

}  // namespace

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

// One launch of the scan kernel: a run of adjacent length partitions that use the same kind.
struct Run {
    KernelType kind;
    int part_id;       // largest partition of the run
    size_t begin, end; // shard-local subject range
    int32_t maxlen;
};

struct DeviceBatch {  // a contiguous shard-local subject range resident (or staged) on the device
    int8_t* chars = nullptr;
    uint64_t* offsets = nullptr;
    int32_t* lengths = nullptr;
    size_t chars_capacity = 0, seq_capacity = 0;
};

struct SearchDriver::Gpu {
    int device = 0;
    sw_ctx* ctx = nullptr;
    hipStream_t stream = nullptr, copyStream = nullptr;
    std::array<ShardRange, kNumLengthPartitions> ranges{};
    std::array<size_t, kNumLengthPartitions + 1> localBegin{};
    size_t numLocal = 0;
    uint64_t localChars = 0;
    int32_t maxLen = 0;
    bool resident = false, wantResident = false;
    DeviceBatch residentDb;
    DeviceBatch staging[2];
    hipEvent_t stagingFree[2] = {nullptr, nullptr}, stagingReady[2] = {nullptr, nullptr};
    int8_t* h_pinnedChars[2] = {nullptr, nullptr};
    uint64_t* h_pinnedOffsets[2] = {nullptr, nullptr};
    int32_t* h_pinnedLengths[2] = {nullptr, nullptr};
    size_t pinnedCharsCap[2] = {0, 0}, pinnedSeqCap[2] = {0, 0};
    int32_t* h_ovfSlot = nullptr;  // pinned [2]: per-batch overflow counts of the two staging slots
    float* d_scores = nullptr;
    int32_t* d_ids = nullptr;
    int32_t* d_ovfPos = nullptr;
    int32_t* d_ovfCount = nullptr;   // [0] per batch, [1] running total of the query
    // Launches that run concurrently need their own stripe-border scratch: slot 0 = work stream,
    // slots 1.. = auxiliary streams (the reference round-robins 10 work streams, cudasw4.cuh:293,1745-1748)
    static constexpr int kAux = 2;
    hipStream_t aux[kAux] = {nullptr, nullptr};
    hipEvent_t forkEvent = nullptr, joinEvent[kAux] = {nullptr, nullptr};
    void* d_temp[kAux + 1] = {nullptr, nullptr, nullptr};
    size_t tempBytes[kAux + 1] = {0, 0, 0};
    void* d_topkTemp = nullptr;
    size_t topkTempBytes = 0;
    float* d_topS = nullptr;
    int32_t* d_topI = nullptr;
    int topCapacity = 0;
    float* h_topS = nullptr;
    int32_t* h_topI = nullptr;
    int32_t* h_ovf = nullptr;
    int lastTop = 0;

    void use() const { HIPCHECK(hipSetDevice(device)); }

    // shard-local index -> global subject id (HostGpuPartitionOffsets, cudasw4.cuh:103-213)
    int64_t toGlobal(int64_t local) const {
        const int p = int(std::upper_bound(localBegin.begin(), localBegin.end(), size_t(local)) - localBegin.begin()) - 1;
        return int64_t(ranges[p].begin + (size_t(local) - localBegin[p]));
    }
    int partitionOfLocal(size_t local) const {
        return int(std::upper_bound(localBegin.begin(), localBegin.end(), local) - localBegin.begin()) - 1;
    }
};

SearchDriver::SearchDriver(std::vector<int> deviceIds, int numTop, MatrixId matrix, KernelTypeConfig kernels,
                           MemoryConfig memory, bool verbose, int gop, int gex)
    : numTop_(numTop), matrix_(substitution_matrix(matrix)), kernels_(kernels), memory_(memory), verbose_(verbose),
      gop_(gop), gex_(gex) {
    std::string why;
    if (!kernels_.valid(&why)) throw std::runtime_error("Invalid kernel type configuration: " + why);
    if (deviceIds.empty()) throw std::runtime_error("No GPU found");
    for (int dev : deviceIds) {
        auto g = std::make_unique<Gpu>();
        g->device = dev;
        g->use();
        SWCHECK(sw_ctx_create(dev, &g->ctx));
        SWCHECK(sw_set_matrix(g->ctx, matrix_.m.data(), kAlphabet));
        HIPCHECK(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
        HIPCHECK(hipStreamCreateWithFlags(&g->copyStream, hipStreamNonBlocking));
        HIPCHECK(hipEventCreateWithFlags(&g->forkEvent, hipEventDisableTiming));
        for (int i = 0; i < Gpu::kAux; i++) {
            HIPCHECK(hipStreamCreateWithFlags(&g->aux[i], hipStreamNonBlocking));
            HIPCHECK(hipEventCreateWithFlags(&g->joinEvent[i], hipEventDisableTiming));
        }
        HIPCHECK(hipMalloc(&g->d_ovfCount, 2 * sizeof(int32_t)));
        HIPCHECK(hipHostMalloc(&g->h_ovf, 2 * sizeof(int32_t)));
        HIPCHECK(hipHostMalloc(&g->h_ovfSlot, 2 * sizeof(int32_t)));
        gpus_.push_back(std::move(g));
    }
}

SearchDriver::~SearchDriver() {
    for (auto& gp : gpus_) {
        Gpu& g = *gp;
        (void)hipSetDevice(g.device);
        (void)hipDeviceSynchronize();
        auto freeBatch = [](DeviceBatch& b) { (void)hipFree(b.chars); (void)hipFree(b.offsets); (void)hipFree(b.lengths); };
        freeBatch(g.residentDb);
        for (int i = 0; i < 2; i++) {
            freeBatch(g.staging[i]);
            (void)hipHostFree(g.h_pinnedChars[i]); (void)hipHostFree(g.h_pinnedOffsets[i]); (void)hipHostFree(g.h_pinnedLengths[i]);
            if (g.stagingFree[i]) (void)hipEventDestroy(g.stagingFree[i]);
            if (g.stagingReady[i]) (void)hipEventDestroy(g.stagingReady[i]);
        }
        (void)hipFree(g.d_scores); (void)hipFree(g.d_ids); (void)hipFree(g.d_ovfPos); (void)hipFree(g.d_ovfCount);
        for (int i = 0; i <= Gpu::kAux; i++) (void)hipFree(g.d_temp[i]);
        for (int i = 0; i < Gpu::kAux; i++) {
            if (g.aux[i]) (void)hipStreamDestroy(g.aux[i]);
            if (g.joinEvent[i]) (void)hipEventDestroy(g.joinEvent[i]);
        }
        if (g.forkEvent) (void)hipEventDestroy(g.forkEvent);
        (void)hipFree(g.d_topkTemp); (void)hipFree(g.d_topS); (void)hipFree(g.d_topI);
        (void)hipHostFree(g.h_topS); (void)hipHostFree(g.h_topI); (void)hipHostFree(g.h_ovf); (void)hipHostFree(g.h_ovfSlot);
        if (g.stream) (void)hipStreamDestroy(g.stream);
        if (g.copyStream) (void)hipStreamDestroy(g.copyStream);
        if (g.ctx) sw_ctx_destroy(g.ctx);
    }
}

void SearchDriver::setDatabase(std::shared_ptr<Database> db) {
    db_ = std::move(db);
    if (db_->num_sequences() > size_t(INT32_MAX) - 1) throw std::runtime_error("Too many sequences in DB");
    const auto shards = shard_database(*db_, int(gpus_.size()));
    for (size_t gi = 0; gi < gpus_.size(); gi++) {
        Gpu& g = *gpus_[gi];
        g.use();
        g.ranges = shards[gi];
        g.localBegin[0] = 0;
        g.localChars = 0;
        g.maxLen = 0;
        for (int p = 0; p < kNumLengthPartitions; p++) {
            g.localBegin[p + 1] = g.localBegin[p] + g.ranges[p].size();
            if (g.ranges[p].size()) {
                g.localChars += db_->offsets()[g.ranges[p].end] - db_->offsets()[g.ranges[p].begin];
                g.maxLen = std::max(g.maxLen, db_->length(g.ranges[p].end - 1));
            }
        }
        g.numLocal = g.localBegin[kNumLengthPartitions];
        g.resident = false;
        const size_t n = std::max<size_t>(g.numLocal, 1);
        (void)hipFree(g.d_scores); (void)hipFree(g.d_ids); (void)hipFree(g.d_ovfPos);
        HIPCHECK(hipMalloc(&g.d_scores, n * sizeof(float)));
        HIPCHECK(hipMalloc(&g.d_ids, n * sizeof(int32_t)));
        HIPCHECK(hipMalloc(&g.d_ovfPos, n * sizeof(int32_t)));
        // residency decision (GpuWorkingSet, cudasw4.cuh:317-392): whole shard if it fits the limit
        size_t freeMem = 0, totalMem = 0;
        HIPCHECK(hipMemGetInfo(&freeMem, &totalMem));
        size_t limit = std::min(memory_.maxGpuMem, freeMem);
        const size_t safety = size_t(256) << 20;  // cudasw4.cuh:1020-1026
        limit = limit > safety ? limit - safety : 0;
        const size_t shardBytes = g.localChars + 64 + (g.numLocal + 1) * sizeof(uint64_t) + g.numLocal * sizeof(int32_t);
        const size_t fixed = n * (sizeof(float) + 2 * sizeof(int32_t)) + std::min(memory_.maxTempBytes, size_t(1) << 30);
        g.wantResident = shardBytes + fixed <= limit;
        if (verbose_) {
            std::cout << "gpu " << g.device << ": " << g.numLocal << " sequences, " << g.localChars << " chars, "
                      << (g.wantResident ? "resident" : "streamed in batches") << "\n";
        }
    }
}

namespace {

void allocBatch(DeviceBatch& b, size_t chars, size_t seqs) {
    if (chars + 64 > b.chars_capacity) {
        (void)hipFree(b.chars);
        HIPCHECK(hipMalloc(&b.chars, chars + 64));
        b.chars_capacity = chars + 64;
    }
    if (seqs + 1 > b.seq_capacity) {
        (void)hipFree(b.offsets); (void)hipFree(b.lengths);
        HIPCHECK(hipMalloc(&b.offsets, (seqs + 1) * sizeof(uint64_t)));
        HIPCHECK(hipMalloc(&b.lengths, std::max<size_t>(seqs, 1) * sizeof(int32_t)));
        b.seq_capacity = seqs + 1;
    }
}

}  // namespace

void SearchDriver::uploadShard(Gpu& g) {
    g.use();
    allocBatch(g.residentDb, g.localChars, g.numLocal);
    std::vector<uint64_t> offsets(g.numLocal + 1);
    uint64_t charPos = 0;
    for (int p = 0; p < kNumLengthPartitions; p++) {
        const ShardRange r = g.ranges[p];
        if (!r.size()) continue;
        const uint64_t* off = db_->offsets();
        const uint64_t bytes = off[r.end] - off[r.begin];
        HIPCHECK(hipMemcpyAsync(g.residentDb.chars + charPos, db_->chars() + (off[r.begin] - off[0]), bytes, hipMemcpyHostToDevice, g.stream));
        HIPCHECK(hipMemcpyAsync(g.residentDb.lengths + g.localBegin[p], db_->lengths() + r.begin, r.size() * sizeof(int32_t), hipMemcpyHostToDevice, g.stream));
        for (size_t i = 0; i < r.size(); i++) offsets[g.localBegin[p] + i] = charPos + (off[r.begin + i] - off[r.begin]);
        charPos += bytes;
    }
    offsets[g.numLocal] = charPos;
    HIPCHECK(hipMemsetAsync(g.residentDb.chars + charPos, kOtherCode, 64, g.stream));
    HIPCHECK(hipMemcpyAsync(g.residentDb.offsets, offsets.data(), offsets.size() * sizeof(uint64_t), hipMemcpyHostToDevice, g.stream));
    HIPCHECK(hipStreamSynchronize(g.stream));
    g.resident = true;
}

void SearchDriver::prefetchDBToGpus() {
    if (!db_) throw std::runtime_error("setDatabase first");
    for (auto& g : gpus_)
        if (g->wantResident && !g->resident) uploadShard(*g);
}

namespace {

// Partition walk of runAlignmentKernels (cudasw4.cuh:1742-2103) over the shard-local range
// [begin, end): largest partition first, adjacent partitions of equal kind merged into one launch.
std::vector<Run> plan_runs(const KernelTypeConfig& kt,
                           const std::array<size_t, kNumLengthPartitions + 1>& localBegin, size_t begin, size_t end,
                           const Database& db, const std::array<ShardRange, kNumLengthPartitions>& ranges) {
    std::vector<Run> runs;
    for (int p = kNumLengthPartitions - 1; p >= 0; p--) {
        const size_t b = std::max(begin, localBegin[p]), e = std::min(end, localBegin[p + 1]);
        if (e <= b) continue;
        const KernelType kind = kt.for_partition(p);
        const int32_t maxlen = db.length(ranges[p].begin + (e - 1 - localBegin[p]));
        // partitions 34/35 (long subjects) use the wave-wide group shape: never merged with 0..33
        const bool sameShape = !runs.empty() && (runs.back().part_id >= kNumLengthPartitions - 2) == (p >= kNumLengthPartitions - 2);
        if (!runs.empty() && runs.back().kind == kind && runs.back().begin == e && sameShape) {
            runs.back().begin = b;
        } else {
            runs.push_back(Run{kind, p, b, e, maxlen});
        }
    }
    return runs;
}

}  // namespace

// Enqueue the scan of the runs of one batch whose data sits in `batch` starting at batch-local
// position 0 (shard-local position batchBegin).  The run with the most subjects goes to the work
// stream; the others (few long subjects) are launched FIRST on auxiliary streams so that they hold
// their handful of workgroups while the bulk run fills the rest of the GPU.
static void* ensure_temp(void*& ptr, size_t& have, size_t need, size_t cap) {
    need = std::min(need, cap);
    if (need > have) {
        (void)hipFree(ptr);
        ptr = nullptr;
        have = 0;
        HIPCHECK(hipMalloc(&ptr, need));
        have = need;
    }
    return ptr;
}

template <class GpuT>
static void enqueue_batch(GpuT& g, const DeviceBatch& batch, size_t batchBegin, const std::vector<Run>& runs,
                          const KernelTypeConfig& kt, const MemoryConfig& mem, int gop, int gex, int32_t maxLen) {
    bool packedUsed = false;
    HIPCHECK(hipMemsetAsync(g.d_ovfCount, 0, sizeof(int32_t), g.stream));
    size_t mainIdx = 0;
    for (size_t i = 1; i < runs.size(); i++)
        if (runs[i].end - runs[i].begin > runs[mainIdx].end - runs[mainIdx].begin) mainIdx = i;
    const bool fork = runs.size() > 1;
    if (fork) HIPCHECK(hipEventRecord(g.forkEvent, g.stream));
    auto launch = [&](const Run& r, hipStream_t stream, int slot) {
        packedUsed |= is_packed(r.kind);
        const int32_t n = int32_t(r.end - r.begin);
        const size_t need = sw_scan_temp_bytes(g.ctx, int(r.kind), r.part_id, n, r.maxlen);
        void* temp = ensure_temp(g.d_temp[slot], g.tempBytes[slot], need, mem.maxTempBytes);
        SWCHECK(sw_scan_partition(g.ctx, int(r.kind), r.part_id, batch.chars, batch.offsets, batch.lengths,
                                  int32_t(r.begin - batchBegin), n, r.maxlen, gop, gex, g.d_scores + batchBegin,
                                  g.d_ids + batchBegin, int64_t(batchBegin), g.d_ovfPos, g.d_ovfCount,
                                  is_packed(r.kind) ? 1 : 0, temp, g.tempBytes[slot], stream));
    };
    int auxUsed = 0;
    bool auxBusy[GpuT::kAux] = {};
    for (size_t i = 0; i < runs.size(); i++) {
        if (i == mainIdx) continue;
        const int a = auxUsed++ % GpuT::kAux;
        if (!auxBusy[a]) HIPCHECK(hipStreamWaitEvent(g.aux[a], g.forkEvent, 0));
        auxBusy[a] = true;
        launch(runs[i], g.aux[a], a + 1);
    }
    if (!runs.empty()) launch(runs[mainIdx], g.stream, 0);
    for (int a = 0; a < GpuT::kAux; a++) {
        if (!auxBusy[a]) continue;
        HIPCHECK(hipEventRecord(g.joinEvent[a], g.aux[a]));
        HIPCHECK(hipStreamWaitEvent(g.stream, g.joinEvent[a], 0));
    }
    if (packedUsed) {
        size_t n = 0;
        for (const Run& r : runs) n += r.end - r.begin;
        const size_t need = sw_scan_temp_bytes(g.ctx, int(kt.overflowType), -1, int32_t(n), maxLen);
        void* temp = ensure_temp(g.d_temp[0], g.tempBytes[0], need, mem.maxTempBytes);
        SWCHECK(sw_rescore_overflow(g.ctx, int(kt.overflowType), g.d_ovfPos, g.d_ovfCount, int32_t(n), batch.chars,
                                    batch.offsets, batch.lengths, maxLen, gop, gex, g.d_scores + batchBegin,
                                    g.d_ids + batchBegin, int64_t(batchBegin), temp, g.tempBytes[0], g.stream));
    }
}

void SearchDriver::scanResident(Gpu& g, int32_t /*qlen*/) {
    const auto runs = plan_runs(kernels_, g.localBegin, 0, g.numLocal, *db_, g.ranges);
    enqueue_batch(g, g.residentDb, 0, runs, kernels_, memory_, gop_, gex_, g.maxLen);
    // running total of the query (addKernel, cudasw4.cuh:46-49,2175): a single batch -> copy
    HIPCHECK(hipMemcpyAsync(g.d_ovfCount + 1, g.d_ovfCount, sizeof(int32_t), hipMemcpyDeviceToDevice, g.stream));
}

// DB shard larger than the memory limit: stream it in batches through two staging buffers
// (pinned host -> device on the copy stream, scan on the work stream), cf. cudasw4.cuh:1560-1712.
void SearchDriver::scanStreamed(Gpu& g, int32_t /*qlen*/) {
    const uint64_t* off = db_->offsets();
    // batch limits
    const size_t maxSeq = std::max<size_t>(1, memory_.maxBatchSequences);
    const size_t maxBytes = std::max<size_t>(memory_.maxBatchBytes, size_t(g.maxLen) + 4);
    for (int i = 0; i < 2; i++) {
        if (!g.stagingFree[i]) {
            HIPCHECK(hipEventCreateWithFlags(&g.stagingFree[i], hipEventDisableTiming));
            HIPCHECK(hipEventCreateWithFlags(&g.stagingReady[i], hipEventDisableTiming));
        }
    }
    int32_t total = 0;
    int slot = 0;
    std::vector<bool> slotUsed(2, false);
    for (int p = 0; p < kNumLengthPartitions; p++) {
        const ShardRange r = g.ranges[p];
        size_t cur = r.begin;
        while (cur < r.end) {
            // batch = [cur, e) of partition p within the byte / sequence limits
            size_t e = cur;
            while (e < r.end && e - cur < maxSeq && (off[e + 1] - off[cur]) <= maxBytes) e++;
            if (e == cur) e = cur + 1;
            const size_t nseq = e - cur;
            const uint64_t bytes = off[e] - off[cur];
            const size_t localBegin = g.localBegin[p] + (cur - r.begin);
            DeviceBatch& b = g.staging[slot];
            if (slotUsed[slot]) {
                // the scan that last used this slot has finished: device and pinned buffers are free again
                HIPCHECK(hipEventSynchronize(g.stagingFree[slot]));
                total += g.h_ovfSlot[slot];
            }
            allocBatch(b, bytes, nseq);
            if (bytes + 64 > g.pinnedCharsCap[slot]) {
                (void)hipHostFree(g.h_pinnedChars[slot]);
                HIPCHECK(hipHostMalloc(&g.h_pinnedChars[slot], bytes + 64));
                g.pinnedCharsCap[slot] = bytes + 64;
            }
            if (nseq + 1 > g.pinnedSeqCap[slot]) {
                (void)hipHostFree(g.h_pinnedOffsets[slot]); (void)hipHostFree(g.h_pinnedLengths[slot]);
                HIPCHECK(hipHostMalloc(&g.h_pinnedOffsets[slot], (nseq + 1) * sizeof(uint64_t)));
                HIPCHECK(hipHostMalloc(&g.h_pinnedLengths[slot], nseq * sizeof(int32_t)));
                g.pinnedSeqCap[slot] = nseq + 1;
            }
            {   // page cache / mmap -> pinned staging, in parallel chunks (one thread saturates ~6 GB/s only)
                const int8_t* src = db_->chars() + (off[cur] - off[0]);
                const size_t chunk = size_t(4) << 20;
                const long nchunks = long((bytes + chunk - 1) / chunk);
#pragma omp parallel for schedule(static) num_threads(16)
                for (long c = 0; c < nchunks; c++) {
                    const size_t b = size_t(c) * chunk;
                    std::memcpy(g.h_pinnedChars[slot] + b, src + b, std::min(chunk, size_t(bytes) - b));
                }
            }
            std::memset(g.h_pinnedChars[slot] + bytes, kOtherCode, 64);
#pragma omp parallel for schedule(static) num_threads(8)
            for (long i = 0; i <= long(nseq); i++) g.h_pinnedOffsets[slot][i] = off[cur + size_t(i)] - off[cur];
            std::memcpy(g.h_pinnedLengths[slot], db_->lengths() + cur, nseq * sizeof(int32_t));
            HIPCHECK(hipMemcpyAsync(b.chars, g.h_pinnedChars[slot], bytes + 64, hipMemcpyHostToDevice, g.copyStream));
            HIPCHECK(hipMemcpyAsync(b.offsets, g.h_pinnedOffsets[slot], (nseq + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, g.copyStream));
            HIPCHECK(hipMemcpyAsync(b.lengths, g.h_pinnedLengths[slot], nseq * sizeof(int32_t), hipMemcpyHostToDevice, g.copyStream));
            HIPCHECK(hipEventRecord(g.stagingReady[slot], g.copyStream));
            HIPCHECK(hipStreamWaitEvent(g.stream, g.stagingReady[slot], 0));
            const KernelType kind = kernels_.for_partition(p);
            std::vector<Run> runs{Run{kind, p, localBegin, localBegin + nseq, db_->length(e - 1)}};
            enqueue_batch(g, b, localBegin, runs, kernels_, memory_, gop_, gex_, db_->length(e - 1));
            HIPCHECK(hipMemcpyAsync(g.h_ovfSlot + slot, g.d_ovfCount, sizeof(int32_t), hipMemcpyDeviceToHost, g.stream));
            HIPCHECK(hipEventRecord(g.stagingFree[slot], g.stream));
            slotUsed[slot] = true;
            slot ^= 1;
            cur = e;
        }
    }
    HIPCHECK(hipStreamSynchronize(g.stream));
    for (int i = 0; i < 2; i++)
        if (slotUsed[i]) total += g.h_ovfSlot[i];
    g.h_ovf[1] = total;
    HIPCHECK(hipMemcpyAsync(g.d_ovfCount + 1, g.h_ovf + 1, sizeof(int32_t), hipMemcpyHostToDevice, g.stream));
}

ScanResult SearchDriver::scan(const char* query, int32_t queryLength) {
    if (!db_) throw std::runtime_error("setDatabase first");
    if (queryLength <= 0) throw std::runtime_error("empty query");
    if (queryLength > INT32_MAX - 132) throw std::runtime_error("query too long");  // cudasw4.cuh:1281-1285
    encodedQuery_.resize(size_t(queryLength));
    for (int32_t i = 0; i < queryLength; i++) encodedQuery_[size_t(i)] = encode_residue(query[i]);

    for (auto& gp : gpus_) { gp->use(); HIPCHECK(hipStreamSynchronize(gp->stream)); }
    const double t0 = now_seconds();

    const int k = numTop_;
    // the query goes to every GPU first (a short synchronous copy each), then the scans are enqueued: the
    // GPUs start within microseconds of each other instead of one set-up time apart
    for (auto& gp : gpus_) {
        Gpu& g = *gp;
        g.use();
        g.lastTop = 0;
        if (g.numLocal == 0) continue;
        if (g.wantResident && !g.resident) uploadShard(g);  // first query pays the upload unless --uploadFull
        SWCHECK(sw_set_query(g.ctx, encodedQuery_.data(), queryLength, g.stream));
    }
    for (auto& gp : gpus_) {
        Gpu& g = *gp;
        g.use();
        if (g.numLocal == 0) continue;
        // thrust::fill(scores, -1) (cudasw4.cuh:405-409) is not needed: every slot is written by a scan or a re-score
        if (g.resident) scanResident(g, queryLength);
        else scanStreamed(g, queryLength);
        const int kk = int(std::min<size_t>(size_t(std::max(k, 0)), g.numLocal));
        if (kk > 0) {
            if (kk > g.topCapacity) {
                (void)hipFree(g.d_topS); (void)hipFree(g.d_topI); (void)hipHostFree(g.h_topS); (void)hipHostFree(g.h_topI);
                HIPCHECK(hipMalloc(&g.d_topS, kk * sizeof(float)));
                HIPCHECK(hipMalloc(&g.d_topI, kk * sizeof(int32_t)));
                HIPCHECK(hipHostMalloc(&g.h_topS, kk * sizeof(float)));
                HIPCHECK(hipHostMalloc(&g.h_topI, kk * sizeof(int32_t)));
                g.topCapacity = kk;
            }
            const size_t tb = sw_topk_temp_bytes(int64_t(g.numLocal), kk);
            if (tb > g.topkTempBytes) {
                (void)hipFree(g.d_topkTemp);
                HIPCHECK(hipMalloc(&g.d_topkTemp, tb));
                g.topkTempBytes = tb;
            }
            SWCHECK(sw_topk(g.ctx, g.d_scores, g.d_ids, int64_t(g.numLocal), kk, g.d_topS, g.d_topI, g.d_topkTemp,
                            g.topkTempBytes, g.stream));
            HIPCHECK(hipMemcpyAsync(g.h_topS, g.d_topS, kk * sizeof(float), hipMemcpyDeviceToHost, g.stream));
            HIPCHECK(hipMemcpyAsync(g.h_topI, g.d_topI, kk * sizeof(int32_t), hipMemcpyDeviceToHost, g.stream));
            g.lastTop = kk;
        }
        HIPCHECK(hipMemcpyAsync(g.h_ovf + 1, g.d_ovfCount + 1, sizeof(int32_t), hipMemcpyDeviceToHost, g.stream));
    }

    ScanResult result;
    struct Hit { int score; int64_t id; };
    std::vector<Hit> hits;
    for (auto& gp : gpus_) {
        Gpu& g = *gp;
        if (g.numLocal == 0) continue;
        g.use();
        HIPCHECK(hipStreamSynchronize(g.stream));
        result.stats.numOverflows += g.h_ovf[1];
        for (int i = 0; i < g.lastTop; i++) hits.push_back(Hit{int(g.h_topS[i]), g.toGlobal(g.h_topI[i])});
    }
    // host merge of the per-GPU lists (replaces cudasw4.cuh:1415-1463): score desc, id asc
    std::sort(hits.begin(), hits.end(), [](const Hit& a, const Hit& b) { return a.score != b.score ? a.score > b.score : a.id < b.id; });
    if (int(hits.size()) > k) hits.resize(size_t(std::max(k, 0)));
    for (const Hit& h : hits) { result.scores.push_back(h.score); result.referenceIds.push_back(h.id); }

    const double t1 = now_seconds();
    result.stats.seconds = t1 - t0;
    const double cells = double(queryLength) * double(db_->total_residues());
    result.stats.gcups = cells / 1e9 / result.stats.seconds;  // cudasw4.cuh:2264-2271
    if (totalRunning_) { totalCells_ += cells; totalOverflows_ += result.stats.numOverflows; }
    return result;
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
