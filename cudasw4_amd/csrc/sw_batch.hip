// sw_batch.hip — sw_scan_batch: ONE call for the whole body of the reference's runAlignmentKernels (cudasw4.cuh:1742-2103)
// and its overflow block (cudasw4.cuh:2134-2172), on top of the launchers of sw_api.hip.
//
// Host code only.  What a batch needs beyond "one launch per partition, one behind the other":
//   * PLAN: adjacent partitions of one arithmetic kind are one launch (the kernels take any subject length); partition 35
//     (> 8000 residues) and a small partition 34 keep launches of their own on wave-wide groups — side launches;
//   * SIDE LAUNCHES run BESIDE the bulk grid: on the engine's high-priority auxiliary streams, each announcing itself through
//     the start signal the bulk launch waits for (a persistent grid that is dispatched first keeps every workgroup slot to
//     its end: sw_set_start_signal);
//   * the LONGEST subjects — those whose lone walk on an alignment group would take a good part of the bulk launch's time —
//     leave the scan launches and run as pipelines of one-wave stages (sw_scan_rows_pipelined), or, for short queries, as
//     overlapping windows (sw_window_overlap);
//   * every packed launch has its own OVERFLOW LIST and re-score launch; the bulk launch's list is re-scored while it is
//     filled (sw_rescore_service), the long entries of a list pipelined (sw_rescore_overflow_pipelined).
// Rounds 2-5 grew this in the host driver (search_driver.cpp: enqueue_batch, 500 lines on the wrong side of the boundary:
// VERDICT r5 item 3); round 6 moved it here, minus what had been measured and rejected on the way (latency mode, the
// one-workgroup row kernel as a driver choice, stream-creation orders).
// Structure: sw_scan_batch validates and then drives a BatchJob — plan() (host arithmetic: pipeline cuts, runs, which run is
// the bulk launch, the service decision), launch_side() (pipeline parts, side runs, service: each adds 1 to the start signal),
// launch_bulk() (behind the start signal and the caller's dry-signal gate), launch_rescores().
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/cudasw4_amd.h"
#include "../../include/cudasw4_amd_engine.h"
#include "sw_internal.hpp"

namespace {

constexpr int kParts = SW_NUM_LENGTH_PARTITIONS;
constexpr int kSmallLong = kParts - 2, kLargeLong = kParts - 1;
constexpr int kAux = 2;
constexpr int kLists = 4;
constexpr int kWindowBufs = 4;
// partition 34 joins the bulk launch from this many subjects up (sw_api.hip: lanes_for_partition gives it the bulk's shape)
constexpr int32_t kLongPartitionMergeMin = 512;
// a subject leaves the scan launches for the pipeline when its lone walk would take more than this share of the bulk launch's
// estimated time (packed partition 34: it may have to be walked twice, the re-score follows the bulk launch) ...
constexpr double kPipelineWalkShare = 0.3;
// ... or this share, for a subject of partition 35 in a 32-bit kind (its walk is final and hides beside the bulk launch)
constexpr double kPipelineWalkShareFinal = 0.8;
// ... and an entry of an overflow list when its re-score walk would take more than this share
constexpr double kPipelineRescoreShare = 0.1;
constexpr int32_t kPipelineMaxSubjects = 256;

bool packed(int kind) { return kind == SW_KIND_F16X2 || kind == SW_KIND_I16X2; }

struct Run {
    int kind, part_id;
    int32_t begin, end, maxlen;
};

struct WindowBuf {
    char* h = nullptr; size_t hcap = 0;
    char* d = nullptr; size_t dcap = 0;
    hipEvent_t copied = nullptr;
    bool used = false;
};

}  // namespace

struct sw_batch {
    sw_ctx* ctx = nullptr;
    int device = 0;
    hipStream_t aux[kAux] = {nullptr, nullptr};
    hipStream_t svc = nullptr;
    hipEvent_t fork[2] = {nullptr, nullptr};
    hipEvent_t joinEv[kAux + 1] = {nullptr, nullptr, nullptr};
    bool auxUsed[kAux] = {false, false};      // since the last join
    bool svcUsed = false;
    bool lastAux[kAux] = {false, false};      // by the batch enqueued last (sw_batch_side_events)
    bool lastSvc = false;
    uint32_t* startSignal = nullptr;          // signal memory
    uint32_t sideLaunches = 0;
    uint32_t* doneSignal = nullptr;
    uint32_t doneSeq = 0;
    bool handshake = false;
    bool svcConcurrent = false;
    // scratch: work stream slots 0 / 1, auxiliary streams, service stream
    void* temp[kAux + 3] = {};
    size_t tempBytes[kAux + 3] = {};
    WindowBuf win[kAux][kWindowBufs];
    int winNext[kAux] = {0, 0};
    // knobs (environment: A/B measurements and tests)
    bool pipelines = true;                    // CUDASW4_AMD_PIPELINES=0: never
    bool pipelinesAlways = false;             // CUDASW4_AMD_PIPELINES=always: every subject of partition 35 (tests)
    double walkShare = kPipelineWalkShare, walkShareFinal = kPipelineWalkShareFinal, rescoreShare = kPipelineRescoreShare;
    int32_t pipelineMaxSubjects = kPipelineMaxSubjects;
    int windows = 1;                          // CUDASW4_AMD_WINDOWS=0|1|always (0 / 1 / 2)
    int svcForce = -1;                        // CUDASW4_AMD_RESCORE_SERVICE=0|1
    int split34MaxLanes = 4;
    // heuristics fed by sw_batch_feedback
    int quietScans = 0;
    double rescoredEma = 8.0;
    int testLoseSide = 0;
    int64_t stats[6] = {};
    int serviceWorkgroups() const { return rescoredEma < 8.0 ? 2 : rescoredEma < 64.0 ? 4 : 8; }
};

namespace {

#define SWB_HIP(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t e__ = (expr);                                                                            \
        if (e__ != hipSuccess) return swi::fail(SW_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)
#define SWB_OK(expr)                  \
    do {                              \
        const int rc__ = (expr);      \
        if (rc__ != SW_OK) return rc__; \
    } while (0)

int ensure_temp(sw_batch* b, int slot, size_t need, size_t cap, void** out) {
    need = std::min(need, cap);
    if (need > b->tempBytes[slot]) {
        (void)hipFree(b->temp[slot]);
        b->temp[slot] = nullptr;
        b->tempBytes[slot] = 0;
        SWB_HIP(hipMalloc(&b->temp[slot], need));
        b->tempBytes[slot] = need;
    }
    *out = b->temp[slot];
    return SW_OK;
}

bool env_on(const char* name) {
    const char* e = getenv(name);
    return e && e[0] && e[0] != '0' && e[0] != 'f' && e[0] != 'F';
}

}  // namespace

extern "C" {

int32_t sw_query_length(const sw_ctx* ctx) { return swi::query_length(ctx); }

int sw_batch_create(sw_ctx* ctx, void* work_stream, sw_batch** out) {
    if (!ctx || !out) return swi::fail(SW_ERR_INVALID, "null argument");
    *out = nullptr;
    const int dev = swi::device_of(ctx);
    SWB_HIP(hipSetDevice(dev));
    sw_batch* b = new sw_batch;
    b->ctx = ctx;
    b->device = dev;
    auto bail = [&](int rc) { sw_batch_destroy(b); return rc; };
    int prioLow = 0, prioHigh = 0;
    if (hipDeviceGetStreamPriorityRange(&prioLow, &prioHigh) != hipSuccess) { (void)hipGetLastError(); prioHigh = 0; }
    // The side streams are HIGH priority: the runtime multiplexes the streams of one priority onto a few hardware queues,
    // and a side launch queued behind the bulk launch on the same queue would run behind it whatever the handshake says
    // ... and every side stream gets a hardware queue of its own: two side launches of one scan on streams that share a queue
    // run one behind the other, and the bulk launch, which waits until all of them hold their slots, starts late by the
    // first one's whole duration (seen on the second engine of a device: profiles/r06_shard_queues.txt).  The runtime hands
    // out queues round-robin at creation, so a stream that turns out to share one (sw_streams_run_concurrently) is set
    // aside and the next one tried; the ones set aside are destroyed at the end.
    hipStream_t const work0 = static_cast<hipStream_t>(work_stream);
    std::vector<hipStream_t> setAside;
    for (int a = 0; a < kAux; a++) {
        for (int tries = 0; tries < 8 && !b->aux[a]; tries++) {
            hipStream_t s = nullptr;
            if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prioHigh) != hipSuccess) return bail(swi::fail(SW_ERR_HIP, "hipStreamCreateWithPriority"));
            bool alone = sw_streams_run_concurrently(ctx, s, work0) == 1;
            for (int o = 0; o < a && alone; o++) alone = sw_streams_run_concurrently(ctx, s, b->aux[o]) == 1;
            if (alone) b->aux[a] = s; else setAside.push_back(s);
        }
        if (!b->aux[a]) { b->aux[a] = setAside.back(); setAside.pop_back(); }   // (no queue to itself to be had: plain stream order)
        if (hipEventCreateWithFlags(&b->joinEv[a], hipEventDisableTiming) != hipSuccess) return bail(swi::fail(SW_ERR_HIP, "hipEventCreate"));
    }
    for (hipStream_t s : setAside) (void)hipStreamDestroy(s);
    for (int i = 0; i < 2; i++)
        if (hipEventCreateWithFlags(&b->fork[i], hipEventDisableTiming) != hipSuccess) return bail(swi::fail(SW_ERR_HIP, "hipEventCreate"));
    if (hipEventCreateWithFlags(&b->joinEv[kAux], hipEventDisableTiming) != hipSuccess) return bail(swi::fail(SW_ERR_HIP, "hipEventCreate"));
    if (const char* e = getenv("CUDASW4_AMD_PIPELINES")) { b->pipelines = !(e[0] == '0'); b->pipelinesAlways = std::string(e) == "always"; }
    if (const char* e = getenv("CUDASW4_AMD_PIPELINE_SHARE")) b->walkShare = std::max(0.01, atof(e));
    if (const char* e = getenv("CUDASW4_AMD_PIPELINE_RESCORE_SHARE")) b->rescoreShare = std::max(0.001, atof(e));
    if (const char* e = getenv("CUDASW4_AMD_WINDOWS")) b->windows = std::string(e) == "always" ? 2 : (e[0] == '0' ? 0 : 1);
    if (const char* e = getenv("CUDASW4_AMD_RESCORE_SERVICE")) b->svcForce = e[0] == '1' ? 1 : 0;
    if (const char* e = getenv("CUDASW4_AMD_SPLIT34_MAX_LANES")) b->split34MaxLanes = atoi(e);
    if (const char* e = getenv("CUDASW4_AMD_TEST_LOSE_SIDE_LAUNCH")) b->testLoseSide = std::max(0, atoi(e));
    // start handshake: needs wait-value packets and signal memory; probed once, bounded (sw_probe_handshake).  Where kernels
    // are serialised — rocprofv3 counter collection, AMD_SERIALIZE_KERNEL, HIP_LAUNCH_BLOCKING — a launch that waits for a
    // value only device code raises would hang its stream without a diagnostic: plain stream order there
    int canWait = 0;
    (void)hipDeviceGetAttribute(&canWait, hipDeviceAttributeCanUseStreamWaitValue, dev);
    b->handshake = canWait != 0 && !env_on("CUDASW4_AMD_NO_HANDSHAKE");
    if (b->handshake) {
        if (hipExtMallocWithFlags(reinterpret_cast<void**>(&b->startSignal), 8, hipMallocSignalMemory) != hipSuccess ||
            hipExtMallocWithFlags(reinterpret_cast<void**>(&b->doneSignal), 8, hipMallocSignalMemory) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(b->startSignal);
            b->startSignal = nullptr;
            b->doneSignal = nullptr;
            b->handshake = false;
        } else {
            *b->startSignal = 0;
            *b->doneSignal = 0;
            if (hipStreamCreateWithPriority(&b->svc, hipStreamNonBlocking, prioHigh) != hipSuccess) return bail(swi::fail(SW_ERR_HIP, "hipStreamCreateWithPriority"));
        }
    }
    hipStream_t work = static_cast<hipStream_t>(work_stream);
    if (b->svc) {
        // the service's polling kernel must not share a hardware queue with the side launches the bulk launch waits for
        b->svcConcurrent = sw_streams_run_concurrently(ctx, b->svc, b->aux[0]) == 1 && sw_streams_run_concurrently(ctx, b->svc, b->aux[1]) == 1 &&
                           sw_streams_run_concurrently(ctx, b->svc, work) == 1;
    }
    if (b->handshake) {
        if (env_on("ROCPROF_COUNTER_COLLECTION") || env_on("AMD_SERIALIZE_KERNEL") || env_on("HIP_LAUNCH_BLOCKING") ||
            sw_probe_handshake(ctx, b->aux[0], work, b->startSignal) != 1)
            b->handshake = false;
    }
    *out = b;
    return SW_OK;
}

int sw_batch_destroy(sw_batch* b) {
    if (!b) return SW_OK;
    (void)hipSetDevice(b->device);
    (void)hipDeviceSynchronize();
    for (int a = 0; a < kAux; a++) {
        if (b->aux[a]) (void)hipStreamDestroy(b->aux[a]);
        for (auto& wb : b->win[a]) {
            (void)hipHostFree(wb.h);
            (void)hipFree(wb.d);
            if (wb.copied) (void)hipEventDestroy(wb.copied);
        }
    }
    if (b->svc) (void)hipStreamDestroy(b->svc);
    for (hipEvent_t e : b->fork) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : b->joinEv) if (e) (void)hipEventDestroy(e);
    if (b->startSignal) (void)hipFree(b->startSignal);
    if (b->doneSignal) (void)hipFree(b->doneSignal);
    for (void* t : b->temp) (void)hipFree(t);
    delete b;
    return SW_OK;
}

int sw_batch_handshake_active(const sw_batch* b) { return b && b->handshake ? 1 : 0; }

int sw_batch_feedback(sw_batch* b, int32_t rescored) {
    if (!b) return swi::fail(SW_ERR_INVALID, "null engine");
    b->quietScans = rescored > 0 ? 0 : std::min(b->quietScans + 1, 1000);
    b->rescoredEma = 0.7 * b->rescoredEma + 0.3 * double(rescored);
    return SW_OK;
}

int sw_batch_stats(const sw_batch* b, int64_t* out, int n) {
    if (!b || !out) return swi::fail(SW_ERR_INVALID, "null argument");
    for (int i = 0; i < n; i++) out[i] = i < 6 ? b->stats[i] : 0;
    return SW_OK;
}

int sw_batch_signal_state(const sw_batch* b, uint32_t* start_now, uint32_t* start_target, uint32_t* done_now, uint32_t* done_target) {
    if (!b) return swi::fail(SW_ERR_INVALID, "null engine");
    if (start_now) *start_now = b->startSignal ? *reinterpret_cast<volatile uint32_t*>(b->startSignal) : 0;
    if (start_target) *start_target = b->sideLaunches;
    if (done_now) *done_now = b->doneSignal ? *reinterpret_cast<volatile uint32_t*>(b->doneSignal) : 0;
    if (done_target) *done_target = b->doneSeq;
    return SW_OK;
}

int sw_batch_open_gates(sw_batch* b) {
    if (!b) return swi::fail(SW_ERR_INVALID, "null engine");
    if (b->startSignal) *reinterpret_cast<volatile uint32_t*>(b->startSignal) = b->sideLaunches;
    if (b->doneSignal) *reinterpret_cast<volatile uint32_t*>(b->doneSignal) = b->doneSeq;
    for (bool& u : b->auxUsed) u = false;
    b->svcUsed = false;
    return SW_OK;
}

int sw_batch_reset(sw_batch* b) {
    if (!b) return swi::fail(SW_ERR_INVALID, "null engine");
    if (b->startSignal) *reinterpret_cast<volatile uint32_t*>(b->startSignal) = 0;
    b->sideLaunches = 0;
    for (bool& u : b->auxUsed) u = false;
    b->svcUsed = false;
    return SW_OK;
}

int sw_batch_test_lose_side_launch(sw_batch* b, int nth) {
    if (!b) return swi::fail(SW_ERR_INVALID, "null engine");
    b->testLoseSide = std::max(0, nth);
    return SW_OK;
}

int sw_batch_join(sw_batch* b, void* stream_) {
    if (!b) return swi::fail(SW_ERR_INVALID, "null engine");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    for (int a = 0; a < kAux; a++) {
        if (!b->auxUsed[a]) continue;
        SWB_HIP(hipEventRecord(b->joinEv[a], b->aux[a]));
        SWB_HIP(hipStreamWaitEvent(stream, b->joinEv[a], 0));
        b->auxUsed[a] = false;
    }
    if (b->svcUsed) {
        SWB_HIP(hipEventRecord(b->joinEv[kAux], b->svc));
        SWB_HIP(hipStreamWaitEvent(stream, b->joinEv[kAux], 0));
        b->svcUsed = false;
    }
    return SW_OK;
}

int sw_batch_side_events(sw_batch* b, void* const* events, int* used) {
    if (!b || !events || !used) return swi::fail(SW_ERR_INVALID, "null argument");
    for (int a = 0; a < kAux; a++) {
        used[a] = b->lastAux[a] ? 1 : 0;
        if (b->lastAux[a]) SWB_HIP(hipEventRecord(static_cast<hipEvent_t>(events[a]), b->aux[a]));
    }
    used[kAux] = b->lastSvc ? 1 : 0;
    if (b->lastSvc) SWB_HIP(hipEventRecord(static_cast<hipEvent_t>(events[kAux]), b->svc));
    return SW_OK;
}

// The partition walk of runAlignmentKernels (cudasw4.cuh:1742-2103) over the positions [begin, end): largest partition
// first, adjacent partitions of equal kind merged into one launch.  Partition 35 always keeps a launch of its own (few
// giant subjects: wave-wide groups, a side stream); partition 34 keeps one while it holds fewer than mergeMin subjects of
// the range and otherwise MERGES with the partitions below it when the kinds agree (one grid takes the long subjects first
// instead of two grids sharing the CUs).  A merged run reports part_id 33.  (The host driver's planner of rounds 2-5,
// cudasw4_amd/csrc/host/search_driver.hpp: plan_launch_runs — the CPU tests compare the two.)
static std::vector<Run> plan_runs(const int kinds[4], const int32_t* pb, const int32_t* pmax, int32_t begin, int32_t end, int64_t mergeMin) {
    std::vector<Run> runs;
    auto kind_of = [&](int p) { return p < kSmallLong ? kinds[0] : p == kSmallLong ? kinds[1] : kinds[2]; };
    int lastShape = -1;
    for (int p = kParts - 1; p >= 0; p--) {
        const int32_t bb = std::max(begin, pb[p]), ee = std::min(end, pb[p + 1]);
        if (ee <= bb) continue;
        const int kind = kind_of(p);
        const int shape = p == kLargeLong ? 2 : (p == kSmallLong && int64_t(ee - bb) < mergeMin) ? 1 : 0;
        if (!runs.empty() && runs.back().kind == kind && runs.back().begin == ee && shape == 0 && lastShape == 0) {
            runs.back().begin = bb;
        } else {
            runs.push_back(Run{kind, (p == kSmallLong && shape == 0) ? kSmallLong - 1 : p, bb, ee, pmax[p]});
        }
        lastShape = shape;
    }
    return runs;
}

}  // extern "C"

namespace {

// One batch on its way through the engine: what sw_scan_batch was given, what plan() decided, and the launches in the order
// they go in — side work (launch_side), the bulk launch behind its gates (launch_bulk), the re-scores (launch_rescores).
struct PipePart { int32_t begin, end; int part_id; int32_t maxlen; size_t need; };
struct BatchJob {
    sw_batch* const b;
    const sw_batch_args* const a;
    sw_ctx* const ctx;
    const int32_t qlen;
    const hipStream_t work;
    const int workTemp;
    const hipEvent_t fork;
    const size_t tempCap;
    const int gop, gex;
    const int32_t* const pb;
    const int32_t n, b34, b35;
    int32_t* const counters;
    // plan()
    double colSeconds = 0.0, bulkSeconds = 0.0;
    bool pipelineOk = false, split34 = false, shareLast = false, useService = false;
    int32_t cut34 = 0, cut35 = 0;
    std::vector<PipePart> pipeParts;
    std::vector<Run> runs;
    size_t mainIdx = 0;
    std::vector<int> ovfList;
    // launches
    int recUsed = 0;
    int auxNext = 0;
    bool auxBusy[kAux] = {};
    std::vector<int> streamOf;
    bool anySide = false;

    BatchJob(sw_batch* b_, const sw_batch_args* a_, int32_t qlen_)
        : b(b_), a(a_), ctx(b_->ctx), qlen(qlen_), work(static_cast<hipStream_t>(a_->stream)), workTemp(a_->work_slot ? kAux + 1 : 0),
          fork(b_->fork[a_->work_slot ? 1 : 0]), tempCap(a_->max_temp_bytes ? a_->max_temp_bytes : (size_t(4) << 30)), gop(a_->gop), gex(a_->gex),
          pb(a_->part_begin), n(a_->n), b34(std::min(a_->n, a_->part_begin[kSmallLong])), b35(std::min(a_->n, a_->part_begin[kLargeLong])),
          counters(a_->counters) {}

    // true length of a position of the long partitions (ascending inside a partition)
    int32_t length_at(int32_t pos) const {
        if (a->long_lengths && pos >= b34) return a->long_lengths[pos - b34];
        int p = 0;
        while (p < kParts - 1 && pos >= pb[p + 1]) p++;
        return a->part_maxlen[p];
    }

    // ---- plan: host arithmetic only (and the library's planning queries) — which subjects are pipelined, which runs there are,
    // which of them is the bulk launch, whether the re-score service runs beside it
    int plan() {
        int32_t batchMax = 0;
        for (int p = 0; p < kParts; p++) {
            if (pb[p + 1] <= pb[p]) continue;
            // (the long partitions: the host view knows better than a nominal boundary)
            const int32_t m = a->long_lengths && p >= kSmallLong ? length_at(std::min(n, pb[p + 1]) - 1) : a->part_maxlen[p];
            if (m < 0 || m > SW_MAX_SUBJECT_LEN)
                return swi::fail(SW_ERR_INVALID, "sw_scan_batch: part_maxlen[" + std::to_string(p) + "] = " + std::to_string(m) +
                                                     ": pass the longest subject of the partition (lengths[last]), not its nominal boundary");
            batchMax = std::max(batchMax, m);
        }

        // ---- which subjects leave the scan launches (pipelines of one-wave stages)
        // a wave-wide group's step of R rows per lane: ~(6.5 R + 19) instructions at ~6 cycles each beside a busy grid
        const double qrows64 = std::ceil(double(qlen) / 64.0);
        colSeconds = std::ceil(qrows64 / 8.0) * (6.5 * std::min(qrows64, 8.0) + 19.0) * 6.0 / 2.4e9;
        bulkSeconds = double(a->batch_bytes) * double(qlen) / 10e12;
        pipelineOk = b->pipelines && gop <= gex && a->long_lengths != nullptr && b->windows != 2;
        cut34 = b35; cut35 = n;   // the scan launches cover [0, cut34) and [b35, cut35)
        if (pipelineOk) {
            auto first_longer = [&](int32_t lo, int32_t hi, double maxWalk) {
                while (lo < hi) {
                    const int32_t mid = lo + (hi - lo) / 2;
                    if (double(length_at(mid)) > maxWalk) hi = mid; else lo = mid + 1;
                }
                return lo;
            };
            const double share35 = packed(a->kinds[2]) ? b->walkShare : std::max(b->walkShare, b->walkShareFinal);
            const double walks = bulkSeconds / colSeconds;
            cut34 = std::max(first_longer(b34, b35, b->pipelinesAlways ? 8000.0 : b->walkShare * walks), b35 - std::min(b35 - b34, b->pipelineMaxSubjects));
            cut35 = std::max(first_longer(b35, n, b->pipelinesAlways ? 8000.0 : share35 * walks), n - std::min(n - b35, b->pipelineMaxSubjects));
            if (int64_t(batchMax) * int64_t(-gex) >= (int64_t(1) << 28)) { cut34 = b35; cut35 = n; }
        }
        {
            const int32_t pbeg[2] = {cut35, cut34}, pend[2] = {n, b35};
            for (int k = 0; k < 2; k++) {
                if (pend[k] <= pbeg[k]) continue;
                const int32_t maxlen = length_at(pend[k] - 1);
                const size_t need = sw_scan_rows_pipelined_temp_bytes(ctx, pend[k] - pbeg[k], maxlen);
                if (need == 0 || need > tempCap || pipeParts.size() >= size_t(kAux)) continue;
                pipeParts.push_back(PipePart{pbeg[k], pend[k], kLargeLong - k, maxlen, need});
            }
        }
        cut34 = b35; cut35 = n;
        for (const PipePart& pp : pipeParts) (pp.part_id == kLargeLong ? cut35 : cut34) = pp.begin;

        // ---- the runs.  Below a bulk launch on 4-lane groups (very short queries) partition 34 keeps a launch of its own on
        // 16-lane groups: a quad's column costs ~(6.5 R + 19) instructions with R a quarter of the query, and the longest
        // subject's walk bounds a launch
        split34 = false;
        if (b35 > b34 && b34 > 0 && a->kinds[0] == a->kinds[1]) {
            int32_t ek = 0, r33 = 0, ns33 = 0, l33 = 16;
            SWB_OK(sw_plan_launch(ctx, a->kinds[1], kSmallLong - 1, b34, length_at(b34 - 1), &ek, &r33, &ns33, &l33));
            split34 = l33 <= b->split34MaxLanes;
        }
        SWB_OK(sw_set_long16_min(ctx, -1));
        const int64_t mergeMin = split34 ? INT64_MAX : kLongPartitionMergeMin;
        std::vector<int32_t> pmax(a->part_maxlen, a->part_maxlen + kParts);
        if (a->long_lengths) {   // the long partitions' runs end at their cut: the host view has the length there
            if (cut34 > b34) pmax[kSmallLong] = length_at(cut34 - 1);
            if (cut35 > b35) pmax[kLargeLong] = length_at(cut35 - 1);
        }
        if (cut34 == b35) {
            runs = plan_runs(a->kinds, pb, pmax.data(), 0, cut35, mergeMin);
        } else {   // the longest of partition 34 are pipelined, the shortest of partition 35 are not: two ranges
            runs = plan_runs(a->kinds, pb, pmax.data(), 0, cut34, mergeMin);
            if (cut35 > b35) {
                const auto tail = plan_runs(a->kinds, pb, pmax.data(), b35, cut35, mergeMin);
                runs.insert(runs.end(), tail.begin(), tail.end());
            }
        }
        mainIdx = 0;
        for (size_t i = 1; i < runs.size(); i++)
            if (runs[i].end - runs[i].begin > runs[mainIdx].end - runs[mainIdx].begin) mainIdx = i;
        // a pipeline stage takes exactly the register-file slot of a wave of the bulk launch it runs beside, so that it leaves no
        // hole behind in which no wave of that persistent grid fits (sw_launch_vgpr_slot)
        if (pipelineOk) {
            int vslot = 0;
            if (!runs.empty()) {
                const Run& m = runs[mainIdx];
                vslot = sw_launch_vgpr_slot(ctx, m.kind, m.part_id, m.end - m.begin, m.maxlen);
            }
            SWB_OK(sw_set_rows_pipeline_slot(ctx, vslot));
        }
        ovfList.assign(runs.size(), -1);
        int numLists = 0;
        for (size_t i = 0; i < runs.size(); i++)
            if (packed(runs[i].kind)) ovfList[i] = std::min(numLists++, kLists - 1);
        shareLast = numLists > kLists;   // (never with the reference's partitions: at most three runs per batch)

        // ---- re-score service for the bulk run's overflow list: a few workgroups that re-score the list while the bulk launch
        // fills it (sw_rescore_service); worth it where one re-scored subject takes about as long as a launch does at all
        const bool serviceWanted = b->svcForce >= 0 ? b->svcForce == 1 : b->quietScans < 3;
        useService = b->handshake && b->svc && b->svcConcurrent && a->allow_service && serviceWanted && !runs.empty() &&
                                packed(runs[mainIdx].kind) && double(qlen) * double(runs[mainIdx].maxlen) >= 5e5;
        return SW_OK;
    }

    int rec_begin(hipStream_t stream, bool onWork, int kind, int part_id, int32_t begin, int32_t end, int32_t maxlen, bool rescore, sw_launch_record** out) {
        *out = nullptr;
        const bool record = a->records && (a->record_mode == 1 || (a->record_mode == 2 && onWork)) && recUsed < a->records_cap;
        if (!record) return SW_OK;
        sw_launch_record* r = &a->records[recUsed++];
        r->kind = kind; r->part_id = part_id; r->begin = begin; r->end = end; r->rescore = rescore ? 1 : 0;
        SWB_OK(sw_plan_launch(ctx, kind, part_id, end - begin, maxlen, &r->eff_kind, &r->rows, &r->nstripes, &r->lanes));
        SWB_HIP(hipEventRecord(static_cast<hipEvent_t>(r->ev0), stream));
        *out = r;
        return SW_OK;
    }
    int rec_end(sw_launch_record* r, hipStream_t stream) {
        if (r) SWB_HIP(hipEventRecord(static_cast<hipEvent_t>(r->ev1), stream));
        return SW_OK;
    }
    int launch(size_t ri, hipStream_t stream, int tslot) {
        const Run& r = runs[ri];
        const int32_t cnt = r.end - r.begin;
        void* temp = nullptr;
        SWB_OK(ensure_temp(b, tslot, sw_scan_temp_bytes(ctx, r.kind, r.part_id, cnt, r.maxlen), tempCap, &temp));
        sw_launch_record* rec = nullptr;
        SWB_OK(rec_begin(stream, tslot == workTemp, r.kind, r.part_id, r.begin, r.end, r.maxlen, false, &rec));
        const bool pk = ovfList[ri] >= 0;
        SWB_OK(sw_scan_partition(ctx, r.kind, r.part_id, a->chars, a->offsets, a->lengths, r.begin, cnt, r.maxlen, gop, gex, a->scores, a->ids,
                                 a->id_offset, pk ? a->ovf_pos + r.begin : nullptr, pk ? counters + SW_BATCH_CNT_LIST0 + ovfList[ri] : nullptr,
                                 pk ? 1 : 0, temp, b->tempBytes[tslot], stream));
        return rec_end(rec, stream);
    }
    // rough clocks of a side launch of long subjects and of the bulk launch beside it
    double giant_seconds(const Run& r) const {
        const double rows = std::ceil(double(qlen) / 64.0), stripes = std::ceil(rows / 8.0);
        return double(r.maxlen) * stripes * (6.5 * std::min(rows, 8.0) + 19.0) * 8.0 / 2.4e9;
    }
    int launch_pipeline(const PipePart& pp, hipStream_t stream, int tslot) {
        const int32_t cnt = pp.end - pp.begin;
        void* temp = nullptr;
        SWB_OK(ensure_temp(b, tslot, pp.need, tempCap, &temp));
        // the reference's statistic counts the subjects whose exact score reaches the packed kind's limit — also where no
        // packed launch ever saw them
        const int pk = pp.part_id == kSmallLong ? a->kinds[1] : a->kinds[2];
        const int32_t limit = pk == SW_KIND_F16X2 ? SW_MAX_ACC_F16 : pk == SW_KIND_I16X2 ? SW_MAX_ACC_I16 : 0;
        sw_launch_record* rec = nullptr;
        const bool record = a->records && a->record_mode == 1 && recUsed < a->records_cap;
        if (record) {
            rec = &a->records[recUsed++];
            rec->kind = pk; rec->part_id = pp.part_id; rec->begin = pp.begin; rec->end = pp.end; rec->rescore = 0;
            rec->eff_kind = SW_KIND_I32; rec->rows = (pp.maxlen + 1023) / 1024; rec->nstripes = 1; rec->lanes = 64;
            SWB_HIP(hipEventRecord(static_cast<hipEvent_t>(rec->ev0), stream));
        }
        SWB_OK(sw_scan_rows_pipelined(ctx, a->chars, a->offsets, a->lengths, pp.begin, cnt, pp.maxlen, gop, gex, a->scores, a->ids, a->id_offset,
                                      counters + SW_BATCH_CNT_FAILED, limit > 0 ? counters + SW_BATCH_CNT_OVERFLOWS : nullptr,
                                      limit > 0 ? counters + SW_BATCH_CNT_PIPE_OVER : nullptr, limit, temp, b->tempBytes[tslot], stream));
        SWB_OK(rec_end(rec, stream));
        b->stats[0]++;
        return SW_OK;
    }
    // A side launch of a 32-bit kind whose long subjects the current query allows to cut into windows (sw_window_overlap:
    // exact — an alignment with a positive score spans fewer than W subject columns).  *done = false: launch the run as it is.
    int launch_windows(size_t ri, hipStream_t stream, int ax, bool* done) {
        *done = false;
        const Run& r = runs[ri];
        if (b->windows == 0 || packed(r.kind) || !a->long_lengths || !a->long_offsets || r.begin < b34) return SW_OK;
        const int32_t W = sw_window_overlap(ctx, gop, gex);
        if (W < 0) return SW_OK;
        const int64_t W4 = (int64_t(W) + 3) / 4 * 4, C = std::max<int64_t>(W4, 2048);
        if (int64_t(r.maxlen) <= C + W4) return SW_OK;   // nothing to cut
        // worth it only while the longest subject's lone group would outlast the bulk launch (windows double the side
        // launch's work): short queries
        if (b->windows != 2 && giant_seconds(r) < 1.15 * bulkSeconds) return SW_OK;
        const size_t nreal = size_t(r.end - r.begin);
        size_t nwin = 0;
        for (int32_t pos = r.begin; pos < r.end; pos++) {
            const int64_t len = length_at(pos);
            nwin += len <= C + W4 ? 1 : size_t((len + C - 1) / C);
        }
        if (nwin > size_t(INT32_MAX) / 2) return SW_OK;
        auto al = [](size_t x) { return (x + 15) / 16 * 16; };
        const size_t offOff = 0, lenOff = al(nwin * 8), firstOff = lenOff + al(nwin * 4), posOff = firstOff + al((nreal + 1) * 4);
        const size_t hostBytes = posOff + al(nreal * 4);
        const size_t scoreOff = hostBytes, idOff = scoreOff + al(nwin * 4), devBytes = idOff + al(nwin * 4);
        WindowBuf& wb = b->win[ax][b->winNext[ax]];
        b->winNext[ax] = (b->winNext[ax] + 1) % kWindowBufs;
        if (!wb.copied) SWB_HIP(hipEventCreateWithFlags(&wb.copied, hipEventDisableTiming));
        if (wb.used) SWB_HIP(hipEventSynchronize(wb.copied));
        if (hostBytes > wb.hcap) {
            (void)hipHostFree(wb.h); wb.h = nullptr; wb.hcap = 0;
            SWB_HIP(hipHostMalloc(reinterpret_cast<void**>(&wb.h), hostBytes * 2));
            wb.hcap = hostBytes * 2;
        }
        if (devBytes > wb.dcap) {
            (void)hipFree(wb.d); wb.d = nullptr; wb.dcap = 0;
            SWB_HIP(hipMalloc(reinterpret_cast<void**>(&wb.d), devBytes * 2));
            wb.dcap = devBytes * 2;
        }
        uint64_t* hOff = reinterpret_cast<uint64_t*>(wb.h + offOff);
        int32_t* hLen = reinterpret_cast<int32_t*>(wb.h + lenOff);
        int32_t* hFirst = reinterpret_cast<int32_t*>(wb.h + firstOff);
        int32_t* hPos = reinterpret_cast<int32_t*>(wb.h + posOff);
        size_t w = 0;
        int32_t maxWin = 0;
        for (int32_t pos = r.begin; pos < r.end; pos++) {
            const int64_t len = length_at(pos);
            const uint64_t base = a->long_offsets[pos - b34] - a->long_offsets_bias;   // the subject's first byte, relative to `chars`
            hFirst[pos - r.begin] = int32_t(w);
            hPos[pos - r.begin] = pos;
            const int64_t k = len <= C + W4 ? 1 : (len + C - 1) / C;
            for (int64_t i = 0; i < k; i++) {
                const int64_t wb0 = k == 1 ? 0 : std::max<int64_t>(0, i * C - W4), we = k == 1 ? len : std::min(len, (i + 1) * C);
                hOff[w] = base + uint64_t(wb0);
                hLen[w] = int32_t(we - wb0);
                maxWin = std::max(maxWin, hLen[w]);
                w++;
            }
        }
        hFirst[nreal] = int32_t(w);
        SWB_HIP(hipMemcpyAsync(wb.d, wb.h, hostBytes, hipMemcpyHostToDevice, stream));
        SWB_HIP(hipEventRecord(wb.copied, stream));
        wb.used = true;
        const int32_t cnt = int32_t(nwin);
        const int tslot = ax + 1;
        void* temp = nullptr;
        SWB_OK(ensure_temp(b, tslot, sw_scan_temp_bytes(ctx, r.kind, r.part_id, cnt, maxWin), tempCap, &temp));
        sw_launch_record* rec = nullptr;
        if (a->records && a->record_mode == 1 && recUsed < a->records_cap) {
            rec = &a->records[recUsed++];
            rec->kind = r.kind; rec->part_id = r.part_id; rec->begin = r.begin; rec->end = r.end; rec->rescore = 0;
            SWB_OK(sw_plan_launch(ctx, r.kind, r.part_id, cnt, maxWin, &rec->eff_kind, &rec->rows, &rec->nstripes, &rec->lanes));
            SWB_HIP(hipEventRecord(static_cast<hipEvent_t>(rec->ev0), stream));
        }
        // the windows are subjects of their own: offsets relative to the first window's, whose first byte `chars + hOff[0]` is
        SWB_OK(sw_scan_partition(ctx, r.kind, r.part_id, a->chars + hOff[0], reinterpret_cast<const uint64_t*>(wb.d + offOff),
                                 reinterpret_cast<const int32_t*>(wb.d + lenOff), 0, cnt, maxWin, gop, gex, reinterpret_cast<float*>(wb.d + scoreOff),
                                 reinterpret_cast<int32_t*>(wb.d + idOff), 0, nullptr, nullptr, 0, temp, b->tempBytes[tslot], stream));
        SWB_OK(sw_reduce_windows(ctx, reinterpret_cast<const float*>(wb.d + scoreOff), reinterpret_cast<const int32_t*>(wb.d + firstOff),
                                 reinterpret_cast<const int32_t*>(wb.d + posOff), int32_t(nreal), a->scores, a->ids, a->id_offset, stream));
        SWB_OK(rec_end(rec, stream));
        b->stats[2]++;
        b->stats[3] += int64_t(nwin);
        *done = true;
        return SW_OK;
    }
    int rescore(size_t ri, hipStream_t stream, int tslot) {   // cudasw4.cuh:2134-2169
        if (ovfList[ri] < 0) return SW_OK;
        const Run& r = runs[ri];
        const int32_t cnt = r.end - r.begin;
        const int okind = a->kinds[3];
        const int32_t limit = r.kind == SW_KIND_F16X2 ? SW_MAX_ACC_F16 : SW_MAX_ACC_I16;
        int32_t* const list = a->ovf_pos + r.begin;
        int32_t* const count = counters + SW_BATCH_CNT_LIST0 + ovfList[ri];
        void* temp = nullptr;
        const size_t need = sw_scan_temp_bytes(ctx, okind, -1, cnt, r.maxlen);
        SWB_OK(ensure_temp(b, tslot, need, tempCap, &temp));
        sw_launch_record* rec = nullptr;
        SWB_OK(rec_begin(stream, tslot == workTemp, okind, -1, r.begin, r.end, r.maxlen, true, &rec));
        // the long subjects of the list first, pipelined: a flagged subject is one group's walk, and on a shard that walk
        // outlasts the bulk launch; the ordinary launch behind claims what is left
        bool picked = false;
        if (pipelineOk) {
            const double minLen = std::max(256.0, b->rescoreShare * bulkSeconds / colSeconds);
            const size_t need2 = double(r.maxlen) >= minLen && int64_t(r.maxlen) * int64_t(-gex) < (int64_t(1) << 28)
                                     ? sw_rescore_overflow_pipelined_temp_bytes(ctx, r.maxlen) : 0;
            if (need2 > 0 && need2 <= tempCap) {
                SWB_OK(ensure_temp(b, tslot, std::max(need, need2), tempCap, &temp));
                SWB_OK(sw_rescore_overflow_pipelined(ctx, list, count, cnt, a->chars, a->offsets, a->lengths, r.maxlen, int32_t(minLen), gop, gex,
                                                     a->scores, a->ids, a->id_offset, counters + SW_BATCH_CNT_FAILED, limit,
                                                     counters + SW_BATCH_CNT_OVERFLOWS, temp, b->tempBytes[tslot], stream));
                picked = true;
                b->stats[1]++;
            }
        }
        if ((useService && ri == mainIdx) || picked)   // what the service / the pipelined launch has not taken
            SWB_OK(sw_rescore_overflow_claim(ctx, okind, list, count, cnt, a->chars, a->offsets, a->lengths, r.maxlen, gop, gex, a->scores, a->ids,
                                             a->id_offset, temp, b->tempBytes[tslot], limit, counters + SW_BATCH_CNT_OVERFLOWS, stream));
        else
            SWB_OK(sw_rescore_overflow_stat(ctx, okind, list, count, cnt, a->chars, a->offsets, a->lengths, r.maxlen, gop, gex, a->scores, a->ids,
                                            a->id_offset, temp, b->tempBytes[tslot], limit, counters + SW_BATCH_CNT_OVERFLOWS, stream));
        return rec_end(rec, stream);
    }


    int launch_side() {
        if (useService) {
            const Run& r = runs[mainIdx];   // the list starts empty (-1) for the compare-and-swap of its takers
            SWB_HIP(hipMemsetAsync(a->ovf_pos + r.begin, 0xFF, size_t(r.end - r.begin) * sizeof(int32_t), work));
        }
        if (runs.size() > 1 || useService || !pipeParts.empty()) SWB_HIP(hipEventRecord(fork, work));

        // ---- side work first: pipeline parts, then the side runs, then the service; every one of them adds 1 to the start signal
        // when its workgroups are resident
        auxNext = a->alt_side_stream ? 1 : 0;
        streamOf.assign(runs.size(), -1);   // auxiliary stream of a run, -1: work stream
        anySide = false;
        // The bulk launch waits until every side launch holds its slots, and a stream runs its launches in order: with two
        // pipeline parts and two side runs on two auxiliary streams the last side launch starts behind a pipeline part.  When
        // the re-score service is not in play its stream takes the first pipeline part.
        const bool svcFree = b->svc && b->svcConcurrent && b->handshake && !useService && a->allow_service &&
                             pipeParts.size() + (runs.empty() ? 0 : runs.size() - 1) > size_t(kAux);
        bool svcTaken = false;
        for (const PipePart& pp : pipeParts) {
            if (b->handshake && !runs.empty()) {
                SWB_OK(sw_set_start_signal(ctx, b->startSignal));
                b->sideLaunches++;
                anySide = true;
            }
            if (svcFree && !svcTaken) {
                svcTaken = true;
                SWB_HIP(hipStreamWaitEvent(b->svc, fork, 0));
                SWB_OK(launch_pipeline(pp, b->svc, kAux + 2));
                b->svcUsed = b->lastSvc = true;
                continue;
            }
            const int ax = auxNext++ % kAux;
            if (!auxBusy[ax]) SWB_HIP(hipStreamWaitEvent(b->aux[ax], fork, 0));
            auxBusy[ax] = true;
            SWB_OK(launch_pipeline(pp, b->aux[ax], ax + 1));
        }
        for (size_t i = 0; i < runs.size(); i++) {
            if (i == mainIdx || (shareLast && ovfList[i] == kLists - 1)) continue;
            const int ax = auxNext++ % kAux;
            if (!auxBusy[ax]) SWB_HIP(hipStreamWaitEvent(b->aux[ax], fork, 0));
            auxBusy[ax] = true;
            streamOf[i] = ax;
            if (b->handshake) {
                SWB_OK(sw_set_start_signal(ctx, b->startSignal));
                b->sideLaunches++;
                anySide = true;
            }
            b->stats[5]++;
            if (b->testLoseSide > 0 && --b->testLoseSide == 0) {   // (tests of a caller's watchdog: counted, never enqueued)
                SWB_OK(sw_set_start_signal(ctx, nullptr));
                continue;
            }
            bool windowed = false;
            SWB_OK(launch_windows(i, b->aux[ax], ax, &windowed));
            if (!windowed) SWB_OK(launch(i, b->aux[ax], ax + 1));
        }
        if (useService) {
            const Run& r = runs[mainIdx];
            const int tslot = kAux + 2;
            void* temp = nullptr;
            SWB_OK(ensure_temp(b, tslot, sw_rescore_service_temp_bytes(ctx, a->kinds[3], r.maxlen, b->serviceWorkgroups()), tempCap, &temp));
            SWB_HIP(hipStreamWaitEvent(b->svc, fork, 0));
            SWB_OK(sw_set_start_signal(ctx, b->startSignal));
            b->sideLaunches++;
            anySide = true;
            b->doneSeq++;
            SWB_OK(sw_rescore_service(ctx, a->kinds[3], a->ovf_pos + r.begin, counters + SW_BATCH_CNT_LIST0 + ovfList[mainIdx], r.end - r.begin, a->chars,
                                      a->offsets, a->lengths, r.maxlen, gop, gex, a->scores, a->ids, a->id_offset, temp, b->tempBytes[tslot],
                                      r.kind == SW_KIND_F16X2 ? SW_MAX_ACC_F16 : SW_MAX_ACC_I16, counters + SW_BATCH_CNT_OVERFLOWS, b->doneSignal,
                                      b->doneSeq, b->serviceWorkgroups(), b->svc));
            b->svcUsed = b->lastSvc = true;
            b->stats[4]++;
        }
        return SW_OK;
    }

    int launch_bulk() {
        SWB_OK(sw_set_grid_reserve(ctx, (!pipeParts.empty() || runs.size() > 1) ? a->grid_reserve_side : 0));
        // ---- the bulk launch goes in only after the side launches hold their workgroup slots ...
        if (anySide) SWB_HIP(hipStreamWaitValue32(work, b->startSignal, b->sideLaunches, hipStreamWaitValueGte, 0xffffffffu));
        // ... and, when the query before is still running on the other lane, only when that one's work counter has run dry
        if (a->wait_signal && a->wait_value) SWB_HIP(hipStreamWaitValue32(work, a->wait_signal, a->wait_value, hipStreamWaitValueGte, 0xffffffffu));
        for (size_t i = 0; i < runs.size(); i++)
            if (streamOf[i] < 0) {
                if (i == mainIdx && a->arm_signal) SWB_OK(sw_set_dry_signal(ctx, a->arm_signal, a->arm_value));
                SWB_OK(launch(i, work, workTemp));
                // the service leaves once the list's producer has finished
                if (useService && i == mainIdx) SWB_HIP(hipStreamWriteValue32(work, b->doneSignal, b->doneSeq, 0));
            }
        return SW_OK;
    }

    int launch_rescores() {
        for (size_t i = 0; i < runs.size(); i++) {
            if (streamOf[i] >= 0) SWB_OK(rescore(i, b->aux[streamOf[i]], streamOf[i] + 1));
            else SWB_OK(rescore(i, work, workTemp));
        }
        return SW_OK;
    }
};

}  // namespace

extern "C" {

// what both entry points check first; *qlen == 0: an empty batch (nothing to do)
static int check_batch(sw_batch* b, const sw_batch_args* a, bool buffers, int32_t* qlen) {
    *qlen = 0;
    if (!b || !a) return swi::fail(SW_ERR_INVALID, "null argument");
    if (a->n < 0 || !a->part_begin || !a->part_maxlen) return swi::fail(SW_ERR_INVALID, "sw_scan_batch: partition tables missing");
    for (int i = 0; i < 4; i++)
        if (a->kinds[i] < 0 || a->kinds[i] > 3) return swi::fail(SW_ERR_INVALID, "sw_scan_batch: unknown kind");
    if (!packed(a->kinds[1]) || packed(a->kinds[2]) || packed(a->kinds[3]))
        return swi::fail(SW_ERR_INVALID, "sw_scan_batch: manyPass_small must be a packed kind, manyPass_large and overflow 32-bit kinds (cudasw4.cuh:841-855)");
    if (a->n == 0) return SW_OK;
    if (buffers && (!a->chars || !a->offsets || !a->lengths || !a->scores || !a->ids || !a->ovf_pos || !a->counters)) return swi::fail(SW_ERR_INVALID, "null buffer");
    const int32_t q = swi::query_length(b->ctx);
    if (q <= 0) return swi::fail(SW_ERR_NO_QUERY, "sw_set_query has not been called");
    *qlen = q;
    return SW_OK;
}

int sw_batch_describe_plan(sw_batch* b, const sw_batch_args* a, char* out, size_t cap) {
    if (!out || cap == 0) return swi::fail(SW_ERR_INVALID, "null argument");
    out[0] = 0;
    int32_t qlen = 0;
    SWB_OK(check_batch(b, a, false, &qlen));
    if (qlen == 0) return SW_OK;
    SWB_HIP(hipSetDevice(b->device));
    BatchJob job(b, a, qlen);
    SWB_OK(job.plan());
    std::string t;
    for (const PipePart& pp : job.pipeParts)
        t += "pipeline p" + std::to_string(pp.part_id) + " [" + std::to_string(pp.begin) + "," + std::to_string(pp.end) + ") maxlen " + std::to_string(pp.maxlen) + "; ";
    for (size_t i = 0; i < job.runs.size(); i++) {
        const Run& r = job.runs[i];
        t += std::string(i == job.mainIdx ? "bulk" : "side") + " kind " + std::to_string(r.kind) + " p" + std::to_string(r.part_id) + " [" + std::to_string(r.begin) + "," +
             std::to_string(r.end) + ") maxlen " + std::to_string(r.maxlen) + " list " + std::to_string(job.ovfList[i]) + "; ";
    }
    t += std::string("service ") + (job.useService ? "1" : "0") + "; split34 " + (job.split34 ? "1" : "0");
    std::snprintf(out, cap, "%s", t.c_str());
    return SW_OK;
}

int sw_scan_batch(sw_batch* b, const sw_batch_args* a) {
    int32_t qlen = 0;
    SWB_OK(check_batch(b, a, true, &qlen));
    b->lastAux[0] = b->lastAux[1] = b->lastSvc = false;
    if (qlen == 0) return SW_OK;
    SWB_HIP(hipSetDevice(b->device));
    BatchJob job(b, a, qlen);
    if (a->zero_counters) SWB_HIP(hipMemsetAsync(a->counters, 0, SW_BATCH_COUNTERS * sizeof(int32_t), job.work));
    SWB_OK(sw_set_dirty_counter(b->ctx, a->counters + SW_BATCH_CNT_DIRTY));
    SWB_OK(job.plan());
    SWB_OK(job.launch_side());
    SWB_OK(job.launch_bulk());
    SWB_OK(job.launch_rescores());
    for (int ax = 0; ax < kAux; ax++)
        if (job.auxBusy[ax]) b->auxUsed[ax] = b->lastAux[ax] = true;
    if (a->records_used) *a->records_used = job.recUsed;
    return SW_OK;
}

}  // extern "C"
