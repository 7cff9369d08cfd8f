// tests/boundary/binding_harness.cpp — TEST INFRASTRUCTURE: compiles the reference-side binding that INTEGRATION.md §2
// documents, against the reference's own headers (types.hpp, length_partitions.hpp, config.hpp — included from
// /root/reference/src where they lie, never copied) and include/cudasw4_amd.h, and links it against libcudasw4_amd.so.
// tests/test_boundary_cpu.py replaces the marker line below with the code block of INTEGRATION.md §2 VERBATIM, so the
// documented patch cannot rot.  The surrounding declarations stand for the members of the reference's CudaSW4 /
// GpuWorkingSet (cudasw4.cuh:251-480) that the patch touches; their types are the reference's.
#include <cstdio>
#include <cstdlib>
#include <vector>
#ifdef BINDING_ON_GPU   // the same block on real device buffers (tests/boundary/Makefile -> _build/binding_gpu)
#include <hip/hip_runtime_api.h>
#include <algorithm>
#include <chrono>
#include <cstring>
#include <fstream>
#include <string>
#endif

#include "types.hpp"              // cudasw4::KernelType, BlosumType, BLOSUM62_20
#include "length_partitions.hpp"  // cudasw4::getLengthPartitionBoundaries
#include "config.hpp"             // ReferenceIdT, SequenceLengthT
#include <cudasw4_amd.h>

using cudasw4::KernelType;
using cudasw4::ReferenceIdT;
using cudasw4::SequenceLengthT;

// the C ABI's kinds are the reference's KernelType values, its partition count the reference's
static_assert(int(KernelType::Half2) == SW_KIND_F16X2 && int(KernelType::DPXs16) == SW_KIND_I16X2 &&
              int(KernelType::DPXs32) == SW_KIND_I32 && int(KernelType::Float) == SW_KIND_F32, "kind numbering");
static_assert(cudasw4::getLengthPartitionBoundaries().size() == SW_NUM_LENGTH_PARTITIONS, "length partitions");
static_assert(cudasw4::BLOSUM62_20::dim == 21, "21-letter tables");
static_assert(sizeof(ReferenceIdT) == sizeof(int32_t) && sizeof(SequenceLengthT) == sizeof(int32_t), "index widths");

static void check(int rc) {  // the reference's CUERR convention: print and exit(1) (hpc_helpers/cuda_helpers.cuh:23-31)
    if (rc != SW_OK) {
        std::fprintf(stderr, "cudasw4_amd error %d: %s\n", rc, sw_last_error());
        std::exit(1);
    }
}

struct KernelTypeConfig { KernelType singlePassType, manyPassType_small, manyPassType_large, overflowType; };

struct DeviceBatchCopyToPinnedPlan { std::vector<int> h_partitionIds; std::vector<int> h_numPerPartition; };  // dbbatching.cuh:16-37

struct GpuWorkingSet {  // cudasw4.cuh:251-480 (the members the patch uses)
    float* d_scores; ReferenceIdT* d_ids; ReferenceIdT* d_overflow_positions; int* d_overflow_number;
    char* d_tempStorageHE; size_t numTempBytes;
};

#ifndef BINDING_ON_GPU
int main() {
    const int numGpus = 1;
    std::vector<int> deviceIds{0};
    std::vector<void*> gpuStreams{nullptr};
    const KernelTypeConfig kernelTypeConfig{KernelType::Half2, KernelType::Half2, KernelType::Float, KernelType::Float};
    const int numLengthPartitions = SW_NUM_LENGTH_PARTITIONS;
    const auto boundaries = cudasw4::getLengthPartitionBoundaries();
    auto kindForPartition = [&](int lp) {
        return int(lp < numLengthPartitions - 2 ? kernelTypeConfig.singlePassType
                   : lp == numLengthPartitions - 2 ? kernelTypeConfig.manyPassType_small : kernelTypeConfig.manyPassType_large);
    };
    DeviceBatchCopyToPinnedPlan plan{std::vector<int>(36), std::vector<int>(36, 0)};
    GpuWorkingSet ws{};
    const char* inputChars = nullptr; const size_t* inputOffsets = nullptr; const SequenceLengthT* inputLengths = nullptr;
    const size_t hostOffsetsStorage[2] = {0, 0};
    const size_t* hostOffsets = hostOffsetsStorage; const SequenceLengthT* hostLengths = nullptr;   // DBdataView::offsets() / lengths()
    const int8_t encodedQueryStorage[4] = {0, 1, 2, 3};
    const int8_t* encodedQuery = encodedQueryStorage;
    const SequenceLengthT queryLength = 4;
    const int gop = -11, gex = -1, gpu = 0;
    void* stream = nullptr;
    const int64_t globalOffsetOfBatch = 0;
    const int32_t maxOverflows = 0, maxLen = 0;
    const int64_t numResults = 0; const int results_per_query = 0;
    float* d_topS = nullptr; int32_t* d_topI = nullptr; void* d_tmp = nullptr; size_t tmpBytes = 0;
    static_assert(sizeof(size_t) == sizeof(uint64_t), "offsets are 64-bit");
#else
template <class T>
static std::vector<T> read_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) { std::fprintf(stderr, "cannot read %s\n", path.c_str()); std::exit(2); }
    const size_t bytes = size_t(f.tellg());
    std::vector<T> v(bytes / sizeof(T));
    f.seekg(0);
    f.read(reinterpret_cast<char*>(v.data()), std::streamsize(v.size() * sizeof(T)));
    return v;
}
#define HIPOK(x) do { if ((x) != hipSuccess) { std::fprintf(stderr, "%s failed\n", #x); std::exit(3); } } while (0)
template <class T>
static T* to_device(const std::vector<T>& v, size_t extra = 0) {
    T* d = nullptr;
    HIPOK(hipMalloc(reinterpret_cast<void**>(&d), (v.size() + extra) * sizeof(T) + 64));
    HIPOK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

// binding_gpu <dbdata prefix> <query = subject index>: the documented call sequence on the reference's dbdata files
int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: binding_gpu dbprefix subjectIndexAsQuery [DPX|-] [--bench queries.bin reps]\n"); return 2; }
    const char* benchQueries = nullptr;
    int benchReps = 2;
    for (int i = 3; i + 1 < argc; i++)
        if (std::string(argv[i]) == "--bench") { benchQueries = argv[i + 1]; if (i + 2 < argc) benchReps = std::atoi(argv[i + 2]); }
    const std::string prefix = argv[1];
    const auto hChars = read_file<char>(prefix + "0chars");
    const auto hOffsets = read_file<size_t>(prefix + "0offsets");
    const auto hLengths = read_file<SequenceLengthT>(prefix + "0lengths");
    const int numSubjects = int(hLengths.size());
    const int qi = std::atoi(argv[2]);
    const bool dpx = argc > 3 && std::string(argv[3]) == "DPX";
    const int numGpus = 1;
    std::vector<int> deviceIds{0};
    std::vector<void*> gpuStreams{nullptr};
    const KernelTypeConfig kernelTypeConfig = dpx ? KernelTypeConfig{KernelType::DPXs16, KernelType::DPXs16, KernelType::DPXs32, KernelType::DPXs32}
                                                  : KernelTypeConfig{KernelType::Half2, KernelType::Half2, KernelType::Float, KernelType::Float};
    const int numLengthPartitions = SW_NUM_LENGTH_PARTITIONS;
    const auto boundaries = cudasw4::getLengthPartitionBoundaries();
    auto kindForPartition = [&](int lp) {
        return int(lp < numLengthPartitions - 2 ? kernelTypeConfig.singlePassType
                   : lp == numLengthPartitions - 2 ? kernelTypeConfig.manyPassType_small : kernelTypeConfig.manyPassType_large);
    };
    // the batch plan of a DB that is one batch: subjects per length partition (dbdata.cpp:91-115)
    DeviceBatchCopyToPinnedPlan plan{std::vector<int>(36), std::vector<int>(36, 0)};
    for (int i = 0; i < numSubjects; i++) {
        int lp = 0;
        while (lp < numLengthPartitions - 1 && hLengths[size_t(i)] > boundaries[size_t(lp)]) lp++;
        plan.h_numPerPartition[size_t(lp)]++;
    }
    GpuWorkingSet ws{};
    HIPOK(hipSetDevice(0));
    const char* inputChars = to_device(hChars);
    const size_t* inputOffsets = to_device(hOffsets);
    const SequenceLengthT* inputLengths = to_device(hLengths);
    HIPOK(hipMalloc(reinterpret_cast<void**>(&ws.d_scores), size_t(numSubjects) * 4));
    HIPOK(hipMalloc(reinterpret_cast<void**>(&ws.d_ids), size_t(numSubjects) * 4));
    HIPOK(hipMalloc(reinterpret_cast<void**>(&ws.d_overflow_positions), size_t(numSubjects) * 4));
    HIPOK(hipMalloc(reinterpret_cast<void**>(&ws.d_overflow_number), SW_BATCH_COUNTERS * sizeof(int)));
    HIPOK(hipMemset(ws.d_overflow_number, 0, SW_BATCH_COUNTERS * sizeof(int)));
    const size_t* hostOffsets = hOffsets.data(); const SequenceLengthT* hostLengths = hLengths.data();   // DBdataView::offsets() / lengths()
    ws.numTempBytes = size_t(1) << 30;
    HIPOK(hipMalloc(reinterpret_cast<void**>(&ws.d_tempStorageHE), ws.numTempBytes));
    // the query: subject qi of the same files (already encoded with ConvertAA_20)
    const int8_t* encodedQuery = reinterpret_cast<const int8_t*>(hChars.data() + (hOffsets[size_t(qi)] - hOffsets[0]));
    const SequenceLengthT queryLength = hLengths[size_t(qi)];
    const int gop = -11, gex = -1, gpu = 0;
    void* stream = nullptr;
    const int64_t globalOffsetOfBatch = 0;
    const int32_t maxOverflows = numSubjects, maxLen = *std::max_element(hLengths.begin(), hLengths.end());
    const int64_t numResults = numSubjects; const int results_per_query = 5;
    float* d_topS = nullptr; int32_t* d_topI = nullptr; void* d_tmp = nullptr;
    size_t tmpBytes = sw_topk_temp_bytes(numResults, results_per_query);
    HIPOK(hipMalloc(reinterpret_cast<void**>(&d_topS), 64)); HIPOK(hipMalloc(reinterpret_cast<void**>(&d_topI), 64));
    HIPOK(hipMalloc(&d_tmp, tmpBytes + 64));
    static_assert(sizeof(size_t) == sizeof(uint64_t), "offsets are 64-bit");
#endif

//@@INTEGRATION_MD_PATCH@@

#ifdef BINDING_ON_GPU
    HIPOK(hipDeviceSynchronize());
    if (benchQueries) {
        // --bench: every query of the file through the documented per-query block, `reps` passes, wall clock around them
        // (top-K inside, like the host driver's benchmark); the launcher-by-launcher form of INTEGRATION.md beside it
        std::vector<std::vector<int8_t>> qs;
        {
            const auto raw = read_file<char>(benchQueries);
            size_t at = 0;
            while (at + 4 <= raw.size()) {
                int32_t len = 0;
                std::memcpy(&len, raw.data() + at, 4);
                at += 4;
                qs.emplace_back(raw.data() + at, raw.data() + at + len);
                at += size_t(len);
            }
        }
        double residues = 0, sumq = 0;
        for (SequenceLengthT l : hLengths) residues += double(l);
        for (auto& q : qs) sumq += double(q.size());
        auto oneByOne = [&](const int8_t* q, SequenceLengthT qlen) {   // INTEGRATION.md section 2, second block
            check(sw_set_query(swCtx[gpu], q, qlen, stream));
            HIPOK(hipMemsetAsync(ws.d_overflow_number, 0, sizeof(int), (hipStream_t)stream));
            for (int lp = numLengthPartitions - 1; lp >= 0; lp--) {
                const int n = plan.h_numPerPartition[size_t(lp)];  if (n == 0) continue;
                check(sw_scan_partition(swCtx[gpu], kindForPartition(lp), lp, (const int8_t*)inputChars, (const uint64_t*)inputOffsets, inputLengths,
                        partBegin[size_t(lp)], n, partMaxLen[size_t(lp)], gop, gex, ws.d_scores, ws.d_ids, globalOffsetOfBatch,
                        ws.d_overflow_positions, ws.d_overflow_number, 1, ws.d_tempStorageHE, ws.numTempBytes, stream));
            }
            check(sw_rescore_overflow(swCtx[gpu], dpx ? SW_KIND_I32 : SW_KIND_F32, ws.d_overflow_positions, ws.d_overflow_number, maxOverflows,
                    (const int8_t*)inputChars, (const uint64_t*)inputOffsets, inputLengths, maxLen, gop, gex,
                    ws.d_scores, ws.d_ids, globalOffsetOfBatch, ws.d_tempStorageHE, ws.numTempBytes, stream));
            check(sw_topk(swCtx[gpu], ws.d_scores, ws.d_ids, numResults, results_per_query, d_topS, d_topI, d_tmp, tmpBytes, stream));
        };
        for (int form = 0; form < 2; form++) {
            std::vector<float> top1(qs.size(), 0.0f);
            double best = 1e30;
            for (int rep = 0; rep < benchReps + 1; rep++) {   // (first pass: warm-up, and the top scores for the checker)
                HIPOK(hipDeviceSynchronize());
                const auto t0 = std::chrono::steady_clock::now();
                for (size_t i = 0; i < qs.size(); i++) {
                    if (form == 0) scanOneQuery(qs[i].data(), SequenceLengthT(qs[i].size()));
                    else oneByOne(qs[i].data(), SequenceLengthT(qs[i].size()));
                    if (rep == 0) { HIPOK(hipDeviceSynchronize()); HIPOK(hipMemcpy(&top1[i], d_topS, 4, hipMemcpyDeviceToHost)); }
                }
                HIPOK(hipDeviceSynchronize());
                const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (rep > 0) best = std::min(best, dt);
            }
            std::printf("BENCH %s gcups=%.1f seconds=%.4f queries=%zu TOP1", form == 0 ? "sw_scan_batch" : "one_launch_per_partition",
                        sumq * residues / 1e9 / best, best, qs.size());
            for (float v : top1) std::printf(" %d", int(v));
            std::printf("\n");
        }
        for (auto* c : swBatch) sw_batch_destroy(c);
        for (auto* c : swCtx) sw_ctx_destroy(c);
        std::puts("binding ok");
        return 0;
    }
    std::vector<float> sc(static_cast<size_t>(numSubjects), 0.0f), ts(5, 0.0f);
    std::vector<int32_t> ti(5, 0);
    int novf = 0;
    HIPOK(hipMemcpy(sc.data(), ws.d_scores, sc.size() * 4, hipMemcpyDeviceToHost));
    HIPOK(hipMemcpy(ts.data(), d_topS, 20, hipMemcpyDeviceToHost));
    HIPOK(hipMemcpy(ti.data(), d_topI, 20, hipMemcpyDeviceToHost));
    HIPOK(hipMemcpy(&novf, ws.d_overflow_number, 4, hipMemcpyDeviceToHost));
    std::printf("SCORES");
    for (float v : sc) std::printf(" %d", int(v));
    std::printf("\nTOP");
    for (int i = 0; i < 5; i++) std::printf(" %d:%d", int(ts[size_t(i)]), ti[size_t(i)]);
    std::printf("\nOVERFLOWS %d\n", novf);
#endif
    for (auto* c : swBatch) sw_batch_destroy(c);
    for (auto* c : swCtx) sw_ctx_destroy(c);
    std::puts("binding ok");
    return 0;
}
