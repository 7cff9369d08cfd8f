// search_driver.hpp — the C++ host driver: DB sharding over the GPUs of a node, residency, the
// partition walk, overflow re-score, per-GPU top-K and the host-side merge.
//
// Mirrors the public surface of the reference's class CudaSW4 (cudasw4.cuh:496-839):
//   SearchDriver(deviceIds, numTop, matrix, KernelTypeConfig, MemoryConfig, verbose)
//   setDatabase / prefetchDBToGpus / scan / totalTimerStart / totalTimerStop / printDBInfo ...
// but drives the GPUs only through the C ABI of include/cudasw4_amd.h.  One host thread issues
// asynchronous work to every GPU (as the reference does); results are merged on the host instead of
// peer copies to GPU 0 (cudasw4.cuh:1415-1463).
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <string_view>
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
    int numOverflows = 0;
    double seconds = 0;
    double gcups = 0;
};

struct ScanResult {  // cudasw4.cuh:82-86
    std::vector<int> scores;
    std::vector<int64_t> referenceIds;
    BenchmarkStats stats;
};

class SearchDriver {
public:
    SearchDriver(std::vector<int> deviceIds, int numTop, MatrixId matrix, KernelTypeConfig kernels, MemoryConfig memory,
                 bool verbose, int gop, int gex);
    ~SearchDriver();
    SearchDriver(const SearchDriver&) = delete;
    SearchDriver& operator=(const SearchDriver&) = delete;

    void setDatabase(std::shared_ptr<Database> db);  // cudasw4.cuh:552-568
    void prefetchDBToGpus();                         // cudasw4.cuh:651-696 (--uploadFull)
    void setNumTop(int k) { numTop_ = k; }           // cudasw4.cuh:574-587

    // raw residue letters, as read from the query file (cudasw4.cuh:698-765)
    ScanResult scan(const char* query, int32_t queryLength);

    void totalTimerStart();          // cudasw4.cuh:818-824
    BenchmarkStats totalTimerStop(); // cudasw4.cuh:826-839

    std::string_view getReferenceHeader(int64_t id) const { return db_->header(size_t(id)); }
    int32_t getReferenceLength(int64_t id) const { return db_->length(size_t(id)); }
    std::string getReferenceSequence(int64_t id) const { return db_->sequence_letters(size_t(id)); }
    void printDBInfo() const;              // cudasw4.cuh:799-807
    void printDBLengthPartitions() const;  // cudasw4.cuh:809-816
    int numGpus() const { return int(gpus_.size()); }

private:
    struct Gpu;
    void uploadShard(Gpu& g);
    void scanResident(Gpu& g, int32_t qlen);
    void scanStreamed(Gpu& g, int32_t qlen);
    std::vector<std::unique_ptr<Gpu>> gpus_;
    std::shared_ptr<Database> db_;
    int numTop_;
    const SubstitutionMatrix& matrix_;
    KernelTypeConfig kernels_;
    MemoryConfig memory_;
    bool verbose_;
    int gop_, gex_;
    std::vector<int8_t> encodedQuery_;
    // total timer
    double totalSeconds_ = 0;
    double totalCells_ = 0;
    int totalOverflows_ = 0;
    bool totalRunning_ = false;
};

}  // namespace swh
