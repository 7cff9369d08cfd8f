// tests/boundary/binding_harness.cpp — TEST INFRASTRUCTURE: compiles the reference-side binding that INTEGRATION.md §2
// documents, against the reference's own headers (types.hpp, length_partitions.hpp, config.hpp — included from
// /root/reference/src where they lie, never copied) and include/cudasw4_amd.h, and links it against libcudasw4_amd.so.
// tests/test_boundary_cpu.py replaces the marker line below with the code block of INTEGRATION.md §2 VERBATIM, so the
// documented patch cannot rot.  The surrounding declarations stand for the members of the reference's CudaSW4 /
// GpuWorkingSet (cudasw4.cuh:251-480) that the patch touches; their types are the reference's.
#include <cstdio>
#include <cstdlib>
#include <vector>

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
    const int8_t encodedQueryStorage[4] = {0, 1, 2, 3};
    const int8_t* encodedQuery = encodedQueryStorage;
    const SequenceLengthT queryLength = 4;
    const int gop = -11, gex = -1, gpu = 0;
    void* stream = nullptr;
    int32_t exclPs = 0;
    const int64_t globalOffsetOfBatch = 0;
    const int32_t maxOverflows = 0, maxLen = 0;
    const int64_t numResults = 0; const int results_per_query = 0;
    float* d_topS = nullptr; int32_t* d_topI = nullptr; void* d_tmp = nullptr; size_t tmpBytes = 0;
    static_assert(sizeof(size_t) == sizeof(uint64_t), "offsets are 64-bit");

//@@INTEGRATION_MD_PATCH@@

    for (auto* c : swCtx) sw_ctx_destroy(c);
    std::puts("binding ok");
    return 0;
}
