// cli_options.hpp — command line of `align`, flag-for-flag the reference's (options.hpp:9-51,
// options.cpp:47-267).
#pragma once
#include <cstddef>
#include <cstdint>
#include <limits>
#include <string>
#include <vector>

#include "search_driver.hpp"
#include "sequence_codec.hpp"

namespace swh {

struct ProgramOptions {
    enum class OutputMode { Plain, TSV };
    bool help = false;
    bool loadFullDBToGpu = false;
    bool usePseudoDB = false;
    bool printLengthPartitions = false;
    bool interactive = false;
    bool verbose = false;
    bool prefetchDBFile = false;
    int numTopOutputs = 10;
    int gop = -11;
    int gex = -1;
    // --refCompat (or CUDASW4_AMD_REF_COMPAT=1): score exactly like the reference BINARY.  There --gop / --gex and the
    // per-matrix default gap scores are parsed and printed (options.cpp:179-194) but never reach the kernels
    // (cudasw4.cuh:539-550: the setters have no call site), which always run gop -11 / gex -1 — so the same inputs give the
    // reference's output for every --mat.  The reference's other quirk, the fp16 overflow check that is off for partitions
    // 0-12, is NOT reproduced: it returns inexact scores.
    bool refCompat = false;
    int effectiveGop() const { return refCompat ? -11 : gop; }
    int effectiveGex() const { return refCompat ? -1 : gex; }
    int pseudoDBLength = 0;
    int pseudoDBSize = 0;
    MatrixId matrix = MatrixId::Blosum62;
    KernelTypeConfig kernels;
    OutputMode outputMode = OutputMode::Plain;
    MemoryConfig memory;
    std::string outputfile = "/dev/stdout";
    std::string dbPrefix;
    std::vector<std::string> queryFiles;
};

void printOptions(const ProgramOptions& options);
bool parseArgs(int argc, char** argv, ProgramOptions& options);
void printHelp(char** argv);
size_t parseMemoryString(const std::string& s);

}  // namespace swh
