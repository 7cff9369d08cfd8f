#include "cli_options.hpp"

#include <cstdlib>
#include <iostream>

namespace swh {

size_t parseMemoryString(const std::string& s) {
    if (s.empty()) return 0;
    size_t factor = 1;
    std::string digits = s;
    switch (s.back()) {
        case 'K': factor = size_t(1) << 10; digits.pop_back(); break;
        case 'M': factor = size_t(1) << 20; digits.pop_back(); break;
        case 'G': factor = size_t(1) << 30; digits.pop_back(); break;
        default: break;
    }
    return factor * std::stoull(digits);
}

void printOptions(const ProgramOptions& o) {
    std::cout << "Selected options:\n";
    std::cout << "verbose: " << o.verbose << "\n";
    std::cout << "interactive: " << o.interactive << "\n";
    std::cout << "loadFullDBToGpu: " << o.loadFullDBToGpu << "\n";
    std::cout << "prefetchDBFile: " << o.prefetchDBFile << "\n";
    std::cout << "numTopOutputs: " << o.numTopOutputs << "\n";
    std::cout << "gop: " << o.gop << "\n";
    std::cout << "gex: " << o.gex << "\n";
    if (o.refCompat) std::cout << "refCompat: gap scores applied are " << o.effectiveGop() << " / " << o.effectiveGex() << " like the reference binary's\n";
    std::cout << "maxBatchBytes: " << o.memory.maxBatchBytes << "\n";
    std::cout << "maxBatchSequences: " << o.memory.maxBatchSequences << "\n";
    std::cout << "maxTempBytes: " << o.memory.maxTempBytes << "\n";
    for (size_t i = 0; i < o.queryFiles.size(); i++) std::cout << "queryFile " << i << " : " << o.queryFiles[i] << "\n";
    std::cout << "blosum: " << substitution_matrix(o.matrix).name << "\n";
    std::cout << "singlePassType: " << to_string(o.kernels.singlePassType) << "\n";
    std::cout << "manyPassType_small: " << to_string(o.kernels.manyPassType_small) << "\n";
    std::cout << "manyPassType_large: " << to_string(o.kernels.manyPassType_large) << "\n";
    std::cout << "overflowType: " << to_string(o.kernels.overflowType) << "\n";
    if (o.usePseudoDB) {
        std::cout << "Using built-in pseudo db with " << o.pseudoDBSize << " sequences of length " << o.pseudoDBLength << "\n";
    } else {
        std::cout << "Using db file: " << o.dbPrefix << "\n";
    }
    std::cout << "memory limit per gpu: "
              << (o.memory.maxGpuMem == std::numeric_limits<size_t>::max() ? std::string("unlimited") : std::to_string(o.memory.maxGpuMem))
              << "\n";
    std::cout << "Output mode: " << (o.outputMode == ProgramOptions::OutputMode::TSV ? "TSV" : "Plain") << "\n";
    std::cout << "Output file: " << o.outputfile << "\n";
}

bool parseArgs(int argc, char** argv, ProgramOptions& o) {
    bool gotQuery = false, gotDB = false, gotGex = false, gotGop = false, gotDPX = false;
    o.queryFiles.clear();
    auto value = [&](int& i) -> std::string {
        if (i + 1 >= argc) { std::cout << "Missing value for " << argv[i] << "\n"; return std::string(); }
        return argv[++i];
    };
    auto kernelType = [&](int& i, KernelType& dst) {
        const std::string v = value(i);
        if (!parse_kernel_type(v, dst)) std::cout << "Unknown kernel type " << v << " (valid: Half2, DPXs16, DPXs32, Float)\n";
    };
    for (int i = 1; i < argc; i++) {
        const std::string arg = argv[i];
        if (arg == "--help") o.help = true;
        else if (arg == "--uploadFull") o.loadFullDBToGpu = true;
        else if (arg == "--verbose") o.verbose = true;
        else if (arg == "--interactive") o.interactive = true;
        else if (arg == "--printLengthPartitions") o.printLengthPartitions = true;
        else if (arg == "--prefetchDBFile") o.prefetchDBFile = true;
        else if (arg == "--top") o.numTopOutputs = std::atoi(value(i).c_str());
        else if (arg == "--gop") { o.gop = std::atoi(value(i).c_str()); gotGop = true; }
        else if (arg == "--gex") { o.gex = std::atoi(value(i).c_str()); gotGex = true; }
        else if (arg == "--maxBatchBytes") o.memory.maxBatchBytes = parseMemoryString(value(i));
        else if (arg == "--maxBatchSequences") o.memory.maxBatchSequences = size_t(std::atoll(value(i).c_str()));
        else if (arg == "--maxTempBytes") o.memory.maxTempBytes = parseMemoryString(value(i));
        else if (arg == "--maxGpuMem") o.memory.maxGpuMem = parseMemoryString(value(i));
        else if (arg == "--query") { o.queryFiles.push_back(value(i)); gotQuery = true; }
        else if (arg == "--db") { o.dbPrefix = value(i); gotDB = true; }
        else if (arg == "--mat") {
            const std::string v = value(i);
            if (!parse_matrix_name(v, o.matrix)) std::cout << "Unknown matrix " << v << "\n";
        }
        else if (arg == "--singlePassType") kernelType(i, o.kernels.singlePassType);
        else if (arg == "--manyPassType_small") kernelType(i, o.kernels.manyPassType_small);
        else if (arg == "--manyPassType_large") kernelType(i, o.kernels.manyPassType_large);
        else if (arg == "--overflowType") kernelType(i, o.kernels.overflowType);
        else if (arg == "--pseudodb") {
            o.usePseudoDB = true;
            o.pseudoDBSize = std::atoi(value(i).c_str());
            o.pseudoDBLength = std::atoi(value(i).c_str());
            gotDB = true;
        }
        else if (arg == "--dpx") gotDPX = true;
        else if (arg == "--refCompat") o.refCompat = true;
        else if (arg == "--tsv") o.outputMode = ProgramOptions::OutputMode::TSV;
        else if (arg == "--of") o.outputfile = value(i);
        else std::cout << "Unexpected arg " << arg << "\n";
    }
    // matrix-specific default gap scores unless given (options.cpp:179-194).  Unlike the reference,
    // where these never reach the kernels, they are forwarded to the scan.
    const SubstitutionMatrix& m = substitution_matrix(o.matrix);
    if (!gotGop) o.gop = m.default_gop;
    if (!gotGex) o.gex = m.default_gex;
    if (const char* e = std::getenv("CUDASW4_AMD_REF_COMPAT")) o.refCompat = o.refCompat || e[0] == '1';
    if (gotDPX) {
        o.kernels.singlePassType = KernelType::DPXs16;
        o.kernels.manyPassType_small = KernelType::DPXs16;
        o.kernels.manyPassType_large = KernelType::DPXs32;
        o.kernels.overflowType = KernelType::DPXs32;
    }
    if (!gotQuery && !o.interactive && !o.help) { std::cout << "Query is missing\n"; return false; }
    if (!gotDB && !o.help) { std::cout << "DB prefix is missing\n"; return false; }
    return true;
}

void printHelp(char** argv) {
    ProgramOptions d;
    std::cout << "Usage: " << argv[0] << " [options]\n";
    std::cout << "The GPUs to use are set via HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES environment variables.\n";
    std::cout << "Options: \n";
    std::cout << "   Mandatory\n";
    std::cout << "      --query queryfile : Mandatory. Fasta or Fastq. Can be gzip'ed. Repeat this option for multiple query files\n";
    std::cout << "      --db dbPrefix : Mandatory. The DB to query against. The same dbPrefix as used for makedb\n\n";
    std::cout << "   Scoring\n";
    std::cout << "      --top val : Output the val best scores. Default val = " << d.numTopOutputs << "\n";
    std::cout << "      --gop val : Gap open score. Overwrites our blosum-dependent default score.\n";
    std::cout << "      --gex val : Gap extend score. Overwrites our blosum-dependent default score.\n";
    std::cout << "      --mat val: Set substitution matrix. Supported values: blosum45, blosum50, blosum62, blosum80. Default: blosum62\n"
                 "        (with suffix _25: the full 25-letter table for the query letters B, J, Z, X, *)\n\n";
    std::cout << "   Memory\n";
    std::cout << "      --maxGpuMem val : Try not to use more than val bytes of gpu memory per gpu. Uses all available gpu memory by default\n";
    std::cout << "      --maxTempBytes val : Size of temp storage in GPU memory. Can use suffix K,M,G. Default val = " << d.memory.maxTempBytes << "\n";
    std::cout << "      --maxBatchBytes val : Process DB in batches of at most val bytes. Can use suffix K,M,G. Default val = " << d.memory.maxBatchBytes << "\n";
    std::cout << "      --maxBatchSequences val : Process DB in batches of at most val sequences. Default val = " << d.memory.maxBatchSequences << "\n\n";
    std::cout << "   Misc\n";
    std::cout << "      --dpx : Use the packed int16 / int32 kernels (the reference's DPX kernels).\n";
    std::cout << "      --refCompat : Apply gap scores -11 / -1 whatever --gop, --gex and --mat say, like the reference binary does.\n";
    std::cout << "      --of : Result output file. Parent directory must exist. Default: console output (/dev/stdout)\n";
    std::cout << "      --tsv : Print results as tab-separated values instead of plain text. \n";
    std::cout << "      --verbose : More console output. Shows timings. \n";
    std::cout << "      --printLengthPartitions : Print number of sequences per length partition in db.\n";
    std::cout << "      --interactive : Loads DB, then waits for sequence input by user\n";
    std::cout << "      --help : Print this message\n\n";
    std::cout << "   Performance and benchmarking\n";
    std::cout << "      --prefetchDBFile : Load DB into RAM immediately at program start instead of waiting for the first access.\n";
    std::cout << "      --uploadFull : If enough GPU memory is available to store full db, copy full DB to GPU before processing queries.\n";
    std::cout << "      --pseudodb num length : Use a generated DB which contains `num` equal sequences of length `length`.\n";
    std::cout << "      --singlePassType val, --manyPassType_small val, --manyPassType_large val, --overflowType val :\n";
    std::cout << "           Select kernel types for different length partitions. Valid values: Half2, DPXs16, DPXs32, Float.\n";
    std::cout << "           Misc option --dpx is equivalent to --singlePassType DPXs16 --manyPassType_small DPXs16 --manyPassType_large DPXs32 --overflowType DPXs32.\n";
    std::cout << "           Default is --singlePassType Half2 --manyPassType_small Half2 --manyPassType_large Float --overflowType Float.\n\n";
}

}  // namespace swh
