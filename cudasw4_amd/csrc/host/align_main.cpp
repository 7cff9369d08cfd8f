// align — search query sequences against a DB on all visible MI355X GPUs.
// Command line, console output and result formats follow the reference (main.cu:98-426,
// SURVEY.md Appendix D); the work is done by SearchDriver through libcudasw4_amd.so.
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdlib>
#include <deque>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <string_view>
#include <vector>

#include "cli_options.hpp"
#include "db_format.hpp"
#include "search_driver.hpp"
#include "sequence_reader.hpp"

using namespace swh;

namespace {

void printScanResultPlain(std::ostream& os, const ScanResult& r, const SearchDriver& d) {
    for (size_t i = 0; i < r.scores.size(); i++) {
        const int64_t id = r.referenceIds[i];
        os << "Result " << i << ". Score: " << r.scores[i] << ". Length: " << d.getReferenceLength(id) << ". Header "
           << d.getReferenceHeader(id) << ". referenceId " << id << "\n";
    }
}

void printTSVHeader(std::ostream& os) {
    os << "Query number\tQuery length\tQuery header\tResult number\tResult score\tReference length\tReference header\tReference ID in DB\n";
}

void printScanResultTSV(std::ostream& os, const ScanResult& r, const SearchDriver& d, int64_t queryId, int64_t queryLength,
                        std::string_view queryHeader) {
    for (size_t i = 0; i < r.scores.size(); i++) {
        const int64_t id = r.referenceIds[i];
        os << queryId << '\t' << queryLength << '\t' << queryHeader << '\t' << i << '\t' << r.scores[i] << '\t'
           << d.getReferenceLength(id) << '\t' << d.getReferenceHeader(id) << '\t' << id << "\n";
    }
}

struct Stopwatch {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void print(const char* label) const {
        std::cout << "# elapsed time (" << label << "): " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() << "s\n";
    }
};

void reportScan(const ProgramOptions& o, const ScanResult& r) {
    if (o.verbose) std::cout << "Done. Scan time: " << r.stats.seconds << " s, " << r.stats.gcups << " GCUPS\n";
    else std::cout << "Done.\n";
}

// All queries of a file (main.cu:217-260), one at a time like the reference — unless the DB's shards are small enough for
// the tail hand-over (SearchDriver::prefersTwoInFlight: resident shards of a few rounds of workgroups, or a query that is
// scanned in a few milliseconds; the next query's launch fills the slots the current one's last round leaves idle: +4 %
// on 125 000 subjects, +26 % for a stream of 48-residue queries on a Swiss-Prot-sized DB): then, and with
// CUDASW4_AMD_PIPELINE=1, the next query is submitted before the current one is collected (SearchDriver::submit /
// collect).  On large DBs two queries in flight bring nothing (10^6 subjects) or cost (-0.8 % on a Swiss-Prot-like one);
// CUDASW4_AMD_PIPELINE=0 keeps one query at a time everywhere.  Output order and format are the reference's either way; a
// query's line is printed when its results are in.
void processQueryFile(const std::string& file, const ProgramOptions& o, SearchDriver& driver, std::ostream& out, bool interactive) {
    SequenceReader reader(file);
    struct Pending { int64_t num; std::string header, sequence; };
    std::deque<Pending> pending;
    int64_t query_num = 0;
    const char* pipe = std::getenv("CUDASW4_AMD_PIPELINE");
    // how many queries may be pending once a query of this length has been submitted
    auto max_in_flight = [&](size_t queryLength) {
        if (pipe) return pipe[0] == '1' ? 2 : 1;
        return driver.preferredInFlight(int32_t(std::min<size_t>(queryLength, size_t(INT32_MAX))));
    };
    if (!interactive) driver.totalTimerStart();
    auto finish_oldest = [&]() {
        const Pending q = std::move(pending.front());
        pending.pop_front();
        std::cout << "Processing query " << q.num << " ... ";
        std::cout.flush();
        ScanResult r = driver.collect();
        reportScan(o, r);
        if (o.numTopOutputs > 0 || interactive) {
            if (o.outputMode == ProgramOptions::OutputMode::Plain) {
                (interactive ? std::cout : out) << "Query " << q.num << ", header" << q.header << ", length " << q.sequence.size()
                                                << ", num overflows " << r.stats.numOverflows << "\n";
                printScanResultPlain(out, r, driver);
            } else {
                printScanResultTSV(out, r, driver, interactive ? -1 : q.num, int64_t(q.sequence.size()), interactive ? "-" : q.header);
            }
            out.flush();
        }
    };
    try {
        while (reader.next()) {
            pending.push_back(Pending{query_num++, reader.header(), reader.sequence()});
            driver.submit(pending.back().sequence.data(), int32_t(pending.back().sequence.size()));
            const int maxInFlight = max_in_flight(pending.back().sequence.size());
            while (driver.inFlight() >= maxInFlight) finish_oldest();
        }
        while (driver.inFlight() > 0) finish_oldest();
    } catch (...) {
        while (driver.inFlight() > 0) {  // leave the driver usable (interactive mode goes on after an error)
            try { (void)driver.collect(); } catch (...) {}
        }
        throw;
    }
    if (!interactive) {
        const BenchmarkStats total = driver.totalTimerStop();
        if (o.verbose) std::cout << "Total time: " << total.seconds << " s, " << total.gcups << " GCUPS\n";
    }
}

}  // namespace

int main(int argc, char** argv) {
    ProgramOptions options;
    const bool ok = parseArgs(argc, argv, options);
    if (!ok || options.help) {
        printHelp(argv);
        return 0;
    }
    printOptions(options);
    try {
        std::vector<int> deviceIds;
        int num = 0;
        if (hipGetDeviceCount(&num) != hipSuccess) num = 0;
        for (int i = 0; i < num; i++) deviceIds.push_back(i);
        if (deviceIds.empty()) throw std::runtime_error("No GPU found");
        if (options.verbose) {
            std::cout << "Will use GPU";
            for (int x : deviceIds) std::cout << " " << x;
            std::cout << "\n";
        }
        if (deviceIds.size() == 1) {
            // one GPU: this thread issues everything (with several, every GPU has a worker thread on its own node): run it
            // on the CPUs next to the GPU (CUDASW4_AMD_NO_NUMA_BIND=1: leave the affinity alone)
            const char* no = std::getenv("CUDASW4_AMD_NO_NUMA_BIND");
            const int node = numa_node_of_device(deviceIds[0]);
            if (!(no && no[0] == '1') && bind_thread_to_numa_node(node) && options.verbose)
                std::cout << "GPU " << deviceIds[0] << " is on NUMA node " << node << ": host threads bound to it\n";
        }
        std::ofstream outputfile(options.outputfile);
        if (!outputfile) throw std::runtime_error("Cannot open file " + options.outputfile);
        if (options.outputMode == ProgramOptions::OutputMode::TSV) printTSVHeader(outputfile);

        SearchDriver driver(deviceIds, options.numTopOutputs, options.matrix, options.kernels, options.memory, options.verbose,
                            options.effectiveGop(), options.effectiveGex());
        if (!options.usePseudoDB) {
            if (options.verbose) std::cout << "Reading Database: \n";
            Stopwatch t;
            bool mapped = true;
            auto db = Database::open_or_read(options.dbPrefix, options.prefetchDBFile, &mapped);
            if (options.verbose && !mapped) std::cout << "Failed to map db files. Using fallback db.\n";  // main.cu:180-183
            if (options.verbose) t.print("Read DB");
            driver.setDatabase(db);
        } else {
            if (options.verbose) std::cout << "Generating pseudo db\n";
            Stopwatch t;
            auto db = Database::pseudo(size_t(options.pseudoDBSize), options.pseudoDBLength);
            if (options.verbose) t.print("Generate DB");
            driver.setDatabase(db);
        }
        if (options.verbose) {
            driver.printDBInfo();
            if (options.printLengthPartitions) driver.printDBLengthPartitions();
        }
        if (options.loadFullDBToGpu) driver.prefetchDBToGpus();

        if (!options.interactive) {
            for (const auto& queryFile : options.queryFiles) {
                std::cout << "Processing query file " << queryFile << "\n";
                processQueryFile(queryFile, options, driver, outputfile, false);
            }
        } else {
            // main.cu:336-424
            std::cout << "Interactive mode ready\n";
            std::cout << "Use 's inputsequence' to query inputsequence against the database. Press ENTER twice to begin.\n";
            std::cout << "Use 'f inputfile' to query all sequences in inputfile\n";
            std::cout << "Use 'exit' to terminate\n";
            std::cout << "Waiting for command...\n";
            std::string line;
            while (std::getline(std::cin, line)) {
                std::stringstream ss(line);
                std::string command, argument;
                ss >> command;
                if (command.empty()) continue;
                if (command == "exit") break;
                if (command == "s") {
                    if (ss >> argument) {
                        std::string sequence = argument;
                        while (std::getline(std::cin, line)) {
                            if (line.empty()) break;
                            sequence += line;
                        }
                        std::cout << "sequence: " << sequence << "\n";
                        std::cout << "Processing query " << 0 << " ... ";
                        std::cout.flush();
                        ScanResult r = driver.scan(sequence.data(), int32_t(sequence.size()));
                        reportScan(options, r);
                        if (options.outputMode == ProgramOptions::OutputMode::Plain) printScanResultPlain(outputfile, r, driver);
                        else printScanResultTSV(outputfile, r, driver, -1, int64_t(sequence.size()), "-");
                        outputfile.flush();
                    } else {
                        std::cout << "Missing argument for command 's'\n";
                    }
                } else if (command == "f") {
                    if (ss >> argument) {
                        try {
                            processQueryFile(argument, options, driver, outputfile, true);
                        } catch (...) {
                            std::cout << "Error\n";
                        }
                    } else {
                        std::cout << "Missing argument for command 'f' \n";
                    }
                } else {
                    std::cout << "Unrecognized command: " << command << "\n";
                }
                std::cout << "Waiting for command...\n";
            }
        }
    } catch (const std::exception& e) {
        std::cerr << "align: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
