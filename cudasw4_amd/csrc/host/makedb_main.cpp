// makedb — FASTA/FASTQ(.gz) -> length-sorted, encoded, 4-byte padded DB files.
// Command line and console output follow the reference's makedb (makedb.cpp:279-374):
//     makedb <FASTA/FASTQ filename> pathtodb/dbname [--mem val] [--tempdir val]
#include <chrono>
#include <cstdio>
#include <iostream>
#include <string>

#include "db_format.hpp"
#include "sequence_reader.hpp"

namespace {

struct Stopwatch {
    const char* label;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit Stopwatch(const char* l) : label(l) {}
    void print() const {
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::cout << "# elapsed time (" << label << "): " << s << "s\n";
    }
};

size_t parse_size(const std::string& s) {  // K / M / G suffix (makedb.cpp:293-321)
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

}  // namespace

int main(int argc, char** argv) {
    if (argc < 3) {
        std::cout << "Usage:\n  " << argv[0] << " <FASTA/FASTQ filename> pathtodb/dbname [options]\n";
        std::cout << "Input file may be gzip'ed. pathtodb must exist.\n";
        std::cout << "Options:\n";
        std::cout << "    --mem val : Memory limit. Can use suffix K,M,G. Default all available memory.\n";
        std::cout << "    --tempdir val : Temp directory for temporary files. Must exist. Default is db output directory.\n";
        return 0;
    }
    const std::string input = argv[1];
    const std::string prefix = argv[2];
    size_t mem_limit = 0;
    std::string tempdir = prefix;
    for (int i = 3; i < argc; i++) {
        const std::string arg = argv[i];
        if (arg == "--mem" && i + 1 < argc) {
            mem_limit = parse_size(argv[++i]);
        } else if (arg == "--tempdir" && i + 1 < argc) {
            tempdir = argv[++i];
            if (tempdir.back() != '/') tempdir += '/';
        } else {
            std::cout << "Unexpected arg " << arg << "\n";
        }
    }
    if (mem_limit) std::cout << "availableMem: " << mem_limit << "\n";
    try {
        swh::SequenceBatch batch(tempdir, mem_limit);
        std::cout << "Parsing file\n";
        Stopwatch t1("file parsing");
        {
            swh::SequenceReader reader(input);
            while (reader.next()) batch.add(reader.header(), reader.sequence());
        }
        t1.print();
        std::cout << "Number of input sequences:  " << batch.size() << '\n';
        std::cout << "Number of input characters: " << batch.chars.size() << '\n';
        if (batch.spilled()) std::cout << "Memory limit reached: using temp files " << tempdir << "_cudasw4tmp*\n";
        std::cout << "Converting amino acids\nCreating DB files\n";
        Stopwatch t3("db creation");
        swh::write_database(prefix, batch);
        t3.print();
    } catch (const std::exception& e) {
        std::cerr << "makedb: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
