// dbinspect — prints what the host driver derives from a DB on disk (no GPU needed): sequence count, residues,
// per-length-partition counts and the per-GPU shard ranges of shard_database().  Used by the CPU tests to check
// the C++ reader / sharding against the Python mirror and the reference's metadata file.
//     dbinspect <dbprefix> [num_shards]
#include <cstdlib>
#include <iostream>

#include "db_format.hpp"

int main(int argc, char** argv) {
    if (argc < 2) {
        std::cout << "Usage: " << argv[0] << " dbprefix [num_shards]\n";
        return 0;
    }
    try {
        auto db = swh::Database::open_or_read(argv[1], false);
        const int shards = argc > 2 ? std::atoi(argv[2]) : 1;
        std::cout << "{\"num_sequences\": " << db->num_sequences() << ", \"num_chars\": " << db->num_chars()
                  << ", \"residues\": " << db->total_residues() << ", \"partition_counts\": [";
        for (int p = 0; p < swh::kNumLengthPartitions; p++) std::cout << (p ? "," : "") << db->partition_counts()[p];
        std::cout << "], \"first_header\": \"" << (db->num_sequences() ? std::string(db->header(0)) : std::string()) << "\", \"shards\": [";
        const auto ranges = swh::shard_database(*db, shards);
        for (size_t s = 0; s < ranges.size(); s++) {
            std::cout << (s ? "," : "") << "[";
            for (int p = 0; p < swh::kNumLengthPartitions; p++)
                std::cout << (p ? "," : "") << "[" << ranges[s][p].begin << "," << ranges[s][p].end << "]";
            std::cout << "]";
        }
        std::cout << "]}\n";
    } catch (const std::exception& e) {
        std::cerr << "dbinspect: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
