// db_format.hpp — the dbdata on-disk layout (kept byte-compatible with the reference) and the
// in-memory database views the search driver works on.
//
// Files of a DB with prefix P (names dbdata.hpp:21-28, writer makedb.cpp:228-275,
// reader dbdata.cpp:46-116):
//   Pmetadata        empty marker file
//   P0chars          int8 codes, every sequence padded with code 20 to a multiple of 4, ascending length
//   P0offsets        uint64[N+1] byte offsets into chars
//   P0lengths        int32[N] true lengths
//   P0headers        concatenated header bytes,  P0headeroffsets  uint64[N+1]
//   P0metadata       int32 n=36, int32 bounds[36], uint64 counts[36]
#pragma once
#include <array>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <string_view>
#include <vector>

#include "spill_vector.hpp"

namespace swh {

constexpr int kNumLengthPartitions = 36;

// length_partitions.hpp:75-113; length L is in partition i iff bound[i-1] < L <= bound[i]
const std::array<int32_t, kNumLengthPartitions>& length_partition_bounds();
int length_partition_of(int32_t length);

class DbLoadError : public std::runtime_error {
public:
    using std::runtime_error::runtime_error;
};
// the files exist but cannot be memory-mapped: callers fall back to Database::open_with_vectors (main.cu:180-191)
class DbMapError : public DbLoadError {
public:
    using DbLoadError::DbLoadError;
};

// Sequences collected by makedb before sorting (HybridBatch, makedb.cpp:80-133): raw letters, padded to 4
// with ' '.  With a memory limit (--mem) the five arrays share the budget 50/7/7/29/7 % like the reference
// (makedb.cpp:84-88) and spill to `<tempprefix>_cudasw4tmp*` files (makedb.cpp:90-94) when they outgrow it.
struct SequenceBatch {
    explicit SequenceBatch(const std::string& tempprefix = std::string(), size_t mem_limit = 0)
        : chars(tempprefix + "_cudasw4tmpchars", mem_limit / 100 * 50),
          offsets(tempprefix + "_cudasw4tmpoffsets", mem_limit / 100 * 7),
          lengths(tempprefix + "_cudasw4tmplengths", mem_limit / 100 * 7),
          headers(tempprefix + "_cudasw4tmpheaders", mem_limit / 100 * 29),
          header_offsets(tempprefix + "_cudasw4tmpheaderOffsets", mem_limit / 100 * 7) {
        offsets.push_back(0);
        header_offsets.push_back(0);
    }
    SpillVector<char> chars;
    SpillVector<uint64_t> offsets;
    SpillVector<int32_t> lengths;
    SpillVector<char> headers;
    SpillVector<uint64_t> header_offsets;

    void add(std::string_view header, std::string_view sequence);
    size_t size() const { return lengths.size(); }
    bool spilled() const { return chars.spilled() || headers.spilled() || offsets.spilled(); }
};

// Encodes the batch (ConvertAA_20), sorts by length and writes the DB files (makedb.cpp:182-276,361).
void write_database(const std::string& prefix, SequenceBatch& batch);

// A read-only database: memory-mapped files, or generated in memory (pseudo DB).
class Database {
public:
    // loadDB (dbdata.cpp:46-116,207-222); prefetch == MAP_POPULATE (mapped_file.hpp:76-78)
    static std::shared_ptr<Database> open(const std::string& prefix, bool prefetch);
    // loadDBWithVectors (dbdata.cpp:118-190): the files read into memory, for when they cannot be mapped
    static std::shared_ptr<Database> open_with_vectors(const std::string& prefix);
    // open(), falling back to open_with_vectors() on a DbMapError like the reference's main (main.cu:172-191)
    static std::shared_ptr<Database> open_or_read(const std::string& prefix, bool prefetch, bool* mapped = nullptr);
    // PseudoDBdata (dbdata.hpp:222-272): one std::mt19937(seed) sequence of `length` replicated `num` times
    static std::shared_ptr<Database> pseudo(size_t num, int32_t length, int seed = 42);
    // from arrays already in dbdata layout (used by tests / embedding)
    static std::shared_ptr<Database> from_vectors(std::vector<int8_t> chars, std::vector<uint64_t> offsets,
                                                  std::vector<int32_t> lengths, std::vector<char> headers,
                                                  std::vector<uint64_t> header_offsets);
    ~Database();

    size_t num_sequences() const { return n_; }
    size_t num_chars() const { return n_ ? size_t(offsets_[n_] - offsets_[0]) : 0; }
    const int8_t* chars() const { return chars_; }
    const uint64_t* offsets() const { return offsets_; }
    const int32_t* lengths() const { return lengths_; }
    int32_t length(size_t i) const { return lengths_[i]; }
    std::string_view header(size_t i) const {
        return std::string_view(headers_ + header_offsets_[i], size_t(header_offsets_[i + 1] - header_offsets_[i]));
    }
    std::string sequence_letters(size_t i) const;  // decoded residues (debug output)

    // number of sequences per length partition, recomputed from the sorted lengths (dbdata.cpp:91-115)
    const std::array<size_t, kNumLengthPartitions>& partition_counts() const { return counts_; }
    size_t partition_begin(int p) const { return begins_[p]; }
    uint64_t total_residues() const { return residues_; }
    // every letter code was checked (0..20) when the DB was loaded.  Large memory-mapped DBs that are not prefetched skip
    // that pass over the whole file; the search driver then checks the chars on the device as it uploads them.
    bool codes_validated() const { return codes_validated_; }

private:
    Database() = default;
    void finish();  // validates offsets / lengths / ordering, computes partition tables
    void validate_codes();  // every letter code 0..20
    bool codes_validated_ = false;
    struct Storage;
    std::unique_ptr<Storage> storage_;
    const int8_t* chars_ = nullptr;
    const uint64_t* offsets_ = nullptr;
    const int32_t* lengths_ = nullptr;
    const char* headers_ = nullptr;
    const uint64_t* header_offsets_ = nullptr;
    size_t n_ = 0;
    uint64_t residues_ = 0;
    std::array<size_t, kNumLengthPartitions> counts_{};
    std::array<size_t, kNumLengthPartitions + 1> begins_{};
};

// A contiguous range of subjects [begin, end) of one length partition assigned to one shard.
struct ShardRange {
    size_t begin = 0, end = 0;
    size_t size() const { return end - begin; }
};

// partitionDBAmongstGpus (cudasw4.cuh:928-1004): every length partition is cut into <= num_shards
// contiguous, char-balanced ranges (dbdata.cpp:265-292) so that each GPU sees every length class.
// result[shard][partition]
std::vector<std::array<ShardRange, kNumLengthPartitions>> shard_database(const Database& db, int num_shards);
// the same on raw arrays: offsets[n+1] and the first subject of every length partition (partBegin[36] = n)
std::vector<std::array<ShardRange, kNumLengthPartitions>> shard_ranges(const uint64_t* offsets, const size_t* partBegin, int num_shards);

}  // namespace swh
