#include "db_format.hpp"
#include "parallel_blocks.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <fstream>
#include <limits>
#include <numeric>
#include <parallel/algorithm>
#include <cstdlib>
#include <random>
#include <type_traits>

#include "sequence_codec.hpp"

namespace swh {

const std::array<int32_t, kNumLengthPartitions>& length_partition_bounds() {
    static const std::array<int32_t, kNumLengthPartitions> bounds = [] {
        std::array<int32_t, kNumLengthPartitions> b{};
        int k = 0;
        b[k++] = 48;
        b[k++] = 64;
        for (int v = 80; v <= 256; v += 16) b[k++] = v;
        for (int v = 288; v <= 512; v += 32) b[k++] = v;
        for (int v = 576; v <= 1280; v += 64) b[k++] = v;
        b[k++] = 8000;
        b[k++] = std::numeric_limits<int32_t>::max() - 1;
        return b;
    }();
    return bounds;
}

int length_partition_of(int32_t length) {
    const auto& b = length_partition_bounds();
    return int(std::lower_bound(b.begin(), b.end(), length) - b.begin());
}

// ------------------------------------------------------------------ makedb side

void SequenceBatch::add(std::string_view header, std::string_view sequence) {
    chars.append(sequence.data(), sequence.size());
    const size_t pad = (4 - sequence.size() % 4) % 4;
    chars.append_fill(' ', pad);
    offsets.push_back(chars.size());
    lengths.push_back(int32_t(sequence.size()));
    headers.append(header.data(), header.size());
    header_offsets.push_back(headers.size());
}

namespace {

std::ofstream open_out(const std::string& path) {
    std::ofstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("Cannot open output file " + path);
    return f;
}

template <class T>
void put(std::ofstream& f, const T& v) {
    f.write(reinterpret_cast<const char*>(&v), sizeof(T));
}

}  // namespace

void write_database(const std::string& prefix, SequenceBatch& batch) {
    {   // thrust::transform(omp::par, ConvertAA_20) (makedb.cpp:361): encode in parallel chunks
        const size_t total = batch.chars.size();
        const size_t chunk = size_t(1) << 24;
        const long nchunks = long((total + chunk - 1) / chunk);
#pragma omp parallel for schedule(dynamic)
        for (long c = 0; c < nchunks; c++) {
            const size_t b = size_t(c) * chunk;
            encode_in_place(batch.chars.data() + b, std::min(chunk, total - b));
        }
    }

    const size_t n = batch.size();
    // same call as the reference (std::sort over an iota permutation, compared by length only), so
    // that equal-length ties land in the same libstdc++-specific order and the files match byte for byte
    std::vector<int32_t> order(n);
    std::iota(order.begin(), order.end(), 0);
    // Large inputs (UniRef / TrEMBL scale: 10^7..10^8 sequences) are sorted in parallel (libstdc++ parallel mode,
    // OpenMP): a stable multiway merge sort, i.e. equal lengths keep their input order.  The reference's single-threaded
    // std::sort leaves ties in an unspecified order, so any tie order is a valid DB; below the threshold the very same
    // call is kept for byte-equal files.  CUDASW4_AMD_PARALLEL_SORT_MIN overrides the threshold (tests).
    size_t parallel_min = size_t(1) << 20;
    if (const char* e = std::getenv("CUDASW4_AMD_PARALLEL_SORT_MIN")) parallel_min = size_t(std::strtoull(e, nullptr, 10));
    auto by_length = [&](const auto& l, const auto& r) { return batch.lengths[l] < batch.lengths[r]; };
    if (n >= parallel_min) __gnu_parallel::stable_sort(order.begin(), order.end(), by_length);
    else std::sort(order.begin(), order.end(), by_length);

    const auto& bounds = length_partition_bounds();
    std::array<uint64_t, kNumLengthPartitions> counts{};
    for (size_t i = 0; i < n; i++) counts[length_partition_of(batch.lengths[order[i]])]++;

    const std::string chunk = prefix + "0";
    {
        auto meta = open_out(chunk + "metadata");
        put(meta, int32_t(kNumLengthPartitions));
        for (int32_t b : bounds) put(meta, b);
        for (uint64_t c : counts) put(meta, c);
    }
    auto headers = open_out(chunk + "headers");
    auto header_offsets = open_out(chunk + "headeroffsets");
    auto chars = open_out(chunk + "chars");
    auto offsets = open_out(chunk + "offsets");
    auto lengths = open_out(chunk + "lengths");
    uint64_t hoff = 0, coff = 0;
    put(header_offsets, hoff);
    put(offsets, coff);
    for (size_t k = 0; k < n; k++) {
        const size_t i = size_t(order[k]);
        const uint64_t hlen = batch.header_offsets[i + 1] - batch.header_offsets[i];
        headers.write(batch.headers.data() + batch.header_offsets[i], std::streamsize(hlen));
        hoff += hlen;
        put(header_offsets, hoff);
        const uint64_t clen = batch.offsets[i + 1] - batch.offsets[i];
        chars.write(batch.chars.data() + batch.offsets[i], std::streamsize(clen));
        put(lengths, batch.lengths[i]);
        coff += clen;
        put(offsets, coff);
    }
    open_out(prefix + "metadata");  // writeGlobalDbInfo (dbdata.cpp:192-197): empty marker
}

// ------------------------------------------------------------------ reading

namespace {

struct Mapping {
    void* ptr = nullptr;
    size_t bytes = 0;
    ~Mapping() {
        if (ptr && bytes) munmap(ptr, bytes);
    }
    void map(const std::string& path, bool populate) {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) throw DbLoadError("Cannot open " + path);
        struct stat st;
        if (fstat(fd, &st) != 0) { ::close(fd); throw DbLoadError("Cannot stat " + path); }
        bytes = size_t(st.st_size);
        if (bytes) {
            int flags = MAP_PRIVATE;
            if (populate) flags |= MAP_POPULATE;
            ptr = mmap(nullptr, bytes, PROT_READ, flags, fd, 0);
            if (ptr == MAP_FAILED) { ptr = nullptr; ::close(fd); throw DbMapError("Cannot mmap " + path); }
        }
        ::close(fd);
    }
};

}  // namespace

struct Database::Storage {
    Mapping chars, offsets, lengths, headers, header_offsets;
    std::vector<int8_t> vchars;
    std::vector<uint64_t> voffsets, vheader_offsets;
    std::vector<int32_t> vlengths;
    std::vector<char> vheaders;
};

Database::~Database() = default;

void Database::finish() {
    // The scan kernels load 4-byte words at chars + offset + j for j < round4(length) and turn every letter byte into
    // an LDS row offset, so a truncated, foreign or corrupt DB must be refused here, not discovered as out-of-bounds
    // device reads or garbage scores: offsets non-decreasing with room for every padded sequence, lengths >= 0 and
    // ascending, letter codes 0..20.
    for (size_t i = 0; i < n_; i++) {
        if (lengths_[i] < 0) throw DbLoadError("DB has a negative sequence length");
        if (i > 0 && lengths_[i] < lengths_[i - 1]) throw DbLoadError("DB is not sorted by sequence length");
        if (offsets_[i + 1] < offsets_[i]) throw DbLoadError("DB offsets are not monotonic");
        const uint64_t padded = (uint64_t(lengths_[i]) + 3) / 4 * 4;
        if (offsets_[i + 1] - offsets_[i] < padded) throw DbLoadError("DB offsets leave no room for a padded sequence");
    }
    const auto& b = length_partition_bounds();
    begins_[0] = 0;
    const int32_t* first = lengths_;
    const int32_t* last = lengths_ + n_;
    for (int p = 0; p < kNumLengthPartitions; p++) {
        const int32_t* e = std::upper_bound(first, last, b[p]);
        begins_[p + 1] = size_t(e - lengths_);
        counts_[p] = begins_[p + 1] - begins_[p];
        first = e;
    }
    residues_ = 0;
    for (size_t i = 0; i < n_; i++) residues_ += uint64_t(lengths_[i]);
}

// every letter code of the DB is 0..20 (a parallel pass over the chars; for a memory-mapped file it reads the whole file,
// which loadDB's prefetch does anyway)
void Database::validate_codes() {
    const size_t total = num_chars();
    const int8_t* c = chars_;
    std::atomic<bool> bad{false};
    // threads that exit with the loop, not OpenMP workers that spin afterwards (parallel_blocks.hpp)
    parallel_blocks((total + (size_t(1) << 22) - 1) >> 22, 16, [&](size_t blk) {
        const size_t b = blk << 22, e = std::min(total, b + (size_t(1) << 22));
        unsigned char m = 0;
        for (size_t i = b; i < e; i++) m |= (unsigned char)(c[i]) > 20 ? 1 : 0;
        if (m) bad.store(true, std::memory_order_relaxed);
    });
    if (bad.load()) throw DbLoadError("DB chars hold letter codes outside 0..20 (not a cudasw4 DB, or corrupt)");
    codes_validated_ = true;
}

std::shared_ptr<Database> Database::open(const std::string& prefix, bool prefetch) {
    {
        std::ifstream marker(prefix + "metadata", std::ios::binary);  // dbdata.cpp:199-204
        if (!marker) throw DbLoadError("Cannot open DB " + prefix + " (missing " + prefix + "metadata)");
    }
    if (const char* e = std::getenv("CUDASW4_AMD_DB_NO_MMAP")) {
        if (e[0] == '1') throw DbMapError("memory mapping disabled by CUDASW4_AMD_DB_NO_MMAP");
    }
    std::shared_ptr<Database> db(new Database);
    db->storage_ = std::make_unique<Storage>();
    Storage& s = *db->storage_;
    const std::string chunk = prefix + "0";
    s.chars.map(chunk + "chars", prefetch);
    s.offsets.map(chunk + "offsets", prefetch);
    s.lengths.map(chunk + "lengths", prefetch);
    s.headers.map(chunk + "headers", false);
    s.header_offsets.map(chunk + "headeroffsets", false);
    if (s.lengths.bytes % sizeof(int32_t) || s.offsets.bytes % sizeof(uint64_t)) throw DbLoadError("Corrupt DB files");
    db->n_ = s.lengths.bytes / sizeof(int32_t);
    if (s.offsets.bytes != (db->n_ + 1) * sizeof(uint64_t) || s.header_offsets.bytes != (db->n_ + 1) * sizeof(uint64_t))
        throw DbLoadError("DB offset files do not match the number of sequences");
    db->chars_ = static_cast<const int8_t*>(s.chars.ptr);
    db->offsets_ = static_cast<const uint64_t*>(s.offsets.ptr);
    db->lengths_ = static_cast<const int32_t*>(s.lengths.ptr);
    db->headers_ = static_cast<const char*>(s.headers.ptr);
    db->header_offsets_ = static_cast<const uint64_t*>(s.header_offsets.ptr);
    if (db->n_ && db->offsets_[db->n_] > s.chars.bytes) throw DbLoadError("DB chars file is too short");
    db->finish();
    // letter codes: checked here when the file is read in full anyway (prefetch) or small; otherwise the search driver
    // checks them on the device as it uploads them (codes_validated()).  CUDASW4_AMD_VALIDATE_DB=1|0 forces / skips the
    // pass here
    const char* v = std::getenv("CUDASW4_AMD_VALIDATE_DB");
    const bool check = v ? v[0] == '1' : (prefetch || db->num_chars() <= (size_t(4) << 30));
    if (check) db->validate_codes();
    return db;
}

// loadDBWithVectors (dbdata.cpp:118-190): the same files read into memory with plain file I/O — the fallback `align`
// takes when the files cannot be memory-mapped (main.cu:180-191), and what CUDASW4_AMD_DB_NO_MMAP=1 forces (tests)
std::shared_ptr<Database> Database::open_with_vectors(const std::string& prefix) {
    {
        std::ifstream marker(prefix + "metadata", std::ios::binary);
        if (!marker) throw DbLoadError("Cannot open DB " + prefix + " (missing " + prefix + "metadata)");
    }
    auto slurp = [](const std::string& path, auto& vec) {
        using T = typename std::remove_reference_t<decltype(vec)>::value_type;
        std::ifstream in(path, std::ios::binary | std::ios::ate);
        if (!in) throw DbLoadError("Cannot open " + path);
        const std::streamsize bytes = in.tellg();
        if (bytes % std::streamsize(sizeof(T))) throw DbLoadError("Corrupt DB file " + path);
        vec.resize(size_t(bytes) / sizeof(T));
        in.seekg(0);
        if (bytes && !in.read(reinterpret_cast<char*>(vec.data()), bytes)) throw DbLoadError("Cannot read " + path);
    };
    const std::string chunk = prefix + "0";
    std::vector<int8_t> chars;
    std::vector<uint64_t> offsets, header_offsets;
    std::vector<int32_t> lengths;
    std::vector<char> headers;
    slurp(chunk + "chars", chars);
    slurp(chunk + "offsets", offsets);
    slurp(chunk + "lengths", lengths);
    slurp(chunk + "headers", headers);
    slurp(chunk + "headeroffsets", header_offsets);
    if (offsets.size() != lengths.size() + 1 || header_offsets.size() != lengths.size() + 1)
        throw DbLoadError("DB offset files do not match the number of sequences");
    return from_vectors(std::move(chars), std::move(offsets), std::move(lengths), std::move(headers), std::move(header_offsets));
}

std::shared_ptr<Database> Database::open_or_read(const std::string& prefix, bool prefetch, bool* mapped) {
    try {
        auto db = open(prefix, prefetch);
        if (mapped) *mapped = true;
        return db;
    } catch (const DbMapError&) {
        if (mapped) *mapped = false;
        return open_with_vectors(prefix);
    }
}

std::shared_ptr<Database> Database::from_vectors(std::vector<int8_t> chars, std::vector<uint64_t> offsets,
                                                 std::vector<int32_t> lengths, std::vector<char> headers,
                                                 std::vector<uint64_t> header_offsets) {
    std::shared_ptr<Database> db(new Database);
    db->storage_ = std::make_unique<Storage>();
    Storage& s = *db->storage_;
    s.vchars = std::move(chars);
    s.voffsets = std::move(offsets);
    s.vlengths = std::move(lengths);
    s.vheaders = std::move(headers);
    s.vheader_offsets = std::move(header_offsets);
    db->n_ = s.vlengths.size();
    if (s.voffsets.size() != db->n_ + 1 || s.vheader_offsets.size() != db->n_ + 1) throw DbLoadError("inconsistent DB vectors");
    db->chars_ = s.vchars.data();
    db->offsets_ = s.voffsets.data();
    db->lengths_ = s.vlengths.data();
    db->headers_ = s.vheaders.data();
    db->header_offsets_ = s.vheader_offsets.data();
    if (db->n_ && (s.voffsets[db->n_] - s.voffsets[0]) > s.vchars.size()) throw DbLoadError("DB chars array is too short");
    db->finish();
    db->validate_codes();
    return db;
}

std::shared_ptr<Database> Database::pseudo(size_t num, int32_t length, int seed) {
    const size_t stride = (size_t(length) + 3) / 4 * 4;
    std::mt19937 gen(seed);
    std::uniform_int_distribution<> dist(0, 19);
    std::vector<int8_t> one(stride, kOtherCode);
    for (int32_t i = 0; i < length; i++) one[i] = int8_t(dist(gen));  // letters[k] encodes to k
    std::vector<int8_t> chars(num * stride);
    for (size_t i = 0; i < num; i++) std::memcpy(chars.data() + i * stride, one.data(), stride);
    std::vector<uint64_t> offsets(num + 1), header_offsets(num + 1);
    for (size_t i = 0; i <= num; i++) { offsets[i] = i * stride; header_offsets[i] = i; }
    std::vector<int32_t> lengths(num, length);
    std::vector<char> headers(num, 'H');
    return from_vectors(std::move(chars), std::move(offsets), std::move(lengths), std::move(headers), std::move(header_offsets));
}

std::string Database::sequence_letters(size_t i) const {
    std::string s(size_t(lengths_[i]), ' ');
    const int8_t* p = chars_ + (offsets_[i] - offsets_[0]);
    for (size_t k = 0; k < s.size(); k++) s[k] = decode_residue(p[k]);
    return s;
}

// ------------------------------------------------------------------ sharding

std::vector<std::array<ShardRange, kNumLengthPartitions>> shard_ranges(const uint64_t* off, const size_t* partBegin, int num_shards) {
    std::vector<std::array<ShardRange, kNumLengthPartitions>> result(size_t(std::max(num_shards, 1)));
    for (int p = 0; p < kNumLengthPartitions; p++) {
        const size_t pb = partBegin[p], pe = partBegin[p + 1];
        for (auto& r : result) r[p] = ShardRange{pb, pb};
        if (pe == pb) continue;
        const uint64_t chars_total = off[pe] - off[pb];
        const uint64_t quota = chars_total / uint64_t(result.size());
        // a range ends at the first subject boundary past its quota (dbdata.cpp:265-292)
        size_t cur = pb;
        for (size_t s = 0; s < result.size() && cur < pe; s++) {
            size_t end;
            if (s + 1 == result.size()) {
                end = pe;
            } else {
                const uint64_t target = off[cur] + quota;
                end = size_t(std::upper_bound(off + cur, off + pe + 1, target) - off);
                end = std::min(std::max(end, cur + 1), pe);
            }
            result[s][p] = ShardRange{cur, end};
            cur = end;
        }
        // leftovers (rounding) go to the last shard that received something
        if (cur < pe) {
            for (size_t s = result.size(); s-- > 0;)
                if (result[s][p].size() > 0) { result[s][p].end = pe; break; }
        }
    }
    return result;
}

std::vector<std::array<ShardRange, kNumLengthPartitions>> shard_database(const Database& db, int num_shards) {
    size_t partBegin[kNumLengthPartitions + 1];
    for (int p = 0; p < kNumLengthPartitions; p++) partBegin[p] = db.partition_begin(p);
    partBegin[kNumLengthPartitions] = db.num_sequences();
    return shard_ranges(db.offsets(), partBegin, num_shards);
}

}  // namespace swh
