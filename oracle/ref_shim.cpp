// oracle/ref_shim.cpp — TEST INFRASTRUCTURE, not product code.
//
// Thin extern "C" driver around headers of the reference implementation, compiled from the
// sources where they lie under /root/reference (never copied into this repo).  It exposes the
// reference's own data tables and generators so that the C restatement in sw_oracle.c can be
// pinned against them and golden fixtures can be generated (tests/golden/make_golden.py).
//
//   types.hpp:29-156 (+ BLOSUM80_20)   -> ref_blosum21()
//   types.hpp:205-396                  -> ref_blosum25()
//   convert.cuh:6-34                   -> ref_encode()
//   length_partitions.hpp:75-113       -> ref_partition_boundaries()
//   dbdata.hpp:222-272 (PseudoDBdata)  -> ref_pseudodb()
//   kseqpp/kseqpp.hpp:54-118           -> ref_fasta_*()
//
// The reference's scalar DP checker (cudasw4.cuh:2331-2392) is a private member of a CUDA-only
// class and cannot be compiled here; it is restated in sw_oracle.c.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "types.hpp"
#include "convert.cuh"
#include "length_partitions.hpp"
#include "dbdata.hpp"
#include "kseqpp/kseqpp.hpp"

extern "C" {

// which: 45, 50, 62, 80.  out: 21*21 int8. returns dim (21) or -1.
int ref_blosum21(int which, int8_t* out) {
    auto put = [&](auto flat) {
        for (size_t i = 0; i < flat.size(); i++) out[i] = flat[i];
        return int(21);
    };
    switch (which) {
        case 45: return put(cudasw4::BLOSUM45_20::get1D());
        case 50: return put(cudasw4::BLOSUM50_20::get1D());
        case 62: return put(cudasw4::BLOSUM62_20::get1D());
        case 80: return put(cudasw4::BLOSUM80_20::get1D());
    }
    return -1;
}

// which: 45, 50, 62, 80.  out: 25*25 int8 (types.hpp:205-396, letter order ARNDCQEGHILKMFPSTWYVBJZX*). returns dim (25) or -1.
int ref_blosum25(int which, int8_t* out) {
    auto put = [&](auto flat) {
        for (size_t i = 0; i < flat.size(); i++) out[i] = flat[i];
        return int(25);
    };
    switch (which) {
        case 45: return put(cudasw4::BLOSUM45::get1D());
        case 50: return put(cudasw4::BLOSUM50::get1D());
        case 62: return put(cudasw4::BLOSUM62::get1D());
        case 80: return put(cudasw4::BLOSUM80::get1D());
    }
    return -1;
}

void ref_encode(const char* in, int8_t* out, size_t n) {
    cudasw4::ConvertAA_20 conv;
    for (size_t i = 0; i < n; i++) out[i] = conv(in[i]);
}

int ref_partition_boundaries(int32_t* out, int cap) {
    auto b = cudasw4::getLengthPartitionBoundaries();
    int n = int(b.size());
    for (int i = 0; i < n && i < cap; i++) out[i] = b[i];
    return n;
}

// Fills chars (num * ceil4(length) bytes), lengths[num], offsets[num+1]. Returns bytes per sequence.
size_t ref_pseudodb(size_t num, int32_t length, int seed, int8_t* chars, int32_t* lengths, uint64_t* offsets) {
    cudasw4::PseudoDBdata db(num, length, seed);
    std::memcpy(chars, db.chars(), db.numChars());
    std::memcpy(lengths, db.lengths(), sizeof(int32_t) * num);
    for (size_t i = 0; i <= num; i++) offsets[i] = db.offsets()[i];
    return db.numChars() / (num ? num : 1);
}

// FASTA reading through the reference's kseqpp parser.
struct RefFasta { std::vector<std::string> headers, seqs; };

void* ref_fasta_open(const char* path) {
    auto* f = new RefFasta;
    kseqpp::KseqPP reader(path);
    while (reader.next() >= 0) {
        f->headers.push_back(reader.getCurrentHeader());
        f->seqs.push_back(reader.getCurrentSequence());
    }
    return f;
}
int ref_fasta_count(void* h) { return int(static_cast<RefFasta*>(h)->seqs.size()); }
int ref_fasta_seqlen(void* h, int i) { return int(static_cast<RefFasta*>(h)->seqs[i].size()); }
const char* ref_fasta_seq(void* h, int i) { return static_cast<RefFasta*>(h)->seqs[i].c_str(); }
const char* ref_fasta_header(void* h, int i) { return static_cast<RefFasta*>(h)->headers[i].c_str(); }
void ref_fasta_close(void* h) { delete static_cast<RefFasta*>(h); }

}  // extern "C"
