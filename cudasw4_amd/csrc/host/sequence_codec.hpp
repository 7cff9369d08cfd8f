// sequence_codec.hpp — residue encoding and substitution matrices of the host driver.
//
//   encode_residue / encode_in_place : ConvertAA_20 (reference convert.cuh:6-34)
//   decode_residue                   : InverseConvertAA_20 (convert.cuh:36-64)
//   SubstitutionMatrix               : the 21 x 21 "_20" tables (types.hpp:29-270) and the full 25 x 25 tables
//                                      (types.hpp:205-396); generated data in ../blosum_tables.inc (oracle/gen_tables.py)
//   encode_residue25                 : query letters for the 25 x 25 tables, order ARNDCQEGHILKMFPSTWYVBJZX*; anything
//                                      else -> X.  (The reference has no such encoder: its CAN_USE_FULL_BLOSUM build still
//                                      encodes with ConvertAA_20, so only the DB-side alphabet is its; see include/cudasw4_amd.h.)
#pragma once
#include <array>
#include <cstddef>
#include <cstdint>
#include <string>

namespace swh {

constexpr int kAlphabet = 21;       // 20 amino acids + "other"
constexpr int8_t kOtherCode = 20;

struct ResidueCodec {
    std::array<int8_t, 256> table{};
    constexpr ResidueCodec() {
        for (auto& t : table) t = kOtherCode;
        const char* letters = "ARNDCQEGHILKMFPSTWYV";
        for (int i = 0; i < 20; i++) table[(unsigned char)letters[i]] = (int8_t)i;
    }
};
inline constexpr ResidueCodec kCodec{};

inline int8_t encode_residue(char c) { return kCodec.table[(unsigned char)c]; }
inline void encode_in_place(char* data, size_t n) {
    for (size_t i = 0; i < n; i++) data[i] = (char)kCodec.table[(unsigned char)data[i]];
}
struct ResidueCodec25 {
    std::array<int8_t, 256> table{};
    constexpr ResidueCodec25() {
        for (auto& t : table) t = 23;  // X
        const char* letters = "ARNDCQEGHILKMFPSTWYVBJZX*";
        for (int i = 0; i < 25; i++) table[(unsigned char)letters[i]] = (int8_t)i;
    }
};
inline constexpr ResidueCodec25 kCodec25{};
inline int8_t encode_residue25(char c) { return kCodec25.table[(unsigned char)c]; }

inline char decode_residue(int8_t code) { return (code >= 0 && code < 20) ? "ARNDCQEGHILKMFPSTWYV"[code] : '-'; }

// types.hpp:18-27 BlosumType: the "_20" tables first (what the reference's default build maps every --mat name to),
// then the full 25-letter ones
enum class MatrixId { Blosum45, Blosum50, Blosum62, Blosum80, Blosum45Full, Blosum50Full, Blosum62Full, Blosum80Full };

struct SubstitutionMatrix {
    MatrixId id;
    int dim;                                      // 21 or 25
    std::array<int8_t, 25 * 25> m;                // row-major dim x dim
    int default_gop;                              // options.cpp:179-194
    int default_gex;
    const char* name;
};

const SubstitutionMatrix& substitution_matrix(MatrixId id);
// "blosum62", "blosum62_20" -> the 21-letter tables like the reference's default build (options.cpp:144-152);
// "blosum62_25", ... -> the full tables (what a CAN_USE_FULL_BLOSUM build selects with the plain names, options.cpp:135-143);
// returns false for unknown names
bool parse_matrix_name(const std::string& name, MatrixId& out);

}  // namespace swh
