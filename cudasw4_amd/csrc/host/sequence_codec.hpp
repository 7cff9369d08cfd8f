// sequence_codec.hpp — residue encoding and substitution matrices of the host driver.
//
//   encode_residue / encode_in_place : ConvertAA_20 (reference convert.cuh:6-34)
//   decode_residue                   : InverseConvertAA_20 (convert.cuh:36-64)
//   SubstitutionMatrix               : the 21 x 21 "_20" tables (types.hpp:29-270); generated data in
//                                      ../blosum_tables.inc (see oracle/gen_tables.py)
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
inline char decode_residue(int8_t code) { return (code >= 0 && code < 20) ? "ARNDCQEGHILKMFPSTWYV"[code] : '-'; }

enum class MatrixId { Blosum45, Blosum50, Blosum62, Blosum80 };

struct SubstitutionMatrix {
    MatrixId id;
    std::array<int8_t, kAlphabet * kAlphabet> m;  // row-major 21 x 21
    int default_gop;                              // options.cpp:179-194
    int default_gex;
    const char* name;
};

const SubstitutionMatrix& substitution_matrix(MatrixId id);
// "blosum62", "blosum62_20", ... (options.cpp:133-153); returns false for unknown names
bool parse_matrix_name(const std::string& name, MatrixId& out);

}  // namespace swh
