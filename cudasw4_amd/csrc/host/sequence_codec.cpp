#include "sequence_codec.hpp"

namespace swh {
namespace {

#define SW_BLOSUM_TABLE(which, low, ...)                   \
    constexpr int8_t kCore##which[400] = {__VA_ARGS__};    \
    constexpr int8_t kLow##which = (low);
#define SW_BLOSUM_TABLE25(which, ...) constexpr int8_t kFull##which[625] = {__VA_ARGS__};
#include "../blosum_tables.inc"
#undef SW_BLOSUM_TABLE
#undef SW_BLOSUM_TABLE25

SubstitutionMatrix make(MatrixId id, const int8_t* core, int8_t low, int gop, int gex, const char* name) {
    SubstitutionMatrix s{id, kAlphabet, {}, gop, gex, name};
    for (int i = 0; i < kAlphabet; i++)
        for (int j = 0; j < kAlphabet; j++) s.m[i * kAlphabet + j] = (i < 20 && j < 20) ? core[i * 20 + j] : low;
    return s;
}

SubstitutionMatrix make25(MatrixId id, const int8_t* full, int gop, int gex, const char* name) {
    SubstitutionMatrix s{id, 25, {}, gop, gex, name};
    for (int i = 0; i < 625; i++) s.m[i] = full[i];
    return s;
}

}  // namespace

const SubstitutionMatrix& substitution_matrix(MatrixId id) {
    static const SubstitutionMatrix b45 = make(MatrixId::Blosum45, kCore45, kLow45, -13, -2, "blosum45");
    static const SubstitutionMatrix b50 = make(MatrixId::Blosum50, kCore50, kLow50, -13, -2, "blosum50");
    static const SubstitutionMatrix b62 = make(MatrixId::Blosum62, kCore62, kLow62, -11, -1, "blosum62");
    static const SubstitutionMatrix b80 = make(MatrixId::Blosum80, kCore80, kLow80, -10, -1, "blosum80");
    static const SubstitutionMatrix f45 = make25(MatrixId::Blosum45Full, kFull45, -13, -2, "blosum45_25");
    static const SubstitutionMatrix f50 = make25(MatrixId::Blosum50Full, kFull50, -13, -2, "blosum50_25");
    static const SubstitutionMatrix f62 = make25(MatrixId::Blosum62Full, kFull62, -11, -1, "blosum62_25");
    static const SubstitutionMatrix f80 = make25(MatrixId::Blosum80Full, kFull80, -10, -1, "blosum80_25");
    switch (id) {
        case MatrixId::Blosum45: return b45;
        case MatrixId::Blosum50: return b50;
        case MatrixId::Blosum80: return b80;
        case MatrixId::Blosum45Full: return f45;
        case MatrixId::Blosum50Full: return f50;
        case MatrixId::Blosum62Full: return f62;
        case MatrixId::Blosum80Full: return f80;
        default: return b62;
    }
}

bool parse_matrix_name(const std::string& name, MatrixId& out) {
    std::string n = name;
    const std::string full = "_25";
    if (n.size() > full.size() && n.compare(n.size() - full.size(), full.size(), full) == 0) {
        n.resize(n.size() - full.size());
        if (n == "blosum45") { out = MatrixId::Blosum45Full; return true; }
        if (n == "blosum50") { out = MatrixId::Blosum50Full; return true; }
        if (n == "blosum62") { out = MatrixId::Blosum62Full; return true; }
        if (n == "blosum80") { out = MatrixId::Blosum80Full; return true; }
        return false;
    }
    const std::string suffix = "_20";
    if (n.size() > suffix.size() && n.compare(n.size() - suffix.size(), suffix.size(), suffix) == 0) n.resize(n.size() - suffix.size());
    if (n == "blosum45") { out = MatrixId::Blosum45; return true; }
    if (n == "blosum50") { out = MatrixId::Blosum50; return true; }
    if (n == "blosum62") { out = MatrixId::Blosum62; return true; }
    if (n == "blosum80") { out = MatrixId::Blosum80; return true; }
    return false;
}

}  // namespace swh
