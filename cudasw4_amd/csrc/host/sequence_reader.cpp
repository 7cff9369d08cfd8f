#include "sequence_reader.hpp"

#include <zlib.h>

#include <cstring>
#include <stdexcept>
#include <vector>

namespace swh {

struct SequenceReader::Impl {
    gzFile file = nullptr;
    std::vector<unsigned char> buf = std::vector<unsigned char>(1 << 16);
    int begin = 0, end = 0;
    bool eof = false;

    bool refill() {
        if (eof) return false;
        const int n = gzread(file, buf.data(), (unsigned)buf.size());
        begin = 0;
        end = n > 0 ? n : 0;
        if (n <= 0) eof = true;
        return n > 0;
    }
};

SequenceReader::SequenceReader(const std::string& path) : impl_(new Impl) {
    impl_->file = gzopen(path.c_str(), "rb");  // transparent for uncompressed files
    if (!impl_->file) throw std::runtime_error("Cannot open file " + path);
    gzbuffer(impl_->file, 1 << 18);
}

SequenceReader::~SequenceReader() {
    if (impl_ && impl_->file) gzclose(impl_->file);
}

int SequenceReader::getc() {
    Impl& s = *impl_;
    if (s.begin >= s.end && !s.refill()) return -1;
    return s.buf[s.begin++];
}

bool SequenceReader::getline(std::string& out, bool append) {
    Impl& s = *impl_;
    if (!append) out.clear();
    bool got = false;
    for (;;) {
        if (s.begin >= s.end && !s.refill()) break;
        got = true;
        const unsigned char* p = s.buf.data() + s.begin;
        const unsigned char* nl = (const unsigned char*)memchr(p, '\n', s.end - s.begin);
        const int stop = nl ? (int)(nl - s.buf.data()) : s.end;
        out.append((const char*)p, stop - s.begin);
        s.begin = stop + 1;
        if (nl) break;
    }
    if (out.size() > 1 && out.back() == '\r') out.pop_back();
    return got;
}

bool SequenceReader::next() {
    int c = pending_;
    if (c == 0) {
        while ((c = getc()) >= 0 && c != '>' && c != '@') {
        }
        if (c < 0) return false;
    }
    pending_ = 0;
    seq_.clear();
    qual_.clear();
    if (!getline(header_, false)) return false;
    while ((c = getc()) >= 0 && c != '>' && c != '+' && c != '@') {
        if (c == '\n') continue;
        seq_.push_back((char)c);
        getline(seq_, true);
    }
    if (c == '>' || c == '@') pending_ = c;
    if (c != '+') return true;  // FASTA record (or last record)
    while ((c = getc()) >= 0 && c != '\n') {
    }
    if (c < 0) return false;  // no quality string
    while (qual_.size() < seq_.size() && getline(qual_, true)) {
    }
    return qual_.size() == seq_.size();
}

}  // namespace swh
