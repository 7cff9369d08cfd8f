// sequence_reader.hpp — FASTA / FASTQ reader (plain or gzip) for makedb and align.
//
// Behaviour follows the reference's kseq-derived parser (kseqpp/kseqpp.hpp:54-118,247-290):
// records start at '>' or '@'; the header is the whole rest of that line; sequence lines are
// concatenated verbatim (no case folding) until a line starts with '>', '@' or '+'; a '+' line
// starts FASTQ qualities, which are read (and discarded) until they are as long as the sequence;
// empty lines are skipped; a trailing '\r' is dropped.
#pragma once
#include <memory>
#include <string>

namespace swh {

class SequenceReader {
public:
    explicit SequenceReader(const std::string& path);  // throws std::runtime_error if the file cannot be opened
    ~SequenceReader();
    SequenceReader(const SequenceReader&) = delete;
    SequenceReader& operator=(const SequenceReader&) = delete;

    // Advances to the next record. Returns false at end of input (or on a malformed FASTQ tail).
    bool next();
    const std::string& header() const { return header_; }
    const std::string& sequence() const { return seq_; }

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
    std::string header_, seq_, qual_;
    int pending_ = 0;  // record-start character already consumed by the previous call
    int getc();
    // reads up to (not including) '\n'; returns false if nothing could be read at EOF
    bool getline(std::string& out, bool append);
};

}  // namespace swh
