// FASTA / FASTQ (optionally gzipped, multi-line) reader and 50-column FASTA writer with
// the behaviour of the reference's readFastSeqs / writeFastaSeqs
// (src/fastseq.cpp:82-90,123-148; record grammar as klib kseq: name = first word of the
// header, comment = the rest, sequence = all graphic characters up to the next record).
#pragma once
#include <cstdint>
#include <iosfwd>
#include <string>
#include <vector>

namespace dnas {

struct FastSeq {
  std::string name, comment, seq, qual;
};

// Throws std::runtime_error("Couldn't open <path>") when the file is unreadable.
std::vector<FastSeq> readFastSeqs(const std::string& path);
void writeFastaSeqs(std::ostream& out, const std::vector<FastSeq>& seqs, size_t width = 50);

// ACGT (any case) -> 0..3.  Throws std::runtime_error naming the sequence on any other
// character (reference: message on stderr, then terminate -- fastseq.cpp:25-39).
std::vector<uint8_t> tokenizeDNA(const std::string& seq, const std::string& name);

}  // namespace dnas
