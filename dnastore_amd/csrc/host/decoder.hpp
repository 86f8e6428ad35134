// Exact (noise-free) decoder: DNA -> input symbols through a Machine (behaviour of the reference's
// Decoder<Writer>, src/decoder.h:7-191): a Frontier fed along the output tape; and the bit packer of
// BinaryWriter (src/decoder.h:193-240).  CPU only; serves BASELINE config 1 and turns Viterbi output symbol
// strings into payload bytes.
#pragma once
#include <string>
#include <vector>

#include "frontier.hpp"

namespace dnas {

class Decoder {
 public:
  explicit Decoder(const Machine& machine) : frontier_(machine, Frontier::kFeedOutput) {}
  void decodeString(const std::string& seq);       // bases in either case (decoder.h:187-190)
  void close();
  const std::string& symbols() const { return frontier_.settled(); }
  const std::vector<std::string>& warnings() const { return warnings_; }

 private:
  Frontier frontier_;
  std::vector<std::string> warnings_;
  bool closed_ = false;
};

// '0'/'1' symbols -> bytes, first symbol = least significant bit; every other symbol is skipped
// (BinaryWriter, decoder.h:193-240).  `leftover` receives the bits that did not fill a byte, most recent first
// (as the reference's warning prints them).
std::string symbolsToBytes(const std::string& symbols, std::string* leftover = nullptr);

}  // namespace dnas
