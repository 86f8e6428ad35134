// Exact (noise-free) encoder: input symbols -> DNA through a Machine (behaviour of the reference's
// Encoder<Writer>, src/encoder.h:7-243): a Frontier fed along the input tape, plus the framing rules of the input
// stream -- start-of-file is sent first when the machine can take it, a symbol the machine cannot take is
// preceded by a flush, end-of-file is sent on close.  Used to make synthetic reads (bench.py, tests) and by the
// CLI's --encode-* arms.
#pragma once
#include <string>
#include <vector>

#include "frontier.hpp"

namespace dnas {

class Encoder {
 public:
  explicit Encoder(const Machine& machine) : frontier_(machine, Frontier::kFeedInput) {}
  void encodeSymbol(char sym);
  void encodeSymbolString(const std::string& s) { for (char c : s) encodeSymbol(c); }
  void encodeByte(unsigned char byte);         // bits LSB first (encoder.h:222-231)
  void encodeBytes(const std::string& bytes) { for (unsigned char c : bytes) encodeByte(c); }
  void close();
  const std::string& output() const { return frontier_.settled(); }
  const std::vector<std::string>& warnings() const { return warnings_; }

 private:
  Frontier frontier_;
  bool started_ = false, ended_ = false, closed_ = false;
  std::vector<std::string> warnings_;
};

}  // namespace dnas
