#include "decoder.hpp"

#include <cctype>

namespace dnas {

void Decoder::decodeString(const std::string& seq) {
  for (char c : seq) frontier_.feed((char)toupper((unsigned char)c));
}

void Decoder::close() {
  if (closed_) return;
  closed_ = true;
  const std::string ambiguity = frontier_.finish();
  if (!ambiguity.empty()) warnings_.push_back("Decoder unresolved: " + ambiguity);
}

std::string symbolsToBytes(const std::string& symbols, std::string* leftover) {
  std::string bytes;
  unsigned acc = 0, filled = 0;
  for (char c : symbols) {
    if (c != '0' && c != '1') continue;
    acc |= (unsigned)(c - '0') << filled;
    if (++filled == 8) {
      bytes.push_back((char)acc);
      acc = filled = 0;
    }
  }
  if (leftover) {
    leftover->clear();
    for (unsigned bit = filled; bit-- > 0;) leftover->push_back((acc >> bit) & 1u ? '1' : '0');
  }
  return bytes;
}

}  // namespace dnas
