#include "encoder.hpp"

namespace dnas {

void Encoder::encodeSymbol(char sym) {
  // framing: what has to go in front of `sym`
  std::string run;
  if (!started_ && sym != kSOF && frontier_.accepts(kSOF)) run.push_back(kSOF);
  run.push_back(sym);
  for (size_t i = 0; i < run.size(); ++i) {
    const char c = run[i];
    if (c != kFlush && !frontier_.accepts(c)) {
      warnings_.push_back("Sending FLUSH. Depending on the code, this may insert extra bits!");
      frontier_.feed(kFlush);
    }
    started_ = started_ || c == kSOF;
    ended_ = ended_ || c == kEOF;
    frontier_.feed(c);
  }
}

void Encoder::encodeByte(unsigned char byte) {
  for (unsigned bit = 0; bit < 8; ++bit) encodeSymbol((byte >> bit) & 1u ? '1' : '0');
}

void Encoder::close() {
  if (closed_) return;
  closed_ = true;
  if (!ended_) encodeSymbol(kEOF);
  const std::string ambiguity = frontier_.finish();
  if (!ambiguity.empty()) warnings_.push_back("Encoder unresolved: " + ambiguity);
}

}  // namespace dnas
