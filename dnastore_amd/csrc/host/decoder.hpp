// Exact (noise-free) decoder: DNA -> input symbols through a Machine, tracking the set of
// states the transducer may be in with each one's pending input queue, as the reference's
// Decoder<Writer> does (src/decoder.h:7-191), and the bit packer BinaryWriter
// (src/decoder.h:193-240).  CPU only; serves BASELINE config 1 and turns Viterbi output
// symbol strings into payload bytes.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "machine.hpp"

namespace dnas {

class Decoder {
 public:
  explicit Decoder(const Machine& machine);
  void decodeSymbol(char outSym);                  // decoder.h:123-159
  void decodeString(const std::string& seq);       // upper-cases, decoder.h:187-190
  void close();                                    // decoder.h:27-47
  const std::string& symbols() const { return out_; }
  const std::vector<std::string>& warnings() const { return warnings_; }

 private:
  typedef std::map<uint32_t, std::string> StateString;
  static bool isUsable(const MachineTransition& t);   // decoder.h:116-121
  void expand();
  void shiftResolvedSymbols();
  const Machine& machine_;
  StateString current_;
  std::string out_;
  std::vector<std::string> warnings_;
  bool closed_ = false;
};

// BinaryWriter (decoder.h:193-240): '0'/'1' symbols -> bytes, LSB first; other symbols ignored.
// `leftover` receives the bits that did not fill a byte (as the reference's warning prints them).
std::string symbolsToBytes(const std::string& symbols, std::string* leftover = nullptr);

}  // namespace dnas
