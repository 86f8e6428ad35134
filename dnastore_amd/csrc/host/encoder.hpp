// Exact (noise-free) encoder: input symbols -> DNA through a Machine, tracking the set
// of states the transducer may be in together with each one's pending output, as the
// reference's Encoder<Writer> does (src/encoder.h:7-243).  Used to make synthetic reads
// (bench.py, tests) and by the CLI's --encode-* arms.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "machine.hpp"

namespace dnas {

class Encoder {
 public:
  explicit Encoder(const Machine& machine);
  void encodeSymbol(char sym);                 // encoder.h:143-186
  void encodeSymbolString(const std::string& s) { for (char c : s) encodeSymbol(c); }
  void encodeByte(unsigned char byte);         // LSB first, encoder.h:222-231
  void encodeBytes(const std::string& bytes) { for (unsigned char c : bytes) encodeByte(c); }
  void close();                                // encoder.h:33-57
  const std::string& output() const { return out_; }
  const std::vector<std::string>& warnings() const { return warnings_; }

 private:
  typedef std::map<uint32_t, std::string> StateString;
  bool canEncodeSymbol(char sym) const;
  void expand();                               // encoder.h:76-121
  void shiftResolvedSymbols();                 // encoder.h:188-216
  static bool exitsWithInput(const MachineState& ms);
  static bool emitsOutput(const MachineState& ms);

  const Machine& machine_;
  StateString current_;
  bool sentSOF_ = false, sentEOF_ = false, closed_ = false;
  std::string out_;
  std::vector<std::string> warnings_;
};

}  // namespace dnas
