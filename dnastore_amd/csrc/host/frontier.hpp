// Noise-free transduction in either direction, as one mechanism.
//
// A transducer read along one of its two tapes is nondeterministic: after a prefix of the fed tape it may sit in
// several states, and each of them implies what was written on the OTHER tape so far.  A Frontier keeps those
// hypotheses as a flat list of (state, slice) over one shared symbol arena; feeding a symbol moves every hypothesis
// along the transitions that read it, then along the transitions that read nothing, and whatever all surviving
// hypotheses agree on -- the longest common prefix of their slices -- is settled and leaves the arena.
//
//   fed tape = input   ->  the exact encoder  (behaviour of the reference's Encoder<Writer>, src/encoder.h)
//   fed tape = output  ->  the exact decoder  (behaviour of the reference's Decoder<Writer>, src/decoder.h)
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "machine.hpp"

namespace dnas {

class Frontier {
 public:
  enum Fed { kFeedInput, kFeedOutput };
  Frontier(const Machine& machine, Fed fed);

  bool accepts(char sym) const;                 // some hypothesis can read `sym` next
  void feed(char sym);                          // throws std::runtime_error when none can
  // End of the fed tape: settle what the surviving end state implies.  Returns a description of the ambiguity
  // ("2 possible end states", "3 possible states") or an empty string.
  std::string finish();
  const std::string& settled() const { return settled_; }
  bool empty() const { return live_.empty(); }

 private:
  struct Hyp { uint32_t state, begin, end; };   // arena_[begin, end) = what the other tape holds beyond settled_
  char reads(const MachineTransition& t) const { return fed_ == kFeedInput ? t.in : t.out; }
  char writes(const MachineTransition& t) const { return fed_ == kFeedInput ? t.out : t.in; }
  bool usable(const MachineTransition& t) const;
  bool restsAt(uint32_t state) const;           // the walk along silent transitions may stop here
  bool speaksAt(uint32_t state) const;          // a lone hypothesis here has nothing left to wait for
  Hyp extended(const Hyp& h, char written, uint32_t dest);
  void admit(std::vector<Hyp>* into, const Hyp& h, const char* what) const;
  void followSilent();
  void settle();
  void compact();

  const Machine& machine_;
  const Fed fed_;
  std::vector<Hyp> live_;                       // sorted by state, one hypothesis per state
  std::string arena_, settled_;
};

}  // namespace dnas
