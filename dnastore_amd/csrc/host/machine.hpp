// Host-side mirror of the reference's transducer model for the Viterbi path:
// Machine / MachineState / MachineTransition (reference src/trans.h:50-126) and the
// JSON file format of src/trans.cpp:402-469.  Only what the decoder path needs.
#pragma once
#include <cstdint>
#include <iosfwd>
#include <string>
#include <vector>

namespace dnas {

// input-symbol classes, reference src/trans.h:13-37
constexpr char kNull = '\0';
constexpr char kFlush = '.';
constexpr char kSOF = '^';
constexpr char kEOF = '$';
constexpr char kWildContext = '*';

enum InputFlags : int {  // reference src/trans.h:39-45
  kStrictInput = 1, kRelaxedInput = 2, kFlushInput = 4, kControlInput = 8, kSEOFInput = 16
};

struct MachineTransition {
  char in = 0;   // 0 = no input consumed
  char out = 0;  // 0 = nothing emitted
  uint32_t dest = 0;
};

struct MachineState {
  std::string name, leftContext, rightContext;
  std::vector<MachineTransition> trans;
};

struct Machine {
  std::vector<MachineState> state;

  size_t nStates() const { return state.size(); }

  // Throws std::runtime_error with the reference's messages on malformed input.
  static Machine fromJSON(const std::string& text);      // trans.cpp:431-469
  static Machine fromFile(const std::string& path);      // trans.cpp:477-482
  void writeJSON(std::ostream& out) const;               // trans.cpp:402-429

  void verifyContexts() const;                           // trans.cpp:484-496
  size_t maxLeftContext() const;                         // trans.cpp:246-251
  std::string inputAlphabet(int flags) const;            // trans.cpp:280-292
  std::string outputAlphabet() const;                    // trans.cpp:294-301
  // Kahn order over usable non-emitting transitions; throws std::domain_error
  // "Transducer is cyclic, can't toposort" (trans.cpp:604-634).
  std::vector<uint32_t> decoderToposort(const std::string& inputAlphabet) const;

  // Composition first*second (first's output feeds second's input): product states i*|B|+j,
  // B turned into a waiting machine first, unreachable / dead-end states pruned, chains of lone
  // null transitions collapsed (reference Machine::compose, trans.cpp:505-602).  State names,
  // order and transition order follow the reference so that the saved JSON is identical.
  static Machine compose(const Machine& first, const Machine& second);
  Machine waitingMachine() const;                        // trans.cpp:636-670
  bool isWaitingMachine() const;                         // trans.cpp:498-503

  static bool isControl(char c) { return c >= 'A' && c <= 'Z'; }
  static bool isRelaxed(char c) { return c == '0' || c == '1'; }
  static bool isStrict(char c) {
    return c == 'i' || c == 'j' || c == 'x' || c == 'y' || c == 'z' || c == 'p' || c == 'q' || c == 'r' || c == 's';
  }
};

// ACGT -> 0..3, case-insensitive; -1 otherwise (reference kmer.h:40-44, fastseq.cpp:9-15)
inline int charToBase(char c) {
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;
  }
}

}  // namespace dnas
