#include "encoder.hpp"

#include <stdexcept>

namespace dnas {

bool Encoder::exitsWithInput(const MachineState& ms) {
  for (const auto& t : ms.trans)
    if (t.in) return true;
  return false;
}

bool Encoder::emitsOutput(const MachineState& ms) {
  for (const auto& t : ms.trans)
    if (t.out) return true;
  return false;
}

Encoder::Encoder(const Machine& machine) : machine_(machine) {
  if (machine.nStates() == 0) throw std::runtime_error("Machine has no states");
  current_[0] = std::string();
  expand();
}

bool Encoder::canEncodeSymbol(char sym) const {
  for (const auto& ss : current_)
    for (const auto& t : machine_.state[ss.first].trans)
      if (t.in == sym) return true;
  return false;
}

// Follow input-free transitions to a fixed point; keep only states that wait for input
// or are end states.
void Encoder::expand() {
  StateString next, seen;
  bool foundNew;
  do {
    foundNew = false;
    for (const auto& ss : current_) {
      seen.insert(ss);
      const MachineState& ms = machine_.state[ss.first];
      if (ms.trans.empty() || exitsWithInput(ms)) next[ss.first] = ss.second;
    }
    for (const auto& ss : current_) {
      const MachineState& ms = machine_.state[ss.first];
      for (const auto& t : ms.trans)
        if (!t.in) {
          std::string q = ss.second;
          if (t.out) q.push_back(t.out);
          auto it = seen.find(t.dest);
          if (it != seen.end()) {
            if (it->second != q)
              throw std::runtime_error("Encoder error: state " + machine_.state[t.dest].name +
                                       " has two possible output queues (" + it->second + ", " + q + ")");
          } else {
            next[t.dest] = q;
            foundNew = true;
          }
        }
    }
    current_.swap(next);
    next.clear();
  } while (foundNew);
}

void Encoder::shiftResolvedSymbols() {
  for (;;) {
    bool foundQueue = false, queueNonempty = false, firstCharSame = false;
    char firstChar = 0;
    for (const auto& ss : current_) {
      if (!foundQueue) {
        if ((queueNonempty = !ss.second.empty())) firstChar = ss.second[0];
        foundQueue = firstCharSame = true;
      } else if (queueNonempty && (ss.second.empty() || firstChar != ss.second[0])) {
        firstCharSame = false;
        break;
      }
    }
    if (foundQueue && queueNonempty && firstCharSame) {
      out_.push_back(firstChar);
      for (auto& ss : current_) ss.second.erase(ss.second.begin());
    } else {
      break;
    }
  }
}

void Encoder::encodeSymbol(char sym) {
  if (!sentSOF_ && sym != kSOF && canEncodeSymbol(kSOF)) encodeSymbol(kSOF);
  if (sym != kFlush && !canEncodeSymbol(sym)) {
    warnings_.push_back("Sending FLUSH. Depending on the code, this may insert extra bits!");
    encodeSymbol(kFlush);
  }
  if (sym == kSOF) sentSOF_ = true;
  else if (sym == kEOF) sentEOF_ = true;

  StateString next;
  for (const auto& ss : current_)
    for (const auto& t : machine_.state[ss.first].trans)
      if (t.in == sym) {
        std::string q = ss.second;
        if (t.out) q.push_back(t.out);
        auto it = next.find(t.dest);
        if (it != next.end() && it->second != q)
          throw std::runtime_error("Encoder error: state " + machine_.state[t.dest].name +
                                   " has two possible output queues (" + it->second + ", " + q + ")");
        next[t.dest] = q;
      }
  if (next.empty()) throw std::runtime_error(std::string("Can't encode symbol '") + sym + "'");
  current_.swap(next);
  expand();
  if (current_.size() == 1) {
    auto it = current_.begin();
    if (emitsOutput(machine_.state[it->first])) {
      out_ += it->second;
      it->second.clear();
    }
  } else {
    shiftResolvedSymbols();
  }
}

void Encoder::encodeByte(unsigned char byte) {
  for (int n = 0; n <= 7; ++n) encodeSymbol((byte >> n) & 1 ? '1' : '0');
}

void Encoder::close() {
  if (closed_) return;
  closed_ = true;
  if (!sentEOF_) encodeSymbol(kEOF);
  if (!current_.empty()) {
    expand();
    std::vector<StateString::iterator> ends;
    for (auto it = current_.begin(); it != current_.end(); ++it)
      if (machine_.state[it->first].trans.empty()) ends.push_back(it);
    if (ends.size() == 1) {
      out_ += ends.front()->second;
      ends.front()->second.clear();
    } else if (ends.size() > 1) {
      warnings_.push_back("Encoder unresolved: " + std::to_string(ends.size()) + " possible end states");
    } else if (current_.size() > 1) {
      warnings_.push_back("Encoder unresolved: " + std::to_string(current_.size()) + " possible states");
    }
    current_.clear();
  }
}

}  // namespace dnas
