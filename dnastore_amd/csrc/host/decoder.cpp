#include "decoder.hpp"

#include <algorithm>
#include <cctype>
#include <stdexcept>

namespace dnas {

bool Decoder::isUsable(const MachineTransition& t) {
  return t.in == kNull || t.in == '0' || t.in == '1' || t.in == kEOF || t.in == kSOF || Machine::isControl(t.in);
}

Decoder::Decoder(const Machine& machine) : machine_(machine) {
  if (machine.nStates() == 0) throw std::runtime_error("Machine has no states");
  current_[0] = std::string();
  expand();
}

// Follow non-emitting usable transitions to a fixed point; keep end states and states that emit.
void Decoder::expand() {
  StateString next, seen;
  bool foundNew;
  do {
    foundNew = false;
    for (const auto& ss : current_) {
      seen.insert(ss);
      const MachineState& ms = machine_.state[ss.first];
      bool emits = false;
      for (const auto& t : ms.trans) if (t.out) emits = true;
      if (ms.trans.empty() || emits) next[ss.first] = ss.second;
    }
    for (const auto& ss : current_)
      for (const auto& t : machine_.state[ss.first].trans)
        if (isUsable(t) && !t.out) {
          std::string q = ss.second;
          if (t.in) q.push_back(t.in);
          auto it = seen.find(t.dest);
          if (it != seen.end()) {
            if (it->second != q)
              throw std::runtime_error("Decoder error: state " + machine_.state[t.dest].name + " has two possible input queues (" +
                                       it->second + ", " + q + ")");
          } else {
            next[t.dest] = q;
            foundNew = true;
          }
        }
    current_.swap(next);
    next.clear();
  } while (foundNew);
}

void Decoder::shiftResolvedSymbols() {
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

void Decoder::decodeSymbol(char outSym) {
  StateString next;
  for (const auto& ss : current_)
    for (const auto& t : machine_.state[ss.first].trans)
      if (isUsable(t) && t.out == outSym) {
        std::string q = ss.second;
        if (t.in) q.push_back(t.in);
        auto it = next.find(t.dest);
        if (it != next.end() && it->second != q)
          throw std::runtime_error("Decoder error: state " + machine_.state[t.dest].name + " has two possible input queues (" +
                                   it->second + ", " + q + ")");
        next[t.dest] = q;
      }
  if (next.empty()) throw std::runtime_error(std::string("Can't decode '") + outSym + "'");
  current_.swap(next);
  expand();
  if (current_.size() == 1) {
    auto it = current_.begin();
    bool exitsWithInput = false;
    for (const auto& t : machine_.state[it->first].trans) if (t.in) exitsWithInput = true;
    if (exitsWithInput) {
      out_ += it->second;
      it->second.clear();
    }
  } else {
    shiftResolvedSymbols();
  }
}

void Decoder::decodeString(const std::string& seq) {
  for (char c : seq) decodeSymbol((char)toupper((unsigned char)c));
}

void Decoder::close() {
  if (closed_) return;
  closed_ = true;
  if (!current_.empty()) {
    expand();
    std::vector<StateString::iterator> ends;
    for (auto it = current_.begin(); it != current_.end(); ++it)
      if (machine_.state[it->first].trans.empty()) ends.push_back(it);
    if (ends.size() == 1) {
      out_ += ends.front()->second;
      ends.front()->second.clear();
    } else if (ends.size() > 1) {
      warnings_.push_back("Decoder unresolved: " + std::to_string(ends.size()) + " possible end states");
    } else if (current_.size() > 1) {
      warnings_.push_back("Decoder unresolved: " + std::to_string(current_.size()) + " possible states");
    }
    current_.clear();
  }
}

std::string symbolsToBytes(const std::string& symbols, std::string* leftover) {
  std::string bytes;
  std::vector<bool> buf;
  for (char c : symbols) {
    if (c != '0' && c != '1') continue;   // control / SOF / EOF symbols are ignored (decoder.h:225-236)
    buf.push_back(c == '1');
    if (buf.size() == 8) {
      unsigned char b = 0;
      for (size_t n = 0; n < 8; ++n) if (buf[n]) b |= (unsigned char)(1u << n);   // LSB first (decoder.h:211-219)
      bytes.push_back((char)b);
      buf.clear();
    }
  }
  if (leftover) {
    leftover->clear();
    std::reverse(buf.begin(), buf.end());
    for (bool bit : buf) leftover->push_back(bit ? '1' : '0');
  }
  return bytes;
}

}  // namespace dnas
