#include "frontier.hpp"

#include <algorithm>
#include <stdexcept>

namespace dnas {

Frontier::Frontier(const Machine& machine, Fed fed) : machine_(machine), fed_(fed) {
  if (machine.nStates() == 0) throw std::runtime_error("Machine has no states");
  live_.push_back(Hyp{0, 0, 0});
  followSilent();
}

// Reading the output tape (decoding), only the transitions a payload can take count: no input, a bit, a control
// symbol, start or end of file -- the strict-radix and flush inputs are the encoder's business.
bool Frontier::usable(const MachineTransition& t) const {
  if (fed_ == kFeedInput) return true;
  return t.in == kNull || t.in == '0' || t.in == '1' || t.in == kEOF || t.in == kSOF || Machine::isControl(t.in);
}

bool Frontier::restsAt(uint32_t state) const {
  const auto& ts = machine_.state[state].trans;
  return ts.empty() || std::any_of(ts.begin(), ts.end(), [&](const MachineTransition& t) { return reads(t) != 0; });
}

bool Frontier::speaksAt(uint32_t state) const {
  const auto& ts = machine_.state[state].trans;
  return std::any_of(ts.begin(), ts.end(), [&](const MachineTransition& t) { return writes(t) != 0; });
}

Frontier::Hyp Frontier::extended(const Hyp& h, char written, uint32_t dest) {
  // slices are immutable once shared: an extension is a fresh copy at the end of the arena
  const uint32_t at = (uint32_t)arena_.size();
  arena_.append(arena_, h.begin, h.end - h.begin);
  if (written) arena_.push_back(written);
  return Hyp{dest, at, (uint32_t)arena_.size()};
}

// One hypothesis per state: a second way into a state must imply the same symbols.
void Frontier::admit(std::vector<Hyp>* into, const Hyp& h, const char* what) const {
  for (const Hyp& o : *into)
    if (o.state == h.state) {
      if (arena_.compare(o.begin, o.end - o.begin, arena_, h.begin, h.end - h.begin) != 0)
        throw std::runtime_error(std::string(what) + " error: state " + machine_.state[h.state].name + " has two possible " +
                                 (fed_ == kFeedInput ? "output" : "input") + " queues (" + arena_.substr(o.begin, o.end - o.begin) + ", " +
                                 arena_.substr(h.begin, h.end - h.begin) + ")");
      return;
    }
  into->push_back(h);
}

// Everything reachable without reading the fed tape; only resting states survive.
void Frontier::followSilent() {
  const char* what = fed_ == kFeedInput ? "Encoder" : "Decoder";
  std::vector<Hyp> reached, todo(live_.rbegin(), live_.rend());
  while (!todo.empty()) {
    const Hyp h = todo.back();
    todo.pop_back();
    const size_t before = reached.size();
    admit(&reached, h, what);
    if (reached.size() == before) continue;          // seen, with the same symbols
    for (const MachineTransition& t : machine_.state[h.state].trans)
      if (usable(t) && reads(t) == 0) todo.push_back(extended(h, writes(t), t.dest));
  }
  live_.clear();
  for (const Hyp& h : reached)
    if (restsAt(h.state)) live_.push_back(h);
  std::sort(live_.begin(), live_.end(), [](const Hyp& a, const Hyp& b) { return a.state < b.state; });
}

bool Frontier::accepts(char sym) const {
  for (const Hyp& h : live_)
    for (const MachineTransition& t : machine_.state[h.state].trans)
      if (usable(t) && reads(t) == sym) return true;
  return false;
}

void Frontier::feed(char sym) {
  const char* what = fed_ == kFeedInput ? "Encoder" : "Decoder";
  std::vector<Hyp> moved;
  for (const Hyp& h : std::vector<Hyp>(live_))
    for (const MachineTransition& t : machine_.state[h.state].trans)
      if (usable(t) && reads(t) == sym) admit(&moved, extended(h, writes(t), t.dest), what);
  if (moved.empty()) throw std::runtime_error(std::string(fed_ == kFeedInput ? "Can't encode symbol '" : "Can't decode '") + sym + "'");
  live_.swap(moved);
  followSilent();
  settle();
  compact();
}

// What every hypothesis agrees on is final.  A single hypothesis agrees with itself on everything -- but it is
// only trusted once its state can write again (until then the next fed symbol may still revise the picture).
void Frontier::settle() {
  if (live_.empty()) return;
  if (live_.size() == 1) {
    Hyp& h = live_.front();
    if (speaksAt(h.state)) {
      settled_.append(arena_, h.begin, h.end - h.begin);
      h.begin = h.end;
    }
    return;
  }
  uint32_t common = live_.front().end - live_.front().begin;
  for (size_t i = 1; i < live_.size() && common; ++i) {
    const Hyp &a = live_.front(), &b = live_[i];
    const uint32_t n = std::min(common, b.end - b.begin);
    uint32_t k = 0;
    while (k < n && arena_[a.begin + k] == arena_[b.begin + k]) ++k;
    common = k;
  }
  if (common) {
    settled_.append(arena_, live_.front().begin, common);
    for (Hyp& h : live_) h.begin += common;
  }
}

void Frontier::compact() {
  std::string fresh;
  for (Hyp& h : live_) {
    const uint32_t at = (uint32_t)fresh.size();
    fresh.append(arena_, h.begin, h.end - h.begin);
    h.end = at + (h.end - h.begin);
    h.begin = at;
  }
  arena_.swap(fresh);
}

std::string Frontier::finish() {
  std::string ambiguity;
  if (live_.empty()) return ambiguity;
  followSilent();
  std::vector<const Hyp*> finals;
  for (const Hyp& h : live_)
    if (machine_.state[h.state].trans.empty()) finals.push_back(&h);
  if (finals.size() == 1)
    settled_.append(arena_, finals.front()->begin, finals.front()->end - finals.front()->begin);
  else if (finals.size() > 1)
    ambiguity = std::to_string(finals.size()) + " possible end states";
  else if (live_.size() > 1)
    ambiguity = std::to_string(live_.size()) + " possible states";
  live_.clear();
  arena_.clear();
  return ambiguity;
}

}  // namespace dnas
