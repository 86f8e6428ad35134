#include "machine.hpp"

#include <deque>
#include <fstream>
#include <ostream>
#include <set>
#include <unordered_map>
#include <sstream>
#include <stdexcept>

#include "json.hpp"

namespace dnas {

Machine Machine::fromJSON(const std::string& text) {
  // the reference joins the file's lines without newlines before parsing (jsonutil.cpp:159-169)
  std::string flat;
  flat.reserve(text.size());
  for (char c : text)
    if (c != '\n') flat += c;
  const JsonValue root = parseJson(flat);
  Machine m;
  for (const JsonValue& js : root.array("state")) {
    MachineState ms;
    if (const JsonValue* n = js.find("n")) {
      if ((size_t)n->num != m.state.size())
        throw std::runtime_error("State n=" + std::to_string((size_t)n->num) + " out of sequence");
    }
    if (js.find("id")) ms.name = js.string("id");
    if (js.find("l")) ms.leftContext = js.string("l");
    if (js.find("r")) ms.rightContext = js.string("r");
    for (const JsonValue& jt : js.array("trans")) {
      MachineTransition t;
      t.dest = (uint32_t)jt.number("to");
      if (jt.find("in")) {
        const std::string& s = jt.string("in");
        if (s.size() != 1) throw std::runtime_error("Invalid input character: " + s);
        t.in = s[0];
      }
      if (jt.find("out")) {
        const std::string& s = jt.string("out");
        if (s.size() != 1) throw std::runtime_error("Invalid output character: " + s);
        t.out = s[0];
      }
      ms.trans.push_back(t);
    }
    m.state.push_back(std::move(ms));
  }
  for (const auto& ms : m.state)
    for (const auto& t : ms.trans)
      if (t.dest >= m.state.size()) throw std::runtime_error("Transition to nonexistent state " + std::to_string(t.dest));
  m.verifyContexts();
  return m;
}

Machine Machine::fromFile(const std::string& path) {
  std::ifstream in(path);
  if (!in) throw std::runtime_error("File not found: " + path);
  std::stringstream ss;
  ss << in.rdbuf();
  return fromJSON(ss.str());
}

void Machine::writeJSON(std::ostream& out) const {
  out << "{\"state\": [\n";
  for (size_t s = 0; s < state.size(); ++s) {
    const MachineState& ms = state[s];
    out << " {\"n\":" << s << ",";
    if (!ms.name.empty()) out << "\"id\":\"" << ms.name << "\",";
    if (!ms.leftContext.empty()) out << "\"l\":\"" << ms.leftContext << "\",";
    if (!ms.rightContext.empty()) out << "\"r\":\"" << ms.rightContext << "\",";
    out << "\"trans\":[";
    for (size_t i = 0; i < ms.trans.size(); ++i) {
      const MachineTransition& t = ms.trans[i];
      if (i) out << ",";
      out << "{";
      if (t.in) out << "\"in\":\"" << t.in << "\",";
      if (t.out) out << "\"out\":\"" << t.out << "\",";
      out << "\"to\":" << t.dest << "}";
    }
    out << "]}";
    if (s + 1 < state.size()) out << ",";
    out << "\n";
  }
  out << "]}\n";
}

void Machine::verifyContexts() const {
  for (const auto& ms : state)
    for (const auto& t : ms.trans) {
      if (!t.out) continue;
      const auto& md = state[t.dest];
      if (!ms.rightContext.empty() && t.out != ms.rightContext[0])
        throw std::runtime_error("In transition from " + ms.name + " to " + md.name + ": emitted character (" + t.out +
                                 ") does not match source's right context (" + ms.rightContext + ")");
      if (!md.leftContext.empty() && t.out != md.leftContext.back())
        throw std::runtime_error("In transition from " + ms.name + " to " + md.name + ": emitted character (" + t.out +
                                 ") does not match destination's left context (" + md.leftContext + ")");
    }
}

size_t Machine::maxLeftContext() const {
  size_t w = 0;
  for (const auto& ms : state) w = ms.leftContext.size() > w ? ms.leftContext.size() : w;
  return w;
}

std::string Machine::inputAlphabet(int flags) const {
  std::set<char> alph;
  for (const auto& ms : state)
    for (const auto& t : ms.trans)
      if (t.in && ((((t.in == kEOF) || (t.in == kSOF)) && (flags & kSEOFInput)) ||
                   (isControl(t.in) && (flags & kControlInput)) || (t.in == kFlush && (flags & kFlushInput)) ||
                   (isRelaxed(t.in) && (flags & kRelaxedInput)) || (isStrict(t.in) && (flags & kStrictInput))))
        alph.insert(t.in);
  return std::string(alph.begin(), alph.end());
}

std::string Machine::outputAlphabet() const {
  std::set<char> alph;
  for (const auto& ms : state)
    for (const auto& t : ms.trans)
      if (t.out) alph.insert(t.out);
  return std::string(alph.begin(), alph.end());
}

// Order of the states such that every usable transition that emits nothing runs forwards (the first sweep of a
// lattice column visits the states in this order, so that what such a transition carries is final when it is
// read; reference behaviour: Machine::decoderToposort, trans.cpp:604-634).  Built on a compressed adjacency of
// the non-emitting usable transitions: states whose last incoming one has been placed queue up first in, first out.
std::vector<uint32_t> Machine::decoderToposort(const std::string& inAlph) const {
  const size_t n = state.size();
  auto silent = [&](const MachineTransition& t) { return !t.out && (!t.in || inAlph.find(t.in) != std::string::npos); };
  std::vector<uint32_t> start(n + 1, 0), waitingFor(n, 0);
  for (size_t s = 0; s < n; ++s) {
    for (const auto& t : state[s].trans)
      if (silent(t)) { ++start[s + 1]; ++waitingFor[t.dest]; }
  }
  for (size_t s = 0; s < n; ++s) start[s + 1] += start[s];
  std::vector<uint32_t> target(start[n]);
  {
    std::vector<uint32_t> at(start.begin(), start.end() - 1);
    for (size_t s = 0; s < n; ++s)
      for (const auto& t : state[s].trans)
        if (silent(t)) target[at[s]++] = t.dest;
  }
  std::vector<uint32_t> order;
  order.reserve(n);
  for (size_t s = 0; s < n; ++s)
    if (waitingFor[s] == 0) order.push_back((uint32_t)s);
  for (size_t head = 0; head < order.size(); ++head) {     // `order` is its own queue
    const uint32_t u = order[head];
    for (uint32_t e = start[u]; e < start[u + 1]; ++e)
      if (--waitingFor[target[e]] == 0) order.push_back(target[e]);
  }
  if (order.size() < n) throw std::domain_error("Transducer is cyclic, can't toposort");
  return order;
}

namespace {

// how a state leaves: only by reading input (it waits), only without reading (it runs), both (mixed), not at all (final)
enum class Exit { kFinal, kWaits, kRuns, kMixed };
Exit exitOf(const MachineState& s) {
  bool reading = false, free = false;
  for (const auto& t : s.trans) (t.in ? reading : free) = true;
  return reading ? (free ? Exit::kMixed : Exit::kWaits) : (free ? Exit::kRuns : Exit::kFinal);
}

}  // namespace

bool Machine::isWaitingMachine() const {
  for (const auto& ms : state)
    if (exitOf(ms) == Exit::kMixed) return false;
  return true;
}

// Every state either waits or runs: a mixed state X is cut into "X;n", which keeps the transitions that read
// nothing and gains one more to "X;w" directly behind it, which keeps the reading ones.  (A final state is cut the
// same way -- the saved machines of the reference have it so, and its file format is the contract: trans.cpp:636-670.)
Machine Machine::waitingMachine() const {
  auto cut = [](const MachineState& ms) { const Exit e = exitOf(ms); return e == Exit::kMixed || e == Exit::kFinal; };
  std::vector<uint32_t> placeOf(nStates());
  uint32_t place = 0;
  for (uint32_t s = 0; s < nStates(); ++s) {
    placeOf[s] = place;
    place += cut(state[s]) ? 2u : 1u;
  }
  Machine wm;
  wm.state.reserve(place);
  for (uint32_t s = 0; s < nStates(); ++s) {
    const MachineState& ms = state[s];
    if (!cut(ms)) {
      wm.state.push_back(ms);
      for (auto& t : wm.state.back().trans) t.dest = placeOf[t.dest];
      continue;
    }
    MachineState runs, waits;
    runs.name = ms.name + ";n";
    waits.name = ms.name + ";w";
    runs.leftContext = waits.leftContext = ms.leftContext;
    runs.rightContext = waits.rightContext = ms.rightContext;
    for (MachineTransition t : ms.trans) {
      t.dest = placeOf[t.dest];
      (t.in ? waits : runs).trans.push_back(t);
    }
    MachineTransition handOver;
    handOver.dest = placeOf[s] + 1;
    runs.trans.push_back(handOver);
    wm.state.push_back(std::move(runs));
    wm.state.push_back(std::move(waits));
  }
  return wm;
}

// first * second: what `first` writes is what `second` reads.  A state of the product is a pair (i, j); while j
// waits, i moves (and each symbol it writes takes j along), otherwise j runs on its own.  Only the pairs that the
// start pair can reach are ever made (a worklist from (0, 0)); of those, the ones from which the final pair cannot
// be reached are dropped, a pair whose only way on is one transition that neither reads nor writes is folded into
// where that chain ends, and the survivors are numbered by (i, j) -- the numbering, names and transition order of
// the reference's product (trans.cpp:505-602), so that a saved composite is the same file.
Machine Machine::compose(const Machine& first, const Machine& origSecond) {
  const Machine second = origSecond.isWaitingMachine() ? origSecond : origSecond.waitingMachine();
  if (first.state.empty() || second.state.empty()) throw std::runtime_error("Machine has no states");
  if (!first.state.back().trans.empty() || !second.state.back().trans.empty())
    throw std::runtime_error("Last state must be end state");
  const uint64_t nB = second.nStates();
  struct Pair { uint64_t key; std::vector<MachineTransition> trans; std::vector<uint32_t> from; bool reachesEnd = false; };
  std::vector<Pair> pairs;                              // in order of discovery
  std::unordered_map<uint64_t, uint32_t> found;         // key i*nB + j -> index into pairs
  auto visit = [&](uint64_t i, uint64_t j) -> uint32_t {
    const uint64_t key = i * nB + j;
    auto it = found.find(key);
    if (it != found.end()) return it->second;
    found.emplace(key, (uint32_t)pairs.size());
    pairs.push_back(Pair{key, {}, {}, false});
    return (uint32_t)pairs.size() - 1;
  };
  visit(0, 0);
  for (uint32_t at = 0; at < pairs.size(); ++at) {      // `pairs` grows while it is walked
    const uint64_t i = pairs[at].key / nB, j = pairs[at].key % nB;
    const MachineState& a = first.state[i];
    const MachineState& b = second.state[j];
    std::vector<MachineTransition> out;
    auto add = [&](char in, char outSym, uint64_t di, uint64_t dj) {
      MachineTransition t;
      t.in = in; t.out = outSym; t.dest = visit(di, dj);    // index into pairs for now
      out.push_back(t);
    };
    const Exit eb = exitOf(b);
    if (eb == Exit::kWaits || eb == Exit::kFinal) {
      for (const auto& ta : a.trans) {
        if (!ta.out) { add(ta.in, 0, ta.dest, j); continue; }
        for (const auto& tb : b.trans)
          if (tb.in == ta.out) add(ta.in, tb.out, ta.dest, tb.dest);
      }
    } else {
      for (const auto& tb : b.trans) add(0, tb.out, i, tb.dest);
    }
    pairs[at].trans = std::move(out);
  }
  for (uint32_t s = 0; s < pairs.size(); ++s)
    for (const auto& t : pairs[s].trans) pairs[t.dest].from.push_back(s);
  // which pairs can still reach the final pair
  {
    const auto last = found.find((uint64_t)(first.nStates() - 1) * nB + (nB - 1));
    std::vector<uint32_t> todo;
    if (last != found.end()) { pairs[last->second].reachesEnd = true; todo.push_back(last->second); }
    while (!todo.empty()) {
      const uint32_t c = todo.back();
      todo.pop_back();
      for (uint32_t s : pairs[c].from)
        if (!pairs[s].reachesEnd) { pairs[s].reachesEnd = true; todo.push_back(s); }
    }
  }
  // fold the chains of lone empty transitions; number what is left by key
  auto passesOn = [&](uint32_t s) {
    const auto& ts = pairs[s].trans;
    return ts.size() == 1 && !ts.front().in && !ts.front().out;
  };
  std::vector<uint32_t> standsFor(pairs.size());
  for (uint32_t s = 0; s < pairs.size(); ++s) {
    uint32_t d = s;
    while (passesOn(d)) d = pairs[d].trans.front().dest;
    standsFor[s] = d;
  }
  std::vector<uint32_t> kept;
  for (uint32_t s = 0; s < pairs.size(); ++s)
    if (pairs[s].reachesEnd && standsFor[s] == s) kept.push_back(s);
  std::sort(kept.begin(), kept.end(), [&](uint32_t x, uint32_t y) { return pairs[x].key < pairs[y].key; });
  std::vector<uint32_t> number(pairs.size(), 0);
  for (uint32_t n = 0; n < kept.size(); ++n) number[kept[n]] = n;
  Machine out;
  out.state.reserve(kept.size());
  for (uint32_t s : kept) {
    const uint64_t i = pairs[s].key / nB, j = pairs[s].key % nB;
    MachineState ms;
    ms.name = "(" + first.state[i].name + "," + second.state[j].name + ")";
    ms.leftContext = second.state[j].leftContext;
    ms.rightContext = second.state[j].rightContext;
    ms.trans = pairs[s].trans;
    for (auto& t : ms.trans) t.dest = number[standsFor[t.dest]];
    out.state.push_back(std::move(ms));
  }
  return out;
}

}  // namespace dnas
