#include "machine.hpp"

#include <deque>
#include <fstream>
#include <ostream>
#include <set>
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

std::vector<uint32_t> Machine::decoderToposort(const std::string& inAlph) const {
  const size_t n = state.size();
  std::vector<int> nParents(n, 0);
  std::vector<std::vector<uint32_t>> children(n);
  long edges = 0;
  for (size_t s = 0; s < n; ++s)
    for (const auto& t : state[s].trans)
      if (!t.out && (!t.in || inAlph.find(t.in) != std::string::npos)) {
        ++nParents[t.dest];
        ++edges;
        children[s].push_back(t.dest);
      }
  std::deque<uint32_t> ready;
  for (size_t s = 0; s < n; ++s)
    if (!nParents[s]) ready.push_back((uint32_t)s);
  std::vector<uint32_t> order;
  order.reserve(n);
  while (!ready.empty()) {
    const uint32_t u = ready.front();
    ready.pop_front();
    order.push_back(u);
    for (uint32_t c : children[u]) {
      --edges;
      if (--nParents[c] == 0) ready.push_back(c);
    }
  }
  if (edges > 0) throw std::domain_error("Transducer is cyclic, can't toposort");
  return order;
}

namespace {

bool hasInputEdge(const MachineState& s) {
  for (const auto& t : s.trans) if (t.in) return true;
  return false;
}
bool hasFreeEdge(const MachineState& s) {
  for (const auto& t : s.trans) if (!t.in) return true;
  return false;
}
// a state either waits for input on every edge, or moves without input on every edge, or is final
bool isWait(const MachineState& s) { return hasInputEdge(s) && !hasFreeEdge(s); }
bool isNonWait(const MachineState& s) { return !hasInputEdge(s) && hasFreeEdge(s); }

}  // namespace

bool Machine::isWaitingMachine() const {
  for (const auto& ms : state)
    if (!isWait(ms) && !isNonWait(ms) && !ms.trans.empty()) return false;
  return true;
}

Machine Machine::waitingMachine() const {
  // A mixed state X (input edges and free edges) becomes "X;n" holding the free edges plus a null
  // edge to a new state "X;w" (appended at the end) that holds the input edges.
  std::vector<MachineState> pool(state);
  std::vector<uint32_t> oldToNew(nStates()), order;
  for (uint32_t s = 0; s < nStates(); ++s) {
    const MachineState& ms = state[s];
    oldToNew[s] = (uint32_t)order.size();
    order.push_back(s);
    if (!isWait(ms) && !isNonWait(ms)) {   // mixed (the reference also splits end states, harmlessly)
      MachineState freePart, waitPart;
      freePart.name = ms.name + ";n";
      waitPart.name = ms.name + ";w";
      freePart.leftContext = waitPart.leftContext = ms.leftContext;
      freePart.rightContext = waitPart.rightContext = ms.rightContext;
      for (const auto& t : ms.trans) (t.in ? waitPart : freePart).trans.push_back(t);
      MachineTransition hop;
      hop.dest = (uint32_t)pool.size();
      freePart.trans.push_back(hop);
      oldToNew.push_back((uint32_t)order.size());
      order.push_back((uint32_t)pool.size());
      pool[s] = std::move(freePart);
      pool.push_back(std::move(waitPart));
    }
  }
  Machine wm;
  for (uint32_t idx : order) {
    MachineState ms = pool[idx];
    for (auto& t : ms.trans) t.dest = oldToNew[t.dest];
    wm.state.push_back(std::move(ms));
  }
  return wm;
}

Machine Machine::compose(const Machine& first, const Machine& origSecond) {
  const Machine second = origSecond.isWaitingMachine() ? origSecond : origSecond.waitingMachine();
  if (first.state.empty() || second.state.empty()) throw std::runtime_error("Machine has no states");
  if (!first.state.back().trans.empty() || !second.state.back().trans.empty())
    throw std::runtime_error("Last state must be end state");
  const size_t nB = second.nStates(), nAll = first.nStates() * nB;
  auto id = [&](size_t i, size_t j) { return i * nB + j; };
  std::vector<MachineState> prod(nAll);
  for (size_t i = 0; i < first.nStates(); ++i)
    for (size_t j = 0; j < nB; ++j) {
      const MachineState& a = first.state[i];
      const MachineState& b = second.state[j];
      MachineState& ms = prod[id(i, j)];
      ms.name = "(" + a.name + "," + b.name + ")";
      ms.leftContext = b.leftContext;
      ms.rightContext = b.rightContext;
      if (isWait(b) || b.trans.empty()) {
        // B waits: A moves; what A emits must be consumed by an input edge of B
        for (const auto& ta : a.trans) {
          if (!ta.out) {
            MachineTransition t; t.in = ta.in; t.out = 0; t.dest = (uint32_t)id(ta.dest, j);
            ms.trans.push_back(t);
          } else {
            for (const auto& tb : b.trans)
              if (ta.out == tb.in) {
                MachineTransition t; t.in = ta.in; t.out = tb.out; t.dest = (uint32_t)id(ta.dest, tb.dest);
                ms.trans.push_back(t);
              }
          }
        }
      } else {
        // B moves on its own
        for (const auto& tb : b.trans) {
          MachineTransition t; t.in = 0; t.out = tb.out; t.dest = (uint32_t)id(i, tb.dest);
          ms.trans.push_back(t);
        }
      }
    }
  // keep states reachable from the start that can still reach the end
  std::vector<char> fwd(nAll, 0), bwd(nAll, 0);
  std::deque<uint32_t> queue;
  queue.push_back((uint32_t)id(0, 0));
  fwd[queue.front()] = 1;
  while (!queue.empty()) {
    const uint32_t c = queue.front(); queue.pop_front();
    for (const auto& t : prod[c].trans) if (!fwd[t.dest]) { fwd[t.dest] = 1; queue.push_back(t.dest); }
  }
  std::vector<std::vector<uint32_t>> sources(nAll);
  for (uint32_t s = 0; s < nAll; ++s) for (const auto& t : prod[s].trans) sources[t.dest].push_back(s);
  queue.push_back((uint32_t)id(first.nStates() - 1, nB - 1));
  bwd[queue.front()] = 1;
  while (!queue.empty()) {
    const uint32_t c = queue.front(); queue.pop_front();
    for (uint32_t s : sources[c]) if (!bwd[s]) { bwd[s] = 1; queue.push_back(s); }
  }
  auto live = [&](uint32_t s) { return fwd[s] && bwd[s]; };
  // a state whose only move is a null transition is merged into where that chain ends
  std::vector<int64_t> merged(nAll, -1);
  for (uint32_t s = 0; s < nAll; ++s)
    if (live(s)) {
      uint32_t d = s;
      while (prod[d].trans.size() == 1 && !prod[d].trans.front().in && !prod[d].trans.front().out) d = prod[d].trans.front().dest;
      if (d != s) merged[s] = d;
    }
  std::vector<uint32_t> renum(nAll, 0);
  uint32_t kept = 0;
  for (uint32_t s = 0; s < nAll; ++s) if (live(s) && merged[s] < 0) renum[s] = kept++;
  for (uint32_t s = 0; s < nAll; ++s) if (live(s) && merged[s] >= 0) renum[s] = renum[(size_t)merged[s]];
  Machine out;
  out.state.reserve(kept);
  for (uint32_t s = 0; s < nAll; ++s)
    if (live(s) && merged[s] < 0) {
      MachineState ms = prod[s];
      for (auto& t : ms.trans) t.dest = renum[t.dest];
      out.state.push_back(std::move(ms));
    }
  return out;
}

}  // namespace dnas
