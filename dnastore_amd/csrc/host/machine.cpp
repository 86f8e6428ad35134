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

}  // namespace dnas
