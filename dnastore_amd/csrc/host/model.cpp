#include "model.hpp"

#include <cmath>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>

#include "json.hpp"

namespace dnas {

// ---------------------------------------------------------------- MutatorParams

MutatorParams MutatorParams::fromFlags(double subProb, double ivRatio, double dupProb, double delOpen, double delExt,
                                       bool global, int length) {
  MutatorParams m;
  const int n = length / 2;  // initMaxDupLen(len / 2), dnastore.cpp:120; mutator.cpp:51-54
  m.pLen.assign(n > 0 ? n : 0, n > 0 ? 1. / (double)n : 0.);
  m.pTanDup = dupProb;
  m.pDelOpen = delOpen;
  m.pDelExtend = delExt;
  m.pTransition = subProb * ivRatio / (1 + ivRatio);
  m.pTransversion = subProb / (1 + ivRatio);
  m.local = !global;
  return m;
}

MutatorParams MutatorParams::fromJSON(const std::string& text) {
  const JsonValue j = parseJson(text);
  MutatorParams m;
  m.pDelOpen = j.number("pDelOpen");
  m.pDelExtend = j.number("pDelExtend");
  m.pTanDup = j.number("pTanDup");
  m.pTransition = j.number("pTransition");
  m.pTransversion = j.number("pTransversion");
  m.local = j.boolean("local");
  for (const JsonValue& v : j.array("pLen")) m.pLen.push_back(v.num);
  return m;
}

MutatorParams MutatorParams::fromFile(const std::string& path) {
  std::ifstream in(path);
  if (!in) throw std::runtime_error("File not found: " + path);
  std::stringstream ss;
  ss << in.rdbuf();
  return fromJSON(ss.str());
}

std::string MutatorParams::toJSON() const {
  // key order, spacing and the default 6-digit ostream precision of mutator.cpp:6-16
  std::ostringstream out;
  out << "{\n";
  out << " \"pDelOpen\": " << pDelOpen << ",\n";
  out << " \"pDelExtend\": " << pDelExtend << ",\n";
  out << " \"pTanDup\": " << pTanDup << ",\n";
  out << " \"pTransition\": " << pTransition << ",\n";
  out << " \"pTransversion\": " << pTransversion << ",\n";
  out << " \"pLen\": [ ";
  for (size_t i = 0; i < pLen.size(); ++i) out << (i ? ", " : "") << pLen[i];
  out << " ],\n";
  out << " \"local\": " << (local ? "true" : "false") << "\n";
  out << "}\n";
  return out.str();
}

void MutatorParams::toC(dnas_mutator_params* o) const {
  memset(o, 0, sizeof(*o));
  o->p_del_open = pDelOpen;
  o->p_del_extend = pDelExtend;
  o->p_tan_dup = pTanDup;
  o->p_transition = pTransition;
  o->p_transversion = pTransversion;
  if (pLen.size() > 32) throw std::runtime_error("pLen longer than 32 entries");
  o->n_len = (int32_t)pLen.size();
  o->local = local ? 1 : 0;
  for (size_t i = 0; i < pLen.size(); ++i) o->p_len[i] = pLen[i];
}

MutatorParams MutatorParams::fromC(const dnas_mutator_params& p) {
  MutatorParams m;
  m.pDelOpen = p.p_del_open;
  m.pDelExtend = p.p_del_extend;
  m.pTanDup = p.p_tan_dup;
  m.pTransition = p.p_transition;
  m.pTransversion = p.p_transversion;
  m.local = p.local != 0;
  m.pLen.assign(p.p_len, p.p_len + (p.n_len < 0 ? 0 : (p.n_len > 32 ? 32 : p.n_len)));
  return m;
}

// ---------------------------------------------------------------- FlatModel

static bool isTransition(int x, int y) { return x != y && (x & 1) == (y & 1); }  // kmer.h:85-87

void FlatModel::bind() {
  view.ein_ptr = einPtr.data(); view.ein_src = einSrc.data(); view.ein_score = einScore.data();
  view.ein_in = einIn.data(); view.ein_base = einBase.data();
  view.nin_ptr = ninPtr.data(); view.nin_src = ninSrc.data(); view.nin_score = ninScore.data();
  view.nin_in = ninIn.data();
  view.eout_ptr = eoutPtr.data(); view.eout_dst = eoutDst.data(); view.eout_score = eoutScore.data();
  view.nout_ptr = noutPtr.data(); view.nout_dst = noutDst.data(); view.nout_score = noutScore.data();
  view.mdl = mdl.data(); view.ctx = ctx.data(); view.topo = topo.data(); view.len = len.data();
}

FlatModel FlatModel::build(const Machine& machine, const MutatorParams& params) {
  FlatModel f;
  const size_t N = machine.nStates();
  if (N == 0) throw std::runtime_error("Machine has no states");
  if (params.pLen.size() > 32) throw std::runtime_error("pLen longer than 32 entries");
  dnas_flat_model& v = f.view;
  v.n_states = (int32_t)N;
  v.n_len = (int32_t)params.maxDupLen();
  v.local = params.local ? 1 : 0;

  machine.verifyContexts();  // viterbi.cpp:26
  for (char c : machine.outputAlphabet())
    if (charToBase(c) < 0) throw std::runtime_error("Not a DNA-outputting machine");  // viterbi.cpp:27-28

  // InputModel over inputAlphabet(Relaxed|Control|SEOF): weight 1 for data symbols,
  // 4^(-4P) for control symbols, normalised (viterbi.cpp:309-310, 6-14)
  const std::string alph = machine.inputAlphabet(kRelaxedInput | kControlInput | kSEOFInput);
  if (alph.size() >= sizeof(v.alphabet)) throw std::runtime_error("input alphabet too large");
  memset(v.alphabet, 0, sizeof(v.alphabet));
  memcpy(v.alphabet, alph.data(), alph.size());
  const double controlWeight = std::pow(4., -(double)(4 * params.maxDupLen()));
  std::map<char, double> symProb;
  double norm = 0;
  for (char c : alph) norm += (symProb[c] = Machine::isControl(c) ? controlWeight : 1.);
  for (auto& sp : symProb) sp.second /= norm;
  for (int i = 0; i < 128; ++i) v.sym_logp[i] = 0;
  for (const auto& sp : symProb) v.sym_logp[(int)sp.first & 127] = std::log(sp.second);

  // maxDupLen (viterbi.cpp:63): raw left-context width (wildcards included) vs. pLen.size()
  const size_t D = std::min(machine.maxLeftContext(), params.maxDupLen());
  v.max_dup_len = (int32_t)D;

  // MutatorScores (mutator.cpp:56-75)
  v.del_open = std::log(params.pDelOpen);
  v.tan_dup = std::log(params.pTanDup);
  v.no_gap = std::log(params.pNoGap());
  v.del_extend = std::log(params.pDelExtend);
  v.del_end = std::log(params.pDelEnd());
  const double nullScore = std::log(1. / 4.);
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      v.sub[i * 4 + j] = (i == j ? std::log(params.pMatch())
                                 : (isTransition(i, j) ? std::log(params.pTransition) : std::log(params.pTransversion / 2))) -
                         nullScore;
  f.len.resize(params.maxDupLen() ? params.maxDupLen() : 1, 0.);
  for (size_t l = 0; l < params.maxDupLen(); ++l) f.len[l] = std::log(params.pLen[l]);

  // StateScores: stripped left contexts (viterbi.cpp:33-36), mdl (viterbi.h:104), ctx (viterbi.h:105)
  f.mdl.assign(N, 0);
  f.ctx.assign(N * (D ? D : 1), 0);
  for (size_t s = 0; s < N; ++s) {
    std::vector<uint8_t> lc;
    for (char c : machine.state[s].leftContext)
      if (c != kWildContext) {
        const int b = charToBase(c);
        if (b < 0) throw std::runtime_error(std::string(1, c) + " is not a nucleotide character");
        lc.push_back((uint8_t)b);
      }
    const size_t mdl = std::min(D, lc.size());
    f.mdl[s] = (uint8_t)mdl;
    for (size_t k = 0; k < mdl; ++k) f.ctx[s * D + k] = lc[lc.size() - 1 - k];
  }

  // usable transitions (viterbi.cpp:38): no input, EOF, or an input the model knows
  auto usable = [&](const MachineTransition& t) { return !t.in || t.in == kEOF || symProb.count(t.in); };
  f.einPtr.assign(N + 1, 0); f.ninPtr.assign(N + 1, 0); f.eoutPtr.assign(N + 1, 0); f.noutPtr.assign(N + 1, 0);
  for (size_t s = 0; s < N; ++s)
    for (const auto& t : machine.state[s].trans)
      if (usable(t)) {
        if (t.out) { ++f.einPtr[t.dest + 1]; ++f.eoutPtr[s + 1]; }
        else { ++f.ninPtr[t.dest + 1]; ++f.noutPtr[s + 1]; }
      }
  for (size_t s = 0; s < N; ++s) {
    f.einPtr[s + 1] += f.einPtr[s]; f.ninPtr[s + 1] += f.ninPtr[s];
    f.eoutPtr[s + 1] += f.eoutPtr[s]; f.noutPtr[s + 1] += f.noutPtr[s];
  }
  const size_t nE = f.einPtr[N], nN = f.ninPtr[N];
  v.n_emit = (int32_t)nE; v.n_null = (int32_t)nN;
  f.einSrc.assign(nE + 1, 0); f.einScore.assign(nE + 1, 0); f.einIn.assign(nE + 1, 0); f.einBase.assign(nE + 1, 0);
  f.ninSrc.assign(nN + 1, 0); f.ninScore.assign(nN + 1, 0); f.ninIn.assign(nN + 1, 0);
  f.eoutDst.assign(nE + 1, 0); f.eoutScore.assign(nE + 1, 0);
  f.noutDst.assign(nN + 1, 0); f.noutScore.assign(nN + 1, 0);
  std::vector<int32_t> fe(N, 0), fn(N, 0), ge(N, 0), gn(N, 0);
  // ascending source state then transition order == the push_back order of viterbi.cpp:49-56
  for (size_t s = 0; s < N; ++s)
    for (const auto& t : machine.state[s].trans)
      if (usable(t)) {
        const double score = symProb.count(t.in) ? std::log(symProb.at(t.in)) : 0;  // viterbi.cpp:41
        if (!t.out) {
          const size_t i = f.ninPtr[t.dest] + fn[t.dest]++;
          f.ninSrc[i] = (int32_t)s; f.ninScore[i] = score; f.ninIn[i] = (uint8_t)t.in;
          const size_t o = f.noutPtr[s] + gn[s]++;
          f.noutDst[o] = (int32_t)t.dest; f.noutScore[o] = score;
        } else {
          const size_t i = f.einPtr[t.dest] + fe[t.dest]++;
          f.einSrc[i] = (int32_t)s; f.einScore[i] = score; f.einIn[i] = (uint8_t)t.in;
          f.einBase[i] = (uint8_t)charToBase(t.out);
          const size_t o = f.eoutPtr[s] + ge[s]++;
          f.eoutDst[o] = (int32_t)t.dest; f.eoutScore[o] = score;
        }
      }

  const std::vector<uint32_t> order = machine.decoderToposort(alph);  // viterbi.cpp:81
  f.topo.assign(order.begin(), order.end());
  f.bind();
  return f;
}

}  // namespace dnas
