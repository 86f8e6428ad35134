#include "plan.hpp"

#include <algorithm>
#include <cstring>
#include <functional>
#include <numeric>
#include <sstream>

namespace dnas {
namespace {

constexpr int kHeavy = 4;  // destinations with more in-edges than this are fed by pushes

struct Pull { int src; int sc; int base; };
struct Push { int dstCell; int sc; int base; int emit; };

}  // namespace

TierAPlan buildTierAPlan(const dnas_flat_model& fm) {
  TierAPlan p;
  const int N = fm.n_states, D = fm.max_dup_len, T = kTierAThreads;
  p.N = N; p.D = D; p.T = T;
  auto no = [&](const std::string& why) { p.ok = false; p.whyNot = why; return p; };
  if (D > 8) return no("more than 8 duplication lanes");
  const int K = (N + T - 1) / T;
  if (K > kTierAMaxRows) return no("more than " + std::to_string(kTierAMaxRows * T) + " states");
  p.K = K; p.NS = K * T;

  // score table: 0 plus up to three distinct input-symbol log-probabilities
  std::vector<double> scores{0.0};
  auto scoreIdx = [&](double s) -> int {
    for (size_t i = 0; i < scores.size(); ++i)
      if (memcmp(&scores[i], &s, sizeof s) == 0) return (int)i;
    scores.push_back(s);
    return (int)scores.size() - 1;
  };
  std::vector<std::vector<Pull>> emitIn(N), nullIn(N);
  for (int j = 0; j < N; ++j) {
    for (int e = fm.ein_ptr[j]; e < fm.ein_ptr[j + 1]; ++e)
      emitIn[j].push_back({fm.ein_src[e], scoreIdx(fm.ein_score[e]), fm.ein_base[e]});
    for (int e = fm.nin_ptr[j]; e < fm.nin_ptr[j + 1]; ++e)
      nullIn[j].push_back({fm.nin_src[e], scoreIdx(fm.nin_score[e]), 0});
  }
  if (scores.size() > 4) return no("more than three distinct edge scores");
  for (size_t i = 0; i < scores.size(); ++i) p.score[i] = scores[i];

  // heavy destinations and LDS cells
  std::vector<char> heavy(N, 0), hasCell(N, 0);
  for (int j = 0; j < N; ++j) heavy[j] = (int)(emitIn[j].size() + nullIn[j].size()) > kHeavy;
  for (int j = 0; j < N; ++j) {
    if (heavy[j]) hasCell[j] = 1;
    else for (const Pull& q : nullIn[j]) hasCell[q.src] = 1;
  }
  std::vector<int> cellOf(N, -1);
  int nCells = 0;
  for (int j = 0; j < N; ++j) if (hasCell[j]) cellOf[j] = nCells++;
  const int C = nCells + 2;          // + write-dummy (C-2) + read-dummy (C-1)
  const int readDummy = C - 1, writeDummy = C - 2;
  p.C = C;
  p.xDummy = p.NS + 2 * C;           // one extra double behind SN[], always -inf
  // X | DN | SN | -inf | score[4] sub[16] len[8] | red[T/64]
  p.ldsBytes = (size_t)(p.NS + 2 * C + 1 + 28 + T / 64) * sizeof(double);
  if (p.ldsBytes > kTierALdsLimit) return no("LDS working set " + std::to_string(p.ldsBytes) + " B exceeds one CU");

  // pushes, filed under the source state
  std::vector<std::vector<Push>> pushes(N);
  for (int j = 0; j < N; ++j)
    if (heavy[j]) {
      for (const Pull& q : emitIn[j]) pushes[q.src].push_back({cellOf[j], q.sc, q.base, 1});
      for (const Pull& q : nullIn[j]) pushes[q.src].push_back({cellOf[j], q.sc, 0, 0});
    }
  auto nEE = [&](int j) { return heavy[j] ? 0 : (int)emitIn[j].size(); };
  auto nEN = [&](int j) { return heavy[j] ? 1 : (int)nullIn[j].size(); };   // heavy: pull its own cell
  auto nEP = [&](int j) { return (int)pushes[j].size(); };
  auto nEC = [&](int j) { return hasCell[j] ? 1 : 0; };

  // chain depth: states with a single in-edge sit behind their predecessor, so that within a
  // sweep (rows run in order) a value travels down a whole chain
  std::vector<int> depth(N, 0);
  for (int it = 0; it < 64; ++it) {
    bool any = false;
    for (int j = 0; j < N; ++j) {
      if (emitIn[j].size() + nullIn[j].size() != 1) continue;
      const int src = emitIn[j].empty() ? nullIn[j][0].src : emitIn[j][0].src;
      const int d = std::min(depth[src] + 1, 63);
      if (d > depth[j] && src != j) { depth[j] = d; any = true; }
    }
    if (!any) break;
  }

  std::vector<int> order(N);
  std::iota(order.begin(), order.end(), 0);
  auto cost = [&](int j) { return nEE(j) + 2 * nEN(j) + 3 * nEP(j) + nEC(j); };
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
    const int ca = cost(a), cb = cost(b);
    if (ca != cb) return ca > cb;
    if (nEN(a) != nEN(b)) return nEN(a) > nEN(b);
    if (nEP(a) != nEP(b)) return nEP(a) > nEP(b);
    if (nEC(a) != nEC(b)) return nEC(a) > nEC(b);
    return depth[a] < depth[b];
  });
  p.slotOf.assign(N, -1);
  p.stateOf.assign(p.NS, -1);
  for (int i = 0; i < N; ++i) { p.slotOf[order[i]] = i; p.stateOf[i] = order[i]; }

  p.rows.assign(K, RowShape{0, 0, 0, 0});
  long real = 0;
  for (int i = 0; i < N; ++i) {
    const int j = order[i];
    RowShape& r = p.rows[i / T];
    r.ee = std::max(r.ee, nEE(j)); r.en = std::max(r.en, nEN(j));
    r.ep = std::max(r.ep, nEP(j)); r.ec = std::max(r.ec, nEC(j));
    real += nEE(j) + nEN(j) + nEP(j) + nEC(j);
  }
  int nEnt = 0;
  for (const RowShape& r : p.rows) nEnt += r.ee + r.en + r.ep + r.ec;
  p.nEntries = std::max(nEnt, 1);
  if (nEnt > 48) return no("row shape needs more than 48 entry registers per thread");
  p.fillRatio = nEnt ? (double)real / ((double)nEnt * T) : 1.0;

  // entries
  // [0:19) LDS byte address | [19:24) score index << 3 | [24:26) base | [26] flag | [27] has cell
  const unsigned dnBase = (unsigned)p.NS;   // DN[] starts right behind X[] (in doubles)
  auto ent = [](unsigned dblIdx, unsigned sc, unsigned base, unsigned flag, unsigned hasCell) {
    return ((dblIdx * 8u) & 0x7ffffu) | ((sc & 3u) << 22) | ((base & 3u) << 24) | ((flag & 1u) << 26) | ((hasCell & 1u) << 27);
  };
  p.entTab.assign((size_t)p.nEntries * T, 0);
  p.metaTab.assign((size_t)K * T, 0);
  int off = 0;
  for (int k = 0; k < K; ++k) {
    const RowShape& r = p.rows[k];
    for (int t = 0; t < T; ++t) {
      const int j = p.stateOf[(size_t)k * T + t];
      int m = off;
      for (int e = 0; e < r.ee; ++e, ++m) {
        unsigned v = ent((unsigned)p.xDummy, 0, 0, 0, 0);
        if (j >= 0 && e < nEE(j)) v = ent((unsigned)p.slotOf[emitIn[j][e].src], emitIn[j][e].sc, emitIn[j][e].base, 0, 0);
        p.entTab[(size_t)m * T + t] = v;
      }
      for (int e = 0; e < r.en; ++e, ++m) {
        unsigned v = ent(dnBase + (unsigned)readDummy, 0, 0, 0, 0);
        if (j >= 0 && e < nEN(j))
          v = heavy[j] ? ent(dnBase + (unsigned)cellOf[j], 0, 0, 0, 0)
                       : ent(dnBase + (unsigned)cellOf[nullIn[j][e].src], nullIn[j][e].sc, 0, 0, 0);
        p.entTab[(size_t)m * T + t] = v;
      }
      for (int e = 0; e < r.ep; ++e, ++m) {
        unsigned v = ent(dnBase + (unsigned)writeDummy, 0, 0, 0, 0);
        if (j >= 0 && e < nEP(j))
          v = ent(dnBase + (unsigned)pushes[j][e].dstCell, pushes[j][e].sc, pushes[j][e].base, pushes[j][e].emit, 0);
        p.entTab[(size_t)m * T + t] = v;
      }
      for (int e = 0; e < r.ec; ++e, ++m) {
        unsigned v = ent(dnBase + (unsigned)readDummy, 0, 0, 0, 0);
        if (j >= 0 && hasCell[j]) v = ent(dnBase + (unsigned)cellOf[j], 0, 0, heavy[j] ? 1 : 0, 1);
        p.entTab[(size_t)m * T + t] = v;
      }
      unsigned meta = 0;
      if (j >= 0) {
        meta = fm.mdl[j] & 15u;
        for (int q = 0; q < fm.mdl[j] && q < 8; ++q) meta |= (unsigned)(fm.ctx[(size_t)j * D + q] & 3u) << (4 + 2 * q);
        if (j == 0) meta |= 0x80000000u;
        if (j == N - 1) meta |= 0x40000000u;
      }
      p.metaTab[(size_t)k * T + t] = meta;
    }
    off += r.ee + r.en + r.ep + r.ec;
  }

  std::ostringstream rows, defs;
  for (int k = 0; k < K; ++k) {
    if (k) rows << ",";
    rows << "{" << p.rows[k].ee << "," << p.rows[k].en << "," << p.rows[k].ep << "," << p.rows[k].ec << "}";
  }
  defs << "-DDNAS_T=" << T << "\n-DDNAS_K=" << K << "\n-DDNAS_D=" << D << "\n-DDNAS_NS=" << p.NS << "\n-DDNAS_C=" << C
       << "\n-DDNAS_ROWS=" << rows.str();
  p.defines = defs.str();
  p.key = "T" + std::to_string(T) + "K" + std::to_string(K) + "D" + std::to_string(D) + "NS" + std::to_string(p.NS) + "C" +
          std::to_string(C) + "R" + rows.str();
  p.ok = true;
  return p;
}

}  // namespace dnas
