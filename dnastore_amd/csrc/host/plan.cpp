#include "plan.hpp"

#include <algorithm>
#include <array>
#include <cstring>
#include <map>
#include <numeric>
#include <sstream>

namespace dnas {
namespace {

constexpr int kHeavy = 4;  // destinations with more in-edges than this are fed by pushes

struct Pull { int src; int sc; int base; };
struct Push { int dst; int sc; int base; int emit; };

}  // namespace

TierAPlan buildTierAPlan(const dnas_flat_model& fm) {
  TierAPlan p;
  const int N = fm.n_states, D = fm.max_dup_len, T = kTierAThreads;
  p.N = N; p.D = D; p.T = T;
  auto no = [&](const std::string& why) { p.ok = false; p.whyNot = why; return p; };
  if (D > 8) return no("more than 8 duplication lanes");
  // rows come in pairs: a thread's rows 2m and 2m+1 are neighbours in the HBM lattice, so that
  // the column goes out (and the S history comes back) as 16-byte accesses
  const int K = 2 * ((N + 2 * T - 1) / (2 * T));
  if (K > kTierAMaxRows) return no("more than " + std::to_string(kTierAMaxRows * T) + " states");
  p.K = K; p.NS = K * T;

  // score classes: 0.0 plus up to three distinct input-symbol log-probabilities
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
  // more than kHeavy in-edges, or more than one null in-edge (so that no row needs a second null pull)
  for (int j = 0; j < N; ++j) heavy[j] = (int)(emitIn[j].size() + nullIn[j].size()) > kHeavy || nullIn[j].size() > 1;
  for (int j = 0; j < N; ++j) {
    if (heavy[j]) hasCell[j] = 1;
    else for (const Pull& q : nullIn[j]) hasCell[q.src] = 1;
  }
  std::vector<int> cellOf(N, -1);
  int nCells = 0;
  for (int j = 0; j < N; ++j) if (hasCell[j]) ++nCells;
  // cells are numbered after the lanes are known (bank-aware); C is padded to a multiple of 32
  // so that SN[cell] and DN[cell] fall on the same LDS bank pair
  const int C = ((nCells + 2 + 31) / 32) * 32;   // + write-dummy (C-2) + read-dummy (C-1)
  const int readDummy = C - 1, writeDummy = C - 2;
  p.C = C;
  p.xDummy = p.NS + 2 * C;           // one extra double behind SN[], always -inf
  // X | DN | SN | -inf | score[4] sub[16] len[8] | red[T/64] | epoch, idle[T/64] (u32)
  p.ldsBytes = (size_t)(p.NS + 2 * C + 1 + 28 + T / 64 + (T / 64 + 2) / 2 + 1) * sizeof(double);
  if (p.ldsBytes > kTierALdsLimit) return no("LDS working set " + std::to_string(p.ldsBytes) + " B exceeds one CU");

  // pushes, filed under the source state
  std::vector<std::vector<Push>> pushes(N);
  for (int j = 0; j < N; ++j)
    if (heavy[j]) {
      for (const Pull& q : emitIn[j]) pushes[q.src].push_back({j, q.sc, q.base, 1});
      for (const Pull& q : nullIn[j]) pushes[q.src].push_back({j, q.sc, 0, 0});
    }

  // per-state entry counts: emit/null pulls by score class, pushes, publish
  typedef std::array<int, 10> Counts;   // e0..e3, n0..n3, ep, ec
  std::vector<Counts> cnt(N);
  for (int j = 0; j < N; ++j) {
    Counts c{};
    if (heavy[j]) {
      c[4] = 1;                           // heavy: one class-0 null pull of its own cell
    } else {
      for (const Pull& q : emitIn[j]) c[q.sc]++;
      for (const Pull& q : nullIn[j]) c[4 + q.sc]++;
    }
    c[8] = (int)pushes[j].size();
    c[9] = hasCell[j] ? 1 : 0;
    cnt[j] = c;
  }
  // ---- which state goes to which row ------------------------------------------------------
  // Two things decide what a sweep costs.  (1) Row shapes: a row pays, for all T threads, the
  // largest pull list any of its states has, so rows should hold states of one kind ("class":
  // which kinds of pull a state has).  (2) The ORDER in which rows are evaluated: a thread walks
  // its rows 0..K-1, so a value crosses an edge within the same sweep when the destination sits
  // in a LATER row than the source, and needs another sweep otherwise.  The in-column recursion
  // runs along the machine's chains (deletions follow the emit edges), so the number of sweeps
  // to the fixpoint is about (how far a value travels) x (share of backward edges on its way).
  //
  // The rows are therefore laid out as a "program" of shapes that follows the machine's cycle:
  // classes in the order that minimises backward edges between them (exhaustive search over the
  // eight largest), every class on rows of its own where the slack allows, shapes ascending
  // inside a class; then the states are dealt onto that program along a depth-first walk of the
  // machine, each state into the first row behind its parent's row whose shape admits it -- a
  // chain runs down the rows of one sweep instead of along one row.  The candidates (which class
  // boundaries are padded to a fresh row) are scored by  (LDS reads per sweep) x (sweeps, as
  // the largest number of backward edges on any walk of kWalk edges)  and the best one is kept.
  std::vector<int> rowOfState(N, -1);
  {
    std::map<Counts, int> classId;
    std::vector<Counts> classes;
    std::vector<int> cls(N), size;
    for (int j = 0; j < N; ++j) {
      Counts sig{};
      for (int q = 0; q < 8; ++q) sig[q] = cnt[j][q] > 0 ? 1 : 0;
      auto it = classId.find(sig);
      if (it == classId.end()) { it = classId.emplace(sig, (int)classes.size()).first; classes.push_back(sig); size.push_back(0); }
      cls[j] = it->second;
      ++size[cls[j]];
    }
    const int nC = (int)classes.size();
    std::vector<std::vector<long>> w(nC, std::vector<long>(nC, 0));   // edges class a -> class b
    std::vector<int> eSrc, eDst;
    for (int j = 0; j < N; ++j) {
      for (const Pull& q : emitIn[j]) { ++w[cls[q.src]][cls[j]]; eSrc.push_back(q.src); eDst.push_back(j); }
      for (const Pull& q : nullIn[j]) { ++w[cls[q.src]][cls[j]]; eSrc.push_back(q.src); eDst.push_back(j); }
    }
    std::vector<int> bySize(nC);
    std::iota(bySize.begin(), bySize.end(), 0);
    std::stable_sort(bySize.begin(), bySize.end(), [&](int a, int b) { return size[a] > size[b]; });
    const int nTop = std::min(nC, 8);
    std::vector<int> top(bySize.begin(), bySize.begin() + nTop), best;
    std::sort(top.begin(), top.end());
    long bestCost = -1;
    do {
      long cost = 0;
      for (int a = 0; a < nTop; ++a)
        for (int b = 0; b < a; ++b) cost += w[top[a]][top[b]];        // from a later class back to an earlier one
      if (bestCost < 0 || cost < bestCost) { bestCost = cost; best = top; }
    } while (std::next_permutation(top.begin(), top.end()));
    std::vector<int> rank(nC, 0);
    for (int i = 0; i < nTop; ++i) rank[best[i]] = i;
    {
      std::vector<int> rest(bySize.begin() + nTop, bySize.end());
      auto key = [&](const Counts& c) { return std::array<int, 10>{c[1], c[2], c[3], c[5], c[6], c[7], c[4], c[0], c[8], c[9]}; };
      std::stable_sort(rest.begin(), rest.end(), [&](int a, int b) { return key(classes[a]) > key(classes[b]); });
      for (size_t i = 0; i < rest.size(); ++i) rank[rest[i]] = nTop + (int)i;
    }
    // depth-first pre-order over all usable edges, from state 0, then from whatever is left
    std::vector<int> pre(N, -1), parent(N, -1), walk;
    walk.reserve(N);
    {
      std::vector<std::vector<int>> succ(N);
      for (size_t e = 0; e < eSrc.size(); ++e) succ[eSrc[e]].push_back(eDst[e]);
      int next = 0;
      std::vector<std::pair<int, size_t>> stack;
      for (int root = 0; root < N; ++root) {
        if (pre[root] >= 0) continue;
        pre[root] = next++;
        walk.push_back(root);
        stack.emplace_back(root, 0);
        while (!stack.empty()) {
          const int u = stack.back().first;
          if (stack.back().second < succ[u].size()) {
            const int v2 = succ[u][stack.back().second++];
            if (pre[v2] < 0) { pre[v2] = next++; parent[v2] = u; walk.push_back(v2); stack.emplace_back(v2, 0); }
          } else {
            stack.pop_back();
          }
        }
      }
    }
    // push / publish entries cost a register, not a read.  Where a class needs them often they
    // are allowed on all of its rows (a chain of such states must not be forced backwards);
    // where they are rare they sort to the end of their group and land on few rows.
    std::vector<std::array<char, 2>> common(nC, std::array<char, 2>{0, 0});
    for (int c = 0; c < nC; ++c)
      for (int q = 0; q < 2; ++q) {
        long have = 0;
        for (int j = 0; j < N; ++j) if (cls[j] == c && cnt[j][8 + q] > 0) ++have;
        common[c][q] = have * 4 >= size[c];
      }
    std::vector<int> order(N);
    std::iota(order.begin(), order.end(), 0);
    auto pulls = [&](int j) { int v = 0; for (int q = 0; q < 8; ++q) v += cnt[j][q]; return v; };
    auto rare = [&](int j, int q) { return common[cls[j]][q] ? 0 : cnt[j][8 + q]; };
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
      if (rank[cls[a]] != rank[cls[b]]) return rank[cls[a]] < rank[cls[b]];
      if (pulls(a) != pulls(b)) return pulls(a) < pulls(b);
      if (rare(a, 0) != rare(b, 0)) return rare(a, 0) < rare(b, 0);
      if (rare(a, 1) != rare(b, 1)) return rare(a, 1) < rare(b, 1);
      return pre[a] < pre[b];
    });
    std::vector<int> boundary;        // sorted indices where a new class starts
    for (int i = 1; i < N; ++i) if (cls[order[i]] != cls[order[i - 1]]) boundary.push_back(i);

    // the row program of one candidate: shapes of the row-major layout with the chosen gaps
    auto program = [&](const std::vector<char>& gap, std::vector<Counts>* caps, std::vector<int>* rows) -> bool {
      caps->assign(K, Counts{});
      int pos = 0, last = 0;
      size_t nb = 0;
      for (int i = 0; i < N; ++i) {
        if (nb < boundary.size() && boundary[nb] == i) {
          if (gap[nb] && pos % T) pos += T - pos % T;
          ++nb;
        }
        if (pos >= K * T) return false;
        const int j = order[i];
        Counts& r = (*caps)[pos / T];
        for (int q = 0; q < 10; ++q) r[q] = std::max(r[q], cnt[j][q]);
        for (int q = 0; q < 2; ++q) if (common[cls[j]][q]) r[8 + q] = 1;
        (*rows)[j] = last = pos / T;
        ++pos;
      }
      for (int k = last + 1; k < K; ++k) (*caps)[k] = (*caps)[last];   // spare rows repeat the last shape
      return true;
    };
    // deal the states onto a program along the depth-first walk
    std::map<Counts, int> typeId;
    std::vector<int> typeOf(N);
    std::vector<Counts> types;
    for (int j = 0; j < N; ++j) {
      auto it = typeId.find(cnt[j]);
      if (it == typeId.end()) { it = typeId.emplace(cnt[j], (int)types.size()).first; types.push_back(cnt[j]); }
      typeOf[j] = it->second;
    }
    auto deal = [&](const std::vector<Counts>& caps, std::vector<int>* rows) -> bool {
      std::vector<unsigned> admits(types.size(), 0);
      for (size_t t = 0; t < types.size(); ++t)
        for (int k = 0; k < K; ++k) {
          bool ok = true;
          for (int q = 0; q < 10; ++q) ok = ok && types[t][q] <= caps[k][q];
          if (ok) admits[t] |= 1u << k;
        }
      std::vector<std::vector<int>> members(K);
      unsigned freeRows = (1u << K) - 1u;
      rows->assign(N, -1);
      auto put = [&](int j, int k) {
        (*rows)[j] = k;
        members[k].push_back(j);
        if ((int)members[k].size() == T) freeRows &= ~(1u << k);
      };
      auto firstBehind = [&](unsigned avail, int j) {
        const int par = parent[j];
        const int start = (par >= 0 && (*rows)[par] >= 0) ? (*rows)[par] + 1 : 0;
        const unsigned fw = start < 32 ? avail & ~((1u << start) - 1u) : 0u;
        return __builtin_ctz(fw ? fw : avail);
      };
      for (int j : walk) {
        unsigned avail = admits[typeOf[j]] & freeRows;
        if (!avail) {
          // every row that admits j is full: move a more flexible resident of one of them elsewhere
          bool moved = false;
          for (int k = 0; k < K && !moved; ++k) {
            if (!(admits[typeOf[j]] >> k & 1u)) continue;
            for (size_t m = 0; m < members[k].size(); ++m) {
              const int i = members[k][m];
              const unsigned alt = admits[typeOf[i]] & freeRows;
              if (!alt) continue;
              members[k].erase(members[k].begin() + (long)m);
              freeRows |= 1u << k;
              put(i, firstBehind(alt, i));
              moved = true;
              break;
            }
          }
          if (!moved) return false;
          avail = admits[typeOf[j]] & freeRows;
        }
        put(j, firstBehind(avail, j));
      }
      return true;
    };
    // score: gathers per sweep (emit pulls + 2 per null pull) x estimated sweeps
    constexpr int kWalk = 30;
    auto score = [&](const std::vector<int>& rows, long* readsOut, int* backOut) -> double {
      std::vector<Counts> shape(K, Counts{});
      for (int j = 0; j < N; ++j)
        for (int q = 0; q < 10; ++q) shape[rows[j]][q] = std::max(shape[rows[j]][q], cnt[j][q]);
      long reads = 0, entries = 0;
      for (const Counts& r : shape) {
        reads += r[0] + r[1] + r[2] + r[3] + 2 * (r[4] + r[5] + r[6] + r[7]);
        for (int q = 0; q < 10; ++q) entries += r[q];
      }
      std::vector<int> f(N, 0), g(N);
      for (int h = 0; h < kWalk; ++h) {
        std::fill(g.begin(), g.end(), 0);
        for (size_t e = 0; e < eSrc.size(); ++e) {
          const int v = f[eSrc[e]] + (rows[eDst[e]] <= rows[eSrc[e]] ? 1 : 0);
          if (v > g[eDst[e]]) g[eDst[e]] = v;
        }
        f.swap(g);
      }
      int back = 0;
      for (int v : f) back = std::max(back, v);
      if (readsOut) *readsOut = reads;
      if (backOut) *backOut = back;
      if (entries > 56) return 1e30;
      return (double)(reads + 10) * (double)(back + 1);
    };
    const int nGap = (int)std::min<size_t>(boundary.size(), 10);   // boundaries between the first classes; later ones stay packed
    double bestScore = 1e31;
    std::vector<Counts> caps;
    std::vector<int> rowsMajor(N), rowsDealt;
    for (unsigned bits = 0; bits < (1u << nGap); ++bits) {
      std::vector<char> gap(boundary.size(), 0);
      for (int b = 0; b < nGap; ++b) gap[b] = (bits >> b) & 1u;
      if (!program(gap, &caps, &rowsMajor)) continue;
      const std::vector<int>* cand = &rowsMajor;
      if (deal(caps, &rowsDealt)) cand = &rowsDealt;
      long reads = 0;
      int back = 0;
      const double sc = score(*cand, &reads, &back);
      if (sc < bestScore) { bestScore = sc; rowOfState = *cand; p.sweepReads = (int)reads; p.backEdgesOnWalk = back; }
    }
    if (bestScore >= 1e30) return no("row shapes need more than 56 entry registers per thread");
  }

  // Lane placement inside each row.  Two goals.  (1) The waves of a work-group sweep without a
  // barrier and drift apart, so an edge into a later row is only CERTAIN to be relaxed in the
  // same sweep when source and destination belong to the same wave (a wave runs its rows in
  // order): a state goes to the wave of its "lead" -- the source in an earlier row it hangs on.
  // (2) LDS is 64 banks of 4 bytes and a ds_read_b64 is served in two 32-lane halves, so a
  // gather is conflict-free when the 32 source slots of a half fall on 32 different bank pairs,
  // i.e. have different (lane mod 32): best is the lead's own lane (the chain stays inside one
  // thread), then the other lane of that wave with the same residue, then any lane of the wave
  // whose half does not yet read that bank pair, then the same residue in another wave.
  std::vector<int> laneOf(N, -1);
  std::vector<std::vector<int>> rowMembers(K);                    // states of each row
  for (int j = 0; j < N; ++j) { laneOf[j] = (int)rowMembers[rowOfState[j]].size(); rowMembers[rowOfState[j]].push_back(j); }
  auto leadOf = [&](int j) {
    // first source in an earlier row (emit edges first), else the first source at all
    int any = -1;
    if (heavy[j]) return -1;
    for (const Pull& q : emitIn[j]) { if (q.src == j) continue; if (rowOfState[q.src] < rowOfState[j]) return q.src; if (any < 0) any = q.src; }
    for (const Pull& q : nullIn[j]) { if (q.src == j) continue; if (rowOfState[q.src] < rowOfState[j]) return q.src; if (any < 0) any = q.src; }
    return any;
  };
  for (int pass = 0; pass < 2; ++pass) {
    for (int k = 0; k < K; ++k) {
      const std::vector<int>& mem = rowMembers[k];
      if (mem.empty()) continue;          // padding row
      const int n = (int)mem.size();
      std::vector<char> lanesFree(T, 1);
      std::vector<int> newLane(n, -1), lead(n, -1);
      std::vector<std::array<unsigned char, 32>> bankUse(T / 32);   // per 32-lane half: gathers per bank pair
      for (auto& h : bankUse) h.fill(0);
      auto take = [&](int i, int t) { newLane[i] = t; lanesFree[t] = 0; if (lead[i] >= 0) ++bankUse[t / 32][laneOf[lead[i]] % 32]; };
      for (int i = 0; i < n; ++i) lead[i] = leadOf(mem[i]);
      for (int i = 0; i < n; ++i)          // the lead's own lane
        if (lead[i] >= 0 && lanesFree[laneOf[lead[i]]]) take(i, laneOf[lead[i]]);
      for (int i = 0; i < n; ++i) {        // same wave, same residue
        if (newLane[i] >= 0 || lead[i] < 0) continue;
        const int t = laneOf[lead[i]] ^ 32;
        if (lanesFree[t]) take(i, t);
      }
      for (int i = 0; i < n; ++i) {        // same wave, a half that does not read this bank pair yet
        if (newLane[i] >= 0 || lead[i] < 0) continue;
        const int w0 = laneOf[lead[i]] & ~63, r = laneOf[lead[i]] % 32;
        int bestT = -1, bestUse = 1 << 30;
        for (int t = w0; t < w0 + 64; ++t)
          if (lanesFree[t] && bankUse[t / 32][r] < bestUse) { bestUse = bankUse[t / 32][r]; bestT = t; }
        if (bestT >= 0) take(i, bestT);
      }
      for (int i = 0; i < n; ++i) {        // another wave, same residue
        if (newLane[i] >= 0 || lead[i] < 0) continue;
        const int r = laneOf[lead[i]] % 32;
        for (int t = r; t < T; t += 32)
          if (lanesFree[t]) { take(i, t); break; }
      }
      int cursor = 0;
      for (int i = 0; i < n; ++i) {
        if (newLane[i] >= 0) continue;
        while (!lanesFree[cursor]) ++cursor;
        take(i, cursor);
      }
      for (int i = 0; i < n; ++i) laneOf[mem[i]] = newLane[i];
    }
  }
  // two index spaces: LDS index row*T + lane (consecutive lanes -> consecutive bank pairs), and
  // the lattice slot (row/2)*2T + 2*lane + (row&1) used in HBM and by the traceback
  p.slotOf.assign(N, -1);
  p.stateOf.assign(p.NS, -1);       // by LDS index
  std::vector<int> ldsIdx(N, -1);
  for (int j = 0; j < N; ++j) {
    const int row = rowOfState[j], lane = laneOf[j];
    ldsIdx[j] = row * T + lane;
    p.stateOf[ldsIdx[j]] = j;
    p.slotOf[j] = (row >> 1) * 2 * T + 2 * lane + (row & 1);
  }
  // cell numbering: a null pull reads DN/SN[cell(src)]; give the cell the bank pair of its first
  // consumer's lane, buckets balanced so that C does not grow
  {
    std::vector<int> want(N, -1);
    for (int j = 0; j < N; ++j)
      if (!heavy[j])
        for (const Pull& q : nullIn[j])
          if (want[q.src] < 0) want[q.src] = laneOf[j] % 32;
    const int perBucket = (C - 2) / 32;                   // capacity of each residue class below the dummies
    std::vector<int> fill(32, 0);
    auto take = [&](int r) { const int id = r + 32 * fill[r]; ++fill[r]; return id; };
    std::vector<int> later;
    for (int j = 0; j < N; ++j) {
      if (!hasCell[j]) continue;
      const int r = want[j];
      if (r >= 0 && fill[r] < perBucket) cellOf[j] = take(r); else later.push_back(j);
    }
    int r = 0;
    for (int j : later) {
      while (fill[r] >= perBucket + (r < (C - 2) % 32 ? 1 : 0)) r = (r + 1) % 32;
      cellOf[j] = take(r);
    }
    for (int j = 0; j < N; ++j)
      if (hasCell[j] && cellOf[j] >= C - 2) return no("internal: cell numbering overflow");
  }

  p.rows.assign(K, RowShape{{0, 0, 0, 0}, {0, 0, 0, 0}, 0, 0});
  long real = 0;
  for (int j = 0; j < N; ++j) {
    const Counts& c = cnt[j];
    RowShape& r = p.rows[rowOfState[j]];
    for (int s = 0; s < 4; ++s) { r.e[s] = std::max(r.e[s], c[s]); r.n[s] = std::max(r.n[s], c[4 + s]); }
    r.ep = std::max(r.ep, c[8]); r.ec = std::max(r.ec, c[9]);
    for (int q = 0; q < 10; ++q) real += c[q];
  }
  auto rowEntries = [](const RowShape& r) { return r.e[0] + r.e[1] + r.e[2] + r.e[3] + r.n[0] + r.n[1] + r.n[2] + r.n[3] + r.ep + r.ec; };
  int nEnt = 0;
  for (const RowShape& r : p.rows) {
    nEnt += rowEntries(r);
    if (r.e[0] + r.e[1] + r.e[2] + r.e[3] > 16) return no("more than 16 emit pulls in one row");
  }
  p.nEntries = std::max(nEnt, 1);
  if (nEnt > 56) return no("row shape needs more than 56 entry registers per thread");
  p.fillRatio = nEnt ? (double)real / ((double)nEnt * T) : 1.0;

  // entries.  Pulls are bare LDS byte addresses (no decode in the sweep); pushes and
  // publishes carry flags:  [0:19) address of DN[cell] | [22:24) score class | [24:26) base |
  // [26] push: emit edge / publish: heavy cell | [27] publish: has a cell
  const unsigned dnBase = (unsigned)p.NS;   // DN[] starts right behind X[] (in doubles)
  auto packed = [](unsigned dblIdx, unsigned sc, unsigned base, unsigned flag, unsigned hasCellBit) {
    return ((dblIdx * 8u) & 0x7ffffu) | ((sc & 3u) << 22) | ((base & 3u) << 24) | ((flag & 1u) << 26) | ((hasCellBit & 1u) << 27);
  };
  p.entTab.assign((size_t)p.nEntries * T, 0);
  p.metaTab.assign((size_t)K * T, 0);
  int totalEmitSlots = 0;
  for (const RowShape& r : p.rows) totalEmitSlots += r.e[0] + r.e[1] + r.e[2] + r.e[3];
  p.nBaseWords = std::max(1, (totalEmitSlots + 15) / 16);
  p.baseTab.assign((size_t)p.nBaseWords * T, 0);
  int emitSlotBase = 0;
  int off = 0;
  for (int k = 0; k < K; ++k) {
    const RowShape& r = p.rows[k];
    for (int t = 0; t < T; ++t) {
      const int j = p.stateOf[(size_t)k * T + t];
      int m = off;
      int epos = emitSlotBase;
      for (int s = 0; s < 4; ++s) {
        std::vector<const Pull*> mine;            // this state's emit pulls of class s, reference edge order
        if (j >= 0 && !heavy[j])
          for (const Pull& q : emitIn[j]) if (q.sc == s) mine.push_back(&q);
        for (int e = 0; e < r.e[s]; ++e, ++m, ++epos) {
          unsigned v = (unsigned)p.xDummy * 8u;
          if (e < (int)mine.size()) {
            v = (unsigned)ldsIdx[mine[e]->src] * 8u;
            p.baseTab[(size_t)(epos / 16) * T + t] |= (unsigned)(mine[e]->base & 3) << (2 * (epos % 16));
          }
          p.entTab[(size_t)m * T + t] = v;
        }
      }
      for (int s = 0; s < 4; ++s) {
        std::vector<unsigned> mine;               // cells this state null-pulls with class s
        if (j >= 0) {
          if (heavy[j]) { if (s == 0) mine.push_back((unsigned)cellOf[j]); }
          else for (const Pull& q : nullIn[j]) if (q.sc == s) mine.push_back((unsigned)cellOf[q.src]);
        }
        for (int e = 0; e < r.n[s]; ++e, ++m) {
          unsigned v = (dnBase + (unsigned)readDummy) * 8u;
          if (e < (int)mine.size()) v = (dnBase + mine[e]) * 8u;
          p.entTab[(size_t)m * T + t] = v;
        }
      }
      for (int e = 0; e < r.ep; ++e, ++m) {
        unsigned v = packed(dnBase + (unsigned)writeDummy, 0, 0, 0, 0);
        if (j >= 0 && e < (int)pushes[j].size())
          v = packed(dnBase + (unsigned)cellOf[pushes[j][e].dst], pushes[j][e].sc, pushes[j][e].base, pushes[j][e].emit, 0);
        p.entTab[(size_t)m * T + t] = v;
      }
      for (int e = 0; e < r.ec; ++e, ++m) {
        unsigned v = packed(dnBase + (unsigned)readDummy, 0, 0, 0, 0);
        if (j >= 0 && hasCell[j]) v = packed(dnBase + (unsigned)cellOf[j], 0, 0, heavy[j] ? 1 : 0, 1);
        p.entTab[(size_t)m * T + t] = v;
      }
      unsigned meta = 0;
      if (j >= 0) {
        meta = fm.mdl[j] & 15u;
        for (int q = 0; q < fm.mdl[j] && q < 8; ++q) meta |= (unsigned)(fm.ctx[(size_t)j * D + q] & 3u) << (4 + 2 * q);
        if (j == 0) meta |= 0x80000000u;
        if (j == N - 1) meta |= 0x40000000u;
        meta |= 0x20000000u;   // slot holds a real state
      }
      p.metaTab[(size_t)k * T + t] = meta;
    }
    off += rowEntries(r);
    emitSlotBase += r.e[0] + r.e[1] + r.e[2] + r.e[3];
  }

  // LDS cost model of one sweep's gathers: a ds_read_b64 is served per 32-lane half in as many
  // cycles as the most loaded bank pair ((addr/8) mod 32) has distinct addresses
  {
    long cyc = 0, ideal = 0;
    int o2 = 0;
    for (int k = 0; k < K; ++k) {
      const RowShape& r = p.rows[k];
      const int pulls = r.e[0] + r.e[1] + r.e[2] + r.e[3] + r.n[0] + r.n[1] + r.n[2] + r.n[3];
      const int nE = r.e[0] + r.e[1] + r.e[2] + r.e[3];
      for (int e = 0; e < pulls; ++e) {
        for (int h = 0; h < T / 32; ++h) {
          int load[32] = {0};
          std::vector<unsigned> seen[32];
          for (int l = 0; l < 32; ++l) {
            const unsigned addr = p.entTab[(size_t)(o2 + e) * T + h * 32 + l];
            const unsigned b = (addr / 8) % 32;
            bool dup = false;
            for (unsigned a2 : seen[b]) if (a2 == addr) dup = true;
            if (!dup) { seen[b].push_back(addr); ++load[b]; }
          }
          int mx = 1;
          for (int b = 0; b < 32; ++b) mx = std::max(mx, load[b]);
          cyc += (e < nE ? 1 : 2) * mx;      // a null pull reads DN and SN
          ideal += (e < nE ? 1 : 2);
        }
      }
      o2 += pulls + r.ep + r.ec;
    }
    p.ldsCycles = cyc;
    p.ldsCyclesIdeal = ideal;
  }

  std::ostringstream rows, defs;
  for (int k = 0; k < K; ++k) {
    const RowShape& r = p.rows[k];
    if (k) rows << ",";
    rows << "{{" << r.e[0] << "," << r.e[1] << "," << r.e[2] << "," << r.e[3] << "},{" << r.n[0] << "," << r.n[1] << ","
         << r.n[2] << "," << r.n[3] << "}," << r.ep << "," << r.ec << "}";
  }
  defs << "-DDNAS_BASEWORDS=" << p.nBaseWords << "\n-DDNAS_T=" << T << "\n-DDNAS_K=" << K << "\n-DDNAS_D=" << D << "\n-DDNAS_NS=" << p.NS << "\n-DDNAS_C=" << C
       << "\n-DDNAS_ROWS=" << rows.str();
  p.defines = defs.str();
  p.key = "T" + std::to_string(T) + "K" + std::to_string(K) + "D" + std::to_string(D) + "NS" + std::to_string(p.NS) + "C" +
          std::to_string(C) + "R" + rows.str();
  p.ok = true;
  return p;
}

}  // namespace dnas
