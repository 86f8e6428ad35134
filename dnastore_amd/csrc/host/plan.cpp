#include "plan.hpp"

#include <algorithm>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <numeric>
#include <sstream>

namespace dnas {
namespace {

constexpr int kMaxEntries = 40;   // entry registers per thread the kernel can afford (at 1024 threads: 128 registers each)
constexpr int kWalk = 30;         // length of the walks the sweep estimate looks at

struct Edge { int src, dst, sc, base, isNull; };
// out-edges, has null in-edges, not plain (some out-edge is a null edge or carries a score), score class of the
// out-edges (4: they differ, 7: no out-edge) -- in a row's caps: the mask of classes it admits --, has an out-edge
// into another member of the cluster (its entries are decoded per lane: kept to a few rows at the end of the program)
typedef std::array<int, 5> Type;

// Which member of the cluster owns which state.  The in-column recursion runs along the machine's chains, and every
// edge between two members costs a trip through the exchange buffer (microseconds, against nanoseconds inside a CU):
// the states are cut into G runs of the depth-first walk (a run is a set of subtrees: the composites here are long
// chains with a branch every few states, so few edges leave a run), then single states move to the member most of
// their neighbours are in while that lowers the cut and the member has room.
std::vector<int> partitionStates(int N, const std::vector<Edge>& edges, const std::vector<int>& walk, int G, int cap) {
  std::vector<int> part((size_t)N, 0);
  for (int i = 0; i < N; ++i) part[(size_t)walk[(size_t)i]] = (int)((long)i * G / N);
  std::vector<int> size((size_t)G, 0);
  for (int j = 0; j < N; ++j) ++size[(size_t)part[(size_t)j]];
  std::vector<std::vector<int>> nb((size_t)N);
  for (const Edge& e : edges)
    if (e.src != e.dst) { nb[(size_t)e.src].push_back(e.dst); nb[(size_t)e.dst].push_back(e.src); }
  std::vector<int> cnt((size_t)G);
  for (int pass = 0; pass < 4; ++pass) {
    int moved = 0;
    for (int i = 0; i < N; ++i) {
      const int j = walk[(size_t)i];
      if (nb[(size_t)j].empty()) continue;
      std::fill(cnt.begin(), cnt.end(), 0);
      for (int v : nb[(size_t)j]) ++cnt[(size_t)part[(size_t)v]];
      const int cur = part[(size_t)j];
      int best = cur;
      for (int g = 0; g < G; ++g)
        if (g != cur && size[(size_t)g] < cap && cnt[(size_t)g] > cnt[(size_t)best]) best = g;
      if (best != cur) { part[(size_t)j] = best; --size[(size_t)cur]; ++size[(size_t)best]; ++moved; }
    }
    if (moved < N / 2000 + 1) break;
  }
  return part;
}

// Pre-order of a depth-first walk over the out-edges (every state, roots in state order) and each state's parent in it.
void depthFirstWalk(int N, const std::vector<Edge>& edges, const std::vector<std::vector<int>>& outOf, std::vector<int>* parent,
                    std::vector<int>* walk) {
  std::vector<int> pre((size_t)N, -1);
  parent->assign((size_t)N, -1);
  walk->clear();
  walk->reserve((size_t)N);
  int next = 0;
  std::vector<std::pair<int, size_t>> stack;
  for (int root = 0; root < N; ++root) {
    if (pre[(size_t)root] >= 0) continue;
    pre[(size_t)root] = next++;
    walk->push_back(root);
    stack.emplace_back(root, 0);
    while (!stack.empty()) {
      const int u = stack.back().first;
      if (stack.back().second < outOf[(size_t)u].size()) {
        const int v2 = edges[(size_t)outOf[(size_t)u][stack.back().second++]].dst;
        if (pre[(size_t)v2] < 0) { pre[(size_t)v2] = next++; (*parent)[(size_t)v2] = u; walk->push_back(v2); stack.emplace_back(v2, 0); }
      } else {
        stack.pop_back();
      }
    }
  }
}

// Lane placement inside each row.  Two goals.  (1) The waves of a work-group sweep without a
// barrier and drift apart, so an edge into a later row is only CERTAIN to be relaxed in the
// same sweep when source and destination belong to the same wave (a wave runs its rows in
// order): a state goes to the wave of its "lead" -- the source in an earlier row it hangs on.
// (2) LDS is 64 banks of 4 bytes and a 64-bit LDS operation is served in two 32-lane halves,
// so a push is conflict-free when the 32 destination cells of a half fall on 32 different
// bank pairs, i.e. have different (lane mod 32): best is the lead's own lane (the chain stays
// inside one thread), then the other lane of that wave with the same residue, then any lane
// of the wave whose half does not yet use that bank pair, then the same residue elsewhere.
// Returns the lane of every state; rowMembers[(member * K + row)] lists the states of each row.
std::vector<int> placeLanes(int N, int G, int K, int T, const std::vector<Edge>& edges, const std::vector<std::vector<int>>& inOf,
                            const std::vector<int>& part, const std::vector<int>& rowOfState, std::vector<std::vector<int>>* rowMembersOut) {
  std::vector<int> laneOf(N, -1);
  std::vector<std::vector<int>>& rowMembers = *rowMembersOut;
  rowMembers.assign((size_t)G * K, {});
  auto bucket = [&](int j) { return (size_t)part[j] * K + rowOfState[j]; };
  for (int j = 0; j < N; ++j) { laneOf[j] = (int)rowMembers[bucket(j)].size(); rowMembers[bucket(j)].push_back(j); }
  auto leadOf = [&](int j) {
    int any = -1;
    for (int e : inOf[j]) {
      const int s = edges[e].src;
      if (s == j || part[s] != part[j]) continue;
      if (rowOfState[s] < rowOfState[j]) return s;
      if (any < 0) any = s;
    }
    return any;
  };
  for (int pass = 0; pass < 2; ++pass) {
    for (size_t b = 0; b < rowMembers.size(); ++b) {
      const std::vector<int>& mem = rowMembers[b];
      if (mem.empty()) continue;          // padding row
      const int n = (int)mem.size();
      std::vector<char> lanesFree(T, 1);
      std::vector<int> newLane(n, -1), lead(n, -1);
      std::vector<std::array<unsigned char, 32>> bankUse(T / 32);   // per 32-lane half: leads per bank pair
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
      for (int i = 0; i < n; ++i) {        // same wave, a half that does not use this bank pair yet
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
  return laneOf;
}

TierAPlan buildPlan(const dnas_flat_model& fm, const int G, const int T, const PlanChoice& choice = PlanChoice()) {
  TierAPlan p;
  const int N0 = fm.n_states, D = fm.max_dup_len;   // N0: the machine's states
  int N = N0;                                       // ... plus the proxies of a cluster (below)
  if (T != 1024 && T != 512) { TierAPlan bad; bad.whyNot = "work-groups of 512 or 1024 threads"; return bad; }
  // a work-group fills a CU either way: 16 waves of 128 registers, or 8 waves of 256 with twice the rows per thread
  const int maxRows = kTierAMaxRows * 1024 / T, maxEntries = kMaxEntries * 1024 / T;
  p.N = N0; p.D = D; p.T = T; p.G = G;
  auto no = [&](const std::string& why) { p.ok = false; p.whyNot = why; return p; };
  if (D > 8) return no("more than 8 duplication lanes");
  if (G < 1 || G > kTierCMaxMembers) return no("cluster size out of range");
  // rows come in pairs: a thread's rows 2m and 2m+1 are neighbours in the HBM lattice, so that
  // the column goes out (and the S history comes back) as 16-byte accesses
  int K;
  if (G == 1) {
    K = 2 * ((N + 2 * T - 1) / (2 * T));
  } else {
    // a member's rows are dealt by shape (S rows, G rows, entry counts): leave them some air
    const long perMember = (N + G - 1) / G;
    K = 2 * (int)((perMember * 100 / 86 + 2 * T - 1) / (2 * T));
    K = std::max(K, 2);
  }
  if (K > maxRows) return no("more than " + std::to_string((long)maxRows * T * G * (G == 1 ? 100 : 86) / 100) + " states");
  p.K = K; p.NSm = K * T; p.NS = G * K * T;

  // score classes: 0.0 plus up to three distinct input-symbol log-probabilities
  std::vector<double> scores{0.0};
  auto scoreIdx = [&](double s) -> int {
    for (size_t i = 0; i < scores.size(); ++i)
      if (memcmp(&scores[i], &s, sizeof s) == 0) return (int)i;
    scores.push_back(s);
    return (int)scores.size() - 1;
  };
  std::vector<Edge> edges;
  for (int j = 0; j < N; ++j) {
    for (int e = fm.ein_ptr[j]; e < fm.ein_ptr[j + 1]; ++e)
      edges.push_back({fm.ein_src[e], j, scoreIdx(fm.ein_score[e]), fm.ein_base[e], 0});
    for (int e = fm.nin_ptr[j]; e < fm.nin_ptr[j + 1]; ++e)
      edges.push_back({fm.nin_src[e], j, scoreIdx(fm.nin_score[e]), 0, 1});
  }
  if (scores.size() > 4) return no("more than three distinct edge scores");
  for (size_t i = 0; i < scores.size(); ++i) p.score[i] = scores[i];
  p.nClasses = (int)scores.size();

  std::vector<std::vector<int>> outOf(N), inOf(N);   // edge indices
  for (size_t e = 0; e < edges.size(); ++e) { outOf[edges[e].src].push_back((int)e); inOf[edges[e].dst].push_back((int)e); }

  // depth-first walk of the machine (the partition and the row dealing both follow it)
  std::vector<int> parent, walk;
  depthFirstWalk(N, edges, outOf, &parent, &walk);

  // ---- which member of the cluster owns which state.  A state with an in-edge from another member gets a slot in
  // its member's INBOX: cells of the cluster's exchange buffer that the other members offer into and that the
  // member folds into the state's LDS accumulators, slot r*T + t by thread t.
  std::vector<int> part(N, 0);
  if (G > 1) part = partitionStates(N, edges, walk, G, (int)((long)K * T * 93 / 100));
  // PROXIES.  A state that many states of ANOTHER member reach over null edges (the 258 538-state composite's state 0:
  // 2 245 null edges from the 20 other members) would need a cell per edge.  Instead those edges end at a proxy INSIDE the
  // source member -- an extra "state" without emission or context, fed over the same null edges with the same scores --
  // whose one out-edge, a null edge of score class 0 (nothing is added), carries the maximum to the real destination: the
  // sources combine in LDS and one thread talks to the other member.  max is associative and the proxy adds nothing, so the
  // destination's D and S cells receive exactly max over the sources of (D + score) and (S + score) as before; what the
  // proxy adds of its own (S >= D + delEnd, like every state) the destination derives from its D cell anyway.
  if (G > 1 && !getenv("DNAS_PLAN_NO_PROXIES")) {
    int kProxyMin = 3;     // edges of one member into one state from which a proxy pays (a place, and a row's delay)
    if (const char* e = getenv("DNAS_PLAN_PROXY_MIN")) kProxyMin = std::max(2, atoi(e));   // tests
    std::map<std::pair<int, int>, std::vector<int>> byPair;   // (source member, destination) -> the null edges between them
    for (size_t ei = 0; ei < edges.size(); ++ei) {
      const Edge& e = edges[ei];
      if (e.isNull && part[e.src] != part[e.dst]) byPair[{part[e.src], e.dst}].push_back((int)ei);
    }
    for (const auto& kv : byPair) {
      if ((int)kv.second.size() < kProxyMin) continue;
      const int P = N++;
      part.push_back(kv.first.first);
      parent.push_back(edges[(size_t)kv.second.front()].src);
      walk.push_back(P);
      for (int ei : kv.second) edges[(size_t)ei].dst = P;
      edges.push_back(Edge{P, kv.first.second, 0, 0, 1});
    }
    if (N > N0) {
      outOf.assign(N, {}); inOf.assign(N, {});
      for (size_t e = 0; e < edges.size(); ++e) { outOf[edges[e].src].push_back((int)e); inOf[edges[e].dst].push_back((int)e); }
    }
  }
  // Every edge between two members gets a cell of its own in the destination member's inbox (a MAILBOX: one writer, the
  // thread that owns the edge's source state; one reader, the thread of the owner that folds it), so that an offer is a
  // plain 8-byte store of a value that only grows within a column -- no atomic, and on a cluster that sits on one XCD the
  // cell never leaves that XCD's L2.  The cells of NULL edges come first: only they need an S cell as well (null edges carry
  // S as well as D, viterbi.cpp:137-151), and the kernel polls S cells for those rows only.
  std::vector<int> inboxSlot(edges.size(), -1);
  std::vector<int> inboxCount(G, 0), inboxSCount(G, 0);
  long nCross = 0;
  for (int pass = 0; pass < 2; ++pass)
    for (size_t ei = 0; ei < edges.size(); ++ei) {
      const Edge& e = edges[ei];
      if (part[e.src] != part[e.dst] && (pass == 0) == (e.isNull != 0)) {
        ++nCross;
        inboxSlot[ei] = inboxCount[part[e.dst]]++;
        if (pass == 0) ++inboxSCount[part[e.dst]];
      }
    }
  p.crossEdges = edges.empty() ? 0. : (double)nCross / (double)edges.size();
  for (int j = 0; j < N; ++j)
    if (parent[j] >= 0 && part[parent[j]] != part[j]) parent[j] = -1;   // the dealing follows a member's own subtrees
  std::vector<std::vector<int>> walkOf(G);
  for (int j : walk) walkOf[part[j]].push_back(j);
  // The order the states are dealt in.  Depth first keeps a chain together; BREADTH first (inside each member, from the
  // member's entry points in depth-first order) deals the states of one depth next to each other, so that what grows in
  // the same sweep sits in the same rows and waves -- measured on MI355X: 0.556 against 0.538 of the roofline on
  // s16h74l4c4 (13.8 sweeps per column against 14.2), 0.407 against 0.364 on water64.1*l4c4 (23 against 28).
  int orderVariant = choice.order;
  if (orderVariant < 0) orderVariant = getenv("DNAS_PLAN_ORDER") ? atoi(getenv("DNAS_PLAN_ORDER")) : 1;
  bool breadthFirst = orderVariant != 0;
  if (breadthFirst && (orderVariant == 2 || orderVariant == 3)) {
    // Longest-path layering (order 2).  Back edges = edges to a state that is still on the depth-first stack; over the
    // rest (a DAG) a state's level is 1 + the highest level of its predecessors, and it is dealt after all of them, right
    // behind the deepest one.  (3, an experiment: forward = towards a greater breadth-first depth; behaves like order 1.)
    std::vector<int> pos((size_t)N, 0);
    for (size_t i = 0; i < walk.size(); ++i) pos[walk[i]] = (int)i;
    // post-order numbers tell ancestors: u is an ancestor of v iff pre[u] <= pre[v] and post[u] >= post[v]
    std::vector<int> last((size_t)N, 0);      // last preorder index inside the subtree
    for (size_t i = walk.size(); i-- > 0;) {
      const int v = walk[i];
      last[v] = std::max(last[v], (int)i);
      // (parent[] here is still the depth-first parent, restricted to the member)
    }
    std::vector<int> dfsParent(parent);
    for (size_t i = walk.size(); i-- > 0;) { const int v = walk[i]; if (dfsParent[v] >= 0) last[dfsParent[v]] = std::max(last[dfsParent[v]], last[v]); }
    // variant 3: an edge is a forward edge when it leads to a greater breadth-first depth
    std::vector<int> bfsDepth((size_t)N, -1);
    if (orderVariant == 3) {
      std::vector<int> q;
      for (int root : walk) {
        if (bfsDepth[root] >= 0) continue;
        bfsDepth[root] = 0;
        size_t h = q.size();
        q.push_back(root);
        while (h < q.size()) {
          const int u = q[h++];
          for (int ei : outOf[u]) { const int v = edges[ei].dst; if (part[v] == part[u] && bfsDepth[v] < 0) { bfsDepth[v] = bfsDepth[u] + 1; q.push_back(v); } }
        }
      }
    }
    auto isBack = [&](const Edge& e) {
      if (orderVariant == 3) return bfsDepth[e.dst] <= bfsDepth[e.src];
      return part[e.src] == part[e.dst] && pos[e.dst] <= pos[e.src] && last[e.dst] >= pos[e.src];
    };
    std::vector<int> level((size_t)N, 0), indeg((size_t)N, 0);
    for (const Edge& e : edges) if (part[e.src] == part[e.dst] && !isBack(e) && e.src != e.dst) ++indeg[e.dst];
    std::vector<int> ready;
    for (int j : walk) if (indeg[j] == 0) ready.push_back(j);
    std::vector<int> topo;
    for (size_t h = 0; h < ready.size(); ++h) {
      const int u = ready[h];
      topo.push_back(u);
      for (int ei : outOf[u]) {
        const Edge& e = edges[ei];
        if (part[e.src] != part[e.dst] || isBack(e) || e.src == e.dst) continue;
        if (level[u] + 1 > level[e.dst]) { level[e.dst] = level[u] + 1; parent[e.dst] = u; }
        if (--indeg[e.dst] == 0) ready.push_back(e.dst);
      }
    }
    if ((int)topo.size() == N) {
      // A state may sit anywhere between its earliest level (above) and its latest (the deepest level minus the longest
      // forward path below it); `slack` eighths of that room are used, i.e. the state moves towards the states it feeds.
      int slackShare = choice.slack;
      if (slackShare < 0) slackShare = getenv("DNAS_PLAN_SLACK") ? atoi(getenv("DNAS_PLAN_SLACK")) : 0;
      slackShare = std::max(0, std::min(8, slackShare));
      if (slackShare > 0) {
        std::vector<int> down((size_t)N, 0);
        int deepest = 0;
        for (size_t i = topo.size(); i-- > 0;) {
          const int u = topo[i];
          for (int ei : outOf[u]) {
            const Edge& e = edges[ei];
            if (part[e.src] != part[e.dst] || isBack(e) || e.src == e.dst) continue;
            down[u] = std::max(down[u], down[e.dst] + 1);
          }
          deepest = std::max(deepest, level[u]);
        }
        for (int j = 0; j < N; ++j) { const int room = std::max(0, deepest - down[j] - level[j]); level[j] += room * slackShare / 8; }
      }
      std::stable_sort(topo.begin(), topo.end(), [&](int a2, int b2) { return level[a2] < level[b2]; });
      for (int g = 0; g < G; ++g) walkOf[g].clear();
      for (int j : topo) walkOf[part[j]].push_back(j);
      for (int j = 0; j < N; ++j) if (level[j] == 0) parent[j] = -1;
      walk = topo;
    }
    breadthFirst = false;
  }
  if (breadthFirst) {
    std::vector<char> seen((size_t)N, 0);
    for (int g = 0; g < G; ++g) {
      std::vector<int> order;
      order.reserve(walkOf[g].size());
      for (int root : walkOf[g]) {
        if (seen[root]) continue;
        seen[root] = 1;
        parent[root] = -1;
        size_t head = order.size();
        order.push_back(root);
        while (head < order.size()) {
          const int u = order[head++];
          for (int e : outOf[u]) {
            const int v2 = edges[e].dst;
            if (part[v2] != g || seen[v2]) continue;
            seen[v2] = 1; parent[v2] = u; order.push_back(v2);
          }
        }
      }
      walkOf[g].swap(order);
    }
    walk.clear();
    for (int g = 0; g < G; ++g) walk.insert(walk.end(), walkOf[g].begin(), walkOf[g].end());
  }
  int nInboxRows = 0;
  for (int g = 0; g < G; ++g) nInboxRows = std::max(nInboxRows, (inboxCount[g] + T - 1) / T);
  if (G > 1 && nInboxRows == 0) nInboxRows = 1;   // (a cluster whose members never talk: keep the kernel's shape)
  nInboxRows = (nInboxRows + 1) & ~1;             // the kernel loads the rows in pairs (cell (r, t) at (r/2)*2T + 2t + (r&1), 16 bytes per thread)
  int nInboxSRows = 0;
  for (int g = 0; g < G; ++g) nInboxSRows = std::max(nInboxSRows, (inboxSCount[g] + T - 1) / T);
  p.nGRows = nInboxRows;
  p.nGSRows = nInboxSRows;
  if ((long)G * nInboxRows * T > (1l << 20)) return no("more than 2^20 exchange cells");
  if (nInboxRows > 12) return no("more than " + std::to_string(12 * T) + " edges from other members into one member");

  std::vector<Type> type(N);
  int maxOut = 0;
  std::vector<int> nNullDestOf(G, 0), remoteOutOf(G, 0), remoteOutSOf(G, 0);
  for (int j = 0; j < N; ++j) {
    int hasS = 0;
    for (int e : inOf[j]) hasS |= edges[e].isNull;
    int notPlain = 0;
    for (int e : outOf[j]) notPlain |= (edges[e].isNull || edges[e].sc != 0) ? 1 : 0;
    if (outOf[j].empty()) notPlain = 1;   // would leave an empty entry in an otherwise full plain row
    int cls = 7;
    for (int e : outOf[j]) cls = cls == 7 ? edges[e].sc : (cls == edges[e].sc ? cls : 4);
    int remoteOut = 0;
    for (int e : outOf[j]) remoteOut |= part[edges[e].dst] != part[j] ? 1 : 0;
    type[j] = Type{(int)outOf[j].size(), hasS, notPlain, cls, remoteOut};
    remoteOutOf[part[j]] += remoteOut;
    remoteOutSOf[part[j]] += remoteOut && hasS;
    maxOut = std::max(maxOut, type[j][0]);
    nNullDestOf[part[j]] += hasS;
  }
  int nNullDestMax = 0, nRemoteRows = 0;
  for (int g = 0; g < G; ++g) nNullDestMax = std::max(nNullDestMax, nNullDestOf[g]);
  // the states that offer into another member sit in the last rows of the program (they are where a member's chains
  // end); the other rows keep the cheap all-LDS entry decode
  int nRemoteSRows = 0;     // ... of which this many carry S cells (such a state may have null in-edges too)
  for (int g = 0; g < G; ++g) {
    const int withS = (remoteOutSOf[g] + T - 1) / T;
    nRemoteSRows = std::max(nRemoteSRows, withS);
    nRemoteRows = std::max(nRemoteRows, withS + (remoteOutOf[g] - remoteOutSOf[g] + T - 1) / T);
  }
  nRemoteRows = std::max(nRemoteRows, nRemoteSRows);
  // Measured: reserved rows pay with 512-thread work-groups (+10 % on the 258 538-state machine, 21 members) and under
  // load, but cost a lone read on 1024-thread work-groups 20 % (more entry registers): on for 512 threads only.
  bool reserveRows = T == 512;
  if (const char* e = getenv("DNAS_PLAN_REMOTE_ROWS")) reserveRows = atoi(e) != 0;
  if (!reserveRows) nRemoteRows = nRemoteSRows = 0;
  if (getenv("DNAS_PLAN_DEBUG") && G > 1) {
    for (int g = 0; g < G; ++g)
      fprintf(stderr, "plan member %d: %zu states, %d edges from other members, %d states with null in-edges\n", g, walkOf[g].size(), inboxCount[g], nNullDestOf[g]);
    fprintf(stderr, "plan: K %d inbox rows %d cross edges %.4f\n", K, nInboxRows, p.crossEdges);
  }

  // ---- which state goes to which row ------------------------------------------------------
  // A thread walks its rows 0..K-1 in every sweep, so a value crosses an edge within the same
  // sweep when the destination sits in a LATER row than the source and needs another sweep
  // otherwise.  The in-column recursion runs along the machine's chains (deletions follow the
  // emit edges), so the number of sweeps to the fixpoint is about (how far a value travels) x
  // (share of backward edges on its way).  The rows therefore form a "program" of shapes --
  // how many out-edge entries a row's states may have, and whether the row carries S cells
  // (states with null in-edges) -- and the states are dealt onto it along a depth-first walk
  // of the machine, each state into the first row behind its parent's row whose shape admits
  // it: a chain runs down the rows of one sweep instead of along one row.  Candidate programs
  // (how many S rows, in how many groups, uniform or ascending entry counts, rows reserved for
  // "plain" states, S rows of one score class each, last row left empty) are scored by (cost of a
  // sweep: accumulator reads + what the offers of its entries cost) x (sweeps, estimated from the
  // largest number of backward edges on any walk of kWalk edges); the best one that fits the
  // registers and the LDS is kept.  DNAS_PLAN_DEBUG=1 prints the candidates, DNAS_PLAN_PICK=
  // "rows,S-rows,groups,ascending,plain,typedS" forces one (experiments).
  // The members of a cluster share the program.
  std::map<Type, int> typeId;
  std::vector<int> typeOf(N);
  std::vector<Type> types;
  for (int j = 0; j < N; ++j) {
    auto it = typeId.find(type[j]);
    if (it == typeId.end()) { it = typeId.emplace(type[j], (int)types.size()).first; types.push_back(type[j]); }
    typeOf[j] = it->second;
  }
  // deal the states of every member onto a program (caps per row) along the walk
  auto deal = [&](const std::vector<Type>& caps, std::vector<int>* rows) -> bool {
    std::vector<unsigned> admits(types.size(), 0), own(types.size(), 0);
    for (size_t t = 0; t < types.size(); ++t)
      for (int k = 0; k < K; ++k)
        if (types[t][0] <= caps[k][0] && types[t][1] <= caps[k][1] && types[t][2] <= caps[k][2] && ((caps[k][3] >> types[t][3]) & 1) &&
            (types[t][4] == 0 || caps[k][4] >= 1)) {
          // caps[k][4]: 0 no state that offers into another member, 1 any state, 2 a row reserved for such states
          admits[t] |= 1u << k;
          // S rows and the reserved rows are kept for the states that need them
          const bool reserved = caps[k][4] == 2;
          if (reserved ? types[t][4] == 1 : types[t][1] == caps[k][1]) own[t] |= 1u << k;
        }
    rows->assign(N, -1);
    for (int g = 0; g < G; ++g) {
      std::vector<std::vector<int>> members(K);
      unsigned freeRows = (1u << K) - 1u;
      auto put = [&](int j, int k) {
        (*rows)[j] = k;
        members[k].push_back(j);
        if ((int)members[k].size() == T) freeRows &= ~(1u << k);
      };
      auto pick = [&](int j, unsigned exclude) -> int {
        const int par = parent[j];
        const int start = (par >= 0 && (*rows)[par] >= 0) ? (*rows)[par] + 1 : 0;
        const unsigned base = freeRows & ~exclude;
        const unsigned sets[2] = {own[typeOf[j]], admits[typeOf[j]]};
        for (unsigned set : sets) {
          const unsigned avail = set & base;
          if (!avail) continue;
          const unsigned fw = start < 32 ? avail & ~((1u << start) - 1u) : 0u;
          return __builtin_ctz(fw ? fw : avail);
        }
        return -1;
      };
      for (int j : walkOf[g]) {
        int k = pick(j, 0u);
        if (k < 0) {
          // every row that admits j is full: move a more flexible resident of one of them elsewhere
          bool moved = false;
          for (int r = 0; r < K && !moved; ++r) {
            if (!(admits[typeOf[j]] >> r & 1u)) continue;
            for (size_t m = 0; m < members[r].size(); ++m) {
              const int i = members[r][m];
              if (!(admits[typeOf[i]] & freeRows & ~(1u << r))) continue;
              members[r].erase(members[r].begin() + (long)m);
              freeRows |= 1u << r;
              (*rows)[i] = -1;
              put(i, pick(i, 1u << r));
              moved = true;
              break;
            }
          }
          if (!moved) return false;
          k = pick(j, 0u);
          if (k < 0) return false;
        }
        put(j, k);
      }
    }
    return true;
  };
  auto score = [&](const std::vector<int>& rows, int* readsOut, int* backOut, int* entriesOut) -> double {
    std::vector<Type> shape(K, Type{0, 0, 0, 0, 0});
    std::vector<char> inUse(K, 0);
    for (int j = 0; j < N; ++j) {
      const int k = rows[j];
      inUse[k] = 1;
      shape[k][0] = std::max(shape[k][0], type[j][0]);
      shape[k][1] = std::max(shape[k][1], type[j][1]);
      shape[k][2] = std::max(shape[k][2], type[j][2]);
    }
    int reads = 0, entries = 0;
    double offerCost = 0;   // in units of one fully decoded entry
    for (int k = 0; k < K; ++k)
      if (inUse[k]) {
        int clsMask = 0;
        for (int j = 0; j < N; ++j) if (rows[j] == k && type[j][3] != 7) clsMask |= 1 << type[j][3];
        const bool oneClass = clsMask != 0 && (clsMask & (clsMask - 1)) == 0 && clsMask < 16;
        reads += 1 + shape[k][1];
        entries += shape[k][0];
        offerCost += !shape[k][2] ? 0.25 * shape[k][0] : (oneClass ? 0.55 * shape[k][0] : shape[k][0]);
      }
    std::vector<int> f(N, 0), g(N);
    for (int h = 0; h < kWalk; ++h) {
      std::fill(g.begin(), g.end(), 0);
      for (const Edge& e : edges) {
        const int v = f[e.src] + ((rows[e.dst] <= rows[e.src] || part[e.dst] != part[e.src]) ? 1 : 0);
        if (v > g[e.dst]) g[e.dst] = v;
      }
      f.swap(g);
    }
    int back = 0;
    for (int v : f) back = std::max(back, v);
    *readsOut = reads; *backOut = back; *entriesOut = entries;
    // A sweep costs its accumulator reads plus, where cells grew, one offer per entry (an offer is what
    // most of a sweep's instructions go to; entries whose kind / class is a property of the row are far
    // cheaper).  Sweeps: the GPU needs about 8 + back/2 (measured on s16h74l4c4 layouts with back 12..17).
    return (0.5 * (double)reads + 1.5 * offerCost + 5.0) * (8.0 + 0.5 * (double)back);
  };
  auto ldsNeed = [&](int nS) { return (size_t)(p.NSm + nS * T + 8 + 28 + T / 64 + (T / 64 + 2) / 2 + 1 + 2) * sizeof(double); };

  std::vector<int> rowOfState;
  std::vector<Type> bestCaps;
  double bestScore = -1;
  std::string why = "no row program fits";
  // out-degrees in ascending order, per kind, for the quantile shapes (of the largest member: the others fit below it)
  std::vector<int> outS, outP, outPlainP;   // outPlainP: plain states without null in-edges
  {
    std::vector<std::vector<int>> oS(G), oP(G), oPP(G);
    for (int j = 0; j < N; ++j) {
      (type[j][1] ? oS : oP)[part[j]].push_back(type[j][0]);
      if (!type[j][1] && !type[j][2]) oPP[part[j]].push_back(type[j][0]);
    }
    auto merge = [&](std::vector<std::vector<int>>& per, std::vector<int>* out, bool everyMember) {
      // element i = the largest i-th smallest out-degree of any member: a row shaped for it serves every member;
      // everyMember: only as many elements as the smallest member has (a census every member must meet)
      size_t n = everyMember ? (size_t)-1 : 0;
      for (auto& v : per) { std::sort(v.begin(), v.end()); n = everyMember ? std::min(n, v.size()) : std::max(n, v.size()); }
      out->assign(n, 0);
      for (size_t i = 0; i < n; ++i)
        for (auto& v : per)
          if (i < v.size()) (*out)[i] = std::max((*out)[i], v[i]);
    };
    merge(oS, &outS, false);
    merge(oP, &outP, false);
    merge(oPP, &outPlainP, true);
  }
  long biggest = 0;
  for (int g = 0; g < G; ++g) biggest = std::max(biggest, (long)walkOf[g].size());
  // Candidate programs: S rows, plain rows, quantile caps ...
  auto tryPrograms = [&](int minS) {
    const int VK = K;
    int nRemoteRowsNow = nRemoteRows, nRemoteSRowsNow = nRemoteSRows;
    // K is even (lattice pairs); when the states fit K-1 rows the last one may stay empty and
    // costs nothing in a sweep -- tried both ways
    // (when the rows reserved for the states that offer into other members leave no program -- tiny machines cut into
    //  clusters -- every row may hold them instead)
    for (int attempt = 0; attempt < 2 && bestScore < 0; ++attempt, nRemoteRowsNow = nRemoteSRowsNow = 0)
    for (int KU = VK; KU >= std::max(1, VK - 1); --KU)
    for (int nS = minS; nS <= std::min(KU, minS + 2); ++nS) {
      if (ldsNeed(nS + nRemoteSRowsNow) > kTierALdsLimit) { why = "LDS working set " + std::to_string(ldsNeed(nS + nRemoteSRowsNow)) + " B exceeds one CU"; continue; }
      if ((long)KU * T < biggest) continue;
      if (nS + nRemoteRowsNow > KU) continue;
      const int KUL = KU - nRemoteRowsNow;                 // rows in front of them
      for (int groups = 1; groups <= std::max(1, std::min(3, nS)); ++groups) {
        for (int ascending = 0; ascending < 2; ++ascending)
        for (int plainRows = 0; plainRows < 2; ++plainRows)
        for (int typedS = 0; typedS < 2; ++typedS) {
          // kinds: the S rows in `groups` runs spread evenly over the program
          std::vector<int> isS(VK, 0);
          for (int g = 0, left = nS; g < groups && nS > 0; ++g) {
            const int len = left / (groups - g);
            const int at = g * KUL / groups;
            for (int i = 0; i < len; ++i) isS[std::min(KUL - 1, at + i)] = 1;
            left -= len;
          }
          if (std::accumulate(isS.begin(), isS.end(), 0) != nS) continue;   // runs collided
          std::vector<Type> vcaps(VK);
          int seenS = 0, seenP = 0, nPlainRows = 0;
          for (int k = 0; k < VK; ++k) {
            if (k >= KUL && k < KU) { vcaps[k] = Type{maxOut, k - KUL < nRemoteSRowsNow ? 1 : 0, 1, 0xff, 2}; continue; }
            const std::vector<int>& sorted = isS[k] ? outS : outP;
            int& seen = isS[k] ? seenS : seenP;
            int cap = maxOut;
            if (ascending && !sorted.empty()) {
              const size_t q = std::min(sorted.size() - 1, (size_t)(seen + 1) * T - 1);
              cap = (size_t)(seen + 1) * T >= sorted.size() ? maxOut : sorted[q];
            }
            // plain rows: rows without S cells that only admit plain states, as long as there are enough
            // plain states of that out-degree to fill them
            int generic = 1;
            if (plainRows && !isS[k] && k < KU) {
              const size_t have = (size_t)(std::upper_bound(outPlainP.begin(), outPlainP.end(), cap) - outPlainP.begin());
              if (have >= (size_t)(nPlainRows + 1) * T) { generic = 0; ++nPlainRows; }
            }
            vcaps[k] = k < KU ? Type{cap, isS[k], generic, 0xff, nRemoteRowsNow == 0 ? 1 : 0} : Type{-1, -1, -1, 0, -1};   // closed rows admit nothing
            ++seen;
          }
          if (typedS && nS > 0) {
            // S rows of one score class each (the kernel then adds the score without decoding it): rows are
            // allotted to the classes by their share of the states with null in-edges, largest class first;
            // what is left over stays open to every class
            std::array<long, 8> cnt{};
            for (int j = 0; j < N; ++j) if (type[j][1]) ++cnt[(size_t)type[j][3]];
            for (auto& c : cnt) c = (c + G - 1) / G;
            std::vector<int> sRows;
            for (int k = 0; k < KU; ++k) if (isS[k]) sRows.push_back(k);
            std::vector<int> order{0, 1, 2, 3};
            std::stable_sort(order.begin(), order.end(), [&](int a2, int b2) { return cnt[(size_t)a2] > cnt[(size_t)b2]; });
            size_t next = 0;
            for (int c : order) {
              const long need = (cnt[(size_t)c] + T - 1) / T;
              for (long r = 0; r < need && next < sRows.size(); ++r) vcaps[sRows[next++]][3] = (1 << c) | (1 << 7);
            }
          }
          const std::vector<Type>& caps = vcaps;
          std::vector<int> rows;
          if (!deal(caps, &rows)) {
            if (getenv("DNAS_PLAN_DEBUG"))
              fprintf(stderr, "plan candidate: rows %d S-rows %d groups %d ascending %d plain %d typedS %d -> states do not fit\n", KU, nS, groups,
                      ascending, plainRows, typedS);
            continue;
          }
          int reads = 0, back = 0, entries = 0;
          const double sc = score(rows, &reads, &back, &entries);
          if (getenv("DNAS_PLAN_DEBUG"))
            fprintf(stderr, "plan candidate: rows %d S-rows %d groups %d ascending %d plain %d typedS %d -> reads %d entries %d back %d score %.0f\n", KU, nS,
                    groups, ascending, plainRows, typedS, reads, entries, back, sc);
          if (const char* pick = getenv("DNAS_PLAN_PICK")) {   // experiments: "rows,S-rows,groups,ascending"
            int a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0;
            if (sscanf(pick, "%d,%d,%d,%d,%d,%d", &a0, &a1, &a2, &a3, &a4, &a5) == 6 &&
                (a0 != KU || a1 != nS || a2 != groups || a3 != ascending || a4 != plainRows || a5 != typedS)) continue;
          }
          if (entries > maxEntries) { why = "row shapes need " + std::to_string(entries) + " entry registers per thread"; continue; }
          if (bestScore < 0 || sc < bestScore) {
            bestScore = sc; rowOfState = rows; bestCaps = caps;
            p.sweepReads = reads; p.backEdgesOnWalk = back;
          }
        }
      }
    }
  };
  {
    const int minS = (nNullDestMax + T - 1) / T;
    tryPrograms(minS);
    if (bestScore < 0) return no(why);
  }

  // A handful of odd states can spoil what the entries of a row have in common (its score class, emit /
  // null kind): move such minorities (at most 2 % of a row) to a row that admits them and is mixed anyway.
  const std::vector<std::vector<int>>& entOut = outOf;       // every out-edge becomes an entry
  {
    auto clsOf = [&](int j) { int c = -2; for (int e : entOut[j]) c = c == -2 ? edges[e].sc : (c == edges[e].sc ? c : -1); return c; };
    auto kindOf = [&](int j) { int k = -1; for (int e : entOut[j]) { const int q = edges[e].isNull ? 2 : 1; k = k < 0 ? q : (k == q ? q : 0); } return k; };
    auto rowEntries = [&]() {
      std::vector<int> nOut(K, 0);
      for (int j = 0; j < N; ++j) nOut[rowOfState[j]] = std::max(nOut[rowOfState[j]], (int)entOut[j].size());
      return nOut;
    };
    for (int attr = 0; attr < 2; ++attr) {
      std::vector<int> fill((size_t)G * K, 0);
      for (int j = 0; j < N; ++j) ++fill[(size_t)part[j] * K + rowOfState[j]];
      std::vector<int> nOutNow = rowEntries();
      // what each row has in common right now (-1 / 0: mixed), per attribute -- over all members: they share the program
      auto common = [&](int k) {
        int c = -2;
        for (int j = 0; j < N; ++j)
          if (rowOfState[j] == k && !entOut[j].empty()) {
            const int v = attr == 0 ? clsOf(j) : kindOf(j);
            c = c == -2 ? v : (c == v ? c : (attr == 0 ? -1 : 0));
          }
        return c;
      };
      for (int k = 0; k < K; ++k) {
        std::map<int, std::vector<int>> byVal;
        for (int j = 0; j < N; ++j)
          if (rowOfState[j] == k && !entOut[j].empty()) byVal[attr == 0 ? clsOf(j) : kindOf(j)].push_back(j);
        if (byVal.size() < 2) continue;
        size_t most = 0, total = 0;
        for (const auto& kv : byVal) { most = std::max(most, kv.second.size()); total += kv.second.size(); }
        if (total - most > (size_t)G * T / 50) continue;
        for (const auto& kv : byVal) {
          if (kv.second.size() == most) continue;
          for (int j : kv.second) {
            for (int k2 = 0; k2 < K; ++k2) {
              if (k2 == k || fill[(size_t)part[j] * K + k2] >= T) continue;
              const Type& c2 = bestCaps[k2];
              if (type[j][0] > c2[0] || type[j][1] > c2[1] || type[j][2] > c2[2] || !((c2[3] >> type[j][3]) & 1) || (type[j][4] && c2[4] == 0) || (!type[j][4] && c2[4] == 2)) continue;
              if ((int)entOut[j].size() > nOutNow[k2]) continue;      // would grow the row's entry registers
              const int have = common(k2);
              const int mine = attr == 0 ? clsOf(j) : kindOf(j);
              if (!(have == (attr == 0 ? -1 : 0) || have == mine || have == -2)) continue;   // would spoil k2
              rowOfState[j] = k2; --fill[(size_t)part[j] * K + k]; ++fill[(size_t)part[j] * K + k2];
              break;
            }
          }
        }
      }
    }
  }

  // which lane of its row every state gets (placeLanes above)
  std::vector<std::vector<int>> rowMembers;                    // states of each (member, row)
  const std::vector<int> laneOf = placeLanes(N, G, K, T, edges, inOf, part, rowOfState, &rowMembers);

  // row shapes as used, S stripes
  p.rows.assign(K, RowShape{0, -1, -1, -2, 0, -1});   // kind / cls / gOut: -1 / -2 / -1 = no entry seen yet
  std::vector<int> needS(K, 0);
  long real = 0;
  for (int j = 0; j < N; ++j) {
    RowShape& r = p.rows[rowOfState[j]];
    r.nOut = std::max(r.nOut, (int)entOut[j].size());
    for (int e : entOut[j]) {
      const int kind = edges[e].isNull ? 2 : 1;
      r.kind = r.kind < 0 ? kind : (r.kind == kind ? kind : 0);
      r.cls = r.cls == -2 ? edges[e].sc : (r.cls == edges[e].sc ? r.cls : -1);
      const int go = part[edges[e].dst] != part[j] ? 1 : 0;
      r.gOut = r.gOut < 0 ? go : (r.gOut == go ? go : 2);
    }
    needS[rowOfState[j]] |= type[j][1];
    real += (long)entOut[j].size();
  }
  p.nSRows = 0;
  for (int k = 0; k < K; ++k) if (needS[k]) p.rows[k].sIdx = p.nSRows++;
  std::vector<char> rowUsed(K, 0);
  for (int j = 0; j < N; ++j) rowUsed[rowOfState[j]] = 1;
  for (int k = 0; k < K; ++k) if (!rowUsed[k]) p.rows[k].nOut = -1;   // the kernel skips the row
  for (RowShape& r : p.rows) { if (r.kind < 0) r.kind = 0; if (r.cls == -2) r.cls = -1; if (r.gOut < 0) r.gOut = 0; }
  for (int k = 0; k < K; ++k) {
    bool full = true;
    for (int g = 0; g < G; ++g) {
      const std::vector<int>& mem = rowMembers[(size_t)g * K + k];
      full = full && (int)mem.size() == T;
      for (int j : mem) full = full && (int)entOut[j].size() == p.rows[k].nOut;
    }
    p.rows[k].full = full ? 1 : 0;
  }
  p.ldsBytes = ldsNeed(p.nSRows);
  if (p.ldsBytes > kTierALdsLimit) return no("LDS working set " + std::to_string(p.ldsBytes) + " B exceeds one CU");
  int nEnt = 0;
  std::vector<int> rowOff(K, 0);
  for (int k = 0; k < K; ++k) { rowOff[k] = nEnt; nEnt += std::max(p.rows[k].nOut, 0); }
  if (nEnt > maxEntries) return no("row shapes need " + std::to_string(nEnt) + " entry registers per thread");
  p.nEntries = std::max(nEnt, 1);
  p.fillRatio = nEnt ? (double)real / ((double)nEnt * T * G) : 1.0;

  // Which two rows of a thread share a 16-byte cell pair of the lattice.  A thread with a state in only ONE row of a pair
  // still moves the pair with every store and history load; the rows of a program are not equally full (and not equally
  // in every member), so rows 2m and 2m+1 as partners padded the lattice traffic of the bench machines by 13-16 %.  The
  // rows are paired by fill instead: start from the rows sorted by their fill over all members, then swap partners
  // between two pairs as long as the sum over members of |fill(a) - fill(b)| goes down (K <= 28: a few hundred trials).
  std::vector<int> pairRows(K), pairSlot(K);     // pairRows[2m], [2m+1]: the rows of pair m; pairSlot[row] = 2m + side
  {
    std::vector<std::vector<long>> fill(G, std::vector<long>(K, 0));
    for (int j = 0; j < N0; ++j) ++fill[part[j]][rowOfState[j]];
    auto waste = [&](int a2, int b2) { long w = 0; for (int g = 0; g < G; ++g) w += std::labs(fill[g][a2] - fill[g][b2]); return w; };
    for (int k = 0; k < K; ++k) pairRows[k] = k;
    if (G > 1 && !getenv("DNAS_PLAN_NO_PAIRING")) {   // (one work-group per read keeps rows 2m, 2m+1: measured, its kernel is 1 % slower otherwise)
      auto total = [&](int k) { long t = 0; for (int g = 0; g < G; ++g) t += fill[g][k]; return t; };
      std::stable_sort(pairRows.begin(), pairRows.end(), [&](int a2, int b2) { return total(a2) > total(b2); });
      for (bool better = true; better;) {
        better = false;
        for (int m = 0; m + 1 < K / 2; ++m)
          for (int n = m + 1; n < K / 2; ++n) {
            int& a2 = pairRows[2 * m]; int& b2 = pairRows[2 * m + 1]; int& c2 = pairRows[2 * n]; int& d2 = pairRows[2 * n + 1];
            const long now = waste(a2, b2) + waste(c2, d2);
            if (waste(a2, c2) + waste(b2, d2) < now) { std::swap(b2, c2); better = true; }
            else if (waste(a2, d2) + waste(b2, c2) < now) { std::swap(b2, d2); better = true; }
          }
      }
      // tidy: the lower row first inside a pair, the pairs by their first row
      std::vector<std::pair<int, int>> ps;
      for (int m = 0; m < K / 2; ++m) ps.emplace_back(std::min(pairRows[2 * m], pairRows[2 * m + 1]), std::max(pairRows[2 * m], pairRows[2 * m + 1]));
      std::sort(ps.begin(), ps.end());
      for (int m = 0; m < K / 2; ++m) { pairRows[2 * m] = ps[(size_t)m].first; pairRows[2 * m + 1] = ps[(size_t)m].second; }
    }
    for (int i = 0; i < K; ++i) pairSlot[pairRows[i]] = i;
    p.pairRows.assign(pairRows.begin(), pairRows.end());
  }

  // index spaces: a member's LDS index row*T + lane (consecutive lanes -> consecutive bank pairs), and
  // the lattice slot member*K*T + pair*2T + 2*lane + side used in HBM and by the traceback
  p.memberOf.assign(part.begin(), part.begin() + N0);
  p.slotOf.assign(N, -1);
  p.stateOf.assign((size_t)p.NS, -1);       // by (member*K + row)*T + lane
  for (int j = 0; j < N; ++j) {
    const int row = rowOfState[j], lane = laneOf[j];
    if (j < N0) p.stateOf[((size_t)part[j] * K + row) * T + lane] = j;      // (a proxy's place holds no state of the machine)
    else { p.proxyMember.push_back(part[j]); p.proxyLds.push_back(row * T + lane); }
    p.slotOf[j] = part[j] * p.NSm + (pairSlot[row] >> 1) * 2 * T + 2 * lane + (pairSlot[row] & 1);
  }
  p.slotOf.resize(N0);

  // entries, one per out-edge (layout: viterbi_tiera.hip).  0: no edge.
  const unsigned dcBase = (unsigned)p.nSRows * T * 8 + 64;     // byte address of DC[0]: behind the S stripes and a pad
  if (p.nSRows * T > 0x1ffc) return no("more than 8188 S cells");
  p.entTab.assign((size_t)G * p.nEntries * T, 0u);
  long fwd = 0, fwdSameWave = 0;
  for (int j = 0; j < N; ++j) {
    const int row = rowOfState[j], lane = laneOf[j];
    for (size_t i = 0; i < entOut[j].size(); ++i) {
      const Edge& e = edges[entOut[j][i]];
      const int drow = rowOfState[e.dst], dlane = laneOf[e.dst];
      unsigned ent;
      if (part[e.dst] != part[j]) {
        // inbox cell of the destination:  [0:2) class | bit 2 | [3:23) cell | bit 23 null edge | [24:26) emitted base
        const int slot = inboxSlot[entOut[j][i]], sr = slot / T, st = slot % T;
        const unsigned cell = (unsigned)(part[e.dst] * nInboxRows * T + (sr >> 1) * 2 * T + 2 * st + (sr & 1));
        if (e.isNull && p.rows[drow].sIdx < 0) return no("internal: null edge into a row without S cells");
        ent = (unsigned)e.sc | 4u | (cell << 3) | (e.isNull ? 1u << 23 : (unsigned)(e.base & 3) << 24);
      } else {
        ent = (unsigned)e.sc | (dcBase + (unsigned)(drow * T + dlane) * 8u);
        if (e.isNull) {
          if (p.rows[drow].sIdx < 0) return no("internal: null edge into a row without S cells");
          ent |= (unsigned)(p.rows[drow].sIdx * T + dlane) << 19;
        } else {
          ent |= (0x1ffcu | (unsigned)(e.base & 3)) << 19;
        }
      }
      p.entTab[((size_t)part[j] * p.nEntries + (size_t)(rowOff[row] + (int)i)) * T + lane] = ent;
      if (drow > row && part[e.dst] == part[j]) { ++fwd; if (dlane / 64 == lane / 64) ++fwdSameWave; }
    }
  }
  p.sameWave = fwd ? (double)fwdSameWave / (double)fwd : 1.0;

  // fold table: inbox slot r*T + t of a member -> LDS cells of the state behind it: DC byte address >> 3 | SC byte
  // address >> 3 << 16 (0xffff: the state has no S cell); 0: slot unused
  p.foldTab.assign((size_t)G * nInboxRows * T, 0u);
  for (size_t ei = 0; ei < edges.size(); ++ei)
    if (inboxSlot[ei] >= 0) {
      const Edge& e = edges[ei];
      const int j = e.dst, row = rowOfState[j], lane = laneOf[j];
      const unsigned dc = (dcBase + (unsigned)(row * T + lane) * 8u) >> 3;
      if (e.isNull && (p.rows[row].sIdx < 0 || inboxSlot[ei] >= nInboxSRows * T)) return no("internal: inbox S cell");
      const unsigned scc = e.isNull ? (unsigned)(p.rows[row].sIdx * T + lane) : 0xffffu;
      p.foldTab[(size_t)part[j] * nInboxRows * T + (size_t)inboxSlot[ei]] = dc | (scc << 16);
    }

  // meta: mdl | ctx << 4 | bit29 real state | bit30 reference's last state | bit31 reference's state 0
  p.metaTab.assign((size_t)G * K * T, 0);
  for (size_t idx = 0; idx < p.metaTab.size(); ++idx) {
    const int j = p.stateOf[idx];
    unsigned meta = 0;
    if (j >= 0) {
      meta = fm.mdl[j] & 15u;
      for (int q = 0; q < fm.mdl[j] && q < 8; ++q) meta |= (unsigned)(fm.ctx[(size_t)j * D + q] & 3u) << (4 + 2 * q);
      if (j == 0) meta |= 0x80000000u;
      if (j == N0 - 1) meta |= 0x40000000u;
      meta |= 0x20000000u;   // slot holds a real state
    }
    p.metaTab[idx] = meta;
  }

  std::ostringstream rows, defs;
  for (int k = 0; k < K; ++k) {
    if (k) rows << ",";
    rows << "{" << p.rows[k].nOut << "," << p.rows[k].sIdx << "," << p.rows[k].kind << "," << p.rows[k].cls << "," << p.rows[k].full << ","
         << p.rows[k].gOut << "}";
  }
  defs << "-DDNAS_T=" << T << "\n-DDNAS_K=" << K << "\n-DDNAS_D=" << D << "\n-DDNAS_NS=" << p.NSm << "\n-DDNAS_SROWS=" << p.nSRows
       << "\n-DDNAS_NCLS=" << p.nClasses << "\n-DDNAS_G=" << G << "\n-DDNAS_GROWS=" << nInboxRows << "\n-DDNAS_GSROWS=" << nInboxSRows << "\n-DDNAS_ROWS=" << rows.str();
  {
    bool identity = true;
    for (int i = 0; i < K; ++i) identity = identity && p.pairRows[(size_t)i] == i;
    if (!identity) {
      defs << "\n-DDNAS_PAIRS=";
      for (int i = 0; i < K; ++i) defs << (i ? "," : "") << p.pairRows[(size_t)i];
    }
  }
  // Small row programs (water64.1*l4c4: 8 rows, 12 entries, 66 KB of LDS) leave half a CU idle: compiled for 64 registers per
  // thread, two work-groups of 1024 threads share a CU and hide each other's waits (measured on MI355X: 0.41 -> 0.565 of the
  // roofline; phase C then loads one cell pair at a time).  DNAS_PLAN_OCCUPANCY=1 / 2 overrides the rule.
  {
    bool two = G == 1 && T == 1024 && K <= 8 && p.nEntries <= 14 && 2 * (p.ldsBytes + 1024) <= 160 * 1024;
    if (const char* e = getenv("DNAS_PLAN_OCCUPANCY")) two = atoi(e) == 2 && G == 1 && T == 1024 && 2 * (p.ldsBytes + 1024) <= 160 * 1024;
    if (two) {
      p.wavesPerSimd = 8;
      defs << "\n-DDNAS_WAVES_PER_EU=8\n-DDNAS_CGROUP=2";
    }
  }
  p.defines = defs.str();
  p.key = "T" + std::to_string(T) + "K" + std::to_string(K) + "D" + std::to_string(D) + "S" + std::to_string(p.nSRows) + "C" +
          std::to_string(p.nClasses) + "G" + std::to_string(G) + "X" + std::to_string(nInboxRows) + "x" + std::to_string(nInboxSRows) + "R" + rows.str();
  if (p.defines.find("-DDNAS_PAIRS=") != std::string::npos) p.key += "P" + p.defines.substr(p.defines.find("-DDNAS_PAIRS=") + 13);
  if (p.wavesPerSimd) p.key += "W" + std::to_string(p.wavesPerSimd);
  p.ok = true;
  return p;
}

}  // namespace

TierAPlan buildTierAPlan(const dnas_flat_model& fm, int threads, const PlanChoice& choice) { return buildPlan(fm, 1, threads, choice); }

TierAPlan buildClusterPlan(const dnas_flat_model& fm, int G, int threads, const PlanChoice& choice) {
  if (G < 2) { TierAPlan p; p.whyNot = "a cluster has at least two members"; return p; }
  return buildPlan(fm, G, threads, choice);
}

TierAPlan buildSmallestClusterPlan(const dnas_flat_model& fm, int gMin, int threads, const PlanChoice& choice) {
  TierAPlan last;
  const int lo = std::max(2, std::max(gMin, (int)(((long)fm.n_states * 100 / 93 + (long)kTierAMaxRows * kTierAThreads - 1) / ((long)kTierAMaxRows * kTierAThreads))));
  for (int G = lo; G <= kTierCMaxMembers; ++G) {
    last = buildPlan(fm, G, threads, choice);
    if (last.ok) return last;
  }
  if (last.whyNot.empty()) last.whyNot = "more than " + std::to_string(kTierCMaxMembers) + " work-groups per read";
  return last;
}

// Tier C as the runtime asks for it: members = 0 -> the smallest cluster; threads = 0 -> work-groups of 512 threads
// (8 waves of 256 registers, twice the rows per thread: no register spills, and the machine fits fewer CUs) when that
// needs no more work-groups per read than 1024-thread ones (measured faster at equal size), else 1024.
TierAPlan chooseClusterPlan(const dnas_flat_model& fm, int members, int threads, const PlanChoice& choice) {
  auto build = [&](int t) { return members >= 2 ? buildClusterPlan(fm, members, t, choice) : buildSmallestClusterPlan(fm, 2, t, choice); };
  if (threads == 512 || threads == 1024) return build(threads);
  TierAPlan narrow = build(512);
  // both shapes hold the same number of states per work-group: when the narrow one already gets by with the fewest
  // work-groups that can hold the machine, the wide one cannot do better (and planning a 258 538-state machine takes seconds)
  const int fewest = std::max(2, (int)(((long)fm.n_states * 100 / 93 + (long)kTierAMaxRows * kTierAThreads - 1) / ((long)kTierAMaxRows * kTierAThreads)));
  if (narrow.ok && (members >= 2 || narrow.G <= fewest)) return narrow;
  TierAPlan wide = build(1024);
  if (!narrow.ok) return wide;
  if (!wide.ok) return narrow;
  return narrow.G <= wide.G ? narrow : wide;
}

}  // namespace dnas
