// Tier-A / tier-C schedule: how one machine's states and edges are laid over the 1024 threads of
// the register/LDS-resident fill kernel (csrc/viterbi_tiera.hip) -- of ONE work-group (tier A,
// machines that fit one CU) or of a CLUSTER of G work-groups that share a read (tier C, machines
// beyond one CU).  Pure host code.
//
//  * every state gets a place = (member, row, thread); a thread evaluates its rows in order inside a
//    sweep, and keeps the S and D cells of its states in registers
//  * every state owns an LDS accumulator DC[row*T + thread] that its in-edges are pushed into
//    (ds_max_f64), and -- if it has null in-edges -- a second one, SC, in a stripe of T cells
//    that its row shares ("S rows")
//  * in a cluster, an EDGE into another member owns a cell of that member's inbox: a cell of the cluster's
//    exchange buffer in global memory with one writer (the thread that holds the edge's source: an 8-byte
//    store of a value that only grows within a column) and one reader (the member folds cell r*T + t into the
//    destination state's LDS cells by thread t, once per sweep)
//  * a state's OUT-edges become per-thread 32-bit entries (kept in registers by the kernel):
//      emit edge  D(dst) >= max(D+delExtend, S+delOpen) + score        (viterbi.cpp:123-125)
//                 and, between columns, S(dst) >= S + score + noGap + sub   (viterbi.cpp:92-95)
//      null edge  D(dst) >= D + score,  S(dst) >= S + score             (viterbi.cpp:137-151)
//  * the per-row maxima (out-edges, needs an S cell, G row) form the row shape the kernel is compiled
//    for; the members of a cluster share ONE row program (one code object), only their tables differ
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/dnastore_amd.h"

namespace dnas {

// per row: out-edge entries per state (-1: row left empty), the row's S stripe (-1: no state of the row has
// null in-edges), and what all entries of the row have in common, which the kernel then does not decode per
// lane: kind 1 = emit edges only, 2 = null edges only, 0 = both; cls = the common score class or -1;
// full = every lane of the row holds a state with exactly nOut out-edges (no entry is empty);
// gOut = 0: every entry of the row points into LDS, 1: every entry into the exchange buffer, 2: mixed
struct RowShape { int nOut, sIdx, kind, cls, full, gOut; };

struct TierAPlan {
  bool ok = false;
  std::string whyNot;
  int T = 1024, K = 0, D = 0, N = 0, nSRows = 0, nClasses = 1, nEntries = 0;
  int G = 1;                      // work-groups per read (1: tier A)
  int nGRows = 0;                 // inbox rows: a member's inbox holds nGRows * T cells (0 when G == 1)
  int nGSRows = 0;                // ... of which the first nGSRows * T also carry an S cell (states reached over a null edge)
  int NSm = 0;                    // lattice slots per member (= K*T)
  int NS = 0;                     // lattice slots per column (= G*K*T)
  std::vector<RowShape> rows;
  std::string defines;            // "-DDNAS_T=.. -DDNAS_K=.. -DDNAS_D=.. -DDNAS_ROWS=.." joined by '\n'
  std::string key;                // cache key of the specialisation
  std::vector<int32_t> memberOf;  // [N]  state -> member of the cluster
  std::vector<int32_t> slotOf;    // [N]  state -> lattice slot member*NSm + (row/2)*2T + 2*thread + (row&1)
  std::vector<int32_t> stateOf;   // [G*K*T] index (member*K + row)*T + thread -> state or -1
  std::vector<uint32_t> entTab;   // [G][nEntries][T]  out-edges, see viterbi_tiera.hip
  std::vector<uint32_t> metaTab;  // [G][K][T]  mdl | ctx<<4 | flags
  std::vector<int32_t> pairRows;  // [K] the rows of lattice cell pair m: pairRows[2m], pairRows[2m+1]
  std::vector<int32_t> proxyMember, proxyLds;   // cluster proxies (plan.cpp): member and row*T + lane of each
  std::vector<uint32_t> foldTab;  // [G][nGRows][T]  inbox slot -> LDS cells of its state (DC addr >> 3 | SC addr >> 3 << 16), 0: unused
  double score[4] = {0, 0, 0, 0};
  size_t ldsBytes = 0;
  double fillRatio = 0;           // real entries / padded entries
  int sweepReads = 0;             // accumulator reads per thread and sweep (own cells)
  int backEdgesOnWalk = 0;        // most backward edges (destination row <= source row, or another member) on any walk of 30 edges
  double sameWave = 0;            // share of forward edges whose ends sit in the same wave
  double crossEdges = 0;          // share of edges that go through the exchange buffer
  int wavesPerSimd = 0;           // 8: the kernel is compiled for 64 registers per thread, so that TWO work-groups of 1024 threads share a
                                  // CU (small row programs: they leave half the registers and more than half the LDS of a CU unused); 0: no limit
  long exchangeCells() const { return (long)G * nGRows * T; }   // cells of one exchange array
  long exchangeStride() const { return 3 * exchangeCells() + 2 * ((G + 15) & ~15); }   // doubles per cluster: XA.dc | XB.dc | XB.sc | reduction cells (viterbi_tiera.hip kXStride)
};

constexpr int kPlanVersion = 6;      // bumped when the planner changes what it produces: recorded tuning verdicts name it

// What the caller (a tuning record, an option, an experiment) decides about the row program; -1: as the environment says
// (DNAS_PLAN_ORDER, DNAS_PLAN_SLACK), else the default.
//   order   the order the states are dealt onto the program in: 0 depth first, 1 breadth first (default: what grows in the
//           same sweep sits in the same rows and waves), 2 by longest-path level over the machine without the edges that
//           close a cycle of the depth-first walk (a state is dealt after ALL its forward predecessors).  Which of 1 and 2 is
//           faster depends on the machine (s16h74l4c4: 0.565 / 0.581 of the roofline, water64.1*l4c4: 0.405 / 0.367):
//           dnastore_amd/tune/ holds the measured verdicts.
//   slack   order 2 only: how far (in eighths, 0 .. 8) a state moves from its earliest level towards its latest one
//           (s16h74l4c4: 0.573 of the roofline at 0, 0.599 at 8)
struct PlanChoice {
  int order = -1;
  int slack = -1;
};
constexpr int kTierAThreads = 1024;
constexpr int kTierAMaxRows = 14;
constexpr int kTierCMaxMembers = 32;                  // one XCD
constexpr size_t kTierALdsLimit = 160 * 1024 - 1024;  // leave room for the static reduction buffer

// One work-group per read; fails (ok = false) when the machine does not fit one CU.
// threads: 1024 (16 waves of 128 registers) or 512 (8 waves of 256 registers, twice the rows per thread).
TierAPlan buildTierAPlan(const dnas_flat_model& fm, int threads = kTierAThreads, const PlanChoice& choice = PlanChoice());
// G work-groups per read (G >= 2); fails when the states do not fit G CUs or no common row program exists.
TierAPlan buildClusterPlan(const dnas_flat_model& fm, int G, int threads = kTierAThreads, const PlanChoice& choice = PlanChoice());
// The smallest cluster that fits (tries G = gMin .. kTierCMaxMembers).
TierAPlan buildSmallestClusterPlan(const dnas_flat_model& fm, int gMin = 2, int threads = kTierAThreads, const PlanChoice& choice = PlanChoice());
// What the runtime uses for tier C: members = 0 -> smallest cluster, threads = 0 -> 512 when that needs fewer members.
TierAPlan chooseClusterPlan(const dnas_flat_model& fm, int members, int threads, const PlanChoice& choice = PlanChoice());

}  // namespace dnas
