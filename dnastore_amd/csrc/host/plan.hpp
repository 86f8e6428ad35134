// Tier-A schedule: how one machine's states and edges are laid over the 1024 threads of
// the register/LDS-resident fill kernel (csrc/viterbi_tiera.hip).  Pure host code.
//
//  * every state gets a slot = row*T + thread; rows are processed in order inside a sweep
//  * a state's in-edges become per-thread 32-bit entries (kept in registers by the kernel):
//      emit pull  D(j) >= X(src)+score                      (viterbi.cpp:123-125, pulled)
//      null pull  D(j) >= DN(src)+score, S(j) >= SN(src)+score   (viterbi.cpp:137-151)
//      push       in-edges of "heavy" destinations (in-degree > kHeavy) are executed by the
//                 source's thread as LDS atomic max into the destination's cell
//      publish    states that own an LDS cell (null-edge sources, heavy destinations)
//  * states are sorted so that each row is (nearly) homogeneous in entry counts; the per-row
//    maxima form the row shape that the kernel is compiled for.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/dnastore_amd.h"

namespace dnas {

// per row: emit pulls / null pulls by score class (class 0 = score 0.0), pushes, publishes
struct RowShape { int e[4], n[4], ep, ec; };

struct TierAPlan {
  bool ok = false;
  std::string whyNot;
  int T = 1024, K = 0, D = 0, N = 0, NS = 0, C = 0, xDummy = 0, nEntries = 0;
  std::vector<RowShape> rows;
  std::string defines;            // "-DDNAS_T=.. -DDNAS_K=.. -DDNAS_D=.. -DDNAS_ROWS=.." joined by '\n'
  std::string key;                // cache key of the specialisation
  std::vector<int32_t> slotOf;    // [N]  state -> slot
  std::vector<int32_t> stateOf;   // [NS] LDS index (row*T + lane) -> state or -1
  std::vector<uint32_t> entTab;   // [nEntries][T]
  std::vector<uint32_t> metaTab;  // [K][T]  mdl | ctx<<4 | flags
  std::vector<uint32_t> baseTab;  // [nBaseWords][T]  emitted base of the thread's i-th emit pull, 2 bits each, 16 per word
  int nBaseWords = 1;
  double score[4] = {0, 0, 0, 0};
  size_t ldsBytes = 0;
  double fillRatio = 0;           // real entries / padded entries
  int sweepReads = 0;             // LDS gathers per thread and sweep (emit pulls + 2 per null pull)
  int backEdgesOnWalk = 0;        // most backward edges (destination row <= source row) on any walk of 30 edges
  long ldsCycles = 0, ldsCyclesIdeal = 0;   // modelled LDS cycles of one sweep's gathers (with / without bank conflicts)
};

constexpr int kTierAThreads = 1024;
constexpr int kTierAMaxRows = 14;
constexpr size_t kTierALdsLimit = 160 * 1024 - 1024;  // leave room for the static reduction buffer

TierAPlan buildTierAPlan(const dnas_flat_model& fm);

}  // namespace dnas
