// Tier-A schedule: how one machine's states and edges are laid over the 1024 threads of
// the register/LDS-resident fill kernel (csrc/viterbi_tiera.hip).  Pure host code.
//
//  * every state gets a place = (row, thread); a thread evaluates its rows in order inside a
//    sweep, and keeps the S and D cells of its states in registers
//  * every state owns an LDS accumulator DC[row*T + thread] that its in-edges are pushed into
//    (ds_max_f64), and -- if it has null in-edges -- a second one, SC, in a stripe of T cells
//    that its row shares ("S rows")
//  * a state's OUT-edges become per-thread 32-bit entries (kept in registers by the kernel):
//      emit edge  D(dst) >= max(D+delExtend, S+delOpen) + score        (viterbi.cpp:123-125)
//                 and, between columns, S(dst) >= S + score + noGap + sub   (viterbi.cpp:92-95)
//      null edge  D(dst) >= D + score,  S(dst) >= S + score             (viterbi.cpp:137-151)
//  * the per-row maxima (out-edges, needs an S cell) form the row shape the kernel is compiled for
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/dnastore_amd.h"

namespace dnas {

// per row: out-edge entries per state (-1: row left empty), the row's S stripe (-1: no state of the row has
// null in-edges), and what all entries of the row have in common, which the kernel then does not decode per
// lane: kind 1 = emit edges only, 2 = null edges only, 0 = both; cls = the common score class or -1;
// full = every lane of the row holds a state with exactly nOut out-edges (no entry is empty)
struct RowShape { int nOut, sIdx, kind, cls, full; };

struct TierAPlan {
  bool ok = false;
  std::string whyNot;
  int T = 1024, K = 0, D = 0, N = 0, NS = 0, nSRows = 0, nClasses = 1, nEntries = 0;
  std::vector<RowShape> rows;
  std::string defines;            // "-DDNAS_T=.. -DDNAS_K=.. -DDNAS_D=.. -DDNAS_ROWS=.." joined by '\n'
  std::string key;                // cache key of the specialisation
  std::vector<int32_t> slotOf;    // [N]  state -> lattice slot (row/2)*2T + 2*thread + (row&1)
  std::vector<int32_t> stateOf;   // [NS] LDS index (row*T + thread) -> state or -1
  std::vector<uint32_t> entTab;   // [nEntries][T]  out-edges, see viterbi_tiera.hip
  std::vector<uint32_t> metaTab;  // [K][T]  mdl | ctx<<4 | flags
  double score[4] = {0, 0, 0, 0};
  size_t ldsBytes = 0;
  double fillRatio = 0;           // real entries / padded entries
  int sweepReads = 0;             // LDS reads per thread and sweep (own cells)
  int backEdgesOnWalk = 0;        // most backward edges (destination row <= source row) on any walk of 30 edges
  double sameWave = 0;            // share of forward edges whose ends sit in the same wave
};

constexpr int kTierAThreads = 1024;
constexpr int kTierAMaxRows = 14;
constexpr size_t kTierALdsLimit = 160 * 1024 - 1024;  // leave room for the static reduction buffer

TierAPlan buildTierAPlan(const dnas_flat_model& fm);

}  // namespace dnas
