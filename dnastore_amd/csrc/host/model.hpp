// Error model (reference src/mutator.h) and the flattening of
// (Machine, MutatorParams) into the CSR tables the kernels consume
// (what reference InputModel + MachineScores + MutatorScores hold, viterbi.h:11-40).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/dnastore_amd.h"
#include "machine.hpp"

namespace dnas {

struct MutatorParams {  // reference src/mutator.h:9-31
  double pDelOpen = .001, pDelExtend = .01, pTanDup = .001, pTransition = 0, pTransversion = 0;
  std::vector<double> pLen;
  bool local = true;

  double pMatch() const { return 1. - pTransition - pTransversion; }
  double pNoGap() const { return 1. - pDelOpen - pTanDup; }
  double pDelEnd() const { return 1. - pDelExtend; }
  size_t maxDupLen() const { return pLen.size(); }

  // t/dnastore.cpp:119-129
  static MutatorParams fromFlags(double subProb, double ivRatio, double dupProb, double delOpen, double delExt,
                                 bool global, int length);
  static MutatorParams fromJSON(const std::string& text);  // mutator.cpp:18-30
  static MutatorParams fromFile(const std::string& path);  // mutator.cpp:44-49
  std::string toJSON() const;                              // mutator.cpp:6-16

  void toC(dnas_mutator_params* out) const;
  static MutatorParams fromC(const dnas_mutator_params& p);
};

// Owns the arrays a dnas_flat_model points into.
struct FlatModel {
  dnas_flat_model view{};
  std::vector<int32_t> einPtr, einSrc, ninPtr, ninSrc, eoutPtr, eoutDst, noutPtr, noutDst, topo;
  std::vector<double> einScore, ninScore, eoutScore, noutScore, len;
  std::vector<uint8_t> einIn, einBase, ninIn, mdl, ctx;

  // Throws std::runtime_error ("Not a DNA-outputting machine", context mismatch) or
  // std::domain_error (cyclic null transitions).
  static FlatModel build(const Machine& machine, const MutatorParams& params);
  void bind();  // refresh view pointers after a move
};

}  // namespace dnas
