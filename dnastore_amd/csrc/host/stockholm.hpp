// Minimal Stockholm reader for the forward-backward path: a database of two-row alignments
// (original, read) as the reference's readStockholmDatabase + Alignment + GuideAlignmentEnvelope
// produce them (src/stockholm.cpp:30-68,154-167; src/alignpath.cpp:189-204,237-265), flattened
// into the arrays dnas_fwdback_estep takes.  Mark-up lines (#=GF/GC/GR/GS) are skipped.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace dnas {

struct AlignmentPairs {
  int64_t n = 0;
  std::vector<int8_t> inSeqs, outSeqs;            // base tokens 0..3
  std::vector<int64_t> inOff, outOff;             // n+1
  std::vector<int32_t> cmIn, cmOut;               // cumulative matches at each position's column
  std::vector<int64_t> cmInOff, cmOutOff;         // n+1
  std::vector<std::string> inName, outName;
};

// Throws std::runtime_error: "File <path> not found" (stockholm.cpp:158-159), a non 2-row
// alignment (fwdback.cpp:25), rows of unequal length, or a non-ACGT residue (fastseq.cpp:25-39).
AlignmentPairs readStockholmPairs(const std::string& path);

}  // namespace dnas
