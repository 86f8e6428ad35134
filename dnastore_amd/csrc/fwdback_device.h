// Kernel-argument view of the error model for the forward-backward kernels.
#pragma once
#include <stdint.h>

constexpr int kFbThreads = 64;       // one wave of independent alignment pairs per block
constexpr int kFbMaxLen = 32;
constexpr int kFbMaxCounts = 21 + kFbMaxLen;

struct FbArgs {
  int P;             // pLen.size()
  int maxDistance;   // envelope half-width: P, or 0 with --strict-guides (fwdback.cpp:17)
  int rowCap;        // widest envelope row in the batch
  double noGap, delOpen, delExtend, delEnd, tanDup;
  double sub[16];
  double len[kFbMaxLen];
};
