// Kernel-argument view of the error model for the forward-backward kernels.
#pragma once
#include <stddef.h>
#include <stdint.h>

constexpr int kFbThreads = 64;       // one wave of independent alignment pairs per block
constexpr int kFbMaxLen = 32;
constexpr int kFbMaxCounts = 21 + kFbMaxLen;

// on-chip kernels (fwdback_onchip.hip): W = 8, 16 or 32 lanes per pair, rows of up to RW = 16 or 32 cells, a wave per work-group
constexpr int kFbWave = 64;
constexpr size_t kFbOnchipLdsLimit = 64 * 1024;   // dynamic LDS a work-group of the on-chip kernel may ask for (several fit a CU)
// doubles of LDS one pair needs there: substitution counts and scores [16 + 16], length scores [8], the other counts [16: five
// transition counts, up to eight length counts], envelope bounds as int16
__host__ __device__ constexpr size_t fbOnchipPairDoubles(int W, int maxInLen) {
  return 16 + 16 + 8 + 16 + ((size_t)(maxInLen + 2) * 2 * sizeof(short) + 7) / 8 + 1;
}
// doubles of global scratch per WAVE: the Forward cells of its pairs, [S, D, T[0..7]][maxSteps steps][64 lanes]
__host__ __device__ constexpr size_t fbOnchipWaveDoubles(int maxSteps) {
  return (size_t)10 * (size_t)maxSteps * 64;
}

struct FbArgs {
  int P;             // pLen.size()
  int maxDistance;   // envelope half-width: P, or 0 with --strict-guides (fwdback.cpp:17)
  int rowCap;        // widest envelope row in the batch
  double noGap, delOpen, delExtend, delEnd, tanDup;
  double sub[16];
  double len[kFbMaxLen];
};
