// Kernel-argument view of the error model for the forward-backward kernels.
#pragma once
#include <stddef.h>
#include <stdint.h>

constexpr int kFbThreads = 64;       // one wave of independent alignment pairs per block
constexpr int kFbMaxLen = 32;
constexpr int kFbMaxCounts = 21 + kFbMaxLen;

// on-chip kernel (fwdback_onchip.hip): sixteen lanes per pair, two pairs per work-group
constexpr int kFbLanes = 16;
constexpr int kFbPairsPerGroup = 2;
constexpr size_t kFbOnchipLdsLimit = 150 * 1024;   // dynamic LDS a work-group of the on-chip kernel may ask for
// doubles of LDS one pair needs there (Forward block, checkpoints, two Backward rows per lane + 1, substitution
// counts, envelope bounds as int16)
__host__ __device__ constexpr size_t fbOnchipPairDoubles(int maxInLen) {
  return (size_t)kFbLanes * (kFbLanes * 8 + 2) + (size_t)((maxInLen + 1 + kFbLanes - 1) / kFbLanes) * kFbLanes * 2 +
         (size_t)(kFbLanes + 1) * kFbLanes * 2 + 16 + ((size_t)(maxInLen + 2) * 2 * sizeof(short) + 7) / 8 + 1;
}

struct FbArgs {
  int P;             // pLen.size()
  int maxDistance;   // envelope half-width: P, or 0 with --strict-guides (fwdback.cpp:17)
  int rowCap;        // widest envelope row in the batch
  double noGap, delOpen, delExtend, delEnd, tanDup;
  double sub[16];
  double len[kFbMaxLen];
};
