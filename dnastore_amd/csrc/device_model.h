// Kernel-argument view of the flattened model (device pointers), shared by the
// kernels (viterbi_kernels.hip) and the launcher (runtime.hip).
#pragma once
#include <stdint.h>

constexpr int kFillThreads = 1024;  // work-group size of viterbi_fill_kernel (16 waves: one work-group fills a CU)
constexpr int kTraceThreads = 128;  // independent reads per block of the thread-per-read traceback (see runtime.hip; launch bounds: 1024)
constexpr int kMaxLen = 32;         // pLen entries (dnas_mutator_params.p_len)

struct DevModel {
  int N, Npad, D, P, local;
  int storedLanes;        // lanes per column in HBM: D+2 (tier B: S, D, T1..TD) or 2 (tier A: S, D; T recomputed)
  // in-edges per destination, reference enumeration order (traceback tie-break order)
  const int32_t* einPtr; const int32_t* einSrc; const double* einScore; const uint8_t* einBase; const uint8_t* einIn;
  const int32_t* ninPtr; const int32_t* ninSrc; const double* ninScore; const uint8_t* ninIn;
  // out-edges per source (dirty marking)
  const int32_t* eoutPtr; const int32_t* eoutDst;
  const int32_t* noutPtr; const int32_t* noutDst;
  const int32_t* slotOf;  // tier A: state -> lattice slot (nullptr: identity)
  const int32_t* einSlot; const int32_t* ninSlot;   // lattice slot of every in-edge's source (saves the traceback a dependent load)
  const uint8_t* mdl;   // [N]
  const uint8_t* ctx;   // [N*D]
  double noGap, delOpen, delExtend, delEnd, tanDup;
  double sub[16];
  double len[kMaxLen];
};

// Bounded-memory decode: a traceback that left the lattice segment it was given, to be picked up over the segment before
// (viterbi_traceback_wave_kernel).
struct TracebackWalk {
  int state, pos, mut;
  int phase;            // 0: not started, 1: parked, 2: finished
  double curCell;
  long long n, nEv;     // symbols / events written so far
};
