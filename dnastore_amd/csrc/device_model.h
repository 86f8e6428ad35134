// Kernel-argument view of the flattened model (device pointers), shared by the
// kernels (viterbi_kernels.hip) and the launcher (runtime.hip).
#pragma once
#include <stdint.h>

constexpr int kFillThreads = 1024;  // work-group size of viterbi_fill_kernel (16 waves: one work-group fills a CU)
constexpr int kTraceLanes = 16;     // reads per wave of the thread-per-read traceback (option tb_lanes): a step costs a wave the union of
                                    // what its lanes do -- five pairs of runs, ms per 10 000-read step: 16 reads per wave 118-149, 64 reads 147-221
constexpr int kTraceThreads = 128;  // independent reads per block of the thread-per-read traceback (see runtime.hip; launch bounds: 256)
constexpr int kMaxLen = 32;         // pLen entries (dnas_mutator_params.p_len)
constexpr int kRecEmit = 3, kRecNull = 1;   // in-edges a node record holds inline

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
  // Node records for the thread-per-read traceback (nullptr: none): everything a step needs to know about a state in one
  // 64-byte line -- word 0: emit in-edges (bits 0-3; 15 = more than kRecEmit: use the CSR arrays), null in-edges (bits 4-7;
  // 15 = more than kRecNull), mdl (bits 8-11), the left context's bases (bits 12-19, 2 each); word 1: lattice slot;
  // words 2+3i..4+3i: emit in-edge i = source state, its lattice slot, input symbol | base << 8 | score index << 16;
  // words 11..13: null in-edge 0 likewise; recScore[index] = the edge's score (the distinct values of a machine are few).
  const uint32_t* rec;       // [N][16]
  const double* recScore;    // [<= 256]
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
