// Forward-backward E-step of the mutator pair-HMM as a wavefront per alignment pair: the banded Forward and Backward
// matrices (reference src/fwdback.cpp:43-116) and the posterior counts (fwdback.cpp:154-188, fwdback.h:92-112).
// (The file keeps its round-2 name: the first version held a block of the matrices in LDS.)
//
// W lanes work on one pair as a systolic wavefront.  Lane l owns the rows ip = l, l + W, l + 2W ...; at step a it computes the cell
// (ip, a - ip) if that lies in the row's envelope [lo(ip), hi(ip)].  A cell needs (ip-1, op-1) and (ip-1, op) -- what the lane
// below it computed two steps and one step ago: they come over by a lane shuffle, not through memory -- and (ip, op-1), its own
// previous step, kept in registers.  A lane must have left its row before its next one comes up: row ip is worked on at the steps
// ip + lo(ip) .. ip + hi(ip), row ip + W from step ip + W + lo(ip + W) on, so what is needed is hi(ip) - lo(ip + W) < W for every
// row.  A row of at most W cells always meets that (round 3: W = 16 or 32 = the widest row served); but the rows of an alignment
// run down a DIAGONAL -- lo(ip + W) is about lo(ip) + W --, so half as many lanes do (W = 8 for rows of up to RW = 16 cells, W = 16
// for up to 32: the host checks the inequality per pair, fwdback_runtime.hip): a lane then works 13 steps of 16 instead of 13 of
// 32, and a wave carries twice the pairs for the same instructions.
//
//   pass 1   Forward over all rows; every finished cell (S, D and the P duplication lanes) goes to the slot's scratch in HBM
//            (330 KB for a 256-nt pair: bandwidth this chip has to spare -- 0.8 TB/s at 1.4 * 10^6 pairs/s -- bought for not
//            evaluating any Forward cell twice; the round-3 version before this one recomputed blocks of W rows from
//            checkpoints and was 1.5 x slower).
//   pass 2   Backward, one continuous wavefront up the rows: the row below comes over by shuffle from the lane above; the
//            Forward cells a Backward cell needs (its own, the duplication lanes of the one to its left, two of the row above)
//            are loaded at the top of its step, long before the counts use them; every finished Backward cell adds its seven
//            posterior terms (fwdback.h:92-112) to the pair's counts.
//
// LDS per pair: the envelope bounds (4 B per input position) and the substitution counts, so what a CU holds is bounded by
// registers: 168 per lane without spills = 12 waves = 48 pairs per CU (round 2: 6 pairs on 3 half-filled waves, 26 KB of LDS
// each).  A work-group is ONE wave (64 / W pairs) and walks the list of pairs with a stride of the grid; nothing in it needs a
// work-group barrier.
//
// The arithmetic of a cell is the reference's, operation for operation (lse() with the reference's 100 001-entry
// table, uploaded once per handle and L2 resident; the two divisions by the table step are formed with a reciprocal and one
// fused correction, which gives the correctly rounded quotient -- checked against the division on 4 * 10^8 arguments):
// per-pair log-likelihoods are bit-identical to the CPU oracle's.  Counts are sums of exp() terms added in another order
// (per lane, then over the lanes in a fixed tree), so they agree to ~1e-12 relative and are reproducible run to run.
//
// What bounds it: the instruction stream of a cell (9 table-interpolated log-sum-exps forward, the same plus 16 fp64 exp
// backward) -- fp64 vector issue, not bytes: bench.py --config 4 reports the share of the chip's vector issue slots the
// kernel's instructions take (roofline.issue).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fwdback_device.h"

namespace {

// DNAS_FB_WHATIF (timing experiments only, never in the shipped library: the results are wrong): 1 = the counts' exp() replaced by a
// single-precision one, 2 = no counts at all, 3 = the log-sum-exp table not read (a constant instead), 4 = the bases not loaded
// (the substitution scores indexed by the cell's coordinates), 5 = ... and the score table in LDS not read either
#ifndef DNAS_FB_WHATIF
#define DNAS_FB_WHATIF 0
#endif
#if DNAS_FB_WHATIF == 1
#define exp(x) ((double)__expf((float)(x)))
#elif DNAS_FB_WHATIF == 2
#define exp(x) (0.0 * (x))
#endif

constexpr double kNegInf = -__builtin_huge_val();
// (kMaxP, a template parameter below: the duplication lanes the kernel keeps in registers -- 6 for the CLI's default model, 8 at most)

// x / .0001, correctly rounded, without the division sequence (Markstein: q0 = x * r, e = x - c * q0 exactly, q = q0 + e * r
// with r the correctly rounded reciprocal; checked against the division on 4 * 10^8 arguments).  Below 1e-280 the residual
// underflows and q may differ from x / c in its last bits -- which nobody sees: both uses below take such an x to the same
// result as the division does ((int) of a number below 1e-276 is 0; f0 + df * q with f0 >= log(1 + e^-10) = 4.5e-5 and
// df * q < 1e-280 is f0), so there is no special case (round 4 took one out: two comparisons and a branch per call, twice per
// log-sum-exp).
__device__ __forceinline__ double divStep(double x) {
  constexpr double c = .0001, r = 1.0 / .0001;
  const double q0 = x * r;
  const double e = __builtin_fma(-c, q0, x);
  return __builtin_fma(e, r, q0);
}

// The reference's log(1 + exp(-x)) (logsumexp.h:38-54): 0 for x >= 10 (and for the +inf / NaN that a difference of -inf's
// gives), else the table value at n = (int)(x / .0001) interpolated to x.  Written without branches: a lane whose x is out
// of range looks up entry 0 and throws the result away.  The table is read the buffer way (base in scalar registers, one
// 16-byte load for the two neighbouring entries: no 64-bit address arithmetic per look-up).
typedef __amdgpu_buffer_rsrc_t fb_rsrc_t;
typedef unsigned fb_u32x4 __attribute__((ext_vector_type(4)));
typedef double fb_dbl2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double lse_unary(fb_rsrc_t tab, double x) {
  const bool in = x < 10.;                        // false for NaN and +inf as well (logsumexp.h:41-42); x is never negative here
  const double xs = in ? x : 0.;
  const int n = (int)divStep(xs);                 // (int)(x / .0001)
  const double dx = xs - (n * .0001);
#if DNAS_FB_WHATIF == 3
  fb_dbl2 f; f.x = 0.5 + n * 1e-9; f.y = 0.25;
#else
  const fb_dbl2 f = __builtin_bit_cast(fb_dbl2, __builtin_amdgcn_raw_buffer_load_b128(tab, n * 8, 0, 0));   // tab[n], tab[n + 1]
#endif
  const double df = f.y - f.x;
  const double v = f.x + df * divStep(dx);        // f0 + df * (dx / .0001)
  return in ? v : 0.;
}

// log_sum_exp (logsumexp.h:56-74): max + f(|a - b|).  The reference takes a == b apart to avoid (-inf) - (-inf); here that
// difference is NaN, for which f is 0 like for every x >= 10, so the three cases fold into max / min / one subtraction
// (b - a and a - b are the same number as max - min).
__device__ __forceinline__ double lse(fb_rsrc_t tab, double a, double b) {
  double mx, mn;       // (the instructions themselves: no NaN comes in, so the canonicalising copies around fmax / fmin are waste)
  asm("v_max_f64 %0, %1, %2" : "=v"(mx) : "v"(a), "v"(b));
  asm("v_min_f64 %0, %1, %2" : "=v"(mn) : "v"(a), "v"(b));
  return mx + lse_unary(tab, mx - mn);
}

// One wave = 64 / W pairs.  pairList[i] = index of the pair in the database.  Scratch of a wave: the Forward cells of its pairs,
// [S, D, T[0..7]][step][64 lanes] -- a cell is filed under the STEP it was computed in (a - first step of its pair) and the lane
// that computed it, so every store and every load of a wave goes to consecutive addresses (round 3 filed it under (row, column):
// 64 lanes, 64 cache lines per access); the Backward pass meets cell (ip, op) in the same step a = ip + op and the same lane, its
// left neighbour one step earlier, the two cells of the row above one and two steps earlier in the lane below.
// fbOnchipWaveDoubles(maxSteps) doubles per wave.
template <int W, int RW, int kMaxP>
__device__ __forceinline__ void fwdback_onchip_body(const FbArgs& a, const int8_t* __restrict__ inSeqs, const int64_t* __restrict__ inOff,
                                                    const int8_t* __restrict__ outSeqs, const int64_t* __restrict__ outOff,
                                                    const int32_t* __restrict__ cmIn, const int64_t* __restrict__ cmInOff,
                                                    const int32_t* __restrict__ cmOut, const int64_t* __restrict__ cmOutOff,
                                                    const double* __restrict__ lseTab, const int64_t* __restrict__ pairList, int64_t nList,
                                                    double* __restrict__ pairCounts, double* __restrict__ pairLL, int maxInLen,
                                                    unsigned long long* __restrict__ lseOps, double* __restrict__ scratch, int maxSteps) {
  extern __shared__ double fbLds[];
  constexpr int PPG = 64 / W;                                       // pairs per wave
  const int g = threadIdx.x / W, l = threadIdx.x % W;
  const int P = a.P, Dm = a.maxDistance;
  // LDS of this pair
  double* const base = fbLds + (size_t)g * fbOnchipPairDoubles(W, maxInLen);
  double* const SUBC = base;                                         // [16] substitution counts
  double* const SUBS_ = SUBC + 16;                                   // [16] the substitution scores
  double* const LENS_ = SUBS_ + 16;                                  // [8]
  double* const ACC = LENS_ + 8;                                     // [16] the pair's other counts: c0..c4, then the length counts
  short* const LO = reinterpret_cast<short*>(ACC + 16);              // [maxInLen + 2]
  short* const HI = LO + (maxInLen + 2);
  // scratch of this slot: every Forward cell of the pair
  const size_t compStride = (size_t)maxSteps * 64;                   // doubles between two components of a cell
  double* const FW = scratch + (size_t)blockIdx.x * fbOnchipWaveDoubles(maxSteps) + threadIdx.x;   // + (component * maxSteps + step) * 64
  const int belowT = (int)threadIdx.x - l + (l + W - 1) % W;         // the lane that owns the row above this lane's row
  for (int i = l; i < 16; i += W) SUBS_[i] = a.sub[i];
  for (int i = l; i < kMaxP; i += W) LENS_[i] = a.len[i];
  unsigned long long nLse = 0;
  const fb_rsrc_t lseRsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(lseTab), 0, 100001 * 8, 0x00020000);
#define LSE(x, y) (++nLse, lse(lseRsrc, (x), (y)))
#if DNAS_FB_WHATIF == 4      /* the bases not loaded */
#define SUBS(i, o) SUBS_[((i) & 3) * 4 + ((o) & 3)]
#define DUPS(i, o, k) SUBS_[(((i) - (k)) & 3) * 4 + ((o) & 3)]
#elif DNAS_FB_WHATIF == 5    /* ... and the score table not read either */
#define SUBS(i, o) (-0.01 * (((i) + (o)) & 3))
#define DUPS(i, o, k) (-0.01 * (((i) + (o) + (k)) & 3))
#else
#define SUBS(i, o) SUBS_[in[(i) - 1] * 4 + out[(o) - 1]]                 /* cellSubScore, fwdback.h:65-67 */
#define DUPS(i, o, k) SUBS_[in[(i) - 1 - (k)] * 4 + out[(o) - 1]]        /* cellTanDupScore, fwdback.h:69-71 */
#endif
#define WAVE_SYNC() __builtin_amdgcn_wave_barrier()                      /* LDS traffic of one wave is in order: a compiler fence */
#define FROM_LANE(v, src) __shfl((v), (src), W)

  for (int64_t item0 = (int64_t)blockIdx.x * PPG; item0 < nList; item0 += (int64_t)gridDim.x * PPG) {
  const int64_t item = item0 + g;
  const bool live = item < nList;
  const int64_t pair = live ? pairList[item] : 0;
  const int8_t* in = inSeqs + inOff[pair];
  const int8_t* out = outSeqs + outOff[pair];
  const int I = live ? (int)(inOff[pair + 1] - inOff[pair]) : -1;
  const int O = live ? (int)(outOff[pair + 1] - outOff[pair]) : -1;
  const int32_t* ci = cmIn + cmInOff[pair];
  const int32_t* co = cmOut + cmOutOff[pair];

  // ---- the envelope of every row (alignpath.h:48-53): op in [lo, hi] <=> |cm(ip) - cm(op)| <= maxDistance; cm is
  // non-decreasing along both sequences, so lo and hi are two binary searches per row
  WAVE_SYNC();
  for (int i = l; i < 16; i += W) { SUBC[i] = 0; ACC[i] = 0; }
  for (int ip = l; ip <= I; ip += W) {
    const int lowKey = ci[ip] - Dm, highKey = ci[ip] + Dm;
    int x = 0, y = O + 1;
    while (x < y) { const int mid = (x + y) >> 1; if (co[mid] < lowKey) x = mid + 1; else y = mid; }
    const int lo = x;
    y = O + 1;
    while (x < y) { const int mid = (x + y) >> 1; if (co[mid] <= highKey) x = mid + 1; else y = mid; }
    LO[ip] = (short)lo;
    HI[ip] = (short)(x - 1);
  }
  WAVE_SYNC();

  const int aStart = live ? LO[0] : 0;                             // the pair's first step
  double ll = kNegInf;
  double T[kMaxP];                                                 // duplication lanes of this lane's previous cell
  double leftS = kNegInf;                                          // (Backward: S of the previous cell of the row)
#pragma unroll
  for (int k = 0; k < kMaxP; ++k) T[k] = kNegInf;

  // One Forward cell.  upS/upD/diagS: row ip-1 (or -inf outside its envelope); T[]: cell (ip, op-1) when hasIns.
  auto forwardCell = [&](int ip, int op, bool hasIns, double diagS, bool diagIn, double upS, double upD, bool upIn, double& s, double& d) {
    const int mdl = ip < P ? ip : P;                               // maxDupLenAt, fwdback.h:59
    s = (ip == 0 && op == 0) ? 0. : kNegInf;
    d = kNegInf;
    if (ip > 0 && op > 0) {
      if (diagIn) s = diagS + a.noGap + SUBS(ip, op);
      if (hasIns) s = LSE(s, T[0] + DUPS(ip, op, 0));
    }
    if (ip > 0 && upIn) d = LSE(upS + a.delOpen, upD + a.delExtend);
    s = LSE(s, d + a.delEnd);
    double tn[kMaxP];
#pragma unroll
    for (int k = 0; k < kMaxP; ++k) {
      double t = kNegInf;
      if (k < P) {
        if (hasIns && k < mdl - 1) t = T[k + 1] + DUPS(ip, op, k + 1);
        if (k < mdl) t = LSE(t, s + a.tanDup + LENS_[k]);
      }
      tn[k] = t;
    }
#pragma unroll
    for (int k = 0; k < kMaxP; ++k) T[k] = tn[k];
  };
  // is (row r, column op) inside the envelope
  auto inRow = [&](int r, int op) -> bool { return r >= 0 && r <= I && op >= LO[r] && op <= HI[r]; };

  // ---------------- pass 1: Forward (fwdback.cpp:43-78), every cell to the slot's scratch.  What the lane below computed one and
  // two steps ago is this lane's (ip-1, op) and (ip-1, op-1): the steps are global (a = ip + op), every cell has its one step
  {
    int ip = l;                                                    // this lane's row
    int aNow = aStart;                                             // global step = ip + op
    const int aLast = live ? I + HI[I] : -1;
    double curS = kNegInf, curD = kNegInf, prevS = kNegInf;        // this lane's cell of the step before, S of the one before that
    for (; __any(live && aNow <= aLast) && aNow <= 2 * 65536; ++aNow) {
      const int below = (l + W - 1) % W;                           // (lane 0 follows lane W-1: its row W*b comes after row W*b - 1)
      const double uS = FROM_LANE(curS, below), uD = FROM_LANE(curD, below), dS = FROM_LANE(prevS, below);
      prevS = curS;
      if (live && ip <= I) {
        const int lo = LO[ip], hi = HI[ip];
        const int op = aNow - ip;
        if (op >= lo && op <= hi) {
          const bool hasIns = ip > 0 && op > 0 && op - 1 >= lo;
          const bool dIn = ip > 0 && op > 0 && inRow(ip - 1, op - 1), uIn = ip > 0 && inRow(ip - 1, op);
          double s, d;
          forwardCell(ip, op, hasIns, dS, dIn, uS, uD, uIn, s, d);
          curS = s; curD = d;
          double* cell = FW + (size_t)(aNow - aStart) * 64;
          cell[0] = s;
          cell[compStride] = d;
#pragma unroll
          for (int k = 0; k < kMaxP; ++k) if (k < P) cell[(size_t)(2 + k) * compStride] = T[k];
          if (ip == I && op == O) ll = s;                          // loglike = sCell(inLen, outLen), fwdback.cpp:76
          if (op == hi) ip += W;                                   // row done: on to this lane's next row
        } else if (op > hi) {
          ip += W;                                                 // (an empty row)
        }
      }
    }
  }
  // every lane of the pair needs the log-likelihood; the lane that owned (I, O) has it
  for (int offs = W / 2; offs > 0; offs >>= 1) { const double o2 = __shfl_xor(ll, offs, W); ll = ll < o2 ? o2 : ll; }
  if (live && l == 0) pairLL[pair] = ll;
  __threadfence_block();                                           // the Forward cells are read back by this lane and the one above it
  WAVE_SYNC();

  // ---------------- pass 2: Backward (fwdback.cpp:80-116), anti-diagonals downwards, with the counts.  The row below (ip+1)
  // belongs to the lane above: its cells (ip+1, op+1) and (ip+1, op) were finished two steps and one step ago
  // (the counts are added up in LDS, like the substitution counts: as registers they were 22 more of a cell that spills as it is)
  {
    int ip = (live && l <= I) ? l + ((I - l) / W) * W : -1;        // this lane's last row
    const int aFirst = live ? I + HI[I] : -1, aEnd = live ? LO[0] : 0;
    double bcurS = kNegInf, bcurD = kNegInf, bprevS = kNegInf;     // Backward cell of the step before, S of the one before that
    for (int aNow = aFirst; __any(live && aNow >= aEnd); --aNow) {
      const int above = (l + 1) % W;                               // (lane W-1 follows lane 0: its row W*b - 1 comes after row W*b)
      const double nS = FROM_LANE(bprevS, above), nD = FROM_LANE(bcurD, above);   // (ip+1, op+1).S and (ip+1, op).D
      bprevS = bcurS;
      if (live && ip >= 0) {
        const int lo = LO[ip], hi = HI[ip];
        const int op = aNow - ip;
        if (op >= lo && op <= hi) {
          const int mdl = ip < P ? ip : P;
          const int nlo = ip < I ? LO[ip + 1] : 0, nhi = ip < I ? HI[ip + 1] : -1;
          // the Forward cells this cell's counts look at -- its own, the duplication lanes of the one to its left, two of the row
          // above -- are on their way while the Backward cell is computed
          const int t = aNow - aStart;                             // this cell's step: Forward cell (ip, op) sits at (t, this lane)
          const double* fc = FW + (size_t)t * 64;
          double ft[kMaxP];
          const bool fIns = op - 1 >= lo;
          {
            const double* tb = fc - (fIns ? 64 : 0) + 2 * compStride;        // (ip, op-1): one step earlier, this lane
#pragma unroll
            for (int k = 0; k < kMaxP; ++k) ft[k] = (k < P && fIns) ? tb[(size_t)k * compStride] : kNegInf;
          }
          const bool upDg = ip > 0 && op > 0 && inRow(ip - 1, op - 1), upU = ip > 0 && inRow(ip - 1, op);
          const double* upF = fc + (belowT - (int)threadIdx.x);              // the lane below, same step
          const double fUpDiagS = upDg ? upF[-2 * 64] : kNegInf;             // (ip-1, op-1): two steps earlier
          const double fUpS = upU ? upF[-64] : kNegInf, fUpD = upU ? upF[-64 + (long)compStride] : kNegInf;   // (ip-1, op): one step earlier
          const double fOwnS = fc[0], fOwnD = fc[compStride];

          double s = (ip == I && op == O) ? 0. : kNegInf, d = kNegInf;
          const bool hasIns = op < O && ip > 0 && op + 1 <= hi;    // (ip, op+1) in range: T[] and leftS are that cell's
          const bool sIn = op < O && ip < I && op + 1 >= nlo && op + 1 <= nhi, dInB = ip < I && op >= nlo && op <= nhi;
          if (sIn) s = a.noGap + SUBS(ip + 1, op + 1) + nS;
          double bt[kMaxP];
#pragma unroll
          for (int k = 0; k < kMaxP; ++k) {
            double t = kNegInf;
            if (k < P && hasIns && k < mdl) t = (k == 0) ? DUPS(ip, op + 1, 0) + leftS : DUPS(ip, op + 1, k) + T[k - 1 < 0 ? 0 : k - 1];
            bt[k] = t;
          }
          if (dInB) {
            s = LSE(s, a.delOpen + nD);
            d = a.delExtend + nD;
          }
#pragma unroll
          for (int k = 0; k < kMaxP; ++k) if (k < mdl) s = LSE(s, bt[k] + a.tanDup + LENS_[k]);
          d = LSE(d, s + a.delEnd);
          bcurS = s; bcurD = d;
#pragma unroll
          for (int k = 0; k < kMaxP; ++k) T[k] = bt[k];
          leftS = s;

          // ---- posterior counts at (ip, op) (fwdback.h:92-112)
          if (ip > 0 && op > 0) {
            const double cS = exp(fUpDiagS + a.noGap + SUBS(ip, op) + s - ll);                     // pS2S
            atomicAdd(&ACC[2], cS);
            double subAdd = cS;
#pragma unroll
            for (int k = 0; k < kMaxP - 1; ++k)
              if (k < mdl - 1)
                atomicAdd(&SUBC[in[ip - 1 - (k + 1)] * 4 + out[op - 1]], exp(ft[k + 1] + DUPS(ip, op, k + 1) + bt[k] - ll));   // pT2T
            subAdd += exp(ft[0] + DUPS(ip, op, 0) + s - ll);                                       // pT2S
            atomicAdd(&SUBC[in[ip - 1] * 4 + out[op - 1]], subAdd);
          }
          if (ip > 0) {
            atomicAdd(&ACC[0], exp(fUpS + a.delOpen + d - ll));                                    // pS2D
            atomicAdd(&ACC[3], exp(fUpD + a.delExtend + d - ll));                                  // pD2D
          }
          atomicAdd(&ACC[4], exp(fOwnD + a.delEnd + s - ll));                                      // pD2S
          const double fs = fOwnS;
          double cTsum = 0;
#pragma unroll
          for (int k = 0; k < kMaxP; ++k)
            if (k < mdl) {
              const double cT = exp(fs + a.tanDup + LENS_[k] + bt[k] - ll);                        // pS2T
              cTsum += cT;
              atomicAdd(&ACC[5 + k], cT);
            }
          atomicAdd(&ACC[1], cTsum);
          if (op == lo) ip -= W;                                   // row done: on to this lane's next row up
        } else if (op < lo) {
          ip -= W;                                                 // (an empty row)
        }
      }
    }
  }
  WAVE_SYNC();

  // ---- the pair's counts
  if (live) {
    double* pc = pairCounts + (size_t)pair * (21 + P);
    if (l == 0) {
      pc[0] = ACC[0]; pc[1] = ACC[1]; pc[2] = ACC[2]; pc[3] = ACC[3]; pc[4] = ACC[4];
#pragma unroll
      for (int k = 0; k < kMaxP; ++k) if (k < P) pc[21 + k] = ACC[5 + k];
    }
    for (int i = l; i < 16; i += W) pc[5 + i] = SUBC[i];
  }
  }   // pairs of this slot
  if (lseOps) {
    unsigned long long tot = nLse;
    for (int offs = 32; offs > 0; offs >>= 1) tot += __shfl_xor(tot, offs, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(lseOps, tot);
  }
#undef LSE
#undef SUBS
#undef DUPS
}

}  // namespace

#ifndef DNAS_FB_MIN_WAVES
#define DNAS_FB_MIN_WAVES 3      // waves per SIMD the register allocation leaves room for: the cell lives in ~235 registers; at 168 it spills 68 of them, and 12 waves per CU still run 1.34 x faster than 8 without spills (the kernel waits on memory)
#endif
#define FB_KERNEL(name, W, RW, MP)                                                                                                \
  extern "C" __global__ void __launch_bounds__(64, DNAS_FB_MIN_WAVES)                                                             \
  name(FbArgs a, const int8_t* __restrict__ inSeqs, const int64_t* __restrict__ inOff, const int8_t* __restrict__ outSeqs,        \
       const int64_t* __restrict__ outOff, const int32_t* __restrict__ cmIn, const int64_t* __restrict__ cmInOff,                 \
       const int32_t* __restrict__ cmOut, const int64_t* __restrict__ cmOutOff, const double* __restrict__ lseTab,                \
       const int64_t* __restrict__ pairList, int64_t nList, double* __restrict__ pairCounts, double* __restrict__ pairLL,         \
       int maxInLen, unsigned long long* __restrict__ lseOps, double* __restrict__ scratch, int maxSteps) {                       \
    fwdback_onchip_body<W, RW, MP>(a, inSeqs, inOff, outSeqs, outOff, cmIn, cmInOff, cmOut, cmOutOff, lseTab, pairList, nList, pairCounts, \
                           pairLL, maxInLen, lseOps, scratch, maxSteps);                                                          \
  }
// name: lanes per pair x cells per row served; p6: up to 6 duplication lengths (the CLI's default model) in registers
FB_KERNEL(fwdback_onchip8x16_kernel, 8, 16, 8)
FB_KERNEL(fwdback_onchip16x16_kernel, 16, 16, 8)
FB_KERNEL(fwdback_onchip16x32_kernel, 16, 32, 8)
FB_KERNEL(fwdback_onchip32x32_kernel, 32, 32, 8)
FB_KERNEL(fwdback_onchip8x16p6_kernel, 8, 16, 6)
FB_KERNEL(fwdback_onchip16x16p6_kernel, 16, 16, 6)
FB_KERNEL(fwdback_onchip16x32p6_kernel, 16, 32, 6)
FB_KERNEL(fwdback_onchip32x32p6_kernel, 32, 32, 6)
