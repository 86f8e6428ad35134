// Forward-backward E-step of the mutator pair-HMM, ON CHIP: the banded Forward and Backward matrices of a pair
// (reference src/fwdback.cpp:43-116) and its posterior counts (fwdback.cpp:154-188, fwdback.h:92-112) never leave
// the CU.  HBM sees the two sequences, the guide columns and 21+P counts + one log-likelihood per pair.
//
// Sixteen lanes work on one pair as a systolic wavefront.  Lane l owns the rows ip = 16b + l of the current block b
// of sixteen rows; at step a it computes the cell (ip, a - ip) if that lies in the row's envelope [lo(ip), hi(ip)].
// A cell needs (ip-1, op-1), (ip-1, op) -- the row of lane l-1, one and two steps ago, read back from LDS -- and
// (ip, op-1), its own previous step, kept in registers.  A row is at most sixteen cells wide, so a lane has left
// its row before the next block hands it another one.
//
//   pass 1   Forward over all rows; only the S and D lanes of the last row of every block are kept (a "checkpoint",
//            2 * 16 doubles per block): a row is a function of the S and D lanes of the row above it, because the
//            duplication lanes T only run along a row (fwdback.cpp:57-60).
//   pass 2   blocks from the last to the first: the block's Forward rows are recomputed from the checkpoint above it
//            and kept whole (16 rows x 16 cells x P+2 lanes, 16 KB); then the Backward wavefront runs up the block,
//            two S/D rows of it alive at a time, and every finished Backward cell adds its seven posterior terms
//            (fwdback.h:92-112) to the pair's counts.
//
// The arithmetic of a cell is the reference's, operation for operation (lse() with the reference's 100 001-entry
// table, uploaded once per handle and L2 resident): per-pair log-likelihoods are bit-identical to the CPU oracle's.
// Counts are sums of exp() terms added in another order (per lane, then over the sixteen lanes in a fixed tree),
// so they agree to ~1e-12 relative and are reproducible run to run.
//
// LDS per pair ~26 KB -> six pairs per CU: the kernel is bound by the latency chain of a cell (four dependent table
// look-ups), not by bytes; what is reported is pairs/s, nt/s and log-sum-exp operations per second.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fwdback_device.h"

namespace {

constexpr double kNegInf = -__builtin_huge_val();
constexpr int kL = kFbLanes;            // lanes per pair = rows per block = widest envelope row served
constexpr int kRS = kL * 8 + 2;         // doubles per Forward row of the block (padded against bank conflicts)
constexpr int kMaxP = 8;                // duplication lanes this kernel keeps in registers

__device__ __forceinline__ double lse_unary(const double* __restrict__ tab, double x) {
  if (x >= 10. || x != x || x == __builtin_huge_val()) return 0;   // logsumexp.h:41-42
  if (x < 0) return -x;
  const int n = (int)(x / .0001);
  const double dx = x - (n * .0001);
  const double f0 = tab[n], f1 = tab[n + 1];
  const double df = f1 - f0;
  return f0 + df * (dx / .0001);
}

__device__ __forceinline__ double lse(const double* __restrict__ tab, double a, double b) {   // logsumexp.h:56-74
  double mx, diff;
  if (a == b) { mx = a; diff = 0; }
  else if (a < b) { mx = b; diff = b - a; }
  else { mx = a; diff = a - b; }
  return mx + lse_unary(tab, diff);
}

}  // namespace

// One work-group = kFbPairsPerGroup pairs (16 lanes each).  pairList[i] = index of the pair in the database.
// Dynamic LDS: kFbPairsPerGroup * pairDoubles doubles, pairDoubles = fbOnchipPairDoubles(maxInLen).
extern "C" __global__ void __launch_bounds__(kFbLanes * kFbPairsPerGroup)
fwdback_onchip_kernel(FbArgs a, const int8_t* __restrict__ inSeqs, const int64_t* __restrict__ inOff,
                      const int8_t* __restrict__ outSeqs, const int64_t* __restrict__ outOff,
                      const int32_t* __restrict__ cmIn, const int64_t* __restrict__ cmInOff,
                      const int32_t* __restrict__ cmOut, const int64_t* __restrict__ cmOutOff,
                      const double* __restrict__ lseTab, const int64_t* __restrict__ pairList, int64_t nList,
                      double* __restrict__ pairCounts, double* __restrict__ pairLL, int maxInLen,
                      unsigned long long* __restrict__ lseOps) {
  extern __shared__ double fbLds[];
  __shared__ double subS[16], lenS[kMaxP];
  if (threadIdx.x < 16) subS[threadIdx.x] = a.sub[threadIdx.x];
  if (threadIdx.x < kMaxP) lenS[threadIdx.x] = a.len[threadIdx.x];
  const int g = threadIdx.x / kL, l = threadIdx.x % kL;
  const int64_t item = (int64_t)blockIdx.x * kFbPairsPerGroup + g;
  const bool live = item < nList;
  const int64_t pair = live ? pairList[item] : 0;
  const int P = a.P, Dm = a.maxDistance;
  const int nCk = (maxInLen + 1 + kL - 1) / kL;                    // blocks (and checkpoints) of the longest pair
  // LDS of this pair
  double* const base = fbLds + (size_t)g * fbOnchipPairDoubles(maxInLen);
  double* const FB = base;                                           // [kL rows][kRS]: (cell j)*8 + lane
  double* const CK = FB + kL * kRS;                                  // [nCk][kL cells][2]
  double* const BR = CK + (size_t)nCk * kL * 2;                      // [kL + 1 rows][kL cells][2]
  double* const SUBC = BR + (kL + 1) * kL * 2;                       // [16] substitution counts
  short* const LO = reinterpret_cast<short*>(SUBC + 16);             // [maxInLen + 2]
  short* const HI = LO + (maxInLen + 2);

  const int8_t* in = inSeqs + inOff[pair];
  const int8_t* out = outSeqs + outOff[pair];
  const int I = live ? (int)(inOff[pair + 1] - inOff[pair]) : -1;
  const int O = live ? (int)(outOff[pair + 1] - outOff[pair]) : -1;
  const int32_t* ci = cmIn + cmInOff[pair];
  const int32_t* co = cmOut + cmOutOff[pair];
  unsigned long long nLse = 0;
#define LSE(x, y) (++nLse, lse(lseTab, (x), (y)))
#define SUBS(i, o) subS[in[(i) - 1] * 4 + out[(o) - 1]]                 /* cellSubScore, fwdback.h:65-67 */
#define DUPS(i, o, k) subS[in[(i) - 1 - (k)] * 4 + out[(o) - 1]]        /* cellTanDupScore, fwdback.h:69-71 */

  // ---- the envelope of every row (alignpath.h:48-53): op in [lo, hi] <=> |cm(ip) - cm(op)| <= maxDistance; cm is
  // non-decreasing along both sequences, so lo and hi are two binary searches per row
  if (l < 16) SUBC[l] = 0;
  for (int ip = l; ip <= I; ip += kL) {
    const int lowKey = ci[ip] - Dm, highKey = ci[ip] + Dm;
    int x = 0, y = O + 1;
    while (x < y) { const int mid = (x + y) >> 1; if (co[mid] < lowKey) x = mid + 1; else y = mid; }
    const int lo = x;
    y = O + 1;
    while (x < y) { const int mid = (x + y) >> 1; if (co[mid] <= highKey) x = mid + 1; else y = mid; }
    LO[ip] = (short)lo;
    HI[ip] = (short)(x - 1);
  }
  __syncthreads();

  const int nBlocks = live ? (I + 1 + kL - 1) / kL : 0;
  double ll = kNegInf;
  double T[kMaxP];                                                 // duplication lanes of this lane's previous cell
  double leftS = kNegInf;                                          // (Backward: S of the previous cell of the row)

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
        if (k < mdl) t = LSE(t, s + a.tanDup + lenS[k]);
      }
      tn[k] = t;
    }
#pragma unroll
    for (int k = 0; k < kMaxP; ++k) T[k] = tn[k];
  };

  // The S and D lanes of row r-1 for a cell of row r at column op: from `rowBuf` (cells of 8 doubles, the Forward
  // block) or from a checkpoint (cells of 2 doubles).
  auto upRow = [&](const double* rowBuf, int stride, int rUp, int op, double& s, double& d) -> bool {
    const int lo = LO[rUp], hi = HI[rUp];
    if (op < lo || op > hi) { s = d = kNegInf; return false; }
    s = rowBuf[(op - lo) * stride];
    d = rowBuf[(op - lo) * stride + 1];
    return true;
  };

  // ---------------- pass 1: Forward, checkpoints only
  {
    int ip = l;                                                    // this lane's row
    int aNow = live ? LO[0] : 0;                                   // global step = ip + op
    const int aLast = live ? I + HI[I] : -1;
    // steps are global over all rows: row ip is worked at steps ip + lo(ip) .. ip + hi(ip)
    for (; __any(live && aNow <= aLast) && aNow <= 2 * 65536; ++aNow) {
      if (live && ip <= I) {
        const int lo = LO[ip], hi = HI[ip];
        const int op = aNow - ip;
        if (op >= lo && op <= hi) {
          const bool hasIns = ip > 0 && op > 0 && op - 1 >= lo;
          double dS = kNegInf, dD, uS = kNegInf, uD = kNegInf;
          bool dIn = false, uIn = false;
          if (ip > 0) {
            const double* up = FB + ((ip - 1) % kL) * kRS;
            if (op > 0) dIn = upRow(up, 8, ip - 1, op - 1, dS, dD);
            uIn = upRow(up, 8, ip - 1, op, uS, uD);
          }
          double s, d;
          forwardCell(ip, op, hasIns, dS, dIn, uS, uD, uIn, s, d);
          double* cell = FB + (ip % kL) * kRS + (op - lo) * 8;
          cell[0] = s;
          cell[1] = d;
          if (ip % kL == kL - 1 || ip == I) {                     // the block's last row is its checkpoint
            double* ck = CK + ((size_t)(ip / kL) * kL + (op - lo)) * 2;
            ck[0] = s;
            ck[1] = d;
          }
          if (ip == I && op == O) ll = s;                          // loglike = sCell(inLen, outLen), fwdback.cpp:76
          if (op == hi) ip += kL;                                  // row done: on to this lane's row of the next block
        } else if (op > hi) {
          ip += kL;                                                // (an empty row)
        }
      }
    }
  }
  // every lane of the pair needs the log-likelihood; the lane that owned (I, O) has it
  for (int offs = kL / 2; offs > 0; offs >>= 1) { const double o2 = __shfl_xor(ll, offs, kL); ll = ll < o2 ? o2 : ll; }
  if (live && l == 0) pairLL[pair] = ll;

  // ---------------- pass 2: per block, Forward again (kept whole), then Backward + counts
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, cl[kMaxP];
#pragma unroll
  for (int k = 0; k < kMaxP; ++k) cl[k] = 0;
  for (int b = nBlocks - 1; __any(b >= 0); --b) {
    const bool on = live && b >= 0;
    const int r0 = b * kL, rLast = on ? (r0 + kL - 1 < I ? r0 + kL - 1 : I) : -1;
    const int ip = r0 + l;
    const bool mine = on && ip <= rLast;
    const int lo = mine ? LO[ip] : 0, hi = mine ? HI[ip] : -1;
    // ---- Forward of the block
    {
      const int aFirst = on ? r0 + LO[r0] : 0, aEnd = on ? rLast + HI[rLast] : -1;
      for (int aNow = aFirst; __any(on && aNow <= aEnd); ++aNow) {
        const int op = aNow - ip;
        if (mine && op >= lo && op <= hi) {
          const bool hasIns = ip > 0 && op > 0 && op - 1 >= lo;
          double dS = kNegInf, dD, uS = kNegInf, uD = kNegInf;
          bool dIn = false, uIn = false;
          if (ip > 0) {
            const double* up = l == 0 ? CK + (size_t)(b - 1) * kL * 2 : FB + (l - 1) * kRS;
            const int stride = l == 0 ? 2 : 8;
            if (op > 0) dIn = upRow(up, stride, ip - 1, op - 1, dS, dD);
            uIn = upRow(up, stride, ip - 1, op, uS, uD);
          }
          double s, d;
          forwardCell(ip, op, hasIns, dS, dIn, uS, uD, uIn, s, d);
          double* cell = FB + l * kRS + (op - lo) * 8;
          cell[0] = s;
          cell[1] = d;
#pragma unroll
          for (int k = 0; k < kMaxP; ++k) if (k < P) cell[2 + k] = T[k];
        }
      }
    }
    // ---- Backward of the block (fwdback.cpp:80-116), anti-diagonals downwards, with the counts
    {
      const int aFirst = on ? rLast + HI[rLast] : -1, aEnd = on ? r0 + LO[r0] : 0;
      const int mdl = ip < P ? ip : P;
      const int nlo = (mine && ip < I) ? LO[ip + 1] : 0, nhi = (mine && ip < I) ? HI[ip + 1] : -1;
      const double* down = BR + (size_t)(l + 1) * kL * 2;          // row ip+1: the lane above, or the block above (row kL)
      double* mineB = BR + (size_t)l * kL * 2;
      const double* upF = l == 0 ? CK + (size_t)(b - 1) * kL * 2 : FB + (l - 1) * kRS;
      const int upStride = l == 0 ? 2 : 8;
      for (int aNow = aFirst; __any(on && aNow >= aEnd); --aNow) {
        const int op = aNow - ip;
        if (mine && op >= lo && op <= hi) {
          const int j = op - lo;
          double s = (ip == I && op == O) ? 0. : kNegInf, d = kNegInf;
          const bool hasIns = op < O && ip > 0 && op + 1 <= hi;    // (ip, op+1) in range: T[] and leftS are that cell's
          if (op < O && ip < I && op + 1 >= nlo && op + 1 <= nhi)
            s = a.noGap + SUBS(ip + 1, op + 1) + down[(op + 1 - nlo) * 2];
          double bt[kMaxP];
#pragma unroll
          for (int k = 0; k < kMaxP; ++k) {
            double t = kNegInf;
            if (k < P && hasIns && k < mdl) t = (k == 0) ? DUPS(ip, op + 1, 0) + leftS : DUPS(ip, op + 1, k) + T[k - 1 < 0 ? 0 : k - 1];
            bt[k] = t;
          }
          if (ip < I && op >= nlo && op <= nhi) {
            const double dd = down[(op - nlo) * 2 + 1];
            s = LSE(s, a.delOpen + dd);
            d = a.delExtend + dd;
          }
#pragma unroll
          for (int k = 0; k < kMaxP; ++k) if (k < mdl) s = LSE(s, bt[k] + a.tanDup + lenS[k]);
          d = LSE(d, s + a.delEnd);
          mineB[j * 2] = s;
          mineB[j * 2 + 1] = d;
#pragma unroll
          for (int k = 0; k < kMaxP; ++k) T[k] = bt[k];
          leftS = s;

          // ---- posterior counts at (ip, op) (fwdback.h:92-112)
          const double* fc = FB + l * kRS + j * 8;                 // Forward cell (ip, op); fc - 8: (ip, op-1)
          if (ip > 0 && op > 0) {
            double fS, fD;
            (void)upRow(upF, upStride, ip - 1, op - 1, fS, fD);
            const double cS = exp(fS + a.noGap + SUBS(ip, op) + s - ll);                           // pS2S
            c2 += cS;
            double subAdd = cS;
            const bool fIns = op - 1 >= lo;
#pragma unroll
            for (int k = 0; k < kMaxP - 1; ++k)
              if (k < mdl - 1) {
                const double ft = fIns ? fc[-8 + 2 + k + 1] : kNegInf;
                atomicAdd(&SUBC[in[ip - 1 - (k + 1)] * 4 + out[op - 1]], exp(ft + DUPS(ip, op, k + 1) + bt[k] - ll));   // pT2T
              }
            const double f0 = fIns ? fc[-8 + 2] : kNegInf;
            subAdd += exp(f0 + DUPS(ip, op, 0) + s - ll);                                          // pT2S
            atomicAdd(&SUBC[in[ip - 1] * 4 + out[op - 1]], subAdd);
          }
          if (ip > 0) {
            double uS, uD;
            (void)upRow(upF, upStride, ip - 1, op, uS, uD);
            c0 += exp(uS + a.delOpen + d - ll);                                                    // pS2D
            c3 += exp(uD + a.delExtend + d - ll);                                                  // pD2D
          }
          c4 += exp(fc[1] + a.delEnd + s - ll);                                                    // pD2S
          const double fs = fc[0];
#pragma unroll
          for (int k = 0; k < kMaxP; ++k)
            if (k < mdl) {
              const double cT = exp(fs + a.tanDup + lenS[k] + bt[k] - ll);                        // pS2T
              c1 += cT;
              cl[k] += cT;
            }
        }
      }
      // the first row of this block is the "row above" of the next (lower) block
      if (on && mine && l == 0)
        for (int j = 0; j <= hi - lo; ++j) { BR[(size_t)kL * kL * 2 + j * 2] = mineB[j * 2]; BR[(size_t)kL * kL * 2 + j * 2 + 1] = mineB[j * 2 + 1]; }
    }
  }
  __syncthreads();

  // ---- the pair's counts: lanes summed in a fixed tree
  auto sum16 = [&](double v) {
    for (int offs = kL / 2; offs > 0; offs >>= 1) v += __shfl_xor(v, offs, kL);
    return v;
  };
  c0 = sum16(c0); c1 = sum16(c1); c2 = sum16(c2); c3 = sum16(c3); c4 = sum16(c4);
#pragma unroll
  for (int k = 0; k < kMaxP; ++k) cl[k] = sum16(cl[k]);
  if (live) {
    double* pc = pairCounts + (size_t)pair * (21 + P);
    if (l == 0) {
      pc[0] = c0; pc[1] = c1; pc[2] = c2; pc[3] = c3; pc[4] = c4;
#pragma unroll
      for (int k = 0; k < kMaxP; ++k) if (k < P) pc[21 + k] = cl[k];
    }
    pc[5 + l] = SUBC[l];
  }
  // the substitution counts of pS2S belong to cell (in[ip-1], out[op-1]) as well: they were added through subAdd
  if (lseOps) {
    unsigned long long tot = nLse;
    for (int offs = 32; offs > 0; offs >>= 1) tot += __shfl_xor(tot, offs, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(lseOps, tot);
  }
}
