// Forward-backward E-step of the mutator pair-HMM on gfx950: replaces FwdBackMatrix +
// counts() (reference src/fwdback.cpp:43-188) and the per-alignment loop of expectedCounts
// (fwdback.cpp:190-209).
//
// This is the STREAMING kernel, the fallback for pairs the on-chip kernel (fwdback_onchip.hip) does not take: an
// envelope row wider than 16 cells, or more than 8 duplication lengths.
//
// One THREAD per alignment pair (the DP of one pair is a short banded scan with almost no
// internal parallelism: ~2P+1 cells per row; a database holds 10^5..10^6 independent pairs).
// The banded Forward matrix of a batch lives in HBM interleaved across the batch,
//     fwd[(cell * (P+2) + lane) * B + b],   b = pair within the batch,
// so that the threads of a wave -- each at its own cell of its own pair, but at the same
// running cell index -- read and write consecutive addresses.  The Backward sweep keeps two
// rows per pair (same interleaving) and accumulates the posterior counts as soon as a
// Backward cell is final, so the Backward matrix is never stored.
//
// fp64, reference operand order, -ffp-contract=off.  log_sum_exp uses the reference's
// 100 001-entry interpolation table (logsumexp.h:19-54), computed on the host with the same
// libm call as the reference and uploaded (800 KB, L2 resident): the cut-off at x >= 10 and
// the linear interpolation are part of the reference's numbers.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fwdback_device.h"

namespace {

constexpr double kNegInf = -__builtin_huge_val();

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

// grid*block >= nBatch threads; thread b handles pair (first + b).
extern "C" __global__ void __launch_bounds__(kFbThreads)
fwdback_estep_kernel(FbArgs a, const int8_t* __restrict__ inSeqs, const int64_t* __restrict__ inOff,
                     const int8_t* __restrict__ outSeqs, const int64_t* __restrict__ outOff,
                     const int32_t* __restrict__ cmIn, const int64_t* __restrict__ cmInOff,
                     const int32_t* __restrict__ cmOut, const int64_t* __restrict__ cmOutOff,
                     const double* __restrict__ lseTab, double* __restrict__ fwd, double* __restrict__ rows,
                     double* __restrict__ pairCounts, double* __restrict__ pairLL, int64_t first, int nBatch,
                     int64_t cellCap, const int64_t* __restrict__ pairList) {
  __shared__ double subS[16], lenS[kFbMaxLen];
  if (threadIdx.x < 16) subS[threadIdx.x] = a.sub[threadIdx.x];
  if (threadIdx.x < kFbMaxLen) lenS[threadIdx.x] = a.len[threadIdx.x];
  __syncthreads();
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nBatch) return;
  const int64_t pair = pairList ? pairList[first + b] : first + b;
  const int P = a.P, W = P + 2;
  const size_t B = (size_t)nBatch;
  const int8_t* in = inSeqs + inOff[pair];
  const int8_t* out = outSeqs + outOff[pair];
  const int inLen = (int)(inOff[pair + 1] - inOff[pair]);
  const int outLen = (int)(outOff[pair + 1] - outOff[pair]);
  const int32_t* ci = cmIn + cmInOff[pair];
  const int32_t* co = cmOut + cmOutOff[pair];
  const int Dm = a.maxDistance;
#define FW(cell, lane) fwd[((size_t)(cell) * W + (size_t)(lane)) * B + (size_t)b]
#define SUBS(i, o) subS[in[(i) - 1] * 4 + out[(o) - 1]]                 /* cellSubScore, fwdback.h:65-67 */
#define DUPS(i, o, k) subS[in[(i) - 1 - (k)] * 4 + out[(o) - 1]]        /* cellTanDupScore, fwdback.h:69-71 */

  // ---------------- Forward (fwdback.cpp:43-78), rows in order, cells of a row contiguous
  // row ip covers op in [lo, hi]: cmOut is non-decreasing, so the envelope of a row is an interval
  int plo = 0, phi = -1;      // previous row
  int64_t pstart = 0, cstart = 0;
  int lo = 0, hi = -1;
  double ll = kNegInf;
  for (int ip = 0; ip <= inLen; ++ip) {
    while (lo <= outLen && co[lo] < ci[ip] - Dm) ++lo;
    if (hi < lo - 1) hi = lo - 1;
    while (hi + 1 <= outLen && co[hi + 1] <= ci[ip] + Dm) ++hi;
    const int mdl = ip < P ? ip : P;                                       // maxDupLenAt, fwdback.h:59
    for (int op = lo; op <= hi; ++op) {
      const int64_t c = cstart + (op - lo);
      if (c >= cellCap) { pairLL[pair] = __builtin_nan(""); return; }       // arena sized on the host: cannot happen
      double s = (ip == 0 && op == 0) ? 0. : kNegInf, d = kNegInf;
      const bool hasIns = ip > 0 && op > 0 && op - 1 >= lo;                 // (ip, op-1) in range
      if (ip > 0 && op > 0) {
        if (op - 1 >= plo && op - 1 <= phi) s = FW(pstart + (op - 1 - plo), 0) + a.noGap + SUBS(ip, op);
        if (hasIns) s = lse(lseTab, s, FW(c - 1, 2) + DUPS(ip, op, 0));
      }
      if (ip > 0 && op >= plo && op <= phi) {
        const int64_t dc = pstart + (op - plo);
        d = lse(lseTab, FW(dc, 0) + a.delOpen, FW(dc, 1) + a.delExtend);
      }
      s = lse(lseTab, s, d + a.delEnd);
      FW(c, 0) = s;
      FW(c, 1) = d;
      for (int k = 0; k < P; ++k) {
        double t = kNegInf;
        if (hasIns && k < mdl - 1) t = FW(c - 1, 2 + k + 1) + DUPS(ip, op, k + 1);
        if (k < mdl) t = lse(lseTab, t, s + a.tanDup + lenS[k]);
        FW(c, 2 + k) = t;
      }
      if (ip == inLen && op == outLen) ll = s;
    }
    plo = lo; phi = hi; pstart = cstart;
    cstart += hi - lo + 1;
  }
  // loglike = sCell(inLen, outLen) (fwdback.cpp:76); -inf when that cell is outside the envelope
  pairLL[pair] = ll;

  // ---------------- Backward (fwdback.cpp:80-116) + counts (fwdback.cpp:154-188), rows in
  // reverse; two Backward rows per pair: rows[((r*rowCap + j) * W + lane) * B + b]
  const int rowCap = a.rowCap;
#define BK(r, j, lane) rows[(((size_t)(r) * rowCap + (size_t)(j)) * W + (size_t)(lane)) * B + (size_t)b]
  // per-thread counters in LDS, [counter][thread]: the sub[][] counters are indexed by data
  __shared__ double cntS[kFbMaxCounts][kFbThreads];
#define cnt(k) cntS[k][threadIdx.x]
  for (int k = 0; k < 21 + P; ++k) cnt(k) = 0;
  // the forward loop left (lo, hi, pstart) describing row inLen
  int nlo = 0, nhi = -1;           // row ip+1
  int64_t rstart = pstart;         // start of row ip in the Forward arena
  int cur = 0;
  for (int ip = inLen; ip >= 0; --ip) {
    if (ip < inLen) {
      // envelope of row ip from that of row ip+1 (pointers move back monotonically)
      while (hi >= 0 && co[hi] > ci[ip] + Dm) --hi;
      if (lo > hi + 1) lo = hi + 1;
      while (lo - 1 >= 0 && co[lo - 1] >= ci[ip] - Dm) --lo;
      rstart -= hi - lo + 1;
    }
    const int mdl = ip < P ? ip : P;
    // Forward row ip-1 (for the counts): its envelope and start
    int qlo = lo, qhi = hi;
    int64_t qstart = rstart;
    if (ip > 0) {
      while (qhi >= 0 && co[qhi] > ci[ip - 1] + Dm) --qhi;
      if (qlo > qhi + 1) qlo = qhi + 1;
      while (qlo - 1 >= 0 && co[qlo - 1] >= ci[ip - 1] - Dm) --qlo;
      qstart = rstart - (qhi - qlo + 1);
    }
    for (int op = hi; op >= lo; --op) {
      const int j = op - lo;
      if (j >= rowCap) { pairLL[pair] = __builtin_nan(""); return; }
      double s = (ip == inLen && op == outLen) ? 0. : kNegInf, d = kNegInf;
      const bool hasIns = op < outLen && ip > 0 && op + 1 <= hi;            // (ip, op+1) in range
      if (op < outLen) {
        if (ip < inLen && op + 1 >= nlo && op + 1 <= nhi) s = a.noGap + SUBS(ip + 1, op + 1) + BK(cur ^ 1, op + 1 - nlo, 0);
      }
      // t lanes first (they feed s)
      for (int k = 0; k < P; ++k) {
        double t = kNegInf;
        if (hasIns && k < mdl) t = (k == 0) ? DUPS(ip, op + 1, 0) + BK(cur, j + 1, 0) : DUPS(ip, op + 1, k) + BK(cur, j + 1, 2 + k - 1);
        BK(cur, j, 2 + k) = t;
      }
      if (ip < inLen && op >= nlo && op <= nhi) {
        const double dd = BK(cur ^ 1, op - nlo, 1);
        s = lse(lseTab, s, a.delOpen + dd);
        d = a.delExtend + dd;
      }
      for (int k = 0; k < mdl; ++k) s = lse(lseTab, s, BK(cur, j, 2 + k) + a.tanDup + lenS[k]);
      d = lse(lseTab, d, s + a.delEnd);
      BK(cur, j, 0) = s;
      BK(cur, j, 1) = d;

      // ---- posterior counts at (ip, op) (fwdback.h:92-112)
      const int64_t c = rstart + j;
      if (ip > 0 && op > 0) {
        const double fS = (op - 1 >= qlo && op - 1 <= qhi) ? FW(qstart + (op - 1 - qlo), 0) : kNegInf;
        const double cS = exp(fS + a.noGap + SUBS(ip, op) + s - ll);                           // pS2S
        cnt(2) += cS;
        cnt(5 + in[ip - 1] * 4 + out[op - 1]) += cS;
        const bool fIns = op - 1 >= lo;
        for (int k = 0; k < mdl - 1; ++k) {
          const double ft = fIns ? FW(c - 1, 2 + k + 1) : kNegInf;
          cnt(5 + in[ip - 1 - (k + 1)] * 4 + out[op - 1]) += exp(ft + DUPS(ip, op, k + 1) + BK(cur, j, 2 + k) - ll);   // pT2T
        }
        const double f0 = fIns ? FW(c - 1, 2) : kNegInf;
        cnt(5 + in[ip - 1] * 4 + out[op - 1]) += exp(f0 + DUPS(ip, op, 0) + s - ll);           // pT2S
      }
      if (ip > 0) {
        const bool up = op >= qlo && op <= qhi;
        const double uS = up ? FW(qstart + (op - qlo), 0) : kNegInf, uD = up ? FW(qstart + (op - qlo), 1) : kNegInf;
        cnt(0) += exp(uS + a.delOpen + d - ll);                                                // pS2D
        cnt(3) += exp(uD + a.delExtend + d - ll);                                              // pD2D
      }
      cnt(4) += exp(FW(c, 1) + a.delEnd + s - ll);                                             // pD2S
      const double fs = FW(c, 0);
      for (int k = 0; k < mdl; ++k) {
        const double cT = exp(fs + a.tanDup + lenS[k] + BK(cur, j, 2 + k) - ll);              // pS2T
        cnt(1) += cT;
        cnt(21 + k) += cT;
      }
    }
    nlo = lo; nhi = hi;
    cur ^= 1;
  }
  for (int k = 0; k < 21 + P; ++k) pairCounts[(size_t)pair * (21 + P) + k] = cnt(k);
}

// Deterministic reduction of the per-pair results: each block sums a contiguous slice in a
// fixed order; the host adds the (few) block partials in order.
extern "C" __global__ void __launch_bounds__(256)
fwdback_reduce_kernel(const double* __restrict__ pairCounts, const double* __restrict__ pairLL, int64_t nPairs, int nc,
                      double* __restrict__ partial /* [gridDim][nc+1] */) {
  __shared__ double sh[256];
  const int64_t per = (nPairs + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < nPairs ? lo + per : nPairs;
  for (int k = 0; k <= nc; ++k) {
    double v = 0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) v += (k < nc) ? pairCounts[(size_t)i * nc + k] : pairLL[i];
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
      __syncthreads();
    }
    if (threadIdx.x == 0) partial[(size_t)blockIdx.x * (nc + 1) + k] = sh[0];
    __syncthreads();
  }
}

// ---- database checks and the envelope census on the GPU (dnas_fb_load_pairs / the routing of dnas_fb_estep) ----------------
// What the host did in loops over every base and guide column of the database (a quarter of a second for 125 000 pairs).

// flags[0] |= 1: a base outside 0..3; |= 2: a guide column array that decreases.  One thread per pair.
extern "C" __global__ void __launch_bounds__(256)
fwdback_validate_kernel(int64_t nPairs, const int8_t* __restrict__ inSeqs, const int64_t* __restrict__ inOff,
                        const int8_t* __restrict__ outSeqs, const int64_t* __restrict__ outOff, const int32_t* __restrict__ cmIn,
                        const int64_t* __restrict__ cmInOff, const int32_t* __restrict__ cmOut, const int64_t* __restrict__ cmOutOff,
                        unsigned* __restrict__ flags) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nPairs) return;
  unsigned bad = 0;
  const int64_t I = inOff[i + 1] - inOff[i], O = outOff[i + 1] - outOff[i];
  const int8_t* in = inSeqs + inOff[i];
  const int8_t* out = outSeqs + outOff[i];
  const int32_t* ci = cmIn + cmInOff[i];
  const int32_t* co = cmOut + cmOutOff[i];
  for (int64_t k = 0; k < I; ++k) { if (in[k] < 0 || in[k] > 3) bad |= 1u; if (ci[k + 1] < ci[k]) bad |= 2u; }
  for (int64_t k = 0; k < O; ++k) { if (out[k] < 0 || out[k] > 3) bad |= 1u; if (co[k + 1] < co[k]) bad |= 2u; }
  if (bad) atomicOr(flags, bad);
}

// The envelope of every pair (alignpath.h:48-53: op in [lo(ip), hi(ip)] <=> |cm(ip) - cm(op)| <= maxDistance) in one pass per
// pair: cells = sum of the row widths, width = the widest row, steps = the anti-diagonals of its wavefront, and whether half the
// lanes of the wavefront kernels do (fwdback_onchip.hip: hi(ip) - lo(ip + W) < W for every row; W = 8 and 16).
// census[i] = {cells, width | fits8 << 16 | fits16 << 17, steps}.  One thread per pair; three two-pointer scans in step
// (rows ip, ip - 8 and ip - 16).
extern "C" __global__ void __launch_bounds__(256)
fwdback_census_kernel(int64_t nPairs, int maxDistance, const int64_t* __restrict__ inOff, const int64_t* __restrict__ outOff,
                      const int32_t* __restrict__ cmIn, const int64_t* __restrict__ cmInOff, const int32_t* __restrict__ cmOut,
                      const int64_t* __restrict__ cmOutOff, int64_t* __restrict__ census) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nPairs) return;
  const int64_t I = inOff[i + 1] - inOff[i], O = outOff[i + 1] - outOff[i];
  const int32_t* ci = cmIn + cmInOff[i];
  const int32_t* co = cmOut + cmOutOff[i];
  const int Dm = maxDistance;
  // scan q = 0: row ip; q = 1: row ip - 8; q = 2: row ip - 16 (only its hi is needed)
  int64_t lo[3] = {0, 0, 0}, hi[3] = {-1, -1, -1};
  auto advance = [&](int q, int64_t ip) {
    while (lo[q] <= O && co[lo[q]] < ci[ip] - Dm) ++lo[q];
    if (hi[q] < lo[q] - 1) hi[q] = lo[q] - 1;
    while (hi[q] + 1 <= O && co[hi[q] + 1] <= ci[ip] + Dm) ++hi[q];
  };
  int64_t cells = 0, lo0 = 0;
  int width = 1;
  bool fits8 = true, fits16 = true;
  for (int64_t ip = 0; ip <= I; ++ip) {
    advance(0, ip);
    if (ip == 0) lo0 = lo[0];
    cells += hi[0] - lo[0] + 1;
    width = max(width, (int)(hi[0] - lo[0] + 1));
    if (ip >= 8) {
      advance(1, ip - 8);
      if (hi[1] >= lo[1] && hi[1] - lo[0] >= 8) fits8 = false;       // (an empty row holds no lane)
    }
    if (ip >= 16) {
      advance(2, ip - 16);
      if (hi[2] >= lo[2] && hi[2] - lo[0] >= 16) fits16 = false;
    }
  }
  census[3 * i] = cells > 0 ? cells : 1;
  census[3 * i + 1] = (int64_t)width | ((int64_t)(fits8 ? 1 : 0) << 16) | ((int64_t)(fits16 ? 1 : 0) << 17);
  census[3 * i + 2] = I + (hi[0] > 0 ? hi[0] : 0) - lo0 + 1;
}
