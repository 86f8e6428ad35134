// Hand-written gfx950 kernels for the Viterbi error decoder.
//
//   viterbi_fill_kernel       replaces the ViterbiMatrix constructor's lattice fill
//                             (reference src/viterbi.cpp:62-176)
//   viterbi_traceback_kernel  replaces ViterbiMatrix::traceback (src/viterbi.cpp:195-304)
//
// One work-group per read.  The lattice lives in HBM as [pos][lane][state] fp64
// (lanes S, D, T1..TD; the reference's AoS [pos][state][lane], viterbi.h:65-67, turned
// SoA so that every lane of a column is one coalesced stream).  The current column's S
// and D rows double as the working storage of the in-column max-plus fixpoint.
//
// All arithmetic is fp64 max/+ in the reference's operand order; no contraction, no
// fast-math (build with -ffp-contract=off).  max() is spelled (a < b ? b : a) == std::max.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_model.h"

namespace {

__device__ __forceinline__ double dmax(double a, double b) { return a < b ? b : a; }

constexpr double kNegInf = -__builtin_huge_val();

// Pull relaxation of one state inside the current column (the body of the reference's
// worklist loop, viterbi.cpp:112-158, seen from the destination side):
//   D(j) = max( D(j), max(D(src)+delExtend, S(src)+delOpen) + score   over emit-in,
//                     D(src)+score                                    over null-in )
//   S(j) = max( S(j), S(src)+score over null-in, D(j)+delEnd )
// Returns true when either cell grew.
__device__ __forceinline__ bool pull_state(const DevModel& m, int j, double* __restrict__ S, double* __restrict__ D) {
  const double d0 = D[j], s0 = S[j];
  double d = d0, s = s0;
  const int eb = m.einPtr[j], ee = m.einPtr[j + 1];
  for (int e = eb; e < ee; ++e) {
    const int src = m.einSrc[e];
    const double cand = dmax(D[src] + m.delExtend, S[src] + m.delOpen) + m.einScore[e];
    d = dmax(d, cand);
  }
  const int nb = m.ninPtr[j], ne = m.ninPtr[j + 1];
  for (int e = nb; e < ne; ++e) {
    const int src = m.ninSrc[e];
    const double sc = m.ninScore[e];
    d = dmax(d, D[src] + sc);
    s = dmax(s, S[src] + sc);
  }
  s = dmax(s, d + m.delEnd);
  bool grew = false;
  if (d > d0) { D[j] = d; grew = true; }
  if (s > s0) { S[j] = s; grew = true; }
  return grew;
}

// Mark every out-neighbour of j dirty for the next round.
__device__ __forceinline__ void mark_successors(const DevModel& m, int j, unsigned* maskNext) {
  const int eb = m.eoutPtr[j], ee = m.eoutPtr[j + 1];
  for (int e = eb; e < ee; ++e) {
    const int d = m.eoutDst[e];
    atomicOr(&maskNext[d >> 5], 1u << (d & 31));
  }
  const int nb = m.noutPtr[j], ne = m.noutPtr[j + 1];
  for (int e = nb; e < ne; ++e) {
    const int d = m.noutDst[e];
    atomicOr(&maskNext[d >> 5], 1u << (d & 31));
  }
}

// One lattice cell, whatever the storage tier.  With storedLanes == 2 the duplication lanes
// are not in memory; T(p,q) = max(T(p-1,q+1) + sub[ctx[q+1]][x_p], (S(p)+tanDup)+len[q]),
// T(0,.) = -inf (viterbi.cpp:105-106,161-168) is rebuilt from the state's own S cells, deepest
// element first -- the same fp64 operations in the same order as the fill, hence the same bits.
__device__ __forceinline__ double lattice_cell(const DevModel& m, const double* __restrict__ lat,
                                               const uint8_t* __restrict__ seq, int st, int ps, int ln) {
  const size_t stride = (size_t)m.Npad;
  const size_t slot = (size_t)(m.slotOf ? m.slotOf[st] : st);
  if (ln < 2 || m.storedLanes > 2) return lat[((size_t)ps * m.storedLanes + (size_t)ln) * stride + slot];
  const int k = ln - 2, mdl = m.mdl[st];
  if (ps < 1 || k >= mdl) return kNegInf;
  const uint8_t* ctx = m.ctx + (size_t)st * m.D;
  int I = mdl - 1 - k;
  if (ps - 1 < I) I = ps - 1;
  double v = (lat[((size_t)(ps - I) * 2) * stride + slot] + m.tanDup) + m.len[k + I];
  for (int i = I - 1; i >= 0; --i) {
    const int p = ps - i, q = k + i;
    v = dmax(v + m.sub[ctx[q + 1] * 4 + seq[p - 1]], (lat[((size_t)p * 2) * stride + slot] + m.tanDup) + m.len[q]);
  }
  return v;
}

// the same with the lattice slot already known
__device__ __forceinline__ double cell_at(const DevModel& m, const double* __restrict__ lat, const uint8_t* __restrict__ seq,
                                          int st, int slot, int ps, int ln) {
  const size_t stride = (size_t)m.Npad;
  if (ln < 2 || m.storedLanes > 2) return lat[((size_t)ps * m.storedLanes + (size_t)ln) * stride + (size_t)slot];
  const int k = ln - 2, mdl = m.mdl[st];
  if (ps < 1 || k >= mdl) return kNegInf;
  const uint8_t* ctx = m.ctx + (size_t)st * m.D;
  int I = mdl - 1 - k;
  if (ps - 1 < I) I = ps - 1;
  double v = (lat[((size_t)(ps - I) * 2) * stride + (size_t)slot] + m.tanDup) + m.len[k + I];
  for (int i = I - 1; i >= 0; --i) {
    const int p = ps - i, q = k + i;
    v = dmax(v + m.sub[ctx[q + 1] * 4 + seq[p - 1]], (lat[((size_t)p * 2) * stride + (size_t)slot] + m.tanDup) + m.len[q]);
  }
  return v;
}

// Duplication lane k of a state at column ps from the state's node record (device_model.h: mdl, the context's bases 2 bits each,
// the lattice slot): lattice_cell's chain for at most four S cells (records exist for D <= 4), every one of them loaded before
// the chain is evaluated -- a loop over them would wait for each load in turn.  SL = lanes stored per column.
__device__ __forceinline__ double dup_cell_from_record(const double* __restrict__ lat, const uint8_t* __restrict__ seq, size_t SL, size_t stride,
                                                       int slot, int mdl, unsigned ctxBits, int ps, int k, double tanDup,
                                                       const double* subT, const double* lenT) {
  if (SL > 2) return lat[((size_t)ps * SL + (size_t)(2 + k)) * stride + (size_t)slot];
  if (ps < 1 || k >= mdl) return kNegInf;
  int I = mdl - 1 - k;
  if (ps - 1 < I) I = ps - 1;
  double sv[4];
  int xb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    sv[i] = i <= I ? lat[((size_t)(ps - i) * SL) * stride + (size_t)slot] : 0.;
    xb[i] = i < I ? (int)seq[ps - i - 1] : 0;
  }
  double v = kNegInf;
#pragma unroll
  for (int i = 3; i >= 0; --i) {
    if (i == I) v = (sv[i] + tanDup) + lenT[k + i];
    else if (i < I) v = dmax(v + subT[((ctxBits >> (2 * (k + i + 1))) & 3u) * 4 + xb[i]], (sv[i] + tanDup) + lenT[k + i]);
  }
  return v;
}

}  // namespace

// Device-side guard of dnas_viterbi_batch_device: counts base codes outside 0..3.
extern "C" __global__ void check_bases_kernel(const uint8_t* __restrict__ bases, size_t n, unsigned long long* __restrict__ bad) {
  unsigned mine = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) mine += bases[i] > 3;
  if (mine) atomicAdd(bad, (unsigned long long)mine);
}

// -inf into every cell of the tier-C exchange buffers before a launch.
// Where should a cluster's sync words sit?  The members of a cluster agree on every lattice column through device-scope
// atomics and loads on a few words (viterbi_tiera.hip): operations performed at the memory side, whose round trip from an XCD
// depends on which memory channel the address belongs to (measured on MI355X: one ~980-nt read of the 46 670-state machine on
// 16 work-groups fills in 41.6 ms with its sync block at one address and in 54 ms 4 KB further on, same box, same process).
// Block b of this kernel (b = 0 .. 7: the XCD that block b, b + 8, ... of a launch are dispatched to) times `reps` dependent
// atomic round trips to each of nCand candidate addresses, pool + k * strideWords, and files the ticks under out[b * nCand + k].
extern "C" __global__ void sync_latency_kernel(unsigned* __restrict__ pool, int nCand, int strideWords, int reps,
                                               unsigned long long* __restrict__ out, unsigned* __restrict__ xccOut) {
  if (threadIdx.x != 0) return;
  const int b = (int)blockIdx.x;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xccOut[b] = xcc & 15u;
  for (int i = 0; i < nCand; ++i) {
    const int k = (i + b * 5) % nCand;      // (the blocks walk the candidates out of step)
    unsigned* const p = pool + (size_t)k * (size_t)strideWords;
    unsigned dep = __hip_atomic_fetch_add(p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // warm: page tables, the line
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) dep = __hip_atomic_fetch_add(p + (dep & 0u), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[(size_t)b * nCand + k] = t1 - t0 + (dep & 0u);
  }
}

// Which XCD is next in the dispatcher's round?  (One block: the first block of the kernel launched right behind this one goes to
// the same XCD -- observed; a matter of speed only.)
extern "C" __global__ void xcc_probe_kernel(unsigned* __restrict__ out) {
  if (threadIdx.x != 0) return;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  *out = xcc & 15u;
}

extern "C" __global__ void fill_neginf_kernel(double* __restrict__ p, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = kNegInf;
}

// Bounded-memory decode: `n` doubles from src + row * srcStride to dst + row * dstStride for every row (one row per read
// of the group).  grid = (chunks, rows).
extern "C" __global__ void copy_rows_kernel(double* __restrict__ dst, const double* __restrict__ src, size_t dstStride,
                                            size_t srcStride, size_t n) {
  const double* s = src + (size_t)blockIdx.y * srcStride;
  double* d = dst + (size_t)blockIdx.y * dstStride;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}

// Test/diagnostic aid: the full (D+2)-lane lattice of one read in reference state order,
// out[(pos*(D+2)+lane)*N + state], whatever the storage tier.  grid = L+1, any block size.
extern "C" __global__ void expand_lattice_kernel(DevModel m, const uint8_t* __restrict__ seq, const double* __restrict__ lat,
                                                 double* __restrict__ out) {
  const int ps = blockIdx.x, lanes = m.D + 2, L = (int)gridDim.x - 1;
  for (int st = threadIdx.x; st < m.N; st += blockDim.x)
    for (int ln = 0; ln < lanes; ++ln) {
      double v = lattice_cell(m, lat, seq, st, ps, ln);
      if (m.local && m.storedLanes == 2 && ln >= 2 && ps == L && st == m.N - 1 && L >= 1) {
        // local mode: the reference formed T(N-1, L, .) from S(N-1, L) BEFORE overwriting that cell with the
        // column maximum (viterbi.cpp:161-173); the fill kernel kept the earlier value behind the lattice
        const int k = ln - 2, mdl = m.mdl[st];
        if (k < mdl) {
          const double s0 = lat[(size_t)(L + 1) * 2 * (size_t)m.Npad];
          const double opened = (s0 + m.tanDup) + m.len[k];
          // T(L,k) = max( T(L-1,k+1) + sub[ctx[k+1]][x_L],  opened ):  the first term does not involve S(L)
          double shifted = kNegInf;
          if (k + 1 < mdl) shifted = lattice_cell(m, lat, seq, st, ps - 1, ln + 1) + m.sub[m.ctx[(size_t)st * m.D + k + 1] * 4 + seq[ps - 1]];
          v = dmax(shifted, opened);
        }
      }
      out[((size_t)ps * lanes + ln) * m.N + st] = v;
    }
}

// grid = reads in this batch, block = kFillThreads.
// Dynamic LDS: two dirty-state bitmasks of maskWords 32-bit words each.
extern "C" __global__ void __launch_bounds__(kFillThreads)
viterbi_fill_kernel(DevModel m, const uint8_t* __restrict__ bases, const uint64_t* __restrict__ readOff,
                    const int32_t* __restrict__ batchRead, const uint64_t* __restrict__ slotOff,
                    double* __restrict__ arena, double* __restrict__ outLoglike,
                    unsigned long long* __restrict__ roundsTotal, int maskWords,
                    const int* __restrict__ colRange) {   // [reads][2] first and last column to fill (bounded-memory decode), or null: 0 .. L
  extern __shared__ unsigned ldsMask[];
  __shared__ double redBuf[kFillThreads / 64];
  const int tid = threadIdx.x;
  const int T = blockDim.x;
  const int N = m.N, D_ = m.D, lanes = m.D + 2;
  const size_t Npad = (size_t)m.Npad;
  const int read = batchRead[blockIdx.x];
  const uint8_t* seq = bases + readOff[read];
  const int L = (int)(readOff[read + 1] - readOff[read]);
  double* lat = arena + slotOff[blockIdx.x];
  unsigned* mask0 = ldsMask;
  unsigned* mask1 = ldsMask + maskWords;
  const int maskBytes = (N + 7) >> 3;
  unsigned rounds = 0;

  for (int w = tid; w < 2 * maskWords; w += T) ldsMask[w] = 0;
  __syncthreads();

  // a segment c0 .. c1 of the read: the column before c0 is in the lattice (every lane is stored in this tier)
  int c0 = 0, c1 = L;
  if (colRange) { c0 = colRange[2 * blockIdx.x]; c1 = colRange[2 * blockIdx.x + 1]; }
  for (int pos = c0; pos <= c1; ++pos) {
    double* col = lat + (size_t)pos * lanes * Npad;
    double* S = col;
    double* Dc = col + Npad;
    const double* prev = col - (size_t)lanes * Npad;  // valid for pos > 0
    const int x = pos > 0 ? seq[pos - 1] : 0;

    // ---- phase A: S from the previous column (viterbi.cpp:92-95,101-103); D = -inf
    for (int j = tid; j < N; j += T) {
      double s;
      if (pos == 0) {
        s = (m.local || j == 0) ? 0. : kNegInf;  // viterbi.cpp:75-79
      } else {
        s = kNegInf;
        const int eb = m.einPtr[j], ee = m.einPtr[j + 1];
        for (int e = eb; e < ee; ++e)
          s = dmax(s, ((prev[m.einSrc[e]] + m.einScore[e]) + m.noGap) + m.sub[m.einBase[e] * 4 + x]);
        if (m.mdl[j] > 0) s = dmax(s, prev[2 * Npad + j] + m.sub[m.ctx[(size_t)j * D_] * 4 + x]);
      }
      S[j] = s;
      Dc[j] = kNegInf;
    }
    __syncthreads();

    // ---- phase B: in-column fixpoint (viterbi.cpp:97-99,110-159).  Round 1 visits every
    // state (the reference seeds its worklist with all of them); later rounds visit the
    // states whose predecessors grew.  Monotone max-plus => the least fixpoint reached is
    // independent of the visiting order.
    unsigned* cur = mask0;
    unsigned* nxt = mask1;
    {
      int marked = 0;
      for (int j = tid; j < N; j += T)
        if (pull_state(m, j, S, Dc)) { mark_successors(m, j, nxt); marked = 1; }
      ++rounds;
      int more = __syncthreads_or(marked);
      while (more) {
        unsigned* t = cur; cur = nxt; nxt = t;
        marked = 0;
        uint8_t* curBytes = reinterpret_cast<uint8_t*>(cur);
        for (int b = tid; b < maskBytes; b += T) {
          unsigned bits = curBytes[b];
          if (!bits) continue;
          curBytes[b] = 0;
          while (bits) {
            const int j = (b << 3) + __builtin_ctz(bits);
            bits &= bits - 1;
            if (pull_state(m, j, S, Dc)) { mark_successors(m, j, nxt); marked = 1; }
          }
        }
        ++rounds;
        more = __syncthreads_or(marked);
      }
    }

    // ---- phase C: duplication lanes (viterbi.cpp:105-106,161-168)
    for (int j = tid; j < N; j += T) {
      const double s = S[j];
      const int mdl = m.mdl[j];
      for (int k = 0; k < D_; ++k) {
        double t = kNegInf;
        if (pos > 0 && k < mdl) {
          if (k < mdl - 1) t = prev[(size_t)(3 + k) * Npad + j] + m.sub[m.ctx[(size_t)j * D_ + k + 1] * 4 + x];
          t = dmax(t, (s + m.tanDup) + m.len[k]);
        }
        col[(size_t)(2 + k) * Npad + j] = t;
      }
    }
    __syncthreads();
  }

  if (c1 < L) {   // not the read's last segment: no log-likelihood yet
    if (tid == 0) atomicAdd(roundsTotal, (unsigned long long)rounds);
    return;
  }
  // ---- local mode: loglike = best end state (viterbi.cpp:171-173)
  double* lastS = lat + (size_t)L * lanes * Npad;
  if (m.local) {
    double best = kNegInf;
    for (int j = tid; j < N; j += T) best = dmax(best, lastS[j]);
    for (int off = 32; off > 0; off >>= 1) best = dmax(best, __shfl_down(best, off, 64));
    if ((tid & 63) == 0) redBuf[tid >> 6] = best;
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < T / 64; ++w) best = dmax(best, redBuf[w]);
      lastS[N - 1] = best;
      outLoglike[read] = best;
    }
  } else if (tid == 0) {
    outLoglike[read] = lastS[N - 1];
  }
  if (tid == 0) atomicAdd(roundsTotal, (unsigned long long)rounds);
}

// One thread per read: the reference's sequential pointer chase, candidate order and
// strict '>' tie-breaking included (viterbi.cpp:217-228,247-301).
extern "C" __global__ void __launch_bounds__(256)
viterbi_traceback_kernel(DevModel m, const uint8_t* __restrict__ bases, const uint64_t* __restrict__ readOff,
                         const int32_t* __restrict__ batchRead, const uint64_t* __restrict__ slotOff,
                         const double* __restrict__ arena, char* __restrict__ outSym,
                         const uint64_t* __restrict__ outOff, uint32_t* __restrict__ outLen,
                         uint8_t* __restrict__ outStatus, int nBatch,
                         unsigned long long* __restrict__ events, const uint64_t* __restrict__ evOff, uint32_t* __restrict__ evLen,
                         int readsPerWave) {
  // the model's small tables, indexed per lane: from LDS (a per-lane index into the kernel-argument block is a load from
  // wherever the runtime keeps kernel arguments -- host memory, as a rule -- in the middle of every step)
  __shared__ double subT[16], lenT[kMaxLen];
  if (threadIdx.x < 16) subT[threadIdx.x] = m.sub[threadIdx.x];
  if (threadIdx.x < kMaxLen) lenT[threadIdx.x] = m.len[threadIdx.x];
  __syncthreads();
  // readsPerWave of a wave's 64 lanes walk a read each: the lanes of a wave are at different places of different machines'
  // states, and a step costs the wave the union of what its lanes do -- fewer reads per wave, fewer paths per step
  const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = (int)(threadIdx.x & 63);
  if (lane >= readsPerWave) return;
  const int b = wave * readsPerWave + lane;
  if (b >= nBatch) return;
  const int read = batchRead[b];
  const uint8_t* seq = bases + readOff[read];
  const int L = (int)(readOff[read + 1] - readOff[read]);
  const double* lat = arena + slotOff[b];
  const int N = m.N, D_ = m.D;
  char* out = outSym + outOff[read];
  const long cap = (long)(outOff[read + 1] - outOff[read]);
  long n = 0;
  // optional log of what the traceback found (the reference's level-3 messages, viterbi.cpp:266-293), in the order
  // it walks (from the end of the read): type << 62 | pos << 32 | payload.  1 substitution at pos: emitted base << 2 |
  // read base; 2 deletion between pos-1 and pos: the deleted base; 3 duplication at pos: count << 26 | bases (2 bits each, at most 13: kEventDupBases)
  unsigned long long* ev = events ? events + evOff[read] : nullptr;
  const long evCap = events ? (long)(evOff[read + 1] - evOff[read]) : 0;
  long nEv = 0;
  bool bestEmit = false;     // the winning candidate is an emit in-edge (the reference's bestIts with a defined base)
  uint8_t bestBase = 0;
#define EVENT(type, p, payload) { if (nEv < evCap) ev[nEv] = ((unsigned long long)(type) << 62) | ((unsigned long long)(unsigned)(p) << 32) | (unsigned long long)(payload); ++nEv; }

#define CELL(st, ps, ln) lattice_cell(m, lat, seq, (st), (ps), (ln))
  if (!(CELL(N - 1, L, 0) > kNegInf)) {  // viterbi.cpp:198-201
    outLen[read] = 0;
    outStatus[read] = 1;  // DNAS_READ_NO_PATH
    return;
  }
  int state = N - 1, pos = L, mut = 0;
  int bestState = 0, bestPos = 0, bestMut = 0;
  double best, bestCell = kNegInf;
  bool found;
  uint8_t bestIn;
  uint8_t status = 0;
  double curCell;   // the lattice cell the walk stands on (read when it was chosen as a candidate)

  // A step's candidates are independent loads: they are gathered CH at a time -- edge fields, then
  // the cells, then the compares in the reference's order (first strictly greater wins) -- so that a
  // step costs a few memory latencies instead of one chain per candidate.
  constexpr int CH = 4;
  int cSt[CH], cSlot[CH], cPs[CH], cLn[CH];
  double cTr[CH];
  uint8_t cIn[CH], cEm[CH];   // cEm: 0 not an emit in-edge, else 1 + emitted base
#define INIT_BEST() { best = kNegInf; found = false; bestIn = 0; bestEmit = false; }
#define FLUSH(cnt) { \
    double v_[CH]; \
    _Pragma("unroll") for (int i_ = 0; i_ < CH; ++i_) v_[i_] = i_ < (cnt) ? cell_at(m, lat, seq, cSt[i_], cSlot[i_], cPs[i_], cLn[i_]) : kNegInf; \
    _Pragma("unroll") for (int i_ = 0; i_ < CH; ++i_) if (i_ < (cnt)) { \
      const double sc_ = v_[i_] + cTr[i_]; \
      if (sc_ > best) { best = sc_; bestCell = v_[i_]; bestState = cSt[i_]; bestPos = cPs[i_]; bestMut = cLn[i_]; bestIn = cIn[i_]; found = true; \
                        bestEmit = cEm[i_] != 0; bestBase = (uint8_t)(cEm[i_] - 1); } } }
#define SET(i, st_, slot_, ps_, ln_, tr_, in_) { cSt[i] = (st_); cSlot[i] = (slot_); cPs[i] = (ps_); cLn[i] = (ln_); cTr[i] = (tr_); cIn[i] = (in_); cEm[i] = 0; }
#define SET_EMIT(i, st_, slot_, ps_, ln_, tr_, in_, base_) { SET(i, st_, slot_, ps_, ln_, tr_, in_) cEm[i] = (uint8_t)(1 + (base_)); }
#define CHECK_BEST() { \
    const double den_ = fabs(curCell) < 1e-6 ? 1. : curCell; \
    if (!(fabs((best - curCell) / den_) < 1e-6) || !found) { status = 3; break; } \
    state = bestState; pos = bestPos; mut = bestMut; curCell = bestCell; }
  const int32_t* slotOf = m.slotOf;
#define SLOT(st) (slotOf ? slotOf[st] : (st))
  const bool useRec = m.rec != nullptr && ev == nullptr;
  bool haveRec = false;
  uint4 cur0 = {0, 0, 0, 0}, cur1 = cur0, cur2 = cur0, cur3 = cur0;   // the node record of `state`
#define LOAD_REC(dst, st) { const uint4* r_ = (const uint4*)(m.rec + (size_t)(st) * 16); dst##0 = r_[0]; dst##1 = r_[1]; dst##2 = r_[2]; dst##3 = r_[3]; }

  do {  // single-pass block so CHECK_BEST can break out on failure
    INIT_BEST();
    curCell = CELL(N - 1, L, 0);
    if (m.local) {
      for (int s0 = 0; s0 < N; s0 += CH) {
        _Pragma("unroll") for (int i = 0; i < CH; ++i) if (s0 + i < N) SET(i, s0 + i, SLOT(s0 + i), L, 0, 0., 0)
        FLUSH(N - s0 < CH ? N - s0 : CH)
      }
    } else {
      SET(0, N - 1, SLOT(N - 1), L, 0, 0., 0)
      FLUSH(1)
    }
    CHECK_BEST();

    while (pos >= 0 && state > 0) {
      INIT_BEST();
      // ---- the step from the state's node record (device_model.h): the record names the in-edges, so the candidates' cells
      // are loaded at once -- and with them the RECORDS of the candidate states, one of which is the next step's: a step
      // costs one memory latency instead of three (row pointers, edge fields, cells).  States with more in-edges than a
      // record holds, the local start column and the event log take the walk through the CSR arrays below.
      const int stateWas = state;
      int adopt = -1;            // fast step taken: 0..3 = the next state's record is pre[adopt], 4 = the state stays
      if (useRec) {
        if (!haveRec) { LOAD_REC(cur, state) haveRec = true; }
        const unsigned hd = cur0.x;
        const int nE = (int)(hd & 15u), nN = (int)((hd >> 4) & 15u);
        if (nE != 15 && nN != 15 && !(mut == 0 && pos == 0 && m.local)) {
          const int mdlR = (int)((hd >> 8) & 15u), ownSlotR = (int)cur0.y;
          const unsigned ctxR = (hd >> 12) & 0xffu;
          const size_t strideR = (size_t)m.Npad, SL = (size_t)m.storedLanes;
          auto LAT = [&](int ps, int ln, int slot) -> double { return lat[((size_t)ps * SL + (size_t)ln) * strideR + (size_t)slot]; };
          auto ownT = [&](int ps, int k) -> double {
            return dup_cell_from_record(lat, seq, SL, strideR, ownSlotR, mdlR, ctxR, ps, k, m.tanDup, subT, lenT);
          };
          const int eSrc[3] = {(int)cur0.z, (int)cur1.y, (int)cur2.x}, eSlot[3] = {(int)cur0.w, (int)cur1.z, (int)cur2.y};
          const unsigned eMisc[3] = {cur1.x, cur1.w, cur2.z};
          const int nSrc = (int)cur2.w, nSlot = (int)cur3.x;
          const unsigned nMisc = cur3.y;
          uint4 pre0[4], pre1[4], pre2[4], pre3[4];   // pre<word>[candidate]: the records of the emit sources 0..2 and of the null source
#define LOAD_PRE(i, st) { const uint4* r_ = (const uint4*)(m.rec + (size_t)(st) * 16); pre0[i] = r_[0]; pre1[i] = r_[1]; pre2[i] = r_[2]; pre3[i] = r_[3]; }
          // the candidates in the reference's order: value, transition score, (state, pos, lane), input symbol, record to adopt
#define TAKE(val, trans, st_, ps_, ln_, in_, adopt_) { const double v_ = (val); const double sc_ = v_ + (trans); \
            if (sc_ > best) { best = sc_; bestCell = v_; bestState = (st_); bestPos = (ps_); bestMut = (ln_); bestIn = (uint8_t)(in_); found = true; adopt = (adopt_); } }
          if (mut == 0) {
            const int x = pos > 0 ? seq[pos - 1] : 0;
            double ve[3], te[3];
            _Pragma("unroll") for (int i = 0; i < 3; ++i) {
              ve[i] = kNegInf; te[i] = 0.;
              if (pos > 0 && i < nE) {
                ve[i] = LAT(pos - 1, 0, eSlot[i]);
                te[i] = (m.recScore[(eMisc[i] >> 16) & 255u] + m.noGap) + subT[((eMisc[i] >> 8) & 255u) * 4 + x];
                LOAD_PRE(i, eSrc[i])
              }
            }
            double vn = kNegInf, tn = 0.;
            if (nN > 0) { vn = LAT(pos, 0, nSlot); tn = m.recScore[(nMisc >> 16) & 255u]; LOAD_PRE(3, nSrc) }
            const double vd = LAT(pos, 1, ownSlotR);
            const bool dup = mdlR > 0 && pos > 0;
            const double vt = dup ? ownT(pos - 1, 0) : kNegInf;
            _Pragma("unroll") for (int i = 0; i < 3; ++i)
              if (pos > 0 && i < nE) TAKE(ve[i], te[i], eSrc[i], pos - 1, 0, eMisc[i] & 255u, i)
            if (nN > 0) TAKE(vn, tn, nSrc, pos, 0, nMisc & 255u, 3)
            TAKE(vd, m.delEnd, state, pos, 1, 0, 4)
            if (dup) TAKE(vt, subT[(ctxR & 3u) * 4 + x], state, pos - 1, 2, 0, 4)
          } else if (mut == 1) {
            double vD[3], vS[3], sc[3];
            _Pragma("unroll") for (int i = 0; i < 3; ++i) {
              vD[i] = vS[i] = kNegInf; sc[i] = 0.;
              if (i < nE) {
                vD[i] = LAT(pos, 1, eSlot[i]);
                vS[i] = LAT(pos, 0, eSlot[i]);
                sc[i] = m.recScore[(eMisc[i] >> 16) & 255u];
                LOAD_PRE(i, eSrc[i])
              }
            }
            double vn = kNegInf, tn = 0.;
            if (nN > 0) { vn = LAT(pos, 1, nSlot); tn = m.recScore[(nMisc >> 16) & 255u]; LOAD_PRE(3, nSrc) }
            _Pragma("unroll") for (int i = 0; i < 3; ++i)
              if (i < nE) {
                TAKE(vD[i], sc[i] + m.delExtend, eSrc[i], pos, 1, eMisc[i] & 255u, i)
                TAKE(vS[i], sc[i] + m.delOpen, eSrc[i], pos, 0, eMisc[i] & 255u, i)
              }
            if (nN > 0) TAKE(vn, tn, nSrc, pos, 1, nMisc & 255u, 3)
          } else {
            const int k = mut - 2;
            const bool deeper = k < mdlR - 1;
            const double vt = deeper ? ownT(pos - 1, k + 1) : kNegInf;
            const double vs = LAT(pos, 0, ownSlotR);
            if (deeper) TAKE(vt, subT[((ctxR >> (2 * (k + 1))) & 3u) * 4 + seq[pos - 1]], state, pos - 1, 2 + k + 1, 0, 4)
            TAKE(vs, m.tanDup + lenT[k], state, pos, 0, 0, 4)
            if (adopt < 0) adopt = 4;
          }
          if (adopt < 0) adopt = 4;     // no candidate at all: CHECK_BEST below reports it
          if (adopt == 0) { cur0 = pre0[0]; cur1 = pre1[0]; cur2 = pre2[0]; cur3 = pre3[0]; }
          else if (adopt == 1) { cur0 = pre0[1]; cur1 = pre1[1]; cur2 = pre2[1]; cur3 = pre3[1]; }
          else if (adopt == 2) { cur0 = pre0[2]; cur1 = pre1[2]; cur2 = pre2[2]; cur3 = pre3[2]; }
          else if (adopt == 3) { cur0 = pre0[3]; cur1 = pre1[3]; cur2 = pre2[3]; cur3 = pre3[3]; }
#undef TAKE
#undef LOAD_PRE
        }
      }
      if (adopt < 0) {
      const int mdl = m.mdl[state];
      const uint8_t* ctx = m.ctx + (size_t)state * D_;
      const int ownSlot = SLOT(state);
      if (mut == 0) {
        if (pos > 0) {
          const int x = seq[pos - 1];
          const int e1 = m.einPtr[state + 1];
          for (int e0 = m.einPtr[state]; e0 < e1; e0 += CH) {
            _Pragma("unroll") for (int i = 0; i < CH; ++i) if (e0 + i < e1)
              SET_EMIT(i, m.einSrc[e0 + i], m.einSlot[e0 + i], pos - 1, 0, (m.einScore[e0 + i] + m.noGap) + subT[m.einBase[e0 + i] * 4 + x], m.einIn[e0 + i], m.einBase[e0 + i])
            FLUSH(e1 - e0 < CH ? e1 - e0 : CH)
          }
        }
        {
          const int e1 = m.ninPtr[state + 1];
          for (int e0 = m.ninPtr[state]; e0 < e1; e0 += CH) {
            _Pragma("unroll") for (int i = 0; i < CH; ++i) if (e0 + i < e1)
              SET(i, m.ninSrc[e0 + i], m.ninSlot[e0 + i], pos, 0, m.ninScore[e0 + i], m.ninIn[e0 + i])
            FLUSH(e1 - e0 < CH ? e1 - e0 : CH)
          }
        }
        int c = 0;
        SET(0, state, ownSlot, pos, 1, m.delEnd, 0)
        c = 1;
        if (mdl > 0 && pos > 0) { SET(1, state, ownSlot, pos - 1, 2, subT[ctx[0] * 4 + seq[pos - 1]], 0) c = 2; }
        if (pos == 0 && m.local) {
          if (c == 1) SET(1, 0, SLOT(0), 0, 0, 0., 0) else SET(2, 0, SLOT(0), 0, 0, 0., 0)
          ++c;
        }
        FLUSH(c)
        if (ev && bestEmit && bestPos < pos && seq[pos - 1] != bestBase) EVENT(1, pos - 1, (bestBase << 2) | seq[pos - 1])   // viterbi.cpp:266-267
      } else if (mut == 1) {
        const int e1 = m.einPtr[state + 1];
        for (int e0 = m.einPtr[state]; e0 < e1; e0 += CH / 2) {
          _Pragma("unroll") for (int i = 0; i < CH / 2; ++i) if (e0 + i < e1) {
            SET_EMIT(2 * i, m.einSrc[e0 + i], m.einSlot[e0 + i], pos, 1, m.einScore[e0 + i] + m.delExtend, m.einIn[e0 + i], m.einBase[e0 + i])
            SET_EMIT(2 * i + 1, m.einSrc[e0 + i], m.einSlot[e0 + i], pos, 0, m.einScore[e0 + i] + m.delOpen, m.einIn[e0 + i], m.einBase[e0 + i])
          }
          FLUSH(2 * (e1 - e0 < CH / 2 ? e1 - e0 : CH / 2))
        }
        const int n1 = m.ninPtr[state + 1];
        for (int e0 = m.ninPtr[state]; e0 < n1; e0 += CH) {
          _Pragma("unroll") for (int i = 0; i < CH; ++i) if (e0 + i < n1)
            SET(i, m.ninSrc[e0 + i], m.ninSlot[e0 + i], pos, 1, m.ninScore[e0 + i], m.ninIn[e0 + i])
          FLUSH(n1 - e0 < CH ? n1 - e0 : CH)
        }
        // viterbi.cpp:278-279 (for a null in-edge the reference prints an uninitialised base: not reproduced)
        if (ev && bestEmit) EVENT(2, pos, bestBase)
      } else {
        const int k = mut - 2;
        int c = 0;
        if (k < mdl - 1) { SET(0, state, ownSlot, pos - 1, 2 + k + 1, subT[ctx[k + 1] * 4 + seq[pos - 1]], 0) c = 1; }
        if (c == 0) SET(0, state, ownSlot, pos, 0, m.tanDup + lenT[k], 0) else SET(1, state, ownSlot, pos, 0, m.tanDup + lenT[k], 0)
        ++c;
        FLUSH(c)
        if (ev && bestMut == 0) {        // viterbi.cpp:288-293: the duplicated bases, outermost first
          unsigned bases = 0;
          for (int q = k; q >= 0; --q) bases = (bases << 2) | ctx[q];
          EVENT(3, pos, ((unsigned)(k + 1) << 26) | (bases & 0x3ffffffu))
        }
      }
      }   // the step through the CSR arrays
      CHECK_BEST();
      if (adopt < 0 && state != stateWas) haveRec = false;   // (a fast step has taken the next state's record along)
      if (bestIn) {  // trace.push_front (viterbi.cpp:299-300): fill the slot from its end
        if (n < cap) out[cap - 1 - n] = (char)bestIn;
        ++n;
      }
    }
  } while (false);
#undef CELL
#undef INIT_BEST
#undef FLUSH
#undef SET
#undef SLOT
#undef CHECK_BEST
#undef SET_EMIT
#undef EVENT
#undef LOAD_REC

  if (evLen) evLen[read] = (uint32_t)(nEv < evCap ? nEv : evCap);
  if (status == 0 && n > cap) status = 2;  // DNAS_READ_OUT_OVERFLOW
  if (status != 0) {
    outLen[read] = 0;
    outStatus[read] = status;
    return;
  }
  for (long k = 0; k < n; ++k) out[k] = out[cap - n + k];  // forward order, moved to the slot's front
  outLen[read] = (uint32_t)n;
  outStatus[read] = 0;
}

// The same walk with one WAVE per read: the candidates of a step -- in-edges of the state, its own D and T cells --
// are dealt over the lanes, each lane fetches its candidate's edge fields and lattice cell, and the wave keeps the
// first strictly greater one (= the largest value, lowest candidate index among equals: the reference's updateBest
// order, viterbi.cpp:217-228).  A step then costs three rounds of memory latency (the state's row pointers and
// context, the edge fields, the cells) instead of one chain per candidate.  grid = ceil(nBatch / 4), block = 256.
extern "C" __global__ void __launch_bounds__(256)
viterbi_traceback_wave_kernel(DevModel m, const uint8_t* __restrict__ bases, const uint64_t* __restrict__ readOff,
                              const int32_t* __restrict__ batchRead, const uint64_t* __restrict__ slotOff,
                              const double* __restrict__ arena, char* __restrict__ outSym,
                              const uint64_t* __restrict__ outOff, uint32_t* __restrict__ outLen,
                              uint8_t* __restrict__ outStatus, int nBatch,
                              unsigned long long* __restrict__ events, const uint64_t* __restrict__ evOff, uint32_t* __restrict__ evLen,
                              const int* __restrict__ colRange, TracebackWalk* __restrict__ walks) {
  // Bounded-memory decode (colRange / walks not null): the lattice holds columns colRange[2b] - (D + 1) .. colRange[2b + 1]
  // of read b only.  The walk runs while it stands on a column >= colRange[2b] (a step looks at most D columns back), is
  // then parked in walks[b] and picked up by the launch over the segment before.
  __shared__ double subT[16], lenT[kMaxLen];   // (as in the thread-per-read kernel: no per-lane index into the kernel-argument block)
  if (threadIdx.x < 16) subT[threadIdx.x] = m.sub[threadIdx.x];
  if (threadIdx.x < kMaxLen) lenT[threadIdx.x] = m.len[threadIdx.x];
  __syncthreads();
  const int b = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
  const int ln = threadIdx.x & 63;
  if (b >= nBatch) return;
  const int read = batchRead[b];
  const uint8_t* seq = bases + readOff[read];
  const int L = (int)(readOff[read + 1] - readOff[read]);
  const double* lat = arena + slotOff[b];
  const int N = m.N, D_ = m.D;
  char* out = outSym + outOff[read];
  const long cap = (long)(outOff[read + 1] - outOff[read]);
  long n = 0;
  unsigned long long* ev = events ? events + evOff[read] : nullptr;
  const long evCap = events ? (long)(evOff[read + 1] - evOff[read]) : 0;
  long nEv = 0;
  const int32_t* slotOf = m.slotOf;
#define WSLOT(st) (slotOf ? slotOf[st] : (st))
#define WEVENT(type, p, payload) { if (ln == 0 && nEv < evCap) ev[nEv] = ((unsigned long long)(type) << 62) | ((unsigned long long)(unsigned)(p) << 32) | (unsigned long long)(payload); ++nEv; }

  TracebackWalk* const walk = walks ? walks + b : nullptr;
  const int stopBelow = colRange ? colRange[2 * b] : 0;
  const int phase = walk ? walk->phase : 0;      // 0: not started, 1: parked, 2: finished
  if (phase == 2) return;
  int state = N - 1, pos = L, mut = 0;
  uint8_t status = 0;
  double curCell = kNegInf;
  if (phase == 1) {
    state = walk->state; pos = walk->pos; mut = walk->mut; curCell = walk->curCell; n = (long)walk->n; nEv = (long)walk->nEv;
  } else {
    if (!(lattice_cell(m, lat, seq, N - 1, L, 0) > kNegInf)) {  // viterbi.cpp:198-201
      if (ln == 0) { outLen[read] = 0; outStatus[read] = 1; if (walk) walk->phase = 2; }   // DNAS_READ_NO_PATH
      return;
    }
    curCell = lattice_cell(m, lat, seq, N - 1, L, 0);
  }

  const bool useRec = m.rec != nullptr && ev == nullptr;
  bool haveW = false;
  uint4 w0 = {0, 0, 0, 0}, w1 = w0, w2 = w0, w3 = w0;   // the node record of `state`, the same in every lane
  // running best of a step (uniform over the wave)
  double best; bool found; int bState, bPos, bMut; double bCell; uint8_t bIn, bEm;
  // one candidate per lane -> the wave's first strictly greater one, merged into the running best
  auto offer = [&](bool has, int st, int slot, int ps, int lane_, double tr, uint8_t in, uint8_t em) {
    double v = kNegInf, sc = kNegInf;
    if (has) { v = cell_at(m, lat, seq, st, slot, ps, lane_); sc = v + tr; }
    double bs = sc;
    int bi = ln;
    for (int off = 32; off > 0; off >>= 1) {
      const double os = __shfl_xor(bs, off, 64);
      const int oi = __shfl_xor(bi, off, 64);
      if (os > bs || (os == bs && oi < bi)) { bs = os; bi = oi; }
    }
    if (bs > best) {            // candidates of earlier calls come first: strictly greater only
      best = bs; found = true;
      bState = __shfl(st, bi, 64); bPos = __shfl(ps, bi, 64); bMut = __shfl(lane_, bi, 64); bCell = __shfl(v, bi, 64);
      bIn = (uint8_t)__shfl((int)in, bi, 64); bEm = (uint8_t)__shfl((int)em, bi, 64);
    }
  };
#define W_INIT() { best = kNegInf; found = false; bIn = 0; bEm = 0; bState = 0; bPos = 0; bMut = 0; bCell = kNegInf; }
#define W_CHECK() { \
    const double den_ = fabs(curCell) < 1e-6 ? 1. : curCell; \
    if (!(fabs((best - curCell) / den_) < 1e-6) || !found) { status = 3; break; } \
    state = bState; pos = bPos; mut = bMut; curCell = bCell; }

  do {
    if (phase == 0) {
      W_INIT();
      if (m.local) {
        for (int s0 = 0; s0 < N; s0 += 64) { const int st = s0 + ln; offer(st < N, st < N ? st : 0, st < N ? WSLOT(st) : 0, L, 0, 0., 0, 0); }
      } else {
        offer(ln == 0, N - 1, WSLOT(N - 1), L, 0, 0., 0, 0);
      }
      W_CHECK();
    }

    while (pos >= stopBelow && pos >= 0 && state > 0) {
      // ---- the step from the state's node record (device_model.h), where its in-edges fit one: every lane forms its candidate
      // from the record (the same in all lanes), loads the candidate's cell AND the candidate state's record, and the winner's
      // record goes to all lanes with its cell -- one round of memory latency per step instead of three.  The CSR rounds below
      // serve the other states, the local start column and the event log.
      const int stateWas = state;
      bool fastStep = false;
      if (useRec) {
        if (!haveW) { const uint4* r_ = (const uint4*)(m.rec + (size_t)state * 16); w0 = r_[0]; w1 = r_[1]; w2 = r_[2]; w3 = r_[3]; haveW = true; }
        const unsigned hd = w0.x;
        const int nEr = (int)(hd & 15u), nNr = (int)((hd >> 4) & 15u);
        if (nEr != 15 && nNr != 15 && !(mut == 0 && pos == 0 && m.local)) {
          fastStep = true;
          const int mdlR = (int)((hd >> 8) & 15u), ownSlotR = (int)w0.y;
          const unsigned ctxR = (hd >> 12) & 0xffu;
          const size_t strideR = (size_t)m.Npad, SL = (size_t)m.storedLanes;
          const int x = pos > 0 ? seq[pos - 1] : 0;
          W_INIT();
          bool has = false, edge = false;          // edge: the candidate is another state (its record is fetched with its cell)
          int st = state, slot = ownSlotR, ps = pos, lane_ = 0;
          double tr = 0.;
          unsigned in = 0;
          // in-edge e of the record: emitting 0..2, null 3
          auto edgeOf = [&](int e, int& src, int& sl, unsigned& misc) {
            src = e == 0 ? (int)w0.z : e == 1 ? (int)w1.y : e == 2 ? (int)w2.x : (int)w2.w;
            sl = e == 0 ? (int)w0.w : e == 1 ? (int)w1.z : e == 2 ? (int)w2.y : (int)w3.x;
            misc = e == 0 ? w1.x : e == 1 ? w1.w : e == 2 ? w2.z : w3.y;
          };
          if (mut == 0) {
            const int nEe = pos > 0 ? nEr : 0;
            const bool hasT = mdlR > 0 && pos > 0;
            const int total = nEe + nNr + 1 + (hasT ? 1 : 0);
            has = ln < total;
            if (ln < nEe + nNr) {
              unsigned misc;
              edgeOf(ln < nEe ? ln : 3, st, slot, misc);
              edge = true;
              in = misc & 255u;
              if (ln < nEe) { ps = pos - 1; tr = (m.recScore[(misc >> 16) & 255u] + m.noGap) + subT[((misc >> 8) & 255u) * 4 + x]; }
              else tr = m.recScore[(misc >> 16) & 255u];
            } else if (ln == nEe + nNr) { lane_ = 1; tr = m.delEnd; }
            else if (has) { ps = pos - 1; lane_ = 2; tr = subT[(ctxR & 3u) * 4 + x]; }
          } else if (mut == 1) {
            const int total = 2 * nEr + nNr;
            has = ln < total;
            if (has) {
              unsigned misc;
              edgeOf(ln < 2 * nEr ? (ln >> 1) : 3, st, slot, misc);
              edge = true;
              in = misc & 255u;
              const double sc = m.recScore[(misc >> 16) & 255u];
              if (ln < 2 * nEr) { if ((ln & 1) == 0) { lane_ = 1; tr = sc + m.delExtend; } else { lane_ = 0; tr = sc + m.delOpen; } }
              else { lane_ = 1; tr = sc; }
            }
          } else {
            const int k = mut - 2;
            const bool shift = k < mdlR - 1;
            has = ln < (shift ? 2 : 1);
            const bool first = shift && ln == 0;
            ps = first ? pos - 1 : pos;
            lane_ = first ? 2 + k + 1 : 0;
            tr = first ? subT[((ctxR >> (2 * (k + 1))) & 3u) * 4 + x] : m.tanDup + lenT[k < kMaxLen ? k : 0];
          }
          uint4 p0 = {0, 0, 0, 0}, p1 = p0, p2 = p0, p3 = p0;
          double v = kNegInf, sc = kNegInf;
          if (has) {
            if (edge) { const uint4* r_ = (const uint4*)(m.rec + (size_t)st * 16); p0 = r_[0]; p1 = r_[1]; p2 = r_[2]; p3 = r_[3]; }
            v = lane_ < 2 ? lat[((size_t)ps * SL + (size_t)lane_) * strideR + (size_t)slot]
                          : dup_cell_from_record(lat, seq, SL, strideR, ownSlotR, mdlR, ctxR, ps, lane_ - 2, m.tanDup, subT, lenT);
            sc = v + tr;
          }
          double bs = sc;
          int bi = ln;
          for (int off = 32; off > 0; off >>= 1) {
            const double os = __shfl_xor(bs, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (os > bs || (os == bs && oi < bi)) { bs = os; bi = oi; }
          }
          if (bs > best) {
            best = bs; found = true;
            bState = __shfl(st, bi, 64); bPos = __shfl(ps, bi, 64); bMut = __shfl(lane_, bi, 64); bCell = __shfl(v, bi, 64);
            bIn = (uint8_t)__shfl((int)in, bi, 64); bEm = 0;
            if (__shfl((int)edge, bi, 64)) {       // the next state's record, from the lane that fetched it
#define W_BCAST(q) { q.x = (unsigned)__shfl((int)q.x, bi, 64); q.y = (unsigned)__shfl((int)q.y, bi, 64); q.z = (unsigned)__shfl((int)q.z, bi, 64); q.w = (unsigned)__shfl((int)q.w, bi, 64); }
              W_BCAST(p0) W_BCAST(p1) W_BCAST(p2) W_BCAST(p3)
#undef W_BCAST
              w0 = p0; w1 = p1; w2 = p2; w3 = p3;
            }
          }
        }
      }
      if (!fastStep) {
      // round 1: what the step needs to know about the state, fetched by different lanes
      int meta = 0;
      if (ln == 0) meta = m.einPtr[state]; else if (ln == 1) meta = m.einPtr[state + 1];
      else if (ln == 2) meta = m.ninPtr[state]; else if (ln == 3) meta = m.ninPtr[state + 1];
      else if (ln == 4) meta = m.mdl[state]; else if (ln == 5) meta = WSLOT(state);
      else if (ln >= 8 && ln < 8 + D_) meta = m.ctx[(size_t)state * D_ + (ln - 8)];
      const int e0 = __shfl(meta, 0, 64), e1 = __shfl(meta, 1, 64), n0 = __shfl(meta, 2, 64), n1 = __shfl(meta, 3, 64);
      const int mdl = __shfl(meta, 4, 64), ownSlot = __shfl(meta, 5, 64);
      const int x = pos > 0 ? seq[pos - 1] : 0;
      const int ctx0 = __shfl(meta, 8, 64);
      W_INIT();
      if (mut == 0) {
        const int nE = pos > 0 ? e1 - e0 : 0, nN = n1 - n0;
        int extra = 1;                                     // own D
        const bool hasT = mdl > 0 && pos > 0, hasStart = pos == 0 && m.local;
        extra += hasT ? 1 : 0;
        extra += hasStart ? 1 : 0;
        const int total = nE + nN + extra;
        for (int c0 = 0; c0 < total; c0 += 64) {
          const int c = c0 + ln;
          bool has = c < total;
          int st = 0, slot = 0, ps = pos, lane_ = 0;
          double tr = 0.;
          uint8_t in = 0, em = 0;
          if (has) {
            if (c < nE) {
              const int e = e0 + c;
              st = m.einSrc[e]; slot = m.einSlot[e]; ps = pos - 1; lane_ = 0;
              tr = (m.einScore[e] + m.noGap) + subT[m.einBase[e] * 4 + x];
              in = m.einIn[e]; em = (uint8_t)(1 + m.einBase[e]);
            } else if (c < nE + nN) {
              const int e = n0 + (c - nE);
              st = m.ninSrc[e]; slot = m.ninSlot[e]; ps = pos; lane_ = 0; tr = m.ninScore[e]; in = m.ninIn[e];
            } else {
              const int q = c - nE - nN;                   // 0: own D, then own T1 (if any), then the local start
              if (q == 0) { st = state; slot = ownSlot; ps = pos; lane_ = 1; tr = m.delEnd; }
              else if (q == 1 && hasT) { st = state; slot = ownSlot; ps = pos - 1; lane_ = 2; tr = subT[ctx0 * 4 + x]; }
              else { st = 0; slot = WSLOT(0); ps = 0; lane_ = 0; tr = 0.; }
            }
          }
          offer(has, st, slot, ps, lane_, tr, in, em);
        }
        if (ev && bEm && bPos < pos && seq[pos - 1] != (uint8_t)(bEm - 1)) WEVENT(1, pos - 1, ((unsigned)(bEm - 1) << 2) | seq[pos - 1])   // viterbi.cpp:266-267
      } else if (mut == 1) {
        const int nE = e1 - e0, nN = n1 - n0, total = 2 * nE + nN;
        for (int c0 = 0; c0 < total; c0 += 64) {
          const int c = c0 + ln;
          const bool has = c < total;
          int st = 0, slot = 0, lane_ = 0;
          double tr = 0.;
          uint8_t in = 0, em = 0;
          if (has) {
            if (c < 2 * nE) {
              const int e = e0 + (c >> 1);
              st = m.einSrc[e]; slot = m.einSlot[e]; in = m.einIn[e]; em = (uint8_t)(1 + m.einBase[e]);
              if ((c & 1) == 0) { lane_ = 1; tr = m.einScore[e] + m.delExtend; } else { lane_ = 0; tr = m.einScore[e] + m.delOpen; }
            } else {
              const int e = n0 + (c - 2 * nE);
              st = m.ninSrc[e]; slot = m.ninSlot[e]; lane_ = 1; tr = m.ninScore[e]; in = m.ninIn[e];
            }
          }
          offer(has, st, slot, pos, lane_, tr, in, em);
        }
        if (ev && bEm) WEVENT(2, pos, (unsigned)(bEm - 1))                                          // viterbi.cpp:278-279
      } else {
        const int k = mut - 2;
        const bool shift = k < mdl - 1;
        const int ctxNext = __shfl(meta, 8 + (k + 1 < D_ ? k + 1 : 0), 64);
        // candidate 0: T(k+1) one column back (if any); then S of this column
        const int total = shift ? 2 : 1;
        const bool has = ln < total;
        const bool first = shift && ln == 0;
        offer(has, state, ownSlot, first ? pos - 1 : pos, first ? 2 + k + 1 : 0,
              first ? subT[ctxNext * 4 + x] : m.tanDup + lenT[k], 0, 0);
        if (ev && bMut == 0) {        // viterbi.cpp:288-293: the duplicated bases, outermost first
          unsigned basesDup = 0;
          for (int q = k; q >= 0; --q) basesDup = (basesDup << 2) | (unsigned)__shfl(meta, 8 + q, 64);
          WEVENT(3, pos, ((unsigned)(k + 1) << 26) | (basesDup & 0x3ffffffu))
        }
      }
      }   // the step through the CSR arrays
      W_CHECK();
      if (!fastStep && state != stateWas) haveW = false;   // (a fast step has brought the next state's record along)
      if (bIn) {  // trace.push_front (viterbi.cpp:299-300): fill the slot from its end
        if (ln == 0 && n < cap) out[cap - 1 - n] = (char)bIn;
        ++n;
      }
    }
  } while (false);
#undef W_INIT
#undef W_CHECK
#undef WSLOT

  if (walk && status == 0 && pos >= 0 && state > 0) {   // the walk left this segment: park it
    if (ln == 0) {
      walk->state = state; walk->pos = pos; walk->mut = mut; walk->curCell = curCell; walk->n = (long long)n; walk->nEv = (long long)nEv;
      walk->phase = 1;
    }
    return;
  }
  if (ln == 0 && walk) walk->phase = 2;
  if (ln == 0 && evLen) evLen[read] = (uint32_t)(nEv < evCap ? nEv : evCap);
#undef WEVENT
  if (status == 0 && n > cap) status = 2;  // DNAS_READ_OUT_OVERFLOW
  if (status != 0) {
    if (ln == 0) { outLen[read] = 0; outStatus[read] = status; }
    return;
  }
  __builtin_amdgcn_wave_barrier();
  if (ln == 0) {
    for (long k2 = 0; k2 < n; ++k2) out[k2] = out[cap - n + k2];  // forward order, moved to the slot's front
    outLen[read] = (uint32_t)n;
    outStatus[read] = 0;
  }
}
