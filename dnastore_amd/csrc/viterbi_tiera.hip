// viterbi_tiera.hip -- the fast lattice-fill kernel ("tier A"), specialised at JIT time
// for one machine (see host/plan.hpp).  Replaces the ViterbiMatrix constructor's fill loop
// (reference src/viterbi.cpp:62-176) for machines whose state set fits one CU.
//
// One work-group (DNAS_T threads, one per CU) per read.  Thread t owns the states in
// slots t, t+T, t+2T, ... ("rows" k = 0..K-1); their S and D cells of the current column
// live in REGISTERS for the whole in-column fixpoint.  What other threads need is
// published in LDS:
//     X[slot]  = max(D+delExtend, S+delOpen)    every state  (what an emit edge reads, viterbi.cpp:124)
//     DN[cell], SN[cell]                         states that are sources of null edges, or
//                                                "heavy" destinations (in-degree > threshold)
// The edge lists are compiled by the host into per-thread 32-bit entries that are loaded
// into registers once per read; the row shape (entries per row) is a compile-time constant,
// so every register index is static.  A column is: gather S of the previous column from
// LDS -> sweep all rows until no cell grows (monotone max-plus: the least fixpoint is
// schedule independent, so the cells equal the reference's worklist result bit for bit)
// -> duplication lanes -> coalesced stores of the six lanes in slot order.
//
// fp64 throughout, reference operand order, no contraction / fast-math.
//
// Compile-time parameters (-D):  DNAS_T threads, DNAS_K rows, DNAS_D dup lanes,
//   DNAS_NS slots (= K*T), DNAS_C LDS cells (incl. two dummies),
//   DNAS_ROWS  brace list of {emit pulls, null pulls, pushes, publishes} per row.
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>   // hiprtc provides the device runtime implicitly
#endif

#ifndef DNAS_T
#error "compile with -DDNAS_T= -DDNAS_K= -DDNAS_D= -DDNAS_NS= -DDNAS_C= -DDNAS_ROWS="
#endif

// per row: emit pulls / null pulls by score class (class 0 adds 0.0, i.e. nothing), pushes, publishes
struct RowShape { int e[4], n[4], ep, ec; };
constexpr RowShape kRows[DNAS_K] = {DNAS_ROWS};

constexpr int rowEE(int k) { return kRows[k].e[0] + kRows[k].e[1] + kRows[k].e[2] + kRows[k].e[3]; }
constexpr int rowEN(int k) { return kRows[k].n[0] + kRows[k].n[1] + kRows[k].n[2] + kRows[k].n[3]; }
constexpr int rowOffset(int k) {
  int o = 0;
  for (int i = 0; i < k; ++i) o += rowEE(i) + rowEN(i) + kRows[i].ep + kRows[i].ec;
  return o;
}
// score class of the e-th emit (null) pull of row k
constexpr int emitClass(int k, int e) {
  int c = 0;
  while (e >= kRows[k].e[c]) { e -= kRows[k].e[c]; ++c; }
  return c;
}
constexpr int nullClass(int k, int e) {
  int c = 0;
  while (e >= kRows[k].n[c]) { e -= kRows[k].n[c]; ++c; }
  return c;
}
constexpr int kEntries = rowOffset(DNAS_K) > 0 ? rowOffset(DNAS_K) : 1;
constexpr int emitSlotBase(int k) {
  int o = 0;
  for (int i = 0; i < k; ++i) o += rowEE(i);
  return o;
}
constexpr int maxRowVals() {
  int m = 0;
  for (int k = 0; k < DNAS_K; ++k) m = rowEE(k) + 2 * rowEN(k) > m ? rowEE(k) + 2 * rowEN(k) : m;
  return m;
}
constexpr int kMaxRowVals = maxRowVals();

typedef double dbl2 __attribute__((ext_vector_type(2)));
static_assert(DNAS_K % 2 == 0, "rows come in pairs");

template <int V> struct IntC { static constexpr int value = V; };
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(IntC<I>{});
    static_for<I + 1, N>(f);
  }
}

// kernel-argument block (mirrors runtime.hip TierAArgs)
struct TierAArgs {
  int N;            // real states
  int local;
  double noGap, delOpen, delExtend, delEnd, tanDup;
  double sub[16];
  double len[8];
  double score[4];  // score table, score[0] == 0
};

// LDS map (bytes):  X[NS] | DN[C] | SN[C] | negInf | score[4] | sub[16] | len[8] | red[T/64] | epoch, idle[T/64] (u32)
constexpr int kXBytes = DNAS_NS * 8;
constexpr int kCellBytes = DNAS_C * 8;           // SN[cell] sits kCellBytes behind DN[cell]
constexpr int kTabBase = kXBytes + 2 * kCellBytes + 8;

// entries (host/plan.cpp packs them).  Pull entries are bare LDS byte addresses -- X[src slot]
// for an emit pull, DN[src cell] for a null pull -- and their score class is a compile-time
// property of the entry's position in the row.  Push / publish entries carry flags:
//   [0:19)  byte address of DN[cell]
//   [19:24) score index << 3   (byte offset into the LDS score table)
//   [24:26) emitted base       (emit push)
//   [26]    flag: push = emit edge;  publish = heavy cell (also receives pushes)
//   [27]    publish: state has a cell
#define ENT_ADDR(e) ((e) & 0x7ffffu)
#define ENT_SCOFF(e) (((e) >> 19) & 0x18u)
#define ENT_BASE(e) (((e) >> 24) & 3u)
#define ENT_FLAG(e) (((e) >> 26) & 1u)
#define ENT_HASCELL(e) (((e) >> 27) & 1u)

// v_max_f64 directly: no NaN can occur here (only -inf + finite / -inf + -inf), so the
// canonicalising copies the compiler would add around fmax() are pure overhead
__device__ __forceinline__ double dmax(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double ldsRead(const char* base, unsigned byteOff) {
  return *reinterpret_cast<const double*>(base + byteOff);
}
__device__ __forceinline__ void ldsWrite(char* base, unsigned byteOff, double v) {
  *reinterpret_cast<double*>(base + byteOff) = v;
}
__device__ __forceinline__ void ldsMax(char* base, unsigned byteOff, double v) {
  __hip_atomic_fetch_max(reinterpret_cast<double*>(base + byteOff), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// keep a register-resident entry opaque inside the sweep loop, so that nothing derived
// from it is hoisted out and kept live across iterations
__device__ __forceinline__ unsigned opaque(unsigned v) {
  asm volatile("" : "+v"(v));
  return v;
}

constexpr double kNegInf = -__builtin_huge_val();

#ifndef DNAS_PIPE
#define DNAS_PIPE 1
#endif

extern "C" __global__ void __launch_bounds__(DNAS_T)
viterbi_fill_tiera(TierAArgs a, const unsigned* __restrict__ entTab,   // [kEntries][T]
                   const unsigned* __restrict__ metaTab,                // [K][T]: mdl | ctx<<4 | flags
                   const unsigned* __restrict__ baseTab,                // [DNAS_BASEWORDS][T]: base of each emit pull, 2 bits each
                   const unsigned char* __restrict__ bases, const unsigned long long* __restrict__ readOff,
                   const int* __restrict__ batchRead, const unsigned long long* __restrict__ slotOff,
                   double* __restrict__ arena, double* __restrict__ outLoglike,
                   unsigned long long* __restrict__ roundsTotal) {
  extern __shared__ double lds[];
  extern __shared__ unsigned ldsU[];
  constexpr int T = DNAS_T, K = DNAS_K, D_ = DNAS_D, lanes = 2, NS = DNAS_NS;   // stored lanes: S, D
  const int tid = threadIdx.x;
  char* const ldsB = reinterpret_cast<char*>(lds);
  double* const X = lds;
  const double* const subL = lds + (kTabBase / 8) + 4;
  const double* const lenL = subL + 16;
  // the vote words live in the same dynamic LDS block; a second extern array (same base) keeps
  // the accesses in the LDS address space (a volatile generic pointer would turn them into
  // flat_* operations that wait on every outstanding global store)
  unsigned* const epochL = ldsU + 2 * ((kTabBase / 8) + 28 + DNAS_T / 64);   // termination words: epoch, idle[T/64]
  unsigned* const idleL = epochL + 1;

  const int read = batchRead[blockIdx.x];
  const unsigned char* seq = bases + readOff[read];
  const int L = (int)(readOff[read + 1] - readOff[read]);
  double* const lat = arena + slotOff[blockIdx.x];

  unsigned E[kEntries];
  static_for<0, kEntries>([&](auto m) { E[m.value] = entTab[(size_t)m.value * T + tid]; });
#define META(k) (metaTab[(size_t)(k) * T + tid])
  unsigned baseW[DNAS_BASEWORDS];   // emitted base of every emit pull of this thread, 2 bits each (phase A)
  static_for<0, DNAS_BASEWORDS>([&](auto w) { baseW[w.value] = baseTab[(size_t)w.value * T + tid]; });
  const double scoreC[4] = {0.0, a.score[1], a.score[2], a.score[3]};

  double S[K], Dv[K];   // after phase C, Dv[k] carries the T1 hand-over to the next column's phase A
  unsigned rounds = 0;
#ifdef DNAS_STAMP   // diagnostic build: where does a column spend its cycles (never in the shipped kernel)
  unsigned long long tA = 0, tP = 0, tB = 0, tC = 0, t0 = 0, t1 = 0;
#define STAMP(acc) { t1 = __builtin_amdgcn_s_memtime(); acc += t1 - t0; t0 = t1; }
#else
#define STAMP(acc)
#endif

  // LDS init: everything -inf (dummy cells / slots stay that way), then the small tables
  for (int i = tid; i < NS + 2 * DNAS_C + 1; i += T) lds[i] = kNegInf;
  if (tid < 4) lds[kTabBase / 8 + tid] = a.score[tid];
  if (tid < 16) lds[kTabBase / 8 + 4 + tid] = a.sub[tid];
  if (tid < 8) lds[kTabBase / 8 + 20 + tid] = a.len[tid];
  if (tid < 1 + DNAS_T / 64) epochL[tid] = 0;
  __syncthreads();

  // The S and D lanes of column p leave for HBM from the registers, 16 bytes per lane (rows 2m and
  // 2m+1 of a thread are lattice neighbours).
#define STORE_LANE(p, lane, REG)                                                                 \
  {                                                                                              \
    double* const colp = lat + ((size_t)(p) * lanes + (lane)) * NS;                              \
    static_for<0, K / 2>([&](auto mc) {                                                          \
      constexpr int m2 = mc.value;                                                               \
      if (pairValid & (1u << m2)) {                                                              \
        dbl2 v2;                                                                                 \
        v2.x = REG[2 * m2]; v2.y = REG[2 * m2 + 1];                                              \
        reinterpret_cast<dbl2*>(colp + (size_t)m2 * 2 * T)[tid] = v2;                            \
      }                                                                                          \
    });                                                                                          \
  }
  unsigned pairValid = 0;   // bit m: the lattice pair (rows 2m, 2m+1) of this thread holds a real state
  static_for<0, K / 2>([&](auto mc) {
    if ((META(2 * mc.value) | META(2 * mc.value + 1)) & 0x20000000u) pairValid |= 1u << mc.value;
  });

  for (int pos = 0; pos <= L; ++pos) {
    double* const col = lat + (size_t)pos * lanes * NS;
    const int x = pos > 0 ? seq[pos - 1] : 0;
#ifdef DNAS_STAMP
    t0 = __builtin_amdgcn_s_memtime();
#endif

    // ---- phase A (viterbi.cpp:75-79,92-95,101-103): S of this column from the previous
    // column's S (parked in X[] by phase C) and the T1 lane (handed over in the D registers by phase C).
    // Heavy destinations receive their emit-in candidates by ds_max pushes into SN[cell]
    // (their owners reset the cell in phase C).
    if (pos > 0) {
      static_for<0, K>([&](auto kc) {
        constexpr int k = kc.value, o = rowOffset(k);
        static_for<0, kRows[k].ep>([&](auto ec) {
          const unsigned en = E[o + rowEE(k) + rowEN(k) + ec.value];
          if (ENT_FLAG(en)) {
            const double cand = ((S[k] + ldsRead(ldsB, kTabBase + ENT_SCOFF(en))) + a.noGap) + subL[ENT_BASE(en) * 4 + x];
            if (cand > kNegInf) ldsMax(ldsB, ENT_ADDR(en) + kCellBytes, cand);
          }
        });
      });
      static_for<0, K>([&](auto kc) {
        constexpr int k = kc.value, o = rowOffset(k);
        double s = Dv[k];   // T1(pos-1) + sub[ctx1][x_pos], left there by phase C
        static_for<0, rowEE(k)>([&](auto ec) {
          constexpr int cls = emitClass(k, ec.value);
          // (S(src) + score) + noGap + sub: "+ 0.0" of class 0 is the identity on every value that occurs
          double v = ldsRead(ldsB, E[o + ec.value]);
          if constexpr (cls != 0) v = v + scoreC[cls];
          constexpr int slot = emitSlotBase(k) + ec.value;   // this thread's slot-th emit pull overall
          s = dmax(s, (v + a.noGap) + subL[((baseW[slot / 16] >> (2 * (slot % 16))) & 3u) * 4 + x]);
        });
        S[k] = s;
      });
    } else {
      static_for<0, K>([&](auto kc) {
        constexpr int k = kc.value;
        const unsigned mt = META(k);    // bit29: real state, bit31: reference state 0
        S[k] = ((mt & 0x20000000u) && (a.local || (mt & 0x80000000u))) ? 0.0 : kNegInf;   // viterbi.cpp:75-79
      });
    }
    __syncthreads();   // every gather of the previous column (and every phase-A push) is done
    STAMP(tA)

    // ---- start of the fixpoint: D = -inf, X = max(D+delExtend, S+delOpen); cells published;
    // every push edge fired once with the starting values
    static_for<0, K>([&](auto kc) {
      constexpr int k = kc.value, o = rowOffset(k);
      double s = S[k];
      static_for<0, kRows[k].ec>([&](auto ec) {
        const unsigned en = E[o + rowEE(k) + rowEN(k) + kRows[k].ep + ec.value];
        if (ENT_HASCELL(en)) {
          if (ENT_FLAG(en)) {        // heavy destination: fold in what phase A pushed, never lower the cell
            s = dmax(s, ldsRead(ldsB, ENT_ADDR(en) + kCellBytes));
            ldsMax(ldsB, ENT_ADDR(en) + kCellBytes, s);
          } else {
            ldsWrite(ldsB, ENT_ADDR(en) + kCellBytes, s);
            ldsWrite(ldsB, ENT_ADDR(en), kNegInf);
          }
        }
      });
      S[k] = s;
      Dv[k] = kNegInf;
      const double xv = dmax(kNegInf + a.delExtend, s + a.delOpen);
      X[k * T + tid] = xv;
      static_for<0, kRows[k].ep>([&](auto ec) {
        const unsigned en = E[o + rowEE(k) + rowEN(k) + ec.value];
        const double sc = ldsRead(ldsB, kTabBase + ENT_SCOFF(en));
        if (ENT_FLAG(en)) {
          if (xv + sc > kNegInf) ldsMax(ldsB, ENT_ADDR(en), xv + sc);
        } else {
          if (s + sc > kNegInf) ldsMax(ldsB, ENT_ADDR(en) + kCellBytes, s + sc);
        }
      });
    });
    __syncthreads();
    STAMP(tP)

    // ---- phase B: sweeps to the fixpoint (viterbi.cpp:97-99,110-159), WITHOUT a barrier per
    // sweep.  Every wave keeps sweeping its own rows (chaotic relaxation: a monotone max-plus
    // system reaches the same least fixpoint under any schedule) and the work-group agrees on
    // termination through LDS words:
    //     epoch    bumped by a wave whose sweep grew a cell
    //     idle[w]  = e+1 once wave w has finished a sweep that grew nothing and saw epoch == e
    //              from its first read to its last
    // When all waves are idle at the same epoch, nothing was written while each of them swept:
    // every cell is consistent with its inputs, i.e. the fixpoint.  The LDS reads of row k+1 are
    // issued before row k is evaluated (software pipeline).
    {
      constexpr int NW = DNAS_T / 64;
      // DNAS_PIPE 1: the gathers of a row are issued when its turn comes, so a value crosses every
      // forward edge (source row < destination row) within one sweep -- the plan lays chains out
      // along ascending rows.  2: one row of read-ahead (a hop then needs two rows of distance).
      constexpr int PIPE = DNAS_PIPE;
      const int wv = tid >> 6, ln = tid & 63;
      for (;;) {
        asm volatile("" ::: "memory");   // other waves write LDS between sweeps: reload everything
        const unsigned e0 = __hip_atomic_load(epochL, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (ln == 0) __hip_atomic_store(&idleL[wv], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        int changed = 0;
        double buf[PIPE][kMaxRowVals > 0 ? kMaxRowVals : 1];
        auto issue = [&](auto kc) {
          constexpr int k = kc.value, o = rowOffset(k), b = k % PIPE;
#ifdef DNAS_DIAG_NO_LDS_READS   // timing experiment: what do the sweeps cost without their gathers
          static_for<0, rowEE(k) + 2 * rowEN(k)>([&](auto ec) { buf[b][ec.value] = __uint_as_float(E[o]) > 3.f ? 1.0 : kNegInf; });
#else
          static_for<0, rowEE(k)>([&](auto ec) { buf[b][ec.value] = ldsRead(ldsB, E[o + ec.value]); });
          static_for<0, rowEN(k)>([&](auto ec) {
            const unsigned addr = E[o + rowEE(k) + ec.value];
            buf[b][rowEE(k) + 2 * ec.value] = ldsRead(ldsB, addr);
            buf[b][rowEE(k) + 2 * ec.value + 1] = ldsRead(ldsB, addr + kCellBytes);
          });
#endif
        };
        if constexpr (PIPE == 2) issue(IntC<0>{});
        static_for<0, K>([&](auto kc) {
          constexpr int k = kc.value, o = rowOffset(k), b = k % PIPE;
          if constexpr (PIPE == 1) issue(kc);
          else if constexpr (k + 1 < K) issue(IntC<k + 1>{});
          double s = S[k], d = Dv[k];
          static_for<0, rowEE(k)>([&](auto ec) {
            constexpr int cls = emitClass(k, ec.value);
            double v = buf[b][ec.value];
            if constexpr (cls != 0) v = v + scoreC[cls];
            d = dmax(d, v);
          });
          static_for<0, rowEN(k)>([&](auto ec) {
            constexpr int cls = nullClass(k, ec.value);
            double vd = buf[b][rowEE(k) + 2 * ec.value], vs = buf[b][rowEE(k) + 2 * ec.value + 1];
            if constexpr (cls != 0) { vd = vd + scoreC[cls]; vs = vs + scoreC[cls]; }
            d = dmax(d, vd);
            s = dmax(s, vs);
          });
          s = dmax(s, d + a.delEnd);
          if (s != S[k] || d != Dv[k]) {
            changed = 1;
            S[k] = s;
            Dv[k] = d;
            const double xv = dmax(d + a.delExtend, s + a.delOpen);
            X[k * T + tid] = xv;
            static_for<0, kRows[k].ec>([&](auto ec) {
              const unsigned en = E[o + rowEE(k) + rowEN(k) + kRows[k].ep + ec.value];
              if (ENT_HASCELL(en)) {   // a heavy destination's cell also receives pushes: only ever raise it
                ldsMax(ldsB, ENT_ADDR(en) + kCellBytes, s);
                ldsMax(ldsB, ENT_ADDR(en), d);
              }
            });
            static_for<0, kRows[k].ep>([&](auto ec) {
              const unsigned en = E[o + rowEE(k) + rowEN(k) + ec.value];
              const double sc = ldsRead(ldsB, kTabBase + ENT_SCOFF(en));
              if (ENT_FLAG(en)) {
                ldsMax(ldsB, ENT_ADDR(en), xv + sc);
              } else {
                ldsMax(ldsB, ENT_ADDR(en), d + sc);
                ldsMax(ldsB, ENT_ADDR(en) + kCellBytes, s + sc);
              }
            });
          }
        });
        ++rounds;
        if (__any(changed)) {
          // the data writes above precede the bump (LDS operations of one wave execute in order)
          if (ln == 0) __hip_atomic_fetch_add(epochL, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
          continue;
        }
        if (__hip_atomic_load(epochL, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != e0) continue;
        if (ln == 0) __hip_atomic_store(&idleL[wv], e0 + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        bool done = false;
        for (;;) {   // every wave reaches this exit: once all are idle nobody sweeps, so nobody bumps the epoch
          const unsigned v = ln < NW ? __hip_atomic_load(&idleL[ln], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) : e0 + 1u;
          if (__all(v == e0 + 1u)) { done = true; break; }
          if (__hip_atomic_load(epochL, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != e0) break;   // someone grew a cell: sweep again
          __builtin_amdgcn_s_sleep(1);
        }
        if (done) break;
      }
      __syncthreads();   // all waves are out of the sweeps before phase C reuses X[]
    }
    STAMP(tB)

    // ---- phase C: the column's S and D lanes go to HBM in slot order (coalesced); S is parked
    // in X[] for the next column's gathers; heavy cells are reset.  The duplication lanes
    // T1..TD (viterbi.cpp:105-106,161-168) are NOT stored: T(pos,q) is a function of this
    // state's own S(pos), S(pos-1), ... S(pos-(D-1)) and the read,
    //     T(p,q) = max( T(p-1,q+1) + sub[ctx[q+1]][x_p],  (S(p)+tanDup)+len[q] ),   T(0,.) = -inf,
    // and fp max/+ satisfy max(a,b)+c == max(a+c,b+c) exactly, so evaluating the chain from
    // its deepest element reproduces the reference's cell bit for bit.  Only T1(pos) is needed
    // here (it feeds S of the next column, viterbi.cpp:101-103); the traceback kernel rebuilds
    // any T cell it visits the same way.  The S history comes back from HBM/L2 (this thread
    // wrote it), a group of rows per memory latency.
    {
      // the D lane leaves first: its registers then carry the T1 hand-over (one store latency ahead
      // of the history loads, instead of 28 more live registers)
      STORE_LANE(pos, 1, Dv)
      const int xn = pos < L ? seq[pos] : 0;
      int xh[D_ > 0 ? D_ : 1];   // xh[i] = x_{pos-i}
      static_for<0, D_>([&](auto ic) { xh[ic.value] = pos - ic.value >= 1 ? seq[pos - ic.value - 1] : 0; });
      // rows 2m, 2m+1 of a thread are neighbours in the lattice: 16-byte loads and stores
#ifndef DNAS_CGROUP
#define DNAS_CGROUP 4
#endif
      constexpr int G = DNAS_CGROUP;               // rows per load group (even: whole pairs)
      static_for<0, (K + G - 1) / G>([&](auto gc) {
        constexpr int k0 = gc.value * G, k1 = (k0 + G < K) ? k0 + G : K;
        unsigned metaG[G];
        static_for<k0, k1>([&](auto kc) { metaG[kc.value - k0] = META(kc.value); });
        double sh[G][D_ > 1 ? D_ - 1 : 1];
        static_for<1, D_>([&](auto ic) {
          constexpr int i = ic.value;
          if (pos - i >= 1) {
            static_for<k0 / 2, k1 / 2>([&](auto mc) {
              constexpr int m2 = mc.value;
#ifdef DNAS_DIAG_NO_HIST   // timing experiment
              dbl2 v2; v2.x = S[2 * m2]; v2.y = S[2 * m2 + 1];
#else
              const dbl2 v2 = reinterpret_cast<const dbl2*>(col - (size_t)i * lanes * NS + (size_t)m2 * 2 * T)[tid];
#endif
              sh[2 * m2 - k0][i - 1] = v2.x;
              sh[2 * m2 + 1 - k0][i - 1] = v2.y;
            });
          }
        });
        static_for<k0, k1>([&](auto kc) {
          constexpr int k = kc.value, o = rowOffset(k);
          const double s = S[k];
          const int mdl = (int)(metaG[k - k0] & 15u);
          X[k * T + tid] = s;
          static_for<0, kRows[k].ec>([&](auto ec) {
            const unsigned en = E[o + rowEE(k) + rowEN(k) + kRows[k].ep + ec.value];
            if (ENT_HASCELL(en) && ENT_FLAG(en)) {
              ldsWrite(ldsB, ENT_ADDR(en) + kCellBytes, kNegInf);
              ldsWrite(ldsB, ENT_ADDR(en), kNegInf);
            }
          });
          // T1(pos): chain from the deepest element (i = D-1) to i = 0.  Nearly every state has a
          // full context (mdl == D) and nearly every column a full history: that case is straight
          // line code, chosen per wave.
          const bool valid = (metaG[k - k0] & 0x20000000u) != 0;
          double v = kNegInf;
          if (pos >= D_ && __all(mdl == D_ || !valid)) {
            if constexpr (D_ > 0) {
              double sd = s;
              if constexpr (D_ > 1) sd = sh[k - k0][D_ - 2];
              v = (sd + a.tanDup) + a.len[D_ - 1];
              static_for<1, D_>([&](auto jc) {
                constexpr int i = D_ - 1 - jc.value;         // i = D-2 .. 0
                double sp = s;
                if constexpr (i > 0) sp = sh[k - k0][i - 1];
                v = dmax(v + ldsRead(ldsB, kTabBase + 32 + ((metaG[k - k0] >> (4 + 2 * (i + 1))) & 3u) * 32 + xh[i] * 8),
                         (sp + a.tanDup) + a.len[i]);
              });
            }
          } else {
            static_for<0, D_>([&](auto jc) {
              constexpr int i = D_ - 1 - jc.value;           // lane q = i at column p = pos - i
              if (i < mdl && pos - i >= 1) {
                double sp = s;
                if constexpr (i > 0) sp = sh[k - k0][i - 1];
                const double base = (sp + a.tanDup) + a.len[i];
                if (i + 1 < mdl && pos - i - 1 >= 1)
                  v = dmax(v + subL[((metaG[k - k0] >> (4 + 2 * (i + 1))) & 3u) * 4 + xh[i]], base);
                else
                  v = base;
              }
            });
          }
          // next column: S >= T1(pos) + sub[ctx1][x_{pos+1}]   (viterbi.cpp:101-103)
          Dv[k] = (valid && mdl > 0) ? v + subL[((metaG[k - k0] >> 4) & 3u) * 4 + xn] : kNegInf;   // D(pos) is already on its way to HBM
        });
      });
      // the S lane goes out last: a wave's memory operations return in order, so a store issued
      // between two groups would sit in front of the next group's history loads
      STORE_LANE(pos, 0, S)
    }
    __syncthreads();   // X[] now holds S(pos) for everyone
    STAMP(tC)
  }
#ifdef DNAS_STAMP
  if (tid == 0 && blockIdx.x == 0) { roundsTotal[1] = tA; roundsTotal[2] = tP; roundsTotal[3] = tB; roundsTotal[4] = tC; roundsTotal[5] = (unsigned long long)rounds; }
#endif

  // ---- loglike (viterbi.h:102); local mode overwrites the end state with the column max
  // (viterbi.cpp:171-173).  bit30 of meta marks the reference's last state.
  double* const red = lds + (kTabBase / 8) + 28;
  double* const lastS = lat + (size_t)L * lanes * NS;
  double best = kNegInf;
  static_for<0, K>([&](auto kc) {
    constexpr int k = kc.value;
    const unsigned mt = META(k);
    if (a.local ? (mt & 0x20000000u) != 0 : (mt & 0x40000000u) != 0) best = dmax(best, S[k]);
  });
  for (int off = 32; off > 0; off >>= 1) best = dmax(best, __shfl_down(best, off, 64));
  if ((tid & 63) == 0) red[tid >> 6] = best;
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < T / 64; ++w) best = dmax(best, red[w]);
    red[0] = best;
    outLoglike[read] = best;
    atomicAdd(roundsTotal, (unsigned long long)rounds);
  }
  __syncthreads();
  if (a.local) {
    static_for<0, K>([&](auto kc) {
      constexpr int k = kc.value;
      if (META(k) & 0x40000000u) lastS[(k >> 1) * 2 * T + 2 * tid + (k & 1)] = red[0];
    });
  }
}
