// viterbi_tiera.hip -- the fast lattice-fill kernel ("tier A"), specialised at JIT time
// for one machine (see host/plan.hpp).  Replaces the ViterbiMatrix constructor's fill loop
// (reference src/viterbi.cpp:62-176) for machines whose state set fits one CU.
//
// One work-group (DNAS_T threads, one per CU) per read.  Thread t owns the states in
// (row k, lane t), k = 0..K-1; their S and D cells of the current column live in REGISTERS
// for the whole in-column fixpoint.  Like the reference's worklist (viterbi.cpp:110-159) the
// recursion is PUSHED along the out-edges: every state owns an LDS accumulator
//     DC[k*T + t]   what its in-edges have offered to its D cell       (all states)
//     SC[stripe]    what its null in-edges have offered to its S cell  (rows with such states)
// and a state whose cells grew offers  max(D+delExtend, S+delOpen)+score  to the DC of its emit
// successors,  D+score / S+score  to DC / SC of its null successors (ds_max_f64).  A sweep
// is: every thread walks its rows in order, reads the row's own accumulators (conflict-free:
// consecutive lanes, consecutive cells), and pushes only where a cell grew.  The out-edge
// lists are compiled by the host into per-thread 32-bit entries that are loaded into registers
// once per read; the row shape (entries per row, S stripe) is a compile-time constant, so
// every register index is static.  A column is: push S of the previous column along the emit
// edges (viterbi.cpp:92-95) -> read it back as this column's S -> sweep until no cell grows
// (monotone max-plus: the least fixpoint is schedule independent, so the cells equal the
// reference's worklist result bit for bit) -> duplication lanes -> coalesced stores.
//
// fp64 throughout, reference operand order, no contraction / fast-math.
//
// Machines beyond one CU ("tier C", DNAS_G > 1): a CLUSTER of DNAS_G work-groups shares a read; member g
// owns a part of the states (host/plan.cpp cuts the machine along its depth-first walk).  Edges inside a
// member work as above.  An EDGE into another member owns a cell of that member's INBOX in the cluster's
// exchange buffer in global memory -- a mailbox: the thread that holds the edge's source state is its only
// writer and stores what it offers (the value only grows within a column: no atomic), and thread t of the owner
// folds cell r*T + t into the destination state's LDS accumulators once per sweep (sc1 loads issued behind the
// last row of a sweep, ds_max in front of the first row of the next).  The launcher either places a cluster on one
// XCD (blockIdx b and b+8 share one) or -- large clusters, of which whole ones per XCD would leave CUs idle -- deals its
// members over the XCDs (`spread`: neighbouring blocks; runtime.hip decides, option cluster_spread).  The members compare
// their XCC ids at the first barrier: on one XCD the stores are plain -- the cells live in that XCD's L2 and never travel
// to memory --, a split cluster stores write-through (sc1), so the protocol is correct wherever the work-groups run (a
// cluster that was meant to share an XCD and does not, included).  Termination is agreed in two
// levels: inside a work-group as before, across the cluster through a device-scope epoch GE (bumped after
// every batch of exchange offers has completed) and one idle word per member.
//
// Global memory -- tables, lattice columns, exchange buffer -- is addressed the buffer way (descriptor in scalar
// registers + 32-bit lane offset): no specialisation of this file keeps a pointer in vector registers or spills.
//
// Compile-time parameters (-D):  DNAS_T threads, DNAS_K rows, DNAS_D dup lanes,
//   DNAS_NS slots per member (= K*T), DNAS_SROWS S stripes, DNAS_NCLS distinct edge scores,
//   DNAS_G members per cluster, DNAS_GROWS inbox slots per thread, the first DNAS_GSROWS of them with an S cell,
//   DNAS_POLLS (tier C, optional) how a sweep looks into the inbox, DNAS_SEGMENTS=1 the bounded-memory build,
//   DNAS_ROWS  brace list of {out-edge entries (-1: row left empty), S stripe or -1, kind, cls, full, gOut}
//              per row; what all entries of a row have in common is not decoded per lane.
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>   // hiprtc provides the device runtime implicitly
#endif

#ifndef DNAS_T
#error "compile with -DDNAS_T= -DDNAS_K= -DDNAS_D= -DDNAS_NS= -DDNAS_SROWS= -DDNAS_NCLS= -DDNAS_G= -DDNAS_GROWS= -DDNAS_ROWS="
#endif

// kind: 1 emit edges only, 2 null edges only, 0 both; cls: common score class or -1; full: no empty entry;
// gOut: 0 every entry of the row points into LDS, 1 every entry into another member's inbox, 2 mixed (bit 2
// of the entry tells)
struct RowShape { int nOut, sIdx, kind, cls, full, gOut; };
constexpr RowShape kRows[DNAS_K] = {DNAS_ROWS};

constexpr bool rowLive(int k) { return kRows[k].nOut >= 0; }      // nOut -1: the plan left the row empty
constexpr int rowOut(int k) { return kRows[k].nOut > 0 ? kRows[k].nOut : 0; }
constexpr int rowOffset(int k) {
  int o = 0;
  for (int i = 0; i < k; ++i) o += rowOut(i);
  return o;
}
constexpr int kEntries = rowOffset(DNAS_K) > 0 ? rowOffset(DNAS_K) : 1;

typedef double dbl2 __attribute__((ext_vector_type(2)));
static_assert(DNAS_K % 2 == 0, "rows come in pairs");
// Which two rows of a thread share a 16-byte cell pair of the lattice (DNAS_PAIRS = a0,b0,a1,b1,...: the plan pairs rows that
// are about equally full, so that few threads move a pair for one state; default: rows 2m and 2m+1).
#ifdef DNAS_PAIRS
constexpr int kPairRows[DNAS_K] = {DNAS_PAIRS};
#else
struct PairIdentity { int v[DNAS_K]; constexpr PairIdentity() : v() { for (int i = 0; i < DNAS_K; ++i) v[i] = i; } };
constexpr PairIdentity kPairIdentity{};
#define kPairRows kPairIdentity.v
#endif
constexpr int pairRow(int m2, int side) { return kPairRows[2 * m2 + side]; }
constexpr int pairSlotOfRow(int k) {      // 2 * pair + side of row k
  for (int i = 0; i < DNAS_K; ++i) if (kPairRows[i] == k) return i;
  return -1;
}

template <int V> struct IntC { static constexpr int value = V; };
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(IntC<I>{});
    static_for<I + 1, N>(f);
  }
}


#ifndef DNAS_G
#define DNAS_G 1
#define DNAS_GROWS 0
#endif
#ifndef DNAS_GSROWS
#define DNAS_GSROWS DNAS_GROWS
#endif
#ifndef DNAS_POLLS
#define DNAS_POLLS 0   // tier C: 0 = the inbox is loaded behind the last row of a sweep and folded in front of the first row of the
#endif                 // next one; n > 0 = n polls spread over the rows of a sweep, each folding what the one before it loaded
#ifndef DNAS_POLL_LAG
#define DNAS_POLL_LAG 1   // ... and a poll folds what the poll DNAS_POLL_LAG polls before it loaded (DNAS_POLLS is a multiple of it)
#endif

// kernel-argument block (mirrors runtime.hip TierAArgs)
struct TierAArgs {
  int N;            // real states
  int local;
  double noGap, delOpen, delExtend, delEnd, tanDup;
  double sub[16];
  double len[8];
  double score[4];  // score table, score[0] == 0
};

// LDS map (bytes):  SC[SROWS*T] | pad | DC[NS] | score[4] | sub[16] | len[8] | red[T/64] | epoch, idle[T/64], done, ge, abort, pend (u32)
constexpr int kDCBase = DNAS_SROWS * DNAS_T * 8 + 64;
constexpr int kTabBase = kDCBase + DNAS_NS * 8;

// entries (host/plan.cpp packs them), one per out-edge; every field is one or two VALU
// operations away from its use.  Destination in LDS:
//   [0:2)   score class
//   [3:18)  byte address of the destination's DC cell, >> 3       ->  en & 0x3fff8
//   [19:32) null edge: index of the destination's SC cell          ->  (en >> 16) & 0xfff8 is its byte address
//           emit edge: 0x1ffc | emitted base                       ->  en >= 0xffe00000
// destination in another member's inbox (bit 2 set):
//   [0:2) score class | [3:23) inbox cell -> en & 0x7ffff8 is its byte offset | bit 23 null edge | [24:26) emitted base
//   0: no edge
#define ENT_VALID(e) ((e) != 0u)
#define ENT_EMIT(e) ((e) >= 0xffe00000u)
#define ENT_DC(e) ((e) & 0x3fff8u)
#define ENT_SC(e) (((e) >> 16) & 0xfff8u)
#define ENT_CLS(e) ((e) & 3u)
#define ENT_BASE32(e) (((e) >> 14) & 0x60u)   // emitted base * 32: byte offset of its row in the sub table
#define ENT_GLOBAL(e) (((e) & 4u) != 0u)
#define ENT_GCELL(e) ((e) & 0x7ffff8u)
#define ENT_GNULL(e) (((e) & 0x800000u) != 0u)
#define ENT_GBASE32(e) (((e) >> 19) & 0x60u)

// v_max_f64 directly: no NaN can occur here (only -inf + finite / -inf + -inf), so the
// canonicalising copies the compiler would add around fmax() are pure overhead
__device__ __forceinline__ double dmax(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// LDS by byte address.  The kernel has no static LDS, so the dynamic block starts at address 0 (checked at kernel entry) and an
// accumulator's address is the number the plan computed: no base is added to it.
typedef __attribute__((address_space(3))) double lds_f64;
__device__ __forceinline__ lds_f64* ldsAt(unsigned byteOff) { return (lds_f64*)(__UINTPTR_TYPE__)byteOff; }
__device__ __forceinline__ double ldsRead(unsigned byteOff) { return *ldsAt(byteOff); }
__device__ __forceinline__ void ldsWrite(unsigned byteOff, double v) { *ldsAt(byteOff) = v; }
__device__ __forceinline__ void ldsMax(unsigned byteOff, double v) {
  __hip_atomic_fetch_max(ldsAt(byteOff), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ double ldsMaxRtn(unsigned byteOff, double v) {   // returns what the cell held
  return __hip_atomic_fetch_max(ldsAt(byteOff), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// the DC address of an entry, straight from its register (one instruction; the statement is volatile so that the 28+ decoded
// addresses are not hoisted out of the column loop and kept in registers)
__device__ __forceinline__ unsigned entDc(unsigned e) {
  unsigned r;
  asm volatile("v_and_b32 %0, 0x3fff8, %1" : "=v"(r) : "v"(e));
  return r;
}
// Global memory is addressed the buffer way -- a uniform base in scalar registers (a resource descriptor), a 32-bit
// per-lane byte offset and a uniform byte offset: the address arithmetic of the tables, the lattice columns and the
// exchange buffer runs on the scalar unit and no 64-bit pointer sits in vector registers (as global_* accesses the
// 28-row program held 39 hoisted pointers and spilled 55 registers).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rsrc_t makeRsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)0x80000000u, 0x00020000);   // raw buffer; a lane whose offset is kLaneOff is out of range: it reads 0 and touches no memory
}
constexpr unsigned kLaneOff = 0x80000000u;
constexpr int kAuxNt = 2, kAuxSc1 = 16;   // cache policy bits of a buffer access (gfx940+: bit 0 sc0, bit 1 nt, bit 4 sc1)
__device__ __forceinline__ unsigned bufLoadU32(rsrc_t r, unsigned laneOff, unsigned uniOff) {
  return __builtin_amdgcn_raw_buffer_load_b32(r, (int)laneOff, (int)uniOff, 0);
}
// exchange buffer: every cell has ONE writer at a time (the edge's source thread during a column, the owner when it clears
// the cell between columns; the cluster's barriers order the two) and is read with sc1 loads, which bypass this CU's L1 and
// are served by the L2.  onXcd (uniform): the whole cluster shares one L2, the stores are plain and stay there; otherwise
// they are written through to memory (sc1), where the reader's sc1 load finds them.  Either way a hand-over needs no fence.
__device__ __forceinline__ double xLoad(rsrc_t r, unsigned laneOff, unsigned uniOff) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)laneOff, (int)uniOff, kAuxSc1));
}
__device__ __forceinline__ void xStore(rsrc_t r, bool onXcd, unsigned laneOff, unsigned uniOff, double v) {
  if (onXcd) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, (int)laneOff, (int)uniOff, 0);
  else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, (int)laneOff, (int)uniOff, kAuxSc1);
}
__device__ __forceinline__ unsigned wLoad(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// keep a register-resident entry opaque inside the column loop, so that nothing decoded from
// it is hoisted out and kept live across iterations
__device__ __forceinline__ unsigned opaque(unsigned v) {
  asm volatile("" : "+v"(v));
  return v;
}

constexpr double kNegInf = -__builtin_huge_val();
constexpr double kFresh = __builtin_huge_val();   // "not evaluated in this column yet" in a D register

#ifndef DNAS_SLEEP
#define DNAS_SLEEP 1   // idle waves poll the termination words every 64 * DNAS_SLEEP cycles
#endif
#ifndef DNAS_SEGMENTS
#define DNAS_SEGMENTS 0   // 1: the launch fills a range of columns per read (bounded-memory decode), see colRange
#endif
#ifndef DNAS_NT_H
#define DNAS_NT_H 1   // how many of the oldest history columns are streamed
#endif
#ifndef DNAS_NT_D
#define DNAS_NT_D 1
#endif
// Cluster constants.  Exchange buffer of one cluster: XA.dc | XB.dc | XB.sc, kCells doubles each, cell
// (member * GROWS + r) * T + t = inbox cell r*T + t of that member (XA: the emit offers between columns,
// XB: the offers of the in-column fixpoint -- two sets, because a fast member is already offering into the
// fixpoint while a slow one still folds its between-column cells), then two sets of G cells for the end-of-read
// reduction (one per member, by read parity).  Sync block of one cluster (64 u32): [0] GE, [1] abort,
// [2, 2+G) idle word per member, [40] placement census (OR of 1 << XCC id).
constexpr int G_ = DNAS_G;
constexpr bool kEarlyOffers = DNAS_G == 1;   // one work-group per read: a column's emit offers are made inside phase C of the column before
constexpr unsigned kCells = (unsigned)DNAS_G * DNAS_GROWS * DNAS_T;
constexpr unsigned kXStride = 3u * kCells + 2u * ((unsigned)DNAS_G + 15u & ~15u);   // doubles per cluster (kCells is a multiple of 64: every array starts a 128-byte line)

// DNAS_WAVES_PER_EU (optional, the planner's choice for small row programs): the register budget that lets this many waves share
// a SIMD, i.e. TWO work-groups of 1024 threads share a CU at 8
#ifdef DNAS_WAVES_PER_EU
#define DNAS_OCCUPANCY __attribute__((amdgpu_waves_per_eu(DNAS_WAVES_PER_EU, DNAS_WAVES_PER_EU)))
#else
#define DNAS_OCCUPANCY
#endif
extern "C" __global__ void __launch_bounds__(DNAS_T) DNAS_OCCUPANCY
viterbi_fill_tiera(TierAArgs a, const unsigned* __restrict__ entTab,   // [G][kEntries][T]
                   const unsigned* __restrict__ metaTab,                // [G][K][T]: mdl | ctx<<4 | flags
                   const unsigned char* __restrict__ bases, const unsigned long long* __restrict__ readOff,
                   const int* __restrict__ batchRead, const unsigned long long* __restrict__ slotOff,
                   double* __restrict__ arena, double* __restrict__ outLoglike,
                   unsigned long long* __restrict__ roundsTotal,
                   double* __restrict__ xbuf, unsigned* __restrict__ syncWords, const unsigned* __restrict__ foldTab,   // [G][GROWS][T]
                   int nClusters, int nReads, unsigned long long timeoutTicks,   // watchdog per lattice column (100 MHz ticks)
                   unsigned long long arriveTicks,    // ... and for the members of a cluster to have all started
                   const int* __restrict__ colRange,     // [nReads][2] first and last column to fill, or null: 0 .. L
                   int spread) {                         // tier C: 1 = the members of a cluster are NEIGHBOURING blocks (dealt over the XCDs: 21-member clusters, tests)
  extern __shared__ double lds[];
  extern __shared__ unsigned ldsU[];
  constexpr int T = DNAS_T, K = DNAS_K, D_ = DNAS_D, lanes = 2, NSm = DNAS_NS, NS = DNAS_NS * DNAS_G;   // stored lanes: S, D
  const int tid = threadIdx.x;
  if ((unsigned)(unsigned long long)(void*)lds != 0u) __builtin_trap();   // (see ldsAt: byte addresses are absolute)
  const double* const subL = lds + (kTabBase / 8) + 4;
  // the vote words live in the same dynamic LDS block; a second extern array (same base) keeps
  // the accesses in the LDS address space (a volatile generic pointer would turn them into
  // flat_* operations that wait on every outstanding global store)
  unsigned* const epochL = ldsU + 2 * ((kTabBase / 8) + 28 + DNAS_T / 64);   // termination words: epoch, idle[T/64], done, ge, abort
  unsigned* const idleL = epochL + 1;
  unsigned* const doneL = idleL + DNAS_T / 64;
  unsigned* const geL = doneL + 1;
  unsigned* const abortL = doneL + 2;
  unsigned* const pendL = doneL + 3;      // clusters: waves whose exchange offers have completed since wave 0 last told the cluster (below)

  // ---- who am I: tier A one work-group per read; tier C member `member` of cluster `cluster`, which
  // walks the reads cluster, cluster + nClusters, ...  Blocks b and b + 8 land on the same XCD (observed
  // dispatch order, speed only): the members of a cluster are 8 blocks apart.
  int member = 0, rFirst = (int)blockIdx.x, rStep = 1, rEnd = (int)blockIdx.x + 1;
  const double* xBase = xbuf;  // this cluster's exchange buffer
  unsigned* SY = nullptr;      // this cluster's sync block
  if constexpr (G_ > 1) {
    // (spread: block b is member b % G of cluster b / G -- consecutive blocks go to different XCDs, so every cluster is split
    //  over several; the protocol does not care, only the exchange is slower: tests/test_gpu_tier_c.py)
    const int b = (int)blockIdx.x, q = b >> 3;
    member = spread ? b % G_ : q % G_;
    const int cluster = spread ? b / G_ : (q / G_) * 8 + (b & 7);
    if (cluster >= nClusters) return;
    rFirst = cluster; rStep = nClusters; rEnd = nReads;
    xBase = xbuf + (size_t)cluster * kXStride;
    SY = syncWords + (size_t)cluster * 64;
    entTab += (size_t)member * kEntries * T;
    metaTab += (size_t)member * K * T;
    foldTab += (size_t)member * DNAS_GROWS * T;
    if (tid == 0) {
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      __hip_atomic_fetch_or(&SY[40], 1u << (xcc & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  constexpr unsigned kXA = 0u, kXBd = kCells * 8u, kXBs = 2u * kCells * 8u, kXRed = 3u * kCells * 8u;
  const rsrc_t xB = makeRsrc(xBase), rEnt = makeRsrc(entTab), rMeta = makeRsrc(metaTab), rFold = makeRsrc(foldTab);
  const unsigned tid4 = (unsigned)tid * 4u, tid8 = (unsigned)tid * 8u, tid16 = (unsigned)tid * 16u;
  // own inbox cells: rows 2m, 2m+1 of a thread are neighbours (cell (r, t) of a member sits at (r/2)*2T + 2t + (r&1), the lattice's
  // layout), so a pair of rows is one 16-byte load -- array base + X_PAIR_U(m) uniform, tid16 per lane
  static_assert(DNAS_GROWS % 2 == 0, "inbox rows come in pairs");
  bool onXcd = false;      // the cluster's members share an XCD (known after the first barrier; no exchange store precedes it)
#define X_PAIR_U(m2) ((unsigned)(((unsigned)member * DNAS_GROWS + 2u * (unsigned)(m2)) * T) * 8u)
  auto xLoad2 = [&](unsigned arrayOff, auto mc, double& lo, double& hi) {
    const dbl2 v2 = __builtin_bit_cast(dbl2, __builtin_amdgcn_raw_buffer_load_b128(xB, (int)tid16, (int)(arrayOff + X_PAIR_U(mc.value)), kAuxSc1));
    lo = v2.x; hi = v2.y;
  };
  auto xClear2 = [&](unsigned arrayOff, auto mc) {
    dbl2 v2;
    v2.x = kNegInf; v2.y = kNegInf;
    if (onXcd) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v2), xB, (int)tid16, (int)(arrayOff + X_PAIR_U(mc.value)), 0);
    else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v2), xB, (int)tid16, (int)(arrayOff + X_PAIR_U(mc.value)), kAuxSc1);
  };
  // where slot r*T + tid folds into: byte addresses of the state's DC and SC cells (0 / 0x7fff8: none)
  constexpr int R_ = DNAS_GROWS > 0 ? DNAS_GROWS : 1;
  unsigned FT[R_];
  unsigned pairUsed = 0;   // bit m: some cell of inbox rows 2m, 2m+1 of this WAVE has a writer (uniform; the other pairs are never loaded)
  if constexpr (G_ > 1) {
    static_for<0, DNAS_GROWS>([&](auto rc) { FT[rc.value] = bufLoadU32(rFold, tid4, (unsigned)rc.value * T * 4u); });
    static_for<0, DNAS_GROWS / 2>([&](auto mc) { if (__any((FT[2 * mc.value] | FT[2 * mc.value + 1]) != 0u)) pairUsed |= 1u << mc.value; });
  }
#define FOLD_DC(f) (((f) & 0xffffu) << 3)
#define FOLD_SC(f) (((f) >> 16) << 3)
#define FOLD_HAS_SC(f) ((f) != 0u && ((f) >> 16) != 0xffffu)

  // the out-edge entries: in registers for the whole launch
  unsigned E[kEntries];
  static_for<0, kEntries>([&](auto m) { E[m.value] = bufLoadU32(rEnt, tid4, (unsigned)m.value * T * 4u); });
#define ENTRY(i) opaque(E[i])
#define META(k) bufLoadU32(rMeta, tid4, (unsigned)(k) * T * 4u)
  // score of an edge by class: class 0 is 0.0 (adding it is the identity on every value that occurs)
  auto withScore = [&](double v, unsigned cls) -> double {
    if constexpr (DNAS_NCLS <= 1) return v;
    else if constexpr (DNAS_NCLS == 2) return cls ? v + a.score[1] : v;
    else return v + ldsRead(kTabBase + cls * 8);
  };
  // ... of an entry of row k: the row's common class is a compile-time constant
  auto withScoreRow = [&](auto kc, double v, unsigned en) -> double {
    constexpr int c = kRows[kc.value].cls;
    if constexpr (c == 0) return v;
    else if constexpr (c > 0) return v + a.score[c];
    else return withScore(v, ENT_CLS(en));
  };
  // own accumulators: byte addresses
  const unsigned ownB = (unsigned)tid * 8u;
#define DC_OWN(k) (ownB + (unsigned)kDCBase + (unsigned)(k) * T * 8u)
#define SC_OWN(k) (ownB + (unsigned)kRows[k].sIdx * T * 8u)

  double S[K], Dv[K];   // after phase C, Dv[k] carries the T1 hand-over to the next column's phase A
  unsigned rounds = 0;
#ifdef DNAS_STAMP   // diagnostic build: where does a column spend its cycles (never in the shipped kernel)
  unsigned long long tA = 0, tP = 0, tB = 0, tC = 0, tX = 0, tW = 0, t0 = 0, t1 = 0, tw0 = 0;
#define STAMP(acc) { t1 = __builtin_amdgcn_s_memtime(); acc += t1 - t0; t0 = t1; }
#else
#define STAMP(acc)
#endif

  // LDS init: every accumulator -inf, then the small tables
  for (int i = tid; i < kTabBase / 8; i += T) lds[i] = kNegInf;
  if (tid < 4) lds[kTabBase / 8 + tid] = a.score[tid];
  if (tid < 16) lds[kTabBase / 8 + 4 + tid] = a.sub[tid];
  if (tid < 8) lds[kTabBase / 8 + 20 + tid] = a.len[tid];
  if (tid < 5 + DNAS_T / 64) epochL[tid] = 0;
  __syncthreads();

  // The S and D lanes of column p leave for HBM from the registers, 16 bytes per lane (rows 2m and
  // 2m+1 of a thread are lattice neighbours).
  // (column p of this member starts at latM + p * lanes * NS doubles: a descriptor per column; the lane and the pair are
  //  the uniform offset, 16 * tid the lane's)
#define COL_RSRC(p) makeRsrc(latM + (size_t)(p) * lanes * NS)
#define PAIR_OFF(lane, m2) ((unsigned)(((unsigned)(lane) * NS + (unsigned)(m2) * 2u * T) * 8u))
#define STORE_LANE(p, lane, REG)                                                                 \
  {                                                                                              \
    const rsrc_t colp = COL_RSRC(p);                                                             \
    static_for<0, K / 2>([&](auto mc) {                                                          \
      constexpr int m2 = mc.value;                                                               \
      if (pairValid & (1u << m2)) {                                                              \
        dbl2 v2;                                                                                 \
        v2.x = REG[pairRow(m2, 0)]; v2.y = REG[pairRow(m2, 1)];                                  \
        /* the D lane is not read again by this kernel: past the caches */                       \
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v2), colp, (int)tid16, (int)PAIR_OFF(lane, m2), \
                                               ((lane) == 1 && DNAS_NT_D) ? kAuxNt : 0);         \
      }                                                                                          \
    });                                                                                          \
  }
  // ... and come back from it when a read is resumed at a checkpoint (bounded-memory decode, runtime.hip)
#define LOAD_LANE(p, lane, REG)                                                                  \
  {                                                                                              \
    const rsrc_t colp = COL_RSRC(p);                                                             \
    static_for<0, K / 2>([&](auto mc) {                                                          \
      constexpr int m2 = mc.value;                                                               \
      dbl2 v2;                                                                                   \
      v2.x = kNegInf; v2.y = kNegInf;                                                            \
      if (pairValid & (1u << m2))                                                                \
        v2 = __builtin_bit_cast(dbl2, __builtin_amdgcn_raw_buffer_load_b128(colp, (int)tid16, (int)PAIR_OFF(lane, m2), 0)); \
      REG[pairRow(m2, 0)] = v2.x; REG[pairRow(m2, 1)] = v2.y;                                    \
    });                                                                                          \
  }
  unsigned pairValid = 0;   // bit m: the lattice pair (rows 2m, 2m+1) of this thread holds a real state
  static_for<0, K / 2>([&](auto mc) {
    if ((META(pairRow(mc.value, 0)) | META(pairRow(mc.value, 1))) & 0x20000000u) pairValid |= 1u << mc.value;
  });

  // ---- cluster synchronisation (tier C).  GE only grows.  geBase = its value when the cluster last agreed
  // (every member holds the same number); a column starts with one bump per member, which is the
  // barrier behind the between-column offers AND what makes the idle words of the previous column stale.
  unsigned geBase = 0, colSeq = 0, xDirty = 0;
  bool aborted = false;
  unsigned long long tStart = 0ull;   // watchdog: a column that takes longer than timeoutTicks (100 MHz) aborts the launch
  // The first barrier of a launch is where a cluster's members find each other.  A member may not have been STARTED yet (the
  // work-groups of another kernel, of another process, of this model's own traceback hold its CU): that is no fault of the
  // protocol, and it is waited for far longer (arriveTicks) than a lattice column may take once everybody is there.
  bool arrived = false;
  // all offers of this work-group into the exchange buffer have completed -> bump -> wait for every member
  auto clusterBarrier = [&]() {
    if constexpr (G_ > 1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's exchange atomics and clears are done
      __syncthreads();
      geBase += (unsigned)G_;
      if (tid < 64) {
        if (tid == 0) __hip_atomic_fetch_add(&SY[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (unsigned spin = 0;; ++spin) {
          const unsigned v = tid < 2 ? wLoad(&SY[tid]) : 0u;
          const unsigned ge = __shfl(v, 0, 64), ab = __shfl(v, 1, 64);
          if ((int)(ge - geBase) >= 0) break;
          if (ab || ((spin & 255u) == 255u && __builtin_amdgcn_s_memrealtime() - tStart > (arrived ? timeoutTicks : arriveTicks))) {
            if (tid == 0) { __hip_atomic_store(&SY[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); *abortL = 1u; }
            break;
          }
          if (arrived) __builtin_amdgcn_s_sleep(2); else __builtin_amdgcn_s_sleep(20);
        }
      }
      __syncthreads();
      if (!arrived) {
        // every member has OR-ed its XCC id into the census word before it bumped GE: one bit set = one XCD = one L2
        const unsigned census = wLoad(&SY[40]);
        onXcd = __builtin_amdgcn_readfirstlane((census & (census - 1u)) == 0u ? 1 : 0) != 0;
      }
      arrived = true;
      aborted = *abortL != 0u;
    }
  };

  for (int r = rFirst; r < rEnd && !aborted; ) {
  const int read = batchRead[r];
  const unsigned char* seq = bases + readOff[read];
  const int L = (int)(readOff[read + 1] - readOff[read]);
  double* const lat = arena + slotOff[r];
  double* const latM = lat + (size_t)member * NSm;   // this member's slots of a column
  const unsigned redOff = kXRed + (unsigned)(((r - rFirst) / rStep) & 1) * ((unsigned)G_ + 15u & ~15u) * 8u;   // this read's reduction cells

  // Bounded-memory decode: this launch fills columns c0 .. c1 of the read only.  A segment that stops short of the read's
  // end leaves, next to its last column, what the next column's phase A needs besides S(c1): the T1 hand-over, in the D
  // lane of column c1 + 1 (which the launch that resumes there overwrites with the real D lane afterwards).
  // Only the kernel built with -DDNAS_SEGMENTS=1 knows about segments: the one that fills whole reads stays as lean as it was.
#if DNAS_SEGMENTS
  int c0 = 0, c1 = L;
  if (colRange) { c0 = colRange[2 * r]; c1 = colRange[2 * r + 1]; }
  if (c0 > 0) {
    LOAD_LANE(c0 - 1, 0, S)
    LOAD_LANE(c0, 1, Dv)
  }
#else
  constexpr int c0 = 0;
  const int c1 = L;
#endif

  bool earlyOffered = false;     // the offers of the column about to start have been made already (phase C of the column before)
  for (int pos = c0; pos <= c1 && !aborted; ++pos) {
    const int x = pos > 0 ? seq[pos - 1] : 0;
    ++colSeq;
    if constexpr (G_ > 1) tStart = __builtin_amdgcn_s_memrealtime();
#ifdef DNAS_STAMP
    t0 = __builtin_amdgcn_s_memtime();
#endif

    // ---- phase A (viterbi.cpp:75-79,92-95,101-103): S of this column.  Every state offers
    // ((S(pos-1) + score) + noGap) + sub[base][x] along its emit edges into the destinations'
    // DC (all -inf since phase C); after the barrier each state takes what it was offered,
    // folds in the T1 lane (handed over in the D registers by phase C) and clears the cell for
    // the D offers of the fixpoint.
    auto emitOffers = [&](int xCol) {
      const unsigned subRow = (unsigned)kTabBase + 32u + (unsigned)xCol * 8u;   // &sub[0][x]
      static_for<0, K>([&](auto kc) {
        constexpr int k = kc.value, o = rowOffset(k);
        if (rowLive(k) && S[k] > kNegInf) {
          static_for<0, rowOut(k)>([&](auto ec) {
            const unsigned en = ENTRY(o + ec.value);
            if constexpr (kRows[k].kind == 2) return;             // no emit edge in this row
            if constexpr (kRows[k].gOut != 0) {
              if (kRows[k].gOut == 1 ? ENT_VALID(en) : ENT_GLOBAL(en)) {
                if (kRows[k].kind == 1 || !ENT_GNULL(en))
                  xStore(xB, onXcd, ENT_GCELL(en), kXA, (withScoreRow(kc, S[k], en) + a.noGap) + ldsRead(subRow + ENT_GBASE32(en)));
                return;
              }
              if constexpr (kRows[k].gOut == 1) return;
            }
            const bool emit = kRows[k].kind == 1 ? (kRows[k].full != 0 || ENT_VALID(en)) : ENT_EMIT(en);
            if (emit) ldsMax(ENT_DC(en), (withScoreRow(kc, S[k], en) + a.noGap) + ldsRead(subRow + ENT_BASE32(en)));
          });
        }
      });
    };
    // (one work-group per read: the offers of this column were made inside phase C of the column before, while its first
    //  history loads were in flight, and the barrier that ends phase C has seen them land -- see there)
    const bool offersMade = kEarlyOffers && earlyOffered;
    if (pos > 0 && !offersMade) emitOffers(x);
    if constexpr (G_ > 1) {
      STAMP(tA)
      clusterBarrier();            // every member's offers of the previous column have landed
      STAMP(tX)
      if (aborted) break;
    } else if (pos > 0 && !offersMade) {
      __syncthreads();             // every offer of the previous column has landed
    }
    STAMP(tA)
    if (pos > 0) {
      if constexpr (G_ > 1) {
        // what the other members offered lands in the LDS cells of the states it was meant for
        double xa[R_];
        static_for<0, DNAS_GROWS / 2>([&](auto mc) {   // all loads in flight together
          constexpr int m2 = mc.value;
          xa[2 * m2] = kNegInf; xa[2 * m2 + 1] = kNegInf;
          if (pairUsed & (1u << m2)) xLoad2(kXA, mc, xa[2 * m2], xa[2 * m2 + 1]);
        });
        static_for<0, DNAS_GROWS / 2>([&](auto mc) {
          constexpr int m2 = mc.value;
          if (xa[2 * m2] > kNegInf) ldsMax(FOLD_DC(FT[2 * m2]), xa[2 * m2]);
          if (xa[2 * m2 + 1] > kNegInf) ldsMax(FOLD_DC(FT[2 * m2 + 1]), xa[2 * m2 + 1]);
          if (xa[2 * m2] > kNegInf || xa[2 * m2 + 1] > kNegInf) xClear2(kXA, mc);      // nobody offers here again before the next column's barrier
        });
        __syncthreads();
      }
      static_for<0, K>([&](auto kc) {
        constexpr int k = kc.value;
        if constexpr (rowLive(k)) {
          S[k] = dmax(ldsRead(DC_OWN(k)), Dv[k]);   // Dv: T1(pos-1) + sub[ctx1][x_pos], left there by phase C
          ldsWrite(DC_OWN(k), kNegInf);
          Dv[k] = kFresh;
        }
      });
    } else {
      static_for<0, K>([&](auto kc) {
        constexpr int k = kc.value;
        const unsigned mt = META(k);    // bit29: real state, bit31: reference state 0
        S[k] = ((mt & 0x20000000u) && (a.local || (mt & 0x80000000u))) ? 0.0 : kNegInf;   // viterbi.cpp:75-79
        Dv[k] = rowLive(k) ? kFresh : kNegInf;   // an empty row stays (-inf, -inf) for the whole read
      });
    }
    __syncthreads();   // every DC is cleared before the first offer of the fixpoint
    STAMP(tP)

    // ---- phase B: sweeps to the fixpoint (viterbi.cpp:97-99,110-159), WITHOUT a barrier per
    // sweep.  Every wave keeps sweeping its own rows (chaotic relaxation: a monotone max-plus
    // system reaches the same least fixpoint under any schedule) and the work-group agrees on
    // termination through LDS words:
    //     epoch    bumped by a wave whose sweep grew a cell
    //     idle[w]  = e+1 once wave w has finished a sweep that grew nothing and saw epoch == e
    //              from its first read to its last
    // When all waves are idle at the same epoch, nothing was written while each of them swept:
    // every cell is consistent with its inputs, i.e. the fixpoint.  A row's accumulators are
    // read when its turn comes, so a value crosses every forward edge (source row < destination
    // row) within one sweep -- the plan lays the machine's chains out along ascending rows.
    //
    // Cluster: the same one level up.  GE is bumped once the exchange offers of a wave have completed: the wave says so in LDS
    // (`pend`, in its next sweep, behind an s_waitcnt vmcnt(0); a sweep that offered something is never the idle one) and wave 0
    // of the work-group passes it on with ONE device-scope atomic per sweep -- sixteen members of eight waves each bumping GE
    // themselves made 1 157 atomics per lattice column on one word, one every 36 ns: the word's atomic unit was what a lone read's
    // sweep waited for (s_waitcnt vmcnt(0) again before wave 0 may call the work-group idle, so that its bump has landed; it looks
    // at `pend` once more there, behind the idle words of the other waves).  Wave 0 reads GE in every sweep and turns a change into a bump of the work-group's
    // epoch: every wave then sweeps once more, i.e. reads its exchange cells AFTER that value of GE was
    // seen.  A work-group whose waves are all idle, and whose wave 0 still reads the GE it last imported,
    // writes GE+1 into its idle word; when every member's word says GE+1 and GE is still the same,
    // every offer ever made was read by its owner in a sweep that grew nothing: the fixpoint.
    {
      constexpr int NW = DNAS_T / 64;
      const int wv = tid >> 6, ln = tid & 63;
      unsigned geSeen = geBase;
      bool pendingBump = false;
      xDirty = 0;
      // How a sweep looks into the inbox.  Every load of a cell goes to the memory side (the atomics of the other members are
      // performed there), and a wait for load data also waits for every older memory operation of the wave -- its own exchange
      // atomics included --, so each look costs a stall of about a microsecond whatever its place.  The default (DNAS_POLLS 0)
      // takes one per sweep where it hurts least and helps most: the cells are loaded behind the last row of a sweep and
      // folded in front of the first row of the next, so that what lands during a sweep is in its state's accumulator before
      // that state's row is evaluated again (13.7 sweeps per column on the 46 670-state machine against 17.3 with the load at
      // the start and the fold at the end of the same sweep; tools/cluster_sim.py predicts both).  DNAS_POLLS = n > 0: n
      // polls spread over the rows, each folding what the poll before it loaded (fewer sweeps still -- 12.1 at n = 4 -- but a
      // stall per poll: slower in all).  Whatever the mode, a wave that thinks itself idle first folds the loads it ISSUED IN
      // THAT SWEEP (the confirmation below): the cluster's termination rests on every wave having looked, after the last value of
      // GE was imported, at cells loaded after that import.
      constexpr bool kSplit = DNAS_POLLS == 0;
      constexpr int kPollStride = kSplit ? K + 1 : (K + DNAS_POLLS - 1) / (DNAS_POLLS > 0 ? DNAS_POLLS : 1);
      constexpr int RS_ = DNAS_GSROWS > 0 ? DNAS_GSROWS : 1;
      double xd[1][R_], xs[1][RS_], lastD[R_], lastS[RS_];   // last*: what has been folded (the cells were cleared in phase C)
      static_for<0, R_>([&](auto rc) { lastD[rc.value] = kNegInf; xd[0][rc.value] = kNegInf; });
      static_for<0, RS_>([&](auto rc) { lastS[rc.value] = kNegInf; xs[0][rc.value] = kNegInf; });
      for (;;) {
        asm volatile("" ::: "memory");   // other waves write LDS between sweeps: reload everything
        const unsigned e0 = __hip_atomic_load(epochL, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (ln == 0) __hip_atomic_store(&idleL[wv], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        int changed = 0, sentX = 0;
        unsigned geNow = geSeen;
        if constexpr (G_ > 1) { if (wv == 0) geNow = wLoad(&SY[0]); }
        auto loadInbox = [&](auto setc) {
          constexpr int q = setc.value;
          static_for<0, DNAS_GROWS / 2>([&](auto mc) {
            constexpr int m2 = mc.value;
            if (pairUsed & (1u << m2)) {
              xLoad2(kXBd, mc, xd[q][2 * m2], xd[q][2 * m2 + 1]);
              if constexpr (2 * m2 + 1 < DNAS_GSROWS) xLoad2(kXBs, mc, xs[q][2 * m2], xs[q][2 * m2 + 1]);
              else if constexpr (2 * m2 < DNAS_GSROWS) xs[q][2 * m2] = xLoad(xB, tid16, kXBs + X_PAIR_U(m2));
            }
          });
        };
        auto foldInbox = [&](auto setc) {
          constexpr int q = setc.value;
          // An inbox cell only grows within a column, and lastD / lastS hold what this thread has folded of its cells: a poll that
          // finds nothing new -- most do -- costs the loads and one comparison per cell.  Unused slots (FT == 0) are real cells
          // that nobody offers into: they read -inf for ever, so the loads need no mask.  A cell that raises its state's
          // accumulator counts like an offer of this wave.
          bool anyNew = false;
          static_for<0, DNAS_GROWS>([&](auto rc) {
            constexpr int r = rc.value;
            anyNew = anyNew || xd[q][r] != lastD[r];
            if constexpr (r < DNAS_GSROWS) anyNew = anyNew || xs[q][r] != lastS[r];
          });
          if (__any(anyNew)) {
            static_for<0, DNAS_GROWS>([&](auto rc) {
              constexpr int r = rc.value;
              if (xd[q][r] != lastD[r]) { lastD[r] = xd[q][r]; if (ldsMaxRtn(FOLD_DC(FT[r]), xd[q][r]) < xd[q][r]) changed = 1; }
              if constexpr (r < DNAS_GSROWS) {
                if (xs[q][r] != lastS[r]) { lastS[r] = xs[q][r]; if (ldsMaxRtn(FOLD_SC(FT[r]), xs[q][r]) < xs[q][r]) changed = 1; }
              }
            });
          }
        };
        if constexpr (G_ > 1 && kSplit) foldInbox(IntC<0>{});      // what was loaded behind the last row of the sweep before
        // One row of the sweep; dIn / scIn: its accumulators as read.
        auto rowEval = [&](auto kc, double dIn, double scIn) {
          constexpr int k = kc.value, o = rowOffset(k);
          if constexpr (!rowLive(k)) return;

          // D is exactly what the in-edges have offered; a row with no S cells can only move when D
          // moved.  The first sweep of a column finds Dv == kFresh (no D cell is ever +inf) and offers
          // the starting values.
          double d = dIn, s = S[k];
          constexpr bool sFed = kRows[k].sIdx >= 0;   // something offers into S
          if constexpr (sFed) s = dmax(s, scIn);
          bool grew = d != Dv[k];
          if constexpr (sFed) grew = grew || s != S[k];
          if (grew) {
            changed = 1;
            s = dmax(s, d + a.delEnd);                                 // viterbi.cpp:114-115
            S[k] = s;
            Dv[k] = d;
            const double xv = dmax(d + a.delExtend, s + a.delOpen);    // viterbi.cpp:124
            // a row of one score class adds the score once, ahead of the first atomic, not per entry
            constexpr bool oneCls = kRows[k].cls >= 0;
            [[maybe_unused]] double xvC = xv, dC = d, sC = s;
            if constexpr (oneCls && kRows[k].cls > 0) {
              if constexpr (kRows[k].kind != 2) xvC = xv + a.score[kRows[k].cls];
              if constexpr (kRows[k].kind != 1) { dC = d + a.score[kRows[k].cls]; sC = s + a.score[kRows[k].cls]; }
            }
            auto scored = [&](double plain, double withCls, unsigned en) -> double {
              if constexpr (oneCls) return withCls; else return withScore(plain, ENT_CLS(en));
            };
            static_for<0, rowOut(k)>([&](auto ec) {
              if constexpr (kRows[k].kind == 1 && kRows[k].gOut == 0) {
                // emit edges into LDS only: the address is all that is needed of the entry (an empty entry decodes to address 0,
                // which is no accumulator)
                const unsigned dc = entDc(E[o + ec.value]);
                const double v = scored(xv, xvC, oneCls ? 0u : ENTRY(o + ec.value));
                if (kRows[k].full != 0 || dc != 0u) ldsMax(dc, v);
                return;
              }
              const unsigned en = ENTRY(o + ec.value);
              if (kRows[k].full != 0 || ENT_VALID(en)) {
                if constexpr (kRows[k].gOut != 0) {
                  if (kRows[k].gOut == 1 || ENT_GLOBAL(en)) {
                    sentX = 1;
                    if (kRows[k].kind == 1 || (kRows[k].kind != 2 && !ENT_GNULL(en))) {
                      xStore(xB, onXcd, ENT_GCELL(en), kXBd, scored(xv, xvC, en));
                    } else {                                           // viterbi.cpp:137-151
                      xStore(xB, onXcd, ENT_GCELL(en), kXBd, scored(d, dC, en));
                      xStore(xB, onXcd, ENT_GCELL(en), kXBs, scored(s, sC, en));
                    }
                    return;
                  }
                }
                if constexpr (kRows[k].gOut != 1) {
                  if constexpr (kRows[k].kind == 1) {
                    const double v = scored(xv, xvC, en);
                    ldsMax(ENT_DC(en), v);
                  } else if constexpr (kRows[k].kind == 2) {            // viterbi.cpp:137-151
                    const double vd = scored(d, dC, en), vs = scored(s, sC, en);
                    ldsMax(ENT_DC(en), vd); ldsMax(ENT_SC(en), vs);
                  } else if (ENT_EMIT(en)) {
                    const double v = scored(xv, xvC, en);
                    ldsMax(ENT_DC(en), v);
                  } else {
                    const double vd = scored(d, dC, en), vs = scored(s, sC, en);
                    ldsMax(ENT_DC(en), vd); ldsMax(ENT_SC(en), vs);
                  }
                }
              }
            });
          }
        };
        static_for<0, K>([&](auto kc) {
          constexpr int k = kc.value;
          if constexpr (G_ > 1 && !kSplit && k % kPollStride == 0) { foldInbox(IntC<0>{}); loadInbox(IntC<0>{}); }
          if constexpr (!rowLive(k)) return;
          double sc = kNegInf;
          const double d = ldsRead(DC_OWN(k));
          if constexpr (kRows[k].sIdx >= 0) sc = ldsRead(SC_OWN(k));
          rowEval(kc, d, sc);
        });
        ++rounds;
        if constexpr (G_ > 1) {
          // the exchange offers of the sweep before have had a sweep to complete: GE may say so now (the wait is for the
          // stragglers, and for this sweep's own offers -- which is why the inbox loads of this sweep go out BEHIND it: in front
          // of it the wait was for their round trip as well, in every sweep that followed one with offers)
          if (pendingBump) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (ln == 0) __hip_atomic_fetch_add(pendL, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            pendingBump = false;
          }
          if (__any(sentX)) pendingBump = true;   // GE is bumped once these offers have completed: in the next sweep
          if constexpr (kSplit) loadInbox(IntC<0>{});
          // wave 0 tells the cluster what the work-group's waves have reported (the count it reads is taken off, what comes in
          // behind it stays for the next sweep)
          if (wv == 0) {
            const unsigned pendSeen = __hip_atomic_load(pendL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (pendSeen != 0u && ln == 0) {
              __hip_atomic_fetch_sub(pendL, pendSeen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              __hip_atomic_fetch_add(&SY[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
          if (wv == 0 && geNow != geSeen) { geSeen = geNow; changed = 1; }   // import: everybody sweeps once more
          // confirmation: nothing moved in this sweep -- then the cells loaded IN this sweep must show nothing new either
          if (!__any(changed)) foldInbox(IntC<0>{});
        }
        if (__any(changed)) {
          // the offers above precede the bump (LDS operations of one wave execute in order)
          if (ln == 0) __hip_atomic_fetch_add(epochL, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
          continue;
        }
        if (__hip_atomic_load(epochL, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != e0) continue;
        if constexpr (G_ > 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's bumps of GE have landed
        if (ln == 0) __hip_atomic_store(&idleL[wv], e0 + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        bool done = false;
        for (;;) {   // every wave reaches this exit: once all are idle nobody sweeps, so nobody bumps the epoch
          const unsigned v = ln < NW ? __hip_atomic_load(&idleL[ln], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) : e0 + 1u;
          if (__all(v == e0 + 1u)) {
            if constexpr (G_ == 1) { done = true; break; }
            // the work-group is quiet; wave 0 asks the cluster, the others wait for its verdict
            if (wv != 0) {
              for (;;) {
                if (__hip_atomic_load(doneL, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == colSeq) { done = true; break; }
                if (__hip_atomic_load(epochL, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != e0) break;
                __builtin_amdgcn_s_sleep(4);
              }
              break;
            }
            bool declared = false;
#ifdef DNAS_STAMP
            tw0 = __builtin_amdgcn_s_memtime();
#endif
            {
              // every wave of the work-group is idle: what they reported last (before their idle words) goes out first -- GE then
              // differs from what was imported, and the loop below sends everybody through one more sweep
              const unsigned pnd = __hip_atomic_load(pendL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              if (pnd != 0u) {
                if (ln == 0) {
                  __hip_atomic_fetch_sub(pendL, pnd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                  __hip_atomic_fetch_add(&SY[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              }
            }
            for (unsigned spin = 0;; ++spin) {
              const unsigned w = ln < 2 + G_ ? wLoad(&SY[ln]) : 0u;
              const unsigned ge = __shfl(w, 0, 64), ab = __shfl(w, 1, 64);
              if (ab || ((spin & 255u) == 255u && __builtin_amdgcn_s_memrealtime() - tStart > timeoutTicks)) {
                if (ln == 0) {
                  __hip_atomic_store(&SY[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  *abortL = 1u;
                  __hip_atomic_store(doneL, colSeq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                done = true;
                break;
              }
              if (ge != geSeen) {          // somebody offered since: import and sweep again
                geSeen = ge;
                if (ln == 0) __hip_atomic_fetch_add(epochL, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                break;
              }
              if (!declared) {
                if (ln == 0) __hip_atomic_store(&SY[2 + member], ge + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                declared = true;
              } else if (__all(ln < 2 || ln >= 2 + G_ || w == ge + 1u)) {
                // every member idle at ge: the verdict stands if GE has not moved since those words were read.  GE and
                // up to 14 idle words share one 64-byte line, which one wave-wide load reads as a unit: the snapshot
                // itself shows GE == ge; larger clusters read GE once more.
                if (G_ <= 14 || wLoad(&SY[0]) == ge) {
                  if (ln == 0) {
                    *geL = ge;
                    __hip_atomic_store(doneL, colSeq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                  }
                  done = true;
                  break;
                }
              }
              __builtin_amdgcn_s_sleep(2);
            }
#ifdef DNAS_STAMP
            tW += __builtin_amdgcn_s_memtime() - tw0;
#endif
            break;
          }
          if (__hip_atomic_load(epochL, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != e0) break;   // someone grew a cell: sweep again
          __builtin_amdgcn_s_sleep(DNAS_SLEEP);
        }
        if (done) break;
      }
      // the inbox cells that were offered into (every offer was folded: the fixpoint was agreed) are cleared in phase C
      static_for<0, DNAS_GROWS>([&](auto rc) {
        constexpr int r = rc.value;
        if (lastD[r] > kNegInf) xDirty |= 1u << r;
        if constexpr (r < DNAS_GSROWS) { if (lastS[r] > kNegInf) xDirty |= 0x10000u << r; }
      });
      __syncthreads();   // all waves are out of the sweeps before phase C clears the accumulators
      if constexpr (G_ > 1) {
        aborted = *abortL != 0u;
        geBase = *geL;
        if (aborted) break;
      }
    }
    STAMP(tB)

    // ---- phase C: the column's S and D lanes go to HBM in slot order (coalesced); the
    // accumulators are cleared for the next column's offers.  The duplication lanes
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
      if constexpr (G_ > 1) {
        // the inbox cells that were offered into are cleared for the next column's fixpoint (every offer was seen by
        // the last sweep; nobody offers into XB again before the next column's barrier)
        static_for<0, DNAS_GROWS / 2>([&](auto mc) {
          constexpr int m2 = mc.value;
          if (xDirty & (3u << (2 * m2))) xClear2(kXBd, mc);
          if constexpr (2 * m2 < DNAS_GSROWS) { if (xDirty & (0x30000u << (2 * m2))) xClear2(kXBs, mc); }
        });
      }
      const int xn = pos < L ? seq[pos] : 0;
      if constexpr (kEarlyOffers) {
        // every accumulator is cleared NOW (not row by row below), so that the offers of the next column can go out while
        // the history of the first row group is on its way
        static_for<0, K>([&](auto kc) {
          constexpr int k = kc.value;
          if constexpr (rowLive(k)) {
            ldsWrite(DC_OWN(k), kNegInf);
            if constexpr (kRows[k].sIdx >= 0) ldsWrite(SC_OWN(k), kNegInf);
          }
        });
        __syncthreads();
        earlyOffered = pos < c1;
      }
      int xh[D_ > 0 ? D_ : 1];   // xh[i] = x_{pos-i}
      static_for<0, D_>([&](auto ic) { xh[ic.value] = pos - ic.value >= 1 ? seq[pos - ic.value - 1] : 0; });
      // The two rows of a cell pair are neighbours in the lattice (rows 2m, 2m+1; clusters: the plan's choice, DNAS_PAIRS): 16-byte
      // loads and stores.  The history comes back in GROUPS of DNAS_CGROUP rows, DNAS_CDEPTH groups in flight: the loads of group
      // g + DNAS_CDEPTH go out when group g has been turned into its hand-over, so that only the first group's latency is
      // waited for in full (all waves of a work-group are here at the same time: nothing else hides it).
      // Measured (profiles/experiments/r4_ab_runs.txt): one work-group per read 4 rows x 1 group (0.640; 2 x 2: 0.632, 2 x 3: 0.624 --
      // a CU's memory path, not the latency of one group, bounds the phase), clusters 2 rows x 3 groups (0.260 against 0.253).
#ifndef DNAS_CGROUP
#define DNAS_CGROUP (DNAS_G > 1 ? 2 : 4)
#endif
#ifndef DNAS_CDEPTH
#define DNAS_CDEPTH (DNAS_G > 1 ? 3 : 1)
#endif
      {
      constexpr int GP = DNAS_CGROUP / 2;          // cell pairs per load group
      static_assert(GP >= 1 && DNAS_CDEPTH >= 1, "a load group is at least one pair");
      constexpr int NG = (K / 2 + GP - 1) / GP, PD = DNAS_CDEPTH < NG ? DNAS_CDEPTH : NG;
      unsigned metaBuf[PD][2 * GP];                // a ring of PD groups: group g lives in slot g % PD
      double shBuf[PD][2 * GP][D_ > 1 ? D_ - 1 : 1];
      auto issueGroup = [&](auto gc) __attribute__((always_inline)) {
        constexpr int p0 = gc.value * GP, p1 = (p0 + GP < K / 2) ? p0 + GP : K / 2;
        static_for<2 * p0, 2 * p1>([&](auto qc) {   // (address rebuilt here: 14 hoisted pointers would cost 28 registers)
          constexpr int k = kPairRows[qc.value];
          metaBuf[gc.value % PD][qc.value - 2 * p0] = rowLive(k) ? META(k) : 0u;
        });
        // clusters: a thread without a state in a pair is switched off by its offset (it reads 0, which nobody uses, and moves no
        // bytes) -- formed here, from an opaque copy of the mask, so that the fourteen offsets are not kept in registers; one
        // work-group per read: every thread loads (switching threads off costs more than the bytes)
        const unsigned pv = G_ > 1 ? opaque(pairValid) : ~0u;
        static_for<1, D_>([&](auto ic) {
          constexpr int i = ic.value;
          // (unconditional loads, so that a slot of the ring is dead from its last use to its next load: in the first columns of a
          //  read, where column pos - i does not exist, they fetch column pos instead and nobody uses what they bring)
          const int colH = pos - i >= 1 ? pos - i : pos;
          static_for<p0, p1>([&](auto mc) {
            constexpr int m2 = mc.value;
            if constexpr (!rowLive(pairRow(m2, 0)) && !rowLive(pairRow(m2, 1))) return;
            const unsigned off = (G_ == 1 || (pv & (1u << m2))) ? tid16 : kLaneOff;
            // the oldest column is read for the last time: stream it past the caches
            const dbl2 v2 = __builtin_bit_cast(dbl2, __builtin_amdgcn_raw_buffer_load_b128(COL_RSRC(colH), (int)off, (int)PAIR_OFF(0, m2),
                                                                                            (DNAS_NT_H && i >= D_ - DNAS_NT_H) ? kAuxNt : 0));
            shBuf[gc.value % PD][2 * (m2 - p0)][i - 1] = v2.x;
            shBuf[gc.value % PD][2 * (m2 - p0) + 1][i - 1] = v2.y;
          });
        });
      };
      auto computeGroup = [&](auto gc) __attribute__((always_inline)) {
        constexpr int p0 = gc.value * GP, p1 = (p0 + GP < K / 2) ? p0 + GP : K / 2;
        static_for<2 * p0, 2 * p1>([&](auto qc) {
          constexpr int k = kPairRows[qc.value], q = qc.value - 2 * p0, b = gc.value % PD;
          if constexpr (!rowLive(k)) { Dv[k] = kNegInf; return; }
          const double s = S[k];
          const unsigned mt = metaBuf[b][q];
          const int mdl = (int)(mt & 15u);
          if constexpr (!kEarlyOffers) {
            ldsWrite(DC_OWN(k), kNegInf);
            if constexpr (kRows[k].sIdx >= 0) ldsWrite(SC_OWN(k), kNegInf);
          }
          // T1(pos): chain from the deepest element (i = D-1) to i = 0.  Nearly every state has a
          // full context (mdl == D) and nearly every column a full history: that case is straight
          // line code, chosen per wave.
          const bool valid = (mt & 0x20000000u) != 0;
          double v = kNegInf;
          if (pos >= D_ && __all(mdl == D_ || !valid)) {
            if constexpr (D_ > 0) {
              double sd = s;
              if constexpr (D_ > 1) sd = shBuf[b][q][D_ - 2];
              v = (sd + a.tanDup) + a.len[D_ - 1];
              static_for<1, D_>([&](auto jc) {
                constexpr int i = D_ - 1 - jc.value;         // i = D-2 .. 0
                double sp = s;
                if constexpr (i > 0) sp = shBuf[b][q][i - 1];
                v = dmax(v + ldsRead(kTabBase + 32 + ((mt >> (4 + 2 * (i + 1))) & 3u) * 32 + xh[i] * 8),
                         (sp + a.tanDup) + a.len[i]);
              });
            }
          } else {
            static_for<0, D_>([&](auto jc) {
              constexpr int i = D_ - 1 - jc.value;           // lane q = i at column p = pos - i
              if (i < mdl && pos - i >= 1) {
                double sp = s;
                if constexpr (i > 0) sp = shBuf[b][q][i - 1];
                const double base = (sp + a.tanDup) + a.len[i];
                if (i + 1 < mdl && pos - i - 1 >= 1)
                  v = dmax(v + subL[((mt >> (4 + 2 * (i + 1))) & 3u) * 4 + xh[i]], base);
                else
                  v = base;
              }
            });
          }
          // next column: S >= T1(pos) + sub[ctx1][x_{pos+1}]   (viterbi.cpp:101-103)
          Dv[k] = (valid && mdl > 0) ? v + subL[((mt >> 4) & 3u) * 4 + xn] : kNegInf;   // D(pos) is already on its way to HBM
        });
      };
      static_for<0, PD>([&](auto gc) { issueGroup(gc); });
      if constexpr (kEarlyOffers) {
        if (earlyOffered) emitOffers(xn);       // column pos + 1: ((S(pos) + score) + noGap) + sub[base][x_{pos+1}]
      }
      static_for<0, NG>([&](auto gc) {
        computeGroup(gc);
        if constexpr (gc.value + PD < NG) issueGroup(IntC<gc.value + PD>{});
      });
      }
      // the S lane goes out last: a wave's memory operations return in order, so a store issued
      // between two groups would sit in front of the next group's history loads
      STORE_LANE(pos, 0, S)
    }
    __syncthreads();   // every accumulator is -inf again
    STAMP(tC)
  }
  if (aborted) break;
#if DNAS_SEGMENTS
  if (c1 < L) {                // a segment: park the hand-over, no log-likelihood yet
    STORE_LANE(c1 + 1, 1, Dv)
    if constexpr (G_ > 1) { r += rStep; continue; }
    else break;
  }
#endif
#ifdef DNAS_STAMP
  if (tid == 0 && blockIdx.x == 0) { roundsTotal[1] = tA; roundsTotal[2] = tP; roundsTotal[3] = tB; roundsTotal[4] = tC; roundsTotal[5] = (unsigned long long)rounds; roundsTotal[6] = tX; roundsTotal[7] = tW; }
#endif

  // ---- loglike (viterbi.h:102); local mode overwrites the end state with the column max
  // (viterbi.cpp:171-173).  bit30 of meta marks the reference's last state.
  double* const red = lds + (kTabBase / 8) + 28;
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
    if constexpr (G_ > 1) xStore(xB, onXcd, (unsigned)member * 8u, redOff, best);
    else { red[0] = best; outLoglike[read] = best; }
  }
  if constexpr (G_ > 1) {
    clusterBarrier();          // every member's best has landed (and the last column's accumulators are clear)
    if (aborted) break;
    if (tid < 64) {
      double b = tid < G_ ? xLoad(xB, tid8, redOff) : kNegInf;
      for (int g = 64; g < G_; g += 64) b = dmax(b, tid + g < G_ ? xLoad(xB, tid8 + (unsigned)g * 8u, redOff) : kNegInf);
      for (int off = 32; off > 0; off >>= 1) b = dmax(b, __shfl_down(b, off, 64));
      if (tid == 0) red[0] = b;
    }
  }
  __syncthreads();
  {
    bool owner = false;
    static_for<0, K>([&](auto kc) {
      constexpr int k = kc.value;
      if (META(k) & 0x40000000u) {
        owner = true;
        if (a.local) {
          // the reference overwrites S(N-1, L) AFTER the duplication lanes of the last column were formed
          // from it (viterbi.cpp:161-173); those lanes are not stored here, so the value they came from is
          // kept in the spare cell behind the read's lattice for whoever rebuilds them (expand_lattice_kernel)
          // (buffer stores: the two addresses live in scalar registers, not in a vector register pair kept for the whole read)
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, S[k]), makeRsrc(lat + (size_t)(L + 1) * lanes * NS), 0, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, red[0]), COL_RSRC(L), (int)tid16,
                                                (int)(PAIR_OFF(0, pairSlotOfRow(k) >> 1) + (unsigned)(pairSlotOfRow(k) & 1) * 8u), 0);
        }
      }
    });
    if constexpr (G_ > 1) { if (owner) outLoglike[read] = red[0]; }
  }
  __syncthreads();             // red[] is free again
  r += rStep;
  }   // reads of this work-group / cluster
  if (tid == 0) atomicAdd(roundsTotal, (unsigned long long)rounds);
}
