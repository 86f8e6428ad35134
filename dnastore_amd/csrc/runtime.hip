// Device runtime behind the C ABI: model upload, lattice arena, batch scheduling and
// kernel launches for the Viterbi path (include/dnastore_amd.h, "device side").
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/dnastore_amd.h"
#include "device_model.h"
#include "errors.hpp"
#include "host/plan.hpp"
#include <cmath>

#include "jit.hpp"

extern "C" __global__ void copy_rows_kernel(double*, const double*, size_t, size_t, size_t);
extern "C" __global__ void viterbi_fill_kernel(DevModel, const uint8_t*, const uint64_t*, const int32_t*,
                                               const uint64_t*, double*, double*, unsigned long long*, int, const int*);
extern "C" __global__ void expand_lattice_kernel(DevModel, const uint8_t*, const double*, double*);
extern "C" __global__ void viterbi_traceback_wave_kernel(DevModel, const uint8_t*, const uint64_t*, const int32_t*,
                                                         const uint64_t*, const double*, char*, const uint64_t*,
                                                         uint32_t*, uint8_t*, int, unsigned long long*, const uint64_t*, uint32_t*,
                                                         const int*, TracebackWalk*);
extern "C" __global__ void fill_neginf_kernel(double*, size_t);
extern "C" __global__ void sync_latency_kernel(unsigned*, int, int, int, unsigned long long*, unsigned*);
extern "C" __global__ void xcc_probe_kernel(unsigned*);
extern "C" __global__ void check_bases_kernel(const uint8_t*, size_t, unsigned long long*);
extern "C" __global__ void viterbi_traceback_kernel(DevModel, const uint8_t*, const uint64_t*, const int32_t*,
                                                    const uint64_t*, const double*, char*, const uint64_t*,
                                                    uint32_t*, uint8_t*, int, unsigned long long*, const uint64_t*, uint32_t*, int);

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return dnas::fail(DNAS_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));     \
  } while (0)

// kernel-argument block of viterbi_fill_tiera (must mirror csrc/viterbi_tiera.hip)
struct TierAArgs {
  int N, local;
  double noGap, delOpen, delExtend, delEnd, tanDup;
  double sub[16];
  double len[8];
  double score[4];
};
struct TierALaunch {
  TierAArgs a;
  const unsigned* entTab;
  const unsigned* metaTab;
  const uint8_t* bases;
  const uint64_t* readOff;
  const int32_t* batchRead;
  const uint64_t* slotOff;
  double* arena;
  double* outLoglike;
  unsigned long long* roundsTotal;
  // tier C (a cluster of work-groups per read): exchange buffers, sync blocks, clusters in this launch, reads in this launch
  double* xbuf;
  unsigned* syncWords;
  const unsigned* foldTab;
  int nClusters;
  int nReads;
  unsigned long long timeoutTicks;
  unsigned long long arriveTicks;
  const int* colRange;          // bounded-memory decode: [nReads][2] first and last column of this launch (null: whole reads)
  int spread;                   // tier C: 1 = a cluster's members are neighbouring blocks (dealt over the XCDs; option cluster_spread)
};

struct dnas_model {
  int device = 0;
  int tier = 0;                 // 1 = tier A (register/LDS-resident JIT kernel, one work-group per read), 2 = tier C (the same
                                // kernel, a cluster of work-groups per read), 0 = tier B (global-memory kernel)
  // bounded-memory decode: reads whose lattice does not fit the arena (checkpoint=auto), or every read (always), are
  // filled in segments of `segmentCols` columns (0: chosen from the arena) from checkpoints, twice -- see DESIGN.md 3.7
  int checkpointMode = 0;       // 0 auto, 1 always, 2 never
  int segmentCols = 0;
  int64_t lastCheckpointed = 0; // reads (in sorted order: the longest) of the last call that went that way
  int* dColRange = nullptr;
  uint64_t* dSegSlot = nullptr;
  TracebackWalk* dWalks = nullptr;
  bool waveTraceback = true;    // one wave per read for batches of up to 256 reads (option traceback=thread: never)
  int tbThreads = kTraceThreads;     // threads per block of the thread-per-read traceback (option tb_threads)
  int tbLanes = kTraceLanes;         // reads per wave there (option tb_lanes): the other lanes idle
  int maxClusters = 1;          // tier C: clusters that fit the GPU at once
  double* dXbuf = nullptr;      // tier C: exchange buffers, one per cluster
  unsigned* dSync = nullptr;    // tier C: sync blocks (64 u32 per cluster)
  unsigned* dSyncBase = nullptr;   // ... and the allocation they sit in: the blocks are placed inside it by measured latency
  std::string syncPlaceNote;
  std::vector<double> syncLat;     // [8 XCCs][syncCand] measured round trips (ticks), by the XCC id the measuring block ran on; empty: not measured
  int syncCand = 0;
  size_t syncOffNow = 0;           // where the blocks sit in their window right now
  unsigned* xccProbe = nullptr;    // pinned host word the probe kernel writes
  unsigned* dFoldTab = nullptr; // tier C: inbox slot -> LDS cells, per member
  size_t xStride = 0;           // doubles per cluster in dXbuf
  unsigned long long timeoutTicks = 0, arriveTicks = 0;
  int clusterSpread = 0;        // members of a cluster dealt over the XCDs (option cluster_spread; default: when that fits a quarter more clusters)
  unsigned* syncCheck = nullptr;     // pinned host copies of the sync blocks of every launch of the last call (watchdog, placement census)
  size_t syncCheckWords = 0, syncCheckCap = 0;
  size_t syncLaunches = 0;
  unsigned clustersSeen = 0, clustersSplit = 0;   // last call: clusters that ran, clusters whose members sat on more than one XCD
  std::string tierNote;
  dnas::TierAPlan plan;
  hipModule_t module = nullptr;
  hipFunction_t fillA = nullptr;
  hipModule_t moduleSeg = nullptr;  // the same kernel built with -DDNAS_SEGMENTS=1 (bounded-memory decode), compiled at first use
  hipFunction_t fillSeg = nullptr;
  std::string jitDefs;
  TierAArgs argsA{};
  unsigned *dEntTab = nullptr, *dMetaTab = nullptr;
  int32_t* dSlotOf = nullptr;
  hipStream_t stream = nullptr;     // fill kernels
  hipStream_t stream2 = nullptr;    // traceback kernels (batch i traces back while batch i+1 fills)
  std::vector<hipEvent_t> sync;     // 2 per batch: fill done, traceback done
  DevModel dm{};
  std::vector<void*> owned;   // device allocations of the tables
  double* arena = nullptr;
  size_t arenaBytes = 0, arenaCap = 0;
  int maxSlots = 512;
  unsigned long long* dRounds = nullptr;
  // per-call scheduling arrays (device), grown on demand
  int32_t* dBatchRead = nullptr;
  uint64_t *dSlotOff = nullptr, *dReadOff = nullptr, *dOutOff = nullptr;
  size_t schedCap = 0;
  size_t halfDoubles = 0;              // size of one arena half (doubles)
  std::vector<uint64_t> lastSlotOff;   // host copy, sorted-batch order of the last call
  std::vector<int32_t> lastBatchRead;
  std::vector<int64_t> lastBatchStart; // first read (sorted order) of every batch of the last call, + n_reads
  std::vector<uint64_t> lastReadOff;   // host copy of the last call's read offsets
  const uint8_t* lastBases = nullptr;  // device pointer of the last call's bases (valid while the caller keeps it)
  // dnas_viterbi_batch's device copies of the caller's host arrays: owned by the model and only ever grown (hipMalloc / hipFree
  // per call is what another tenant of the card can hold up for seconds: tools/alloc_probe.py); the bases stay valid until the
  // next call (lattice export)
  uint8_t* ioBases = nullptr; char* ioSym = nullptr; uint32_t* ioLen = nullptr; double* ioLL = nullptr; uint8_t* ioSt = nullptr;
  size_t ioBasesCap = 0, ioSymCap = 0, ioReadsCap = 0;
  std::vector<hipEvent_t> events;      // 4 per batch: fill start/end (stream), traceback start/end (stream2)
  dnas_batch_stats stats{};
  bool statsPending = false;
  // optional traceback event log (the reference's level-3 messages): device buffers of the last call
  bool eventLog = false;
  unsigned long long* dEvents = nullptr;
  uint64_t* dEvOff = nullptr;
  uint32_t* dEvLen = nullptr;
  std::vector<uint64_t> evOff;
};

constexpr size_t kSyncWindow = 64 * 1024, kSyncStep = 1024;   // the sync blocks of a tier-C model sit somewhere in a window this much longer than they are

// Where in their window the sync blocks serve a launch best, given the measured round trips (m->syncLat) and the XCC the launch's
// block 0 -- cluster 0 -- is expected on: cluster c of a launch runs on the XCC c after it (whole clusters per XCD), the clusters of
// a spread plan on all of them.  The first cluster counts four times: a launch of ONE cluster is what latency is asked of.
static size_t place_sync_words(const dnas_model* m, unsigned firstXcc) {
  if (m->syncLat.empty()) return 0;
  const int first = std::min(8, m->maxClusters), nC = m->syncCand;
  double best = -1;
  size_t at = 0;
  for (size_t off = 0; off <= kSyncWindow; off += kSyncStep) {
    double cost = 0;
    for (int c = 0; c < first; ++c) {
      const size_t k = (off + (size_t)c * 256) / kSyncStep;
      if ((int)k >= nC) { cost = -1; break; }
      double t = 0;
      if (m->clusterSpread) { for (int x = 0; x < 8; ++x) t += m->syncLat[(size_t)x * nC + k] / 8; }
      else t = m->syncLat[(size_t)((firstXcc + (unsigned)c) & 7u) * nC + k];
      cost += (c == 0 ? 4.0 : 1.0) * t;
    }
    if (cost >= 0 && (best < 0 || cost < best)) { best = cost; at = off; }
  }
  return at;
}

namespace {

// roctx ranges around the phases of a call (visible in rocprofv3 --marker-trace): the marker library is looked up at run
// time, so that the shared library does not depend on it
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    for (const char* lib : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
      if (void* h = dlopen(lib, RTLD_LAZY | RTLD_LOCAL)) {
        push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
        pop = (int (*)())dlsym(h, "roctxRangePop");
        if (push && pop) return;
        push = nullptr; pop = nullptr;
      }
    }
  }
};
const Roctx& roctx() { static Roctx r; return r; }
struct RoctxRange {
  bool on;
  explicit RoctxRange(const char* name) : on(roctx().push != nullptr) { if (on) roctx().push(name); }
  ~RoctxRange() { if (on) roctx().pop(); }
};

template <class T>
int upload(dnas_model* m, const T* host, size_t n, const T** out) {
  T* d = nullptr;
  HIP_TRY(hipMalloc((void**)&d, std::max<size_t>(n, 1) * sizeof(T)));
  m->owned.push_back(d);
  if (n) HIP_TRY(hipMemcpy(d, host, n * sizeof(T), hipMemcpyHostToDevice));
  *out = d;
  return DNAS_OK;
}

int collect_stats(dnas_model* m) {
  if (!m->statsPending) return DNAS_OK;
  m->stats.fill_ms = m->stats.traceback_ms = 0;
  for (size_t i = 0; i + 4 <= m->events.size(); i += 4) {
    float a = 0, b = 0;
    HIP_TRY(hipEventElapsedTime(&a, m->events[i], m->events[i + 1]));
    HIP_TRY(hipEventElapsedTime(&b, m->events[i + 2], m->events[i + 3]));
    m->stats.fill_ms += a;
    m->stats.traceback_ms += b;
  }
  unsigned long long r = 0;
  HIP_TRY(hipMemcpy(&r, m->dRounds, sizeof r, hipMemcpyDeviceToHost));
  m->stats.rounds = (int64_t)r;
  m->statsPending = false;
  if (m->tier == 2) {
    unsigned xccMixed = 0, clusters = 0;
    for (size_t c = 0; c * 64 < m->syncCheckWords; ++c) {
      const unsigned* w = m->syncCheck + c * 64;
      if (w[1]) return dnas::fail(DNAS_E_DEVICE, "tier C: a cluster did not agree on a lattice column within the watchdog time, or its work-groups were "
                                                 "not all started within the arrival time (launch aborted; options cluster_timeout_s, cluster_arrive_s)");
      if (w[40]) { ++clusters; if (w[40] & (w[40] - 1)) ++xccMixed; }
      if (c == 0 && getenv("DNAS_SYNC_DEBUG")) fprintf(stderr, "sync debug: cluster 0 ran on XCC mask 0x%x, its epoch word counted %u bumps\n", w[40], w[0]);
    }
    m->clustersSeen = clusters; m->clustersSplit = xccMixed;
  }
  return DNAS_OK;
}

}  // namespace

namespace {

// Which row program serves a machine fastest?  The planner can deal the states breadth first or by longest-path level
// (host/plan.hpp: PlanChoice); which is best depends on the machine and the plan's cost model does not predict it, so it is
// measured and RECORDED: dnastore_amd/tune/ ships the verdicts of the fixture and bench machines (tools/make_tune_records.sh:
// bench-like reads), and a model follows its machine's record.  The record's name hashes the machine's graph (not the error
// model: the row programs do not depend on it), the work-group shape and the planner version; the kernel source it was
// measured with is named INSIDE the record ("kernel=<hash>"), so that an edit of the kernel leaves the records in force and
// tests/test_tune_records.py says which ones to measure again.
std::string tune_record_name(const dnas_flat_model* fm, int members, int threads) {
  std::string graph((const char*)&fm->n_states, sizeof fm->n_states);
  auto add = [&](const void* ptr, size_t bytes) { graph.append((const char*)ptr, bytes); };
  add(&fm->max_dup_len, sizeof fm->max_dup_len); add(&threads, sizeof threads); add(&members, sizeof members);
  add(&dnas::kPlanVersion, sizeof dnas::kPlanVersion);
  add(fm->ein_ptr, ((size_t)fm->n_states + 1) * sizeof(int32_t)); add(fm->ein_src, (size_t)fm->n_emit * sizeof(int32_t));
  add(fm->ein_score, (size_t)fm->n_emit * sizeof(double)); add(fm->ein_base, (size_t)fm->n_emit);
  add(fm->nin_ptr, ((size_t)fm->n_states + 1) * sizeof(int32_t)); add(fm->nin_src, (size_t)fm->n_null * sizeof(int32_t));
  add(fm->nin_score, (size_t)fm->n_null * sizeof(double));
  char name[64];
  snprintf(name, sizeof name, "tune_%016llx.txt", dnas::textHash(graph));
  return name;
}

// "order=<o> slack=<s> kernel=<hash> ..." -> the choice a record names (false: no record / not readable)
bool parse_plan_record(const std::string& note, dnas::PlanChoice* choice, std::string* kernel = nullptr) {
  int o = -1, sl = 0;
  char k[32] = "";
  if (sscanf(note.c_str(), "order=%d slack=%d kernel=%31s", &o, &sl, k) < 2 || o < 0 || o > 2 || sl < 0 || sl > 8) return false;
  choice->order = o;
  choice->slack = sl;
  if (kernel) *kernel = k;
  return true;
}

// the record of a machine, if there is one: the choice, and a note for dnas_model_tier ("record tune_x.txt" / "... stale: measured
// with another kernel source")
bool follow_plan_record(const dnas_flat_model* fm, int members, int threads, dnas::PlanChoice* choice, std::string* note) {
  const std::string name = tune_record_name(fm, members, threads);
  std::string kernel;
  if (!parse_plan_record(dnas::cacheNoteRead(name), choice, &kernel)) { *note = "no tuning record (default row program)"; return false; }
  char now[32];
  snprintf(now, sizeof now, "%016llx", dnas::kernelSourceHash());
  *note = "record " + name + " order=" + std::to_string(choice->order) + " slack=" + std::to_string(choice->slack) +
          (kernel == now ? "" : " (stale: measured with kernel " + kernel + ", this is " + now + ")");
  return true;
}

// The choice a model of this machine would be planned with when nothing is forced by an option: the environment (DNAS_PLAN_ORDER,
// DNAS_PLAN_SLACK: left to the planner), else the machine's tuning record unless DNAS_RECORDS=0, else the default.  The
// precompile and analysis entry points use it, so that what they compile or describe is the program a model runs.
dnas::PlanChoice env_or_recorded_choice(const dnas_flat_model* fm, int members, int threads) {
  dnas::PlanChoice choice;
  if (!getenv("DNAS_PLAN_ORDER") && !getenv("DNAS_PLAN_SLACK") && !(getenv("DNAS_RECORDS") && atoi(getenv("DNAS_RECORDS")) == 0))
    (void)parse_plan_record(dnas::cacheNoteRead(tune_record_name(fm, members, threads)), &choice);
  return choice;
}
int env_threads(int fallback) { return getenv("DNAS_THREADS") && atoi(getenv("DNAS_THREADS")) > 0 ? atoi(getenv("DNAS_THREADS")) : fallback; }

// autotune=1 and no record: time the candidate programs on synthetic reads once (one launch each), keep the verdict in the
// kernel cache.  The reads are what a random walk through the machine emits (a code word sequence) with one base in a hundred
// substituted; the lattice arena of the timing models is the caller's (at most 8 GiB).
int tune_row_program(const dnas_flat_model* fm, int device_id, int threads, size_t arena_bytes, dnas::PlanChoice* choice) {
  *choice = dnas::PlanChoice{1, 0};
  const std::string name = tune_record_name(fm, 1, threads);
  const size_t arena = std::min<size_t>(arena_bytes ? arena_bytes : (size_t)8 << 30, (size_t)8 << 30);
  const int L = 480;
  // a read's lattice: (L + 1) columns of 2 lanes of (about) n_states doubles; half the arena holds a launch
  const int nReads = (int)std::max<size_t>(4, std::min<size_t>(240, arena / 2 / ((size_t)(L + 1) * 2 * 8 * ((size_t)fm->n_states + 2048))));
  std::vector<uint8_t> bases((size_t)L * nReads);
  {
    // (destination, emitted base or -1); the walk stays inside one message: it does not take an edge that reads the
    // end-of-message symbol '$' while another one is there
    std::vector<std::vector<std::pair<int, int>>> out((size_t)fm->n_states), outEnd((size_t)fm->n_states);
    for (int j = 0; j < fm->n_states; ++j) {
      for (int e = fm->ein_ptr[j]; e < fm->ein_ptr[j + 1]; ++e) (fm->ein_in[e] == '$' ? outEnd : out)[(size_t)fm->ein_src[e]].emplace_back(j, (int)fm->ein_base[e]);
      for (int e = fm->nin_ptr[j]; e < fm->nin_ptr[j + 1]; ++e) (fm->nin_in[e] == '$' ? outEnd : out)[(size_t)fm->nin_src[e]].emplace_back(j, -1);
    }
    for (int j = 0; j < fm->n_states; ++j) if (out[(size_t)j].empty()) out[(size_t)j] = outEnd[(size_t)j];
    unsigned long long x = 88172645463325252ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (int r = 0; r < nReads; ++r) {
      int state = 0, got = 0;
      for (long steps = 0; got < L && steps < 64l * L; ++steps) {
        if (out[(size_t)state].empty()) { state = 0; continue; }
        const std::pair<int, int>& e = out[(size_t)state][rnd() % out[(size_t)state].size()];
        state = e.first;
        if (e.second >= 0) bases[(size_t)r * L + got++] = (uint8_t)((rnd() % 100 == 0) ? (e.second + 1 + rnd() % 3) & 3 : e.second);
      }
      for (; got < L; ++got) bases[(size_t)r * L + got] = (uint8_t)(rnd() & 3);       // (a machine that emits too little)
    }
  }
  std::vector<uint64_t> readOff(nReads + 1), outOff(nReads + 1);
  const size_t cap = 4 * (size_t)L + 64;
  for (int r = 0; r <= nReads; ++r) { readOff[(size_t)r] = (uint64_t)r * L; outOff[(size_t)r] = (uint64_t)r * cap; }
  std::vector<char> sym(cap * nReads);
  std::vector<uint32_t> len(nReads);
  std::vector<double> ll(nReads);
  std::vector<uint8_t> st(nReads);
  // the default first: another candidate has to beat it by 1.5 % (run-to-run differences of one program stay below 0.5 %)
  const dnas::PlanChoice candidates[] = {{1, 0}, {2, 0}, {2, 8}};
  double best = 0;
  std::string report;
  for (const dnas::PlanChoice& c : candidates) {
    dnas_model* t = nullptr;
    const std::string options = "tier=A,autotune=0,plan_order=" + std::to_string(c.order) + ",plan_slack=" + std::to_string(c.slack) + ",threads=" + std::to_string(threads);
    int rc = dnas_model_create_ex(fm, device_id, arena, options.c_str(), &t);
    if (rc != DNAS_OK) { if (c.order == 1) return DNAS_OK; continue; }   // (the creation that asked reports what is wrong)
    dnas_batch_stats s{};
    double fastest = 0;
    for (int rep = 0; rep < 3 && rc == DNAS_OK; ++rep) {          // the first run warms up; the faster of the other two counts
      rc = dnas_viterbi_batch(t, nReads, readOff.data(), bases.data(), sym.data(), outOff.data(), len.data(), ll.data(), st.data());
      if (rc == DNAS_OK) rc = dnas_model_last_stats(t, &s);
      if (rc == DNAS_OK && rep > 0 && (fastest == 0 || s.fill_ms < fastest)) fastest = s.fill_ms;
    }
    dnas_model_destroy(t);
    s.fill_ms = fastest;
    if (rc != DNAS_OK || !(s.fill_ms > 0)) continue;
    char item[96];
    snprintf(item, sizeof item, "  %d/%d: %.3f ms", c.order, c.slack, s.fill_ms);
    report += item;
    if (best == 0 || s.fill_ms < 0.985 * best) { if (best == 0 || s.fill_ms < best) best = s.fill_ms; *choice = c; }
  }
  if (best == 0) return DNAS_OK;
  char head[200];
  snprintf(head, sizeof head, "order=%d slack=%d kernel=%016llx   (fill of %d synthetic reads of %d bases;", choice->order, choice->slack,
           dnas::kernelSourceHash(), nReads, L);
  dnas::cacheNoteWrite(name, std::string(head) + report + ")\n");
  return DNAS_OK;
}

}  // namespace

extern "C" int dnas_has_device_code(void) { return 1; }

extern "C" int dnas_device_count(void) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess) return 0;
  return count;
}

extern "C" int dnas_model_create(const dnas_flat_model* fm, int device_id, size_t arena_bytes, dnas_model** out) {
  return dnas_model_create_ex(fm, device_id, arena_bytes, nullptr, out);
}

// options: "key=value,key=value"; keys tier (A|B|C), cluster (work-groups per read), threads (512 | 1024 per work-group),
// max_clusters, max_slots, cluster_timeout_s, traceback, arena_fraction, checkpoint, segment, plan_order (dealing order: 0 depth
// first, 1 breadth first, 2 longest-path levels), plan_slack, records (0: ignore the machine's tuning record), autotune (1: a
// tier-A machine without a record is timed once and the verdict kept in the kernel cache).  A key that is absent falls back to
// the environment variable DNAS_<KEY>.
extern "C" int dnas_model_create_ex(const dnas_flat_model* fm, int device_id, size_t arena_bytes, const char* options,
                                    dnas_model** out) {
  if (!fm || !out) return dnas::fail(DNAS_E_INVALID, "dnas_model_create: null argument");
  *out = nullptr;
  const std::string optStr = options ? options : "";
  std::string optHold;
  auto opt = [&](const char* key) -> const char* {
    const std::string k = std::string(key) + "=";
    size_t at = 0;
    while (at < optStr.size()) {
      size_t end = optStr.find(',', at);
      if (end == std::string::npos) end = optStr.size();
      if (optStr.compare(at, k.size(), k) == 0) { optHold = optStr.substr(at + k.size(), end - at - k.size()); return optHold.c_str(); }
      at = end + 1;
    }
    std::string env = "DNAS_" + std::string(key);
    for (char& c : env) c = (char)toupper((unsigned char)c);
    return getenv(env.c_str());
  };
  if (fm->n_len > kMaxLen) return dnas::fail(DNAS_E_UNSUPPORTED, "pLen longer than 32 entries");
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return dnas::fail(DNAS_E_DEVICE, "no HIP device available");
  if (device_id < 0 || device_id >= count) return dnas::fail(DNAS_E_INVALID, "device_id out of range");
  HIP_TRY(hipSetDevice(device_id));
  dnas_model* m = new dnas_model();
  m->device = device_id;
  auto bail = [&](int rc) { dnas_model_destroy(m); return rc; };
  if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&m->stream2, hipStreamNonBlocking) != hipSuccess)
    return bail(dnas::fail(DNAS_E_DEVICE, "hipStreamCreate failed"));
  const int N = fm->n_states, D = fm->max_dup_len;
  DevModel& d = m->dm;
  d.N = N;
  d.Npad = (N + 31) & ~31;  // 256-byte aligned rows
  d.D = D;
  d.P = fm->n_len;
  d.local = fm->local;
  d.storedLanes = D + 2;
  int rc;
#define UP(field, src, n) if ((rc = upload(m, src, (size_t)(n), &d.field)) != DNAS_OK) return bail(rc)
  UP(einPtr, fm->ein_ptr, N + 1); UP(einSrc, fm->ein_src, fm->n_emit); UP(einScore, fm->ein_score, fm->n_emit);
  UP(einBase, fm->ein_base, fm->n_emit); UP(einIn, fm->ein_in, fm->n_emit);
  UP(ninPtr, fm->nin_ptr, N + 1); UP(ninSrc, fm->nin_src, fm->n_null); UP(ninScore, fm->nin_score, fm->n_null);
  UP(ninIn, fm->nin_in, fm->n_null);
  UP(eoutPtr, fm->eout_ptr, N + 1); UP(eoutDst, fm->eout_dst, fm->n_emit);
  UP(noutPtr, fm->nout_ptr, N + 1); UP(noutDst, fm->nout_dst, fm->n_null);
  UP(mdl, fm->mdl, N); UP(ctx, fm->ctx, (size_t)N * (D ? D : 1));
#undef UP
  d.noGap = fm->no_gap; d.delOpen = fm->del_open; d.delExtend = fm->del_extend; d.delEnd = fm->del_end;
  d.tanDup = fm->tan_dup;
  memcpy(d.sub, fm->sub, sizeof d.sub);
  for (int k = 0; k < kMaxLen; ++k) d.len[k] = k < fm->n_len ? fm->len[k] : 0.;
  d.slotOf = nullptr;
  if (hipMalloc((void**)&m->dRounds, 16 * sizeof(unsigned long long)) != hipSuccess)
    return bail(dnas::fail(DNAS_E_DEVICE, "hipMalloc failed"));
  // ---- tier A: specialise the register/LDS-resident kernel for this machine
  auto uploadEdgeSlots = [&](const int32_t* slotOf) -> int {   // after the tier (and with it the slot map) is known
    std::vector<int32_t> es((size_t)std::max(fm->n_emit, 1)), ns((size_t)std::max(fm->n_null, 1));
    for (int e = 0; e < fm->n_emit; ++e) es[(size_t)e] = slotOf ? slotOf[fm->ein_src[e]] : fm->ein_src[e];
    for (int e = 0; e < fm->n_null; ++e) ns[(size_t)e] = slotOf ? slotOf[fm->nin_src[e]] : fm->nin_src[e];
    int r = upload(m, es.data(), es.size(), &d.einSlot);
    if (r == DNAS_OK) r = upload(m, ns.data(), ns.size(), &d.ninSlot);
    if (r != DNAS_OK) return r;
    // node records (device_model.h): only when the machine's edge scores are few enough to be named by a byte
    d.rec = nullptr; d.recScore = nullptr;
    if (D > 4 || getenv("DNAS_NO_NODE_RECORDS")) return DNAS_OK;
    std::vector<double> scores;
    auto scoreIndex = [&](double v) -> int {
      for (size_t i = 0; i < scores.size(); ++i) if (memcmp(&scores[i], &v, sizeof v) == 0) return (int)i;
      if (scores.size() >= 256) return -1;
      scores.push_back(v);
      return (int)scores.size() - 1;
    };
    std::vector<uint32_t> rec((size_t)N * 16, 0u);
    for (int j = 0; j < N; ++j) {
      uint32_t* w = &rec[(size_t)j * 16];
      const int e0 = fm->ein_ptr[j], ne = fm->ein_ptr[j + 1] - e0, n0 = fm->nin_ptr[j], nn = fm->nin_ptr[j + 1] - n0;
      uint32_t ctxBits = 0;
      for (int q = 0; q < D; ++q) ctxBits |= (uint32_t)(fm->ctx[(size_t)j * D + q] & 3u) << (2 * q);
      w[0] = (uint32_t)(ne > kRecEmit ? 15 : ne) | (uint32_t)(nn > kRecNull ? 15 : nn) << 4 | (uint32_t)fm->mdl[j] << 8 | ctxBits << 12;
      w[1] = (uint32_t)(slotOf ? slotOf[j] : j);
      for (int i = 0; i < ne && ne <= kRecEmit; ++i) {
        const int k = scoreIndex(fm->ein_score[e0 + i]);
        if (k < 0) return DNAS_OK;
        w[2 + 3 * i] = (uint32_t)fm->ein_src[e0 + i];
        w[3 + 3 * i] = (uint32_t)es[(size_t)(e0 + i)];
        w[4 + 3 * i] = (uint32_t)fm->ein_in[e0 + i] | (uint32_t)fm->ein_base[e0 + i] << 8 | (uint32_t)k << 16;
      }
      for (int i = 0; i < nn && nn <= kRecNull; ++i) {
        const int k = scoreIndex(fm->nin_score[n0 + i]);
        if (k < 0) return DNAS_OK;
        w[11 + 3 * i] = (uint32_t)fm->nin_src[n0 + i];
        w[12 + 3 * i] = (uint32_t)ns[(size_t)(n0 + i)];
        w[13 + 3 * i] = (uint32_t)fm->nin_in[n0 + i] | (uint32_t)k << 16;
      }
    }
    if (scores.empty()) scores.push_back(0.);
    r = upload(m, scores.data(), scores.size(), &d.recScore);
    if (r == DNAS_OK) r = upload(m, rec.data(), rec.size(), &d.rec);
    if (r != DNAS_OK) { d.rec = nullptr; d.recScore = nullptr; }
    return r;
  };
  {
    // DNAS_TIER=A|B|C forces a tier (and makes its failure an error); DNAS_CLUSTER=<G> forces the cluster size
    const char* forceOpt = opt("tier");
    const std::string force = forceOpt ? forceOpt : "";
    const char want = !force.empty() ? (char)(force[0] & ~0x20) : 0;
    int wantG = 0, wantT = 0;
    dnas::PlanChoice want_;               // -1: not said
    bool autotune = false, records = true;
    std::string recordNote = "row program forced by option";
    if (const char* s = opt("cluster")) wantG = atoi(s);
    if (const char* s = opt("threads")) wantT = atoi(s);
    if (const char* s = opt("plan_order")) want_.order = std::max(0, std::min(2, atoi(s)));
    if (const char* s = opt("plan_slack")) want_.slack = std::max(0, std::min(8, atoi(s)));
    if (const char* s = opt("autotune")) autotune = atoi(s) != 0;
    if (const char* s = opt("records")) records = atoi(s) != 0;
    const bool free_ = want_.order < 0 && want_.slack < 0;     // nothing forced: the record decides
    if (want == 'B') {
      m->tierNote = "tier B forced by DNAS_TIER";
    } else {
      std::string whyNotA;
      if (want != 'C' && wantG < 2) {
        const int threadsA = wantT ? wantT : dnas::kTierAThreads;
        dnas::PlanChoice choiceA = want_;
        if (free_ && !records) recordNote = "tuning records off (default row program)";
        if (free_ && records && !follow_plan_record(fm, 1, threadsA, &choiceA, &recordNote) && autotune) {
          // no record for this machine: time the candidates once (if the machine fits one work-group at all)
          if (dnas::buildTierAPlan(*fm, threadsA, choiceA).ok) {
            int rcTune = tune_row_program(fm, device_id, threadsA, arena_bytes, &choiceA);
            if (rcTune != DNAS_OK) return bail(rcTune);
            if (hipSetDevice(device_id) != hipSuccess) return bail(dnas::fail(DNAS_E_DEVICE, "hipSetDevice failed"));
            recordNote = "timed at model creation: order=" + std::to_string(choiceA.order) + " slack=" + std::to_string(choiceA.slack);
          }
        }
        m->plan = dnas::buildTierAPlan(*fm, threadsA, choiceA);
        if (!m->plan.ok) whyNotA = m->plan.whyNot;
      } else {
        m->plan.ok = false;
        whyNotA = "cluster forced";
      }
      if (!m->plan.ok && want != 'A') {
        // (clusters are not timed here -- planning a 258 538-state machine takes seconds per candidate -- but a recorded
        //  verdict is followed: tools/make_tune_records.py times the dealing orders for the bench machines)
        dnas::PlanChoice choiceC = want_;
        if (free_ && !records) recordNote = "tuning records off (default row program)";
        if (free_ && records) (void)follow_plan_record(fm, wantG >= 2 ? wantG : 0, wantT, &choiceC, &recordNote);
        m->plan = dnas::chooseClusterPlan(*fm, wantG, wantT, choiceC);
        if (!m->plan.ok) m->plan.whyNot = "one work-group: " + whyNotA + "; cluster: " + m->plan.whyNot;
      }
      if (!m->plan.ok) {
        m->tierNote = "tier B: " + m->plan.whyNot;
        if (want) return bail(dnas::fail(DNAS_E_UNSUPPORTED, "tier " + force + " was asked for: " + m->plan.whyNot));
      } else {
        try {
          std::string defs = m->plan.defines;
          if (const char* extra = getenv("DNAS_TIERA_DEFS")) {      // diagnostics, e.g. -DDNAS_STAMP; several: separated by ':', ' ' or newlines
            std::string more = extra;
            for (char& c : more) if (c == ':' || c == ' ') c = '\n';
            defs += "\n" + more;
          }
          m->jitDefs = defs;
          const std::vector<char> code = dnas::jitCompile(defs, m->plan.key);
          if (hipModuleLoadData(&m->module, code.data()) != hipSuccess ||
              hipModuleGetFunction(&m->fillA, m->module, "viterbi_fill_tiera") != hipSuccess)
            throw std::runtime_error("hipModuleLoadData/GetFunction failed");
          const dnas::TierAPlan& p = m->plan;
          if (hipMalloc((void**)&m->dEntTab, p.entTab.size() * 4) != hipSuccess ||
              hipMalloc((void**)&m->dMetaTab, p.metaTab.size() * 4) != hipSuccess ||
              hipMalloc((void**)&m->dSlotOf, p.slotOf.size() * 4) != hipSuccess ||
              hipMemcpy(m->dEntTab, p.entTab.data(), p.entTab.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
              hipMemcpy(m->dMetaTab, p.metaTab.data(), p.metaTab.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
              hipMemcpy(m->dSlotOf, p.slotOf.data(), p.slotOf.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
            throw std::runtime_error("tier A table upload failed");
          TierAArgs& a = m->argsA;
          a.N = N; a.local = fm->local;
          a.noGap = fm->no_gap; a.delOpen = fm->del_open; a.delExtend = fm->del_extend; a.delEnd = fm->del_end;
          a.tanDup = fm->tan_dup;
          memcpy(a.sub, fm->sub, sizeof a.sub);
          for (int k = 0; k < 8; ++k) a.len[k] = k < fm->n_len ? fm->len[k] : 0.;
          memcpy(a.score, p.score, sizeof a.score);
          d.slotOf = m->dSlotOf;
          d.Npad = p.NS;          // lattice row stride = slots
          d.storedLanes = 2;      // tier A keeps S and D in HBM; T lanes are recomputed where needed
          if (p.G == 1) {
            m->tier = 1;
            m->tierNote = "tier A: " + p.key + "; " + recordNote;
          } else {
            // a cluster lives on one XCD (32 CUs): floor(32 / G) clusters per XCD.  Every cluster owns an exchange
            // buffer (3 arrays of G * GROWS * T cells + the end-of-read reduction cells) and a sync block.
            // The members of a cluster wait for each other, so all of them must be resident together: the grid is sized
            // from what the device says it can hold of THIS kernel (one work-group per CU is what the plan aims at; were it
            // none, the launch could never finish).  Work-groups are dispatched in index order and a cluster's members are
            // neighbours in it (groups of 8 clusters), so whatever else holds CUs -- another process, this model's traceback --
            // delays at most the last clusters of a launch until CUs come free; the kernel gives late members time (below).
            int cus = 256, perCu = 0;
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id);
            if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, m->fillA, p.T, p.ldsBytes) != hipSuccess || perCu < 1)
              throw std::runtime_error("the cluster kernel cannot be resident on this device (occupancy " + std::to_string(perCu) + ")");
            const int xcds = std::max(1, cus / 32);
            m->maxClusters = std::max(1, xcds * ((cus / xcds) / p.G));      // one work-group per CU, whole clusters per XCD
            // Whole clusters per XCD leave CUs idle when a cluster is large (21 members: one cluster and 11 idle CUs per XCD).
            // Dealt over the XCDs instead (block b = member b % G of cluster b / G) half as many clusters again fit, and the
            // exchange through memory (write-through stores, the loads find them there) costs 2 %: measured 12 clusters of
            // 21 at 152 ms per read against 8 at 144 -- taken when it buys a quarter more clusters.
            if (const char* s = opt("cluster_spread")) m->clusterSpread = atoi(s) != 0;
            else if ((cus / p.G) * 4 >= m->maxClusters * 5) m->clusterSpread = 1;
            if (m->clusterSpread) m->maxClusters = std::max(1, cus / p.G);
            if (const char* s = opt("max_clusters")) m->maxClusters = std::max(1, atoi(s));
            m->xStride = (size_t)p.exchangeStride();
            // the sync blocks sit in a window of 64 KB more than they need: their place in it is chosen below
            size_t syncOff = kSyncWindow;
            if (hipMalloc((void**)&m->dXbuf, m->xStride * (size_t)m->maxClusters * sizeof(double)) != hipSuccess ||
                hipMalloc((void**)&m->dSyncBase, (size_t)m->maxClusters * 64 * sizeof(unsigned) + syncOff) != hipSuccess ||
                hipMalloc((void**)&m->dFoldTab, std::max<size_t>(p.foldTab.size(), 1) * 4) != hipSuccess ||
                hipMemcpy(m->dFoldTab, p.foldTab.data(), p.foldTab.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
              throw std::runtime_error("tier C exchange buffer allocation failed");
            // WHERE in the window: the members of a cluster agree on every column through device-scope atomics on their sync block,
            // performed at the memory side -- the round trip from an XCD depends on the memory channel the address belongs to (one
            // read alone on 16 work-groups: 41.6 ms or 54 ms with the block 4 KB apart; a full launch of 64 clusters does not care,
            // its blocks spread over the channels).  sync_latency_kernel times the round trip from every XCD to a candidate
            // every KB; the offset is taken that serves the first clusters of a launch best (cluster c runs on the XCD of block
            // c % 8; the clusters of a spread plan on all of them), the first one -- a single read's -- above all.
            // DNAS_SYNC_OFFSET=<bytes> forces a place (experiments), sync_place=0 takes the start of the window.
            syncOff = 0;
            if (const char* e = getenv("DNAS_SYNC_OFFSET")) syncOff = std::min((size_t)atol(e) & ~(size_t)255, kSyncWindow);
            else if (!(opt("sync_place") && atoi(opt("sync_place")) == 0)) {
              const int nCand = (int)(kSyncWindow / kSyncStep) + 16, reps = 12;    // (the blocks of the first clusters reach past the offset)
              unsigned long long* dLat = nullptr;
              unsigned* dXcc = nullptr;
              std::vector<unsigned long long> lat((size_t)8 * nCand, 0ull);
              if (hipMalloc((void**)&dLat, lat.size() * sizeof(unsigned long long)) == hipSuccess && hipMalloc((void**)&dXcc, 8 * sizeof(unsigned)) == hipSuccess &&
                  hipMemsetAsync(m->dSyncBase, 0, (size_t)m->maxClusters * 64 * sizeof(unsigned) + kSyncWindow, m->stream) == hipSuccess) {
                const int candAvail = (int)std::min<size_t>((size_t)nCand, ((size_t)m->maxClusters * 64 * sizeof(unsigned) + kSyncWindow) / kSyncStep);
                hipLaunchKernelGGL(sync_latency_kernel, dim3(8), dim3(64), 0, m->stream, m->dSyncBase, candAvail, (int)(kSyncStep / sizeof(unsigned)), reps, dLat, dXcc);
                if (hipGetLastError() == hipSuccess &&
                    hipMemcpyAsync(lat.data(), dLat, (size_t)8 * candAvail * sizeof(unsigned long long), hipMemcpyDeviceToHost, m->stream) == hipSuccess &&
                    hipStreamSynchronize(m->stream) == hipSuccess) {
                  std::vector<unsigned> xcc(8, 0);
                  (void)hipMemcpy(xcc.data(), dXcc, 8 * sizeof(unsigned), hipMemcpyDeviceToHost);
                  m->syncCand = candAvail;
                  m->syncLat.assign((size_t)8 * candAvail, 0.);
                  for (int b = 0; b < 8; ++b)
                    for (int k = 0; k < candAvail; ++k) m->syncLat[(size_t)(xcc[(size_t)b] & 7u) * candAvail + k] = (double)lat[(size_t)b * candAvail + k];
                  syncOff = place_sync_words(m, 0);
                  m->syncPlaceNote = "sync words at +" + std::to_string(syncOff) + " B";
                  if (getenv("DNAS_SYNC_DEBUG")) {
                    fprintf(stderr, "sync debug: calibration blocks ran on XCCs");
                    for (int b = 0; b < 8; ++b) fprintf(stderr, " %u", xcc[(size_t)b]);
                    fprintf(stderr, "; offset %zu; block 0 ticks per candidate:", syncOff);
                    for (int k = 0; k < candAvail; k += 4) fprintf(stderr, " %llu", lat[(size_t)k]);
                    fprintf(stderr, "\n");
                  }
                }
              }
              if (dLat) (void)hipFree(dLat);
              if (dXcc) (void)hipFree(dXcc);
            }
            m->syncOffNow = syncOff;
            m->dSync = m->dSyncBase + syncOff / sizeof(unsigned);
            // watchdog per lattice column.  A column takes tens of microseconds, but the clock keeps running while the device's
            // scheduler lets another queue's kernel run: the limit only has to turn a protocol failure into an error instead
            // of a hung GPU.  (Two cluster launches at once on one card take turns launch by launch; the stalls of 2-6 s once
            // seen beside them were hipMalloc / hipFree of the arenas: tools/alloc_probe.py)
            double seconds = 30.0;
            if (const char* s = opt("cluster_timeout_s")) seconds = std::max(0.001, atof(s));
            m->timeoutTicks = (unsigned long long)(seconds * 1e8);   // s_memrealtime counts at 100 MHz
            double arrive = 120.0;  // ... and for the work-groups of a cluster to have all been started (CUs held by others)
            if (const char* s = opt("cluster_arrive_s")) arrive = std::max(0.001, atof(s));
            m->arriveTicks = (unsigned long long)(arrive * 1e8);
            m->tier = 2;
            m->tierNote = "tier C: " + std::to_string(p.G) + " work-groups per read, " + std::to_string(m->maxClusters) +
                          (m->clusterSpread ? " clusters dealt over the XCDs, exchange edges " : " clusters, exchange edges ") + std::to_string(p.crossEdges) + ", " + p.key + "; " + recordNote +
                          (m->syncPlaceNote.empty() ? "" : "; " + m->syncPlaceNote);
          }
        } catch (const std::exception& e) {
          m->tier = 0;
          m->tierNote = std::string("tier B: tier A/C unavailable: ") + e.what();
          // no silent fallback: a machine the register/LDS kernel can serve is served by it or not at all,
          // unless the caller asked for the fallback explicitly (DNAS_ALLOW_TIER_B_FALLBACK=1)
          if (!getenv("DNAS_ALLOW_TIER_B_FALLBACK")) return bail(dnas::fail(DNAS_E_DEVICE, m->tierNote));
        }
      }
    }
  }
  if ((rc = uploadEdgeSlots(m->tier >= 1 ? m->plan.slotOf.data() : nullptr)) != DNAS_OK) return bail(rc);
  size_t freeB = 0, totalB = 0;
  if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) return bail(dnas::fail(DNAS_E_DEVICE, "hipMemGetInfo failed"));
  m->arenaCap = arena_bytes ? arena_bytes : (size_t)((double)freeB * 0.6);
  if (m->tier == 1) {
    // A tier-A work-group owns a whole CU (all of its vector registers and nearly all of its LDS), so
    // a launch runs in rounds of one read per CU -- per XCD: work-groups are dealt round-robin to the
    // XCDs (32 CUs each), and an XCD that gets one work-group more than rounds x its free CUs adds a
    // whole round to the launch.  The traceback of the previous batch runs beside the fill (blocks of
    // 512 reads, one CU each), and a round with a few CUs of an XCD idle is faster than a full one
    // (the XCD's 4 MB L2 holds the S history of ~17 work-groups).  Measured on MI355X with the bench
    // workload: three rounds of 30 work-groups per XCD (720 reads) per launch, 48.8 ms, is the best
    // point; 2 x 255 = 510 reads cost 48 ms (a third round on the traceback's XCD), 500 reads 37 ms.
    // (Round 2 measured persistent work-groups that pull reads from a queue: a round takes 16.0-16.4 ms for any number
    // from 208 to 240 and 18 ms at 248, with or without the queue -- 3 x 240 stays the best point; the queue is gone.)
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 1) {
      const int xcds = std::max(1, cus / 32);
      m->maxSlots = std::max(1, 3 * (cus - 2 * xcds));
      // A small row program is compiled so that two work-groups share a CU (plan.cpp, wavesPerSimd): what the device says it holds
      // of this kernel decides, and a launch is then ONE round of two reads per CU (measured on water64.1*l4c4, ~1050-nt reads:
      // 480 reads per launch 0.565 of the roofline, 448: 0.53, 512: 0.51; one work-group per CU, 695 per launch: 0.41)
      int perCu = 1;
      if (m->plan.wavesPerSimd && hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, m->fillA, m->plan.T, m->plan.ldsBytes) == hipSuccess && perCu >= 2) {
        m->maxSlots = std::max(1, 2 * (cus - 2 * xcds));
        m->tierNote += "; 2 work-groups per CU";
      }
    }
  }
  if (m->tier == 2) m->maxSlots = 1 << 20;   // persistent clusters walk any number of reads: a launch is bounded by the arena only
  if (const char* s = opt("max_slots")) m->maxSlots = std::max(1, atoi(s));
  if (const char* s = opt("traceback")) m->waveTraceback = !(s[0] == 't' || s[0] == 'T');
  if (const char* s = opt("tb_threads")) m->tbThreads = std::max(64, std::min(256, atoi(s) / 64 * 64));
  if (const char* s = opt("tb_lanes")) m->tbLanes = std::max(1, std::min(64, atoi(s)));
  if (const char* s = opt("checkpoint")) m->checkpointMode = (s[0] == 'a' && s[1] == 'l') ? 1 : (s[0] == 'n' ? 2 : 0);   // auto | always | never
  if (const char* s = opt("segment")) m->segmentCols = std::max(0, atoi(s));
  if (const char* s = opt("arena_fraction")) {
    const double f = atof(s);
    if (f > 0.05 && f < 0.95 && !arena_bytes) m->arenaCap = (size_t)((double)freeB * f);
  }
  *out = m;
  return DNAS_OK;
}

extern "C" void dnas_model_destroy(dnas_model* m) {
  if (!m) return;
  (void)hipSetDevice(m->device);
  if (m->stream) (void)hipStreamSynchronize(m->stream);
  if (m->stream2) (void)hipStreamSynchronize(m->stream2);
  for (hipEvent_t e : m->sync) (void)hipEventDestroy(e);
  if (m->stream2) (void)hipStreamDestroy(m->stream2);
  for (void* p : m->owned) (void)hipFree(p);
  if (m->arena) (void)hipFree(m->arena);
  if (m->dRounds) (void)hipFree(m->dRounds);
  if (m->ioBases) (void)hipFree(m->ioBases);
  if (m->ioSym) (void)hipFree(m->ioSym);
  if (m->ioLen) (void)hipFree(m->ioLen);
  if (m->ioLL) (void)hipFree(m->ioLL);
  if (m->ioSt) (void)hipFree(m->ioSt);
  if (m->dEntTab) (void)hipFree(m->dEntTab);
  if (m->dMetaTab) (void)hipFree(m->dMetaTab);
  if (m->dSlotOf) (void)hipFree(m->dSlotOf);
  if (m->dXbuf) (void)hipFree(m->dXbuf);
  if (m->dSyncBase) (void)hipFree(m->dSyncBase);
  if (m->dFoldTab) (void)hipFree(m->dFoldTab);
  if (m->syncCheck) (void)hipHostFree(m->syncCheck);
  if (m->xccProbe) (void)hipHostFree(m->xccProbe);
  if (m->dEvents) (void)hipFree(m->dEvents);
  if (m->dEvOff) (void)hipFree(m->dEvOff);
  if (m->dEvLen) (void)hipFree(m->dEvLen);
  if (m->dColRange) (void)hipFree(m->dColRange);
  if (m->dSegSlot) (void)hipFree(m->dSegSlot);
  if (m->dWalks) (void)hipFree(m->dWalks);
  if (m->module) (void)hipModuleUnload(m->module);
  if (m->moduleSeg) (void)hipModuleUnload(m->moduleSeg);
  if (m->dBatchRead) (void)hipFree(m->dBatchRead);
  if (m->dSlotOff) (void)hipFree(m->dSlotOff);
  if (m->dReadOff) (void)hipFree(m->dReadOff);
  if (m->dOutOff) (void)hipFree(m->dOutOff);
  for (hipEvent_t e : m->events) (void)hipEventDestroy(e);
  if (m->stream) (void)hipStreamDestroy(m->stream);
  delete m;
}

extern "C" int dnas_model_sync(dnas_model* m) {
  if (!m) return dnas::fail(DNAS_E_INVALID, "null model");
  HIP_TRY(hipSetDevice(m->device));
  HIP_TRY(hipStreamSynchronize(m->stream));
  HIP_TRY(hipStreamSynchronize(m->stream2));
  return collect_stats(m);
}

extern "C" int dnas_model_last_stats(const dnas_model* m, dnas_batch_stats* out) {
  if (!m || !out) return dnas::fail(DNAS_E_INVALID, "null argument");
  if (m->statsPending) return dnas::fail(DNAS_E_INVALID, "call dnas_model_sync first");
  *out = m->stats;
  return DNAS_OK;
}

namespace {

// ---- one call of dnas_viterbi_batch_device, planned on the host -----------------------------------------------------------
// Bounded-memory decode (the reference holds every read's whole lattice, viterbi.h:48-50): the reads whose lattice does not
// fit half the arena -- the longest, first in sorted order -- are decoded in groups, in segments of C columns:
//   pass 1  fill segment after segment into a work buffer of H + C + 1 columns per read, keeping of every segment only its
//           last H columns and the hand-over lane (what the fill of the next segment and a traceback step look back at)
//   pass 2  from the last segment to the first: restore the checkpoint in front of the segment, fill it again (the last one
//           is still there), and let the traceback walk it; a walk that leaves the segment is parked until the next launch
// Fill work doubles; memory per read drops from L + 1 columns to about 2 sqrt((L + 1)(H + 1)).
struct SegmentGroup {
  int64_t first, n;          // reads [first, first + n) of the sorted order
  int64_t C, nSeg;           // columns per segment, segments of the longest read
  size_t workStride, ckStride;   // doubles per read: work buffer (H + C + 1 columns + a spare cell), checkpoint store
  size_t tabAt;              // where the group's per-segment tables start (entries of n reads per segment)
};
struct CallPlan {
  std::vector<int32_t> order;        // reads, longest first
  std::vector<uint64_t> slotOff;     // lattice of read i of the sorted order inside its arena half
  std::vector<int64_t> batchStart;   // whole-lattice batches: first read of each, + n_reads
  std::vector<SegmentGroup> groups;  // reads [0, nSegmented) in groups
  int64_t nSegmented = 0, columns = 0;
  size_t peak = 0;                   // doubles of the largest batch
  size_t groupPeak = 0, tabEntries = 0, groupLaunches = 0;
};

int plan_call(const dnas_model* m, int64_t n_reads, const uint64_t* read_offsets, CallPlan* cp) {
  const DevModel& d = m->dm;
  const size_t colDoubles = (size_t)d.storedLanes * (size_t)d.Npad;
  // longest reads first: a batch's work-groups then finish together
  cp->order.resize((size_t)n_reads);
  std::iota(cp->order.begin(), cp->order.end(), 0);
  std::stable_sort(cp->order.begin(), cp->order.end(), [&](int32_t a, int32_t b) {
    return read_offsets[a + 1] - read_offsets[a] > read_offsets[b + 1] - read_offsets[b];
  });
  auto lenOf = [&](int64_t i) { return (int64_t)(read_offsets[cp->order[(size_t)i] + 1] - read_offsets[cp->order[(size_t)i]]); };
  for (int64_t i = 0; i < n_reads; ++i)
    if (lenOf(i) > 0x7ffffff0ll) return dnas::fail(DNAS_E_UNSUPPORTED, "read too long");
  // two arena halves: batch i fills half (i & 1) while the traceback of batch i-1 still reads the other
  const size_t arenaCapDoubles = m->arenaCap / sizeof(double) / 2;

  // ---- the reads that go through segments, and their groups
  const size_t H = (size_t)d.D + 1;
  int64_t nSeg = 0;
  if (m->checkpointMode == 1) nSeg = n_reads;
  else if (m->checkpointMode == 0)
    while (nSeg < n_reads && colDoubles * (size_t)(lenOf(nSeg) + 1) + 8 > arenaCapDoubles) ++nSeg;
  cp->nSegmented = nSeg;
  const size_t budget = m->arenaCap / sizeof(double);
  const int64_t groupMax = m->tier == 2 ? m->maxClusters : m->maxSlots;
  for (int64_t g0 = 0; g0 < nSeg;) {
    const int64_t Lmax = lenOf(g0);
    int64_t nG = std::min(nSeg - g0, groupMax), C = 0;
    auto perRead = [&](int64_t c) {
      return (H + (size_t)c + 1) * colDoubles + 8 + (size_t)(Lmax / c + 1) * (H + 1) * colDoubles;
    };
    const int64_t cMin = (int64_t)H + 1;
    for (;;) {
      const size_t per = budget / (size_t)nG;
      if (m->segmentCols > 0) {
        C = std::max<int64_t>(cMin, m->segmentCols);
        if (perRead(C) <= per) break;
      } else {
        C = std::max<int64_t>(cMin, (int64_t)std::ceil(std::sqrt((double)(Lmax + 1) * (double)(H + 1))));
        if (perRead(C) <= per) {
          while (C < Lmax + 1 && perRead(std::min(2 * C, Lmax + 1)) <= per) C = std::min(2 * C, Lmax + 1);
          break;
        }
      }
      if (nG == 1)
        return dnas::fail(DNAS_E_NOMEM, "a read of " + std::to_string(Lmax) + " bases needs " + std::to_string(perRead(C) * 8) +
                                            " bytes of lattice segments and checkpoints; the lattice arena has " + std::to_string(m->arenaCap));
      nG = (nG + 1) / 2;
    }
    SegmentGroup g{g0, nG, C, Lmax / C + 1, (H + (size_t)C + 1) * colDoubles + 8, (size_t)(Lmax / C + 1) * (H + 1) * colDoubles, cp->tabEntries};
    cp->groupPeak = std::max(cp->groupPeak, (size_t)nG * (g.workStride + g.ckStride));
    cp->tabEntries += (size_t)g.nSeg * (size_t)nG;
    cp->groupLaunches += 2 * (size_t)g.nSeg;
    cp->groups.push_back(g);
    g0 += nG;
  }

  // ---- the other reads, in whole-lattice batches
  cp->slotOff.assign((size_t)n_reads, 0);
  cp->batchStart.assign(1, nSeg);
  size_t used = 0;
  for (int64_t i = 0; i < nSeg; ++i) cp->columns += lenOf(i) + 1;
  // one work-group per read: equal batches rather than full ones and a remainder (a launch costs whole rounds of
  // work-groups)
  const int64_t nPlain = n_reads - nSeg;
  const int64_t nFull = std::max<int64_t>(1, (nPlain + m->maxSlots - 1) / m->maxSlots);
  int64_t perBatch = (nPlain + nFull - 1) / nFull;
  // clusters: a launch runs whole rounds of maxClusters reads -- a batch that the arena cuts at 14 reads on 12 clusters would
  // take two rounds for 14; it is cut at 12, and the other two open the next batch
  const int64_t round = m->tier == 2 ? std::max(1, m->maxClusters) : 1;
  if (round > 1 && perBatch > round) perBatch = (perBatch + round - 1) / round * round;
  auto needOf = [&](int64_t i) { return colDoubles * (size_t)((uint64_t)lenOf(i) + 1) + 8; };   // + a spare cell (tier A, local mode: S(N-1, L) before its overwrite)
  for (int64_t i = nSeg; i < n_reads; ++i) {
    const uint64_t L = (uint64_t)lenOf(i);
    const size_t need = needOf(i);
    if (need > arenaCapDoubles)
      return dnas::fail(DNAS_E_NOMEM, "a single read's lattice (" + std::to_string(need * 8) +
                                          " bytes) exceeds the lattice arena (" + std::to_string(m->arenaCap) + ") and checkpoint=never");
    if (i - cp->batchStart.back() >= perBatch || used + need > arenaCapDoubles) {
      const int64_t start = cp->batchStart.back(), count = i - start;
      const int64_t cut = (round > 1 && count > round && count % round) ? start + count / round * round : i;
      cp->batchStart.push_back(cut);
      used = 0;
      for (int64_t t = cut; t < i; ++t) { cp->slotOff[(size_t)t] = used; used += needOf(t); }   // the reads that moved on
    }
    cp->slotOff[(size_t)i] = used;
    used += need;
    cp->peak = std::max(cp->peak, used);
    cp->columns += (int64_t)L + 1;
  }
  if (nPlain > 0) cp->batchStart.push_back(n_reads);
  return DNAS_OK;
}

// the kernels index their substitution tables with the base codes: anything but 0..3 must not reach them
int check_device_bases(dnas_model* m, const uint8_t* d_bases, size_t nBases) {
  if (!nBases) return DNAS_OK;
  HIP_TRY(hipMemsetAsync(m->dRounds + 8, 0, sizeof(unsigned long long), m->stream));
  hipLaunchKernelGGL(check_bases_kernel, dim3((unsigned)std::min<size_t>((nBases + 255) / 256, 4096)), dim3(256), 0, m->stream,
                     d_bases, nBases, m->dRounds + 8);
  HIP_TRY(hipGetLastError());
  unsigned long long bad = 0;
  HIP_TRY(hipMemcpyAsync(&bad, m->dRounds + 8, sizeof bad, hipMemcpyDeviceToHost, m->stream));
  HIP_TRY(hipStreamSynchronize(m->stream));
  if (bad) return dnas::fail(DNAS_E_BAD_BASE, "base code > 3 in the device buffer (bases are 0..3 = ACGT)");
  return DNAS_OK;
}

// One fill launch over nB reads (their indices at batchRead, their lattices at arena + slots[.]) on the model's fill stream;
// colRange: the columns to fill per read (segments), or null for whole reads.
struct FillLauncher {
  dnas_model* m;
  const uint8_t* d_bases;
  double* d_out_loglike;
  size_t syncAt = 0, launches = 0;

  int operator()(const int32_t* batchRead, const uint64_t* slots, int nB, const int* colRange) {
    ++launches;
    RoctxRange fillRange(m->tier == 2 ? "viterbi fill (tier C)" : (m->tier == 1 ? "viterbi fill (tier A)" : "viterbi fill (tier B)"));
    if (m->tier == 0) {
      const int maskWords = (m->dm.N + 31) / 32 + 1;
      hipLaunchKernelGGL(viterbi_fill_kernel, dim3(nB), dim3(kFillThreads), 2 * (size_t)maskWords * sizeof(unsigned), m->stream, m->dm, d_bases,
                         (const uint64_t*)m->dReadOff, batchRead, slots, m->arena, d_out_loglike, m->dRounds, maskWords, colRange);
      HIP_TRY(hipGetLastError());
      return DNAS_OK;
    }
    TierALaunch la{m->argsA, m->dEntTab, m->dMetaTab, d_bases, m->dReadOff, batchRead, slots,
                   m->arena, d_out_loglike, m->dRounds, nullptr, nullptr, nullptr, 0, nB, 0ull, 0ull, colRange, m->clusterSpread};
    unsigned grid = (unsigned)nB;
    int nClusters = 0;
    if (m->tier == 2) {
      // a cluster of G work-groups per read, persistent over the reads of the launch.  Blocks b and b + 8 share
      // an XCD (observed dispatch order; the kernel is correct under any placement): the members of a cluster are
      // 8 blocks apart, cluster = (b / 8 / G) * 8 + b % 8.
      const int G = m->plan.G;
      nClusters = std::min(nB, m->maxClusters);
      la.xbuf = m->dXbuf; la.syncWords = m->dSync; la.foldTab = m->dFoldTab; la.nClusters = nClusters; la.timeoutTicks = m->timeoutTicks; la.arriveTicks = m->arriveTicks;
      grid = m->clusterSpread ? (unsigned)(G * nClusters) : (unsigned)(8 * G * ((nClusters + 7) / 8));
      const size_t nX = m->xStride * (size_t)nClusters;
      hipLaunchKernelGGL(fill_neginf_kernel, dim3((unsigned)((nX + 255) / 256)), dim3(256), 0, m->stream, m->dXbuf, nX);
      HIP_TRY(hipGetLastError());
      if (!m->syncLat.empty() && !m->clusterSpread && nClusters <= 8) {
        // A launch of a few clusters -- one read alone -- is a latency matter, and which XCD its first block goes to is not fixed:
        // the dispatcher deals the work-groups of successive kernels round the XCDs in one sequence.  A one-block probe says where
        // that sequence stands right now (the fill is the next kernel: its block 0 goes where the probe's one block went -- observed
        // over and over, `DNAS_SYNC_DEBUG=1` prints both), and the
        // sync blocks move to the place in their window that this XCD reaches soonest.  Speed only: a wrong guess costs what a
        // badly placed block costs, 42 against 54 ms for a ~980-nt read of the 46 670-state machine on 16 work-groups.
        if (!m->xccProbe) HIP_TRY(hipHostMalloc((void**)&m->xccProbe, sizeof(unsigned), hipHostMallocDefault));
        HIP_TRY(hipMemsetAsync(m->dSyncBase, 0, (size_t)m->maxClusters * 64 * sizeof(unsigned) + kSyncWindow, m->stream));
        *m->xccProbe = 0xffu;
        hipLaunchKernelGGL(xcc_probe_kernel, dim3(1), dim3(64), 0, m->stream, m->xccProbe);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(m->stream));
        if (*m->xccProbe < 8u) {
          m->syncOffNow = place_sync_words(m, *m->xccProbe & 7u);
          m->dSync = m->dSyncBase + m->syncOffNow / sizeof(unsigned);
          la.syncWords = m->dSync;
          if (getenv("DNAS_SYNC_DEBUG")) fprintf(stderr, "sync debug: probe on XCC %u, sync words at +%zu\n", *m->xccProbe, m->syncOffNow);
        }
      } else {
        HIP_TRY(hipMemsetAsync(m->dSync, 0, (size_t)nClusters * 64 * sizeof(unsigned), m->stream));
      }
    }
    size_t laSize = sizeof la;
    void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &la, HIP_LAUNCH_PARAM_BUFFER_SIZE, &laSize, HIP_LAUNCH_PARAM_END};
    HIP_TRY(hipModuleLaunchKernel(colRange ? m->fillSeg : m->fillA, grid, 1, 1, (unsigned)m->plan.T, 1, 1, (unsigned)m->plan.ldsBytes,
                                  m->stream, nullptr, config));
    if (m->tier == 2) {
      // the watchdog words of this launch: [1] of every sync block (checked in dnas_model_sync)
      HIP_TRY(hipMemcpyAsync(m->syncCheck + syncAt * (size_t)m->maxClusters * 64, m->dSync,
                             (size_t)nClusters * 64 * sizeof(unsigned), hipMemcpyDeviceToHost, m->stream));
      ++syncAt;
    }
    return DNAS_OK;
  }
};

// the register/LDS kernel built with -DDNAS_SEGMENTS=1, compiled when a call first needs it
int ensure_segment_kernel(dnas_model* m) {
  if (m->tier == 0 || m->fillSeg) return DNAS_OK;
  try {
    const std::vector<char> code = dnas::jitCompile(m->jitDefs + "\n-DDNAS_SEGMENTS=1", m->plan.key + "+segments");
    if (hipModuleLoadData(&m->moduleSeg, code.data()) != hipSuccess ||
        hipModuleGetFunction(&m->fillSeg, m->moduleSeg, "viterbi_fill_tiera") != hipSuccess)
      throw std::runtime_error("hipModuleLoadData/GetFunction failed");
  } catch (const std::exception& e) {
    m->fillSeg = nullptr;
    return dnas::fail(DNAS_E_DEVICE, std::string("bounded-memory decode: the segment kernel is unavailable: ") + e.what());
  }
  return DNAS_OK;
}

// The groups of the bounded-memory decode, everything in order on the fill stream.  events: 4 per group (pass 1, pass 2).
int run_segment_groups(dnas_model* m, const CallPlan& cp, const uint64_t* read_offsets, FillLauncher& fill, const uint8_t* d_bases,
                       char* d_out_sym, uint32_t* d_out_len, uint8_t* d_out_status) {
  const DevModel& d = m->dm;
  const size_t colDoubles = (size_t)d.storedLanes * (size_t)d.Npad, H = (size_t)d.D + 1;
  auto lenOf = [&](int64_t i) { return (int64_t)(read_offsets[cp.order[(size_t)i] + 1] - read_offsets[cp.order[(size_t)i]]); };
  // per group and segment: the column range and the (virtual) lattice origin of every read
  std::vector<int> ranges(2 * cp.tabEntries);
  std::vector<uint64_t> segSlot(cp.tabEntries);
  for (const SegmentGroup& g : cp.groups)
    for (int64_t sg = 0; sg < g.nSeg; ++sg)
      for (int64_t j = 0; j < g.n; ++j) {
        const size_t at = g.tabAt + (size_t)sg * (size_t)g.n + (size_t)j;
        const int64_t L = lenOf(g.first + j), c0 = sg * g.C;
        ranges[2 * at] = (int)c0;
        ranges[2 * at + 1] = (int)std::min(L, c0 + g.C - 1);
        // column c of the segment sits at work(j) + (c - c0 + H) columns: the origin the kernels add c * column to
        segSlot[at] = (uint64_t)((size_t)j * g.workStride + H * colDoubles) - (uint64_t)((size_t)c0 * colDoubles);
      }
  if (m->dColRange) { (void)hipFree(m->dColRange); (void)hipFree(m->dSegSlot); (void)hipFree(m->dWalks); }
  m->dColRange = nullptr; m->dSegSlot = nullptr; m->dWalks = nullptr;
  HIP_TRY(hipMalloc((void**)&m->dColRange, ranges.size() * sizeof(int)));
  HIP_TRY(hipMalloc((void**)&m->dSegSlot, segSlot.size() * sizeof(uint64_t)));
  HIP_TRY(hipMalloc((void**)&m->dWalks, (size_t)cp.nSegmented * sizeof(TracebackWalk)));
  HIP_TRY(hipMemcpy(m->dColRange, ranges.data(), ranges.size() * sizeof(int), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(m->dSegSlot, segSlot.data(), segSlot.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(m->dWalks, 0, (size_t)cp.nSegmented * sizeof(TracebackWalk)));
  const size_t headDoubles = (H + 1) * colDoubles;
  for (size_t gi = 0; gi < cp.groups.size(); ++gi) {
    const SegmentGroup& g = cp.groups[gi];
    RoctxRange groupRange("viterbi bounded-memory group");
    double* const work = m->arena;
    double* const ckpt = m->arena + (size_t)g.n * g.workStride;
    // reads of the group that reach segment sg (sorted longest first: a prefix)
    auto reach = [&](int64_t sg) {
      int64_t k = 0;
      while (k < g.n && lenOf(g.first + k) >= sg * g.C) ++k;
      return (int)k;
    };
    auto copyRows = [&](double* dst, size_t dstStride, const double* src, size_t srcStride, int rows) -> int {
      hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)std::min<size_t>((headDoubles + 255) / 256, 256), (unsigned)rows), dim3(256), 0,
                         m->stream, dst, src, dstStride, srcStride, headDoubles);
      HIP_TRY(hipGetLastError());
      return DNAS_OK;
    };
    int rc;
    HIP_TRY(hipEventRecord(m->events[4 * gi], m->stream));
    for (int64_t sg = 0; sg < g.nSeg; ++sg) {            // pass 1
      const int nAct = reach(sg);
      if (nAct == 0) break;
      if (sg > 0) {
        // the last H columns of the segment before and the hand-over lane behind them: kept, and moved to the front
        if ((rc = copyRows(ckpt + (size_t)sg * headDoubles, g.ckStride, work + (size_t)g.C * colDoubles, g.workStride, nAct)) != DNAS_OK) return rc;
        if ((rc = copyRows(work, g.workStride, ckpt + (size_t)sg * headDoubles, g.ckStride, nAct)) != DNAS_OK) return rc;
      }
      const size_t at = g.tabAt + (size_t)sg * (size_t)g.n;
      if ((rc = fill(m->dBatchRead + g.first, m->dSegSlot + at, nAct, m->dColRange + 2 * at)) != DNAS_OK) return rc;
    }
    HIP_TRY(hipEventRecord(m->events[4 * gi + 1], m->stream));
    HIP_TRY(hipEventRecord(m->events[4 * gi + 2], m->stream));
    for (int64_t sg = g.nSeg - 1; sg >= 0; --sg) {        // pass 2
      const int nAct = reach(sg), nAgain = sg + 1 < g.nSeg ? reach(sg + 1) : 0;
      if (nAct == 0) continue;
      const size_t at = g.tabAt + (size_t)sg * (size_t)g.n;
      if (nAgain > 0) {
        if (sg > 0 && (rc = copyRows(work, g.workStride, ckpt + (size_t)sg * headDoubles, g.ckStride, nAgain)) != DNAS_OK) return rc;
        if ((rc = fill(m->dBatchRead + g.first, m->dSegSlot + at, nAgain, m->dColRange + 2 * at)) != DNAS_OK) return rc;
      }
      hipLaunchKernelGGL(viterbi_traceback_wave_kernel, dim3((nAct + 3) / 4), dim3(256), 0, m->stream, d, d_bases,
                         (const uint64_t*)m->dReadOff, (const int32_t*)(m->dBatchRead + g.first), (const uint64_t*)(m->dSegSlot + at),
                         (const double*)m->arena, d_out_sym, (const uint64_t*)m->dOutOff, d_out_len, d_out_status, nAct, m->dEvents,
                         (const uint64_t*)m->dEvOff, m->dEvLen, (const int*)(m->dColRange + 2 * at), m->dWalks + g.first);
      HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(m->events[4 * gi + 3], m->stream));
  }
  return DNAS_OK;
}

}  // namespace

extern "C" int dnas_viterbi_batch_device(dnas_model* m, int64_t n_reads, const uint64_t* read_offsets,
                                         const uint8_t* d_bases, char* d_out_sym, const uint64_t* out_offsets,
                                         uint32_t* d_out_len, double* d_out_loglike, uint8_t* d_out_status) {
  if (!m || n_reads < 0 || (n_reads > 0 && (!read_offsets || !d_bases || !d_out_sym || !out_offsets || !d_out_len ||
                                            !d_out_loglike || !d_out_status)))
    return dnas::fail(DNAS_E_INVALID, "dnas_viterbi_batch_device: bad argument");
  RoctxRange callRange("dnas_viterbi_batch_device");
  if (n_reads > 0x7fffffffll) return dnas::fail(DNAS_E_UNSUPPORTED, "more than 2^31-1 reads in one call");
  HIP_TRY(hipSetDevice(m->device));
  // the previous call's events/stat buffers are about to be reused
  HIP_TRY(hipStreamSynchronize(m->stream));
  HIP_TRY(hipStreamSynchronize(m->stream2));
  m->stats = dnas_batch_stats{};
  m->statsPending = false;
  if (n_reads == 0) return DNAS_OK;
  int rc = check_device_bases(m, d_bases + read_offsets[0], (size_t)(read_offsets[n_reads] - read_offsets[0]));
  if (rc != DNAS_OK) return rc;
  const DevModel& d = m->dm;

  // plan the call against the arena cap; if the device cannot give that much any more (the cap was taken from the free memory
  // when the model was created -- another process may have come since), plan once more against what is free now: smaller
  // batches, or segments for the reads that no longer fit
  CallPlan cp;
  for (int attempt = 0;; ++attempt) {
    cp = CallPlan();
    if ((rc = plan_call(m, n_reads, read_offsets, &cp)) != DNAS_OK) return rc;
    const size_t need = std::max((cp.batchStart.size() > 2 ? 2 : 1) * cp.peak, cp.groupPeak) * sizeof(double);
    if (need <= m->arenaBytes) break;
    if (m->arena) HIP_TRY(hipFree(m->arena));
    m->arena = nullptr;
    m->arenaBytes = 0;
    const hipError_t e = hipMalloc((void**)&m->arena, need);
    if (e == hipSuccess) { m->arenaBytes = need; break; }
    (void)hipGetLastError();
    m->arena = nullptr;
    size_t freeB = 0, totalB = 0;
    if (attempt > 0 || e != hipErrorOutOfMemory || hipMemGetInfo(&freeB, &totalB) != hipSuccess || freeB / 10 * 8 >= m->arenaCap)
      return dnas::fail(DNAS_E_DEVICE, "hipMalloc of the lattice arena (" + std::to_string(need) + " bytes): " + hipGetErrorString(e));
    m->arenaCap = freeB / 10 * 8;
  }
  std::vector<uint64_t>& slotOff = cp.slotOff;
  const std::vector<int64_t>& batchStart = cp.batchStart;
  const size_t nBatches = batchStart.size() - 1, nGroups = cp.groups.size();
  m->lastCheckpointed = cp.nSegmented;
  m->lastBatchStart = batchStart;
  const bool pingPong = nBatches > 1;
  m->halfDoubles = cp.peak;
  if ((size_t)n_reads + 1 > m->schedCap) {
    if (m->dBatchRead) { (void)hipFree(m->dBatchRead); (void)hipFree(m->dSlotOff); (void)hipFree(m->dReadOff); (void)hipFree(m->dOutOff); }
    m->dBatchRead = nullptr; m->dSlotOff = m->dReadOff = m->dOutOff = nullptr;
    m->schedCap = 0;
    const size_t cap = (size_t)n_reads + 1;
    HIP_TRY(hipMalloc((void**)&m->dBatchRead, cap * sizeof(int32_t)));
    HIP_TRY(hipMalloc((void**)&m->dSlotOff, cap * sizeof(uint64_t)));
    HIP_TRY(hipMalloc((void**)&m->dReadOff, cap * sizeof(uint64_t)));
    HIP_TRY(hipMalloc((void**)&m->dOutOff, cap * sizeof(uint64_t)));
    m->schedCap = cap;
  }
  // slot offsets of odd batches point into the second half
  if (pingPong)
    for (size_t b = 1; b < nBatches; b += 2)
      for (int64_t i = batchStart[b]; i < batchStart[b + 1]; ++i) slotOff[(size_t)i] += m->halfDoubles;
  m->lastSlotOff = slotOff;
  m->lastBatchRead = cp.order;
  m->lastReadOff.assign(read_offsets, read_offsets + n_reads + 1);
  m->lastBases = d_bases;
  // host vectors stay alive until the copies complete (synchronous copies keep this simple)
  HIP_TRY(hipMemcpy(m->dBatchRead, cp.order.data(), (size_t)n_reads * sizeof(int32_t), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(m->dSlotOff, slotOff.data(), (size_t)n_reads * sizeof(uint64_t), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(m->dReadOff, read_offsets, ((size_t)n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(m->dOutOff, out_offsets, ((size_t)n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
  HIP_TRY(hipMemsetAsync(m->dRounds, 0, 8 * sizeof(unsigned long long), m->stream));

  if (m->dEvents) { (void)hipFree(m->dEvents); (void)hipFree(m->dEvOff); (void)hipFree(m->dEvLen); m->dEvents = nullptr; m->dEvOff = nullptr; m->dEvLen = nullptr; }
  if (m->eventLog) {
    // at most one event per traceback step: a read of L bases takes fewer than 2L + 8 + (null depth) steps
    m->evOff.assign((size_t)n_reads + 1, 0);
    for (int64_t i = 0; i < n_reads; ++i) m->evOff[(size_t)i + 1] = m->evOff[(size_t)i] + 3 * (read_offsets[i + 1] - read_offsets[i]) + 64;
    HIP_TRY(hipMalloc((void**)&m->dEvents, std::max<size_t>(m->evOff.back(), 1) * sizeof(unsigned long long)));
    HIP_TRY(hipMalloc((void**)&m->dEvOff, ((size_t)n_reads + 1) * sizeof(uint64_t)));
    HIP_TRY(hipMalloc((void**)&m->dEvLen, (size_t)n_reads * sizeof(uint32_t)));
    HIP_TRY(hipMemcpy(m->dEvOff, m->evOff.data(), ((size_t)n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(m->dEvLen, 0, (size_t)n_reads * sizeof(uint32_t)));
  }
  const size_t nTimed = nBatches + nGroups;              // 4 timing events each: the groups first, then the batches
  while (m->events.size() < 4 * nTimed) {
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    m->events.push_back(e);
  }
  while (m->sync.size() < 2 * nBatches) {
    hipEvent_t e;
    HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    m->sync.push_back(e);
  }
  if (m->tier == 2) {
    const size_t wantWords = (nBatches + cp.groupLaunches) * (size_t)m->maxClusters * 64;
    if (wantWords > m->syncCheckCap) {
      if (m->syncCheck) (void)hipHostFree(m->syncCheck);
  if (m->xccProbe) (void)hipHostFree(m->xccProbe);
      m->syncCheck = nullptr; m->syncCheckCap = 0; m->syncCheckWords = 0;
      HIP_TRY(hipHostMalloc((void**)&m->syncCheck, wantWords * sizeof(unsigned), hipHostMallocDefault));   // pinned: the copies after each fill stay asynchronous
      m->syncCheckCap = wantWords;
    }
    m->syncCheckWords = wantWords;
    memset(m->syncCheck, 0, m->syncCheckWords * sizeof(unsigned));
    m->syncLaunches = nBatches + cp.groupLaunches;
  }

  FillLauncher fill{m, d_bases, d_out_loglike};
  if (nGroups) {
    if ((rc = ensure_segment_kernel(m)) != DNAS_OK) return rc;
    if ((rc = run_segment_groups(m, cp, read_offsets, fill, d_bases, d_out_sym, d_out_len, d_out_status)) != DNAS_OK) return rc;
  }

  for (size_t b = 0; b < nBatches; ++b) {
    const int64_t s = batchStart[b];
    const int nB = (int)(batchStart[b + 1] - s);
    const size_t ev = 4 * (nGroups + b);
    // the half this batch fills was last read by the traceback of batch b-2
    if (b >= 2) HIP_TRY(hipStreamWaitEvent(m->stream, m->sync[2 * (b - 2) + 1], 0));
    HIP_TRY(hipEventRecord(m->events[ev], m->stream));
    if ((rc = fill(m->dBatchRead + s, m->dSlotOff + s, nB, nullptr)) != DNAS_OK) return rc;
    HIP_TRY(hipEventRecord(m->events[ev + 1], m->stream));
    HIP_TRY(hipEventRecord(m->sync[2 * b], m->stream));
    RoctxRange tbRange("viterbi traceback");
    HIP_TRY(hipStreamWaitEvent(m->stream2, m->sync[2 * b], 0));
    HIP_TRY(hipEventRecord(m->events[ev + 2], m->stream2));
    // one wave per read finishes a read 4-5x sooner but costs about three times the CU time: for batches small enough
    // that the traceback is what the caller waits for (tier C, short jobs); large batches trace back thread-per-read in
    // blocks of 128 threads, 16 reads per wave (round 4; ~9 ms beside the next batch's fill.  Round 3: 128 reads per block, six CUs
    // for ~18 ms; as two blocks of 512 it held its CUs for 22-37 ms,
    // and every fifth fill launch paid a fourth round of work-groups on the shader engine it sat on: 51 ms instead of 43.  Also
    // tried: a wave per read launched when the next fill opens its last round -- the work-group that opens it bumped a signal
    // word the traceback stream waited on --: its 180 blocks then crowd the forty idle CUs and every launch took 48 ms) --
    // except the last batch of a call, which has the GPU to itself
    if (m->waveTraceback && (nB <= 256 || b + 1 == nBatches))
      hipLaunchKernelGGL(viterbi_traceback_wave_kernel, dim3((nB + 3) / 4), dim3(256), 0, m->stream2, d, d_bases,
                         (const uint64_t*)m->dReadOff, (const int32_t*)(m->dBatchRead + s), (const uint64_t*)(m->dSlotOff + s),
                         (const double*)m->arena, d_out_sym, (const uint64_t*)m->dOutOff, d_out_len, d_out_status, nB, m->dEvents,
                         (const uint64_t*)m->dEvOff, m->dEvLen, (const int*)nullptr, (TracebackWalk*)nullptr);
    else
    {
      const int perBlock = (m->tbThreads / 64) * m->tbLanes;     // reads a block walks: tbLanes of every wave's 64 lanes
      hipLaunchKernelGGL(viterbi_traceback_kernel, dim3((nB + perBlock - 1) / perBlock), dim3(m->tbThreads), 0,
                         m->stream2, d, d_bases, (const uint64_t*)m->dReadOff, (const int32_t*)(m->dBatchRead + s),
                         (const uint64_t*)(m->dSlotOff + s), (const double*)m->arena, d_out_sym,
                         (const uint64_t*)m->dOutOff, d_out_len, d_out_status, nB, m->dEvents, (const uint64_t*)m->dEvOff, m->dEvLen,
                         m->tbLanes);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(m->events[ev + 3], m->stream2));
    HIP_TRY(hipEventRecord(m->sync[2 * b + 1], m->stream2));
  }
  // trim so collect_stats sees exactly this call's events
  while (m->events.size() > 4 * nTimed) {
    (void)hipEventDestroy(m->events.back());
    m->events.pop_back();
  }
  m->stats.fill_launches = (int64_t)fill.launches;
  m->stats.checkpointed_reads = cp.nSegmented;
  m->stats.columns = cp.columns;
  m->stats.lattice_bytes = (int64_t)(8 * ((size_t)d.D + 2) * (size_t)d.N) * cp.columns;
  m->statsPending = true;
  return DNAS_OK;
}

extern "C" int dnas_viterbi_batch(dnas_model* m, int64_t n_reads, const uint64_t* read_offsets, const uint8_t* bases,
                                  char* out_sym, const uint64_t* out_offsets, uint32_t* out_len, double* out_loglike,
                                  uint8_t* out_status) {
  if (!m || n_reads < 0) return dnas::fail(DNAS_E_INVALID, "dnas_viterbi_batch: bad argument");
  if (n_reads == 0) return DNAS_OK;
  if (!read_offsets || !bases || !out_sym || !out_offsets || !out_len || !out_loglike || !out_status)
    return dnas::fail(DNAS_E_INVALID, "dnas_viterbi_batch: null argument");
  HIP_TRY(hipSetDevice(m->device));
  const size_t nBases = (size_t)(read_offsets[n_reads] - read_offsets[0]);
  if (read_offsets[0] != 0) return dnas::fail(DNAS_E_INVALID, "read_offsets[0] must be 0");
  for (size_t i = 0; i < nBases; ++i)
    if (bases[i] > 3) return dnas::fail(DNAS_E_BAD_BASE, "base code > 3 at offset " + std::to_string(i));
  const size_t nOut = (size_t)out_offsets[n_reads];
  // (the streams are idle: the call before was synchronised before it returned its results)
  auto grow = [&](void** p, size_t* cap, size_t need, size_t elem) -> int {
    if (need <= *cap && *p) return DNAS_OK;
    if (*p) HIP_TRY(hipFree(*p));
    *p = nullptr; *cap = 0;
    const size_t want = std::max<size_t>(need + need / 4, 256);
    HIP_TRY(hipMalloc(p, want * elem));
    *cap = want;
    return DNAS_OK;
  };
  int rc;
  m->lastBases = nullptr;
  if ((rc = grow((void**)&m->ioBases, &m->ioBasesCap, nBases, 1)) != DNAS_OK) return rc;
  if ((rc = grow((void**)&m->ioSym, &m->ioSymCap, nOut, 1)) != DNAS_OK) return rc;
  if ((size_t)n_reads > m->ioReadsCap || !m->ioLen) {
    if (m->ioLen) { (void)hipFree(m->ioLen); (void)hipFree(m->ioLL); (void)hipFree(m->ioSt); }
    m->ioLen = nullptr; m->ioLL = nullptr; m->ioSt = nullptr; m->ioReadsCap = 0;
    const size_t want = (size_t)n_reads + (size_t)n_reads / 4 + 64;
    HIP_TRY(hipMalloc((void**)&m->ioLen, want * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void**)&m->ioLL, want * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&m->ioSt, want));
    m->ioReadsCap = want;
  }
  if (nBases) HIP_TRY(hipMemcpy(m->ioBases, bases, nBases, hipMemcpyHostToDevice));
  rc = dnas_viterbi_batch_device(m, n_reads, read_offsets, m->ioBases, m->ioSym, out_offsets, m->ioLen, m->ioLL, m->ioSt);
  if (rc == DNAS_OK) rc = dnas_model_sync(m);
  if (rc != DNAS_OK) return rc;
  if (nOut) HIP_TRY(hipMemcpy(out_sym, m->ioSym, nOut, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(out_len, m->ioLen, (size_t)n_reads * sizeof(uint32_t), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(out_loglike, m->ioLL, (size_t)n_reads * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(out_status, m->ioSt, (size_t)n_reads, hipMemcpyDeviceToHost));
  return DNAS_OK;
}

extern "C" int dnas_model_read_lattice(dnas_model* m, int64_t slot, int64_t len, double* out) {
  if (!m || !out || slot < 0 || len < 0 || m->lastReadOff.empty() || (size_t)slot + 1 >= m->lastReadOff.size())
    return dnas::fail(DNAS_E_INVALID, "dnas_model_read_lattice: bad argument");
  HIP_TRY(hipSetDevice(m->device));
  HIP_TRY(hipStreamSynchronize(m->stream));
  HIP_TRY(hipStreamSynchronize(m->stream2));
  if (!m->lastBases) return dnas::fail(DNAS_E_INVALID, "dnas_model_read_lattice: the read bases of the last call are gone");
  const DevModel& d = m->dm;
  const size_t lanes = (size_t)d.D + 2;
  // `slot` indexes the caller's read order; find its arena slot.  The arena has two halves that the batches of a call
  // use in turn: only the lattices of the last two batches still exist.
  size_t pos = 0;
  for (; pos < m->lastBatchRead.size(); ++pos)
    if (m->lastBatchRead[pos] == (int32_t)slot) break;
  if (pos == m->lastBatchRead.size()) return dnas::fail(DNAS_E_INVALID, "no such read in the last call");
  if ((int64_t)pos < m->lastCheckpointed)
    return dnas::fail(DNAS_E_INVALID, "dnas_model_read_lattice: that read was decoded in segments (bounded-memory decode); its lattice was never whole");
  {
    size_t batchOf = 0;
    while (batchOf + 1 < m->lastBatchStart.size() && (int64_t)pos >= m->lastBatchStart[batchOf + 1]) ++batchOf;
    const size_t nB = m->lastBatchStart.size() - 1;
    if (batchOf + 2 < nB) return dnas::fail(DNAS_E_INVALID, "dnas_model_read_lattice: that read's lattice has been overwritten by a later batch of the call");
  }
  if ((uint64_t)len != m->lastReadOff[slot + 1] - m->lastReadOff[slot]) return dnas::fail(DNAS_E_INVALID, "length mismatch");
  const size_t n = (size_t)(len + 1) * lanes * (size_t)d.N;
  double* dOut = nullptr;
  HIP_TRY(hipMalloc((void**)&dOut, n * sizeof(double)));
  hipLaunchKernelGGL(expand_lattice_kernel, dim3((unsigned)(len + 1)), dim3(256), 0, m->stream, d,
                     m->lastBases + m->lastReadOff[slot], (const double*)(m->arena + m->lastSlotOff[pos]), dOut);
  hipError_t e = hipStreamSynchronize(m->stream);
  if (e == hipSuccess) e = hipMemcpy(out, dOut, n * sizeof(double), hipMemcpyDeviceToHost);
  (void)hipFree(dOut);
  if (e != hipSuccess) return dnas::fail(DNAS_E_DEVICE, hipGetErrorString(e));
  return DNAS_OK;
}

// Analysis / test aid: the tier-A tables of a machine exactly as the kernel receives them (no GPU
// needed): row shapes [K][2] = {out-edge entries, S stripe or -1}, the entry table [n_entries][T]
// and the meta table [K][T].  Either output may be NULL; *n_entries / *n_s_rows are always set.
extern "C" int dnas_tiera_plan_tables(const dnas_flat_model* fm, int32_t* row_shapes, uint32_t* entries, size_t entries_cap,
                                      uint32_t* meta, int32_t* n_entries, int32_t* n_s_rows) {
  if (!fm) return dnas::fail(DNAS_E_INVALID, "null argument");
  try {
    const int threads = env_threads(dnas::kTierAThreads);
    const dnas::TierAPlan p = dnas::buildTierAPlan(*fm, threads, env_or_recorded_choice(fm, 1, threads));
    if (!p.ok) return dnas::fail(DNAS_E_UNSUPPORTED, p.whyNot);
    if (n_entries) *n_entries = p.nEntries;
    if (n_s_rows) *n_s_rows = p.nSRows;
    if (row_shapes) for (int k = 0; k < p.K; ++k) { row_shapes[2 * k] = p.rows[k].nOut; row_shapes[2 * k + 1] = p.rows[k].sIdx; }
    if (entries) {
      if (entries_cap < p.entTab.size()) return dnas::fail(DNAS_E_INVALID, "entry buffer too small");
      memcpy(entries, p.entTab.data(), p.entTab.size() * sizeof(uint32_t));
    }
    if (meta) memcpy(meta, p.metaTab.data(), p.metaTab.size() * sizeof(uint32_t));
    return DNAS_OK;
  } catch (const std::exception& e) {
    return dnas::fail(DNAS_E_DEVICE, e.what());
  }
}

// Build-time helper: specialise and compile the tier-A kernel for a machine without touching
// a GPU (fills dnastore_amd/kcache so that the GPU box finds the code object ready).
extern "C" int dnas_tiera_precompile(const dnas_flat_model* fm, char* note, size_t note_cap) {
  if (!fm) return dnas::fail(DNAS_E_INVALID, "null argument");
  try {
    // the row program a model of this machine will run: as the environment says, else as its tuning record says, else the
    // default one
    const int threads = env_threads(dnas::kTierAThreads);   // as dnas_model_create_ex reads it
    const dnas::TierAPlan p = dnas::buildTierAPlan(*fm, threads, env_or_recorded_choice(fm, 1, threads));
    std::string msg;
    if (!p.ok) {
      msg = "tier B: " + p.whyNot;
    } else {
      (void)dnas::jitCompile(p.defines, p.key);
      msg = "tier A: " + p.key + " lds=" + std::to_string(p.ldsBytes) + " fill=" + std::to_string(p.fillRatio) +
            " reads=" + std::to_string(p.sweepReads) + " entries=" + std::to_string(p.nEntries) + " back=" +
            std::to_string(p.backEdgesOnWalk) + " sameWave=" + std::to_string(p.sameWave);
    }
    if (note && note_cap) { strncpy(note, msg.c_str(), note_cap - 1); note[note_cap - 1] = 0; }
    return DNAS_OK;
  } catch (const std::exception& e) {
    return dnas::fail(DNAS_E_DEVICE, e.what());
  }
}

extern "C" const char* dnas_model_tier(const dnas_model* m) { return m ? m->tierNote.c_str() : ""; }

extern "C" int dnas_model_set_event_log(dnas_model* m, int on) {
  if (!m) return dnas::fail(DNAS_E_INVALID, "null model");
  if (on && m->dm.D > 13)     // a duplication event carries its bases in 26 bits (viterbi_kernels.hip)
    return dnas::fail(DNAS_E_UNSUPPORTED, "the traceback event log holds at most 13 duplicated bases per event; this model has " + std::to_string(m->dm.D) + " duplication lanes");
  m->eventLog = on != 0;
  return DNAS_OK;
}

// The traceback events of read `read_index` of the last call, in the order the traceback met them.
extern "C" int dnas_model_read_events(dnas_model* m, int64_t read_index, uint64_t* out, int64_t cap, int64_t* n_events) {
  if (!m || !n_events || read_index < 0) return dnas::fail(DNAS_E_INVALID, "dnas_model_read_events: bad argument");
  *n_events = 0;
  if (!m->dEvents || (size_t)read_index + 1 >= m->evOff.size()) return dnas::fail(DNAS_E_INVALID, "no event log for that read (dnas_model_set_event_log before the call)");
  HIP_TRY(hipSetDevice(m->device));
  HIP_TRY(hipStreamSynchronize(m->stream2));
  uint32_t n = 0;
  HIP_TRY(hipMemcpy(&n, m->dEvLen + read_index, sizeof n, hipMemcpyDeviceToHost));
  *n_events = n;
  if (out && cap > 0) {
    const size_t take = (size_t)std::min<int64_t>(cap, n);
    if (take) HIP_TRY(hipMemcpy(out, m->dEvents + m->evOff[(size_t)read_index], take * sizeof(uint64_t), hipMemcpyDeviceToHost));
  }
  return DNAS_OK;
}

// Tier C: compile the cluster kernel for a machine ahead of time (members = 0: the smallest cluster that fits).
extern "C" int dnas_tierc_precompile(const dnas_flat_model* fm, int32_t members, char* note, size_t note_cap) {
  if (!fm) return dnas::fail(DNAS_E_INVALID, "null argument");
  try {
    const int threads = env_threads(0);   // (the same thread count names the record and shapes the plan)
    const dnas::TierAPlan p = dnas::chooseClusterPlan(*fm, members, threads, env_or_recorded_choice(fm, members >= 2 ? members : 0, threads));
    if (!p.ok) return dnas::fail(DNAS_E_UNSUPPORTED, p.whyNot);
    (void)dnas::jitCompile(p.defines, p.key);
    const std::string msg = "tier C: G=" + std::to_string(p.G) + " K=" + std::to_string(p.K) + " inbox rows " + std::to_string(p.nGRows) +
                            " exchange edges " + std::to_string(p.crossEdges) + " lds=" + std::to_string(p.ldsBytes) + " entries=" +
                            std::to_string(p.nEntries) + " back=" + std::to_string(p.backEdgesOnWalk) + " " + p.key;
    if (note && note_cap) { strncpy(note, msg.c_str(), note_cap - 1); note[note_cap - 1] = 0; }
    return DNAS_OK;
  } catch (const std::exception& e) {
    return dnas::fail(DNAS_E_DEVICE, e.what());
  }
}

namespace {
// members = 1: the tier-A plan, else the cluster plan -- under the row program a model of this machine follows (record, environment)
dnas::TierAPlan plan_as_a_model_would(const dnas_flat_model* fm, int members) {
  if (members == 1) {
    const int threads = env_threads(dnas::kTierAThreads);
    return dnas::buildTierAPlan(*fm, threads, env_or_recorded_choice(fm, 1, threads));
  }
  const int threads = env_threads(0);
  return dnas::chooseClusterPlan(*fm, members, threads, env_or_recorded_choice(fm, members >= 2 ? members : 0, threads));
}
}  // namespace

// Analysis / test aid: the tier-C tables of a machine exactly as the kernel receives them (no GPU needed).
// info[8] = {G, K, T, entries per member, S stripes, exchange rows, proxies, 0}; every other output may be NULL:
// row_shapes[K][6], entries[G][n_entries][T], meta[G][K][T], member_of[N], lds_index[N] = row*T + lane inside
// the member, lattice_slot[N], fold[G][inbox rows][T].
extern "C" int dnas_tierc_plan(const dnas_flat_model* fm, int32_t members, int32_t* info, int32_t* row_shapes, uint32_t* entries,
                               size_t entries_cap, uint32_t* meta, int32_t* member_of, int32_t* lds_index, int32_t* lattice_slot,
                               uint32_t* fold) {
  if (!fm || !info) return dnas::fail(DNAS_E_INVALID, "null argument");
  try {
    const dnas::TierAPlan p = plan_as_a_model_would(fm, members);
    if (!p.ok) return dnas::fail(DNAS_E_UNSUPPORTED, p.whyNot);
    info[0] = p.G; info[1] = p.K; info[2] = p.T; info[3] = p.nEntries; info[4] = p.nSRows; info[5] = p.nGRows; info[6] = (int32_t)p.proxyMember.size(); info[7] = 0;
    if (row_shapes)
      for (int k = 0; k < p.K; ++k) {
        const dnas::RowShape& r = p.rows[k];
        const int v[6] = {r.nOut, r.sIdx, r.kind, r.cls, r.full, r.gOut};
        memcpy(row_shapes + 6 * k, v, sizeof v);
      }
    if (entries) {
      if (entries_cap < p.entTab.size()) return dnas::fail(DNAS_E_INVALID, "entry buffer too small");
      memcpy(entries, p.entTab.data(), p.entTab.size() * sizeof(uint32_t));
    }
    if (meta) memcpy(meta, p.metaTab.data(), p.metaTab.size() * sizeof(uint32_t));
    if (member_of) memcpy(member_of, p.memberOf.data(), (size_t)p.N * sizeof(int32_t));
    if (lds_index)
      for (size_t idx = 0; idx < p.stateOf.size(); ++idx)
        if (p.stateOf[idx] >= 0) lds_index[p.stateOf[idx]] = (int32_t)(idx % ((size_t)p.K * p.T));
    if (lattice_slot) memcpy(lattice_slot, p.slotOf.data(), (size_t)p.N * sizeof(int32_t));
    if (fold && !p.foldTab.empty()) memcpy(fold, p.foldTab.data(), p.foldTab.size() * sizeof(uint32_t));
    return DNAS_OK;
  } catch (const std::exception& e) {
    return dnas::fail(DNAS_E_DEVICE, e.what());
  }
}

// ... and the places of its proxies (plan.cpp: extra places that combine a member's null edges into one state of another
// member): member and row*T + lane of each, info[6] of them.
extern "C" int dnas_tierc_plan_proxies(const dnas_flat_model* fm, int32_t members, int32_t* proxy_member, int32_t* proxy_lds_index, size_t cap) {
  if (!fm || !proxy_member || !proxy_lds_index) return dnas::fail(DNAS_E_INVALID, "null argument");
  try {
    const dnas::TierAPlan p = plan_as_a_model_would(fm, members);
    if (!p.ok) return dnas::fail(DNAS_E_UNSUPPORTED, p.whyNot);
    if (cap < p.proxyMember.size()) return dnas::fail(DNAS_E_INVALID, "proxy buffer too small");
    if (!p.proxyMember.empty()) {
      memcpy(proxy_member, p.proxyMember.data(), p.proxyMember.size() * sizeof(int32_t));
      memcpy(proxy_lds_index, p.proxyLds.data(), p.proxyLds.size() * sizeof(int32_t));
    }
    return DNAS_OK;
  } catch (const std::exception& e) {
    return dnas::fail(DNAS_E_DEVICE, e.what());
  }
}

// The name of the tuning record of a tier-A machine (tune_row_program above; tools/make_tune_records.py writes the records
// that ship with the library from bench-like reads).
extern "C" int dnas_tune_record_name(const dnas_flat_model* fm, int32_t members, int32_t threads, char* out, size_t cap) {
  if (!fm || !out || cap < 40) return dnas::fail(DNAS_E_INVALID, "dnas_tune_record_name: bad argument");
  // members 1: tier A (threads 0 = 1024); 0 or >= 2: tier C with the smallest / that cluster (threads 0 = chosen by the planner)
  const std::string name = tune_record_name(fm, members, members == 1 && threads <= 0 ? dnas::kTierAThreads : threads);
  strncpy(out, name.c_str(), cap - 1);
  out[cap - 1] = 0;
  return DNAS_OK;
}

// The hash of the fill kernel's source as this library carries it (what a tuning record names as "kernel=").
extern "C" int dnas_kernel_source_hash(char* out, size_t cap) {
  if (!out || cap < 17) return dnas::fail(DNAS_E_INVALID, "dnas_kernel_source_hash: bad argument");
  snprintf(out, cap, "%016llx", dnas::kernelSourceHash());
  return DNAS_OK;
}

extern "C" int dnas_model_cluster_census(dnas_model* m, int32_t* clusters, int32_t* split) {
  if (!m || !clusters || !split) return dnas::fail(DNAS_E_INVALID, "null argument");
  if (m->statsPending) return dnas::fail(DNAS_E_INVALID, "call dnas_model_sync first");
  *clusters = (int32_t)m->clustersSeen; *split = (int32_t)m->clustersSplit;
  return DNAS_OK;
}

// Diagnostic: the 8 words of the rounds/stamps buffer of the last call (word 0 = total rounds).
extern "C" int dnas_model_debug_words(dnas_model* m, unsigned long long* out8) {
  if (!m || !out8) return dnas::fail(DNAS_E_INVALID, "null argument");
  HIP_TRY(hipSetDevice(m->device));
  HIP_TRY(hipStreamSynchronize(m->stream));
  HIP_TRY(hipMemcpy(out8, m->dRounds, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return DNAS_OK;
}

// Analysis / test aid: the tier-A placement of a machine (no GPU needed).  lds_index[N] receives
// row*T + lane of every state; returns T in *threads and K in *rows; DNAS_E_UNSUPPORTED when the
// machine does not fit tier A.
extern "C" int dnas_tiera_plan_slots(const dnas_flat_model* fm, int32_t* lds_index, int32_t* lattice_slot, int32_t* threads,
                                     int32_t* rows) {
  if (!fm) return dnas::fail(DNAS_E_INVALID, "null argument");
  try {
    const dnas::TierAPlan p = plan_as_a_model_would(fm, 1);
    if (!p.ok) return dnas::fail(DNAS_E_UNSUPPORTED, p.whyNot);
    if (threads) *threads = p.T;
    if (rows) *rows = p.K;
    for (int slot = 0; slot < p.NS; ++slot)
      if (p.stateOf[slot] >= 0 && lds_index) lds_index[p.stateOf[slot]] = slot;
    if (lattice_slot) for (int j = 0; j < p.N; ++j) lattice_slot[j] = p.slotOf[j];
    return DNAS_OK;
  } catch (const std::exception& e) {
    return dnas::fail(DNAS_E_DEVICE, e.what());
  }
}
