// Host side of the forward-backward path: dnas_fwdback_estep (expectedCounts,
// reference src/fwdback.cpp:190-209) and dnas_baum_welch (baumWelchParams,
// fwdback.cpp:211-230 with MutatorCounts::mlParams / logPrior, mutator.cpp:167-214).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/dnastore_amd.h"
#include "errors.hpp"
#include "fwdback_device.h"
#include "host/model.hpp"

extern "C" __global__ void fwdback_estep_kernel(FbArgs, const int8_t*, const int64_t*, const int8_t*, const int64_t*,
                                                const int32_t*, const int64_t*, const int32_t*, const int64_t*,
                                                const double*, double*, double*, double*, double*, int64_t, int, int64_t, const int64_t*);
#define FB_ONCHIP_ARGS FbArgs, const int8_t*, const int64_t*, const int8_t*, const int64_t*, const int32_t*, const int64_t*, const int32_t*, \
                       const int64_t*, const double*, const int64_t*, int64_t, double*, double*, int, unsigned long long*, double*, int
// <lanes per pair>x<cells per row served>; p6: up to 6 duplication lengths (the CLI's default model) in registers
extern "C" __global__ void fwdback_onchip8x16_kernel(FB_ONCHIP_ARGS);
extern "C" __global__ void fwdback_onchip16x16_kernel(FB_ONCHIP_ARGS);
extern "C" __global__ void fwdback_onchip16x32_kernel(FB_ONCHIP_ARGS);
extern "C" __global__ void fwdback_onchip32x32_kernel(FB_ONCHIP_ARGS);
extern "C" __global__ void fwdback_onchip8x16p6_kernel(FB_ONCHIP_ARGS);
extern "C" __global__ void fwdback_onchip16x16p6_kernel(FB_ONCHIP_ARGS);
extern "C" __global__ void fwdback_onchip16x32p6_kernel(FB_ONCHIP_ARGS);
extern "C" __global__ void fwdback_onchip32x32p6_kernel(FB_ONCHIP_ARGS);
extern "C" __global__ void fwdback_reduce_kernel(const double*, const double*, int64_t, int, double*);
extern "C" __global__ void fwdback_validate_kernel(int64_t, const int8_t*, const int64_t*, const int8_t*, const int64_t*, const int32_t*,
                                                   const int64_t*, const int32_t*, const int64_t*, unsigned*);
extern "C" __global__ void fwdback_census_kernel(int64_t, int, const int64_t*, const int64_t*, const int32_t*, const int64_t*, const int32_t*,
                                                 const int64_t*, int64_t*);

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      cleanup();                                                                               \
      return dnas::fail(DNAS_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    }                                                                                          \
  } while (0)

#ifndef DNAS_FB_WAVES_PER_CU
#define DNAS_FB_WAVES_PER_CU 12     // 4 SIMDs x DNAS_FB_MIN_WAVES of fwdback_onchip.hip
#endif

namespace {

// log(1 + exp(-x)) at x = n * 1e-4, n = 0..100000: the reference's static table
// (logsumexp.cpp:5-19), computed with the host libm exactly as the reference does.
const std::vector<double>& lseTable() {
  static std::vector<double> tab;
  static std::once_flag once;
  std::call_once(once, [] {
    const int n = ((int)(10 / .0001)) + 1;
    tab.resize(n);
    for (int i = 0; i < n; ++i) tab[i] = std::log(1. + std::exp(-(i * .0001)));
  });
  return tab;
}

bool isTransition(int x, int y) { return x != y && (x & 1) == (y & 1); }

constexpr int kFbLanes[4] = {8, 16, 16, 32}, kFbRowCells[4] = {16, 16, 32, 32};   // the four wavefront kernels

}  // namespace

// ---- the persistent handle ---------------------------------------------------------------------------
// What is per device (the log-sum-exp table, a stream), what is per database (the pairs, their envelope widths per
// guide mode, the result buffers) and what is per E-step (the scores) are uploaded / computed once each; an EM run
// calls dnas_fb_estep up to 100 times on the same handle.
struct dnas_fb {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double* dTab = nullptr;
  // database
  int64_t nPairs = 0;
  int P = -1;                    // pLen size the per-pair buffers are sized for
  int maxInLen = 0;
  std::vector<int64_t> inOff, outOff;                        // host copies of the sequence offsets (routing by length, statistics)
  int8_t *dIn = nullptr, *dOut = nullptr;
  int64_t *dInOff = nullptr, *dOutOff = nullptr, *dCiOff = nullptr, *dCoOff = nullptr;
  int32_t *dCi = nullptr, *dCo = nullptr;
  double *dCounts = nullptr, *dLL = nullptr, *dPartial = nullptr;
  unsigned long long* dLseOps = nullptr;
  // per guide mode (0: the envelope is maxDistance = P wide, 1: strict) and P: which kernel takes which pair.  onchip[q]: the
  // wavefront kernels, kFbLanes[q] lanes per pair for envelope rows of up to kFbRowCells[q] cells; the narrow ones (q = 0, 2) take
  // the pairs whose rows meet hi(ip) - lo(ip + W) < W (fwdback_onchip.hip), i.e. every alignment that runs down a diagonal
  struct Route { int P = -1; int maxInOnchip[4] = {0, 0, 0, 0}, maxSteps[4] = {1, 1, 1, 1}; std::vector<int64_t> onchip[4], streaming; std::vector<int64_t> cells; std::vector<int> width;
                 int64_t* dOnchip[4] = {nullptr, nullptr, nullptr, nullptr}; int64_t* dStreaming = nullptr; };
  Route route[2];
  // streaming kernel arenas
  double *dFwd = nullptr, *dRows = nullptr;
  size_t fwdBytes = 0, rowsBytes = 0;
  double* dScratch = nullptr;        // on-chip kernels: checkpoints and duplication lanes of the pair slots
  size_t scratchBytes = 0;
  int cus = 256;
  dnas_fb_stats stats{};
};

namespace {

void fbFreeDatabase(dnas_fb* h) {
  for (void* q : {(void*)h->dIn, (void*)h->dOut, (void*)h->dInOff, (void*)h->dOutOff, (void*)h->dCiOff, (void*)h->dCoOff, (void*)h->dCi,
                  (void*)h->dCo, (void*)h->dCounts, (void*)h->dLL, (void*)h->dPartial})
    if (q) (void)hipFree(q);
  h->dIn = h->dOut = nullptr; h->dInOff = h->dOutOff = h->dCiOff = h->dCoOff = nullptr; h->dCi = h->dCo = nullptr;
  h->dCounts = h->dLL = h->dPartial = nullptr;
  for (auto& r : h->route) {
    for (int64_t* q : r.dOnchip) if (q) (void)hipFree(q);
    if (r.dStreaming) (void)hipFree(r.dStreaming);
    r = dnas_fb::Route{};
  }
  h->nPairs = 0; h->P = -1;
}

}  // namespace

extern "C" int dnas_fb_create(int device_id, dnas_fb** out) {
  auto cleanup = [] {};
  if (!out) return dnas::fail(DNAS_E_INVALID, "dnas_fb_create: null argument");
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return dnas::fail(DNAS_E_DEVICE, "no HIP device available");
  if (device_id < 0 || device_id >= count) return dnas::fail(DNAS_E_INVALID, "device_id out of range");
  HIP_TRY(hipSetDevice(device_id));
  dnas_fb* h = new dnas_fb();
  h->device = device_id;
  auto cleanup2 = [&] { dnas_fb_destroy(h); };
#define cleanup cleanup2
  HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  HIP_TRY(hipEventCreate(&h->ev0));
  HIP_TRY(hipEventCreate(&h->ev1));
  const std::vector<double>& tab = lseTable();
  HIP_TRY(hipMalloc((void**)&h->dTab, tab.size() * sizeof(double)));
  HIP_TRY(hipMemcpy(h->dTab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc((void**)&h->dLseOps, sizeof(unsigned long long)));
  (void)hipDeviceGetAttribute(&h->cus, hipDeviceAttributeMultiprocessorCount, device_id);
#undef cleanup
  *out = h;
  return DNAS_OK;
}

extern "C" void dnas_fb_destroy(dnas_fb* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  fbFreeDatabase(h);
  for (void* q : {(void*)h->dTab, (void*)h->dFwd, (void*)h->dRows, (void*)h->dLseOps, (void*)h->dScratch})
    if (q) (void)hipFree(q);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

extern "C" int dnas_fb_load_pairs(dnas_fb* h, int64_t n_pairs, const int8_t* in_seqs, const int64_t* in_off, const int8_t* out_seqs,
                                  const int64_t* out_off, const int32_t* cm_in, const int64_t* cm_in_off, const int32_t* cm_out,
                                  const int64_t* cm_out_off) {
  auto cleanup = [] {};
  if (!h || n_pairs < 0) return dnas::fail(DNAS_E_INVALID, "dnas_fb_load_pairs: bad argument");
  if (n_pairs > 0 && (!in_seqs || !in_off || !out_seqs || !out_off || !cm_in || !cm_in_off || !cm_out || !cm_out_off))
    return dnas::fail(DNAS_E_INVALID, "dnas_fb_load_pairs: null argument");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->stream));
  fbFreeDatabase(h);
  if (n_pairs == 0) return DNAS_OK;
  // ---- validate (the kernels trust these): the offsets here, every base and guide column on the GPU once they are there
  int maxIn = 0;
  if (in_off[0] != 0 || out_off[0] != 0 || cm_in_off[0] != 0 || cm_out_off[0] != 0) return dnas::fail(DNAS_E_INVALID, "offset arrays must start at 0");
  for (int64_t i = 0; i < n_pairs; ++i) {
    const int64_t inLen = in_off[i + 1] - in_off[i], outLen = out_off[i + 1] - out_off[i];
    if (inLen < 0 || outLen < 0 || cm_in_off[i + 1] - cm_in_off[i] != inLen + 1 || cm_out_off[i + 1] - cm_out_off[i] != outLen + 1)
      return dnas::fail(DNAS_E_INVALID, "pair " + std::to_string(i) + ": inconsistent offsets");
    if (inLen > 30000 || outLen > 30000) return dnas::fail(DNAS_E_UNSUPPORTED, "pair " + std::to_string(i) + ": sequences longer than 30000");
    maxIn = std::max<int>(maxIn, (int)inLen);
  }
  auto cleanup2 = [&] { fbFreeDatabase(h); };
#define cleanup cleanup2
  const size_t nIn = (size_t)in_off[n_pairs], nOut = (size_t)out_off[n_pairs];
  const size_t nCi = (size_t)cm_in_off[n_pairs], nCo = (size_t)cm_out_off[n_pairs];
#define UPLOAD(dst, src, n, T)                                                     \
  HIP_TRY(hipMalloc((void**)&dst, std::max<size_t>((n), 1) * sizeof(T)));         \
  if (n) HIP_TRY(hipMemcpy(dst, src, (n) * sizeof(T), hipMemcpyHostToDevice));
  UPLOAD(h->dIn, in_seqs, nIn, int8_t) UPLOAD(h->dOut, out_seqs, nOut, int8_t)
  UPLOAD(h->dInOff, in_off, (size_t)n_pairs + 1, int64_t) UPLOAD(h->dOutOff, out_off, (size_t)n_pairs + 1, int64_t)
  UPLOAD(h->dCi, cm_in, nCi, int32_t) UPLOAD(h->dCo, cm_out, nCo, int32_t)
  UPLOAD(h->dCiOff, cm_in_off, (size_t)n_pairs + 1, int64_t) UPLOAD(h->dCoOff, cm_out_off, (size_t)n_pairs + 1, int64_t)
#undef UPLOAD
  HIP_TRY(hipMalloc((void**)&h->dLL, (size_t)n_pairs * sizeof(double)));
  {
    // every base in 0..3, every guide column array non-decreasing: one thread per pair (h->dLseOps doubles as the flag word)
    HIP_TRY(hipMemsetAsync(h->dLseOps, 0, sizeof(unsigned long long), h->stream));
    hipLaunchKernelGGL(fwdback_validate_kernel, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, h->stream, n_pairs, h->dIn, h->dInOff, h->dOut,
                       h->dOutOff, h->dCi, h->dCiOff, h->dCo, h->dCoOff, (unsigned*)h->dLseOps);
    HIP_TRY(hipGetLastError());
    unsigned flags = 0;
    HIP_TRY(hipMemcpyAsync(&flags, h->dLseOps, sizeof flags, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (flags & 1u) { cleanup(); return dnas::fail(DNAS_E_BAD_BASE, "bad base"); }
    if (flags & 2u) { cleanup(); return dnas::fail(DNAS_E_INVALID, "cm_in / cm_out must be non-decreasing"); }
  }
#undef cleanup
  h->nPairs = n_pairs;
  h->maxInLen = maxIn;
  h->inOff.assign(in_off, in_off + n_pairs + 1);
  h->outOff.assign(out_off, out_off + n_pairs + 1);
  return DNAS_OK;
}

extern "C" int dnas_fb_estep(dnas_fb* h, const dnas_mutator_params* p, int strict, double* out_counts, double* out_ll,
                             double* out_pair_ll) {
  auto cleanup = [] {};
  if (!h || !p || !out_counts || !out_ll) return dnas::fail(DNAS_E_INVALID, "dnas_fb_estep: null argument");
  const int P = p->n_len, nc = 21 + P;
  if (P < 0 || P > kFbMaxLen) return dnas::fail(DNAS_E_UNSUPPORTED, "pLen longer than 32 entries");
  for (int k = 0; k < nc; ++k) out_counts[k] = 0;
  *out_ll = 0;
  h->stats = dnas_fb_stats{};
  const int64_t n_pairs = h->nPairs;
  if (n_pairs == 0) return DNAS_OK;
  HIP_TRY(hipSetDevice(h->device));

  // ---- scores (MutatorScores, mutator.cpp:56-75)
  FbArgs a{};
  a.P = P;
  a.maxDistance = strict ? 0 : P;                            // fwdback.cpp:17
  a.delOpen = std::log(p->p_del_open);
  a.tanDup = std::log(p->p_tan_dup);
  a.noGap = std::log(1. - p->p_del_open - p->p_tan_dup);
  a.delExtend = std::log(p->p_del_extend);
  a.delEnd = std::log(1. - p->p_del_extend);
  const double nullScore = std::log(1. / 4.), pMatch = 1. - p->p_transition - p->p_transversion;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      a.sub[i * 4 + j] = (i == j ? std::log(pMatch) : (isTransition(i, j) ? std::log(p->p_transition) : std::log(p->p_transversion / 2))) - nullScore;
  for (int k = 0; k < P; ++k) a.len[k] = std::log(p->p_len[k]);

  // ---- which kernel takes which pair: the on-chip kernels serve envelope rows of up to 16 or 32 cells (16 or 32 lanes per
  // pair) and up to 8 duplication lengths; the census depends on the envelope half-width (P, or 0 with strict guides) and is kept
  dnas_fb::Route& rt = h->route[strict ? 1 : 0];
  if (rt.P != P) {
    for (int64_t*& q : rt.dOnchip) { if (q) (void)hipFree(q); q = nullptr; }
    if (rt.dStreaming) { (void)hipFree(rt.dStreaming); rt.dStreaming = nullptr; }
    for (auto& list : rt.onchip) list.clear();
    rt.streaming.clear();
    rt.cells.assign((size_t)n_pairs, 1);
    rt.width.assign((size_t)n_pairs, 1);
    const int Dm = a.maxDistance;
    const bool forceStreaming = getenv("DNAS_FB_STREAMING") != nullptr;
    // the on-chip kernels keep a pair's envelope bounds in LDS: a pair whose input is too long for that goes to the streaming
    // kernel like the pairs with wider envelope rows (never fail the call for it)
    int longest[4] = {0, 0, 0, 0};
    for (int q = 0; q < 4; ++q)
      while ((size_t)(kFbWave / kFbLanes[q]) * fbOnchipPairDoubles(kFbLanes[q], longest[q] + 64) * sizeof(double) <= kFbOnchipLdsLimit) longest[q] += 64;
    for (int& v : rt.maxInOnchip) v = 0;
    for (int& v : rt.maxSteps) v = 1;
    const bool noNarrow = getenv("DNAS_FB_NO_NARROW") != nullptr;       // (measurement: the round-3 routing, W = the row capacity)
    // the envelope census of every pair -- cells, widest row, wavefront steps, whether half the lanes do -- on the GPU
    // (fwdback_census_kernel: one thread per pair; on the host this loop took longer than the E-step itself)
    std::vector<int64_t> census((size_t)n_pairs * 3);
    {
      int64_t* dCensus = nullptr;
      HIP_TRY(hipMalloc((void**)&dCensus, census.size() * sizeof(int64_t)));
      hipLaunchKernelGGL(fwdback_census_kernel, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, h->stream, n_pairs, Dm, h->dInOff, h->dOutOff,
                         h->dCi, h->dCiOff, h->dCo, h->dCoOff, dCensus);
      hipError_t e = hipGetLastError();
      if (e == hipSuccess) e = hipMemcpyAsync(census.data(), dCensus, census.size() * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
      (void)hipFree(dCensus);
      HIP_TRY(e);
    }
    for (int64_t i = 0; i < n_pairs; ++i) {
      const int64_t inLen = h->inOff[i + 1] - h->inOff[i], outLen = h->outOff[i + 1] - h->outOff[i];
      const int w = (int)(census[3 * (size_t)i + 1] & 0xffff);
      const bool fits8 = (census[3 * (size_t)i + 1] >> 16) & 1, fits16 = (census[3 * (size_t)i + 1] >> 17) & 1;
      rt.cells[(size_t)i] = census[3 * (size_t)i];
      rt.width[(size_t)i] = w;
      int kind = w <= 16 ? 1 : (w <= 32 ? 3 : 4);            // the full-width kernel of the row capacity ...
      // ... or half the lanes, when every lane has left its row before its next one comes up (Forward: rows ip, ip + W;
      // Backward walks the same rows the other way: the same inequalities)
      if (kind < 4 && !noNarrow && (kind == 1 ? fits8 : fits16)) --kind;
      const bool chip = kind < 4 && P <= 8 && inLen <= longest[kind] && outLen < 32000 && !forceStreaming;
      (chip ? rt.onchip[kind] : rt.streaming).push_back(i);
      if (chip) {
        rt.maxInOnchip[kind] = std::max<int>(rt.maxInOnchip[kind], (int)inLen);
        rt.maxSteps[kind] = std::max<int>(rt.maxSteps[kind], (int)census[3 * (size_t)i + 2]);   // a = ip + op from lo(0) to inLen + hi(inLen)
      }
    }
    // the pairs of a wave walk in step: neighbours in the list should be of a length (longest first)
    // (a stable counting sort by input length: the lengths are at most 30 000, the lists 10^5 .. 10^6 long)
    for (auto& list : rt.onchip) {
      if (list.size() < 2) continue;
      int64_t longestIn = 0;
      for (int64_t x : list) longestIn = std::max(longestIn, h->inOff[x + 1] - h->inOff[x]);
      std::vector<int64_t> start((size_t)longestIn + 2, 0);
      for (int64_t x : list) ++start[(size_t)(longestIn - (h->inOff[x + 1] - h->inOff[x])) + 1];
      for (size_t k = 1; k < start.size(); ++k) start[k] += start[k - 1];
      std::vector<int64_t> sorted(list.size());
      for (int64_t x : list) sorted[(size_t)start[(size_t)(longestIn - (h->inOff[x + 1] - h->inOff[x]))]++] = x;
      list.swap(sorted);
    }
    auto put = [&](const std::vector<int64_t>& v, int64_t** d) -> hipError_t {
      hipError_t e = hipMalloc((void**)d, std::max<size_t>(v.size(), 1) * sizeof(int64_t));
      if (e == hipSuccess && !v.empty()) e = hipMemcpy(*d, v.data(), v.size() * sizeof(int64_t), hipMemcpyHostToDevice);
      return e;
    };
    for (int q = 0; q < 4; ++q) HIP_TRY(put(rt.onchip[q], &rt.dOnchip[q]));
    HIP_TRY(put(rt.streaming, &rt.dStreaming));
    rt.P = P;
  }
  if (h->P != P) {   // result buffers are sized by the number of counts
    if (h->dCounts) { (void)hipFree(h->dCounts); h->dCounts = nullptr; }
    if (h->dPartial) { (void)hipFree(h->dPartial); h->dPartial = nullptr; }
    HIP_TRY(hipMalloc((void**)&h->dCounts, (size_t)n_pairs * nc * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&h->dPartial, (size_t)256 * (nc + 1) * sizeof(double)));
    h->P = P;
  }
  HIP_TRY(hipMemsetAsync(h->dLseOps, 0, sizeof(unsigned long long), h->stream));
  HIP_TRY(hipEventRecord(h->ev0, h->stream));

  // ---- on-chip kernels: a wave per work-group (8 pairs of 8 lanes, 4 of 16, or 2 of 32), persistent over the list
  for (int w = 0; w < 4; ++w) {
    if (rt.onchip[w].empty()) continue;
    const int W = kFbLanes[w], RW = kFbRowCells[w], ppg = kFbWave / W;
    const size_t lds = (size_t)ppg * fbOnchipPairDoubles(W, rt.maxInOnchip[w]) * sizeof(double);   // <= kFbOnchipLdsLimit by the routing
    const int64_t nL = (int64_t)rt.onchip[w].size();
    // work-groups (= waves) a CU holds: 160 KB of LDS, and 16 waves of the kernel's 128 registers per lane
    const int perCu = std::max(1, std::min(DNAS_FB_WAVES_PER_CU, (int)((size_t)(160 * 1024) / lds)));
    // every wave keeps the Forward cells of the pairs it works on in HBM (2.7 MB for 256-nt pairs): the waves of a launch are
    // bounded by a scratch budget (16 GB, a quarter of what is free) when the inputs are long
    (void)RW;
    const size_t waveBytes = fbOnchipWaveDoubles(rt.maxSteps[w]) * sizeof(double);
    size_t freeB = 0, totalB = 0;
    HIP_TRY(hipMemGetInfo(&freeB, &totalB));
    const size_t budget = std::max<size_t>(h->scratchBytes, std::min<size_t>((size_t)16 << 30, (freeB + h->scratchBytes) / 4));
    const int64_t slotsMax = std::max<int64_t>(1, (int64_t)(budget / waveBytes));
    const unsigned grid = (unsigned)std::min<int64_t>(std::min<int64_t>((nL + ppg - 1) / ppg, (int64_t)h->cus * perCu), slotsMax);
    const size_t need = (size_t)grid * waveBytes;
    if (need > h->scratchBytes) {
      HIP_TRY(hipStreamSynchronize(h->stream));
      if (h->dScratch) { (void)hipFree(h->dScratch); h->dScratch = nullptr; h->scratchBytes = 0; }
      HIP_TRY(hipMalloc((void**)&h->dScratch, need));
      h->scratchBytes = need;
    }
    void (*const kernels[2][4])(FB_ONCHIP_ARGS) = {
        {fwdback_onchip8x16_kernel, fwdback_onchip16x16_kernel, fwdback_onchip16x32_kernel, fwdback_onchip32x32_kernel},
        {fwdback_onchip8x16p6_kernel, fwdback_onchip16x16p6_kernel, fwdback_onchip16x32p6_kernel, fwdback_onchip32x32p6_kernel}};
    auto kernel = kernels[P <= 6 ? 1 : 0][w];
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kFbWave), lds, h->stream, a, h->dIn, h->dInOff,
                       h->dOut, h->dOutOff, h->dCi, h->dCiOff, h->dCo, h->dCoOff, h->dTab, rt.dOnchip[w], nL, h->dCounts, h->dLL, rt.maxInOnchip[w],
                       h->dLseOps, h->dScratch, rt.maxSteps[w]);
    HIP_TRY(hipGetLastError());
  }
  // ---- streaming kernel for the rest: the interleaved Forward arena holds cellCap cells for each of B pairs
  if (!rt.streaming.empty()) {
    const size_t W = (size_t)P + 2;
    size_t freeB = 0, totalB = 0;
    HIP_TRY(hipMemGetInfo(&freeB, &totalB));
    const size_t budget = std::max<size_t>(h->fwdBytes + h->rowsBytes, std::min<size_t>((size_t)((double)freeB * 0.5), (size_t)16 << 30));
    const int64_t nS = (int64_t)rt.streaming.size();
    int64_t start = 0;
    while (start < nS) {
      int64_t end = start, cap = 0;
      int rowCap = 1;
      while (end < nS && end - start < (1 << 16)) {
        const int64_t c2 = std::max(cap, rt.cells[(size_t)rt.streaming[(size_t)end]]);
        const int r2 = std::max(rowCap, rt.width[(size_t)rt.streaming[(size_t)end]]);
        const size_t need = ((size_t)c2 + 2 * (size_t)r2) * W * sizeof(double) * (size_t)(end - start + 1);
        if (need > budget && end > start) break;
        cap = c2; rowCap = r2; ++end;
      }
      const int nB = (int)(end - start);
      const size_t fwdNeed = (size_t)cap * W * sizeof(double) * nB, rowNeed = 2 * (size_t)rowCap * W * sizeof(double) * nB;
      if (fwdNeed > h->fwdBytes) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (h->dFwd) { (void)hipFree(h->dFwd); h->dFwd = nullptr; h->fwdBytes = 0; }
        HIP_TRY(hipMalloc((void**)&h->dFwd, fwdNeed));
        h->fwdBytes = fwdNeed;
      }
      if (rowNeed > h->rowsBytes) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (h->dRows) { (void)hipFree(h->dRows); h->dRows = nullptr; h->rowsBytes = 0; }
        HIP_TRY(hipMalloc((void**)&h->dRows, rowNeed));
        h->rowsBytes = rowNeed;
      }
      a.rowCap = rowCap;
      hipLaunchKernelGGL(fwdback_estep_kernel, dim3((nB + kFbThreads - 1) / kFbThreads), dim3(kFbThreads), 0, h->stream, a, h->dIn, h->dInOff,
                         h->dOut, h->dOutOff, h->dCi, h->dCiOff, h->dCo, h->dCoOff, h->dTab, h->dFwd, h->dRows, h->dCounts, h->dLL, start, nB, cap,
                         (const int64_t*)rt.dStreaming);
      HIP_TRY(hipGetLastError());
      start = end;     // (the next batch reuses the arenas: same stream, in order)
    }
  }
  HIP_TRY(hipEventRecord(h->ev1, h->stream));

  // ---- reduction over pairs: fixed-shape tree, block partials added in order on the host
  const int nBlocks = (int)std::min<int64_t>(256, (n_pairs + 255) / 256);
  hipLaunchKernelGGL(fwdback_reduce_kernel, dim3(nBlocks), dim3(256), 0, h->stream, h->dCounts, h->dLL, n_pairs, nc, h->dPartial);
  HIP_TRY(hipGetLastError());
  std::vector<double> partial((size_t)nBlocks * (nc + 1));
  HIP_TRY(hipMemcpyAsync(partial.data(), h->dPartial, partial.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (out_pair_ll) HIP_TRY(hipMemcpyAsync(out_pair_ll, h->dLL, (size_t)n_pairs * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  unsigned long long ops = 0;
  HIP_TRY(hipMemcpyAsync(&ops, h->dLseOps, sizeof ops, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (int b = 0; b < nBlocks; ++b) {
    for (int k = 0; k < nc; ++k) out_counts[k] += partial[(size_t)b * (nc + 1) + k];
    *out_ll += partial[(size_t)b * (nc + 1) + nc];
  }
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  h->stats.kernel_ms = ms;
  h->stats.pairs_onchip = (int64_t)(rt.onchip[0].size() + rt.onchip[1].size() + rt.onchip[2].size() + rt.onchip[3].size());
  h->stats.pairs_narrow = (int64_t)(rt.onchip[0].size() + rt.onchip[2].size());
  h->stats.pairs_streaming = (int64_t)rt.streaming.size();
  h->stats.lse_ops = (int64_t)ops;
  int64_t ntOut = 0;
  for (int64_t i = 0; i < n_pairs; ++i) ntOut += h->outOff[i + 1] - h->outOff[i];
  h->stats.out_nt = ntOut;
  return DNAS_OK;
}

extern "C" int dnas_fb_last_stats(const dnas_fb* h, dnas_fb_stats* out) {
  if (!h || !out) return dnas::fail(DNAS_E_INVALID, "null argument");
  *out = h->stats;
  return DNAS_OK;
}

// expectedCounts in one call (host pointers in, counts out): a handle for the length of the call
extern "C" int dnas_fwdback_estep(const dnas_mutator_params* p, int strict, int64_t n_pairs, const int8_t* in_seqs,
                                  const int64_t* in_off, const int8_t* out_seqs, const int64_t* out_off,
                                  const int32_t* cm_in, const int64_t* cm_in_off, const int32_t* cm_out,
                                  const int64_t* cm_out_off, int device_id, double* out_counts, double* out_ll,
                                  double* out_pair_ll) {
  if (!p || n_pairs < 0 || !out_counts || !out_ll) return dnas::fail(DNAS_E_INVALID, "dnas_fwdback_estep: null argument");
  dnas_fb* h = nullptr;
  int rc = dnas_fb_create(device_id, &h);
  if (rc == DNAS_OK) rc = dnas_fb_load_pairs(h, n_pairs, in_seqs, in_off, out_seqs, out_off, cm_in, cm_in_off, cm_out, cm_out_off);
  if (rc == DNAS_OK) rc = dnas_fb_estep(h, p, strict, out_counts, out_ll, out_pair_ll);
  dnas_fb_destroy(h);
  return rc;
}

// ---- Baum-Welch driver (host): the EM loop around the GPU E-step -------------------------------
namespace {

double nMatch(const double* c) { return c[5] + c[10] + c[15] + c[20]; }
double nTransition(const double* c) {
  double n = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) if (isTransition(i, j)) n += c[5 + i * 4 + j];
  return n;
}
double nTransversion(const double* c) {
  double n = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) if (i != j && !isTransition(i, j)) n += c[5 + i * 4 + j];
  return n;
}
double logBetaPdfCounts(double prob, double yes, double no) {   // logsumexp.cpp:59-61,67-69
  const double al = yes + 1, be = no + 1;
  return std::lgamma(al + be) - std::lgamma(al) - std::lgamma(be) + (al - 1) * std::log(prob) + (be - 1) * std::log(1 - prob);
}
double logDirichletPdfCounts3(const double* prob, const double* count) {   // logsumexp.cpp:62-66,70-76
  double alpha[3];
  for (int n = 0; n < 3; ++n) alpha[n] = count[n] + 1;
  double ld = std::lgamma(0. + alpha[0] + alpha[1] + alpha[2]);
  for (int n = 0; n < 3; ++n) ld += (alpha[n] - 1) * std::log(prob[n]) - std::lgamma(alpha[n]);
  return ld;
}
double logPrior(const double* prior, const dnas_mutator_params& p) {   // MutatorCounts::logPrior, mutator.cpp:204-214
  const double pGap[3] = {p.p_del_open, p.p_tan_dup, 1. - p.p_del_open - p.p_tan_dup};
  const double nGap[3] = {prior[0], prior[1], prior[2]};
  const double pSub[3] = {p.p_transition, p.p_transversion, 1. - p.p_transition - p.p_transversion};
  const double nSub[3] = {nTransition(prior), nTransversion(prior), nMatch(prior)};
  return logBetaPdfCounts(p.p_del_extend, prior[3], prior[4]) + logDirichletPdfCounts3(pGap, nGap) + logDirichletPdfCounts3(pSub, nSub);
}

}  // namespace

extern "C" int dnas_baum_welch(const dnas_mutator_params* init, int strict, int64_t n_pairs, const int8_t* in_seqs,
                               const int64_t* in_off, const int8_t* out_seqs, const int64_t* out_off, const int32_t* cm_in,
                               const int64_t* cm_in_off, const int32_t* cm_out, const int64_t* cm_out_off, int device_id,
                               dnas_mutator_params* out, int32_t* out_iterations) {
  if (!init || !out) return dnas::fail(DNAS_E_INVALID, "dnas_baum_welch: null argument");
  dnas_mutator_params cur = *init;
  const int P = cur.n_len, nc = 21 + P;
  // one handle for the whole EM run: the database and the log-sum-exp table go to the GPU once
  dnas_fb* h = nullptr;
  int rc0 = dnas_fb_create(device_id, &h);
  if (rc0 == DNAS_OK) rc0 = dnas_fb_load_pairs(h, n_pairs, in_seqs, in_off, out_seqs, out_off, cm_in, cm_in_off, cm_out, cm_out_off);
  if (rc0 != DNAS_OK) { dnas_fb_destroy(h); return rc0; }
  std::vector<double> counts(nc), prior(nc, 1.);          // prior.initLaplace(), dnastore.cpp:137-138
  double best = -INFINITY;
  int iter = 0;
  for (; iter < 100; ++iter) {                            // BaumWelchMaxIter, fwdback.cpp:8
    double ll = 0;
    const int rc = dnas_fb_estep(h, &cur, strict, counts.data(), &ll, nullptr);
    if (rc != DNAS_OK) { dnas_fb_destroy(h); return rc; }
    ll += logPrior(prior.data(), cur);
    if ((ll - best) / std::fabs(best) < .001) break;      // BaumWelchMinFracInc, fwdback.cpp:7,221
    best = ll;
    for (int k = 0; k < nc; ++k) counts[k] += prior[k];   // counts.mlParams(prior), mutator.cpp:198-202
    // MutatorCounts::mlParams (mutator.cpp:167-178): pLen back to uniform, local kept from init
    for (int k = 0; k < P; ++k) cur.p_len[k] = 1. / (double)P;
    cur.p_del_open = counts[0] / (counts[0] + counts[1] + counts[2]);
    cur.p_tan_dup = counts[1] / (counts[0] + counts[1] + counts[2]);
    cur.p_del_extend = counts[3] / (counts[3] + counts[4]);
    const double ni = nTransition(counts.data()), nv = nTransversion(counts.data()), nm = nMatch(counts.data());
    cur.p_transition = ni / (ni + nv + nm);
    cur.p_transversion = nv / (ni + nv + nm);
    cur.local = init->local;
  }
  dnas_fb_destroy(h);
  *out = cur;
  if (out_iterations) *out_iterations = iter;
  return DNAS_OK;
}
