// Host side of the forward-backward path: dnas_fwdback_estep (expectedCounts,
// reference src/fwdback.cpp:190-209) and dnas_baum_welch (baumWelchParams,
// fwdback.cpp:211-230 with MutatorCounts::mlParams / logPrior, mutator.cpp:167-214).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
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
                                                const double*, double*, double*, double*, double*, int64_t, int, int64_t);
extern "C" __global__ void fwdback_reduce_kernel(const double*, const double*, int64_t, int, double*);

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      cleanup();                                                                               \
      return dnas::fail(DNAS_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    }                                                                                          \
  } while (0)

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

}  // namespace

extern "C" int dnas_fwdback_estep(const dnas_mutator_params* p, int strict, int64_t n_pairs, const int8_t* in_seqs,
                                  const int64_t* in_off, const int8_t* out_seqs, const int64_t* out_off,
                                  const int32_t* cm_in, const int64_t* cm_in_off, const int32_t* cm_out,
                                  const int64_t* cm_out_off, int device_id, double* out_counts, double* out_ll,
                                  double* out_pair_ll) {
  auto cleanup = [] {};
  if (!p || n_pairs < 0 || !out_counts || !out_ll) return dnas::fail(DNAS_E_INVALID, "dnas_fwdback_estep: null argument");
  const int P = p->n_len, nc = 21 + P;
  if (P < 0 || P > kFbMaxLen) return dnas::fail(DNAS_E_UNSUPPORTED, "pLen longer than 32 entries");
  for (int k = 0; k < nc; ++k) out_counts[k] = 0;
  *out_ll = 0;
  if (n_pairs == 0) return DNAS_OK;
  if (!in_seqs || !in_off || !out_seqs || !out_off || !cm_in || !cm_in_off || !cm_out || !cm_out_off)
    return dnas::fail(DNAS_E_INVALID, "dnas_fwdback_estep: null argument");
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return dnas::fail(DNAS_E_DEVICE, "no HIP device available");
  if (device_id < 0 || device_id >= count) return dnas::fail(DNAS_E_INVALID, "device_id out of range");

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

  // ---- per-pair envelope sizes (cells, widest row); validates the inputs
  const int Dm = a.maxDistance;
  std::vector<int64_t> cells((size_t)n_pairs);
  std::vector<int> width((size_t)n_pairs);
  for (int64_t i = 0; i < n_pairs; ++i) {
    const int64_t inLen = in_off[i + 1] - in_off[i], outLen = out_off[i + 1] - out_off[i];
    if (inLen < 0 || outLen < 0 || cm_in_off[i + 1] - cm_in_off[i] != inLen + 1 || cm_out_off[i + 1] - cm_out_off[i] != outLen + 1)
      return dnas::fail(DNAS_E_INVALID, "pair " + std::to_string(i) + ": inconsistent offsets");
    const int32_t* ci = cm_in + cm_in_off[i];
    const int32_t* co = cm_out + cm_out_off[i];
    for (int64_t k = 0; k < inLen; ++k) if (in_seqs[in_off[i] + k] < 0 || in_seqs[in_off[i] + k] > 3) return dnas::fail(DNAS_E_BAD_BASE, "bad base");
    for (int64_t k = 0; k < outLen; ++k) if (out_seqs[out_off[i] + k] < 0 || out_seqs[out_off[i] + k] > 3) return dnas::fail(DNAS_E_BAD_BASE, "bad base");
    for (int64_t k = 0; k < inLen; ++k) if (ci[k + 1] < ci[k]) return dnas::fail(DNAS_E_INVALID, "cm_in must be non-decreasing");
    for (int64_t k = 0; k < outLen; ++k) if (co[k + 1] < co[k]) return dnas::fail(DNAS_E_INVALID, "cm_out must be non-decreasing");
    int64_t lo = 0, hi = -1, tot = 0;
    int w = 1;
    for (int64_t ip = 0; ip <= inLen; ++ip) {
      while (lo <= outLen && co[lo] < ci[ip] - Dm) ++lo;
      if (hi < lo - 1) hi = lo - 1;
      while (hi + 1 <= outLen && co[hi + 1] <= ci[ip] + Dm) ++hi;
      tot += hi - lo + 1;
      w = std::max<int>(w, (int)(hi - lo + 1));
    }
    cells[i] = std::max<int64_t>(tot, 1);
    width[i] = w;
  }

  HIP_TRY(hipSetDevice(device_id));
  const size_t W = (size_t)P + 2;
  int8_t *dIn = nullptr, *dOut = nullptr;
  int64_t *dInOff = nullptr, *dOutOff = nullptr, *dCiOff = nullptr, *dCoOff = nullptr;
  int32_t *dCi = nullptr, *dCo = nullptr;
  double *dTab = nullptr, *dFwd = nullptr, *dRows = nullptr, *dCounts = nullptr, *dLL = nullptr, *dPartial = nullptr;
  auto cleanup2 = [&] {
    for (void* q : {(void*)dIn, (void*)dOut, (void*)dInOff, (void*)dOutOff, (void*)dCiOff, (void*)dCoOff, (void*)dCi, (void*)dCo,
                    (void*)dTab, (void*)dFwd, (void*)dRows, (void*)dCounts, (void*)dLL, (void*)dPartial})
      if (q) (void)hipFree(q);
  };
#define cleanup cleanup2
  const size_t nIn = (size_t)in_off[n_pairs], nOut = (size_t)out_off[n_pairs];
  const size_t nCi = (size_t)cm_in_off[n_pairs], nCo = (size_t)cm_out_off[n_pairs];
  const std::vector<double>& tab = lseTable();
#define UPLOAD(dst, src, n, T)                                                     \
  HIP_TRY(hipMalloc((void**)&dst, std::max<size_t>((n), 1) * sizeof(T)));         \
  if (n) HIP_TRY(hipMemcpy(dst, src, (n) * sizeof(T), hipMemcpyHostToDevice));
  UPLOAD(dIn, in_seqs, nIn, int8_t) UPLOAD(dOut, out_seqs, nOut, int8_t)
  UPLOAD(dInOff, in_off, (size_t)n_pairs + 1, int64_t) UPLOAD(dOutOff, out_off, (size_t)n_pairs + 1, int64_t)
  UPLOAD(dCi, cm_in, nCi, int32_t) UPLOAD(dCo, cm_out, nCo, int32_t)
  UPLOAD(dCiOff, cm_in_off, (size_t)n_pairs + 1, int64_t) UPLOAD(dCoOff, cm_out_off, (size_t)n_pairs + 1, int64_t)
  UPLOAD(dTab, tab.data(), tab.size(), double)
#undef UPLOAD
  HIP_TRY(hipMalloc((void**)&dCounts, (size_t)n_pairs * nc * sizeof(double)));
  HIP_TRY(hipMalloc((void**)&dLL, (size_t)n_pairs * sizeof(double)));

  // ---- batches: the interleaved Forward arena holds cellCap cells for each of B pairs
  size_t freeB = 0, totalB = 0;
  HIP_TRY(hipMemGetInfo(&freeB, &totalB));
  const size_t budget = std::min<size_t>((size_t)((double)freeB * 0.5), (size_t)16 << 30);
  int64_t start = 0;
  size_t arenaBytes = 0, rowsBytes = 0;
  while (start < n_pairs) {
    // grow the batch while it fits the budget
    int64_t end = start, cap = 0;
    int rowCap = 1;
    while (end < n_pairs && end - start < (1 << 16)) {
      const int64_t c2 = std::max(cap, cells[end]);
      const int r2 = std::max(rowCap, width[end]);
      const size_t need = ((size_t)c2 + 2 * (size_t)r2) * W * sizeof(double) * (size_t)(end - start + 1);
      if (need > budget && end > start) break;
      cap = c2; rowCap = r2; ++end;
    }
    const int nB = (int)(end - start);
    const size_t fwdNeed = (size_t)cap * W * sizeof(double) * nB, rowNeed = 2 * (size_t)rowCap * W * sizeof(double) * nB;
    if (fwdNeed > arenaBytes) {
      if (dFwd) { (void)hipFree(dFwd); dFwd = nullptr; }
      HIP_TRY(hipMalloc((void**)&dFwd, fwdNeed));
      arenaBytes = fwdNeed;
    }
    if (rowNeed > rowsBytes) {
      if (dRows) { (void)hipFree(dRows); dRows = nullptr; }
      HIP_TRY(hipMalloc((void**)&dRows, rowNeed));
      rowsBytes = rowNeed;
    }
    a.rowCap = rowCap;
    hipLaunchKernelGGL(fwdback_estep_kernel, dim3((nB + kFbThreads - 1) / kFbThreads), dim3(kFbThreads), 0, 0, a, dIn, dInOff,
                       dOut, dOutOff, dCi, dCiOff, dCo, dCoOff, dTab, dFwd, dRows, dCounts, dLL, start, nB, cap);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    start = end;
  }

  // ---- reduction over pairs: fixed-shape tree, block partials added in order on the host
  const int nBlocks = (int)std::min<int64_t>(256, (n_pairs + 255) / 256);
  HIP_TRY(hipMalloc((void**)&dPartial, (size_t)nBlocks * (nc + 1) * sizeof(double)));
  hipLaunchKernelGGL(fwdback_reduce_kernel, dim3(nBlocks), dim3(256), 0, 0, dCounts, dLL, n_pairs, nc, dPartial);
  HIP_TRY(hipGetLastError());
  std::vector<double> partial((size_t)nBlocks * (nc + 1));
  HIP_TRY(hipMemcpy(partial.data(), dPartial, partial.size() * sizeof(double), hipMemcpyDeviceToHost));
  for (int b = 0; b < nBlocks; ++b) {
    for (int k = 0; k < nc; ++k) out_counts[k] += partial[(size_t)b * (nc + 1) + k];
    *out_ll += partial[(size_t)b * (nc + 1) + nc];
  }
  if (out_pair_ll) HIP_TRY(hipMemcpy(out_pair_ll, dLL, (size_t)n_pairs * sizeof(double), hipMemcpyDeviceToHost));
  cleanup();
#undef cleanup
  return DNAS_OK;
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
  std::vector<double> counts(nc), prior(nc, 1.);          // prior.initLaplace(), dnastore.cpp:137-138
  double best = -INFINITY;
  int iter = 0;
  for (; iter < 100; ++iter) {                            // BaumWelchMaxIter, fwdback.cpp:8
    double ll = 0;
    const int rc = dnas_fwdback_estep(&cur, strict, n_pairs, in_seqs, in_off, out_seqs, out_off, cm_in, cm_in_off, cm_out,
                                      cm_out_off, device_id, counts.data(), &ll, nullptr);
    if (rc != DNAS_OK) return rc;
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
  *out = cur;
  if (out_iterations) *out_iterations = iter;
  return DNAS_OK;
}
