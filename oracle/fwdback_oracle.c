/*
 * fwdback_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded, fp64 restatement of the reference's banded pair-HMM
 * forward-backward E-step and Baum-Welch driver (ihh/dnastore src/fwdback.cpp,
 * src/logsumexp.{h,cpp}, src/mutator.cpp).  Only tests/, smoke() and bench.py's
 * cpu_baseline leg may load it.
 *
 * Pinned (tests/test_oracle_golden.py) against the reference's own goldens: the three
 * `--error-counts` cases and the two `--fit-error` cases of reference Makefile:156-163
 * (data/dup*.counts*.json, data/{tiny,test}.params.json, printed at 6 significant digits)
 * and the fp64 log-likelihoods recorded in SURVEY.md section 8(c).
 *
 * An alignment pair arrives as: in/out base tokens (0..3) and, per sequence position, the
 * guide's cumulative match count at that position's alignment column
 * (GuideAlignmentEnvelope, alignpath.h:35-54, alignpath.cpp:237-265):
 *   cmIn[ip]  = cumulativeMatches[row1PosToCol[ip]],  ip = 0..inLen
 *   cmOut[op] = cumulativeMatches[row2PosToCol[op]],  op = 0..outLen
 * so that inRange(ip,op) == |cmIn[ip] - cmOut[op]| <= maxDistance.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- log_sum_exp with the reference's lookup table (logsumexp.h:19-74, logsumexp.cpp:5-19) */
#define LSE_MAX 10
#define LSE_PRECISION .0001
#define LSE_ENTRIES (((int)(LSE_MAX / LSE_PRECISION)) + 1)

static double *lseTable = NULL;

static void lse_init(void) {
  int n;
  if (lseTable) return;
  lseTable = (double *)malloc(sizeof(double) * LSE_ENTRIES);
  for (n = 0; n < LSE_ENTRIES; ++n) {
    const double x = n * LSE_PRECISION;
    lseTable[n] = log(1. + exp(-x));   /* log_sum_exp_unary_slow, logsumexp.cpp:44-46 */
  }
}

const double *orc_lse_table(int *entries) {
  lse_init();
  if (entries) *entries = LSE_ENTRIES;
  return lseTable;
}

static double lse_unary(double x) {   /* logsumexp.h:34-54 */
  int n;
  double dx, f0, f1, df;
  if (x >= LSE_MAX || isnan(x) || isinf(x)) return 0;
  if (x < 0) return -x;
  n = (int)(x / LSE_PRECISION);
  dx = x - (n * LSE_PRECISION);
  f0 = lseTable[n];
  f1 = lseTable[n + 1];
  df = f1 - f0;
  return f0 + df * (dx / LSE_PRECISION);
}

double orc_log_sum_exp(double a, double b) {   /* logsumexp.h:56-74 */
  double mx, diff;
  lse_init();
  if (a == b) { mx = a; diff = 0; }
  else if (a < b) { mx = b; diff = b - a; }
  else { mx = a; diff = a - b; }
  return mx + lse_unary(diff);
}
#define LSE(a, b) orc_log_sum_exp((a), (b))

typedef struct {
  double pDelOpen, pDelExtend, pTanDup, pTransition, pTransversion;
  int nLen, local;
  const double *pLen;
} orc_params;

typedef struct {   /* MutatorScores, mutator.cpp:56-75 */
  double delOpen, tanDup, noGap, delExtend, delEnd, sub[4][4], len[32];
} fb_scores;

static int is_transition(int x, int y) { return x != y && (x & 1) == (y & 1); }

static void make_scores(const orc_params *p, fb_scores *s) {
  int i, j;
  const double nullScore = log(1. / 4.);
  const double pMatch = 1. - p->pTransition - p->pTransversion;
  s->delOpen = log(p->pDelOpen);
  s->tanDup = log(p->pTanDup);
  s->noGap = log(1. - p->pDelOpen - p->pTanDup);
  s->delExtend = log(p->pDelExtend);
  s->delEnd = log(1. - p->pDelExtend);
  for (i = 0; i < 4; ++i)
    for (j = 0; j < 4; ++j)
      s->sub[i][j] = (i == j ? log(pMatch) : (is_transition(i, j) ? log(p->pTransition) : log(p->pTransversion / 2))) - nullScore;
  for (i = 0; i < p->nLen && i < 32; ++i) s->len[i] = log(p->pLen[i]);
}

/* counts layout (MutatorCounts, mutator.h:43-50):
 *   [0] nDelOpen [1] nTanDup [2] nNoGap [3] nDelExtend [4] nDelEnd [5..20] nSub[4][4] [21..21+P) nLen */
#define C_DELOPEN 0
#define C_TANDUP 1
#define C_NOGAP 2
#define C_DELEXT 3
#define C_DELEND 4
#define C_SUB 5
#define C_LEN 21

/*
 * FwdBackMatrix (fwdback.cpp:118-128) + counts() (fwdback.cpp:154-188) for one pair.
 * counts[21+P] is OVERWRITTEN with this pair's expected counts; returns the forward
 * log-likelihood (fwdback.h:87).  *backLL receives the backward score.
 */
double orc_fwdback_pair(const orc_params *p, int strict, const int8_t *inSeq, int inLen, const int8_t *outSeq,
                        int outLen, const int *cmIn, const int *cmOut, double *counts, double *backLL) {
  const int P = p->nLen, W = P + 2;
  const int maxDistance = strict ? 0 : P;             /* fwdback.cpp:17 */
  const size_t nCells = (size_t)(inLen + 1) * (size_t)(outLen + 1);
  double *F, *B;
  fb_scores sc;
  double ll;
  size_t c;
  int ip, op, k;
  const double NEG = -INFINITY;
  lse_init();
  make_scores(p, &sc);
  F = (double *)malloc(sizeof(double) * nCells * W);
  B = (double *)malloc(sizeof(double) * nCells * W);
  for (c = 0; c < nCells * W; ++c) F[c] = B[c] = NEG;
#define INR(i, o) (abs(cmIn[i] - cmOut[o]) <= maxDistance)   /* alignpath.h:48-53 */
#define CELLF(i, o) (F + ((size_t)(i) * (outLen + 1) + (size_t)(o)) * W)
#define CELLB(i, o) (B + ((size_t)(i) * (outLen + 1) + (size_t)(o)) * W)
#define S_ 0
#define D_ 1
#define T_(k) (2 + (k))
#define MDL(i) ((i) < P ? (i) : P)                            /* maxDupLenAt, fwdback.h:59 */
#define INB(i) inSeq[(i) - 1]                                 /* cellInBase, fwdback.h:61 */
#define OUTB(o) outSeq[(o) - 1]
#define SUBS(i, o) sc.sub[INB(i)][OUTB(o)]                    /* cellSubScore, fwdback.h:65-67 */
#define DUPS(i, o, k) sc.sub[inSeq[(i) - 1 - (k)]][OUTB(o)]   /* cellTanDupScore, fwdback.h:69-71 */

  /* ForwardMatrix (fwdback.cpp:43-78) */
  CELLF(0, 0)[S_] = 0;
  for (ip = 0; ip <= inLen; ++ip)
    for (op = 0; op <= outLen; ++op)
      if (INR(ip, op)) {
        double *cell = CELLF(ip, op);
        if (ip > 0 && op > 0) {
          if (INR(ip - 1, op - 1)) cell[S_] = CELLF(ip - 1, op - 1)[S_] + sc.noGap + SUBS(ip, op);
          if (INR(ip, op - 1)) {
            const double *ins = CELLF(ip, op - 1);
            for (k = 0; k < MDL(ip) - 1; ++k) cell[T_(k)] = ins[T_(k + 1)] + DUPS(ip, op, k + 1);
            cell[S_] = LSE(cell[S_], ins[T_(0)] + DUPS(ip, op, 0));
          }
        }
        if (ip > 0 && INR(ip - 1, op)) {
          const double *del = CELLF(ip - 1, op);
          cell[D_] = LSE(del[S_] + sc.delOpen, del[D_] + sc.delExtend);
        }
        cell[S_] = LSE(cell[S_], cell[D_] + sc.delEnd);
        for (k = 0; k < MDL(ip); ++k) cell[T_(k)] = LSE(cell[T_(k)], cell[S_] + sc.tanDup + sc.len[k]);
      }
  ll = CELLF(inLen, outLen)[S_];

  /* BackwardMatrix (fwdback.cpp:80-116) */
  CELLB(inLen, outLen)[S_] = 0;
  for (ip = inLen; ip >= 0; --ip)
    for (op = outLen; op >= 0; --op)
      if (INR(ip, op)) {
        double *cell = CELLB(ip, op);
        if (op < outLen) {
          if (ip < inLen && INR(ip + 1, op + 1)) cell[S_] = sc.noGap + SUBS(ip + 1, op + 1) + CELLB(ip + 1, op + 1)[S_];
          if (ip > 0 && INR(ip, op + 1)) {
            const double *ins = CELLB(ip, op + 1);
            for (k = 1; k < MDL(ip); ++k) cell[T_(k)] = DUPS(ip, op + 1, k) + ins[T_(k - 1)];
            cell[T_(0)] = DUPS(ip, op + 1, 0) + ins[S_];
          }
        }
        if (ip < inLen && INR(ip + 1, op)) {
          const double *del = CELLB(ip + 1, op);
          cell[S_] = LSE(cell[S_], sc.delOpen + del[D_]);
          cell[D_] = sc.delExtend + del[D_];
        }
        for (k = 0; k < MDL(ip); ++k) cell[S_] = LSE(cell[S_], cell[T_(k)] + sc.tanDup + sc.len[k]);
        cell[D_] = LSE(cell[D_], cell[S_] + sc.delEnd);
      }
  if (backLL) *backLL = CELLB(0, 0)[S_];

  /* counts (fwdback.cpp:154-188; posterior formulas fwdback.h:92-112).  Cells outside the
     envelope read as -inf (the const getCell returns the dummy cell, fwdback.h:51-55). */
  for (k = 0; k < 21 + P; ++k) counts[k] = 0;
  for (ip = 0; ip <= inLen; ++ip)
    for (op = 0; op <= outLen; ++op)
      if (INR(ip, op)) {
        const double *bc = CELLB(ip, op);
        if (ip > 0 && op > 0) {
          /* pS2S */
          const double cS = exp(CELLF(ip - 1, op - 1)[S_] + sc.noGap + SUBS(ip, op) + bc[S_] - ll);
          double c0;
          counts[C_NOGAP] += cS;
          counts[C_SUB + INB(ip) * 4 + OUTB(op)] += cS;
          for (k = 0; k < MDL(ip) - 1; ++k) {
            /* pT2T */
            const double ci = exp(CELLF(ip, op - 1)[T_(k + 1)] + DUPS(ip, op, k + 1) + bc[T_(k)] - ll);
            counts[C_SUB + inSeq[ip - 1 - (k + 1)] * 4 + OUTB(op)] += ci;
          }
          /* pT2S */
          c0 = exp(CELLF(ip, op - 1)[T_(0)] + DUPS(ip, op, 0) + bc[S_] - ll);
          counts[C_SUB + inSeq[ip - 1] * 4 + OUTB(op)] += c0;
        }
        if (ip > 0) {
          counts[C_DELOPEN] += exp(CELLF(ip - 1, op)[S_] + sc.delOpen + bc[D_] - ll);     /* pS2D */
          counts[C_DELEXT] += exp(CELLF(ip - 1, op)[D_] + sc.delExtend + bc[D_] - ll);    /* pD2D */
        }
        counts[C_DELEND] += exp(CELLF(ip, op)[D_] + sc.delEnd + bc[S_] - ll);              /* pD2S */
        for (k = 0; k < MDL(ip); ++k) {
          const double cT = exp(CELLF(ip, op)[S_] + sc.tanDup + sc.len[k] + bc[T_(k)] - ll);   /* pS2T */
          counts[C_TANDUP] += cT;
          counts[C_LEN + k] += cT;
        }
      }
  free(F);
  free(B);
  return ll;
}

/*
 * expectedCounts (fwdback.cpp:190-209): counts and log-likelihoods summed over the database
 * in order.  Pairs are concatenated: seqs[inOff[i]..inOff[i+1]) etc.; cm arrays have one more
 * entry per pair than the sequences (offsets cmInOff/cmOutOff).
 */
double orc_expected_counts(const orc_params *p, int strict, int nPairs, const int8_t *inSeqs, const int64_t *inOff,
                           const int8_t *outSeqs, const int64_t *outOff, const int *cmIn, const int64_t *cmInOff,
                           const int *cmOut, const int64_t *cmOutOff, double *counts, double *perPairLL) {
  const int nc = 21 + p->nLen;
  double *tmp = (double *)malloc(sizeof(double) * nc);
  double ll = 0;
  int i, k;
  for (k = 0; k < nc; ++k) counts[k] = 0;
  for (i = 0; i < nPairs; ++i) {
    const double l = orc_fwdback_pair(p, strict, inSeqs + inOff[i], (int)(inOff[i + 1] - inOff[i]), outSeqs + outOff[i],
                                      (int)(outOff[i + 1] - outOff[i]), cmIn + cmInOff[i], cmOut + cmOutOff[i], tmp, NULL);
    for (k = 0; k < nc; ++k) counts[k] += tmp[k];   /* MutatorCounts::operator+=, mutator.cpp:139-152 */
    ll += l;
    if (perPairLL) perPairLL[i] = l;
  }
  free(tmp);
  return ll;
}

/* ---- Baum-Welch (fwdback.cpp:211-230; mutator.cpp:167-220; logsumexp.cpp:59-72) */
static double n_match(const double *c) { return c[C_SUB + 0] + c[C_SUB + 5] + c[C_SUB + 10] + c[C_SUB + 15]; }
static double n_transition(const double *c) {
  double n = 0; int i, j;
  for (i = 0; i < 4; ++i) for (j = 0; j < 4; ++j) if (is_transition(i, j)) n += c[C_SUB + i * 4 + j];
  return n;
}
static double n_transversion(const double *c) {
  double n = 0; int i, j;
  for (i = 0; i < 4; ++i) for (j = 0; j < 4; ++j) if (i != j && !is_transition(i, j)) n += c[C_SUB + i * 4 + j];
  return n;
}
static double log_beta_pdf_counts(double prob, double yes, double no) {   /* logsumexp.cpp:59-61,67-69 */
  const double a = yes + 1, b = no + 1;
  return lgamma(a + b) - lgamma(a) - lgamma(b) + (a - 1) * log(prob) + (b - 1) * log(1 - prob);
}
static double log_dirichlet_pdf_counts3(const double *prob, const double *count) {   /* logsumexp.cpp:62-66,70-76 */
  double alpha[3], ld; int n;
  for (n = 0; n < 3; ++n) alpha[n] = count[n] + 1;
  ld = lgamma(0. + alpha[0] + alpha[1] + alpha[2]);
  for (n = 0; n < 3; ++n) ld += (alpha[n] - 1) * log(prob[n]) - lgamma(alpha[n]);
  return ld;
}
static double log_prior(const double *prior, const double *pr /* pDelOpen,pDelExtend,pTanDup,pTransition,pTransversion */) {
  /* MutatorCounts::logPrior, mutator.cpp:204-214 */
  const double pGap[3] = {pr[0], pr[2], 1. - pr[0] - pr[2]};
  const double nGap[3] = {prior[C_DELOPEN], prior[C_TANDUP], prior[C_NOGAP]};
  const double pSub[3] = {pr[3], pr[4], 1. - pr[3] - pr[4]};
  const double nSub[3] = {n_transition(prior), n_transversion(prior), n_match(prior)};
  return log_beta_pdf_counts(pr[1], prior[C_DELEXT], prior[C_DELEND]) + log_dirichlet_pdf_counts3(pGap, nGap) +
         log_dirichlet_pdf_counts3(pSub, nSub);
}

/*
 * baumWelchParams (fwdback.cpp:211-230) with the Laplace prior of dnastore.cpp:137-138.
 * `params5` holds pDelOpen, pDelExtend, pTanDup, pTransition, pTransversion and is updated in
 * place; pLen is reset to uniform by mlParams (mutator.cpp:169).  Returns the iteration count.
 */
int orc_baum_welch(double *params5, int P, int local, int strict, int nPairs, const int8_t *inSeqs, const int64_t *inOff,
                   const int8_t *outSeqs, const int64_t *outOff, const int *cmIn, const int64_t *cmInOff, const int *cmOut,
                   const int64_t *cmOutOff, double *pLenOut) {
  const int nc = 21 + P;
  double *counts = (double *)malloc(sizeof(double) * nc), *prior = (double *)malloc(sizeof(double) * nc);
  double *pLen = (double *)malloc(sizeof(double) * (P > 0 ? P : 1));
  double best = -INFINITY;
  int iter, k;
  for (k = 0; k < nc; ++k) prior[k] = 1;                 /* initLaplace, mutator.cpp:126-137 */
  for (k = 0; k < P; ++k) pLen[k] = pLenOut[k];
  for (iter = 0; iter < 100; ++iter) {                   /* BaumWelchMaxIter, fwdback.cpp:8 */
    orc_params p;
    double ll, ni, nv, nm;
    p.pDelOpen = params5[0]; p.pDelExtend = params5[1]; p.pTanDup = params5[2];
    p.pTransition = params5[3]; p.pTransversion = params5[4];
    p.nLen = P; p.local = local; p.pLen = pLen;
    ll = orc_expected_counts(&p, strict, nPairs, inSeqs, inOff, outSeqs, outOff, cmIn, cmInOff, cmOut, cmOutOff, counts, NULL);
    ll += log_prior(prior, params5);
    if ((ll - best) / fabs(best) < .001) break;          /* BaumWelchMinFracInc, fwdback.cpp:7,221 */
    best = ll;
    for (k = 0; k < nc; ++k) counts[k] += prior[k];      /* mlParams(prior), mutator.cpp:198-202 */
    /* MutatorCounts::mlParams, mutator.cpp:167-178 */
    for (k = 0; k < P; ++k) pLen[k] = 1. / (double)P;
    params5[0] = counts[C_DELOPEN] / (counts[C_DELOPEN] + counts[C_TANDUP] + counts[C_NOGAP]);
    params5[2] = counts[C_TANDUP] / (counts[C_DELOPEN] + counts[C_TANDUP] + counts[C_NOGAP]);
    params5[1] = counts[C_DELEXT] / (counts[C_DELEXT] + counts[C_DELEND]);
    ni = n_transition(counts); nv = n_transversion(counts); nm = n_match(counts);
    params5[3] = ni / (ni + nv + nm);
    params5[4] = nv / (ni + nv + nm);
  }
  for (k = 0; k < P; ++k) pLenOut[k] = pLen[k];
  free(counts); free(prior); free(pLen);
  return iter;
}
