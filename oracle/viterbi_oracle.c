/*
 * viterbi_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-threaded, fp64 CPU restatement of the reference's Viterbi
 * error decoder (ihh/dnastore src/viterbi.cpp, src/mutator.cpp, src/trans.cpp).
 * It exists only as the checker for the HIP path: tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product (dnastore_amd/) never
 * does.  Every function cites the reference file:line it follows.
 *
 * Parity pinning: this restatement is checked (tests/test_oracle_golden.py)
 * against the reference's own golden vectors for the path -- the 11 Viterbi
 * `make test` cases of reference Makefile:146-186 (decoded strings
 * data/hello.{exact,padded}.bits) and the fp64 log-likelihoods recorded in
 * SURVEY.md section 8(c).  The reference itself is unbuildable in this image
 * (src/logger.h:15 needs Boost, which is absent), so there is no oracle/_ref.
 *
 * The machine arrives flattened (per-state transition lists in file order,
 * raw left-context strings); the JSON reading is done by the caller
 * (oracle/oracle.py), independently of the product's C++ loader.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>

#define ORC_OK 0
#define ORC_ERR_CONTEXT 1     /* verifyContexts failed        (trans.cpp:484-496) */
#define ORC_ERR_NOT_DNA 2     /* non-ACGT output symbol       (viterbi.cpp:27-28) */
#define ORC_ERR_CYCLIC 3      /* null graph cyclic            (trans.cpp:631-632) */
#define ORC_ERR_BAD_BASE 4    /* non-ACGT read character      (fastseq.cpp:25-39) */
#define ORC_ERR_TRACEBACK 5   /* checkBest assertion          (viterbi.cpp:230-233) */
#define ORC_ERR_ALLOC 6
#define ORC_ERR_OUTCAP 7

typedef struct {
  double pDelOpen, pDelExtend, pTanDup, pTransition, pTransversion;
  int nLen;            /* P = pLen.size() = maxDupLen()  (mutator.h:30) */
  int local;
  const double *pLen;
} orc_params;

/* IncomingTransScore (viterbi.h:18-23) in CSR form */
typedef struct {
  int32_t src;
  double score;
  char in;
  int8_t base;
} orc_in_edge;

/* OutgoingTransScore (viterbi.h:25-28) */
typedef struct {
  int32_t dest;
  double score;
} orc_out_edge;

typedef struct {
  int nStates;
  int D;                      /* maxDupLen = min(maxLeftContext, P) (viterbi.cpp:63) */
  int P;
  int local;
  /* StateScores (viterbi.h:30-35) */
  int *ctxLen;                /* leftContext.size() after '*' stripping (viterbi.cpp:33-36) */
  int8_t *ctx;                /* [N][D]: ctx[k] = leftContext[size-1-k] (viterbi.h:105) */
  int *mdl;                   /* maxDupLenAt (viterbi.h:104) */
  int *einPtr, *ninPtr, *eoutPtr, *noutPtr;
  orc_in_edge *ein, *nin;
  orc_out_edge *eout, *nout;
  int *topo;                  /* decoderToposort order (trans.cpp:604-634) */
  /* MutatorScores (mutator.cpp:56-75) */
  double delOpen, tanDup, noGap, delExtend, delEnd;
  double sub[4][4];
  double *len;
  /* InputModel (viterbi.cpp:6-14) */
  char alph[64];
  double symLogP[128];
  int symIn[128];
  /* instrumentation */
  long long pops;             /* worklist pops of the last fill */
} orc_model;

static int char_to_base(char c) {
  /* kmer.h:40-44 charToBase; fastseq.cpp:9-15 tokenize: case-insensitive index in "ACGT" */
  switch (toupper((unsigned char)c)) {
    case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3;
  }
  return -1;
}

static int is_control(char c) { return c >= 'A' && c <= 'Z'; }       /* trans.cpp:224-226 */
static int is_relaxed(char c) { return c == '0' || c == '1'; }        /* trans.cpp:234-236 */
static int is_transition(int x, int y) { return x != y && (x & 1) == (y & 1); } /* kmer.h:85-87 */

void orc_model_free(orc_model *m) {
  if (!m) return;
  free(m->ctxLen); free(m->ctx); free(m->mdl);
  free(m->einPtr); free(m->ninPtr); free(m->eoutPtr); free(m->noutPtr);
  free(m->ein); free(m->nin); free(m->eout); free(m->nout);
  free(m->topo); free(m->len);
  free(m);
}

/*
 * Build everything decodeFastSeqs/ViterbiMatrix derive from (machine, params):
 * inputAlphabet + InputModel (viterbi.cpp:309-310, 6-14; trans.cpp:280-292),
 * MachineScores (viterbi.cpp:23-60), MutatorScores (mutator.cpp:56-75),
 * decoderToposort (trans.cpp:604-634).  The reference rebuilds these per read;
 * they depend only on (machine, params) so one build serves a batch.
 */
int orc_model_create(int nStates, const int *transPtr, const char *transIn,
                     const char *transOut, const int *transDest,
                     const int *lctxPtr, const char *lctxChars,
                     const int *rctxPtr, const char *rctxChars,
                     const orc_params *p, orc_model **out) {
  int N = nStates, s, t, i, j, k;
  orc_model *m = (orc_model *)calloc(1, sizeof(orc_model));
  if (!m) return ORC_ERR_ALLOC;
  *out = NULL;
  m->nStates = N; m->P = p->nLen; m->local = p->local;

  /* verifyContexts (trans.cpp:484-496) */
  for (s = 0; s < N; ++s)
    for (t = transPtr[s]; t < transPtr[s + 1]; ++t)
      if (transOut[t]) {
        int d = transDest[t];
        if (rctxPtr[s + 1] > rctxPtr[s] && transOut[t] != rctxChars[rctxPtr[s]]) { orc_model_free(m); return ORC_ERR_CONTEXT; }
        if (lctxPtr[d + 1] > lctxPtr[d] && transOut[t] != lctxChars[lctxPtr[d + 1] - 1]) { orc_model_free(m); return ORC_ERR_CONTEXT; }
      }
  /* output alphabet must be DNA (viterbi.cpp:27-28) */
  for (t = 0; t < transPtr[N]; ++t)
    if (transOut[t] && char_to_base(transOut[t]) < 0) { orc_model_free(m); return ORC_ERR_NOT_DNA; }

  /* inputAlphabet(Relaxed|Control|SEOF) (trans.cpp:280-292): a sorted set of chars */
  {
    int present[128]; int na = 0;
    memset(present, 0, sizeof present);
    for (t = 0; t < transPtr[N]; ++t) {
      char c = transIn[t];
      if (c && (c == '$' || c == '^' || is_control(c) || is_relaxed(c))) present[(int)c] = 1;
    }
    for (i = 0; i < 128; ++i) if (present[i]) m->alph[na++] = (char)i;
    m->alph[na] = 0;
    /* InputModel (viterbi.cpp:6-14) with symWeight 1, controlWeight 4^-(4P) (viterbi.cpp:310) */
    {
      double controlWeight = pow(4., -(double)(4 * p->nLen));
      double norm = 0, w[128];
      for (i = 0; i < na; ++i) { w[i] = is_control(m->alph[i]) ? controlWeight : 1.; norm += w[i]; }
      for (i = 0; i < 128; ++i) { m->symIn[i] = 0; m->symLogP[i] = 0; }
      for (i = 0; i < na; ++i) {
        double pr = w[i] / norm;
        m->symIn[(int)m->alph[i]] = 1;
        m->symLogP[(int)m->alph[i]] = log(pr);
      }
    }
  }

  /* maxDupLen = min(machine.maxLeftContext(), params.maxDupLen()) (viterbi.cpp:63);
     maxLeftContext counts the raw string incl. '*' (trans.cpp:246-251) */
  {
    int maxLC = 0;
    for (s = 0; s < N; ++s) if (lctxPtr[s + 1] - lctxPtr[s] > maxLC) maxLC = lctxPtr[s + 1] - lctxPtr[s];
    m->D = maxLC < p->nLen ? maxLC : p->nLen;
  }

  /* MutatorScores (mutator.cpp:56-75) */
  m->delOpen = log(p->pDelOpen);
  m->tanDup = log(p->pTanDup);
  m->noGap = log(1. - p->pDelOpen - p->pTanDup);
  m->delExtend = log(p->pDelExtend);
  m->delEnd = log(1. - p->pDelExtend);
  {
    const double nullScore = log(1. / 4.);
    const double pMatch = 1. - p->pTransition - p->pTransversion;
    for (i = 0; i < 4; ++i)
      for (j = 0; j < 4; ++j)
        m->sub[i][j] = (i == j ? log(pMatch)
                        : (is_transition(i, j) ? log(p->pTransition) : log(p->pTransversion / 2))) - nullScore;
  }
  m->len = (double *)malloc(sizeof(double) * (p->nLen > 0 ? p->nLen : 1));
  for (k = 0; k < p->nLen; ++k) m->len[k] = log(p->pLen[k]);

  /* MachineScores (viterbi.cpp:23-60) */
  m->ctxLen = (int *)calloc(N, sizeof(int));
  m->mdl = (int *)calloc(N, sizeof(int));
  m->ctx = (int8_t *)calloc((size_t)N * (m->D > 0 ? m->D : 1), 1);
  m->einPtr = (int *)calloc(N + 1, sizeof(int));
  m->ninPtr = (int *)calloc(N + 1, sizeof(int));
  m->eoutPtr = (int *)calloc(N + 1, sizeof(int));
  m->noutPtr = (int *)calloc(N + 1, sizeof(int));
  for (s = 0; s < N; ++s) {
    int n = 0;
    for (i = lctxPtr[s]; i < lctxPtr[s + 1]; ++i) if (lctxChars[i] != '*') ++n;
    m->ctxLen[s] = n;
    m->mdl[s] = m->D < n ? m->D : n;
    /* stripped context, then ctx[k] = stripped[size-1-k] */
    {
      int8_t tmp[64]; int q = 0;
      for (i = lctxPtr[s]; i < lctxPtr[s + 1]; ++i)
        if (lctxChars[i] != '*') { int b = char_to_base(lctxChars[i]); if (b < 0) { orc_model_free(m); return ORC_ERR_NOT_DNA; } if (q < 64) tmp[q++] = (int8_t)b; }
      for (k = 0; k < m->mdl[s]; ++k) m->ctx[(size_t)s * m->D + k] = tmp[q - 1 - k];
    }
  }
#define USABLE(t) (transIn[t] == 0 || transIn[t] == '$' || m->symIn[(int)transIn[t]])
  for (s = 0; s < N; ++s)
    for (t = transPtr[s]; t < transPtr[s + 1]; ++t)
      if (USABLE(t)) {
        if (transOut[t]) { m->einPtr[transDest[t] + 1]++; m->eoutPtr[s + 1]++; }
        else { m->ninPtr[transDest[t] + 1]++; m->noutPtr[s + 1]++; }
      }
  for (s = 0; s < N; ++s) {
    m->einPtr[s + 1] += m->einPtr[s]; m->ninPtr[s + 1] += m->ninPtr[s];
    m->eoutPtr[s + 1] += m->eoutPtr[s]; m->noutPtr[s + 1] += m->noutPtr[s];
  }
  m->ein = (orc_in_edge *)calloc(m->einPtr[N] + 1, sizeof(orc_in_edge));
  m->nin = (orc_in_edge *)calloc(m->ninPtr[N] + 1, sizeof(orc_in_edge));
  m->eout = (orc_out_edge *)calloc(m->eoutPtr[N] + 1, sizeof(orc_out_edge));
  m->nout = (orc_out_edge *)calloc(m->noutPtr[N] + 1, sizeof(orc_out_edge));
  {
    int *fe = (int *)calloc(N, sizeof(int)), *fn = (int *)calloc(N, sizeof(int));
    int *ge = (int *)calloc(N, sizeof(int)), *gn = (int *)calloc(N, sizeof(int));
    /* ascending src state, then transition order: this is the push_back order of
       viterbi.cpp:49-56 and therefore the traceback tie-break order */
    for (s = 0; s < N; ++s)
      for (t = transPtr[s]; t < transPtr[s + 1]; ++t)
        if (USABLE(t)) {
          int d = transDest[t];
          double sc = m->symIn[(int)transIn[t]] ? m->symLogP[(int)transIn[t]] : 0;   /* viterbi.cpp:41 */
          orc_in_edge ie; orc_out_edge oe;
          ie.src = s; ie.score = sc; ie.in = transIn[t]; ie.base = -1;
          oe.dest = d; oe.score = sc;
          if (!transOut[t]) {
            m->nin[m->ninPtr[d] + fn[d]++] = ie;
            m->nout[m->noutPtr[s] + gn[s]++] = oe;
          } else {
            ie.base = (int8_t)char_to_base(transOut[t]);
            m->ein[m->einPtr[d] + fe[d]++] = ie;
            m->eout[m->eoutPtr[s] + ge[s]++] = oe;
          }
        }
    free(fe); free(fn); free(ge); free(gn);
  }

  /* decoderToposort (trans.cpp:604-634): Kahn over non-emitting transitions whose
     input is empty or in the input alphabet; FIFO queue seeded in ascending order */
  {
    int *nParents = (int *)calloc(N, sizeof(int));
    int *chPtr = (int *)calloc(N + 1, sizeof(int));
    int *ch, *fill = (int *)calloc(N, sizeof(int));
    int *queue = (int *)malloc(sizeof(int) * (N + 1));
    int qh = 0, qt = 0, edges = 0, nL = 0;
    m->topo = (int *)malloc(sizeof(int) * (N + 1));
#define TOPO_EDGE(t) (transOut[t] == 0 && (transIn[t] == 0 || strchr(m->alph, transIn[t]) != NULL))
    for (s = 0; s < N; ++s)
      for (t = transPtr[s]; t < transPtr[s + 1]; ++t)
        if (TOPO_EDGE(t)) { nParents[transDest[t]]++; edges++; chPtr[s + 1]++; }
    for (s = 0; s < N; ++s) chPtr[s + 1] += chPtr[s];
    ch = (int *)malloc(sizeof(int) * (chPtr[N] + 1));
    for (s = 0; s < N; ++s)
      for (t = transPtr[s]; t < transPtr[s + 1]; ++t)
        if (TOPO_EDGE(t)) ch[chPtr[s] + fill[s]++] = transDest[t];
    for (s = 0; s < N; ++s) if (nParents[s] == 0) queue[qt++] = s;
    while (qh < qt) {
      int n = queue[qh++];
      m->topo[nL++] = n;
      for (i = chPtr[n]; i < chPtr[n + 1]; ++i) {
        --edges;
        if (--nParents[ch[i]] == 0) queue[qt++] = ch[i];
      }
    }
    free(nParents); free(chPtr); free(ch); free(fill); free(queue);
    if (edges > 0) { orc_model_free(m); return ORC_ERR_CYCLIC; }
  }
  *out = m;
  return ORC_OK;
}

int orc_model_D(const orc_model *m) { return m->D; }
int orc_model_nstates(const orc_model *m) { return m->nStates; }
long long orc_model_pops(const orc_model *m) { return m->pops; }
const char *orc_model_alphabet(const orc_model *m) { return m->alph; }
double orc_model_sym_logp(const orc_model *m, int c) { return m->symLogP[c & 127]; }
void orc_model_edge_counts(const orc_model *m, int *nEmit, int *nNull) { *nEmit = m->einPtr[m->nStates]; *nNull = m->ninPtr[m->nStates]; }
void orc_model_scores(const orc_model *m, double *out /* 5+16+P */) {
  int i, j, k = 0;
  out[k++] = m->delOpen; out[k++] = m->tanDup; out[k++] = m->noGap; out[k++] = m->delExtend; out[k++] = m->delEnd;
  for (i = 0; i < 4; ++i) for (j = 0; j < 4; ++j) out[k++] = m->sub[i][j];
  for (i = 0; i < m->P; ++i) out[k++] = m->len[i];
}

static inline double dmax(double a, double b) { return a < b ? b : a; }   /* std::max */

/* cellIndex (viterbi.h:65-67): [pos][state][lane], lanes = S, D, T1..TD */
#define CELL(st, ps, lane) cell[(size_t)lanes * ((size_t)(ps) * N + (size_t)(st)) + (size_t)(lane)]
#define S_(st, ps) CELL(st, ps, 0)
#define D_(st, ps) CELL(st, ps, 1)
#define T_(st, ps, k) CELL(st, ps, 2 + (k))

/*
 * ViterbiMatrix ctor, lattice fill (viterbi.cpp:62-176).  `cell` must hold
 * (D+2)*N*(L+1) doubles (the reference over-allocates with P+2, viterbi.h:48-50,
 * but indexes with D+2).
 */
static void orc_fill(orc_model *m, const int8_t *seq, int L, double *cell) {
  const int N = m->nStates, D = m->D, lanes = D + 2;
  const double NEG = -INFINITY;
  size_t nc = (size_t)lanes * N * ((size_t)L + 1), ci;
  int pos, s, k, ti, e;
  int *stack = (int *)malloc(sizeof(int) * (size_t)(N + 1));
  unsigned char *onStack = (unsigned char *)malloc((size_t)N + 1);
  for (ci = 0; ci < nc; ++ci) cell[ci] = NEG;
  if (m->local) for (s = 0; s < N; ++s) S_(s, 0) = 0;      /* viterbi.cpp:75-79 */
  else S_(0, 0) = 0;
  m->pops = 0;

  for (pos = 0; pos <= L; ++pos) {
    const int x = pos > 0 ? seq[pos - 1] : 0;
    /* sweep 1 in topological order (viterbi.cpp:88-108) */
    for (ti = 0; ti < N; ++ti) {
      const int st = m->topo[ti];
      const int mdl = m->mdl[st];
      const int8_t *ctx = m->ctx + (size_t)st * D;
      if (pos > 0)
        for (e = m->einPtr[st]; e < m->einPtr[st + 1]; ++e) {
          const orc_in_edge *its = &m->ein[e];
          S_(st, pos) = dmax(S_(st, pos), S_(its->src, pos - 1) + its->score + m->noGap + m->sub[its->base][x]);
        }
      for (e = m->ninPtr[st]; e < m->ninPtr[st + 1]; ++e) {
        const orc_in_edge *its = &m->nin[e];
        S_(st, pos) = dmax(S_(st, pos), S_(its->src, pos) + its->score);
      }
      if (mdl > 0 && pos > 0) {
        S_(st, pos) = dmax(S_(st, pos), T_(st, pos - 1, 0) + m->sub[ctx[0]][x]);
        for (k = 0; k < mdl - 1; ++k)
          T_(st, pos, k) = T_(st, pos - 1, k + 1) + m->sub[ctx[k + 1]][x];
      }
    }
    /* sweep 2: LIFO worklist seeded with the topological order (viterbi.cpp:110-159) */
    {
      int sp = N;
      memcpy(stack, m->topo, sizeof(int) * (size_t)N);
      memset(onStack, 1, (size_t)N);
      while (sp > 0) {
        const int st = stack[--sp];
        double dsrc, ssrc;
        onStack[st] = 0;
        m->pops++;
        dsrc = D_(st, pos);
        ssrc = dmax(S_(st, pos), dsrc + m->delEnd);
        S_(st, pos) = ssrc;
        for (e = m->eoutPtr[st]; e < m->eoutPtr[st + 1]; ++e) {
          const orc_out_edge *ots = &m->eout[e];
          const double dsc = dmax(dsrc + m->delExtend, ssrc + m->delOpen) + ots->score;
          double *ddest = &D_(ots->dest, pos);
          if (dsc > *ddest) {
            *ddest = dsc;
            if (!onStack[ots->dest]) { stack[sp++] = ots->dest; onStack[ots->dest] = 1; }
          }
        }
        for (e = m->noutPtr[st]; e < m->noutPtr[st + 1]; ++e) {
          const orc_out_edge *ots = &m->nout[e];
          int push = 0;
          const double dsc = dsrc + ots->score;
          double *ddest = &D_(ots->dest, pos);
          double ssc, *sdest;
          if (dsc > *ddest) { *ddest = dsc; push = 1; }
          ssc = ssrc + ots->score;
          sdest = &S_(ots->dest, pos);
          if (ssc > *sdest) { *sdest = ssc; push = 1; }
          if (push && !onStack[ots->dest]) { stack[sp++] = ots->dest; onStack[ots->dest] = 1; }
        }
      }
    }
    /* sweep 3: open duplications (viterbi.cpp:161-168) */
    if (pos > 0)
      for (s = 0; s < N; ++s) {
        const int mdl = m->mdl[s];
        for (k = 0; k < mdl; ++k)
          T_(s, pos, k) = dmax(T_(s, pos, k), S_(s, pos) + m->tanDup + m->len[k]);
      }
  }
  /* local mode: best end state (viterbi.cpp:171-173) */
  if (m->local)
    for (s = 0; s < N; ++s)
      S_(N - 1, L) = dmax(S_(N - 1, L), S_(s, L));
  free(stack); free(onStack);
}

/*
 * ViterbiMatrix::traceback (viterbi.cpp:195-304).  Returns the decoded symbol
 * string in out[0..*outLen).  Empty string + ORC_OK when loglike is -inf
 * (viterbi.cpp:198-201).  nSteps (optional) counts loop iterations.
 */
static int orc_traceback(const orc_model *m, const int8_t *seq, int L, const double *cell,
                         char *out, int outCap, int *outLen, int *nSteps) {
  const int N = m->nStates, D = m->D, lanes = D + 2;
  int state = N - 1, pos = L, mut = 0;
  int bestState = 0, bestPos = 0, bestMut = 0, found;
  double best;
  char bestIn;
  int n = 0, steps = 0, s, e;
  char *rev;
  *outLen = 0;
  if (nSteps) *nSteps = 0;
  if (!(S_(N - 1, L) > -INFINITY)) return ORC_OK;
  rev = (char *)malloc((size_t)outCap + 1);
  if (!rev) return ORC_ERR_ALLOC;

#define INIT_BEST() do { best = -INFINITY; found = 0; bestIn = 0; } while (0)
  /* updateBest (viterbi.cpp:217-228): cell + transScore, first strictly greater wins */
#define UPDATE_BEST(ss, pp, mm, transScore, insym) do { \
    const double sc_ = CELL(ss, pp, mm) + (transScore); \
    if (sc_ > best) { best = sc_; bestState = (ss); bestPos = (pp); bestMut = (mm); bestIn = (insym); found = 1; } } while (0)
  /* checkBest (viterbi.cpp:230-237) */
#define CHECK_BEST() do { \
    const double exp_ = CELL(state, pos, mut); \
    if (!(fabs((best - exp_) / (fabs(exp_) < 1e-6 ? 1 : exp_)) < 1e-6) || !found) { free(rev); return ORC_ERR_TRACEBACK; } \
    state = bestState; pos = bestPos; mut = bestMut; } while (0)

  INIT_BEST();
  if (m->local) { for (s = 0; s < N; ++s) UPDATE_BEST(s, L, 0, 0., 0); }
  else UPDATE_BEST(N - 1, L, 0, 0., 0);
  CHECK_BEST();

  while (pos >= 0 && state > 0) {
    const int mdl = m->mdl[state];
    const int8_t *ctx = m->ctx + (size_t)state * D;
    ++steps;
    INIT_BEST();
    if (mut == 0) {
      if (pos > 0)
        for (e = m->einPtr[state]; e < m->einPtr[state + 1]; ++e) {
          const orc_in_edge *its = &m->ein[e];
          UPDATE_BEST(its->src, pos - 1, 0, its->score + m->noGap + m->sub[its->base][seq[pos - 1]], its->in);
        }
      for (e = m->ninPtr[state]; e < m->ninPtr[state + 1]; ++e) {
        const orc_in_edge *its = &m->nin[e];
        UPDATE_BEST(its->src, pos, 0, its->score, its->in);
      }
      UPDATE_BEST(state, pos, 1, m->delEnd, 0);
      if (mdl > 0 && pos > 0)
        UPDATE_BEST(state, pos - 1, 2, m->sub[ctx[0]][seq[pos - 1]], 0);
      if (pos == 0 && m->local)
        UPDATE_BEST(0, 0, 0, 0., 0);
    } else if (mut == 1) {
      for (e = m->einPtr[state]; e < m->einPtr[state + 1]; ++e) {
        const orc_in_edge *its = &m->ein[e];
        UPDATE_BEST(its->src, pos, 1, its->score + m->delExtend, its->in);
        UPDATE_BEST(its->src, pos, 0, its->score + m->delOpen, its->in);
      }
      for (e = m->ninPtr[state]; e < m->ninPtr[state + 1]; ++e) {
        const orc_in_edge *its = &m->nin[e];
        UPDATE_BEST(its->src, pos, 1, its->score, its->in);
      }
    } else {
      const int k = mut - 2;
      if (k < mdl - 1)
        UPDATE_BEST(state, pos - 1, 2 + k + 1, m->sub[ctx[k + 1]][seq[pos - 1]], 0);
      UPDATE_BEST(state, pos, 0, m->tanDup + m->len[k], 0);
    }
    CHECK_BEST();
    if (bestIn) {
      if (n >= outCap) { free(rev); return ORC_ERR_OUTCAP; }
      rev[n++] = bestIn;          /* trace.push_front (viterbi.cpp:299-300) */
    }
  }
  for (s = 0; s < n; ++s) out[s] = rev[n - 1 - s];
  *outLen = n;
  if (nSteps) *nSteps = steps;
  free(rev);
  return ORC_OK;
}

/*
 * One iteration of decodeFastSeqs' loop (viterbi.cpp:312-318) for one read:
 * tokenise (fastseq.cpp:25-39), fill, traceback, loglike (viterbi.h:102).
 * latticeOut (optional): the reference-layout lattice, (D+2)*N*(L+1) doubles.
 */
int orc_viterbi_read(orc_model *m, const char *read, int L,
                     char *outSym, int outCap, int *outLen, double *outLoglike,
                     double *latticeOut, int *nSteps) {
  const int N = m->nStates, lanes = m->D + 2;
  int8_t *seq = (int8_t *)malloc((size_t)L + 1);
  double *cell;
  int i, rc;
  *outLen = 0;
  for (i = 0; i < L; ++i) {
    int b = char_to_base(read[i]);
    if (b < 0) { free(seq); return ORC_ERR_BAD_BASE; }
    seq[i] = (int8_t)b;
  }
  cell = latticeOut ? latticeOut : (double *)malloc(sizeof(double) * (size_t)lanes * N * ((size_t)L + 1));
  if (!cell) { free(seq); return ORC_ERR_ALLOC; }
  orc_fill(m, seq, L, cell);
  *outLoglike = S_(N - 1, L);
  rc = orc_traceback(m, seq, L, cell, outSym, outCap, outLen, nSteps);
  if (!latticeOut) free(cell);
  free(seq);
  return rc;
}
