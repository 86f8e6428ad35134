"""Experiment (profiles/EXPERIMENTS.md, round 4, "a quick look before a sweep"): writes a copy of csrc/viterbi_tiera.hip in which a
sweep from the DNAS_QUICK_FROM-th of a column on first reads ALL its rows' accumulators in one go (one LDS latency instead of one
per row), skips the rows in front of the first one that has something new -- all of them when nothing is new: a quiet sweep costs a
few hundred cycles instead of three thousand -- and sweeps on from there as before.

    python tools/quick_source.py <output file>
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s = open(os.path.join(ROOT, "dnastore_amd", "csrc", "viterbi_tiera.hip")).read()


def replace(old, new):
    global s
    assert s.count(old) == 1, old
    s = s.replace(old, new)


replace("      for (;;) {\n        asm volatile(\"\" ::: \"memory\");   // other waves write LDS between sweeps: reload everything\n",
        "      int sweepIdx = 0;\n      bool lastQuiet = false;\n      for (;;) {\n        asm volatile(\"\" ::: \"memory\");   // other waves write LDS between sweeps: reload everything\n")
replace("        static_for<0, K>([&](auto kc) {\n          constexpr int k = kc.value;\n          if constexpr (G_ > 1 && !kSplit && k % kPollStride == 0) { foldInbox(IntC<0>{}); loadInbox(IntC<0>{}); }\n          if constexpr (!rowLive(k)) return;\n",
        """        int firstNew = 0;
#ifndef DNAS_QUICK_FROM
#define DNAS_QUICK_FROM 2
#endif
#ifndef DNAS_QUICK_ALWAYS
#define DNAS_QUICK_ALWAYS 0      // 0: only behind a sweep of this wave that grew nothing
#endif
        if (kSplit && sweepIdx >= DNAS_QUICK_FROM && (DNAS_QUICK_ALWAYS || lastQuiet)) {
          bool isNew[K];
          constexpr int KH = (K + 1) / 2;
          static_for<0, 2>([&](auto hc) {          // two halves: the registers of half the rows' accumulators at a time
            constexpr int k0 = hc.value * KH, k1 = (k0 + KH < K) ? k0 + KH : K;
            double dq[KH], sq[KH];
            static_for<k0, k1>([&](auto kc) {
              constexpr int k = kc.value;
              dq[k - k0] = kNegInf; sq[k - k0] = kNegInf;
              if constexpr (rowLive(k)) {
                dq[k - k0] = ldsRead(DC_OWN(k));
                if constexpr (kRows[k].sIdx >= 0) sq[k - k0] = ldsRead(SC_OWN(k));
              }
            });
            __builtin_amdgcn_sched_barrier(0);
            static_for<k0, k1>([&](auto kc) {
              constexpr int k = kc.value;
              isNew[k] = false;
              if constexpr (rowLive(k)) {
                bool x = dq[k - k0] != Dv[k];
                if constexpr (kRows[k].sIdx >= 0) x = x | (sq[k - k0] > S[k]);
                isNew[k] = __any(x);
              }
            });
          });
          firstNew = K;
          static_for<0, K>([&](auto kc) { if (isNew[K - 1 - kc.value]) firstNew = K - 1 - kc.value; });
        }
        static_for<0, K>([&](auto kc) {
          constexpr int k = kc.value;
          if constexpr (G_ > 1 && !kSplit && k % kPollStride == 0) { foldInbox(IntC<0>{}); loadInbox(IntC<0>{}); }
          if constexpr (!rowLive(k)) return;
          if (k < firstNew) return;
""")
replace("        ++rounds;\n        if constexpr (G_ > 1 && kSplit) loadInbox(IntC<0>{});\n", "        ++rounds;\n        ++sweepIdx;\n        lastQuiet = !__any(changed);\n        if constexpr (G_ > 1 && kSplit) loadInbox(IntC<0>{});\n")
open(sys.argv[1], "w").write(s)
