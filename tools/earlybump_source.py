"""Experiment (profiles/EXPERIMENTS.md, round 4, "the epoch bumped during a sweep"): a copy of csrc/viterbi_tiera.hip in which a wave
that has grown a cell bumps the work-group's epoch already behind rows K/4, K/2 and 3K/4 of its sweep (once per sweep, besides the bump
at the end), so that idle waves wake before the offering wave has finished its sweep.

    python tools/earlybump_source.py <output file>
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s = open(os.path.join(ROOT, "dnastore_amd", "csrc", "viterbi_tiera.hip")).read()


def replace(old, new):
    global s
    assert s.count(old) == 1, old
    s = s.replace(old, new)


replace("          double sc = kNegInf;\n          const double d = ldsRead(DC_OWN(k));\n          if constexpr (kRows[k].sIdx >= 0) sc = ldsRead(SC_OWN(k));\n          rowEval(kc, d, sc);\n",
        """          double sc = kNegInf;
          const double d = ldsRead(DC_OWN(k));
          if constexpr (kRows[k].sIdx >= 0) sc = ldsRead(SC_OWN(k));
          rowEval(kc, d, sc);
#ifndef DNAS_BUMP_EVERY
#define DNAS_BUMP_EVERY ((K + 3) / 4)
#endif
          if constexpr ((k + 1) % DNAS_BUMP_EVERY == 0 && k + 1 < K) {
            if (!bumpedEarly && __any(changed)) {
              if (ln == 0) __hip_atomic_fetch_add(epochL, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
              bumpedEarly = true;
            }
          }
""")
replace("        int changed = 0, sentX = 0;\n", "        int changed = 0, sentX = 0;\n        bool bumpedEarly = false;\n")
open(sys.argv[1], "w").write(s)
