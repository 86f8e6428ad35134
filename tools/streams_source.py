"""Experiment (profiles/EXPERIMENTS.md, round 4, "row streams"): a copy of csrc/viterbi_tiera.hip whose sweep walks DNAS_STREAMS
(default 2) far-apart rows at a time -- rows k, k + K/STREAMS, ... : their accumulators are read together, then the rows are
evaluated one after the other -- so that one wave has several LDS round trips in flight (a wave's row visit is a chain of
latencies; with two waves per SIMD nothing else hides it).  No planner change: the rows of a stream keep their order.

    python tools/streams_source.py <output file>
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s = open(os.path.join(ROOT, "dnastore_amd", "csrc", "viterbi_tiera.hip")).read()

old = """        static_for<0, K>([&](auto kc) {
          constexpr int k = kc.value;
          if constexpr (G_ > 1 && !kSplit && k % kPollStride == 0) { foldInbox(IntC<0>{}); loadInbox(IntC<0>{}); }
          if constexpr (!rowLive(k)) return;
          double sc = kNegInf;
          const double d = ldsRead(DC_OWN(k));
          if constexpr (kRows[k].sIdx >= 0) sc = ldsRead(SC_OWN(k));
          rowEval(kc, d, sc);
        });
"""
new = """#ifndef DNAS_STREAMS
#define DNAS_STREAMS 2
#endif
        static_assert(kSplit, "row streams: inbox polls inside the sweep are not supported");
        constexpr int NST = DNAS_STREAMS, KS = (K + NST - 1) / NST;
        static_for<0, KS>([&](auto jc) {
          double dq[NST], sq[NST];
          static_for<0, NST>([&](auto sc_) {
            constexpr int k = sc_.value * KS + jc.value;
            dq[sc_.value] = kNegInf; sq[sc_.value] = kNegInf;
            if constexpr (k < K) {
              if constexpr (rowLive(k)) {
                dq[sc_.value] = ldsRead(DC_OWN(k));
                if constexpr (kRows[k].sIdx >= 0) sq[sc_.value] = ldsRead(SC_OWN(k));
              }
            }
          });
          static_for<0, NST>([&](auto sc_) {
            constexpr int k = sc_.value * KS + jc.value;
            if constexpr (k < K) {
              if constexpr (rowLive(k)) rowEval(IntC<k>{}, dq[sc_.value], sq[sc_.value]);
            }
          });
        });
"""
assert s.count(old) == 1
s = s.replace(old, new)
open(sys.argv[1], "w").write(s)
