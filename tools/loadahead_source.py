"""Experiment (profiles/EXPERIMENTS.md, round 4, second session): a copy of csrc/viterbi_tiera.hip in which a cluster member reports
the completion of its exchange offers at the START of the next sweep (behind the fold) instead of at its end, and issues the inbox
loads -DDNAS_AHEAD=n rows before the end of the sweep (default 3; 0: behind the last row as shipped), so that the next sweep's fold
does not wait for their round trip.

    python tools/loadahead_source.py <output file>       # then DNAS_TIERA_SRC=<output file>, DNAS_TIERA_DEFS=-DDNAS_AHEAD=n
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s = open(os.path.join(ROOT, "dnastore_amd", "csrc", "viterbi_tiera.hip")).read()

def sub(old, new):
    global s
    assert s.count(old) == 1, old[:60]
    s = s.replace(old, new)

sub("""        if constexpr (G_ > 1 && kSplit) foldInbox(IntC<0>{});      // what was loaded behind the last row of the sweep before
""", """        if constexpr (G_ > 1 && kSplit) foldInbox(IntC<0>{});      // what was loaded behind the last row of the sweep before
#ifndef DNAS_AHEAD
#define DNAS_AHEAD 3
#endif
        if constexpr (G_ > 1) {
          if (pendingBump) {      // the offers of the sweep before: completed -> reported
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (ln == 0) __hip_atomic_fetch_add(pendL, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            pendingBump = false;
          }
        }
""")
sub("""          if constexpr (!rowLive(k)) return;
          double sc = kNegInf;
          const double d = ldsRead(DC_OWN(k));
""", """          if constexpr (G_ > 1 && kSplit && DNAS_AHEAD > 0 && k == (K > DNAS_AHEAD ? K - DNAS_AHEAD : 0)) loadInbox(IntC<0>{});
          if constexpr (!rowLive(k)) return;
          double sc = kNegInf;
          const double d = ldsRead(DC_OWN(k));
""")
sub("""          if (pendingBump) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (ln == 0) __hip_atomic_fetch_add(pendL, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            pendingBump = false;
          }
          if (__any(sentX)) pendingBump = true;   // GE is bumped once these offers have completed: in the next sweep
          if constexpr (kSplit) loadInbox(IntC<0>{});
""", """          if (__any(sentX)) pendingBump = true;   // reported at the start of the next sweep
          if constexpr (kSplit && DNAS_AHEAD <= 0) loadInbox(IntC<0>{});
""")
open(sys.argv[1], "w").write(s)
print("wrote", sys.argv[1])
