"""Writes a copy of csrc/viterbi_tiera.hip with cycle stamps INSIDE a cluster's sweep (tier C): built with -DDNAS_STAMP
-DDNAS_SSPLIT=n, wave 0 of work-group 0 adds the cycles from the start of every sweep to split point n into the stamp word that
otherwise holds the cluster vote's time (tools/tierc_probe.py prints it as "wave 0 in the cluster vote").
Points: 1 inbox folded (rows start), 2 rows done, 3 inbox loads issued, 4 exchange bump / import done, 5 end of the sweep body.

    python tools/csweep_source.py <output file>
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s = open(os.path.join(ROOT, "dnastore_amd", "csrc", "viterbi_tiera.hip")).read()


def replace(old, new):
    global s
    assert s.count(old) == 1, old
    s = s.replace(old, new)


def stamp(n):
    return "#if DNAS_SSPLIT == %d\n        tW += __builtin_amdgcn_s_memtime() - tS0;\n#endif\n" % n


# the vote's own stamp must not add into the same word
replace("            tW += __builtin_amdgcn_s_memtime() - tw0;\n", "#ifndef DNAS_SSPLIT\n            tW += __builtin_amdgcn_s_memtime() - tw0;\n#endif\n")
replace("        asm volatile(\"\" ::: \"memory\");   // other waves write LDS between sweeps: reload everything\n",
        "        asm volatile(\"\" ::: \"memory\");   // other waves write LDS between sweeps: reload everything\n        const unsigned long long tS0 = __builtin_amdgcn_s_memtime();\n")
replace("        if constexpr (G_ > 1 && kSplit) foldInbox(IntC<0>{});      // what was loaded behind the last row of the sweep before\n",
        "        if constexpr (G_ > 1 && kSplit) foldInbox(IntC<0>{});      // what was loaded behind the last row of the sweep before\n" + stamp(1))
replace("        ++rounds;\n        if constexpr (G_ > 1 && kSplit) loadInbox(IntC<0>{});\n",
        stamp(2) + "        ++rounds;\n        if constexpr (G_ > 1 && kSplit) loadInbox(IntC<0>{});\n" + stamp(3))
replace("          if (!__any(changed)) foldInbox(IntC<0>{});\n        }\n", "          " + stamp(4).replace("\n        tW", "\n          tW") + "          if (!__any(changed)) foldInbox(IntC<0>{});\n        }\n" + stamp(5))
open(sys.argv[1], "w").write(s)
