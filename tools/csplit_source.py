"""Writes a copy of csrc/viterbi_tiera.hip with extra cycle stamps INSIDE phase C (for tools/csplit.sh): built with
-DDNAS_STAMP -DDNAS_CSPLIT=n, the kernel adds the cycles from the start of phase C to split point n into the spare stamp word
(tools/stamp_gpu.py prints it as "extra stamp").  Points: 1 D lane stored, 2 accumulators cleared + barrier, 3 first load groups
issued, 4 next column's emit offers made, 10 + g load group g turned into its hand-over, 5 all groups done, 6 S lane stored,
7 every memory operation of the wave complete.

    python tools/csplit_source.py <output file>
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s = open(os.path.join(ROOT, "dnastore_amd", "csrc", "viterbi_tiera.hip")).read()
STAMP = "{ t1 = __builtin_amdgcn_s_memtime(); tX += t1 - t0; }"


def after(anchor, cond, body=STAMP):
    global s
    assert s.count(anchor) == 1, anchor
    s = s.replace(anchor, anchor + "#if %s\n      %s\n#endif\n" % (cond, body))


after("      const int xn = pos < L ? seq[pos] : 0;\n", "DNAS_CSPLIT == 1")
after("        __syncthreads();\n        earlyOffered = pos < c1;\n      }\n", "DNAS_CSPLIT == 2")
after("      static_for<0, PD>([&](auto gc) { issueGroup(gc); });\n", "DNAS_CSPLIT == 3")
after("        if (earlyOffered) emitOffers(xn);       // column pos + 1: ((S(pos) + score) + noGap) + sub[base][x_{pos+1}]\n      }\n", "DNAS_CSPLIT == 4")
after("        computeGroup(gc);\n", "DNAS_CSPLIT >= 10", "if constexpr (gc.value == DNAS_CSPLIT - 10) " + STAMP)
after("        if constexpr (gc.value + PD < NG) issueGroup(IntC<gc.value + PD>{});\n      });\n      }\n", "DNAS_CSPLIT == 5")
after("      STORE_LANE(pos, 0, S)\n", "DNAS_CSPLIT == 6")
after("#if DNAS_CSPLIT == 6\n      " + STAMP + "\n#endif\n", "DNAS_CSPLIT == 7", '{ asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t1 = __builtin_amdgcn_s_memtime(); tX += t1 - t0; }')
# what-if builds (wrong results, timing only): -DDNAS_EXP_NO_D leaves the D lane's stores out, -DDNAS_EXP_HCOLS=n loads n of the D-1
# history columns (the others reuse the first one's registers)
anchor = "      STORE_LANE(pos, 1, Dv)\n      if constexpr (G_ > 1) {"
assert s.count(anchor) == 1
s = s.replace(anchor, "#ifndef DNAS_EXP_NO_D\n      STORE_LANE(pos, 1, Dv)\n#endif\n      if constexpr (G_ > 1) {")
anchor = "            const unsigned off = (G_ == 1 || (pv & (1u << m2))) ? tid16 : kLaneOff;\n"
assert s.count(anchor) == 1
s = s.replace(anchor, anchor + "#ifdef DNAS_EXP_HCOLS\n            if constexpr (i > DNAS_EXP_HCOLS) { shBuf[gc.value % PD][2 * (m2 - p0)][i - 1] = shBuf[gc.value % PD][2 * (m2 - p0)][0]; shBuf[gc.value % PD][2 * (m2 - p0) + 1][i - 1] = shBuf[gc.value % PD][2 * (m2 - p0) + 1][0]; return; }\n#endif\n")
open(sys.argv[1], "w").write(s)
