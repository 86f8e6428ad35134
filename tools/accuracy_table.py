"""Reproduce the reference's published accuracy table doc/len4.ham.subs.tab (fixture: tests/golden/ref_doc/) on the GPU decoder with
the method of doc/errdecode.pl as restated in tests/accuracy_tables.py: 20 payloads of 8192 bits per substitution rate through
h74l4c4.json, edits per bit.  Prints our row next to the table's.
  python tools/accuracy_table.py [rows, e.g. 1-20] > profiles/r4_len4_ham_subs.txt"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import dnastore_amd as da
import accuracy_tables as AT

table = AT.read_table(os.path.join(ROOT, "tests", "golden", "ref_doc", "len4.ham.subs.tab"))
lo, hi = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1-20").split("-"))
machine = da.Machine.fromFile(os.path.join(ROOT, "tests", "golden", "ref_data", "h74l4c4.json"))
print("# doc/len4.ham.subs.tab against dnastore_amd (GPU Viterbi, error model given its rates: --error-sub-prob <rate> --error-dup-prob 0 "
      "--error-del-open 0 --error-del-ext 0.2 --error-global --length 4; the table's model was fitted per row), 20 x 8192 bits per row")
print("row SubProb  table_mean table_sd  ours_mean ours_sd  (ours-table)/SE  nt_decoded decode_s")
for n in range(lo, hi + 1):
    row = table[n - 1]
    rate = row["SubProb"]
    cases = [AT.make_case(machine, rate, rep) for rep in range(20)]
    dec = da.ViterbiDecoder(machine, da.MutatorParams.fromFlags(sub=rate, dup=0.0, del_open=0.0, del_ext=0.2, global_=True, length=4))
    t0 = time.time()
    out, ll, st = dec.decode([c[1] for c in cases])
    dt = time.time() - t0
    tier = dec.tier[:40]
    dec.close()
    per_bit = np.array([AT.edit_distance(c[0], s.replace("^", "").replace("$", "")) / AT.BITS for c, s in zip(cases, out)])
    se = np.hypot(row["StDevEditsPerBit"], per_bit.std()) / np.sqrt(20)
    z = (per_bit.mean() - row["MeanEditsPerBit"]) / se if se > 0 else 0.0
    print("%2d %-7g  %.4e %.3e  %.4e %.3e  %+.2f  %d %.2f" % (n, rate, row["MeanEditsPerBit"], row["StDevEditsPerBit"], per_bit.mean(), per_bit.std(),
                                                       z, sum(len(c[1]) for c in cases), dt), flush=True)
print("# served by:", tier)
