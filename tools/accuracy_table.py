"""Reproduce the reference's published accuracy tables doc/len4[.mix2|.ham].{subs,dels,dups}.tab (fixtures: tests/golden/ref_doc/)
on the GPU with the method of doc/errdecode.pl as restated in tests/accuracy_tables.py: 20 payloads of 8192 bits per rate through
l4c4.json / mr2l4c4.json / h74l4c4.json, edits per bit; the error model FITTED per row on ten simulated alignments (GPU
Baum-Welch, as the tables were made) or, with --exact, told its rates (the script's -exacterrs).  Prints our row next to the table's.
  python tools/accuracy_table.py len4.ham.subs [rows, e.g. 1-20] [--exact] > profiles/r4_len4_ham_subs[_exact].txt"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import dnastore_amd as da
import accuracy_tables as AT

args = [a for a in sys.argv[1:] if not a.startswith("--")]
which = args[0] if args else "len4.ham.subs"
if which in ("subs", "dels"):
    which = "len4.ham." + which
exact = "--exact" in sys.argv
table = AT.read_table(os.path.join(ROOT, "tests", "golden", "ref_doc", "%s.tab" % which))
lo, hi = (int(v) for v in (args[1] if len(args) > 1 else "1-%d" % len(table)).split("-"))
machine = da.Machine.fromFile(os.path.join(ROOT, "tests", "golden", "ref_data", AT.machine_file(which)))
work = tempfile.mkdtemp()
print("# doc/%s.tab against dnastore_amd (GPU Viterbi; error model %s), 20 x 8192 bits per row" % (
    which, "given its rates: --error-sub-prob / --error-del-open <rate> --error-del-ext 0.2 --error-dup-prob 0 --error-global --length 4" if exact
    else "fitted per row on ten simulated 8192-base alignments by the GPU Baum-Welch, written as JSON and read back, as the table's was"))
print("row rate  table_mean table_sd  ours_mean ours_sd  (ours-table)/SE  nt_decoded decode_s  model")
for n in range(lo, hi + 1):
    row = table[n - 1]
    sub, dele, dup = row["SubProb"], row["DelProb"], row["DupProb"]
    tag = which[len("len4.ham."):] if which.startswith("len4.ham.") else which      # (the seeds of round 4's first two tables)
    t0 = time.time()
    if exact:
        params, note = da.MutatorParams.fromFlags(sub=sub, dup=dup, del_open=dele, del_ext=0.2, global_=True, length=4), "exact"
    else:
        params, text, iters = AT.fit_model(da, sub, dele, "%s%d" % (tag, n), work, dup_rate=dup)
        note = "fitted in %d iterations (%.1f s): %s" % (iters, time.time() - t0, " ".join(text.split()))
    cases = [AT.make_case_general(machine, sub, dele, rep, tag, dup_rate=dup) for rep in range(20)]
    dec = da.ViterbiDecoder(machine, params)
    t0 = time.time()
    out, ll, st = dec.decode([c[1] for c in cases])
    dt = time.time() - t0
    tier = dec.tier[:40]
    dec.close()
    per_bit = np.array([AT.edit_distance(c[0], s.replace("^", "").replace("$", "")) / AT.BITS for c, s in zip(cases, out)])
    se = np.hypot(row["StDevEditsPerBit"], per_bit.std()) / np.sqrt(20)
    z = (per_bit.mean() - row["MeanEditsPerBit"]) / se if se > 0 else 0.0
    print("%2d %-7g  %.4e %.3e  %.4e %.3e  %+.2f  %d %.2f  %s" % (n, max(sub, dele, dup), row["MeanEditsPerBit"], row["StDevEditsPerBit"], per_bit.mean(),
                                                             per_bit.std(), z, sum(len(c[1]) for c in cases), dt, note), flush=True)
print("# served by:", tier)
