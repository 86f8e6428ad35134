"""How does the fill time per column change with the number of busy CUs?  Decodes 1, 32, 128, 256,
512 reads of the bench machine in one launch each and prints fill time per column per CU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dnastore_amd as da
from oracle import oracle as O
from synth import synthetic_reads
G = os.path.join(ROOT, "tests", "golden", "ref_data", "s16h74l4c4.json")
m = da.Machine.fromFile(G)
dec = da.ViterbiDecoder(m, da.MutatorParams.fromFlags(global_=True))
reads = synthetic_reads(O.Machine.from_file(G), 512, 29, seed=1000, sub=0.01)
dec.decode(reads[:8])
for nb in (1, 128, 255, 510, 510):
    out, ll, st = dec.decode(reads[:nb])
    s = dec.stats()
    cols = s["columns"]
    per_cu = cols / min(nb, 255)
    print("%4d reads: fill %.2f ms, launches %d, %.1f us per column per CU, traceback %.2f ms, rounds/col %.1f" % (
        nb, s["fill_ms"], s["fill_launches"], s["fill_ms"] * 1e3 / per_cu, s["traceback_ms"], s["rounds"] / cols), flush=True)
