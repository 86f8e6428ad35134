"""Diagnostic: configs[1] reads decoded in batches of different sizes must agree with each other (status, string, log-likelihood)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dnastore_amd as da, bench
wl = bench.workload(da, 1, "a")
m = wl["machine"]
reads = bench.make_reads(m, 0, 64, payload_bytes=wl["payload_bytes"])
dec = da.ViterbiDecoder(m, da.MutatorParams.fromFlags(global_=True), options=sys.argv[1] if len(sys.argv) > 1 else None)
print(dec.tier[:90])
ref = dec.decode(reads)
print("64 reads: status", np.bincount(ref[2], minlength=4))
for n in (1, 2, 3, 5, 8, 9, 16):
    for rep in range(2):
        out, ll, st = dec.decode(reads[:n])
        same = sum(out[i] == ref[0][i] and ll[i] == ref[1][i] for i in range(n))
        print("%2d reads (run %d): status %s, %d of %d equal to the 64-read batch" % (n, rep, np.bincount(st, minlength=4), same, n), flush=True)
