"""Diagnostic: per-phase cycle shares of the tier-A fill kernel (DNAS_STAMP build), one read."""
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["DNAS_TIERA_DEFS"] = "-DDNAS_STAMP" + "".join("\n" + d for d in sys.argv[2:])
import numpy as np
import dnastore_amd as da
from dnastore_amd import lib as L
G = os.path.join(ROOT, "tests", "golden", "ref_data")
m = da.Machine.fromFile(os.path.join(G, "s16h74l4c4.json"))
dec = da.ViterbiDecoder(m, da.MutatorParams.fromFlags(global_=True))
print(dec.tier)
import random
reads = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
    rng = random.Random(1000 + i)
    reads.append(m.encodeBytes(bytes(rng.randrange(256) for _ in range(29))))
out, ll, st = dec.decode(reads)
s = dec.stats()
w = (ctypes.c_ulonglong * 8)()
L.check(L.lib().dnas_model_debug_words(dec._h, w))
w = list(w)
cols = len(reads[0]) + 1
tot = sum(w[1:5])
print("stats", s)
print("block0: cycles/col  A %.0f  publish %.0f  sweeps %.0f  C %.0f  (total %.0f)  rounds/col %.1f  extra stamp %.0f" % (
    w[1] / cols, w[2] / cols, w[3] / cols, w[4] / cols, tot / cols, w[5] / cols, w[6] / cols))
print("shares: A %.1f%% P %.1f%% B %.1f%% C %.1f%%; cycles per sweep %.0f" % (100*w[1]/tot, 100*w[2]/tot, 100*w[3]/tot, 100*w[4]/tot, w[3]/max(w[5],1)))
print("fill_ms %.2f -> %.1f us/col" % (s["fill_ms"], s["fill_ms"] * 1e3 / cols))
