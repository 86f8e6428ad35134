"""Tier C timings: `reads` ~1 kb reads through the 46 670-state flusher*mixradar6*l4c4 composite (BASELINE configs[1])
or the 258 538-state hamming74*dropdot*water64.1*l4c4 composite (configs[3] as written), from the library's HIP events.
  python tools/tierc_probe.py [2|4b] [reads] [options e.g. cluster=8,max_clusters=32] [repeat]"""
import os, sys, random, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import dnastore_amd as da
from test_gpu_tier_c import _compose, _substitute, DROPDOT
G = os.path.join(ROOT, "tests", "golden", "ref_data")
which = sys.argv[1] if len(sys.argv) > 1 else "2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
options = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != "-" else None
repeat = int(sys.argv[4]) if len(sys.argv) > 4 else 2
params = da.MutatorParams.fromFlags(global_=True)
if which == "2":
    m, nbytes = _compose(da, G, "flusher.json", "mixradar6.json", "l4c4.json"), 128
else:
    m, nbytes = _compose(da, G, "hamming74.json", da.Machine.fromJSON(DROPDOT), "water64.1.json", "l4c4.json"), 32
t0 = time.time()
dec = da.ViterbiDecoder(m, params, options=options)
print("model: %d states, %s (create %.1f s)" % (m.nStates(), dec.tier[:100], time.time() - t0), flush=True)
reads = []
for i in range(n):
    r = random.Random(7000 + i)
    reads.append(_substitute(r, m.encodeBytes(bytes(r.randrange(256) for _ in range(nbytes))), 0.01))
nt = sum(map(len, reads))
dec.decode(reads[:min(n, 8)])
for it in range(repeat):
    t0 = time.perf_counter(); out, ll, st = dec.decode(reads); dt = time.perf_counter() - t0
    s = dec.stats()
    frac = s["lattice_bytes"] / (s["fill_ms"] / 1e3) / 8e12
    print("%d reads, %d nt: wall %.1f ms, fill %.1f ms (%d launches), traceback %.1f ms; %.0f nt/s by fill; %.1f us/column/cluster-slot; sweeps/col/wg %.1f; "
          "canonical roofline frac %.3f; census %s; status ok %s" % (
              n, nt, dt * 1e3, s["fill_ms"], s["fill_launches"], s["traceback_ms"], nt / (s["fill_ms"] / 1e3),
              s["fill_ms"] * 1e3 / s["columns"] * min(n, (int(dec.tier.split(",")[1].split()[0]) if dec.tier.startswith("tier C") else 256)), s["rounds"] / s["columns"], frac, dec.cluster_census(), not st.any()), flush=True)
if "DNAS_STAMP" in os.environ.get("DNAS_TIERA_DEFS", ""):
    import ctypes
    from dnastore_amd import lib as L
    w = (ctypes.c_ulonglong * 8)()
    L.check(L.lib().dnas_model_debug_words(dec._h, w))
    w = list(w)
    cols = len(reads[0]) + 1     # block 0 = member 0 of cluster 0 = read 0 (last of its reads: stamps accumulate over them)
    nreads0 = (n + min(n, (int(dec.tier.split(",")[1].split()[0]) if dec.tier.startswith("tier C") else 256)) - 1) // min(n, (int(dec.tier.split(",")[1].split()[0]) if dec.tier.startswith("tier C") else 256))
    cols *= nreads0
    print("block 0, per column (100 MHz ticks -> us): offers %.2f  barrier %.2f  take %.2f  fixpoint %.2f (of which wave 0 in the cluster vote %.2f)  phase C %.2f; sweeps/col %.1f" % (
        w[1] / cols / 100, w[6] / cols / 100, w[2] / cols / 100, w[3] / cols / 100, w[7] / cols / 100, w[4] / cols / 100, w[5] / cols))
dec.close()
