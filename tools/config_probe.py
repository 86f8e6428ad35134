"""Timings of the BASELINE configurations that are not the bench line: config 2 (one ~1 kb read through the
46 670-state flusher*mixradar6*l4c4 composite, tier B) and config 4a (1024-nt reads through water64.1*l4c4,
7 066 states, tier A).  Prints time-to-decode / nt/s from the library's own HIP-event stats."""
import os, sys, random, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dnastore_amd as da
G = os.path.join(ROOT, "tests", "golden", "ref_data")


def compose(*names):
    m = da.Machine.fromFile(os.path.join(G, names[-1]))
    for n in reversed(names[:-1]):
        m = da.Machine.compose(da.Machine.fromFile(os.path.join(G, n)), m)
    return m


def substitute(rng, dna, rate):
    return "".join(rng.choice([b for b in "ACGT" if b != c]) if rng.random() < rate else c for c in dna)


params = da.MutatorParams.fromFlags(global_=True)
rng = random.Random(7)
m2 = compose("flusher.json", "mixradar6.json", "l4c4.json")
dec = da.ViterbiDecoder(m2, params)
read = substitute(rng, m2.encodeBytes(bytes(rng.randrange(256) for _ in range(128))), 0.01)
dec.decode([read])
t0 = time.perf_counter(); out, ll, st = dec.decode([read]); dt = time.perf_counter() - t0
s = dec.stats()
print("config 2: N=%d %s; one read of %d nt: wall %.1f ms, fill %.1f ms, traceback %.1f ms, %.0f nt/s; rounds/col %.1f; lattice %.2f GB" % (
    m2.nStates(), dec.tier[:6], len(read), dt * 1e3, s["fill_ms"], s["traceback_ms"], len(read) / dt, s["rounds"] / s["columns"], s["lattice_bytes"] / 1e9), flush=True)
many = [substitute(rng, m2.encodeBytes(bytes(rng.randrange(256) for _ in range(128))), 0.01) for _ in range(64)]
t0 = time.perf_counter(); out, ll, st = dec.decode(many); dt = time.perf_counter() - t0
print("          64 such reads: wall %.1f ms -> %.0f nt/s" % (dt * 1e3, sum(map(len, many)) / dt), flush=True)
dec.close()
m4 = compose("water64.1.json", "l4c4.json")
dec = da.ViterbiDecoder(m4, params)
reads = []
for i in range(1020):
    r = random.Random(5000 + i)
    reads.append(substitute(r, m4.encodeBytes(bytes(r.randrange(256) for _ in range(64))), 0.01))
dec.decode(reads[:16])
t0 = time.perf_counter(); out, ll, st = dec.decode(reads); dt = time.perf_counter() - t0
s = dec.stats()
nt = sum(map(len, reads))
print("config 4a: N=%d %s; %d reads of ~%d nt: wall %.1f ms, fill %.1f ms (%d launches) -> %.0f nt/s by fill, %.0f nt/s wall (host copies included); rounds/col %.1f" % (
    m4.nStates(), dec.tier[:6], len(reads), nt // len(reads), dt * 1e3, s["fill_ms"], s["fill_launches"], nt / (s["fill_ms"] / 1e3), nt / dt, s["rounds"] / s["columns"]))
print(dec.tier)
