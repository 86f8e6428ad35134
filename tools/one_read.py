"""Decode a few synthetic reads of the bench machine once (profiling target for rocprofv3 --pmc)."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 2:
    os.environ["DNAS_TIERA_DEFS"] = "\n".join(sys.argv[2:])
import dnastore_amd as da
G = os.path.join(ROOT, "tests", "golden", "ref_data")
m = da.Machine.fromFile(os.path.join(G, "s16h74l4c4.json"))
dec = da.ViterbiDecoder(m, da.MutatorParams.fromFlags(global_=True))
reads = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
    rng = random.Random(1000 + i)
    reads.append(m.encodeBytes(bytes(rng.randrange(256) for _ in range(29))))
out, ll, st = dec.decode(reads)
print(dec.tier, dec.stats())
