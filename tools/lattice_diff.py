"""Debug aid: decode one reference read on the GPU, compare every lattice cell with the oracle and
print the first mismatches.   python tools/lattice_diff.py l4c4.json hello.fa [global]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dnastore_amd as da
from oracle import oracle as O
G = os.path.join(ROOT, "tests", "golden", "ref_data")
mach, fa = sys.argv[1], sys.argv[2]
flags = dict(global_=True) if len(sys.argv) > 3 else {}
path = os.path.join(G, mach)
dec = da.ViterbiDecoder(da.Machine.fromFile(path), da.MutatorParams.fromFlags(**flags))
orc = O.ViterbiOracle(O.Machine.from_file(path), O.MutatorParams.from_cli(**flags))
read = da.read_fastseqs(os.path.join(G, fa))[0][1]
out, ll, st = dec.decode([read])
s, oll, olat = orc.decode(read, want_lattice=True)
lat = np.ascontiguousarray(dec.lattice(0, len(read)).transpose(0, 2, 1))
print(dec.tier, "gpu:", repr(out[0]), ll[0], "oracle:", repr(s), oll)
bad = np.argwhere(lat.view(np.uint64) != olat.view(np.uint64))
print("mismatching cells:", len(bad), "of", lat.size)
for pos, st_, lane in bad[:20]:
    print("pos %d state %d lane %d: gpu %r oracle %r" % (pos, st_, lane, lat[pos, st_, lane], olat[pos, st_, lane]))
if len(bad):
    import collections
    print("stats", dec.stats())
    print("by lane", sorted(collections.Counter(int(b[2]) for b in bad).items()))
    print("by pos (first 12)", sorted(collections.Counter(int(b[0]) for b in bad).items())[:12])
    fin = np.isfinite(lat)
    print("finite gpu cells by lane", fin.sum(axis=(0, 1)), "oracle", np.isfinite(olat).sum(axis=(0, 1)))
    print("gpu S(pos 0) finite:", fin[0, :, 0].sum(), " S(pos 1) finite:", fin[1, :, 0].sum(), "D(pos0) finite", fin[0, :, 1].sum())
