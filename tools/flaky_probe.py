"""Debug aid: decode the same read repeatedly (fresh decoder each time / same decoder) and report
how often the GPU result deviates from the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dnastore_amd as da
from oracle import oracle as O
G = os.path.join(ROOT, "tests", "golden", "ref_data")
mach, fa = sys.argv[1], sys.argv[2]
path = os.path.join(G, mach)
orc = O.ViterbiOracle(O.Machine.from_file(path), O.MutatorParams.from_cli())
read = da.read_fastseqs(os.path.join(G, fa))[0][1]
s, oll, olat = orc.decode(read, want_lattice=True)
m = da.Machine.fromFile(path)
for trial in range(int(sys.argv[3])):
    dec = da.ViterbiDecoder(m, da.MutatorParams.fromFlags())
    res = []
    for rep in range(4):
        out, ll, st = dec.decode([read])
        lat = np.ascontiguousarray(dec.lattice(0, len(read)).transpose(0, 2, 1))
        nbad = int((lat.view(np.uint64) != olat.view(np.uint64)).sum())
        res.append((out[0] == s, ll[0] == oll, nbad))
        if nbad and not any(r[2] for r in res[:-1]):
            fin = np.isfinite(lat)
            print("  first bad: finite S(pos0) %d D(pos0) %d S(pos1) %d; stats %s" % (fin[0, :, 0].sum(), fin[0, :, 1].sum(), fin[1, :, 0].sum(), dec.stats()))
    print("decoder", trial, res, flush=True)
    dec.close()
