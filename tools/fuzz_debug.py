import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import dnastore_amd as da
from oracle import oracle as O
from random_machines import random_machine, random_read
seed, n, glob = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3] == "g"
text = random_machine(seed, n)
flags = dict(global_=glob, sub=.02, dup=.01, del_open=.02, del_ext=.1)
dec = da.ViterbiDecoder(da.Machine.fromJSON(text), da.MutatorParams.fromFlags(**flags))
print(dec.tier[:120])
orc = O.ViterbiOracle(O.Machine.from_json(text), O.MutatorParams.from_cli(**flags))
reads = [random_read(100 * seed + r, text, max_len=30) for r in range(4)]
out, ll, st = dec.decode(reads)
for i, r in enumerate(reads):
    s, oll, olat = orc.decode(r, want_lattice=True)
    lat = np.ascontiguousarray(dec.lattice(i, len(r)).transpose(0, 2, 1))
    bad = np.argwhere(lat.view(np.uint64) != olat.view(np.uint64))
    print(i, r, "gpu", repr(out[i]), ll[i], st[i], "| oracle", repr(s), oll, "| lattice mismatches", len(bad), bad[:3].tolist())
