"""Fingerprint of the row programs and tables host/plan.cpp produces for the fixture machines (tier A and clusters of 2 and 3,
default dealing order): a refactoring of the planner must leave every line unchanged.
  python tools/plan_hash.py > /tmp/before.txt ; ... ; python tools/plan_hash.py | diff /tmp/before.txt -"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dnastore_amd as da
G = os.path.join(ROOT, "tests", "golden", "ref_data")


def machines():
    for n in ("l4c4.json", "mr2l4c4.json", "h74l4c4.json", "s16mr2l4c4.json", "s16h74l4c4.json"):
        yield n, da.Machine.fromFile(os.path.join(G, n))
    yield "water64.1*l4c4", da.Machine.compose(da.Machine.fromFile(os.path.join(G, "water64.1.json")), da.Machine.fromFile(os.path.join(G, "l4c4.json")))
    if "--big" in sys.argv:
        m = da.Machine.fromFile(os.path.join(G, "l4c4.json"))
        for part in ("mixradar6.json", "flusher.json"):
            m = da.Machine.compose(da.Machine.fromFile(os.path.join(G, part)), m)
        yield "flusher*mixradar6*l4c4", m


for name, m in machines():
    for glob in (True, False):
        fm = da.FlatModel(m, da.MutatorParams.fromFlags(global_=glob))
        for fwd in ("0",):
            for members in ((1, 2, 3) if "mixradar6" not in name else (0,)):
                try:
                    pl = fm.cluster_plan(members)
                except da.DnasError as e:
                    print(name, "global" if glob else "local", "fwd", fwd, "members", members, "->", str(e)[:80])
                    continue
                h = hashlib.sha256()
                for key in ("shapes", "entries", "meta", "member_of", "lds_index", "lattice_slot", "fold"):
                    h.update(np.ascontiguousarray(pl[key]).tobytes())
                print(name, "global" if glob else "local", "fwd", fwd, "members", members, "G", pl["G"], "K", pl["K"], "T", pl["T"], "entries", pl["n_entries"],
                      "S", pl["n_s_rows"], h.hexdigest()[:20], flush=True)
