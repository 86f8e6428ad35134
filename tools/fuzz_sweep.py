"""Wider fuzz run than the test suite's: many random machines / error models / reads, GPU against the oracle -- decoded
string, log-likelihood, status and every lattice cell -- under every way the fill can run: tier A, tier B, tier C (clusters
of 2-4 work-groups), the bounded-memory decode in segments (tiers A and C; no lattice to compare), and the experimental row
program dealt by longest-path level (plan_order=2).
  python tools/fuzz_sweep.py 40 [first seed]"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import dnastore_amd as da
from oracle import oracle as O
from random_machines import random_machine, random_read

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
BASE = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
bad = 0
for case in range(n_cases):
    rng = random.Random(BASE + case)
    n_states = rng.choice([12, 40, 200, 700, 1500, 2300, 4200, 6500, 9000])
    text = random_machine(BASE + 500 + case, n_states)
    flags = dict(global_=rng.random() < 0.5, sub=rng.choice([0., .01, .05]), dup=rng.choice([0., .001, .02]),
                 del_open=rng.choice([0., .001, .03]), del_ext=rng.choice([.01, .2]), length=rng.choice([4, 8, 12]))
    orc = O.ViterbiOracle(O.Machine.from_json(text), O.MutatorParams.from_cli(**flags))
    reads = [random_read(7000 + 10 * case + r, text, max_len=rng.choice([5, 25, 60]), noise=rng.choice([0., .1, .3])) for r in range(3)]
    reads.append("")                       # an empty read in every batch
    want = [orc.decode(r, want_lattice=True) for r in reads]
    members = 2 + case % 3
    seg = max(flags["length"] + 2, rng.choice([6, 14, 33]))      # at least D + 2 columns
    modes = [("A", "tier=A", False, True), ("B", "tier=B", False, True), ("C", "tier=C,cluster=%d" % members, False, True),
             ("A segments", "tier=A,checkpoint=always,segment=%d" % seg, False, False),
             ("C segments", "tier=C,cluster=%d,checkpoint=always,segment=%d" % (members, seg), False, False),
             ("A by level", "tier=A,plan_order=2,plan_slack=%d" % (case % 9), False, True)]
    if flags["length"] > 8:
        modes = [mo for mo in modes if mo[0] == "B"]             # more than 8 duplication lanes: the general kernel only
    for mode, options, fwd, has_lattice in modes:
        try:
            dec = da.ViterbiDecoder(da.Machine.fromJSON(text), da.MutatorParams.fromFlags(**flags), options=options)
        except da.DnasError as e:
            if "was asked for" in str(e):                        # the machine does not fit that tier (e.g. too many edge scores)
                print("  (%s: %s)" % (mode, str(e)[:90]), flush=True)
                continue
            raise
        out, ll, st = dec.decode(reads)
        for i, r in enumerate(reads):
            s, oll, olat = want[i]
            ok = out[i] == s and (ll[i] == oll or (np.isinf(ll[i]) and np.isinf(oll)))
            nbad = 0
            if has_lattice:
                lat = np.ascontiguousarray(dec.lattice(i, len(r)).transpose(0, 2, 1))
                nbad = int((lat.view(np.uint64) != olat.view(np.uint64)).sum())
            if not ok or nbad:
                bad += 1
                print("MISMATCH case %d mode %s (%s) N=%d flags %s read %d %r: gpu %r %r st %d | oracle %r %r | cells %d" % (
                    case, mode, dec.tier[:6], n_states, flags, i, r, out[i], ll[i], st[i], s, oll, nbad), flush=True)
        dec.close()
    print("case %d N=%d %s ok" % (case, n_states, "global" if flags["global_"] else "local"), flush=True)
print("%d cases, %d mismatches" % (n_cases, bad))
sys.exit(1 if bad else 0)
