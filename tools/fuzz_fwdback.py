"""Wider fuzz run for the forward-backward E-step: random error models (zero probabilities included), random
alignments (degenerate ones included), strict and non-strict guides; GPU against the oracle: per-pair
log-likelihoods bit-identical, counts to 1e-9 relative.
  python tools/fuzz_fwdback.py 30"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import dnastore_amd as da
from oracle import oracle as O
from synth import synthetic_alignment

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = 0
for case in range(n_cases):
    rng = random.Random(4000 + case)
    flags = dict(sub=rng.choice([1e-9, .01, .1]), dup=rng.choice([1e-9, .001, .05]), del_open=rng.choice([1e-9, .001, .05]),
                 del_ext=rng.choice([1e-9, .01, .3]), length=rng.choice([2, 6, 12]))
    strict = rng.random() < 0.4
    rows = [synthetic_alignment(rng, rng.choice([1, 2, 7, 33, 100, 256]), sub=rng.choice([0., .05]), dele=rng.choice([0., .03]),
                                dup=rng.choice([0., .03])) for _ in range(rng.choice([1, 5, 70]))]
    pairs = [O.alignment_pair(r) for r in rows]
    want = O.expected_counts(O.MutatorParams.from_cli(**flags), pairs, strict=strict)
    pk = O.pack_pairs(pairs)
    got = da.expectedCounts(da.MutatorParams.fromFlags(**flags), pk, strict=strict)
    ll_ok = np.array_equal(np.asarray(got[2]).view(np.uint64), np.asarray(want[2]).view(np.uint64))
    scale = np.maximum(np.abs(want[0]), 1e-300)
    rel = np.abs(got[0] - want[0]) / scale
    c_ok = bool(np.all((rel < 1e-9) | (np.abs(got[0] - want[0]) < 1e-12)))
    if not (ll_ok and c_ok):
        bad += 1
        print("MISMATCH case %d flags %s strict %s pairs %d: ll_ok %s max rel %.3g" % (case, flags, strict, len(pairs), ll_ok, rel.max()), flush=True)
    else:
        print("case %d ok (%d pairs, strict %s)" % (case, len(pairs), strict), flush=True)
print("%d cases, %d mismatches" % (n_cases, bad))
sys.exit(1 if bad else 0)
