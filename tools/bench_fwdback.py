#!/usr/bin/env python3
"""Secondary measurement (BASELINE configs[4], SURVEY 8d "Config 5"): forward-backward E-step
(expectedCounts) over synthetic (original, read) pairs, 256-nt originals with tandem duplications
(len 1-3, rate .01), substitutions (.02), deletions (.01), true alignment as guide, CLI default model
(P = 6).  Prints one JSON line: pairs/s and nt/s on the GPU, and the CPU oracle timed on a sample."""
import argparse, json, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import dnastore_amd as da
from synth import synthetic_alignment
from oracle import oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=100000)
ap.add_argument("--unique", type=int, default=2000, help="distinct synthetic pairs (tiled to --pairs)")
ap.add_argument("--cpu-pairs", type=int, default=300)
args = ap.parse_args()

rng = random.Random(5)
uniq = [O.alignment_pair(synthetic_alignment(rng, 256, sub=.02, dele=.01, dup=.01)) for _ in range(args.unique)]
pairs = [uniq[i % args.unique] for i in range(args.pairs)]
pk = O.pack_pairs(pairs)
nt = int(pk["out_off"][-1])
params = da.MutatorParams.fromFlags()
da.expectedCounts(params, O.pack_pairs(uniq[:64]))          # warm-up (context, table upload)
t0 = time.perf_counter()
counts, ll, per = da.expectedCounts(params, pk)
dt = time.perf_counter() - t0
oparams = O.MutatorParams.from_cli()
t1 = time.perf_counter()
oc, oll, oper = O.expected_counts(oparams, pairs[:args.cpu_pairs])
cdt = time.perf_counter() - t1
assert np.array_equal(per[:args.cpu_pairs], oper), "per-pair log-likelihood parity"
gc, gll, gper = da.expectedCounts(params, O.pack_pairs(pairs[:args.cpu_pairs]))
assert np.allclose(gc, oc, rtol=1e-9, atol=1e-300), "counts parity"
cpu_nt = int(sum(len(p[1]) for p in pairs[:args.cpu_pairs]))
print(json.dumps({"metric": "forward-backward E-step, read nt/s (1 MI355X, host buffers in, counts out)", "value": nt / dt, "unit": "nt/s",
                  "pairs": args.pairs, "pairs_per_s": args.pairs / dt, "seconds": dt, "dtype": "f64", "data": "synthetic",
                  "cpu_baseline": {"value": cpu_nt / cdt, "unit": "nt/s", "cores": 1, "kind": "port",
                                   "sample": "%d pairs, oracle/fwdback_oracle.c, %.1f s" % (args.cpu_pairs, cdt)},
                  "parity": "per-pair ll bit-exact on the sample; counts within 1e-9"}))
