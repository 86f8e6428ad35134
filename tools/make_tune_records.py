"""Row-program verdicts for the fixture and bench machines, from bench-like reads: for every machine the candidate programs
(dealing order 1 / 2, with / without F rows: options plan_order, plan_fwd) decode 720 reads -- encoded random payloads with
1 % substitutions, as bench.py makes them -- three times; the fastest fill of the last two runs counts, and a candidate has
to beat the default (order=1 fwd=0) by 1.5 %.  The records go to the directory given (tools/make_tune_records.sh copies them
to dnastore_amd/tune/, where the library finds them); their names hash the kernel source and the planner version, so they
are made again whenever either changes.  (The library's own tuning run, for machines without a record, has no encoder at
hand and uses what a random walk through the machine emits; for s16h74l4c4 that ranks the two dealing orders the other way
round than real reads do, by 2 % either way.)
  python tools/make_tune_records.py <output directory>"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DNAS_AUTOTUNE"] = "0"
for k in ("DNAS_PLAN_FWD", "DNAS_PLAN_ORDER"):
    os.environ.pop(k, None)
import dnastore_amd as da
import bench
OUT = sys.argv[1]
os.makedirs(OUT, exist_ok=True)
G = os.path.join(ROOT, "tests", "golden", "ref_data")
machines = [(n, da.Machine.fromFile(os.path.join(G, n)), 29) for n in ("l4c4.json", "mr2l4c4.json", "h74l4c4.json", "s16mr2l4c4.json", "s16h74l4c4.json")]
wl = bench.workload(da, 3, "a")
machines.append(("water64.1*l4c4", wl["machine"], wl["payload_bytes"]))
params = da.MutatorParams.fromFlags(global_=True)
for name, m, payload in machines:
    reads = bench.make_reads(m, 0, 720, payload_bytes=payload)
    fm = da.FlatModel(m, params)
    results = []
    for order, fwd in ((1, 0), (2, 0), (1, 1)):
        try:
            dec = da.ViterbiDecoder(m, params, options="tier=A,autotune=0,plan_order=%d,plan_fwd=%d" % (order, fwd))
        except da.DnasError as e:
            continue
        ms = []
        for rep in range(3):
            dec.decode(reads)
            ms.append(dec.stats()["fill_ms"])
        dec.close()
        results.append((order, fwd, min(ms[1:])))
    best = results[0]
    for r in results[1:]:
        if r[2] < 0.985 * best[2] and r[2] < min(x[2] for x in results if x is not r) + 1e-9:
            best = r
    text = "order=%d fwd=%d   (fill of 720 bench reads, %s;%s)\n" % (best[0], best[1], name, "".join("  order=%d fwd=%d: %.2f ms" % r for r in results))
    path = os.path.join(OUT, fm.tune_record_name())
    open(path, "w").write(text)
    print(os.path.basename(path), text.strip(), flush=True)
