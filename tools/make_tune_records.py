"""Row-program verdicts for the fixture and bench machines, from bench-like reads: for every machine the candidate programs
(dealing order 1, order 2 with every slack share 0 .. 8: options plan_order, plan_slack) decode 2160 reads (three launches) -- encoded random payloads with
1 % substitutions, as bench.py makes them -- three times; the fastest fill of the last two runs counts, and a candidate has
to beat the default (order=1) by 4 %.  The records go to the directory given (tools/make_tune_records.sh copies them
to dnastore_amd/tune/, where the library finds them); their names hash the machine and the planner version, and each names the
kernel source it was measured with (tests/test_tune_records.py fails when that is no longer the library's).  (The library's own tuning run, for machines without a record, has no encoder at
hand and uses what a random walk through the machine emits; for s16h74l4c4 that ranks the two dealing orders the other way
round than real reads do, by 2 % either way.)
  python tools/make_tune_records.py <output directory> [--no-4b] [--only <part of a machine's name>]"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
for k in ("DNAS_PLAN_SLACK", "DNAS_PLAN_ORDER"):
    os.environ.pop(k, None)
import dnastore_amd as da
import bench
OUT = sys.argv[1]
os.makedirs(OUT, exist_ok=True)
G = os.path.join(ROOT, "tests", "golden", "ref_data")
# (name, machine, payload bytes per read, reads, members: 1 = tier A, 0 = tier C with the smallest cluster)
machines = [(n, da.Machine.fromFile(os.path.join(G, n)), 29, 2160, 1) for n in ("l4c4.json", "mr2l4c4.json", "h74l4c4.json", "s16mr2l4c4.json", "s16h74l4c4.json")]
wl = bench.workload(da, 3, "a")
machines.append(("water64.1*l4c4", wl["machine"], wl["payload_bytes"], 2160, 1))
wl = bench.workload(da, 1, "a")
machines.append(("configs[1], 46 670 states", wl["machine"], wl["payload_bytes"], 64, 0))
if "--no-4b" not in sys.argv:
    wl = bench.workload(da, 3, "b")
    machines.append(("configs[3] as written, 258 538 states", wl["machine"], wl["payload_bytes"], 24, 0))
if "--only" in sys.argv:                         # one machine again, e.g. --only s16h74l4c4
    machines = [mm for mm in machines if sys.argv[sys.argv.index("--only") + 1] in mm[0]]
params = da.MutatorParams.fromFlags(global_=True)
for name, m, payload, n_reads, members in machines:
    # tier A: three launches -- the traceback of one runs beside the fill of the next, as in a long job; clusters: one
    reads = bench.make_reads(m, 0, n_reads, payload_bytes=payload)
    fm = da.FlatModel(m, params)
    results = []
    candidates = [(1, 0)] + [(2, sl) for sl in (range(9) if members == 1 else (0, 4, 8))]
    for order, slack in candidates:
        try:
            dec = da.ViterbiDecoder(m, params, options="tier=%s,plan_order=%d,plan_slack=%d" % ("A" if members == 1 else "C", order, slack))
        except da.DnasError as e:
            continue
        ms = []
        for rep in range(3):
            dec.decode(reads)
            ms.append(dec.stats()["fill_ms"])
        dec.close()
        results.append((order, slack, min(ms[1:])))
    best = results[0]
    fastest = min(results, key=lambda r: r[2])
    confirm = ""
    if fastest[2] < 0.96 * best[2]:      # the default dealing stays unless another is 4 % faster: the timings repeat to 1-3 %
        # ... and the verdict must REPRODUCE: a fresh model of the winner and of the default, three timed runs each, medians (one fast
        # sample among slower neighbours -- round 3's s16mr2l4c4 record -- does not replace the default)
        def median_ms(order, slack):
            dec = da.ViterbiDecoder(m, params, options="tier=%s,plan_order=%d,plan_slack=%d" % ("A" if members == 1 else "C", order, slack))
            ms = []
            for rep in range(4):
                dec.decode(reads)
                ms.append(dec.stats()["fill_ms"])
            dec.close()
            return sorted(ms[1:])[1]
        again_w, again_d = median_ms(fastest[0], fastest[1]), median_ms(best[0], best[1])
        confirm = "  confirmation (medians of three): %d/%d %.2f ms, %d/%d %.2f ms" % (fastest[0], fastest[1], again_w, best[0], best[1], again_d)
        if again_w < 0.97 * again_d:
            best = fastest
    text = "order=%d slack=%d kernel=%s   (fill of %d bench reads, %s;%s;%s)\n" % (best[0], best[1], da.FlatModel.kernel_source_hash(), n_reads, name,
                                                                                  "".join("  %d/%d: %.2f ms" % r for r in results), confirm)
    path = os.path.join(OUT, fm.tune_record_name(members))
    open(path, "w").write(text)
    print(os.path.basename(path), text.strip(), flush=True)
