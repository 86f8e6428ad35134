"""One-off confidence run: N reads of a bench workload decoded on the GPU and, in parallel on the host cores, by the
CPU oracle; every decoded string and fp64 log-likelihood must be identical.
  python tools/bulk_parity.py 2000 14            (headline workload, configs[2])
  python tools/bulk_parity.py 64 14 --config 1   (the 46 670-state composite, tier C)
  python tools/bulk_parity.py 1000 14 --options checkpoint=always,segment=48   (the bounded-memory decode)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multiprocessing as mp


def _oracle_chunk(args):
    machine_json, reads = args
    from oracle import oracle as O
    orc = O.ViterbiOracle(O.Machine.from_json(machine_json), O.MutatorParams.from_cli(global_=True))
    return [orc.decode(r) for r in reads]


if __name__ == "__main__":
    options = None
    if "--options" in sys.argv:
        at = sys.argv.index("--options")
        options = sys.argv[at + 1]
        del sys.argv[at:at + 2]
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = int(argv[0]) if len(argv) > 0 else 500
    workers = int(argv[1]) if len(argv) > 1 else 8
    config = int(sys.argv[sys.argv.index("--config") + 1]) if "--config" in sys.argv else 2
    variant = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else "a"
    import bench
    import dnastore_amd as da
    wl = bench.workload(da, config, variant)
    m = wl["machine"]
    reads = bench.make_reads(m, 0, n, payload_bytes=wl["payload_bytes"])
    mj = m.toJSON()
    chunks = [reads[i::workers] for i in range(workers)]
    t0 = time.time()
    with mp.get_context("spawn").Pool(workers) as pool:
        fut = pool.map_async(_oracle_chunk, [(mj, c) for c in chunks])
        dec = da.ViterbiDecoder(m, da.MutatorParams.fromFlags(global_=True), options=options)
        out, ll, st = dec.decode(reads)
        print("%s; %s; options %s" % (wl["name"], dec.tier[:60], options))
        print("gpu done in %.1fs: %s" % (time.time() - t0, dec.stats()), flush=True)
        res = fut.get()
    bad = 0
    for w in range(workers):
        for k, (s_ref, ll_ref) in enumerate(res[w]):
            i = w + k * workers
            if out[i] != s_ref or float(ll[i]) != ll_ref or st[i] != 0:
                bad += 1
                if bad < 5:
                    print("MISMATCH read", i, repr(out[i][:40]), ll[i], "oracle", repr(s_ref[:40]), ll_ref)
    print("%d reads, %d mismatches, %.1fs" % (n, bad, time.time() - t0))
    sys.exit(1 if bad else 0)
