"""One-off confidence run: N reads of the bench workload decoded on the GPU and, in parallel on the host
cores, by the CPU oracle; every decoded string and fp64 log-likelihood must be identical.
  python tools/bulk_parity.py 2000 14"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multiprocessing as mp


def _oracle_chunk(args):
    reads, = args
    from oracle import oracle as O
    import bench
    orc = O.ViterbiOracle(O.Machine.from_file(bench.MACHINE), O.MutatorParams.from_cli(global_=True))
    return [orc.decode(r) for r in reads]


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    import bench
    import dnastore_amd as da
    m = da.Machine.fromFile(bench.MACHINE)
    reads = bench.make_reads(m, 0, n)
    chunks = [reads[i::workers] for i in range(workers)]
    t0 = time.time()
    with mp.get_context("spawn").Pool(workers) as pool:
        fut = pool.map_async(_oracle_chunk, [(c,) for c in chunks])
        dec = da.ViterbiDecoder(m, da.MutatorParams.fromFlags(global_=True))
        out, ll, st = dec.decode(reads)
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
