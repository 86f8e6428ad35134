"""Two models of ONE process decode on clusters (tier C) on the same card from two threads.
python tools/loop_two_clusters.py [extra options] [iterations] [warm]
Without "warm" every iteration's calls allocate their 34-GB lattice arenas inside the timed region: hipMalloc / hipFree of such
blocks from two threads takes seconds now and then (tools/alloc_probe.py) -- the stalls once blamed on the cluster launches."""
import os, sys, random, threading, time
EXTRA = ("," + sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] else ""
ITER = int(sys.argv[2]) if len(sys.argv) > 2 else 12
WARM = len(sys.argv) > 3 and sys.argv[3] == "warm"
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import torch; torch.cuda.init()
import dnastore_amd as da
from test_gpu_checkpoint import _reads
m = da.Machine.fromFile("tests/golden/ref_data/s16h74l4c4.json")
params = da.MutatorParams.fromFlags(global_=True)
reads = _reads(da, m, random.Random(4), [29] * 300, rate=0.01)
lone = da.ViterbiDecoder(m, params, options="tier=C,cluster=2"); want = lone.decode(reads); lone.close()
for it in range(ITER):
    decs = [da.ViterbiDecoder(m, params, options="tier=C,cluster=2,arena_fraction=0.3" + EXTRA) for _ in range(2)]
    got, errs = [None, None], [None, None]
    if WARM: [d.decode(reads) for d in decs]
    walls = [0.0, 0.0]
    def work(i):
        t1 = time.time()
        try: got[i] = decs[i].decode(reads)
        except Exception as e: errs[i] = e
        walls[i] = time.time() - t1
    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    t0 = time.time()
    [t.start() for t in ts]; [t.join() for t in ts]
    ok = [g is not None and g[0] == want[0] and np.array_equal(g[1].view(np.uint64), want[1].view(np.uint64)) for g in got]
    print(it, "%.2fs" % (time.time() - t0), "errs", [str(e)[:300] if e else None for e in errs], "ok", ok, "census", [d.cluster_census() if e is None else None for d, e in zip(decs, errs)],
          "fill/traceback ms", [("%.0f/%.0f" % (d.stats()["fill_ms"], d.stats()["traceback_ms"])) if e is None else None for d, e in zip(decs, errs)], "call s", ["%.2f" % w for w in walls], flush=True)
    [d.close() for d in decs]
