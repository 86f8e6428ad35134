import os, sys, random, threading, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import torch; torch.cuda.init()
import dnastore_amd as da
from test_gpu_checkpoint import _reads
m = da.Machine.fromFile("tests/golden/ref_data/s16h74l4c4.json")
params = da.MutatorParams.fromFlags(global_=True)
reads = _reads(da, m, random.Random(4), [29] * 300, rate=0.01)
lone = da.ViterbiDecoder(m, params, options="tier=C,cluster=2"); want = lone.decode(reads); lone.close()
for it in range(12):
    decs = [da.ViterbiDecoder(m, params, options="tier=C,cluster=2,arena_fraction=0.3") for _ in range(2)]
    got, errs = [None, None], [None, None]
    def work(i):
        try: got[i] = decs[i].decode(reads)
        except Exception as e: errs[i] = e
    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    t0 = time.time()
    [t.start() for t in ts]; [t.join() for t in ts]
    ok = [g is not None and g[0] == want[0] and np.array_equal(g[1].view(np.uint64), want[1].view(np.uint64)) for g in got]
    print(it, "%.2fs" % (time.time() - t0), "errs", [str(e)[:300] if e else None for e in errs], "ok", ok, "census", [d.cluster_census() if e is None else None for d, e in zip(decs, errs)], flush=True)
    [d.close() for d in decs]
