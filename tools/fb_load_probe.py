"""Where the time of the one-call E-step (dnas_fwdback_estep: handle + database upload + census + kernels) goes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dnastore_amd as da
from oracle import oracle as O
import bench_fwdback as BF
n = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
pairs = BF.make_pairs(O, 0, n, 2000)
pk = O.pack_pairs(pairs)
params = da.MutatorParams.fromFlags()
for rep in range(3):
    t0 = time.perf_counter(); fb = da.ForwardBackward(pk, device=0); t1 = time.perf_counter()
    c, ll, _ = fb.expectedCounts(params, want_pair_ll=False); t2 = time.perf_counter()
    c, ll, _ = fb.expectedCounts(params, want_pair_ll=False); t3 = time.perf_counter()
    st = fb.stats(); fb.close()
    t4 = time.perf_counter(); da.expectedCounts(params, pk, device=0); t5 = time.perf_counter()
    print("create+load %.1f ms, first E-step %.1f ms (kernels %.1f), second E-step %.1f ms; one-call form %.1f ms" % (
        (t1 - t0) * 1e3, (t2 - t1) * 1e3, st["kernel_ms"], (t3 - t2) * 1e3, (t5 - t4) * 1e3), flush=True)
