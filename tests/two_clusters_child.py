"""One of two PROCESSES that decode on clusters of work-groups (tier C) on the same card at the same time
(tests/test_gpu_two_cluster_processes.py, tools/two_cluster_processes.py).  Prints one JSON line: the wall time of every
decode call, the fill kernels' time in it (HIP events) and a digest of the results.

    python tests/two_clusters_child.py <start file> <calls> <options>
"""
import hashlib
import json
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def reads_and_model(da):
    from test_gpu_checkpoint import _reads
    data = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_data")
    m = da.Machine.fromFile(os.path.join(data, "s16h74l4c4.json"))
    params = da.MutatorParams.fromFlags(global_=True)
    # 300 reads on clusters of two work-groups: a launch asks for 256 work-groups, every CU of the card
    return m, params, _reads(da, m, random.Random(4), [29] * 300, rate=0.01)


def digest(strings, loglikes):
    h = hashlib.sha256()
    for s in strings:
        h.update(s.encode() + b"\n")
    h.update(loglikes.tobytes())
    return h.hexdigest()


def main():
    start_file, calls, options = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    import dnastore_amd as da
    m, params, reads = reads_and_model(da)
    # The model and the arena of a full call are made ONE PROCESS AT A TIME: hipMalloc of tens of GB stalls whoever else talks to the
    # driver -- a kernel of the other process included, for seconds (tools/alloc_probe.py) --, and a cluster launch that is held up
    # for longer than its watchdog allows gives up with DNAS_E_DEVICE (seen once in the warm-up of a full test run).  The calls that
    # the test is about, below, run side by side with everything allocated.
    import fcntl
    with open(start_file + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        dec = da.ViterbiDecoder(m, params, options=options)
        dec.decode(reads)                       # code object loaded, the arena of a full call allocated
        fcntl.flock(lock, fcntl.LOCK_UN)
    open(start_file + ".%d" % os.getpid(), "w").close()
    while not os.path.exists(start_file):        # both processes start their calls together
        time.sleep(0.002)
    walls, fills, digests = [], [], set()
    for _ in range(calls):
        t0 = time.time()
        strings, ll = dec.decode(reads)[:2]
        walls.append(time.time() - t0)
        fills.append(dec.stats()["fill_ms"])
        digests.add(digest(strings, ll))
    print(json.dumps({"walls_s": walls, "fill_ms": fills, "digests": sorted(digests), "tier": dec.tier,
                      "t_first": t0 - sum(walls[:-1]), "t_last": time.time()}), flush=True)
    # the arena goes back only when the other process is through as well: hipFree of tens of GB stalls whoever else talks to
    # the driver at that moment (tools/alloc_probe.py)
    deadline = time.time() + 120
    while not os.path.exists(start_file + ".done") and time.time() < deadline:
        time.sleep(0.01)
    dec.close()


if __name__ == "__main__":
    main()
