"""Does the time of ONE read on a cluster (configs[1], 16 x 512 threads) depend on WHERE the cluster's sync words sit in memory?
Every line is a fresh decoder in the same process on the same box: DNAS_SYNC_OFFSET=<bytes> forces the place inside the 64-KB window,
`placed` = the library's own choice (sync_latency_kernel: measured round trips from every XCD), `start` = option sync_place=0.
  python tools/sync_offset_probe.py [offsets in bytes ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dnastore_amd as da
import bench
wl = bench.workload(da, 1, "a")
m = wl["machine"]
params = da.MutatorParams.fromFlags(global_=True)
reads = bench.make_reads(m, 0, 1, payload_bytes=wl["payload_bytes"])
offs = sys.argv[1:] or ["placed", "start", "0", "1024", "4096", "8192", "16384", "32768", "65536", "placed", "placed"]
for off in offs:
    os.environ.pop("DNAS_SYNC_OFFSET", None)
    opts = os.environ.get("PROBE_PLAN", "threads=512,cluster=16")     # PROBE_PLAN="threads=512,cluster=4": the throughput plan's clusters
    if off == "start":
        opts += ",sync_place=0"
    elif off != "placed":
        os.environ["DNAS_SYNC_OFFSET"] = off
    t = time.perf_counter()
    dec = da.ViterbiDecoder(m, params, options=opts)
    t_create = (time.perf_counter() - t) * 1e3
    dec.decode(reads)
    fills, walls = [], []
    for _ in range(3):
        t = time.perf_counter(); dec.decode(reads); walls.append((time.perf_counter() - t) * 1e3); fills.append(dec.stats()["fill_ms"])
    print("%8s: fill %.2f ms (%s), wall %.2f ms, split over XCDs: %s, decoder made in %.0f ms, %s" % (off, sorted(fills)[1], " ".join("%.1f" % f for f in fills), sorted(walls)[1], dec.cluster_census(), t_create, dec.tier[-40:]), flush=True)
    dec.close()
