"""Two processes decode on clusters of work-groups (tier C) on the same card at the same time: wall time and fill-kernel time
per call (the launches share the card: about 100 ms each instead of 60; results unchanged).

    python tools/two_cluster_processes.py [calls]
"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "tests", "two_clusters_child.py")


def run_pair(options, calls, timeout=300):
    """-> the JSON lines of two children that started their decode calls together"""
    with tempfile.TemporaryDirectory() as tmp:
        start = os.path.join(tmp, "go")
        procs = [subprocess.Popen([sys.executable, CHILD, start, str(calls), options], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
                 for _ in range(2)]
        t0 = time.time()
        while sum(os.path.exists(start + ".%d" % p.pid) for p in procs) < 2:     # both have their model on the card
            if any(p.poll() is not None for p in procs) or time.time() - t0 > timeout:
                for p in procs:
                    if p.poll() is None:
                        p.kill()
                raise RuntimeError("a child did not come up: " + " | ".join(p.communicate()[1][-2000:] for p in procs))
            time.sleep(0.01)
        open(start, "w").close()
        # a child prints its line when its calls are through and keeps its arena until both are (the ".done" file)
        lines = [p.stdout.readline() for p in procs]
        open(start + ".done", "w").close()
        out = []
        for p, line in zip(procs, lines):
            so, se = p.communicate(timeout=timeout)
            if p.returncode != 0 or not line.strip():
                raise RuntimeError("child failed (exit %s): %s" % (p.returncode, se[-2000:]))
            out.append(json.loads(line))
        return out


if __name__ == "__main__":
    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    t0 = time.time()
    a, b = run_pair("tier=C,cluster=2,arena_fraction=0.3", calls)
    print("both done after %.2f s" % (time.time() - t0))
    for r in (a, b):
        print("   calls: " + " ".join("%.3f" % w for w in r["walls_s"]) + " s;  fill kernels " + " ".join("%.0f" % w for w in r["fill_ms"]) +
              " ms;  results " + ("equal" if len(r["digests"]) == 1 else "DIFFER"))
    print("   the two processes agree:", a["digests"] == b["digests"], flush=True)
