#!/bin/bash
# How many reads per fill launch?  Runs bench.py with DNAS_MAX_SLOTS = each argument on exactly 20 full
# launches and prints nt/s and the average launch time (the scan behind the 720 of runtime.hip).
#   bash tools/slots_scan.sh 464 500 510 696 720 744
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for s in "$@"; do
  DNAS_MAX_SLOTS=$s python $R/bench.py --reads $((s * 14)) --steps 2 --warmup 1 --cpu-seconds 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('%4d reads per launch: %8.0f nt/s, %6.2f ms per launch, %.2f reads/ms' % ($s, d['value'], r['avg_launch_ms'], $s / r['avg_launch_ms']))"
done
