#!/bin/bash
# GPU box: tier-C poll variants on configs[1] (64 reads), then the parity suite
set -o pipefail
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r3b
run() {  # name, defs
  DNAS_TIERA_DEFS="$2" timeout -k 10 200 python bench.py --config 1 --reads 64 --steps 2 --warmup 1 --cpu-seconds 0 --timed-only > gpurun_out/r3b/c1_$1.json 2> gpurun_out/r3b/c1_$1.err || { echo "$1 failed"; tail -3 gpurun_out/r3b/c1_$1.err; return; }
  python - <<PY
import json
j=json.load(open("gpurun_out/r3b/c1_$1.json")); r=j["roofline"]
print("$1: frac %.3f launch %.2f ms sweeps/col/member %.1f" % (r["frac"], r["avg_launch_ms"], r["rounds_per_column"]/4))
PY
}
run p4l1 $'-DDNAS_POLLS=4\n-DDNAS_POLL_LAG=1'
run p4l2 $'-DDNAS_POLLS=4\n-DDNAS_POLL_LAG=2'
run p2l1 $'-DDNAS_POLLS=2\n-DDNAS_POLL_LAG=1'
run p2l2 $'-DDNAS_POLLS=2\n-DDNAS_POLL_LAG=2'
run p1l1 $'-DDNAS_POLLS=1\n-DDNAS_POLL_LAG=1'
run p7l1 $'-DDNAS_POLLS=7\n-DDNAS_POLL_LAG=1'
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3b/pytest.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r3b/pytest.log
