#!/bin/bash
# GPU box: parity suite, then short benches of every Viterbi configuration (timed steps only)
set -o pipefail
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r3a
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3a/pytest.log
tail -5 gpurun_out/r3a/pytest.log
for cfg in "2 a 2880" "1 a 64" "3 a 2085" "3 b 16"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --config $1 --variant $2 --reads $3 --steps 2 --warmup 1 --cpu-seconds 0 --timed-only > gpurun_out/r3a/bench_$1$2.json 2> gpurun_out/r3a/bench_$1$2.err; echo "bench $1$2 rc=$?"
  python - <<PY
import json
try:
    j=json.load(open("gpurun_out/r3a/bench_$1$2.json"))
    r=j["roofline"]; print("config $1$2: value %.3g nt/s frac %.3f launch %.2f ms rounds/col %.1f tier %s" % (j["value"], r["frac"], r["avg_launch_ms"], r["rounds_per_column"], r["tier"]))
except Exception as e: print("config $1$2: no line", e)
PY
done
