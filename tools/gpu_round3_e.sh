#!/bin/bash
set -o pipefail
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r3e
timeout -k 10 900 python -m pytest tests/test_gpu_fwdback.py -q -x > gpurun_out/r3e/pytest_fb.log 2>&1; echo "pytest fwdback rc=$?"
tail -6 gpurun_out/r3e/pytest_fb.log
timeout -k 10 600 python bench.py --config 4 --steps 2 --warmup 1 --cpu-seconds 4 > gpurun_out/r3e/c4.json 2> gpurun_out/r3e/c4.err; echo "bench c4 rc=$?"; tail -2 gpurun_out/r3e/c4.err
python - <<PY
import json
try:
    j=json.load(open("gpurun_out/r3e/c4.json")); r=j["roofline"]
    print("config 4: pairs/s %.3g value %.3g nt/s launch %.1f ms onchip %d streaming %d lse/s %.3g cpu %s" % (j["pairs_per_s"], j["value"], r["avg_launch_ms"], r["pairs_onchip"], r["pairs_streaming"], r["lse_ops_per_s"], (j.get("cpu_baseline") or {}).get("value")))
except Exception as e: print("no c4 line", e)
PY
run() {  # name, config, variant, reads, options
  timeout -k 10 400 python bench.py --config $2 --variant $3 --reads $4 --steps 2 --warmup 1 --cpu-seconds 0 --timed-only --options "$5" > gpurun_out/r3e/$1.json 2> gpurun_out/r3e/$1.err || { echo "$1 failed"; tail -3 gpurun_out/r3e/$1.err; return; }
  python - <<PY
import json
j=json.load(open("gpurun_out/r3e/$1.json")); r=j["roofline"]
print("$1: value %.3g frac %.3f launch %.2f ms rounds/col %.1f  %s" % (j["value"], r["frac"], r["avg_launch_ms"], r["rounds_per_column"], j["config"]["program"][:60]))
PY
}
run c1_t1024 1 a 64 "threads=1024"
run c1_g5 1 a 64 "cluster=5"
run c1_g6 1 a 64 "cluster=6"
run c1_g8_t1024 1 a 64 "cluster=8,threads=1024"
