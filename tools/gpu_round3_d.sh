#!/bin/bash
set -o pipefail
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r3d
timeout -k 10 900 python -m pytest tests/test_gpu_tier_c.py tests/test_gpu_fuzz_machines.py tests/test_gpu_checkpoint.py -q > gpurun_out/r3d/pytest.log 2>&1; echo "pytest rc=$?"
tail -6 gpurun_out/r3d/pytest.log
run() {  # name, config, variant, reads, members-divisor, defs
  DNAS_TIERA_DEFS="$6" timeout -k 10 400 python bench.py --config $2 --variant $3 --reads $4 --steps 2 --warmup 1 --cpu-seconds 4 --no-other-configs > gpurun_out/r3d/$1.json 2> gpurun_out/r3d/$1.err || { echo "$1 failed"; tail -3 gpurun_out/r3d/$1.err; return; }
  python - <<PY
import json
j=json.load(open("gpurun_out/r3d/$1.json")); r=j["roofline"]
print("$1: value %.3g frac %.3f launch %.2f ms sweeps/col/member %.1f parity %s" % (j["value"], r["frac"], r["avg_launch_ms"], r["rounds_per_column"]/$5, (j.get("cpu_baseline") or {}).get("parity_checked_reads")))
PY
}
run c1_split 1 a 64 4 ""
run c1_p1 1 a 64 4 "-DDNAS_POLLS=1"
run c3b_split 3 b 16 21 ""
