#!/bin/bash
set -o pipefail
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r3c
run() {  # name, config, variant, reads, members-divisor, defs
  DNAS_TIERA_DEFS="$6" timeout -k 10 300 python bench.py --config $2 --variant $3 --reads $4 --steps 2 --warmup 1 --cpu-seconds 0 --timed-only > gpurun_out/r3c/$1.json 2> gpurun_out/r3c/$1.err || { echo "$1 failed"; tail -3 gpurun_out/r3c/$1.err; return; }
  python - <<PY
import json
j=json.load(open("gpurun_out/r3c/$1.json")); r=j["roofline"]
print("$1: value %.3g frac %.3f launch %.2f ms sweeps/col/member %.1f tb %.1f ms" % (j["value"], r["frac"], r["avg_launch_ms"], r["rounds_per_column"]/$5, r["traceback_ms_per_step"]))
PY
}
run c1_p1_cg4 1 a 64 4 $'-DDNAS_POLLS=1\n-DDNAS_POLL_LAG=1\n-DDNAS_CGROUP=4'
run c1_p1_cg8 1 a 64 4 $'-DDNAS_POLLS=1\n-DDNAS_POLL_LAG=1\n-DDNAS_CGROUP=8'
run c1_p1_cg10 1 a 64 4 $'-DDNAS_POLLS=1\n-DDNAS_POLL_LAG=1\n-DDNAS_CGROUP=10'
run c1_split_cg8 1 a 64 4 $'-DDNAS_POLLS=1\n-DDNAS_POLL_LAG=1\n-DDNAS_POLL_SPLIT=1\n-DDNAS_CGROUP=8'
run c1_p2_cg8 1 a 64 4 $'-DDNAS_POLLS=2\n-DDNAS_POLL_LAG=1\n-DDNAS_CGROUP=8'
run c2_cg4 2 a 2880 1 $'-DDNAS_CGROUP=4'
run c2_cg6 2 a 2880 1 $'-DDNAS_CGROUP=6'
run c2_cg8 2 a 2880 1 $'-DDNAS_CGROUP=8'
run c3a_cg4 3 a 2085 1 $'-DDNAS_CGROUP=4'
run c3a_cg6 3 a 2085 1 $'-DDNAS_CGROUP=6'
run c3a_cg8 3 a 2085 1 $'-DDNAS_CGROUP=8'
run c3b_p1_cg8 3 b 16 21 $'-DDNAS_POLLS=1\n-DDNAS_POLL_LAG=1\n-DDNAS_CGROUP=8'
run c3b_p1_cg4 3 b 16 21 $'-DDNAS_POLLS=1\n-DDNAS_POLL_LAG=1\n-DDNAS_CGROUP=4'
timeout -k 10 1200 python -m pytest tests -m gpu -q > gpurun_out/r3c/pytest.log 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/r3c/pytest.log
