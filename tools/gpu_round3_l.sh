#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
mkdir -p $R/gpurun_out/r3l
cd $R
run() {
  timeout -k 10 300 python bench.py --config $3 --steps 3 --warmup 1 --cpu-seconds 0 --timed-only --options "$2" > gpurun_out/r3l/$1.json 2> gpurun_out/r3l/$1.err || { echo "$1 failed"; tail -2 gpurun_out/r3l/$1.err; return; }
  python3 -c "
import json; j=json.load(open('gpurun_out/r3l/$1.json')); print('%-28s value %.4g frac %.3f launch %.2f ms ms/step %.1f tb %.1f' % ('$1', j['value'], j['roofline']['frac'], j['roofline']['avg_launch_ms'], j['ms_per_step'], j['roofline']['traceback_ms_per_step']))"
}
run base "tb_threads=128" 2
run s696 "max_slots=696" 2
run s732 "max_slots=732" 2
run s744 "max_slots=744" 2
run s768 "max_slots=768" 2
run t192 "tb_threads=192" 2
run c3_base "tb_threads=128" 3
run c3_t512 "tb_threads=512" 3
run c3_t64 "tb_threads=64" 3
