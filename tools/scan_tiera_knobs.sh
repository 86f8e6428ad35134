#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r3s
run() {
  DNAS_TIERA_DEFS="$2" timeout -k 10 300 python bench.py --config 2 --steps 3 --warmup 1 --cpu-seconds 0 --timed-only > gpurun_out/r3s/$1.json 2> gpurun_out/r3s/$1.err || { echo "$1 failed"; tail -2 gpurun_out/r3s/$1.err; return; }
  python3 -c "
import json; j=json.load(open('gpurun_out/r3s/$1.json')); print('%-22s value %.4g frac %.4f launch %.2f ms' % ('$1', j['value'], j['roofline']['frac'], j['roofline']['avg_launch_ms']))"
}
run base ""
run nth0 "-DDNAS_NT_H=0"
run nth2 "-DDNAS_NT_H=2"
run ntd0 "-DDNAS_NT_D=0"
run sleep0 "-DDNAS_SLEEP=0"
run sleep2 "-DDNAS_SLEEP=2"
run sleep4 "-DDNAS_SLEEP=4"
run base2 ""
