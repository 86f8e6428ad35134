#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r3n
one() {
  timeout -k 10 600 python bench.py --config 4 --steps 3 --warmup 1 --cpu-seconds 0 --timed-only > gpurun_out/r3n/c4_$1.json 2> gpurun_out/r3n/c4_$1.err; echo "bench $1 rc=$?"
  python3 -c "
import json
j=json.load(open('gpurun_out/r3n/c4_$1.json')); r=j['roofline']
print('$1: pairs/s %.4g launch %.1f ms' % (j['pairs_per_s'], r['avg_launch_ms']))"
}
cp dnastore_amd/libdnastore_amd.so /tmp/lib_keep.so
for v in "2 8" "4 16" "3 16" "4 20"; do
  set -- $v
  touch dnastore_amd/csrc/fwdback_onchip.hip dnastore_amd/csrc/fwdback_runtime.hip
  make -s -C dnastore_amd/csrc EXTRA="-DDNAS_FB_MIN_WAVES=$1 -DDNAS_FB_WAVES_PER_CU=$2" ../libdnastore_amd.so > gpurun_out/r3n/make_$1_$2.log 2>&1 || { echo "make $v failed"; tail -3 gpurun_out/r3n/make_$1_$2.log; continue; }
  one w$1x$2
done
cp /tmp/lib_keep.so dnastore_amd/libdnastore_amd.so
