#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r3n
timeout -k 10 600 python -m pytest tests/test_gpu_fwdback.py -q 2>&1 | tail -3
timeout -k 10 600 python bench.py --config 4 --steps 3 --warmup 1 --cpu-seconds 4 > gpurun_out/r3n/c4.json 2> gpurun_out/r3n/c4.err; echo "bench rc=$?"; tail -2 gpurun_out/r3n/c4.err | cut -c1-200
python3 -c "
import json
j=json.load(open('gpurun_out/r3n/c4.json')); r=j['roofline']
print('pairs/s %.4g launch %.1f ms onchip %d streaming %d parity %s' % (j['pairs_per_s'], r['avg_launch_ms'], r['pairs_onchip'], r['pairs_streaming'], (j.get('cpu_baseline') or {}).get('parity_checked_pairs')))"
