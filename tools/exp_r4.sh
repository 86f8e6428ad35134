#!/bin/bash
# round 4: A/B runs of fill-kernel variants.  usage: tools/exp_r4.sh <out-subdir> ; reads lines "name|config|variant|reads|env assignments" from stdin
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/$1; mkdir -p $OUT
while IFS='|' read -r name config variant reads envs; do
  [ -z "$name" ] && continue
  env $envs timeout -k 10 400 python bench.py --config $config --variant $variant --reads $reads --steps 2 --warmup 1 --cpu-seconds 0 --timed-only > $OUT/$name.json 2> $OUT/$name.err || { echo "$name FAILED"; tail -3 $OUT/$name.err; continue; }
  python3 -c "
import json; j=json.load(open('$OUT/$name.json')); r=j['roofline']; print('%-28s value %.4g frac %.4f launch %.2f ms sweeps/col %.2f tb %.1f ms  %s' % ('$name', j['value'], r['frac'], r['avg_launch_ms'], r['rounds_per_column'], r['traceback_ms_per_step'], j['config']['program'][:60]))"
done
