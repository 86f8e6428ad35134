#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
mkdir -p $R/gpurun_out/r3j
cd /tmp && export TMPDIR=/tmp
for v in defer nodefer; do
  OPT=""; [ $v = nodefer ] && OPT="--options defer_traceback=0"
  rm -rf $R/gpurun_out/r3j/trace_$v
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3j/trace_$v -o run --output-format csv -- python3 $R/bench.py --config 2 --steps 3 --warmup 1 --cpu-seconds 0 --timed-only $OPT > $R/gpurun_out/r3j/bench_$v.json 2> $R/gpurun_out/r3j/trace_$v.log
  echo "== $v"; python3 $R/tools/launch_times.py $R/gpurun_out/r3j/trace_$v | cut -c1-400
  python3 -c "
import json; j=json.load(open('$R/gpurun_out/r3j/bench_$v.json')); print('value %.4g frac %.3f launch %.2f ms ms/step %.1f tb %.1f' % (j['value'], j['roofline']['frac'], j['roofline']['avg_launch_ms'], j['ms_per_step'], j['roofline']['traceback_ms_per_step']))"
done
cd $R; timeout -k 10 300 python bench.py --config 2 --steps 3 --warmup 1 --cpu-seconds 6 --no-other-configs > gpurun_out/r3j/bench_parity.json 2> gpurun_out/r3j/bench_parity.err; echo "parity run rc=$?"; python3 -c "
import json; j=json.load(open('gpurun_out/r3j/bench_parity.json')); print('value %.4g frac %.3f parity reads %s' % (j['value'], j['roofline']['frac'], j['cpu_baseline']['parity_checked_reads']))"
