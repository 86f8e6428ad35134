#!/bin/bash
# rocprofv3 evidence kept under profiles/ (round 3): per configuration one `--kernel-trace --stats` run of a shortened
# bench.py command (timed steps only), and the two HBM-byte counters in separate `--pmc` passes (never combined with trace
# flags).  Usage: tools/profile_round3.sh <name> <bench args...>   e.g.  config2 --config 2 --reads 4320
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
NAME=$1; shift
OUT=$R/gpurun_out/prof${ROUND:-3}/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "$(date +%T) == $NAME: kernel trace" >> $R/gpurun_out/prof${ROUND:-3}/progress.log
rm -rf $OUT/trace
rocprofv3 --kernel-trace --stats -d $OUT/trace -o run --output-format csv -- python3 $R/bench.py "$@" --cpu-seconds 0 --timed-only --steps 2 --warmup 1 > $OUT/trace_bench.json 2> $OUT/trace.log || echo "trace failed" >> $R/gpurun_out/prof${ROUND:-3}/progress.log
for C in FETCH_SIZE WRITE_SIZE; do
  echo "$(date +%T) == $NAME: pmc $C" >> $R/gpurun_out/prof${ROUND:-3}/progress.log
  rm -rf $OUT/pmc_$C
  rocprofv3 --pmc $C -d $OUT/pmc_$C -o run --output-format csv -- python3 $R/bench.py "$@" --cpu-seconds 0 --timed-only --steps 1 --warmup 0 > $OUT/pmc_${C}_bench.json 2> $OUT/pmc_$C.log || echo "pmc $C failed" >> $R/gpurun_out/prof${ROUND:-3}/progress.log
done
echo "$(date +%T) == $NAME done" >> $R/gpurun_out/prof${ROUND:-3}/progress.log
