#!/bin/bash
# rocprofv3 evidence kept under profiles/ (round 2): per configuration one `--kernel-trace --stats` run of a shortened
# bench.py command, and for the Viterbi configurations the two HBM-byte counters in separate `--pmc` passes (never
# combined with trace flags).  Usage: tools/profile_round2.sh <name> <bench args...>   e.g.  c2 --config 2 --reads 4320
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
NAME=$1; shift
OUT=$R/gpurun_out/prof2/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "== $NAME: kernel trace" >> $R/gpurun_out/prof2/progress.log
rocprofv3 --kernel-trace --stats -d $OUT/trace -o run --output-format csv -- python3 $R/bench.py "$@" --cpu-seconds 0 > $OUT/trace_bench.json 2> $OUT/trace.log || echo "trace failed" >> $R/gpurun_out/prof2/progress.log
if [ -z "$NO_PMC" ]; then
  for C in FETCH_SIZE WRITE_SIZE; do
    echo "== $NAME: pmc $C" >> $R/gpurun_out/prof2/progress.log
    rocprofv3 --pmc $C -d $OUT/pmc_$C -o run --output-format csv -- python3 $R/bench.py "$@" --cpu-seconds 0 --timed-only --steps 1 --warmup 0 > $OUT/pmc_${C}_bench.json 2> $OUT/pmc_$C.log || echo "pmc $C failed" >> $R/gpurun_out/prof2/progress.log
  done
fi
find $OUT -name "*.csv" >> $R/gpurun_out/prof2/progress.log
echo "== $NAME done" >> $R/gpurun_out/prof2/progress.log
