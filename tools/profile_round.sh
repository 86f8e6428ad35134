#!/bin/bash
# Collect the rocprofv3 evidence kept under profiles/: kernel trace + stats of a shortened bench run,
# and the two HBM-byte counters in separate PMC passes (no trace flags together with --pmc).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o run --output-format csv -- python3 $R/bench.py --reads 4320 --steps 2 --warmup 1 --cpu-seconds 0 > $OUT/trace_bench.json 2> $OUT/trace.log || echo "trace failed"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C -d $OUT/pmc_$C -o run --output-format csv -- python3 $R/bench.py --reads 720 --steps 1 --warmup 0 --cpu-seconds 0 > $OUT/pmc_${C}_bench.json 2> $OUT/pmc_$C.log || echo "pmc $C failed"
done
find $OUT -name "*.csv" | head -20
