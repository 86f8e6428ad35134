#!/bin/bash
# Everything under profiles/r2_* in one GPU session: the bench line of every BASELINE configuration, rocprofv3 kernel
# stats and PMC byte counters per configuration, the 100k-read run and the bulk parity run.  Progress goes to
# gpurun_out/prof2/progress.log (a long run must keep writing).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
P=$R/gpurun_out/prof2
mkdir -p $P
log() { echo "$(date +%T) $*" >> $P/progress.log; }
cd $R
for cfg in "2" "1" "3" "3 --variant b" "4"; do
  name=config$(echo $cfg | tr -d ' -' | sed 's/variant//')
  log "bench $name"
  timeout -k 10 500 python bench.py --config $cfg > $P/bench_$name.json 2> $P/bench_$name.err || log "bench $name FAILED"
done
log "profiles"
bash tools/profile_round2.sh config2 --config 2 --reads 4320 --steps 2
bash tools/profile_round2.sh config1 --config 1 --reads 48 --steps 2
bash tools/profile_round2.sh config3 --config 3 --reads 2160 --steps 2
NO_PMC=1 bash tools/profile_round2.sh config3b --config 3 --variant b --reads 8 --steps 1
bash tools/profile_round2.sh config4 --config 4 --reads 125000 --steps 2
log "100k reads"
timeout -k 10 900 python bench.py --config 2 --reads 100000 --steps 1 --warmup 0 --cpu-seconds 0 > $P/bench_config2_100k.json 2> $P/bench_config2_100k.err || log "100k FAILED"
log "bulk parity"
timeout -k 10 600 python tools/bulk_parity.py 2000 14 > $P/bulk_parity.txt 2>&1 || log "bulk parity FAILED"
log "done"
