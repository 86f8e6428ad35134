#!/bin/bash
# Second evidence pass of round 2 (after the tier-C kernel moved to 512-thread work-groups): bench lines and kernel stats
# of the configurations that changed, PMC passes of every configuration with --timed-only, the 1M-pair E-step.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
P=$R/gpurun_out/prof2
mkdir -p $P
log() { echo "$(date +%T) $*" >> $P/progress.log; }
cd $R
for cfg in "1" "3 --variant b"; do
  name=config$(echo $cfg | tr -d ' -' | sed 's/variant//')
  log "bench $name"
  timeout -k 10 500 python bench.py --config $cfg > $P/bench_$name.json 2> $P/bench_$name.err || log "bench $name FAILED"
done
log "profiles"
bash tools/profile_round2.sh config1 --config 1 --reads 64 --steps 2
NO_PMC=1 bash tools/profile_round2.sh config3b --config 3 --variant b --reads 8 --steps 1
for c in config2 config3 config4; do mkdir -p $P/$c; done
for c in "config2 --config 2 --reads 4320" "config3 --config 3 --reads 2160" "config4 --config 4 --reads 125000"; do
  set -- $c; name=$1; shift
  for C in FETCH_SIZE WRITE_SIZE; do
    log "$name pmc $C"
    ( cd /tmp && export TMPDIR=/tmp && rm -rf $P/$name/pmc_$C && rocprofv3 --pmc $C -d $P/$name/pmc_$C -o run --output-format csv -- python3 $R/bench.py "$@" --cpu-seconds 0 --timed-only --steps 1 --warmup 0 > $P/$name/pmc_${C}_bench.json 2> $P/$name/pmc_$C.log ) || log "pmc failed"
  done
done
log "1M pairs"
timeout -k 10 900 python bench.py --config 4 --reads 1000000 --steps 2 --warmup 1 --cpu-seconds 0 > $P/bench_config4_1M.json 2> $P/bench_config4_1M.err || log "1M FAILED"
log "done b"
