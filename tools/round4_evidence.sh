#!/bin/bash
# Everything under profiles/r4_* in one GPU session (final kernels and tuning records): the default bench run (the driver's command),
# rocprofv3 kernel stats and PMC byte counters per configuration, SQ counters, the 100k-read and 1M-pair runs, bulk parity, fuzz sweep.
# Progress: gpurun_out/prof4/progress.log.  Then, back home: ROUND=4 bash tools/collect_profiles.sh
export ROUND=4
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
P=$R/gpurun_out/prof4
mkdir -p $P
log() { echo "$(date +%T) $*" | tee -a $P/progress.log; }
cd $R
PART=${1:-all}
if [ $PART = all ] || [ $PART = a ]; then
log "default bench"
timeout -k 10 900 python bench.py > $P/bench_default.json 2> $P/bench_default.err || log "default bench FAILED"
log "profiles"
bash tools/profile_round3.sh config2 --config 2 --reads 4320
bash tools/profile_round3.sh config1 --config 1 --reads 64
bash tools/profile_round3.sh config3 --config 3 --reads 2400
bash tools/profile_round3.sh config3b --config 3 --variant b --reads 12
bash tools/profile_round3.sh config4 --config 4 --reads 125000
fi
if [ $PART = all ] || [ $PART = a2 ]; then
log "100k reads"
timeout -k 10 900 python bench.py --config 2 --reads 100000 --steps 1 --warmup 1 --cpu-seconds 0 --no-other-configs > $P/bench_config2_100k.json 2> $P/bench_config2_100k.err || log "100k FAILED"
log "1M pairs"
timeout -k 10 900 python bench.py --config 4 --reads 1000000 --steps 2 --warmup 1 --cpu-seconds 0 > $P/bench_config4_1M.json 2> $P/bench_config4_1M.err || log "1M FAILED"
fi
if [ $PART = all ] || [ $PART = b ]; then
log "sq counters tier C"
bash tools/sq_counters.sh c > $P/sq_c.log 2>&1
log "sq counters tier A"
bash tools/sq_counters.sh a > $P/sq_a.log 2>&1
log "sq counters forward-backward"
bash tools/pmc_fwdback.sh > $P/sq_fb.log 2>&1
log "bulk parity"
timeout -k 10 600 python tools/bulk_parity.py 2000 14 > $P/bulk_parity.txt 2>&1 || log "bulk parity FAILED"
timeout -k 10 600 python tools/bulk_parity.py 96 14 --config 1 > $P/bulk_parity_config1.txt 2>&1 || log "bulk parity config1 FAILED"
timeout -k 10 900 python tools/bulk_parity.py 24 14 --config 3 --variant b > $P/bulk_parity_config3b.txt 2>&1 || log "bulk parity config3b FAILED"
timeout -k 10 600 python tools/bulk_parity.py 700 14 --config 3 > $P/bulk_parity_config3a.txt 2>&1 || log "bulk parity config3a FAILED"
log "fuzz sweep"
DNAS_PLAN_PROXY_MIN=2 timeout -k 10 900 python tools/fuzz_sweep.py 60 71000 > $P/fuzz_sweep.txt 2>&1 || log "fuzz sweep FAILED"
timeout -k 10 300 python tools/fuzz_fwdback.py 60 > $P/fuzz_fwdback.txt 2>&1 || log "fuzz fwdback FAILED"
fi
log "done"
