#!/bin/bash
# Copy what tools/round2_evidence.sh left under gpurun_out/prof2/ into profiles/ (tracked), named per round.
R=$(cd "$(dirname "$0")/.." && pwd)
P=$R/gpurun_out/prof2
O=$R/profiles
for c in config1 config2 config3 config3b config4; do
  [ -f $P/$c/trace/run_kernel_stats.csv ] && cp $P/$c/trace/run_kernel_stats.csv $O/r2_${c}_kernel_stats.csv
  [ -f $P/$c/trace_bench.json ] && tail -1 $P/$c/trace_bench.json > $O/r2_${c}_trace_bench.json
  for k in FETCH_SIZE WRITE_SIZE; do
    f=$P/$c/pmc_$k/run_counter_collection.csv
    [ -f $f ] && (head -1 $f; grep -E "viterbi_fill|fwdback_onchip|fwdback_estep" $f) > $O/r2_${c}_pmc_$k.csv
  done
done
for f in $P/bench_config*.json; do tail -1 $f > $O/r2_$(basename $f); done
[ -f $P/bench_config4_1M.json ] && tail -1 $P/bench_config4_1M.json > $O/r2_bench_config4_1M.json
cp $P/bulk_parity.txt $O/r2_bulk_parity.txt
python $R/tools/traffic_from_pmc.py $P/config2 viterbi_fill_tiera $O/r2_traffic_config2.json
python $R/tools/traffic_from_pmc.py $P/config1 viterbi_fill_tiera $O/r2_traffic_config1.json
python $R/tools/traffic_from_pmc.py $P/config3 viterbi_fill_tiera $O/r2_traffic_config3a.json
python $R/tools/traffic_from_pmc.py $P/config4 fwdback_onchip_kernel $O/r2_traffic_config4.json
