#!/bin/bash
# Copy what tools/round3_evidence.sh left under gpurun_out/prof${ROUND:-3}/ (and gpurun_out/sq_*, gpurun_out/pmc_fb) into profiles/ (tracked), named per round.
R=$(cd "$(dirname "$0")/.." && pwd)
P=$R/gpurun_out/prof${ROUND:-3}
O=$R/profiles
for c in config1 config2 config3 config3b config4; do
  [ -f $P/$c/trace/run_kernel_stats.csv ] && cp $P/$c/trace/run_kernel_stats.csv $O/r${ROUND:-3}_${c}_kernel_stats.csv
  [ -f $P/$c/trace_bench.json ] && tail -1 $P/$c/trace_bench.json > $O/r${ROUND:-3}_${c}_trace_bench.json
  for k in FETCH_SIZE WRITE_SIZE; do
    f=$P/$c/pmc_$k/run_counter_collection.csv
    [ -f $f ] && (head -1 $f; grep -E "viterbi_fill|fwdback_onchip|fwdback_estep" $f) > $O/r${ROUND:-3}_${c}_pmc_$k.csv
  done
done
tail -1 $P/bench_default.json > $O/r${ROUND:-3}_bench_default.json
[ -s $P/bench_config2_100k.json ] && tail -1 $P/bench_config2_100k.json > $O/r${ROUND:-3}_bench_config2_100k.json
[ -s $P/bench_config4_1M.json ] && tail -1 $P/bench_config4_1M.json > $O/r${ROUND:-3}_bench_config4_1M.json
for f in bulk_parity bulk_parity_config1 bulk_parity_config3a bulk_parity_config3b fuzz_sweep fuzz_fwdback; do [ -s $P/$f.txt ] && grep -v amdgpu.ids $P/$f.txt > $O/r${ROUND:-3}_$f.txt; done
python $R/tools/traffic_from_pmc.py $P/config2 viterbi_fill_tiera $O/r${ROUND:-3}_traffic_config2.json
python $R/tools/traffic_from_pmc.py $P/config1 viterbi_fill_tiera $O/r${ROUND:-3}_traffic_config1.json
python $R/tools/traffic_from_pmc.py $P/config3 viterbi_fill_tiera $O/r${ROUND:-3}_traffic_config3a.json
python $R/tools/traffic_from_pmc.py $P/config3b viterbi_fill_tiera $O/r${ROUND:-3}_traffic_config3b.json
python $R/tools/traffic_from_pmc.py $P/config4 fwdback_onchip $O/r${ROUND:-3}_traffic_config4.json
[ -s $R/gpurun_out/sq_c.json ] && cp $R/gpurun_out/sq_c.json $O/r${ROUND:-3}_sq_counters_tierC.json
[ -s $R/gpurun_out/sq_a.json ] && cp $R/gpurun_out/sq_a.json $O/r${ROUND:-3}_sq_counters_tierA.json
[ -s $R/gpurun_out/pmc_fb/fb_sq_counters.json ] && cp $R/gpurun_out/pmc_fb/fb_sq_counters.json $O/r${ROUND:-3}_sq_counters_fwdback.json
ls $O | grep r${ROUND:-3}_
