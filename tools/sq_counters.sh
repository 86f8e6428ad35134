#!/bin/bash
# SQ counters of the fill kernel for ONE read alone on the GPU, one rocprofv3 --pmc pass per counter group (never with
# trace flags), summarised by tools/sq_summary.py.
#   bash tools/sq_counters.sh a    one ~490-nt read through s16h74l4c4 (tier A, one work-group)
#   bash tools/sq_counters.sh c    one ~980-nt read through the 46 670-state composite (tier C, one cluster)
# Output: gpurun_out/sq_<a|c>/<group>/run_counter_collection.csv, gpurun_out/sq_<a|c>.json
W=${1:-a}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/sq_$W
mkdir -p $OUT
if [ "$W" = "a" ]; then PROG="$R/tools/one_read.py 1"; else PROG="$R/tools/tierc_probe.py 2 1 - 1"; fi
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
         "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM"; do
  D=$OUT/$(echo $C | cut -d' ' -f1)
  rocprofv3 --pmc $C --kernel-include-regex "viterbi_fill" -d $D -o run --output-format csv -- python3 $PROG > $D.log 2>&1 || { echo "failed $C"; tail -3 $D.log; }
  echo "done $C" >> $OUT/progress.log
done
if [ "$W" = "a" ]; then WAVES=16; else WAVES=32; fi    # tier C, this machine: 4 work-groups of 8 waves
python3 $R/tools/sq_summary.py $OUT $W $WAVES > $R/gpurun_out/sq_$W.json
cat $R/gpurun_out/sq_$W.json
