#!/bin/bash
# SQ counters of the tier-C fill kernel for ONE ~1 kb read through the 46 670-state composite (one cluster alone on the
# GPU), one rocprofv3 --pmc pass per counter group.  Output: gpurun_out/pmc_tierc/<group>/...csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_tierc
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM"; do
  D=$R/gpurun_out/pmc_tierc/$(echo $C | cut -d' ' -f1)
  rocprofv3 --pmc $C --kernel-include-regex "viterbi_fill" -d $D -o run --output-format csv -- python3 $R/tools/tierc_probe.py 2 1 - 1 > $D.log 2>&1 || echo "failed $C"
  echo "done $C" >> $R/gpurun_out/pmc_tierc/progress.log
done
