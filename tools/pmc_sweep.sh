cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --list-avail > $R/gpurun_out/avail.txt 2>&1
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN"; do
  D=$R/gpurun_out/pmc_$(echo $C | cut -d' ' -f1)
  rocprofv3 --pmc $C --kernel-include-regex "viterbi_fill" -d $D -o run -- python3 $R/tools/one_read.py 1 > $D.log 2>&1 || echo "failed $C"
done
ls $R/gpurun_out | head -30
