#!/bin/bash
# phase C's length with fewer bytes through the CU's memory path (timing only, results are wrong): tools/csplit_whatif.sh <reads>
cd ${GRAFT_REPO_ROOT:-.}
N=${1:-1}; mkdir -p /tmp/kc_csplit
python tools/csplit_source.py /tmp/kc_csplit/tiera_csplit.hip || exit 1
for v in "" "-DDNAS_EXP_NO_D" "-DDNAS_EXP_HCOLS=2" "-DDNAS_EXP_HCOLS=1" "-DDNAS_EXP_HCOLS=1 -DDNAS_EXP_NO_D"; do
  DNAS_TIERA_SRC=/tmp/kc_csplit/tiera_csplit.hip DNAS_KCACHE_DIR=/tmp/kc_csplit timeout -k 10 120 python tools/stamp_gpu.py $N -DDNAS_CSPLIT=5 $v 2>&1 | grep "block0" | sed "s/^/$N reads [$v]: /"
done
