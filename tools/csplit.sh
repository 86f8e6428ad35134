#!/bin/bash
# Where inside phase C does a column of the tier-A fill kernel spend its cycles?  tools/csplit.sh <reads> [rows per load group] [groups in flight]
# One short run per split point (tools/csplit_source.py), work-group 0's wave 0; results: profiles/experiments/r4_phase_c_timeline.txt
cd ${GRAFT_REPO_ROOT:-.}
N=${1:-1}; ROWS=${2:-4}; DEPTH=${3:-1}
mkdir -p /tmp/kc_csplit
python tools/csplit_source.py /tmp/kc_csplit/tiera_csplit.hip || exit 1
NG=$(( (7 + ROWS / 2 - 1) / (ROWS / 2) ))
for n in 1 2 3 4 $(seq 10 $((9 + NG))) 5 6 7; do
  DNAS_TIERA_SRC=/tmp/kc_csplit/tiera_csplit.hip DNAS_KCACHE_DIR=/tmp/kc_csplit timeout -k 10 120 python tools/stamp_gpu.py $N -DDNAS_CSPLIT=$n -DDNAS_CGROUP=$ROWS -DDNAS_CDEPTH=$DEPTH 2>&1 | grep "block0" | sed "s/^/$N reads, $ROWS rows x $DEPTH in flight, split $n: /"
done
