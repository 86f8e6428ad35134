#!/bin/bash
# Where inside a sweep does a cluster member (tier C, configs[1]) spend its cycles?  tools/csweep.sh <reads> [options]
# (tools/csweep_source.py; the figure printed as "cluster vote" is cycles / 100 per column from sweep start to the split point, summed over the column's sweeps)
cd ${GRAFT_REPO_ROOT:-.}
N=${1:-64}; OPT=${2:--}
mkdir -p /tmp/kc_csweep
python tools/csweep_source.py /tmp/kc_csweep/tiera_csweep.hip || exit 1
for n in 1 2 3 4 5; do
  DNAS_TIERA_SRC=/tmp/kc_csweep/tiera_csweep.hip DNAS_KCACHE_DIR=/tmp/kc_csweep DNAS_TIERA_DEFS="-DDNAS_STAMP:-DDNAS_SSPLIT=$n" timeout -k 10 200 python tools/tierc_probe.py 2 $N $OPT 1 2>&1 | grep "block 0" | sed "s/^/split $n: /" | cut -c1-260
done
