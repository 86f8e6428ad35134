#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for opt in "threads=1024" "threads=1024,cluster=8" "threads=1024,cluster=16" "threads=1024,cluster=32,cluster_spread=0" "threads=512,cluster=8" "threads=512,cluster=16"; do
  echo "== $opt"; DNAS_TIERA_DEFS=-DDNAS_MARKS=0 timeout -k 10 200 python tools/tierc_probe.py 2 1 "$opt" 2 2>&1 | grep -E "model:|reads," | cut -c1-260
done
