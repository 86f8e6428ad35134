#!/bin/bash
# every published table that needs no LDPC wrapper, fitted models: gpurun_out/tables/<stem>.txt
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/tables
for t in "$@"; do
  timeout -k 10 500 python tools/accuracy_table.py $t > gpurun_out/tables/$t.txt 2> gpurun_out/tables/$t.err || { echo "$t FAILED"; tail -3 gpurun_out/tables/$t.err; }
  cut -c1-100 gpurun_out/tables/$t.txt
done
