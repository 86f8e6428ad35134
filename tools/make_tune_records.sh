#!/bin/bash
# On a GPU box (from the repository root):  bash tools/make_tune_records.sh   -> gpurun_out/tune_records/tune_*.txt ;
# then, back home:  rm -f dnastore_amd/tune/tune_*.txt && cp gpurun_out/tune_records/tune_*.txt dnastore_amd/tune/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
rm -rf $R/gpurun_out/tune_records    # (gpurun merges into the local gpurun_out/: clear the local copy too before a new run)
python3 $R/tools/make_tune_records.py $R/gpurun_out/tune_records
