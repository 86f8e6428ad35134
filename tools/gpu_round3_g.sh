#!/bin/bash
set -o pipefail
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r3g
timeout -k 10 600 python -m pytest tests/test_gpu_shipped_programs.py -q -x --tb=short -k "bench_machine" > gpurun_out/r3g/pytest1.log 2>&1; echo "pytest1 rc=$?"
tail -30 gpurun_out/r3g/pytest1.log
timeout -k 10 600 python -m pytest tests/test_gpu_shipped_programs.py -q -x --tb=short -k "two_cluster" > gpurun_out/r3g/pytest2.log 2>&1; echo "pytest2 rc=$?"
grep -v "^  File\|^$" gpurun_out/r3g/pytest2.log | head -30
