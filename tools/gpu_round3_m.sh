#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
echo "== 1 read"; timeout -k 10 200 python tools/stamp_gpu.py 1 2>&1 | grep -v amdgpu.ids | tail -4
echo "== 715 reads"; timeout -k 10 200 python tools/stamp_gpu.py 715 2>&1 | grep -v amdgpu.ids | tail -4
