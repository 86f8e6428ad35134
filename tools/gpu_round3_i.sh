#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r3i
export DNAS_TIERA_DEFS="-DDNAS_STAMP"
echo "== 512 threads, 64 reads"; timeout -k 10 300 python tools/tierc_probe.py 2 64 - 2 2>&1 | grep -v amdgpu.ids | tail -3
echo "== 512 threads, 1 read"; timeout -k 10 300 python tools/tierc_probe.py 2 1 - 2 2>&1 | grep -v amdgpu.ids | tail -2
echo "== 1024 threads, 48 reads"; timeout -k 10 300 python tools/tierc_probe.py 2 48 threads=1024 2 2>&1 | grep -v amdgpu.ids | tail -3
echo "== 1024 threads, 1 read"; timeout -k 10 300 python tools/tierc_probe.py 2 1 threads=1024 2 2>&1 | grep -v amdgpu.ids | tail -2
echo "== 4b 8 reads"; timeout -k 10 400 python tools/tierc_probe.py 4b 8 - 2 2>&1 | grep -v amdgpu.ids | tail -2
