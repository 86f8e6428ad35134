#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r3h
ls -la dnastore_amd/kcache | head -20
for v in A B C; do
  case $v in
    A) export DNAS_TIERA_DEFS="";;
    B) unset DNAS_TIERA_DEFS;;
    C) unset DNAS_TIERA_DEFS; export DNAS_KCACHE_DIR=/tmp/kc_fresh;;
  esac
  timeout -k 10 400 python bench.py --config 1 --reads 64 --steps 1 --warmup 1 --cpu-seconds 3 --no-other-configs > gpurun_out/r3h/c1_$v.json 2> gpurun_out/r3h/c1_$v.err; echo "variant $v: bench rc=$?"; grep -c PARITY gpurun_out/r3h/c1_$v.err
done
ls -la /tmp/kc_fresh
