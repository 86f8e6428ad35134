#!/bin/bash
# The CPU tests (-m "not gpu": host C++ of the library -- machine, composer, planner, exact coder, readers -- and the oracle) under
# AddressSanitizer + UndefinedBehaviorSanitizer, plus the planner fuzz (random machines through the tier-A / tier-C planner).
#   bash tools/run_asan.sh [pytest args]      -> exit code of pytest; the log goes to stdout
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
make -s -C $R/dnastore_amd/csrc asan
make -s -C $R/oracle asan/liboracle.so
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
export LD_PRELOAD=$RT
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=0:detect_odr_violation=0
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export DNAS_LIBRARY=$R/dnastore_amd/asan/libdnastore_amd.so DNAS_ORACLE_LIBRARY=$R/oracle/asan/liboracle.so
export DNAS_KCACHE_DIR=${DNAS_KCACHE_DIR:-$R/dnastore_amd/kcache}
cd $R
python -m pytest tests -m "not gpu" -x -q -p no:cacheprovider "$@"
