#!/bin/bash
# Kernel development: compile csrc/viterbi_tiera.hip offline with the options of a cached specialisation and keep the ISA.
#   DNAS_JIT_DUMP=1 makes the library write <code object>.defs next to every code object it compiles;
#   tools/compile_kernel.sh <file.defs> <output dir> [extra -D options]  ->  <output dir>/*.s, resource usage on stdout
set -e
DEFS=$1; OUT=$2; shift 2
SRC=${DNAS_TIERA_SRC:-$(cd "$(dirname "$0")/.." && pwd)/dnastore_amd/csrc/viterbi_tiera.hip}
mkdir -p "$OUT"
ARGS=()
while IFS= read -r line; do [ -n "$line" ] && ARGS+=("$line"); done < "$DEFS"
cd "$OUT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 --cuda-device-only -save-temps -Rpass-analysis=kernel-resource-usage \
  "${ARGS[@]}" "$@" -c "$SRC" -o kernel.o 2>&1 | grep -E "remark|error" | sed 's/.*remark: //'
