#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python - <<'PY'
import dnastore_amd as da
m = da.Machine.fromFile("tests/golden/ref_data/s16h74l4c4.json")
dec = da.ViterbiDecoder(m, da.MutatorParams.fromFlags(global_=True))
print(dec.tier[-120:])
PY
