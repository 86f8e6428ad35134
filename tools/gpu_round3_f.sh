#!/bin/bash
set -o pipefail
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r3f
timeout -k 10 300 python tools/tierc_small_batch.py 2>&1 | grep -v amdgpu.ids | tail -6
timeout -k 10 1500 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r3f/pytest.log 2>&1; echo "pytest rc=$?"
tail -22 gpurun_out/r3f/pytest.log
T0=$(date +%s)
timeout -k 10 900 python bench.py > gpurun_out/r3f/bench_default.json 2> gpurun_out/r3f/bench_default.err; echo "bench rc=$? in $(( $(date +%s) - T0 )) s"
python - <<PY
import json
j=json.load(open("gpurun_out/r3f/bench_default.json"))
r=j["roofline"]; print("headline: value %.4g frac %.3f ms/step %.1f cpu %s" % (j["value"], r["frac"], j["ms_per_step"], (j.get("cpu_baseline") or {}).get("value")))
for k,o in j.get("other_configs",{}).items():
    if "error" in o: print(k, "ERROR", o["error"][:200]); continue
    r=o["roofline"]; print("%s: value %.4g frac %.3g launch %.1f ms cpu %s parity %s  (%.0f s)" % (k, o["value"], r["frac"], r["avg_launch_ms"], (o.get("cpu_baseline") or {}).get("value"), (o.get("cpu_baseline") or {}).get("parity_checked_reads", (o.get("cpu_baseline") or {}).get("parity_checked_pairs")), o.get("bench_seconds",0)))
print("other configs took %.0f s" % j.get("other_configs_seconds",0))
PY
