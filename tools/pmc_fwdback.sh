#!/bin/bash
# SQ counters of the on-chip forward-backward kernel for one E-step over 125 000 pairs (bench.py --config 4), one rocprofv3 --pmc
# pass per counter group.  Output: gpurun_out/pmc_fb/<group>/...csv, then tools/fb_sq_summary.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_fb
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM"; do
  D=$R/gpurun_out/pmc_fb/$(echo $C | cut -d' ' -f1)
  rocprofv3 --pmc $C --kernel-include-regex "fwdback_onchip" -d $D -o run --output-format csv -- python3 $R/bench.py --config 4 --steps 1 --warmup 0 --cpu-seconds 0 --timed-only > $D.log 2>&1 || echo "failed $C"
  echo "done $C" >> $R/gpurun_out/pmc_fb/progress.log
done
python3 - <<PY
import csv, glob, json, os
root = "$R/gpurun_out/pmc_fb"
tot, meta = {}, {}
for f in sorted(glob.glob(os.path.join(root, "*", "run_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        if "fwdback_onchip" not in r["Kernel_Name"]: continue
        tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        meta = {"grid": int(r["Grid_Size"]), "workgroup": int(r["Workgroup_Size"]), "vgpr": int(r["VGPR_Count"]), "agpr": r.get("Accum_VGPR_Count"), "lds": int(r["LDS_Block_Size"]), "scratch": int(r["Scratch_Size"]),
                "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])}
out = {"workload": "one E-step over 125 000 pairs (bench.py --config 4 --steps 1 --warmup 0 --timed-only), all wavefront kernels (8x16: 93 % of the pairs)", "dispatch": meta, "counters": tot}
if "SQ_WAVE_CYCLES" in tot:
    pairs = 125000.0
    out["derived"] = {"valu_per_pair": tot.get("SQ_INSTS_VALU", 0) / pairs * 1.0, "salu_per_pair": tot.get("SQ_INSTS_SALU", 0) / pairs, "lds_per_pair": tot.get("SQ_INSTS_LDS", 0) / pairs,
                      "vmem_rd_per_pair": tot.get("SQ_INSTS_VMEM_RD", 0) / pairs, "vmem_wr_per_pair": tot.get("SQ_INSTS_VMEM_WR", 0) / pairs, "smem_per_pair": tot.get("SQ_INSTS_SMEM", 0) / pairs,
                      "valu_issue_share_of_wave_cycles": tot.get("SQ_ACTIVE_INST_VALU", 0) / tot["SQ_WAVE_CYCLES"],
                      "wait_any_share_of_wave_cycles": tot.get("SQ_WAIT_ANY", 0) / tot["SQ_WAVE_CYCLES"], "wait_inst_any_share": tot.get("SQ_WAIT_INST_ANY", 0) / tot["SQ_WAVE_CYCLES"]}
json.dump(out, open(os.path.join(root, "fb_sq_counters.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
