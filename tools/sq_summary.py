"""Sum the rocprofv3 --pmc csv files of tools/sq_counters.sh over the dispatches of the fill kernel (the last dispatch is the
measured call; earlier ones are the probe's warm-up) and print them per wave and lattice column.
  python tools/sq_summary.py gpurun_out/sq_a a [working waves]"""
import csv, glob, json, os, sys
root, which = sys.argv[1], sys.argv[2]
tot, meta = {}, {}
for f in sorted(glob.glob(os.path.join(root, "*", "run_counter_collection.csv"))):
    rows = [r for r in csv.DictReader(open(f)) if "viterbi_fill" in r["Kernel_Name"]]
    if not rows:
        continue
    last = max(int(r["Dispatch_Id"]) for r in rows)
    for r in rows:
        if int(r["Dispatch_Id"]) != last:
            continue
        tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        meta = {"grid": int(r["Grid_Size"]), "workgroup": int(r["Workgroup_Size"]), "scratch_per_lane": int(r["Scratch_Size"]),
                "vgpr": int(r["VGPR_Count"]), "sgpr": int(r["SGPR_Count"]), "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])}
waves = tot.get("SQ_WAVES", 0) or (meta.get("grid", 0) // 64)
launched = waves
if len(sys.argv) > 3:      # tier C launches whole groups of 8 clusters: the waves of the clusters without a read leave at once
    waves = float(sys.argv[3])
out = {"workload": "one ~490-nt read through s16h74l4c4 (tier A: one work-group)" if which == "a" else
       "one ~980-nt read through flusher*mixradar6*l4c4, 46 670 states (tier C: one cluster of work-groups)",
       "command": "tools/sq_counters.sh %s: rocprofv3 --pmc <group> --kernel-include-regex viterbi_fill -- python3 tools/%s" % (
           which, "one_read.py 1" if which == "a" else "tierc_probe.py 2 1 - 1"),
       "dispatch": meta, "waves_launched": launched, "waves_working": waves, "counters": tot}
if "SQ_BUSY_CYCLES" in tot and "SQ_WAVE_CYCLES" in tot and waves:
    out["derived"] = {
        "valu_issue_share_of_wave_cycles": tot.get("SQ_ACTIVE_INST_VALU", 0) / tot["SQ_WAVE_CYCLES"],
        "lds_issue_share_of_wave_cycles": tot.get("SQ_ACTIVE_INST_LDS", 0) / tot["SQ_WAVE_CYCLES"],
        "wait_any_share_of_wave_cycles": tot.get("SQ_WAIT_ANY", 0) / tot["SQ_WAVE_CYCLES"],
        "valu_per_wave": tot.get("SQ_INSTS_VALU", 0) / waves, "salu_per_wave": tot.get("SQ_INSTS_SALU", 0) / waves,
        "lds_per_wave": tot.get("SQ_INSTS_LDS", 0) / waves, "vmem_rd_per_wave": tot.get("SQ_INSTS_VMEM_RD", 0) / waves,
        "vmem_wr_per_wave": tot.get("SQ_INSTS_VMEM_WR", 0) / waves,
    }
print(json.dumps(out, indent=1))
