"""profiles/r2_traffic_<name>.json from the two rocprofv3 --pmc passes of tools/profile_round2.sh: HBM bytes of the dominant
kernel per unit of work (lattice column, or alignment pair), with the gfx950 correction MI355X_MICROARCH.md prescribes
(FETCH_SIZE counts 64 B per 128-B request of a wide coalesced streaming read: doubled; WRITE_SIZE is exact for 16-B-per-lane
stores and float atomics).  Usage: traffic_from_pmc.py <prof dir of one name> <kernel name> <out json>"""
import csv, glob, json, os, sys

d, kernel, out = sys.argv[1], sys.argv[2], sys.argv[3]


def counter(name):
    rows = []
    for f in glob.glob(os.path.join(d, "pmc_" + name, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith(kernel) and r["Counter_Name"] == name:
                rows.append(r)
    return rows


fetch, write = counter("FETCH_SIZE"), counter("WRITE_SIZE")
bench = json.loads(open(os.path.join(d, "pmc_FETCH_SIZE_bench.json")).read().strip().splitlines()[-1])
res = {"command": "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> -- python3 bench.py <args> --steps 1 --warmup 0 --cpu-seconds 0 --timed-only (one pass per counter; tools/profile_round3.sh)",
       "workload": bench["config"]["workload"], "kernel": kernel, "launches": len(fetch),
       "FETCH_SIZE_kb": sum(float(r["Counter_Value"]) for r in fetch), "WRITE_SIZE_kb": sum(float(r["Counter_Value"]) for r in write),
       "scratch_bytes_per_lane": fetch[0]["Scratch_Size"] if fetch else None, "vgprs": fetch[0]["VGPR_Count"] if fetch else None,
       "gfx950_fetch_correction": 2.0}
rf = bench["roofline"]
if "algorithmic_bytes_per_column" in rf:
    cols = rf["algorithmic_bytes_per_launch"] / rf["algorithmic_bytes_per_column"] * max(len(fetch), 1)
    res.update(tier=rf["tier"], columns=cols, fetch_bytes_per_column_raw=res["FETCH_SIZE_kb"] * 1024 / cols,
               write_bytes_per_column=res["WRITE_SIZE_kb"] * 1024 / cols,
               hbm_bytes_per_column_corrected=(2 * res["FETCH_SIZE_kb"] + res["WRITE_SIZE_kb"]) * 1024 / cols,
               algorithmic_bytes_per_column=rf["algorithmic_bytes_per_column"])
    res["traffic_over_algorithmic"] = res["hbm_bytes_per_column_corrected"] / res["algorithmic_bytes_per_column"]
else:
    pairs = bench["config"]["pairs_per_gpu"] * max(len(fetch), 1)      # every launch of the run processed the whole shard
    res.update(pairs=pairs, fetch_bytes_per_pair_raw=res["FETCH_SIZE_kb"] * 1024 / pairs, write_bytes_per_pair=res["WRITE_SIZE_kb"] * 1024 / pairs,
               hbm_bytes_per_pair_corrected=(2 * res["FETCH_SIZE_kb"] + res["WRITE_SIZE_kb"]) * 1024 / pairs,
               algorithmic_bytes_per_pair=rf["algorithmic_bytes_per_launch"] / bench["config"]["pairs_per_gpu"])
    res["traffic_over_algorithmic"] = res["hbm_bytes_per_pair_corrected"] / res["algorithmic_bytes_per_pair"]
    res["traffic_over_algorithmic_uncorrected"] = (res["fetch_bytes_per_pair_raw"] + res["write_bytes_per_pair"]) / res["algorithmic_bytes_per_pair"]
    res["note"] = ("the x2 FETCH_SIZE correction is calibrated for wide (16 B per lane) streaming reads; this kernel reads 1-B bases and 4-B guide "
                   "columns, for which the counter is uncalibrated: both ratios are given")
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
