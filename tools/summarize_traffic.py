"""Turns the PMC passes of tools/profile_traffic.sh into per-launch HBM-side byte counts for the SVGF kernels.
FETCH_SIZE / WRITE_SIZE are in KiB-units of 1024 bytes per the rocprofv3 convention; FETCH_SIZE is scaled by the
factor measured on the calibration kernel of the matching access width (8 B per lane for the RGBA16F streams)."""
import collections
import csv
import glob
import json
import sys

root = sys.argv[1]


def collect(sub, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]) * 1024.0)
    return acc


known = 8192 * 8192 * 8
cal = collect("cal_fetch", "FETCH_SIZE")
factors = {}
for name, vals in cal.items():
    if "calibration_read_kernel" in name:
        w = 8 if "2u>" in name else (16 if "4u>" in name else 4)
        factors[w] = known / (sum(vals[-1:]) / 1.0)        # last launch of each width (first one includes cold effects)
fetch = collect("bench_fetch", "FETCH_SIZE")
write = collect("bench_write", "WRITE_SIZE")
out = {"fetch_size_correction_by_bytes_per_lane": factors, "kernels": {}}
f8 = factors.get(8, 1.0)
for name in sorted(set(fetch) | set(write)):
    if "vhr::" not in name:
        continue
    fr = sum(fetch.get(name, [0])) / max(1, len(fetch.get(name, [0])))
    wr = sum(write.get(name, [0])) / max(1, len(write.get(name, [0])))
    out["kernels"][name] = {"fetch_raw_bytes": fr, "fetch_corrected_bytes": fr * f8, "write_bytes": wr, "traffic_bytes": fr * f8 + wr,
                            "launches_sampled": len(fetch.get(name, []))}
atrous = [v["traffic_bytes"] for k, v in out["kernels"].items() if "svgf_atrous" in k]
if atrous:
    out["svgf_atrous_mean_traffic_bytes_per_launch"] = sum(atrous) / len(atrous)
out["workload"] = "bench.py default: sponza_proc 1920x1080, 1 GPU"
print(json.dumps(out, indent=1))
