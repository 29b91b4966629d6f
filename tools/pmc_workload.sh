#!/bin/bash
# The PMC-derived fields of bench.py's line for ONE workload, tied to the library they were collected on:
#   tools/pmc_workload.sh [bench.py workload arguments, e.g. --scene bistro_proc --reflections]
# -> gpurun_out/pmc/pmc_<workload>.json (copy it to profiles/): the library's source fingerprint (vhr_source_fingerprint), HBM-side bytes per
# a-trous launch (FETCH_SIZE x the correction calibrated in the same run + WRITE_SIZE; separate --pmc passes, --kernel-trace only, as
# MI355X_MICROARCH.md prescribes), wave-level vector instructions per a-trous launch, the ray-tracing kernel's address-unit counters.
# bench.py quotes a file only when its fingerprint equals the loaded library's and its workload is the one being measured.
R=${GRAFT_REPO_ROOT:-$(pwd)}
KEY=$(python3 $R/bench.py --print-workload-key "$@") || exit 1
OUT=$R/gpurun_out/pmc/$KEY
rm -rf $OUT; mkdir -p $OUT
rm -f $R/gpurun_out/pmc/pmc_$KEY.json      # a failed pass must not leave an earlier run's file for `cp gpurun_out/pmc/pmc_*.json profiles/`
cd /tmp && export TMPDIR=/tmp
B="--steps 8 --warmup 2 --no-cpu-baseline --no-extras --min-seconds 0 $*"
echo "[$KEY] FETCH_SIZE calibration"; rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/cal_fetch -- python3 $R/tools/calibrate_fetch.py > $OUT/cal_fetch.log 2>&1 || { echo "calibration failed"; tail -3 $OUT/cal_fetch.log; exit 1; }
echo "[$KEY] FETCH_SIZE"; rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/bench_fetch -- python3 $R/bench.py $B > $OUT/bench_fetch.log 2>&1 || { echo "fetch pass failed"; tail -3 $OUT/bench_fetch.log; exit 1; }
echo "[$KEY] WRITE_SIZE"; rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/bench_write -- python3 $R/bench.py $B > $OUT/bench_write.log 2>&1 || { echo "write pass failed"; exit 1; }
echo "[$KEY] SQ"; rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq -- python3 $R/bench.py $B > $OUT/sq.log 2>&1 || { echo "SQ pass failed"; exit 1; }
echo "[$KEY] TA"; rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/ta -- python3 $R/bench.py $B > $OUT/ta.log 2>&1 || { echo "TA pass failed"; exit 1; }
python3 $R/tools/summarize_traffic.py $OUT > $OUT/traffic.json
python3 - <<PY
import csv, glob, json, collections, sys
sys.path.insert(0, "$R")
from vulkanhybridrenderer_amd import lib
def collect(sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/" + sub + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
sq, ta = collect("sq"), collect("ta")
traffic = json.load(open("$OUT/traffic.json"))
atrous = {k: v for k, v in sq.items() if "svgf_atrous_tile_kernel" in k}
out = {"fingerprint": lib.source_fingerprint(), "workload": "$KEY", "bench_arguments": "$*",
       "svgf_atrous_mean_traffic_bytes_per_launch": traffic.get("svgf_atrous_mean_traffic_bytes_per_launch"),
       "fetch_size_correction_by_bytes_per_lane": traffic.get("fetch_size_correction_by_bytes_per_lane"),
       "traffic_per_kernel": {k: v["traffic_bytes"] for k, v in traffic.get("kernels", {}).items()},
       "traffic_split_per_kernel": {k: {"fetch_corrected_bytes": v["fetch_corrected_bytes"], "write_bytes": v["write_bytes"]} for k, v in traffic.get("kernels", {}).items()},
       "svgf_atrous_valu_insts_per_launch": round(sum(v["SQ_INSTS_VALU"] for v in atrous.values()) / max(1, len(atrous))) if atrous else None,
       "svgf_atrous_wait_inst_any_per_launch": round(sum(v["SQ_WAIT_INST_ANY"] for v in atrous.values()) / max(1, len(atrous))) if atrous else None,
       "svgf_atrous_wave_cycles_per_launch": round(sum(v["SQ_WAVE_CYCLES"] for v in atrous.values()) / max(1, len(atrous))) if atrous else None,
       "sq_per_kernel": sq,
       "source": "tools/pmc_workload.sh: rocprofv3 --kernel-trace --pmc, one pass per counter group (FETCH_SIZE with its calibration, WRITE_SIZE, SQ_*, TA_*); per-dispatch averages"}
for k, c in ta.items():
    targs = k.split("<", 1)[1].rsplit(">", 1)[0].split(", ") if "<" in k else []
    if "raygen_queue_kernel" in k and len(targs) >= 4 and targs[3] == "false" and "TA_TA_BUSY_sum" in c:      # the timed flavour (<WAVES, COMPACT, SPILL, STATS, FUSE>: STATS = false)
        cycles = c["GRBM_GUI_ACTIVE"] / 8
        out["raygen_ta"] = {"kernel": k, "wave_level_load_instructions": round(c["TA_FLAT_READ_WAVEFRONTS_sum"]), "ta_busy_cycles_sum": round(c["TA_TA_BUSY_sum"]),
                            "kernel_cycles": round(cycles), "ta_busy_frac": round(c["TA_TA_BUSY_sum"] / (256 * cycles), 4),
                            "ta_cycles_per_load_instruction": round(c["TA_TA_BUSY_sum"] / c["TA_FLAT_READ_WAVEFRONTS_sum"], 2)}
    if "reflection_queue_kernel" in k and len(targs) >= 3 and targs[2] == "false" and "TA_TA_BUSY_sum" in c and "reflection_ta" not in out:      # (<SPILL, BOUNCES, STATS>)
        cycles = c["GRBM_GUI_ACTIVE"] / 8
        out["reflection_ta"] = {"kernel": k, "wave_level_load_instructions": round(c["TA_FLAT_READ_WAVEFRONTS_sum"]), "ta_busy_cycles_sum": round(c["TA_TA_BUSY_sum"]),
                                "kernel_cycles": round(cycles), "ta_busy_frac": round(c["TA_TA_BUSY_sum"] / (256 * cycles), 4)}
json.dump(out, open("$R/gpurun_out/pmc/pmc_$KEY.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("fingerprint", "workload", "svgf_atrous_mean_traffic_bytes_per_launch", "svgf_atrous_valu_insts_per_launch")}))
PY
