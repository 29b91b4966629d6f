#!/bin/bash
# The ray-tracing kernel's vector-memory path on the bench's own workload: wave-level load instructions, the address unit's (TA) busy
# cycles, L1 tag lookups, kernel cycles.  One rocprofv3 --pmc pass of its own (--kernel-trace only).  Writes profiles-ready
# gpurun_out/ta/raygen_ta.json (bench.py reads profiles/raygen_ta.json for `traversal.address_unit`).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/ta
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
C="TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU"
rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p1 -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --min-seconds 0 > $OUT/p1.log 2>&1 || { echo "pass failed"; tail -5 $OUT/p1.log; exit 1; }
python3 - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/p1/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        # the timed flavour (not the statistics one) of the default kernel: on the 32-byte nodes (r3c) or, where those do not exist, on the 48-byte ones
        if "raygen_queue_kernel<false, 2, true, false, true, false, true, false, false>" not in k and "raygen_queue_kernel<false, 2, false, false, true, false, true, false, false>" not in k: continue
        a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
c = {k: v[0] / v[1] for k, v in acc.items()}
cus, xcds = 256, 8
cycles = c["GRBM_GUI_ACTIVE"] / xcds
out = {"kernel": "raygen_queue_kernel (any-hit shadow + AO rays), bench.py default workload", "launches_sampled": acc["TA_TA_BUSY_sum"][1],
       "wave_level_load_instructions": round(c["TA_FLAT_READ_WAVEFRONTS_sum"]), "ta_busy_cycles_sum": round(c["TA_TA_BUSY_sum"]),
       "kernel_cycles": round(cycles), "ta_busy_frac": round(c["TA_TA_BUSY_sum"] / (cus * cycles), 4),
       "ta_cycles_per_load_instruction": round(c["TA_TA_BUSY_sum"] / c["TA_FLAT_READ_WAVEFRONTS_sum"], 2),
       "l1_tag_lookups": round(c["TCP_TOTAL_CACHE_ACCESSES_sum"]), "valu_instructions": round(c["SQ_INSTS_VALU"]), "salu_instructions": round(c["SQ_INSTS_SALU"]),
       "source": "tools/pmc_ta.sh: rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE SQ_INSTS_*; GRBM_GUI_ACTIVE is summed over the 8 XCDs, TA_TA_BUSY over the 256 CUs"}
json.dump(out, open("$OUT/raygen_ta.json", "w"), indent=1)
print(json.dumps(out))
PY
