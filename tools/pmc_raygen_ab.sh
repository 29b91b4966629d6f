#!/bin/bash
# The ray-tracing kernel under two option sets, same counters: address unit (TA) busy, wave-level loads, VALU / SALU instructions, and the
# split of the waves' cycles (parked at s_waitcnt / stalled at issue / issuing).  Two rocprofv3 --pmc passes per arm (--kernel-trace only).
# usage (GPU box): tools/pmc_raygen_ab.sh <tag> "<bench args of arm A>" "<bench args of arm B>"
TAG=${1:-r3}; A=${2:-}; B=${3:---option bvh_wide=1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/raygen_ab_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
C1="TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS"
C2="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU"
for arm in a b; do
  if [ $arm = a ]; then ARGS="$A"; else ARGS="$B"; fi
  rocprofv3 --kernel-trace --pmc $C1 --output-format csv -d $OUT/${arm}1 -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --min-seconds 0 $ARGS > $OUT/${arm}1.log 2>&1 || { echo "pass ${arm}1 failed"; tail -5 $OUT/${arm}1.log; exit 1; }
  rocprofv3 --kernel-trace --pmc $C2 --output-format csv -d $OUT/${arm}2 -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --min-seconds 0 $ARGS > $OUT/${arm}2.log 2>&1 || { echo "pass ${arm}2 failed"; tail -5 $OUT/${arm}2.log; exit 1; }
done
python3 - <<PY > $OUT/${TAG}_raygen_ab.txt
import csv, glob, collections
def arm(tag, args):
    acc = collections.defaultdict(lambda: [0.0, 0])
    names = set()
    for f in glob.glob("$OUT/%s[12]/**/*counter_collection.csv" % tag, recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "raygen_queue_kernel" not in k: continue
            names.add(k.split("(")[0])
            a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
    c = {k: v[0] / v[1] for k, v in acc.items()}
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    print("arm %s: bench.py %s" % (tag, args or "(defaults)"))
    for n in sorted(names): print("   kernel:", n)
    print("   kernel cycles (GRBM_GUI_ACTIVE / 8 XCDs)      %12.0f  (%.1f us at 2.4 GHz; under the profiler)" % (cyc, cyc / 2400.0))
    print("   wave-level load instructions (TA)            %12.0f" % c["TA_FLAT_READ_WAVEFRONTS_sum"])
    print("   TA busy / (256 CUs x kernel cycles)           %12.4f   (%.1f TA cycles per load instruction)" % (c["TA_TA_BUSY_sum"] / (256 * cyc), c["TA_TA_BUSY_sum"] / c["TA_FLAT_READ_WAVEFRONTS_sum"]))
    print("   L1 tag lookups                                %12.0f" % c["TCP_TOTAL_CACHE_ACCESSES_sum"])
    print("   SQ_INSTS_VALU / SALU / LDS / VMEM_RD          %12.0f %12.0f %12.0f %12.0f" % (c["SQ_INSTS_VALU"], c["SQ_INSTS_SALU"], c["SQ_INSTS_LDS"], c["SQ_INSTS_VMEM_RD"]))
    w = c["SQ_WAVE_CYCLES"]
    print("   wave cycles: parked (WAIT_ANY) %.1f %%, stalled at issue (WAIT_INST_ANY) %.1f %%, issuing (ACTIVE_INST_ANY) %.1f %%" % (100 * c["SQ_WAIT_ANY"] / w, 100 * c["SQ_WAIT_INST_ANY"] / w, 100 * c["SQ_ACTIVE_INST_ANY"] / w))
    print("   VALU busy: SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x kernel cycles) = %.3f" % (c["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc)))
    print("   active lanes over all VALU instructions: %.1f %%" % (100 * c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64)))
    return c
a = arm("a", """$A"""); b = arm("b", """$B""")
PY
cat $OUT/${TAG}_raygen_ab.txt
