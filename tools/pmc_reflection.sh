#!/bin/bash
# The mirror-ray kernel (reflection_queue_kernel: raygen.rgen:59-65 + reflection_hit.rchit) under the counters the any-hit kernel has:
# address unit (TA) busy cycles and wave-level loads, where the waves' cycles go, instruction counts and active lanes.  Three rocprofv3
# --pmc passes of their own (--kernel-trace only).  Usage on the GPU box: tools/pmc_reflection.sh <tag> [scene]
TAG=${1:-r4}
SCENE=${2:-sponza_proc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/refl_pmc_${TAG}_${SCENE}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--scene $SCENE --reflections --steps 8 --warmup 2 --no-cpu-baseline --no-extras --min-seconds 0"
rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -- python3 $R/bench.py $ARGS > $OUT/p1.log 2>&1 || { echo "pass 1 failed"; tail -5 $OUT/p1.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/p2 -- python3 $R/bench.py $ARGS > $OUT/p2.log 2>&1 || { echo "pass 2 failed"; tail -5 $OUT/p2.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $OUT/p3 -- python3 $R/bench.py $ARGS > $OUT/p3.log 2>&1 || { echo "pass 3 failed"; tail -5 $OUT/p3.log; exit 1; }
python3 - <<PY > $OUT/${TAG}_refl_pmc_${SCENE}.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "reflection_queue_kernel" not in k and "reflection_walk_kernel" not in k and "reflection_shade_kernel" not in k and "raygen_queue_kernel" not in k: continue
        a = acc[k][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
print("scene $SCENE 1920x1080, bench.py --reflections; per-dispatch averages (rocprofv3 --pmc, three passes)")
for k, cs in acc.items():
    c = {n: v[0] / v[1] for n, v in cs.items()}
    print(k)
    for n in sorted(c): print(f"   {n:34s} {c[n]:16.1f}   (dispatches {cs[n][1]})")
    if "GRBM_GUI_ACTIVE" in c and "TA_TA_BUSY_sum" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8
        print(f"   -> kernel cycles {cyc:.0f}; TA busy {c['TA_TA_BUSY_sum'] / (256 * cyc):.3f} of the 256 CUs' address units; {c['TA_TA_BUSY_sum'] / max(1, c['TA_FLAT_READ_WAVEFRONTS_sum']):.1f} TA cycles per wave-level load; TA floor {c['TA_TA_BUSY_sum'] / 256 / 2.4e3:.1f} us")
    if "SQ_WAVE_CYCLES" in c:
        w = c["SQ_WAVE_CYCLES"]
        print(f"   -> of the waves' cycles: waiting (s_waitcnt / barrier) {c['SQ_WAIT_ANY'] / w:.3f}, issue stalls {c['SQ_WAIT_INST_ANY'] / w:.3f}, issuing {c['SQ_ACTIVE_INST_ANY'] / w:.3f} (VALU {c['SQ_ACTIVE_INST_VALU'] / w:.3f}, VMEM {c['SQ_ACTIVE_INST_VMEM'] / w:.3f}, LDS {c['SQ_ACTIVE_INST_LDS'] / w:.3f}, scalar {c['SQ_ACTIVE_INST_SCA'] / w:.3f})")
    if "SQ_THREAD_CYCLES_VALU" in c and "SQ_INSTS_VALU" in c:
        print(f"   -> active lanes over all VALU instructions {c['SQ_THREAD_CYCLES_VALU'] / (c['SQ_INSTS_VALU'] * 64):.3f}; VALU instructions per wave {c['SQ_INSTS_VALU'] / max(1, c.get('SQ_WAVES', 1)):.0f}, loads per wave {c['SQ_INSTS_VMEM_RD'] / max(1, c.get('SQ_WAVES', 1)):.0f}")
PY
cat $OUT/${TAG}_refl_pmc_${SCENE}.txt
