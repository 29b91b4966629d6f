#!/bin/bash
# TA / TCP / instruction counters of scratch/ab_loads.py runs: default library, the +2-loads variant, compact nodes.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/ta
mkdir -p $OUT/default $OUT/x2 $OUT/compact
cd /tmp && export TMPDIR=/tmp
C="TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU"
rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/default/p1 -- python3 $R/scratch/ab_loads.py > $OUT/default.log 2>&1 || { echo "default failed"; tail -5 $OUT/default.log; exit 1; }
# (the +2-loads library was a build-time experiment: this leg runs only where scratch/_variants/libvhr_x2.so was built)
export VHR_LIB_VARIANT=$R/scratch/_variants/libvhr_x2.so
if [ -f $VHR_LIB_VARIANT ]; then
rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/x2/p1 -- python3 $R/scratch/ab_loads.py > $OUT/x2.log 2>&1 || { echo "x2 failed"; tail -5 $OUT/x2.log; exit 1; }
fi
unset VHR_LIB_VARIANT
rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/compact/p1 -- python3 $R/scratch/ab_loads.py compact_nodes=1 > $OUT/compact.log 2>&1 || { echo "compact failed"; tail -5 $OUT/compact.log; exit 1; }
for v in default x2 compact; do echo "== $v"; python3 $R/tools/pmc_summary.py $OUT/$v | grep -A8 raygen_queue; done
