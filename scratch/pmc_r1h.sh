#!/bin/bash
# r1h PMC passes (each in its own run, --kernel-trace only): hardware lane utilisation of the traversal kernels and the
# SVGF traffic (tools/profile_traffic.sh).  Usage on the GPU box: scratch/pmc_${TAG:-r1j}.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_${TAG:-r1j}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3-avail list 2>/dev/null | grep -i -o "SQ_THREAD_CYCLES_VALU\|SQ_ACTIVE_INST_VALU\b" | sort | uniq -c > $OUT/avail.txt
i=0
for C in "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --reflections > $OUT/p$i.log 2>&1
done
python3 $R/scratch/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/avail.txt; head -60 $OUT/summary.txt
