#!/bin/bash
# Where the waves' cycles go (MI355X_MICROARCH.md: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES, quad-cycles), per kernel.
# One rocprofv3 --pmc pass of its own, --kernel-trace only.  Usage on the GPU box: tools/pmc_stalls.sh <tag>
TAG=${1:-r2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/stalls_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/p1 -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --min-seconds 0 > $OUT/p1.log 2>&1 || { echo "pass 1 failed"; tail -5 $OUT/p1.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p2 -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --min-seconds 0 > $OUT/p2.log 2>&1 || { echo "pass 2 failed"; tail -5 $OUT/p2.log; }
python3 $R/tools/pmc_summary.py $OUT > $OUT/${TAG}_pmc_stalls.txt
grep -A18 "raygen_queue_kernel<2, true, true, false>\|atrous_tile_kernel<2" $OUT/${TAG}_pmc_stalls.txt | head -80
