#!/bin/bash
# Where the ray-tracing kernels' loads are served from: L1 (TCP) accesses and the requests it passes on to L2, L2 (TCC) hits and misses.
# Two rocprofv3 --pmc passes (--kernel-trace only).  Usage on the GPU box: tools/pmc_cache.sh <tag> [bench.py arguments]
TAG=${1:-r5}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/cache_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 8 --warmup 2 --no-cpu-baseline --no-extras --min-seconds 0 $*"
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum --output-format csv -d $OUT/p1 -- python3 $R/bench.py $ARGS > $OUT/p1.log 2>&1 || { echo "pass 1 failed"; tail -5 $OUT/p1.log; exit 1; }
rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum --output-format csv -d $OUT/p2 -- python3 $R/bench.py $ARGS > $OUT/p2.log 2>&1 || { echo "pass 2 failed"; tail -5 $OUT/p2.log; exit 1; }
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum --output-format csv -d $OUT/p3 -- python3 $R/bench.py $ARGS > $OUT/p3.log 2>&1 || { echo "pass 3 failed"; tail -5 $OUT/p3.log; }
python3 $R/tools/pmc_summary.py $OUT > $OUT/${TAG}_pmc_cache.txt
# the timed flavours: raygen_queue_kernel<WAVES, COMPACT, SPILL, STATS = false, FUSE>, reflection_queue_kernel<SPILL, BOUNCES, STATS = false>
grep -E -A14 "raygen_queue_kernel<[0-9]+, (true|false), (true|false), false|reflection_queue_kernel<(true|false), [0-9]+, false" $OUT/${TAG}_pmc_cache.txt || { echo "no ray-tracing kernel in the summary"; exit 1; }
