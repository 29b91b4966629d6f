#!/bin/bash
# The memory side of the kernels (L1 = TCP, its address / data units TA / TD, L2 = TCC): accesses, hit rates, busy and stall cycles,
# summed request latencies.  Four rocprofv3 --pmc passes of their own, --kernel-trace only.  Usage on the GPU box: tools/pmc_memory.sh <tag>
TAG=${1:-r2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/memory_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --min-seconds 0"
rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -- $B > $OUT/p1.log 2>&1 || { echo "pass 1 failed"; tail -5 $OUT/p1.log; exit 1; }
rocprofv3 --kernel-trace --pmc TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum --output-format csv -d $OUT/p2 -- $B > $OUT/p2.log 2>&1 || { echo "pass 2 failed"; tail -5 $OUT/p2.log; }
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum --output-format csv -d $OUT/p3 -- $B > $OUT/p3.log 2>&1 || { echo "pass 3 failed"; tail -5 $OUT/p3.log; }
rocprofv3 --kernel-trace --pmc TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TD_TD_BUSY_sum TD_TC_STALL_sum --output-format csv -d $OUT/p4 -- $B > $OUT/p4.log 2>&1 || { echo "pass 4 failed"; tail -5 $OUT/p4.log; }
python3 $R/tools/pmc_summary.py $OUT > $OUT/${TAG}_pmc_memory.txt
grep -A26 "raygen_queue_kernel<false, 2, false, false, true, false" $OUT/${TAG}_pmc_memory.txt | head -60
