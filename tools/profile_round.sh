#!/bin/bash
# Everything profiles/ holds for one round, taken on the GPU box in ONE gpurun call:   tools/profile_round.sh <tag>
#   1. python bench.py (default flags)                           -> <tag>_bench_1gpu.json          (the line the driver measures)
#   2. the same command under rocprofv3 --kernel-trace --stats   -> <tag>_kernel_stats_bench_1080p.csv, <tag>_bench_under_rocprof.json
#   3. one SQ counter pass (--kernel-trace only) with the mirror ray on                     -> <tag>_pmc_sq.txt
#      (traffic / instruction / address-unit counters per workload: tools/pmc_workload.sh -> profiles/pmc_<workload>.json)
# The program itself follows `--` (python3 bench.py ...): no env / bash -c hop under the profiler.
TAG=${1:-r2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-extras --min-seconds 0"
echo "[1] bench"; python3 $R/bench.py > $OUT/${TAG}_bench_1gpu.json 2> $OUT/bench.err || { echo "bench failed"; tail -5 $OUT/bench.err; exit 1; }
echo "[2] kernel trace"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/trace.err || { echo "trace failed"; tail -5 $OUT/trace.err; exit 1; }
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats_bench_1080p.csv
echo "[3] SQ"; rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/p1 -- python3 $R/bench.py --steps 8 --warmup 2 $B --reflections > $OUT/p1.log 2>&1 || { echo "SQ pass failed"; exit 1; }
python3 $R/tools/pmc_summary.py $OUT > $OUT/${TAG}_pmc_sq.txt
# (the PMC files bench.py quotes -- traffic, a-trous instructions, address unit -- are tools/pmc_workload.sh's, one per workload, fingerprinted)
tail -n 3 $OUT/${TAG}_bench_1gpu.json | cut -c1-600
grep -E "svgf_atrous|raygen|temporal|copy_rows" $OUT/${TAG}_kernel_stats_bench_1080p.csv | cut -c1-200
# [4] the same command with every dispatch in recorded order on the one stream (--option svgf_async_unread=0): the kernel summary then has
#     all five a-trous rows undisturbed (by default the step-16 dispatch runs on the side stream beside raygen_queue_kernel and its row
#     shows the time it shares the chip)
cd /tmp
echo "[4] kernel trace, in order"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_in_order -- python3 $R/bench.py --no-cpu-baseline --no-extras --option svgf_async_unread=0 > $OUT/${TAG}_bench_in_order_under_rocprof.json 2> $OUT/trace_in_order.err || { echo "in-order trace failed"; tail -5 $OUT/trace_in_order.err; exit 1; }
cp $(find $OUT/trace_in_order -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats_bench_1080p_in_order.csv
grep -E "svgf_atrous|raygen|temporal" $OUT/${TAG}_kernel_stats_bench_1080p_in_order.csv | cut -c1-200
