#!/bin/bash
# HBM-side traffic of the SVGF kernels from PMC counters, collected as MI355X_MICROARCH.md prescribes:
# separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), --kernel-trace only, plus a calibration
# pass that reads a known byte count with the same access widths.  Usage (on the GPU box): tools/profile_traffic.sh <tag>
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/traffic_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/cal_fetch -- python3 $R/tools/calibrate_fetch.py > $OUT/cal_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/bench_fetch -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/bench_write -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench_write.log 2>&1
python3 $R/tools/summarize_traffic.py $OUT > $OUT/summary.json
cat $OUT/summary.json
