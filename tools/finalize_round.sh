#!/bin/bash
# Everything profiles/ holds about the library as it is NOW, in one gpurun call (about three GPU-minutes):   tools/finalize_round.sh <tag>
#   1. the counter file of each BASELINE configuration measured on one GPU (tools/pmc_workload.sh) -> profiles/pmc_<workload>.json
#   2. tools/profile_round.sh <tag> (bench line, the same under rocprofv3 --kernel-trace --stats, in order, one SQ pass) -> profiles/<tag>_*
#   3. tools/other_configs.sh -> profiles/<tag>_other_configs_1gpu.json
# Run it after the LAST change under csrc/ or include/: the counter files carry the library's source fingerprint and bench.py quotes them
# only on that library.  On the GPU box the results land in gpurun_out/; copy them into profiles/ afterwards with `tools/finalize_round.sh <tag> collect`.
set -o pipefail      # a failed counter pass must stop the round: `... | tail -1` alone reports tail's status
TAG=${1:-r5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
if [ "$2" = "collect" ]; then
    cp $R/gpurun_out/pmc/pmc_*.json $R/profiles/ || exit 1
    for f in ${TAG}_bench_1gpu.json ${TAG}_bench_under_rocprof.json ${TAG}_bench_in_order_under_rocprof.json ${TAG}_kernel_stats_bench_1080p.csv ${TAG}_kernel_stats_bench_1080p_in_order.csv ${TAG}_pmc_sq.txt; do
        cp $R/gpurun_out/profile_$TAG/$f $R/profiles/ || exit 1
    done
    cp $R/gpurun_out/oc_$TAG/other_configs_1gpu.json $R/profiles/${TAG}_other_configs_1gpu.json || exit 1
    python3 - <<PY
import json
d = json.loads(open("$R/profiles/${TAG}_bench_1gpu.json").read().strip().splitlines()[-1])
print("collected:", d["ms_per_step"], "ms,", d["value"], d["unit"], "roofline.frac", d["roofline"]["frac"], "library", d["roofline"]["library_fingerprint"], "counters", d["roofline"]["pmc_file"])
PY
    exit 0
fi
bash $R/tools/pmc_workload.sh | tail -1 &&
bash $R/tools/pmc_workload.sh --width 3840 --height 2160 --ao-spp 4 --max-gbuffers 20 | tail -1 &&
bash $R/tools/pmc_workload.sh --scene bistro_proc --reflections | tail -1 &&
bash $R/tools/pmc_workload.sh --scene bistro_proc --width 3840 --height 2160 --ao-spp 16 --refl-bounces 2 --max-gbuffers 12 | tail -1 || exit 1
# (the bench line quotes the counter files: they have to be in profiles/ of THIS copy before it runs)
cp $R/gpurun_out/pmc/pmc_*.json $R/profiles/
bash $R/tools/profile_round.sh $TAG > $R/gpurun_out/profile_$TAG.log 2>&1 || { tail -5 $R/gpurun_out/profile_$TAG.log; exit 1; }
bash $R/tools/other_configs.sh oc_$TAG | tail -4
