#!/bin/bash
# The other BASELINE.json configurations on ONE GPU (configs 4 and 5 name 8 GPUs: the single-GPU numbers are the per-frame work).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
run() { name=$1; shift; timeout -k 10 400 python3 $R/bench.py --no-cpu-baseline --no-extras --min-seconds 0.3 "$@" > $OUT/cfg_$name.json 2> $OUT/cfg_$name.err || echo "FAILED $name"; }
run cfg2_1080p_shadows_only --ao-spp 0 --steps 16
run cfg3_4k_4spp --width 3840 --height 2160 --ao-spp 4 --steps 16 --max-gbuffers 20
run cfg4_bistro_1080p_full_hybrid --scene bistro_proc --reflections --steps 16
run extra_sponza_hard_1080p --scene sponza_hard --steps 16
run extra_sponza_hard_rot_1080p --scene sponza_hard_rot --steps 16
run extra_sponza_hard_rot_1080p_world_axes --scene sponza_hard_rot --bvh-frame 0 --steps 16
run extra_bistro_rot_1080p_full_hybrid --scene bistro_proc_rot --reflections --steps 16
run extra_bistro_rot_1080p_full_hybrid_world_axes --scene bistro_proc_rot --reflections --bvh-frame 0 --steps 16
run cfg5_bistro_4k_16spp_2bounce --scene bistro_proc --width 3840 --height 2160 --ao-spp 16 --refl-bounces 2 --steps 8 --max-gbuffers 12
python3 - <<PY
import json, glob, os
out = {}
for f in sorted(glob.glob("$OUT/cfg_*.json")):
    try:
        out[os.path.basename(f)[4:-5]] = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        out[os.path.basename(f)[4:-5]] = {"error": str(e)}
json.dump(out, open("$OUT/other_configs_1gpu.json", "w"), indent=1)
for k, v in out.items():
    print(k, v.get("value"), v.get("ms_per_step"), v.get("kernels_us"))
PY
