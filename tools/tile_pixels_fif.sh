#!/bin/bash
# VERDICT r2 #1 (last sentence): wide ray-tracing tiles (raygen_tile_pixels 128: 16x8 pixels and one queue per wave) lost at 1080p because half
# as many waves left a longer launch tail -- with two frames in flight the next frame's waves fill that tail.  Whole-frame ms at 1080p and
# 4K (4 AO samples), frames in flight 1 and 2, tile 64 and 128.  usage (GPU box): tools/tile_pixels_fif.sh > profiles/r3_tile_pixels_fif.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
for size in "1920 1080 2" "3840 2160 4"; do
  set -- $size
  for fif in 1 2; do
    for tp in 64 128; do
      line=$(python3 $R/bench.py --no-cpu-baseline --no-extras --min-seconds 0.5 --width $1 --height $2 --ao-spp $3 --frames-in-flight $fif --option raygen_tile_pixels=$tp 2>/dev/null | grep '^{')
      echo "$1x$2 ao_spp $3 frames_in_flight $fif raygen_tile_pixels $tp: $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/frame,", d["value"], "Mrays/s, raygen kernel", d["traversal"]["avg_launch_ms"], "ms, lanes", d["traversal"]["active_lane_utilisation"])')"
    done
  done
done
