#!/bin/bash
# The mirror-ray launch on several builds of the library (two rounds, arms alternate): scratch/ab_refl_libs.sh lib1.so lib2.so ...   ("default" = in-tree)
# VHR_BOUNCES, VHR_SCENES, VHR_SIZE as scratch/ab_opts.py
for round in 1 2; do
  for l in "$@"; do
    if [ "$l" = "default" ]; then unset VHR_LIB_VARIANT; else export VHR_LIB_VARIANT=$l; fi
    echo "== $l (round $round)"
    VHR_KERNEL=reflection VHR_REPS=3 python scratch/ab_opts.py "" 2>&1 | grep -v amdgpu.ids
    [ $round = 1 ] && python scratch/refl_counters.py 2>&1 | grep -v amdgpu.ids
  done
done
