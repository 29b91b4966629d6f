#!/bin/bash
# The any-hit launch and the headline frame for several builds of the library, alternated: scratch/ab_k1_libs.sh <variant.so or ""> ...
for rep in 1 2; do
for L in "$@"; do
  [ "$L" = "default" ] && L=""
  echo "== lib [$L] rep $rep"
  VHR_REPS=3 VHR_LIB_VARIANT=$L python scratch/ab_opts.py "" 2>&1 | grep -v amdgpu.ids
  VHR_LIB_VARIANT=$L python scratch/bench_variant.py --steps 400 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('frame', d['ms_per_step'])"
done
done
