#!/bin/bash
# One bench configuration for several builds of the library, alternated: scratch/ab_cfg_libs.sh "<bench args>" <variant.so or default> ...
ARGS=$1; shift
for rep in 1 2 3; do
for L in "$@"; do
  V=$L; [ "$L" = "default" ] && V=""
  VHR_LIB_VARIANT=$V python scratch/bench_variant.py --no-cpu-baseline --no-extras --min-seconds 0.5 $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$L', d['ms_per_step'], d.get('passes_ms'))"
done
done
