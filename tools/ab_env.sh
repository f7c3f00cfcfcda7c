#!/bin/bash
# A/B of an EGK_DISABLE development toggle on ONE box: alternates bench.py runs with and without it.
# Usage: bash tools/ab_env.sh <toggle> [rounds] [extra bench args]
TOG=$1; R=${2:-4}; shift; shift
for i in $(seq $R); do
  for k in "" "$TOG"; do
    EGK_DISABLE=$k python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-roofline --no-f32-leg "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('disable[$k]', round(d['ms_per_step'],4))"
  done
done
