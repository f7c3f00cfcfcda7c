#!/bin/bash
run() { name=$1; shift; "$@" 2>/dev/null | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],4), 'min', round(d['block_ms_min'],4))
except Exception as e: print('$name FAILED', e)"; }
C="--steps 10 --warmup 3 --no-cpu-baseline --no-f32-leg --no-roofline --min-timed-s 0.6 --compute f32"
for rep in 1 2; do
  run f32_base_$rep python3 bench.py $C
  EGK_WGRAD_SCHED=inline run f32_inline_$rep python3 bench.py $C
  EGK_WGRAD_SCHED=rows run f32_rows_$rep python3 bench.py $C
  EGK_DISABLE=wgrad_grouping run f32_nogroup_$rep python3 bench.py $C
  EGK_F32_WGRAD_COUNT=4 run f32_count4_$rep python3 bench.py $C
  EGK_F32_WGRAD_COUNT=12 run f32_count12_$rep python3 bench.py $C
done
