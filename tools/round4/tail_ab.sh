#!/bin/bash
run() { name=$1; shift; "$@" 2>/dev/null | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],4), 'min', round(d['block_ms_min'],4))
except Exception as e: print('$name FAILED', e)"; }
C="--steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline --min-timed-s 0.6"
for rep in 1 2 3; do
  run c3_base_$rep python3 bench.py $C
  EGK_DISABLE=tail_group run c3_notailgroup_$rep python3 bench.py $C
done
EGK_DISABLE=tail_group bash tools/timeline.sh notail > /dev/null 2>&1; awk '$1>1050' gpurun_out/tl_notail/replay.txt | cut -c1-80
