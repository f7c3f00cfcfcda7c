#!/bin/bash
run() { name=$1; shift; "$@" 2>/dev/null | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],4), 'min', round(d['block_ms_min'],4))
except Exception as e: print('$name FAILED', e)"; }
C="--steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline --min-timed-s 0.6 --workload egopack_oscc"
for rep in 1 2; do
  run late_$rep python3 bench.py $C
  EGK_DISABLE=precise_late_fork run first_$rep python3 bench.py $C
  EGK_DISABLE=precise_stream run inline_$rep python3 bench.py $C
done
