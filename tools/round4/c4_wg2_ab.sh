#!/bin/bash
# config 4: the two-wave-group 64-row variant (12) INSTEAD of split-K for the 2048-row contractions (reduce launch priced up)
run() { name=$1; shift; "$@" 2>/dev/null | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],4), 'min', round(d['block_ms_min'],4))
except Exception as e: print('$name FAILED', e)"; }
C="--steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline --min-timed-s 0.6 --workload egopack_oscc"
for rep in 1 2; do
  run base_$rep python3 bench.py $C
  run wg2_$rep python3 bench.py $C --gemm-knob 501
  run wg2_red60_$rep python3 bench.py $C --gemm-knob 501,460
  run wg2_red99_$rep python3 bench.py $C --gemm-knob 501,499
  run red99_$rep python3 bench.py $C --gemm-knob 499
done
