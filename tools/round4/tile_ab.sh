#!/bin/bash
# tile-height sweep of the headline step: 96-row policy (512 tiles = one per resident slot) against 64-row tiles (1024 tiles on
# 768 slots: a second wave of tiles whose K loops overlap the first wave's epilogues -- what a persistent kernel would arrange)
run() { name=$1; shift; "$@" 2>/dev/null | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],4), 'min', round(d['block_ms_min'],4))
except Exception as e: print('$name FAILED', e)"; }
C="--steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline --min-timed-s 0.6"
for rep in 1 2; do
  run policy_$rep python3 bench.py $C
  run rows64_$rep python3 bench.py $C --gemm-knob 11
  run rows128_$rep python3 bench.py $C --gemm-knob 3
  run rows96_$rep python3 bench.py $C --gemm-knob 8
done
