#!/bin/bash
# BASELINE config 4: Adam over everything but the temporal pooling beside the step's last weight-gradient launch (default) against
# Adam alone at the end of the step (EGK_DISABLE=early_adam)
run() { name=$1; shift; "$@" 2>gpurun_out/ab_err_$name.log | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],4), 'min', round(d['block_ms_min'],4))
except Exception as e: print('$name FAILED', e)"; }
C="--workload egopack_oscc --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline --min-timed-s 0.6"
for rep in 1 2 3; do
  run c4_early_$rep python3 bench.py $C
  EGK_DISABLE=early_adam run c4_late_$rep python3 bench.py $C
done
