#!/bin/bash
# BASELINE config 4: Adam over GraphONE's slice beside the backbone's backward (EGK_ENABLE=graphone_adam) against in the step's tail (default)
run() { name=$1; shift; "$@" 2>gpurun_out/ab_err_$name.log | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],4), 'min', round(d['block_ms_min'],4))
except Exception as e: print('$name FAILED', e)"; }
C="--workload egopack_oscc --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline --min-timed-s 0.6"
for rep in 1 2 3; do
  EGK_ENABLE=graphone_adam run c4_g1adam_$rep python3 bench.py $C
  run c4_tailadam_$rep python3 bench.py $C
done
python3 bench.py $C --stamps 2>&1 >/dev/null | grep "\[stamp\]" | grep "step_start\|graphone_bwd\|wgrad_group\|backward_done\|adam_done\|bwd_rowln\[[012]\]" | tail -16
