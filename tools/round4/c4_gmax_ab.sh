#!/bin/bash
# BASELINE config 4: GraphONE's max aggregation with every load of a row up front, all tasks in one launch (default) against the
# generic kernel (--egk-tune gather_max=0: same grouped entry point, one dependent-load chain per row)
run() { name=$1; shift; "$@" 2>gpurun_out/ab_err_$name.log | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],4), 'min', round(d['block_ms_min'],4))
except Exception as e: print('$name FAILED', e)"; }
C="--workload egopack_oscc --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline --min-timed-s 0.6"
for rep in 1 2 3; do
  run c4_upfront_$rep python3 bench.py $C
  run c4_generic_$rep python3 bench.py $C --egk-tune gather_max=0
done
