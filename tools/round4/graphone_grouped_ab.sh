#!/bin/bash
# BASELINE config 4 with GraphONE's stages as ONE grouped chain (default) against one chain per task (EGK_DISABLE=graphone_grouped)
run() { name=$1; shift; "$@" 2>gpurun_out/ab_err_$name.log | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],4), 'min', round(d['block_ms_min'],4), 'nodes', d['config'].get('graph_nodes'))
except Exception as e: print('$name FAILED', e)"; }
C="--workload egopack_oscc --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline --min-timed-s 0.6"
for rep in 1 2 3; do
  run c4_grouped_$rep python3 bench.py $C
  EGK_DISABLE=graphone_grouped run c4_pertask_$rep python3 bench.py $C
done
