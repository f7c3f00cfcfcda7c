#!/bin/bash
# Config 4 (EgoPack OSCC + GraphONE) A/B: grouped weight gradients / late forks / schedule, same box
mkdir -p gpurun_out/c4ab
B="python3 bench.py --workload egopack_oscc --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline --min-timed-s 0.6"
run() { name=$1; shift; env "$@" $B 2> gpurun_out/c4ab/$name.err | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],4), 'min', round(d['block_ms_min'],4))
except Exception as e: print('$name FAILED', e)"; }
for rep in 1 2; do
  run base_$rep X=1
  run group_$rep EGK_ENABLE=wgrad_grouping
  run group_defer_$rep EGK_ENABLE=wgrad_grouping,deferred_forks
  run group_rows_$rep EGK_ENABLE=wgrad_grouping EGK_WGRAD_SCHED=rows
  run group_inline_$rep EGK_ENABLE=wgrad_grouping EGK_WGRAD_SCHED=inline
  run nofuse_$rep EGK_DISABLE=gather_fusion
done 2>&1 | tee gpurun_out/c4ab/summary.txt
