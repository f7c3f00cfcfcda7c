#!/bin/bash
# A/B of the backward schedule of the parked weight gradients (ops._wq_sched) and of the gather-in-epilogue fusion, same box
mkdir -p gpurun_out/sched
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline --min-timed-s 0.6"
run() { name=$1; shift; env "$@" $B 2> gpurun_out/sched/$name.err | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],4), 'min', round(d['block_ms_min'],4))
except Exception as e: print('$name FAILED', e)"; }
for rep in 1 2 3; do
  run free_$rep EGK_WGRAD_SCHED=free
  run rows_$rep EGK_WGRAD_SCHED=rows
  run inline_$rep EGK_WGRAD_SCHED=inline
  run rows_nofuse_$rep EGK_WGRAD_SCHED=rows EGK_DISABLE=gather_fusion
  run free_nofuse_$rep EGK_WGRAD_SCHED=free EGK_DISABLE=gather_fusion
  run rows_count8_$rep EGK_WGRAD_SCHED=rows EGK_WGRAD_COUNT=8
done 2>&1 | tee gpurun_out/sched/summary.txt
