#!/bin/bash
mkdir -p gpurun_out/c4ab
B="python3 bench.py --workload egopack_oscc --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline --min-timed-s 0.6"
run() { name=$1; shift; env "$@" 2> gpurun_out/c4ab/$name.err | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],4), 'min', round(d['block_ms_min'],4))
except Exception as e: print('$name FAILED', e)"; }
for rep in 1 2; do
  run base_$rep $B
  run pktcap1_$rep DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 $B
  run pktcap0_$rep DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 $B
  run hwq8_$rep GPU_MAX_HW_QUEUES=8 $B
  run red80_$rep $B --gemm-knob 480
  run red150_$rep $B --gemm-knob 550
  run red80_group_inline_$rep EGK_ENABLE=wgrad_grouping EGK_WGRAD_SCHED=inline $B --gemm-knob 480
done 2>&1 | tee gpurun_out/c4ab/summary2.txt
# the headline with the reduce-cost knob (heads at M = 2048 are the only launches it can touch there)
for rep in 1 2; do
  run c3_base_$rep python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline --min-timed-s 0.6
  run c3_red80_$rep python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline --min-timed-s 0.6 --gemm-knob 480
done 2>&1 | tee -a gpurun_out/c4ab/summary2.txt
