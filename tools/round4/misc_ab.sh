#!/bin/bash
mkdir -p gpurun_out/misc
python3 -m pytest tests/test_gpu_kernels.py -q -x -k "gather_in_the_contraction or rows1024" -p no:cacheprovider 2>&1 | tail -4
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline --min-timed-s 0.6"
run() { name=$1; shift; env "$@" 2> gpurun_out/misc/$name.err | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],4), 'min', round(d['block_ms_min'],4), d['config']['capture'])
except Exception as e: print('$name FAILED', e)"; }
for rep in 1 2; do
  run base_$rep $B
  run unfused_$rep $B --no-fused-backbone
  run taskstreams_$rep EGK_ENABLE=task_streams $B --no-fused-backbone
done 2>&1 | tee gpurun_out/misc/summary.txt
bash tools/round4/c4_ab.sh
