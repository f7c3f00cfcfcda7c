#!/bin/bash
mkdir -p gpurun_out/c4ab
python3 -m pytest tests/test_gpu_configs.py tests/test_gpu_precise.py tests/test_gpu_blockwise.py -x -q -m gpu -p no:cacheprovider -k "c4 or config4 or precise or GraphONE or OSCC" 2>&1 | tail -4
B="python3 bench.py --workload egopack_oscc --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline --min-timed-s 0.6"
run() { name=$1; shift; env "$@" 2> gpurun_out/c4ab/$name.err | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],4), 'min', round(d['block_ms_min'],4))
except Exception as e: print('$name FAILED', e)"; }
for rep in 1 2 3; do
  run inter_$rep $B
  run no_precise_inter_$rep EGK_DISABLE=precise_interleave $B
  run no_g1_inter_$rep EGK_DISABLE=graphone_interleave $B
  run neither_$rep EGK_DISABLE=precise_interleave,graphone_interleave $B
done 2>&1 | tee gpurun_out/c4ab/summary3.txt
bash tools/timeline.sh r4c4 --workload egopack_oscc > gpurun_out/c4ab/timeline.txt 2>&1; head -6 gpurun_out/tl_r4c4/timeline.txt
