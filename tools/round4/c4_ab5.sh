#!/bin/bash
mkdir -p gpurun_out/c4ab
python3 -m pytest tests/test_gpu_precise.py tests/test_gpu_configs.py -x -q -m gpu -p no:cacheprovider -k "tee or c4 or config4 or precise" 2>&1 | tail -3
B="python3 bench.py --workload egopack_oscc --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline --min-timed-s 0.6"
run() { name=$1; shift; env "$@" 2> gpurun_out/c4ab/$name.err | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],4), 'min', round(d['block_ms_min'],4))
except Exception as e: print('$name FAILED', e)"; }
for rep in 1 2 3; do
  run tee_$rep $B
  run notee_$rep EGK_DISABLE=x3_tee $B
done 2>&1 | tee gpurun_out/c4ab/summary5.txt
for rep in 1 2; do
  run x3_tee_$rep python3 bench.py --compute bf16x3 --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline --min-timed-s 0.6
  run x3_notee_$rep EGK_DISABLE=x3_tee python3 bench.py --compute bf16x3 --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline --min-timed-s 0.6
done 2>&1 | tee -a gpurun_out/c4ab/summary5.txt
