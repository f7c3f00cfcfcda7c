#!/bin/bash
mkdir -p gpurun_out/c5
python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -x -q -m gpu -p no:cacheprovider -k "segment_max or segmax or oscc or OSCC" 2>&1 | tail -3
run() { name=$1; shift; env "$@" 2> gpurun_out/c5/$name.err | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],4), 'min', round(d['block_ms_min'],4))
except Exception as e: print('$name FAILED', e)"; }
C="--steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline --min-timed-s 0.6"
for rep in 1 2; do
  run c5_$rep python3 bench.py --workload mtl4 --T 256 --batch 16 $C
  run c4_$rep python3 bench.py --workload egopack_oscc $C
done 2>&1 | tee gpurun_out/c5/summary.txt
python3 bench.py --compute f32 --steps 20 --warmup 3 --no-cpu-baseline --no-f32-leg --kernel-table > gpurun_out/c5/f32.json 2> gpurun_out/c5/f32_table.txt; tail -c 400 gpurun_out/c5/f32.json
