#!/bin/bash
# 64-row tiles with TWO wave groups per workgroup (variant 12) against the lone 4-wave workgroup (11) and 128-row two-group (5)
# at the 2048-row shapes of configs 2 and 4; then those configs with the policy's use of (12) off / on.
python3 -m pytest tests/test_gpu_kernels.py -q -x -k "gemm_pipelined" 2>&1 | tail -3
python3 tools/gemm_bench.py --variants 11,12,5,1 2>&1 | grep -E "shape|2048"
run() { name=$1; shift; "$@" 2>/dev/null | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],4), 'min', round(d['block_ms_min'],4))
except Exception as e: print('$name FAILED', e)"; }
C="--steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline --min-timed-s 0.6"
for rep in 1 2; do
  run c2_off_$rep python3 bench.py $C --workload ar --gemm-knob 500
  run c2_on_$rep python3 bench.py $C --workload ar --gemm-knob 501
  run c4_off_$rep python3 bench.py $C --workload egopack_oscc --gemm-knob 500
  run c4_on_$rep python3 bench.py $C --workload egopack_oscc --gemm-knob 501
done
run c3_on python3 bench.py $C
run c5_on python3 bench.py $C --workload mtl4 --T 256 --batch 16
