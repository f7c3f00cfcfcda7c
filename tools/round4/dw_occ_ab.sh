#!/bin/bash
# queued weight-gradient groups held to ONE workgroup per CU (extra dynamic LDS) so that the backward chain's launches always
# find registers and LDS on every CU
run() { name=$1; shift; "$@" 2>/dev/null | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],4), 'min', round(d['block_ms_min'],4))
except Exception as e: print('$name FAILED', e)"; }
C="--steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline --min-timed-s 0.6"
for rep in 1 2; do
  run c3_base_$rep python3 bench.py $C
  run c3_pad20_$rep python3 bench.py $C --gemm-knob 620
  run c3_pad48_$rep python3 bench.py $C --gemm-knob 648
  run c5_base_$rep python3 bench.py $C --workload mtl4 --T 256 --batch 16
  run c5_pad20_$rep python3 bench.py $C --workload mtl4 --T 256 --batch 16 --gemm-knob 620
done
