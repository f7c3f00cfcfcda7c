#!/bin/bash
# split-K finished inside the launch (egk_gemm_desc.sk_tickets) against the separate reduce launch: EGK_DISABLE=splitk_in_launch
python3 -m pytest tests/test_gpu_kernels.py -q -x -k "splitk_finished or gemm_pipelined or gemm_f32" 2>&1 | tail -3
run() { name=$1; shift; "$@" 2>/dev/null | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],4), 'min', round(d['block_ms_min'],4))
except Exception as e: print('$name FAILED', e)"; }
C="--steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline --min-timed-s 0.6"
for rep in 1 2; do
  for w in "c4 --workload egopack_oscc" "c2 --workload ar" "c3 --workload mtl" "c5 --workload mtl4 --T 256 --batch 16"; do
    set -- $w; tag=$1; shift
    run ${tag}_inlaunch_$rep python3 bench.py $C "$@"
    EGK_DISABLE=splitk_in_launch run ${tag}_reduce_$rep python3 bench.py $C "$@"
  done
done
