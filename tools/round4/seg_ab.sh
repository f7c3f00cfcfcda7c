#!/bin/bash
# segmented replay (egk_graph_plan_*) against the runtime's replay of the same capture: every BASELINE configuration, alternating
python3 -m pytest tests/test_gpu_step_structures.py -q -x -k segmented 2>&1 | tail -3
run() { name=$1; shift; "$@" 2>/dev/null | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],4), 'min', round(d['block_ms_min'],4))
except Exception as e: print('$name FAILED', e)"; }
C="--steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline --min-timed-s 0.6"
for rep in 1 2; do
  for w in "c4 --workload egopack_oscc" "c3 --workload mtl" "c2 --workload ar" "c5 --workload mtl4 --T 256 --batch 16"; do
    set -- $w; tag=$1; shift
    run ${tag}_runtime_$rep python3 bench.py $C "$@"
    for n in 2 3 4 6; do
      EGK_ENABLE=segmented_replay=$n run ${tag}_seg${n}_$rep python3 bench.py $C "$@"
    done
  done
done
