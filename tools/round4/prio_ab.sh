#!/bin/bash
mkdir -p gpurun_out/prio
python3 -c "import torch; print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else None)"
run() { name=$1; shift; env "$@" 2> gpurun_out/prio/$name.err | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],4), 'min', round(d['block_ms_min'],4), d['config']['capture'])
except Exception as e: print('$name FAILED', e)"; }
C="--steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline --min-timed-s 0.6"
for rep in 1 2; do
  run eager_$rep python3 bench.py --mode eager $C
  run eager_mainhi_$rep EGK_MAIN_PRIO=-1 python3 bench.py --mode eager $C
  run eager_mainhi_sidelo_$rep EGK_MAIN_PRIO=-1 EGK_SIDE_PRIO=0 python3 bench.py --mode eager $C
  run eager_sidehi_$rep EGK_SIDE_PRIO=-1 python3 bench.py --mode eager $C
  run graph_mainhi_$rep EGK_MAIN_PRIO=-1 python3 bench.py $C
  run graph_$rep python3 bench.py $C
done 2>&1 | tee gpurun_out/prio/summary.txt
