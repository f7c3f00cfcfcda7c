#!/bin/bash
run() { name=$1; shift; "$@" 2>/dev/null | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],4), 'min', round(d['block_ms_min'],4))
except Exception as e: print('$name FAILED', e)"; }
C="--steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-roofline --min-timed-s 0.6"
for rep in 1 2; do
  for n in 1 2 3 4 6; do
    EGK_WGRAD_KCHUNKS=$n run c3_k${n}_$rep python3 bench.py $C
  done
done
EGK_WGRAD_KCHUNKS=3 bash tools/timeline.sh kchunk3 > /dev/null 2>&1; head -6 gpurun_out/tl_kchunk3/timeline.txt; awk '$1>560' gpurun_out/tl_kchunk3/replay.txt | cut -c1-80
