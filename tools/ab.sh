#!/bin/bash
# A/B of a development knob on ONE box (boxes differ by several % in clocks): alternates bench.py runs with and without
# `--gemm-knob $1`.  Usage: bash tools/ab.sh <knob> [rounds] [extra bench args]
KNOB=$1; R=${2:-3}; shift; shift
for i in $(seq $R); do
  for k in "" "--gemm-knob $KNOB"; do
    python bench.py --no-cpu-baseline --no-roofline $k "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('knob[$k]', round(d['ms_per_step'],4))"
  done
done
