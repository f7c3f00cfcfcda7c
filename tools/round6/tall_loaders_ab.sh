#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --steps 40 --warmup 10 --no-cpu-baseline --no-f32-leg --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" 2>/dev/null || echo failed; }
for r in 1 2; do
  echo -n "base (tall tail only):       "; run
  echo -n "tail tall + loaders (883):   "; run --gemm-knob 883
  echo -n "tall for all groups (852):   "; run --gemm-knob 852
  echo -n "tall all + loaders (852,883): "; run --gemm-knob 852,883
  echo -n "128 loaders 3-stage (882):   "; run --gemm-knob 882
done
