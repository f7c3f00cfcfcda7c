#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --steps 40 --warmup 10 --no-cpu-baseline --no-f32-leg --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" 2>/dev/null || echo failed; }
for r in 1 2 3; do
  echo -n "c2 default: "; run --workload ar
  echo -n "c2 loaders off (870): "; run --workload ar --gemm-knob 870
  echo -n "c2 r192 off (900): "; run --workload ar --gemm-knob 900
done
